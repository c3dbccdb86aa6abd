#!/usr/bin/env python3
"""Compile sdrplusplus_amd/csrc/sdrpp_gpu.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per
kernel (VGPRs, SGPRs, scratch, occupancy, static LDS)."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "sdrplusplus_amd", "csrc", "sdrpp_gpu.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-I" + os.path.join(os.path.dirname(src), "gfx950"), "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/tmp/sdrpp_gpu_res.o", src]
out = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
cur, rows = None, {}
for l in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill): (\S+)", l)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for f, r in rows.items():
    name = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("sdrpp_k::", "").replace("void ", "")
    print("%-34s VGPR %4s SGPR %4s scratch %3s spill %2s occ %2s LDS %6s" % (name[:34], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
          r.get("VGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
