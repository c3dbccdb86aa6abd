#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06v_seam_hostprof.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
A="sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1"
{
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -o /tmp/bench_blocks_prof tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
echo "---- prof build + SDRPP_GPU_HOSTPROF, pinned ----"
SDRPP_GPU_HOSTPROF=1 taskset -c $cpus /tmp/bench_blocks_prof $A 1 2>&1 | grep -v "amdgpu.ids\|passes 0" | cut -c1-260
echo "---- the same pushes from python (no stream hand-overs): SDRPP_GPU_HOSTPROF on the by-push loop ----"
SDRPP_GPU_HOSTPROF=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, time
sys.path.insert(0, ".")
import numpy as np
from sdrplusplus_amd import capi, workloads
B = 50000
ctx = capi.Context(0, max_push=B)
info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=32)
ctx.set_pipelined(True, 3)
x = workloads.synth(3, B, seed=1, nvfo=32)
t0 = time.perf_counter()
n = 4000
for i in range(n):
    ctx.push(x)
    if i >= 8:
        t = ctx.ticket() - 8
        ctx.result_wait(t, copy=False); ctx.result_release(t)
ctx.sync()
print("python push loop: %.1f us per block" % ((time.perf_counter() - t0) / n * 1e6))
ctx.close()
PY
} > $out 2>&1
cat $out
