#!/usr/bin/env python3
"""Per-kernel-family HIP-event times of the VFO bank ALONE (no FFT branch on the second stream), cfg 3 geometry.
   tools/vfo_only_time.py [push] [nvfo] [steps]     (SDRPP_GPU_LIB selects an experimental build of the library)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sdrplusplus_amd import capi, radio, workloads

if os.environ.get("SDRPP_GPU_LIB"):  # experimental builds of the library (a switch of this TOOL, not of the binding)
    capi.DEFAULT_LIB = os.environ["SDRPP_GPU_LIB"]
push = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
nvfo = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
x = (torch.randn(push * 2, device="cuda:0", dtype=torch.float32) * 0.1)
ctx = capi.Context(0, max_push=push)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
if os.environ.get("SDRPP_TOOL_PIPELINE"):  # 0: one launch per back-end stage
    ctx.set_backend_pipeline(int(os.environ["SDRPP_TOOL_PIPELINE"]))
for mode, if_rate, bw, centre, _ in workloads.vfo_plan(3, nvfo):
    d, keep = radio.vfo_desc(10e6, if_rate, bw, centre, mode)
    ctx.vfo_add(d, keep)
for _ in range(3):
    ctx.push_device(x.data_ptr(), push)
torch.cuda.synchronize()
ctx.timing_enable(True)
for _ in range(steps):
    ctx.push_device(x.data_ptr(), push)
torch.cuda.synchronize()
fam = ctx.timing_read()
print(os.path.basename(capi.lib_path()), {k: round(v[0] / steps, 4) for k, v in fam.items() if v[0] > 0})
ctx.close()
