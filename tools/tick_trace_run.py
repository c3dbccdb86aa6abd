#!/usr/bin/env python3
"""Run N pipelined blocks of a configuration on the `make ticktrace` build and dump the per-workgroup timeline of the tick kernel.
   tools/tick_trace_run.py <cfg> <block> <nblocks> <dump.bin> [host|af]   (af: the radio's AF chain behind every VFO)
   TICK_GROUP=<k>: up to k blocks per launch (sdrpp_set_pipeline_group, fixed; the blocks are consecutive slices of one device tensor)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdrplusplus_amd import capi, workloads  # noqa: E402

capi.DEFAULT_LIB = os.path.join(ROOT, "sdrplusplus_amd", "csrc", "libsdrpp_gpu_ticktrace.so")  # a switch of this TOOL, not of the binding
import torch  # noqa: E402

cfg, B, n, path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
from_host = len(sys.argv) > 5 and sys.argv[5] == "host"  # blocks fetched from page-locked host memory by the tick's landing copy
os.environ["SDRPP_TICK_TRACE_FILE"] = path
if os.path.exists(path):
    os.remove(path)
nvfo = workloads.CFG[cfg]["nvfo"]
G = int(os.environ.get("TICK_GROUP", "1"))
ctx = capi.Context(0, max_push=B * G)
info = workloads.setup(ctx, cfg, dense_fft=True, data_width=1024, nvfo=nvfo or None)
if len(sys.argv) > 5 and sys.argv[5] == "af":
    from sdrplusplus_amd import radio
    _keep = []
    for vid, (m_, r_, _b, _c, _x) in zip(info["vids"], info["plan"]):
        a_, k_ = radio.af_desc(r_, 48000.0, 50e-6 if m_ == "WFM" else None, m_ == "NFM")
        ctx.vfo_set_af(vid, a_, k_)
        _keep.append(k_)
nslice = max(4, G)
xall = torch.from_numpy(np.concatenate([workloads.synth(cfg, B, seed=7 + i, nvfo=nvfo or None) for i in range(nslice)]).view(np.float32)).to("cuda")
xd = [xall[2 * i * B:2 * (i + 1) * B] for i in range(nslice)]
ctx.set_pipelined(True, 0)
if G > 1:
    ctx.set_pipeline_group(G, False)
if from_host:
    import ctypes as C
    ptrs = []
    for i in range(4):
        x = workloads.synth(cfg, B, seed=7 + i, nvfo=nvfo or None)
        p = ctx.L.sdrpp_host_alloc(B * 8)
        C.memmove(p, x.ctypes.data, B * 8)
        ptrs.append(p)
for i in range(n):
    if from_host:
        ctx.push_host_ptr_async(ptrs[i % 4], B)
    else:
        ctx.push_device(xd[i % nslice].data_ptr(), B)
ctx.sync()
ctx.close()
print("dumped", os.path.getsize(path) // 72, "records")
