#!/usr/bin/env python3
"""Ingest rate of cfg 3 (65536-pt FFT + 32 WFM VFOs) as a function of the push size and of where the samples come from.
   tools/push_rate.py [push ...]      default: 50000 (sr/200, file_source/src/main.cpp:157), 1000000 (stream.h:9), 2^24
For every size: device-resident pushes (sdrpp_push_device), pinned host memory and pageable host memory (sdrpp_push, H2D included).
Prints one JSON line per size: enqueue time per push (host side), wall time per push, Msamples/s."""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from sdrplusplus_amd import capi, workloads

sizes = [int(a) for a in sys.argv[1:]] or [50000, 1000000, 1 << 24]
for push in sizes:
    n_push = max(8, min(400, (1 << 26) // push))
    ctx = capi.Context(0, max_push=push)
    workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=32)
    xd = torch.randn(push * 2, device="cuda:0", dtype=torch.float32) * 0.1
    xp = (torch.randn(push * 2, dtype=torch.float32) * 0.1).pin_memory()
    xh = np.ascontiguousarray(xp.numpy().copy())
    res = {"push": push, "pushes": n_push}
    for name in ("device", "pinned", "pageable"):
        def one():
            if name == "device":
                ctx.push_device(xd.data_ptr(), push)
            elif name == "pinned":
                ctx.push_host_ptr(xp.data_ptr(), push)
            else:
                ctx.push_host_ptr(xh.ctypes.data, push)
        for _ in range(5):
            one()
        ctx.sync()
        best = None
        for _trial in range(3):
            t0 = time.perf_counter()
            for _ in range(n_push):
                one()
            t1 = time.perf_counter()
            ctx.sync()
            t2 = time.perf_counter()
            r = {"enqueue_us": 1e6 * (t1 - t0) / n_push, "wall_us": 1e6 * (t2 - t0) / n_push, "msps": push * n_push / (t2 - t0) / 1e6}
            if best is None or r["msps"] > best["msps"]:
                best = r
        res[name] = {k: round(v, 2) for k, v in best.items()}
    print(json.dumps(res), flush=True)
    ctx.close()
