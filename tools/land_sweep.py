#!/usr/bin/env python3
"""sr/200 blocks fetched from page-locked host memory, pipelined: rate against the landing copy's workgroup count (SDRPP_GPU_TICK_LAND_BLOCKS), its
position in the tick's grid (SDRPP_GPU_TICK_L0_AT) and the front-end shape (SDRPP_GPU_FCM16_MAX_TILES), read from the environment by the library.
Prints pinned_no_read and pinned_results_delivered (flags 3, lag 8) in Msamples/s."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from sdrplusplus_amd import capi, workloads

B, nv = 50000, 32
ctx = capi.Context(0, max_push=B)
info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=nv)
ptrs = []
for i in range(4):
    x = workloads.synth(3, B, seed=7 + i, nvfo=nv)
    p = ctx.L.sdrpp_host_alloc(B * 8)
    C.memmove(p, x.ctypes.data, B * 8)
    ptrs.append(p)
def rate(fn, end, n=600):
    for i in range(40): fn(i)
    end()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(n): fn(i)
        end()
        best = max(best, B * n / (time.perf_counter() - t0) / 1e6)
    return best
ctx.set_pipelined(True, 0)
a = rate(lambda i: ctx.push_host_ptr_async(ptrs[i % 4], B), ctx.sync)
ctx.set_pipelined(False)
ctx.set_pipelined(True, 3)
st = {"next": ctx.ticket() + 1}
def collect(upto):
    while st["next"] <= upto:
        r = capi.Result()
        ctx._chk(ctx.L.sdrpp_result_wait(ctx.h, C.c_uint64(st["next"]), C.byref(r)))
        ctx._chk(ctx.L.sdrpp_result_release(ctx.h, C.c_uint64(st["next"])))
        st["next"] += 1
def step(i):
    ctx.push_host_ptr_async(ptrs[i % 4], B)
    collect(ctx.ticket() - 8)
b = rate(step, lambda: collect(ctx.ticket()))
print("%s no_read %.0f delivered %.0f   roles %s" % (" ".join("%s=%s" % (k[10:], v) for k, v in sorted(os.environ.items()) if k.startswith("SDRPP_GPU_")), a, b, [r for r in ctx.pipeline_stats()["roles"] if r.startswith("fcm")]))
ctx.close()
