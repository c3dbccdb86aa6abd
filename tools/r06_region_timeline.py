#!/usr/bin/env python3
"""Where the wall time of ONE 20-step timed region goes (headline workload, launch groups): host time stamps around every push, the flush, every
collected ticket and the final synchronisation, next to the device's tick completions (polled from the page-locked flag).
   python tools/r06_region_timeline.py [group] [adaptive 0|1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdrplusplus_amd import capi, multi, workloads
import bench

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
adaptive = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
push, nvfo, N, width = 1000000, 32, 65536, 1024
bufs, keep = bench.make_inputs(torch, np, dev, 3, push, 1, nvfo)
ctx = capi.Context(0, max_push=push * G)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
info = workloads.setup(ctx, 3, dense_fft=True, data_width=width, nvfo=nvfo)
ctx.set_reference_block(50000)
ctx.set_pipelined(True, 2)
ctx.set_pipeline_group(G, adaptive)
max_lines = (push + N - 1) // N + 1
lines = torch.zeros((4, max_lines + 1, width), dtype=torch.float32, device=dev)
r = multi.StreamRunner(ctx, bufs, push, lines, sync=torch.cuda.synchronize, pipelined=True, lag=8, gather_every=4)
for i in range(25):
    r.step(i)
r.finish(); torch.cuda.synchronize()
r.lag = max(r.lag, min(capi.RESULT_SLOTS - 2, int(ctx.pipeline_stats()["depth"]) + 1) * G)  # as bench.py sets it
print("lag", r.lag, "depth", ctx.pipeline_stats()["depth"])
for rep in range(3):
    ev = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        r.step(100 + rep * 20 + i)
        ev.append(("push %d" % i, time.perf_counter() - t0, ctx.pipeline_stats()["ticks"]))
    t1 = time.perf_counter() - t0
    ctx.pipeline_flush()
    t2 = time.perf_counter() - t0
    n = len(r.tickets)
    first = None
    while r.tickets:
        r._collect(r.tickets.pop(0))
        if first is None:
            first = time.perf_counter() - t0
    t3 = time.perf_counter() - t0
    r._next_ticket = None
    r._flush_batch()
    if r.side is not None:
        r.side.synchronize()
    t4 = time.perf_counter() - t0
    torch.cuda.synchronize()
    t5 = time.perf_counter() - t0
    print("rep %d: pushes done %.0f us | flush (launch drains) %.0f | first of %d tickets collected %.0f | all collected %.0f | batches on the device %.0f | device idle %.0f us -> %.0f MS/s" % (
        rep, t1 * 1e6, t2 * 1e6, n, first * 1e6, t3 * 1e6, t4 * 1e6, t5 * 1e6, 20 * push / t5 / 1e6))
    print("   pushes (us, ticks launched):", " ".join("%.0f/%d" % (t * 1e6, k) for _, t, k in ev))
ctx.close()
