#!/usr/bin/env python3
"""cfg 3 at the reference's block size (sr/200 = 50 000 samples), device-resident blocks, pipelined, nothing read back: ingest rate against the
number of blocks one launch may carry (sdrpp_set_pipeline_group, adaptive).   python tools/r06_group_sweep.py [B] [k ...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdrplusplus_amd import capi, workloads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ks = [int(a) for a in sys.argv[2:]] or [1, 4, 8, 12, 16, 20, 24, 32]
nvfo = 32
dev = torch.device("cuda", 0)
nb = 96
xs = np.concatenate([workloads.synth(3, B, seed=7 + i, nvfo=nvfo) for i in range(nb)])
xd = torch.from_numpy(xs.view(np.float32)).to(dev)
base = xd.data_ptr()
for k in ks:
    ctx = capi.Context(0, max_push=B * k)
    info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=nvfo)
    ctx.set_pipelined(True, 0)
    ctx.set_pipeline_group(k, True)
    ring = nb // k * k  # the wrap-around of the ring of resident blocks falls on a group boundary
    npush = 1200
    out = []
    for mode in ("no_read", "delivered"):
        if mode == "delivered":
            ctx.set_pipeline_group(1, False); ctx.set_pipelined(False); ctx.set_pipelined(True, 3); ctx.set_pipeline_group(k, True)
        lag = min(capi.RESULT_SLOTS - 2, 7) * k
        nxt = [ctx.ticket() + 1]
        res = capi.Result()

        def collect(upto):
            while nxt[0] <= upto:
                t = C.c_uint64(nxt[0])
                ctx._chk(ctx.L.sdrpp_result_wait(ctx.h, t, C.byref(res)))
                ctx._chk(ctx.L.sdrpp_result_release(ctx.h, t))
                nxt[0] += 1

        best = 0.0
        for trial in range(4):
            t0 = time.perf_counter()
            for i in range(npush):
                ctx.push_device(base + 8 * B * (i % ring), B)
                if mode == "delivered":
                    collect(ctx.ticket() - lag)
            if mode == "delivered":
                collect(ctx.ticket())
            ctx.sync()
            dt = time.perf_counter() - t0
            if trial:
                best = max(best, B * npush / dt / 1e6)
        gs = ctx.pipeline_group_stats()
        out.append("%s %.0f MS/s" % (mode, best))
    print("B=%d k=%2d: %s  (blocks per multi-block launch %.2f, depth %d)" % (B, k, ", ".join(out), gs["multi_blocks"] / max(1, gs["multi_groups"]), ctx.pipeline_stats()["depth"]), flush=True)
    ctx.close()
