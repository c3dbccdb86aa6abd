#!/usr/bin/env python3
"""Ingest rate of the headline workload (cfg 3) block by block at API-legal block sizes: the ordinary pass against pipelined mode
(sdrpp_set_pipelined: one launch per block), with the outputs left on the device and with every VFO block + the zoomed lines delivered
into page-locked host memory.  Prints one JSON object per block size.  (bench.py's by_push section reports the same modes.)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from sdrplusplus_amd import capi, workloads

    if os.environ.get("SDRPP_TOOL_LIB"):  # (a switch of this TOOL for A / B runs of library builds; the package itself has no override)
        capi.DEFAULT_LIB = os.path.join(ROOT, "sdrplusplus_amd", "csrc", os.environ["SDRPP_TOOL_LIB"])

    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    sizes = [int(a) for a in sys.argv[2:]] or [int(workloads.CFG[cfg]["sr"] / 200), 1000000]
    nvfo = workloads.CFG[cfg]["nvfo"]
    dev = torch.device("cuda", 0)
    for B in sizes:
        ctx = capi.Context(0, max_push=B)
        info = workloads.setup(ctx, cfg, dense_fft=True, data_width=1024, nvfo=nvfo or None, fft=True)
        nb = 4
        xs = [workloads.synth(cfg, B, seed=7 + i, nvfo=nvfo or None) for i in range(nb)]
        xd = [torch.from_numpy(x.view(np.float32)).to(dev) for x in xs]
        ptrs = []
        for x in xs:
            p = ctx.L.sdrpp_host_alloc(B * 8)
            C.memmove(p, x.ctypes.data, B * 8)
            ptrs.append(p)
        out = {"cfg": cfg, "push": B}
        npush = max(24, min(2000, (1 << 27) // B))

        def rate(fn, n, end):
            for i in range(8):
                fn(i)
            end()
            best = 0.0
            for _trial in range(3):
                t0 = time.perf_counter()
                for i in range(n):
                    fn(i)
                end()
                best = max(best, B * n / (time.perf_counter() - t0) / 1e6)
            return round(best, 1)

        out["ordinary_device_no_read"] = rate(lambda i: ctx.push_device(xd[i % nb].data_ptr(), B), npush, ctx.sync)
        ctx.set_pipelined(True, 0)
        out["pipelined_device_no_read"] = rate(lambda i: ctx.push_device(xd[i % nb].data_ptr(), B), npush, ctx.sync)
        out["pipelined_pinned_no_read"] = rate(lambda i: ctx.push_host_ptr_async(ptrs[i % nb], B), npush, ctx.sync)
        # host enqueue cost alone: time the loop without the final synchronisation
        t0 = time.perf_counter()
        for i in range(npush):
            ctx.push_device(xd[i % nb].data_ptr(), B)
        t1 = time.perf_counter()
        ctx.sync()
        out["pipelined_host_us_per_push"] = round((t1 - t0) / npush * 1e6, 2)
        ctx.set_pipelined(False)
        ctx.set_pipelined(True, 3)
        lag = 8
        state = {"next": ctx.ticket() + 1}  # first block whose results have not been collected yet

        def collect(upto):
            while state["next"] <= upto:
                t = C.c_uint64(state["next"])
                res = capi.Result()
                ctx._chk(ctx.L.sdrpp_result_wait(ctx.h, t, C.byref(res)))
                ctx._chk(ctx.L.sdrpp_result_release(ctx.h, t))
                state["next"] += 1

        def with_results(i):
            ctx.push_host_ptr_async(ptrs[i % nb], B)
            collect(ctx.ticket() - lag)

        def drain_results():
            collect(ctx.ticket())

        out["pipelined_pinned_results_delivered"] = rate(with_results, npush, drain_results)
        ctx.set_pipelined(False)
        tm = ctx.timing_read()
        for p in ptrs:
            ctx.L.sdrpp_host_free(p)
        ctx.close()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
