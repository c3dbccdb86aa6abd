#!/usr/bin/env python3
"""FFT branch alone against the transform size: one ordinary pass over 2^24 samples (dense framing, Nuttall window, 1024-pixel full
view), per-family HIP-event times.  tools/fft_sizes.py [lg ...]   (default 13 .. 20)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from sdrplusplus_amd import capi, workloads

    if os.environ.get("SDRPP_TOOL_LIB"):  # a switch of this TOOL (A/B builds of the library), not of the binding
        capi.DEFAULT_LIB = os.path.join(ROOT, "sdrplusplus_amd", "csrc", os.environ["SDRPP_TOOL_LIB"])

    lgs = [int(a) for a in sys.argv[1:]] or list(range(13, 21))
    push = 1 << 24
    x = workloads.synth(2, 1 << 22, seed=3)
    xd = torch.from_numpy(np.tile(x, 4).view(np.float32)).to("cuda")
    for lg in lgs:
        N = 1 << lg
        ctx = capi.Context(0, max_push=push)
        ctx.fft_configure(N, N, 0, capi.design_fft_window(2, N))
        start, size = capi.design_waterfall_view(0.0, 10e6, 10e6, N)
        ctx.fft_set_view(start, size, 1024, -120.0, 0.0)
        for _ in range(3):
            ctx.push_device(xd.data_ptr(), push)
        ctx.sync()
        ctx.timing_enable(True)
        n = 10
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.push_device(xd.data_ptr(), push)
        ctx.sync()
        dt = (time.perf_counter() - t0) / n
        fam = {k: round(v[0] / n, 4) for k, v in ctx.timing_read().items() if v[0] > 0}
        ctx.timing_enable(False)
        ctx.close()
        print(json.dumps({"lg": lg, "Gsamples_per_s": round(push / dt / 1e9, 2), "ms_per_pass": round(dt * 1e3, 4), "ps_per_sample": round(dt / push * 1e12, 2), "family_ms": fam}), flush=True)


if __name__ == "__main__":
    main()
