#!/usr/bin/env python3
"""Soak: the same long stream through the pipelined path three times — one block per launch, adaptive groups, fixed groups of GROUP_MAX — every VFO block and
every line of every block delivered and hashed on the host; the three runs must deliver identical bytes for every block (launch grouping follows the timing of
host and device, the results may not).   python tools/r06_soak.py [block] [blocks] [nvfo]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, xxhash
from sdrplusplus_amd import capi, workloads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
nvfo = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)
ring = 64 if B <= 100000 else 16
xs = np.concatenate([workloads.synth(3, B, seed=100 + i, nvfo=nvfo) for i in range(ring)])
xd = torch.from_numpy(xs.view(np.float32)).to(dev)
base = xd.data_ptr()


def run(k, adaptive):
    ctx = capi.Context(0, max_push=B * k)
    info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=nvfo)
    ctx.set_reference_block(min(B, 50000))
    ctx.set_pipelined(True, 3)
    ctx.set_pipeline_group(k, adaptive)
    lag = min(capi.RESULT_SLOTS - 2, 8) * k
    res = capi.Result()
    digests = []
    nxt = 1
    t0 = time.perf_counter()

    def collect(upto):
        nonlocal nxt
        while nxt <= upto:
            t = C.c_uint64(nxt)
            ctx._chk(ctx.L.sdrpp_result_wait(ctx.h, t, C.byref(res)))
            h = xxhash.xxh64()
            for i in range(res.n_vfo):
                n = res.counts[i]
                if n > 0:
                    h.update(C.string_at(C.addressof(res.samples.contents) + 8 * res.offsets[i], 8 * n))
            if res.n_lines > 0:
                h.update(C.string_at(res.zoomed, 4 * res.n_lines * res.data_width))
                h.update(C.string_at(res.index, 4 * res.n_lines * res.data_width))
            digests.append((res.n_lines, h.intdigest()))
            ctx._chk(ctx.L.sdrpp_result_release(ctx.h, t))
            nxt += 1

    for i in range(NB):
        ctx.push_device(base + 8 * B * (i % ring), B)
        collect(i + 1 - lag)
    collect(NB)
    dt = time.perf_counter() - t0
    gs = ctx.pipeline_group_stats()
    st = ctx.pipeline_stats()
    ctx.close()
    return digests, dt, gs, st


runs = [("one block per launch", 1, False), ("adaptive, up to 4", 4, True), ("fixed groups of %d" % min(capi.GROUP_MAX, 8 if B > 100000 else capi.GROUP_MAX), min(capi.GROUP_MAX, 8 if B > 100000 else capi.GROUP_MAX), False)]
ref = None
for name, k, ad in runs:
    d, dt, gs, st = run(k, ad)
    print("%-24s %d blocks of %d samples in %.2f s = %.0f MS/s delivered; launches %d (groups of several blocks %d, largest %d); tick blocks %d, pass blocks %d; lines %d"
          % (name, NB, B, dt, NB * B / dt / 1e6, gs["groups"], gs["multi_groups"], gs["largest"], st["tick_blocks"], st["pass_blocks"], sum(x[0] for x in d)), flush=True)
    if ref is None:
        ref = d
    else:
        bad = [i + 1 for i, (a, b) in enumerate(zip(ref, d)) if a != b]
        print("   identical to the first run: %s" % ("yes, all %d blocks" % NB if not bad and len(d) == len(ref) else "NO — first differing blocks %s" % bad[:8]), flush=True)
        if bad or len(d) != len(ref):
            sys.exit(1)
print("soak ok")
