#!/usr/bin/env python3
"""Randomised comparison of the pipelined FM back end with one launch per stage: random push lengths, retunes, resets, channel-bandwidth
   changes, VFOs added and removed between pushes — outputs must be bit-identical.   tools/pipe_fuzz.py emu|gpu [seed] [npush]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sdrplusplus_amd import capi, radio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "emu":
    capi.DEFAULT_LIB = os.path.join(ROOT, "tests", "emu", "libsdrpp_gpu_emu.so")
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
npush = int(sys.argv[3]) if len(sys.argv) > 3 else 120
big = 200000 if sys.argv[1] == "emu" else 3000000


def scenario(sr, modes, seed):
    r = np.random.default_rng(seed)
    cuts, ops = [], []
    for k in range(npush):
        u = r.random()
        cuts.append(int(r.integers(0, 40)) if u < 0.3 else (int(r.integers(40, 3000)) if u < 0.8 else (int(r.integers(3000, 60000)) if u < 0.97 else big)))
        v = r.random()
        ops.append(("retune", int(r.integers(0, len(modes))), float(r.uniform(-0.3, 0.3) * sr)) if v < 0.15 else
                   (("reset", int(r.integers(0, len(modes)))) if v < 0.2 else
                    (("bw", int(r.integers(0, len(modes))), float(r.uniform(0.3, 0.9))) if v < 0.25 else
                     (("readd", int(r.integers(0, len(modes)))) if v < 0.29 else None))))
    n = sum(cuts)
    t = np.arange(n)
    x = ((r.standard_normal(n) + 1j * r.standard_normal(n)) * 0.05 + 0.4 * np.exp(2j * np.pi * (0.11 * t + 2.0 * np.sin(2 * np.pi * 3e-4 * t)))).astype(np.complex64)
    return cuts, ops, x


def run(sr, modes, cuts, ops, x, pipe):
    ctx = capi.Context(0, max_push=max(max(cuts), 1))
    ctx.set_backend_pipeline(pipe)
    keepalive, vids, descs = [], [], []
    for m, off in modes:
        if_rate, bw = radio.RADIO_DEFAULTS.get(m, (250e3, 250e3))
        d, keep = radio.vfo_desc(sr, if_rate, bw, off, m)
        keepalive.append(keep)
        descs.append((m, if_rate, bw, off))
        vids.append(ctx.vfo_add(d, keep))
    out = [[] for _ in modes]
    pos = 0
    for c, op in zip(cuts, ops):
        if op:
            k = op[1]
            m, if_rate, bw, off = descs[k]
            if op[0] == "retune":
                ctx.vfo_set_phase_delta(vids[k], *capi.design_phase_delta(-op[2], sr))
            elif op[0] == "reset":
                ctx.vfo_reset(vids[k])
            elif op[0] == "bw" and bw < if_rate:
                ctx.vfo_set_channel_taps(vids[k], capi.design_low_pass(bw * op[2] / 2, bw * op[2] / 2 * 0.1, if_rate))
            elif op[0] == "readd":
                ctx.vfo_remove(vids[k])
                d, keep = radio.vfo_desc(sr, if_rate, bw, off, m)
                keepalive.append(keep)
                vids[k] = ctx.vfo_add(d, keep)
        ctx.push(x[pos:pos + c])
        pos += c
        for k, v in enumerate(vids):
            out[k].append(ctx.vfo_read(v).copy())
            out[k].append(ctx.vfo_read_if(v).view(np.float32).reshape(-1, 2).copy())
    ctx.close()
    return [np.concatenate(o) if o else np.zeros((0, 2), np.float32) for o in out]


bad = 0
for sr, modes in ((10e6, [("WFM", 1.35e6), ("WFM", -2.0e6), ("AM", 0.3e6)]), (2.4e6, [("WFM", 0.3e6), ("NFM", -0.5e6)])):
    cuts, ops, x = scenario(sr, modes, seed)
    ref = run(sr, modes, cuts, ops, x, 0)
    for pipe in (1, 3):
        got = run(sr, modes, cuts, ops, x, pipe)
        for k, (a, b) in enumerate(zip(ref, got)):
            same = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
            bad += 0 if same else 1
            print("sr %.1f MS/s %s pipeline %d: %d values %s" % (sr / 1e6, modes[k][0], pipe, a.size, "identical" if same else "DIFFERENT"))
sys.exit(1 if bad else 0)
