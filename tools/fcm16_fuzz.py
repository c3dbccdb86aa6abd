#!/usr/bin/env python3
"""Randomised comparison of the small-block front end (vfo_frontcm16_body, SDRPP_GPU_FCM16_MAX_TILES=4096) with the 32 x 32 x 2 form (=0) on the
CPU emulator: 17 / 20 / 32 WFM VFOs, random push lengths, retunes and resets between pushes, ordinary passes (even seeds) or pipelined mode
(odd seeds); every output block must be bit-identical.   tools/fcm16_fuzz.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sdrplusplus_amd import capi, radio, workloads
capi.DEFAULT_LIB = os.path.join(ROOT, "tests", "emu", "libsdrpp_gpu_emu.so")  # a switch of this TOOL, not of the binding
seed = int(sys.argv[1])
r = np.random.default_rng(seed)
nv = int(r.choice([17, 20, 32]))
plan = workloads.vfo_plan(3, nv)
npush = 14
cuts = [int(r.integers(1, 40)) if r.random() < 0.2 else int(r.integers(40, 30000)) for _ in range(npush)]
ops = [("retune", int(r.integers(0, nv)), float(r.uniform(-0.2, 0.2) * 10e6)) if r.random() < 0.2 else (("reset", int(r.integers(0, nv))) if r.random() < 0.1 else None) for _ in range(npush)]
x = workloads.synth(3, sum(cuts), seed=seed, nvfo=nv)
pipelined = bool(seed & 1)
outs = []
for small in ("0", "4096"):
    os.environ["SDRPP_GPU_FCM16_MAX_TILES"] = small
    ctx = capi.Context(0, max_push=max(cuts))
    vids = []
    for m, rate, bw, c, _ in plan:
        d, keep = radio.vfo_desc(10e6, rate, bw, c, m)
        vids.append(ctx.vfo_add(d, keep))
    if pipelined:
        ctx.set_pipelined(True, 1)
    got, pos = [], 0
    for n, op in zip(cuts, ops):
        if op and op[0] == "retune":
            ctx.vfo_set_phase_delta(vids[op[1]], *capi.design_phase_delta(op[2], 10e6))
        elif op and op[0] == "reset":
            ctx.vfo_reset(vids[op[1]])
        ctx.push(x[pos:pos + n]); pos += n
        if pipelined:
            res = ctx.result_wait(ctx.ticket(), copy=True)
            got.append([res["vfo"][v] for v in vids]); ctx.result_release(res["ticket"])
        else:
            got.append([a.copy() for a in ctx.vfo_read_many(vids)] + [a.copy() for a in ctx.vfo_read_many(vids, which=[1] * nv)])
    outs.append(got); ctx.close()
bad = 0
for ga, gb in zip(*outs):
    for a, b in zip(ga, gb):
        if a.shape != b.shape or not np.array_equal(a, b): bad += 1
print("seed", seed, "nv", nv, "pipelined", pipelined, "pushes", cuts[:6], "mismatches", bad)
sys.exit(1 if bad else 0)
