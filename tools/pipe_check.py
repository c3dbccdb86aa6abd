#!/usr/bin/env python3
"""Outputs of an FM VFO bank over a sequence of ragged pushes as hashes, for sdrpp_set_backend_pipeline = 0 / 1 / forced segment counts (the
   pipelined back end must be bit-identical to the separate launches).   tools/pipe_check.py emu|gpu [n_total] [mode ...]"""
import hashlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sdrplusplus_amd import capi, radio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "emu":
    capi.DEFAULT_LIB = os.path.join(ROOT, "tests", "emu", "libsdrpp_gpu_emu.so")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
modes = [int(a) for a in sys.argv[3:]] or [0, 1, 3]
sr = 10e6
r = np.random.default_rng(5)
x = ((r.standard_normal(n) + 1j * r.standard_normal(n)) * 0.05).astype(np.complex64)
t = np.arange(n)
x += (0.5 * np.exp(2j * np.pi * (1.35e6 / sr * t + 3.0 * np.sin(2 * np.pi * 3e3 / sr * t)))).astype(np.complex64)
cuts = [50000, 1, 7, 4096, 65536, 100, 50000, 131072, 333]
cuts.append(n - sum(cuts))
assert cuts[-1] > 0
for pm in modes:
    print('pipeline mode', pm)
    ctx = capi.Context(0, max_push=max(cuts))
    ctx.set_backend_pipeline(pm)
    specs = [("WFM", 1.35e6), ("WFM", -2.0e6), ("NFM", 0.4e6)]
    vids = []
    for mode, off in specs:
        if_rate, bw = radio.RADIO_DEFAULTS.get(mode, (250e3, 250e3))
        d, keep = radio.vfo_desc(sr, if_rate, bw, off, mode)
        vids.append(ctx.vfo_add(d, keep))
    pos = 0
    ha, hi = [hashlib.sha256() for _ in vids], [hashlib.sha256() for _ in vids]
    tot = [0] * len(vids)
    for c in cuts:
        ctx.push(x[pos:pos + c])
        pos += c
        for k, v in enumerate(vids):
            a, i = ctx.vfo_read(v), ctx.vfo_read_if(v)
            ha[k].update(np.ascontiguousarray(a).tobytes())
            hi[k].update(np.ascontiguousarray(i).tobytes())
            tot[k] += len(a)
    for k in range(len(vids)):
        print(specs[k][0], tot[k], ha[k].hexdigest()[:16], hi[k].hexdigest()[:16])
    ctx.close()
