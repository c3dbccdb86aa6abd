#!/usr/bin/env python3
"""Diagnostic: cfg 3 at bench geometry, pipelined (result flags 1) against ordinary passes, block by block; where do they differ?
usage: pipelined_diff_probe.py [B] [nblk] [lag]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sdrplusplus_amd import capi, workloads
from test_full_configs_gpu import _synth_threaded

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 13
lag = int(sys.argv[3]) if len(sys.argv) > 3 else 9
x = _synth_threaded(3, B * nblk, seed=0x3A + nblk)
t = torch.from_numpy(x.view(np.float32)).to("cuda:0")
torch.cuda.synchronize()
ptr = lambda b: t.data_ptr() + 8 * b * B

def setup(pipelined):
    ctx = capi.Context(0, max_push=B)
    info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=32)
    if B > 50000:
        ctx.set_reference_block(50000)
    if pipelined:
        ctx.set_pipelined(True, 7)
    return ctx, info

ca, ia = setup(False)
ref = []
for b in range(nblk):
    ca.push_device(ptr(b), B)
    ref.append({v: ca.vfo_read(v).copy() for v in ia["vids"]})
ca.close()
cb, ib = setup(True)
got = []
for b in range(nblk):
    cb.push_device(ptr(b), B)
    if b + 1 > lag:
        got.append(cb.result_wait(b + 1 - lag)); cb.result_release(b + 1 - lag)
for tk in range(len(got) + 1, nblk + 1):
    got.append(cb.result_wait(tk)); cb.result_release(tk)
print("stats", cb.pipeline_stats())
for b in range(nblk):
    bad = []
    for va, vb in zip(ia["vids"], ib["vids"]):
        a, g = ref[b][va], got[b]["vfo"][vb]
        if a.shape != g.shape:
            bad.append((vb, "shape", a.shape, g.shape)); continue
        d = np.nonzero(a.view(np.uint32) != g.view(np.uint32))[0]
        if len(d):
            bad.append((vb, len(d), int(d[0]), int(d[-1]), float(np.max(np.abs(a - g)))))
    print("block", b + 1, "differing VFOs:", len(bad), bad[:6])
cb.close()
