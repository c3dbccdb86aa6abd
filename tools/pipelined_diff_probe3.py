#!/usr/bin/env python3
"""Diagnostic: pipelined blocks collected at once through the completion flag only (no stream synchronisation in the loop); on a mismatch
with the ordinary pass: the slot again after a synchronisation, and the device buffers."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sdrplusplus_amd import capi, workloads
from test_full_configs_gpu import _synth_threaded

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 20
FLAGS = int(sys.argv[3]) if len(sys.argv) > 3 else 7
LAG = int(sys.argv[4]) if len(sys.argv) > 4 else 0
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
        ctx.set_pipelined(True, FLAGS)
    return ctx, info

ca, ia = setup(False)
ref = []
for b in range(nblk):
    ca.push_device(ptr(b), B)
    ref.append([ca.vfo_read(v).copy() for v in ia["vids"]])
ca.close()
cb, ib = setup(True)
def ranges(idx):
    if len(idx) == 0: return []
    cuts = np.nonzero(np.diff(idx) > 1)[0]
    starts = np.concatenate([[idx[0]], idx[cuts + 1]]); ends = np.concatenate([idx[cuts], [idx[-1]]])
    return list(zip(starts.tolist(), ends.tolist()))
nbad = 0
def check(tk):
    global nbad
    r = cb.result_wait(tk, copy=False)
    for k, v in enumerate(ib["vids"]):
        a = r["vfo"][v]
        rf = ref[tk - 1][k]
        bad = np.nonzero(rf.view(np.uint32).ravel() != a.view(np.uint32).ravel())[0]
        if len(bad):
            nbad += 1
            A = a.copy()
            cb.sync()
            bad2 = np.nonzero(rf.view(np.uint32).ravel() != a.view(np.uint32).ravel())[0]
            rr = ranges(bad)
            print("block %d vfo idx %d: %d floats differ at the flag in %d ranges %s ; after sync %d differ; got values %s ref %s" % (
                tk, k, len(bad), len(rr), rr[:8], len(bad2), A.ravel()[bad[:6]].tolist(), rf.ravel()[bad[:6]].tolist()))
            if tk == cb.ticket():
                c = cb.vfo_read(v)
                print("    device buffer differs in %d floats" % int(np.count_nonzero(rf.view(np.uint32) != c.view(np.uint32))))
    cb.result_release(tk)
for b in range(nblk):
    cb.push_device(ptr(b), B)
    if b + 1 > LAG:
        check(b + 1 - LAG)
for tk in range(max(1, nblk - LAG + 1), nblk + 1):
    check(tk)
print("flags %d lag %d: %d (block, vfo) pairs differed; stats %s" % (FLAGS, LAG, nbad, cb.pipeline_stats()["ticks"]))
cb.close()
