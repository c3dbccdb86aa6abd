#!/usr/bin/env python3
"""Diagnostic: is a sporadic difference in a pipelined block's result slot a HOST-VISIBILITY race (data arrives after the completion flag)
or wrong data on the device?  Block by block, lag 0: A = slot right after sdrpp_result_wait, B = the same slot after a stream synchronisation,
C = the device buffers (sdrpp_vfo_read), all against ordinary passes."""
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
def ndiff(a, g):
    return int(np.count_nonzero(a.view(np.uint32) != g.view(np.uint32)))
tot = [0, 0, 0]
for b in range(nblk):
    cb.push_device(ptr(b), B)
    r = cb.result_wait(b + 1, copy=False)
    A = [r["vfo"][v].copy() for v in ib["vids"]]
    time_zero = [int(np.count_nonzero(a == 0.0)) for a in A]
    cb.sync()
    Bv = [r["vfo"][v].copy() for v in ib["vids"]]
    cb.result_release(b + 1)
    Cv = [cb.vfo_read(v) for v in ib["vids"]]
    da = [ndiff(p, q) for p, q in zip(ref[b], A)]
    db = [ndiff(p, q) for p, q in zip(ref[b], Bv)]
    dc = [ndiff(p, q) for p, q in zip(ref[b], Cv)]
    tot[0] += sum(da); tot[1] += sum(db); tot[2] += sum(dc)
    if sum(da) or sum(db) or sum(dc):
        k = int(np.argmax(np.array(da) + np.array(db) + np.array(dc)))
        bad = np.nonzero(ref[b][k].view(np.uint32).ravel() != A[k].view(np.uint32).ravel())[0]
        print("block %d: slot-at-flag %s  slot-after-sync %s  device %s  | worst vfo idx %d: zeros in A %d, first bad floats %s" % (
            b + 1, [d for d in da if d], [d for d in db if d], [d for d in dc if d], k, time_zero[k], bad[:12].tolist()))
print("total differing floats: slot-at-flag %d, slot-after-sync %d, device buffers %d" % tuple(tot))
cb.close()
