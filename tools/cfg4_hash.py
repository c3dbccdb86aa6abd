#!/usr/bin/env python3
"""Outputs of a cfg-4 style bank (long /64 first stages of three different lengths) as one hash: run with an experiment switch of the
   library set either way (e.g. SDRPP_GPU_FCL_FAST=0 / 1) and compare — variants of the front kernels must give the same bits.
   tools/fcs_check.py emu|gpu [nvfo] [n_total]"""
import hashlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sdrplusplus_amd import capi, radio, workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "emu":
    capi.DEFAULT_LIB = os.path.join(ROOT, "tests", "emu", "libsdrpp_gpu_emu.so")
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 54
n = int(sys.argv[3]) if len(sys.argv) > 3 else 400000
sr = 61.44e6
x = workloads.synth(4, n, seed=13, nvfo=nv)
plan = workloads.vfo_plan(4, nv)
cuts = [150000, 3, 100003]
cuts.append(n - sum(cuts))
ctx = capi.Context(0, max_push=max(cuts))
if os.environ.get("SDRPP_TOOL_PIPELINE"):  # 0: one launch per back-end stage, >= 2: forced segment count
    ctx.set_backend_pipeline(int(os.environ["SDRPP_TOOL_PIPELINE"]))
vids = []
for mode, if_rate, bw, centre, _ in plan:
    d, keep = radio.vfo_desc(sr, if_rate, bw, centre, mode)
    vids.append(ctx.vfo_add(d, keep))
h = hashlib.sha256()
pos = 0
for c in cuts:
    ctx.push(x[pos:pos + c])
    pos += c
    for v in vids:
        h.update(np.ascontiguousarray(ctx.vfo_read_if(v)).tobytes())
        h.update(np.ascontiguousarray(ctx.vfo_read(v)).tobytes())
print({k: v for k, v in os.environ.items() if k.startswith("SDRPP_GPU_") or k.startswith("SDRPP_TOOL_")}, nv, "VFOs", h.hexdigest()[:24])
ctx.close()
