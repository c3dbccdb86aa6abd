"""Helper of test_opt_in_kernel_variants_bit_identical: one process = one setting of the opt-in switches (the library reads them once).
   python tests/variant_scenario.py OUT.npz NVFO [LIBRARY]      (LIBRARY: the emulator build for the CPU leg of the test)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from sdrplusplus_amd import capi, radio, workloads

nv = int(sys.argv[2])
if len(sys.argv) > 3:
    capi.DEFAULT_LIB = sys.argv[3]
sizes = [65536, 50000, 1000, 131072, 7, 90001]  # ragged pushes: windows, carries and the resampler phase straddle every cut
ctx = capi.Context(0, max_push=max(sizes))
vids = []
for mode, if_rate, bw, centre, _ in workloads.vfo_plan(3, nv):
    d, keep = radio.vfo_desc(10e6, if_rate, bw, centre, mode)
    vids.append(ctx.vfo_add(d, keep))
rng = np.random.default_rng(1)
outs = {v: [] for v in vids}
for n in sizes:
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.05).astype(np.complex64)
    ctx.push(x)
    for v in vids:
        outs[v].append(ctx.vfo_read(v))
np.savez(sys.argv[1], **{"v%d" % v: np.concatenate(outs[v]) for v in vids})
ctx.close()
