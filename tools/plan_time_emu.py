"""Host planner timing WITHOUT a device: the emulator build of the library (tests/emu) with SDRPP_EMU_NO_EXEC=1 launches nothing, so the wall time of
a pipelined push is the host work of a block (chains, grouping, job tables, role queue).  Diagnostic; numbers are of THIS machine's CPU.
    python tools/plan_time_emu.py [cfg] [push] [blocks]"""
import os, sys, time
os.environ["SDRPP_EMU_NO_EXEC"] = "1"
os.environ.setdefault("SDRPP_GPU_HOSTPROF", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sdrplusplus_amd import capi, workloads
capi.DEFAULT_LIB = os.path.join(ROOT, "tests", "emu", "libsdrpp_gpu_emu.so")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ctx = capi.Context(0, max_push=B)
info = workloads.setup(ctx, cfg, dense_fft=True, data_width=1024)
ctx.set_pipelined(True, 2)
x = np.zeros(B, np.complex64)
ptr = x.ctypes.data  # the emulator's "device" memory is host memory
for t in range(20):
    ctx.push_device(ptr, B)
t0 = time.perf_counter()
for t in range(nblk):
    ctx.push_device(ptr, B)
dt = (time.perf_counter() - t0) / nblk
st = ctx.pipeline_stats()
print("cfg %d, %d-sample blocks: %.1f us of host work per block (%d blocks as ticks, %d levels, %d bytes of job tables per block)" % (cfg, B, dt * 1e6, st["tick_blocks"], st["depth"], st["table_bytes"]))
ctx.close()
