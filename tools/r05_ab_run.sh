#!/bin/bash
# A / B on one box (tools/ab_tick.py): variants as arguments after the tag, cfg 4 at both block sizes
mkdir -p gpurun_out
T=${1:-r05zk}
L=libsdrpp_gpu.so
timeout 500 python tools/ab_tick.py --cfg 4 --push 1000000 307200 --rounds 3 wide=$L narrow=$L,SDRPP_GPU_TICK_FIR_WIDE=0 > gpurun_out/${T}_ab_cfg4.log 2>&1
grep -A6 summary gpurun_out/${T}_ab_cfg4.log
