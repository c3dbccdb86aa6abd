#!/bin/bash
# A / B on one box (tools/ab_tick.py): the long first stages' tiles per wavefront (32-row / 16-row jobs), cfg 4 at both block sizes
mkdir -p gpurun_out
T=${1:-r05zl}
L=libsdrpp_gpu.so
timeout 700 python tools/ab_tick.py --cfg 4 --push 1000000 --rounds 2 def=$L t3_6=$L,SDRPP_GPU_FCL_TPW=3,SDRPP_GPU_FCL_TPW16=6 t3_6w95=$L,SDRPP_GPU_FCL_TPW=3,SDRPP_GPU_FCL_TPW16=6,SDRPP_GPU_TICK_FCL_WEIGHT=95 t2_6=$L,SDRPP_GPU_FCL_TPW=2,SDRPP_GPU_FCL_TPW16=6 t3_4=$L,SDRPP_GPU_FCL_TPW=3,SDRPP_GPU_FCL_TPW16=4 t2_3=$L,SDRPP_GPU_FCL_TPW=2,SDRPP_GPU_FCL_TPW16=3 t1_4=$L,SDRPP_GPU_FCL_TPW=1,SDRPP_GPU_FCL_TPW16=4 > gpurun_out/${T}_ab_cfg4.log 2>&1
grep -A8 summary gpurun_out/${T}_ab_cfg4.log
