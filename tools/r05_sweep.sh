#!/bin/bash
# re-tune sweep of the tick's grid knobs after the round-5 role changes (one box, variants interleaved)
mkdir -p gpurun_out
T=${1:-r05zf}
L=libsdrpp_gpu.so
timeout 500 python tools/ab_tick.py --cfg 3 --push 1000000 --rounds 2 def=$L tb128=$L,SDRPP_GPU_TICK_TOEP_BLOCKS=128 tb192=$L,SDRPP_GPU_TICK_TOEP_BLOCKS=192 tb384=$L,SDRPP_GPU_TICK_TOEP_BLOCKS=384 tb512=$L,SDRPP_GPU_TICK_TOEP_BLOCKS=512 fw512=$L,SDRPP_GPU_TICK_FCM_WAVES=512 fw1024=$L,SDRPP_GPU_TICK_FCM_WAVES=1024 zg16=$L,SDRPP_GPU_TICK_ZOOM_GROUPS=16 > gpurun_out/${T}_sweep_cfg3.log 2>&1
timeout 500 python tools/ab_tick.py --cfg 4 --push 1000000 307200 --rounds 1 def=$L tb128=$L,SDRPP_GPU_TICK_TOEP_BLOCKS=128 tb512=$L,SDRPP_GPU_TICK_TOEP_BLOCKS=512 fclw60=$L,SDRPP_GPU_TICK_FCL_WEIGHT=60 fclw95=$L,SDRPP_GPU_TICK_FCL_WEIGHT=95 > gpurun_out/${T}_sweep_cfg4.log 2>&1
grep -A12 summary gpurun_out/${T}_sweep_cfg3.log gpurun_out/${T}_sweep_cfg4.log
