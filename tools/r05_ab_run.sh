#!/bin/bash
# A / B of the base build (libsdrpp_gpu_base.so) against the working build on one box
mkdir -p gpurun_out
T=${1:-r05ze}
timeout 400 python tools/ab_tick.py --cfg 4 --push 1000000 307200 --rounds 2 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so > gpurun_out/${T}_ab_cfg4.log 2>&1
timeout 300 python tools/ab_tick.py --cfg 3 --push 1000000 50000 --rounds 2 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so > gpurun_out/${T}_ab_cfg3.log 2>&1
timeout 300 python tools/ab_tick.py --cfg 4 --push 1000000 --af --rounds 1 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so > gpurun_out/${T}_ab_cfg4_af.log 2>&1
tail -4 gpurun_out/${T}_ab_cfg4.log gpurun_out/${T}_ab_cfg3.log gpurun_out/${T}_ab_cfg4_af.log
