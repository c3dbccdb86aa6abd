#!/bin/bash
# A / B on one box (tools/ab_tick.py): wavefront priority in the Toeplitz roles (high outside their matrix loops, as the front ends have it)
mkdir -p gpurun_out
T=${1:-r05zn}
timeout 500 python tools/ab_tick.py --cfg 3 --push 1000000 50000 --rounds 3 def=libsdrpp_gpu.so tprio=libsdrpp_gpu_tprio.so > gpurun_out/${T}_ab_cfg3.log 2>&1
timeout 300 python tools/ab_tick.py --cfg 4 --push 1000000 --rounds 2 def=libsdrpp_gpu.so tprio=libsdrpp_gpu_tprio.so > gpurun_out/${T}_ab_cfg4.log 2>&1
grep -A5 summary gpurun_out/${T}_ab_cfg3.log gpurun_out/${T}_ab_cfg4.log
