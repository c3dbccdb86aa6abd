#!/bin/bash
# round 3, session 5, call 10: where FFT pass 1 stands in the order of a tick
set -u
O=gpurun_out/r03z
mkdir -p $O
for rep in 1 2; do
for w in 70 55 45; do
  echo "== SDRPP_GPU_TICK_P1_WEIGHT=$w"
  SDRPP_GPU_TICK_P1_WEIGHT=$w timeout 200 python tools/tick_rate.py 3 1000000 200000 50000 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
done 2>&1 | tee $O/tick_p1_weight.log
