#!/bin/bash
# round 3, session 5: where the long first stages (cfg 4) stand in the order of a crowded tick
set -u
O=gpurun_out/r03zj
mkdir -p $O
for rep in 1 2; do
for w in 100 58 40 15; do
  echo "== SDRPP_GPU_TICK_FCL_WEIGHT=$w"
  SDRPP_GPU_TICK_FCL_WEIGHT=$w timeout 200 python tools/tick_rate.py 4 1000000 307200 2>&1 | grep -v amdgpu.ids | cut -c1-140
done
done 2>&1 | tee $O/tick_fcl_weight.log
