#!/bin/bash
# round 3, session 5, call 12: grid rules of the tick's matrix roles under the new order
set -u
O=gpurun_out/r03zb
mkdir -p $O
run() { echo "== $*"; env "$@" timeout 200 python tools/tick_rate.py 3 1000000 500000 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['push'], d['pipelined_device_no_read'], d['pipelined_pinned_results_delivered'])"; }
( run X=0
  run SDRPP_GPU_TICK_TOEP_BLOCKS=192
  run SDRPP_GPU_TICK_TOEP_BLOCKS=160
  run SDRPP_GPU_TICK_TOEP_BLOCKS=128
  run SDRPP_GPU_TICK_FCM_WAVES=640
  run SDRPP_GPU_TICK_FCM_WAVES=640 SDRPP_GPU_TICK_TOEP_BLOCKS=160
  run SDRPP_GPU_TICK_FCM_WAVES=1024 SDRPP_GPU_TICK_TOEP_BLOCKS=192
  run X=0 ) 2>&1 | tee $O/tick_grid_rules_new_order.log
