#!/bin/bash
# round 3, session 5: copy role with four loads in flight per work-item (landing copy from host memory); small-block front end with it
set -u
O=gpurun_out/r03zm
mkdir -p $O
timeout 600 python -m pytest tests/test_pipelined.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2; do
for t in 0 256; do
  echo "== SDRPP_GPU_FCM16_MAX_TILES=$t"
  SDRPP_GPU_FCM16_MAX_TILES=$t timeout 200 python tools/tick_rate.py 3 50000 1000000 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['push'], d['pipelined_device_no_read'], d['pipelined_pinned_no_read'], d['pipelined_pinned_results_delivered'])"
done
done 2>&1 | tee $O/tick_rate_copy_batched.log
