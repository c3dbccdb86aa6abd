#!/bin/bash
# round 3, session 5, last call: all GPU tests on the committed state; tick timelines with blocks fetched from host memory, the front end in both shapes
set -u
O=gpurun_out/r03zn
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
for t in 0 256; do
  SDRPP_GPU_FCM16_MAX_TILES=$t timeout 100 python tools/tick_trace_run.py 3 50000 300 $O/tt.bin host 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/tick_trace.py $O/tt.bin 20 2>/dev/null > $O/tick_timeline_cfg3_B50000_host_fcm16_$t.txt
  rm -f $O/tt.bin
  head -9 $O/tick_timeline_cfg3_B50000_host_fcm16_$t.txt | cut -c1-150
done
