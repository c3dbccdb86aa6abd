#!/bin/bash
# Round 3, session 2, call 1: state check (GPU tests), rates of pipelined mode, tick timelines at the stream cap and for cfg 4
set -u
O=gpurun_out/r03c
mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
for spec in "3 1000000 80" "3 50000 300" "4 307200 80" "4 1000000 40"; do
  set -- $spec
  timeout 200 python tools/tick_trace_run.py $1 $2 $3 $O/tt_$1_$2.bin 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/tick_trace.py $O/tt_$1_$2.bin 20 > $O/tick_timeline_cfg$1_B$2.txt 2>&1
  rm -f $O/tt_$1_$2.bin
  cat $O/tick_timeline_cfg$1_B$2.txt
done
timeout 300 python tools/tick_rate.py 3 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg3.log | tail -3
timeout 300 python tools/tick_rate.py 4 307200 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg4.log | tail -3
