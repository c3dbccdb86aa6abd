#!/bin/bash
set -u
O=gpurun_out/r03b
mkdir -p $O
timeout 300 python -m pytest tests/test_pipelined.py -m gpu -x -q 2>&1 | tail -3
timeout 120 python tools/tick_trace_run.py 3 50000 300 $O/tt_50000.bin 2>&1 | grep -v amdgpu.ids
timeout 60 python tools/tick_trace.py $O/tt_50000.bin 60 > $O/tick_timeline_cfg3_B50000.txt 2>&1
cat $O/tick_timeline_cfg3_B50000.txt
timeout 300 python tools/tick_rate.py 3 2>&1 | grep -v amdgpu.ids | tail -3
