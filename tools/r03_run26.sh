#!/bin/bash
# round 3, session 5, call 11: tick order with FFT pass 1 late in crowded ticks: pipelined tests, timeline, rates
set -u
O=gpurun_out/r03za
mkdir -p $O
timeout 600 python -m pytest tests/test_pipelined.py -m gpu -x -q 2>&1 | tail -1
timeout 200 python tools/tick_trace_run.py 3 1000000 80 $O/tt.bin 2>&1 | grep -v amdgpu.ids
timeout 100 python tools/tick_trace.py $O/tt.bin 20 2>/dev/null > $O/tick_timeline_cfg3_B1000000.txt
rm -f $O/tt.bin
head -22 $O/tick_timeline_cfg3_B1000000.txt
timeout 200 python tools/tick_rate.py 3 1000000 500000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg3.log
timeout 200 python tools/tick_rate.py 4 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg4.log
timeout 200 python tools/tick_rate.py 2 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg2.log
