#!/bin/bash
# round 3, session 5, call 17: where a front-end workgroup's first 6 us go at sr/200 blocks (mark 3 = the first tile's samples are in LDS)
set -u
O=gpurun_out/r03zg
mkdir -p $O
timeout 200 python tools/tick_trace_run.py 3 50000 300 $O/tt.bin 2>&1 | grep -v amdgpu.ids
timeout 100 python tools/tick_trace.py $O/tt.bin 20 2>/dev/null > $O/tick_timeline_cfg3_B50000.txt
rm -f $O/tt.bin
head -8 $O/tick_timeline_cfg3_B50000.txt
