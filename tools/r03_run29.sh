#!/bin/bash
# round 3, session 5, call 16: + the front end keeps its NCO constants in registers
set -u
O=gpurun_out/r03zf
mkdir -p $O
timeout 900 python -m pytest tests/test_pipelined.py tests/test_parity_vfo.py tests/test_full_configs_gpu.py -m gpu -x -q 2>&1 | tail -1
timeout 200 python tools/tick_rate.py 3 1000000 200000 50000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg3.log
timeout 200 python tools/tick_rate.py 4 1000000 307200 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg4.log
timeout 200 python tools/tick_trace_run.py 3 50000 300 $O/tt.bin 2>&1 | grep -v amdgpu.ids
timeout 100 python tools/tick_trace.py $O/tt.bin 20 2>/dev/null > $O/tick_timeline_cfg3_B50000.txt
rm -f $O/tt.bin
head -14 $O/tick_timeline_cfg3_B50000.txt
timeout 300 python bench.py --mode ordinary --push 16777216 --ref-block 0 --steps 10 --warmup 2 --no-others --no-by-push --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ceiling', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
