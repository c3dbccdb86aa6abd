#!/bin/bash
set -u
O=gpurun_out/r03g
mkdir -p $O
run() { echo "== $*" | tee -a $O/sweep.log; env "$@" timeout 300 python tools/tick_rate.py 3 1000000 500000 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['push'], d['pipelined_device_no_read'], d['pipelined_pinned_results_delivered'])
" | tee -a $O/sweep.log; }
run X=0
run SDRPP_GPU_TICK_FCM_WAVES=512
run SDRPP_GPU_TICK_FCM_WAVES=1024
run SDRPP_GPU_TICK_TOEP_BLOCKS=192
run SDRPP_GPU_TICK_TOEP_BLOCKS=384
run SDRPP_GPU_TICK_FCM_WAVES=512 SDRPP_GPU_TICK_TOEP_BLOCKS=192
run SDRPP_GPU_TICK_ORDER=0
timeout 200 python tools/tick_trace_run.py 3 1000000 80 $O/tt.bin 2>&1 | grep -v amdgpu.ids
timeout 100 python tools/tick_trace.py $O/tt.bin 20 2>/dev/null > $O/tick_timeline_cfg3_B1000000.txt
rm -f $O/tt.bin
cat $O/tick_timeline_cfg3_B1000000.txt
timeout 300 python tools/tick_rate.py 4 307200 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg4.log
