#!/bin/bash
# per-workgroup timelines of the tick (ticktrace build) for a few workloads
T=${1:-r05zg}
O=gpurun_out
mkdir -p $O
for spec in "4 1000000 60 fcl_pf:62" "4 307200 80 fcl_pf:38"; do
  set -- $spec
  timeout 200 python tools/tick_trace_run.py $1 $2 $3 $O/tt.bin 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/tick_trace.py $O/tt.bin 20 $4 2>/dev/null > $O/${T}_tick_timeline_cfg$1_B$2.txt
  rm -f $O/tt.bin
done
grep -A8 "per job" $O/${T}_tick_timeline_cfg4_B*.txt
timeout 300 python tools/ab_tick.py --cfg 4 --push 1000000 307200 --rounds 2 prev=libsdrpp_gpu_base.so new=libsdrpp_gpu.so > $O/${T}_ab.log 2>&1; grep -A5 summary $O/${T}_ab.log
timeout 600 python -m pytest tests -m gpu -x -q -k "long_first or cfg4 or mixed_modes" 2>&1 | tail -3
