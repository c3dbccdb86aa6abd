#!/bin/bash
# stream-allocation skew experiment (channel camping hypothesis)
set -u
O=gpurun_out/r02c9
mkdir -p $O
for sk in 0 256 4352 12544 33024 0; do
    echo "-- VFO bank alone, skew $sk" | tee -a $O/log.txt
    SDRPP_GPU_STREAM_SKEW=$sk timeout 120 python tools/vfo_only_time.py 16777216 32 10 2>&1 | tail -1 | tee -a $O/log.txt
done
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernel_ms_per_step"])'
for sk in 0 4352 12544 0; do
    echo "-- bench, skew $sk" | tee -a $O/log.txt
    SDRPP_GPU_STREAM_SKEW=$sk timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-by-push 2>/dev/null | python -c "$P" 2>&1 | tee -a $O/log.txt
done
