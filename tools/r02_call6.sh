#!/bin/bash
# cfg 4 front kernel with register prefetch: parity (GPU tests touching it) + cfg 4 / cfg 3 bench lines
set -u
O=gpurun_out/r02c6
mkdir -p $O
( timeout 900 python -m pytest tests/ -x -q -m gpu ) 2>&1 | tail -4 | tee $O/log.txt
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["kernel_ms_per_step"])'
for c in 4 4 3; do
    echo "-- bench cfg $c" | tee -a $O/log.txt
    timeout 300 python bench.py --cfg $c --steps 20 --no-cpu-baseline --no-by-push 2>/dev/null | python -c "$P" 2>&1 | tee -a $O/log.txt
done
