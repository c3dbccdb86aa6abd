#!/bin/bash
# Round-2 GPU call 2: the whole -m gpu suite (new NCO / reference-block / retune tests, full-size cfg 4 and 2^24 parity) + bench regression.
set -u
mkdir -p gpurun_out/r02c2
O=gpurun_out/r02c2
( time timeout 1500 python -m pytest tests/ -x -q -m gpu -s --durations=15 ) > $O/pytest_gpu.log 2>&1
tail -40 $O/pytest_gpu.log
python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
