#!/bin/bash
# after the small-bank front ends went into the tick kernel: device suite of the pipelined tests, then base build vs working build on the default ticks
mkdir -p gpurun_out
T=${1:-r05zt}
( timeout 900 python -m pytest tests/test_pipelined.py -m gpu -x -q 2>&1 | tail -3 ) | tee gpurun_out/${T}_pytest_pipelined.log
timeout 400 python tools/ab_tick.py --cfg 3 --push 1000000 50000 --rounds 3 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so > gpurun_out/${T}_ab_cfg3.log 2>&1
grep -A5 summary gpurun_out/${T}_ab_cfg3.log
