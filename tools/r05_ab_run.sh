#!/bin/bash
# after the reference-rotator roles went into the tick kernel: device suite of the touched areas, then base build vs working build on the default ticks
mkdir -p gpurun_out
T=${1:-r05zp}
( timeout 900 python -m pytest tests -m gpu -x -q -k "rotator or nco or pipelined or cfg4" 2>&1 | tail -3 ) | tee gpurun_out/${T}_pytest_rotator.log
timeout 400 python tools/ab_tick.py --cfg 3 --push 1000000 50000 --rounds 2 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so > gpurun_out/${T}_ab_cfg3.log 2>&1
timeout 300 python tools/ab_tick.py --cfg 4 --push 1000000 --rounds 2 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so > gpurun_out/${T}_ab_cfg4.log 2>&1
grep -A5 summary gpurun_out/${T}_ab_cfg3.log gpurun_out/${T}_ab_cfg4.log
