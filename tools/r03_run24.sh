#!/bin/bash
# round 3, session 5: final state — GPU tests, then tools/profile_all.sh (default bench line, cfg 2 / cfg 4 lines, kernel trace, PMC passes, tick timelines)
set -u
O=gpurun_out/r03zh
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
bash tools/profile_all.sh r03zh
