#!/bin/bash
# round 3, session 5: the committed state once more — all GPU tests, cfg 4 line with the new tick order
set -u
O=gpurun_out/r03zk
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 300 python bench.py --cfg 4 --no-others --no-by-push --no-cpu-baseline > $O/bench_cfg4.json 2> $O/err.log
python - <<PY
import json
d=json.load(open("$O/bench_cfg4.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernel_ms_per_step"])
PY
