#!/bin/bash
# round 3, session 5, call 7: rotator with every 4th phase handed over (the applying wavefronts take the steps in between)
set -u
O=gpurun_out/r03w
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
B="python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline"
for v in 16 8; do
echo "== cfg4 ssb-exact VPW=$v"
SDRPP_GPU_ROTX_VPW=$v timeout 300 $B > $O/bench_cfg4_ssb_exact_vpw$v.json 2> $O/err.log
python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact_vpw$v.json")); print(d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items() if v > 0.2})
PY
done
