#!/bin/bash
# round 3, session 5, call 8: rotator hand-over stride 4 / 8 / 16
set -u
O=gpurun_out/r03x
mkdir -p $O
timeout 300 python -m pytest tests/test_parity_vfo.py -m gpu -x -q -k "four_wavefront" 2>&1 | tail -1
B="python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline"
for spec in "4 16" "8 16" "8 8" "16 8" "16 4"; do
set -- $spec
echo "== cfg4 ssb-exact SKIP=$1 VPW=$2"
SDRPP_GPU_ROTX_SKIP=$1 SDRPP_GPU_ROTX_VPW=$2 timeout 300 $B > $O/bench_cfg4_ssb_exact_skip$1_vpw$2.json 2> $O/err.log
SDRPP_GPU_ROTX_SKIP=$1 SDRPP_GPU_ROTX_VPW=$2 timeout 300 python -m pytest tests/test_parity_vfo.py -m gpu -x -q -k "four_wavefront" 2>&1 | tail -1
python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact_skip$1_vpw$2.json")); print(d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items() if v > 0.2})
PY
done
