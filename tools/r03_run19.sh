#!/bin/bash
# round 3, session 5, call 4: rotator with 16 VFOs per workgroup (and the sweep), result copies first in a tick (A/B on the headline line)
set -u
O=gpurun_out/r03t
mkdir -p $O
timeout 600 python -m pytest tests/test_parity_vfo.py tests/test_pipelined.py -m gpu -x -q -k "rotator or nco_mode or pipelined" 2>&1 | tail -2
B="python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline"
for v in 16 8 24 64; do
  echo "== cfg4 ssb-exact, SDRPP_GPU_ROTX_VPW=$v"
  SDRPP_GPU_ROTX_VPW=$v timeout 300 $B > $O/bench_cfg4_ssb_exact_vpw$v.json 2> $O/err.log
  python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact_vpw$v.json")); print(d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items() if v > 0.2})
PY
done
H="python bench.py --no-others --no-by-push --no-cpu-baseline"
for rep in 1 2; do
for cf in 1 0; do
  echo "== headline, SDRPP_GPU_TICK_COPY_FIRST=$cf"
  SDRPP_GPU_TICK_COPY_FIRST=$cf timeout 300 $H > $O/bench_headline_copyfirst$cf.json 2> $O/err.log || tail -3 $O/err.log
  python - <<PY
import json
d=json.load(open("$O/bench_headline_copyfirst$cf.json")); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
done
done
