#!/bin/bash
# round 3, session 5, call 1: state after the FFT tile-walk / rotator commits (results of the previous session's calls were lost with the container)
set -u
O=gpurun_out/r03q
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 300 python tools/fft_sizes.py 2>&1 | grep -v amdgpu.ids | tee $O/fft_sizes.log
timeout 200 python tools/tick_rate.py 3 1000000 50000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg3.log
timeout 200 python tools/tick_rate.py 2 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg2.log
timeout 200 python tools/tick_rate.py 4 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg4.log
echo "== cfg4 ssb-exact"
timeout 400 python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline > $O/bench_cfg4_ssb_exact.json 2> $O/bench_cfg4_ssb_exact.err
python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact.json")); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
echo "== default bench"
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"]); print(json.dumps(d["other_configs"])[:1500]); print(json.dumps(d["by_push"])[:1500])
PY
