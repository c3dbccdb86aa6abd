#!/bin/bash
# round 3, session 4, call 3: new FFT defaults (16 columns, pass 1 walking from 1024 workgroups, pass 2 one tile each), rotator with requests six chunks ahead
set -u
O=gpurun_out/r03p
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 300 python tools/fft_sizes.py 2>&1 | grep -v amdgpu.ids | tee $O/fft_sizes.log
fs() { echo "== $*"; env "$@" timeout 120 python tools/fft_sizes.py 16 2>&1 | grep -v amdgpu.ids; }
( fs SDRPP_GPU_FFT_P2_GRID=2048
  fs SDRPP_GPU_FFT_P1_C32=1
  fs SDRPP_GPU_FFT_P1_GRID=0
  fs SDRPP_GPU_FFT_P1_GRID=768
  fs SDRPP_GPU_FFT_P1_GRID=1536 ) | tee $O/fft16_sweeps.log
timeout 200 python tools/tick_rate.py 3 1000000 50000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg3.log
timeout 200 python tools/tick_rate.py 2 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg2.log
echo "== cfg4 ssb-exact"
timeout 400 python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline > $O/bench_cfg4_ssb_exact.json 2> $O/bench_cfg4_ssb_exact.err
python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact.json")); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
echo "== cfg2 bench"
timeout 400 python bench.py --cfg 2 --no-by-push --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python - <<PY
import json
d=json.load(open("$O/bench_cfg2.json")); print(d["value"], d["ms_per_step"], d["roofline"]); print(json.dumps(d.get("ceiling"))[:600])
PY
