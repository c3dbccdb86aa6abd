#!/bin/bash
# round 3, session 4, call 2: FFT tile walks (grids / columns per workgroup), tick rates with walking FFT roles, the reworked reference-rotator kernel, the C++ worker staging first
set -u
O=gpurun_out/r03o
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
( time timeout 900 python -m pytest tests/test_parity_fft.py tests/test_golden.py tests/test_pipelined.py tests/test_host_cpp.py -m gpu -x -q ) > $O/pytest_gpu_subset.log 2>&1
tail -4 $O/pytest_gpu_subset.log
timeout 600 python -m pytest tests/test_parity_vfo.py tests/test_full_configs_gpu.py -m gpu -x -q -k "rotator or nco_mode" 2>&1 | tail -2
fs() { echo "== $*"; env "$@" timeout 120 python tools/fft_sizes.py 16 2>&1 | grep -v amdgpu.ids; }
( fs X=0
  fs SDRPP_GPU_FFT_P1_GRID=0 SDRPP_GPU_FFT_P2_GRID=0
  fs SDRPP_GPU_FFT_P1_GRID=256
  fs SDRPP_GPU_FFT_P1_GRID=1024
  fs SDRPP_GPU_FFT_P1_C16=1 SDRPP_GPU_FFT_P1_GRID=0
  fs SDRPP_GPU_FFT_P1_C16=1 SDRPP_GPU_FFT_P1_GRID=512
  fs SDRPP_GPU_FFT_P1_C16=1 SDRPP_GPU_FFT_P1_GRID=1024
  fs SDRPP_GPU_FFT_P1_C16=1 SDRPP_GPU_FFT_P1_GRID=2048
  fs SDRPP_GPU_FFT_P2_GRID=512
  fs SDRPP_GPU_FFT_P2_GRID=2048 ) | tee $O/fft16_sweeps.log
timeout 300 python tools/fft_sizes.py 2>&1 | grep -v amdgpu.ids | tee $O/fft_sizes.log
for g in 0 64 128 256; do echo "== SDRPP_GPU_FFT_TICK_GRID=$g"; SDRPP_GPU_FFT_TICK_GRID=$g timeout 200 python tools/tick_rate.py 3 1000000 2>&1 | grep -v amdgpu.ids; done | tee $O/tick_rate_fft_tick_grid.log
timeout 200 python tools/tick_rate.py 3 50000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_50k.log
for g in 0 128 512; do echo "== cfg2 SDRPP_GPU_FFT_TICK_GRID=$g"; SDRPP_GPU_FFT_TICK_GRID=$g timeout 200 python tools/tick_rate.py 2 1000000 2>&1 | grep -v amdgpu.ids; done | tee $O/tick_rate_cfg2.log
echo "== cfg4 ssb-exact"
timeout 400 python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline > $O/bench_cfg4_ssb_exact.json 2> $O/bench_cfg4_ssb_exact.err
python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact.json")); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
echo "== C++ worker"
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -o /tmp/bench_blocks_prof $R/tests/host_cpp/bench_blocks.cpp -I$R/tests/host_cpp/standalone -L$R/sdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
g++ -std=c++17 -O2 -w -o /tmp/bench_blocks $R/tests/host_cpp/bench_blocks.cpp -I$R/tests/host_cpp/standalone -L$R/sdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
( for i in 1 2 3 4; do /tmp/bench_blocks $R/sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 3 0 1; done
  /tmp/bench_blocks_prof $R/sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 3 0 1
  /tmp/bench_blocks $R/sdrplusplus_amd/data/decim_plans.bin 10000000 200000 65536 32 3 0 1
  /tmp/bench_blocks $R/sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 3 1 0 ) 2>&1 | grep -v amdgpu.ids | tee $O/cpp_pipelined.log
