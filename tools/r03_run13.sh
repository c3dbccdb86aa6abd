#!/bin/bash
# round 3, session 4, call 1: GPU tests after the rotator split / zoom lanes / staged_when; tick rates + timeline; cfg 4 with the SSB channels on the
# reference rotator (four-wavefront form against the one-wavefront form); the C++ worker's per-block profile
set -u
O=gpurun_out/r03n
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 300 python tools/tick_rate.py 3 50000 200000 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg3.log
for zg in 1 2 4; do echo "== ZOOM_GROUPS=$zg"; SDRPP_GPU_TICK_ZOOM_GROUPS=$zg timeout 200 python tools/tick_rate.py 3 1000000 2>&1 | grep -v amdgpu.ids; done | tee $O/tick_rate_zoomgroups.log
timeout 200 python tools/tick_trace_run.py 3 1000000 64 /tmp/tt1m.bin && python tools/tick_trace.py /tmp/tt1m.bin > $O/tick_timeline_cfg3_B1000000.txt 2>&1
head -24 $O/tick_timeline_cfg3_B1000000.txt
timeout 200 python tools/tick_trace_run.py 3 50000 300 /tmp/tt50k.bin && python tools/tick_trace.py /tmp/tt50k.bin > $O/tick_timeline_cfg3_B50000.txt 2>&1
head -8 $O/tick_timeline_cfg3_B50000.txt
echo "== cfg4 ssb-exact (four wavefronts)"
timeout 400 python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline > $O/bench_cfg4_ssb_exact.json 2> $O/bench_cfg4_ssb_exact.err
python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact.json")); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
echo "== cfg4 ssb-exact (one wavefront)"
SDRPP_GPU_ROT_EXACT_SINGLE=1 timeout 400 python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline > $O/bench_cfg4_ssb_exact_single.json 2> $O/bench_cfg4_ssb_exact_single.err
python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact_single.json")); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
echo "== C++ worker"
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -o /tmp/bench_blocks_prof $R/tests/host_cpp/bench_blocks.cpp -I$R/tests/host_cpp/standalone -L$R/sdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
g++ -std=c++17 -O2 -w -o /tmp/bench_blocks $R/tests/host_cpp/bench_blocks.cpp -I$R/tests/host_cpp/standalone -L$R/sdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
( for i in 1 2 3; do /tmp/bench_blocks $R/sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 3 0 1; done
  SDRPP_GPU_HOSTPROF=1 /tmp/bench_blocks_prof $R/sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 3 0 1
  /tmp/bench_blocks $R/sdrplusplus_amd/data/decim_plans.bin 10000000 200000 65536 32 3 0 1 ) 2>&1 | grep -v amdgpu.ids | tee $O/cpp_pipelined.log
nproc; lscpu | grep -E "Model name|^CPU\(s\)" 
