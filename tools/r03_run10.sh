#!/bin/bash
set -u
O=gpurun_out/r03j
mkdir -p $O
timeout 600 python -m pytest tests/test_host_cpp.py -m gpu -x -q 2>&1 | tail -3
g++ -std=c++17 -O2 -w -o /tmp/bench_blocks tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$PWD/sdrplusplus_amd/csrc -lpthread
for spec in "0 1" "0 1" "0 1" "0 0" "1 0" "1 0"; do
  set -- $spec
  /tmp/bench_blocks sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 3 $1 $2 2>&1 | tail -1 | tee -a $O/cpp_blocks.log
done
/tmp/bench_blocks sdrplusplus_amd/data/decim_plans.bin 10000000 200000 65536 32 3 0 1 2>&1 | tail -1 | tee -a $O/cpp_blocks.log
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -o /tmp/bench_blocks_prof tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$PWD/sdrplusplus_amd/csrc -lpthread
SDRPP_GPU_HOSTPROF=1 /tmp/bench_blocks_prof sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 3 0 1 2>&1 | tail -25 | tee -a $O/cpp_blocks_prof.log
