#!/bin/bash
# where the C++ IQFrontEnd worker's time goes at the reference block size (diagnostic build of tests/host_cpp/bench_blocks.cpp)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -o /tmp/bench_blocks_prof $R/tests/host_cpp/bench_blocks.cpp -I$R/tests/host_cpp/standalone -L$R/sdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
for buffered in 0 1; do
    /tmp/bench_blocks_prof $R/sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 3 $buffered 2>&1 | grep -v amdgpu.ids
done
