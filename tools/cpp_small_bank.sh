#!/bin/bash
# the C++ front end (IQFrontEnd::run, real dsp::stream hand-overs) on a session-sized bank: cfg 1 (2.4 MS/s, 4096-point FFT) with 1 and 4 radios at
# sr/200 blocks, one pass per block against setPipelining — pipelined since the vector-unit front ends are roles of the tick (round 5)
mkdir -p gpurun_out
T=${1:-r05zy}
R=${GRAFT_REPO_ROOT:-$PWD}
g++ -std=c++17 -O2 -w -o /tmp/bench_blocks $R/tests/host_cpp/bench_blocks.cpp -I$R/tests/host_cpp/standalone -L$R/sdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
for nv in 1 4; do for pipe in 0 1; do
  timeout 60 /tmp/bench_blocks $R/sdrplusplus_amd/data/decim_plans.bin 2400000 12000 4096 $nv 2 0 $pipe 2>&1 | grep '^{' | tail -1
done; done | tee gpurun_out/${T}_cpp_small_bank.log
