#!/bin/bash
# round 6: where the C++ IQFrontEnd worker's time goes at sr/200 in pipelined bypass mode (diagnostic build) + the seam figure against the reference's own stream.h
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06d_seam_prof.log
{
echo "nproc $(nproc)"; lscpu | grep -E "Model name|Socket|Core|Thread|MHz|L2|L3" 
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -o /tmp/bench_blocks_prof tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
for i in 1 2 3; do
  /tmp/bench_blocks_prof sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 2>&1 | grep -v amdgpu.ids
done
echo "---- against the reference's stream.h (oracle/_ref/bench_blocks_ref) ----"
for i in 1 2 3 4 5; do
  oracle/_ref/bench_blocks_ref sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 2>&1 | grep -v amdgpu.ids
done
echo "---- a memcpy of 400 KB, one thread (what SpeedTester::writeWorker pays per block) ----"
python - <<'PY'
import numpy as np, time
src=[np.ones(100000,np.float32) for _ in range(4)]; dst=np.empty(2000000,np.float32)
t0=time.perf_counter()
n=20000
for i in range(n): dst[:100000]=src[i&3]
dt=time.perf_counter()-t0
print("%.1f us per 400 KB copy (numpy slice assignment)"%(dt/n*1e6))
PY
} > $out 2>&1
cat $out
