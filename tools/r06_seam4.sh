#!/bin/bash
# round 6: the seam with the worker looking for its next block before it sleeps (IQFrontEnd::setSpinWait, default 100 us)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06h_seam_spin.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
{
echo "GPU numa node $node cpus $cpus"
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -o /tmp/bench_blocks_prof tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
echo "---- prof build, pinned ----"
for i in 1 2 3; do taskset -c $cpus /tmp/bench_blocks_prof sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 1 2>&1 | grep -v "amdgpu.ids\|passes 0"; done
echo "---- reference stream.h build, pinned ----"
for i in 1 2 3 4 5; do taskset -c $cpus oracle/_ref/bench_blocks_ref sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 1 2>&1 | grep -v "amdgpu.ids"; done
echo "---- reference stream.h build, pinned, blocks per launch 8 ----"
for i in 1 2 3 4 5; do taskset -c $cpus oracle/_ref/bench_blocks_ref sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 8 2>&1 | grep -v "amdgpu.ids"; done
echo "---- reference stream.h build, NOT pinned ----"
for i in 1 2 3 4 5; do oracle/_ref/bench_blocks_ref sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 1 2>&1 | grep -v "amdgpu.ids"; done
} > $out 2>&1
cat $out
