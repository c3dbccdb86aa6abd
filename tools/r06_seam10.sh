#!/bin/bash
# round 6: the seam with host-side launch groups (held pushes that do not wait for their copy; a group goes out full or after SDRPP_GPU_KICK_US without a new block)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06w_seam_held.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
A="sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1"
{
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -o /tmp/bench_blocks_prof tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
for g in 1 2 4 8 16; do
echo "---- reference stream.h build, pinned, blocks per launch $g ----"
for i in 1 2 3; do taskset -c $cpus oracle/_ref/bench_blocks_ref $A $g 2>&1 | grep -v "amdgpu.ids" | cut -c60-420; done
done
for g in 1 4; do
echo "---- prof build, blocks per launch $g ----"
for i in 1 2; do SDRPP_GPU_HOSTPROF=1 taskset -c $cpus /tmp/bench_blocks_prof $A $g 2>&1 | grep -v "amdgpu.ids\|passes 0" | cut -c1-260; done
done
timeout 600 python -m pytest tests/test_host_cpp.py -q -m gpu 2>&1 | tail -3
} > $out 2>&1
cat $out
