#!/bin/bash
# round 6: the pipelined C++ worker's time with launch groups, on the GPU's NUMA node, 8 / 16 hand-over helpers
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06f_seam_prof.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
{
echo "GPU numa node $node cpus $cpus"
for h in 8 16; do
  g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -DSDRPP_GPU_HELPERS=$h -o /tmp/bench_blocks_prof$h tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
  for grp in 1 8; do
    echo "---- helpers $h, blocks per launch $grp ----"
    for i in 1 2 3 4; do
      taskset -c $cpus /tmp/bench_blocks_prof$h sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 $grp 2>&1 | grep -v "amdgpu.ids\|passes 0"
    done
  done
done
echo "---- physical cores only (0-63 of node 0) helpers 16 group 8 ----"
first=$(echo $cpus | cut -d, -f1)
for i in 1 2 3 4; do
  taskset -c $first /tmp/bench_blocks_prof16 sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 8 2>&1 | grep -v "amdgpu.ids\|passes 0"
done
} > $out 2>&1
cat $out
