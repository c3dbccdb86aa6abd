#!/bin/bash
# round 6: the same seam with the cores kept out of deep idle states (PM QoS /dev/cpu_dma_latency)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06g_seam_qos.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
{
echo "GPU numa node $node cpus $cpus; idle states: $(cat /sys/devices/system/cpu/cpu0/cpuidle/state*/name 2>/dev/null | tr '\n' ' ') latencies $(cat /sys/devices/system/cpu/cpu0/cpuidle/state*/latency 2>/dev/null | tr '\n' ' ')"
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -o /tmp/bench_blocks_prof tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
for q in none 0; do
  echo "---- cpu_dma_latency $q, pinned ----"
  for i in 1 2 3 4 5; do
    if [ $q = none ]; then taskset -c $cpus /tmp/bench_blocks_prof sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 1 2>&1 | grep -v "amdgpu.ids\|passes 0"
    else SDRPP_BENCH_CPU_DMA_LATENCY=$q taskset -c $cpus /tmp/bench_blocks_prof sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 1 2>&1 | grep -v "amdgpu.ids\|passes 0"; fi
  done
done
echo "---- cpu_dma_latency 0, NOT pinned ----"
for i in 1 2 3; do
  SDRPP_BENCH_CPU_DMA_LATENCY=0 /tmp/bench_blocks_prof sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 1 2>&1 | grep -v "amdgpu.ids\|passes 0"
done
} > $out 2>&1
cat $out
