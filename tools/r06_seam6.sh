#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06v_seam_evict.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
A="sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 1"
{
for e in 0 1 0 1; do
  echo "---- reference stream.h build, pinned, SDRPP_GPU_STAGE_EVICT=$e ----"
  for i in 1 2 3; do SDRPP_GPU_STAGE_EVICT=$e taskset -c $cpus oracle/_ref/bench_blocks_ref $A 2>&1 | grep -v "amdgpu.ids" | cut -c100-400; done
done
} > $out 2>&1
cat $out
