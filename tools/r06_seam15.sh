#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06x_seam_kick_ema.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
A="sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1"
{
for g in 1 4 8 16 1 8; do
echo "---- reference stream.h build, pinned, blocks per launch $g ----"
for i in 1 2 3; do taskset -c $cpus oracle/_ref/bench_blocks_ref $A $g 2>&1 | grep -v "amdgpu.ids" | sed 's/.*"msps": \([0-9.]*\).*fill": \([0-9.]*\), "source_us_per_block_swap": \([0-9.]*\).*/   msps \1 fill \2 swap \3/'; done
done
echo "---- fixed kick of 20 us (SDRPP_GPU_KICK_FIXED=1), blocks per launch 8 ----"
for i in 1 2 3; do SDRPP_GPU_KICK_FIXED=1 taskset -c $cpus oracle/_ref/bench_blocks_ref $A 8 2>&1 | grep -v "amdgpu.ids" | sed 's/.*"msps": \([0-9.]*\).*fill": \([0-9.]*\), "source_us_per_block_swap": \([0-9.]*\).*/   msps \1 fill \2 swap \3/'; done
} > $out 2>&1
cat $out
