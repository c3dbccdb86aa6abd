#!/bin/bash
# round 6: the source thread's side of the seam — time to fill writeBuf (the SpeedTester's memcpy) and time inside swap(), against the number of copy helpers
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06v_seam_source.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
A="sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 1"
{
echo "GPU numa node $node cpus $cpus"
for h in 8 4 2 1; do
  echo "---- reference stream.h build, pinned, SDRPP_GPU_HELPERS=$h ----"
  for i in 1 2 3; do SDRPP_GPU_HELPERS=$h taskset -c $cpus oracle/_ref/bench_blocks_ref $A 2>&1 | grep -v "amdgpu.ids" | cut -c1-400; done
done
echo "---- no VFOs (FFT only) helpers 8 ----"
for i in 1 2; do taskset -c $cpus oracle/_ref/bench_blocks_ref sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 0 2 0 1 1 2>&1 | grep -v "amdgpu.ids" | cut -c1-400; done
} > $out 2>&1
cat $out
