#!/bin/bash
# round 6: the whole device suite + the C++ seam legs (launch groups, NUMA node of the GPU)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r06e_pytest_gpu.log 2>&1
tail -5 gpurun_out/r06e_pytest_gpu.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
echo "GPU numa node: $node; cpus $(cat /sys/devices/system/node/node$node/cpulist)" | tee gpurun_out/r06e_seam.log
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
for grp in 1 8; do
  for i in 1 2 3 4 5; do
    taskset -c $cpus oracle/_ref/bench_blocks_ref sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 $grp 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06e_seam.log
  done
done
echo "---- not pinned ----" | tee -a gpurun_out/r06e_seam.log
for i in 1 2 3 4 5; do
  oracle/_ref/bench_blocks_ref sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1 8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06e_seam.log
done
