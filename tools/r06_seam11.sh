#!/bin/bash
# round 6: hand-over helpers 8 / 12 / 16 / 32 with host-side launch groups of 8 (the hand-over is now the longest part of a block's cycle)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06w_seam_helpers.log
node=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip())
PY
)
cpus=$(cat /sys/devices/system/node/node$node/cpulist)
A="sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1"
{
for h in 8 12 16 32; do
g++ -std=c++17 -O2 -w -DSDRPP_GPU_BLOCKS_PROF -DSDRPP_GPU_HELPERS=$h -o /tmp/bench_blocks_h$h tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$R/sdrplusplus_amd/csrc -lpthread || exit 1
for g in 1 8; do
echo "---- prof build (test-double stream.h), helpers $h, blocks per launch $g ----"
for i in 1 2 3; do taskset -c $cpus /tmp/bench_blocks_h$h $A $g 2>&1 | grep -v "amdgpu.ids\|passes 0" | cut -c1-330 | sed 's/"buffered.*"msps":/"msps":/; s/"audio_frames.*fill"/fill"/'; done
done
done
} > $out 2>&1
cat $out
