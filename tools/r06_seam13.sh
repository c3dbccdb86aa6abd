#!/bin/bash
# round 6: hand-over helpers with STATIC job assignment (one cache line per helper instead of one claim word polled by all): 8 / 12 / 16 / 24 helpers
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06x_seam_static_helpers.log
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
for h in 8 12 16 24; do
for g in 1 8; do
echo "---- reference stream.h build, helpers $h, blocks per launch $g ----"
for i in 1 2 3; do SDRPP_GPU_HELPERS=$h taskset -c $cpus oracle/_ref/bench_blocks_ref $A $g 2>&1 | grep -v "amdgpu.ids" | sed 's/.*"msps": \([0-9.]*\).*fill": \([0-9.]*\), "source_us_per_block_swap": \([0-9.]*\).*/   msps \1 fill \2 swap \3/'; done
done
echo "---- prof build, helpers $h, blocks per launch 8 ----"
SDRPP_GPU_HELPERS=$h taskset -c $cpus /tmp/bench_blocks_prof $A 8 2>&1 | grep "blocks prof" | cut -c1-220
done
} > $out 2>&1
cat $out
