#!/bin/bash
# round 3, session 5, call 6: the phase chain inside the product kernel's four-wavefront structure (probe variants G-K)
set -u
O=gpurun_out/r03v
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/chain_latency_probe tools/probe/chain_latency_probe.hip 2>/dev/null
/tmp/chain_latency_probe 2>&1 | grep -v amdgpu.ids | tee $O/chain_latency_probe.log
