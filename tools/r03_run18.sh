#!/bin/bash
# round 3, session 5, call 3: latency probe of the rotator's phase chain; tick timelines of the current state
set -u
O=gpurun_out/r03s
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/chain_latency_probe tools/probe/chain_latency_probe.hip 2>/dev/null
/tmp/chain_latency_probe 2>&1 | grep -v amdgpu.ids | tee $O/chain_latency_probe.log
for spec in "3 1000000 80" "3 50000 300" "4 1000000 40"; do
  set -- $spec
  timeout 200 python tools/tick_trace_run.py $1 $2 $3 $O/tt.bin 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/tick_trace.py $O/tt.bin 20 2>/dev/null > $O/tick_timeline_cfg$1_B$2.txt
  rm -f $O/tt.bin
  head -24 $O/tick_timeline_cfg$1_B$2.txt
done
