#!/bin/bash
# round 6: per-workgroup timelines of the tick with launch groups (make ticktrace build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
make -C sdrplusplus_amd/csrc -s ticktrace 2>&1 | grep -i error
for spec in "3 1000000 80 1" "3 1000000 160 4" "3 50000 400 8" "4 1000000 80 4"; do
  set -- $spec
  TICK_GROUP=$4 timeout 300 python tools/tick_trace_run.py $1 $2 $3 gpurun_out/tt.bin 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/tick_trace.py gpurun_out/tt.bin 12 2>/dev/null > gpurun_out/r06q_tick_timeline_cfg$1_B$2_group$4.txt
  rm -f gpurun_out/tt.bin
  head -40 gpurun_out/r06q_tick_timeline_cfg$1_B$2_group$4.txt
done
