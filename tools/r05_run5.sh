#!/bin/bash
# round 5, GPU call 5: AGPR / VGPR accumulators under streaming loads (probe), the NCO window test, tick timelines of cfg 4 and cfg 3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 tools/probe/mfma_mem_overlap_probe 2>&1 ) > gpurun_out/r05e_mfma_mem_overlap_agpr_vgpr.log
cat gpurun_out/r05e_mfma_mem_overlap_agpr_vgpr.log
( timeout 900 python -m pytest tests/test_bench_geometry_gpu.py -m gpu -q -k "nco_validity" -s 2>&1 | tail -30 ) > gpurun_out/r05e_nco_window.log
tail -6 gpurun_out/r05e_nco_window.log
for spec in "4 1000000 50" "4 307200 60" "3 1000000 60"; do
  set -- $spec
  timeout 300 python tools/tick_trace_run.py $1 $2 $3 /tmp/tt.bin 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/tick_trace.py /tmp/tt.bin 20 2>/dev/null > gpurun_out/r05e_tick_timeline_cfg$1_B$2.txt
  rm -f /tmp/tt.bin
done
grep -v "in 1 ticks\|in 2 ticks\|in 3 ticks" gpurun_out/r05e_tick_timeline_cfg4_B1000000.txt | head -40
