#!/bin/bash
# round 5, GPU call 7: the 16-row shape of the long first stage on the device (parity + rates + timeline)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_parity_vfo.py tests/test_bench_geometry_gpu.py tests/test_full_configs_gpu.py tests/test_pipelined.py -m gpu -q -k "long_first or cfg4 or mixed" 2>&1 | tail -30 ) > gpurun_out/r05g_pytest_gpu_cfg4.log
tail -3 gpurun_out/r05g_pytest_gpu_cfg4.log
( timeout 400 python tools/ab_tick.py --cfg 4 --push 1000000 307200 --rounds 2 new=libsdrpp_gpu.so 2>&1 ) > gpurun_out/r05g_ab_cfg4.log
grep -A3 summary gpurun_out/r05g_ab_cfg4.log
timeout 300 python tools/tick_trace_run.py 4 1000000 50 /tmp/tt.bin 2>&1 | grep -v amdgpu.ids
timeout 100 python tools/tick_trace.py /tmp/tt.bin 20 2>/dev/null > gpurun_out/r05g_tick_timeline_cfg4_B1000000.txt
grep -v "in 1 ticks\|in 2 ticks\|in 3 ticks" gpurun_out/r05g_tick_timeline_cfg4_B1000000.txt | head -14
