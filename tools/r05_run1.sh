#!/bin/bash
# round 5, GPU call 1: parity of the pair-per-half front ends on the device + A/B against the round-4 library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05a_pytest_gpu.log
tail -3 gpurun_out/r05a_pytest_gpu.log
( timeout 600 python tools/ab_tick.py --cfg 3 --push 1000000 50000 --rounds 3 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so earlyiq=libsdrpp_gpu_earlyiq.so 2>&1 ) > gpurun_out/r05a_ab_cfg3.log
tail -8 gpurun_out/r05a_ab_cfg3.log
( timeout 600 python tools/ab_tick.py --cfg 4 --push 1000000 307200 --rounds 2 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so 2>&1 ) > gpurun_out/r05a_ab_cfg4.log
tail -6 gpurun_out/r05a_ab_cfg4.log
( timeout 300 python tools/tick_trace_run.py 3 1000000 40 /tmp/tt.bin && python tools/tick_trace.py /tmp/tt.bin 2>&1 | head -40 ) > gpurun_out/r05a_tick_timeline_cfg3_B1000000.txt 2>&1
head -24 gpurun_out/r05a_tick_timeline_cfg3_B1000000.txt
