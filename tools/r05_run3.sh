#!/bin/bash
# round 5, GPU call 3: the whole device suite with the hazard fix + the A/B again (the first A/B timed a wrong kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > gpurun_out/r05c_pytest_gpu.log
tail -4 gpurun_out/r05c_pytest_gpu.log
( timeout 600 python tools/ab_tick.py --cfg 3 --push 1000000 50000 --rounds 3 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so earlyiq=libsdrpp_gpu_earlyiq.so 2>&1 ) > gpurun_out/r05c_ab_cfg3.log
tail -7 gpurun_out/r05c_ab_cfg3.log
( timeout 600 python tools/ab_tick.py --cfg 4 --push 1000000 307200 --rounds 2 base=libsdrpp_gpu_base.so new=libsdrpp_gpu.so 2>&1 ) > gpurun_out/r05c_ab_cfg4.log
tail -5 gpurun_out/r05c_ab_cfg4.log
