#!/bin/bash
# round 5, GPU call 6: vectorised carries on the device (suite subset + rates), grid-rule sweep with the faster front end
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_pipelined.py tests/test_parity_fft.py tests/test_bench_geometry_gpu.py tests/test_full_configs_gpu.py -m gpu -q 2>&1 | tail -30 ) > gpurun_out/r05f_pytest_gpu_subset.log
tail -3 gpurun_out/r05f_pytest_gpu_subset.log
( timeout 400 python tools/ab_tick.py --cfg 4 --push 1000000 307200 --rounds 2 new=libsdrpp_gpu.so 2>&1 ) > gpurun_out/r05f_ab_cfg4.log
grep -A3 summary gpurun_out/r05f_ab_cfg4.log
( timeout 900 python tools/ab_tick.py --cfg 3 --push 1000000 --rounds 2 new=libsdrpp_gpu.so \
   w512=libsdrpp_gpu.so,SDRPP_GPU_TICK_FCM_WAVES=512 w640=libsdrpp_gpu.so,SDRPP_GPU_TICK_FCM_WAVES=640 w1024=libsdrpp_gpu.so,SDRPP_GPU_TICK_FCM_WAVES=1024 \
   t192=libsdrpp_gpu.so,SDRPP_GPU_TICK_TOEP_BLOCKS=192 t320=libsdrpp_gpu.so,SDRPP_GPU_TICK_TOEP_BLOCKS=320 t384=libsdrpp_gpu.so,SDRPP_GPU_TICK_TOEP_BLOCKS=384 \
   w1024t320=libsdrpp_gpu.so,SDRPP_GPU_TICK_FCM_WAVES=1024,SDRPP_GPU_TICK_TOEP_BLOCKS=320 2>&1 ) > gpurun_out/r05f_grid_sweep_cfg3.log
grep -A10 summary gpurun_out/r05f_grid_sweep_cfg3.log
( timeout 300 python tools/ab_tick.py --cfg 3 --push 50000 --rounds 2 new=libsdrpp_gpu.so 2>&1 ) > gpurun_out/r05f_ab_cfg3_sr200.log
grep -A2 summary gpurun_out/r05f_ab_cfg3_sr200.log
