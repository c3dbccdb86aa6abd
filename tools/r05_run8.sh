#!/bin/bash
# round 5, GPU call 8: tiles per wavefront of the long first stage (16- and 32-row shapes)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python tools/ab_tick.py --cfg 4 --push 1000000 --rounds 2 \
   n1w1=libsdrpp_gpu.so,SDRPP_GPU_FCL_TPW16=1,SDRPP_GPU_FCL_TPW=1 n2w1=libsdrpp_gpu.so,SDRPP_GPU_FCL_TPW16=2,SDRPP_GPU_FCL_TPW=1 n4w1=libsdrpp_gpu.so,SDRPP_GPU_FCL_TPW16=4,SDRPP_GPU_FCL_TPW=1 \
   n2w2=libsdrpp_gpu.so,SDRPP_GPU_FCL_TPW16=2,SDRPP_GPU_FCL_TPW=2 n4w2=libsdrpp_gpu.so,SDRPP_GPU_FCL_TPW16=4,SDRPP_GPU_FCL_TPW=2 n8w2=libsdrpp_gpu.so,SDRPP_GPU_FCL_TPW16=8,SDRPP_GPU_FCL_TPW=2 2>&1 ) > gpurun_out/r05h_fcl_tiles_per_wave.log
grep -A8 summary gpurun_out/r05h_fcl_tiles_per_wave.log
( timeout 600 python tools/ab_tick.py --cfg 4 --push 307200 --rounds 2 n1w1=libsdrpp_gpu.so,SDRPP_GPU_FCL_TPW16=1,SDRPP_GPU_FCL_TPW=1 n2w1=libsdrpp_gpu.so,SDRPP_GPU_FCL_TPW16=2,SDRPP_GPU_FCL_TPW=1 n2w2=libsdrpp_gpu.so,SDRPP_GPU_FCL_TPW16=2,SDRPP_GPU_FCL_TPW=2 2>&1 ) > gpurun_out/r05h_fcl_tiles_per_wave_307200.log
grep -A4 summary gpurun_out/r05h_fcl_tiles_per_wave_307200.log
