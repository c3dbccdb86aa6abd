#!/bin/bash
# per-phase cycle counters of the Toeplitz kernel (VFO bank alone and with the FFT branch beside it)
set -u
O=gpurun_out/r02c8
mkdir -p $O
make -C sdrplusplus_amd/csrc -s prof
export SDRPP_GPU_LIB=$PWD/sdrplusplus_amd/csrc/libsdrpp_gpu_prof.so
echo "== VFO bank alone, 2^24, 32 VFOs" | tee $O/log.txt
timeout 120 python tools/vfo_only_time.py 16777216 32 10 2>&1 | tail -6 | tee -a $O/log.txt
echo "== VFO bank alone, 2^22" | tee -a $O/log.txt
timeout 120 python tools/vfo_only_time.py 4194304 32 10 2>&1 | tail -6 | tee -a $O/log.txt
