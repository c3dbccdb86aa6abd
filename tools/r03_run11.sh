#!/bin/bash
set -u
O=gpurun_out/r03k
mkdir -p $O
timeout 600 python -m pytest tests/test_parity_fft.py tests/test_pipelined.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_parity_vfo.py -m gpu -x -q -k "failed_push" 2>&1 | tail -2
timeout 300 python tools/fft_sizes.py 2>&1 | grep -v amdgpu.ids | tee $O/fft_sizes.log
timeout 300 python tools/tick_rate.py 4 307200 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg4.log
timeout 300 python tools/tick_rate.py 2 1000000 200000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg2.log
