#!/bin/bash
# Round-2 GPU call 4: LDS-DMA / wide-store variants of the Toeplitz kernel: parity on the device, per-family times, whole-bench A/B.
set -u
O=gpurun_out/r02c4
mkdir -p $O
( timeout 600 python -m pytest tests/test_parity_vfo.py tests/test_host_cpp.py -x -q -m gpu -k "variants or bank or deferred" ) 2>&1 | tail -5 | tee $O/log.txt
for sw in "SDRPP_GPU_TOEP_DMA=0" "SDRPP_GPU_TOEP_DMA=1" "SDRPP_GPU_TOEP_DMA=2" "SDRPP_GPU_WIDE_STORE=1" "SDRPP_GPU_TOEP_DMA=0"; do
    echo "-- VFO bank alone, $sw" | tee -a $O/log.txt
    env $sw timeout 120 python tools/vfo_only_time.py 16777216 32 10 2>&1 | tail -1 | tee -a $O/log.txt
done
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernel_ms_per_step"])'
for sw in "SDRPP_GPU_TOEP_DMA=0" "SDRPP_GPU_TOEP_DMA=1" "SDRPP_GPU_TOEP_DMA=2" "SDRPP_GPU_WIDE_STORE=1" "SDRPP_GPU_TOEP_DMA=0"; do
    echo "-- bench, $sw" | tee -a $O/log.txt
    env $sw timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-by-push 2>/dev/null | python -c "$P" 2>&1 | tee -a $O/log.txt
done
