#!/bin/bash
# round 6: FFT roles as FEW long-lived workgroups that start with the tick (walking their tiles) instead of ~1000 short ones behind the matrix roles
cd /root/repo
mkdir -p gpurun_out
B="python bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check --regions 3 --steps 20 --warmup 5"
pick() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], '| value', d['value'], '| steady', (d.get('steady_state') or {}).get('value'))
" "$1"; }
(
for rep in 1 2; do
timeout 200 $B 2>/dev/null | pick "default"
for g in 64 128 256; do
  SDRPP_GPU_FFT_TICK_GRID=$g SDRPP_GPU_TICK_P1_WEIGHT=100 SDRPP_GPU_TICK_P2_WEIGHT=100 timeout 200 $B 2>/dev/null | pick "grid $g, FFT roles first"
done
SDRPP_GPU_FFT_TICK_GRID=128 SDRPP_GPU_TICK_P1_WEIGHT=100 timeout 200 $B 2>/dev/null | pick "grid 128, pass 1 first, pass 2 as before"
SDRPP_GPU_FFT_TICK_GRID=128 timeout 200 $B 2>/dev/null | pick "grid 128, order as before"
done
) > gpurun_out/r06x_fft_throttle.log 2>&1
cat gpurun_out/r06x_fft_throttle.log
