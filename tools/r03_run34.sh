#!/bin/bash
# round 3, session 5: the ratio-32 front end in its small-block shape (16 x 16 x 4, a workgroup per tile): device bit-identity, sr/200 tick rate with and without
set -u
O=gpurun_out/r03zl
mkdir -p $O
timeout 600 python -m pytest tests/test_parity_vfo.py -m gpu -x -q -k "small_block" 2>&1 | tail -1
for rep in 1 2; do
for t in 0 128 512; do
  echo "== SDRPP_GPU_FCM16_MAX_TILES=$t"
  SDRPP_GPU_FCM16_MAX_TILES=$t timeout 200 python tools/tick_rate.py 3 50000 100000 2>&1 | grep -v amdgpu.ids | cut -c1-260
done
done 2>&1 | tee $O/tick_rate_small_block_shape.log
SDRPP_GPU_FCM16_MAX_TILES=128 timeout 200 python tools/tick_trace_run.py 3 50000 300 $O/tt.bin 2>&1 | grep -v amdgpu.ids
timeout 100 python tools/tick_trace.py $O/tt.bin 20 2>/dev/null > $O/tick_timeline_cfg3_B50000_fcm16.txt
rm -f $O/tt.bin
head -12 $O/tick_timeline_cfg3_B50000_fcm16.txt
