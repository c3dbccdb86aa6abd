#!/bin/bash
# round 5, GPU call 4: device suite with the new boundary tests; what the tick costs without its FFT roles; bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -80 ) > gpurun_out/r05d_pytest_gpu.log
tail -4 gpurun_out/r05d_pytest_gpu.log
( timeout 300 python tools/ab_tick.py --cfg 3 --push 1000000 --rounds 2 with_fft=libsdrpp_gpu.so 2>&1; timeout 300 python tools/ab_tick.py --cfg 3 --push 1000000 --rounds 2 --no-fft vfo_only=libsdrpp_gpu.so 2>&1; timeout 300 python tools/ab_tick.py --cfg 2 --push 1000000 --rounds 2 fft_only=libsdrpp_gpu.so 2>&1 ) > gpurun_out/r05d_tick_parts.log
grep -A3 summary gpurun_out/r05d_tick_parts.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r05d_bench.err | tail -1 ) > gpurun_out/r05d_bench_default.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05d_bench_default.json'))
print('value', d['value'], 'steady', d.get('steady_state'), 'frac', d['roofline']['frac'], 'avg_launch_ms', d['roofline'].get('avg_launch_ms'))
oc=d.get('other_configs',{})
for k,v in oc.items():
    if isinstance(v,dict) and 'pipelined_stream_cap' in v: print(k, v['pipelined_stream_cap']['value'], v.get('ceiling_2p24_ordinary',{}).get('value'))
bp=d.get('by_push',{})
for k,v in bp.items():
    print(k, json.dumps(v)[:300])
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
