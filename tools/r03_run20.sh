#!/bin/bash
# round 3, session 5, call 5: rotator with the phase writes one round late, probe variant F, packed log2 polynomial against the scalar build
set -u
O=gpurun_out/r03u
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/chain_latency_probe tools/probe/chain_latency_probe.hip 2>/dev/null
/tmp/chain_latency_probe 2>&1 | grep -v amdgpu.ids | tee $O/chain_latency_probe.log
timeout 600 python -m pytest tests/test_parity_vfo.py tests/test_parity_fft.py tests/test_golden.py -m gpu -x -q -k "rotator or nco_mode or fft or golden or extreme" 2>&1 | tail -2
B="python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline"
echo "== cfg4 ssb-exact"
timeout 300 $B > $O/bench_cfg4_ssb_exact.json 2> $O/err.log
python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact.json")); print(d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items() if v > 0.2})
PY
for rep in 1 2; do
for lib in libsdrpp_gpu.so libsdrpp_gpu_log2scalar.so; do
  echo "== fft sizes 16 12 20, $lib"
  SDRPP_TOOL_LIB=$lib timeout 200 python tools/fft_sizes.py 16 12 20 2>&1 | grep -v amdgpu.ids | tee -a $O/fft_sizes_$lib.log
done
done
