#!/bin/bash
# round 3, session 5, call 2: reference rotator with the chain wavefront as the only loader (two role loops), FFT lean paths (dense-frame loads,
# 32-bit offsets, branch-free log2 main path, unrolled pass-2 write-out)
set -u
O=gpurun_out/r03r
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 300 python tools/fft_sizes.py 2>&1 | grep -v amdgpu.ids | tee $O/fft_sizes.log
timeout 200 python tools/tick_rate.py 3 1000000 50000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg3.log
timeout 200 python tools/tick_rate.py 2 1000000 2>&1 | grep -v amdgpu.ids | tee $O/tick_rate_cfg2.log
echo "== cfg4 ssb-exact"
timeout 400 python bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline > $O/bench_cfg4_ssb_exact.json 2> $O/bench_cfg4_ssb_exact.err
python - <<PY
import json
d=json.load(open("$O/bench_cfg4_ssb_exact.json")); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_ssb -o t -- python $R/bench.py --cfg 4 --nco ssb-exact --mode ordinary --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline > $R/$O/trace_ssb.log 2>&1
cd $R
T=$(find $O/trace_ssb -name "*kernel_stats.csv" | head -1)
[ -n "$T" ] && head -14 "$T" | cut -c1-220
find $O -name "*.db" -size +8M -delete
