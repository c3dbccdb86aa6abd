#!/bin/bash
# Round-2 GPU call 1: baseline at HEAD of round 1 + what round 1 left unmeasured.
set -u
mkdir -p gpurun_out/r02c1
O=gpurun_out/r02c1
export TMPDIR=/tmp
echo "== default bench" | tee $O/log.txt
python bench.py --steps 20 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json | tee -a $O/log.txt
echo "== push-size sweep" | tee -a $O/log.txt
timeout 300 python tools/push_rate.py 50000 1000000 16777216 2>&1 | tee -a $O/log.txt
echo "== wide store A/B (VFO bank alone)" | tee -a $O/log.txt
for sw in "SDRPP_GPU_WIDE_STORE=0" "SDRPP_GPU_WIDE_STORE=1"; do
    echo "-- $sw" | tee -a $O/log.txt
    env $sw timeout 120 python tools/vfo_only_time.py 16777216 32 10 2>&1 | tail -1 | tee -a $O/log.txt
done
echo "== wide-store parity on device" | tee -a $O/log.txt
env SDRPP_GPU_WIDE_STORE=0 python tests/variant_scenario.py /tmp/v0.npz 18 && env SDRPP_GPU_WIDE_STORE=1 python tests/variant_scenario.py /tmp/v1.npz 18 && python - <<PY 2>&1 | tee -a $O/log.txt
import numpy as np
a, b = np.load("/tmp/v0.npz"), np.load("/tmp/v1.npz")
print("WIDE_STORE", "bit-identical" if all(np.array_equal(a[k], b[k]) for k in a.files) else "DIFFERENT")
PY
echo "== kernel trace at B=50000 (device pushes)" | tee -a $O/log.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof50k -o p50k -- python $GRAFT_REPO_ROOT/tools/hosttime.py 50000 > $GRAFT_REPO_ROOT/$O/prof50k.log 2>&1
cd $GRAFT_REPO_ROOT
tail -5 $O/prof50k.log | tee -a $O/log.txt
DB=$(find $O/prof50k -name "*.db" | head -1); python tools/rocpd_summary.py $DB > $O/prof50k_summary.md 2>&1 || true; find $O/prof50k -name "*.db" -size +20M -delete
head -40 $O/prof50k_summary.md | tee -a $O/log.txt
ls -R $O | head -30
