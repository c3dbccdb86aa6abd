#!/bin/bash
# kernel trace of the headline command for the FINAL library of round 5 (the PMC passes of r05zm stay valid: the default roles did not change)
TAG=${1:-r05zv}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
BENCH="python $R/bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o t -- $BENCH --steps 200 > $R/$O/trace.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) --out $O/${TAG}_cfg3_pipelined_1M_kernel_trace.md \
    --title "round 5, FINAL library (reference-rotator and small-bank roles in the tick kernel): headline workload, python bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check --steps 200" \
    --meta push=1000000 cfg=3 nvfo=32 mode=pipelined 2>&1 | tail -2
head -12 $O/${TAG}_cfg3_pipelined_1M_kernel_trace.md
tail -1 $O/trace.log | cut -c1-600
find $O -name "*.db" -size +8M -delete
