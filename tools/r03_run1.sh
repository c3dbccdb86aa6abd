#!/bin/bash
# Round 3, first device run of the pipelined mode: parity with the ordinary pass, rates, host-side profile, kernel trace
set -u
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( time timeout 600 python -m pytest tests/test_pipelined.py -m gpu -x -q ) > $O/pytest_pipelined.log 2>&1
tail -6 $O/pytest_pipelined.log
echo "== rates cfg 3"
SDRPP_GPU_HOSTPROF=1 timeout 300 python tools/tick_rate.py 3 > $O/tick_rate_cfg3.log 2>&1
grep -v amdgpu.ids $O/tick_rate_cfg3.log | tail -40
echo "== rocprofv3 kernel trace, B = 50000"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o t -- python $R/tools/tick_rate.py 3 50000 > $R/$O/trace.log 2>&1
cd $R
T=$(find $O/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $T --out $O/r03a_tick_trace_B50000.md --title "round 3: pipelined mode (tick kernel), cfg 3, 50 000-sample blocks (tools/tick_rate.py)" 2>&1 | tail -3
head -30 $O/r03a_tick_trace_B50000.md
find $O -name "*.db" -size +8M -delete
