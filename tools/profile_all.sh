#!/bin/bash
# Round 3, final-state GPU call: the default bench line exactly as the driver runs it, bench lines for cfg 2 / cfg 4, rocprofv3 kernel trace +
# PMC passes (FETCH_SIZE / WRITE_SIZE / matrix-pipe busy cycles, one pass each) of the headline workload (pipelined mode, 10^6-sample blocks),
# tick timelines.   usage: bash tools/profile_all.sh [tag]
set -u
TAG=${1:-r03zh}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
tail -c 600 $O/bench_default.json; echo
for c in 2 4; do
    timeout 600 python bench.py --cfg $c --no-others > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err
    tail -c 300 $O/bench_cfg$c.json; echo
done
echo "== rocprofv3 kernel trace"
BENCH="python $R/bench.py --no-others --no-by-push --no-cpu-baseline"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o t -- $BENCH --steps 200 > $R/$O/trace.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/pmc_$ctr -o p -- $BENCH --steps 60 --warmup 10 > $R/$O/pmc_$ctr.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $R/$O/pmc_SQ -o p -- $BENCH --steps 60 --warmup 10 > $R/$O/pmc_SQ.log 2>&1
cd $R
T=$(find $O/trace -name "*.db" | head -1); F=$(find $O/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*.db" | head -1); Q=$(find $O/pmc_SQ -name "*.db" | head -1)
python tools/rocpd_summary.py $T --pmc $F $W $Q --out $O/${TAG}_cfg3_pipelined_1M.md --json $O/pmc_traffic_cfg3_push1000000.json \
    --title "round 3, final state: headline workload (cfg 3, pipelined mode, 10^6-sample blocks, zoomed lines delivered), python bench.py --no-others --no-by-push --no-cpu-baseline" \
    --meta push=1000000 cfg=3 nvfo=32 mode=pipelined 2>&1 | tail -3
head -24 $O/${TAG}_cfg3_pipelined_1M.md
grep -h "\"value\"" $O/trace.log | tail -c 400; echo
find $O -name "*.db" -size +8M -delete
for spec in "3 1000000 80" "3 50000 300" "4 307200 80"; do
  set -- $spec
  timeout 200 python tools/tick_trace_run.py $1 $2 $3 $O/tt.bin 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/tick_trace.py $O/tt.bin 20 2>/dev/null > $O/tick_timeline_cfg$1_B$2.txt
  rm -f $O/tt.bin
done
head -8 $O/tick_timeline_cfg3_B1000000.txt
ls $O
