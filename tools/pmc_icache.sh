#!/bin/bash
# Instruction-cache behaviour of the tick kernel (403 KB of code, 64 KB of instruction cache per CU pair).   usage: bash tools/pmc_icache.sh [tag] [bench args]
set -u
TAG=${1:-r04s}; shift || true
O=gpurun_out/$TAG
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check --steps 60 --warmup 10 $*"
i=0
DBS=""
for ctr in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/g$i -o p -- $BENCH > $R/$O/g$i.log 2>&1
    db=$(find $R/$O/g$i -name "*.db" | head -1)
    [ -n "$db" ] && DBS="$DBS $db" || { echo "group $i ($ctr): no database"; tail -3 $R/$O/g$i.log; }
done
cd $R
first=$(echo $DBS | cut -d' ' -f1)
python tools/rocpd_summary.py $first --pmc $DBS --out $O/${TAG}_tick_icache.md --title "instruction cache of the tick ($*)" 2>&1 | tail -2
grep -E "tick_kernel" $O/${TAG}_tick_icache.md | head -40
find $O -name "*.db" -size +4M -delete
