#!/bin/bash
# Per-ROLE instruction mix of the headline workload: the roles of a tick are the bodies of the ordinary kernels, so one ordinary pass per 10^6-sample
# block (one launch per stage) under the PMC counters gives every role its own row — who owns the vector / LDS instructions per matrix instruction.
#   usage: bash tools/pmc_per_role.sh [tag]
set -u
TAG=${1:-r05p}
O=gpurun_out/$TAG
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --mode ordinary --push 1000000 --no-others --no-by-push --no-cpu-baseline --no-self-check --steps 40 --warmup 5"
i=0
DBS=""
for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/g$i -o p -- $BENCH > $R/$O/g$i.log 2>&1
    db=$(find $R/$O/g$i -name "*.db" | head -1)
    [ -n "$db" ] && DBS="$DBS $db" || { echo "group $i ($ctr): no database"; tail -3 $R/$O/g$i.log; }
done
cd $R
first=$(echo $DBS | cut -d' ' -f1)
python tools/rocpd_summary.py $first --pmc $DBS --out $O/${TAG}_per_role_instruction_mix.md --title "per-role instruction mix: headline workload as ONE ORDINARY PASS per 10^6-sample block (python bench.py --mode ordinary --push 1000000)" 2>&1 | tail -2
head -80 $O/${TAG}_per_role_instruction_mix.md
find $O -name "*.db" -size +4M -delete
