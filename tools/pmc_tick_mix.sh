#!/bin/bash
# Instruction mix / issue accounting of the headline tick (cfg 3, 10^6-sample blocks, pipelined): rocprofv3 PMC passes, one counter group per run.
#   usage: bash tools/pmc_tick_mix.sh [tag] [bench args]
set -u
TAG=${1:-r04r}; shift || true
O=gpurun_out/$TAG
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check --steps 60 --warmup 10 $*"
i=0
DBS=""
for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VALU" \
           "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/g$i -o p -- $BENCH > $R/$O/g$i.log 2>&1
    db=$(find $R/$O/g$i -name "*.db" | head -1)
    [ -n "$db" ] && DBS="$DBS $db" || { echo "group $i ($ctr): no database"; tail -3 $R/$O/g$i.log; }
done
cd $R
first=$(echo $DBS | cut -d' ' -f1)
python tools/rocpd_summary.py $first --pmc $DBS --out $O/${TAG}_tick_instruction_mix.md --title "instruction mix of the tick ($*)" 2>&1 | tail -2
grep -E "tick_kernel" $O/${TAG}_tick_instruction_mix.md | head -40
find $O -name "*.db" -size +4M -delete
