#!/bin/bash
# Counters of the FFT passes in one ordinary pass over 2^24 samples (cfg 2's ceiling run): instruction mix, waits, LDS, traffic.   usage: bash tools/pmc_fft_passes.sh [tag]
set -u
TAG=${1:-r04fft}
O=gpurun_out/$TAG
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --cfg 2 --mode ordinary --push 16777216 --ref-block 0 --steps 12 --warmup 3 --no-others --no-by-push --no-cpu-baseline --no-self-check"
DBS=""; i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/g$i -o p -- $BENCH > $R/$O/g$i.log 2>&1
    db=$(find $R/$O/g$i -name "*.db" | head -1)
    [ -n "$db" ] && DBS="$DBS $db" || { echo "group $i ($ctr): no database"; tail -3 $R/$O/g$i.log; }
done
cd $R
first=$(echo $DBS | cut -d' ' -f1)
python tools/rocpd_summary.py $first --pmc $DBS --out $O/${TAG}_fft_passes_2p24.md --title "FFT passes, one ordinary pass over 2^24 samples (cfg 2 ceiling)" 2>&1 | tail -1
cat $O/${TAG}_fft_passes_2p24.md | grep -v "^$" | head -70
find $O -name "*.db" -size +4M -delete
