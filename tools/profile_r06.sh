#!/bin/bash
# Round 6 profile call: rocprofv3 kernel trace + PMC passes (FETCH_SIZE / WRITE_SIZE, one pass each) of the headline workload with launch groups, of cfg 2,
# cfg 4 and the sr/200 blocks; instruction mix of the headline tick.   usage: bash tools/profile_r06.sh [tag]
set -u
TAG=${1:-r06p}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
BENCH="python $R/bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check --regions 1"
cd /tmp
echo "== rocprofv3 kernel trace (headline, driver geometry: --steps 20 --warmup 5, then 200 steady steps in the same process)"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o t -- $BENCH --steps 20 --warmup 5 > $R/$O/trace.log 2>&1
blocks_of() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["pipeline"]["blocks_pushed_by_this_context"])
except Exception:
    print(0)
PY
}
prof() {  # name, bench args [, steps + warmup]
    name=$1; shift; args=$1; shift; sw=${1:---steps 60 --warmup 10}
    for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/pmc_${name}_$ctr -o p -- $BENCH $args $sw > $R/$O/pmc_${name}_$ctr.log 2>&1
    done
}
prof cfg3 ""
prof cfg3_g1 "--group 1"
prof cfg2 "--cfg 2"
prof cfg4 "--cfg 4"
prof cfg3_sr200 "--push 50000 --group 8"
prof cfg3_sr200_g32 "--push 50000 --group 32" "--steps 960 --warmup 96"
MIX=""
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/pmc_mix_$i -o p -- $BENCH --steps 60 --warmup 10 > $R/$O/pmc_mix_$i.log 2>&1
    MIX="$MIX $(find $R/$O/pmc_mix_$i -name '*.db' | head -1)"
done
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db trace) --pmc $(db pmc_cfg3_FETCH_SIZE) $(db pmc_cfg3_WRITE_SIZE) $MIX \
    --out $O/${TAG}_cfg3_pipelined_1M_group4.md --json $O/pmc_traffic_cfg3_push1000000_group4.json \
    --title "round 6: headline workload (cfg 3, pipelined mode, 10^6-sample blocks, up to 4 blocks per launch (adaptive), zoomed lines delivered); kernel trace = python bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check --regions 1 --steps 20 --warmup 5 (20 timed + 200 steady blocks), PMC passes --steps 60 --warmup 10" \
    --meta push=1000000 cfg=3 nvfo=32 mode=pipelined group=4 blocks=$(blocks_of $O/pmc_cfg3_FETCH_SIZE.log) 2>&1 | tail -2
python tools/rocpd_summary.py $(db pmc_cfg3_g1_FETCH_SIZE) --pmc $(db pmc_cfg3_g1_FETCH_SIZE) $(db pmc_cfg3_g1_WRITE_SIZE) --out $O/${TAG}_cfg3_pipelined_1M_group1.md --json $O/pmc_traffic_cfg3_push1000000.json \
    --title "round 6: the same with ONE block per launch (--group 1)" --meta push=1000000 cfg=3 nvfo=32 mode=pipelined group=1 blocks=$(blocks_of $O/pmc_cfg3_g1_FETCH_SIZE.log) 2>&1 | tail -1
for spec in "cfg2 2 1000000 0 4" "cfg4 4 1000000 128 4" "cfg3_sr200 3 50000 32 8" "cfg3_sr200_g32 3 50000 32 32"; do
    set -- $spec
    python tools/rocpd_summary.py $(db pmc_$1_FETCH_SIZE) --pmc $(db pmc_$1_FETCH_SIZE) $(db pmc_$1_WRITE_SIZE) --out $O/${TAG}_$1_pipelined.md --json $O/pmc_traffic_cfg$2_push$3_group$5.json \
        --title "round 6: $1, pipelined mode, $3-sample blocks, up to $5 blocks per launch (kernel durations here are those of the PMC pass)" --meta push=$3 cfg=$2 nvfo=$4 mode=pipelined group=$5 blocks=$(blocks_of $O/pmc_$1_FETCH_SIZE.log) 2>&1 | tail -1
done
head -24 $O/${TAG}_cfg3_pipelined_1M_group4.md
tail -4 $O/${TAG}_cfg3_pipelined_1M_group4.md
grep -h "per block" -A2 $O/${TAG}_*.md | grep "tick launches"
find $O -name "*.db" -delete
find $O -name "*.csv" -size +1M -delete
ls $O
