#!/bin/bash
# Round 5, final-state GPU call: the default bench line exactly as the driver runs it, bench lines for cfg 2 / cfg 4, rocprofv3 kernel trace + PMC
# passes (FETCH_SIZE / WRITE_SIZE, one pass each) of the headline workload AND of cfg 2, cfg 4 and the sr/200 blocks (pipelined mode), the
# instruction mix of the headline tick, tick timelines.   usage: bash tools/profile_r04.sh [tag]
set -u
TAG=${1:-r05zc}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
tail -c 400 $O/bench_default.json; echo
for c in 2 4; do
    timeout 600 python bench.py --cfg $c --no-others --no-by-push > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err
    tail -c 200 $O/bench_cfg$c.json; echo
done
BENCH="python $R/bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check"
cd /tmp
echo "== rocprofv3 kernel trace (headline)"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o t -- $BENCH --steps 200 > $R/$O/trace.log 2>&1
prof() {  # name, bench args, meta
    name=$1; shift; args=$1; shift
    for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/pmc_${name}_$ctr -o p -- $BENCH $args --steps 60 --warmup 10 > $R/$O/pmc_${name}_$ctr.log 2>&1
    done
}
prof cfg3 ""
prof cfg2 "--cfg 2"
prof cfg4 "--cfg 4"
prof cfg3_sr200 "--push 50000"
MIX=""
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/pmc_mix_$i -o p -- $BENCH --steps 60 --warmup 10 > $R/$O/pmc_mix_$i.log 2>&1
    MIX="$MIX $(find $R/$O/pmc_mix_$i -name '*.db' | head -1)"
done
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db trace) --pmc $(db pmc_cfg3_FETCH_SIZE) $(db pmc_cfg3_WRITE_SIZE) $MIX \
    --out $O/${TAG}_cfg3_pipelined_1M.md --json $O/pmc_traffic_cfg3_push1000000.json \
    --title "round 5, final state: headline workload (cfg 3, pipelined mode, 10^6-sample blocks, zoomed lines delivered), python bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check" \
    --meta push=1000000 cfg=3 nvfo=32 mode=pipelined 2>&1 | tail -2
for spec in "cfg2 2 1000000 0" "cfg4 4 1000000 128" "cfg3_sr200 3 50000 32"; do
    set -- $spec
    python tools/rocpd_summary.py $(db pmc_$1_FETCH_SIZE) --pmc $(db pmc_$1_FETCH_SIZE) $(db pmc_$1_WRITE_SIZE) --out $O/${TAG}_$1_pipelined.md --json $O/pmc_traffic_cfg$2_push$3.json \
        --title "round 5, final state: $1, pipelined mode, $3-sample blocks (kernel durations here are those of the PMC pass)" --meta push=$3 cfg=$2 nvfo=$4 mode=pipelined 2>&1 | tail -1
done
head -30 $O/${TAG}_cfg3_pipelined_1M.md
find $O -name "*.db" -size +8M -delete
for spec in "3 1000000 80" "3 50000 300" "4 1000000 60" "4 307200 80"; do
  set -- $spec
  timeout 200 python tools/tick_trace_run.py $1 $2 $3 $O/tt.bin 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/tick_trace.py $O/tt.bin 20 2>/dev/null > $O/tick_timeline_cfg$1_B$2.txt
  rm -f $O/tt.bin
done
head -8 $O/tick_timeline_cfg3_B1000000.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
ls $O
