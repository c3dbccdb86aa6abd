#!/bin/bash
# Round-2 GPU call 3: full -m gpu suite, bench lines for cfg 3 (default, with by_push + cpu_baseline), cfg 2, cfg 4, host-side enqueue profile
# at the reference block size, rocprofv3 kernel trace + PMC passes of the default workload.
set -u
O=gpurun_out/r02m
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/ -x -q -m gpu -s --durations=8 ) > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
for c in 3 2 4; do
    echo "== bench cfg $c"
    timeout 600 python bench.py --cfg $c --steps 20 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err
    tail -c 1500 $O/bench_cfg$c.json; echo
done
echo "== host enqueue profile, B = 50000"
SDRPP_GPU_HOSTPROF=1 timeout 120 python tools/hosttime.py 50000 2>&1 | tail -25 | tee $O/hostprof_50k.log
echo "== rocprofv3 kernel trace"
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o t -- python $R/bench.py --steps 10 --no-cpu-baseline --no-by-push > $R/$O/trace.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/pmc_$ctr -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-by-push > $R/$O/pmc_$ctr.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $R/$O/pmc_SQ -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-by-push > $R/$O/pmc_SQ.log 2>&1
cd $R
T=$(find $O/trace -name "*.db" | head -1); F=$(find $O/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*.db" | head -1); Q=$(find $O/pmc_SQ -name "*.db" | head -1)
python tools/rocpd_summary.py $T --pmc $F $W $Q --out $O/r02m_cfg3_16Mi.md --json $O/pmc_traffic.json --title "round 2, final state (pipelined FM back end, doZoom groups, FFT branch at low stream priority), cfg 3, 2^24 samples per step" --meta push=16777216 cfg=3 nvfo=32 2>&1 | tail -3
head -30 $O/r02m_cfg3_16Mi.md
find $O -name "*.db" -size +8M -delete
ls -la $O
echo "== VFO bank alone: pipelined back end on / off"
for m in 1 0 1 0; do SDRPP_TOOL_PIPELINE=$m timeout 120 python tools/vfo_only_time.py 16777216 32 10 2>&1 | tail -1 | sed "s/^/pipeline $m: /" | tee -a $O/vfo_only.log; done
echo "== per-role cycles of the pipelined kernel (prof build)"
SDRPP_GPU_LIB=$PWD/sdrplusplus_amd/csrc/libsdrpp_gpu_prof.so timeout 120 python tools/vfo_only_time.py 16777216 32 10 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/pipe_prof.log
echo "== C++ IQFrontEnd worker profile"
bash tools/blocks_prof.sh 2>&1 | tee $O/blocks_prof.log
