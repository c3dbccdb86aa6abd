#!/bin/bash
# the round's closing call: GPU suite, smoke, the driver's bench command
cd /root/repo
O=gpurun_out/r06z
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
t0=$(date +%s); timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? wall $(( $(date +%s) - t0 )) s" >> $O/bench_default.err
tail -3 $O/pytest_gpu.log; tail -3 $O/smoke.log; tail -2 $O/bench_default.err; head -c 600 $O/bench_default.json
