#!/bin/bash
cd /root/repo
O=gpurun_out/r06z
mkdir -p $O
t0=$(date +%s.%N)
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/bench_default.err
t1=$(date +%s.%N)
echo "bench wall $(echo "$t1 - $t0" | bc) s" >> $O/bench_default.err
tail -3 $O/bench_default.err; head -c 400 $O/bench_default.json
