#!/bin/bash
# the N > 1 code path on the one GPU there is: a process group of one rank (RCCL gather of the lines in the timed loop), then torchrun with one rank
cd /root/repo
O=gpurun_out/r06z
mkdir -p $O
SDRPP_BENCH_FORCE_RCCL=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-others --no-by-push --no-cpu-baseline > $O/bench_forced_rccl.json 2> $O/bench_forced_rccl.err; echo "rc $?" >> $O/bench_forced_rccl.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-others --no-by-push --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "rc $?" >> $O/bench_torchrun1.err
tail -2 $O/bench_forced_rccl.err; head -c 300 $O/bench_forced_rccl.json; echo; tail -2 $O/bench_torchrun1.err; head -c 300 $O/bench_torchrun1.json
