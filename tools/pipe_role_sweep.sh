# pipelined mode: FM back ends as one role of the tick (SDRPP_GPU_TICK_PIPE=1, default) against four Toeplitz roles on four ticks (=0)
run() { python bench.py --steps $2 --warmup 10 --push ${3:-1000000} --no-others --no-by-push --no-cpu-baseline --no-self-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 steps $2 push ${3:-1000000}:', d['value'], 'MS/s', d['ms_per_step'], 'ms/step, tick', d['roofline']['avg_launch_ms'], 'ms, frac', d['roofline']['frac'], 'depth', d['pipeline']['depth_levels'], d['pipeline']['roles'])"; }
export SDRPP_GPU_TICK_PIPE=0; run four_roles 200; run four_roles 20; run four_roles 400 50000
export SDRPP_GPU_TICK_PIPE=1; run pipe_role 200; run pipe_role 20; run pipe_role 400 50000
for b in 128 192 384 512; do export SDRPP_GPU_TICK_PIPE_BLOCKS=$b; run pipe_blocks_$b 200; done
