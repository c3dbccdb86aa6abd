run() { python bench.py --steps 200 --warmup 10 --no-others --no-by-push --no-cpu-baseline --no-self-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1:', d['value'], 'MS/s', d['ms_per_step'], 'ms/step, tick', d['roofline']['avg_launch_ms'], 'ms, frac', d['roofline']['frac'], d['pipeline']['roles'][:4])"; }
export SDRPP_GPU_TICK_FCM16W=1
export SDRPP_GPU_TICK_FCM16W_BLOCKS=1024
SDRPP_GPU_TICK_SET2=0 run fcm16w_1024_three_waves
SDRPP_GPU_TICK_SET2=1 run fcm16w_1024_four_waves
export SDRPP_GPU_TICK_FCM16W_BLOCKS=768
SDRPP_GPU_TICK_SET2=0 run fcm16w_768_three_waves
SDRPP_GPU_TICK_SET2=1 run fcm16w_768_four_waves
SDRPP_GPU_TICK_SET2=1 SDRPP_GPU_TICK_TOEP_BLOCKS=384 run fcm16w_768_four_waves_toep384
SDRPP_GPU_TICK_SET2=1 SDRPP_GPU_TICK_TOEP_BLOCKS=192 run fcm16w_768_four_waves_toep192
