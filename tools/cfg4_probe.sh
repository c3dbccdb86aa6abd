run() { python bench.py --cfg 4 --steps $2 --warmup 12 --push ${3:-1000000} --no-others --no-by-push --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 push ${3:-1000000}:', d['value'], 'MS/s', d['ms_per_step'], 'ms/step, tick', d['roofline']['avg_launch_ms'], 'ms, frac', d['roofline']['frac'], 'identical', (d.get('self_check') or {}).get('identical'), d['pipeline']['depth_levels'])"; }
run cfg4 100
run cfg4 100 307200
