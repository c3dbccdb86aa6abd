run() { python bench.py --steps $2 --warmup 10 --no-others --no-by-push --no-cpu-baseline --no-self-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('frac_full_launch'))"; }
echo "== default (ext stop event, timing through the launch's own events)"; run ext 200; run ext 200; run ext 20; run ext 20
