#!/bin/bash
# round 6, first device call: the grouped-launch tests, the driver's bench command, a sweep of the blocks per launch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pipelined.py -m gpu -x -q -k "grouped" > gpurun_out/r06a_pytest_grouped.log 2>&1
tail -3 gpurun_out/r06a_pytest_grouped.log
for g in 1 2 4 8; do
  timeout 600 python bench.py --steps 20 --warmup 5 --group $g --no-others --no-cpu-baseline --no-by-push > gpurun_out/r06a_bench_group$g.json 2> gpurun_out/r06a_bench_group$g.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06a_bench_group$g.json").read().strip().splitlines()[-1])
    print("group $g: value", d["value"], "regions", d.get("timed_regions", {}).get("Msamples_per_s"), "steady", d.get("steady_state", {}).get("value"), "frac", d["roofline"]["frac"], "avg_launch_ms", d["roofline"]["avg_launch_ms"], d.get("blocks_per_launch", {}).get("first_timed_region"))
except Exception as e:
    print("group $g: failed", e)
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 --group 4 --group-fixed --no-others --no-cpu-baseline --no-by-push > gpurun_out/r06a_bench_group4_fixed.json 2> gpurun_out/r06a_bench_group4_fixed.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06a_bench_group4_fixed.json").read().strip().splitlines()[-1])
print("group 4 fixed: value", d["value"], "regions", d.get("timed_regions", {}).get("Msamples_per_s"), "steady", d.get("steady_state", {}).get("value"), d.get("blocks_per_launch", {}).get("first_timed_region"))
PY
timeout 900 python bench.py --steps 20 --warmup 5 --no-others --no-cpu-baseline > gpurun_out/r06a_bench_bypush.json 2> gpurun_out/r06a_bench_bypush.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06a_bench_bypush.json").read().strip().splitlines()[-1])
for k, v in d.get("by_push", {}).items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v.items() if not isinstance(b, (dict, str))})
PY
