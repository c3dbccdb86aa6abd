#!/bin/bash
# round 6: by_push with launch groups, other group sizes, full default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pipelined.py -m gpu -x -q -k "grouped" > gpurun_out/r06b_pytest_grouped.log 2>&1
tail -2 gpurun_out/r06b_pytest_grouped.log
for g in 5 6; do
  timeout 600 python bench.py --steps 20 --warmup 5 --group $g --group-fixed --no-others --no-cpu-baseline --no-by-push > gpurun_out/r06b_bench_group${g}_fixed.json 2> gpurun_out/r06b_bench_group${g}_fixed.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06b_bench_group${g}_fixed.json").read().strip().splitlines()[-1])
    print("group $g fixed: value", d["value"], "regions", d.get("timed_regions", {}).get("Msamples_per_s"), "steady", d.get("steady_state", {}).get("value"), d.get("blocks_per_launch", {}).get("first_timed_region"))
except Exception as e:
    print("group $g: failed", e)
PY
done
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06b_bench_default.json 2> gpurun_out/r06b_bench_default.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06b_bench_default.json").read().strip().splitlines()[-1])
print("default: value", d["value"], d["timed_regions"]["Msamples_per_s"], "steady", d.get("steady_state"), "frac", d["roofline"]["frac"])
for k, v in d.get("by_push", {}).items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v.items() if not isinstance(b, (dict, str))})
    else:
        print(k, str(v)[:300])
for k, v in d.get("other_configs", {}).items():
    if "pipelined_stream_cap" in v:
        print(k, "pipelined", v["pipelined_stream_cap"]["value"], v["pipelined_stream_cap"].get("roofline", {}).get("frac"), v["pipelined_stream_cap"].get("blocks_per_launch", {}).get("first_timed_region"))
    else:
        print(k, str(v)[:300])
print("cpu", d.get("cpu_baseline"))
PY
