#!/bin/bash
set -u
O=gpurun_out/r03h
mkdir -p $O
timeout 600 python -m pytest tests/test_pipelined.py tests/test_host_cpp.py -m gpu -x -q 2>&1 | tail -5
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
tail -5 $O/bench_default.err | grep -v amdgpu.ids
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03h/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms")} if d.get("roofline") else None)
print("ceiling", {k: d["ceiling"].get(k) for k in ("value", "ms_per_step", "error")})
for k, v in d.get("other_configs", {}).items():
    print(k, {kk: (vv.get("value"), vv.get("ms_per_step"), (vv.get("roofline") or {}).get("frac")) if isinstance(vv, dict) else vv for kk, vv in v.items() if kk != "workload"})
print("cpu", d.get("cpu_baseline", {}).get("value"), "gpu/cpu", d.get("gpu_over_cpu"))
for k, v in d.get("by_push", {}).items():
    print(k, v if k != "note" else "")
PY
SDRPP_BENCH_FORCE_RCCL=1 timeout 300 python bench.py --no-others --no-by-push --no-cpu-baseline --steps 50 > $O/bench_rccl1.json 2> $O/bench_rccl1.err; tail -3 $O/bench_rccl1.err | grep -v amdgpu.ids; python3 -c "
import json; d=json.loads(open('gpurun_out/r03h/bench_rccl1.json').read().strip().splitlines()[-1]); print('rccl run', d['value'], d.get('rccl'))"
