#!/bin/bash
set -u
O=gpurun_out/r03i
mkdir -p $O
timeout 600 python -m pytest tests/test_pipelined.py tests/test_host_cpp.py -m gpu -x -q 2>&1 | tail -3
g++ -std=c++17 -O2 -w -o /tmp/bench_blocks tests/host_cpp/bench_blocks.cpp -Itests/host_cpp/standalone -Lsdrplusplus_amd/csrc -lsdrpp_gpu -Wl,-rpath,$PWD/sdrplusplus_amd/csrc -lpthread
nproc
for spec in "0 1" "0 1" "0 0" "1 0"; do
  set -- $spec
  /tmp/bench_blocks sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 3 $1 $2 2>&1 | tail -1 | tee -a $O/cpp_blocks.log
done
/tmp/bench_blocks sdrplusplus_amd/data/decim_plans.bin 10000000 200000 65536 32 3 0 1 2>&1 | tail -1 | tee -a $O/cpp_blocks.log
( time timeout 900 python bench.py --no-by-push --no-cpu-baseline > $O/bench_nobp.json 2> $O/bench_nobp.err ) 2>&1 | tail -4
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03i/bench_nobp.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms")} if d.get("roofline") else None)
print("ceiling", {k: d["ceiling"].get(k) for k in ("value", "ms_per_step", "error")})
for k, v in d.get("other_configs", {}).items():
    print(k, {kk: (vv.get("value"), vv.get("ms_per_step"), (vv.get("roofline") or {}).get("frac"), (vv.get("roofline") or {}).get("avg_launch_ms")) if isinstance(vv, dict) else vv for kk, vv in v.items() if kk != "workload"})
PY
