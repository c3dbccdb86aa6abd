#!/bin/bash
# round 6: device suite + the driver's bench command (full line) with the current tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r06i_pytest_gpu.log 2>&1
tail -4 gpurun_out/r06i_pytest_gpu.log
timeout 1800 python bench.py --steps 20 --warmup 5 > gpurun_out/r06i_bench_default.json 2> gpurun_out/r06i_bench_default.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06i_bench_default.json").read().strip().splitlines()[-1])
print("default: value", d["value"], d["timed_regions"]["Msamples_per_s"], "steady", d.get("steady_state"), "frac", d["roofline"]["frac"], "avg launch", d["roofline"]["avg_launch_ms"])
print("config sr/200:", json.dumps(d["config"].get("at_the_reference_block_size_sr_200")))
print("cap delivered:", d["config"].get("at_the_stream_cap_every_VFO_block_and_lines_delivered"))
cb = d.get("cpu_baseline", {})
print("cpu:", cb.get("value"), cb.get("cores"), cb.get("threaded_graph"), cb.get("per_stage_single_thread"))
print("gpu_over_cpu", d.get("gpu_over_cpu"), d.get("gpu_over_cpu_threaded_graph"))
bp = d.get("by_push", {})
for k in ("cpp_iqfrontend_run_bypass_pipelined", "cpp_iqfrontend_run_bypass_pipelined_one_block_per_launch", "cpp_iqfrontend_built_against", "cpp_iqfrontend_cpus"):
    print(k, bp.get(k))
for k, v in d.get("other_configs", {}).items():
    if isinstance(v, dict):
        for kk, vv in v.items():
            if isinstance(vv, dict) and "value" in vv:
                print(k, kk, vv["value"], (vv.get("roofline") or {}).get("frac"))
PY
