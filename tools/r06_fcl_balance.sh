#!/bin/bash
# round 6: cfg 4, walk lengths of the long first stages by a cost model (SDRPP_GPU_FCL_BALANCE) — A/B + the cost model's two constants
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r06s_fcl_balance.log
: > $out
run() {
  timeout 300 env "$@" python bench.py --cfg 4 --steps 100 --warmup 10 --no-others --no-cpu-baseline --no-by-push --no-self-check --regions 1 > /tmp/b.json 2>/tmp/b.err
  python - "$*" <<'PY' >> gpurun_out/r06s_fcl_balance.log
import json, sys
try:
    d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "| value", d["value"], "steady", (d.get("steady_state") or {}).get("value"), "avg_launch_ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
except Exception as e:
    print(sys.argv[1], "| failed", e, open("/tmp/b.err").read()[-300:])
PY
}
for rep in 1 2; do
run SDRPP_GPU_FCL_BALANCE=0
run SDRPP_GPU_FCL_BALANCE=1
run SDRPP_GPU_FCL_BALANCE=1 SDRPP_GPU_FCL_C0=1.5
run SDRPP_GPU_FCL_BALANCE=1 SDRPP_GPU_FCL_C0=6
run SDRPP_GPU_FCL_BALANCE=1 SDRPP_GPU_FCL_C0=0.5 SDRPP_GPU_FCL_C1=0.04
done
run SDRPP_GPU_FCL_BALANCE=0 X=group1 
cat $out
make -C sdrplusplus_amd/csrc -s ticktrace 2>&1 | grep -i error
TICK_GROUP=4 timeout 300 python tools/tick_trace_run.py 4 1000000 80 gpurun_out/tt.bin 2>&1 | grep -v amdgpu.ids
timeout 100 python tools/tick_trace.py gpurun_out/tt.bin 12 2>/dev/null > gpurun_out/r06s_tick_timeline_cfg4_B1000000_group4_balanced.txt
rm -f gpurun_out/tt.bin
head -12 gpurun_out/r06s_tick_timeline_cfg4_B1000000_group4_balanced.txt
