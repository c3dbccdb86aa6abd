#!/bin/bash
# round 6: where the FFT roles stand among a 4-block tick's workgroups (role order = dispatch order): steady-state rate, two runs each
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r06r_order_sweep.log
: > $out
run() {
  timeout 300 env "$@" python bench.py --steps 20 --warmup 5 --no-others --no-cpu-baseline --no-by-push --no-self-check --regions 3 > /tmp/b.json 2>/tmp/b.err
  python - "$*" <<'PY' >> gpurun_out/r06r_order_sweep.log
import json, sys
try:
    d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "| value", d["value"], "steady", d["steady_state"]["value"], "avg_launch_ms", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[1], "| failed", e, open("/tmp/b.err").read()[-300:])
PY
}
for rep in 1 2; do
run X=base
run SDRPP_GPU_TICK_P1_WEIGHT=62 SDRPP_GPU_TICK_P2_WEIGHT=61
run SDRPP_GPU_TICK_P1_WEIGHT=95 SDRPP_GPU_TICK_P2_WEIGHT=94
run SDRPP_GPU_TICK_P1_WEIGHT=62
run SDRPP_GPU_TICK_P2_WEIGHT=62
run SDRPP_GPU_TICK_P1_WEIGHT=88 SDRPP_GPU_TICK_P2_WEIGHT=64
done
cat $out
