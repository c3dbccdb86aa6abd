#!/bin/bash
# round 6: grid knobs of the roles with launch groups of four 10^6-sample blocks (a tick carries 4x the work per role)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r06c_grid_sweep.log
: > $out
run() {
  timeout 300 env "$@" python bench.py --steps 20 --warmup 5 --group 4 --group-fixed --no-others --no-cpu-baseline --no-by-push --no-self-check > /tmp/b.json 2>/tmp/b.err
  python - "$*" <<'PY' >> gpurun_out/r06c_grid_sweep.log
import json, sys
try:
    d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "| value", d["value"], "regions", d["timed_regions"]["Msamples_per_s"], "steady", d["steady_state"]["value"], "avg_launch_ms", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[1], "| failed", e, open("/tmp/b.err").read()[-300:])
PY
}
run X=0
run SDRPP_GPU_TICK_TOEP_BLOCKS=512
run SDRPP_GPU_TICK_TOEP_BLOCKS=1024
run SDRPP_GPU_TICK_FCM_WAVES=1536
run SDRPP_GPU_TICK_FCM_WAVES=3072
run SDRPP_GPU_TICK_TOEP_BLOCKS=512 SDRPP_GPU_TICK_FCM_WAVES=1536
run SDRPP_GPU_TICK_TOEP_BLOCKS=1024 SDRPP_GPU_TICK_FCM_WAVES=3072
run SDRPP_GPU_FFT_TICK_GRID=512
run SDRPP_GPU_TICK_TOEP_BLOCKS=512 SDRPP_GPU_FFT_TICK_GRID=512
run X=1
cat $out
