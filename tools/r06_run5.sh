#!/bin/bash
# round 6: A/B of the sparse-tick grid boost (SDRPP_GPU_TICK_SPARSE), then the profile call
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r06j_sparse_ab.log
: > $out
run() {
  timeout 300 env "$@" python bench.py --steps 20 --warmup 5 --no-others --no-cpu-baseline --no-by-push --no-self-check > /tmp/b.json 2>/tmp/b.err
  python - "$*" <<'PY' >> gpurun_out/r06j_sparse_ab.log
import json, sys
try:
    d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "| value", d["value"], "regions", d["timed_regions"]["Msamples_per_s"], "steady", d["steady_state"]["value"], "avg_launch_ms", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[1], "| failed", e, open("/tmp/b.err").read()[-300:])
PY
}
for i in 1 2 3; do
  run SDRPP_GPU_TICK_SPARSE=0
  run SDRPP_GPU_TICK_SPARSE=1
done
run SDRPP_GPU_TICK_SPARSE=0 X=push50000
cat $out
for s in 0 1; do
  SDRPP_GPU_TICK_SPARSE=$s python - <<'PY'
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sdrplusplus_amd import capi, workloads
B = 50000
ctx = capi.Context(0, max_push=B * 8)
info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=32)
xs = np.concatenate([workloads.synth(3, B, seed=7 + i, nvfo=32) for i in range(8)])
xd = torch.from_numpy(xs.view(np.float32)).to("cuda:0")
for g in (1, 8):
    ctx.set_pipelined(True, 0)
    ctx.set_pipeline_group(g, True)
    for i in range(64): ctx.push_device(xd.data_ptr() + 8 * B * (i % 8), B)
    ctx.sync()
    best = 0
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(1200): ctx.push_device(xd.data_ptr() + 8 * B * (i % 8), B)
        ctx.sync()
        best = max(best, B * 1200 / (time.perf_counter() - t0) / 1e6)
    print("SPARSE=%s sr/200 device no read, group %d: %.0f MS/s" % (os.environ["SDRPP_GPU_TICK_SPARSE"], g, best))
    ctx.set_pipelined(False)
ctx.close()
PY
done 2>&1 | grep -v amdgpu.ids | tee -a $out
bash tools/profile_r06.sh r06p 2>&1 | tail -40
