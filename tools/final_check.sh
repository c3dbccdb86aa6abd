#!/bin/bash
# the state the round ends in: device suite + the default bench line as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05zb}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/${TAG}_pytest_gpu.log
tail -2 gpurun_out/${TAG}_pytest_gpu.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${TAG}_bench.err | tail -1 ) > gpurun_out/${TAG}_bench_default.json
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_default.json'))
print('value', d['value'], 'steady', d.get('steady_state',{}).get('value'), 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'))
oc=d.get('other_configs',{})
for k,v in oc.items():
    if isinstance(v,dict): print(k, {kk:(vv.get('value') if isinstance(vv,dict) else None) for kk,vv in v.items() if kk!='workload'})
print(d.get('by_push',{}).get('cpp_iqfrontend_run_bypass_pipelined'))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
# what pipelined mode gives a small bank (cfg 1: 2.4 MS/s, one WFM VFO, 4096-point FFT) at the reference's block size
( timeout 200 python tools/tick_rate.py 1 12000 2>&1 | grep '^{' ) | tee gpurun_out/${TAG}_tick_rate_cfg1.json
