# AF chain in pipelined mode: timelines of the ticks and the two bench entries
python tools/tick_trace_run.py 3 1000000 40 /tmp/af1m.bin af > /dev/null 2>&1; python tools/tick_trace.py /tmp/af1m.bin 20 2>&1 | head -45
python tools/tick_trace_run.py 3 50000 200 /tmp/af50k.bin af > /dev/null 2>&1; python tools/tick_trace.py /tmp/af50k.bin 60 2>&1 | head -45
python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from sdrplusplus_amd import capi, workloads
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
r, _ = bench.run_workload(torch, np, dev, 0, 3, 1000000, "pipelined", 100, 14, 32, af=True, check=True)
print("cfg3 + AF pipelined 10^6:", r["value"], "MS/s", r["ms_per_step"], "ms/step tick", r["roofline"]["avg_launch_ms"], r.get("self_check", {}).get("identical"), r["pipeline"])
print("sr/200 + AF delivered:", bench.af_sr200_delivered(torch, capi, workloads, 10e6, 32))
PY
