#!/bin/bash
# Kernel trace of 20-block timed regions (tools/r06_region_timeline.py): every tick launch of the last region with its start (relative to the region's first
# launch), duration and the gap in front of it.   usage: bash tools/r06_region_trace.sh [group] [adaptive]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06u_region_trace
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/tools/r06_region_timeline.py ${1:-4} ${2:-1} > $O/run.log 2>&1
python - "$(find $O/trace -name '*.db' | head -1)" <<'PY' > $O/launches.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels where name like '%tick_kernel%' order by start").fetchall()
# regions: a pause of more than 150 us in front of a launch starts a new one
regs, cur = [], []
for s, e, n in rows:
    if cur and s - cur[-1][1] > 150000:
        regs.append(cur); cur = []
    cur.append((s, e, n))
regs.append(cur)
for r in regs[-3:]:
    t0 = r[0][0]
    print("region of %d launches, first start -> last end %.1f us, sum of launch durations %.1f us" % (len(r), (r[-1][1] - t0) / 1e3, sum(e - s for s, e, _ in r) / 1e3))
    prev = None
    for s, e, n in r:
        print("   start %8.1f  dur %7.1f  gap %5.1f" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
        prev = e
PY
tail -5 $O/run.log; cat $O/launches.txt
