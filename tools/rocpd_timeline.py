#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, stream/queue) of the last N product kernels in a rocprofv3 rocpd database —
used to look for idle gaps between launches.   tools/rocpd_timeline.py trace.db [N]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
q = "select name, start, end, %s from kernels where name like '%%sdrpp_k::%%' order by start" % (qcol or "0")
rows = db.execute(q).fetchall()[-n:]
t0 = rows[0][1]
for name, st, en, qid in rows:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name).replace("sdrpp_k::", "")
    print("%9.1f us  +%8.1f us  q%-4s %s" % ((st - t0) / 1e3, (en - st) / 1e3, qid, name))
