#!/usr/bin/env python3
"""Timeline of the tick kernel from a `make ticktrace` build (SDRPP_TICK_TRACE_FILE dump): per role, when its workgroups start and end
inside a tick (100 MHz wall clock), averaged over the steady-state ticks.   tools/tick_trace.py dump.bin [skip_ticks] [role:grid_x]
(role:grid_x, e.g. fcl_pf:62 — that role's workgroups also per JOB, job = workgroup index // grid_x)"""
import sys

import numpy as np

ROLES = ["none", "copy", "carry", "rot", "fcm_132_4", "fcm_6", "fcm_10", "fcm_16", "fcm16_132_4", "fcl_0", "fcl_pf", "toep_c", "toep_r", "toep_q", "firb_c", "firb_r", "firb_s", "firb_q",
         "pre", "seq", "fft_s10", "fft_s11", "fft_s12", "p1_5", "p1_6", "p1_7", "p1_8", "p1_9", "p1_10", "p2_7", "p2_8", "p2_9", "p2_10", "p2row", "transp", "zoom16", "zoom4", "zoom1", "fcm16w_132_4", "polyc", "deemp_p0", "deemp_p1", "dc_p0", "dc_p1", "wf_ring", "wf_trace", "pipe", "rotx16", "fird", "ssbx", "s1_1", "s1d_1", "f2_1", "poly"]
dt = np.dtype({"names": ["tick", "role", "entry", "block", "t0", "t1", "hwid", "xcc", "m"], "formats": ["<u4", "<i2", "<i2", "<i4", "<u8", "<u8", "<u4", "<u4", ("<u8", 4)], "offsets": [0, 4, 6, 8, 16, 24, 32, 36, 40], "itemsize": 72})
a = np.fromfile(sys.argv[1], dtype=dt)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ticks = np.unique(a["tick"])
ticks = ticks[skip:-5] if len(ticks) > skip + 10 else ticks
print("%d records, %d ticks analysed" % (len(a), len(ticks)))
rows = {}
span, nblk = [], []
prev_end = None
gaps = []
for t in ticks:
    r = a[a["tick"] == t]
    base = r["t0"].min()
    end = r["t1"].max()
    span.append((end - base) / 100.0)
    nblk.append(len(r))
    if prev_end is not None:
        gaps.append((float(base) - float(prev_end)) / 100.0)
    prev_end = end
    for key in set(zip(r["role"].tolist(), r["entry"].tolist())):
        m = r[(r["role"] == key[0]) & (r["entry"] == key[1])]
        mk = [float(np.mean((m["m"][:, q].astype(np.int64) - m["t0"].astype(np.int64))[m["m"][:, q] > 0])) / 100.0 if np.any(m["m"][:, q] > 0) else float("nan") for q in range(4)]
        rows.setdefault(key, []).append(((m["t0"].min() - base) / 100.0, (m["t0"].max() - base) / 100.0, (m["t1"].max() - base) / 100.0, float(np.mean(m["t1"] - m["t0"])) / 100.0, len(m)) + tuple(mk))
hw = a["hwid"]
cu_key = (a["xcc"].astype(np.int64) & 0xf) * 10000 + ((hw >> 13) & 7) * 100 + ((hw >> 8) & 0xf)
one = a["tick"] == ticks[len(ticks) // 2]
u, cnt = np.unique(cu_key[one], return_counts=True)
print("one tick: %d workgroups on %d distinct CUs (xcc, se, cu), per CU min %d max %d" % (one.sum(), len(u), cnt.min(), cnt.max()))
print("tick span: avg %.2f us  median %.2f  p90 %.2f   workgroups per tick avg %.0f   gap between ticks (end -> next start): median %.2f us" % (np.mean(span), np.median(span), np.percentile(span, 90), np.mean(nblk), np.median(gaps) if gaps else 0))
print("%-12s %5s %6s | first start  last start  last end  | mean workgroup life (us) | marks 0..3 (us after the workgroup's start; thread 0)" % ("role", "entry", "wgs"))
for key in sorted(rows, key=lambda k: (k[1], k[0])):
    v = np.array(rows[key])
    print("%-12s %5d %6.0f | %10.2f %11.2f %9.2f | %8.2f   (in %d ticks) | %s" % (ROLES[key[0]] if 0 <= key[0] < len(ROLES) else ("L0" if key[0] < 0 else str(key[0])), key[1], v[:, 4].mean(), v[:, 0].mean(), v[:, 1].mean(), v[:, 2].mean(), v[:, 3].mean(), len(v),
          "  ".join("%6.2f" % x for x in np.nanmean(v[:, 5:9], axis=0))))

if len(sys.argv) > 3:  # one role per job (grid.y index): life and marks of its workgroups
    rname, gx = sys.argv[3].split(":")
    ri, gx = ROLES.index(rname), int(gx)
    sel = a[(a["role"] == ri) & np.isin(a["tick"], ticks)]
    t0s = {int(t): int(a[a["tick"] == t]["t0"].min()) for t in ticks}
    print("%s per job (grid.x = %d):  job  wgs/tick | mean life  max life  last end | marks" % (rname, gx))
    for j in sorted(set((sel["block"] // gx).tolist())):
        m = sel[sel["block"] // gx == j]
        life = (m["t1"] - m["t0"]) / 100.0
        ends = np.array([(int(x["t1"]) - t0s[int(x["tick"])]) / 100.0 for x in m])
        mk = [float(np.mean((m["m"][:, q].astype(np.int64) - m["t0"].astype(np.int64))[m["m"][:, q] > 0])) / 100.0 if np.any(m["m"][:, q] > 0) else float("nan") for q in range(4)]
        print("   %3d  %7.1f | %8.2f %9.2f %9.2f | %s" % (j, len(m) / max(1, len(ticks)), life.mean(), life.max(), ends.max(), "  ".join("%6.2f" % x for x in mk)))
