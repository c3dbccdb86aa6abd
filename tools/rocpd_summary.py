#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs for the product's kernels (names containing `sdrpp_k::`).

  tools/rocpd_summary.py kernel-trace.db [--pmc fetch.db write.db ...] [--out profiles/rNN_xxx.md] [--json profiles/pmc_traffic.json]

Kernel-trace: calls / total / average / min / max duration per kernel.  PMC databases: per-kernel average of each counter.
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; MI355X_MICROARCH.md (HBM section) says FETCH_SIZE under-counts
wide coalesced streaming reads by exactly 2x on gfx950 in this ROCm — both the raw and the corrected (x2) read bytes are
printed; WRITE_SIZE is uncalibrated and printed raw."""
import argparse
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name.replace("sdrpp_k::", "")


def kernel_stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels where name like '%sdrpp_k::%' group by name order by sum(duration) desc").fetchall()
    return [(short(n), c, s, a, mi, ma) for n, c, s, a, mi, ma in rows]


def pmc_stats(path):
    db = sqlite3.connect(path)
    # one row per (dispatch, counter, dimension instance): sum the instances of a dispatch first, then average over dispatches
    rows = db.execute("select kernel_name, counter_name, count(*), avg(v), avg(d) from (select kernel_name, counter_name, dispatch_id, sum(value) as v, "
                      "max(duration) as d from counters_collection where kernel_name like '%sdrpp_k::%' group by kernel_name, counter_name, dispatch_id) "
                      "group by kernel_name, counter_name").fetchall()
    return [(short(k), c, n, v, d) for k, c, n, v, d in rows]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--out")
    ap.add_argument("--json")
    ap.add_argument("--title", default="rocprofv3 summary")
    ap.add_argument("--meta", nargs="*", default=[], help="key=value pairs stored under _meta in the JSON (e.g. push=4194304 cfg=3)")
    a = ap.parse_args()
    lines = ["# " + a.title, "", "## kernel trace (`rocprofv3 --kernel-trace --stats`), product kernels only", "",
             "| kernel | calls | total us | avg us | min us | max us |", "|---|---:|---:|---:|---:|---:|"]
    for n, c, s, av, mi, ma in kernel_stats(a.trace):
        lines.append("| %s | %d | %.1f | %.2f | %.2f | %.2f |" % (n, c, s / 1e3, av / 1e3, mi / 1e3, ma / 1e3))
    traffic = {}
    if a.pmc:
        lines += ["", "## PMC passes (`rocprofv3 --pmc <counter> --kernel-trace`, one counter per pass)", "",
                  "| kernel | counter | dispatches | avg value | avg bytes | note |", "|---|---|---:|---:|---:|---|"]
        per = {}
        for p in a.pmc:
            for k, c, n, v, d in pmc_stats(p):
                note, b = "", None
                per.setdefault(k, {})["n"] = n
                if c == "FETCH_SIZE":
                    b = v * 1024.0
                    note = "raw; x2 = %.0f (gfx950 FETCH_SIZE correction)" % (2 * b)
                    per.setdefault(k, {})["fetch_raw"] = b
                    per[k]["fetch_x2"] = 2 * b
                elif c == "WRITE_SIZE":
                    b = v * 1024.0
                    note = "raw (uncalibrated)"
                    per.setdefault(k, {})["write"] = b
                lines.append("| %s | %s | %d | %.1f | %s | %s |" % (k, c, n, v, ("%.0f" % b) if b is not None else "", note))
        for k, d in per.items():
            if "fetch_x2" in d and "write" in d:
                traffic[k] = {"hbm_bytes_per_launch": d["fetch_x2"] + d["write"], "fetch_raw": d["fetch_raw"], "fetch_x2": d["fetch_x2"], "write": d["write"], "dispatches": d.get("n")}
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    else:
        sys.stdout.write(text)
    if a.json:
        traffic["_meta"] = dict(kv.split("=", 1) for kv in a.meta)
        # launch groups (round 6): a launch carries several blocks — with blocks=<pushes of the profiled run> the bytes per BLOCK follow
        if "blocks" in traffic["_meta"]:
            nb = float(traffic["_meta"]["blocks"])
            tick = [k for k in traffic if k != "_meta" and "tick_kernel" in k and traffic[k].get("dispatches")]
            if tick and nb > 0:
                total = sum(traffic[k]["hbm_bytes_per_launch"] * traffic[k]["dispatches"] for k in tick)
                traffic["_meta"]["tick_hbm_bytes_per_block"] = total / nb
                traffic["_meta"]["tick_launches"] = sum(traffic[k]["dispatches"] for k in tick)
                lines_extra = "\n## per block\n\n%d tick launches carried %d blocks: %.0f HBM bytes per block (FETCH_SIZE x 2 + WRITE_SIZE over all tick launches / blocks)\n" % (traffic["_meta"]["tick_launches"], int(nb), total / nb)
                if a.out:
                    open(a.out, "a").write(lines_extra)
        json.dump(traffic, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
