#!/usr/bin/env python3
"""A / B of library builds and planner switches on ONE box: the pipelined ingest rate (blocks resident on the device, outputs left
there) and the tick kernel's own duration (HIP events on the launch) for every variant, variants interleaved over several rounds so
that box drift shows as scatter instead of as a difference.

    tools/ab_tick.py --cfg 3 --push 1000000 --rounds 3  base=libsdrpp_gpu_base.so  new=libsdrpp_gpu.so  "new_w640=libsdrpp_gpu.so,SDRPP_GPU_TICK_FCM_WAVES=640"

A variant is  label=<library under sdrplusplus_amd/csrc>[,ENV=value ...].  The library switch is a switch of THIS TOOL (capi.DEFAULT_LIB,
as tools/tick_trace_run.py does) — the package itself has no override."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cfg, push, lib, nblocks, af, fft=True, nvfo_override=None):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    from sdrplusplus_amd import capi, workloads

    capi.DEFAULT_LIB = os.path.join(ROOT, "sdrplusplus_amd", "csrc", lib)
    nvfo = workloads.CFG[cfg]["nvfo"] if nvfo_override is None else nvfo_override
    ctx = capi.Context(0, max_push=push)
    info = workloads.setup(ctx, cfg, dense_fft=True, data_width=1024, nvfo=nvfo or None, fft=fft)
    keep = []
    if af and nvfo:
        from sdrplusplus_amd import radio
        for vid, (m_, r_, _b, _c, _x) in zip(info["vids"], info["plan"]):
            a_, k_ = radio.af_desc(r_, 48000.0, 50e-6 if m_ == "WFM" else None, m_ == "NFM")
            ctx.vfo_set_af(vid, a_, k_)
            keep.append(k_)
    sr = workloads.CFG[cfg]["sr"]
    if int(sr / 200) < push:
        ctx.set_reference_block(int(sr / 200))
    x0 = workloads.synth(cfg, push, seed=7, nvfo=nvfo or None)  # (one generated block + five rotations of it: the generator costs seconds per 10^6 samples, the timing does not depend on the content)
    xd = [torch.from_numpy(np.roll(x0, 1009 * i).view(np.float32)).to("cuda") for i in range(6)]
    ctx.set_pipelined(True, 0)
    for i in range(16):
        ctx.push_device(xd[i % 6].data_ptr(), push)
    ctx.sync()
    best = 0.0
    for _trial in range(3):
        t0 = time.perf_counter()
        for i in range(nblocks):
            ctx.push_device(xd[i % 6].data_ptr(), push)
        ctx.sync()
        best = max(best, push * nblocks / (time.perf_counter() - t0) / 1e6)
    ctx.timing_enable(True, families=[ctx.family_index("tick")])
    for i in range(nblocks):
        ctx.push_device(xd[i % 6].data_ptr(), push)
    ctx.sync()
    ms, n = ctx.timing_read()["tick"]
    st = ctx.pipeline_stats()
    ctx.close()
    print(json.dumps({"Msps": round(best, 1), "tick_us": round(ms / max(1, n) * 1e3, 2), "launches": int(n), "pass_blocks": st["pass_blocks"], "depth": st["depth"],
                      "roles": sorted(st["roles"])}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=3)
    ap.add_argument("--push", type=int, nargs="+", default=[1000000])
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--af", action="store_true")
    ap.add_argument("--child", default=None)
    ap.add_argument("--no-fft", action="store_true", help="VFO bank only (what the tick costs without the FFT roles)")
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a.cfg, a.push[0], a.child, a.blocks, a.af, fft=not a.no_fft)
    res = {}
    for rnd in range(a.rounds):
        for v in a.variants:
            label, spec = v.split("=", 1)
            parts = spec.split(",")
            env = dict(os.environ)
            for kv in parts[1:]:
                k, val = kv.split("=", 1)
                env[k] = val
            for push in a.push:
                nb = a.blocks or max(60, min(2000, (1 << 28) // push))
                cmd = [sys.executable, os.path.abspath(__file__), "--cfg", str(a.cfg), "--push", str(push), "--blocks", str(nb), "--child", parts[0]] + (["--af"] if a.af else []) + (["--no-fft"] if a.no_fft else [])
                r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not line:
                    print("## %s push %d FAILED rc %d: %s" % (label, push, r.returncode, (r.stderr or r.stdout)[-600:]), flush=True)
                    continue
                o = json.loads(line[-1])
                res.setdefault((label, push), []).append(o)
                print("round %d  %-28s push %8d  %9.1f MS/s  tick %7.2f us  (%d launches, %d ordinary, depth %d)" % (rnd, label, push, o["Msps"], o["tick_us"], o["launches"], o["pass_blocks"], o["depth"]), flush=True)
    print("---- summary (median over rounds) ----")
    for (label, push), rs in res.items():
        ms = sorted(x["Msps"] for x in rs)
        tk = sorted(x["tick_us"] for x in rs)
        print("%-28s push %8d  %9.1f MS/s  tick %7.2f us   roles %s" % (label, push, ms[len(ms) // 2], tk[len(tk) // 2], ",".join(rs[0]["roles"])))


if __name__ == "__main__":
    main()
