#!/usr/bin/env python3
"""bench.py — BASELINE.json headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload by --cfg (default: 3 on one GPU, 5 on several):
  2  10 MS/s, 65536-pt Nuttall FFT + log-power + waterfall line only (dense framing: every sample transformed)   [HBM-bound]
  3  cfg 2 + 32 VFOs x WFM (translation, 8/2/2 decimating FIR cascade, 4/5 polyphase resampler, 126-tap channel filter, FM
     discriminator, 237-tap audio low-pass) — the configuration BASELINE.json's metric is quoted on            [FP32 MFMA-bound]
  4  61.44 MS/s, 2^20-pt FFT + 128 VFOs mixed NFM / AM / USB
  5  one cfg-4 stream per GPU, seeds 0..N-1, RCCL gather of the finished (zoomed) waterfall lines on rank 0
A "step" = one pass of the hot path (`sdrpp_push_device`) over one batch of `--push` complex samples already resident in HBM.
Inputs come from sdrplusplus_amd/workloads.synth — the numpy generator the CPU baseline and the parity tests use too.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline      dominant kernel family, HIP events on its launch stream inside the timed region (algorithmic flops or bytes / time)
  cpu_baseline  the reference's own RxVFO / demodulator / FFT code (oracle/_ref) timed on this host on a bounded sample
  by_push       (cfg 3, one GPU) ingest rate as a function of the push size and of how the host drives the C-ABI, incl. the
                reference's block size sr/200 through host pointers and through the C++ IQFrontEnd::run loop
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense FP32 MFMA peak (= packed FP32 VALU peak)

METRIC = {2: "IQ Msamples/s ingested (65536-pt FFT only, cfg2)", 3: "IQ Msamples/s ingested (65536-pt FFT + 32 VFO WFM)",
          4: "IQ Msamples/s ingested (2^20-pt FFT + 128 VFO mixed NFM/AM/USB, cfg4)", 5: "IQ Msamples/s ingested (cfg5: one 61.44 MS/s x 128-VFO stream per GPU, RCCL line gather)"}

# kernel family (sdrpp_kernel_family_name) -> kernel-name prefixes as rocprofv3 reports them (profiles/pmc_traffic*.json keys)
FAMILY_KERNELS = {
    "fft_pass1": ["fft_pass1_kernel"], "fft_pass2": ["fft_pass2_kernel"], "fft_single": ["fft_single_kernel"], "zoom_palette": ["zoom_palette_kernel"],
    "vfo_stage1": ["vfo_frontcm_kernel", "vfo_frontcl_kernel", "vfo_front2_kernel", "vfo_stage1_kernel", "vfo_stage1_direct_kernel", "vfo_rotate_kernel"],
    "vfo_decim": ["vfo_toep_kernel<2, 2, false"], "vfo_poly": ["vfo_toep_kernel<2, 2, false", "vfo_polyc_kernel", "vfo_polyb_kernel", "vfo_poly_kernel"],
    "vfo_fir": ["vfo_toep_kernel<1, 2, true", "vfo_toep_kernel<2, 2, false", "vfo_toep_kernel<1, 2, false", "vfo_firb_kernel"],
    "demod": ["vfo_demod_pre_kernel", "vfo_sequential_kernel"], "carry_misc": ["carry_kernel"], "vfo_pipe": ["vfo_pipe_kernel"],
}


def synth_threaded(cfg, n, seed, nvfo=None, chunk=1 << 20, workers=16):
    """workloads.synth in parallel chunks (carriers are phase-continuous across chunks; the noise is drawn per chunk)."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    from sdrplusplus_amd import workloads

    starts = list(range(0, n, chunk))
    with ThreadPoolExecutor(workers) as ex:
        parts = list(ex.map(lambda s0: workloads.synth(cfg, min(chunk, n - s0), seed=seed + s0 // chunk, nvfo=nvfo, start=s0), starts))
    return np.concatenate(parts)


def cpu_baseline(base_cfg, nvfo, fft_size):
    """The reference's own code (oracle/_ref: reference headers compiled -O3 -march=native against the restated VOLK / FFT shims) on this
    host's cores, same workload in blocks of sr/200, bounded to roughly 10-20 s of CPU time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import support as S

    lib = S.ref(fast=True)
    if lib is None:
        return None
    from sdrplusplus_amd import workloads

    sr = workloads.CFG[base_cfg]["sr"]
    block = int(sr / 200)
    plan = workloads.vfo_plan(base_cfg, nvfo) if nvfo else []
    cores = max(1, min(os.cpu_count() or 1, max(nvfo, 1)))  # one worker per VFO at most; thread 0 also runs the FFT branch
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    offs = np.array([c for _, _, _, c, _ in plan] or [0.0], dtype=np.float64)
    rates = np.array([r for _, r, _, _, _ in plan] or [0.0], dtype=np.float64)
    bws = np.array([b for _, _, b, _, _ in plan] or [0.0], dtype=np.float64)
    modes = np.array([S.MODES[m] for m, _, _, _, _ in plan] or [0], dtype=np.int32)
    n0 = max(block * 8, (fft_size * 2 // block + 1) * block)
    x = synth_threaded(base_cfg, n0, seed=21, nvfo=nvfo if nvfo else None)
    xp = S._fp(x.view(np.float32))

    def run(n, repeat):
        return lib.ref_bench_cfg(xp, n, block, sr, len(plan), dp(offs), dp(rates), dp(bws), modes.ctypes.data_as(C.POINTER(C.c_int)), fft_size, cores, repeat)

    run(n0, 1)  # warm caches / tables
    t = run(n0, 1)
    repeat = int(max(1, min(2000, round(12.0 / max(t, 1e-4)))))
    t = run(n0, repeat)
    total = n0 * repeat
    what = ("%d VFOs (%s) via the reference's RxVFO::process + radio demodulators, " % (len(plan), "/".join(sorted({m for m, _, _, _, _ in plan})))) if plan else ""
    return {
        "value": round(total / t / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "reference",
        "sample": "%d samples of cfg%d (%s%d-pt windowed FFT + log-power per %d samples) in blocks of %d, %d worker threads, reference headers compiled "
                  "-O3 -march=native against the restated VOLK (vectorised dot products) / FFT shim — genuine libvolk/libfftw3f are not installed; %.1f s of CPU time"
                  % (total, base_cfg, what, fft_size, fft_size, block, cores, t),
    }


def algorithmic_work(push, plan, sr, nvfo, piped=False):
    """ALGORITHMIC flops / compulsory HBM bytes of ONE launch set of every kernel family for one push: real-tap FIR on
    complex data = 4 flop per tap and output; the fused front end in its tap-pair form (8 flop per pair and output + the NCO's two
    complex products); real audio filter 2 flop per tap.  Families hold several launches (vfo_fir = channel filter + audio low-pass).
    piped: the FM back ends ran as the pipelined launch (family vfo_pipe = last decimator + resampler + channel filter + discriminator /
    audio low-pass of those VFOs, compulsory bytes = front-end stream in, IF + audio out); their work leaves the other three families."""
    from sdrplusplus_amd import radio

    by = {"fft_pass1": push * 16.0, "fft_pass2": push * 12.0, "fft_single": push * 12.0, "zoom_palette": push * 4.0}
    fl = {"fft_pass1": push * (3 * 2 * 8 + 8 + 2.0), "fft_pass2": push * (3 * 2 * 8 + 12.0), "fft_single": push * (5 * 12 + 12.0)}
    bound = {"fft_pass1": "hbm", "fft_pass2": "hbm", "fft_single": "hbm", "zoom_palette": "hbm", "carry_misc": "hbm", "demod": "hbm"}
    if not nvfo:
        return fl, by, bound
    for f in ("vfo_stage1", "vfo_decim", "vfo_poly", "vfo_fir", "vfo_pipe"):
        fl[f], by[f], bound[f] = 0.0, 0.0, "mfma"
    by["vfo_stage1"] += push * 8.0  # the IQ stream is read once for all VFOs of a front-end job
    cache = {}
    for mode, if_rate, bw, centre, _ in plan:
        key = (mode, if_rate, bw)
        if key not in cache:
            d, _keep = radio.vfo_desc(sr, if_rate, bw, centre, mode)
            cache[key] = dict(stages=[(int(d.stage_decim[i]), int(d.stage_ntaps[i])) for i in range(d.n_stages)], interp=int(d.interp), decim=int(d.decim),
                              rtaps=int(d.resamp_ntaps), chan=int(d.chan_ntaps), audio=int(d.audio_ntaps), fm=mode in ("WFM", "NFM"))
        g = cache[key]
        n = float(push)
        st = list(g["stages"])
        if st:
            D1, K1 = st[0]
            fused = len(st) >= 2 and D1 < 32  # stages 0 + 1 as one composite filter (vfo_frontcm / vfo_front2); long first stages run alone
            D2, K2 = st[1] if fused else (1, 1)
            Kc, Dc = K1 + (K2 - 1) * D1, D1 * D2
            n = n / Dc
            fl["vfo_stage1"] += n * (((Kc + 1) // 2) * 8.0 + 16.0)
            by["vfo_stage1"] += n * 8.0
            rest = st[2 if fused else 1:]
            # (the pipelined launch takes the LAST decimator stage of an FM VFO whose plan has a resampler, a channel filter and an audio low-pass)
            in_pipe = piped and g["fm"] and bool(rest) and g["interp"] != g["decim"] and g["interp"] <= 15 and g["chan"] and g["audio"]
            for si, (Ds, Ks) in enumerate(rest):
                fam = "vfo_pipe" if (in_pipe and si == len(rest) - 1) else "vfo_decim"
                by[fam] += n * 8 + (0 if fam == "vfo_pipe" else n / Ds * 8)
                n /= Ds
                fl[fam] += n * Ks * 4.0
        else:
            in_pipe = False
            fl["vfo_stage1"] += n * 8.0
            by["vfo_stage1"] += n * 8.0
        if g["interp"] != g["decim"]:
            tpp = -(-g["rtaps"] // g["interp"])
            n_out = n * g["interp"] / g["decim"]
            if in_pipe:
                fl["vfo_pipe"] += n_out * tpp * 4.0
            else:
                by["vfo_poly"] += n * 8 + n_out * 8
                fl["vfo_poly"] += n_out * tpp * 4.0
            n = n_out
        if g["chan"]:
            if in_pipe:
                by["vfo_pipe"] += n * 8  # the IF stream (RxVFO output) is written
                fl["vfo_pipe"] += n * g["chan"] * 4.0
            else:
                by["vfo_fir"] += n * 16
                fl["vfo_fir"] += n * g["chan"] * 4.0
        if g["audio"]:
            if in_pipe:
                by["vfo_pipe"] += n * 8
                fl["vfo_pipe"] += n * g["audio"] * 2.0
            else:
                by["vfo_fir"] += n * (16 if g["fm"] else 12)
                fl["vfo_fir"] += n * g["audio"] * 2.0
    return fl, by, bound


def pmc_traffic(cfg, push, nvfo, dom):
    """HBM bytes per launch set of the dominant family from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs,
    FETCH_SIZE x2 per MI355X_MICROARCH.md, tools/rocpd_summary.py) — only when that profile was taken on this very workload."""
    for name in ("pmc_traffic_cfg%d.json" % cfg, "pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            prof = json.load(open(path))
            meta = prof.get("_meta", {})
            if int(meta.get("push", 0)) != push or int(meta.get("cfg", 0)) != cfg or int(meta.get("nvfo", -1)) != nvfo:
                continue
            hits = [v["hbm_bytes_per_launch"] * v.get("launches_per_push", 1) for k, v in prof.items() if k != "_meta" and any(k.startswith(p) for p in FAMILY_KERNELS.get(dom, []))]
            if hits:
                return round(sum(hits))
        except Exception:
            pass
    return None


def by_push_report(torch, capi, workloads, sr, nvfo):
    """cfg 3 ingest rate against the push size and the way the host drives the C-ABI.  Host buffers are page-locked (sdrpp_host_alloc —
    what the C++ blocks use for their frame-buffer slots) unless marked pageable; every mode DELIVERS the outputs (all VFO blocks +
    line count) to the host, except `device_no_read` (the step bench.py times: inputs resident, outputs left on the device)."""
    import numpy as np

    res = {}
    dev = torch.device("cuda", torch.cuda.current_device())
    for B in (int(sr / 200), 1000000):
        per_pass = max(1, 1000000 // B)
        ctx = capi.Context(dev.index or 0, max_push=B * per_pass)
        info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=nvfo)
        vids = info["vids"]
        xh = workloads.synth(3, B, seed=7, nvfo=nvfo)
        ptr = ctx.L.sdrpp_host_alloc(B * 8)
        C.memmove(ptr, xh.ctypes.data, B * 8)
        xd = torch.from_numpy(xh.view(np.float32)).to(dev)
        entry = {"push": B}

        def rate(fn, npush):
            for _ in range(3):
                fn()
            ctx.sync()
            best = 0.0
            for _trial in range(3):
                t0 = time.perf_counter()
                for _ in range(npush):
                    fn()
                ctx.sync()
                best = max(best, B * npush / (time.perf_counter() - t0) / 1e6)
            return round(best, 1)

        npush = max(6, min(300, (1 << 25) // B))

        def sync_pinned():
            ctx.push_host_ptr(ptr, B)
            ctx.vfo_read_many(vids)
            ctx.fft_lines()

        def sync_pageable():
            ctx.push(xh)
            ctx.vfo_read_many(vids)
            ctx.fft_lines()

        entry["per_push_read_pinned"] = rate(sync_pinned, npush)
        entry["per_push_read_pageable"] = rate(sync_pageable, npush)
        entry["device_no_read"] = rate(lambda: ctx.push_device(xd.data_ptr(), B), npush)
        ctx.set_deferred(True)

        def deferred_pass(pinned=True):
            for _ in range(per_pass):
                if pinned:
                    ctx.push_host_ptr_async(ptr, B)  # page-locked source: the device fetches the block, one wait per pass
                else:
                    ctx.push(xh)
            if pinned:
                ctx.push_wait()
            ctx.vfo_read_many(vids)
            ctx.fft_lines()

        if per_pass > 1:
            entry["deferred_pushes_per_pass"] = per_pass
            entry["deferred_read_pinned"] = round(rate(lambda: deferred_pass(True), max(3, npush // per_pass)) * per_pass, 1)
            entry["deferred_read_pageable"] = round(rate(lambda: deferred_pass(False), max(3, npush // per_pass)) * per_pass, 1)
        ctx.set_deferred(False)
        ctx.L.sdrpp_host_free(ptr)
        ctx.close()
        res["B=%d" % B] = entry
    # through the C++ host mirror (source thread -> dsp::stream -> IQFrontEnd::run -> one sink thread per VFO), reference block size
    try:
        with tempfile.TemporaryDirectory() as tmp:
            exe = os.path.join(tmp, "bench_blocks")
            csrc = os.path.join(ROOT, "sdrplusplus_amd", "csrc")
            subprocess.run(["g++", "-std=c++17", "-O2", "-w", "-o", exe, os.path.join(ROOT, "tests", "host_cpp", "bench_blocks.cpp"), "-I" + os.path.join(ROOT, "tests", "host_cpp", "standalone"),
                            "-L" + csrc, "-lsdrpp_gpu", "-Wl,-rpath," + csrc, "-lpthread"], check=True, capture_output=True)
            for buffered in (0, 1):
                r = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin"), str(sr), str(int(sr / 200)), "65536", str(nvfo), "3", str(buffered)],
                                   capture_output=True, text=True, timeout=120)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                res["cpp_iqfrontend_run_%s" % ("buffered" if buffered else "bypass")] = json.loads(line[-1]) if line else {"error": (r.stdout + r.stderr)[-300:]}
    except Exception as e:
        res["cpp_iqfrontend_run"] = {"error": repr(e)[:300]}
    res["note"] = ("Msamples/s; per_push_read = sdrpp_push (host pointer, H2D included) + sdrpp_vfo_read_many + sdrpp_fft_lines after EVERY push; deferred_read = "
                   "sdrpp_set_deferred: pushes staged (pinned: sdrpp_push_pinned_async + one sdrpp_push_wait), one pass + one read per `deferred_pushes_per_pass` pushes, every push still its own reference block; "
                   "cpp_iqfrontend_run = tests/host_cpp/bench_blocks.cpp (SpeedTester-style source thread, one sink thread per VFO), buffered = 32-slot frame buffer whose "
                   "backlog is staged without per-block waits and processed as one deferred pass, results handed out by the worker and three helper threads")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--push", type=int, default=1 << 24, help="complex samples per step (multiple of the FFT size); 2^24 = 1.7 s of the 10 MS/s stream per GPU and step")
    ap.add_argument("--nvfo", type=int, default=32)
    ap.add_argument("--nbuf", type=int, default=4, help="distinct input batches rotated through (4 x 128 MiB at the default push: never resident in the 256 MiB MALL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-by-push", action="store_true")
    ap.add_argument("--af", action="store_true", help="also run the radio AF chain (resampler to 48 kHz + 50 us de-emphasis) behind every VFO (SURVEY.md 8f row 1; not part of the headline workload)")
    ap.add_argument("--fft-only", action="store_true", help="same as --cfg 2")
    ap.add_argument("--cfg", type=int, default=0, help="BASELINE config 2 / 3 / 4 / 5 (default: 3 on one GPU, 5 on several)")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)  # backend "nccl" is RCCL on ROCm
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    from sdrplusplus_amd import capi, multi, workloads

    cfg = args.cfg if args.cfg in (2, 3, 4, 5) else (2 if args.fft_only else (3 if world == 1 else 5))
    base = 4 if cfg == 5 else cfg  # cfg 5 = one cfg-4 stream per rank
    sr, N = workloads.CFG[base]["sr"], workloads.CFG[base]["fft"]
    push = max(1, args.push // N) * N
    nvfo = 0 if base == 2 else (args.nvfo if base == 3 else 128)
    stream_index = multi.stream_for_rank(rank, world, world)[0]
    seed0 = multi.stream_seed(0x5D2B0001 if base != 4 else 0, stream_index)  # cfg 5: seeds 0 .. N-1
    x0 = synth_threaded(base, push, seed=seed0 * 1000 + 1, nvfo=nvfo if nvfo else None)
    first = torch.from_numpy(x0.view(np.float32)).to(device)
    del x0
    # further batches: the same signal delayed by a few thousand samples (distinct addresses and contents, same statistics)
    bufs = [first] + [torch.roll(first, 2 * 4097 * b).contiguous() for b in range(1, args.nbuf)]
    ctx = capi.Context(local, max_push=push)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # one ordering domain with torch / RCCL
    info = workloads.setup(ctx, base, dense_fft=True, data_width=1024, nvfo=nvfo)
    af_keep = []
    if args.af and nvfo:
        from sdrplusplus_amd import radio
        for vid, (m_, r_, _b, _c, _x) in zip(info["vids"], info["plan"]):
            a_, k_ = radio.af_desc(r_, 48000.0, 50e-6 if m_ == "WFM" else None, m_ == "NFM")
            ctx.vfo_set_af(vid, a_, k_)
            af_keep.append(k_)
    lines_per_push = push // N
    lines = torch.empty((lines_per_push, 1024), dtype=torch.float32, device=device)
    runner = multi.StreamRunner(ctx, bufs, push, lines, sync=torch.cuda.synchronize)

    for i in range(args.warmup):
        runner.step(i)
    torch.cuda.synchronize()
    # calibration (untimed): every kernel family bracketed by HIP events to find the dominant one and record the per-kernel split
    ctx.timing_enable(True)
    ncal = max(2, min(5, args.steps))
    for i in range(ncal):
        runner.step(i)
    torch.cuda.synchronize()
    fam_all = ctx.timing_read()
    kernel_ms_all = {k: v[0] / ncal for k, v in fam_all.items() if v[0] > 0}
    # dominant family = the longest one on the CRITICAL stream: with VFOs present the FFT branch runs on a second stream as filler
    # behind the VFO bank (its launches stretch while they wait for CUs, which says nothing about the kernels themselves), so only
    # the VFO-bank families compete there; FFT-only runs (cfg 2) have just the FFT families.  The HBM-bound FFT kernels are still
    # reported against 8 TB/s in `roofline_fft` below.
    filler = {"fft_pass1", "fft_pass2", "fft_single", "zoom_palette"} if nvfo else set()
    cand = {k: v for k, v in kernel_ms_all.items() if k not in filler} or kernel_ms_all
    dom = max(cand, key=cand.get) if cand else None
    # timed region: only the dominant family keeps its event pair (two event records per step on its launch stream)
    ctx.timing_enable(True, families=[ctx.family_index(dom)] if dom else [])
    elapsed = runner.timed(args.steps, first=args.warmup)
    fam = ctx.timing_read()
    ctx.timing_enable(False)

    # sanity: the work was really done (outputs have the expected sizes)
    assert ctx.fft_lines() == lines_per_push
    if base == 3:
        for vid in info["vids"][:1]:
            assert abs(ctx.vfo_out_count(vid) - push // 40) <= 2
    if world > 1 and rank == 0:
        assert runner.gathered is not None and tuple(runner.gathered.shape) == (world, lines_per_push, 1024)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_samples = world * push * args.steps
    value = total_samples / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel family (HIP events around its launches, on the launch stream, inside the timed region) ----
    fl, by, bound_of = algorithmic_work(push, info["plan"], sr, nvfo, piped=kernel_ms_all.get("vfo_pipe", 0) > 0)
    kernel_ms = {k: v[0] / args.steps for k, v in fam.items() if v[0] > 0}
    roof = None
    if dom is not None and dom in kernel_ms and dom in by:
        dur = kernel_ms[dom] * 1e-3
        gbs = by[dom] / dur / 1e9
        traffic = pmc_traffic(base, push, nvfo, dom)
        if bound_of.get(dom) == "mfma" and fl.get(dom):
            tf = fl[dom] / dur / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": round(tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP32_PEAK_TFLOPS, 5), "traffic": traffic,
                    "algorithmic_flops_per_launch": fl[dom], "algorithmic_bytes_per_launch": by[dom], "hbm_GBps_at_this_rate": round(gbs, 2), "avg_launch_ms": round(kernel_ms[dom], 4),
                    "note": "family = all launches of this kind in one push (vfo_fir = channel filters + discriminator / audio low-passes; vfo_pipe = the pipelined FM back ends: last decimator + resampler + channel filter + discriminator / audio low-pass in one launch); peak = dense FP32 MFMA"}
        else:
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": by[dom], "avg_launch_ms": round(kernel_ms[dom], 4)}
    # the HBM-bound branch on its own (calibration pass: with VFOs present it shares the CUs with the matrix kernels and stretches)
    roof_fft = {}
    for f in ("fft_pass1", "fft_pass2", "fft_single", "zoom_palette"):
        if f in kernel_ms_all and kernel_ms_all[f] > 0:
            g = by[f] / (kernel_ms_all[f] * 1e-3) / 1e9
            roof_fft[f] = {"bound": "hbm", "achieved": round(g, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(base, push, nvfo, f)}
    # FFT 8 in + 4 out, VFO outputs at their IF rates; the IQ read is shared by both branches
    out_rate = sum(r for _, r, _, _, _ in info["plan"]) if nvfo else 0.0
    path_bytes = 12.0 + out_rate / sr * 8
    per_gpu = value * 1e6 / world
    roof_path = {"bound": "hbm", "achieved": round(per_gpu * path_bytes / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(per_gpu * path_bytes / 1e9 / HBM_PEAK_GBS, 5),
                 "algorithmic_bytes_per_sample": path_bytes, "algorithmic_TFLOPs": round(per_gpu * sum(fl.values()) / push / 1e12, 2),
                 "note": "whole step, per GPU: SURVEY.md 8(d) path figure x ingest rate"}

    mode_names = "/".join(sorted({m for m, _, _, _, _ in info["plan"]})) if nvfo else ""
    out = {
        "metric": METRIC[cfg], "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg%d: %.2f MS/s-format synthetic IQ (workloads.synth), %d-pt dense FFT + log-power waterfall%s" % (cfg, sr / 1e6, N, (" + %d VFO x %s (xlate+FIR+resample+demod)" % (nvfo, mode_names)) if nvfo else ""),
                   "samples_per_step_per_gpu": push, "streams": world, "parallelism": "one independent IQ stream per GPU" + ("; RCCL gather of zoomed waterfall lines to rank 0" if world > 1 else ""),
                   "input_batches_rotated": args.nbuf, "af_chain": bool(args.af and nvfo), "device": ctx.device_info()},
        "roofline": roof, "roofline_fft": roof_fft, "roofline_path": roof_path,
        "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(kernel_ms_all.items(), key=lambda kv: -kv[1])},
        "kernel_ms_note": "per-family HIP-event times from an untimed calibration pass (FFT branch and VFO bank run on two streams and overlap, so the "
                          "entries sum to more than ms_per_step); roofline.avg_launch_ms is the dominant family re-measured inside the timed region",
        "realtime_factor": round(per_gpu / sr, 1),
    }
    ctx.close()
    del bufs, first
    torch.cuda.empty_cache()
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(base, nvfo, N)
            if out["cpu_baseline"]:
                out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU result
            out["cpu_baseline"] = {"error": repr(e)}
    if world == 1 and base == 3 and not args.no_by_push:
        try:
            out["by_push"] = by_push_report(torch, capi, workloads, sr, nvfo)
        except Exception as e:
            out["by_push"] = {"error": repr(e)[:400]}
    try:  # C stdio of anything loaded into this process goes out BEFORE the JSON line, which must be the last line on stdout
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
