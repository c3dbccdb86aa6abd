#!/usr/bin/env python3
"""bench.py — BASELINE.json headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload by --cfg (default: 3 on one GPU, 5 on several):
  2  10 MS/s, 65536-pt Nuttall FFT + log-power + waterfall line only (dense framing: every sample transformed)   [HBM-bound]
  3  cfg 2 + 32 VFOs x WFM (translation, 8/2/2 decimating FIR cascade, 4/5 polyphase resampler, 126-tap channel filter, FM
     discriminator, 237-tap audio low-pass) — the configuration BASELINE.json's metric is quoted on            [FP32 MFMA-bound]
  4  61.44 MS/s, 2^20-pt FFT + 128 VFOs mixed NFM / AM / USB
  5  one cfg-4 stream per GPU, seeds 0..N-1, RCCL gather of the finished (zoomed) waterfall lines on rank 0

A "step" = ONE BLOCK of `--push` complex samples (default 1 000 000 = the most a dsp::stream hand-over can carry,
core/src/dsp/stream.h:9) already resident in HBM, handed to the hot path with `sdrpp_push_device` in PIPELINED mode
(sdrpp_set_pipelined: one launch per block, the stages of consecutive blocks skewed over consecutive launches, include/sdrpp_gpu.h)
with the reference's block semantics inside it (`sdrpp_set_reference_block(sr / 200)`: AGC look-ahead / rotator renormalisation see
the blocks the file source would have cut, file_source/src/main.cpp:157), the zoomed waterfall lines of every block delivered into
page-locked host memory and handed on in batches (multi.StreamRunner; with N > 1 gathered over RCCL).  The timed region = exactly K
blocks pushed AND all their results delivered.  `--mode ordinary` runs the same blocks as ordinary passes (one launch per stage).
Inputs come from sdrplusplus_amd/workloads.synth — the numpy generator the CPU baseline and the parity tests use too.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline       dominant kernel (pipelined: the tick launch = every stage of the path over one block), HIP events on its launch stream
                 inside the timed region (algorithmic flops or bytes / time)
  cpu_baseline   the reference's own RxVFO / demodulator / FFT code (oracle/_ref) timed on this host on a bounded sample
  ceiling        (one GPU) the same workload as ONE ordinary pass over 2^24 samples — what the kernels do with batches the
                 dsp::stream boundary cannot carry; last round's headline, kept for comparison
  other_configs  (one GPU) short runs of cfg 2 and cfg 4: pipelined at the stream cap + the 2^24 ceiling
  by_push        (cfg 3, one GPU) ingest rate against the block size and the way the host drives the C-ABI, incl. the reference's
                 block size sr/200 through host pointers, with results delivered, and through the C++ IQFrontEnd::run loop
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
STREAM_CAP = 1000000      # STREAM_BUFFER_SIZE, core/src/dsp/stream.h:9

METRIC = {2: "IQ Msamples/s ingested (65536-pt FFT only, cfg2)", 3: "IQ Msamples/s ingested (65536-pt FFT + 32 VFO WFM)",
          4: "IQ Msamples/s ingested (2^20-pt FFT + 128 VFO mixed NFM/AM/USB, cfg4)", 5: "IQ Msamples/s ingested (cfg5: one 61.44 MS/s x 128-VFO stream per GPU, RCCL line gather)"}

# kernel family (sdrpp_kernel_family_name) -> kernel-name prefixes as rocprofv3 reports them (profiles/pmc_traffic*.json keys)
FAMILY_KERNELS = {
    "fft_pass1": ["fft_pass1_kernel"], "fft_pass2": ["fft_pass2_kernel"], "fft_single": ["fft_single_kernel"], "zoom_palette": ["zoom_palette_kernel"],
    "vfo_stage1": ["vfo_frontcm_kernel", "vfo_frontcl_kernel", "vfo_front2_kernel", "vfo_stage1_kernel", "vfo_stage1_direct_kernel", "vfo_rotate_kernel"],
    "vfo_decim": ["vfo_toep_kernel<2, 2, false"], "vfo_poly": ["vfo_toep_kernel<2, 2, false", "vfo_polyc_kernel", "vfo_polyb_kernel", "vfo_poly_kernel"],
    "vfo_fir": ["vfo_toep_kernel<1, 2, true", "vfo_toep_kernel<2, 2, false", "vfo_toep_kernel<1, 2, false", "vfo_firb_kernel"],
    "demod": ["vfo_demod_pre_kernel", "vfo_sequential_kernel"], "carry_misc": ["carry_kernel"], "vfo_pipe": ["vfo_pipe_kernel"], "tick": ["tick_kernel", "void sdrpp_k::tick_kernel"],
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
    """The reference's own code (oracle/_ref: reference headers + iq_frontend.cpp compiled -O3 -march=native against the restated VOLK / FFT shims) on this host's
    cores, same workload in blocks of sr/200.  Two figures:
      value            process()-only: the VFO chains dealt round-robin to one worker thread per VFO (at most one per core), the FFT branch on a thread of its own,
                       no stream hand-overs — the reference's arithmetic at its best, generous to the CPU; bounded to ~10 s of CPU time
      threaded_graph   the reference's own THREADED graph (IQFrontEnd with its Splitter / Reshaper / Handler threads, one RxVFO thread + one demodulator thread + one
                       reader per VFO, every hand-over a dsp::stream swap), an unthrottled SpeedTester-style source, 0.5 s warm-up, median of five 2 s runs
      per_stage        single-thread process() rates of one RxVFO, one demodulator, the FFT handler"""
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
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(ncpu, max(nvfo, 1)))  # one worker per VFO at most
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    offs = np.array([c for _, _, _, c, _ in plan] or [0.0], dtype=np.float64)
    rates = np.array([r for _, r, _, _, _ in plan] or [0.0], dtype=np.float64)
    bws = np.array([b for _, _, b, _, _ in plan] or [0.0], dtype=np.float64)
    modes = np.array([S.MODES[m] for m, _, _, _, _ in plan] or [0], dtype=np.int32)
    mp = modes.ctypes.data_as(C.POINTER(C.c_int))
    n0 = max(block * 8, (fft_size * 2 // block + 1) * block)
    x = synth_threaded(base_cfg, n0, seed=21, nvfo=nvfo if nvfo else None)
    xp = S._fp(x.view(np.float32))

    def run(n, repeat):
        return lib.ref_bench_cfg(xp, n, block, sr, len(plan), dp(offs), dp(rates), dp(bws), mp, fft_size, workers, repeat)

    run(n0, 1)  # warm caches / tables
    t = run(n0, 1)
    repeat = int(max(1, min(2000, round(8.0 / max(t, 1e-4)))))
    t = run(n0, repeat)
    total = n0 * repeat
    what = ("%d VFOs (%s) via the reference's RxVFO::process + radio demodulators, " % (len(plan), "/".join(sorted({m for m, _, _, _, _ in plan})))) if plan else ""
    out = {
        "value": round(total / t / 1e6, 4), "unit": "Msamples/s", "cores": workers + (1 if fft_size else 0), "kind": "reference",
        "sample": "process() only: %d samples of cfg%d (%s%d-pt windowed FFT + log-power per %d samples on a thread of its own) in blocks of %d, %d worker threads, reference headers compiled "
                  "-O3 -march=native against the restated VOLK (vectorised dot products) / FFT shim — genuine libvolk/libfftw3f are not installed; %.1f s of wall time"
                  % (total, base_cfg, what, fft_size, fft_size, block, workers, t),
    }
    try:  # the threaded graph, SpeedTester-style (speed_tester.h:31-56, 78-92)
        nruns = 5
        rr = (C.c_double * nruns)()
        af, ln = C.c_longlong(), C.c_longlong()
        got = lib.ref_bench_graph(xp, len(x), block, sr, len(plan), dp(offs), dp(rates), dp(bws), mp, fft_size, sr / float(fft_size), 0.5, 2.0, nruns, rr, C.byref(af), C.byref(ln))
        runs = sorted(rr[i] / 1e6 for i in range(got))
        if runs:
            out["threaded_graph"] = {"value": round(runs[len(runs) // 2], 4), "unit": "Msamples/s", "runs": [round(r, 4) for r in runs], "threads": 8 + 3 * len(plan), "hardware_threads": ncpu,
                                     "what": "the reference's IQFrontEnd (iq_frontend.cpp verbatim) + addVFO()'s RxVFO blocks + one radio demodulator block + one reader per VFO, dsp::stream hand-overs "
                                             "throughout, unthrottled source in blocks of %d, dense framing; 0.5 s warm-up, median of %d runs of 2 s (samples the source got rid of, back-pressured by the "
                                             "slowest branch)" % (block, len(runs)),
                                     "audio_frames": int(af.value), "lines": int(ln.value)}
    except Exception as e:
        out["threaded_graph"] = {"error": repr(e)[:200]}
    # `value` = the reference's threaded graph (what SURVEY.md 8(d) defines as the CPU baseline) when it ran; the process()-only figure stays beside it
    tg = out.get("threaded_graph", {})
    if "value" in tg:
        out["process_only"] = {"value": out["value"], "unit": "Msamples/s", "threads": out["cores"], "sample": out["sample"]}
        out["value"] = tg["value"]
        out["cores"] = min(ncpu, tg["threads"])
        out["sample"] = ("the reference's own threaded graph, SpeedTester-style: " + tg["what"] + "; %d threads on %d hardware threads; reference headers + iq_frontend.cpp compiled -O3 -march=native against "
                         "the restated VOLK (vectorised dot products) / FFT shim — genuine libvolk / libfftw3f are not installed; median of %s; 10.5 s of wall time" % (tg["threads"], ncpu, tg["runs"]))
    try:  # per-stage single-thread rates
        m0, r0, b0, c0 = (modes[0], rates[0], bws[0], offs[len(offs) // 3]) if plan else (0, 250e3, 150e3, 0.0)
        out["per_stage_single_thread"] = {
            "rx_vfo_Msps_in": round(lib.ref_bench_stage(0, xp, len(x), block, sr, r0, b0, c0, int(m0), fft_size, 1.0) / 1e6, 2) if plan else None,
            "demodulator_Msps_if": round(lib.ref_bench_stage(1, xp, len(x), block, sr, r0, b0, c0, int(m0), fft_size, 1.0) / 1e6, 3) if plan else None,
            "fft_handler_Msps_in": round(lib.ref_bench_stage(2, xp, len(x), block, sr, r0, b0, c0, int(m0), fft_size, 1.0) / 1e6, 2) if fft_size else None,
            "what": "process() of ONE RxVFO (translation, decimators, resampler, channel filter) / ONE demodulator fed that VFO's IF / the FFT handler (window, FFT shim, log-power), one thread each, 1 s",
        }
    except Exception as e:
        out["per_stage_single_thread"] = {"error": repr(e)[:200]}
    return out


def algorithmic_work(push, plan, sr, nvfo, piped=False, fft_n=0, data_width=1024):
    """ALGORITHMIC flops / compulsory HBM bytes of ONE launch set of every kernel family for one push: real-tap FIR on
    complex data = 4 flop per tap and output; the fused front end in its tap-pair form (8 flop per pair and output + the NCO's two
    complex products); real audio filter 2 flop per tap.  Families hold several launches (vfo_fir = channel filter + audio low-pass).
    piped: the FM back ends ran as the pipelined launch (family vfo_pipe = last decimator + resampler + channel filter + discriminator /
    audio low-pass of those VFOs, compulsory bytes = front-end stream in, IF + audio out); their work leaves the other three families."""
    from sdrplusplus_amd import radio

    by = {"fft_pass1": push * 16.0, "fft_pass2": push * 12.0, "fft_single": push * 12.0, "zoom_palette": push * 4.0}
    if fft_n > 4096:
        # doZoom over transforms above 4096 points reads the maxima of aligned groups of R bins that FFT pass 2 (or the transpose pass) leaves —
        # R = its rows per workgroup — plus up to 2 R ragged bins per pixel, and writes 8 bytes per pixel: that, not the whole line, is what it must move
        R = 32 if fft_n <= (1 << 14) else 16
        by["zoom_palette"] = push * 4.0 / R + (push / float(fft_n)) * data_width * (8.0 + 2 * R * 4.0)
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


def pmc_traffic(cfg, push, nvfo, dom, af=False, group=1, per_block=False):
    """HBM bytes per launch set of the dominant family from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs,
    FETCH_SIZE x2 per MI355X_MICROARCH.md, tools/rocpd_summary.py) — only when that profile was taken on this very workload (same block size, VFO count
    and blocks per launch).  per_block: the tick's bytes per BLOCK (a launch of pipelined mode carries up to `group` blocks)."""
    for name in ("pmc_traffic_cfg%d_push%d_group%d.json" % (cfg, push, group), "pmc_traffic_cfg%d_push%d.json" % (cfg, push), "pmc_traffic_cfg%d.json" % cfg, "pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            prof = json.load(open(path))
            meta = prof.get("_meta", {})
            if int(meta.get("push", 0)) != push or int(meta.get("cfg", 0)) != cfg or int(meta.get("nvfo", -1)) != nvfo or bool(int(meta.get("af", 0))) != bool(af) or int(meta.get("group", 1)) != group:
                continue
            if per_block:
                return round(float(meta["tick_hbm_bytes_per_block"])) if "tick_hbm_bytes_per_block" in meta else None
            if dom == "tick" and "tick_hbm_bytes_per_block" in meta and meta.get("tick_launches"):
                return round(float(meta["tick_hbm_bytes_per_block"]) * float(meta["blocks"]) / float(meta["tick_launches"]))  # the average launch of that run
            hits = [v["hbm_bytes_per_launch"] * v.get("launches_per_push", 1) for k, v in prof.items() if k != "_meta" and any(k.startswith(p) for p in FAMILY_KERNELS.get(dom, []))]
            if hits:
                return round(max(hits) if dom == "tick" else sum(hits))  # (a tick is ONE launch; cfg 4's profile holds both builds of the tick kernel: the steady-state one is the larger)
        except Exception:
            pass
    return None


def path_work(push, plan, sr, nvfo, N):
    """Whole-path ALGORITHMIC flops and compulsory HBM bytes of one block of `push` samples: what one tick launch of the
    pipelined mode carries in steady state (every stage of the path, each on a different block)."""
    fl, by, _ = algorithmic_work(push, plan, sr, nvfo, piped=False)
    m = N.bit_length() - 1
    fft_flops = push * (5.0 * m + 12.0)
    vfo_flops = sum(v for k, v in fl.items() if k.startswith("vfo_"))
    out_rate = sum(r for _, r, _, _, _ in plan) if nvfo else 0.0
    bytes_per_sample = 12.0 + out_rate / sr * 8
    return fft_flops + vfo_flops, bytes_per_sample * push, bytes_per_sample


def make_inputs(torch, np, device, base, push, seed, nvfo, min_bytes=384 << 20):
    """Resident input blocks: slices of one synthetic signal and of delayed copies of it — more than the 256 MiB Infinity Cache
    holds, so a block is never served from the cache it was left in a few steps ago."""
    total = max(push, 1 << 24)
    total = (total + push - 1) // push * push if push <= (1 << 24) else push
    x0 = synth_threaded(base, total, seed=seed, nvfo=nvfo if nvfo else None)
    first = torch.from_numpy(x0.view(np.float32)).to(device)
    del x0
    copies = [first]
    while len(copies) * total * 8 < min_bytes and len(copies) < 8:
        copies.append(torch.roll(first, 2 * 4097 * len(copies)).contiguous())
    bufs = []
    for t in copies:
        flat = t.view(-1)
        for k in range(total // push):
            bufs.append(flat[2 * k * push:2 * (k + 1) * push])
    return bufs, copies


def self_check(torch, np, local, cfg, push, nvfo, bufs, ref_block, data_width=1024, nblocks=3, af=False, group=1):
    """Before anything is timed: the first `nblocks` input blocks through a PIPELINED context (result flags 7) and through an ORDINARY-pass
    context of the same configuration — every VFO block, raw dB line, zoomed line and palette index of every block must be bit-identical
    (the ordinary pass is what the parity tests compare with the oracle at every size; tests/test_bench_geometry_gpu.py compares the
    pipelined mode with the oracle directly).  Returns a dict for the JSON line; raises if the two paths differ."""
    import hashlib

    from sdrplusplus_amd import capi, workloads

    base = 4 if cfg == 5 else cfg
    if group > 1:
        nblocks = 2 * group + 1  # two full launch groups and a block that goes out alone
    out = {"blocks": nblocks, "samples_per_block": push, "blocks_per_launch": group}
    digests = []
    for pipelined in (True, False):
        ctx = capi.Context(local, max_push=push * (group if pipelined else 1))
        info = workloads.setup(ctx, base, dense_fft=True, data_width=data_width, nvfo=nvfo)
        if ref_block and ref_block < push:
            ctx.set_reference_block(ref_block)
        af_keep = []
        if af and nvfo:
            from sdrplusplus_amd import radio
            for vid, (m_, r_, _b, _c, _x) in zip(info["vids"], info["plan"]):
                a_, k_ = radio.af_desc(r_, 48000.0, 50e-6 if m_ == "WFM" else None, m_ == "NFM")
                ctx.vfo_set_af(vid, a_, k_)
                af_keep.append(k_)
        h = hashlib.sha256()
        counts = [0, 0]
        if pipelined:
            ctx.set_pipelined(True, 7)
            if group > 1:
                ctx.set_pipeline_group(group, False)
            for b in range(nblocks):
                ctx.push_device(bufs[b % len(bufs)].data_ptr(), push)
            if group > 1:
                gs = ctx.pipeline_group_stats()
                out["launch_groups"] = {"of_several_blocks": gs["multi_groups"], "largest": gs["largest"]}
                if gs["multi_groups"] < 2 or gs["largest"] != group:
                    raise RuntimeError("self check: the blocks did not go out as launch groups of %d: %r" % (group, gs))
            for t in range(1, nblocks + 1):
                r = ctx.result_wait(t, copy=False)
                for vid in info["vids"]:
                    h.update(np.ascontiguousarray(r["vfo"][vid]).tobytes())
                    counts[0] += len(r["vfo"][vid])
                for k in ("raw", "zoomed", "index"):
                    if r["n_lines"] > 0 and r[k] is not None:
                        h.update(np.ascontiguousarray(r[k]).tobytes())
                counts[1] += int(r["n_lines"])
                ctx.result_release(t)
            st = ctx.pipeline_stats()
            out["blocks_as_ticks"] = st["tick_blocks"]
            out["blocks_as_ordinary_passes"] = st["pass_blocks"]
            out["roles"] = sorted(st["roles"])
        else:
            for b in range(nblocks):
                ctx.push_device(bufs[b % len(bufs)].data_ptr(), push)
                for vid in info["vids"]:
                    a = ctx.vfo_af_read(vid) if (af and nvfo) else ctx.vfo_read(vid)
                    h.update(np.ascontiguousarray(a).tobytes())
                    counts[0] += len(a)
                if ctx.fft_lines() > 0:
                    raw, zo, ix = ctx.fft_read()
                    for a in (raw, zo, ix):
                        h.update(np.ascontiguousarray(a).tobytes())
                    counts[1] += len(raw)
        ctx.close()
        digests.append((h.hexdigest(), counts))
    out["sha256_pipelined"], out["sha256_ordinary"] = digests[0][0][:32], digests[1][0][:32]
    out["vfo_samples_hashed"], out["lines_hashed"] = digests[0][1]
    out["identical"] = digests[0] == digests[1]
    if not out["identical"]:
        raise RuntimeError("pipelined and ordinary-pass results differ: %r" % (digests,))
    return out


def run_workload(torch, np, device, local, cfg, push, mode, steps, warmup, nvfo, world=1, rank=0, af=False, ref_block=None, inputs=None, data_width=1024, lag=8, gather_every=4, exact_ssb=False, check=False,
                 group=1, adaptive=True, regions=1):
    """One measured run of a configuration.  mode 'pipelined': sdrpp_set_pipelined, one launch per block; 'ordinary': one launch per
    stage.  Returns (result dict for rank 0, inputs) — the inputs can be handed to a second run of the same configuration."""
    from sdrplusplus_amd import capi, multi, workloads

    base = 4 if cfg == 5 else cfg
    sr, N = workloads.CFG[base]["sr"], workloads.CFG[base]["fft"]
    stream_index = multi.stream_for_rank(rank, world, world)[0]
    seed0 = multi.stream_seed(0x5D2B0001 if base != 4 else 0, stream_index)  # cfg 5: seeds 0 .. N-1
    if inputs is None or inputs[2] != (base, push, nvfo):
        bufs, keep = make_inputs(torch, np, device, base, push, seed0 * 1000 + 1, nvfo)
        inputs = (bufs, keep, (base, push, nvfo))
    bufs = inputs[0]
    pipelined = mode == "pipelined"
    group = max(1, min(int(group), capi.GROUP_MAX)) if pipelined else 1
    ctx = capi.Context(local, max_push=push * group)  # (a launch group of `group` blocks is planned as one block: the buffers are sized for it)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # one ordering domain with torch / RCCL
    info = workloads.setup(ctx, base, dense_fft=True, data_width=data_width, nvfo=nvfo, exact_ssb=exact_ssb)
    if ref_block is None:
        ref_block = int(sr / 200)
    if ref_block and ref_block < push:
        ctx.set_reference_block(ref_block)
    af_keep = []
    if af and nvfo:
        from sdrplusplus_amd import radio
        for vid, (m_, r_, _b, _c, _x) in zip(info["vids"], info["plan"]):
            a_, k_ = radio.af_desc(r_, 48000.0, 50e-6 if m_ == "WFM" else None, m_ == "NFM")
            ctx.vfo_set_af(vid, a_, k_)
            af_keep.append(k_)
    checked = None
    if check and pipelined and rank == 0 and not exact_ssb:
        checked = self_check(torch, np, local, cfg, push, nvfo, bufs, ref_block, data_width=data_width, af=af, group=1 if af else group)
    if af and nvfo and pipelined:
        lag = max(lag, 13)  # (the AF chain's five levels behind the demodulator: a block's lines and audio are complete 12 launches after its push)
    max_lines = (push + N - 1) // N + 1
    if pipelined:
        ctx.set_pipelined(True, 2)  # zoomed lines + palette indices of every block into page-locked result slots
        if group > 1:
            # up to `group` blocks per launch (include/sdrpp_gpu.h: sdrpp_set_pipeline_group): the pushes keep their tickets and results, consecutive resident
            # blocks (contiguous in HBM) go out as one launch; adaptive = the group follows what is queued on the device
            ctx.set_pipeline_group(group, adaptive)
        lines = torch.zeros((gather_every, max_lines + 1, data_width), dtype=torch.float32, device=device)
    else:
        lines = torch.empty((max(1, push // N), data_width), dtype=torch.float32, device=device)
    runner = multi.StreamRunner(ctx, bufs, push, lines, sync=torch.cuda.synchronize, pipelined=pipelined, lag=lag, gather_every=gather_every)

    for i in range(warmup):
        runner.step(i)
    runner.finish()
    torch.cuda.synchronize()
    if pipelined:
        # a block's results are complete `depth` launches after its push (sdrpp_pipeline_stats): a host that asks for them EARLIER makes
        # sdrpp_result_wait run the queued stages and wait for the device at every step — host work and device work then take turns instead
        # of overlapping (cfg 4: 10 levels against a lag of 8 cost 213 instead of 165 us per step).  SDRPP_RESULT_SLOTS = 24 result slots exist.
        # (with launch groups: `depth` LAUNCHES of up to `group` blocks each; the result slots count launches)
        need = min(capi.RESULT_SLOTS - 2, int(ctx.pipeline_stats()["depth"]) + 1) * group
        if need > runner.lag:
            runner.lag = lag = need
    # calibration (untimed): every kernel family bracketed by HIP events to find the dominant one and record the per-kernel split
    ctx.timing_enable(True)
    ncal = max(2, min(5, steps)) if not pipelined else max(8, min(24, steps))
    for i in range(ncal):
        runner.step(i)
    runner.finish()
    torch.cuda.synchronize()
    fam_all = ctx.timing_read()
    kernel_ms_all = {k: v[0] / ncal for k, v in fam_all.items() if v[0] > 0}
    # dominant family = the longest one on the CRITICAL stream: in an ordinary pass with VFOs the FFT branch runs on a second stream as filler
    # behind the VFO bank (its launches stretch while they wait for CUs, which says nothing about the kernels themselves), so only
    # the VFO-bank families compete there; FFT-only runs (cfg 2) have just the FFT families.  Pipelined: the tick launch is everything.
    filler = {"fft_pass1", "fft_pass2", "fft_single", "zoom_palette"} if nvfo else set()
    cand = {k: v for k, v in kernel_ms_all.items() if k not in filler} or kernel_ms_all
    dom = max(cand, key=cand.get) if cand else None
    # timed region: only the dominant family keeps its event pair (two event records per launch on its launch stream)
    ctx.timing_enable(True, families=[ctx.family_index(dom)] if dom else [])
    blocks0 = runner.collected
    gs0 = ctx.pipeline_group_stats() if pipelined else None
    ticks0 = ctx.pipeline_stats()["ticks"] if pipelined else 0
    # `regions` timed regions of EXACTLY `steps` steps each (median of 5); the first one carries the dominant family's event pairs
    elapsed = runner.timed(steps, first=warmup)
    fam = ctx.timing_read()
    ctx.timing_enable(False)
    gs1 = ctx.pipeline_group_stats() if pipelined else None
    ticks1 = ctx.pipeline_stats()["ticks"] if pipelined else 0
    region_s = [elapsed]
    for q in range(1, max(1, int(regions))):
        region_s.append(runner.timed(steps, first=warmup + q * steps))
    elapsed = sorted(region_s)[len(region_s) // 2]
    # a short timed region (the driver's 20 steps) is a third pipeline fill and drain — depth - 1 launches at either end carry less than a block's
    # work: the same loop over 200 blocks next to it, so that both numbers come from the same process on the same box
    steady = None
    if pipelined and world == 1 and steps < 100:
        st_steps = 200
        st_elapsed = runner.timed(st_steps, first=warmup + steps)
        steady = {"steps": st_steps, "value": round(push * st_steps / st_elapsed / 1e6, 3), "ms_per_step": round(st_elapsed / st_steps * 1e3, 5), "unit": "Msamples/s"}

    # sanity: the work was really done
    if pipelined:
        assert runner.collected - blocks0 == steps * len(region_s) + (steady["steps"] if steady else 0) and not runner.tickets, (runner.collected, blocks0, steps)
        assert runner.gathered is not None or rank != 0
    else:
        assert ctx.fft_lines() == push // N if push % N == 0 else ctx.fft_lines() in (push // N, push // N + 1), (ctx.fft_lines(), push, N)
        if runner.collective and rank == 0:
            assert runner.gathered is not None and tuple(runner.gathered.shape) == (world, max(1, push // N), data_width)
    if base == 3:
        for vid in info["vids"][:1]:
            n_last = ctx.vfo_out_count(vid)  # (the most recent LAUNCH: one block, or a group of up to `group`)
            assert any(abs(n_last - q * push // 40) <= 2 for q in range(1, group + 1)), (n_last, push, group)
    out = None
    if rank == 0:
        total_samples = world * push * steps
        value = total_samples / elapsed / 1e6
        fl, by, bound_of = algorithmic_work(push, info["plan"], sr, nvfo, piped=kernel_ms_all.get("vfo_pipe", 0) > 0, fft_n=N, data_width=data_width)
        pflops, pbytes, bps = path_work(push, info["plan"], sr, nvfo, N)
        fl["tick"], by["tick"], bound_of["tick"] = pflops, pbytes, ("mfma" if nvfo else "hbm")
        launches = {k: max(1, v[1]) for k, v in fam.items()}
        # average duration of one launch set of the dominant family inside the timed region (pipelined: of one tick launch; the few
        # launches that drain the pipeline at the end carry less than a block's work and are part of the average)
        kernel_ms = {k: (v[0] / (launches[k] if k == "tick" else steps)) for k, v in fam.items() if v[0] > 0}
        roof = None
        if dom is not None and dom in kernel_ms and dom in by:
            dur = kernel_ms[dom] * 1e-3
            gbs = by[dom] / dur / 1e9
            traffic = pmc_traffic(base, push, nvfo, dom, af, group=group)
            traffic_block = pmc_traffic(base, push, nvfo, dom, af, group=group, per_block=True) if dom == "tick" else None
            if bound_of.get(dom) == "mfma" and fl.get(dom):
                tf = fl[dom] / dur / 1e12
                # pipelined: a timed region of K blocks holds K + depth - 1 launches (the last ones drain the pipeline and carry less than a block's
                # work), so "one block's flops / mean launch time" over-states short runs; the work-weighted figure — K blocks' flops over the SUM of
                # the launch times — is the same for a 20-step and a 400-step run and is what `frac` reports.  `frac_full_launch` keeps the old formula.
                tf_full = tf
                if dom == "tick" and fam[dom][0] > 0:
                    tf = fl[dom] * steps / (fam[dom][0] * 1e-3) / 1e12
                roof = {"kernel": dom, "bound": "mfma", "achieved": round(tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP32_PEAK_TFLOPS, 5), "traffic": traffic,
                        "frac_full_launch": round(tf_full / FP32_PEAK_TFLOPS, 5), "sum_launch_ms": round(fam[dom][0], 4), "blocks_timed": steps,
                        "algorithmic_flops_per_launch": fl[dom], "algorithmic_bytes_per_launch": by[dom], "hbm_GBps_at_this_rate": round(gbs, 2), "avg_launch_ms": round(kernel_ms[dom], 5),
                        "launches_timed": launches.get(dom), "traffic_per_block": traffic_block,
                        "note": ("tick = the ONE launch per block of pipelined mode: every stage of the path (front end, three decimator / resampler / channel stages, discriminator + audio filter, "
                                 "FFT pass 1 / pass 2, zoom, carries, result copies), each working on a different block; flops / bytes = SURVEY.md 8(d) whole-path figures x samples per block; "
                                 if dom == "tick" else
                                 "family = all launches of this kind in one push (vfo_fir = channel filters + discriminator / audio low-passes; vfo_pipe = the pipelined FM back ends in one launch); ")
                                + "peak = dense FP32 MFMA; traffic = HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic*.json, builder-run: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                  "passes of this workload), null when no pass of this exact workload is committed"}
            else:
                gbs_full = gbs
                if dom == "tick" and fam[dom][0] > 0:
                    gbs = by[dom] * steps / (fam[dom][0] * 1e-3) / 1e9  # work-weighted, as above
                roof = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "frac_full_launch": round(gbs_full / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": by[dom], "avg_launch_ms": round(kernel_ms[dom], 5), "launches_timed": launches.get(dom)}
        roof_fft = {}
        for f in ("fft_pass1", "fft_pass2", "fft_single", "zoom_palette"):
            if f in kernel_ms_all and kernel_ms_all[f] > 0:
                g = by[f] / (kernel_ms_all[f] * 1e-3) / 1e9
                roof_fft[f] = {"bound": "hbm", "achieved": round(g, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(base, push, nvfo, f, af)}
        per_gpu = value * 1e6 / world
        roof_path = {"bound": "hbm", "achieved": round(per_gpu * bps / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(per_gpu * bps / 1e9 / HBM_PEAK_GBS, 5),
                     "algorithmic_bytes_per_sample": bps, "algorithmic_TFLOPs": round(per_gpu * pflops / push / 1e12, 2), "frac_of_fp32_mfma_peak": round(per_gpu * pflops / push / 1e12 / FP32_PEAK_TFLOPS, 5),
                     "note": "whole job, per GPU: SURVEY.md 8(d) path figures x ingest rate (wall clock of the timed region, result delivery included)"}
        mode_names = "/".join(sorted({m for m, _, _, _, _ in info["plan"]})) if nvfo else ""
        out = {
            "value": round(value, 3), "ms_per_step": round(elapsed / steps * 1e3, 5), "steps": steps, "warmup": warmup,
            "timed_regions": {"n": len(region_s), "value": "median", "Msamples_per_s": [round(world * push * steps / t / 1e6, 1) for t in region_s],
                              "note": "every region = exactly `steps` pushes AND the delivery of all their results, bracketed by device synchronisation; roofline.* comes from the first region (the one with HIP events on the launches)"},
            "workload": workload_string(cfg, sr, N, nvfo, mode_names, pipelined),
            "samples_per_step_per_gpu": push, "mode": mode, "reference_block": ref_block if (ref_block and ref_block < push) else push,
            "input_blocks_rotated": len(bufs), "input_bytes_rotated": len(bufs) * push * 8, "af_chain": bool(af and nvfo), "device": ctx.device_info(),
            "roofline": roof, "roofline_fft": roof_fft, "roofline_path": roof_path,
            "kernel_ms_per_step": {k: round(v, 5) for k, v in sorted(kernel_ms_all.items(), key=lambda kv: -kv[1])},
            "realtime_factor": round(per_gpu / sr, 1), "sr": sr, "fft": N, "nvfo": nvfo,
            "nco": ("SSB / DSB / raw channels: the reference's float rotator recursion on the device (nco_mode 2: every channel inside the north-star tolerance against the "
                    "compiled reference at arbitrary offsets); FM / AM channels closed form") if exact_ssb else ("closed form: FM / AM inside the tolerance for any run length; SSB / DSB audio and the raw IF differ from the reference by ITS rotator's rounding drift — "
                                                         "measured at cfg 4 (tests/test_bench_geometry_gpu.py::test_closed_form_nco_validity_window_vs_pinned_oracle): +2.8e-8 .. 1.05e-5 (audio, median 1.5e-6) / 1.0e-7 .. 3.1e-5 (IF, median 4.5e-6) relative per 10^5 input samples once the channel filter has filled; the worst channel leaves 1e-5 after ~2e5 input samples, the median one after ~1.2e6; "
                                                         "3e-7 for any length against the reference with an exact NCO (include/sdrpp_gpu.h, DESIGN.md 5); --nco ssb-exact runs those channels on the reference's own recursion"),
        }
        if pipelined:
            out["results_delivered"] = "zoomed lines + palette indices of every block in page-locked host memory (result flag 2), batches of %d blocks copied to the device%s; VFO outputs stay on the device" % (
                gather_every, " and gathered on rank 0 over RCCL" if world > 1 else "")
            out["result_lag_blocks"] = lag
            st = ctx.pipeline_stats()
            out["pipeline"] = {"ticks": st["ticks"], "blocks_as_ticks": st["tick_blocks"], "blocks_as_ordinary_passes": st["pass_blocks"], "crowded_ticks": st["crowded_ticks"], "depth_levels": st["depth"],
                               "roles": sorted(st["roles"]), "blocks_pushed_by_this_context": ctx.ticket()}
            out["blocks_per_launch"] = {"max": group, "adaptive": bool(adaptive and group > 1),
                                        "first_timed_region": {"blocks": steps, "launches": ticks1 - ticks0, "launch_groups": gs1["groups"] - gs0["groups"],
                                                               "groups_of_several_blocks": gs1["multi_groups"] - gs0["multi_groups"], "blocks_in_those": gs1["multi_blocks"] - gs0["multi_blocks"]},
                                        "note": "sdrpp_set_pipeline_group: consecutive pushes go out as ONE launch (planned as one block whose reference-block ends are the push ends; every push keeps its ticket "
                                                "and its results, bit-identical to one launch per block: self_check); adaptive: a push goes out at once while the device has fewer than two launches in flight"}
        if checked is not None:
            out["self_check"] = checked
        if steady is not None:
            out["steady_state"] = steady
        out["rank_local_s"] = runner.local_elapsed
    else:
        out = {"rank_local_s": runner.local_elapsed}
    ctx.close()
    return out, inputs


def compact(r):
    if r is None:
        return None
    keep = ("value", "ms_per_step", "steps", "samples_per_step_per_gpu", "mode", "reference_block", "kernel_ms_per_step", "realtime_factor", "nco", "blocks_per_launch")
    o = {k: r[k] for k in keep if k in r}
    o["unit"] = "Msamples/s"
    if r.get("roofline"):
        o["roofline"] = {k: r["roofline"][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms") if k in r["roofline"]}
    if r.get("roofline_fft"):
        o["roofline_fft"] = r["roofline_fft"]
    o["roofline_path"] = {k: r["roofline_path"][k] for k in ("achieved", "frac", "algorithmic_TFLOPs", "frac_of_fp32_mfma_peak")}
    return o


def by_push_report(torch, capi, workloads, sr, nvfo, N, group=4):
    """cfg 3 ingest rate against the block size and the way the host drives the C-ABI.  Host buffers are page-locked (sdrpp_host_alloc —
    what the C++ blocks use for their frame-buffer slots) unless marked pageable.  `*_no_read` leave the outputs on the device; every other
    mode DELIVERS all VFO blocks + the lines of every block to the host.  frac = rate x SURVEY.md 8(d) path flops / FP32 matrix peak."""
    import numpy as np

    res = {}
    dev = torch.device("cuda", torch.cuda.current_device())
    for B in (int(sr / 200), STREAM_CAP):
        per_pass = max(1, STREAM_CAP // B)
        # blocks per launch of the grouped legs: what the headline uses at the stream cap, SDRPP_GROUP_MAX = 32 at the reference's block size (a launch of
        # 8 x 50 000 samples is still a fraction of one at 10^6: 13.1 GS/s at 8, 16.4 at 16, 17.3 at 24-32 where the host's 3 us per push take over,
        # profiles/r06t_group_sweep.log)
        G = max(1, min(capi.GROUP_MAX, group if B >= STREAM_CAP else capi.GROUP_MAX))
        ctx = capi.Context(dev.index or 0, max_push=B * max(per_pass, G))
        info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=nvfo)
        vids = info["vids"]
        nb = max(8, 2 * G)  # consecutive blocks of ONE allocation (device and page-locked host): contiguous, so that they can share a launch; the wrap-around starts a new group
        xs = [workloads.synth(3, B, seed=7 + i, nvfo=nvfo) for i in range(nb)]
        pin_base = ctx.L.sdrpp_host_alloc(nb * B * 8)
        ptrs = []
        for i, x in enumerate(xs):
            C.memmove(pin_base + i * B * 8, x.ctypes.data, B * 8)
            ptrs.append(pin_base + i * B * 8)
        xd_all = torch.from_numpy(np.concatenate(xs).view(np.float32)).to(dev)
        xd = [xd_all[2 * i * B:2 * (i + 1) * B] for i in range(nb)]
        entry = {"push": B, "blocks_per_launch_max": G}

        def rate(fn, npush, end=None):
            end = end or ctx.sync
            for i in range(8):
                fn(i)
            end()
            best = 0.0
            for _trial in range(3):
                t0 = time.perf_counter()
                for i in range(npush):
                    fn(i)
                end()
                best = max(best, B * npush / (time.perf_counter() - t0) / 1e6)
            return round(best, 1)

        npush = max(8, min(400, (1 << 26) // B), 40 * G if B < STREAM_CAP else 0)  # (a launch group of G blocks: 40 launches per trial, the pipeline's fill and drain are 7 of them)
        xp = [t.data_ptr() for t in xd]
        # ---- ordinary passes ----
        def sync_pinned(i):
            ctx.push_host_ptr(ptrs[i % nb], B)
            ctx.vfo_read_many(vids)
            ctx.fft_lines()

        def sync_pageable(i):
            ctx.push(xs[i % nb])
            ctx.vfo_read_many(vids)
            ctx.fft_lines()

        entry["per_push_read_pinned"] = rate(sync_pinned, min(npush, 200))
        entry["per_push_read_pageable"] = rate(sync_pageable, min(npush, 200))
        entry["device_no_read"] = rate(lambda i: ctx.push_device(xp[i % nb], B), npush)
        # ---- pipelined: one launch per block (rounds 3-5), then up to G blocks per launch (sdrpp_set_pipeline_group, adaptive) ----
        ctx.set_pipelined(True, 0)
        entry["pipelined_device_no_read_one_block_per_launch"] = rate(lambda i: ctx.push_device(xp[i % nb], B), npush)
        entry["pipelined_pinned_no_read_one_block_per_launch"] = rate(lambda i: ctx.push_host_ptr_async(ptrs[i % nb], B), npush)
        ctx.set_pipeline_group(G, True)
        entry["pipelined_device_no_read"] = rate(lambda i: ctx.push_device(xp[i % nb], B), npush)
        entry["pipelined_pinned_no_read"] = rate(lambda i: ctx.push_host_ptr_async(ptrs[i % nb], B), npush)
        gs = ctx.pipeline_group_stats()
        entry["pipelined_blocks_per_launch_seen"] = round(gs["multi_blocks"] / max(1, gs["multi_groups"]), 2)
        ctx.set_pipeline_group(1, False)
        ctx.set_pipelined(False)
        ctx.set_pipelined(True, 3)  # every VFO block + zoomed lines / palette indices of every block into page-locked result slots
        lag = 8
        state = {"next": ctx.ticket() + 1}

        def collect(upto):
            while state["next"] <= upto:
                t = C.c_uint64(state["next"])
                r = capi.Result()
                ctx._chk(ctx.L.sdrpp_result_wait(ctx.h, t, C.byref(r)))
                ctx._chk(ctx.L.sdrpp_result_release(ctx.h, t))
                state["next"] += 1

        def with_results(i):
            ctx.push_host_ptr_async(ptrs[i % nb], B)
            collect(ctx.ticket() - lag)

        def with_results_pageable(i):
            ctx.push(xs[i % nb])
            collect(ctx.ticket() - lag)

        entry["pipelined_pinned_results_delivered_one_block_per_launch"] = rate(with_results, npush, lambda: collect(ctx.ticket()))
        entry["pipelined_result_lag_blocks_one_block_per_launch"] = lag
        # the same with launch groups: a block's results are complete `depth` LAUNCHES after its group went out — the host asks (depth + 1) * G blocks behind
        ctx.set_pipeline_group(G, True)
        lag = min(capi.RESULT_SLOTS - 2, int(ctx.pipeline_stats()["depth"]) + 1) * G

        def with_results_device(i):
            ctx.push_device(xp[i % nb], B)
            collect(ctx.ticket() - lag)

        entry["pipelined_pinned_results_delivered"] = rate(with_results, npush, lambda: collect(ctx.ticket()))
        entry["pipelined_pageable_results_delivered"] = rate(with_results_pageable, npush, lambda: collect(ctx.ticket()))
        entry["pipelined_device_results_delivered"] = rate(with_results_device, npush, lambda: collect(ctx.ticket()))
        entry["pipelined_result_lag_blocks"] = lag
        ctx.set_pipeline_group(1, False)
        ctx.set_pipelined(False)
        # ---- deferred: many blocks staged, one ordinary pass ----
        ctx.set_deferred(True)

        def deferred_pass(i, pinned=True):
            for k in range(per_pass):
                if pinned:
                    ctx.push_host_ptr_async(ptrs[(i + k) % nb], B)  # page-locked source: the device fetches the block, one wait per pass
                else:
                    ctx.push(xs[(i + k) % nb])
            if pinned:
                ctx.push_wait()
            ctx.vfo_read_many(vids)
            ctx.fft_lines()

        if per_pass > 1:
            entry["deferred_pushes_per_pass"] = per_pass
            entry["deferred_read_pinned"] = round(rate(lambda i: deferred_pass(i, True), max(3, min(npush, 200) // per_pass)) * per_pass, 1)
            entry["deferred_read_pageable"] = round(rate(lambda i: deferred_pass(i, False), max(3, min(npush, 200) // per_pass)) * per_pass, 1)
        ctx.set_deferred(False)
        pflops, _pb, _bps = path_work(B, info["plan"], sr, nvfo, N)
        entry["frac_of_fp32_mfma_peak"] = {k: round(v * 1e6 * pflops / B / 1e12 / FP32_PEAK_TFLOPS, 5) for k, v in entry.items() if isinstance(v, float) and k not in ("push",)}
        tr = pmc_traffic(3, B, nvfo, "tick", group=G, per_block=True)  # pipelined mode, device-resident blocks (committed PMC passes of exactly this workload, or null)
        entry["pipelined_tick_traffic_bytes_per_block"] = tr
        entry["pipelined_tick_traffic_over_algorithmic"] = round(tr / _pb, 2) if tr else None
        ctx.L.sdrpp_host_free(pin_base)
        ctx.close()
        res["B=%d" % B] = entry
    # through the C++ host mirror (source thread -> dsp::stream -> IQFrontEnd::run -> one sink thread per VFO), reference block size
    try:
        with tempfile.TemporaryDirectory() as tmp:
            # built against the REFERENCE's own dsp/stream.h + dsp/block.h where the reference tree was present at build time (oracle/Makefile:
            # _ref/bench_blocks_ref — the binary travels to the GPU box like the product library); the test doubles only if that binary is missing
            exe = os.path.join(ROOT, "oracle", "_ref", "bench_blocks_ref")
            against = "the reference's dsp/stream.h and dsp/block.h (oracle/_ref/bench_blocks_ref, compiled from /root/reference by oracle/Makefile)"
            if not os.path.exists(exe):
                exe = os.path.join(tmp, "bench_blocks")
                against = "the TEST DOUBLES of dsp/stream.h and dsp/block.h (tests/host_cpp/standalone: oracle/_ref/bench_blocks_ref was not built)"
                csrc = os.path.join(ROOT, "sdrplusplus_amd", "csrc")
                subprocess.run(["g++", "-std=c++17", "-O2", "-w", "-o", exe, os.path.join(ROOT, "tests", "host_cpp", "bench_blocks.cpp"), "-I" + os.path.join(ROOT, "tests", "host_cpp", "standalone"),
                                "-L" + csrc, "-lsdrpp_gpu", "-Wl,-rpath," + csrc, "-lpthread"], check=True, capture_output=True)
            res["cpp_iqfrontend_built_against"] = against
            # The GPU box is a two-socket host (2 x 64 cores, 256 hardware threads): whether the harness's source thread, the front end's worker and the 32
            # sink threads land on the socket the GPU hangs off decides the figure by a factor of 1.7 (profiles/r06d_seam_prof.log: a stream hand-over across
            # the sockets costs the worker 30 us per block of waiting).  The C++ legs therefore run on the cores of the GPU's NUMA node, as a deployment
            # that cares would start its host process (numactl --cpunodebind); `cpp_iqfrontend_cpus` says which.
            cpus = gpu_numa_cpus(torch)
            res["cpp_iqfrontend_cpus"] = ("%d hardware threads of NUMA node %s (the GPU's)" % (len(cpus[1]), cpus[0])) if cpus else "not pinned (no NUMA information)"
            pre = (lambda: os.sched_setaffinity(0, cpus[1])) if cpus else None
            for name, buffered, pipelined, reps, grp in (("bypass_pipelined", 0, 1, 5, 1), ("bypass_pipelined_launch_groups", 0, 1, 5, 8), ("bypass_per_block", 0, 0, 1, 1), ("buffered", 1, 0, 1, 1)):
                runs = []  # (the pipelined figure depends on how the host schedules 34 threads: five runs, median reported, min / max and all five listed)
                for _ in range(reps):
                    r = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin"), str(sr), str(int(sr / 200)), str(N), str(nvfo), "2" if reps > 1 else "3", str(buffered), str(pipelined), str(grp)],
                                       capture_output=True, text=True, timeout=120, preexec_fn=pre)
                    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    runs.append(json.loads(line[-1]) if line else {"error": (r.stdout + r.stderr)[-300:]})
                good = sorted((q for q in runs if "msps" in q), key=lambda q: q["msps"])
                entry = dict(good[len(good) // 2]) if good else runs[-1]
                if len(good) > 1:
                    entry["msps_runs"] = [q["msps"] for q in good]
                    entry["msps_min"], entry["msps_max"] = good[0]["msps"], good[-1]["msps"]
                res["cpp_iqfrontend_run_%s" % name] = entry
    except Exception as e:
        res["cpp_iqfrontend_run"] = {"error": repr(e)[:300]}
    res["note"] = ("Msamples/s; per_push_read = sdrpp_push (host pointer, H2D included) + sdrpp_vfo_read_many + sdrpp_fft_lines after EVERY push (ordinary pass); pipelined_* = sdrpp_set_pipelined + "
                   "sdrpp_set_pipeline_group(blocks_per_launch_max, adaptive): consecutive blocks share a launch while the device has two launches in flight (`*_one_block_per_launch`: the mode of rounds 3-5); "
                   "*_no_read leave the outputs on the device, *_results_delivered deliver EVERY VFO's block + zoomed lines + palette indices of every push into page-locked result slots (pinned / pageable: the "
                   "block is also fetched from host memory; device: resident in HBM), collected with sdrpp_result_wait / _release `pipelined_result_lag_blocks` blocks behind the push; deferred_read = sdrpp_set_deferred: pushes staged, one ordinary pass + one read per "
                   "`deferred_pushes_per_pass` pushes; cpp_iqfrontend_run = tests/host_cpp/bench_blocks.cpp (SpeedTester-style source thread, one sink thread per VFO) through sdrpp_gpu::IQFrontEnd: bypass_pipelined = "
                   "one block per launch, results handed to the streams a few blocks late; bypass_per_block = one ordinary pass per block; buffered = 32-slot frame buffer worked off as deferred passes.  "
                   "`cpp_iqfrontend_built_against` says which dsp/stream.h the C++ legs were compiled with")
    return res


def gpu_numa_cpus(torch):
    """(node, set of cpus) of the NUMA node the current GPU is attached to, None if the host does not say."""
    try:
        p = torch.cuda.get_device_properties(torch.cuda.current_device())
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        return (node, cpus) if len(cpus) >= 8 else None
    except Exception:
        return None


def dry_launch(args, np, torch):
    """`--gpus N --dry-launch` on a machine without GPUs: everything of the N > 1 launch that does not need a device — the ranks spawned by
    the launcher branch of main(), the rendezvous (gloo instead of RCCL), the per-rank device naming, multi.StreamRunner's pipelined protocol
    (barrier, K steps, delivery, MAX over ranks) on a stub context, the per-rank rates and rank 0's ONE JSON line."""
    import torch.distributed as dist

    from sdrplusplus_amd import multi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("gloo")
    names = [None] * world
    dist.all_gather_object(names, "stub device (rank %d, pid %d)" % (rank, os.getpid()))
    push, width, max_lines, every = 1000, 16, 2, 4

    class Stub:
        """What StreamRunner needs of a context: tickets, result slots with a line count that depends on the block and the rank."""

        def __init__(self):
            self.n = 0

        def push_device(self, ptr, count):
            self.n += 1

        def ticket(self):
            return self.n

        def result_wait(self, t, copy=False):
            nl = 1 + (t + rank) % 2
            return {"n_lines": nl, "zoomed": np.full((nl, width), float(1000 * rank + t), np.float32)}

        def result_release(self, t):
            pass

    lines = torch.zeros((every, max_lines + 1, width), dtype=torch.float32)
    runner = multi.StreamRunner(Stub(), [torch.zeros(2 * push)], push, lines, pipelined=True, lag=8, gather_every=every)
    for i in range(args.warmup):
        runner.step(i)
    runner.finish()
    elapsed = runner.timed(args.steps, first=args.warmup)
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "local_s": runner.local_elapsed, "collected": runner.collected})
    ok = True
    if rank == 0:
        g = runner.gathered
        ok = g is not None and tuple(g.shape) == (world, every, max_lines + 1, width) and all(p["collected"] == args.steps + args.warmup for p in per_rank)
        from sdrplusplus_amd import workloads

        cfg = default_cfg(args)  # the SAME choice main() makes with devices: the metric and the workload do not depend on the number of GPUs
        cbase = 4 if cfg == 5 else cfg
        cnv = 0 if cbase == 2 else (args.nvfo if cbase == 3 else 128)
        names_ = "/".join(sorted({m for m, _, _, _, _ in workloads.vfo_plan(cbase, cnv)})) if cnv else ""
        print(json.dumps({"dry_launch": True, "ok": bool(ok), "metric": METRIC[cfg],
                          "config": {"workload": workload_string(cfg, workloads.CFG[cbase]["sr"], workloads.CFG[cbase]["fft"], cnv, names_, args.mode == "pipelined"), "streams": world},
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "backend": dist.get_backend(), "devices": names,
                          "value": round(world * push * args.steps / elapsed / 1e6, 3), "unit": "Msamples/s (stub context: the protocol, not the hot path)",
                          "per_rank": [{"rank": p["rank"], "Msamples_per_s": round(push * args.steps / max(p["local_s"], 1e-9) / 1e6, 1)} for p in per_rank],
                          "gathered_shape": list(g.shape) if g is not None else None}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def af_sr200_delivered(torch, capi, workloads, sr, nvfo):
    """cfg 3 + the AF chain on every VFO at the reference's block size (sr / 200), pipelined: every block fetched from page-locked host memory,
    every VFO's 48 kHz AF block + zoomed lines + palette indices delivered into page-locked result slots 13 blocks behind the push."""
    import numpy as np

    from sdrplusplus_amd import radio

    B, lag = int(sr / 200), 13
    dev = torch.device("cuda", torch.cuda.current_device())
    ctx = capi.Context(dev.index or 0, max_push=B)
    info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=nvfo)
    keep = []
    for vid, (m_, r_, _b, _c, _x) in zip(info["vids"], info["plan"]):
        a_, k_ = radio.af_desc(r_, 48000.0, 50e-6, False)
        ctx.vfo_set_af(vid, a_, k_)
        keep.append(k_)
    nb = 4
    ptrs = []
    for i in range(nb):
        x = workloads.synth(3, B, seed=7 + i, nvfo=nvfo)
        p = ctx.L.sdrpp_host_alloc(B * 8)
        C.memmove(p, x.ctypes.data, B * 8)
        ptrs.append(p)
    ctx.set_pipelined(True, 3)
    state = {"next": ctx.ticket() + 1}

    def collect(upto):
        while state["next"] <= upto:
            t = C.c_uint64(state["next"])
            r = capi.Result()
            ctx._chk(ctx.L.sdrpp_result_wait(ctx.h, t, C.byref(r)))
            ctx._chk(ctx.L.sdrpp_result_release(ctx.h, t))
            state["next"] += 1

    def step(i):
        ctx.push_host_ptr_async(ptrs[i % nb], B)
        collect(ctx.ticket() - lag)

    for i in range(40):
        step(i)
    collect(ctx.ticket())
    best = 0.0
    npush = 400
    for _trial in range(3):
        t0 = time.perf_counter()
        for i in range(npush):
            step(i)
        collect(ctx.ticket())
        best = max(best, B * npush / (time.perf_counter() - t0) / 1e6)
    st = ctx.pipeline_stats()
    for p in ptrs:
        ctx.L.sdrpp_host_free(p)
    ctx.close()
    return {"value": round(best, 1), "unit": "Msamples/s", "push": B, "result_lag_blocks": lag, "blocks_as_ordinary_passes": st["pass_blocks"], "depth_levels": st["depth"]}


def default_cfg(args):
    """The configuration a command line means: --cfg wins, then --fft-only, else the headline workload (cfg 3) at EVERY number of GPUs."""
    return args.cfg if args.cfg in (2, 3, 4, 5) else (2 if args.fft_only else 3)


def workload_string(cfg, sr, N, nvfo, mode_names, pipelined):
    """config.workload of the JSON line — one function so that the 1-GPU line, the N-GPU line and the dry launch say the same thing."""
    return "cfg%d: %.2f MS/s-format synthetic IQ (workloads.synth), %d-pt dense FFT + log-power waterfall%s%s" % (
        cfg, sr / 1e6, N, (" + %d VFO x %s (xlate+FIR+resample+demod)" % (nvfo, mode_names)) if nvfo else "",
        ("; zoomed lines + palette indices of every block DELIVERED to page-locked host memory" + (", VFO outputs left in HBM" if nvfo else "")) if pipelined else "; outputs left in HBM")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--push", type=int, default=STREAM_CAP, help="complex samples per step = per block handed to the hot path (default: the dsp::stream cap, 10^6)")
    ap.add_argument("--mode", choices=("pipelined", "ordinary"), default="pipelined", help="pipelined: one launch per block (sdrpp_set_pipelined); ordinary: one launch per stage")
    ap.add_argument("--group", type=int, default=4, help="pipelined mode: blocks one launch may carry (sdrpp_set_pipeline_group; 1 = one launch per block, as in rounds 3-5)")
    ap.add_argument("--group-fixed", action="store_true", help="always wait for --group blocks (default: adaptive — a push goes out at once while the device has fewer than two launches in flight)")
    ap.add_argument("--regions", type=int, default=5, help="timed regions of --steps steps each; value = their median (SURVEY.md 8d)")
    ap.add_argument("--ref-block", type=int, default=-1, help="reference block inside a push (default sr/200, what the file source cuts; 0: the push is one block)")
    ap.add_argument("--nvfo", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-by-push", action="store_true")
    ap.add_argument("--no-self-check", action="store_true", help="skip the untimed comparison of the pipelined path with ordinary passes on the first blocks")
    ap.add_argument("--no-others", action="store_true", help="skip the ceiling and the cfg 2 / cfg 4 runs")
    ap.add_argument("--af", action="store_true", help="also run the radio AF chain (resampler to 48 kHz + 50 us de-emphasis) behind every VFO (SURVEY.md 8f row 1; not part of the headline workload; pipelined like everything else)")
    ap.add_argument("--fft-only", action="store_true", help="same as --cfg 2")
    ap.add_argument("--nco", choices=("closed", "ssb-exact"), default="closed", help="ssb-exact: SSB / DSB / raw channels on the reference's float rotator recursion (sdrpp_vfo_desc.nco_mode = 2): a role of the tick that bounds it (~26 cycles per sample), every channel stays pipelined")
    ap.add_argument("--cfg", type=int, default=0, help="BASELINE config 2 / 3 / 4 / 5 (default: 3, the headline workload, on ANY number of GPUs — one stream per GPU; 5 = a cfg-4 stream per GPU)")
    ap.add_argument("--dry-launch", action="store_true", help="N > 1 without GPUs: spawn the ranks, rendezvous over gloo, run the StreamRunner protocol on a stub context, print the JSON line (CPU test of the launch path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started like the one-GPU run (`python bench.py --gpus N`): become the launcher — one rank per GPU through torch.distributed.run, the
        # same command line; rank 0's JSON line is the last line of this process's stdout as well
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch

    if args.dry_launch:
        return dry_launch(args, np, torch)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    rccl = None
    if world > 1 or os.environ.get("SDRPP_BENCH_FORCE_RCCL"):
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)  # backend "nccl" is RCCL on ROCm
        names = [None] * world
        dist.all_gather_object(names, "%s (rank %d, cuda:%d)" % (torch.cuda.get_device_name(local), rank, local))
        rccl = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "devices": names}
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))

    from sdrplusplus_amd import capi, workloads

    # ONE metric across N: the headline workload (cfg 3) on every rank's own stream whatever the number of GPUs, so that the driver's 1/2/4/8 curve
    # scales one workload; cfg 5 (a cfg-4 stream per rank) only when asked for (--cfg 5).  The reference's graph is one independent IQFrontEnd
    # per stream (core/src/signal_path/iq_frontend.cpp:140-183).
    cfg = default_cfg(args)
    base = 4 if cfg == 5 else cfg
    sr, N = workloads.CFG[base]["sr"], workloads.CFG[base]["fft"]
    nvfo = 0 if base == 2 else (args.nvfo if base == 3 else 128)
    push = int(args.push)
    mode = args.mode
    if mode == "ordinary":
        push = max(1, push // N) * N  # whole frames per step (the ordinary protocol copies a fixed number of lines)
    ref_block = None if args.ref_block < 0 else args.ref_block
    head, inputs = run_workload(torch, np, device, local, cfg, push, mode, args.steps, args.warmup, nvfo, world=world, rank=rank, af=args.af, ref_block=ref_block, exact_ssb=args.nco == "ssb-exact",
                                check=not args.no_self_check, group=args.group, adaptive=not args.group_fixed, regions=args.regions)
    per_rank = None
    if dist is not None:  # every rank's own rate (its K blocks over its own wall time, before it waits for the slowest rank)
        mine = {"rank": rank, "device": "cuda:%d" % local, "local_s": head["rank_local_s"]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    out = {
        "metric": METRIC[cfg], "value": head["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": head["workload"], "samples_per_step_per_gpu": push,
                   "mode": head["mode"] + ((" (sdrpp_set_pipelined + sdrpp_set_pipeline_group: up to %d consecutive blocks per launch%s, every block with its own ticket and results, collected %d blocks behind the push)"
                                            % (args.group, "" if args.group_fixed else ", fewer while the device would run dry", head.get("result_lag_blocks", 0))) if (mode == "pipelined" and args.group > 1) else
                                           (" (sdrpp_set_pipelined: one launch per block, results %d blocks late)" % head.get("result_lag_blocks", 0) if mode == "pipelined" else "")),
                   "reference_block": head["reference_block"], "streams": world, "parallelism": "one independent IQ stream per GPU" + ("; RCCL gather of zoomed waterfall lines to rank 0" if world > 1 else ""),
                   "input_blocks_rotated": head["input_blocks_rotated"], "input_bytes_rotated": head["input_bytes_rotated"], "af_chain": head["af_chain"], "device": head["device"],
                   "results_delivered": head.get("results_delivered"), "nco": head["nco"]},
        "roofline": head["roofline"], "roofline_fft": head["roofline_fft"], "roofline_path": head["roofline_path"],
        "kernel_ms_per_step": head["kernel_ms_per_step"],
        "kernel_ms_note": "per-family HIP-event times from an untimed calibration pass; pipelined mode has ONE family (tick: the launch per block); ordinary passes: the FFT branch and the VFO bank run on two "
                          "streams and overlap, so the entries sum to more than ms_per_step; roofline.avg_launch_ms is the dominant family re-measured inside the timed region",
        "realtime_factor": head["realtime_factor"],
    }
    if rccl:
        out["rccl"] = rccl
    if per_rank:
        out["per_rank"] = [{"rank": r["rank"], "device": r["device"], "Msamples_per_s": round(push * args.steps / r["local_s"] / 1e6, 1) if r.get("local_s") else None} for r in per_rank]
    for k in ("pipeline", "blocks_per_launch", "timed_regions", "self_check", "steady_state"):
        if head.get(k) is not None:
            out[k] = head[k]
    del inputs
    torch.cuda.empty_cache()
    if world == 1 and not args.no_others:
        others = {}
        try:  # last round's headline geometry: ONE ordinary pass over 2^24 samples (1.7 s of the stream — more than a dsp::stream can hand over)
            r, inp = run_workload(torch, np, device, local, cfg, max(1, (1 << 24) // N) * N, "ordinary", 10, 2, nvfo, af=args.af, ref_block=0)
            out["ceiling"] = compact(r)
            out["ceiling"]["note"] = "one ordinary pass over 2^24 samples per step (not API-legal through dsp::stream: cap 10^6); roofline = dominant kernel family of the pass"
            del inp
        except Exception as e:
            out["ceiling"] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
        for oc in (2, 4):
            if oc == cfg:
                continue
            try:
                ocN = workloads.CFG[oc]["fft"]
                ocv = 0 if oc == 2 else 128
                r1, inp = run_workload(torch, np, device, local, oc, STREAM_CAP, "pipelined", 200, 14, ocv, group=args.group, adaptive=not args.group_fixed)  # (200 steps: fill and drain of a 5 .. 12 level pipeline are 3-6 % of the region, not 10-20 %)
                del inp
                torch.cuda.empty_cache()
                r2, inp = run_workload(torch, np, device, local, oc, max(1, (1 << 24) // ocN) * ocN, "ordinary", 5, 2, ocv, ref_block=0)
                del inp
                torch.cuda.empty_cache()
                others["cfg%d" % oc] = {"workload": r1["workload"], "pipelined_stream_cap": compact(r1), "ceiling_2p24_ordinary": compact(r2)}
                if oc == 2 and cfg == 3:  # the headline workload with the radio module's AF chain behind every VFO (radio_module.h:98-110: the AF resampler is always on)
                    ra, inp = run_workload(torch, np, device, local, 3, STREAM_CAP, "pipelined", 200, 14, nvfo, af=True, group=args.group, adaptive=not args.group_fixed)
                    del inp
                    torch.cuda.empty_cache()
                    others["cfg3_af"] = {"workload": ra["workload"] + " + AF chain (resampler to 48 kHz, 50 us de-emphasis) on every VFO", "pipelined_stream_cap": compact(ra),
                                         "sr200_pinned_results_delivered": af_sr200_delivered(torch, capi, workloads, sr, nvfo)}
                if oc == 4:  # the setting in which cfg 4's SSB channels follow the reference's own rotator (parity at arbitrary offsets): chain-bound
                    # (pipelined since round 5: the rotator recursion is a role of the tick — TR_ROTX16 — and bounds it; the other stages run in its shadow)
                    r3, inp = run_workload(torch, np, device, local, oc, STREAM_CAP, "pipelined", 24, 4, ocv, exact_ssb=True)
                    del inp
                    torch.cuda.empty_cache()
                    # cfg 4 LEADS with the figure at which EVERY channel is inside the north-star tolerance of the compiled reference for any run length; the
                    # closed-form figure holds for NFM / AM always and for the 42 USB channels only during the first 2e5 input samples of a VFO
                    # (profiles/r06_rotator_drift.md: no calibration of the closed form widens that window to 1e7 samples)
                    fast = others["cfg4"]
                    others["cfg4"] = {
                        "workload": fast["workload"],
                        "parity_true": dict(compact(r3), what="SSB channels on the reference's own rotator recursion (sdrpp_vfo_desc.nco_mode = 2, a role of the tick that bounds it): every one of the 128 channels "
                                                               "within 1e-5 of the compiled reference for any run length (tests/test_full_configs_gpu.py::test_cfg4_all_128_vfos_every_mode_within_1e5[reference_rotator])"),
                        "closed_form_nco": dict(fast["pipelined_stream_cap"], what="default NCO: NFM / AM within 1e-5 of the reference for any run length; the 42 USB channels within 1e-5 for the first 2e5 input samples of a "
                                                                                   "VFO, afterwards only against the reference with an exact NCO (the reference rotator's rounding drift is not reproduced)"),
                        "ceiling_2p24_ordinary": fast["ceiling_2p24_ordinary"],
                    }
            except Exception as e:
                others["cfg%d" % oc] = {"error": repr(e)[:300]}
        out["other_configs"] = others
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(base, nvfo, N)
            if out["cpu_baseline"]:
                out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
                if "value" in out["cpu_baseline"].get("process_only", {}):
                    out["gpu_over_cpu_process_only"] = round(out["value"] / out["cpu_baseline"]["process_only"]["value"], 1)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU result
            out["cpu_baseline"] = {"error": repr(e)}
    if world == 1 and base == 3 and not args.no_by_push:
        try:
            out["by_push"] = by_push_report(torch, capi, workloads, sr, nvfo, N, group=args.group)
        except Exception as e:
            out["by_push"] = {"error": repr(e)[:400]}
        # `value` is quoted on 10^6-sample blocks resident in HBM with the lines delivered: the ceiling at the boundary.  What a host sees at the block size the
        # reference's sources produce (sample_rate / 200, file_source/src/main.cpp:157) goes into `config` as well, so that the line cannot be read without it
        bp = out["by_push"]
        k200 = "B=%d" % int(sr / 200)
        if isinstance(bp.get(k200), dict):
            cpp = bp.get("cpp_iqfrontend_run_bypass_pipelined", {})
            out["config"]["at_the_reference_block_size_sr_200"] = {
                "block": int(sr / 200), "unit": "Msamples/s",
                "device_resident_outputs_left_in_HBM": bp[k200].get("pipelined_device_no_read"),
                "device_resident_every_VFO_block_and_lines_delivered": bp[k200].get("pipelined_device_results_delivered"),
                "host_fed_page_locked_every_VFO_block_and_lines_delivered": bp[k200].get("pipelined_pinned_results_delivered"),
                "cpp_IQFrontEnd_run_seam_median_of_5": cpp.get("msps"), "cpp_IQFrontEnd_run_seam_min_max": [cpp.get("msps_min"), cpp.get("msps_max")],
                "cpp_IQFrontEnd_run_seam_launch_groups_of_8_median_of_5": bp.get("cpp_iqfrontend_run_bypass_pipelined_launch_groups", {}).get("msps"),
                "cpp_seam_built_against": bp.get("cpp_iqfrontend_built_against"), "cpp_seam_cpus": bp.get("cpp_iqfrontend_cpus"),
            }
            k1m = "B=%d" % STREAM_CAP
            if isinstance(bp.get(k1m), dict):
                out["config"]["at_the_stream_cap_every_VFO_block_and_lines_delivered"] = {"device_resident": bp[k1m].get("pipelined_device_results_delivered"), "host_fed_page_locked": bp[k1m].get("pipelined_pinned_results_delivered"), "unit": "Msamples/s"}
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
