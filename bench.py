#!/usr/bin/env python3
"""bench.py — BASELINE.json headline metric on MI355X.

Workload (config.workload = "cfg3"): 10 MS/s synthetic IQ, 65536-pt Nuttall FFT + log-power + waterfall line (dense
framing: every sample transformed) AND 32 VFOs x WFM (frequency translation, 8/2/2 decimating FIR cascade, 4/5 polyphase
resampler, 126-tap channel filter, FM discriminator, 237-tap audio low-pass), all inside one `sdrpp_push_device` call.
A "step" = one pass of that hot path over one batch of `--push` complex samples already resident in HBM.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
Multi-GPU: independent IQ streams, one per GPU (weak scaling); the only exchange is the RCCL gather of the finished
(zoomed) waterfall lines to rank 0, inside the timed region.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector


def make_input(torch, n, nvfo, sr, seed, device):
    """cfg 3 signal generated on the device (float64 phase, rounded once): 8 tones + AWGN + nvfo FM carriers."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.arange(n, device=device, dtype=torch.float64)
    x = torch.complex(torch.randn(n, generator=g, device=device, dtype=torch.float32), torch.randn(n, generator=g, device=device, dtype=torch.float32)).to(torch.complex128) * 1e-4
    fr = [0.0625, -0.125, 0.20001, -0.3123, 0.4101, -0.0417, 0.3333, -0.4499]
    amp = [0.1, 0.03, 0.01, 3e-3, 1e-3, 3e-4, 1e-4, 1e-5]
    two_pi = 6.283185307179586
    for f, a in zip(fr, amp):
        x += a * torch.polar(torch.ones_like(t), two_pi * f * t)
    for k in range(nvfo):
        carrier = (k - (nvfo - 1) / 2.0) * 300e3
        tone = 400.0 + 50.0 * k
        ph = two_pi * (carrier / sr) * t + (75e3 / tone) * torch.sin(two_pi * (tone / sr) * t)
        x += 0.05 * torch.polar(torch.ones_like(t), ph)
    return x.to(torch.complex64).contiguous()


def cpu_baseline(sr, nvfo, fft_size, block):
    """Reference code (oracle/_ref, reference headers + restated VOLK/FFTW) timed on this host: same cfg-3 workload,
    bounded sample sized for roughly 10-20 s of CPU work."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import support as S

    lib = S.ref(fast=True)
    kind = "reference"
    if lib is None:
        return None
    from sdrplusplus_amd import workloads

    cores = min(os.cpu_count() or 1, max(nvfo, 1))  # one worker per VFO at most; thread 0 also runs the FFT branch
    offs = np.array([c for _, _, _, c, _ in workloads.vfo_plan(3, nvfo)], dtype=np.float64)
    offs_p = offs.ctypes.data_as(C.POINTER(C.c_double))
    n0 = block * 40  # 2 M samples, streamed `repeat` times
    x = workloads.synth(3, n0, seed=21, nvfo=nvfo)
    xp = S._fp(x.view(np.float32))
    lib.ref_bench_cfg3(xp, block * 2, block, sr, nvfo, offs_p, fft_size, cores, 1)  # warm caches / tables
    t = lib.ref_bench_cfg3(xp, n0, block, sr, nvfo, offs_p, fft_size, cores, 1)
    repeat = int(max(1, min(400, round(15.0 / max(t, 1e-3)))))  # ~15 s of CPU time
    t = lib.ref_bench_cfg3(xp, n0, block, sr, nvfo, offs_p, fft_size, cores, repeat)
    n0 = n0 * repeat
    rate = n0 / t
    return {
        "value": round(rate / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": kind,
        "sample": "%d samples of cfg3 (%d VFO x WFM via the reference's RxVFO::process + BroadcastFM::process, %d-pt FFT + log-power per %d "
                  "samples) in blocks of %d, %d worker threads over VFOs, reference headers compiled -O3 -march=native against the restated "
                  "VOLK (vectorised dot products) / FFT shim — genuine libvolk/libfftw3f are not installed; %.1f s of CPU time"
                  % (n0, nvfo, fft_size, fft_size, block, cores, t),
    }


# kernel family (sdrpp_kernel_family_name) -> kernel-name prefixes as rocprofv3 reports them
FAMILY_KERNELS = {
    "fft_pass1": ["fft_pass1_kernel"], "fft_pass2": ["fft_pass2_kernel"], "fft_single": ["fft_single_kernel"], "zoom_palette": ["zoom_palette_kernel"],
    "vfo_stage1": ["vfo_frontcm_kernel", "vfo_front2_kernel", "vfo_stage1_kernel", "vfo_stage1_direct_kernel", "vfo_rotate_kernel"],
    "vfo_decim": ["vfo_toep_kernel<2, 2, false>", "vfo_firb_kernel<2, false, false>"], "vfo_poly": ["vfo_toep_kernel<2, 2, false>", "vfo_polyb_kernel", "vfo_poly_kernel"],
    "vfo_fir": ["vfo_toep_kernel<1, 2, true>", "vfo_toep_kernel<2, 2, false>", "vfo_firb_kernel"], "demod": ["vfo_demod_pre_kernel", "vfo_sequential_kernel"], "carry_misc": ["carry_kernel"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--push", type=int, default=1 << 24, help="complex samples per step (multiple of the FFT size); 2^24 = 1.7 s of the 10 MS/s stream per GPU and step")
    ap.add_argument("--nvfo", type=int, default=32)
    ap.add_argument("--nbuf", type=int, default=4, help="distinct input batches rotated through (4 x 128 MiB at the default push: never resident in the 256 MiB MALL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--af", action="store_true", help="also run the radio AF chain (resampler to 48 kHz + 50 us de-emphasis) behind every VFO (SURVEY.md 8f row 1; not part of the headline workload)")
    ap.add_argument("--fft-only", action="store_true", help="BASELINE cfg2 (no VFOs) instead of cfg3")
    ap.add_argument("--cfg", type=int, default=0, help="explicit BASELINE config: 2 (FFT only), 3 (headline), 4 (61.44 MS/s, 128 mixed VFOs, 2^20-pt FFT)")
    args = ap.parse_args()

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

    from sdrplusplus_amd import capi, workloads

    cfg = args.cfg if args.cfg in (2, 3, 4) else (2 if args.fft_only else 3)
    sr, N = workloads.CFG[cfg]["sr"], workloads.CFG[cfg]["fft"]
    push = max(1, args.push // N) * N
    nvfo = 0 if cfg == 2 else (args.nvfo if cfg == 3 else 128)
    if cfg == 4:
        import numpy as np
        base = torch.from_numpy(workloads.synth(4, push, seed=0x5D2B + rank)).to(device)  # numpy generator (mixed NFM/AM/USB carriers)
        bufs = [torch.roll(base, 4097 * b).contiguous() for b in range(args.nbuf)]
    else:
        bufs = [make_input(torch, push, nvfo, sr, 0x5D2B0001 + 1000 * rank + b, device) for b in range(args.nbuf)]
    ctx = capi.Context(local, max_push=push)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # one ordering domain with torch / RCCL
    info = workloads.setup(ctx, cfg, dense_fft=True, data_width=1024, nvfo=nvfo)
    af_keep = []
    if args.af and nvfo:
        from sdrplusplus_amd import radio
        for vid, (m_, r_, _b, _c, _x) in zip(info["vids"], info["plan"]):
            a_, k_ = radio.af_desc(r_, 48000.0, 50e-6 if m_ == "WFM" else None, m_ == "NFM")
            ctx.vfo_set_af(vid, a_, k_)
            af_keep.append(k_)
    lines_per_push = push // N
    lines = torch.empty((lines_per_push, 1024), dtype=torch.float32, device=device)
    gathered = [torch.empty_like(lines) for _ in range(world)] if (dist is not None and rank == 0) else None

    def step(i):
        ctx.push_device(bufs[i % args.nbuf].data_ptr(), push)
        if dist is not None:
            ctx.fft_copy_device(0, lines_per_push, zoomed_ptr=lines.data_ptr())
            dist.gather(lines, gathered, dst=0)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # calibration (untimed): every kernel family bracketed by HIP events to find the dominant one and record the per-kernel split
    ctx.timing_enable(True)
    ncal = max(2, min(5, args.steps))
    for i in range(ncal):
        step(i)
    torch.cuda.synchronize()
    fam_all = ctx.timing_read()
    kernel_ms_all = {k: v[0] / ncal for k, v in fam_all.items() if v[0] > 0}
    # dominant family = the longest one on the CRITICAL stream: with VFOs present the FFT branch runs on a second stream as filler
    # behind the VFO bank (its launches stretch while they wait for CUs, which says nothing about the kernels themselves), so only
    # the VFO-bank families compete; FFT-only runs (cfg 2) have just the FFT families
    filler = {"fft_pass1", "fft_pass2", "fft_single", "zoom_palette"} if nvfo else set()
    cand = {k: v for k, v in kernel_ms_all.items() if k not in filler} or kernel_ms_all
    dom = max(cand, key=cand.get) if cand else None
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # timed region: only the dominant family keeps its event pair (two event records per step on its launch stream)
    ctx.timing_enable(True, families=[ctx.family_index(dom)] if dom else [])
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fam = ctx.timing_read()
    ctx.timing_enable(False)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity: the work was really done (outputs have the expected sizes)
    assert ctx.fft_lines() == lines_per_push
    if cfg == 3:
        for vid in info["vids"][:1]:
            assert abs(ctx.vfo_out_count(vid) - push // 40) <= 2

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_samples = world * push * args.steps
    value = total_samples / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel (HIP events around every launch of the family, on the launch stream) ----
    # geometry of the VFO chain (all VFOs of cfg 3 share it): decimation plan, resampler, channel / audio filters
    geo = None
    if nvfo:
        from sdrplusplus_amd import radio
        m0, r0, b0, c0, _ = info["plan"][0]
        d0, _keep = radio.vfo_desc(sr, r0, b0, c0, m0)
        st = [(int(d0.stage_decim[i]), int(d0.stage_ntaps[i])) for i in range(d0.n_stages)]
        geo = dict(stages=st, interp=int(d0.interp), decim=int(d0.decim), resamp_ntaps=int(d0.resamp_ntaps), chan_ntaps=int(d0.chan_ntaps), audio_ntaps=int(d0.audio_ntaps))
    matrix_front = bool(geo) and nvfo >= 17 and len(geo["stages"]) >= 2      # vfo_frontcm_kernel: stages 1+2 as one composite FIR on the MFMA pipe
    matrix_fir = not os.environ.get("SDRPP_GPU_VALU_FIR")                      # vfo_toep_kernel for the filters behind the front end
    bytes_per_launch = {"fft_pass1": push * (8 + 8), "fft_pass2": push * (8 + 4), "zoom_palette": push * 4.0}  # compulsory HBM bytes of ONE launch per family
    flops_per_launch = {
        "fft_pass1": push * (3 * 2 * 8 + 8 + 2),   # 8 radix-2 stages x 6 FMA per butterfly (3 per point) + window + twiddle
        "fft_pass2": push * (3 * 2 * 8 + 12),
    }
    bound_of = {"fft_pass1": "hbm", "fft_pass2": "hbm", "fft_single": "hbm", "zoom_palette": "hbm", "carry_misc": "hbm"}
    if geo:
        (D1, K1) = geo["stages"][0]
        fused = len(geo["stages"]) >= 2
        D2, K2 = geo["stages"][1] if fused else (1, 1)
        Kc, Dc = K1 + (K2 - 1) * D1, D1 * D2
        n_front = push / Dc
        bytes_per_launch["vfo_stage1"] = push * 8 + nvfo * n_front * 8
        if matrix_front:   # composite filter at its output rate: tap PAIRS x (2 matrix rows x re/im) + NCO (two complex products)
            flops_per_launch["vfo_stage1"] = nvfo * n_front * (((Kc + 1) // 2) * 8.0 + 16.0)
        else:              # two-stage VALU form: 4 FMA per stage-1 tap pair, phasor, stage 2 on complex data with real taps
            flops_per_launch["vfo_stage1"] = push * nvfo * (((K1 + 1) // 2) * 8.0 / D1 + 8.0 / D1 + K2 * 4.0 / Dc)
        bound_of["vfo_stage1"] = "mfma" if matrix_front else "fp32_valu"
        n, fl, by = n_front, 0.0, 0.0
        for (Ds, Ks) in geo["stages"][2:]:
            by += nvfo * (n * 8 + n / Ds * 8)
            n /= Ds
            fl += nvfo * n * Ks * 4.0
        bytes_per_launch["vfo_decim"], flops_per_launch["vfo_decim"] = by, fl
        if geo["interp"] != geo["decim"]:
            tpp = -(-geo["resamp_ntaps"] // geo["interp"])
            n_out = n * geo["interp"] / geo["decim"]
            bytes_per_launch["vfo_poly"] = nvfo * (n * 8 + n_out * 8)
            flops_per_launch["vfo_poly"] = nvfo * n_out * tpp * 4.0
            n = n_out
        # channel filter (complex) + discriminator/audio low-pass (real in, stereo out)
        bytes_per_launch["vfo_fir"] = nvfo * n * ((16 if geo["chan_ntaps"] else 0) + 8 + 8)
        flops_per_launch["vfo_fir"] = nvfo * n * (geo["chan_ntaps"] * 4.0 + geo["audio_ntaps"] * 2.0)
        for f in ("vfo_decim", "vfo_poly", "vfo_fir"):
            bound_of[f] = "mfma" if matrix_fir else "fp32_valu"
    kernel_ms = {k: v[0] / args.steps for k, v in fam.items() if v[0] > 0}  # dominant family only, measured inside the timed region
    roof = None
    roof_valu = None
    if dom is not None and dom in bytes_per_launch and cfg == 3:
        dur = kernel_ms[dom] * 1e-3
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
            # rocprofv3 --pmc runs, tools/rocpd_summary.py); only quoted when the profile was taken on this very workload
            try:
                prof = json.load(open(pmc))
                meta = prof.get("_meta", {})
                if int(meta.get("push", 0)) == push and int(meta.get("cfg", 0)) == cfg and int(meta.get("nvfo", -1)) == nvfo:
                    hits = [v["hbm_bytes_per_launch"] for k, v in prof.items() if k != "_meta" and any(k.startswith(pfx) for pfx in FAMILY_KERNELS[dom])]
                    traffic = round(sum(hits)) if hits else None
            except Exception:
                traffic = None
        bound = bound_of.get(dom, "hbm")
        gbs = bytes_per_launch[dom] / dur / 1e9
        if bound != "mfma" or dom not in flops_per_launch:  # HBM roofline (+ roofline_valu below for the FP32-VALU kernels)
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "algorithmic_bytes_per_launch": bytes_per_launch[dom], "avg_launch_ms": round(kernel_ms[dom], 4)}
        else:
            # FP32 filter-bank kernels: the floor is the FP32 multiply-add rate (matrix pipe = vector pipe = 157.3 TFLOP/s on MI355X);
            # `achieved` counts the ALGORITHMIC flops of the filters (DESIGN.md "Kernels"), not the zero band of the Toeplitz tiles
            tf = flops_per_launch[dom] / dur / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": round(tf, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / VALU_PEAK_TFLOPS, 5), "traffic": traffic, "algorithmic_flops_per_launch": flops_per_launch[dom],
                    "algorithmic_bytes_per_launch": bytes_per_launch[dom], "hbm_GBps_at_this_rate": round(gbs, 2), "avg_launch_ms": round(kernel_ms[dom], 4),
                    "note": "family = all launches of this kind in one push (e.g. vfo_fir = channel filter + discriminator/audio low-pass)"}
        if dom in flops_per_launch and roof["bound"] == "hbm":
            tf = flops_per_launch[dom] / dur / 1e12
            roof_valu = {"kernel": dom, "bound": "fp32_valu", "achieved": round(tf, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / VALU_PEAK_TFLOPS, 5)}
    # FFT 8 in + 4 out, VFO outputs at their IF rates; the IQ read is shared by both branches
    out_rate = sum(r for _, r, _, _, _ in info["plan"]) if nvfo else 0.0
    path_bytes = 12.0 + out_rate / sr * 8
    roof_path = {"bound": "hbm", "achieved": round(value * 1e6 / world * path_bytes / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(value * 1e6 / world * path_bytes / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_sample": path_bytes,
                 "note": "whole step, per GPU: SURVEY.md 8(d) path figure x ingest rate"}

    out = {
        "metric": {3: "IQ Msamples/s ingested (65536-pt FFT + 32 VFO WFM)", 2: "IQ Msamples/s ingested (65536-pt FFT only, cfg2)", 4: "IQ Msamples/s ingested (2^20-pt FFT + 128 VFO mixed NFM/AM/USB, cfg4)"}[cfg],
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg%d: %.2f MS/s-format synthetic IQ, %d-pt dense FFT + log-power waterfall%s" % (cfg, sr / 1e6, N, (" + %d VFO x %s (xlate+FIR+resample+demod)" % (nvfo, "WFM" if cfg == 3 else "NFM/AM/USB")) if nvfo else ""),
                   "samples_per_step_per_gpu": push, "streams": world, "parallelism": "one independent IQ stream per GPU" + ("; RCCL gather of zoomed waterfall lines to rank 0" if world > 1 else ""),
                   "input_batches_rotated": args.nbuf, "af_chain": bool(args.af and nvfo), "device": ctx.device_info()},
        "roofline": roof, "roofline_valu": roof_valu, "roofline_path": roof_path,
        "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(kernel_ms_all.items(), key=lambda kv: -kv[1])},
        "kernel_ms_note": "per-family HIP-event times from an untimed calibration pass (FFT branch and VFO bank run on two streams and overlap, so the "
                          "entries sum to more than ms_per_step); roofline.avg_launch_ms is the dominant family re-measured inside the timed region",
        "realtime_factor": round(value * 1e6 / world / sr, 1),
    }
    if world == 1 and not args.no_cpu_baseline and cfg == 3:
        try:
            out["cpu_baseline"] = cpu_baseline(sr, nvfo, N, int(sr / 200))
            if out["cpu_baseline"]:
                out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU result
            out["cpu_baseline"] = {"error": repr(e)}
    try:  # C stdio of anything loaded into this process goes out BEFORE the JSON line, which must be the last line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
