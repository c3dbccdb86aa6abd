"""The mode bench.py measures, at the geometry it measures it in, DIRECTLY against the oracle (not through "pipelined == ordinary pass"):
pipelined execution (sdrpp_set_pipelined: one launch per block), the reference's sr/200 blocks marked inside larger blocks
(sdrpp_set_reference_block), result flags 1 | 2 | 4, enough blocks for the steady state of the role queue (crowded ticks, every level of the
pipeline busy).  Every raw dB line / zoomed line / palette index bit-exact (iq_frontend.cpp:248-267, waterfall.cpp:65-90), every audio stream
within BASELINE.json's 1e-5 RMS (rx_vfo.h:89-100, broadcast_fm.h:146-212, fm.h, am.h, ssb.h).  sdrpp_pipeline_stats says which roles ran,
so a silent fall-back to ordinary passes or to another front-end shape fails the test instead of passing it.

  (a) cfg 3 as bench.py sets it up: 32 x WFM + dense 65536-point FFT, 10^6-sample blocks resident on the device, reference block 50 000
  (b) the same at sr/200 = 50 000-sample blocks: device-resident (front end in its small-block shape) and fetched from host memory
  (c) cfg 4: 128 VFOs NFM / AM / USB + 2^20-point FFT, 10^6- and 307 200-sample blocks (long first stages: the SET = 1 build of the kernel)
  (d) cfg 2: FFT only, 65536 and 2^20 points, 10^6-sample blocks
  (e) cfg 3 + the radio's AF chain on every VFO (`other_configs.cfg3_af`), 10^6-sample and sr/200 blocks: the AF chain's output against the
      oracle's chain"""
import numpy as np
import pytest
import torch  # BEFORE the product library: torch preloads its own copy of the HIP runtime by path, and the second runtime in a process finds no device

import support as S
from test_full_configs_gpu import _oracle_streams, _synth_threaded, rms

pytestmark = pytest.mark.gpu


def _device_blocks(x, B):
    """The signal cut into blocks of B samples resident on the device (what bench.py pushes: sdrpp_push_device reads them in place)."""
    t = torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).to("cuda:0")
    torch.cuda.synchronize()
    return t, [(t.data_ptr() + 8 * i, min(B, len(x) - i)) for i in range(0, len(x), B)]


def _check_lines(spec, view, x, cuts, results):
    """results[t]: sdrpp_result of block t; the oracle's frames do not depend on how the stream is cut, the assignment of lines to blocks does."""
    start, size, width, lo, hi = view
    pos, total = 0, 0
    for t, n in enumerate(cuts):
        ol = spec.push(x[pos:pos + n])
        pos += n
        r = results[t]
        assert r["n_lines"] == len(ol), (t, r["n_lines"], len(ol))
        total += len(ol)
        if len(ol) == 0:
            continue
        assert r["raw"] is not None and np.array_equal(r["raw"].view(np.uint32), ol.view(np.uint32)), "block %d: raw dB lines differ from the oracle" % t
        if r["zoomed"] is not None:
            oz = np.stack([S.oracle_do_zoom(start, size, width, l) for l in ol])
            assert np.array_equal(r["zoomed"].view(np.uint32), oz.view(np.uint32)), "block %d: zoomed lines" % t
            assert np.array_equal(r["index"], np.stack([S.oracle_palette_index(z, lo, hi) for z in oz])), "block %d: palette indices" % t
    return total


def _run_pipelined(ctx, feed, cuts, lag=9):
    """Push every block, collecting results `lag` blocks behind like a streaming host; returns the per-block result dicts."""
    out = []
    for t, n in enumerate(cuts):
        feed(t, n)
        if t + 1 > lag:
            out.append(ctx.result_wait(t + 1 - lag))
            ctx.result_release(t + 1 - lag)
    for t in range(len(out) + 1, len(cuts) + 1):
        out.append(ctx.result_wait(t))
        ctx.result_release(t)
    return out


def _audio_check(info, results, ref, tol=1e-5, what=""):
    worst = {}
    for k, (vid, (m, _, _, _, _)) in enumerate(zip(info["vids"], info["plan"])):
        ga = np.concatenate([r["vfo"][vid] for r in results])
        oa = ref[k][1]
        assert ga.shape == oa.shape, (what, k, m, ga.shape, oa.shape)
        e = rms(ga - oa) / max(1.0, rms(oa))
        worst[m] = max(worst.get(m, 0.0), e)
        assert e < tol, (what, k, m, e)
    return worst


@pytest.mark.parametrize("fed", ["device_1M", "device_sr200", "host_sr200"])
def test_cfg3_pipelined_bench_geometry_vs_oracle(fed):
    """(a) + (b).  32 x WFM, 65536-point dense FFT, 1024-pixel full view, pipelined, flags 7, reference block 50 000."""
    from sdrplusplus_amd import capi, workloads

    B, nblk = (1000000, 13) if fed == "device_1M" else (50000, 48)
    x = _synth_threaded(3, B * nblk, seed=0x3A + nblk)
    ctx = capi.Context(0, max_push=B)
    info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=32)
    if B > 50000:
        ctx.set_reference_block(50000)
    ctx.set_pipelined(True, 1 | 2 | 4)
    cuts = [B] * nblk
    if fed.startswith("device"):
        keep, blocks = _device_blocks(x, B)
        feed = lambda t, n: ctx.push_device(blocks[t][0], n)
    else:
        keep = None
        feed = lambda t, n: ctx.push(x[t * B:t * B + n])  # pageable host memory -> page-locked staging slot -> the tick's landing copy
    results = _run_pipelined(ctx, feed, cuts)
    st = ctx.pipeline_stats()
    assert st["tick_blocks"] == nblk and st["pass_blocks"] == 0, st
    if fed == "device_1M":
        assert st["crowded_ticks"] >= nblk - 2, st  # the order of the roles bench.py's ticks run in
        assert "fcm_132_4" in st["roles"] and "fcm16_132_4" not in st["roles"], st
    elif fed == "device_sr200":
        assert "fcm16_132_4" in st["roles"] and "fcm_132_4" not in st["roles"], st  # small-block shape of the front end
    else:
        assert "fcm_132_4" in st["roles"] and "fcm16_132_4" not in st["roles"], st  # host-fetched blocks keep the 32 x 32 x 2 shape
    assert any(k in st["roles"] for k in ("toep_q", "pipe")) and "fft_p1_8" in st["roles"] and "fft_p2_8" in st["roles"], st
    spec = S.OracleSpectrum(65536, 65536, 0, capi.design_fft_window(2, 65536))
    nlines = _check_lines(spec, info["view"], x, cuts, results)
    assert nlines == (B * nblk) // 65536
    chains = [S.OracleChain(info["sr"], r, bw, c, S.MODES[m]) for m, r, bw, c, _ in info["plan"]]
    ref = _oracle_streams(chains, x, [50000] * (B * nblk // 50000))
    worst = _audio_check(info, results, ref, what=fed)
    print("cfg3 pipelined %s: %d lines bit-exact, worst WFM audio error %.2e over %d samples; roles %s" % (fed, nlines, worst["WFM"], B * nblk, sorted(st["roles"])))
    del keep
    ctx.close()


@pytest.mark.parametrize("B,nblk", [(1000000, 10), (307200, 12)])
def test_cfg4_pipelined_vs_oracle(B, nblk):
    """(c) 128 VFOs NFM / AM / USB at 61.44 MS/s + the 2^20-point FFT, pipelined.  EVERY channel against the PINNED oracle (the reference's own
    code paths, its float rotator included): NFM / AM over the whole stream; USB over the window in which the default closed-form NCO holds
    BASELINE.json's 1e-5 against the reference's rotator — the first 2e5 input samples (measured: test_closed_form_nco_validity_window_vs_pinned_oracle;
    why no calibration of the closed form can widen it: profiles/r06_rotator_drift.md — the rotator's rounding drift is a random walk of ~3e-8 rad
    per step on top of a slope that changes with the phase, 2e-5 .. 1e-3 rad after 10^6 samples with the best linear fit taken out).  Beyond the
    window the USB channels are checked against the oracle with an EXACT NCO in the reference's place (`ideal_nco`: a statement about this
    library's arithmetic, not about parity); parity with the reference's own phase sequence for any length is nco_mode 2
    (test_full_configs_gpu.py::test_cfg4_all_128_vfos_every_mode_within_1e5[reference_rotator])."""
    from sdrplusplus_amd import capi, workloads

    RB, W = 307200, 200000
    x = _synth_threaded(4, B * nblk, seed=0x4C)
    ctx = capi.Context(0, max_push=B)
    info = workloads.setup(ctx, 4, dense_fft=True, data_width=1024)
    if B != RB:
        ctx.set_reference_block(RB)
    ctx.set_pipelined(True, 1 | 2 | 4)
    keep, blocks = _device_blocks(x, B)
    cuts = [B] * nblk
    results = _run_pipelined(ctx, lambda t, n: ctx.push_device(blocks[t][0], n), cuts)
    st = ctx.pipeline_stats()
    assert st["tick_blocks"] == nblk and st["pass_blocks"] == 0, st
    assert "fcl_pf" in st["roles"], st  # long first stages with the register-prefetched window: the 247-register build of the tick kernel
    assert "fft_p2row" in st["roles"] and "fft_tr" in st["roles"] and "seq" in st["roles"], st
    N = 1 << 20
    spec = S.OracleSpectrum(N, N, 0, capi.design_fft_window(2, N))
    nlines = _check_lines(spec, info["view"], x, cuts, results)
    assert nlines == (B * nblk) // N
    # the reference's blocks: RB-sample blocks restarting with every push (sdrpp_set_reference_block cuts every push from its start)
    per = [RB] * (B // RB) + ([B % RB] if B % RB else [])
    pinned = [S.OracleChain(info["sr"], r, bw, c, S.MODES[m]) for m, r, bw, c, _ in info["plan"]]
    ref = _oracle_streams(pinned, x, per * nblk)
    worst = {}
    usb = []
    for k, (vid, (m, _, _, _, _)) in enumerate(zip(info["vids"], info["plan"])):
        ga = np.concatenate([r["vfo"][vid] for r in results])
        oa = ref[k][1]
        assert ga.shape == oa.shape, (k, m, ga.shape, oa.shape)
        if m == "USB":  # the window that holds: audio produced by the first W input samples, relative to the RMS of the whole reference stream
            nwin = int(len(oa) * (W / float(B * nblk)))
            assert nwin >= 70, nwin
            e = rms(ga[:nwin] - oa[:nwin]) / max(1.0, rms(oa))
            usb.append(k)
        else:
            e = rms(ga - oa) / max(1.0, rms(oa))
        worst[m] = max(worst.get(m, 0.0), e)
        assert e < 1e-5, ("cfg4 B=%d vs the pinned oracle" % B, k, m, e)
    assert len(usb) >= 40
    ideal = [S.OracleChain(info["sr"], *info["plan"][k][1:4], S.MODES["USB"], ideal_nco=True) for k in usb]
    ref_i = _oracle_streams(ideal, x, per * nblk)
    worst_i = 0.0
    for q, k in enumerate(usb):
        ga = np.concatenate([r["vfo"][info["vids"][k]] for r in results])
        e = rms(ga - ref_i[q][1]) / max(1.0, rms(ref_i[q][1]))
        worst_i = max(worst_i, e)
        assert e < 1e-5, ("cfg4 B=%d USB vs the exact-NCO oracle, whole stream" % B, k, e)
    print("cfg4 pipelined B=%d: %d 2^20-point lines bit-exact; vs the PINNED oracle, worst relative audio error per mode %s (USB: first %d input samples); USB over all %d samples vs the exact-NCO oracle %.2e"
          % (B, nlines, {m: "%.2e" % v for m, v in worst.items()}, W, B * nblk, worst_i))
    del keep
    ctx.close()


@pytest.mark.parametrize("lgn", [16, 20])
def test_cfg2_pipelined_vs_oracle(lgn):
    """(d) FFT branch alone, pipelined, 10^6-sample blocks: 65536 points (pass 1 / pass 2 roles) and 2^20 points (column pass, 4096-point rows,
    transpose)."""
    from sdrplusplus_amd import capi, workloads

    N, B, nblk = 1 << lgn, 1000000, 12
    x = _synth_threaded(2, B * nblk, seed=0x2D + lgn)
    ctx = capi.Context(0, max_push=B)
    ctx.fft_configure(N, N, 0, capi.design_fft_window(2, N))
    start, size = capi.design_waterfall_view(0.0, 10e6, 10e6, N)
    view = (start, size, 1024, -120.0, 0.0)
    ctx.fft_set_view(*view)
    ctx.set_pipelined(True, 2 | 4)
    keep, blocks = _device_blocks(x, B)
    cuts = [B] * nblk
    results = _run_pipelined(ctx, lambda t, n: ctx.push_device(blocks[t][0], n), cuts)
    st = ctx.pipeline_stats()
    assert st["tick_blocks"] == nblk and st["pass_blocks"] == 0, st
    assert ("fft_p2_8" in st["roles"]) if lgn == 16 else ("fft_p2row" in st["roles"] and "fft_tr" in st["roles"]), st
    spec = S.OracleSpectrum(N, N, 0, capi.design_fft_window(2, N))
    nlines = _check_lines(spec, view, x, cuts, results)
    assert nlines == (B * nblk) // N
    del keep
    ctx.close()


@pytest.mark.parametrize("B,nblk", [(1000000, 8), (50000, 60)])
def test_cfg3_af_pipelined_bench_geometry_vs_oracle(B, nblk):
    """(e) `other_configs.cfg3_af` as bench.py sets it up: 32 x WFM + the radio's AF chain on every VFO (resampler to 48 kHz, 50 us
    de-emphasis: radio_module.h:98-110), 65536-point dense FFT, pipelined, VFO blocks (= the AF chain's output) + zoomed lines delivered, at
    10^6-sample blocks with the reference block marked and at sr/200 blocks.  AF output of every VFO within 1e-5 RMS of the oracle's chain,
    lines bit-exact, no block as an ordinary pass, the AF roles present."""
    from sdrplusplus_amd import capi, radio, workloads
    from test_parity_vfo import _OracleAf

    x = _synth_threaded(3, B * nblk, seed=0xAF + nblk)
    ctx = capi.Context(0, max_push=B)
    info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=32)
    keep_af = []
    for vid, (m, r, _bw, _c, _car) in zip(info["vids"], info["plan"]):
        a, k = radio.af_desc(r, 48000.0, 50e-6, False)
        ctx.vfo_set_af(vid, a, k)
        keep_af.append(k)
    if B > 50000:
        ctx.set_reference_block(50000)
    ctx.set_pipelined(True, 1 | 2 | 4)
    keep, blocks = _device_blocks(x, B)
    cuts = [B] * nblk
    results = _run_pipelined(ctx, lambda t, n: ctx.push_device(blocks[t][0], n), cuts, lag=14)
    st = ctx.pipeline_stats()
    assert st["tick_blocks"] == nblk and st["pass_blocks"] == 0, st
    assert "polyc" in st["roles"] and "deemp_p0" in st["roles"] and "deemp_p1" in st["roles"], st
    spec = S.OracleSpectrum(65536, 65536, 0, capi.design_fft_window(2, 65536))
    nlines = _check_lines(spec, info["view"], x, cuts, results)
    chains = [S.OracleChain(info["sr"], r, bw, c, S.MODES[m]) for m, r, bw, c, _ in info["plan"]]
    ref = _oracle_streams(chains, x, [50000] * (B * nblk // 50000))
    worst = 0.0
    for k, (vid, (m, r, _bw, _c, _car)) in enumerate(zip(info["vids"], info["plan"])):
        oaf = _OracleAf(r, 48000.0, 50e-6, False).process(ref[k][1])
        got = np.concatenate([q["vfo"][vid] for q in results])
        assert got.shape == oaf.shape, (k, got.shape, oaf.shape)
        e = rms(got - oaf) / max(1.0, rms(oaf))
        worst = max(worst, e)
        assert e < 1e-5, (k, e)
    print("cfg3 + AF pipelined B=%d: %d lines bit-exact, worst AF-output error %.2e over %d input samples, %d levels" % (B, nlines, worst, B * nblk, st["depth"]))
    del keep, keep_af
    ctx.close()


def test_closed_form_nco_validity_window_vs_pinned_oracle():
    """The statement in include/sdrpp_gpu.h about the DEFAULT (closed-form) NCO, as a measurement: cfg 4's USB channels, pipelined like bench.py runs
    them, against the PINNED oracle — the reference's own float rotator (frequency_xlator.h:43-50, VOLK rotator2 with its 512-sample
    renormalisation), no `ideal_nco`.  The difference is that rotator's rounding drift (1e-10 .. 2e-9 rad per sample, linear in time), which the
    product detector of ssb.h:77-92 and the raw IF see and FM / AM do not.  Asserted, per channel, relative to the RMS of the reference stream,
    in windows of 10^5 INPUT samples since sdrpp_vfo_add:
      * audio inside BASELINE.json's 1e-5 over the first 2e5 input samples, every channel (the worst channel leaves the tolerance in the third window);
      * from 10^6 input samples on (channel filter filled, AGC settled) the error of window w grows at most like 1.3e-5 * (w + 1) for EVERY channel
        and the median channel's growth rate lies in 7e-7 .. 3e-6 per 10^5 samples (oracle-vs-oracle calibration: 2.8e-8 .. 1.05e-5, median 1.5e-6);
      * the raw IF (RxVFO::out) of the same channels: growth at most 4e-5 * (w + 1), median rate 2e-6 .. 9e-6 (calibration 1.0e-7 .. 3.1e-5, 4.5e-6).
    The header quotes these figures and names this test; channels that need the reference's own phase sequence beyond the window run with
    sdrpp_vfo_desc.nco_mode = 2 (test_full_configs_gpu.py::test_cfg4_all_128_vfos_every_mode_within_1e5[reference_rotator])."""
    from sdrplusplus_amd import capi, radio, workloads

    RB, nblk, W = 307200, 10, 100000
    x = _synth_threaded(4, RB * nblk, seed=0x4C)
    ctx = capi.Context(0, max_push=RB)
    info = workloads.setup(ctx, 4, dense_fft=True, data_width=1024)
    ctx.set_pipelined(True, 1)
    keep, blocks = _device_blocks(x, RB)
    results = _run_pipelined(ctx, lambda t, n: ctx.push_device(blocks[t][0], n), [RB] * nblk, lag=11)
    st = ctx.pipeline_stats()
    assert st["tick_blocks"] == nblk and st["pass_blocks"] == 0 and "fcl_pf" in st["roles"], st
    usb = [(vid, p) for vid, p in zip(info["vids"], info["plan"]) if p[0] == "USB"]
    assert len(usb) >= 40
    chains = [S.OracleChain(info["sr"], r, bw, c, S.MODES[m]) for _, (m, r, bw, c, _c2) in usb]  # PINNED: the reference's rotator
    ref = _oracle_streams(chains, x, [RB] * nblk)
    nw = RB * nblk // W

    def windows(got, want):
        per = len(want) / float(RB * nblk) * W
        r0 = rms(want)
        return np.array([rms(got[int(w * per):int((w + 1) * per)] - want[int(w * per):int((w + 1) * per)]) / r0 for w in range(nw)])

    E = np.stack([windows(np.concatenate([r["vfo"][vid] for r in results]), oa) for (vid, _), (_, oa) in zip(usb, ref)])
    t = np.arange(nw) + 0.5
    assert E[:, :2].max() < 1e-5, ("audio, first 2e5 input samples", float(E[:, :2].max()))
    assert np.all(E[:, 10:] <= 1.3e-5 * (np.arange(10, nw) + 1)), ("audio growth", float((E[:, 10:] / (np.arange(10, nw) + 1)).max()))
    rate = np.median(E[:, 10:] / t[10:])
    assert 7e-7 < rate < 3e-6, ("median audio growth per 1e5 input samples", float(rate))
    worst_cross = int(np.argmax(E.max(axis=0) > 1e-5)) if (E.max(axis=0) > 1e-5).any() else nw
    med_cross = int(np.argmax(np.median(E, axis=0) > 1e-5)) if (np.median(E, axis=0) > 1e-5).any() else nw
    del keep
    ctx.close()
    # the raw IF of eight of those channels (an 8-VFO bank on the long-first-stage matrix kernel, ordinary passes: the same closed-form NCO)
    ctx = capi.Context(0, max_push=RB)
    pick = usb[::5][:8]
    vids = []
    for _, (m, r, bw, c, _c2) in pick:
        d, kp = radio.vfo_desc(info["sr"], r, bw, c, "RAW")
        vids.append(ctx.vfo_add(d, kp))
    raw = [S.OracleChain(info["sr"], r, bw, c, None) for _, (m, r, bw, c, _c2) in pick]
    got = [[] for _ in pick]
    for b in range(nblk):
        ctx.push(x[b * RB:(b + 1) * RB])
        for k, vid in enumerate(vids):
            got[k].append(ctx.vfo_read_if(vid).copy())
    refi = [np.concatenate([ch.process(x[b * RB:(b + 1) * RB])[0] for b in range(nblk)]) for ch in raw]  # (a raw chain has no audio: not _oracle_streams)
    EI = np.stack([windows(np.concatenate(g).view(np.float32), oi.view(np.float32)) for g, oi in zip(got, refi)])
    assert np.all(EI[:, 10:] <= 4e-5 * (np.arange(10, nw) + 1)), ("IF growth", float((EI[:, 10:] / (np.arange(10, nw) + 1)).max()))
    ratei = np.median(EI[:, 10:] / t[10:])
    assert 1e-6 < ratei < 1.2e-5, ("median IF growth per 1e5 input samples", float(ratei))
    ctx.close()
    print("closed-form NCO vs the pinned oracle, %d USB channels, %d input samples: audio first 2e5 samples %.2e; growth per 1e5 samples median %.2e, max %.2e; "
          "worst channel leaves 1e-5 in window %d, the median channel in window %d (windows of 1e5 input samples); raw IF growth median %.2e, max %.2e"
          % (len(usb), RB * nblk, E[:, :2].max(), rate, (E[:, 10:] / t[10:]).max(), worst_cross, med_cross, ratei, (EI[:, 10:] / t[10:]).max()))
