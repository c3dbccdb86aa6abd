"""Per-VFO channeliser + demodulator parity: product kernels vs the oracle (= the reference's RxVFO + radio demodulators,
bit-exactly, see test_oracle_vs_reference.py).  Tolerance per BASELINE.json: audio within 1e-5 RMS.

What differs by design: dot products are summed in a different order, and by default the frequency
translation uses a closed-form NCO at arg(phaseDelta) instead of the reference's fp32 phase recursion, whose own rounding makes
it drift by ~1e-10..2e-9 rad/sample from its nominal increment (test_oracle_kat.py::test_rotator_drift_against_ideal_nco).
FM and AM outputs do not see that drift; a product detector (SSB) and the raw IF do.  Three kinds of check pin this down:
  * closed-form NCO vs the oracle with its ideal-NCO test switch (float64 phase, everything else the pinned restatement): isolates
    the rotator — every mode, arbitrary offsets, inside 1e-5 (test_closed_form_nco_vs_ideal_nco_oracle);
  * reference-rotator mode (sdrpp_set_nco_mode(1): the float recursion itself on the device) vs the PINNED oracle: every mode,
    arbitrary offsets, inside 1e-5 for any length (test_reference_rotator_mode_matches_the_reference);
  * closed-form NCO vs the pinned oracle at arbitrary offsets: what is left is the reference's own drift, bounded in
    test_ssb_arbitrary_offset_is_drift_limited."""
import ctypes as C

import numpy as np
import pytest

import support as S


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2))) if a.size else 0.0


def _setup(sr, specs, max_push, nco_mode=0, ref_block=0, ideal_nco=False):
    """specs: [(mode, offset)] -> (ctx, vids, oracle chains, [(if_rate, bw)])"""
    from sdrplusplus_amd import capi, radio

    ctx = capi.Context(0, max_push=max_push)
    ctx.set_nco_mode(nco_mode)
    ctx.set_reference_block(ref_block)
    vids, chains, rates = [], [], []
    for mode, offset in specs:
        if_rate, bw = radio.RADIO_DEFAULTS.get(mode, (250e3, 250e3))
        d, keep = radio.vfo_desc(sr, if_rate, bw, offset, mode)
        vids.append(ctx.vfo_add(d, keep))
        chains.append(S.OracleChain(sr, if_rate, bw, offset, S.MODES.get(mode), ideal_nco=ideal_nco))
        rates.append((if_rate, bw))
    return ctx, vids, chains, rates


def _audio_tol(ref):
    return 1e-5 * max(1.0, rms(ref))


def test_cfg3_wfm_bank(backend):
    """BASELINE cfg 3 geometry (10 MS/s, WFM VFOs 300 kHz apart, 65536-pt FFT alongside), reference block size sr/200."""
    from sdrplusplus_amd import capi, workloads

    sr, B, nblk, nv = 10e6, 50000, 6, 9
    x = workloads.synth(3, B * nblk, seed=3, nvfo=nv)
    plan = workloads.vfo_plan(3, nv)
    ctx, vids, chains, _ = _setup(sr, [(m, c) for m, _, _, c, _ in plan], B)
    ideal = [S.OracleChain(sr, r, bw, c, None, ideal_nco=True) for _, r, bw, c, _ in plan]  # IF only: isolates the rotator's drift
    N = 65536
    w = capi.design_fft_window(2, N)
    ctx.fft_configure(N, N, 0, w)
    spec = S.OracleSpectrum(N, N, 0, w)
    worst_audio, worst_if, worst_if_ideal = 0.0, 0.0, 0.0
    for b in range(nblk):
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        raw, _, _ = ctx.fft_read(zoomed=False)
        ol = spec.push(blk)
        assert raw.shape == ol.shape and np.array_equal(raw, ol)
        for vid, ch, ich in zip(vids, chains, ideal):
            oi, oa = ch.process(blk)
            ii, _ = ich.process(blk)
            gi, ga = ctx.vfo_read_if(vid), ctx.vfo_read(vid)
            assert gi.shape == oi.shape and ga.shape == oa.shape
            assert len(oa) in (1249, 1251)
            worst_audio = max(worst_audio, rms(ga - oa))
            worst_if = max(worst_if, rms(gi - oi) / rms(oi))
            worst_if_ideal = max(worst_if_ideal, rms(gi - ii) / rms(ii))
    assert worst_audio < 1e-5, worst_audio      # measured ~1e-7
    assert worst_if_ideal < 2e-6, worst_if_ideal  # everything but the rotator: measured ~1.5e-7
    assert worst_if < 2e-3, worst_if             # against the reference's own rotator: its drift over 300k samples (measured ~1e-5)
    ctx.close()


@pytest.mark.parametrize("nv", [20, 33])
def test_matrix_core_front_bank(backend, nv):
    """>= 17 VFOs of one geometry take the MFMA fused-front kernel (jobs of up to 32 VFOs; 20 -> one partly filled job,
    33 -> one full job + the VALU kernel for the last one; more would leave the +-5 MHz band).  Uneven pushes exercise history, tile and push boundaries."""
    from sdrplusplus_amd import workloads

    sr = 10e6
    pushes = [50000, 1031, 20000, 7, 33333]
    x = workloads.synth(3, sum(pushes), seed=11, nvfo=nv)
    plan = workloads.vfo_plan(3, nv)
    ctx, vids, chains, _ = _setup(sr, [(m, c) for m, _, _, c, _ in plan], max(pushes))
    worst_audio, worst_if, pos = 0.0, 0.0, 0
    for n in pushes:
        blk = x[pos:pos + n]
        pos += n
        ctx.push(blk)
        for vid, ch in zip(vids, chains):
            oi, oa = ch.process(blk)
            gi, ga = ctx.vfo_read_if(vid), ctx.vfo_read(vid)
            assert gi.shape == oi.shape and ga.shape == oa.shape
            if len(oa):
                worst_audio = max(worst_audio, rms(ga - oa))
                worst_if = max(worst_if, rms(gi - oi) / max(rms(oi), 1e-9))
    assert worst_audio < 1e-5, worst_audio
    assert worst_if < 2e-3, worst_if
    ctx.close()


def test_small_block_front_end_shape_is_bit_identical(backend, monkeypatch):
    """The ratio-32 front end in its small-block shape (vfo_frontcm16_body: one 32-output tile per workgroup, 16 x 16 x 4 matrix
    instructions, SDRPP_GPU_FCM16_MAX_TILES read when a context is created) against the 32 x 32 x 2 form: IF and audio of every VFO bit
    for bit — full and partly filled jobs (32 and 20 VFOs), pushes that end inside a tile, pushes shorter than a tile, windows that reach
    into the history; ordinary passes and pipelined mode."""
    from sdrplusplus_amd import workloads

    sr = 10e6
    for nv in (32, 20):
        pushes = [50000, 1031, 20000, 7, 33333, 50000]
        x = workloads.synth(3, sum(pushes), seed=13, nvfo=nv)
        plan = workloads.vfo_plan(3, nv)
        for pipelined in (False, True):
            outs = []
            for small in ("0", "4096"):
                monkeypatch.setenv("SDRPP_GPU_FCM16_MAX_TILES", small)
                ctx, vids, _, _ = _setup(sr, [(m, c) for m, _, _, c, _ in plan], max(pushes))
                if pipelined:
                    ctx.set_pipelined(True, 1)
                got, pos = [], 0
                for n in pushes:
                    ctx.push(x[pos:pos + n])
                    pos += n
                    if pipelined:
                        r = ctx.result_wait(ctx.ticket(), copy=True)
                        got.append([r["vfo"][v] for v in vids])
                        ctx.result_release(r["ticket"])
                    else:
                        got.append([a.copy() for a in ctx.vfo_read_many(vids)] + [a.copy() for a in ctx.vfo_read_many(vids, which=[1] * len(vids))])
                outs.append(got)
                ctx.close()
            for ga, gb in zip(*outs):
                assert len(ga) == len(gb)
                for a, b in zip(ga, gb):
                    assert a.shape == b.shape and np.array_equal(a, b)
            assert sum(len(a) for a in outs[0][0]) > 0


@pytest.mark.parametrize("nv,pushes", [(54, [307200, 100003, 204397]), (33, [153600, 70003, 83597])], ids=["18_per_mode", "11_per_mode_16_row_shape"])
def test_long_first_stage_bank(backend, nv, pushes):
    """cfg 4 geometry (61.44 MS/s; plans 1024 / 4096 / 2048 with a /64 first stage of 257 / 400 / 329 taps): 18 VFOs per mode take
    the matrix-core kernel for long first stages (vfo_frontcl_kernel), the stages behind it the Toeplitz kernels; 11 per mode (what is left of
    cfg 4's 43 channels per mode behind a job of 32) take its 16 x 16 x 4 shape (round 5: vfo_frontcl_impl<PF, true>, odd and even tap counts)."""
    from sdrplusplus_amd import workloads

    sr = 61.44e6
    x = workloads.synth(4, sum(pushes), seed=13, nvfo=nv)
    plan = workloads.vfo_plan(4, nv)
    ctx, vids, chains, _ = _setup(sr, [(m, c) for m, _, _, c, _ in plan], max(pushes))
    worst = {"NFM": 0.0, "AM": 0.0, "USB": 0.0}
    pos = 0
    for n in pushes:
        blk = x[pos:pos + n]
        pos += n
        ctx.push(blk)
        for (mode, _, _, _, _), vid, ch in zip(plan, vids, chains):
            oi, oa = ch.process(blk)
            gi, ga = ctx.vfo_read_if(vid), ctx.vfo_read(vid)
            assert gi.shape == oi.shape and ga.shape == oa.shape
            if len(oa) and mode != "USB":  # SSB at arbitrary offsets is drift-limited (see the module docstring); its IF is checked instead
                worst[mode] = max(worst[mode], rms(ga - oa) / max(1.0, rms(oa)))
            if len(oi) and mode == "USB":
                worst[mode] = max(worst[mode], rms(gi - oi) / max(rms(oi), 1e-9))
    assert worst["NFM"] < 1e-5 and worst["AM"] < 1e-5, worst
    assert worst["USB"] < 2e-3, worst  # IF: dominated by the reference rotator's own drift
    ctx.close()


def test_if_tight_with_exact_phase_steps(backend):
    """Offsets at multiples of sr/8: NCO and reference recursion are both exact, what is left is fp32 summation order."""
    sr, B = 10e6, 50000
    specs = [("RAW", sr / 8), ("RAW", -sr / 4), ("RAW", 3 * sr / 8), ("RAW", 0.0)]
    from sdrplusplus_amd import capi, radio

    ctx = capi.Context(0, max_push=B)
    vids, chains = [], []
    for _, off in specs:
        d, keep = radio.vfo_desc(sr, 250e3, 150e3, off, "RAW")
        vids.append(ctx.vfo_add(d, keep))
        chains.append(S.OracleChain(sr, 250e3, 150e3, off, None))
    r = np.random.default_rng(8)
    x = ((r.standard_normal(B * 4) + 1j * r.standard_normal(B * 4)) * 0.1).astype(np.complex64)
    for b in range(4):
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        for vid, ch in zip(vids, chains):
            oi, _ = ch.process(blk)
            gi = ctx.vfo_read_if(vid)
            out = ctx.vfo_read(vid)  # RAW mode: the output IS RxVFO::out
            assert gi.shape == oi.shape
            assert rms(gi - oi) / rms(oi) < 2e-6
            assert np.array_equal(out.view(np.complex64).ravel(), gi)
    ctx.close()


CFG4_CASES = [("NFM", -3.2e6), ("AM", 600e3), ("USB", 61.44e6 / 8), ("LSB", -61.44e6 / 4), ("DSB", 3 * 61.44e6 / 8), ("AM", 0.0)]


def test_cfg4_mixed_modes(backend):
    """BASELINE cfg 4 geometry: 61.44 MS/s, NFM / AM / SSB family, reference block 307 200."""
    sr, B, nblk = 61.44e6, 307200, 4
    n = B * nblk
    t = np.arange(n) / sr
    r = np.random.default_rng(4)
    x = (r.standard_normal(n) + 1j * r.standard_normal(n)) * 1e-3
    for mode, f in CFG4_CASES:
        if mode == "NFM":
            x += 0.05 * np.exp(1j * (2 * np.pi * f * t + 2.5 * np.sin(2 * np.pi * 1000 * t)))
        elif mode == "AM":
            x += 0.05 * (1 + 0.3 * np.cos(2 * np.pi * 1000 * t)) * np.exp(2j * np.pi * (f + 10.0) * t)
        else:
            x += 0.03 * (np.exp(2j * np.pi * (f + 700) * t) + np.exp(2j * np.pi * (f - 1100) * t))
    x = x.astype(np.complex64)
    ctx, vids, chains, _ = _setup(sr, CFG4_CASES, B)
    for b in range(nblk):
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        for (mode, f), vid, ch in zip(CFG4_CASES, vids, chains):
            oi, oa = ch.process(blk)
            ga = ctx.vfo_read(vid)
            assert ga.shape == oa.shape, (mode, ga.shape, oa.shape)
            assert np.array_equal(ga[:, 0], ga[:, 1])
            assert rms(ga - oa) <= _audio_tol(oa), (mode, b, rms(ga - oa), rms(oa))
    ctx.close()


ARB_SPECS = [("USB", 1.0014e6), ("RAW", -2.34567e6), ("LSB", 333333.0), ("AM", 2.2222e6), ("DSB", -4.1e6), ("NFM", 77777.0)]


def _two_tone_mix(sr, n, specs, seed):
    t = np.arange(n) / sr
    r = np.random.default_rng(seed)
    x = (r.standard_normal(n) + 1j * r.standard_normal(n)) * 1e-3
    for _, f in specs:
        x += 0.03 * (np.exp(2j * np.pi * (f + 700) * t) + np.exp(2j * np.pi * (f - 1100) * t))
    return x.astype(np.complex64)


def _compare_streams(ctx, vids, chains, specs, x, pushes, blocks):
    """GPU pushed in `pushes`, oracle driven in `blocks` (the reference's cut); concatenated outputs compared per VFO."""
    oi, oa = [[] for _ in specs], [[] for _ in specs]
    pos = 0
    for n in blocks:
        for k, ch in enumerate(chains):
            i_, a_ = ch.process(x[pos:pos + n])
            oi[k].append(i_)
            if a_ is not None:
                oa[k].append(a_)
        pos += n
    gi, ga = [[] for _ in specs], [[] for _ in specs]
    pos = 0
    for n in pushes:
        ctx.push(x[pos:pos + n])
        pos += n
        for k, vid in enumerate(vids):
            gi[k].append(ctx.vfo_read_if(vid))
            if specs[k][0] != "RAW":
                ga[k].append(ctx.vfo_read(vid))
    res = {}
    for k, (mode, f) in enumerate(specs):
        I, G = np.concatenate(oi[k]), np.concatenate(gi[k])
        assert I.shape == G.shape, (mode, I.shape, G.shape)
        e_if = rms(G - I) / rms(I)
        e_a = 0.0
        if mode != "RAW":
            A, GA = np.concatenate(oa[k]), np.concatenate(ga[k])
            assert A.shape == GA.shape, (mode, A.shape, GA.shape)
            e_a = rms(GA - A) / max(1.0, rms(A))
        res[(mode, f)] = (e_if, e_a)
    return res


def test_closed_form_nco_vs_ideal_nco_oracle(backend):
    """The isolating test: with the reference's float rotator recursion replaced by a float64 NCO at the same arg(phaseDelta_f32)
    (oracle test switch; nothing else changes) the default device path agrees at ARBITRARY offsets in every mode — IF ~1e-7,
    audio < 1e-6 measured; bars 2e-6 / 1e-5.  So the rotator's own rounding is the only thing between the closed-form NCO and
    the reference."""
    sr, B, nblk = 10e6, 50000, 6
    x = _two_tone_mix(sr, B * nblk, ARB_SPECS, 21)
    ctx, vids, chains, _ = _setup(sr, ARB_SPECS, B, ideal_nco=True)
    res = _compare_streams(ctx, vids, chains, ARB_SPECS, x, [B] * nblk, [B] * nblk)
    for key, (e_if, e_a) in res.items():
        assert e_if < 2e-6 and e_a < 1e-5, (key, e_if, e_a)
    ctx.close()


@pytest.mark.parametrize("cut", ["blocks", "big", "ragged"])
def test_reference_rotator_mode_matches_the_reference(backend, cut):
    """sdrpp_set_nco_mode(REFERENCE_ROTATOR): the reference's float recursion (renormalised every 512 samples and at the end of
    every reference block) runs on the device, so the PINNED oracle — not the ideal-NCO variant — is matched at arbitrary offsets in
    every mode: IF ~1e-7, SSB audio ~3e-7 measured.  With sdrpp_set_reference_block the result does not depend on the push size:
    pushes of three blocks, and ragged pushes that are no multiple of a block, give the same streams."""
    sr, B, nblk = 10e6, 50000, 6
    x = _two_tone_mix(sr, B * nblk, ARB_SPECS, 22)
    if cut == "blocks":
        pushes, blocks, ref_block = [B] * nblk, [B] * nblk, 0
    elif cut == "big":
        pushes, blocks, ref_block = [3 * B] * 2, [B] * nblk, B
    else:  # every push is cut into B-sample blocks + a shorter last one: drive the oracle with exactly those blocks
        pushes, ref_block = [120001, 70000, 109999], B
        blocks = []
        for p in pushes:
            blocks += [B] * (p // B) + ([p % B] if p % B else [])
    ctx, vids, chains, _ = _setup(sr, ARB_SPECS, max(pushes), nco_mode=1, ref_block=ref_block)
    res = _compare_streams(ctx, vids, chains, ARB_SPECS, x, pushes, blocks)
    for key, (e_if, e_a) in res.items():
        assert e_if < 2e-6 and e_a < 1e-5, (key, e_if, e_a)
    ctx.close()


@pytest.mark.parametrize("vpw", [None, "3"])
def test_reference_rotator_four_wavefront_kernel_is_bit_identical(backend, monkeypatch, vpw):
    """vfo_rotate_exact4_kernel (one wavefront runs the phase chain and requests the samples, three apply the phases) against the
    one-wavefront form of the same recursion (SDRPP_GPU_ROT_EXACT_SINGLE, read when a context is created): IF and audio streams bit for bit,
    over pushes that end inside a 64-sample chunk, reference blocks that do, pushes shorter than a chunk and seven VFOs (rows dealt to the
    three applying wavefronts unevenly), in one workgroup and (SDRPP_GPU_ROTX_VPW = 3 VFOs per workgroup) in three."""
    if vpw:
        monkeypatch.setenv("SDRPP_GPU_ROTX_VPW", vpw)
    sr = 10e6
    specs = ARB_SPECS + [("RAW", 123456.0)]
    pushes = [50000, 63, 1, 120001, 64, 8 * 64 + 5, 70000, 100000]
    x = _two_tone_mix(sr, sum(pushes), specs, 31)
    outs = []
    for single in (True, False):
        if single:
            monkeypatch.setenv("SDRPP_GPU_ROT_EXACT_SINGLE", "1")
        else:
            monkeypatch.delenv("SDRPP_GPU_ROT_EXACT_SINGLE", raising=False)
        ctx, vids, _, _ = _setup(sr, specs, max(pushes), nco_mode=1, ref_block=50000)
        got, pos = [], 0
        for n in pushes:
            ctx.push(x[pos:pos + n])
            pos += n
            got.append([a.copy() for a in ctx.vfo_read_many(vids)] + [a.copy() for a in ctx.vfo_read_many(vids, which=[1] * len(vids))])
        outs.append(got)
        ctx.close()
    for ga, gb in zip(*outs):
        for a, b in zip(ga, gb):
            assert a.shape == b.shape and np.array_equal(a, b)
    assert sum(len(a) for a in outs[0][0]) > 0


def test_per_vfo_nco_mode_in_one_bank(backend):
    """sdrpp_vfo_desc.nco_mode: the SSB / raw-IF channels of a bank run the reference's rotator recursion (pinned oracle matched at
    arbitrary offsets: IF 2e-6, audio 1e-5) while its FM / AM channels stay on the closed-form fast path (audio 1e-5 against the same
    pinned oracle: nothing they output depends on the absolute phase) — one context, the context's own mode left at closed form."""
    from sdrplusplus_amd import capi, radio

    sr, B, nblk = 10e6, 50000, 5
    x = _two_tone_mix(sr, B * nblk, ARB_SPECS, 29)
    ctx = capi.Context(0, max_push=B)
    vids, chains = [], []
    for mode, offset in ARB_SPECS:
        if_rate, bw = radio.RADIO_DEFAULTS.get(mode, (250e3, 250e3))
        exact = mode in ("USB", "LSB", "DSB", "RAW")
        d, keep = radio.vfo_desc(sr, if_rate, bw, offset, mode, nco_mode=2 if exact else 1)
        vids.append(ctx.vfo_add(d, keep))
        chains.append(S.OracleChain(sr, if_rate, bw, offset, S.MODES.get(mode)))
    res = _compare_streams(ctx, vids, chains, ARB_SPECS, x, [B] * nblk, [B] * nblk)
    for (mode, off), (e_if, e_a) in res.items():
        if mode in ("USB", "LSB", "DSB", "RAW"):
            assert e_if < 2e-6 and e_a < 1e-5, (mode, e_if, e_a)
        else:
            assert e_a < 1e-5, (mode, e_if, e_a)  # (their IF carries the reference rotator's drift: section 5 of DESIGN.md)
    with pytest.raises(capi.SdrppError):
        d, keep = radio.vfo_desc(sr, 250e3, 250e3, 0.0, "RAW", nco_mode=7)
        ctx.vfo_add(d, keep)
    ctx.close()


def test_reference_rotator_mode_retune_and_reset(backend):
    """setOffset in reference-rotator mode only swaps phaseDelta (rx_vfo.h:72-77): the phase state continues, the delay line keeps
    its old-increment samples — exact from the first output on; reset restarts the phase at (1, 0)."""
    from sdrplusplus_amd import capi

    sr, B = 10e6, 50000
    specs = [("USB", 1.0014e6), ("RAW", -2.34567e6)]
    x = _two_tone_mix(sr, B * 4, specs + [("USB", 1.5e6), ("RAW", 2.0e6)], 23)
    ctx, vids, chains, _ = _setup(sr, specs, B, nco_mode=1)
    for b in range(4):
        if b == 1:
            for (mode, _), vid, ch, off in zip(specs, vids, chains, (1.5e6, 2.0e6)):
                ch.set_offset(off)
                ctx.vfo_set_phase_delta(vid, *capi.design_phase_delta(-off, sr))
        if b == 3:
            ctx.vfo_reset(vids[1])
            chains[1] = S.OracleChain(sr, 250e3, 250e3, 2.0e6, None)
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        for (mode, _), vid, ch in zip(specs, vids, chains):
            oi, oa = ch.process(blk)
            gi = ctx.vfo_read_if(vid)
            assert gi.shape == oi.shape and rms(gi - oi) / rms(oi) < 2e-6, (mode, b)
            if oa is not None:
                assert rms(ctx.vfo_read(vid) - oa) < 1e-5 * max(1.0, rms(oa)), (mode, b)
    ctx.close()


def test_agc_look_ahead_follows_reference_blocks(backend):
    """loop::AGC rescans to the end of the CURRENT BLOCK when its output would clip (agc.h:91-104).  Bursts 30 dB above the settled
    level make it do so in AM (audio AGC, carrier AGC) and SSB; with sdrpp_set_reference_block the device cuts its look-ahead at the
    reference's block ends whatever the push size — same audio as the oracle driven block by block."""
    from sdrplusplus_amd import capi, radio

    sr, B, nblk = 10e6, 50000, 8
    n = B * nblk
    t = np.arange(n) / sr
    env = np.where((np.arange(n) // 37000) % 3 == 2, 1.0, 0.03)  # bursts that start and end inside blocks
    specs = [("AM", 1.2e6), ("USB", sr / 8), ("AM", -2.5e6)]
    x = np.zeros(n, dtype=np.complex128)
    x += env * 0.3 * (1 + 0.5 * np.cos(2 * np.pi * 900 * t)) * np.exp(2j * np.pi * 1.2e6 * t)
    x += env * 0.2 * (np.exp(2j * np.pi * (sr / 8 + 300) * t) + np.exp(2j * np.pi * (sr / 8 - 900) * t))
    x += env * 0.3 * (1 + 0.5 * np.cos(2 * np.pi * 1300 * t)) * np.exp(2j * np.pi * -2.5e6 * t)
    x = x.astype(np.complex64)
    outs = {}
    for name, pushes, ref_block in (("blocks", [B] * nblk, 0), ("big", [4 * B] * 2, B), ("ragged", [130000, 20000, 250000], B)):
        ctx = capi.Context(0, max_push=max(pushes))
        ctx.set_reference_block(ref_block)
        vids, chains = [], []
        for k, (mode, off) in enumerate(specs):
            if_rate, bw = radio.RADIO_DEFAULTS[mode]
            carrier = k == 2
            d, keep = radio.vfo_desc(sr, if_rate, bw, off, mode, carrier_agc=carrier)
            vids.append(ctx.vfo_add(d, keep))
            chains.append(S.OracleChain(sr, if_rate, bw, off, S.MODES[mode], carrier_agc=carrier))
        blocks = []
        for p in pushes:
            blocks += [B] * (p // B) + ([p % B] if p % B else [])
        res = _compare_streams(ctx, vids, chains, specs, x, pushes, blocks)
        for key, (e_if, e_a) in res.items():
            assert e_a < 1e-5, (name, key, e_a)  # offsets are multiples of sr/8 or FM/AM-insensitive: no rotator drift in the way
        ctx.close()


def test_ssb_arbitrary_offset_is_drift_limited(backend):
    """With an arbitrary offset the reference's rotator drifts away from its own nominal frequency; the SSB audio then
    differs by that phase drift (documented), still far below audibility."""
    sr, B = 61.44e6, 307200
    f = 1.0014e6
    t = np.arange(B * 3) / sr
    x = (0.03 * (np.exp(2j * np.pi * (f - 700) * t) + np.exp(2j * np.pi * (f + 500) * t))).astype(np.complex64)
    ctx, vids, chains, _ = _setup(sr, [("USB", f)], B)
    for b in range(3):
        ctx.push(x[b * B:(b + 1) * B])
        _, oa = chains[0].process(x[b * B:(b + 1) * B])
        ga = ctx.vfo_read(vids[0])
        assert rms(ga - oa) <= 1e-3 * max(1.0, rms(oa))
    ctx.close()


def test_push_size_invariance(backend):
    """The same stream cut into different pushes gives the same outputs (state carried exactly; only the NCO's double-precision
    phase origin moves)."""
    sr, n = 10e6, 131072
    r = np.random.default_rng(12)
    x = ((r.standard_normal(n) + 1j * r.standard_normal(n)) * 0.05).astype(np.complex64)
    outs = []
    for cuts in ([n], [50000, 50000, 31072], [1, 7, 4096, 65536, 100, 61332], [512] * 256):
        assert sum(cuts) == n
        ctx, vids, _, _ = _setup(sr, [("WFM", 1.35e6), ("RAW", -2.0e6)], 65536 if max(cuts) <= 65536 else n)
        pos, a, i = 0, [], []
        for c in cuts:
            ctx.push(x[pos:pos + c])
            pos += c
            a.append(ctx.vfo_read(vids[0]))
            i.append(ctx.vfo_read_if(vids[1]))
        outs.append((np.concatenate(a), np.concatenate(i)))
        ctx.close()
    for a, i in outs[1:]:
        assert a.shape == outs[0][0].shape and i.shape == outs[0][1].shape
        assert np.max(np.abs(a - outs[0][0])) < 2e-6
        assert np.max(np.abs(i - outs[0][1])) < 2e-7


@pytest.mark.parametrize("pipelined", [False, True])
def test_failed_push_changes_nothing(backend, pipelined, monkeypatch):
    """A push that fails half-way through its planning ("job arena exhausted", forced by the library's test hook on the 3rd pass, from its
    3rd job table on) returns an error and leaves the stream exactly as it was: the same block pushed again, and every block after it,
    is bit-identical to a run without the failure (histories, decimation offsets, resampler phases, NCO phase, frame position)."""
    from sdrplusplus_amd import capi, radio, workloads

    nv, B = 20, 20000
    x = workloads.synth(3, B * 5, seed=31, nvfo=nv)

    def run(fail):
        if fail:
            monkeypatch.setenv("SDRPP_GPU_TEST_FAIL_ARENA", "3:3")
        else:
            monkeypatch.delenv("SDRPP_GPU_TEST_FAIL_ARENA", raising=False)
        ctx = capi.Context(0, max_push=B)
        ctx.fft_configure(4096, 4096, 0, capi.design_fft_window(2, 4096))
        vids = []
        for mode, if_rate, bw, centre, _ in workloads.vfo_plan(3, nv):
            d, keep = radio.vfo_desc(10e6, if_rate, bw, centre, mode)
            vids.append(ctx.vfo_add(d, keep))
        if pipelined:
            ctx.set_pipelined(True, 1 | 4)
        outs, failures = [], 0
        for b in range(5):
            blk = x[b * B:(b + 1) * B]
            try:
                ctx.push(blk)
            except capi.SdrppError as e:
                assert "arena" in str(e)
                failures += 1
                ctx.push(blk)  # the block again: nothing of the failed attempt may be left behind
            if pipelined:
                r = ctx.result_wait(ctx.ticket())
                outs.append(([r["vfo"][v] for v in vids], r["raw"]))
                ctx.result_release(ctx.ticket())
            else:
                outs.append(([ctx.vfo_read(v).copy() for v in vids], ctx.fft_read(zoomed=False)[0].copy()))
        ctx.close()
        return outs, failures

    good, f0 = run(False)
    bad, f1 = run(True)
    assert f0 == 0 and f1 == 1
    for b, ((va, la), (vb, lb)) in enumerate(zip(good, bad)):
        assert la.shape == lb.shape and np.array_equal(la, lb), b
        for a, c in zip(va, vb):
            assert a.shape == c.shape and np.array_equal(a.view(np.uint32), c.view(np.uint32)), b


def test_retune_add_remove_reset(backend):
    sr, B = 10e6, 50000
    r = np.random.default_rng(13)
    x = ((r.standard_normal(B * 6) + 1j * r.standard_normal(B * 6)) * 0.05).astype(np.complex64)
    t = np.arange(len(x)) / sr
    x = (x + 0.2 * np.exp(1j * (2 * np.pi * 2.5e6 * t + 30 * np.sin(2 * np.pi * 2000 * t)))).astype(np.complex64)
    from sdrplusplus_amd import capi, radio

    ctx, vids, chains, _ = _setup(sr, [("WFM", 1.25e6)], B)
    # block 0: as configured; block 1: retuned (phase continuous, only phaseDelta changes: rx_vfo.h:72-77)
    for b, off in enumerate((1.25e6, 2.5e6, 2.5e6)):
        if b == 1:
            chains[0].set_offset(off)
            ctx.vfo_set_phase_delta(vids[0], *capi.design_phase_delta(-off, sr))
        ctx.push(x[b * B:(b + 1) * B])
        _, oa = chains[0].process(x[b * B:(b + 1) * B])
        ga = ctx.vfo_read(vids[0])
        # Right after a retune the reference's first decimator still holds (taps-1) samples rotated with the OLD increment; the
        # outputs whose window reaches them are handed over sample-exactly (vfo_retune_fix_kernel): no transient to skip.
        assert ga.shape == oa.shape and rms(ga - oa) < 1e-5, (b, rms(ga - oa))
        assert np.max(np.abs(ga[:400] - oa[:400])) < 2e-5, (b, np.max(np.abs(ga[:400] - oa[:400])))
    # second VFO added mid-stream starts from reset state while the first keeps streaming
    d, keep = radio.vfo_desc(sr, 250e3, 150e3, 2.5e6, "WFM")
    v2 = ctx.vfo_add(d, keep)
    c2 = S.OracleChain(sr, 250e3, 150e3, 2.5e6, S.MODES["WFM"])
    assert ctx.vfo_count() == 2
    blk = x[3 * B:4 * B]
    ctx.push(blk)
    assert rms(ctx.vfo_read(v2) - c2.process(blk)[1]) < 1e-5
    assert rms(ctx.vfo_read(vids[0]) - chains[0].process(blk)[1]) < 1e-5
    # remove the first; reset the second == brand-new chain
    ctx.vfo_remove(vids[0])
    with pytest.raises(capi.SdrppError):
        ctx.vfo_remove(vids[0])
    ctx.vfo_reset(v2)
    c3 = S.OracleChain(sr, 250e3, 150e3, 2.5e6, S.MODES["WFM"])
    blk = x[4 * B:5 * B]
    ctx.push(blk)
    assert rms(ctx.vfo_read(v2) - c3.process(blk)[1]) < 1e-5
    ctx.close()


def test_rotate_only_and_bandwidth_change(backend):
    """No power-of-two decimation (out rate > in/2 -> PowerDecimator bypassed) and RxVFO::setBandwidth."""
    from sdrplusplus_amd import capi, radio

    sr = 48000.0
    r = np.random.default_rng(14)
    x = ((r.standard_normal(24000) + 1j * r.standard_normal(24000)) * 0.1).astype(np.complex64)
    ctx = capi.Context(0, max_push=8000)
    d, keep = radio.vfo_desc(sr, 48000.0, 12000.0, sr / 8, "RAW")  # NONE mode + channel filter
    assert radio.describe(d)["predec"] == 1 and d.interp == d.decim and d.chan_ntaps > 0
    vid = ctx.vfo_add(d, keep)
    ch = S.OracleChain(sr, 48000.0, 12000.0, sr / 8, None)
    for b in range(2):
        blk = x[b * 8000:(b + 1) * 8000]
        ctx.push(blk)
        oi, _ = ch.process(blk)
        gi = ctx.vfo_read_if(vid)
        assert gi.shape == oi.shape and rms(gi - oi) / rms(oi) < 2e-6
    # resampling only (RESAMP_ONLY): 48k -> 36k, no channel filter (bandwidth == out rate)
    d2, keep2 = radio.vfo_desc(sr, 36000.0, 36000.0, 0.0, "RAW")
    assert radio.describe(d2)["predec"] == 1 and (d2.interp, d2.decim) == (3, 4) and d2.chan_ntaps == 0
    v2 = ctx.vfo_add(d2, keep2)
    c2 = S.OracleChain(sr, 36000.0, 36000.0, 0.0, None)
    blk = x[16000:24000]
    ctx.push(blk)
    o2, _ = c2.process(blk)
    g2 = ctx.vfo_read_if(v2)
    assert g2.shape == o2.shape and rms(g2 - o2) / rms(o2) < 2e-6
    ctx.close()


def test_bandwidth_change_mid_stream(backend):
    """RxVFO::setBandwidth between blocks (rx_vfo.h:60-70 -> sdrpp_vfo_set_channel_taps): wider, narrower, BYPASSED (bandwidth == IF rate exactly),
    bypassed still, back on, bypassed, on.  The channel filter's delay line survives a change of its tap count (fir.h:31-52: fewer taps keep the
    newest samples, more taps start from zeros in front); a bypassed filter is not touched by the reference and continues, switched on again, from
    the STALE delay line it had when it last ran; the demodulator's audio low-pass keeps its own delay line across the switch (its input changes
    from the filter's output to the filter's input).  IF and WFM audio against the oracle, which is pinned to the compiled reference for such
    sequences (test_oracle_vs_reference.py::test_bandwidth_change_mid_stream_bit_exact) — every block, from its first sample."""
    from sdrplusplus_amd import capi

    sr, B = 10e6, 50000
    seq = (150e3, 200e3, 90e3, 250e3, 250e3, 120e3, 250e3, 240e3)
    ctx, vids, chains, _ = _setup(sr, [("WFM", sr / 8), ("WFM", -sr / 4)], B)  # (offsets with exact phase steps: the IF is compared tightly, no rotator drift in the way)
    r = np.random.default_rng(21)
    t = np.arange(len(seq) * B) / sr
    x = (0.3 * np.exp(2j * np.pi * (sr / 8 * t + 75e3 / (2 * np.pi * 1e3) * np.sin(2 * np.pi * 1e3 * t))) + 0.2 * np.exp(2j * np.pi * (-sr / 4 + 20e3) * t)
         + 0.01 * (r.standard_normal(len(t)) + 1j * r.standard_normal(len(t)))).astype(np.complex64)
    if_rate = 250e3
    for b, bw in enumerate(seq):
        if b:
            for vid, ch in zip(vids, chains):
                ch.set_bandwidth(bw)
                fw = bw / 2.0
                ctx.vfo_set_channel_taps(vid, capi.design_low_pass(fw, fw * 0.1, if_rate) if bw != if_rate else np.zeros(0, np.float32))
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        for vid, ch in zip(vids, chains):
            oi, oa = ch.process(blk)
            gi, ga = ctx.vfo_read_if(vid), ctx.vfo_read(vid)
            assert gi.shape == oi.shape and rms(gi - oi) / rms(oi) < 2e-6, (b, bw, rms(gi - oi) / rms(oi))
            assert rms(gi[:300] - oi[:300]) / rms(oi) < 2e-6, (b, bw, "the first outputs behind the change")
            assert ga.shape == oa.shape and rms(ga - oa) < _audio_tol(oa) and rms(ga[:600] - oa[:600]) < _audio_tol(oa), (b, bw, rms(ga - oa))
    ctx.close()


def test_one_tap_filter_then_bypass_then_long_taps(backend):
    """a channel filter of ONE tap (a bandwidth far above the IF rate: taps::lowPass returns a single tap, FIR::setTaps leaves an empty delay line)
    that goes to sleep (bandwidth == IF rate) and wakes up with long taps starts from zeros in the reference — not from the delay line an EARLIER bypass left
    behind.  Sequence with an earlier bypass in it; IF against the oracle (pinned for this sequence: test_one_tap_filter_bypass_long_taps_bit_exact) from the
    first sample of every block."""
    from sdrplusplus_amd import capi

    sr, B, if_rate = 10e6, 50000, 250e3
    seq = (150e3, 250e3, 120e3, 10e6, 250e3, 150e3, 250e3, 10e6, 90e3)
    ctx, vids, chains, _ = _setup(sr, [("WFM", sr / 8), ("WFM", -sr / 4)], B)
    r = np.random.default_rng(27)
    t = np.arange(len(seq) * B) / sr
    x = (0.3 * np.exp(2j * np.pi * (sr / 8 + 2.0e3) * t) + 0.2 * np.exp(2j * np.pi * (-sr / 4 - 1.1e3) * t)
         + 0.01 * (r.standard_normal(len(t)) + 1j * r.standard_normal(len(t)))).astype(np.complex64)
    lens = []
    for b, bw in enumerate(seq):
        if b:
            taps = capi.design_low_pass(bw / 2.0, bw / 2.0 * 0.1, if_rate) if bw != if_rate else np.zeros(0, np.float32)
            lens.append(len(taps))
            for vid, ch in zip(vids, chains):
                ch.set_bandwidth(bw)
                ctx.vfo_set_channel_taps(vid, taps)
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        for vid, ch in zip(vids, chains):
            oi, _ = ch.process(blk)
            gi = ctx.vfo_read_if(vid)
            assert gi.shape == oi.shape and rms(gi - oi) / rms(oi) < 2e-6 and rms(gi[:300] - oi[:300]) / rms(oi) < 2e-6, (b, bw, rms(gi - oi) / rms(oi), rms(gi[:300] - oi[:300]) / rms(oi))
    assert 1 in lens and 0 in lens and max(lens) > 100, lens
    ctx.close()


def test_very_long_channel_filter_histories(backend):
    """Channel filters of thousands of taps (16 kHz, then 5 kHz of bandwidth at a 250 kHz IF: 1 187 and 3 800 taps; the C-ABI takes up to 4 096): the filter runs in
    the VALU form and its delay line — the stream's history, up to 8 190 floats — is carried block to block by the one-wavefront-per-job form of the carry role
    in TWO and FOUR rounds of 2 048 floats; the IF against the oracle from the first sample of every block."""
    from sdrplusplus_amd import capi

    sr, B, if_rate = 10e6, 50000, 250e3
    ctx, vids, chains, _ = _setup(sr, [("WFM", sr / 8), ("WFM", -sr / 4)], B)
    r = np.random.default_rng(23)
    seq = (150e3, 16e3, 16e3, 5e3, 5e3, 150e3)
    t = np.arange(len(seq) * B) / sr
    x = (0.3 * np.exp(2j * np.pi * (sr / 8 + 1.0e3) * t) + 0.2 * np.exp(2j * np.pi * (-sr / 4 - 0.7e3) * t)
         + 0.01 * (r.standard_normal(len(t)) + 1j * r.standard_normal(len(t)))).astype(np.complex64)
    lens = set()
    for b, bw in enumerate(seq):
        if b and bw != seq[b - 1]:
            taps = capi.design_low_pass(bw / 2.0, bw / 2.0 * 0.1, if_rate)
            lens.add(len(taps))
            for vid, ch in zip(vids, chains):
                ch.set_bandwidth(bw)
                ctx.vfo_set_channel_taps(vid, taps)
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        for vid, ch in zip(vids, chains):
            oi, _ = ch.process(blk)
            gi = ctx.vfo_read_if(vid)
            assert gi.shape == oi.shape and rms(gi - oi) / rms(oi) < 3e-6 and rms(gi[:300] - oi[:300]) / rms(oi) < 3e-6, (b, bw, rms(gi - oi) / rms(oi), rms(gi[:300] - oi[:300]) / rms(oi))
    assert 3072 < max(lens) <= 4096 and any(1024 < n <= 2048 for n in lens), lens  # (complex samples: twice that many floats)
    ctx.close()


def test_reset_while_channel_filter_is_bypassed(backend):
    """RxVFO::reset (rx_vfo.h:79-87) clears the channel filter's delay line whether the filter is in the chain or bypassed: a filter that ran, was
    bypassed, was RESET while bypassed and is switched on again starts from zeros — not from the stale delay line a bypass alone would keep
    (test_bandwidth_change_mid_stream).  After the reset the chain equals a brand-new one bypassed before its first block."""
    from sdrplusplus_amd import capi

    sr, B, if_rate = 10e6, 50000, 250e3
    ctx, vids, chains, _ = _setup(sr, [("WFM", sr / 8)], B)
    r = np.random.default_rng(22)
    t = np.arange(5 * B) / sr
    x = (0.3 * np.exp(2j * np.pi * (sr / 8 * t + 75e3 / (2 * np.pi * 1e3) * np.sin(2 * np.pi * 1e3 * t)))
         + 0.01 * (r.standard_normal(len(t)) + 1j * r.standard_normal(len(t)))).astype(np.complex64)
    vid, ch = vids[0], chains[0]

    def taps(bw):
        return capi.design_low_pass(bw / 2.0, bw / 2.0 * 0.1, if_rate) if bw != if_rate else np.zeros(0, np.float32)

    for b, bw in enumerate((150e3, 250e3, 250e3, 120e3, 120e3)):
        if b in (1, 3):
            ch.set_bandwidth(bw)
            ctx.vfo_set_channel_taps(vid, taps(bw))
        if b == 2:  # reset with the filter asleep
            ctx.vfo_reset(vid)
            ch = S.OracleChain(sr, if_rate, 150e3, sr / 8, S.MODES["WFM"])  # (the demodulator keeps the deviation it was built with: only the VFO's bandwidth moves)
            ch.set_bandwidth(if_rate)
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        oi, oa = ch.process(blk)
        gi, ga = ctx.vfo_read_if(vid), ctx.vfo_read(vid)
        assert gi.shape == oi.shape and rms(gi - oi) / rms(oi) < 2e-6 and rms(gi[:300] - oi[:300]) / rms(oi) < 2e-6, (b, bw, rms(gi - oi) / rms(oi), rms(gi[:300] - oi[:300]) / rms(oi))
        assert ga.shape == oa.shape and rms(ga - oa) < _audio_tol(oa) and rms(ga[:600] - oa[:600]) < _audio_tol(oa), (b, bw, rms(ga - oa))
    ctx.close()


def test_am_carrier_agc_and_fm_without_lowpass(backend):
    from sdrplusplus_amd import capi, radio

    sr, B = 61.44e6, 307200
    t = np.arange(2 * B) / sr
    x = (0.05 * (1 + 0.5 * np.cos(2 * np.pi * 800 * t)) * np.exp(2j * np.pi * 0.0 * t) + 0.02 * np.exp(1j * (2 * np.pi * sr / 8 * t + 2.0 * np.sin(2 * np.pi * 900 * t)))).astype(np.complex64)
    ctx = capi.Context(0, max_push=B)
    d, keep = radio.vfo_desc(sr, 15000.0, 10000.0, 0.0, "AM", carrier_agc=True)
    va = ctx.vfo_add(d, keep)
    ca = S.OracleChain(sr, 15000.0, 10000.0, 0.0, S.MODES["AM"], carrier_agc=True)
    d, keep = radio.vfo_desc(sr, 50000.0, 12500.0, sr / 8, "NFM", low_pass=False)
    vf = ctx.vfo_add(d, keep)
    cf = S.OracleChain(sr, 50000.0, 12500.0, sr / 8, S.MODES["NFM"], low_pass=False)
    for b in range(2):
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        for vid, ch in ((va, ca), (vf, cf)):
            oa = ch.process(blk)[1]
            ga = ctx.vfo_read(vid)
            assert ga.shape == oa.shape and rms(ga - oa) <= _audio_tol(oa)
    ctx.close()


class _OracleAf:
    """The radio module's AF chain on the oracle: RationalResampler<stereo_t> -> optional highPass(300,100) FIR -> optional Deemphasis
    (radio_module.h:98-110); each piece is pinned bit-exactly to the reference in test_oracle_vs_reference.py."""

    def __init__(self, af_rate, audio_rate, tau, high_pass):
        import ctypes as C
        from sdrplusplus_amd import capi

        self.o = S.oracle()
        self.rs = self.o.orc_resampler_create(S.plans_handle(), float(af_rate), float(audio_rate), 2)
        self.ratio = float(audio_rate) / float(af_rate)
        self.hp = None
        if high_pass:
            self.hp_taps = capi.design_high_pass(300.0, 100.0, audio_rate)
            self.hp = self.o.orc_fir_create(S._fp(self.hp_taps), len(self.hp_taps), 1, 2)
        self.de = self.o.orc_deemp_create(float(tau), float(audio_rate)) if tau else None

    def process(self, audio):
        a = np.ascontiguousarray(audio, dtype=np.float32)
        out = np.empty((int(len(a) * max(1.0, self.ratio)) + 64, 2), np.float32)  # the AF resampler may interpolate (15 k -> 48 k)
        n = self.o.orc_resampler_process(self.rs, len(a), S._fp(a), S._fp(out)) if len(a) else 0
        y = np.ascontiguousarray(out[:n])
        if self.hp is not None and n:
            z = np.empty_like(y)
            self.o.orc_fir_process(self.hp, n, S._fp(y), S._fp(z))
            y = z
        if self.de is not None and n:
            z = np.empty_like(y)
            self.o.orc_deemp_process(self.de, n, S._fp(y), S._fp(z))
            y = z
        return y


@pytest.mark.parametrize("mode,high_pass,tau", [("WFM", False, 50e-6), ("NFM", True, None), ("AM", False, 75e-6)])
def test_af_chain(backend, mode, high_pass, tau):
    """demodulator output -> AF resampler to 48 kHz -> [high-pass 300 Hz, 1824 taps] -> [de-emphasis] on the
    device (sdrpp_vfo_set_af) against the oracle chain, over uneven pushes; the demodulator output itself stays readable."""
    from sdrplusplus_amd import radio, workloads

    sr = 10e6 if mode == "WFM" else 2.4e6
    pushes = [50000, 12000, 7, 30011, 50000] if mode == "WFM" else [48000, 12000, 7, 30011, 48000]
    if_rate, bw = radio.RADIO_DEFAULTS[mode]
    if mode == "WFM":  # cfg 3 content: listen to one of its FM carriers
        x = workloads.synth(3, sum(pushes), seed=21, nvfo=4)
        offset = workloads.vfo_plan(3, 4)[1][3]
    else:              # a carrier with 1 kHz AM and 700 Hz FM (+ a little noise) where this VFO listens
        offset = 250e3
        rng = np.random.default_rng(21)
        n = np.arange(sum(pushes))
        dev = 2.0 if mode == "NFM" else 0.0
        tone = 0.2 * (1.0 + 0.3 * np.sin(2 * np.pi * 1000.0 * n / sr)) * np.exp(2j * np.pi * offset * n / sr + 1j * dev * np.sin(2 * np.pi * 700.0 * n / sr))
        x = (tone + 0.002 * (rng.standard_normal(len(n)) + 1j * rng.standard_normal(len(n)))).astype(np.complex64)
    ctx, vids, chains, _ = _setup(sr, [(mode, offset)], max(pushes))
    a, keep = radio.af_desc(if_rate, 48000.0, tau, high_pass)
    ctx.vfo_set_af(vids[0], a, keep)
    oaf = _OracleAf(if_rate, 48000.0, tau, high_pass)
    worst, worst_dem, pos, total = 0.0, 0.0, 0, 0
    for npush in pushes:
        blk = x[pos:pos + npush]
        pos += npush
        ctx.push(blk)
        _, oa = chains[0].process(blk)
        ref = oaf.process(oa)
        got = ctx.vfo_af_read(vids[0])
        dem = ctx.vfo_read(vids[0])
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert dem.shape == oa.shape
        total += len(ref)
        if len(ref):
            worst = max(worst, rms(got - ref) / max(1.0, rms(ref)))
        if len(oa):
            worst_dem = max(worst_dem, rms(dem - oa) / max(1.0, rms(oa)))
    assert total > 0
    assert worst < 1e-5, worst
    assert worst_dem < 1e-5, worst_dem
    # detach: the VFO keeps running without the chain
    ctx.vfo_set_af(vids[0], None)
    ctx.push(x[:5000])
    with pytest.raises(Exception):
        ctx.vfo_af_count(vids[0])
    ctx.close()


def test_packed_reads(backend):
    """int16 / int8 conversion and the SDR++-server frame built on the device from the VFO's streams — bit-exact
    against the oracle restatement (pinned to the reference's SampleStreamCompressor) applied to what the plain float reads return."""
    import ctypes as C
    from sdrplusplus_amd import radio, workloads

    sr = 10e6
    x = workloads.synth(3, 60000, seed=41, nvfo=4)
    plan = workloads.vfo_plan(3, 4)
    ctx, vids, _, _ = _setup(sr, [(plan[1][0], plan[1][3])], 60000)
    a, keep = radio.af_desc(250e3, 48000.0, 50e-6, False)
    ctx.vfo_set_af(vids[0], a, keep)
    ctx.push(x)
    o = S.oracle()
    floats = {0: ctx.vfo_read(vids[0]), 1: ctx.vfo_read_if(vids[0]).view(np.float32).reshape(-1, 2), 2: ctx.vfo_af_read(vids[0])}
    u8, fp = C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    for which, f in floats.items():
        f = np.ascontiguousarray(f, dtype=np.float32)
        n = len(f)
        assert n > 0
        # recorder-style int16 (utils/wav.cpp:166) and int8 with an explicit scale
        for pcm, scale, dt, conv in ((1, 32767.0, np.int16, o.orc_convert_16i), (0, 100.0, np.int8, o.orc_convert_8i)):
            got = ctx.vfo_read_pcm(vids[0], which, pcm, scale, n)
            ref = np.empty(n * 2, dt)
            conv.argtypes = [fp, C.c_float, C.c_int, C.c_void_p]
            conv(f.ctypes.data_as(fp), scale, n * 2, ref.ctypes.data_as(C.c_void_p))
            assert got.shape == (n, 2) and np.array_equal(got.reshape(-1), ref)
        # server frames
        for pcm in (0, 1, 2):
            got = ctx.vfo_read_compressed(vids[0], which, pcm, n)
            ref = np.zeros(8 + n * 8 + 16, np.uint8)
            nb = o.orc_compress(n, pcm, f.ctypes.data_as(fp), ref.ctypes.data_as(u8))
            assert len(got) == nb and np.array_equal(got, ref[:nb])
    ctx.close()


def test_read_many_size_query_and_streams(backend):
    """sdrpp_vfo_read_many: a NULL destination is a size query (offsets / counts / total, nothing copied: the binding sizes its buffer from
    it instead of max_push x VFOs); the packed blocks equal the per-VFO reads for every stream (demodulator output, IF, AF); a buffer
    that is too small is refused with the total in the message."""
    import ctypes as C
    from sdrplusplus_amd import capi, radio, workloads

    sr = 10e6
    x = workloads.synth(3, 60000, seed=43, nvfo=4)
    plan = workloads.vfo_plan(3, 4)
    ctx, vids, _, _ = _setup(sr, [(m, c) for m, _, _, c, _ in plan[:3]], 60000)
    a, keep = radio.af_desc(250e3, 48000.0, 50e-6, False)
    for v in vids:
        ctx.vfo_set_af(v, a, keep)
    ctx.push(x)
    n = len(vids)
    ids = (C.c_int * n)(*vids)
    for which in (None, [1] * n, [2] * n, [0, 1, 2]):
        wh = (C.c_int * n)(*which) if which is not None else None
        offs, cnts = (C.c_int64 * n)(), (C.c_int * n)()
        total = ctx._chk(ctx.L.sdrpp_vfo_read_many(ctx.h, n, ids, wh, None, 0, offs, cnts))
        single = []
        for i, v in enumerate(vids):
            w = 0 if which is None else which[i]
            single.append(ctx.vfo_read(v) if w == 0 else (ctx.vfo_read_if(v).view(np.float32).reshape(-1, 2) if w == 1 else ctx.vfo_af_read(v)))
        assert total == sum(len(b) for b in single) > 0
        assert [cnts[i] for i in range(n)] == [len(b) for b in single]
        assert [offs[i] for i in range(n)] == list(np.cumsum([0] + [len(b) for b in single[:-1]]))
        got = ctx.vfo_read_many(vids, which=which)
        for g, b in zip(got, single):
            assert g.shape == b.shape and np.array_equal(g, b)
        small = np.empty((total - 1, 2), np.float32)
        rc = ctx.L.sdrpp_vfo_read_many(ctx.h, n, ids, wh, small.ctypes.data_as(C.POINTER(C.c_float)), len(small), offs, cnts)
        assert rc < 0 and str(total) in ctx.L.sdrpp_last_error(ctx.h).decode()
    ctx.close()


def test_deferred_pushes_equal_block_by_block(backend):
    """sdrpp_set_deferred: pushes are staged and the next observing call processes them as ONE pass.  The result is the concatenation of
    what pushing and reading block by block gives — including the block-dependent AGC look-ahead (bursts make it rescan), because every
    staged push stays a reference block of its own — and the waterfall lines are bit-identical.  Configuration calls flush first."""
    from sdrplusplus_amd import capi, radio

    sr, B, nblk, N = 10e6, 50000, 8, 4096
    n = B * nblk
    t = np.arange(n) / sr
    env = np.where((np.arange(n) // 37000) % 3 == 2, 1.0, 0.03)
    specs = [("WFM", 1.35e6), ("AM", -2.5e6), ("USB", sr / 8), ("RAW", 2.0e6)]
    r = np.random.default_rng(5)
    x = (r.standard_normal(n) + 1j * r.standard_normal(n)) * 1e-3
    x += 0.2 * np.exp(1j * (2 * np.pi * 1.35e6 * t + 30 * np.sin(2 * np.pi * 2000 * t)))
    x += env * 0.3 * (1 + 0.5 * np.cos(2 * np.pi * 900 * t)) * np.exp(2j * np.pi * -2.5e6 * t)
    x += env * 0.2 * (np.exp(2j * np.pi * (sr / 8 + 300) * t) + np.exp(2j * np.pi * (sr / 8 - 900) * t))
    x = x.astype(np.complex64)
    w = capi.design_fft_window(2, N)

    def run(deferred, per_pass, pinned_async=False):
        ctx = capi.Context(0, max_push=B * per_pass)
        ctx.fft_configure(N, N, 1000, w)
        ctx.set_deferred(deferred)
        slots = [ctx.L.sdrpp_host_alloc(B * 8) for _ in range(per_pass)] if pinned_async else []
        vids = []
        for mode, off in specs:
            if_rate, bw = radio.RADIO_DEFAULTS.get(mode, (250e3, 250e3))
            d, keep = radio.vfo_desc(sr, if_rate, bw, off, mode)
            vids.append(ctx.vfo_add(d, keep))
        lines, outs = [], [[] for _ in specs]
        for b in range(nblk):
            if pinned_async:  # sdrpp_push_pinned_async: the device fetches the block itself, no wait per block (one slot per staged block)
                C.memmove(slots[b % per_pass], x[b * B:(b + 1) * B].ctypes.data, B * 8)
                ctx.push_host_ptr_async(slots[b % per_pass], B)
                if (b + 1) % per_pass == 0:
                    ctx.push_wait()
            else:
                ctx.push(x[b * B:(b + 1) * B])
            if deferred and (b + 1) % per_pass:
                assert ctx.pending() == ((b % per_pass) + 1) * B
                continue
            lines.append(ctx.fft_read(zoomed=False)[0])  # observing call: processes what is staged
            assert ctx.pending() == 0
            for k, got in enumerate(ctx.vfo_read_many(vids)):
                outs[k].append(got.copy())
        if deferred:  # staging more than max_push is refused, the staged data stays intact
            for _ in range(per_pass):
                ctx.push(x[:B])
            with pytest.raises(capi.SdrppError):
                ctx.push(x[:B])
            assert ctx.pending() == per_pass * B
        for p in slots:
            ctx.L.sdrpp_host_free(p)
        ctx.close()
        return np.concatenate(lines), [np.concatenate(o) for o in outs]

    l0, o0 = run(False, 1)
    l1, o1 = run(True, 4)
    l2, o2 = run(True, 4, pinned_async=True)
    assert np.array_equal(l1, l2)
    for a, b in zip(o1, o2):
        assert np.array_equal(a, b)
    # sdrpp_push_pinned_async degrades to sdrpp_push where it cannot do better: outside deferred mode, and for memory that is not page-locked
    ctx = capi.Context(0, max_push=B)
    ctx.fft_configure(N, N, 1000, w)
    if_rate, bw = radio.RADIO_DEFAULTS.get("WFM", (250e3, 250e3))
    d, keep = radio.vfo_desc(sr, if_rate, bw, 1.35e6, "WFM")
    vid = ctx.vfo_add(d, keep)
    got = []
    for b in range(nblk):
        blk = np.ascontiguousarray(x[b * B:(b + 1) * B])
        if b == nblk // 2:
            ctx.set_deferred(True)  # second half: deferred, from PAGEABLE memory
        ctx.push_host_ptr_async(blk.ctypes.data, B)
        ctx.push_wait()
        got.append(ctx.vfo_read(vid).copy())
    ctx.close()
    assert np.max(np.abs(np.concatenate(got) - o0[0])) < 5e-6 * max(1.0, float(np.max(np.abs(o0[0]))))
    assert l0.shape == l1.shape and l0.shape[0] == n // (N + 1000) and np.array_equal(l0, l1)
    for (mode, _), a, b in zip(specs, o0, o1):
        assert a.shape == b.shape and len(a) > 100, mode
        assert np.max(np.abs(a - b)) < 5e-6 * max(1.0, float(np.max(np.abs(a)))), (mode, float(np.max(np.abs(a - b))))


def test_pipelined_fm_back_end_is_bit_identical(backend):
    """sdrpp_set_backend_pipeline: last decimator -> resampler -> channel filter -> discriminator + audio low-pass of an FM VFO as ONE
    launch (csrc/pipe_kernels.h: four wavefronts = four stages, the streams between them in LDS rings) gives the BITS of the four
    separate launches — audio and IF — over ragged pushes (history hand-over between pushes, one-sample pushes, pushes that end inside a
    macro tile), alone in a workgroup and cut into segments with recomputed warm-up tiles (the forced segment counts), next to VFOs that
    keep their separate launches (NFM's 16/25 resampler, AM).  Against the oracle the default (pipelined) path is what every other test
    of this file runs."""
    from sdrplusplus_amd import capi, radio

    sr = 10e6
    n = 420000 if backend == "emu" else 6000000
    r = np.random.default_rng(5)
    x = (r.standard_normal(n) + 1j * r.standard_normal(n)) * 0.05
    t = np.arange(n)
    x += 0.5 * np.exp(2j * np.pi * (1.35e6 / sr * t + 3.0 * np.sin(2 * np.pi * 3e3 / sr * t)))
    x = x.astype(np.complex64)
    cuts = [50000, 1, 7, 4096, 65536, 100, 50000, 131072, 333]
    cuts.append(n - sum(cuts))
    specs = [("WFM", 1.35e6), ("WFM", -2.0e6), ("NFM", 0.4e6), ("AM", -0.7e6)]

    def run(mode):
        ctx = capi.Context(0, max_push=max(cuts))
        ctx.set_backend_pipeline(mode)
        vids = []
        for m, off in specs:
            if_rate, bw = radio.RADIO_DEFAULTS.get(m, (250e3, 250e3))
            d, keep = radio.vfo_desc(sr, if_rate, bw, off, m)
            vids.append(ctx.vfo_add(d, keep))
        pos, audio, ifs = 0, [[] for _ in vids], [[] for _ in vids]
        ctx.timing_enable(True)
        for c in cuts:
            ctx.push(x[pos:pos + c])
            pos += c
            for k, v in enumerate(vids):
                audio[k].append(ctx.vfo_read(v))
                ifs[k].append(ctx.vfo_read_if(v))
        launches = ctx.timing_read()["vfo_pipe"][1]
        ctx.close()
        return [np.concatenate(a) for a in audio], [np.concatenate(i) for i in ifs], launches

    a0, i0, l0 = run(0)
    assert l0 == 0
    for mode in (1, 2, 5) if backend == "emu" else (1, 2, 7, 64):
        a1, i1, l1 = run(mode)
        assert l1 > 0, mode  # the pipelined launch really ran
        for k, (m, _) in enumerate(specs):
            assert a1[k].shape == a0[k].shape and i1[k].shape == i0[k].shape and len(a0[k]) > 500
            assert np.array_equal(a1[k].view(np.uint32), a0[k].view(np.uint32)), (mode, m)
            assert np.array_equal(i1[k].view(np.uint32), i0[k].view(np.uint32)), (mode, m)


def test_pipelined_fm_back_end_tiny_ragged_pushes(backend):
    """The pipelined FM back end over hundreds of tiny pushes of random length (1 … 300 samples: pushes that give a stage an output or two
    and the stage behind it none, outputs that end exactly on a macro-tile boundary, empty pushes) — every sample a stage produces must reach
    its stream (the next push needs it as filter history) whether or not a later stage consumes it in this push: bit-identical to one
    launch per stage."""
    from sdrplusplus_amd import capi, radio

    sr = 10e6
    r = np.random.default_rng(23)
    cuts = [int(c) for c in r.integers(0, 300, 400)] + [40000] + [int(c) for c in r.integers(1, 120, 150)]
    n = sum(cuts)
    t = np.arange(n)
    x = ((r.standard_normal(n) + 1j * r.standard_normal(n)) * 0.05 + 0.5 * np.exp(2j * np.pi * (1.35e6 / sr * t + 3.0 * np.sin(2 * np.pi * 3e3 / sr * t)))).astype(np.complex64)

    def run(mode):
        ctx = capi.Context(0, max_push=max(cuts))
        ctx.set_backend_pipeline(mode)
        if_rate, bw = radio.RADIO_DEFAULTS.get("WFM", (250e3, 250e3))
        d, keep = radio.vfo_desc(sr, if_rate, bw, 1.35e6, "WFM")
        vid = ctx.vfo_add(d, keep)
        pos, audio, ifs = 0, [], []
        for c in cuts:
            ctx.push(x[pos:pos + c])
            pos += c
            audio.append(ctx.vfo_read(vid))
            ifs.append(ctx.vfo_read_if(vid))
        ctx.close()
        return np.concatenate(audio), np.concatenate(ifs)

    a0, i0 = run(0)
    assert len(a0) > 2000
    for mode in (1, 2):
        a1, i1 = run(mode)
        assert a1.shape == a0.shape and i1.shape == i0.shape
        assert np.array_equal(a1.view(np.uint32), a0.view(np.uint32)), mode
        assert np.array_equal(i1.view(np.uint32), i0.view(np.uint32)), mode


def test_vfo_replace_follows_the_reference_over_rate_changes(backend):
    """sdrpp_vfo_replace = RxVFO::setInSamplerate / setOutSamplerate (rx_vfo.h:35-58) and the radio module's demodulator switch (vfo_manager.cpp:52,
    radio_module.h:419-563: setOutSamplerate, then a NEW demodulator), against the COMPILED REFERENCE's own RxVFO and demodulators (oracle/_ref: the
    reference's objects carry their state across these calls the way they do, and that is what must come out): the translation's phase and the channel
    filter's delay line survive (keep & 1), the demodulator survives a change of the INPUT rate (keep & 2), decimators and resampler start cleared.
    Sequence on one VFO: WFM -> NFM -> USB -> WFM (the radio's mode switch), then the input rate 10 MS/s -> 5 MS/s (IQFrontEnd::setDecimation(2) as the
    VFO sees it) and back.  Audio of every block from its FIRST sample within 1e-5; for USB the raw IF too (the phase carried over is visible there)."""
    from sdrplusplus_amd import capi, radio

    if not S.ref_available():
        pytest.skip("oracle/_ref not built (needs the reference tree at build time)")
    sr, B = 10e6, 50000
    off = sr / 8  # (an offset whose phase steps are exact in float: no rotator drift between the closed form and the reference's recursion)
    r = np.random.default_rng(77)
    n = 9 * B
    t = np.arange(n) / sr
    x = (0.3 * np.exp(2j * np.pi * (off * t + 5e3 / (2 * np.pi * 1e3) * np.sin(2 * np.pi * 1e3 * t))) * (1.0 + 0.3 * np.cos(2 * np.pi * 700.0 * t))
         + 0.01 * (r.standard_normal(n) + 1j * r.standard_normal(n))).astype(np.complex64)
    ctx = capi.Context(0, max_push=B)
    mode, (if_rate, bw) = "WFM", radio.RADIO_DEFAULTS["WFM"]
    d, keep = radio.vfo_desc(sr, if_rate, bw, off, mode)
    vid = ctx.vfo_add(d, keep)
    ch = S.RefChain(sr, if_rate, bw, off, S.MODES[mode])
    in_sr = sr
    steps = {2: ("out", "NFM"), 4: ("out", "USB"), 6: ("out", "WFM"), 7: ("in", 5e6), 8: ("in", 10e6)}
    pos = 0
    for b in range(9):
        if b in steps:
            kind, arg = steps[b]
            if kind == "out":
                mode, (if_rate, bw) = arg, radio.RADIO_DEFAULTS[arg]
                ch.set_out_samplerate(if_rate, bw, mode=S.MODES[mode])
                d, keep = radio.vfo_desc(in_sr, if_rate, bw, off, mode)
                vid = ctx.vfo_replace(vid, d, 1, keep)
            else:
                in_sr = arg
                ch.set_in_samplerate(in_sr)
                d, keep = radio.vfo_desc(in_sr, if_rate, bw, off, mode)
                vid = ctx.vfo_replace(vid, d, 3, keep)
        # (at 5 MS/s the same samples are simply read as a stream of half the rate: the arithmetic is what is compared)
        blk = x[pos:pos + B]
        pos += B
        ctx.push(blk)
        oi, oa = ch.process(blk)
        gi, ga = ctx.vfo_read_if(vid), ctx.vfo_read(vid)
        assert gi.shape == oi.shape and ga.shape == oa.shape, (b, mode, gi.shape, oi.shape, ga.shape, oa.shape)
        head = min(len(oa), 400)
        assert rms(ga - oa) < _audio_tol(oa) and rms(ga[:head] - oa[:head]) < _audio_tol(oa), (b, mode, in_sr, rms(ga - oa), rms(ga[:head] - oa[:head]))
        assert rms(gi - oi) / max(rms(oi), 1e-9) < 5e-6 and rms(gi[:head] - oi[:head]) / max(rms(oi), 1e-9) < 5e-6, (b, mode, in_sr, "IF", rms(gi - oi) / rms(oi))
    ctx.close()
