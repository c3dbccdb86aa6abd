"""Pipelined execution (sdrpp_set_pipelined: one launch per block, the stages of consecutive blocks skewed over consecutive launches,
csrc/tick_kernels.h) against the ordinary pass, block for block and bit for bit: the roles of the tick kernel are the bodies of the
ordinary kernels, so every result must be IDENTICAL — VFO blocks, raw dB lines, zoomed lines, palette indices — whatever the block
sizes, including blocks that fall back to an ordinary pass in the middle of a pipelined run.  (The ordinary pass itself is what the
other parity tests compare with the oracle.)"""
import numpy as np
import pytest

from conftest import BACKENDS  # noqa: F401  (the `backend` fixture lives in conftest)


def _ctx_pair(cfg, nv, max_push, fft_size, data_width=600, flags=7, ref_block=0):
    from sdrplusplus_amd import capi, radio, workloads

    sr = workloads.CFG[cfg]["sr"]
    out = []
    for pipelined in (False, True):
        ctx = capi.Context(0, max_push=max_push)
        if fft_size:
            ctx.fft_configure(fft_size, fft_size, 0, capi.design_fft_window(2, fft_size))
            start, size = capi.design_waterfall_view(0.0, sr, sr, fft_size)
            ctx.fft_set_view(start, size, data_width, -120.0, 0.0)
        vids = []
        for mode, if_rate, bw, centre, _ in workloads.vfo_plan(cfg, nv):
            d, keep = radio.vfo_desc(sr, if_rate, bw, centre, mode)
            vids.append(ctx.vfo_add(d, keep))
        if ref_block:
            ctx.set_reference_block(ref_block)
        if pipelined:
            ctx.set_pipelined(True, flags)
        out.append((ctx, vids))
    return out


def _ordinary_results(ctx, vids, blk, fft):
    ctx.push(blk)
    r = {"vfo": {v: ctx.vfo_read(v).copy() for v in vids}}
    if fft:
        raw, zo, ix = ctx.fft_read()
        r.update(raw=raw, zoomed=zo, index=ix)
    return r


def _same(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (what, float(np.max(np.abs(a - b))) if a.size else 0.0)


def _compare(ref, got, fft, what):
    for v, a in ref["vfo"].items():
        _same(a, got["vfo"][v], "%s vfo %d" % (what, v))
    if fft:
        n = len(ref["raw"])
        assert got["n_lines"] == n, (what, got["n_lines"], n)
        if n:
            _same(ref["raw"], got["raw"], what + " raw lines")
            _same(ref["zoomed"], got["zoomed"], what + " zoomed")
            assert np.array_equal(ref["index"], got["index"]), what + " palette index"


@pytest.mark.parametrize("fft_size,l0_at", [(4096, None), (16384, None), (4096, "37"), (4096, "-1"), (4096, "pipe"), (4096, "fcm16w")])
def test_pipelined_equals_ordinary_wfm_bank(backend, fft_size, l0_at, monkeypatch):
    """20 WFM VFOs at 10 MS/s (matrix-core front end, four Toeplitz stages behind it) + the FFT branch (one-pass and two-pass sizes):
    uneven blocks — histories, tile and frame boundaries, a block that completes no frame, one-sample-scale blocks.  l0_at: the stage-0
    copies of a tick (landing copy, job-table upload) behind the first 37 role workgroups / behind all roles ("-1") instead of in front
    (SDRPP_GPU_TICK_L0_AT, read when a context is created)."""
    from sdrplusplus_amd import workloads

    if l0_at == "pipe":  # the FM back ends as ONE role of the tick (SDRPP_GPU_TICK_PIPE: a measurement switch, off by default) — blocks shorter than
        monkeypatch.setenv("SDRPP_GPU_TICK_PIPE", "1")  # a filter history (the 1031- and 7-sample ones and the one behind each) fall back to ordinary passes
    elif l0_at == "fcm16w":  # the front end in its 16 x 16 x 4 shape walking its tiles, the tick in the four-wavefronts-per-SIMD build of the kernel
        monkeypatch.setenv("SDRPP_GPU_TICK_FCM16W", "1")
        monkeypatch.setenv("SDRPP_GPU_TICK_FCM16W_BLOCKS", "16")  # (several tiles per workgroup at these block sizes)
    elif l0_at is not None:
        monkeypatch.setenv("SDRPP_GPU_TICK_L0_AT", l0_at)

    nv = 20 if backend == "gpu" else 17
    pushes = [50000, 1031, 20000, 7, 33333, 50000, 50000] if backend == "gpu" else [25000, 1031, 10000, 7, 16667, 25000]
    x = workloads.synth(3, sum(pushes), seed=5, nvfo=nv)
    (ca, va), (cb, vb) = _ctx_pair(3, nv, max(pushes), fft_size)
    refs, pos = [], 0
    for n in pushes:
        blk = x[pos:pos + n]
        pos += n
        refs.append(_ordinary_results(ca, va, blk, True))
        if len(refs) % 3 == 1:
            cb.push(blk)  # returns at once: one launch
        elif len(refs) % 3 == 2:
            cb.push_staged_from(blk)  # sdrpp_push_stage / _staged: the host fills the library's page-locked slot itself
        else:
            cb.push_staged_late_fill(blk)  # sdrpp_push_staged_when: another thread is still filling the slot while the call plans the block
    assert cb.ticket() == len(pushes)
    assert cb.fft_lines() == len(refs[-1]["raw"])  # host knowledge: no flush needed
    if l0_at == "fcm16w":
        cb.pipeline_flush()
        st = cb.pipeline_stats()
        assert "fcm16w_132_4" in st["roles"] and st["set2_ticks"] > 0, st
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "block %d" % t)
        cb.result_release(t)
    # the observing calls see the most recent block, as after an ordinary pass
    for v_a, v_b in zip(va, vb):
        _same(refs[-1]["vfo"][v_a], cb.vfo_read(v_b), "last block through sdrpp_vfo_read")
    raw, zo, ix = cb.fft_read()
    _same(refs[-1]["raw"], raw, "last block through sdrpp_fft_read")
    assert np.array_equal(refs[-1]["index"], ix)
    ca.close()
    cb.close()


def test_pipelined_equals_ordinary_mixed_modes(backend):
    """cfg 4 geometry: 54 VFOs NFM / AM / USB at 61.44 MS/s (long first stages on the matrix cores, three or four decimators, resampler
    or none, AM / SSB recursions with the reference's block ends) — chains of different depth share the ticks."""
    from sdrplusplus_amd import workloads

    nv = 54
    pushes = [307200, 100003, 204397, 307200] if backend == "gpu" else [153600, 50003, 102197, 153600]  # (the emulator leg is a logic check: half the samples)
    x = workloads.synth(4, sum(pushes), seed=7, nvfo=nv)
    (ca, va), (cb, vb) = _ctx_pair(4, nv, max(pushes), 0, flags=1, ref_block=50000)
    refs, pos = [], 0
    for n in pushes:
        blk = x[pos:pos + n]
        pos += n
        refs.append(_ordinary_results(ca, va, blk, False))
        cb.push(blk)
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values()))}, got, False, "block %d" % t)
        cb.result_release(t)
    ca.close()
    cb.close()


def test_pipelined_long_fft(backend):
    """2^17-point transform (the long split: 32-point column pass, 4096-point rows with the dB values in place, transpose into bin order one
    launch later) without VFOs: lines, zoomed lines (group maxima from the transpose pass) and palette indices identical to the ordinary
    pass, frames straddling the blocks."""
    from sdrplusplus_amd import workloads

    N = 1 << 17
    pushes = [100000, 70000, 131072, 50000, 180000]
    x = workloads.synth(2, sum(pushes), seed=12)
    (ca, va), (cb, vb) = _ctx_pair(2, 0, max(pushes), N, data_width=500)
    refs, pos = [], 0
    for n in pushes:
        blk = x[pos:pos + n]
        pos += n
        refs.append(_ordinary_results(ca, va, blk, True))
        cb.push(blk)
    assert sum(len(r["raw"]) for r in refs) == sum(pushes) // N
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        _compare(ref, got, True, "block %d" % t)
        cb.result_release(t)
    ca.close()
    cb.close()


def test_reference_rotator_vfos_stay_in_the_tick(backend):
    """sdrpp_vfo_desc.nco_mode = 2 (the reference's float rotator recursion, frequency_xlator.h:43-50) on the USB channels of a cfg 4 bank, pipelined: the
    recursion over the block, the plain first stages behind it and SSB's second rotator are roles of the tick (TR_ROTX16 / TR_FIRD / TR_SSBX) — no block
    falls back to an ordinary pass, every VFO's results (the closed-form NFM / AM channels' too) arrive through the result slots, bit-identical to the
    ordinary pass; a retune of such a channel between blocks (only phaseDelta changes: rx_vfo.h:72-77) included."""
    from sdrplusplus_amd import capi, radio, workloads

    sr, nv = workloads.CFG[4]["sr"], 54
    pushes = [307200, 100003, 204397, 307200, 307200] if backend == "gpu" else [38400, 12503, 25597, 38400, 38400]
    x = workloads.synth(4, sum(pushes), seed=11, nvfo=nv)
    pair = []
    for pipelined in (False, True):
        ctx = capi.Context(0, max_push=max(pushes))
        vids, usb = [], []
        for mode, if_rate, bw, centre, _ in workloads.vfo_plan(4, nv):
            d, keep = radio.vfo_desc(sr, if_rate, bw, centre, mode, nco_mode=2 if mode == "USB" else 1)
            vids.append(ctx.vfo_add(d, keep))
            if mode == "USB":
                usb.append(vids[-1])
        ctx.set_reference_block(50000)
        if pipelined:
            ctx.set_pipelined(True, 1)
        pair.append((ctx, vids, usb))
    (ca, va, ua), (cb, vb, ub) = pair
    assert len(ua) == 18
    refs, pos = [], 0
    for i, n in enumerate(pushes):
        blk = x[pos:pos + n]
        pos += n
        if i == 3:
            re, im = capi.design_phase_delta(-3.21e6, sr)
            ca.vfo_set_phase_delta(ua[2], re, im)
            cb.vfo_set_phase_delta(ub[2], re, im)
        refs.append(_ordinary_results(ca, va, blk, False))
        cb.push(blk)
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values()))}, got, False, "block %d" % t)
        cb.result_release(t)
    st = cb.pipeline_stats()
    assert st["pass_blocks"] == 0 and st["tick_blocks"] == len(pushes), st
    assert all(st["roles"].get(k, 0) > 0 for k in ("rotx16", "fird", "ssbx")), st["roles"]
    ca.close()
    cb.close()


def test_reference_rotator_context_runs_as_ticks(backend):
    """sdrpp_set_nco_mode(SDRPP_NCO_REFERENCE_ROTATOR) for the whole context, cfg 3 geometry (20 WFM VFOs at 10 MS/s): no matrix front end is left, the
    rotator role (71 KB of LDS in the three-wavefronts-per-SIMD build of the tick kernel) feeds plain first stages — still one launch per block, bit-identical
    to the ordinary pass."""
    from sdrplusplus_amd import capi, radio, workloads

    sr, nv = workloads.CFG[3]["sr"], 20
    pushes = [50000, 12345, 50000, 50000] if backend == "gpu" else [20000, 12345, 20000]
    x = workloads.synth(3, sum(pushes), seed=12, nvfo=nv)
    pair = []
    for pipelined in (False, True):
        ctx = capi.Context(0, max_push=max(pushes))
        ctx.set_nco_mode(1)
        vids = []
        for mode, if_rate, bw, centre, _ in workloads.vfo_plan(3, nv):
            d, keep = radio.vfo_desc(sr, if_rate, bw, centre, mode)
            vids.append(ctx.vfo_add(d, keep))
        if pipelined:
            ctx.set_pipelined(True, 1)
        pair.append((ctx, vids))
    (ca, va), (cb, vb) = pair
    refs, pos = [], 0
    for n in pushes:
        blk = x[pos:pos + n]
        pos += n
        refs.append(_ordinary_results(ca, va, blk, False))
        cb.push(blk)
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values()))}, got, False, "block %d" % t)
        cb.result_release(t)
    st = cb.pipeline_stats()
    assert st["pass_blocks"] == 0 and st["tick_blocks"] == len(pushes) and st["roles"].get("rotx16", 0) > 0, st
    ca.close()
    cb.close()


@pytest.mark.parametrize("sr,modes", [(2.4e6, ("WFM",)), (2.4e6, ("WFM", "NFM", "AM", "USB")), (10e6, ("WFM", "WFM", "NFM", "AM", "USB", "RAW")),
                                      (20e6, ("WFM", "AM", "DSB", "LSB")), (1e6, ("NFM", "USB"))])
def test_small_banks_stay_pipelined(backend, sr, modes):
    """What a user's session usually is — one to a handful of VFOs, far too few for the matrix front end: the VALU front ends (first stage alone, from an
    LDS window or straight from memory; first two stages fused), one VFO per job, and the one-output-per-work-item resampler are roles of the tick too
    (TR_S1_1 / TR_S1D_1 / TR_F2_1 / TR_POLY, round 5).  One launch per block, no ordinary pass, VFO outputs and waterfall lines bit-identical to the
    ordinary pass (whose jobs take up to eight VFOs each: the sums per VFO are the same)."""
    from sdrplusplus_amd import capi, radio

    B = int(sr / 200) if backend == "gpu" else int(sr / 400)  # (the emulator leg is a logic check: half the samples)
    pushes = [B, B // 3 + 1, B, B]
    r = np.random.default_rng(31)
    n = sum(pushes)
    t = np.arange(n) / sr
    x = (0.05 * (r.standard_normal(n) + 1j * r.standard_normal(n))).astype(np.complex64)
    offs = [(i - len(modes) / 2.0) * sr / (2 * len(modes) + 2) for i in range(len(modes))]
    for o in offs:
        x = (x + 0.2 * np.exp(2j * np.pi * ((o + 3e3) * t + 0.5 * np.sin(2 * np.pi * 700.0 * t)))).astype(np.complex64)
    pair = []
    for pipelined in (False, True):
        ctx = capi.Context(0, max_push=B)
        ctx.fft_configure(4096, 4096, 0, capi.design_fft_window(2, 4096))
        start, size = capi.design_waterfall_view(0.0, sr, sr, 4096)
        ctx.fft_set_view(start, size, 600, -120.0, 0.0)
        vids = []
        for mode, o in zip(modes, offs):
            if_rate, bw = radio.RADIO_DEFAULTS.get(mode, (250e3, 250e3))
            d, keep = radio.vfo_desc(sr, if_rate, bw, o, mode)
            vids.append(ctx.vfo_add(d, keep))
        if pipelined:
            ctx.set_pipelined(True, 7)
        pair.append((ctx, vids))
    (ca, va), (cb, vb) = pair
    refs, pos = [], 0
    for k in pushes:
        blk = x[pos:pos + k]
        pos += k
        refs.append(_ordinary_results(ca, va, blk, True))
        cb.push(blk)
    for tkt, ref in enumerate(refs, start=1):
        got = cb.result_wait(tkt)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), "raw": ref["raw"], "zoomed": ref["zoomed"], "index": ref["index"]}, got, True, "block %d" % tkt)
        cb.result_release(tkt)
    st = cb.pipeline_stats()
    assert st["pass_blocks"] == 0 and st["tick_blocks"] == len(pushes), st
    ca.close()
    cb.close()


def test_a_user_session_stays_pipelined(backend):
    """Everything a session switches on at once, on a bank as small as sessions are: the front end's decimation + DC blocking in front (iq_frontend.cpp:
    105-130), four radios of four modes — the USB one on the reference's rotator recursion — each with the radio module's AF chain (radio_module.h:98-110),
    waterfall lines of every block.  One launch per block throughout; AF output, lines and the pre-processed stream bit-identical to ordinary passes."""
    from sdrplusplus_amd import capi, radio

    sr_in, ratio = 9.6e6, 4
    eff = sr_in / ratio
    modes = ("WFM", "NFM", "AM", "USB")
    B = int(sr_in / 200) if backend == "gpu" else int(sr_in / 400)
    pushes = [B, B // 3 + 2, B, B, B]
    r = np.random.default_rng(33)
    n = sum(pushes)
    t = np.arange(n) / sr_in
    x = (0.05 * (r.standard_normal(n) + 1j * r.standard_normal(n)) + (0.02 + 0.01j)).astype(np.complex64)
    offs = [(i - 1.5) * eff / 6 for i in range(len(modes))]
    for o in offs:
        x = (x + 0.2 * np.exp(2j * np.pi * ((o + 2e3) * t + 0.4 * np.sin(2 * np.pi * 600.0 * t)))).astype(np.complex64)
    pair = []
    for pipelined in (False, True):
        ctx = capi.Context(0, max_push=max(pushes))
        ctx.preproc_configure(radio.plans().stages(ratio), 50.0 / eff, False)
        ctx.fft_configure(4096, 4096, 0, capi.design_fft_window(2, 4096))
        start, size = capi.design_waterfall_view(0.0, eff, eff, 4096)
        ctx.fft_set_view(start, size, 600, -120.0, 0.0)
        vids = []
        for mode, o in zip(modes, offs):
            if_rate, bw = radio.RADIO_DEFAULTS.get(mode, (250e3, 250e3))
            d, keep = radio.vfo_desc(eff, if_rate, bw, o, mode, nco_mode=2 if mode == "USB" else 0)
            vid = ctx.vfo_add(d, keep)
            a, akeep = radio.af_desc(if_rate, 48000.0, 50e-6 if mode == "WFM" else None, mode == "NFM")
            ctx.vfo_set_af(vid, a, akeep)
            vids.append(vid)
        if pipelined:
            ctx.set_pipelined(True, 15)
        pair.append((ctx, vids))
    (ca, va), (cb, vb) = pair
    refs, pos = [], 0
    for k in pushes:
        blk = x[pos:pos + k]
        pos += k
        ca.push(blk)
        ref = {"vfo": {v: ca.vfo_af_read(v).copy() for v in va}, "iq": ca.preproc_read().copy()}
        raw, zo, ix = ca.fft_read()
        ref.update(raw=raw, zoomed=zo, index=ix)
        refs.append(ref)
        cb.push(blk)
    for tkt, ref in enumerate(refs, start=1):
        got = cb.result_wait(tkt)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), **{q: ref[q] for q in ("raw", "zoomed", "index")}}, got, True, "block %d" % tkt)
        if len(ref["iq"]):
            _same(ref["iq"].view(np.float32), got["iq"].view(np.float32), "block %d pre-processed IQ" % tkt)
        cb.result_release(tkt)
    st = cb.pipeline_stats()
    assert st["pass_blocks"] == 0 and st["tick_blocks"] == len(pushes), st
    assert st["roles"].get("rotx16", 0) > 0 and st["roles"].get("ssbx", 0) > 0, st["roles"]
    ca.close()
    cb.close()


def test_pipelined_falls_back_to_ordinary_passes(backend):
    """What has no role in the tick kernel runs as an ordinary pass behind everything queued: a retune in the middle of a 20-VFO run (the hand-over
    kernel), then pipelined again — same results throughout; in front of it a two-VFO bank read back call by call after every push (its vector-unit
    front end was the other case until round 5: a role of the tick now, test_small_banks_stay_pipelined)."""
    from sdrplusplus_amd import capi, workloads

    pushes = [50000, 12345, 50000] if backend == "gpu" else [20000, 12345, 20000]
    x = workloads.synth(3, sum(pushes), seed=9, nvfo=2)
    (ca, va), (cb, vb) = _ctx_pair(3, 2, max(pushes), 4096)
    pos = 0
    for n in pushes:
        blk = x[pos:pos + n]
        pos += n
        ref = _ordinary_results(ca, va, blk, True)
        cb.push(blk)
        # an ordinary pass leaves its results on the device AND — like every block of a pipelined run — in the block's result slot
        got = cb.result_wait(cb.ticket())
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), "raw": ref["raw"], "zoomed": ref["zoomed"], "index": ref["index"]}, got, True, "fallback block")
        cb.result_release(cb.ticket())
        for v_a, v_b in zip(va, vb):
            _same(ref["vfo"][v_a], cb.vfo_read(v_b), "vfo")
        _same(ref["raw"], cb.fft_read()[0], "raw lines")
    ca.close()
    cb.close()
    nv = 20 if backend == "gpu" else 17
    pushes = [50000, 50000, 20011, 50000, 50000, 50000] if backend == "gpu" else [20000, 20000, 10011, 20000, 20000, 20000]
    x = workloads.synth(3, sum(pushes), seed=10, nvfo=nv)
    (ca, va), (cb, vb) = _ctx_pair(3, nv, max(pushes), 0, flags=1)
    pos = 0
    tickets = []
    for i, n in enumerate(pushes):
        blk = x[pos:pos + n]
        pos += n
        if i == 2:
            re, im = capi.design_phase_delta(-1.234e6, 10e6)
            ca.vfo_set_phase_delta(va[3], re, im)
            cb.vfo_set_phase_delta(vb[3], re, im)
        ref = _ordinary_results(ca, va, blk, False)
        cb.push(blk)
        tickets.append((cb.ticket(), ref))
    for t, ref in tickets:  # every block has its results, the one that ran as an ordinary pass (the retune hand-over) included
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values()))}, got, False, "block %d" % t)
        cb.result_release(t)
    for v_a, v_b in zip(va, vb):
        _same(tickets[-1][1]["vfo"][v_a], cb.vfo_read(v_b), "last block")
    ca.close()
    cb.close()


def test_pipelined_result_slots_and_modes(backend):
    from sdrplusplus_amd import capi, workloads

    nv, B = (20, 4000) if backend == "gpu" else (3, 2000)  # (gpu: >= 17 VFOs = the matrix-core front end; the emulator: a small bank, which runs as launches of the pipeline as well)
    S, M = capi.RESULT_SLOTS, capi.RESULT_SLOTS * capi.GROUP_MAX
    x = workloads.synth(3, B * 16, seed=3, nvfo=nv)
    blk_of = lambda i: x[(i % 16) * B:(i % 16 + 1) * B]
    (ca, va), (cb, vb) = _ctx_pair(3, nv, B, 0, flags=1)
    with pytest.raises(capi.SdrppError):
        cb.set_deferred(True)
    refs = []
    for i in range(S + 4):
        refs.append(_ordinary_results(ca, va, blk_of(i), False))
        cb.push(blk_of(i))
    # the result ring holds AT LEAST the last S launches of max_push samples (round 6: a byte ring — more when the launches deliver less than the
    # largest possible; the entries themselves are a ring of S x GROUP_MAX blocks): the newest S blocks are all there
    for t in (5, 7, S + 4):
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, refs[t - 1]["vfo"].values()))}, got, False, "block %d" % t)
        if t != 7:
            cb.result_release(t)
    assert cb.result_ready(S + 3) in (True, False)
    cb.pipeline_flush()
    cb.sync()
    assert cb.result_ready(S + 3)
    # block 7 is still held: pushing on until the ring comes round to it must fail (at the latest when its ENTRY comes round, M pushes after it),
    # change nothing, and work again after the release; by then the oldest blocks are gone
    n0 = S + 4
    failed_at = None
    for i in range(n0, M + 12):
        try:
            cb.push(blk_of(i))
        except capi.SdrppError:
            failed_at = i + 1
            break
        _ordinary_results(ca, va, blk_of(i), False)
    assert failed_at is not None and 7 + S <= failed_at <= 7 + M, failed_at
    assert cb.ticket() == failed_at - 1  # the push did not happen
    with pytest.raises(capi.SdrppError):
        cb.result_wait(2)  # long overwritten
    cb.result_release(7)
    ref = _ordinary_results(ca, va, blk_of(failed_at - 1), False)
    cb.push(blk_of(failed_at - 1))
    got = cb.result_wait(failed_at)
    _compare({"vfo": dict(zip(vb, ref["vfo"].values()))}, got, False, "the block after the release")
    S = failed_at - 7  # (the code below names blocks relative to S + 7)
    # leaving the mode flushes; ordinary passes continue the same streams
    cb.result_release(S + 7)
    cb.set_pipelined(False)
    blk = blk_of(S + 7)
    ref = _ordinary_results(ca, va, blk, False)
    cb.push(blk)
    for v_a, v_b in zip(va, vb):
        _same(ref["vfo"][v_a], cb.vfo_read(v_b), "after leaving pipelined mode")
    ca.close()
    cb.close()


def test_results_survive_growing_result_slots(backend):
    """A host keeps tickets across a change of the configuration (IQFrontEnd: pendingTickets across tempStop / addVFO).  Adding a VFO makes
    every block's results larger than the page-locked slots were sized for: the slots grow — and the results of the blocks pushed before,
    complete or still in flight, HELD by the host or not, must all still be there afterwards (they used to be dropped: the next
    sdrpp_result_wait failed with NOT_FOUND and the pipelined worker of the C++ front end died)."""
    from sdrplusplus_amd import capi, radio, workloads

    nv, B = (20, 4000) if backend == "gpu" else (17, 2000)
    plan = workloads.vfo_plan(3, nv + 12)
    x = workloads.synth(3, B * 12, seed=14, nvfo=nv + 12)
    (ca, va), (cb, vb) = _ctx_pair(3, nv, B, 4096, flags=7)
    refs = []
    for i in range(5):
        blk = x[i * B:(i + 1) * B]
        refs.append(_ordinary_results(ca, va, blk, True))
        cb.push(blk)
    held = cb.result_wait(2, copy=False)  # handed out and NOT released: its arrays point into the slot's page-locked memory
    held_copy = {v: a.copy() for v, a in held["vfo"].items()}
    st0 = cb.pipeline_stats()
    # (blocks this small leave the piped back ends less than a filter history per block: every other one runs as an ordinary pass)
    assert st0["tick_blocks"] + st0["pass_blocks"] == 5 and st0["tick_blocks"] >= 1, st0
    # twelve more VFOs on both contexts: 1.6 x the result bytes per block — more than the slots' 12.5 % slack
    for mode, if_rate, bw, centre, _ in plan[nv:]:
        for ctx, vids in ((ca, va), (cb, vb)):
            d, keep = radio.vfo_desc(10e6, if_rate, bw, centre, mode)
            vids.append(ctx.vfo_add(d, keep))
    for i in range(5, 9):
        blk = x[i * B:(i + 1) * B]
        refs.append(_ordinary_results(ca, va, blk, True))
        cb.push(blk)
    # the held block's memory is still the host's (old buffer kept alive), bit for bit
    for v, a in held["vfo"].items():
        assert np.array_equal(a.view(np.uint32), held_copy[v].view(np.uint32))
    cb.result_release(2)
    for t, ref in enumerate(refs, start=1):
        if t == 2:
            continue
        got = cb.result_wait(t)
        n_v = nv if t <= 5 else nv + 12
        assert len(got["vfo"]) == n_v, (t, len(got["vfo"]))
        _compare({"vfo": dict(zip(vb[:n_v], [ref["vfo"][v] for v in va[:n_v]])), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "block %d" % t)
        cb.result_release(t)
    ca.close()
    cb.close()


@pytest.mark.parametrize("high_pass,tau", [(False, 50e-6), (True, None)])
def test_pipelined_equals_ordinary_with_af_chain(backend, high_pass, tau):
    """The radio's AF chain (radio_module.h:98-110: the AF resampler is ALWAYS on) behind every VFO of a pipelined bank: pre-decimators and
    the 96 / 125 resampler, the 1824-tap high-pass or the de-emphasis run as roles of the ticks too (de-emphasis: two levels, its state handed
    from block to block through two alternating slots) — no block falls back to an ordinary pass, and what result flag 1 delivers is the AF
    chain's output (what the radio's audio stream carries), bit-identical to sdrpp_vfo_af_read after ordinary passes."""
    from sdrplusplus_amd import radio, workloads

    nv = 20 if backend == "gpu" else 17
    pushes = [50000, 30011, 50000, 20000, 50000, 50000] if backend == "gpu" else [25000, 15011, 25000, 10000, 25000]
    x = workloads.synth(3, sum(pushes), seed=15, nvfo=nv)
    (ca, va), (cb, vb) = _ctx_pair(3, nv, max(pushes), 4096, flags=7)
    for ctx, vids in ((ca, va), (cb, vb)):
        for vid in vids:
            a, keep = radio.af_desc(250e3, 48000.0, tau, high_pass)
            ctx.vfo_set_af(vid, a, keep)
    refs, pos = [], 0
    for n in pushes:
        blk = x[pos:pos + n]
        pos += n
        ca.push(blk)
        r = {"vfo": {v: ca.vfo_af_read(v).copy() for v in va}}
        raw, zo, ix = ca.fft_read()
        r.update(raw=raw, zoomed=zo, index=ix)
        refs.append(r)
        cb.push(blk)
    total = 0
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "block %d" % t)
        total += sum(len(a) for a in got["vfo"].values())
        cb.result_release(t)
    assert total > 0
    st = cb.pipeline_stats()  # (roles are counted as their ticks are launched: all of them by now)
    assert st["tick_blocks"] == len(pushes) and st["pass_blocks"] == 0, st
    assert "polyc" in st["roles"] and (("firb_c" in st["roles"]) if high_pass else ("deemp_p0" in st["roles"] and "deemp_p1" in st["roles"])), st
    for v_a, v_b in zip(va, vb):  # the observing calls see the most recent block's AF output
        _same(refs[-1]["vfo"][v_a], cb.vfo_af_read(v_b), "last block through sdrpp_vfo_af_read")
    ca.close()
    cb.close()


def test_pipelined_equals_ordinary_with_waterfall_state(backend):
    """The waterfall widget's display state (sdrpp_wf_configure: raw-line ring, FFT trace smoothing / hold — waterfall.cpp:875-941) no longer
    throws a pipelined block onto an ordinary pass: ring store and trace update are roles of the ticks.  Same lines, same trace, same re-raster
    as after ordinary passes, with the ring wrapping, smoothing and hold switched on mid-stream and blocks that complete 0, 1 and several lines."""
    from sdrplusplus_amd import capi, workloads

    sr, N, W, H = 10e6, 4096, 600, 5
    pushes = [N // 2, N, N * 3 + 17, 100, N * 5, N * 2]
    x = workloads.synth(2, sum(pushes), seed=33)
    ctxs = []
    for pipelined in (False, True):
        ctx = capi.Context(0, max_push=N * 5)
        ctx.fft_configure(N, N, 0, capi.design_fft_window(2, N))
        start, size = capi.design_waterfall_view(1.0e6, 4.0e6, sr, N)
        ctx.fft_set_view(start, size, W, -110.0, -10.0)
        ctx.wf_configure(H)
        if pipelined:
            ctx.set_pipelined(True, 6)
        ctxs.append(ctx)
    ca, cb = ctxs
    pos = 0
    for step, npush in enumerate(pushes):
        if step == 2:
            for c_ in ctxs:
                c_.wf_set_smoothing(True, 0.25)
        if step == 3:
            for c_ in ctxs:
                c_.wf_set_hold(True, 1.5)
        blk = x[pos:pos + npush]
        pos += npush
        ca.push(blk)
        cb.push(blk)
        if step in (1, 4, 5):  # observed only now and then: in between the trace roles of consecutive blocks run back to back
            raw, zo, ix = ca.fft_read()
            got = cb.result_wait(cb.ticket())
            _compare({"vfo": {}, "raw": raw, "zoomed": zo, "index": ix}, got, True, "block %d" % (step + 1))
            cb.result_release(cb.ticket())
            la, ha = ca.wf_latest(W)
            lb, hb = cb.wf_latest(W)
            _same(la, lb, "latest FFT trace")
            if step > 3:
                _same(ha, hb, "hold trace")
            s2, z2 = capi.design_waterfall_view(-2.0e6, 1.0e6, sr, N)
            (ra, na), (rb, nb) = ca.wf_raster(s2, z2, 400, -100.0, 0.0), cb.wf_raster(s2, z2, 400, -100.0, 0.0)
            assert na == nb and ra.shape == rb.shape and np.array_equal(ra, rb), "re-raster of the ring"
    st = cb.pipeline_stats()
    assert st["pass_blocks"] == 0 and "wf_ring" in st["roles"] and "wf_trace" in st["roles"], st
    ca.close()
    cb.close()


@pytest.mark.parametrize("ratio,dc,conj", [(2, True, True), (4, False, False), (1, False, True), (1, True, False)])
def test_pipelined_equals_ordinary_with_preproc_chain(backend, ratio, dc, conj):
    """IQFrontEnd's pre-processing chain (setDecimation / setDCBlocking / setInvertIQ, iq_frontend.cpp:32-39 and :105-130) in front of a
    pipelined bank: its decimator stages, the DC blocker's two passes and the conjugate are the first levels of every block, the FFT branch
    and the VFO bank start that many ticks later.  Pre-processed stream (result flag 8: what bindIQStream consumers receive), VFO blocks and
    lines bit-identical to ordinary passes; no block falls back except one the decimator swallows whole."""
    from sdrplusplus_amd import capi, radio, workloads

    nv = 20 if backend == "gpu" else 17
    eff = 10e6
    scale = 1 if backend == "gpu" else 2
    pushes = [n * ratio // scale for n in (50000, 30011, 50000, 20000, 50000)] + [ratio - 1 if ratio > 1 else 7, 50000 * ratio // scale]
    x0 = workloads.synth(3, (sum(pushes) + ratio - 1) // ratio + 8, seed=17, nvfo=nv)
    x = np.repeat(x0, ratio)[:sum(pushes)].astype(np.complex64) + np.complex64(0.05 + 0.03j)  # (a crude oversampled copy: content does not matter here)
    if conj:
        x = np.conj(x)
    ctxs = []
    for pipelined in (False, True):
        ctx = capi.Context(0, max_push=max(pushes))
        stages = radio.plans().stages(ratio) if ratio > 1 else []
        ctx.preproc_configure(stages, 50.0 / eff if dc else 0.0, conj)
        ctx.fft_configure(4096, 4096, 0, capi.design_fft_window(2, 4096))
        start, size = capi.design_waterfall_view(0.0, eff, eff, 4096)
        ctx.fft_set_view(start, size, 600, -120.0, 0.0)
        vids = []
        for mode, if_rate, bw, centre, _ in workloads.vfo_plan(3, nv):
            d, keep = radio.vfo_desc(eff, if_rate, bw, centre, mode)
            vids.append(ctx.vfo_add(d, keep))
        if pipelined:
            ctx.set_pipelined(True, 15)
        ctxs.append((ctx, vids))
    (ca, va), (cb, vb) = ctxs
    refs, pos = [], 0
    for n in pushes:
        blk = x[pos:pos + n]
        pos += n
        r = _ordinary_results(ca, va, blk, True)
        r["iq"] = ca.preproc_read().copy()
        refs.append(r)
        cb.push(blk)
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "block %d" % t)
        if len(ref["iq"]):
            assert got["iq"] is not None and got["iq"].shape == ref["iq"].shape, (t, None if got["iq"] is None else got["iq"].shape, ref["iq"].shape)
            _same(ref["iq"].view(np.float32), got["iq"].view(np.float32), "block %d pre-processed IQ" % t)
        cb.result_release(t)
    _same(ca.preproc_read().view(np.float32), cb.preproc_read().view(np.float32), "last block through sdrpp_preproc_read")
    st = cb.pipeline_stats()
    assert st["pass_blocks"] <= (1 if ratio > 1 else 0) and st["tick_blocks"] >= len(pushes) - 1, st
    if dc:
        assert "dc_p0" in st["roles"] and "dc_p1" in st["roles"], st
    ca.close()
    cb.close()


# ---- several blocks per launch (sdrpp_set_pipeline_group) -----------------------------------------------------------------------------------
def _device_copy_of(ctx, x):
    """x (complex64) in memory of the context's device, through the C-ABI's own allocator / copy (works on the emulator build too)."""
    import ctypes as C

    x = np.ascontiguousarray(x, dtype=np.complex64)
    p = ctx.L.sdrpp_device_alloc(ctx.h, x.nbytes)
    assert p
    ctx._chk(ctx.L.sdrpp_device_copy(ctx.h, C.c_void_p(p), C.c_void_p(x.ctypes.data), x.nbytes, 0))
    return p


@pytest.mark.parametrize("feed", ["host", "device", "staged", "int16"])
def test_grouped_launches_equal_block_by_block(backend, feed):
    """sdrpp_set_pipeline_group(3): the pushes of a group are planned as ONE block whose reference-block ends are the push ends and go out as ONE
    launch — and every push still has its own ticket and results, bit-identical to the ordinary pass block by block: 20 / 17 WFM VFOs + FFT, uneven
    blocks (a block that completes no frame, a 7-sample block, a group the next block does not fit into: max_push), every way a block can arrive."""
    from sdrplusplus_amd import workloads

    nv = 20 if backend == "gpu" else 17
    pushes = [50000, 1031, 20000, 7, 33333, 50000, 50000, 12000] if backend == "gpu" else [25000, 1031, 10000, 7, 16667, 25000, 25000, 6000]
    x = workloads.synth(3, sum(pushes), seed=5, nvfo=nv)
    if feed == "int16":  # (what file_source reads: int16 pairs, converted by the landing copy — the ordinary pass gets the same samples)
        xi = np.round(x.view(np.float32) * 20000.0).astype(np.int16)
        x = (xi.astype(np.float32) / 32768.0).view(np.complex64)
    cap = max(pushes) * 2 + 2000  # room for two large blocks: the third of a would-be group of three goes out with the next one
    (ca, va), (cb, vb) = _ctx_pair(3, nv, cap, 4096)
    cb.set_pipeline_group(3)
    dev = _device_copy_of(cb, x) if feed == "device" else None
    refs, pos = [], 0
    for i, n in enumerate(pushes):
        blk = x[pos:pos + n]
        refs.append(_ordinary_results(ca, va, blk, True))
        if feed == "host":
            cb.push(blk)
        elif feed == "device":
            cb.push_device(dev + 8 * pos, n)
        elif feed == "int16":
            cb.push_int16(xi[2 * pos:2 * (pos + n)])
        elif i % 2 == 0:
            cb.push_staged_from(blk)
        else:
            cb.push_staged_late_fill(blk)
        pos += n
        assert cb.ticket() == i + 1
    gs = cb.pipeline_group_stats()
    assert gs["held"] > 0 and gs["multi_groups"] >= 2, gs  # (8 pushes: the last group is still open)
    assert not cb.result_ready(len(pushes))  # held back: not planned yet
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)  # (the last tickets: the open group goes out as it stands)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "block %d" % t)
        cb.result_release(t)
    gs = cb.pipeline_group_stats()
    st = cb.pipeline_stats()
    assert gs["held"] == 0 and gs["largest"] == 3 and gs["groups"] < len(pushes), gs
    assert st["pass_blocks"] == 0 and st["tick_blocks"] == len(pushes) and st["ticks"] < 2 * len(pushes) + 12, st
    if dev:
        cb.L.sdrpp_device_free(cb.h, dev)
    ca.close()
    cb.close()


@pytest.mark.parametrize("feed", ["host", "device"])
def test_largest_group_of_reference_size_blocks(backend, feed):
    """SDRPP_GROUP_MAX blocks of the reference's size in ONE launch (what a file played faster than the device works comes to): 2 x GROUP_MAX + 3 pushes
    of uneven small blocks, every push with its own ticket, VFO blocks and lines, bit-identical to the ordinary pass block by block."""
    from sdrplusplus_amd import capi, workloads

    K = capi.GROUP_MAX
    nv = 12 if backend == "gpu" else 5
    base = 12000 if backend == "gpu" else 2400
    rng = np.random.default_rng(11)
    pushes = [int(base + rng.integers(-base // 10, base // 10)) for _ in range(2 * K + 3)]
    x = workloads.synth(3, sum(pushes), seed=9, nvfo=nv)
    cap = sum(sorted(pushes)[-K:]) + 16
    (ca, va), (cb, vb) = _ctx_pair(3, nv, cap, 4096)
    cb.set_pipeline_group(K)
    dev = _device_copy_of(cb, x) if feed == "device" else None
    refs, pos = [], 0
    for i, n in enumerate(pushes):
        blk = x[pos:pos + n]
        refs.append(_ordinary_results(ca, va, blk, True))
        if feed == "host":
            cb.push(blk)
        else:
            cb.push_device(dev + 8 * pos, n)
        pos += n
        assert cb.ticket() == i + 1
    gs = cb.pipeline_group_stats()
    assert gs["held"] == 3 and gs["largest"] == K and gs["groups"] == 2, gs
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "block %d" % t)
        cb.result_release(t)
    gs = cb.pipeline_group_stats()
    assert gs["held"] == 0 and gs["groups"] == 3, gs
    if dev:
        cb.L.sdrpp_device_free(cb.h, dev)
    ca.close()
    cb.close()


def test_grouped_launches_mixed_modes_af_and_reference_blocks(backend):
    """Groups of four over a cfg 4 bank (NFM / AM / USB: chains of different depth and rate, AGC look-ahead that follows the reference's blocks inside every
    push) with the radio's AF chain on some VFOs: per-push shares of every output — the AF chain's 48 kHz stream included — identical to the ordinary pass;
    a retune and a bandwidth change between pushes send the open group on its way first."""
    from sdrplusplus_amd import capi, radio, workloads

    sr, nv = workloads.CFG[4]["sr"], 27
    pushes = [153600, 50003, 102197, 153600, 80000, 153600, 153600] if backend == "gpu" else [38400, 12503, 25597, 38400, 20000, 38400, 38400]
    x = workloads.synth(4, sum(pushes), seed=9, nvfo=nv)
    pair = []
    for pipelined in (False, True):
        ctx = capi.Context(0, max_push=sum(pushes))
        vids, with_af, with_deemph = [], set(), set()
        for i, (mode, if_rate, bw, centre, _) in enumerate(workloads.vfo_plan(4, nv)):
            d, keep = radio.vfo_desc(sr, if_rate, bw, centre, mode)
            vids.append(ctx.vfo_add(d, keep))
            if i % 4 == 1:
                a, akeep = radio.af_desc(if_rate, 48000.0, 50e-6 if mode == "NFM" else None, high_pass=(mode == "NFM"))
                ctx.vfo_set_af(vids[-1], a, akeep)
                with_af.add(vids[-1])
                if mode == "NFM":
                    with_deemph.add(vids[-1])
        ctx.set_reference_block(int(sr / 400))
        if pipelined:
            ctx.set_pipelined(True, 1)
            ctx.set_pipeline_group(4)
        pair.append((ctx, vids, with_af, with_deemph))
    (ca, va, afa, _), (cb, vb, _, deb) = pair
    assert len(afa) >= 6 and len(deb) >= 2 and len(deb) < len(afa)
    refs, pos = [], 0
    for i, n in enumerate(pushes):
        blk = x[pos:pos + n]
        pos += n
        if i == 2:
            re, im = capi.design_phase_delta(-1.234e6, sr)
            ca.vfo_set_phase_delta(va[3], re, im)
            cb.vfo_set_phase_delta(vb[3], re, im)
        if i == 5:
            taps = capi.design_low_pass(3000.0, 600.0, 50000.0)
            ca.vfo_set_channel_taps(va[0], taps)
            cb.vfo_set_channel_taps(vb[0], taps)
        ca.push(blk)
        r = {}
        for v_a, v_b in zip(va, vb):
            r[v_b] = (ca.vfo_af_read(v_a) if v_a in afa else ca.vfo_read(v_a)).copy()
        refs.append(r)
        cb.push(blk)
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        for v, a in ref.items():
            if v in deb:
                # the de-emphasis is a recursion evaluated as a two-level scan whose segments start where a LAUNCH starts (vfo_deemph_body): the carry into a
                # segment rounds differently when the cuts move — equal to the last bits, not bit for bit (the same holds for a deferred pass)
                b = got["vfo"][v]
                assert a.shape == b.shape and np.max(np.abs(a - b), initial=0.0) <= 2e-6 * max(1e-6, float(np.max(np.abs(a), initial=0.0))), ("block %d vfo %d (de-emphasis)" % (t, v))
            else:
                _same(a, got["vfo"][v], "block %d vfo %d" % (t, v))
        cb.result_release(t)
    gs = cb.pipeline_group_stats()
    assert gs["multi_groups"] >= 2 and gs["largest"] <= 4, gs
    ca.close()
    cb.close()


def test_held_staged_pushes_with_words_that_live_until_the_launch(backend):
    """sdrpp_set_pipeline_group flag 2 (what sdrpp_gpu::IQFrontEnd runs with): a staged push that is merely HELD returns while its slot is still being
    filled — the word it handed over lives until the launch — and sdrpp_pipeline_launch_held sends the group on its way without draining the pipeline.
    The launch waits for every word; results are those of block-by-block processing."""
    import ctypes as C
    import threading
    import time
    from sdrplusplus_amd import capi, workloads

    nv = 6 if backend == "gpu" else 3
    pushes = [20000, 7000, 20000, 12345, 20000] if backend == "gpu" else [5000, 1700, 5000, 3086, 5000]
    x = workloads.synth(3, sum(pushes), seed=21, nvfo=nv)
    (ca, va), (cb, vb) = _ctx_pair(3, nv, sum(pushes[:3]) + 64, 4096)
    cb.set_pipeline_group(3, adaptive=False, stable_words=True)
    words = [C.c_uint32(1) for _ in pushes]
    fills, refs, pos = [], [], 0
    for i, n in enumerate(pushes):
        blk = np.ascontiguousarray(x[pos:pos + n])
        refs.append(_ordinary_results(ca, va, blk, True))
        slot = capi.c_float_p()
        cb._chk(cb.L.sdrpp_push_stage(cb.h, n, C.byref(slot)))
        dst = C.cast(slot, C.c_void_p).value

        def fill(dst=dst, blk=blk, w=words[i]):
            time.sleep(0.05)
            C.memmove(dst, blk.ctypes.data, blk.nbytes)
            w.value = 0

        th = threading.Thread(target=fill)
        t0 = time.perf_counter()
        th.start()
        cb._chk(cb.L.sdrpp_push_staged_when(cb.h, n, C.byref(words[i])))
        dt = time.perf_counter() - t0
        held = cb.pipeline_group_stats()["held"]
        if (i + 1) % 3:   # held: the call came back although the slot was not filled yet
            assert held == (i + 1) % 3 and dt < 0.04 and words[i].value == 1, (i, held, dt)
        else:             # the third push of a group launches it: it waited for all three words
            assert held == 0 and all(w.value == 0 for w in words[:i + 1]), (i, held)
        fills.append(th)
        pos += n
    assert cb.pipeline_group_stats()["held"] == 2
    ticks0 = cb.pipeline_stats()["ticks"]
    cb.pipeline_launch_held()   # waits for the two words, ONE launch, no drain
    assert cb.pipeline_group_stats()["held"] == 0 and cb.pipeline_stats()["ticks"] == ticks0 + 1
    cb.pipeline_launch_held()   # nothing held: no-op
    assert cb.pipeline_stats()["ticks"] == ticks0 + 1
    for th in fills:
        th.join()
    for t, ref in enumerate(refs, start=1):
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "block %d" % t)
        cb.result_release(t)
    ca.close()
    cb.close()


def test_grouped_results_hold_release_and_adaptive(backend):
    """The result slots belong to launch groups: blocks of a group can be held and released in any order, a slot still held when its turn comes round again
    fails the push (and nothing else); adaptive grouping on an idle device sends every push at once."""
    from sdrplusplus_amd import capi, workloads

    nv, B = (17, 12000) if backend == "gpu" else (3, 3000)  # (the emulator: a small bank — it runs as launches of the pipeline too — and short blocks)
    x = workloads.synth(3, 8 * B, seed=3, nvfo=nv)
    (ca, va), (cb, vb) = _ctx_pair(3, nv, 4 * B, 4096, flags=3)
    cb.set_pipeline_group(2)
    for i in range(4):
        cb.push(x[i * B:(i + 1) * B])
    r2 = cb.result_wait(2, copy=False)
    r1 = cb.result_wait(1, copy=False)
    a1 = {v: a.copy() for v, a in r1["vfo"].items()}
    cb.result_release(2)
    # the result ring holds at least RESULT_SLOTS launches of max_push samples (more of these small groups), the ring of result entries RESULT_SLOTS x
    # GROUP_MAX blocks.  When either comes round to block 1, which is still held, the push that sends that group out fails (and takes the group's
    # tickets back); the stream goes on once the block is released
    i, failed_at = 4, None
    while failed_at is None and i < capi.GROUP_MAX * capi.RESULT_SLOTS + 8:
        t0 = cb.ticket()
        try:
            cb.push(x[(i % 8) * B:(i % 8 + 1) * B])
        except capi.SdrppError:
            failed_at = cb.pipeline_group_stats()["groups"]
            assert cb.ticket() == t0 - 1 and cb.pipeline_group_stats()["held"] == 0  # the group (this push and the one held before it) did not happen
            break
        i += 1
        if i >= 8 and not cb.pipeline_group_stats()["held"]:
            for t in (i - 5, i - 4):  # (blocks of groups that have gone out: asking for them flushes nothing that is held)
                cb.result_wait(t)
                cb.result_release(t)
    assert failed_at is not None and capi.RESULT_SLOTS <= failed_at <= capi.GROUP_MAX * capi.RESULT_SLOTS // 2 + 2, failed_at
    for v in vb:
        assert np.array_equal(a1[v], r1["vfo"][v])  # untouched while held
    cb.result_release(1)
    t0 = cb.ticket()
    cb.push(x[0:B])
    cb.push(x[B:2 * B])
    t = cb.ticket()
    assert t == t0 + 2
    cb.result_wait(t)
    cb.result_release(t)
    # adaptive: the emulator / an idle device has nothing in flight -> every push is its own launch
    g0 = cb.pipeline_group_stats()
    cb.set_pipeline_group(4, adaptive=True)
    for i in range(3):
        cb.push(x[i * B:(i + 1) * B])
        cb.sync()
    g1 = cb.pipeline_group_stats()
    assert g1["groups"] - g0["groups"] == 3 and g1["multi_groups"] == g0["multi_groups"], (g0, g1)
    ca.close()
    cb.close()


def test_result_ring_grows_under_held_and_uncollected_results(backend):
    """A VFO added in the middle of a pipelined run makes the result ring grow (tick_results_ensure): a block the host is HOLDING keeps the address it was
    given (its ring stays alive until the release), blocks pushed before the change and not yet collected follow their bytes into the new ring, blocks
    pushed after it carry the new VFO — everything equal to the ordinary pass block by block."""
    from sdrplusplus_amd import radio, workloads

    nv = 4 if backend == "gpu" else 2
    B = 20000 if backend == "gpu" else 5000
    x = workloads.synth(3, 8 * B, seed=31, nvfo=nv + 6)
    (ca, va), (cb, vb) = _ctx_pair(3, nv, B, 4096)
    cb.set_pipeline_group(2)
    refs = []
    for i in range(4):
        blk = x[i * B:(i + 1) * B]
        refs.append((list(va), list(vb), _ordinary_results(ca, va, blk, True)))
        cb.push(blk)
    held = cb.result_wait(1, copy=False)          # block 1 handed out and HELD across the change (views into the old ring)
    held_copy = {v: a.copy() for v, a in held["vfo"].items()}
    # six more VFOs: every launch now delivers more than twice the bytes the ring was sized for
    sr = workloads.CFG[3]["sr"]
    keep = []
    for mode, if_rate, bw, centre, _ in workloads.vfo_plan(3, nv + 6)[nv:]:
        d, k = radio.vfo_desc(sr, if_rate, bw, centre, mode)
        va.append(ca.vfo_add(d, k))
        vb.append(cb.vfo_add(d, k))
        keep.append(k)
    for i in range(4, 8):
        blk = x[i * B:(i + 1) * B]
        refs.append((list(va), list(vb), _ordinary_results(ca, va, blk, True)))
        cb.push(blk)
    for v, a in held["vfo"].items():               # the held block's memory is still the block's
        _same(held_copy[v], a, "held block vfo %d after the ring grew" % v)
    _compare({"vfo": dict(zip(refs[0][1], refs[0][2]["vfo"].values())), **{k: refs[0][2][k] for k in ("raw", "zoomed", "index")}}, held, True, "held block 1")
    cb.result_release(1)
    for t in range(2, 9):
        ra, rb, ref = refs[t - 1]
        got = cb.result_wait(t)
        _compare({"vfo": dict(zip(rb, ref["vfo"].values())), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "block %d" % t)
        cb.result_release(t)
    ca.close()
    cb.close()


@pytest.mark.parametrize("cfg,seed", [(3, q) for q in range(1, 13)] + [(4, q) for q in range(1, 7)])
def test_random_schedules_equal_block_by_block(backend, cfg, seed):
    """A random schedule against the ordinary pass, block by block, bit for bit: block sizes from 1 sample to a whole group's room, every way a block can
    arrive (host, device and page-locked memory contiguous / not contiguous, staged, staged with a late fill), the group size and rule changed on the way, results collected
    out of step (some held across many pushes, released later), flushes, a retune and a channel-filter change between pushes."""
    from sdrplusplus_amd import capi, workloads

    rng = np.random.default_rng(1000 * cfg + seed)
    # cfg 3: WFM channels; cfg 4: NFM / AM (AGC with look-ahead) / USB (second translation) channels behind long first stages, the reference's blocks
    # (AGC look-ahead, rotator calls) a third of the usual push — launch groups must cut every one of these where block-by-block processing does
    nv = (5 if backend == "gpu" else 3) if cfg == 3 else (9 if backend == "gpu" else 6)
    base = (16000 if backend == "gpu" else 3000) if cfg == 3 else (60000 if backend == "gpu" else 15000)
    sr = workloads.CFG[cfg]["sr"]
    nsteps = 40
    sizes = [int(rng.choice([1, 7, base // 3, base, base, base + 11, 2 * base])) for _ in range(nsteps)]
    x = workloads.synth(cfg, sum(sizes), seed=40 + seed, nvfo=nv)
    cap = 4 * base + 64
    (ca, va), (cb, vb) = _ctx_pair(cfg, nv, cap, 4096, ref_block=0 if cfg == 3 else base // 3)
    dev = _device_copy_of(cb, x)
    import ctypes as C

    pin = cb.L.sdrpp_host_alloc(x.nbytes)
    assert pin
    C.memmove(pin, x.ctypes.data, x.nbytes)
    refs, held, collected, pos = {}, [], 0, 0
    for i, n in enumerate(sizes):
        op = rng.integers(0, 12)
        if op == 0:
            cb.set_pipeline_group(int(rng.integers(1, 6)), adaptive=bool(rng.integers(0, 2)))
        elif op == 1:
            cb.pipeline_flush()
        elif op == 2:
            re, im = capi.design_phase_delta(float(rng.uniform(-2e6, 2e6)), sr)
            k = int(rng.integers(0, nv))
            ca.vfo_set_phase_delta(va[k], re, im)
            cb.vfo_set_phase_delta(vb[k], re, im)
        elif op == 3:
            taps = capi.design_low_pass(float(rng.uniform(40e3, 90e3)), 20e3, 250000.0) if cfg == 3 else capi.design_low_pass(float(rng.uniform(2e3, 5e3)), 1e3, 50000.0)
            k = int(rng.integers(0, nv))
            ca.vfo_set_channel_taps(va[k], taps)
            cb.vfo_set_channel_taps(vb[k], taps)
        blk = np.ascontiguousarray(x[pos:pos + n])
        refs[i + 1] = _ordinary_results(ca, va, blk, True)
        feed = rng.integers(0, 5)
        if feed == 0:
            cb.push(blk)
        elif feed == 1:
            cb.push_device(dev + 8 * pos, n)              # contiguous with the block before when that one came from the device too
        elif feed == 2:
            cb.push_staged_from(blk)
        elif feed == 3:
            cb.push_staged_late_fill(blk, delay_s=0.0005)
        else:
            cb.push_host_ptr_async(pin + 8 * pos, n)      # the caller's page-locked memory, fetched by the launch's landing copy
        pos += n
        assert cb.ticket() == i + 1
        # collect out of step: usually the oldest outstanding block, sometimes held for a while
        while collected + len(held) < i + 1 - int(rng.integers(0, 7)):
            t = collected + len(held) + 1
            got = cb.result_wait(t, copy=False)
            ref = refs.pop(t)
            _compare({"vfo": dict(zip(vb, ref["vfo"].values())), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "cfg %d seed %d block %d" % (cfg, seed, t))
            held.append(t)
            if len(held) > int(rng.integers(0, 4)):
                for h in held:
                    cb.result_release(h)
                collected += len(held)
                held = []
    for h in held:
        cb.result_release(h)
    collected += len(held)
    for t in range(collected + 1, nsteps + 1):
        got = cb.result_wait(t)
        ref = refs.pop(t)
        _compare({"vfo": dict(zip(vb, ref["vfo"].values())), **{k: ref[k] for k in ("raw", "zoomed", "index")}}, got, True, "cfg %d seed %d block %d (tail)" % (cfg, seed, t))
        cb.result_release(t)
    st = cb.pipeline_stats()
    assert st["tick_blocks"] + st["pass_blocks"] == nsteps and st["tick_blocks"] >= nsteps // 2, st  # (blocks fall back to ordinary passes while a retune hand-over is in progress)
    cb.sync()
    cb.L.sdrpp_device_free(cb.h, dev)
    cb.L.sdrpp_host_free(pin)
    ca.close()
    cb.close()


def test_long_stream_is_identical_under_every_launch_grouping(backend):
    """A long stream three times — one block per launch, adaptive groups (their sizes follow the timing of host and device), fixed groups of GROUP_MAX —
    every VFO block and every line of every block delivered: identical bytes for every block (tools/r06_soak.py is the 60 000-block version of this)."""
    import ctypes as C
    import hashlib
    from sdrplusplus_amd import capi, workloads

    nv, B, NB, ring = (32, 50000, 1500, 64) if backend == "gpu" else (3, 2000, 150, 40)
    xs = np.concatenate([workloads.synth(3, B, seed=100 + i, nvfo=nv) for i in range(ring)])

    def run(k, adaptive):
        ctx = capi.Context(0, max_push=B * k)
        workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=nv)
        ctx.set_pipelined(True, 3)
        ctx.set_pipeline_group(k, adaptive)
        dev = _device_copy_of(ctx, xs)
        lag = min(capi.RESULT_SLOTS - 2, 8) * k
        res, digests, nxt = capi.Result(), [], 1
        for i in range(NB + lag):
            if i < NB:
                ctx.push_device(dev + 8 * B * (i % ring), B)
            upto = min(NB, i + 1 - lag) if i < NB else NB
            while nxt <= upto:
                ctx._chk(ctx.L.sdrpp_result_wait(ctx.h, C.c_uint64(nxt), C.byref(res)))
                h = hashlib.blake2b(digest_size=16)
                for q in range(res.n_vfo):
                    if res.counts[q] > 0:
                        h.update(C.string_at(C.addressof(res.samples.contents) + 8 * res.offsets[q], 8 * res.counts[q]))
                if res.n_lines > 0:
                    h.update(C.string_at(res.zoomed, 4 * res.n_lines * res.data_width))
                    h.update(C.string_at(res.index, 4 * res.n_lines * res.data_width))
                digests.append((res.n_lines, h.digest()))
                ctx._chk(ctx.L.sdrpp_result_release(ctx.h, C.c_uint64(nxt)))
                nxt += 1
        st, gs = ctx.pipeline_stats(), ctx.pipeline_group_stats()
        ctx.L.sdrpp_device_free(ctx.h, dev)
        ctx.close()
        assert len(digests) == NB and st["tick_blocks"] == NB and st["pass_blocks"] == 0, (len(digests), st)
        return digests, gs

    ref, _ = run(1, False)
    assert sum(n for n, _ in ref) > 0
    for k, adaptive in ((4, True), (capi.GROUP_MAX, False)):
        got, gs = run(k, adaptive)
        bad = [i + 1 for i, (a, b) in enumerate(zip(ref, got)) if a != b]
        assert not bad, ("blocks per launch up to %d%s: first differing blocks %s" % (k, " (adaptive)" if adaptive else "", bad[:8]))
        if not adaptive:
            assert gs["largest"] == k and gs["multi_groups"] >= NB // k - 1, gs
