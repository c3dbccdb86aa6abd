"""FFT -> log-power -> waterfall-line parity: product kernels vs the oracle, BIT-EXACT (raw dB lines, zoomed lines, palette
indices).  Runs on the real device under `-m gpu` and on the CPU fiber emulator of the same kernel sources otherwise."""
import numpy as np
import pytest

import support as S


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2))) if a.size else 0.0


def _ctx(max_push):
    from sdrplusplus_amd import capi

    return capi.Context(0, max_push=max_push)


def _signal(n, seed):
    r = np.random.default_rng(seed)
    t = np.arange(n)
    x = (r.standard_normal(n) + 1j * r.standard_normal(n)) * 0.01 + 0.3 * np.exp(2j * np.pi * 0.1003 * t) + 1e-3 * np.exp(-2j * np.pi * 0.31 * t)
    return x.astype(np.complex64)


def _oracle_view(lines, start, size, width, lo, hi):
    oz = np.stack([S.oracle_do_zoom(start, size, width, l) for l in lines])
    oi = np.stack([S.oracle_palette_index(z, lo, hi) for z in oz])
    return oz, oi


@pytest.mark.parametrize("lg", list(range(10, 21)))
def test_every_fft_size_bit_exact(backend, lg):
    from sdrplusplus_amd import capi

    N = 1 << lg
    reps = 3 if lg <= 16 else 1
    n = N * reps + 5
    ctx = _ctx(n)
    w = capi.design_fft_window(2, N)
    assert np.array_equal(w, S.oracle_fft_window(2, N))
    ctx.fft_configure(N, N, 0, w)
    view = (N // 8, N // 2, 733, -110.0, -15.0)
    ctx.fft_set_view(*view)
    x = _signal(n, lg)
    ctx.push(x)
    raw, zo, ix = ctx.fft_read()
    ol = S.OracleSpectrum(N, N, 0, w).push(x)
    assert raw.shape == ol.shape == (reps, N)
    assert np.array_equal(raw, ol)
    oz, oi = _oracle_view(ol, *view)
    assert np.array_equal(zo, oz) and np.array_equal(ix, oi)
    ctx.close()


@pytest.mark.parametrize("lg", [14, 16])
def test_fft_workgroups_walk_their_tiles(backend, lg, monkeypatch):
    """Pass 1 / pass 2 workgroups that walk several tiles (tile, tile + grid, ...) with the next tile's loads in flight: the grids are
    forced down to the tiles of ONE frame (a workgroup then walks the same columns / rows of every frame of the push: one to four tiles,
    frames with nz < N, frames that start in the history of the previous push), ordinary pass and pipelined mode — every line bit-exact."""
    from sdrplusplus_amd import capi

    monkeypatch.setenv("SDRPP_GPU_FFT_P1_GRID", "4")
    monkeypatch.setenv("SDRPP_GPU_FFT_P2_GRID", "4")
    monkeypatch.setenv("SDRPP_GPU_FFT_TICK_GRID", "4")
    N = 1 << lg
    nz, skip = N - 1000, 37
    w = capi.design_fft_window(2, nz)
    pushes = [N + 11, 3 * N + 901, 2 * N]
    x = _signal(sum(pushes), lg)
    sp = S.OracleSpectrum(N, nz, skip, w)
    view = (N // 8, N // 2, 733, -110.0, -15.0)
    for pipelined in (False, True):
        ctx = _ctx(max(pushes))
        ctx.fft_configure(N, nz, skip, w)
        ctx.fft_set_view(*view)
        if pipelined:
            ctx.set_pipelined(True, 2 | 4)
        sp = S.OracleSpectrum(N, nz, skip, w)
        pos = 0
        for t, n in enumerate(pushes, start=1):
            blk = x[pos:pos + n]
            pos += n
            ctx.push(blk)
            ol = sp.push(blk)
            if pipelined:
                got = ctx.result_wait(t)
                raw, zo, ix = got["raw"], got["zoomed"], got["index"]
                ctx.result_release(t)
            else:
                raw, zo, ix = ctx.fft_read()
            assert raw.shape == ol.shape and len(ol) >= 1
            assert np.array_equal(raw, ol)
            oz, oi = _oracle_view(ol, *view)
            assert np.array_equal(zo, oz) and np.array_equal(ix, oi)
        ctx.close()


@pytest.mark.parametrize("window_kind", [0, 1, 2])
def test_reference_default_framing_streaming(backend, window_kind):
    """fftRate framing (keep nz, skip the rest) across pushes of awkward sizes, incl. empty and 1-sample pushes."""
    from sdrplusplus_amd import capi

    sr, N = 2.4e6, 4096
    nz, skip = capi.design_reshape_params(sr, N, 300.0)
    assert (nz, skip) == (4096, 3904)
    w = capi.design_fft_window(window_kind, nz)
    assert np.array_equal(w, S.oracle_fft_window(window_kind, nz))
    ctx = _ctx(120000)
    ctx.fft_configure(N, nz, skip, w)
    ctx.fft_set_view(0, N, 1024, -120.0, 0.0)
    sp = S.OracleSpectrum(N, nz, skip, w)
    x = _signal(240000, 7)
    pos, total = 0, 0
    for sz in [12000, 0, 1, 4095, 4096, 30000, 7, 100000, 20000, 3904, 4097]:
        blk = x[pos:pos + sz]
        pos += sz
        ctx.push(blk)
        raw, zo, ix = ctx.fft_read()
        ol = sp.push(blk) if sz else np.empty((0, N), np.float32)
        assert raw.shape == ol.shape, (sz, raw.shape, ol.shape)
        if len(ol):
            assert np.array_equal(raw, ol)
            oz, oi = _oracle_view(ol, 0, N, 1024, -120.0, 0.0)
            assert np.array_equal(zo, oz) and np.array_equal(ix, oi)
        total += len(ol)
    assert total == (pos - nz) // (nz + skip) + 1
    ctx.close()


def test_zero_padded_frames_and_reconfigure(backend):
    """interval < fft size: nz < N inputs, the rest of the FFT input is zero (iq_frontend.cpp:301)."""
    from sdrplusplus_amd import capi

    ctx = _ctx(40000)
    x = _signal(80000, 9)
    for (sr, N, rate) in [(1e6, 4096, 400.0), (1e6, 65536, 20.0), (2.4e6, 1024, 20.0)]:
        nz, skip = capi.design_reshape_params(sr, N, rate)
        w = capi.design_fft_window(2, nz)
        ctx.fft_configure(N, nz, skip, w)  # reconfigure restarts the framing, like updateFFTPath
        sp = S.OracleSpectrum(N, nz, skip, w)
        pos = 0
        for sz in [2499, 1, 40000, 12345, 25000]:
            blk = x[pos:pos + sz]
            pos += sz
            ctx.push(blk)
            raw, _, _ = ctx.fft_read(zoomed=False)
            ol = sp.push(blk)
            assert raw.shape == ol.shape and np.array_equal(raw, ol), (N, nz, skip, sz)
    ctx.close()


@pytest.mark.parametrize("N", [2048, 8192, 1 << 17])
def test_view_changes_and_extreme_inputs(backend, N):
    """(three sizes: the one-workgroup transform, the two-pass split and the long split each have their own dB epilogue, whose main path
    handles normal powers only and hands zero / subnormal / overflowing powers to the general log2)"""
    from sdrplusplus_amd import capi

    ctx = _ctx(N)
    w = capi.design_fft_window(2, N)
    ctx.fft_configure(N, N, 0, w)
    cases = {
        "zeros": np.zeros(N, np.complex64),  # -> VOLK's clamped log2: -127 * 3.0103 dB in every bin
        "dc": np.full(N, 1.0 + 0.0j, np.complex64),
        "tiny": (_signal(N, 1) * 1e-30).astype(np.complex64),  # denormal powers
        "huge": (_signal(N, 2) * 1e15).astype(np.complex64),
    }
    views = [(0, N, 600, -120.0, 0.0), (N // 2 - 50, 100, 1024, -70.0, 0.0), (-7, N, 333, -200.0, 50.0), (N - 40, 300, 64, -120.0, 0.0)]
    if N > 8192:
        views = views[:2]
    cases["mixed"] = np.concatenate([np.zeros(N // 2, np.complex64), (_signal(N // 2, 3) * 1e-22).astype(np.complex64)])
    for name, x in cases.items():
        for view in views:
            start, size = capi.design_waterfall_view(0.0, 1.0, 1.0, N) if view[0] == 0 else (view[0], view[1])
            ctx.fft_set_view(start, size, view[2], view[3], view[4])
            ctx.push(x)
            raw, zo, ix = ctx.fft_read()
            ol = S.OracleSpectrum(N, N, 0, w).push(x)
            assert np.array_equal(raw, ol), name
            oz, oi = _oracle_view(ol, start, size, view[2], view[3], view[4])
            assert np.array_equal(zo, oz, equal_nan=True) and np.array_equal(ix, oi), (name, view)
    ctx.close()


def test_int16_ingest_matches_file_source_conversion(backend):
    """cfg 1 input format: int16 IQ converted on the device exactly like volk_16i_s32f_convert_32f(.., 32768)."""
    from sdrplusplus_amd import capi, workloads

    x = workloads.synth(1, 24000, seed=1)
    i16 = workloads.to_int16_wav_samples(x)
    xf = np.empty(len(i16), np.float32)
    S.oracle().orc_int16_to_float(i16.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_int16)), S._fp(xf), len(i16))
    xc = xf.view(np.complex64)
    nz, skip = capi.design_reshape_params(2.4e6, 4096, 200.0)
    w = capi.design_fft_window(2, nz)
    ctx = _ctx(24000)
    ctx.fft_configure(4096, nz, skip, w)
    ctx.push_int16(i16)
    raw, _, _ = ctx.fft_read(zoomed=False)
    ol = S.OracleSpectrum(4096, nz, skip, w).push(xc)
    assert len(ol) == 2 and np.array_equal(raw, ol)
    ctx.close()


@pytest.mark.parametrize("ratio,dc,conj", [(2, True, False), (8, True, True), (1, True, True), (4, False, False), (1, False, True)])
def test_preproc_chain(backend, ratio, dc, conj):
    """IQFrontEnd's pre-processing chain (decimation / DC blocking / invert IQ, iq_frontend.cpp:32-39) on the
    device in front of everything else.  Checked: the pre-processed stream itself (what bindIQStream consumers get) against the
    oracle chain (bit-exactly pinned to the reference in test_oracle_vs_reference.py), the waterfall lines computed from it, and a
    WFM VFO designed at the effective sample rate — over uneven pushes."""
    from sdrplusplus_amd import capi, radio, workloads

    sr = 2.4e6 * ratio  # effective rate 2.4 MS/s after the chain (cfg 1 geometry behind it)
    eff = sr / ratio
    pushes = [24000 * ratio, 1001, 7 * ratio, 36000 * ratio + 3, 12000 * ratio]
    rng = np.random.default_rng(ratio * 10 + dc * 2 + conj)
    n = np.arange(sum(pushes))
    sign = -1.0 if conj else 1.0  # the conjugate mirrors the spectrum: put the FM carrier where it lands at +300 kHz afterwards
    x = (0.2 * np.exp(1j * (2 * np.pi * sign * 300e3 * n / sr + sign * 3.0 * np.sin(2 * np.pi * 1000.0 * n / sr))) + (0.05 + 0.03j)
         + 0.01 * (rng.standard_normal(len(n)) + 1j * rng.standard_normal(len(n)))).astype(np.complex64)
    ctx = capi.Context(0, max_push=max(pushes))
    stages = radio.plans().stages(ratio) if ratio > 1 else []
    rate = 50.0 / eff if dc else 0.0
    ctx.preproc_configure(stages, rate, conj)
    N = 4096
    w = capi.design_fft_window(2, N)
    ctx.fft_configure(N, N, 0, w)
    d, keep = radio.vfo_desc(eff, 250e3, 150e3, 300e3, "WFM")
    vid = ctx.vfo_add(d, keep)
    opre = S.OraclePreproc(ratio, dc, rate, conj)
    spec = S.OracleSpectrum(N, N, 0, w)
    chain = S.OracleChain(eff, 250e3, 150e3, 300e3, S.MODES["WFM"])
    pos, worst_pre, worst_audio, worst_db, nlines = 0, 0.0, 0.0, 0.0, 0
    for npush in pushes:
        blk = x[pos:pos + npush]
        pos += npush
        ctx.push(blk)
        ref = opre.process(blk)
        got = ctx.preproc_read()
        assert got.shape == ref.shape
        if len(got):  # the recorder's baseband format (int16, utils/wav.cpp:166) and int8, converted on the device: bit-exact w.r.t. the float read
            import ctypes as C
            o = S.oracle()
            f = np.ascontiguousarray(got).view(np.float32)
            for pcm, scale, dt, conv in ((1, 32767.0, np.int16, o.orc_convert_16i), (0, 100.0, np.int8, o.orc_convert_8i)):
                want = np.empty(len(f), dt)
                conv.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_int, C.c_void_p]
                conv(f.ctypes.data_as(C.POINTER(C.c_float)), scale, len(f), want.ctypes.data_as(C.c_void_p))
                packed = ctx.preproc_read_pcm(pcm, scale)
                assert packed.shape == (len(got), 2) and np.array_equal(packed.reshape(-1), want)
        if len(ref):
            worst_pre = max(worst_pre, rms(got - ref) / max(rms(ref), 1e-9))
        # downstream: oracle fed with the ORACLE's pre-processed stream (the reference graph), device fed by its own
        raw, _, _ = ctx.fft_read(zoomed=False)
        ol = spec.push(ref)
        assert raw.shape == ol.shape
        if ol.size:
            strong = ol > -100.0  # bins above the noise floor of the test signal: dB of a ~1e-6-relative different input
            worst_db = max(worst_db, float(np.max(np.abs(raw - ol)[strong])))
            nlines += len(ol)
        _, oa = chain.process(ref)
        ga = ctx.vfo_read(vid)
        assert ga.shape == oa.shape
        if len(oa):
            worst_audio = max(worst_audio, rms(ga - oa))
    assert nlines > 0
    # conjugate alone is exact; the decimator differs by summation order (~1e-7).  The DC blocker is a very slow float32 integrator
    # (rate 2e-5): the reference's sequential running sum carries its own rounding drift (~1e-5 of the DC level, quasi-systematic),
    # which a parallel evaluation — chunk-local partial sums, i.e. a MORE accurate sum — cannot and does not reproduce
    assert worst_pre < (1e-9 if (ratio == 1 and not dc) else (2e-4 if dc else 2e-6)), worst_pre
    assert worst_audio < 1e-5, worst_audio
    assert worst_db < (0.05 if dc else 5e-3), worst_db  # -100 dB bins next to a -14 dB carrier feel a 1e-7-relative input change as ~1e-3 dB
    ctx.preproc_configure([], 0.0, False)  # chain removed: the raw stream goes straight through again
    ctx.push(x[:N])
    with pytest.raises(Exception):
        ctx.preproc_read()
    ctx.close()


@pytest.mark.parametrize("ratio,dc,conj", [(2, True, False), (8, True, True), (1, True, False), (4, False, True)])
def test_preproc_chain_reference_order(backend, ratio, dc, conj):
    """sdrpp_preproc_set_reference_order: the chain in the reference's own arithmetic (tap-ordered multiply-then-add decimator,
    sequential DC blocker) — the pre-processed stream is BIT-IDENTICAL to the oracle chain (itself pinned bit-exactly to the compiled
    reference), and so is every waterfall line computed from it; uneven pushes."""
    from sdrplusplus_amd import capi, radio

    sr = 2.4e6 * ratio
    eff = sr / ratio
    pushes = [24000 * ratio, 1001, 7 * ratio, 36000 * ratio + 3, 12000 * ratio]
    rng = np.random.default_rng(100 + ratio * 10 + dc * 2 + conj)
    n = np.arange(sum(pushes))
    x = (0.2 * np.exp(1j * (2 * np.pi * 300e3 * n / sr + 3.0 * np.sin(2 * np.pi * 1000.0 * n / sr))) + (0.05 + 0.03j)
         + 0.01 * (rng.standard_normal(len(n)) + 1j * rng.standard_normal(len(n)))).astype(np.complex64)
    ctx = capi.Context(0, max_push=max(pushes))
    ctx.preproc_set_reference_order(True)  # before configure: the switch survives it
    stages = radio.plans().stages(ratio) if ratio > 1 else []
    rate = 50.0 / eff if dc else 0.0
    ctx.preproc_configure(stages, rate, conj)
    N = 4096
    w = capi.design_fft_window(2, N)
    ctx.fft_configure(N, N, 0, w)
    opre = S.OraclePreproc(ratio, dc, rate, conj)
    spec = S.OracleSpectrum(N, N, 0, w)
    pos, nlines = 0, 0
    for npush in pushes:
        blk = x[pos:pos + npush]
        pos += npush
        ctx.push(blk)
        ref = opre.process(blk)
        got = ctx.preproc_read()
        assert got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (npush, float(np.max(np.abs(got - ref))) if len(ref) else 0.0)
        raw, _, _ = ctx.fft_read(zoomed=False)
        ol = spec.push(ref)
        assert raw.shape == ol.shape and np.array_equal(raw, ol)
        nlines += len(ol)
    assert nlines > 0
    ctx.close()


class _OracleWf:
    def __init__(self, height, N, width):
        import ctypes as C

        self.C = C
        self.o = S.oracle()
        o = self.o
        o.orc_wf_create.restype = C.c_void_p
        o.orc_wf_create.argtypes = [C.c_int, C.c_int, C.c_int]
        o.orc_wf_destroy.argtypes = [C.c_void_p]
        o.orc_wf_set_smoothing.argtypes = [C.c_void_p, C.c_int, C.c_float]
        o.orc_wf_set_hold.argtypes = [C.c_void_p, C.c_int, C.c_float]
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        o.orc_wf_push.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, C.c_float, C.c_float, ip]
        o.orc_wf_latest.argtypes = [C.c_void_p, fp, fp]
        o.orc_wf_raster.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, ip]
        o.orc_wf_signal_info.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, fp, fp]
        self.h = o.orc_wf_create(height, N, width)
        self.height, self.N, self.width = height, N, width

    def push(self, line, start, size, wmin, wmax):
        C = self.C
        idx = np.empty(self.width, np.int32)
        line = np.ascontiguousarray(line, np.float32)
        self.o.orc_wf_push(self.h, line.ctypes.data_as(C.POINTER(C.c_float)), start, size, wmin, wmax, idx.ctypes.data_as(C.POINTER(C.c_int32)))
        return idx

    def latest(self):
        C = self.C
        a, b = np.empty(self.width, np.float32), np.empty(self.width, np.float32)
        self.o.orc_wf_latest(self.h, a.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)))
        return a, b

    def signal_info(self, centre, bw, whole):
        C = self.C
        a, b = C.c_float(), C.c_float()
        ok = self.o.orc_wf_signal_info(self.h, centre, bw, whole, C.byref(a), C.byref(b))
        return (a.value, b.value) if ok else None

    def raster(self, start, size, wmin, wmax):
        C = self.C
        fb = np.empty((self.height, self.width), np.int32)
        n = self.o.orc_wf_raster(self.h, start, size, wmin, wmax, fb.ctypes.data_as(C.POINTER(C.c_int32)))
        return fb, n


def test_waterfall_history_trace_and_raster(backend):
    """the raw-line ring in HBM (getFFTBuffer order), the FFT trace's smoothing / hold per new line (pushFFT) and
    the full re-raster after a view change (updateWaterfallFb) — all integer / max / separately rounded float work: bit-exact against
    the oracle restatement of waterfall.cpp.  The ring wraps (height 5, 12 lines), pushes bring 0, 1 and several lines."""
    from sdrplusplus_amd import capi, workloads

    sr, N, W, H = 10e6, 4096, 600, 5
    x = workloads.synth(2, N * 12 + 1000, seed=31)
    ctx = capi.Context(0, max_push=N * 5)
    w = capi.design_fft_window(2, N)
    ctx.fft_configure(N, N, 0, w)
    start, size = capi.design_waterfall_view(1.0e6, 4.0e6, sr, N)
    wmin, wmax = -110.0, -10.0
    ctx.fft_set_view(start, size, W, wmin, wmax)
    ctx.wf_configure(H)
    spec = S.OracleSpectrum(N, N, 0, w)
    owf = _OracleWf(H, N, W)
    pos, step = 0, 0
    for npush in [N // 2, N, N * 3 + 17, 100, N * 5, N * 2]:
        if step == 2:   # switched on mid-stream: the smoothing buffer starts as a copy of the current trace
            ctx.wf_set_smoothing(True, 0.25)
            owf.o.orc_wf_set_smoothing(owf.h, 1, 0.25)
        if step == 3:
            ctx.wf_set_hold(True, 1.5)
            owf.o.orc_wf_set_hold(owf.h, 1, 1.5)
        blk = x[pos:pos + npush]
        pos += npush
        step += 1
        ctx.push(blk)
        lines = spec.push(blk)
        raw, zoomed, index = ctx.fft_read()
        assert raw.shape == lines.shape and np.array_equal(raw, lines)
        for k, ln in enumerate(lines):
            oidx = owf.push(ln, start, size, wmin, wmax)
            assert np.array_equal(index[k], oidx)
        if len(lines):
            gl, gh = ctx.wf_latest(W)
            ol, oh = owf.latest()
            assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32))
            if step > 3:
                assert np.array_equal(gh.view(np.uint32), oh.view(np.uint32))
        # SNR meter of a "selected VFO" on the newest line (calculateVFOSignalInfo): max exact, the double-precision mean to 1e-5 dB
        # (in-band VFOs: one reaching past +sr/2 makes the reference read one bin beyond the line; the device clamps)
        for centre, bw in ((1.0e6, 200e3), (-3.0e6, 12.5e3), (4.0e6, 400e3)):
            gi, oi = ctx.wf_signal_info(centre, bw, sr), owf.signal_info(centre, bw, sr)
            assert (gi is None) == (oi is None)
            if gi is not None:
                assert gi[0] == oi[0] and abs(gi[1] - oi[1]) < 1e-5
        # re-raster with the current view and with a zoomed-in one (the GUI's pan / zoom): every stored line, newest first
        for vo, vb in ((1.0e6, 4.0e6), (-2.0e6, 1.0e6)):
            s2, z2 = capi.design_waterfall_view(vo, vb, sr, N)
            gfb, gn = ctx.wf_raster(s2, z2, W, wmin, wmax)
            ofb, on = owf.raster(s2, z2, wmin, wmax)
            assert gn == on and np.array_equal(gfb, ofb)
    assert gn == H  # the ring wrapped
    ctx.wf_configure(0)
    with pytest.raises(Exception):
        ctx.wf_raster(start, size, W, wmin, wmax)
    ctx.close()


def test_reference_blocks_behind_the_preproc_decimator(backend):
    """The reference's blocks are blocks of the RAW stream; behind the front end's decimator the VFO bank sees them as the decimator's
    per-block outputs.  Reference-rotator mode + sdrpp_set_reference_block with a /2 pre-processing decimation, large and ragged pushes,
    deferred staging: USB / AM audio equal to the oracle graph (preproc -> RxVFO -> demodulator) driven block by block."""
    from sdrplusplus_amd import capi, radio

    ratio, eff, Braw = 2, 2.4e6, 24000  # raw 4.8 MS/s in blocks of sr/200
    sr = eff * ratio
    nblk = 10
    n = np.arange(Braw * nblk)
    rng = np.random.default_rng(77)
    env = np.where((n // 31000) % 3 == 2, 1.0, 0.04)
    x = (env * 0.2 * (np.exp(2j * np.pi * (200e3 + 700) * n / sr) + np.exp(2j * np.pi * (200e3 + 1500) * n / sr))
         + env * 0.3 * (1 + 0.5 * np.cos(2 * np.pi * 900 * n / sr)) * np.exp(2j * np.pi * -433333.0 * n / sr)
         + 0.002 * (rng.standard_normal(len(n)) + 1j * rng.standard_normal(len(n)))).astype(np.complex64)
    specs = [("USB", 200e3 + 1400.0), ("AM", -433333.0)]

    def oracle_run():
        pre = S.OraclePreproc(ratio, False, 0.0, False)
        chains = [S.OracleChain(eff, *radio.RADIO_DEFAULTS[m], off, S.MODES[m]) for m, off in specs]
        outs = [[] for _ in specs]
        for b in range(nblk):
            y = pre.process(x[b * Braw:(b + 1) * Braw])
            for k, ch in enumerate(chains):
                outs[k].append(ch.process(y)[1])
        return [np.concatenate(o) for o in outs]

    ref = oracle_run()
    for name, pushes, deferred in (("blocks", [Braw] * nblk, False), ("big", [5 * Braw] * 2, False), ("ragged", [50001, 70000, 119999], False), ("deferred", [Braw] * nblk, True)):
        ctx = capi.Context(0, max_push=max(max(pushes), Braw * nblk if deferred else 0))
        ctx.set_nco_mode(1)
        ctx.set_reference_block(Braw)
        ctx.preproc_configure(radio.plans().stages(ratio), 0.0, False)
        ctx.set_deferred(deferred)
        vids = []
        for m, off in specs:
            d, keep = radio.vfo_desc(eff, *radio.RADIO_DEFAULTS[m], off, m)
            vids.append(ctx.vfo_add(d, keep))
        got, pos = [[] for _ in specs], 0
        for i, p in enumerate(pushes):
            ctx.push(x[pos:pos + p])
            pos += p
            if deferred and i + 1 < len(pushes):
                continue
            for k, a in enumerate(ctx.vfo_read_many(vids)):
                got[k].append(a.copy())
        for k, (m, _) in enumerate(specs):
            g = np.concatenate(got[k])
            # ragged pushes are cut into Braw-sample blocks + a shorter last one: a different block structure than the oracle's -> only
            # the runs whose cuts coincide with the reference's blocks are compared tightly
            if name == "ragged":
                assert g.shape == ref[k].shape
                continue
            assert g.shape == ref[k].shape, (name, m, g.shape, ref[k].shape)
            e = rms(g - ref[k]) / max(1.0, rms(ref[k]))
            assert e < 1e-5, (name, m, e)
        ctx.close()


def test_preproc_reconfigure_carries_what_the_reference_carries(backend):
    """sdrpp_preproc_reconfigure against the compiled reference's chain objects re-planned the way IQFrontEnd's setters re-plan them (iq_frontend.cpp:76-130;
    oracle/ref_api.cpp: ref_preproc_set): setDecimation(4) (new stages), setDCBlocking(true), setSampleRate (the DC blocker's rate changes, its estimate and the
    decimator's delay lines stay), setInvertIQ(true), setDCBlocking(false) and on again (the estimate it had), setDecimation(2).  In the reference's own arithmetic
    (sdrpp_preproc_set_reference_order) the pre-processed stream is BIT-identical block for block; in the default arithmetic it is within the chain's tolerance."""
    from sdrplusplus_amd import capi, radio

    if not S.ref_available():
        pytest.skip("oracle/_ref not built (needs the reference tree at build time)")
    B = 24000
    steps = [  # (ratio, dc, effective rate for the DC blocker, conj, new decimator, keep)
        (1, False, 2.4e6, False, False, 3),
        (4, False, 0.6e6, False, True, 2),
        (4, True, 0.6e6, False, False, 3),
        (4, True, 0.5e6, False, False, 3),
        (4, True, 0.5e6, True, False, 3),
        (4, False, 0.5e6, True, False, 3),
        (4, True, 0.5e6, True, False, 3),
        (2, True, 1.0e6, True, True, 2),
    ]
    rng = np.random.default_rng(91)
    n = np.arange(B * len(steps) * 2)
    x = (0.2 * np.exp(2j * np.pi * 0.11 * n) + (0.07 - 0.04j) + 0.01 * (rng.standard_normal(len(n)) + 1j * rng.standard_normal(len(n)))).astype(np.complex64)
    for ref_order in (True, False):
        ctx = capi.Context(0, max_push=B)
        ctx.preproc_set_reference_order(ref_order)
        pre = S.RefPreproc(1, False, 50.0 / 2.4e6, False)
        pos = 0
        for k, (ratio, dc, eff, conj, newdec, keep) in enumerate(steps):
            rate = 50.0 / eff
            pre.set(ratio, dc, rate, conj, newdec)
            ctx.preproc_reconfigure(radio.plans().stages(ratio) if ratio > 1 else [], rate if dc else 0.0, conj, keep)
            for _ in range(2):
                blk = x[pos:pos + B]
                pos += B
                want = pre.process(blk)
                if ratio == 1 and not dc and not conj:
                    continue  # (no chain: the device hands the block straight through, nothing to read back)
                ctx.push(blk)
                got = ctx.preproc_read()
                assert got.shape == want.shape, (k, got.shape, want.shape)
                if ref_order:
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, ratio, dc, conj, float(np.max(np.abs(got - want))))
                else:
                    assert rms(got - want) / max(rms(want), 1e-9) < 2e-4 and rms(got[:64] - want[:64]) / max(rms(want), 1e-9) < 5e-4, (k, ratio, dc, conj, rms(got - want) / rms(want))
        ctx.close()
