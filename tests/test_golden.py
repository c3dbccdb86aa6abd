"""Golden fixtures generated from the reference's own code (tests/golden/make_golden.py, run where /root/reference
exists): they pin the oracle bit-exactly everywhere, and the product (GPU under -m gpu, emulator otherwise) within the
stated tolerances — waterfall lines bit-exact, audio 1e-5 RMS."""
import os
import sys

import numpy as np
import pytest

import support as S

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import lcg_noise  # noqa: E402  (input regeneration only)


def _load(name):
    return np.load(os.path.join(HERE, "golden", name))


def rms(a):
    return float(np.sqrt(np.mean(np.abs(a) ** 2)))


def _cfg1_input(g):
    i16 = g["iq_int16"]
    return i16, (i16.astype(np.float32) * np.float32(1.0 / 32768.0)).view(np.complex64)


def test_oracle_reproduces_cfg1_golden():
    g = _load("cfg1_wfm_2blocks.npz")
    sr, B, N, rate, off, if_rate, bw = g["meta"]
    _, x = _cfg1_input(g)
    B, N = int(B), int(N)
    ch = S.OracleChain(sr, if_rate, bw, off, S.MODES["WFM"])
    outs = [ch.process(x[b * B:(b + 1) * B]) for b in range(2)]
    assert np.array_equal(np.concatenate([o[0] for o in outs]), g["if_out"])
    assert np.array_equal(np.concatenate([o[1] for o in outs]), g["audio"])
    import ctypes as C

    skip, nz = C.c_int(), C.c_int()
    S.oracle().orc_gen_reshape_params(sr, N, rate, C.byref(skip), C.byref(nz))
    sp = S.OracleSpectrum(N, nz.value, skip.value, S.oracle_fft_window(2, nz.value))
    lines = np.concatenate([sp.push(x[b * B:(b + 1) * B]) for b in range(2)])
    assert np.array_equal(lines, g["lines"])


def _cfg4_signal(mode, f, sr=61.44e6, B=307200):
    t = np.arange(B) / sr
    noise = 1e-3 * lcg_noise(B, 4)
    if mode == "NFM":
        s = 0.05 * np.exp(1j * (2 * np.pi * f * t + 2.5 * np.sin(2 * np.pi * 1000 * t)))
    elif mode == "AM":
        s = 0.05 * (1 + 0.3 * np.cos(2 * np.pi * 1000 * t)) * np.exp(2j * np.pi * (f + 10.0) * t)
    else:
        s = 0.03 * (np.exp(2j * np.pi * (f + 700) * t) + np.exp(2j * np.pi * (f - 1100) * t))
    return (noise + s).astype(np.complex64)


CFG4 = (("NFM", -3.2e6), ("AM", 600e3), ("USB", 61.44e6 / 8), ("LSB", -61.44e6 / 4), ("DSB", 3 * 61.44e6 / 8))


def test_oracle_reproduces_cfg4_golden():
    from sdrplusplus_amd import radio

    g = _load("cfg4_modes_1block.npz")
    for mode, f in CFG4:
        if_rate, bw = radio.RADIO_DEFAULTS[mode]
        i, a = S.OracleChain(61.44e6, if_rate, bw, f, S.MODES[mode]).process(_cfg4_signal(mode, f))
        assert np.array_equal(i, g["if_" + mode]) and np.array_equal(a, g["audio_" + mode]), mode


def _fft_case(lg):
    N = 1 << lg
    x = (0.01 * lcg_noise(N, lg) + 0.3 * np.exp(2j * np.pi * 0.1003 * np.arange(N))).astype(np.complex64)
    k0 = N // 2 + int(0.1003 * N) - 1024
    return N, x, k0


@pytest.mark.parametrize("lg", [16, 20])
def test_oracle_reproduces_fft_golden(lg):
    g = _load("fft_lines.npz")
    N, x, k0 = _fft_case(lg)
    line = S.OracleSpectrum(N, N, 0, S.oracle_fft_window(2, N)).push(x)[0]
    assert np.array_equal(line[::64], g["sub_%d" % lg]) and np.array_equal(line[k0:k0 + 2048], g["peak_%d" % lg])


def test_product_vs_cfg1_golden(backend):
    """cfg 1 end to end through the C-ABI: int16 ingest (file_source format) -> FFT lines + WFM audio."""
    from sdrplusplus_amd import capi, radio

    g = _load("cfg1_wfm_2blocks.npz")
    sr, B, N, rate, off, if_rate, bw = g["meta"]
    B, N = int(B), int(N)
    i16, _ = _cfg1_input(g)
    ctx = capi.Context(0, max_push=B)
    nz, skip = capi.design_reshape_params(sr, N, rate)
    ctx.fft_configure(N, nz, skip, capi.design_fft_window(2, nz))
    d, keep = radio.vfo_desc(sr, if_rate, bw, off, "WFM")
    vid = ctx.vfo_add(d, keep)
    lines, audio, ifs = [], [], []
    for b in range(2):
        ctx.push_int16(i16[2 * b * B:2 * (b + 1) * B])
        lines.append(ctx.fft_read(zoomed=False)[0])
        audio.append(ctx.vfo_read(vid))
        ifs.append(ctx.vfo_read_if(vid))
    lines, audio, ifs = np.concatenate(lines), np.concatenate(audio), np.concatenate(ifs)
    assert np.array_equal(lines, g["lines"])                      # bit-exact waterfall lines
    assert audio.shape == g["audio"].shape and rms(audio - g["audio"]) < 1e-5
    assert ifs.shape == g["if_out"].shape and rms(ifs - g["if_out"]) / rms(g["if_out"]) < 1e-4
    ctx.close()


def test_product_vs_cfg4_golden(backend):
    from sdrplusplus_amd import capi, radio

    g = _load("cfg4_modes_1block.npz")
    for mode, f in CFG4:
        if_rate, bw = radio.RADIO_DEFAULTS[mode]
        ctx = capi.Context(0, max_push=307200)
        d, keep = radio.vfo_desc(61.44e6, if_rate, bw, f, mode)
        vid = ctx.vfo_add(d, keep)
        ctx.push(_cfg4_signal(mode, f))
        a = ctx.vfo_read(vid)
        ref = g["audio_" + mode]
        assert a.shape == ref.shape and rms(a - ref) <= 1e-5 * max(1.0, rms(ref)), (mode, rms(a - ref), rms(ref))
        ctx.close()


@pytest.mark.parametrize("lg", [16, 20])
def test_product_vs_fft_golden(backend, lg):
    from sdrplusplus_amd import capi

    g = _load("fft_lines.npz")
    N, x, k0 = _fft_case(lg)
    ctx = capi.Context(0, max_push=N)
    ctx.fft_configure(N, N, 0, capi.design_fft_window(2, N))
    ctx.push(x)
    line = ctx.fft_read(zoomed=False)[0][0]
    assert np.array_equal(line[::64], g["sub_%d" % lg]) and np.array_equal(line[k0:k0 + 2048], g["peak_%d" % lg])
    ctx.close()
