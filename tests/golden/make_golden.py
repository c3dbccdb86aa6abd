#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE'S OWN code (oracle/_ref = reference headers + iq_frontend.cpp compiled
where they lie under /root/reference, against the restated VOLK/FFTW shim).  Run in the build container only:

    make -C oracle && python tests/golden/make_golden.py

The fixtures pin the oracle (and through it the HIP path) on machines where /root/reference does not exist.  Inputs are
stored (int16 IQ, the file_source on-disk format) or regenerated from a tiny LCG so they do not depend on numpy's RNG."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import support as S  # noqa: E402


def lcg_noise(n, seed):
    """Deterministic complex noise in [-1, 1)^2 from a 64-bit LCG (Knuth MMIX constants)."""
    out = np.empty(2 * n, dtype=np.float64)
    s = np.uint64(seed)
    a, c = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        for i in range(2 * n):
            s = s * a + c
            out[i] = (int(s >> np.uint64(40)) / float(1 << 23)) - 1.0
    return out[0::2] + 1j * out[1::2]


def cfg1():
    """BASELINE cfg 1: 2.4 MS/s int16 IQ, 1 VFO WFM (+300 kHz), 4096-pt FFT; two file_source blocks of 12000."""
    sr, B, N = 2.4e6, 12000, 4096
    n = 2 * B
    t = np.arange(n) / sr
    x = 0.01 * lcg_noise(n, 1) + 0.5 * np.exp(1j * (2 * np.pi * 300e3 * t + 75.0 * np.sin(2 * np.pi * 1000.0 * t)))
    v = np.empty(2 * n)
    v[0::2], v[1::2] = x.real, x.imag
    i16 = np.clip(np.rint(v * 32767.0 * 0.5), -32768, 32767).astype(np.int16)
    xf = (i16.astype(np.float32) * np.float32(1.0 / 32768.0)).view(np.complex64)  # file_source/main.cpp:162
    r = S.ref()
    ch = S.RefChain(sr, 250e3, 150e3, 300e3, S.MODES["WFM"])
    ifs, audio = [], []
    for b in range(2):
        i, a = ch.process(xf[b * B:(b + 1) * B])
        ifs.append(i)
        audio.append(a)
    fe = r.ref_frontend_create(sr, N, 200.0, 2)  # interval 12000: nz 4096, skip 7904
    got = r.ref_frontend_feed(fe, S._fp(np.ascontiguousarray(xf).view(np.float32)), n, B, 2, 10000)
    lines = np.empty((got, N), np.float32)
    r.ref_frontend_lines(fe, S._fp(lines), got)
    r.ref_frontend_destroy(fe)
    assert got == 2
    np.savez_compressed(os.path.join(HERE, "cfg1_wfm_2blocks.npz"), iq_int16=i16, if_out=np.concatenate(ifs), audio=np.concatenate(audio), lines=lines,
                        meta=np.array([sr, B, N, 200.0, 300e3, 250e3, 150e3]))


def cfg4_modes():
    """BASELINE cfg 4 geometry: 61.44 MS/s, one reference block (307200) per mode; input regenerated from the LCG."""
    sr, B = 61.44e6, 307200
    t = np.arange(B) / sr
    noise = 1e-3 * lcg_noise(B, 4)
    out = {}
    for mode, f in (("NFM", -3.2e6), ("AM", 600e3), ("USB", sr / 8), ("LSB", -sr / 4), ("DSB", 3 * sr / 8)):
        if mode == "NFM":
            s = 0.05 * np.exp(1j * (2 * np.pi * f * t + 2.5 * np.sin(2 * np.pi * 1000 * t)))
        elif mode == "AM":
            s = 0.05 * (1 + 0.3 * np.cos(2 * np.pi * 1000 * t)) * np.exp(2j * np.pi * (f + 10.0) * t)
        else:
            s = 0.03 * (np.exp(2j * np.pi * (f + 700) * t) + np.exp(2j * np.pi * (f - 1100) * t))
        x = (noise + s).astype(np.complex64)
        from sdrplusplus_amd import radio

        if_rate, bw = radio.RADIO_DEFAULTS[mode]
        ch = S.RefChain(sr, if_rate, bw, f, S.MODES[mode])
        i, a = ch.process(x)
        out["if_" + mode] = i
        out["audio_" + mode] = a
    np.savez_compressed(os.path.join(HERE, "cfg4_modes_1block.npz"), **out)


def fft_lines():
    """65536-pt and 2^20-pt dense lines of LCG noise + tones (the 2^20 case exceeds the reference's 1e6-sample stream buffers,
    SURVEY.md §7, so it goes through handler()'s arithmetic directly: window multiply, FFT shim, VOLK power spectrum)."""
    o = S.oracle()
    res = {}
    for lg in (16, 20):
        N = 1 << lg
        x = (0.01 * lcg_noise(N, lg) + 0.3 * np.exp(2j * np.pi * 0.1003 * np.arange(N))).astype(np.complex64)
        w = S.oracle_fft_window(2, N)
        if lg == 16:
            fe = S.ref().ref_frontend_create(10e6, N, 10e6 / N, 2)
            got = S.ref().ref_frontend_feed(fe, S._fp(x.view(np.float32)), N, 50000, 1, 10000)
            line = np.empty((1, N), np.float32)
            S.ref().ref_frontend_lines(fe, S._fp(line), 1)
            S.ref().ref_frontend_destroy(fe)
            assert got == 1
            line = line[0]
        else:
            line = S.OracleSpectrum(N, N, 0, w).push(x)[0]
        # keep the fixture small: every 64th bin + the 2048 bins around the tone
        k0 = N // 2 + int(0.1003 * N) - 1024
        res["sub_%d" % lg] = line[::64].copy()
        res["peak_%d" % lg] = line[k0:k0 + 2048].copy()
    np.savez_compressed(os.path.join(HERE, "fft_lines.npz"), **res)


if __name__ == "__main__":
    assert S.ref_available(), "build oracle/_ref first (needs /root/reference)"
    cfg1()
    cfg4_modes()
    fft_lines()
    print("golden fixtures written to", HERE)
