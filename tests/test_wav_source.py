"""cfg 1 from real WAV bytes: the C++ sdrpp_gpu::WavSource — 44-byte header, int16 / float32 payloads, blocks of
sr / 200, loop at the end of the file — pinned against the reference's own wavreader.h + file-source worker arithmetic (oracle/_ref),
and the whole chain WAV -> IQFrontEnd -> FFT lines + WFM audio against the oracle (emulator here, device with -m gpu)."""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import support as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdrplusplus_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "emu")
REFWAV = os.path.join(ROOT, "oracle", "_ref", "libsdrpp_refwav.so")
PLANS = os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin")
SR, B = 2400000, 12000


def _wav_bytes(payload, sr, bits, channels=2, sample_type=1):
    """The canonical 44-byte header file_source expects (wavreader.h:62-76) + payload."""
    bps = bits // 8 * channels
    return (b"RIFF" + struct.pack("<I", len(payload) + 36) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, sample_type, channels, sr, sr * bps, bps, bits)
            + b"data" + struct.pack("<I", len(payload)) + payload)


def _cfg1_wav(tmp, nsamp, f32=False):
    from sdrplusplus_amd import workloads

    x = workloads.synth(1, nsamp, seed=1)
    if f32:
        payload, xs = x.view(np.float32).tobytes(), x
    else:
        i16 = workloads.to_int16_wav_samples(x)
        payload = i16.tobytes()
        xs = (i16.astype(np.float32) / np.float32(32768.0)).view(np.complex64)
    path = os.path.join(tmp, "baseband_100000000Hz.wav")
    with open(path, "wb") as f:
        f.write(_wav_bytes(payload, SR, 32 if f32 else 16, sample_type=3 if f32 else 1))
    return path, xs


def _looped(xs, nblk):
    """What the source hands over: the payload from its start, wrapping to the first sample whenever a block runs past the end."""
    reps = (nblk * B + len(xs) - 1) // len(xs) + 1
    return np.tile(xs, reps)[:nblk * B]


def _build(tmp, lib):
    exe = os.path.join(tmp, "test_wav_" + lib)
    cmd = ["g++", "-std=c++17", "-O2", "-w", "-o", exe, os.path.join(ROOT, "tests", "host_cpp", "test_wav.cpp"), "-I" + os.path.join(ROOT, "tests", "host_cpp", "standalone")]
    if lib == "emu":
        S.locked_make("-C", EMU, "-s")
        cmd += ["-L" + EMU, "-l:libsdrpp_gpu_emu.so", "-Wl,-rpath," + EMU]
    else:
        cmd += ["-L" + CSRC, "-lsdrpp_gpu", "-Wl,-rpath," + CSRC]
    subprocess.run(cmd + ["-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("f32", [False, True])
def test_wav_source_blocks_match_the_reference_reader(f32):
    """Blocks produced by WavSource (dump mode: no device) == blocks of the reference's WavReader + worker loop, over two wraps of a file
    whose length is no multiple of the block; header fields and the block size agree."""
    with tempfile.TemporaryDirectory() as tmp:
        path, xs = _cfg1_wav(tmp, int(B * 3.4), f32)
        nblk = 9
        exe = _build(tmp, "emu")
        r = subprocess.run([exe, PLANS, path, tmp, "dump", str(nblk), "1" if f32 else "0"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.fromfile(os.path.join(tmp, "blocks.f32"), np.float32).view(np.complex64)
        assert np.array_equal(got, _looped(xs, nblk))
        if not os.path.exists(REFWAV):
            pytest.skip("oracle/_ref/libsdrpp_refwav.so not built (needs the reference tree at build time)")
        L = C.CDLL(REFWAV)
        sr, bits, ch = C.c_uint(), C.c_uint(), C.c_uint()
        assert L.ref_wav_info(path.encode(), C.byref(sr), C.byref(bits), C.byref(ch)) == 1
        assert (sr.value, bits.value, ch.value) == (SR, 32 if f32 else 16, 2)
        assert L.ref_wav_block_size(path.encode()) == B
        ref = np.zeros(nblk * B * 2, np.float32)
        assert L.ref_wav_blocks(path.encode(), int(f32), nblk, ref.ctypes.data_as(C.POINTER(C.c_float))) == B
        assert np.array_equal(got.view(np.float32), ref)
        assert L.ref_wav_info(PLANS.encode(), C.byref(sr), C.byref(bits), C.byref(ch)) == 0  # no RIFF/WAVE magic


def _run_chain(exe, tmp, mode, f32, nblk, drain_ms):
    from sdrplusplus_amd import capi

    path, xs = _cfg1_wav(tmp, int(B * 6.5), f32)
    r = subprocess.run([exe, PLANS, path, tmp, mode, str(nblk), "1" if f32 else "0", str(drain_ms)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    N = 4096
    lines = np.fromfile(os.path.join(tmp, "lines.f32"), np.float32).reshape(-1, N)
    audio = np.fromfile(os.path.join(tmp, "audio.f32"), np.float32).reshape(-1, 2)
    x = _looped(xs, nblk)
    nz, skip = capi.design_reshape_params(float(SR), N, 20.0)
    spec = S.OracleSpectrum(N, nz, skip, capi.design_fft_window(2, nz))
    wfm = S.OracleChain(float(SR), 250e3, 150e3, 300e3, S.MODES["WFM"])
    ol = np.concatenate([spec.push(x[b * B:(b + 1) * B]) for b in range(nblk)])
    oa = np.concatenate([wfm.process(x[b * B:(b + 1) * B])[1] for b in range(nblk)])
    assert lines.shape == ol.shape and len(ol) >= 1 and np.array_equal(lines, ol)  # bit-exact waterfall lines
    assert audio.shape == oa.shape and np.sqrt(np.mean((audio - oa) ** 2)) < 1e-5


@pytest.mark.parametrize("mode,f32", [("stream", False), ("direct", False), ("direct", True)])
def test_cfg1_from_wav_bytes_on_the_emulator(mode, f32):
    with tempfile.TemporaryDirectory() as tmp:
        _run_chain(_build(tmp, "emu"), tmp, mode, f32, 11, 2500)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,f32", [("stream", False), ("direct", False), ("stream", True), ("direct", True)])
def test_cfg1_from_wav_bytes_on_the_device(mode, f32):
    """cfg 1 (2.4 MS/s int16 IQ WAV, 4096-point FFT at 20 lines/s, one WFM radio at +300 kHz) from the file's bytes through the C++ graph:
    ~0.2 s of signal, the file wrapping around 6 times; lines bit-exact, audio within 1e-5 RMS."""
    with tempfile.TemporaryDirectory() as tmp:
        _run_chain(_build(tmp, "product"), tmp, mode, f32, 40, 300)
