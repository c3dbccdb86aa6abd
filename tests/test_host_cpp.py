"""C++ host mirror (sdrplusplus_amd/host/sdrpp_gpu_blocks.h): compiles stand-alone and against the reference's real
dsp::block / dsp::stream headers; on a GPU, a threaded source -> IQFrontEnd -> sinks graph matches the oracle."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import support as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdrplusplus_amd", "csrc")


def _build(tmp):
    exe = os.path.join(tmp, "test_blocks")
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "host_cpp", "test_blocks.cpp"), "-L" + CSRC, "-lsdrpp_gpu",
                    "-Wl,-rpath," + CSRC, "-lpthread"], check=True)
    return exe


def test_host_mirror_compiles_standalone():
    with tempfile.TemporaryDirectory() as tmp:
        assert os.path.exists(_build(tmp))


def test_device_math_helpers():
    """fm_phase (the discriminator's polynomial atan2) against double-precision atan2 over 2.5 M points, normalize_phase's range:
    the kernel header compiled for the host against the emulator's headers (tests/host_cpp/test_device_math.cpp)."""
    emu = os.path.join(ROOT, "tests", "emu")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "test_device_math")
        subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-I" + emu, "-I" + os.path.join(emu, "gfx950"), "-I" + CSRC, "-o", exe,
                        os.path.join(ROOT, "tests", "host_cpp", "test_device_math.cpp"), "-lm"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/core/src/dsp"), reason="needs the reference tree")
def test_host_mirror_compiles_inside_sdrpp_tree():
    """-DSDRPP_GPU_USE_SDRPP_DSP: the blocks derive from the reference's own dsp::block and use its dsp::stream<T> and plans.h."""
    src = '#define SDRPP_GPU_USE_SDRPP_DSP\n#include "sdrplusplus_amd/host/sdrpp_gpu_blocks.h"\nint main() { sdrpp_gpu::DecimPlans p; sdrpp_gpu::IQFrontEnd fe; return p.maxRatio == 8192 ? 0 : 1; }\n'
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "t.cpp")
        open(f, "w").write(src)
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I" + ROOT, "-I" + os.path.join(ROOT, "oracle", "shim"), "-I/root/reference/core/src", f], check=True)


@pytest.mark.gpu
def test_threaded_graph_matches_oracle():
    from sdrplusplus_amd import capi, workloads

    sr, B, N, rate = 2.4e6, 12000, 4096, 100.0
    nblk = 12
    x = workloads.synth(1, B * nblk, seed=5)
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp)
        x.view(np.float32).tofile(os.path.join(tmp, "iq.f32"))
        r = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin"), os.path.join(tmp, "iq.f32"), str(sr), str(B), str(N), str(rate), tmp],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = np.fromfile(os.path.join(tmp, "lines.f32"), np.float32).reshape(-1, N)
        ifs = np.fromfile(os.path.join(tmp, "if.f32"), np.float32).view(np.complex64)
        audio = np.fromfile(os.path.join(tmp, "audio.f32"), np.float32).reshape(-1, 2)
        af = np.fromfile(os.path.join(tmp, "af.f32"), np.float32).reshape(-1, 2)
    nz, skip = capi.design_reshape_params(sr, N, rate)
    spec = S.OracleSpectrum(N, nz, skip, capi.design_fft_window(2, nz))
    raw = S.OracleChain(sr, 250e3, 150e3, sr / 8, None)
    wfm = S.OracleChain(sr, 250e3, 150e3, 300e3, S.MODES["WFM"])
    from test_parity_vfo import _OracleAf
    oaf, of = _OracleAf(250e3, 48000.0, 50e-6, False), []
    ol, oi, oa = [], [], []
    for b in range(nblk):
        blk = x[b * B:(b + 1) * B]
        if b == 3:
            raw.set_offset(-sr / 4)  # the C++ test retunes after feeding 3 blocks
        ol.append(spec.push(blk))
        oi.append(raw.process(blk)[0])
        oa.append(wfm.process(blk)[1])
        of.append(oaf.process(oa[-1]))
    ol, oi, oa, of = np.concatenate(ol), np.concatenate(oi), np.concatenate(oa), np.concatenate(of)
    assert lines.shape == ol.shape and np.array_equal(lines, ol)
    assert audio.shape == oa.shape and np.sqrt(np.mean((audio - oa) ** 2)) < 1e-5
    # RxVFO::attachAF: resampler to 48 kHz + 50 us de-emphasis behind the demodulator, delivered on the same `audio` stream
    assert af.shape == of.shape and np.sqrt(np.mean((af - of) ** 2)) < 1e-5
    # setOffset() is called by the source thread right after it handed over the third block, i.e. asynchronously to the worker
    # (exactly like a GUI retune in SDR++): it takes effect from block 2 or 3.  Blocks 0-1 are therefore compared tightly.
    assert ifs.shape == oi.shape
    n2 = 2 * 1250 - 10
    assert np.sqrt(np.mean(np.abs(ifs[:n2] - oi[:n2]) ** 2)) / np.sqrt(np.mean(np.abs(oi[:n2]) ** 2)) < 5e-6
