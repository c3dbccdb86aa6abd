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


EMU = os.path.join(ROOT, "tests", "emu")
REF = "/root/reference"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "test_blocks_ref")


def _build(tmp, lib="product", headers="standalone", source="test_blocks.cpp"):
    """tests/host_cpp/test_blocks.cpp against the test double of dsp::block / dsp::stream or against the reference's REAL headers
    (+ the radio module's demod::Demodulator interface, extracted by oracle/Makefile), linked with the product library or — for runs
    on a machine without a GPU — with the CPU emulator build of the same sources (tests/emu)."""
    exe = os.path.join(tmp, "%s_%s_%s" % (source.split(".")[0], lib, headers))
    cmd = ["g++", "-std=c++17", "-O2", "-w", "-o", exe, os.path.join(ROOT, "tests", "host_cpp", source)]
    if headers == "reference":
        cmd += ["-DSDRPP_GPU_TEST_DEMOD_IFACE", "-I" + os.path.join(ROOT, "oracle", "shim"), "-I" + REF + "/core/src", "-I" + os.path.join(ROOT, "oracle", "_ref")]
    else:
        cmd += ["-I" + os.path.join(ROOT, "tests", "host_cpp", "standalone")]
    if lib == "emu":
        S.locked_make("-C", EMU, "-s")
        cmd += ["-L" + EMU, "-l:libsdrpp_gpu_emu.so", "-Wl,-rpath," + EMU]
    else:
        cmd += ["-L" + CSRC, "-lsdrpp_gpu", "-Wl,-rpath," + CSRC]
    subprocess.run(cmd + ["-lpthread"], check=True)
    return exe


def _run_graph_and_check(exe, mode, tmp, drain_ms=None):
    """Source thread -> IQFrontEnd -> sink threads (the way SDR++ drives it), 12 blocks of cfg 1, outputs against the oracle."""
    from sdrplusplus_amd import capi, workloads

    sr, B, N, rate = 2.4e6, 12000, 4096, 100.0
    nblk = 12
    x = workloads.synth(1, B * nblk, seed=5)
    x.view(np.float32).tofile(os.path.join(tmp, "iq.f32"))
    r = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin"), os.path.join(tmp, "iq.f32"), str(sr), str(B), str(N), str(rate), tmp, mode] +
                       ([str(drain_ms)] if drain_ms else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = np.fromfile(os.path.join(tmp, "lines.f32"), np.float32).reshape(-1, N)
    ifs = np.fromfile(os.path.join(tmp, "if.f32"), np.float32).view(np.complex64)
    audio = np.fromfile(os.path.join(tmp, "audio.f32"), np.float32).reshape(-1, 2)
    af = np.fromfile(os.path.join(tmp, "af.f32"), np.float32).reshape(-1, 2)
    tap = np.fromfile(os.path.join(tmp, "iq_tap.f32"), np.float32).view(np.complex64)
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
    # (setPipelining: the AF chain and the IQ tap are part of the pipelined graph too — same checks in every mode)
    # RxVFO::attachAF: resampler to 48 kHz + 50 us de-emphasis behind the demodulator, delivered on the same `audio` stream
    assert af.shape == of.shape and np.sqrt(np.mean((af - of) ** 2)) < 1e-5
    # bindIQStream: every block of the (here un-pre-processed) wideband IQ, bit for bit, across the setInput() change of source
    assert tap.shape == x.shape and np.array_equal(tap, x)
    # setOffset() is called by the source thread right after it handed over the third block, i.e. asynchronously to the worker
    # (exactly like a GUI retune in SDR++): it takes effect from block 2 or 3 on (later still when the frame buffer queues blocks).
    # Blocks 0-1 are therefore compared tightly.
    assert ifs.shape == oi.shape
    if mode == "buffered":
        return  # the source runs up to 32 blocks ahead of the worker: where the asynchronous retune lands is not defined
    n2 = 2 * 1250 - 10
    assert np.sqrt(np.mean(np.abs(ifs[:n2] - oi[:n2]) ** 2)) / np.sqrt(np.mean(np.abs(oi[:n2]) ** 2)) < 5e-6


@pytest.mark.parametrize("mode", ["bypass", "buffered", "pipelined", "pipelined_groups"])
def test_host_mirror_threaded_graph_on_the_emulator(mode):
    """The C++ mirror built against the test double of dsp::block / dsp::stream, linked with the CPU emulator build of the library:
    source thread, front-end worker (+ frame-buffer worker when buffering is on), sink threads; setInput, bindIQStream,
    flushInputBuffer, retune while running.  A logic check of the host code; the device leg is test_threaded_graph_matches_oracle."""
    # (pipelined: the source is decoupled from the emulated launch by one more block and the 20-VFO bank takes seconds per block there,
    # so "everything handed over has been consumed" needs a longer wait before the change of source; the figures are ~5 x what the emulator needs since its fibers switch in user space)
    with tempfile.TemporaryDirectory() as tmp:
        _run_graph_and_check(_build(tmp, lib="emu"), mode, tmp, drain_ms=6000 if mode.startswith("pipelined") else 3000)


@pytest.mark.skipif(not os.path.isdir(REF + "/core/src/dsp"), reason="needs the reference tree")
def test_host_mirror_links_and_runs_against_the_reference_headers():
    """The same source compiled against SDR++'s REAL core/src/dsp/block.h, stream.h, types.h, multirate/decim/plans.h and the radio
    module's demod::Demodulator interface (decoder_modules/radio/src/demod.h, cut out at build time), LINKED and RUN (CPU emulator
    library): IQFrontEnd derives from the real dsp::block, speaks the real dsp::stream<T>, FusedDemodulator overrides every pure
    virtual of the real interface."""
    S.locked_make("-C", os.path.join(ROOT, "oracle"), "-s", "_ref/demod_iface.h")
    with tempfile.TemporaryDirectory() as tmp:
        _run_graph_and_check(_build(tmp, lib="emu", headers="reference"), "bypass", tmp, drain_ms=3000)


def _run_reconfig_and_check(exe, tmp, nextra, wait_ms):
    """tests/host_cpp/test_reconfig.cpp: setters called between blocks of a RUNNING pipelined graph; the same schedule replayed on the oracle."""
    from sdrplusplus_amd import capi, workloads

    sr, B, nblk = 2.4e6, 12000, 13
    x = workloads.synth(1, B * nblk, seed=9)
    x.view(np.float32).tofile(os.path.join(tmp, "iq.f32"))
    r = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin"), os.path.join(tmp, "iq.f32"), str(sr), str(B), tmp, str(nextra), str(wait_ms)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    ld = lambda name, dt: np.fromfile(os.path.join(tmp, name), dt)
    radio, late, victim = (ld(n + ".f32", np.float32).reshape(-1, 2) for n in ("radio", "late", "victim"))
    rc, lc, vc = (ld(n + "_counts.i32", np.int32) for n in ("radio", "late", "victim"))
    sizes = ld("line_sizes.i32", np.int32)
    lines = ld("lines.f32", np.float32)
    # the oracle under the same schedule
    o_radio = S.OracleChain(sr, 250e3, 150e3, 300e3, S.MODES["WFM"])
    o_victim = S.OracleChain(sr, 250e3, 150e3, -200e3, S.MODES["WFM"])
    o_late = S.OracleChain(sr, 250e3, 150e3, 500e3, S.MODES["WFM"])
    specs = {}
    for N in (4096, 2048):
        nz, skip = capi.design_reshape_params(sr, N, 100.0)
        specs[N] = S.OracleSpectrum(N, nz, skip, capi.design_fft_window(2, nz))
    e_radio, e_late, e_victim, e_l4096, e_l2048 = [], [], [], [], []
    for b in range(nblk):
        blk = x[b * B:(b + 1) * B]
        if b == 5:
            o_radio.set_bandwidth(120e3)  # RxVFO::setBandwidth after block 4
        e_radio.append(o_radio.process(blk)[1])
        if b >= 3:
            e_late.append(o_late.process(blk)[1])  # added after block 2: starts from an all-zero history with block 3
        if b <= 5:
            e_victim.append(o_victim.process(blk)[1])  # removed after block 5
        (e_l4096 if b <= 6 else e_l2048).append(specs[4096 if b <= 6 else 2048].push(blk))  # setFFTSize(2048) after block 6: the framing restarts
    tol = lambda ref: 1e-5 * max(1.0, float(np.sqrt(np.mean(ref ** 2))))
    # "radio" and "late": EVERY block exactly once, in order — across addVFO / removeVFO / setBandwidth / setFFTSize / setPipelining off + on / stop + start
    for name, got, cnt, exp in (("radio", radio, rc, e_radio), ("late", late, lc, e_late)):
        assert [int(c) for c in cnt] == [len(e) for e in exp], (name, cnt.tolist(), [len(e) for e in exp])
        ref = np.concatenate(exp)
        assert got.shape == ref.shape and np.sqrt(np.mean((got - ref) ** 2)) < tol(ref), (name, float(np.sqrt(np.mean((got - ref) ** 2))))
    # "victim": what it delivered before it was removed is a prefix of its stream, whole blocks, nothing twice
    nv = len(vc)
    assert nv <= len(e_victim) and [int(c) for c in vc] == [len(e) for e in e_victim[:nv]], (vc.tolist(), [len(e) for e in e_victim])
    if nv:
        ref = np.concatenate(e_victim[:nv])
        assert victim.shape == ref.shape and np.sqrt(np.mean((victim - ref) ** 2)) < tol(ref)
    # lines: 4096-point lines in order without a gap up to the change of size (those still in flight at the change are dropped with the display's
    # old buffers), then EVERY 2048-point line of the restarted framing; bit-exact
    n_old = int(np.sum(sizes == 4096))
    assert np.all(sizes[:n_old] == 4096) and np.all(sizes[n_old:] == 2048), sizes.tolist()
    o4, o2 = np.concatenate(e_l4096), np.concatenate(e_l2048)
    assert n_old <= len(o4) and len(sizes) - n_old == len(o2), (n_old, len(o4), len(sizes) - n_old, len(o2))
    g4 = lines[:n_old * 4096].reshape(-1, 4096)
    g2 = lines[n_old * 4096:].reshape(-1, 2048)
    assert np.array_equal(g4, o4[:n_old]) and np.array_equal(g2, o2)
    return r.stdout


def test_reconfigure_while_running_on_the_emulator():
    """addVFO / setBandwidth / removeVFO (blocks in flight) / setFFTSize / setPipelining(false / true) / stop + start between the blocks of a RUNNING
    pipelined graph (iq_frontend.cpp:105-183, dsp/block.h:46-94): every delivered block equals the oracle under the same schedule, none lost, none
    twice.  CPU emulator build; three VFOs (a small bank: the vector-unit front ends as roles of the tick since round 5 — ordinary passes before; the
    device leg adds 17 radios for the matrix front end)."""
    with tempfile.TemporaryDirectory() as tmp:
        out = _run_reconfig_and_check(_build(tmp, lib="emu", source="test_reconfig.cpp"), tmp, 0, 60000)
        assert "blocks 13" in out


@pytest.mark.parametrize("san", ["address", "thread"])
def test_reconfigure2_while_running_host_side_under_sanitizers(san):
    """tests/host_cpp/test_reconfig2.cpp (the rest of the control surface) with the HOST side compiled with -fsanitize=address / thread, emulator library: no
    memory error, no data race (the one filtered report: see the test below).  The outputs are checked by the un-instrumented legs; here the run must be clean
    and deliver every block."""
    S.locked_make("-C", EMU, "-s")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "test_reconfig2_" + san)
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-w", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-o", exe, os.path.join(ROOT, "tests", "host_cpp", "test_reconfig2.cpp"),
                            "-I" + os.path.join(ROOT, "tests", "host_cpp", "standalone"), "-L" + EMU, "-l:libsdrpp_gpu_emu.so", "-Wl,-rpath," + EMU, "-lpthread"], capture_output=True, text=True)
        if r.returncode != 0 and ("cannot find" in r.stderr or "sanitizer" in r.stderr.lower()):
            pytest.skip("no %s sanitizer runtime in this toolchain" % san)
        assert r.returncode == 0, r.stderr[-2000:]
        sr, B, nblk = 2.4e6, 12000, 14
        rg = np.random.default_rng(19)
        x = (0.3 * np.exp(2j * np.pi * 300e3 * np.arange(B * nblk) / sr) + 0.01 * (rg.standard_normal(B * nblk) + 1j * rg.standard_normal(B * nblk))).astype(np.complex64)
        x.view(np.float32).tofile(os.path.join(tmp, "iq.f32"))
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=0")
        r = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin"), os.path.join(tmp, "iq.f32"), str(sr), str(B), tmp, "0", "120000"],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "blocks 14" in r.stdout and "in 14 blocks, steady" in r.stdout and r.stdout.rstrip().endswith("in 14"), r.stdout
        assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        reports = [ln for ln in r.stderr.splitlines() if ln.startswith("WARNING: ThreadSanitizer")]
        assert all("double lock of a mutex" in ln for ln in reports), "\n".join(reports) + r.stderr[-3000:]


@pytest.mark.parametrize("san", ["address", "thread"])
def test_reconfigure_while_running_host_side_under_sanitizers(san):
    """The same program with the HOST side (the C++ mirror, the test double of dsp::stream / dsp::block, the test) compiled with
    -fsanitize=address / -fsanitize=thread and linked with the emulator library: no memory error, no data race.  (ThreadSanitizer of GCC 11 does
    not intercept pthread_cond_clockwait, which std::condition_variable::wait_for calls: it then believes the mutex is still held when the wait
    returns and reports a "double lock" at the helpers' timed wait — the one report that is filtered out.)"""
    from sdrplusplus_amd import workloads

    S.locked_make("-C", EMU, "-s")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "test_reconfig_" + san)
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-w", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-o", exe, os.path.join(ROOT, "tests", "host_cpp", "test_reconfig.cpp"),
                            "-I" + os.path.join(ROOT, "tests", "host_cpp", "standalone"), "-L" + EMU, "-l:libsdrpp_gpu_emu.so", "-Wl,-rpath," + EMU, "-lpthread"], capture_output=True, text=True)
        if r.returncode != 0 and ("cannot find" in r.stderr or "sanitizer" in r.stderr.lower()):
            pytest.skip("no %s sanitizer runtime in this toolchain" % san)
        assert r.returncode == 0, r.stderr[-2000:]
        sr, B, nblk = 2.4e6, 12000, 13
        workloads.synth(1, B * nblk, seed=9).view(np.float32).tofile(os.path.join(tmp, "iq.f32"))
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=0")
        r = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin"), os.path.join(tmp, "iq.f32"), str(sr), str(B), tmp, "0", "120000"],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "radio 16250 in 13 blocks, late 12500 in 10" in r.stdout, r.stdout
        assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        reports = [ln for ln in r.stderr.splitlines() if ln.startswith("WARNING: ThreadSanitizer")]
        assert all("double lock of a mutex" in ln for ln in reports), "\n".join(reports) + r.stderr[-3000:]


@pytest.mark.gpu
def test_reconfigure_while_running_on_the_device():
    """The same on the device with 18 more radios: the blocks run as ticks (matrix front end), results 7 blocks behind their pushes."""
    with tempfile.TemporaryDirectory() as tmp:
        out = _run_reconfig_and_check(_build(tmp, source="test_reconfig.cpp"), tmp, 18, 20000)
        assert "blocks 13" in out


def _run_reconfig2_and_check(exe, tmp, nextra, wait_ms):
    """tests/host_cpp/test_reconfig2.cpp: the rest of IQFrontEnd's / RxVFO's setters between the blocks of a RUNNING pipelined graph; the same schedule replayed on
    the compiled reference's own RxVFO / demodulator / pre-processing objects (their state across these calls is the specification) and the oracle's spectrum."""
    from sdrplusplus_amd import capi, radio

    sr0, B, nblk, N = 2.4e6, 12000, 14, 4096
    r = np.random.default_rng(19)
    t = np.arange(B * nblk) / sr0
    x = (0.3 * np.exp(1j * (2 * np.pi * 300e3 * t + (75e3 / 1e3) * np.sin(2 * np.pi * 1e3 * t))) * (1.0 + 0.2 * np.cos(2 * np.pi * 600.0 * t))
         + 0.2 * np.exp(1j * (2 * np.pi * -200e3 * t + (50e3 / 700.0) * np.sin(2 * np.pi * 700.0 * t)))
         + (0.02 + 0.01j) + 0.01 * (r.standard_normal(len(t)) + 1j * r.standard_normal(len(t)))).astype(np.complex64)
    x.view(np.float32).tofile(os.path.join(tmp, "iq.f32"))
    rr = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin"), os.path.join(tmp, "iq.f32"), str(sr0), str(B), tmp, str(nextra), str(wait_ms)],
                        capture_output=True, text=True, timeout=900)
    assert rr.returncode == 0, rr.stdout + rr.stderr
    ld = lambda name, dt: np.fromfile(os.path.join(tmp, name), dt)
    g_radio, g_steady = (ld(n + ".f32", np.float32).reshape(-1, 2) for n in ("radio", "steady"))
    c_radio, c_steady = (ld(n + "_counts.i32", np.int32) for n in ("radio", "steady"))
    g_lines = ld("lines.f32", np.float32).reshape(-1, N)
    # ---- the reference under the same schedule ----
    st = dict(sr=sr0, decim=1, dc=False, conj=False, win=2, rate=100.0)
    eff = lambda: st["sr"] / st["decim"]
    pre = S.RefPreproc(1, False, 50.0 / eff(), False)
    o_radio = S.RefChain(eff(), 250e3, 150e3, 300e3, S.MODES["WFM"])
    o_steady = S.RefChain(eff(), 250e3, 150e3, -200e3, S.MODES["WFM"])

    def new_spec():  # updateFFTPath: the Reshaper restarts (iq_frontend.cpp:269-309)
        nz, skip = capi.design_reshape_params(eff(), N, st["rate"])
        return S.OracleSpectrum(N, nz, skip, capi.design_fft_window(st["win"], nz))

    def rates_changed(new_decimator):
        pre.set(st["decim"], st["dc"], 50.0 / eff(), st["conj"], new_decimator)  # genDCBlockRate: iq_frontend.h:55-57
        o_radio.set_in_samplerate(eff())
        o_steady.set_in_samplerate(eff())
        return new_spec()

    spec = new_spec()
    e_radio, e_steady, e_lines, exact_lines = [], [], [], 0
    for b in range(nblk):
        if b == 2:
            st["win"] = 1  # BLACKMAN
            spec = new_spec()
        if b == 3:
            st["rate"] = 50.0
            spec = new_spec()
        if b == 4:
            o_radio.set_out_samplerate(50e3, 12.5e3, mode=S.MODES["NFM"])
        if b == 5:
            st["conj"] = True
            pre.set(st["decim"], st["dc"], 50.0 / eff(), True, False)
        if b == 6:
            o_radio.set_out_samplerate(24e3, 2.8e3, mode=S.MODES["USB"])
        if b == 9:
            o_radio.set_out_samplerate(250e3, 150e3, mode=S.MODES["WFM"])
        if b == 10:
            st["decim"] = 2
            spec = rates_changed(True)
        if b == 11:
            st["sr"] = 2.0e6
            spec = rates_changed(False)
        if b == 12:
            st["dc"] = True
            pre.set(st["decim"], True, 50.0 / eff(), st["conj"], False)
        if b == 13:
            st["sr"] = 2.4e6
            spec = rates_changed(False)
        y = pre.process(x[b * B:(b + 1) * B])
        ln = spec.push(y)
        e_lines.append(ln)
        if b < 10:
            exact_lines += len(ln)
        e_radio.append(o_radio.process(y)[1])
        e_steady.append(o_steady.process(y)[1])
    # every block of both radios exactly once, in order, every one from its first sample (the first 200 frames behind each change are looked at on their own)
    for name, got, cnt, exp in (("radio", g_radio, c_radio, e_radio), ("steady", g_steady, c_steady, e_steady)):
        assert [int(c) for c in cnt] == [len(e) for e in exp], (name, cnt.tolist(), [len(e) for e in exp])
        pos = 0
        for b, e in enumerate(exp):
            g = got[pos:pos + len(e)]
            pos += len(e)
            tol = 1e-5 * max(1.0, float(np.sqrt(np.mean(e ** 2))))
            h = min(len(e), 200)
            err, err_h = float(np.sqrt(np.mean((g - e) ** 2))), float(np.sqrt(np.mean((g[:h] - e[:h]) ** 2)))
            assert err < tol and err_h < tol, (name, "block", b, err, err_h, tol)
    # lines: bit-exact while the pre-processing chain only conjugates; behind the decimator / DC blocker (default arithmetic: matrix-core sums, a parallel scan —
    # include/sdrpp_gpu.h) within 0.05 dB; none lost, none twice across the five restarts of the framing
    o = np.concatenate([l for l in e_lines if len(l)])
    assert g_lines.shape == o.shape, (g_lines.shape, o.shape)
    assert np.array_equal(g_lines[:exact_lines], o[:exact_lines])
    if len(o) > exact_lines:
        d = np.abs(g_lines[exact_lines:] - o[exact_lines:])
        assert float(d.max()) < 0.05, float(d.max())
    return rr.stdout


@pytest.mark.skipif(not S.ref_available(), reason="oracle/_ref not built (needs the reference tree at build time)")
def test_reconfigure2_while_running_on_the_emulator():
    """setFFTWindow / setFFTRate / setInvertIQ / setBuffering on + off / setDecimation / setSampleRate / setDCBlocking and the radio's demodulator switch WFM -> NFM -> USB ->
    WFM (RxVFO::setOutSamplerate + a new demodulator), RxVFO::setInSamplerate through the front end's rate changes — all between the blocks of a RUNNING pipelined graph
    (iq_frontend.cpp:76-130, rx_vfo.h:35-58, radio_module.h:419-563), replayed on the compiled reference's objects.  CPU emulator build."""
    with tempfile.TemporaryDirectory() as tmp:
        out = _run_reconfig2_and_check(_build(tmp, lib="emu", source="test_reconfig2.cpp"), tmp, 0, 60000)
        assert "blocks 14" in out


@pytest.mark.gpu
@pytest.mark.skipif(not S.ref_available(), reason="oracle/_ref not built")
def test_reconfigure2_while_running_on_the_device():
    """The same on the device with 18 more radios (matrix front end: the blocks run as launches of the pipeline)."""
    with tempfile.TemporaryDirectory() as tmp:
        out = _run_reconfig2_and_check(_build(tmp, source="test_reconfig2.cpp"), tmp, 18, 20000)
        assert "blocks 14" in out


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


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bypass", "buffered", "pipelined", "pipelined_groups"])
def test_threaded_graph_matches_oracle(mode):
    with tempfile.TemporaryDirectory() as tmp:
        _run_graph_and_check(_build(tmp), mode, tmp)


@pytest.mark.gpu
def test_reference_header_build_runs_on_the_device():
    """oracle/_ref/test_blocks_ref: the host mirror compiled HERE against the reference's real dsp headers + Demodulator interface (the
    GPU box has no reference tree; the binary travels like the other oracle/_ref files), linked with the product library."""
    if not os.path.exists(REF_BIN):
        pytest.skip("oracle/_ref/test_blocks_ref not built (needs the reference tree at build time)")
    with tempfile.TemporaryDirectory() as tmp:
        _run_graph_and_check(REF_BIN, "bypass", tmp)


def _run_multi_and_check(exe, tmp, drain_ms):
    """Two streams with different FM modulation through sdrpp_gpu::StreamBank: each stream's lines and audio match its OWN oracle."""
    from sdrplusplus_amd import capi

    r = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin"), tmp, str(drain_ms)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    sr, B, N, nblk = 2.4e6, 12000, 4096, 6
    nz, skip = capi.design_reshape_params(sr, N, 100.0)
    t = np.arange(B * nblk, dtype=np.float64) / sr
    for s in range(2):
        ph = 2.0 * np.pi * 300e3 * t + (20.0 if s else 60.0) * np.sin(2.0 * np.pi * (1700.0 if s else 1000.0) * t)
        amp = 0.2 if s else 0.4
        x = np.empty(len(t), np.complex64)
        x.real = (amp * np.cos(ph)).astype(np.float32)
        x.imag = (amp * np.sin(ph)).astype(np.float32)
        spec = S.OracleSpectrum(N, nz, skip, capi.design_fft_window(2, nz))
        ch = S.OracleChain(sr, 250e3, 150e3, 300e3, S.MODES["WFM"])
        ol = np.concatenate([spec.push(x[b * B:(b + 1) * B]) for b in range(nblk)])
        oa = np.concatenate([ch.process(x[b * B:(b + 1) * B])[1] for b in range(nblk)])
        lines = np.fromfile(os.path.join(tmp, "lines_%d.f32" % s), np.float32).reshape(-1, N)
        audio = np.fromfile(os.path.join(tmp, "audio_%d.f32" % s), np.float32).reshape(-1, 2)
        assert lines.shape == ol.shape and np.array_equal(lines, ol), s
        assert audio.shape == oa.shape and np.sqrt(np.mean((audio - oa) ** 2)) < 1e-5, s


def test_stream_bank_two_streams_on_the_emulator():
    """sdrpp_gpu::StreamBank (one front end + worker per stream / device, one line handler): the C++ multi-device host, on the emulator."""
    with tempfile.TemporaryDirectory() as tmp:
        _run_multi_and_check(_build(tmp, lib="emu", source="test_multi.cpp"), tmp, 4000)


@pytest.mark.gpu
def test_stream_bank_two_streams_on_the_device():
    with tempfile.TemporaryDirectory() as tmp:
        _run_multi_and_check(_build(tmp, source="test_multi.cpp"), tmp, 500)


@pytest.mark.gpu
def test_rccl_line_gather_cpp():
    """sdrpp_gpu::LineGather (host/sdrpp_gpu_rccl.h): the C++ host's RCCL gather of waterfall lines — one context and one RCCL rank per
    visible device (one on the single-GPU test box: the collectives still run), gathered lines == sdrpp_fft_read of every stream."""
    rocm = "/opt/rocm"
    if not os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")):
        pytest.skip("no RCCL headers")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "test_rccl_gather")
        subprocess.run(["g++", "-std=c++17", "-O2", "-w", "-D__HIP_PLATFORM_AMD__", "-I" + rocm + "/include", "-I" + os.path.join(ROOT, "tests", "host_cpp", "standalone"), "-o", exe,
                        os.path.join(ROOT, "tests", "host_cpp", "test_rccl_gather.cpp"),
                        "-L" + CSRC, "-lsdrpp_gpu", "-L" + rocm + "/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath," + rocm + "/lib", "-Wl,-rpath," + CSRC, "-lpthread"], check=True)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "ok" in r.stdout and "ranks" in r.stdout


@pytest.mark.gpu
def test_stream_bank_line_gather_over_rccl():
    """sdrpp_gpu::BankLineGather: StreamBank + LineGather wired together — the newest line of every stream stays on its device
    (IQFrontEnd::keepDeviceLine) until one grouped RCCL send / receive brings them to the display device; equal, bit for bit, to the lines
    the bank's handler received.  Degrades to one rank when the streams share the one GPU of the test box."""
    rocm = "/opt/rocm"
    if not os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")):
        pytest.skip("no RCCL headers")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "test_bank_gather")
        subprocess.run(["g++", "-std=c++17", "-O2", "-w", "-D__HIP_PLATFORM_AMD__", "-I" + rocm + "/include", "-I" + os.path.join(ROOT, "tests", "host_cpp", "standalone"), "-o", exe,
                        os.path.join(ROOT, "tests", "host_cpp", "test_bank_gather.cpp"), "-L" + CSRC, "-lsdrpp_gpu", "-L" + rocm + "/lib", "-lrccl", "-lamdhip64",
                        "-Wl,-rpath," + rocm + "/lib", "-Wl,-rpath," + CSRC, "-lpthread"], check=True)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([exe, os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin")], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "ok" in r.stdout and "ranks" in r.stdout


@pytest.mark.gpu
def test_bench_protocol_over_rccl_with_one_rank():
    """bench.py's N > 1 leg on the one GPU there is: init_process_group("nccl") with a world of one, the pipelined StreamRunner protocol,
    the line batches going through dist.gather on RCCL, barrier + all_reduce(MAX) of the elapsed time — the JSON line says so."""
    import json
    import sys

    env = dict(os.environ, SDRPP_BENCH_FORCE_RCCL="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "6", "--no-others", "--no-by-push", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["rccl"]["rccl_ranks"] == 1 and d["rccl"]["backend"] == "nccl" and len(d["rccl"]["devices"]) == 1
    assert d["value"] > 0 and d["n_gpus"] == 1 and d["roofline"]["kernel"] == "tick"
