"""ctypes access to the test oracle (oracle/liboracle.so) and, when present, the compiled reference
(oracle/_ref/libsdrpp_ref.so).  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import this module; nothing under sdrplusplus_amd/ does."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
PLANS_PATH = os.path.join(ROOT, "sdrplusplus_amd", "data", "decim_plans.bin")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def locked_make(*args):
    """`make` under a file lock: the workers of a parallel run (pytest-xdist) on a fresh tree would otherwise build — and load — the same library at once."""
    import fcntl

    with open(os.path.join(ROOT, "tests", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.run(["make"] + list(args), check=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def build_oracle():
    """(Re)build oracle/liboracle.so (and oracle/_ref when /root/reference exists)."""
    locked_make("-C", ORACLE_DIR, "-s", "all")


_oracle = None
_ref = {}


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.sdrpp_oracle_log2f.restype = C.c_float
        L.sdrpp_oracle_log2f.argtypes = [C.c_float]
        L.sdrpp_oracle_log2f_non_ieee.restype = C.c_float
        L.sdrpp_oracle_log2f_non_ieee.argtypes = [C.c_float]
        L.sdrpp_oracle_fft.argtypes = [C.c_int, c_float_p, c_float_p]
        L.sdrpp_oracle_twiddle.argtypes = [C.c_int, C.c_int, c_float_p, c_float_p]
        L.orc_nuttall.restype = C.c_double
        L.orc_nuttall.argtypes = [C.c_double, C.c_double]
        L.orc_blackman.restype = C.c_double
        L.orc_blackman.argtypes = [C.c_double, C.c_double]
        L.orc_estimate_tap_count.argtypes = [C.c_double, C.c_double]
        for f in (L.orc_low_pass, L.orc_high_pass):
            f.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, c_float_p, C.c_int]
        L.orc_gen_reshape_params.argtypes = [C.c_double, C.c_int, C.c_double, c_int_p, c_int_p]
        L.orc_fft_window.argtypes = [C.c_int, C.c_int, c_float_p]
        L.orc_power_spectrum.argtypes = [c_float_p, C.c_float, C.c_int, c_float_p]
        L.orc_spectrum_create.restype = C.c_void_p
        L.orc_spectrum_create.argtypes = [C.c_int, C.c_int, C.c_int, c_float_p]
        L.orc_spectrum_destroy.argtypes = [C.c_void_p]
        L.orc_spectrum_push.argtypes = [C.c_void_p, c_float_p, C.c_int, c_float_p, C.c_int]
        L.orc_do_zoom.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p]
        L.orc_waterfall_view.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, c_int_p, c_int_p]
        L.orc_palette_index.argtypes = [c_float_p, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int32)]
        L.orc_plans_load.restype = C.c_void_p
        L.orc_plans_load.argtypes = [C.c_char_p]
        L.orc_plans_free.argtypes = [C.c_void_p]
        L.orc_rxvfo_create.restype = C.c_void_p
        L.orc_rxvfo_create.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_rxvfo_destroy.argtypes = [C.c_void_p]
        L.orc_rxvfo_set_offset.argtypes = [C.c_void_p, C.c_double]
        L.orc_rxvfo_set_bandwidth.argtypes = [C.c_void_p, C.c_double]
        L.orc_rxvfo_set_bandwidth.restype = None
        L.orc_rxvfo_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.orc_rxvfo_info.argtypes = [C.c_void_p] + [c_int_p] * 8
        L.orc_rxvfo_phase_delta.argtypes = [C.c_void_p, c_float_p, c_float_p]
        L.orc_rxvfo_set_ideal_nco.argtypes = [C.c_void_p, C.c_int]
        L.orc_demod_set_ideal_nco.argtypes = [C.c_void_p, C.c_int]
        L.orc_xlator_set_ideal.argtypes = [C.c_void_p, C.c_int]
        L.orc_demod_create.restype = C.c_void_p
        L.orc_demod_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_demod_destroy.argtypes = [C.c_void_p]
        L.orc_demod_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.orc_demod_audio_taps.argtypes = [C.c_void_p]
        L.orc_resampler_create.restype = C.c_void_p
        L.orc_resampler_create.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        L.orc_resampler_destroy.argtypes = [C.c_void_p]
        L.orc_resampler_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.orc_resampler_info.argtypes = [C.c_void_p] + [c_int_p] * 6
        L.orc_deemp_create.restype = C.c_void_p
        L.orc_deemp_create.argtypes = [C.c_double, C.c_double]
        L.orc_deemp_destroy.argtypes = [C.c_void_p]
        L.orc_deemp_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.orc_preproc_create.restype = C.c_void_p
        L.orc_preproc_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int]
        L.orc_preproc_destroy.argtypes = [C.c_void_p]
        L.orc_preproc_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.orc_int16_to_float.argtypes = [C.POINTER(C.c_int16), c_float_p, C.c_int]
        L.orc_fir_create.restype = C.c_void_p
        L.orc_fir_create.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int]
        L.orc_fir_destroy.argtypes = [C.c_void_p]
        L.orc_fir_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.orc_xlator_create.restype = C.c_void_p
        L.orc_xlator_create.argtypes = [C.c_double, C.c_double]
        L.orc_xlator_destroy.argtypes = [C.c_void_p]
        L.orc_xlator_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.orc_xlator_state.argtypes = [C.c_void_p] + [c_float_p] * 4
        _oracle = L
    return _oracle


def ref_available(fast=False):
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libsdrpp_ref_fast.so" if fast else "libsdrpp_ref.so"))


def ref(fast=False):
    """The reference's own code compiled against oracle/shim (None if oracle/_ref was never built)."""
    if fast not in _ref:
        path = os.path.join(ORACLE_DIR, "_ref", "libsdrpp_ref_fast.so" if fast else "libsdrpp_ref.so")
        if not os.path.exists(path):
            _ref[fast] = None
            return None
        L = C.CDLL(path)
        for f in (L.ref_low_pass, L.ref_high_pass):
            f.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, c_float_p, C.c_int]
        L.ref_nuttall.restype = C.c_double
        L.ref_nuttall.argtypes = [C.c_double, C.c_double]
        L.ref_blackman.restype = C.c_double
        L.ref_blackman.argtypes = [C.c_double, C.c_double]
        L.ref_rxvfo_create.restype = C.c_void_p
        L.ref_rxvfo_create.argtypes = [C.c_double] * 4
        L.ref_rxvfo_destroy.argtypes = [C.c_void_p]
        L.ref_rxvfo_set_offset.argtypes = [C.c_void_p, C.c_double]
        L.ref_rxvfo_set_bandwidth.argtypes = [C.c_void_p, C.c_double]
        L.ref_rxvfo_set_bandwidth.restype = None
        L.ref_rxvfo_set_in_samplerate.argtypes = [C.c_void_p, C.c_double]
        L.ref_rxvfo_set_in_samplerate.restype = None
        L.ref_rxvfo_set_out_samplerate.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.ref_rxvfo_set_out_samplerate.restype = None
        L.ref_rxvfo_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.ref_demod_create.restype = C.c_void_p
        L.ref_demod_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int]
        L.ref_demod_destroy.argtypes = [C.c_void_p]
        L.ref_demod_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.ref_resampler_create.restype = C.c_void_p
        L.ref_resampler_create.argtypes = [C.c_double, C.c_double]
        L.ref_resampler_destroy.argtypes = [C.c_void_p]
        L.ref_resampler_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.ref_deemp_create.restype = C.c_void_p
        L.ref_deemp_create.argtypes = [C.c_double, C.c_double]
        L.ref_deemp_destroy.argtypes = [C.c_void_p]
        L.ref_deemp_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.ref_preproc_create.restype = C.c_void_p
        L.ref_preproc_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
        L.ref_preproc_destroy.argtypes = [C.c_void_p]
        L.ref_preproc_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.ref_preproc_set.restype = None
        L.ref_preproc_process.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p]
        L.ref_frontend_create.restype = C.c_void_p
        L.ref_frontend_create.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int]
        L.ref_frontend_feed.argtypes = [C.c_void_p, c_float_p, C.c_longlong, C.c_int, C.c_int, C.c_int]
        L.ref_frontend_lines.argtypes = [C.c_void_p, c_float_p, C.c_int]
        L.ref_frontend_destroy.argtypes = [C.c_void_p]
        L.ref_bench_cfg3.restype = C.c_double
        L.ref_bench_cfg3.argtypes = [c_float_p, C.c_longlong, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int]
        L.ref_bench_cfg.restype = C.c_double
        L.ref_bench_cfg.argtypes = [c_float_p, C.c_longlong, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), c_int_p,
                                    C.c_int, C.c_int, C.c_int]
        L.ref_bench_graph.restype = C.c_int
        L.ref_bench_graph.argtypes = [c_float_p, C.c_longlong, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), c_int_p,
                                      C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.ref_bench_stage.restype = C.c_double
        L.ref_bench_stage.argtypes = [C.c_int, c_float_p, C.c_longlong, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double]
        _ref[fast] = L
    return _ref[fast]


# ---------------------------------------------------------------------------------------------------------------------
# Convenience wrappers (numpy in / numpy out).  Complex data is np.complex64; stereo audio is float32 [n, 2].
# ---------------------------------------------------------------------------------------------------------------------
_plans_handle = None


def plans_handle():
    global _plans_handle
    if _plans_handle is None:
        _plans_handle = oracle().orc_plans_load(PLANS_PATH.encode())
        assert _plans_handle, "cannot load " + PLANS_PATH
    return _plans_handle


def c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


def oracle_fft(x):
    x = c64(x)
    out = np.empty_like(x)
    oracle().sdrpp_oracle_fft(len(x), _fp(x.view(np.float32)), _fp(out.view(np.float32)))
    return out


def oracle_low_pass(cutoff, trans, sr, odd=False):
    n = oracle().orc_estimate_tap_count(trans, sr) + 1
    t = np.zeros(n, dtype=np.float32)
    n = oracle().orc_low_pass(cutoff, trans, sr, int(odd), _fp(t), len(t))
    return t[:n].copy()


def oracle_fft_window(kind, nz):
    w = np.empty(nz, dtype=np.float32)
    oracle().orc_fft_window(kind, nz, _fp(w))
    return w


class OracleSpectrum:
    """Reshaper + IQFrontEnd::handler restatement (streaming)."""

    def __init__(self, fft_size, nz, skip, window):
        self.N, self.nz, self.skip = fft_size, nz, skip
        w = np.ascontiguousarray(window, dtype=np.float32)
        self.h = oracle().orc_spectrum_create(fft_size, nz, skip, _fp(w))

    def push(self, iq):
        iq = c64(iq)
        max_lines = len(iq) // max(1, self.nz + self.skip) + 2
        out = np.empty((max_lines, self.N), dtype=np.float32)
        n = oracle().orc_spectrum_push(self.h, _fp(iq.view(np.float32)), len(iq), _fp(out), max_lines)
        return out[:n].copy()

    def __del__(self):
        if getattr(self, "h", None):
            oracle().orc_spectrum_destroy(self.h)
            self.h = None


def oracle_do_zoom(offset, width, out_size, line):
    line = np.ascontiguousarray(line, dtype=np.float32)
    out = np.empty(out_size, dtype=np.float32)
    oracle().orc_do_zoom(offset, width, len(line), out_size, _fp(line), _fp(out))
    return out


def oracle_palette_index(zoomed, wmin, wmax):
    z = np.ascontiguousarray(zoomed, dtype=np.float32)
    idx = np.empty(len(z), dtype=np.int32)
    oracle().orc_palette_index(_fp(z), len(z), wmin, wmax, idx.ctypes.data_as(C.POINTER(C.c_int32)))
    return idx


class _Chain:
    """RxVFO + demodulator pair driven block by block, backed by either the oracle or the compiled reference."""

    def __init__(self, lib, prefix, in_sr, out_sr, bw, offset, mode, low_pass=True, agc_attack=50.0, agc_decay=5.0, carrier_agc=False, ideal_nco=False):
        """ideal_nco (oracle only, a TEST SWITCH): both frequency translations use a float64 NCO at arg(phaseDelta_f32) instead of the
        float recursion; everything else is the pinned restatement.  Isolates what the recursion's rounding contributes."""
        self.lib, self.p = lib, prefix
        g = lambda name: getattr(lib, prefix + name)
        if prefix == "orc_":
            self.vfo = g("rxvfo_create")(plans_handle(), in_sr, out_sr, bw, offset)
        else:
            self.vfo = g("rxvfo_create")(in_sr, out_sr, bw, offset)
        self.dem = None
        if mode is not None:
            self.dem = g("demod_create")(mode, bw, out_sr, int(low_pass), agc_attack, agc_decay, int(carrier_agc))
        if ideal_nco:
            assert prefix == "orc_", "the ideal-NCO switch exists in the oracle only"
            lib.orc_rxvfo_set_ideal_nco(self.vfo, 1)
            if self.dem is not None:
                lib.orc_demod_set_ideal_nco(self.dem, 1)

    def vfo_process(self, iq):
        iq = c64(iq)
        out = np.empty(len(iq) + 16, dtype=np.complex64)
        n = getattr(self.lib, self.p + "rxvfo_process")(self.vfo, len(iq), _fp(iq.view(np.float32)), _fp(out.view(np.float32)))
        return out[:n].copy()

    def demod_process(self, ifs):
        ifs = c64(ifs)
        out = np.empty((len(ifs) + 1, 2), dtype=np.float32)
        n = getattr(self.lib, self.p + "demod_process")(self.dem, len(ifs), _fp(ifs.view(np.float32)), _fp(out))
        return out[:n].copy()

    def process(self, iq):
        ifs = self.vfo_process(iq)
        if self.dem is None:
            return ifs, None
        return ifs, self.demod_process(ifs)

    def set_offset(self, offset):
        getattr(self.lib, self.p + "rxvfo_set_offset")(self.vfo, offset)

    def set_bandwidth(self, bandwidth):
        """RxVFO::setBandwidth (rx_vfo.h:60-70): new channel taps, the filter's delay line kept (fir.h:31-52)."""
        getattr(self.lib, self.p + "rxvfo_set_bandwidth")(self.vfo, float(bandwidth))

    def set_in_samplerate(self, sr):
        """RxVFO::setInSamplerate (rx_vfo.h:35-43) — the compiled reference only (RefChain): the demodulator lives on."""
        assert self.p == "ref_", "only the compiled reference replays this setter"
        self.lib.ref_rxvfo_set_in_samplerate(self.vfo, float(sr))

    def set_out_samplerate(self, sr, bandwidth, mode="same", **demod_kw):
        """RxVFO::setOutSamplerate (rx_vfo.h:45-58) and, with `mode`, the radio module's demodulator switch (a NEW demodulator object, radio_module.h:419-563)."""
        assert self.p == "ref_", "only the compiled reference replays this setter"
        self.lib.ref_rxvfo_set_out_samplerate(self.vfo, float(sr), float(bandwidth))
        if mode != "same":
            if self.dem:
                self.lib.ref_demod_destroy(self.dem)
                self.dem = None
            if mode is not None:
                self.dem = self.lib.ref_demod_create(mode, float(bandwidth), float(sr), int(demod_kw.get("low_pass", True)), demod_kw.get("agc_attack", 50.0), demod_kw.get("agc_decay", 5.0), int(demod_kw.get("carrier_agc", False)))

    def close(self):
        if self.vfo:
            getattr(self.lib, self.p + "rxvfo_destroy")(self.vfo)
            self.vfo = None
        if self.dem:
            getattr(self.lib, self.p + "demod_destroy")(self.dem)
            self.dem = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def OracleChain(*a, **k):
    return _Chain(oracle(), "orc_", *a, **k)


def RefChain(*a, fast=False, **k):
    lib = ref(fast)
    assert lib is not None, "oracle/_ref not built"
    return _Chain(lib, "ref_", *a, **k)


def oracle_rxvfo_info(chain):
    vals = [C.c_int() for _ in range(8)]
    oracle().orc_rxvfo_info(chain.vfo, *[C.byref(v) for v in vals])
    keys = ["mode", "predec", "interp", "decim", "rtaps", "taps_per_phase", "chan_taps", "filter_needed"]
    return dict(zip(keys, [v.value for v in vals]))


MODES = {"WFM": 0, "NFM": 1, "AM": 2, "USB": 3, "LSB": 4, "DSB": 5}


class RefPreproc:
    """The compiled reference's pre-processing chain objects (PowerDecimator -> DCBlocker -> Conjugate) with the re-planning IQFrontEnd's setters do to them
    while the stream runs (oracle/ref_api.cpp: ref_preproc_set)."""

    def __init__(self, ratio=1, dc_blocking=False, dc_rate=1e-5, conjugate=False):
        self.L = ref()
        assert self.L is not None, "oracle/_ref not built"
        self.h = self.L.ref_preproc_create(int(ratio), int(bool(dc_blocking)), float(dc_rate), int(bool(conjugate)))

    def set(self, ratio, dc_blocking, dc_rate, conjugate, new_decimator=False):
        self.L.ref_preproc_set(self.h, int(ratio), int(bool(dc_blocking)), float(dc_rate), int(bool(conjugate)), int(bool(new_decimator)))

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.empty(len(x) + 8, np.complex64)
        n = self.L.ref_preproc_process(self.h, len(x), _fp(x.view(np.float32)), _fp(out.view(np.float32))) if len(x) else 0
        return out[:n].copy()

    def __del__(self):
        try:
            self.L.ref_preproc_destroy(self.h)
        except Exception:
            pass


class OraclePreproc:
    """IQFrontEnd pre-processing chain on the oracle (orc_preproc_*): PowerDecimator -> DCBlocker -> Conjugate."""

    def __init__(self, ratio, dc_blocking, dc_rate, conjugate):
        self.o = oracle()
        self.h = self.o.orc_preproc_create(plans_handle(), int(ratio), int(bool(dc_blocking)), float(dc_rate), int(bool(conjugate)))

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.empty(len(x) + 8, np.complex64)
        n = self.o.orc_preproc_process(self.h, len(x), _fp(x.view(np.float32)), _fp(out.view(np.float32))) if len(x) else 0
        return out[:n].copy()

    def __del__(self):
        try:
            self.o.orc_preproc_destroy(self.h)
        except Exception:
            pass
