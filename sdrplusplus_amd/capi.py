"""ctypes binding of the C-ABI in include/sdrpp_gpu.h (sdrplusplus_amd/csrc/libsdrpp_gpu.so).

The library is the product: every data-path call runs hand-written HIP kernels on a gfx950 device and there is no CPU
fallback — `Context()` raises `SdrppError` (SDRPP_ERR_NO_DEVICE) when no GPU is present, and importing this module
raises `ImportError` when the shared library has not been built (`python -c "import __graft_entry__ as g; g.build()"`).

The library that is loaded is always DEFAULT_LIB below — no environment override.  (The test suite's CPU legs assign
`capi.DEFAULT_LIB` themselves to run host logic against the fiber-emulator build in tests/emu; nothing in this package does.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libsdrpp_gpu.so")

MAX_DECIM_STAGES = 4
NUM_KERNEL_FAMILIES = 13  # SDRPP_NUM_KERNEL_FAMILIES (checked against the header in tests/test_capi_host.py)

DEMOD_RAW, DEMOD_WFM, DEMOD_NFM, DEMOD_AM, DEMOD_USB, DEMOD_LSB, DEMOD_DSB = -1, 0, 1, 2, 3, 4, 5

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_int32_p = C.POINTER(C.c_int32)


class VfoDesc(C.Structure):
    """struct sdrpp_vfo_desc (include/sdrpp_gpu.h)."""

    _fields_ = [
        ("phase_delta_re", C.c_float),
        ("phase_delta_im", C.c_float),
        ("n_stages", C.c_int),
        ("stage_decim", C.c_int * MAX_DECIM_STAGES),
        ("stage_ntaps", C.c_int * MAX_DECIM_STAGES),
        ("stage_taps", c_float_p * MAX_DECIM_STAGES),
        ("interp", C.c_int),
        ("decim", C.c_int),
        ("resamp_ntaps", C.c_int),
        ("resamp_taps", c_float_p),
        ("chan_ntaps", C.c_int),
        ("chan_taps", c_float_p),
        ("demod", C.c_int),
        ("inv_deviation", C.c_float),
        ("audio_ntaps", C.c_int),
        ("audio_taps", c_float_p),
        ("agc_set_point", C.c_float),
        ("agc_attack", C.c_float),
        ("agc_decay", C.c_float),
        ("agc_max_gain", C.c_float),
        ("agc_max_output_amp", C.c_float),
        ("agc_init_gain", C.c_float),
        ("am_carrier_agc", C.c_int),
        ("dc_block_rate", C.c_float),
        ("ssb_phase_delta_re", C.c_float),
        ("ssb_phase_delta_im", C.c_float),
        ("nco_mode", C.c_int),
    ]


class AfDesc(C.Structure):
    """struct sdrpp_af_desc (include/sdrpp_gpu.h): the radio AF chain behind a demodulating VFO."""

    _fields_ = [
        ("n_stages", C.c_int),
        ("stage_decim", C.c_int * MAX_DECIM_STAGES),
        ("stage_ntaps", C.c_int * MAX_DECIM_STAGES),
        ("stage_taps", c_float_p * MAX_DECIM_STAGES),
        ("interp", C.c_int),
        ("decim", C.c_int),
        ("resamp_ntaps", C.c_int),
        ("resamp_taps", c_float_p),
        ("hpf_ntaps", C.c_int),
        ("hpf_taps", c_float_p),
        ("deemph_alpha", C.c_float),
    ]


class Result(C.Structure):
    """struct sdrpp_result (include/sdrpp_gpu.h): one block's results in the library's page-locked host memory (pipelined mode)."""

    _fields_ = [
        ("ticket", C.c_uint64),
        ("n_vfo", C.c_int),
        ("ids", c_int_p),
        ("offsets", C.POINTER(C.c_int64)),
        ("counts", c_int_p),
        ("samples", c_float_p),
        ("n_lines", C.c_int),
        ("fft_size", C.c_int),
        ("data_width", C.c_int),
        ("zoomed", c_float_p),
        ("index", c_int32_p),
        ("raw", c_float_p),
        ("n_iq", C.c_int),
        ("iq", c_float_p),
    ]


ABI_VERSION = 2    # SDRPP_ABI_VERSION (include/sdrpp_gpu.h)
RESULT_SLOTS = 24  # SDRPP_RESULT_SLOTS: launches (one block each, or a group of up to GROUP_MAX: sdrpp_set_pipeline_group) whose pipelined results can exist at a time
GROUP_MAX = 32     # SDRPP_GROUP_MAX


class SdrppError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sdrpp error %d: %s" % (code, msg))
        self.code = code


_lib = None
_lib_path = None


def lib_path():
    return DEFAULT_LIB


def load():
    """Load (once) and prototype the shared library."""
    global _lib, _lib_path
    path = lib_path()
    if _lib is not None and _lib_path == path:
        return _lib
    if not os.path.exists(path):
        raise ImportError("%s not built — run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.sdrpp_create.argtypes = [C.c_int, C.c_int64, C.POINTER(vp)]
    L.sdrpp_destroy.argtypes = [vp]
    L.sdrpp_strerror.restype = C.c_char_p
    L.sdrpp_strerror.argtypes = [C.c_int]
    L.sdrpp_last_error.restype = C.c_char_p
    L.sdrpp_last_error.argtypes = [vp]
    L.sdrpp_set_stream.argtypes = [vp, vp]
    L.sdrpp_sync.argtypes = [vp]
    L.sdrpp_device_info.argtypes = [vp, C.c_char_p, C.c_int]
    L.sdrpp_abi_version.argtypes = [c_int_p]
    sz = C.c_int()
    if L.sdrpp_abi_version(C.byref(sz)) != ABI_VERSION or sz.value != C.sizeof(VfoDesc):
        raise ImportError("sdrpp_vfo_desc layout mismatch: library %d bytes, binding %d" % (sz.value, C.sizeof(VfoDesc)))
    L.sdrpp_abi_sizeof_af_desc.argtypes = []
    if L.sdrpp_abi_sizeof_af_desc() != C.sizeof(AfDesc):
        raise ImportError("sdrpp_af_desc layout mismatch: library %d bytes, binding %d" % (L.sdrpp_abi_sizeof_af_desc(), C.sizeof(AfDesc)))
    L.sdrpp_design_deemphasis_alpha.restype = C.c_float
    L.sdrpp_design_deemphasis_alpha.argtypes = [C.c_double, C.c_double]
    L.sdrpp_vfo_read_pcm.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int]
    L.sdrpp_preproc_read_pcm.argtypes = [vp, C.c_int, C.c_float, vp, C.c_int]
    L.sdrpp_vfo_read_compressed.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int]
    L.sdrpp_wf_configure.argtypes = [vp, C.c_int]
    L.sdrpp_wf_set_smoothing.argtypes = [vp, C.c_int, C.c_float]
    L.sdrpp_wf_set_hold.argtypes = [vp, C.c_int, C.c_float]
    L.sdrpp_wf_latest.argtypes = [vp, c_float_p, c_float_p]
    L.sdrpp_wf_signal_info.argtypes = [vp, C.c_double, C.c_double, C.c_double, c_float_p, c_float_p]
    L.sdrpp_wf_raster.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, c_int32_p, c_int_p]
    L.sdrpp_preproc_configure.argtypes = [vp, C.c_int, c_int_p, c_int_p, C.POINTER(c_float_p), C.c_float, C.c_int]
    L.sdrpp_preproc_reconfigure.argtypes = [vp, C.c_int, c_int_p, c_int_p, C.POINTER(c_float_p), C.c_float, C.c_int, C.c_int]
    L.sdrpp_preproc_out_count.argtypes = [vp]
    L.sdrpp_preproc_read.argtypes = [vp, c_float_p, C.c_int]
    L.sdrpp_preproc_device_buffer.argtypes = [vp, C.POINTER(vp), c_int_p]
    L.sdrpp_vfo_set_af.argtypes = [vp, C.c_int, C.POINTER(AfDesc)]
    L.sdrpp_vfo_af_count.argtypes = [vp, C.c_int]
    L.sdrpp_vfo_af_read.argtypes = [vp, C.c_int, c_float_p, C.c_int]
    L.sdrpp_vfo_af_device_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), c_int_p]
    for f in (L.sdrpp_design_low_pass, L.sdrpp_design_high_pass):
        f.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, c_float_p, C.c_int]
    L.sdrpp_design_fft_window.argtypes = [C.c_int, C.c_int, c_float_p]
    L.sdrpp_design_reshape_params.restype = None
    L.sdrpp_design_reshape_params.argtypes = [C.c_double, C.c_int, C.c_double, c_int_p, c_int_p]
    L.sdrpp_design_phase_delta.restype = None
    L.sdrpp_design_phase_delta.argtypes = [C.c_double, C.c_double, c_float_p, c_float_p]
    L.sdrpp_design_resampler.argtypes = [C.c_double, C.c_double, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p, c_float_p, C.c_int]
    L.sdrpp_design_waterfall_view.restype = None
    L.sdrpp_design_waterfall_view.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, c_int_p, c_int_p]
    L.sdrpp_fft_configure.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_float_p]
    L.sdrpp_fft_disable.argtypes = [vp]
    L.sdrpp_fft_set_view.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
    L.sdrpp_fft_lines.argtypes = [vp]
    L.sdrpp_fft_read.argtypes = [vp, C.c_int, C.c_int, c_float_p, c_float_p, c_int32_p]
    L.sdrpp_fft_copy_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.sdrpp_fft_device_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), c_int_p]
    L.sdrpp_vfo_add.argtypes = [vp, C.POINTER(VfoDesc), c_int_p]
    L.sdrpp_vfo_remove.argtypes = [vp, C.c_int]
    L.sdrpp_vfo_replace.argtypes = [vp, C.c_int, C.POINTER(VfoDesc), C.c_int, c_int_p]
    L.sdrpp_vfo_count.argtypes = [vp]
    L.sdrpp_vfo_set_phase_delta.argtypes = [vp, C.c_int, C.c_float, C.c_float]
    L.sdrpp_vfo_set_channel_taps.argtypes = [vp, C.c_int, c_float_p, C.c_int]
    L.sdrpp_vfo_reset.argtypes = [vp, C.c_int]
    L.sdrpp_set_reference_block.argtypes = [vp, C.c_int]
    L.sdrpp_set_deferred.argtypes = [vp, C.c_int]
    L.sdrpp_push_pinned_async.argtypes = [vp, c_float_p, C.c_int64]
    L.sdrpp_push_wait.argtypes = [vp]
    L.sdrpp_push_stage.argtypes = [vp, C.c_int64, C.POINTER(c_float_p)]
    L.sdrpp_push_staged.argtypes = [vp, C.c_int64]
    L.sdrpp_push_staged_when.argtypes = [vp, C.c_int64, C.POINTER(C.c_uint32)]
    L.sdrpp_host_alloc.restype = vp
    L.sdrpp_host_alloc.argtypes = [C.c_size_t]
    L.sdrpp_host_free.restype = None
    L.sdrpp_host_free.argtypes = [vp]
    L.sdrpp_device_alloc.restype = vp
    L.sdrpp_device_alloc.argtypes = [vp, C.c_size_t]
    L.sdrpp_device_free.restype = None
    L.sdrpp_device_free.argtypes = [vp, vp]
    L.sdrpp_device_copy.restype = C.c_int
    L.sdrpp_device_copy.argtypes = [vp, vp, vp, C.c_size_t, C.c_int]
    L.sdrpp_pending.restype = C.c_int64
    L.sdrpp_pending.argtypes = [vp]
    L.sdrpp_vfo_read_many.argtypes = [vp, C.c_int, c_int_p, c_int_p, c_float_p, C.c_int64, C.POINTER(C.c_int64), c_int_p]
    L.sdrpp_set_nco_mode.argtypes = [vp, C.c_int]
    L.sdrpp_set_backend_pipeline.argtypes = [vp, C.c_int]
    L.sdrpp_vfo_set_ssb_phase_delta.argtypes = [vp, C.c_int, C.c_float, C.c_float]
    L.sdrpp_vfo_out_count.argtypes = [vp, C.c_int]
    L.sdrpp_vfo_read.argtypes = [vp, C.c_int, c_float_p, C.c_int]
    L.sdrpp_vfo_device_buffers.argtypes = [vp, C.c_int, C.POINTER(vp), c_int_p, C.POINTER(vp), c_int_p]
    L.sdrpp_push.argtypes = [vp, c_float_p, C.c_int64]
    L.sdrpp_push_device.argtypes = [vp, vp, C.c_int64]
    L.sdrpp_push_int16.argtypes = [vp, C.POINTER(C.c_int16), C.c_int64]
    L.sdrpp_set_pipelined.argtypes = [vp, C.c_int, C.c_int]
    L.sdrpp_set_pipeline_group.argtypes = [vp, C.c_int, C.c_int]
    L.sdrpp_pipeline_group_stats.argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
    L.sdrpp_ticket.restype = C.c_uint64
    L.sdrpp_ticket.argtypes = [vp]
    L.sdrpp_pipeline_flush.argtypes = [vp]
    L.sdrpp_pipeline_launch_held.argtypes = [vp]
    L.sdrpp_result_ready.argtypes = [vp, C.c_uint64]
    L.sdrpp_result_wait.argtypes = [vp, C.c_uint64, C.POINTER(Result)]
    L.sdrpp_result_release.argtypes = [vp, C.c_uint64]
    L.sdrpp_result_take_lines.argtypes = [vp, C.c_uint64, vp, vp, C.c_int, c_int_p]
    L.sdrpp_pipeline_stats.argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
    L.sdrpp_pipeline_role_name.restype = C.c_char_p
    L.sdrpp_pipeline_role_name.argtypes = [C.c_int]
    L.sdrpp_timing_enable.argtypes = [vp, C.c_int]
    L.sdrpp_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.sdrpp_kernel_family_name.restype = C.c_char_p
    L.sdrpp_kernel_family_name.argtypes = [C.c_int]
    _lib, _lib_path = L, path
    return L


# symbols include/sdrpp_gpu.h declares (checked by the CPU test-suite against the built library)
EXPORTED_SYMBOLS = [
    "sdrpp_create", "sdrpp_destroy", "sdrpp_strerror", "sdrpp_last_error", "sdrpp_set_stream", "sdrpp_sync", "sdrpp_abi_version", "sdrpp_device_info",
    "sdrpp_design_low_pass", "sdrpp_design_high_pass", "sdrpp_design_fft_window", "sdrpp_design_reshape_params",
    "sdrpp_design_phase_delta", "sdrpp_design_resampler", "sdrpp_design_waterfall_view", "sdrpp_design_deemphasis_alpha",
    "sdrpp_vfo_read_pcm", "sdrpp_vfo_read_compressed", "sdrpp_preproc_read_pcm",
    "sdrpp_wf_configure", "sdrpp_wf_set_smoothing", "sdrpp_wf_set_hold", "sdrpp_wf_latest", "sdrpp_wf_raster", "sdrpp_wf_signal_info",
    "sdrpp_preproc_configure", "sdrpp_preproc_reconfigure", "sdrpp_preproc_set_reference_order", "sdrpp_preproc_out_count", "sdrpp_preproc_read", "sdrpp_preproc_device_buffer",
    "sdrpp_vfo_set_af", "sdrpp_vfo_af_count", "sdrpp_vfo_af_read", "sdrpp_vfo_af_device_buffer", "sdrpp_abi_sizeof_af_desc",
    "sdrpp_fft_configure", "sdrpp_fft_disable", "sdrpp_fft_set_view", "sdrpp_fft_lines", "sdrpp_fft_read", "sdrpp_fft_copy_device", "sdrpp_fft_device_buffers",
    "sdrpp_vfo_add", "sdrpp_vfo_remove", "sdrpp_vfo_replace", "sdrpp_vfo_count", "sdrpp_vfo_set_phase_delta", "sdrpp_vfo_set_channel_taps", "sdrpp_vfo_reset",
    "sdrpp_vfo_out_count", "sdrpp_vfo_read", "sdrpp_vfo_device_buffers",
    "sdrpp_set_reference_block", "sdrpp_set_nco_mode", "sdrpp_set_backend_pipeline", "sdrpp_vfo_set_ssb_phase_delta", "sdrpp_vfo_read_many", "sdrpp_set_deferred", "sdrpp_push_pinned_async", "sdrpp_push_wait", "sdrpp_pending", "sdrpp_host_alloc", "sdrpp_host_free", "sdrpp_device_alloc", "sdrpp_device_free", "sdrpp_device_copy", "sdrpp_device_count",
    "sdrpp_push", "sdrpp_push_device", "sdrpp_push_int16", "sdrpp_push_stage", "sdrpp_push_staged", "sdrpp_push_staged_when",
    "sdrpp_set_pipelined", "sdrpp_set_pipeline_group", "sdrpp_pipeline_group_stats", "sdrpp_ticket", "sdrpp_pipeline_flush", "sdrpp_pipeline_launch_held", "sdrpp_result_ready", "sdrpp_result_wait", "sdrpp_result_release", "sdrpp_result_take_lines", "sdrpp_pipeline_stats", "sdrpp_pipeline_role_name",
    "sdrpp_timing_enable", "sdrpp_timing_read", "sdrpp_kernel_family_name",
]


# ---- host-side design helpers (pure CPU maths inside the library) ---------------------------------------------------------
def design_low_pass(cutoff, trans_width, sample_rate, odd=False):
    L = load()
    n = L.sdrpp_design_low_pass(cutoff, trans_width, sample_rate, int(odd), None, 0)
    taps = np.zeros(max(n, 1), dtype=np.float32)
    L.sdrpp_design_low_pass(cutoff, trans_width, sample_rate, int(odd), taps.ctypes.data_as(c_float_p), n)
    return taps[:n]


def design_high_pass(cutoff, trans_width, sample_rate, odd=False):
    L = load()
    n = L.sdrpp_design_high_pass(cutoff, trans_width, sample_rate, int(odd), None, 0)
    taps = np.zeros(max(n, 1), dtype=np.float32)
    L.sdrpp_design_high_pass(cutoff, trans_width, sample_rate, int(odd), taps.ctypes.data_as(c_float_p), n)
    return taps[:n]


def design_fft_window(kind, nz):
    w = np.empty(nz, dtype=np.float32)
    rc = load().sdrpp_design_fft_window(kind, nz, w.ctypes.data_as(c_float_p))
    if rc:
        raise SdrppError(rc, "bad window parameters")
    return w


def design_reshape_params(sample_rate, fft_size, fft_rate):
    skip, nz = C.c_int(), C.c_int()
    load().sdrpp_design_reshape_params(sample_rate, fft_size, fft_rate, C.byref(skip), C.byref(nz))
    return nz.value, skip.value


def design_phase_delta(offset_hz, sample_rate):
    re, im = C.c_float(), C.c_float()
    load().sdrpp_design_phase_delta(offset_hz, sample_rate, C.byref(re), C.byref(im))
    return re.value, im.value


def design_resampler(in_sr, out_sr, max_ratio=8192):
    L = load()
    mode, predec, interp, decim = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    n = L.sdrpp_design_resampler(in_sr, out_sr, max_ratio, C.byref(mode), C.byref(predec), C.byref(interp), C.byref(decim), None, 0)
    taps = np.zeros(max(n, 1), dtype=np.float32)
    if n > 0:
        L.sdrpp_design_resampler(in_sr, out_sr, max_ratio, C.byref(mode), C.byref(predec), C.byref(interp), C.byref(decim), taps.ctypes.data_as(c_float_p), n)
    return dict(mode=mode.value, predec=predec.value, interp=interp.value, decim=decim.value, taps=taps[:n])


def design_deemphasis_alpha(tau, sample_rate):
    return float(load().sdrpp_design_deemphasis_alpha(tau, sample_rate))


def design_waterfall_view(view_offset, view_bandwidth, whole_bandwidth, raw_fft_size):
    start, size = C.c_int(), C.c_int()
    load().sdrpp_design_waterfall_view(view_offset, view_bandwidth, whole_bandwidth, raw_fft_size, C.byref(start), C.byref(size))
    return start.value, size.value


# ---- context ---------------------------------------------------------------------------------------------------------------------
class Context:
    """One wideband IQ stream on one GPU (what one IQFrontEnd owns in the reference)."""

    def __init__(self, device=0, max_push=1_000_000):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.sdrpp_create(device, int(max_push), C.byref(h))
        if rc:
            raise SdrppError(rc, self.L.sdrpp_strerror(rc).decode())
        self.h = h
        self.max_push = int(max_push)
        self._keep = []  # numpy arrays referenced by descriptors during vfo_add
        self.fft_size = 0
        self.data_width = 0
        self.wf_height = 0
        self._res_scratch = Result()
        self._take_n_val = C.c_int(0)
        self._take_n = C.byref(self._take_n_val)

    def _chk(self, rc):
        if rc < 0:
            raise SdrppError(rc, "%s: %s" % (self.L.sdrpp_strerror(rc).decode(), self.L.sdrpp_last_error(self.h).decode()))
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.sdrpp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        buf = C.create_string_buffer(512)
        self._chk(self.L.sdrpp_device_info(self.h, buf, 512))
        return buf.value.decode()

    def set_stream(self, hip_stream_ptr):
        self._chk(self.L.sdrpp_set_stream(self.h, C.c_void_p(hip_stream_ptr)))

    def sync(self):
        self._chk(self.L.sdrpp_sync(self.h))

    def set_reference_block(self, ref_block):
        """Every push = consecutive reference blocks of `ref_block` samples (AGC look-ahead, rotator renormalisation); 0 = one push, one block."""
        self._chk(self.L.sdrpp_set_reference_block(self.h, int(ref_block)))

    def set_deferred(self, on):
        """Pushes are only staged; the next observing call processes them as one pass (results cover all of them)."""
        self._chk(self.L.sdrpp_set_deferred(self.h, int(bool(on))))

    def pending(self):
        return int(self.L.sdrpp_pending(self.h))

    def set_nco_mode(self, mode):
        """0 closed-form NCO (default), 1 the reference's float rotator recursion (parity mode).  Only while no VFO exists."""
        self._chk(self.L.sdrpp_set_nco_mode(self.h, int(mode)))

    def set_backend_pipeline(self, on):
        """FM back ends as one pipelined launch where that pays (default) / one launch per stage.  Bit-identical results."""
        self._chk(self.L.sdrpp_set_backend_pipeline(self.h, int(on)))

    # FFT branch
    def fft_configure(self, fft_size, nz, skip, window):
        w = np.ascontiguousarray(window, dtype=np.float32)
        assert len(w) == nz
        self._chk(self.L.sdrpp_fft_configure(self.h, fft_size, nz, skip, w.ctypes.data_as(c_float_p)))
        self.fft_size = fft_size

    def fft_disable(self):
        self._chk(self.L.sdrpp_fft_disable(self.h))

    def fft_set_view(self, draw_start, draw_size, data_width, wf_min=-120.0, wf_max=0.0):
        self._chk(self.L.sdrpp_fft_set_view(self.h, draw_start, draw_size, data_width, wf_min, wf_max))
        self.data_width = data_width

    def fft_lines(self):
        return self._chk(self.L.sdrpp_fft_lines(self.h))

    def fft_read(self, raw=True, zoomed=True):
        n = self.fft_lines()
        raw_a = np.empty((n, self.fft_size), dtype=np.float32) if raw else None
        zo = np.empty((n, self.data_width), dtype=np.float32) if (zoomed and self.data_width) else None
        ix = np.empty((n, self.data_width), dtype=np.int32) if (zoomed and self.data_width) else None
        if n:
            self._chk(self.L.sdrpp_fft_read(self.h, 0, n, raw_a.ctypes.data_as(c_float_p) if raw_a is not None else None,
                                            zo.ctypes.data_as(c_float_p) if zo is not None else None,
                                            ix.ctypes.data_as(c_int32_p) if ix is not None else None))
        return raw_a, zo, ix

    def fft_copy_device(self, first, n, raw_ptr=None, zoomed_ptr=None, index_ptr=None):
        return self._chk(self.L.sdrpp_fft_copy_device(self.h, first, n, C.c_void_p(raw_ptr), C.c_void_p(zoomed_ptr), C.c_void_p(index_ptr)))

    def fft_device_buffers(self):
        raw, zo, ix, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        self._chk(self.L.sdrpp_fft_device_buffers(self.h, C.byref(raw), C.byref(zo), C.byref(ix), C.byref(n)))
        return raw.value, zo.value, ix.value, n.value

    # VFO bank
    def vfo_add(self, desc, keepalive=()):
        vid = C.c_int()
        self._chk(self.L.sdrpp_vfo_add(self.h, C.byref(desc), C.byref(vid)))
        del keepalive
        return vid.value

    def vfo_remove(self, vid):
        self._chk(self.L.sdrpp_vfo_remove(self.h, vid))

    def vfo_replace(self, vid, desc, keep, keepalive=()):
        """sdrpp_vfo_replace: RxVFO::setInSamplerate / setOutSamplerate — a new description, the RxVFO's own state (keep & 1) and the demodulator's
        (keep & 2) carried over as the reference's objects carry it.  Returns the new handle."""
        nid = C.c_int()
        self._chk(self.L.sdrpp_vfo_replace(self.h, vid, C.byref(desc), int(keep), C.byref(nid)))
        del keepalive
        return nid.value

    def vfo_count(self):
        return self._chk(self.L.sdrpp_vfo_count(self.h))

    def vfo_set_phase_delta(self, vid, re, im):
        self._chk(self.L.sdrpp_vfo_set_phase_delta(self.h, vid, re, im))

    def vfo_set_channel_taps(self, vid, taps):
        t = np.ascontiguousarray(taps, dtype=np.float32)
        self._chk(self.L.sdrpp_vfo_set_channel_taps(self.h, vid, t.ctypes.data_as(c_float_p), len(t)))

    def vfo_set_ssb_phase_delta(self, vid, re, im):
        self._chk(self.L.sdrpp_vfo_set_ssb_phase_delta(self.h, vid, re, im))

    def vfo_reset(self, vid):
        self._chk(self.L.sdrpp_vfo_reset(self.h, vid))

    def vfo_out_count(self, vid):
        return self._chk(self.L.sdrpp_vfo_out_count(self.h, vid))

    def vfo_read(self, vid):
        n = self.vfo_out_count(vid)
        out = np.empty((max(n, 1), 2), dtype=np.float32)
        got = self._chk(self.L.sdrpp_vfo_read(self.h, vid, out.ctypes.data_as(c_float_p), n))
        return out[:got]

    def vfo_read_many(self, vids, which=None):
        """Outputs of many VFOs with one device-to-host copy -> list of [n, 2] float32 arrays (views into one buffer)."""
        n = len(vids)
        ids = (C.c_int * max(n, 1))(*[int(v) for v in vids])
        wh = (C.c_int * max(n, 1))(*[int(w) for w in which]) if which is not None else None
        offs = (C.c_int64 * max(n, 1))()
        cnts = (C.c_int * max(n, 1))()
        total = self._chk(self.L.sdrpp_vfo_read_many(self.h, n, ids, wh, None, 0, offs, cnts))  # size query: the buffer follows the real counts
        buf = np.empty((max(total, 1), 2), dtype=np.float32)
        self._chk(self.L.sdrpp_vfo_read_many(self.h, n, ids, wh, buf.ctypes.data_as(c_float_p), len(buf), offs, cnts))
        return [buf[offs[i]:offs[i] + cnts[i]] for i in range(n)]

    def vfo_read_pcm(self, vid, which, pcm_type, scale, max_frames):
        dt = np.int16 if pcm_type == 1 else np.int8
        out = np.empty((max(max_frames, 1), 2), dtype=dt)
        got = self._chk(self.L.sdrpp_vfo_read_pcm(self.h, vid, which, pcm_type, float(scale), out.ctypes.data_as(C.c_void_p), max_frames))
        return out[:got]

    def vfo_read_compressed(self, vid, which, pcm_type, max_frames):
        buf = np.empty(8 + max(max_frames, 1) * 8, dtype=np.uint8)
        got = self._chk(self.L.sdrpp_vfo_read_compressed(self.h, vid, which, pcm_type, buf.ctypes.data_as(C.POINTER(C.c_uint8)), len(buf)))
        return buf[:got]

    def wf_configure(self, height):
        self._chk(self.L.sdrpp_wf_configure(self.h, int(height)))
        self.wf_height = int(height)  # sdrpp_wf_raster always fills height x data_width entries: size the buffer from THIS value

    def wf_set_smoothing(self, enabled, speed=0.1):
        self._chk(self.L.sdrpp_wf_set_smoothing(self.h, int(bool(enabled)), float(speed)))

    def wf_set_hold(self, enabled, speed=60.0):
        self._chk(self.L.sdrpp_wf_set_hold(self.h, int(bool(enabled)), float(speed)))

    def wf_latest(self, data_width):
        a = np.empty(data_width, np.float32)
        b = np.empty(data_width, np.float32)
        self._chk(self.L.sdrpp_wf_latest(self.h, a.ctypes.data_as(c_float_p), b.ctypes.data_as(c_float_p)))
        return a, b

    def wf_signal_info(self, center_offset, bandwidth, whole_bandwidth):
        a, b = C.c_float(), C.c_float()
        ok = self._chk(self.L.sdrpp_wf_signal_info(self.h, center_offset, bandwidth, whole_bandwidth, C.byref(a), C.byref(b)))
        return (a.value, b.value) if ok else None

    def wf_raster(self, draw_start, draw_size, data_width, wf_min, wf_max):
        fb = np.empty((self.wf_height, data_width), np.int32)
        n = C.c_int()
        self._chk(self.L.sdrpp_wf_raster(self.h, int(draw_start), int(draw_size), int(data_width), float(wf_min), float(wf_max), fb.ctypes.data_as(c_int32_p), C.byref(n)))
        return fb, n.value

    def preproc_configure(self, stages=(), dc_rate=0.0, conjugate=False):
        """IQFrontEnd pre-processing chain: `stages` = [(decimation, taps)] of the PowerDecimator plan, dc_rate (0 = off), conjugate."""
        n = len(stages)
        dec = (C.c_int * max(n, 1))(*[int(d) for d, _ in stages])
        nt = (C.c_int * max(n, 1))(*[len(t) for _, t in stages])
        arrs = [np.ascontiguousarray(t, dtype=np.float32) for _, t in stages]
        ptrs = (c_float_p * max(n, 1))(*[a.ctypes.data_as(c_float_p) for a in arrs])
        self._chk(self.L.sdrpp_preproc_configure(self.h, n, dec, nt, ptrs, float(dc_rate), int(bool(conjugate))))

    def preproc_reconfigure(self, stages=(), dc_rate=0.0, conjugate=False, keep=3):
        """sdrpp_preproc_reconfigure: re-plan the chain while the stream runs; keep & 1 = the decimator's delay lines (same stages), keep & 2 = the DC blocker's estimate."""
        n = len(stages)
        dec = (C.c_int * max(n, 1))(*[int(d) for d, _ in stages])
        nt = (C.c_int * max(n, 1))(*[len(t) for _, t in stages])
        arrs = [np.ascontiguousarray(t, dtype=np.float32) for _, t in stages]
        ptrs = (c_float_p * max(n, 1))(*[a.ctypes.data_as(c_float_p) for a in arrs])
        self._chk(self.L.sdrpp_preproc_reconfigure(self.h, n, dec, nt, ptrs, float(dc_rate), int(bool(conjugate)), int(keep)))

    def preproc_set_reference_order(self, on=True):
        """Parity mode of the pre-processing chain: the reference's own tap-ordered multiply-then-add decimator and sequential DC blocker."""
        self._chk(self.L.sdrpp_preproc_set_reference_order(self.h, int(bool(on))))

    def preproc_read(self):
        n = self._chk(self.L.sdrpp_preproc_out_count(self.h))
        out = np.empty(max(n, 1), dtype=np.complex64)
        got = self._chk(self.L.sdrpp_preproc_read(self.h, out.view(np.float32).ctypes.data_as(c_float_p), n))
        return out[:got]

    def preproc_read_pcm(self, pcm_type, scale, max_samples=None):
        """Pre-processed wideband IQ of the most recent push as int16 (pcm_type 1) / int8 (0) pairs, converted on the device."""
        n = self._chk(self.L.sdrpp_preproc_out_count(self.h)) if max_samples is None else int(max_samples)
        out = np.empty((max(n, 1), 2), dtype=np.int16 if pcm_type == 1 else np.int8)
        got = self._chk(self.L.sdrpp_preproc_read_pcm(self.h, int(pcm_type), float(scale), out.ctypes.data_as(C.c_void_p), n))
        return out[:got]

    def vfo_set_af(self, vid, af_desc, keepalive=None):
        """Attach (or, with None, detach) the radio AF chain: resampler -> high-pass -> de-emphasis (sdrpp_vfo_set_af)."""
        self._chk(self.L.sdrpp_vfo_set_af(self.h, vid, C.byref(af_desc) if af_desc is not None else None))

    def vfo_af_count(self, vid):
        return self._chk(self.L.sdrpp_vfo_af_count(self.h, vid))

    def vfo_af_read(self, vid):
        n = self.vfo_af_count(vid)
        out = np.empty((max(n, 1), 2), dtype=np.float32)
        got = self._chk(self.L.sdrpp_vfo_af_read(self.h, vid, out.ctypes.data_as(c_float_p), n))
        return out[:got]

    def vfo_device_buffers(self, vid):
        o, i = C.c_void_p(), C.c_void_p()
        no, ni = C.c_int(), C.c_int()
        self._chk(self.L.sdrpp_vfo_device_buffers(self.h, vid, C.byref(o), C.byref(no), C.byref(i), C.byref(ni)))
        return o.value, no.value, i.value, ni.value

    def vfo_read_if(self, vid):
        """Complex IF stream (RxVFO::out) of the last push, copied from the device through the host-visible path."""
        import ctypes
        _, _, ptr, n = self.vfo_device_buffers(vid)
        self.sync()
        out = np.empty(max(n, 1), dtype=np.complex64)
        if n:
            _copy_from_device(self.L, out.ctypes.data, ptr, n * 8)
        return out[:n]

    # data path
    def push(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        self._chk(self.L.sdrpp_push(self.h, iq.view(np.float32).ctypes.data_as(c_float_p), len(iq)))

    def push_host_ptr(self, host_ptr, count):
        """sdrpp_push from a raw host address (e.g. pinned memory the caller owns)."""
        self._chk(self.L.sdrpp_push(self.h, C.cast(C.c_void_p(host_ptr), c_float_p), int(count)))

    def push_host_ptr_async(self, host_ptr, count):
        """sdrpp_push_pinned_async: deferred mode, page-locked source, no wait for the copy (the buffer stays untouched until results are read)."""
        self._chk(self.L.sdrpp_push_pinned_async(self.h, C.cast(C.c_void_p(host_ptr), c_float_p), int(count)))

    def push_wait(self):
        self._chk(self.L.sdrpp_push_wait(self.h))

    def push_int16(self, iq_i16):
        a = np.ascontiguousarray(iq_i16, dtype=np.int16)
        self._chk(self.L.sdrpp_push_int16(self.h, a.ctypes.data_as(C.POINTER(C.c_int16)), len(a) // 2))

    def push_device(self, dev_ptr, count):
        self._chk(self.L.sdrpp_push_device(self.h, C.c_void_p(dev_ptr), int(count)))

    # pipelined execution (one launch per block, results a few blocks late)
    def set_pipelined(self, on, result_flags=0):
        """result_flags: 1 = every VFO's output block (AF output where a chain is attached), 2 = zoomed lines + palette indices, 4 = raw dB lines,
        8 = the pre-processed IQ stream (with a pre-processing chain) into page-locked result slots."""
        self._chk(self.L.sdrpp_set_pipelined(self.h, int(bool(on)), int(result_flags)))

    def set_pipeline_group(self, max_blocks, adaptive=False, stable_words=False):
        """sdrpp_set_pipeline_group: up to `max_blocks` pushes per launch (every push keeps its own ticket and results); adaptive: the group follows
        what is queued on the device (1 while the host is the slower side); stable_words (flag 2): the words of sdrpp_push_staged_when live until the
        launch, a held push does not wait for its copy."""
        self._chk(self.L.sdrpp_set_pipeline_group(self.h, int(max_blocks), int(bool(adaptive)) | (2 if stable_words else 0)))

    def pipeline_launch_held(self):
        self._chk(self.L.sdrpp_pipeline_launch_held(self.h))

    def pipeline_group_stats(self):
        buf = (C.c_int64 * 8)()
        self._chk(self.L.sdrpp_pipeline_group_stats(self.h, buf, 8))
        return dict(groups=int(buf[0]), multi_groups=int(buf[1]), multi_blocks=int(buf[2]), largest=int(buf[3]), held=int(buf[4]))

    def ticket(self):
        return int(self.L.sdrpp_ticket(self.h))

    def pipeline_flush(self):
        self._chk(self.L.sdrpp_pipeline_flush(self.h))

    def result_ready(self, ticket):
        return bool(self._chk(self.L.sdrpp_result_ready(self.h, int(ticket))))

    def result_wait(self, ticket, copy=True):
        """-> dict(vfo={id: [n, 2] float32}, zoomed, index, raw); arrays are copies unless copy=False (then valid until result_release)."""
        r = Result()
        self._chk(self.L.sdrpp_result_wait(self.h, int(ticket), C.byref(r)))
        out = {"ticket": int(r.ticket), "vfo": {}, "n_lines": r.n_lines, "zoomed": None, "index": None, "raw": None, "iq": None}
        cp = (lambda a: a.copy()) if copy else (lambda a: a)
        for i in range(r.n_vfo):
            n = r.counts[i]
            if n > 0:
                a = np.ctypeslib.as_array(C.cast(C.addressof(r.samples.contents) + 8 * r.offsets[i], c_float_p), shape=(n, 2))
                out["vfo"][r.ids[i]] = cp(a)
            else:
                out["vfo"][r.ids[i]] = np.zeros((0, 2), np.float32)
        if r.n_lines > 0:
            if r.zoomed:
                out["zoomed"] = cp(np.ctypeslib.as_array(r.zoomed, shape=(r.n_lines, r.data_width)))
                out["index"] = cp(np.ctypeslib.as_array(r.index, shape=(r.n_lines, r.data_width)))
            if r.raw:
                out["raw"] = cp(np.ctypeslib.as_array(r.raw, shape=(r.n_lines, r.fft_size)))
        if r.n_iq > 0 and r.iq:
            out["iq"] = cp(np.ctypeslib.as_array(r.iq, shape=(2 * r.n_iq,)).view(np.complex64))
        return out

    def push_staged_from(self, iq):
        """sdrpp_push_stage + fill + sdrpp_push_staged (pipelined mode): the block goes into the library's page-locked slot by the caller's copy."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        slot = c_float_p()
        self._chk(self.L.sdrpp_push_stage(self.h, len(iq), C.byref(slot)))
        C.memmove(slot, iq.ctypes.data, len(iq) * 8)
        self._chk(self.L.sdrpp_push_staged(self.h, len(iq)))

    def push_staged_late_fill(self, iq, delay_s=0.002):
        """sdrpp_push_stage + sdrpp_push_staged_when: a second thread fills the slot (after `delay_s`, in two parts) while the call is
        already planning the block; the launch waits for the pending word.  Test helper for the C++ worker's staging protocol."""
        import threading
        import time

        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        slot = c_float_p()
        self._chk(self.L.sdrpp_push_stage(self.h, len(iq), C.byref(slot)))
        pending = C.c_uint32(2)
        dst = C.cast(slot, C.c_void_p).value
        half = (len(iq) // 2) * 8

        def fill():
            time.sleep(delay_s)
            C.memmove(dst, iq.ctypes.data, half)
            pending.value = 1
            time.sleep(delay_s)
            C.memmove(dst + half, iq.ctypes.data + half, len(iq) * 8 - half)
            pending.value = 0

        th = threading.Thread(target=fill)
        th.start()
        try:
            self._chk(self.L.sdrpp_push_staged_when(self.h, len(iq), C.byref(pending)))
        finally:
            th.join()

    def result_lines_into(self, ticket, dst_addr, max_lines):
        """sdrpp_result_take_lines: wait + copy + release for the zoomed lines of a block (result flag 2) in ONE call; the lines go to host
        address `dst_addr` (room for max_lines x data_width floats).  Returns the number of lines."""
        n = self._take_n
        rc = self.L.sdrpp_result_take_lines(self.h, ticket, dst_addr, None, max_lines, n)
        if rc < 0:
            self._chk(rc)
        return self._take_n_val.value

    def result_release(self, ticket):
        self._chk(self.L.sdrpp_result_release(self.h, int(ticket)))

    def pipeline_stats(self):
        """sdrpp_pipeline_stats -> dict(ticks, tick_blocks, pass_blocks, crowded_ticks, depth, roles={name: workgroups launched})."""
        buf = (C.c_int64 * 128)()
        n = self._chk(self.L.sdrpp_pipeline_stats(self.h, buf, 128))
        nroles = int(buf[5])
        roles = {}
        for r in range(nroles):
            if 8 + r < n and buf[8 + r]:
                roles[self.L.sdrpp_pipeline_role_name(r).decode()] = int(buf[8 + r])
        return dict(ticks=int(buf[0]), tick_blocks=int(buf[1]), pass_blocks=int(buf[2]), crowded_ticks=int(buf[3]), depth=int(buf[4]), set2_ticks=int(buf[6]), table_bytes=int(buf[7]), roles=roles)

    # measurement
    def timing_enable(self, on=True, families=None):
        """families: iterable of family indices to instrument (None = all)."""
        v = int(bool(on))
        if on and families is not None:
            mask = 0
            for f in families:
                mask |= 1 << int(f)
            v = (1 | (mask << 1)) if mask else 0
        self._chk(self.L.sdrpp_timing_enable(self.h, v))

    def family_index(self, name):
        for i in range(NUM_KERNEL_FAMILIES):
            if self.L.sdrpp_kernel_family_name(i).decode() == name:
                return i
        raise KeyError(name)

    def timing_read(self):
        ms = (C.c_double * NUM_KERNEL_FAMILIES)()
        ln = (C.c_int64 * NUM_KERNEL_FAMILIES)()
        self._chk(self.L.sdrpp_timing_read(self.h, ms, ln))
        return {self.L.sdrpp_kernel_family_name(i).decode(): (ms[i], ln[i]) for i in range(NUM_KERNEL_FAMILIES)}


_hip = None


def _copy_from_device(L, host_ptr, dev_ptr, nbytes):
    """hipMemcpy D2H through the HIP runtime the library itself links (emulator build: plain memmove)."""
    global _hip
    if "emu" in os.path.basename(lib_path()):
        C.memmove(host_ptr, dev_ptr, nbytes)
        return
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rc = _hip.hipMemcpy(C.c_void_p(host_ptr), C.c_void_p(dev_ptr), nbytes, 2)
    if rc:
        raise SdrppError(-4, "hipMemcpy D2H failed (%d)" % rc)
