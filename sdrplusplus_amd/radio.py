"""Host-side parameterisation of one VFO: what RxVFO::init (core/src/dsp/channel/rx_vfo.h:19-36), RationalResampler::
reconfigure (dsp/multirate/rational_resampler.h:120-165), PowerDecimator::reconfigure (power_decimator.h:93-111) and
the radio module's demodulators (decoder_modules/radio/src/demodulators/*.h) compute before any sample flows.
All double-precision design maths is done by the library's sdrpp_design_* functions (C++, libm) so that the constants
are bit-identical to the reference's; this module only wires them into a `sdrpp_vfo_desc`."""
import math
import os
import struct

import numpy as np

from . import capi

PLANS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "decim_plans.bin")

# radio-module defaults: (IF sample rate, default bandwidth) — wfm.h:268-270, nfm.h:56-58, am.h:76-78, usb.h:70-72, lsb.h, dsb.h
RADIO_DEFAULTS = {
    "WFM": (250000.0, 150000.0),
    "NFM": (50000.0, 12500.0),
    "AM": (15000.0, 10000.0),
    "USB": (24000.0, 2800.0),
    "LSB": (24000.0, 2800.0),
    "DSB": (24000.0, 4600.0),
}
DEMOD_CODES = {"RAW": capi.DEMOD_RAW, "WFM": capi.DEMOD_WFM, "NFM": capi.DEMOD_NFM, "AM": capi.DEMOD_AM,
               "USB": capi.DEMOD_USB, "LSB": capi.DEMOD_LSB, "DSB": capi.DEMOD_DSB}


class DecimPlans:
    """The reference's power-of-two decimation plans (numbers extracted by tools/extract_decim_plans.cpp)."""

    def __init__(self, path=PLANS_PATH):
        with open(path, "rb") as f:
            blob = f.read()
        assert blob[:4] == b"SDPL", "bad plan file"
        ver, n = struct.unpack_from("<II", blob, 4)
        assert ver == 1
        off = 12
        self.plans = {}
        for _ in range(n):
            ratio, ns = struct.unpack_from("<II", blob, off)
            off += 8
            stages = []
            for _ in range(ns):
                d, nt = struct.unpack_from("<II", blob, off)
                off += 8
                taps = np.frombuffer(blob, dtype="<f4", count=nt, offset=off).copy()
                off += 4 * nt
                stages.append((d, taps))
            self.plans[ratio] = stages
        self.max_ratio = 1 << n  # power_decimator.h:29-31

    def stages(self, ratio):
        return [] if ratio == 1 else self.plans[ratio]


_plans = None


def plans():
    global _plans
    if _plans is None:
        _plans = DecimPlans()
    return _plans


def hz_to_rads(freq, samplerate):
    return 2.0 * math.pi * (freq / samplerate)  # dsp/math/hz_to_rads.h:6-8


def _fp(a):
    return a.ctypes.data_as(capi.c_float_p)


def vfo_desc(in_sr, out_sr, bandwidth, offset, mode="RAW", low_pass=True, agc_attack=50.0, agc_decay=5.0, carrier_agc=False, nco_mode=0):
    """Build a sdrpp_vfo_desc for RxVFO(in_sr -> out_sr, bandwidth, offset) followed by radio demodulator `mode`.
    Returns (desc, keepalive) — keepalive holds the numpy arrays the descriptor points into (sdrpp_vfo_add copies them)."""
    d = capi.VfoDesc()
    d.nco_mode = int(nco_mode)  # 0: the context's NCO mode, 1: closed form, 2: the reference's float rotator recursion (this VFO only)
    keep = []
    # FrequencyXlator: xlator.init(NULL, -_offset, _inSamplerate) (rx_vfo.h:27)
    d.phase_delta_re, d.phase_delta_im = capi.design_phase_delta(-offset, in_sr)
    # RationalResampler
    rs = capi.design_resampler(in_sr, out_sr, plans().max_ratio)
    stages = plans().stages(rs["predec"]) if rs["mode"] in (0, 1) else []
    d.n_stages = len(stages)
    for i, (dec, taps) in enumerate(stages):
        t = np.ascontiguousarray(taps, dtype=np.float32)
        keep.append(t)
        d.stage_decim[i] = dec
        d.stage_ntaps[i] = len(t)
        d.stage_taps[i] = _fp(t)
    if rs["mode"] in (0, 2):
        rt = np.ascontiguousarray(rs["taps"], dtype=np.float32)
        keep.append(rt)
        d.interp, d.decim = rs["interp"], rs["decim"]
        d.resamp_ntaps = len(rt)
        d.resamp_taps = _fp(rt)
    else:
        d.interp, d.decim = 1, 1
        d.resamp_ntaps = 0
    # channel filter: generateTaps (rx_vfo.h:117-121), used when bandwidth != out rate (rx_vfo.h:24)
    if bandwidth != out_sr:
        fw = bandwidth / 2.0
        ct = capi.design_low_pass(fw, fw * 0.1, out_sr)
        keep.append(ct)
        d.chan_ntaps = len(ct)
        d.chan_taps = _fp(ct)
    else:
        d.chan_ntaps = 0
    # demodulator
    d.demod = DEMOD_CODES[mode]
    d.agc_set_point, d.agc_max_gain, d.agc_max_output_amp, d.agc_init_gain = 1.0, 10e6, 10.0, float("inf")  # am.h:30-31, ssb.h:27
    d.agc_attack = np.float32(agc_attack / out_sr)
    d.agc_decay = np.float32(agc_decay / out_sr)
    d.am_carrier_agc = int(carrier_agc)
    d.dc_block_rate = np.float32(100.0 / out_sr)  # radio am.h:34
    d.inv_deviation = 0.0
    d.audio_ntaps = 0
    d.ssb_phase_delta_re, d.ssb_phase_delta_im = 1.0, 0.0
    if mode == "WFM":
        d.inv_deviation = np.float32(1.0 / hz_to_rads(bandwidth / 2.0, out_sr))  # wfm.h:78, quadrature.h:19-26
        if low_pass:
            at = capi.design_low_pass(15000.0, 4000.0, out_sr)  # broadcast_fm.h:49
            keep.append(at)
            d.audio_ntaps = len(at)
            d.audio_taps = _fp(at)
    elif mode == "NFM":
        d.inv_deviation = np.float32(1.0 / hz_to_rads(bandwidth / 2.0, out_sr))  # fm.h:32
        if low_pass:
            at = capi.design_low_pass(bandwidth / 2.0, (bandwidth / 2.0) * 0.1, out_sr)  # fm.h:156
            keep.append(at)
            d.audio_ntaps = len(at)
            d.audio_taps = _fp(at)
    elif mode == "AM":
        at = capi.design_low_pass(bandwidth / 2.0, (bandwidth / 2.0) * 0.1, out_sr)  # demod/am.h:33
        keep.append(at)
        d.audio_ntaps = len(at)
        d.audio_taps = _fp(at)
    elif mode in ("USB", "LSB", "DSB"):
        tr = {"USB": bandwidth / 2.0, "LSB": -bandwidth / 2.0, "DSB": 0.0}[mode]  # ssb.h:106-117
        d.ssb_phase_delta_re, d.ssb_phase_delta_im = capi.design_phase_delta(tr, out_sr)
    return d, keep


def describe(desc):
    """Plan summary (used by tests to compare against the oracle / reference printout)."""
    predec = 1
    for i in range(desc.n_stages):
        predec *= desc.stage_decim[i]
    return dict(predec=predec, interp=desc.interp, decim=desc.decim, rtaps=desc.resamp_ntaps, chan_taps=desc.chan_ntaps,
                audio_taps=desc.audio_ntaps, stages=[(desc.stage_decim[i], desc.stage_ntaps[i]) for i in range(desc.n_stages)])


def af_desc(af_rate, audio_rate=48000.0, deemp_tau=50e-6, high_pass=False):
    """sdrpp_af_desc for the radio module's AF chain (radio_module.h:98-110): RationalResampler<stereo_t>(af_rate -> audio_rate),
    optional highPass(300, 100, audio_rate) FIR, optional Deemphasis(tau, audio_rate) (tau None/0 = off).
    Returns (desc, keepalive)."""
    a = capi.AfDesc()
    keep = []
    rs = capi.design_resampler(af_rate, audio_rate, plans().max_ratio)
    stages = plans().stages(rs["predec"]) if rs["mode"] in (0, 1) else []
    a.n_stages = len(stages)
    for i, (dec, taps) in enumerate(stages):
        t = np.ascontiguousarray(taps, dtype=np.float32)
        keep.append(t)
        a.stage_decim[i] = dec
        a.stage_ntaps[i] = len(t)
        a.stage_taps[i] = _fp(t)
    if rs["mode"] in (0, 2):
        rt = np.ascontiguousarray(rs["taps"], dtype=np.float32)
        keep.append(rt)
        a.interp, a.decim = rs["interp"], rs["decim"]
        a.resamp_ntaps = len(rt)
        a.resamp_taps = _fp(rt)
    else:
        a.interp, a.decim, a.resamp_ntaps = 1, 1, 0
    if high_pass:
        ht = capi.design_high_pass(300.0, 100.0, audio_rate)
        keep.append(ht)
        a.hpf_ntaps = len(ht)
        a.hpf_taps = _fp(ht)
    a.deemph_alpha = capi.design_deemphasis_alpha(deemp_tau, audio_rate) if deemp_tau else 0.0
    return a, keep
