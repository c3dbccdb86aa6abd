"""BASELINE.json configurations as concrete synthetic inputs + context set-up.

cfg 1: 2.4 MS/s int16 IQ WAV (file_source format), 1 VFO WFM, 4096-pt FFT at 20 lines/s
cfg 2: 10 MS/s, 65536-pt FFT + log-power waterfall only
cfg 3: cfg 2 + 32 VFO x WFM (the headline metric's configuration)
cfg 4: 61.44 MS/s, 128 VFO mixed NFM/AM/USB, 2^20-pt FFT
cfg 5: 8 x cfg 4 with seeds 0..7, one stream per GPU
Generators are numpy (float64 phase, rounded once to complex64) so CPU oracle and GPU see identical bits."""
import numpy as np

from . import capi, radio

CFG = {
    1: dict(sr=2.4e6, fft=4096, fft_rate=20.0, nvfo=1, spacing=0.0, first_offset=300e3),
    2: dict(sr=10e6, fft=65536, fft_rate=None, nvfo=0),
    3: dict(sr=10e6, fft=65536, fft_rate=None, nvfo=32, spacing=300e3),
    4: dict(sr=61.44e6, fft=1 << 20, fft_rate=None, nvfo=128, spacing=400e3),
}
CFG4_MODES = ("NFM", "AM", "USB")


def vfo_plan(cfg, nvfo=None):
    """[(mode, if_rate, bandwidth, vfo_centre_offset_hz, carrier_hz)] for a configuration."""
    c = CFG[cfg]
    n = c["nvfo"] if nvfo is None else nvfo
    out = []
    for k in range(n):
        if cfg == 1:
            mode, carrier = "WFM", c["first_offset"]
        elif cfg == 3:
            mode, carrier = "WFM", (k - (c["nvfo"] - 1) / 2.0) * c["spacing"]
        else:
            mode, carrier = CFG4_MODES[k % 3], (k - (c["nvfo"] - 1) / 2.0) * c["spacing"]
        if_rate, bw = radio.RADIO_DEFAULTS[mode]
        # USB/LSB: the GUI offset is the carrier, the RxVFO is tuned to the band centre (waterfall.cpp:1219-1240)
        centre = carrier + (bw / 2.0 if mode == "USB" else (-bw / 2.0 if mode == "LSB" else 0.0))
        out.append((mode, if_rate, bw, centre, carrier))
    return out


def tones_and_noise(n, sr, seed, start=0):
    """cfg 2 background: 8 complex tones (amplitudes 1 .. 1e-4, on and off bin centres) + complex AWGN sigma 1e-3."""
    rng = np.random.default_rng(seed)
    t = np.arange(start, start + n, dtype=np.float64)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 1e-3
    fr = np.array([0.0625, -0.125, 0.20001, -0.3123, 0.4101, -0.0417, 0.3333, -0.4499])
    amp = np.array([1.0, 0.3, 0.1, 0.03, 0.01, 3e-3, 1e-3, 1e-4])
    for f, a in zip(fr, amp):
        x += a * np.exp(2j * np.pi * f * t)
    return x


def synth(cfg, n, seed=1, nvfo=None, start=0):
    """complex64 IQ for a configuration (float64 maths, one rounding)."""
    c = CFG[cfg]
    sr = c["sr"]
    t = np.arange(start, start + n, dtype=np.float64) / sr
    if cfg == 1:
        rng = np.random.default_rng(seed)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.01
    else:
        x = tones_and_noise(n, sr, seed, start)
        if cfg in (3, 4):
            x *= 0.1  # leave headroom for the carriers
    for k, (mode, if_rate, bw, centre, carrier) in enumerate(vfo_plan(cfg, nvfo)):
        if mode == "WFM":
            tone = 1000.0 if cfg == 1 else 400.0 + 50.0 * k
            a = 0.5 if cfg == 1 else 0.05
            x += a * np.exp(1j * (2 * np.pi * carrier * t + (75e3 / tone) * np.sin(2 * np.pi * tone * t)))
        elif mode == "NFM":
            x += 0.02 * np.exp(1j * (2 * np.pi * carrier * t + (2500.0 / 1000.0) * np.sin(2 * np.pi * 1000.0 * t)))
        elif mode == "AM":
            x += 0.02 * (1.0 + 0.3 * np.cos(2 * np.pi * 1000.0 * t)) * np.exp(2j * np.pi * carrier * t)
        else:  # USB two-tone
            x += 0.01 * (np.exp(2j * np.pi * (carrier + 700.0) * t) + np.exp(2j * np.pi * (carrier + 1900.0) * t))
    return x.astype(np.complex64)


def to_int16_wav_samples(x):
    """cfg 1 on-disk format: interleaved int16 IQ, round(x * 32767 * 0.5) (file_source reads them back / 32768)."""
    v = np.empty(2 * len(x), dtype=np.float64)
    v[0::2] = x.real
    v[1::2] = x.imag
    return np.clip(np.rint(v * 32767.0 * 0.5), -32768, 32767).astype(np.int16)


def setup(ctx, cfg, dense_fft=True, data_width=1024, nvfo=None, window_kind=2, wf_min=-120.0, wf_max=0.0, fft=True, exact_ssb=False):
    """Configure a capi.Context for a BASELINE configuration (fft=False: VFO bank only).  Returns dict(vids, plan, nz, skip, fft).
    exact_ssb: the SSB / DSB / raw channels run the reference's float rotator recursion (sdrpp_vfo_desc.nco_mode = 2) — the setting in which
    EVERY channel matches the compiled reference inside the north-star tolerance at arbitrary offsets; FM / AM channels stay closed form."""
    c = CFG[cfg]
    sr, N = c["sr"], c["fft"]
    if dense_fft and c["fft_rate"] is None:
        nz, skip = N, 0  # every sample transformed (headline framing)
    else:
        nz, skip = capi.design_reshape_params(sr, N, c["fft_rate"] or 20.0)
    start, size = capi.design_waterfall_view(0.0, sr, sr, N)  # full view
    if fft:
        ctx.fft_configure(N, nz, skip, capi.design_fft_window(window_kind, nz))
        ctx.fft_set_view(start, size, data_width, wf_min, wf_max)
    vids = []
    plan = vfo_plan(cfg, nvfo)
    for mode, if_rate, bw, centre, _ in plan:
        d, keep = radio.vfo_desc(sr, if_rate, bw, centre, mode, nco_mode=2 if (exact_ssb and mode in ("USB", "LSB", "DSB", "RAW")) else 0)
        vids.append(ctx.vfo_add(d, keep))
    return dict(vids=vids, plan=plan, nz=nz, skip=skip, fft=N, sr=sr, view=(start, size, data_width, wf_min, wf_max))
