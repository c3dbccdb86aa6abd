"""MI355X-native implementation of SDR++'s streaming-DSP hot path (FFT/waterfall pipeline + per-VFO channeliser).

Product layout:
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (include/sdrpp_gpu.h) -> csrc/libsdrpp_gpu.so
  host/      C++ mirror of the reference's dsp::block / dsp::stream<T> plugin surface wrapping the C-ABI
  capi.py    ctypes binding of the C-ABI (used by tests and bench.py; torch only supplies device memory / streams)
  radio.py   host-side parameterisation of RxVFO + radio-module demodulators (what VFOManager / RadioModule compute)
  workloads.py  BASELINE.json configurations as synthetic inputs
There is no CPU fallback anywhere in this package.
"""
from . import capi  # noqa: F401
