import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libsdrpp_gpu_emu.so")
REAL_LIB = os.path.join(ROOT, "sdrplusplus_amd", "csrc", "libsdrpp_gpu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(*args):
    from support import locked_make  # (tests/support.py: one lock for every build a test may trigger)

    locked_make(*args)


def _ensure_built():
    """CPU-side artefacts: oracle (+ compiled reference where /root/reference exists), product library (cross-compiled),
    and the test-only emulator build of the same kernel sources."""
    # always through make: a no-op when the library is newer than its sources (on the GPU box hipcc is present as well).  The product library
    # first: oracle/_ref links one of its reference-side test programs against it.
    _make("-C", os.path.join(ROOT, "sdrplusplus_amd", "csrc"), "-s", "all")
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")) or os.path.isdir("/root/reference"):
        _make("-C", os.path.join(ROOT, "oracle"), "-s", "all")


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()


BACKENDS = [pytest.param("emu", id="emu"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
    """'gpu' = the product library on a real device (parity tests proper).  'emu' = the SAME kernel sources compiled
    against the fiber emulator in tests/emu, run on the CPU: a logic check of indexing/streaming state, never a product
    path."""
    from sdrplusplus_amd import capi

    if request.param == "emu":
        _make("-C", os.path.join(ROOT, "tests", "emu"), "-s")
        capi.DEFAULT_LIB = EMU_LIB  # test-only: the product binding itself has no override
    else:
        capi.DEFAULT_LIB = REAL_LIB
    yield request.param
    capi.DEFAULT_LIB = REAL_LIB
