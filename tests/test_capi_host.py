"""C-ABI surface and host-side logic that needs no GPU: the product library loads, exports every symbol the header
declares, its double-precision design maths equals the oracle's (and the reference's) bit for bit, and it refuses to
compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import support as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _real_capi():
    from sdrplusplus_amd import capi

    capi.DEFAULT_LIB = os.path.join(ROOT, "sdrplusplus_amd", "csrc", "libsdrpp_gpu.so")
    return capi


def test_library_exports_every_declared_symbol():
    capi = _real_capi()
    L = capi.load()
    header = open(os.path.join(ROOT, "include", "sdrpp_gpu.h")).read()
    declared = sorted(set(re.findall(r"\b(sdrpp_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(L, name), "libsdrpp_gpu.so does not export " + name
    assert sorted(capi.EXPORTED_SYMBOLS) == declared
    sz = C.c_int()
    assert L.sdrpp_abi_version(C.byref(sz)) == capi.ABI_VERSION and sz.value == C.sizeof(capi.VfoDesc)  # ctypes mirror == C struct layout


def test_no_cpu_fallback_without_device():
    import torch

    capi = _real_capi()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.SdrppError) as e:
        capi.Context(0, max_push=1024)
    assert e.value.code == -1  # SDRPP_ERR_NO_DEVICE


def test_design_maths_matches_oracle_bit_for_bit():
    capi = _real_capi()
    o = S.oracle()
    for args in [(75000.0, 7500.0, 250000.0), (15000.0, 4000.0, 250000.0), (5000.0, 500.0, 15000.0), (125000.0, 12500.0, 1250000.0)]:
        assert np.array_equal(capi.design_low_pass(*args), S.oracle_low_pass(*args))
    hp = capi.design_high_pass(300.0, 100.0, 48000.0)
    buf = np.zeros(len(hp) + 4, np.float32)
    n = o.orc_high_pass(300.0, 100.0, 48000.0, 0, S._fp(buf), len(buf))
    assert n == len(hp) and np.array_equal(buf[:n], hp)
    for kind in (0, 1, 2):
        for nz in (1024, 4096, 50000):
            assert np.array_equal(capi.design_fft_window(kind, nz), S.oracle_fft_window(kind, nz))
    for sr, N, rate in [(2.4e6, 4096, 20.0), (10e6, 65536, 20.0), (61.44e6, 1 << 20, 20.0), (1e6, 4096, 400.0)]:
        skip, nz = C.c_int(), C.c_int()
        o.orc_gen_reshape_params(sr, N, rate, C.byref(skip), C.byref(nz))
        assert capi.design_reshape_params(sr, N, rate) == (nz.value, skip.value)
    assert capi.design_reshape_params(10e6, 65536, 20.0) == (65536, 434464)
    for vo, vb, wb, n in [(0.0, 10e6, 10e6, 65536), (1.2e6, 2e6, 10e6, 65536), (-3e6, 0.5e6, 8e6, 4096)]:
        a, b = C.c_int(), C.c_int()
        o.orc_waterfall_view(vo, vb, wb, n, C.byref(a), C.byref(b))
        assert capi.design_waterfall_view(vo, vb, wb, n) == (a.value, b.value)


def test_vfo_descriptors_match_oracle_plans():
    capi = _real_capi()
    from sdrplusplus_amd import radio

    for sr, mode in [(2.4e6, "WFM"), (10e6, "WFM"), (61.44e6, "NFM"), (61.44e6, "AM"), (61.44e6, "USB"), (61.44e6, "DSB")]:
        if_rate, bw = radio.RADIO_DEFAULTS[mode]
        d, keep = radio.vfo_desc(sr, if_rate, bw, 123456.0, mode)
        info = S.oracle_rxvfo_info(S.OracleChain(sr, if_rate, bw, 123456.0, S.MODES[mode]))
        desc = radio.describe(d)
        assert desc["predec"] == info["predec"] and desc["chan_taps"] == info["chan_taps"]
        if info["rtaps"]:
            assert (desc["interp"], desc["decim"], desc["rtaps"]) == (info["interp"], info["decim"], info["rtaps"])
        else:
            assert d.interp == d.decim
        ch = S.OracleChain(sr, if_rate, bw, 123456.0, None)
        dr, di = C.c_float(), C.c_float()
        S.oracle().orc_rxvfo_phase_delta(ch.vfo, C.byref(dr), C.byref(di))
        assert (d.phase_delta_re, d.phase_delta_im) == (dr.value, di.value)
    p = radio.plans()
    assert p.max_ratio == 8192 and [dd for dd, _ in p.stages(32)] == [8, 2, 2] and [len(t) for _, t in p.stages(32)] == [44, 12, 69]


def test_emulator_is_never_the_default():
    capi = _real_capi()
    assert os.path.basename(capi.lib_path()) == "libsdrpp_gpu.so"
    src = open(os.path.join(ROOT, "sdrplusplus_amd", "capi.py")).read() + open(os.path.join(ROOT, "bench.py")).read() + open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "libsdrpp_gpu_emu" not in src
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sdrplusplus_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in text.replace("oracle/_ref", "").replace("test oracle", "") or f in ("__init__.py",), "product file %s references the oracle" % f


def test_binding_constants_match_header():
    """Buffers handed to sdrpp_timing_read are sized from these constants: they must track include/sdrpp_gpu.h."""
    import re
    from sdrplusplus_amd import capi

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sdrpp_gpu.h")).read()
    assert int(re.search(r"#define\s+SDRPP_NUM_KERNEL_FAMILIES\s+(\d+)", hdr).group(1)) == capi.NUM_KERNEL_FAMILIES
    assert int(re.search(r"#define\s+SDRPP_MAX_DECIM_STAGES\s+(\d+)", hdr).group(1)) == capi.MAX_DECIM_STAGES
    names = [capi.load().sdrpp_kernel_family_name(i).decode() for i in range(capi.NUM_KERNEL_FAMILIES)]
    assert len(set(names)) == capi.NUM_KERNEL_FAMILIES and "?" not in names


def test_timeline_tool_knows_every_role_of_the_tick_kernel():
    """tools/tick_trace.py labels a dump's records by role index: its table has to follow the library's (sdrpp_pipeline_role_name — itself pinned to the
    kernel's enum by a static_assert); it had fallen one role behind once."""
    lib = _real_capi().load()
    names = []
    while lib.sdrpp_pipeline_role_name(len(names)) is not None:
        names.append(lib.sdrpp_pipeline_role_name(len(names)).decode())
    src = open(os.path.join(ROOT, "tools", "tick_trace.py")).read()
    roles = eval(src[src.index("ROLES = ") + 8:src.index("]", src.index("ROLES = ")) + 1])

    def tool_name(n):  # (the tool abbreviates a few)
        if re.fullmatch(r"fft_s1[0-2]", n):
            return n
        n = n.replace("fft_", "")
        return {"tr": "transp"}.get(n, n).replace("zoom_", "zoom")

    assert len(names) > 40 and [tool_name(n) for n in names] == roles, [(a, b) for a, b in zip(names, roles) if tool_name(a) != b]
    buf_len = 128  # capi.Context.pipeline_stats: head + one count per role must fit its buffer
    assert 8 + len(names) <= buf_len
