"""BASELINE.json configurations at FULL size on the real device: every VFO of cfg 3, sampled VFOs + the 2^20-pt line of
cfg 4 against the oracle, and size-independent properties at bench-sized pushes (determinism, push-cut invariance,
Parseval, tone placement)."""
import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def rms(a):
    return float(np.sqrt(np.mean(np.abs(a) ** 2)))


def test_cfg3_all_32_vfos_and_fft_vs_oracle():
    from sdrplusplus_amd import capi, workloads

    B, nblk = 50000, 4
    x = workloads.synth(3, B * nblk, seed=31)
    ctx = capi.Context(0, max_push=B)
    info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024)
    assert len(info["vids"]) == 32
    chains = [S.OracleChain(info["sr"], r, bw, c, S.MODES[m]) for m, r, bw, c, _ in info["plan"]]
    w = capi.design_fft_window(2, 65536)
    spec = S.OracleSpectrum(65536, 65536, 0, w)
    start, size, width, lo, hi = info["view"]
    worst = 0.0
    for b in range(nblk):
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        raw, zo, ix = ctx.fft_read()
        ol = spec.push(blk)
        assert raw.shape == ol.shape and np.array_equal(raw, ol)
        if len(ol):
            oz = np.stack([S.oracle_do_zoom(start, size, width, l) for l in ol])
            assert np.array_equal(zo, oz)
            assert np.array_equal(ix, np.stack([S.oracle_palette_index(z, lo, hi) for z in oz]))
        for vid, ch in zip(info["vids"], chains):
            oa = ch.process(blk)[1]
            ga = ctx.vfo_read(vid)
            assert ga.shape == oa.shape
            worst = max(worst, rms(ga - oa))
    assert worst < 1e-5, worst
    ctx.close()


def test_cfg4_128_vfos_sampled_and_1m_point_line_vs_oracle():
    from sdrplusplus_amd import capi, workloads

    B, nblk = 307200, 4
    x = workloads.synth(4, B * nblk, seed=41)
    ctx = capi.Context(0, max_push=B)
    info = workloads.setup(ctx, 4, dense_fft=True, data_width=1024)
    assert len(info["vids"]) == 128
    sample = [0, 1, 2, 63, 64, 65, 125, 126, 127]
    chains = {k: S.OracleChain(info["sr"], info["plan"][k][1], info["plan"][k][2], info["plan"][k][3], S.MODES[info["plan"][k][0]]) for k in sample}
    N = 1 << 20
    spec = S.OracleSpectrum(N, N, 0, capi.design_fft_window(2, N))
    nlines = 0
    for b in range(nblk):
        blk = x[b * B:(b + 1) * B]
        ctx.push(blk)
        raw, _, _ = ctx.fft_read()
        ol = spec.push(blk)
        assert raw.shape == ol.shape and np.array_equal(raw, ol)
        nlines += len(ol)
        for k in sample:
            mode = info["plan"][k][0]
            oa = chains[k].process(blk)[1]
            ga = ctx.vfo_read(info["vids"][k])
            assert ga.shape == oa.shape, (k, mode)
            # FM/AM are insensitive to the reference rotator's drift; the SSB product detector sees it
            tol = 1e-5 if mode in ("NFM", "AM") else 2e-3
            assert rms(ga - oa) <= tol * max(1.0, rms(oa)), (k, mode, b, rms(ga - oa), rms(oa))
        for vid in info["vids"]:
            n = ctx.vfo_out_count(vid)
            assert n in (75, 120, 250)  # 307200 / 4096, *4/5/2048, *5/6/1024
    assert nlines == 1
    ctx.close()


def test_bench_sized_push_properties():
    """4 Mi-sample pushes (what bench.py times): bit-identical lines and near-identical audio whether the stream is pushed in
    one piece or in four; repeated runs are bit-identical; Parseval holds; tones sit on the expected bins."""
    import torch
    from sdrplusplus_amd import capi, workloads

    n = 1 << 22
    x = workloads.synth(3, n, seed=77, nvfo=8)

    def run(cuts):
        ctx = capi.Context(0, max_push=max(cuts))
        info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=8)
        lines, audio = [], [[] for _ in info["vids"]]
        pos = 0
        for c in cuts:
            ctx.push(x[pos:pos + c])
            pos += c
            lines.append(ctx.fft_read(zoomed=False)[0])
            for i, vid in enumerate(info["vids"]):
                audio[i].append(ctx.vfo_read(vid))
        ctx.close()
        return np.concatenate(lines), [np.concatenate(a) for a in audio]

    l1, a1 = run([n])
    l2, a2 = run([n])
    l4, a4 = run([n // 4] * 4)
    assert l1.shape == (64, 65536)
    assert np.array_equal(l1, l2) and all(np.array_equal(p, q) for p, q in zip(a1, a2))  # deterministic
    assert np.array_equal(l1, l4)                                                        # frames do not care about push cuts
    for p, q in zip(a1, a4):
        assert p.shape == q.shape and np.max(np.abs(p - q)) < 2e-6
    # Parseval on the first frame: sum |X_k|^2 / N^2 == sum |x w|^2 / N
    w = capi.design_fft_window(2, 65536).astype(np.float64)
    lhs = np.sum(10.0 ** (l1[0].astype(np.float64) / 10.0))
    rhs = np.sum(np.abs(x[:65536].astype(np.complex128) * w) ** 2) / 65536.0
    assert abs(lhs / rhs - 1.0) < 1e-4
    # strongest background tone: 0.1 * 1.0 at +0.0625 fs -> bin N/2 + N/16 (on a bin centre), level 20log10(0.1) - 8.98 dB
    k = 32768 + 4096
    assert abs(int(np.argmax(l1[0])) - k) <= 0 and abs(l1[0][k] - (-20.0 - 8.98)) < 0.05
    assert torch.cuda.is_available()
