// FFT -> log-power -> waterfall-line kernels for gfx950 (wave64, LDS-staged radix-2^4 register rounds).
//
// What the reference does per frame on one CPU thread (core/src/signal_path/iq_frontend.cpp:248-267):
//   fftIn[0:nz] = x * window (volk_32fc_32f_multiply_32fc), fftIn[nz:N] = 0; X = FFT_N(fftIn) (fftwf_execute);
//   dB[k] = 10*log10(|X[k]|^2 / N^2) (volk_32fc_s32f_power_spectrum_32f) written into the waterfall's raw line;
// then WaterFall::pushFFT max-decimates the line to `dataWidth` pixels and maps each to a palette index
// (core/src/gui/widgets/waterfall.cpp:65-90, 889-906).
//
// FFT algorithm (identical, operation for operation, to the test oracle so palette indices can be compared bit-exactly):
//   * sub-FFT of length L <= 4096: radix-2 decimation in time on the bit-reversed sequence; every butterfly is the
//     6-FMA form  X0 = u + w*v (two nested fmaf per component),  X1 = fmaf(2, u, -X0);  twiddles come from an exactly
//     symmetric float table tw(e, L) (host generated, sdrpp_host::twiddle).  Groups of four consecutive stages run in
//     registers (16 points per work-item); between groups the points are exchanged through LDS.  The grouping does not
//     change any rounding: each butterfly sees the same operands whatever the schedule.
//   * N > 4096: four-step N = N1 x N2 (N1 = 2^floor(m/2) up to N = 65536; N1 = N / 4096 above — the largest row transform one
//     workgroup holds, so that the column pass keeps >= 128-byte row segments: with the even split a 2^20-point pass 1 read 32-byte
//     segments at an 8 KB stride and cost 5x the 65536-point transform per sample): pass 1 = N1-point column FFTs (stride N2) + multiplication by
//     tw(n2*k1, N) (re = fmaf(a.re, w.re, -(a.im*w.im)), im = fmaf(a.re, w.im, a.im*w.re)) into a scratch matrix
//     A[k1][n2]; pass 2 = N2-point row FFTs, X[k1 + N1*k2], fused with the dB conversion.
// All floating-point contraction is disabled for this translation unit (-ffp-contract=off); FMAs are explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <sdrpp_gfx950.h>

namespace sdrpp_k {

// Workgroup coordinates handed to a kernel BODY.  Every kernel of the per-block path is written as a body function of (bid, gdim,
// ...) plus a thin __global__ wrapper that passes blockIdx / gridDim: the same body then also runs as a ROLE inside the one-launch-
// per-block kernel (tick_kernels.h), where a workgroup's role and coordinates come from a table instead of the launch grid.
struct KIdx { int x, y; };
template <class T> __device__ __forceinline__ KIdx kidx(const T& v) { return KIdx{ (int)v.x, (int)v.y }; }

// Two-source sample accessor: index i is relative to the first sample of the current push; i < 0 reaches into the
// history kept from earlier pushes (hist holds the most recent hist_len samples, oldest first).
struct IqSrc {
    const float2* cur;
    const float2* hist;
    int hist_len;
    long long n_cur;  // valid samples in `cur`
};
__device__ __forceinline__ float2 iq_load(const IqSrc& s, long long i) {
    return (i >= 0) ? s.cur[i] : s.hist[s.hist_len + i];
}
// Tile loaders may run past the last sample a (partial) tile really needs: those reads return zero instead of touching
// memory beyond the caller's buffer.
__device__ __forceinline__ float2 iq_load_clamped(const IqSrc& s, long long i) {
    if (i >= s.n_cur) { return make_float2(0.0f, 0.0f); }
    return iq_load(s, i);
}

// Branch-free form for loops that fetch several samples per work-item (every load unconditional, the address clamped into the buffers,
// the value selected afterwards: the loads of a loop are all in flight before the first wait).  `valid` false -> zero.
__device__ __forceinline__ float2 iq_load_nb(const IqSrc& s, long long i, bool valid) {
    const bool cur = i >= 0;
    valid = valid && i < s.n_cur;
    long long ic = cur ? i : (long long)s.hist_len + i;
    ic = (valid && ic >= 0) ? ic : 0;
    const float2* p = (cur && valid) ? s.cur : s.hist;
    const float2 v = global_load_f32x2(p, ic);
    return valid ? v : make_float2(0.0f, 0.0f);
}

// ---- butterflies -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bfly(float2& u, float2& v, const float2 w) {
    const float x0r = fmaf(-w.y, v.y, fmaf(w.x, v.x, u.x));
    const float x0i = fmaf(w.y, v.x, fmaf(w.x, v.y, u.y));
    v.x = fmaf(2.0f, u.x, -x0r);
    v.y = fmaf(2.0f, u.y, -x0i);
    u.x = x0r;
    u.y = x0i;
}
// w == tw(0) = (1, -0): X0 degenerates to u + v (same rounding as the general form)
__device__ __forceinline__ void bfly_one(float2& u, float2& v) {
    const float x0r = u.x + v.x;
    const float x0i = u.y + v.y;
    v.x = fmaf(2.0f, u.x, -x0r);
    v.y = fmaf(2.0f, u.y, -x0i);
    u.x = x0r;
    u.y = x0i;
}
// w == tw(M/4) = (-0, -1) = -j
__device__ __forceinline__ void bfly_mj(float2& u, float2& v) {
    const float x0r = u.x + v.y;
    const float x0i = u.y - v.x;
    v.x = fmaf(2.0f, u.x, -x0r);
    v.y = fmaf(2.0f, u.y, -x0i);
    u.x = x0r;
    u.y = x0i;
}

// ---- one group of G <= 4 consecutive radix-2 stages on 16 register-resident points ----------------------------------------
// Stages S0+1 .. S0+G of an L = 2^LG point sub-FFT.  A work-item with index t (0 .. L/16-1) owns NG = 16 >> G groups of
// 2^G points; group i is q = t + i*(L/16) -> lo = q mod 2^S0, hi = q >> S0, and point j of the group sits at position
// p = (hi << (S0+G)) | (j << S0) | lo.  `tw` holds tw(e, L) for e < L/2 (LDS).
template <int LG, int S0, int G>
struct FftRound {
    static constexpr int NG = 16 >> G;
    static constexpr int TPF = (1 << LG) / 16;
    static constexpr int GS = 1 << G;

    __device__ static __forceinline__ int pos(int t, int i, int j) {
        const int q = t + i * TPF;
        const int lo = q & ((1 << S0) - 1);
        const int hi = q >> S0;
        return (hi << (S0 + G)) | (j << S0) | lo;
    }

    __device__ static __forceinline__ void compute(float2 (&r)[16], int t, const float2* tw) {
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const int q = t + i * TPF;
            const int lo = q & ((1 << S0) - 1);
#pragma unroll
            for (int u = 1; u <= G; u++) {
                const int half = 1 << (u - 1);
                const int sh = LG - S0 - u;  // table index = k << sh, k = (kj << S0) | lo
                float2 wst[8];
#pragma unroll
                for (int kj = 0; kj < half; kj++) {
                    float2 w;
                    bool one = false, mj = false;
                    if (half >= 2 && kj >= half / 2) {
                        // tw(k + M/4) = -j * tw(k), exact by construction of the table
                        const float2 wp = wst[kj - half / 2];
                        w = make_float2(wp.y, -wp.x);
                        if (S0 == 0 && kj == half / 2) { mj = true; }
                    }
                    else {
                        if (S0 == 0 && kj == 0) {
                            one = true;
                            w = make_float2(1.0f, -0.0f);
                        }
                        else {
                            w = tw[(((kj << S0) | lo)) << sh];
                        }
                    }
                    wst[kj] = w;
#pragma unroll
                    for (int b = 0; b < GS / (2 * half); b++) {
                        const int j0 = b * 2 * half + kj;
                        const int j1 = j0 + half;
                        if (one) { bfly_one(r[i * GS + j0], r[i * GS + j1]); }
                        else if (mj) { bfly_mj(r[i * GS + j0], r[i * GS + j1]); }
                        else { bfly(r[i * GS + j0], r[i * GS + j1], w); }
                    }
                }
            }
        }
    }
};

// Number of stages in round `R` (rounds start at stage 0, 4, 8) of an LG-stage sub-FFT.
constexpr int fft_round_stages(int LG, int S0) { return (LG - S0) >= 4 ? 4 : (LG - S0); }

// Runs every round after the first on a [positions] LDS image.  `IDX` maps a position to an LDS element index (padding /
// column interleave); after the call r[] holds the points of the LAST round: point (i, j) is output bin
// FftRound<LG, S0_last, G_last>::pos(t, i, j).
template <int LG, int S0, class IDX>
__device__ __forceinline__ void fft_rounds_after_first(float2 (&r)[16], int t, const float2* tw, float2* data, const IDX& idx) {
    constexpr int Gprev = fft_round_stages(LG, S0 - 4);
    constexpr int G = fft_round_stages(LG, S0);
    using Prev = FftRound<LG, S0 - 4, Gprev>;
    using Cur = FftRound<LG, S0, G>;
    __syncthreads();  // previous readers of `data` are done
#pragma unroll
    for (int i = 0; i < Prev::NG; i++) {
#pragma unroll
        for (int j = 0; j < Prev::GS; j++) { data[idx(Prev::pos(t, i, j))] = r[i * Prev::GS + j]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < Cur::NG; i++) {
#pragma unroll
        for (int j = 0; j < Cur::GS; j++) { r[i * Cur::GS + j] = data[idx(Cur::pos(t, i, j))]; }
    }
    Cur::compute(r, t, tw);
    if constexpr (S0 + G < LG) { fft_rounds_after_first<LG, S0 + 4, IDX>(r, t, tw, data, idx); }
}

// Last round's geometry for an LG-stage sub-FFT.
template <int LG>
struct FftLast {
    static constexpr int S0 = (LG <= 4) ? 0 : ((LG <= 8) ? 4 : 8);
    static constexpr int G = LG - S0;
    using Round = FftRound<LG, S0, G>;
};

// ---- log-power (VOLK power_spectrum generic; log2 per the shared specification) -------------------------------------------
__device__ __forceinline__ float spec_log2_poly(float f, float e) {
    float p = -0.107497893f;
    p = fmaf(p, f, 0.184760332f);
    p = fmaf(p, f, -0.191388384f);
    p = fmaf(p, f, 0.204857647f);
    p = fmaf(p, f, -0.239608124f);
    p = fmaf(p, f, 0.288552552f);
    p = fmaf(p, f, -0.360696554f);
    p = fmaf(p, f, 0.480898529f);
    p = fmaf(p, f, -0.721347332f);
    p = fmaf(p, f, 1.44269502f);
    return fmaf(f, p, e);
}
__device__ __forceinline__ float spec_log2f_non_ieee(float x) {
    // log2 of a non-negative float: subnormals rescaled by 2^23; mantissa folded into [sqrt(1/2), sqrt(2)); degree-9
    // polynomial in f = m - 1 evaluated with fmaf (Horner); result = fmaf(f, P, e).  An infinite result becomes +-127
    // (VOLK's log2f_non_ieee).
    if (x != x) { return x; }
    if (x < 0.0f) { return __uint_as_float(0x7fc00000u); }
    if (x == 0.0f) { return -127.0f; }
    unsigned u = __float_as_uint(x);
    if (u == 0x7f800000u) { return 127.0f; }
    int e = 0;
    if (u < 0x00800000u) {
        u = __float_as_uint(x * 8388608.0f);
        e = -23;
    }
    e += (int)(u >> 23) - 127;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m >= 1.41421354f) {
        m = m * 0.5f;
        e += 1;
    }
    return spec_log2_poly(m - 1.0f, (float)e);
}
// The same function for a POSITIVE NORMAL x (bit pattern in [0x00800000, 0x7f800000)) without a compare or a select: with m in [1, 2) the
// test m >= 1.41421354f (0x3fb504f3) is a test of the mantissa field, halving m is exact (the exponent field drops from 127 to 126) and both
// are integer arithmetic: adj = (mantissa + (0x800000 - 0x3504f3)) >> 23.  Bit for bit the value spec_log2f_non_ieee returns.
// `abnormal` collects (max) the distance of the bit pattern from that range: >= 0x7f000000 means zero / subnormal / inf / NaN / negative,
// and the caller then evaluates the general function instead (once for all its values: it practically never happens).
__device__ __forceinline__ float spec_log2f_normal(float x, unsigned& abnormal) {
    const unsigned u = __float_as_uint(x);
    const unsigned d = u - 0x00800000u;
    abnormal = d > abnormal ? d : abnormal;
    const unsigned mant = u & 0x007fffffu;
    const unsigned adj = (mant + 0x004afb0du) >> 23;
    const float m = __uint_as_float((mant | 0x3f800000u) - (adj << 23));
    const int e = (int)(u >> 23) - 127 + (int)adj;
    return spec_log2_poly(m - 1.0f, (float)e);
}
#define SDRPP_LOG2_ABNORMAL 0x7f000000u
__device__ __forceinline__ float power_of(const float2 X, const float inv_norm) {
    const float re = X.x * inv_norm;
    const float im = X.y * inv_norm;
    return (re * re) + (im * im);
}
__device__ __forceinline__ float power_db(const float2 X, const float inv_norm) {
    return 3.01029995663981209120f * spec_log2f_non_ieee(power_of(X, inv_norm));
}
__device__ __forceinline__ float power_db_normal(const float2 X, const float inv_norm, unsigned& abnormal) {
    return 3.01029995663981209120f * spec_log2f_normal(power_of(X, inv_norm), abnormal);
}

struct FrameGeom {
    long long first_start;  // push-relative index of sample 0 of frame 0 of this launch (may be negative: history)
    int stride;             // nz + skip
    int nz;                 // windowed samples per frame; the rest of the N inputs is zero
    int nframes;
};

// (branch-free: the 16 loads of a work-item are in flight together; beyond the frame's nz samples the FFT input is zero)
__device__ __forceinline__ float2 load_windowed(const IqSrc& src, const FrameGeom& g, const float* __restrict__ window, int frame, int i) {
    const bool in = i < g.nz;
    const float2 x = iq_load_nb(src, g.first_start + (long long)frame * g.stride + i, in);
    const float w = window[in ? i : 0];
    return in ? make_float2(x.x * w, x.y * w) : make_float2(0.0f, 0.0f);
}

// ---- N <= 4096: whole transform in one workgroup ------------------------------------------------------------------------------
struct IdxPad16 {
    __device__ __forceinline__ int operator()(int p) const { return p + (p >> 4); }
};

template <int LG, int FPW>
struct FftSingleLds {
    static constexpr int L = 1 << LG;
    static constexpr int PITCH = L + L / 16;
    static constexpr int TW = L / 2, DATA = FPW * PITCH;  // float2 elements: twiddles, then the frames' images
};
template <int LG, int FPW>
__device__ __forceinline__ void fft_single_body(const KIdx bid, float2* tw, float2* data, const IqSrc& src, const FrameGeom& g, const float* __restrict__ window,
                                                const float2* __restrict__ tw_g, float* __restrict__ out_db) {
    constexpr int L = 1 << LG;
    constexpr int TPF = L / 16;
    constexpr int PITCH = L + L / 16;
    const int t = threadIdx.x % TPF;
    const int fl = threadIdx.x / TPF;
    const int frame = bid.x * FPW + fl;
    const bool live = frame < g.nframes;
    for (int e = threadIdx.x; e < L / 2; e += TPF * FPW) { tw[e] = tw_g[e]; }
    float2 r[16];
    using R0 = FftRound<LG, 0, 4>;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int p = R0::pos(t, 0, j);
        const int n = (int)(__brev((unsigned)p) >> (32 - LG));
        r[j] = live ? load_windowed(src, g, window, frame, n) : make_float2(0.0f, 0.0f);
    }
    __syncthreads();  // twiddle table visible
    R0::compute(r, t, tw);
    float2* mydata = &data[fl * PITCH];
    fft_rounds_after_first<LG, 4, IdxPad16>(r, t, tw, mydata, IdxPad16());
    using RL = typename FftLast<LG>::Round;
    if (live) {
        const float inv = 1.0f / (float)L;
        float* dst = out_db + (size_t)frame * L;
        unsigned abn = 0;
#pragma unroll
        for (int i = 0; i < RL::NG; i++) {
#pragma unroll
            for (int j = 0; j < RL::GS; j++) { dst[RL::pos(t, i, j)] = power_db_normal(r[i * RL::GS + j], inv, abn); }
        }
        if (abn >= SDRPP_LOG2_ABNORMAL) {  // a zero / subnormal / non-finite power among this work-item's 16: the general function, all again
#pragma unroll
            for (int i = 0; i < RL::NG; i++) {
#pragma unroll
                for (int j = 0; j < RL::GS; j++) { dst[RL::pos(t, i, j)] = power_db(r[i * RL::GS + j], inv); }
            }
        }
    }
}
template <int LG, int FPW>
__global__ __launch_bounds__(((1 << LG) / 16) * FPW) void fft_single_kernel(IqSrc src, FrameGeom g, const float* __restrict__ window,
                                                                           const float2* __restrict__ tw_g, float* __restrict__ out_db) {
    __shared__ float2 tw[FftSingleLds<LG, FPW>::TW];
    __shared__ float2 data[FftSingleLds<LG, FPW>::DATA];
    fft_single_body<LG, FPW>(kidx(blockIdx), tw, data, src, g, window, tw_g, out_db);
}

// ---- N > 4096, pass 1: column FFTs + inter-pass twiddle ---------------------------------------------------------------------------
struct IdxCols {
    int ncols, c;
    __device__ __forceinline__ int operator()(int p) const { return p * ncols + c; }
};

// block = (N1/16) * C work-items, column index fastest; a workgroup walks the tiles bid.x, bid.x + step, ... of the ntiles = nframes * (N2 / C)
// tiles (frame-major) and keeps the NEXT tile's 16 IQ samples per work-item in flight while it transforms the current one (round 3b: one
// tile per workgroup left the memory system idle during every workgroup's compute phase and the matrix of phases — load, 8 stages, LDS
// exchange, twiddles, store — in lock-step on all CUs: 0.31-0.40 of the HBM rate with ~1 050 vector instructions per work-item being only
// a quarter of the time).  With step a multiple of the tiles per frame a workgroup stays on its columns: window values are loaded once.
// scratch layout: A[frame][k1][n2] (n2 contiguous); tw_n2k1 has the same [k1][n2] layout and holds tw(n2*k1, N).
template <int LG1, int C>
__device__ __forceinline__ void fft_pass1_body(const KIdx bid, float2* tw, float2* data, const IqSrc& src, const FrameGeom& g, const float* __restrict__ window,
                                               const float2* __restrict__ tw1_g, const float2* __restrict__ tw_n2k1, float2* __restrict__ scratch, int lg2, int ntiles, int step) {
    constexpr int L1 = 1 << LG1;
    constexpr int TPF = L1 / 16;
    const int N2 = 1 << lg2;
    const int tiles = N2 / C;
    const int c_ = threadIdx.x % C;
    const int t_ = threadIdx.x / C;
    for (int e = threadIdx.x; e < L1 / 2; e += TPF * C) { tw[e] = global_load_f32x2(tw1_g, e); }
    using R0 = FftRound<LG1, 0, 4>;
    using RL = typename FftLast<LG1>::Round;
    // raw samples of a tile (branch-free: the 16 loads of a work-item are in flight together; beyond the frame's nz samples the FFT input is zero).
    // A frame that lies inside the current push and has no zero padding — every frame of a dense framing but the few that begin in the
    // history — takes the lean path: uniform base pointer + one 32-bit offset per load, no bounds to test (the general loader spends ~25
    // vector instructions per load on 64-bit compares and selects: more than the transform itself).
    const bool dense = g.nz == (L1 << lg2);
    auto fetch = [&](float2 (&x)[16], int tile) {
        const int t = opaque(t_), c = opaque(c_);
        const int frame = tile / tiles, n2 = (tile % tiles) * C + c;
        const long long s0 = g.first_start + (long long)frame * g.stride;
        if (dense && s0 >= 0 && s0 + ((long long)L1 << lg2) <= src.n_cur) {  // (uniform)
            const float2* base = src.cur + s0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int n1 = (int)(__brev((unsigned)R0::pos(t, 0, j)) >> (32 - LG1));
                x[j] = global_load_f32x2_boff(base, (unsigned)((n1 << lg2) + n2) * 8u);
            }
        }
        else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int n1 = (int)(__brev((unsigned)R0::pos(t, 0, j)) >> (32 - LG1));
                const int i = (n1 << lg2) + n2;
                x[j] = iq_load_nb(src, s0 + i, i < g.nz);
            }
        }
    };
    float wv[16];
    int wv_n2 = -1;
    auto transform = [&](float2 (&r)[16], int tile) {
        const int t = opaque(t_), c = opaque(c_);  // (per-tile copies the optimiser cannot see through: it would otherwise hoist every LDS / global
                                                   // address of the transform out of the tile loop and keep ~60 more registers alive across it)
        const int frame = tile / tiles, n2 = (tile % tiles) * C + c;
        if (n2 != wv_n2) {  // (uniform: all work-items of a workgroup change columns together)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int n1 = (int)(__brev((unsigned)R0::pos(t, 0, j)) >> (32 - LG1));
                const int i = (n1 << lg2) + n2;
                wv[j] = global_load_f32(window, i < g.nz ? i : 0);
            }
            wv_n2 = n2;
        }
        if (dense) {  // (uniform) every input of the transform is a windowed sample
#pragma unroll
            for (int j = 0; j < 16; j++) { r[j] = make_float2(r[j].x * wv[j], r[j].y * wv[j]); }
        }
        else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int n1 = (int)(__brev((unsigned)R0::pos(t, 0, j)) >> (32 - LG1));
                const bool in = ((n1 << lg2) + n2) < g.nz;
                r[j] = in ? make_float2(r[j].x * wv[j], r[j].y * wv[j]) : make_float2(0.0f, 0.0f);
            }
        }
        R0::compute(r, t, tw);
        IdxCols idx{ C, c };
        fft_rounds_after_first<LG1, 4, IdxCols>(r, t, tw, data, idx);
        float2* dst = scratch + ((size_t)frame << (LG1 + lg2));  // (uniform; a frame's scratch is at most 8 MiB: 32-bit byte offsets)
#pragma unroll
        for (int i = 0; i < RL::NG; i++) {
#pragma unroll
            for (int j = 0; j < RL::GS; j++) {
                const int k1 = RL::pos(t, i, j);
                const unsigned o = (unsigned)((k1 << lg2) + n2) * 8u;
                const float2 a = r[i * RL::GS + j];
                const float2 w = global_load_f32x2_boff(tw_n2k1, o);
                const float pp = a.y * w.y;
                const float qq = a.y * w.x;
                global_store_f32x2_boff(dst, o, make_float2(fmaf(a.x, w.x, -pp), fmaf(a.x, w.y, qq)));
            }
        }
    };
    float2 r[16], rn[16];
    int tile = bid.x;
    if (tile >= ntiles) { return; }  // (no barrier has been passed yet)
    fetch(r, tile);
    __syncthreads();  // twiddle table visible
#ifdef SDRPP_FFT_NO_PREFETCH  // (diagnostic build: the tile walk without the loads in flight)
    while (true) {
        transform(r, tile);
        tile += step;
        if (tile >= ntiles) { break; }
        fetch(r, tile);
    }
#else
    while (true) {
        const int next = tile + step;
        const bool more = next < ntiles;
        if (more) { fetch(rn, next); }  // in flight during the transform below
        transform(r, tile);
        if (!more) { break; }
#pragma unroll
        for (int j = 0; j < 16; j++) { r[j] = rn[j]; }
        tile = next;
    }
#endif
}
template <int LG1, int C>
__global__ __launch_bounds__(((1 << LG1) / 16) * C, (((1 << LG1) / 16) * C) <= 256 ? 4 : 2) void fft_pass1_kernel(IqSrc src, FrameGeom g, const float* __restrict__ window,
                                                                         const float2* __restrict__ tw1_g, const float2* __restrict__ tw_n2k1,
                                                                         float2* __restrict__ scratch, int lg2, int ntiles) {
    __shared__ float2 tw[(1 << LG1) / 2];
    __shared__ float2 data[(1 << LG1) * C];
    fft_pass1_body<LG1, C>(kidx(blockIdx), tw, data, src, g, window, tw1_g, tw_n2k1, scratch, lg2, ntiles, (int)gridDim.x);
}

// ---- N > 4096, pass 2: row FFTs + dB, output bin k = k1 + N1*k2 ------------------------------------------------------------------
struct IdxRowPad {
    int base;
    __device__ __forceinline__ int operator()(int p) const { return base + p + (p >> 4); }
};

// block = (N2/16) * R work-items, t fastest (coalesced row reads); a workgroup walks the tiles bid.x, bid.x + step, ... of the
// ntiles = nframes * (N1 / R) row tiles and keeps the next tile's rows in flight while it transforms the current one (see fft_pass1_body).
template <int LG2, int R>
__device__ __forceinline__ void fft_pass2_body(const KIdx bid, float2* tw, float2* data, const float2* __restrict__ scratch, const float2* __restrict__ tw2_g,
                                               float* __restrict__ out_db, int lg1, int ntiles, float* __restrict__ grp_max, int step) {
    constexpr int L2 = 1 << LG2;
    constexpr int TPF = L2 / 16;
    constexpr int PITCH = L2 + L2 / 16;
    const int N1 = 1 << lg1;
    const int tiles = N1 / R;
    const int t_ = threadIdx.x % TPF;
    const int row_ = threadIdx.x / TPF;
    for (int e = threadIdx.x; e < L2 / 2; e += TPF * R) { tw[e] = global_load_f32x2(tw2_g, e); }
    using R0 = FftRound<LG2, 0, 4>;
    using RL = typename FftLast<LG2>::Round;
    auto fetch = [&](float2 (&x)[16], int tile) {
        const int t = opaque(t_), row = opaque(row_);
        const int frame = tile / tiles, r0 = (tile % tiles) * R;
        const float2* srcrow = scratch + ((size_t)frame << (LG2 + lg1)) + ((size_t)(r0 + row) << LG2);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int n = (int)(__brev((unsigned)R0::pos(t, 0, j)) >> (32 - LG2));
            x[j] = global_load_f32x2(srcrow, n);  // (explicit GLOBAL: as a tick role the pointer comes out of a table in memory and a plain dereference is a FLAT
                                                  // load, which the LDS wait counter counts too — every LDS wait of the transform then waited for the NEXT tile's rows)
        }
    };
    const float inv = 1.0f / (float)((size_t)1 << (LG2 + lg1));
    auto transform = [&](float2 (&r)[16], int tile) {
        const int t = opaque(t_), row = opaque(row_), tid = opaque((int)threadIdx.x);  // (see fft_pass1_body)
        const int frame = tile / tiles, r0 = (tile % tiles) * R;
        R0::compute(r, t, tw);
        // (fft_rounds_after_first opens with a barrier: the previous tile's dB values have been read out of `data` before it is rewritten)
        IdxRowPad idx{ row * PITCH };
        fft_rounds_after_first<LG2, 4, IdxRowPad>(r, t, tw, data, idx);
        // dB values are transposed through LDS so that consecutive lanes write consecutive k1 (k = k1 + N1*k2).
        __syncthreads();
        float* tile_db = reinterpret_cast<float*>(data);  // [k2][R + 1]
        unsigned abn = 0;
#pragma unroll
        for (int i = 0; i < RL::NG; i++) {
#pragma unroll
            for (int j = 0; j < RL::GS; j++) {
                const int k2 = RL::pos(t, i, j);
                tile_db[k2 * (R + 1) + row] = power_db_normal(r[i * RL::GS + j], inv, abn);
            }
        }
        if (abn >= SDRPP_LOG2_ABNORMAL) {  // a zero / subnormal / non-finite power among this work-item's 16: the general function, all again
#pragma unroll
            for (int i = 0; i < RL::NG; i++) {
#pragma unroll
                for (int j = 0; j < RL::GS; j++) {
                    const int k2 = RL::pos(t, i, j);
                    tile_db[k2 * (R + 1) + row] = power_db(r[i * RL::GS + j], inv);
                }
            }
        }
        __syncthreads();
        float* dst = out_db + ((size_t)frame << (LG2 + lg1)) + r0;  // (uniform; a line is at most 4 MiB: 32-bit byte offsets)
        {   // element e = tid + it * (TPF * R), it = 0 .. 15 -> k2 = e / R = tid / R + it * TPF, rr = e % R = tid % R: every step adds constants
            const int k2_0 = tid / R, rr = tid % R;
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const int k2 = k2_0 + it * TPF;
                global_store_f32_boff(dst, (unsigned)((k2 << lg1) + rr) * 4u, tile_db[k2 * (R + 1) + rr]);
            }
        }
        // doZoom's maximum over the R consecutive bins k1 = r0 .. r0 + R - 1 of every k2, while the tile is in LDS: the zoom kernel then
        // reads one value per aligned group of R bins instead of R (waterfall.cpp:65-90 takes a maximum, which does not care how it is split)
        if (grp_max) {
            float* gdst = grp_max + (((size_t)frame << (LG2 + lg1)) + r0) / R;
            static_assert(TPF * R == 256 && L2 * R == 4096, "256 work-items, 4096 values per tile");
#pragma unroll
            for (int k2 = tid; k2 < L2; k2 += 256) {
                float m = __uint_as_float(0xff800000u);
#pragma unroll
                for (int rr = 0; rr < R; rr++) {
                    const float v = tile_db[k2 * (R + 1) + rr];
                    if (v > m) { m = v; }
                }
                global_store_f32_boff(gdst, (unsigned)((k2 << lg1) / R) * 4u, m);
            }
        }
    };
    float2 r[16], rn[16];
    int tile = bid.x;
    if (tile >= ntiles) { return; }  // (no barrier has been passed yet)
    fetch(r, tile);
    __syncthreads();  // twiddle table visible
#ifdef SDRPP_FFT_NO_PREFETCH  // (diagnostic build: the tile walk without the loads in flight)
    while (true) {
        transform(r, tile);
        tile += step;
        if (tile >= ntiles) { break; }
        fetch(r, tile);
    }
#else
    while (true) {
        const int next = tile + step;
        const bool more = next < ntiles;
        if (more) { fetch(rn, next); }  // in flight during the transform below
        transform(r, tile);
        if (!more) { break; }
#pragma unroll
        for (int j = 0; j < 16; j++) { r[j] = rn[j]; }
        tile = next;
    }
#endif
}
template <int LG2, int R>
__global__ __launch_bounds__(((1 << LG2) / 16) * R, 4) void fft_pass2_kernel(const float2* __restrict__ scratch, const float2* __restrict__ tw2_g,
                                                                         float* __restrict__ out_db, int lg1, int ntiles, float* __restrict__ grp_max) {
    __shared__ float2 tw[(1 << LG2) / 2];
    __shared__ float2 data[R * ((1 << LG2) + (1 << LG2) / 16)];
    fft_pass2_body<LG2, R>(kidx(blockIdx), tw, data, scratch, tw2_g, out_db, lg1, ntiles, grp_max, (int)gridDim.x);
}

// ---- N > 65536, pass 2: ONE 4096-point row per workgroup ------------------------------------------------------------------------------
// Row k1 of the scratch matrix (4096 complex values, contiguous) -> its 4096 dB values, written IN PLACE over the first half of the row
// ([k1][k2] order: every access of this pass is contiguous).  A workgroup has all of its row in registers before the first LDS exchange,
// i.e. long before it stores anything, and no other workgroup touches the row.  fft_transpose_body then puts the values into bin
// order k = k1 + N1 * k2.  Same butterflies, same dB expression as fft_pass2_body: only where the results go differs.
template <int LG2>
__device__ __forceinline__ void fft_pass2row_body(const KIdx bid, float2* tw, float2* data, float2* __restrict__ scratch, const float2* __restrict__ tw2_g, int lg1) {
    constexpr int L2 = 1 << LG2;
    constexpr int TPF = L2 / 16;
    const int t = threadIdx.x;
    for (int e = threadIdx.x; e < L2 / 2; e += TPF) { tw[e] = global_load_f32x2(tw2_g, e); }
    float2* row = scratch + ((size_t)bid.x << LG2);  // bid.x = frame * N1 + k1
    float2 r[16];
    using R0 = FftRound<LG2, 0, 4>;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int p = R0::pos(t, 0, j);
        const int n = (int)(__brev((unsigned)p) >> (32 - LG2));
        r[j] = global_load_f32x2(row, n);  // (explicit GLOBAL accesses throughout: FLAT ones as a tick role, see fft_pass2_body)
    }
    __syncthreads();
    R0::compute(r, t, tw);
    fft_rounds_after_first<LG2, 4, IdxPad16>(r, t, tw, data, IdxPad16());
    using RL = typename FftLast<LG2>::Round;
    __syncthreads();
    float* tile = reinterpret_cast<float*>(data);  // [k2]
    const float inv = 1.0f / (float)((size_t)1 << (LG2 + lg1));
    unsigned abn = 0;
#pragma unroll
    for (int i = 0; i < RL::NG; i++) {
#pragma unroll
        for (int j = 0; j < RL::GS; j++) { tile[RL::pos(t, i, j)] = power_db_normal(r[i * RL::GS + j], inv, abn); }
    }
    if (abn >= SDRPP_LOG2_ABNORMAL) {  // (see fft_pass2_body)
#pragma unroll
        for (int i = 0; i < RL::NG; i++) {
#pragma unroll
            for (int j = 0; j < RL::GS; j++) { tile[RL::pos(t, i, j)] = power_db(r[i * RL::GS + j], inv); }
        }
    }
    __syncthreads();
    float* dst = reinterpret_cast<float*>(row);
    for (int e = threadIdx.x; e < L2; e += TPF) { global_store_f32_boff(dst, (unsigned)e * 4u, tile[e]); }
}
template <int LG2>
__global__ __launch_bounds__((1 << LG2) / 16) void fft_pass2row_kernel(float2* __restrict__ scratch, const float2* __restrict__ tw2_g, int lg1) {
    __shared__ float2 tw[(1 << LG2) / 2];
    __shared__ float2 data[(1 << LG2) + (1 << LG2) / 16];
    fft_pass2row_body<LG2>(kidx(blockIdx), tw, data, scratch, tw2_g, lg1);
}

// ---- N > 65536, pass 3: dB rows [k1][k2] -> bin order k = k1 + N1 * k2 (+ doZoom's group maxima) ---------------------------------------------
// A workgroup moves a tile of all N1 rows x TK2 = 8192 / N1 consecutive k2 through LDS: it reads TK2 consecutive floats of every row
// (the rows sit 2 * 4096 floats apart: the first half of each complex scratch row) and writes, for every k2 of the tile, the N1 consecutive
// bins k1 + N1 * k2 — whole tiles of 32 KB are contiguous in the output line.  While the tile is in LDS it also leaves the maximum of every
// aligned group of `gsz` bins for the zoom kernel (gsz divides N1; a maximum does not care how it is split: waterfall.cpp:65-90).
#define SDRPP_FFT_TR_TILE 8192
__device__ __forceinline__ void fft_transpose_body(const KIdx bid, float* tile, const float* __restrict__ rows, float* __restrict__ out_db, float* __restrict__ grp_max,
                                                   int lg1, int lg2, int gsz) {
    const int N1 = 1 << lg1, TK2 = SDRPP_FFT_TR_TILE >> lg1, tiles = (1 << lg2) / TK2;
    const int frame = bid.x / tiles, k2_0 = (bid.x % tiles) * TK2;
    const int pitch = N1 + 1;
    const float* src = rows + (((size_t)frame << (lg1 + lg2)) << 1) + k2_0;
    const int lgt = 13 - lg1;  // log2(TK2)
    for (int e = threadIdx.x; e < SDRPP_FFT_TR_TILE; e += 256) {
        const int k1 = e >> lgt, kk = e & (TK2 - 1);
        tile[kk * pitch + k1] = global_load_f32(src, ((size_t)k1 << (lg2 + 1)) + kk);
    }
    __syncthreads();
    float* dst = out_db + ((size_t)frame << (lg1 + lg2)) + ((size_t)k2_0 << lg1);
    for (int e = threadIdx.x; e < SDRPP_FFT_TR_TILE; e += 256) {
        const int kk = e >> lg1, k1 = e & (N1 - 1);
        global_store_f32_boff(dst, (unsigned)e * 4u, tile[kk * pitch + k1]);
    }
    if (grp_max) {
        const int gpr = N1 / gsz;  // groups per k2
        float* gdst = grp_max + (((size_t)frame << (lg1 + lg2)) + ((size_t)k2_0 << lg1)) / gsz;
        for (int g = threadIdx.x; g < TK2 * gpr; g += 256) {
            const int kk = g / gpr, gi = g % gpr;
            float m = __uint_as_float(0xff800000u);
            for (int q = 0; q < gsz; q++) {
                const float v = tile[kk * pitch + gi * gsz + q];
                if (v > m) { m = v; }
            }
            global_store_f32_boff(gdst, (unsigned)g * 4u, m);
        }
    }
}
__global__ __launch_bounds__(256) void fft_transpose_kernel(const float* __restrict__ rows, float* __restrict__ out_db, float* __restrict__ grp_max, int lg1, int lg2, int gsz) {
    __shared__ float tile[SDRPP_FFT_TR_TILE + 256];
    fft_transpose_body(kidx(blockIdx), tile, rows, out_db, grp_max, lg1, lg2, gsz);
}

// ---- doZoom max-decimation + palette index (waterfall.cpp:65-90, 899-905) -------------------------------------------------------------
// The float32 running index of doZoom is evaluated on the host once per view change (sdrpp_host::zoomTable: first bin and bin
// count of every pixel).  TP consecutive work-items share a pixel and walk its bin range with stride TP, so the lanes of a wavefront
// read consecutive bins (the host picks TP = 16 / 4 / 1 from the bins-per-pixel of the view); each keeps the reference's comparison
// (`if (in > max) max = in`, -inf start, so a NaN never wins) and the partial maxima meet in LDS.  max is order independent: the
// result is bit-identical to the sequential scan.
template <int TP>
__device__ __forceinline__ void zoom_palette_body(const KIdx bid, float* part, const float* __restrict__ lines, int fft_size, int data_width,
                                                  const int32_t* __restrict__ zstart, const int32_t* __restrict__ zcount, float wf_min, float wf_max,
                                                  float* __restrict__ zoomed, int32_t* __restrict__ index, const float* __restrict__ grp_max, int gsz) {
    constexpr int PPB = 256 / TP;  // pixels per block
    const int line = bid.y;
    const int p = threadIdx.x / TP, q = threadIdx.x % TP;
    const int px = bid.x * PPB + p;
    const float* in = lines + (size_t)line * fft_size;
    float m = __uint_as_float(0xff800000u);  // -inf
    if (px < data_width) {
        const int s = global_load_i32(zstart, px), e = s + global_load_i32(zcount, px);
        if (grp_max && e - s >= 2 * gsz) {
            // pass 2 left the maximum of every aligned group of gsz bins: the pixel's range = ragged head + whole groups + ragged tail
            const int a = ((s + gsz - 1) / gsz) * gsz, bnd = (e / gsz) * gsz;
            const int nh = a - s, ng = (bnd - a) / gsz, nt = e - bnd;
            const float* g = grp_max + (size_t)line * (fft_size / gsz) + a / gsz;
            for (int i = q; i < nh + ng + nt; i += TP) {
                const float v = (i < nh) ? global_load_f32(in, s + i) : ((i < nh + ng) ? global_load_f32(g, i - nh) : global_load_f32(in, bnd + (i - nh - ng)));
                if (v > m) { m = v; }
            }
        }
        else {
            for (int b = s + q; b < e; b += TP) {
                const float v = global_load_f32(in, b);
                if (v > m) { m = v; }
            }
        }
    }
    if constexpr (TP > 1) {
        part[p * (TP + 1) + q] = m;
        __syncthreads();
    }
    if (q == 0 && px < data_width) {
        if constexpr (TP > 1) {
#pragma unroll
            for (int i = 1; i < TP; i++) {
                const float v = part[p * (TP + 1) + i];
                if (v > m) { m = v; }
            }
        }
        global_store_f32_boff(zoomed, (unsigned)(((size_t)line * data_width + px) * 4u), m);
        const float range = wf_max - wf_min;
        const float v = (m < wf_min) ? wf_min : ((wf_max < m) ? wf_max : m);
        const float pixel = (v - wf_min) / range;
        global_store_f32_boff(index, (unsigned)(((size_t)line * data_width + px) * 4u), __int_as_float((int32_t)(pixel * 999999.0f)));
    }
}
template <int TP>
__global__ __launch_bounds__(256) void zoom_palette_kernel(const float* __restrict__ lines, int fft_size, int data_width,
                                                          const int32_t* __restrict__ zstart, const int32_t* __restrict__ zcount,
                                                          float wf_min, float wf_max, float* __restrict__ zoomed, int32_t* __restrict__ index,
                                                          const float* __restrict__ grp_max, int gsz) {
    __shared__ float part[(256 / TP) * (TP + 1)];
    zoom_palette_body<TP>(kidx(blockIdx), part, lines, fft_size, data_width, zstart, zcount, wf_min, wf_max, zoomed, index, grp_max, gsz);
}

// ---- WaterFall display state (waterfall.cpp:875-941): raw-line ring, FFT trace smoothing / hold -----------------------------------------
// getFFTBuffer (:875-886): every new line moves currentFFTLine one slot DOWN (mod height) and is written there.  Line f of this
// push (0 = oldest) therefore lands in slot (cur0 - 1 - f) mod H; when a push brings more than H lines only the last H survive.
__device__ __forceinline__ void wf_ring_store_body(const KIdx bid, const KIdx gdim, const float* __restrict__ lines, int nframes, int fft_size, float* __restrict__ ring, int height, int cur0) {
    const int f = bid.y;
    if (f < nframes - height) { return; }
    int slot = (cur0 - 1 - f) % height;
    if (slot < 0) { slot += height; }
    const float4* src = reinterpret_cast<const float4*>(lines + (size_t)f * fft_size);
    float4* dst = reinterpret_cast<float4*>(ring + (size_t)slot * fft_size);
    for (int i = bid.x * 256 + (int)threadIdx.x; i < fft_size / 4; i += gdim.x * 256) { dst[i] = src[i]; }
}
__global__ __launch_bounds__(256) void wf_ring_store_kernel(const float* __restrict__ lines, int nframes, int fft_size, float* __restrict__ ring, int height, int cur0) {
    wf_ring_store_body(kidx(blockIdx), kidx(gridDim), lines, nframes, fft_size, ring, height, cur0);
}
// pushFFT tail (:913-939), one work-item per pixel walking over the new lines in order: smoothing = three separately rounded passes
// (latest *= alpha; buf *= beta; buf += latest; latest = buf), hold[i] = max(latest[i], hold[i] - speed) for i >= 1.
__device__ __forceinline__ void wf_trace_body(const KIdx bid, const float* __restrict__ zoomed, int nframes, int data_width, float* __restrict__ latest,
                                              float* __restrict__ smooth, float alpha, float beta, float* __restrict__ hold, float hold_speed) {
    const int j = bid.x * 256 + (int)threadIdx.x;
    if (j >= data_width) { return; }
    float s = smooth ? smooth[j] : 0.0f;
    float h = hold ? hold[j] : 0.0f;
    float l = 0.0f;
    for (int f = 0; f < nframes; f++) {
        l = zoomed[(size_t)f * data_width + j];
        if (smooth) {
            l = l * alpha;
            s = s * beta;
            s = s + l;
            l = s;
        }
        if (hold && j >= 1) {
            const float b = h - hold_speed;
            h = (l < b) ? b : l;
        }
    }
    latest[j] = l;
    if (smooth) { smooth[j] = s; }
    if (hold) { hold[j] = h; }
}
__global__ __launch_bounds__(256) void wf_trace_kernel(const float* __restrict__ zoomed, int nframes, int data_width, float* __restrict__ latest,
                                                      float* __restrict__ smooth, float alpha, float beta, float* __restrict__ hold, float hold_speed) {
    wf_trace_body(kidx(blockIdx), zoomed, nframes, data_width, latest, smooth, alpha, beta, hold, hold_speed);
}

// calculateVFOSignalInfo (waterfall.cpp:558-598) on one raw line: out[0] = max over [o1, o2], out[1] = max - mean of [o0, o1) and (o2, o3)
// (double accumulation; a tree instead of the reference's bin order: the double sums agree to ~1e-13 relative, the float snr to 1 ulp).
__global__ __launch_bounds__(256) void wf_signal_info_kernel(const float* __restrict__ line, int o0, int o1, int o2, int o3, float* __restrict__ out) {
    __shared__ double ssum[256];
    __shared__ float smax[256];
    double acc = 0.0;
    float m = __uint_as_float(0xff800000u);
    for (int i = o0 + threadIdx.x; i < o1; i += 256) { acc += (double)line[i]; }
    for (int i = o2 + 1 + threadIdx.x; i < o3; i += 256) { acc += (double)line[i]; }
    for (int i = o1 + threadIdx.x; i <= o2; i += 256) {
        const float v = line[i];
        if (v > m) { m = v; }
    }
    ssum[threadIdx.x] = acc;
    smax[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            ssum[threadIdx.x] += ssum[threadIdx.x + d];
            if (smax[threadIdx.x + d] > smax[threadIdx.x]) { smax[threadIdx.x] = smax[threadIdx.x + d]; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int cnt = (o1 > o0 ? o1 - o0 : 0) + (o3 > o2 + 1 ? o3 - (o2 + 1) : 0);
        const double avg = ssum[0] / (double)cnt;
        out[0] = smax[0];
        out[1] = (float)((double)smax[0] - avg);
    }
}

// ---- sink-side sample packing: f32 -> int16 / int8 on the device, so the D2H copy carries 2 or 1 byte per value ----
// VOLK generic conversion (volk_32f_s32f_convert_16i / _8i): r = x * scalar, clamp to the integer range, rintf (round half to even), cast.
template <typename T>
__global__ __launch_bounds__(256) void pack_convert_kernel(const float* __restrict__ in, float scalar, long long n, T* __restrict__ out) {
    constexpr float hi = sizeof(T) == 2 ? 32767.0f : 127.0f, lo = sizeof(T) == 2 ? -32768.0f : -128.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float r = in[i] * scalar;
        if (r > hi) { r = hi; }
        else if (r < lo) { r = lo; }
        out[i] = (T)rintf(r);
    }
}
// largest VALUE of a float array (volk_32f_index_max_32u as SampleStreamCompressor uses it: only the value matters); partial[blockIdx.x]
__global__ __launch_bounds__(256) void pack_max_kernel(const float* __restrict__ in, long long n, float* __restrict__ partial) {
    __shared__ float sm[256];
    float m = __uint_as_float(0xff800000u);  // -inf; `src[i] > max` never lets a NaN win, like the reference loop (whose seed is src[0])
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = in[i];
        if (v > m) { m = v; }
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            const float o = sm[threadIdx.x + d];
            if (o > sm[threadIdx.x]) { sm[threadIdx.x] = o; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[blockIdx.x] = sm[0]; }
}

// int16 IQ -> float (file_source/main.cpp:162: volk_16i_s32f_convert_32f(out, in, 32768.0f, n))
__global__ __launch_bounds__(256) void int16_to_float_kernel(const int16_t* __restrict__ in, float* __restrict__ out, long long n) {
    const float inv = 1.0f / 32768.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { out[i] = ((float)in[i]) * inv; }
}

}  // namespace sdrpp_k
