// Per-VFO channeliser + demodulator kernels for gfx950.
//
// Reference data flow per VFO and per input block (core/src/dsp/channel/rx_vfo.h:89-100):
//   FrequencyXlator (VOLK rotator, full input rate)  ->  PowerDecimator cascade of DecimatingFIR stages
//   (dsp/filter/decimating_fir.h:45-68)  ->  PolyphaseResampler (dsp/multirate/polyphase_resampler.h:69-99)  ->
//   channel FIR (dsp/filter/fir.h:62-83)  ->  demodulator (dsp/demod/{quadrature,fm,broadcast_fm,am,ssb}.h).
// The reference runs one thread per block and one VOLK dot product per output sample; every VFO re-reads its own copy of
// the input (Splitter memcpy).  Here the time axis is split across workgroups (each stream keeps the (taps-1)-sample
// history of its consumer in a small side buffer, so results do not depend on how the input is cut into pushes), all
// VFOs of a launch are processed by the same grid, and the full-rate stage reads the shared IQ buffer once per tile.
//
// Numerics: summation order inside a dot product and the NCO differ from the reference's sequential fp32 recursion
// (VOLK's own SIMD kernels differ from its generic ones in the same way); parity is by tolerance (1e-5 RMS), see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <sdrpp_gfx950.h>
#include "fft_kernels.h"

namespace sdrpp_k {

// A stream of `width`-float samples: this push's samples in `data`, the previous `hist_len` samples in `hist`.
struct StreamIn {
    const float* data;
    const float* hist;
    int hist_len;
    int n;  // valid samples in `data`
};
__device__ __forceinline__ float2 stream_load2(const StreamIn& s, int i) {
    const float2* d = reinterpret_cast<const float2*>(s.data);
    const float2* h = reinterpret_cast<const float2*>(s.hist);
    if (i >= s.n) { return make_float2(0.0f, 0.0f); }  // tile over-read past the end of this push
    return (i >= 0) ? global_load_f32x2(d, i) : global_load_f32x2(h, s.hist_len + i);  // (explicit GLOBAL loads: a plain dereference of a job-table pointer is FLAT)
}
__device__ __forceinline__ float stream_load1(const StreamIn& s, int i) {
    if (i >= s.n) { return 0.0f; }
    return (i >= 0) ? global_load_f32(s.data, i) : global_load_f32(s.hist, s.hist_len + i);
}
// The same without a branch, for loops that fetch several samples per lane: the load is unconditional (the address is clamped into the
// stream, the value selected afterwards), so the compiler issues all loads of the loop before the first wait — behind a per-element
// branch every load costs its own memory round trip (measured: 18 x 0.75 us for the first window of the audio filter of a 50 000-sample
// block).  Same values; needs i >= -hist_len like the functions above.
__device__ __forceinline__ float2 stream_load2_nb(const StreamIn& s, int i, bool ok = true) {  // ok false: zero (no load is ever guarded by a branch)
    const bool use = ok && i < s.n, cur = i >= 0;
    int ic = cur ? i : (s.hist_len + i);
    ic = (use && ic >= 0) ? ic : 0;
    const float2 v = global_load_f32x2(reinterpret_cast<const float2*>((cur || !use) ? s.data : s.hist), ic);  // (not wanted: element 0 of the data buffer, which always exists)
    return use ? v : make_float2(0.0f, 0.0f);
}
__device__ __forceinline__ float stream_load1_nb(const StreamIn& s, int i, bool ok = true) {
    const bool use = ok && i < s.n, cur = i >= 0;
    int ic = cur ? i : (s.hist_len + i);
    ic = (use && ic >= 0) ? ic : 0;
    const float v = global_load_f32((cur || !use) ? s.data : s.hist, ic);
    return use ? v : 0.0f;
}

// =====================================================================================================================
// Stage 1: frequency translation folded into the first decimating FIR, VT VFOs per work-item sharing one LDS input tile
// =====================================================================================================================
// Reference:  r[n] = x[n] * e^{j(phi0 + n*theta)}  (rotator), then  y[j] = sum_k h[k] * r[i0 + k],  i0 = off0 + j*D - (K-1).
// Same sum:   y[j] = e^{j(phi0 + (i0 + kc)*theta)} * sum_k g[k] * x[i0 + k],   g[k] = h[k] * e^{j(k - kc)*theta}
// with complex taps g (host, double precision -> float) and ONE phasor per output instead of one per input sample.
// theta = arg(phaseDelta) of the reference's float phaseDelta, phi0 accumulated on the host in double.
#define SDRPP_S1_MAX_VT 8
struct Stage1Job {
    int nv;                  // VFOs handled by this job (<= VT of the launch)
    int ntaps, log2_decim, off0, nout;
    int min_idx;             // samples before this push-relative index read as zero (a VFO added or reset mid-stream starts
                             // from an all-zero history: fir.h:24-26 clears the delay line)
    int anchor;              // the index phi0 belongs to (0 = the block's first sample; a push of a launch group: where ITS samples start)
    const float2* ctaps;     // [(ntaps+1)/2][VT] modulated tap pairs (see stage1_accumulate), VFO index fastest
    double theta[SDRPP_S1_MAX_VT];  // turns per input sample
    double phi0[SDRPP_S1_MAX_VT];   // turns at push-relative sample index 0
    float2* out[SDRPP_S1_MAX_VT];
};


// Symmetric-tap form of the translated FIR.  Every stage of the reference's decimation plans is linear phase (h[k] == h[K-1-k],
// checked on the host; asymmetric taps fall back to nothing here — the host refuses them), so with the modulation centred on
// the filter, g[K-1-k] = conj(g[k]) and
//     g[k]*a + conj(g[k])*b = g.re * (a + b) + j * g.im * (a - b)            (a = x[i0+k], b = x[i0+K-1-k])
// i.e. FOUR FMAs per tap PAIR and VFO instead of eight; the sum/difference are shared by all VT VFOs of the work-item.
// ctaps: [npairs][VT] float2 (g.re, g.im), npairs = (K+1)/2; an odd K has its centre tap as a last "pair" with b = 0, g.im = 0.
template <int VT>
__device__ __forceinline__ void stage1_accumulate(const float2* xs, int pitch, int lgD, int K, int j, const UniformF32 g, float2 (&acc)[VT]) {
    const int D = 1 << lgD;
    const int npairs = (K + 1) >> 1;
    const bool odd = (K & 1) != 0;
    for (int k = 0; k < npairs; k++) {
        const int kb = K - 1 - k;
        const float2 a = xs[(k & (D - 1)) * pitch + (k >> lgD) + j];
        float2 b = xs[(kb & (D - 1)) * pitch + (kb >> lgD) + j];
        if (odd && k == npairs - 1) { b = make_float2(0.0f, 0.0f); }
        const float sr = a.x + b.x, si = a.y + b.y, dr = a.x - b.x, di = a.y - b.y;
#pragma unroll
        for (int v = 0; v < VT; v++) {
            const float gr = g[2 * (k * VT + v)], gi = g[2 * (k * VT + v) + 1];
            acc[v].x = fmaf(gr, sr, acc[v].x);
            acc[v].x = fmaf(-gi, di, acc[v].x);
            acc[v].y = fmaf(gr, si, acc[v].y);
            acc[v].y = fmaf(gi, dr, acc[v].y);
        }
    }
}


// Compile-time (K, log2 D) variant: fully unrolled, so every LDS offset is an instruction immediate and the tap fetches are
// s_load_dwordx16 with constant offsets that the scheduler can hoist ahead of their use — no scalar address arithmetic at all
// (the generic loop spends as many SALU as VALU instructions; one scalar unit serves the four SIMDs of a CU).
template <int VT, int K, int LGD>
__device__ __forceinline__ void stage1_accumulate_static(const float2* xs, int pitch, int j, const UniformF32 g, float2 (&acc)[VT]) {
    constexpr int D = 1 << LGD;
    constexpr int NP = (K + 1) / 2;
    const float2* xj = xs + j;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const int kb = K - 1 - k;
        const float2 a = xj[(k & (D - 1)) * pitch + (k >> LGD)];
        float2 b = xj[(kb & (D - 1)) * pitch + (kb >> LGD)];
        if ((K & 1) && k == NP - 1) { b = make_float2(0.0f, 0.0f); }
        const float sr = a.x + b.x, si = a.y + b.y, dr = a.x - b.x, di = a.y - b.y;
#pragma unroll
        for (int v = 0; v < VT; v++) {
            const float gr = g[2 * (k * VT + v)], gi = g[2 * (k * VT + v) + 1];
            acc[v].x = fmaf(gr, sr, acc[v].x);
            acc[v].x = fmaf(-gi, di, acc[v].x);
            acc[v].y = fmaf(gr, si, acc[v].y);
            acc[v].y = fmaf(gi, dr, acc[v].y);
        }
    }
}

// grid = (ceil(max nout / TILE), njobs); block = TILE work-items; dynamic LDS = D * pitch float2 with
// pitch = TILE + ceil((K-1)/D) + 1.  LDS image is de-interleaved by decimation phase: sample s of the tile lives at
// [s mod D][s div D], so lane j reads x[j*D + k] at [k mod D][j + k div D] — consecutive lanes, consecutive addresses.
// (also a role of the tick kernel — TR_S1_1, round 5: banks too small for the matrix front end stay pipelined; `tile` work-items compute, all `nall` of the
// workgroup load)
template <int VT>
__device__ __forceinline__ void vfo_stage1_body(const KIdx bid, float2* xs, const int tile, const int nall, const IqSrc& src, const Stage1Job* __restrict__ jobs) {
    const Stage1Job& job = jobs[bid.y];
    const int j0 = bid.x * tile;
    if (j0 >= job.nout) { return; }
    const int K = job.ntaps, lgD = job.log2_decim, D = 1 << lgD;
    const int extra = (K - 1 + D - 1) >> lgD;
    const int pitch = tile + extra + 1;
    const int nsamp = (tile - 1) * D + K;
    const long long base = (long long)job.off0 + (long long)j0 * D - (K - 1);  // push-relative index of tile sample 0
    for (int s = threadIdx.x; s < nsamp; s += nall) {
        const long long gi = base + s;
        xs[(s & (D - 1)) * pitch + (s >> lgD)] = (gi < job.min_idx) ? make_float2(0.0f, 0.0f) : iq_load_clamped(src, gi);
    }
    __syncthreads();
    const int j = threadIdx.x;
    if (j >= tile) { return; }
    float2 acc[VT];
#pragma unroll
    for (int v = 0; v < VT; v++) { acc[v] = make_float2(0.0f, 0.0f); }
    stage1_accumulate<VT>(xs, pitch, lgD, K, j, as_uniform(job.ctaps), acc);  // taps are wave-uniform: scalar loads
    if (j0 + j >= job.nout) { return; }
    const double centre = (double)(base + (long long)j * D - job.anchor) + 0.5 * (double)(K - 1);
#pragma unroll
    for (int v = 0; v < VT; v++) {
        if (v < job.nv) {
            double ph = fma(centre, job.theta[v], job.phi0[v]);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            float2 y;
            y.x = fmaf(acc[v].x, cs, -(acc[v].y * sn));
            y.y = fmaf(acc[v].x, sn, acc[v].y * cs);
            job.out[v][j0 + j] = y;
        }
    }
}
template <int VT>
__global__ __launch_bounds__(256) void vfo_stage1_kernel(IqSrc src, const Stage1Job* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float2, xs)
    vfo_stage1_body<VT>(kidx(blockIdx), xs, (int)blockDim.x, (int)blockDim.x, src, jobs);
}

// Large first-stage decimation (D >= 32: the 61.44 MS/s plans decimate by 64 with 257..400 taps).  An LDS tile for even 64
// outputs would be ~36 KiB, leaving one wavefront per SIMD.  Consecutive outputs start D samples apart, so there is almost
// no overlap between neighbouring lanes to exploit anyway: every lane streams its own K contiguous samples straight from
// global memory (each 64-byte line is consumed over 8 iterations and stays in L1), no LDS, full occupancy.  Reuse is across
// the VT VFOs of the work-item, exactly as in the tiled kernel.
template <int VT>
__device__ __forceinline__ void vfo_stage1_direct_body(const KIdx bid, const IqSrc& src, const Stage1Job* __restrict__ jobs) {  // (256 work-items; role TR_S1D_1)
    const Stage1Job& job = jobs[bid.y];
    const int j = bid.x * 256 + (int)threadIdx.x;
    const int K = job.ntaps, lgD = job.log2_decim;
    const int jc = (j < job.nout) ? j : (job.nout - 1);  // lanes past the end redo the last output (no divergence), never store
    if (job.nout <= 0) { return; }
    const long long i0 = (long long)job.off0 + ((long long)jc << lgD) - (K - 1);
    const int npairs = (K + 1) >> 1;
    const bool odd = (K & 1) != 0;
    const UniformF32 g = as_uniform(job.ctaps);
    float2 acc[VT];
#pragma unroll
    for (int v = 0; v < VT; v++) { acc[v] = make_float2(0.0f, 0.0f); }
    // block-uniform fast path: every window of this block lies inside the current push
    const long long blk_first = (long long)job.off0 + ((long long)(bid.x * 256) << lgD) - (K - 1);
    const long long blk_last = (long long)job.off0 + ((long long)min(bid.x * 256 + 255, job.nout - 1) << lgD);
    const bool inside = blk_first >= 0 && blk_first >= job.min_idx && blk_last < src.n_cur;
    if (inside) {
        const float2* __restrict__ xa = src.cur + i0;
        for (int k = 0; k < npairs; k++) {
            const float2 a = xa[k];
            float2 b = xa[K - 1 - k];
            if (odd && k == npairs - 1) { b = make_float2(0.0f, 0.0f); }
            const float sr = a.x + b.x, si = a.y + b.y, dr = a.x - b.x, di = a.y - b.y;
#pragma unroll
            for (int v = 0; v < VT; v++) {
                const float gr = g[2 * (k * VT + v)], gi = g[2 * (k * VT + v) + 1];
                acc[v].x = fmaf(gr, sr, acc[v].x);
                acc[v].x = fmaf(-gi, di, acc[v].x);
                acc[v].y = fmaf(gr, si, acc[v].y);
                acc[v].y = fmaf(gi, dr, acc[v].y);
            }
        }
    }
    else {
        for (int k = 0; k < npairs; k++) {
            const long long ia = i0 + k, ib = i0 + K - 1 - k;
            const float2 a = (ia < job.min_idx) ? make_float2(0.0f, 0.0f) : iq_load_clamped(src, ia);
            float2 b = (ib < job.min_idx) ? make_float2(0.0f, 0.0f) : iq_load_clamped(src, ib);
            if (odd && k == npairs - 1) { b = make_float2(0.0f, 0.0f); }
            const float sr = a.x + b.x, si = a.y + b.y, dr = a.x - b.x, di = a.y - b.y;
#pragma unroll
            for (int v = 0; v < VT; v++) {
                const float gr = g[2 * (k * VT + v)], gi = g[2 * (k * VT + v) + 1];
                acc[v].x = fmaf(gr, sr, acc[v].x);
                acc[v].x = fmaf(-gi, di, acc[v].x);
                acc[v].y = fmaf(gr, si, acc[v].y);
                acc[v].y = fmaf(gi, dr, acc[v].y);
            }
        }
    }
    if (j >= job.nout) { return; }
    const double centre = (double)(i0 - job.anchor) + 0.5 * (double)(K - 1);
#pragma unroll
    for (int v = 0; v < VT; v++) {
        if (v < job.nv) {
            double ph = fma(centre, job.theta[v], job.phi0[v]);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            job.out[v][j] = make_float2(fmaf(acc[v].x, cs, -(acc[v].y * sn)), fmaf(acc[v].x, sn, acc[v].y * cs));
        }
    }
}
template <int VT>
__global__ __launch_bounds__(256) void vfo_stage1_direct_kernel(IqSrc src, const Stage1Job* __restrict__ jobs) { vfo_stage1_direct_body<VT>(kidx(blockIdx), src, jobs); }

// Rotation only (VFOs whose output rate is above half the input rate have no decimation stage: power_decimator.h:53-56).
struct RotJob {
    double theta, phi0;
    float2* out;
    int n;
};
__device__ __forceinline__ void vfo_rotate_body(const KIdx bid, const KIdx gdim, const IqSrc& src, const RotJob* __restrict__ jobs) {
    const RotJob& job = jobs[bid.y];
    for (int i = bid.x * blockDim.x + threadIdx.x; i < job.n; i += gdim.x * blockDim.x) {
        double ph = fma((double)i, job.theta, job.phi0);
        ph -= rint(ph);
        float sn, cs;
        sincospif(2.0f * (float)ph, &sn, &cs);
        const float2 x = iq_load(src, i);
        job.out[i] = make_float2(fmaf(x.x, cs, -(x.y * sn)), fmaf(x.x, sn, x.y * cs));
    }
}
__global__ __launch_bounds__(256) void vfo_rotate_kernel(IqSrc src, const RotJob* __restrict__ jobs) { vfo_rotate_body(kidx(blockIdx), kidx(gridDim), src, jobs); }

#include "vfo_rot_kernels.h"

// =====================================================================================================================
// Polyphase rational resampler (polyphase_resampler.h:75-93):
//   A_n = phase0 + n*M;  out[n] = sum_k bank[A_n mod L][k] * in[offset0 + A_n div L + k - (tpp-1)]
// bank[(L-1) - (i mod L)][i div L] = taps[i] (polyphase_bank.h:31-34) is laid out [phase][tpp] on the host.
// =====================================================================================================================
struct PolyJob {
    StreamIn in;
    float2* out;
    const float* bank;  // [interp][tpp]
    int interp, decim, tpp, phase0, off0, nout;
};

__device__ __forceinline__ void vfo_poly_body(const KIdx bid, float2* xs, const PolyJob* __restrict__ jobs) {  // 256 work-items, one output each (role TR_POLY)
    const PolyJob& job = jobs[bid.y];
    constexpr int tile = 256;
    const int n0 = bid.x * tile;
    if (n0 >= job.nout) { return; }
    const int L = job.interp, M = job.decim, tpp = job.tpp;
    const long long a0 = (long long)job.phase0 + (long long)n0 * M;
    int nlast = n0 + tile - 1;
    if (nlast >= job.nout) { nlast = job.nout - 1; }
    const long long a1 = (long long)job.phase0 + (long long)nlast * M;
    const int first = job.off0 + (int)(a0 / L) - (tpp - 1);  // stream index of the first sample this tile needs
    const int nsamp = (int)(a1 / L) - (int)(a0 / L) + tpp;
    for (int s = threadIdx.x; s < nsamp; s += tile) { xs[s] = stream_load2(job.in, first + s); }
    __syncthreads();
    const int n = n0 + threadIdx.x;
    if (n >= job.nout) { return; }
    const long long a = (long long)job.phase0 + (long long)n * M;
    const int ph = (int)(a % L);
    const int rel = (int)(a / L) - (int)(a0 / L);
    const float* __restrict__ t = job.bank + (size_t)ph * tpp;
    float2 acc = make_float2(0.0f, 0.0f);
    for (int k = 0; k < tpp; k++) {
        const float2 x = xs[rel + k];
        const float h = t[k];
        acc.x = fmaf(h, x.x, acc.x);
        acc.y = fmaf(h, x.y, acc.y);
    }
    job.out[n] = acc;
}
__global__ __launch_bounds__(256) void vfo_poly_kernel(const PolyJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float2, xs)
    vfo_poly_body(kidx(blockIdx), xs, jobs);
}

// =====================================================================================================================
// FM discriminator (quadrature.h:39-46): out[i] = normalizePhase(atan2f(x[i]) - atan2f(x[i-1])) * invDeviation — fused into the loads
// of the audio low-pass kernels (QUAD); this is its phase wrap.
// =====================================================================================================================
// atan2f for the discriminator: |error| <= 3e-7 rad against double precision (tests/host_cpp/test_device_math.cpp; libm's is ~1 ulp = 2.4e-7 at pi) in ~23 vector instructions instead of the
// ~53 of the library routine — the phase of every IF sample is taken on the way into the audio low-pass, which made this the
// largest single cost of that kernel.  Octant reduction to z = min/max in [0, 1], odd polynomial z * P(z^2) of degree 17
// (least-squares fit on Chebyshev nodes, weighted by z; max error 8.9e-8 in float arithmetic), then the usual reflections.
__device__ __forceinline__ float fm_phase(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1.17549435e-38f), mn = fminf(ax, ay);  // (0, 0) -> z = 0 -> phase 0 like atan2f
    const float z = mn * fast_rcp(mx);
    const float w = z * z;
    float p = 0.0023981390986591578f;
    p = fmaf(p, w, -0.014152348041534424f);
    p = fmaf(p, w, 0.03934541344642639f);
    p = fmaf(p, w, -0.07194384187459946f);
    p = fmaf(p, w, 0.10477539151906967f);
    p = fmaf(p, w, -0.1415480673313141f);
    p = fmaf(p, w, 0.19984884560108185f);
    p = fmaf(p, w, -0.33332523703575134f);
    p = fmaf(p, w, 0.9999998807907104f);
    float r = z * p;
    r = (ay > ax) ? 1.57079632679489662f - r : r;
    r = (x < 0.0f) ? 3.14159265358979324f - r : r;
    return copysignf(r, y);
}
__device__ __forceinline__ float normalize_phase(float d) {
    const float FL_PI = 3.1415926535f;  // math/constants.h:4, math/normalize_phase.h:6-9
    if (d > FL_PI) { d -= 2.0f * FL_PI; }
    else if (d <= -FL_PI) { d += 2.0f * FL_PI; }
    return d;
}

// =====================================================================================================================
// Sequential tails at IF rate — one work-item per VFO, exactly the reference's per-sample recursions:
//   AM  (am.h:101-131): [carrier AGC] -> |x| -> DC blocker (dc_blocker.h:54-60) -> [audio AGC] -> (LPF runs afterwards as a FIR job)
//   SSB (ssb.h:77-92) : second translation (closed-form NCO) -> Re{} -> AGC (agc.h:70-109) -> {v, v}
// The AGC look-ahead on clipping scans to the end of the reference block (SeqJob::bounds; without them: to the end of the push).
// =====================================================================================================================
struct AgcState {
    float set_point, attack, inv_attack, decay, inv_decay, max_gain, max_output_amp, amp;
};
// Parallel part of AM / SSB: everything before the first per-sample recursion.
//   AM (audio AGC):  pre[i] = |x[i]|                         (volk_32fc_magnitude_32f, am.h:120)
//   SSB:             pre[i] = Re{ x[i] * e^{j(phi2 + i*theta2)} }   (ssb.h:79-88: second translation + ComplexToReal)
struct PreJob {
    int mode, n;
    const float2* in;
    float* out;
    double theta2, phi2;
};
__device__ __forceinline__ void vfo_demod_pre_body(const KIdx bid, const KIdx gdim, const PreJob* __restrict__ jobs) {
    const PreJob& job = jobs[bid.y];
    for (int i = bid.x * blockDim.x + threadIdx.x; i < job.n; i += gdim.x * blockDim.x) {
        const float2 x = global_load_f32x2(job.in, i);  // (explicit GLOBAL accesses: FLAT ones as a tick role)
        if (job.mode == 2) { global_store_f32_boff(job.out, (unsigned)i * 4u, sqrtf((x.x * x.x) + (x.y * x.y))); }
        else {
            double ph = fma((double)i, job.theta2, job.phi2);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            global_store_f32_boff(job.out, (unsigned)i * 4u, fmaf(x.x, cs, -(x.y * sn)));
        }
    }
}
__global__ __launch_bounds__(256) void vfo_demod_pre_kernel(const PreJob* __restrict__ jobs) { vfo_demod_pre_body(kidx(blockIdx), kidx(gridDim), jobs); }

struct SeqJob {
    int mode;  // 2 AM, 3/4/5 SSB family
    int n;
    const float2* in;  // complex IF samples of this push (AM carrier-AGC mode only)
    float* pre;        // real samples from vfo_demod_pre_kernel; AM overwrites them in place with the low-pass input
    float* out;        // SSB: stereo float2 output
    AgcState* agc;     // persistent (device)
    AgcState* carrier_agc;
    float* dc_offset;  // persistent
    float dc_rate;
    int carrier_mode;
    // reference blocks inside this push (cumulative sample counts; nullptr: the push is one block).  loop::AGC's look-ahead on
    // clipping scans to the end of the CURRENT BLOCK (agc.h:91-104), so its result depends on how the reference cut the stream.
    const int* bounds;
    int nb;
};

__device__ __forceinline__ float agc_gain(AgcState& a, float inAmp) {
    float gain;
    if (inAmp != 0.0f) {
        a.amp = (inAmp > a.amp) ? ((a.amp * a.inv_attack) + (inAmp * a.attack)) : ((a.amp * a.inv_decay) + (inAmp * a.decay));
        const float g = a.set_point / a.amp;
        gain = (a.max_gain < g) ? a.max_gain : g;
    }
    else {
        gain = 1.0f;
    }
    return gain;
}

// loop::AGC's amplitude tracker alone (agc.h:79-83): the part of the recursion that is really sequential.  The gain — a division per
// sample — depends on it but nothing depends on the gain, so it is taken out of the chain and evaluated for 64 samples at once.
__device__ __forceinline__ float agc_track(float amp, float inAmp, const AgcState& a) {
    if (inAmp != 0.0f) {
        const bool up = inAmp > amp;
        const float c1 = up ? a.inv_attack : a.inv_decay, c2 = up ? a.attack : a.decay;
        amp = (amp * c1) + (inAmp * c2);
    }
    return amp;
}
__device__ __forceinline__ float agc_gain_of(float amp, float inAmp, const AgcState& a) {
    if (inAmp == 0.0f) { return 1.0f; }
    const float g = a.set_point / amp;
    return (a.max_gain < g) ? a.max_gain : g;
}

// One WAVEFRONT per VFO: only the recursions (DC blocker, AGC) are left here.  The lanes fetch 64 consecutive samples with one
// coalesced load; every lane then evaluates the same (uniform) recursion, taking sample i from lane i with v_readlane — a
// one-work-item loop over global memory pays ~1 us of load latency per sample.  The AGC's look-ahead to the end of the push
// (agc.h:91-104) is a wave-wide max reduction where it is a plain maximum, and the same chunked loop where it has to re-run the
// DC blocker forward (AM, audio AGC).
__device__ __forceinline__ void vfo_sequential_body(const KIdx bid, const SeqJob* __restrict__ jobs, int njobs) {
    const int id = bid.x;
    if (id >= njobs) { return; }
    const SeqJob job = jobs[id];
    const int lane = threadIdx.x;
    const int nblk = job.bounds ? job.nb : 1;
    if (job.mode == 2) {
        AgcState agc = *job.agc;
        AgcState cagc = *job.carrier_agc;
        float off = *job.dc_offset;
        int blk_lo = 0;
        for (int blk = 0; blk < nblk; blk++) {
        const int n = job.bounds ? job.bounds[blk] : job.n;  // end of this reference block
        for (int base = blk_lo; base < n; base += 64) {
            const int cnt = (n - base < 64) ? n - base : 64;
            float2 xin = make_float2(0.0f, 0.0f);
            float pv = 0.0f;
            if (lane < cnt) {
                if (job.carrier_mode) { xin = job.in[base + lane]; }
                else { pv = job.pre[base + lane]; }
            }
            const float amp_l = sqrtf((xin.x * xin.x) + (xin.y * xin.y));  // carrier mode: |x| of this lane's sample
            float outv = 0.0f;
            if (job.carrier_mode) {
                // carrier AGC on the complex IF (am.h:103-106), then envelope and DC blocker: sample by sample
                for (int i = 0; i < cnt; i++) {
                    float2 x = make_float2(wave_bcast(xin.x, i), wave_bcast(xin.y, i));
                    const float inAmp = wave_bcast(amp_l, i);
                    float gain = agc_gain(cagc, inAmp);
                    if (inAmp * gain > cagc.max_output_amp) {
                        float m = (lane >= i && lane < cnt) ? amp_l : 0.0f;  // rest of this chunk, then the rest of the block
                        for (int b2 = base + 64 + lane; b2 < n; b2 += 64) {
                            const float2 y = job.in[b2];
                            const float a = sqrtf((y.x * y.x) + (y.y * y.y));
                            if (a > m) { m = a; }
                        }
                        cagc.amp = wave_max(m);
                        const float g = cagc.set_point / cagc.amp;
                        gain = (cagc.max_gain < g) ? cagc.max_gain : g;
                    }
                    x.x = x.x * gain;
                    x.y = x.y * gain;
                    const float mag = sqrtf((x.x * x.x) + (x.y * x.y));
                    const float v = mag - off;
                    off += v * job.dc_rate;
                    if (lane == i) { outv = v; }
                }
            }
            else {
                // envelope (already in `pre`) -> DC blocker -> audio AGC.  Sequential per chunk: only the DC blocker and the AGC's amplitude
                // tracker (lane i keeps v, the tracker and the blocker's offset after sample i); gains and the clip test in one parallel step.
                // A clip is handled at its sample as the reference does: the look-ahead needs the not-yet-computed future samples of the same
                // recursion, so it re-runs the DC blocker forward to the end of the BLOCK from the state behind that sample (exactly what
                // the reference's in-place buffer holds at that moment), and the scan restarts behind it.
                int i0 = 0;
                while (i0 < cnt) {
                    float o = off, amp = agc.amp, my_v = 0.0f, my_amp = 0.0f, my_off = 0.0f;
                    for (int i = i0; i < cnt; i++) {
                        const float v = wave_bcast(pv, i) - o;
                        o += v * job.dc_rate;
                        amp = agc_track(amp, fabsf(v), agc);
                        if (lane == i) {
                            my_v = v;
                            my_amp = amp;
                            my_off = o;
                        }
                    }
                    const bool mine = lane >= i0 && lane < cnt;
                    const float a_l = fabsf(my_v);
                    const float g_l = mine ? agc_gain_of(my_amp, a_l, agc) : 1.0f;
                    const int f = wave_first(mine && (a_l * g_l > agc.max_output_amp));
                    if (mine && lane < f) { outv = my_v * g_l; }
                    if (f >= 64) {
                        off = o;
                        agc.amp = amp;
                        break;
                    }
                    float maxAmp = wave_bcast(a_l, f);
                    float o2 = wave_bcast(my_off, f);
                    off = o2;
                    for (int jn = f + 1; jn < cnt; jn++) {
                        const float v2 = wave_bcast(pv, jn) - o2;
                        o2 += v2 * job.dc_rate;
                        const float a2 = fabsf(v2);
                        if (a2 > maxAmp) { maxAmp = a2; }
                    }
                    for (int b2 = base + 64; b2 < n; b2 += 64) {
                        const int c2 = (n - b2 < 64) ? n - b2 : 64;
                        const float q = (lane < c2) ? job.pre[b2 + lane] : 0.0f;
                        for (int jn = 0; jn < c2; jn++) {
                            const float v2 = wave_bcast(q, jn) - o2;
                            o2 += v2 * job.dc_rate;
                            const float a2 = fabsf(v2);
                            if (a2 > maxAmp) { maxAmp = a2; }
                        }
                    }
                    agc.amp = maxAmp;
                    const float g = agc.set_point / agc.amp;
                    const float gain = (agc.max_gain < g) ? agc.max_gain : g;
                    if (lane == f) { outv = my_v * gain; }
                    i0 = f + 1;
                }
            }
            if (lane < cnt) { job.pre[base + lane] = outv; }
        }
        blk_lo = n;
        }
        if (lane == 0) {
            *job.agc = agc;
            *job.carrier_agc = cagc;
            *job.dc_offset = off;
        }
    }
    else {
        AgcState agc = *job.agc;
        float2* out = reinterpret_cast<float2*>(job.out);
        int blk_lo = 0;
        for (int blk = 0; blk < nblk; blk++) {
        const int n = job.bounds ? job.bounds[blk] : job.n;
        for (int base = blk_lo; base < n; base += 64) {
            const int cnt = (n - base < 64) ? n - base : 64;
            const float pv = (lane < cnt) ? job.pre[base + lane] : 0.0f;
            const float a_l = fabsf(pv);
            float outv = 0.0f;
            // Chunk of 64 samples: the amplitude tracker runs sequentially (uniform, ~10 instructions per sample), lane i keeps the value
            // after sample i; gains and the clip test are then one parallel step.  A clip (rare: the start of a burst) is handled at its
            // sample exactly as the reference does — amp = maximum over the rest of the BLOCK — and the scan restarts behind it.
            int i0 = 0;
            while (i0 < cnt) {
                float amp = agc.amp, my_amp = 0.0f;
                for (int i = i0; i < cnt; i++) {
                    amp = agc_track(amp, wave_bcast(a_l, i), agc);
                    if (lane == i) { my_amp = amp; }
                }
                const bool mine = lane >= i0 && lane < cnt;
                const float g_l = mine ? agc_gain_of(my_amp, a_l, agc) : 1.0f;
                const int f = wave_first(mine && (a_l * g_l > agc.max_output_amp));
                if (mine && lane < f) { outv = pv * g_l; }
                if (f >= 64) {
                    agc.amp = amp;
                    break;
                }
                float m = (lane >= f && lane < cnt) ? a_l : 0.0f;  // rest of this chunk, then the rest of the block
                for (int b2 = base + 64 + lane; b2 < n; b2 += 64) {
                    const float a2 = fabsf(job.pre[b2]);
                    if (a2 > m) { m = a2; }
                }
                agc.amp = wave_max(m);
                const float g = agc.set_point / agc.amp;
                const float gain = (agc.max_gain < g) ? agc.max_gain : g;
                if (lane == f) { outv = pv * gain; }
                i0 = f + 1;
            }
            if (lane < cnt) { out[base + lane] = make_float2(outv, outv); }
        }
        blk_lo = n;
        }
        if (lane == 0) { *job.agc = agc; }
    }
}
__global__ __launch_bounds__(64) void vfo_sequential_kernel(const SeqJob* __restrict__ jobs, int njobs) { vfo_sequential_body(kidx(blockIdx), jobs, njobs); }

// =====================================================================================================================
// History carry: after a push of n samples, the new history of a stream is the last hist_len samples of (old history ++ data).
// Written to the stream's alternate history buffer (ping-pong), so the update is race-free for any n.
// =====================================================================================================================
struct CarryJob {
    const float* data;
    const float* old_hist;
    float* new_hist;
    int hist_len, n, width;
    int need;  // only the most recent `need` samples will be read by the next push: older entries are not copied
};
// njw > 0: ONE WAVEFRONT per job (job 4 bid.y + wavefront of njw) — the per-VFO histories are a few hundred samples, a workgroup's life is
// the chain of round trips to its job and back whatever it moves, and in a tick workgroup SLOTS are what the roles compete for (cfg 4:
// 1 300-1 500 carry workgroups of 4.5 us were an eighth of the tick's slot time); njw = 0: grid.x workgroups stride over job bid.y.
__device__ __forceinline__ void carry_body(const KIdx bid, const KIdx gdim, const CarryJob* __restrict__ jobs, int njw) {
    const int jidx = njw > 0 ? bid.y * 4 + ((int)threadIdx.x >> 6) : bid.y;
    if (njw > 0 && jidx >= njw) { return; }
    const CarryJob job = jobs[jidx];
    const int first = (job.hist_len - job.need) * job.width;
    const int total = job.hist_len * job.width;
    // new_hist[e] = (old_hist ++ data)[n * width + e]: elements below `eb` still come from the old history (a push shorter than the history),
    // the rest from the data of this push at data[e - eb]
    const long long nw = (long long)job.n * job.width;
    const long long ebl = (long long)total - nw;
    const int eb = ebl < 0 ? 0 : (ebl > total ? total : (int)ebl);
    // Round 5: FOUR floats per access (one dwordx4 load / store, 4-byte alignment is all global memory asks for) and eight accesses in flight per
    // work-item before the first store — the carries of a tick were thousands of workgroups of one 4-byte load per work-item each (cfg 4: ~2 300
    // workgroups of 3.9 us, the whole tail of the tick), their life a memory round trip whatever they carry: fewer, fatter workgroups.
    const int first4 = (first + 3) & ~3;
    const int nthreads = njw > 0 ? 64 : gdim.x * 256, t = njw > 0 ? ((int)threadIdx.x & 63) : bid.x * 256 + (int)threadIdx.x;
    for (int e = first + t; e < first4 && e < total; e += nthreads) {  // (the up to three elements in front of the first whole quad)
        const long long sx = nw + e;
        global_store_f32_boff(job.new_hist, (unsigned)e * 4u, global_load_f32(e < eb ? job.old_hist : job.data, e < eb ? sx : (long long)e - ebl));
    }
    constexpr int U = 8;
    for (int q0 = first4 + 4 * t; q0 < total; q0 += 4 * nthreads * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int e = q0 + 4 * nthreads * u;
            if (e >= total) { e = first4; }  // (beyond the end: some quad that exists — never a guarded load; nothing is stored for it below)
            if (e + 3 < eb) { v[u] = global_load_f32x4_unaligned(job.old_hist, nw + e); }
            else if (e >= eb && e + 3 < total) { v[u] = global_load_f32x4_unaligned(job.data, (long long)e - ebl); }
            else {  // the quad that straddles the seam between the two sources, or the last, partial one: element by element
                float w4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int ek = e + k < total ? e + k : total - 1;
                    w4[k] = global_load_f32(ek < eb ? job.old_hist : job.data, ek < eb ? nw + ek : (long long)ek - ebl);
                }
                v[u] = make_float4(w4[0], w4[1], w4[2], w4[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = q0 + 4 * nthreads * u;
            if (e + 3 < total) { global_store_f32x4_unaligned(job.new_hist, e, v[u]); }
            else if (e < total) {  // the last, partial quad
                const float w4[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
                for (int k = 0; k < 4 && e + k < total; k++) { global_store_f32_boff(job.new_hist, (unsigned)(e + k) * 4u, w4[k]); }
            }
        }
    }
}
__global__ __launch_bounds__(256) void carry_kernel(const CarryJob* __restrict__ jobs, int njw) { carry_body(kidx(blockIdx), kidx(gridDim), jobs, njw); }

// =====================================================================================================================
// Output gather (sdrpp_vfo_read_many): the per-VFO output blocks of one push packed back to back, so that the host gets all of them
// with ONE device-to-host copy instead of one small copy (and stream synchronisation) per VFO.
// =====================================================================================================================
struct GatherJob {
    const float2* src;
    long long dst_off;  // samples
    int n;
};
__global__ __launch_bounds__(256) void gather_kernel(const GatherJob* __restrict__ jobs, float2* __restrict__ dst) {
    const GatherJob job = jobs[blockIdx.y];
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < job.n; i += (int)(gridDim.x * blockDim.x)) { dst[job.dst_off + i] = job.src[i]; }
}
// the same with the job table in the kernel arguments (up to 128 VFOs: 3 KB of the 4 KB the launch packet carries): no upload of the table,
// which for a read after every reference-sized block was a staged host-to-device copy of its own
#define SDRPP_GATHER_INLINE 128
struct GatherArgs { GatherJob j[SDRPP_GATHER_INLINE]; };
__global__ __launch_bounds__(256) void gather_inline_kernel(GatherArgs args, float2* __restrict__ dst) {
    const GatherJob job = args.j[blockIdx.y];
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < job.n; i += (int)(gridDim.x * blockDim.x)) { dst[job.dst_off + i] = job.src[i]; }
}

// =====================================================================================================================
// Register-blocked kernels (round-1 optimisation of the measured bottleneck).
//
// The generic FIR above issues one ds_read per two FMAs and is LDS-bound at ~10 TFLOP/s.  Here every work-item computes
// R = 8 consecutive outputs with a circular window of R registers: each input sample is read from LDS once and used for
// R outputs (R*R FMAs per R reads), taps are wave-uniform and arrive through scalar loads, R at a time.
//
// Decimation by D is handled as D ordinary FIRs over the polyphase components c_p[i] = x[base + D*i + p] with taps
// h_p[q] = h[D*q + p] (host lays them out phase-major, zero-padded to a multiple of R):
//      out[j] = sum_p sum_q h_p[q] * c_p[j + q]
// LDS image: component p, element e (tile-relative) at [p][e mod R][e div R]; work-item t reads elements t*R + m, i.e.
// [p][m mod R][t + m div R] — consecutive lanes, consecutive addresses.
// =====================================================================================================================
#define SDRPP_FIR_R 8
struct FirBJob {
    StreamIn in;
    float* out;
    const float* taps;  // [D][kp_pad], phase-major, zero padded
    int ntaps, log2_decim, off0, nout, kp_pad;
    float inv_deviation;  // QUAD only
};

// Decimating FIR on a complex stream whose window fits neither the matrix-core table nor an LDS tile (decimation 32 / 64 with hundreds
// of taps as a PLAIN filter: only in reference-rotator mode, where the first stage cannot be fused with the translation).  One output
// per work-item straight from global memory, k-ordered fmaf chain.  Correctness path of a parity mode, not tuned.
// REFORDER: the reference's own arithmetic — VOLK's generic dot product as DecimatingFIR::process calls it (decimating_fir.h:51-61):
// taps in order, product rounded, then added (two roundings per tap, no fused multiply-add).  The parity mode of the front end's
// pre-processing decimator (sdrpp_preproc_set_reference_order): bit-identical to the compiled reference.
template <bool REFORDER>
__device__ __forceinline__ void vfo_fir_direct_body(const KIdx bid, const KIdx gdim, const FirBJob* __restrict__ jobs) {
    const FirBJob& job = jobs[bid.y];
    const int D = 1 << job.log2_decim, kp = job.kp_pad;
    for (int j = bid.x * 256 + (int)threadIdx.x; j < job.nout; j += gdim.x * 256) {
        const int i0 = job.off0 + (j << job.log2_decim) - (job.ntaps - 1);
        float2 acc = make_float2(0.0f, 0.0f);
        for (int k = 0; k < job.ntaps; k++) {
            const float h = job.taps[(size_t)(k & (D - 1)) * kp + (size_t)(k >> job.log2_decim)];
            const float2 x = stream_load2(job.in, i0 + k);
            if constexpr (REFORDER) {
                const float pr = x.x * h, pi = x.y * h;  // (the translation unit is compiled with -ffp-contract=off: these stay products)
                acc.x = acc.x + pr;
                acc.y = acc.y + pi;
            }
            else {
                acc.x = fmaf(h, x.x, acc.x);
                acc.y = fmaf(h, x.y, acc.y);
            }
        }
        reinterpret_cast<float2*>(job.out)[j] = acc;
    }
}
template <bool REFORDER>
__global__ __launch_bounds__(256) void vfo_fir_direct_kernel(const FirBJob* __restrict__ jobs) { vfo_fir_direct_body<REFORDER>(kidx(blockIdx), kidx(gridDim), jobs); }

// The reference's DC blocker recursion itself (dc_blocker.h:54-60: out = in - offset; offset += out * rate, product rounded, then added)
// over the wideband stream, for the parity mode of the pre-processing chain: ONE wavefront walks the block, 64 samples per coalesced
// load, every lane evaluating the same recursion with sample i taken from lane i (v_readlane).  ~40 cycles per sample: a few times real
// time for a 10 MS/s stream — the default (a two-level scan of affine maps, vfo_deemph_kernel<1, *>) is the fast one and agrees to ~5e-5.
__global__ __launch_bounds__(64) void iq_dc_block_exact_kernel(const float2* __restrict__ in, float2* __restrict__ out, int n, float rate, float2* __restrict__ state, int conj) {
    const int lane = (int)threadIdx.x;
    float offr = state->x, offi = state->y;
    for (int base = 0; base < n; base += 64) {
        const int cnt = (n - base < 64) ? n - base : 64;
        const float2 v = (lane < cnt) ? in[base + lane] : make_float2(0.0f, 0.0f);
        float2 res = make_float2(0.0f, 0.0f);
        for (int i = 0; i < cnt; i++) {
            const float xr = wave_bcast(v.x, i), xi = wave_bcast(v.y, i);
            const float orr = xr - offr, oi = xi - offi;
            const float pr = orr * rate, pi = oi * rate;
            offr = offr + pr;
            offi = offi + pi;
            if (lane == i) { res = make_float2(orr, conj ? -oi : oi); }
        }
        if (lane < cnt) { out[base + lane] = res; }
    }
    if (lane == 0) { *state = make_float2(offr, offi); }
}

// QUAD (WIDTH 1, decimation 1): the input stream is the complex IF and the FM discriminator (quadrature.h:39-46) runs while the
// tile is loaded — d[i] = normalizePhase(atan2f(x[i]) - atan2f(x[i-1])) * invDeviation — so the demodulated stream never goes
// to memory.  The reference keeps the previous phase as state; here it is recomputed from the IF history (atan2f(0, 0) = 0
// reproduces the reset state).
template <int WIDTH, bool STEREO, bool QUAD = false>
__device__ __forceinline__ void vfo_firb_body(const KIdx bid, float* smem, const int nthreads, const FirBJob* __restrict__ jobs) {  // nthreads: work-items of the workgroup that take part (a multiple of 64)
    constexpr int R = SDRPP_FIR_R;
    const FirBJob& job = jobs[bid.y];
    const int nall = (int)blockDim.x;  // every work-item of the workgroup loads, `nthreads` of them compute
    const int tile = nthreads * R;
    const int j0 = bid.x * tile;
    if (j0 >= job.nout) { return; }
    const int K = job.ntaps, lgD = job.log2_decim, D = 1 << lgD, kp = job.kp_pad;
    const int P1 = nthreads + kp / R + 1;  // columns per (phase, residue) row
    const int P2 = R * P1;
    // component elements needed per phase: tile + kp - 1 (+R-1 preload slack) -> all inside R * P1
    const int ncomp = R * P1;
    const int base = job.off0 + j0 * D - (K - 1);  // stream index of component 0, element 0
    const int nvalid = (tile - 1) * D + K;         // samples a full tile really needs; the rest is zero-filled
    typedef typename std::conditional<WIDTH == 2, float2, float>::type T;
    T* xs = reinterpret_cast<T*>(smem);
    if constexpr (QUAD) {
        float* phase = smem + ncomp;  // phase[i] = atan2f(x[base - 1 + i]), i = 0 .. nvalid
        constexpr int UQ = 4;
        for (int s0 = threadIdx.x; s0 <= nvalid; s0 += nall * UQ) {
            float2 x[UQ];
#pragma unroll
            for (int u = 0; u < UQ; u++) { x[u] = stream_load2_nb(job.in, base - 1 + s0 + u * nall, s0 + u * nall <= nvalid); }
#pragma unroll
            for (int u = 0; u < UQ; u++) {
                if (s0 + u * nall <= nvalid) { phase[s0 + u * nall] = fm_phase(x[u].y, x[u].x); }
            }
        }
        __syncthreads();
        for (int s = threadIdx.x; s < ncomp; s += nall) {
            const float v = (s < nvalid) ? normalize_phase(phase[s + 1] - phase[s]) * job.inv_deviation : 0.0f;
            xs[(s & (R - 1)) * P1 + (s >> 3)] = v;
        }
    }
    else {
        // Eight loads in flight per work-item before the first LDS store, none behind a branch (stream_load*_nb): a tile of a decimator by 8 is
        // ~17 samples per work-item, and one guarded load per loop iteration made that 17 memory round trips one after the other — the whole
        // 18 us life of this role's workgroups in cfg 4's tick, 512 of them (round 5; same values, same order of everything that is rounded).
        constexpr int U = 8;
        for (int s0 = threadIdx.x; s0 < ncomp * D; s0 += nall * U) {
            T v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + u * nall;
                if constexpr (WIDTH == 2) { v[u] = stream_load2_nb(job.in, base + s, s < nvalid); }
                else { v[u] = stream_load1_nb(job.in, base + s, s < nvalid); }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + u * nall;
                if (s < ncomp * D) {
                    const int p = s & (D - 1), e = s >> lgD;
                    xs[p * P2 + (e & (R - 1)) * P1 + (e >> 3)] = v[u];
                }
            }
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= nthreads) { return; }  // (a role of the tick kernel: the workgroup is wider than the tile; everybody helped to load it and met the barriers)
    const UniformF32 taps = as_uniform(job.taps);
    T acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        if constexpr (WIDTH == 2) { acc[r] = make_float2(0.0f, 0.0f); }
        else { acc[r] = 0.0f; }
    }
    for (int p = 0; p < D; p++) {
        const T* xp = xs + p * P2 + t;
        T w[R];
#pragma unroll
        for (int m = 0; m < R - 1; m++) { w[m] = xp[m * P1]; }  // elements 0 .. R-2 (m div R == 0)
        for (int q0 = 0; q0 < kp; q0 += R) {
            const int col = (q0 >> 3);
#pragma unroll
            for (int u = 0; u < R; u++) {
                // element m = q0 + u + R - 1 -> residue (u - 1) mod R, column col + (u >= 1)
                const int res = (u + R - 1) & (R - 1);
                w[res] = xp[res * P1 + col + (u >= 1 ? 1 : 0)];
                const float h = taps[p * kp + q0 + u];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const T x = w[(u + r) & (R - 1)];
                    if constexpr (WIDTH == 2) {
                        acc[r].x = fmaf(h, x.x, acc[r].x);
                        acc[r].y = fmaf(h, x.y, acc[r].y);
                    }
                    else { acc[r] = fmaf(h, x, acc[r]); }
                }
            }
        }
    }
    const int jo = j0 + t * R;
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (jo + r < job.nout) {
            if constexpr (WIDTH == 2) { global_store_f32x2(reinterpret_cast<float2*>(job.out), jo + r, acc[r]); }  // (explicit GLOBAL stores: FLAT ones as a tick role)
            else if constexpr (STEREO) { global_store_f32x2(reinterpret_cast<float2*>(job.out), jo + r, make_float2(acc[r], acc[r])); }
            else { global_store_f32_boff(job.out, (unsigned)(jo + r) * 4u, acc[r]); }
        }
    }
}
template <int WIDTH, bool STEREO, bool QUAD = false>
__global__ __launch_bounds__(256) void vfo_firb_kernel(const FirBJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float, smem)
    vfo_firb_body<WIDTH, STEREO, QUAD>(kidx(blockIdx), smem, (int)blockDim.x, jobs);
}

// Polyphase resampler, register-blocked over one full phase cycle per work-item: outputs n = c*L + r (r = 0..L-1) of cycle c
// use phases (phase0 + r*M) mod L and input offsets c*M + o_r, o_r = (phase0 + r*M) div L — the SAME (phase, o_r) pattern for
// every cycle, so the taps are wave-uniform.  The host tabulates, for every phase0, cyc[m][r] = bank[phase_r][m - o_r] (0
// outside the filter), m = 0 .. tpp + M - 1; a work-item walks its tpp + M inputs once, doing LMAX FMAs (complex: 2x) per read.
struct PolyBJob {
    StreamIn in;
    float2* out;
    const float* cyc;  // [rows][LMAX] for this push's phase0
    int interp, decim, tpp, off0, nout, rows;
};

template <int LMAX, bool LINEAR>
__global__ __launch_bounds__(256) void vfo_polyb_kernel(const PolyBJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float2, xs)
    const PolyBJob& job = jobs[blockIdx.y];
    const int nthreads = blockDim.x;
    const int L = job.interp, M = job.decim, rows = job.rows;
    const int c0 = blockIdx.x * nthreads;  // first cycle of this block
    if (c0 * L >= job.nout) { return; }
    const int P1 = nthreads + rows / M + 2;  // columns per residue row (de-interleaved layout)
    const int first = job.off0 + c0 * M - (job.tpp - 1);
    const int need = (nthreads - 1) * M + rows;
    if constexpr (LINEAR) {
        // odd M: lanes read t*M + m, a stride of 2*M dwords — conflict-free for ds_read_b64 (gcd(2M, 64) = 2), so the tile is
        // stored as is and the row loop needs no address arithmetic
        for (int s = threadIdx.x; s < need; s += nthreads) { xs[s] = stream_load2(job.in, first + s); }
    }
    else {
        for (int s = threadIdx.x; s < M * P1; s += nthreads) {
            const float2 v = (s < need) ? stream_load2(job.in, first + s) : make_float2(0.0f, 0.0f);
            xs[(s % M) * P1 + (s / M)] = v;  // element s of the tile lives at [s mod M][s div M]
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    const UniformF32 cyc = as_uniform(job.cyc);
    float2 acc[LMAX];
#pragma unroll
    for (int r = 0; r < LMAX; r++) { acc[r] = make_float2(0.0f, 0.0f); }
    if constexpr (LINEAR) {
        const float2* xp = xs + t * M;
#pragma unroll 4
        for (int m = 0; m < rows; m++) {
            const float2 x = xp[m];
#pragma unroll
            for (int r = 0; r < LMAX; r++) {
                const float h = cyc[m * LMAX + r];
                acc[r].x = fmaf(h, x.x, acc[r].x);
                acc[r].y = fmaf(h, x.y, acc[r].y);
            }
        }
    }
    else {
        int res = 0, col = t;  // element t*M + m -> residue m mod M, column t + m div M
        for (int m = 0; m < rows; m++) {
            const float2 x = xs[res * P1 + col];
#pragma unroll
            for (int r = 0; r < LMAX; r++) {
                const float h = cyc[m * LMAX + r];
                acc[r].x = fmaf(h, x.x, acc[r].x);
                acc[r].y = fmaf(h, x.y, acc[r].y);
            }
            if (++res == M) { res = 0; col++; }
        }
    }
    const int n0 = (c0 + t) * L;
#pragma unroll
    for (int r = 0; r < LMAX; r++) {
        if (r < L && n0 + r < job.nout) { job.out[n0 + r] = acc[r]; }
    }
}

#include "vfo_front_kernels.h"
#include "vfo_toep_kernels.h"

// =====================================================================================================================
// AF chain: Deemphasis<stereo_t> (filter/deephasis.h:58-77): y[i] = alpha * x[i] + (1 - alpha) * y[i-1] per channel, state carried
// across pushes.  A first-order linear recurrence: one workgroup per VFO walks the push in super chunks of 256 * 8 frames; every
// work-item runs the recursion over its 8 frames from a zero carry, the chunk-end values are combined with a workgroup scan of
// the affine maps (m, a): y_end = m * y_in + a, and each work-item then re-runs the reference's exact expression from its true
// carry-in.  Only the carry-in differs in rounding from the sequential loop (~1e-7 relative; the filter is contractive).
// =====================================================================================================================
struct DeempJob {
    const float2* in;
    float2* out;
    int n;
    float alpha;      // KIND 0: de-emphasis alpha; KIND 1: DC-blocker rate
    const float2* state_in;  // KIND 0: lastOut (deephasis.h:72-73); KIND 1: offset (dc_blocker.h:57) as the block before left it, device resident
    float2* state_out;       // ... as this block leaves it (the host alternates two slots block by block: in pipelined mode pass 1 of block n + 1
                             // runs one launch behind pass 1 of block n and must neither wait for a third launch nor overwrite what is being read)
    float4* seg;      // [nseg] scratch: per segment (m, a.l, a.r, -): state_end = m * state_in + a (two buffers, alternating like the state)
    int nseg;         // segments of SDRPP_DEEMP_SEG frames
    int conj;         // KIND 1: negate the imaginary part of the output (dsp/math/conjugate.h) after the DC blocker
};
#define SDRPP_DEEMP_C 16
#define SDRPP_DEEMP_SEG (256 * SDRPP_DEEMP_C)

// Workgroup-wide composition of the per-work-item affine maps (Hillis-Steele): on return sm_m/sm_a[t] hold the map of work-items
// 0..t applied in order: (m2, a2) o (m1, a1) = (m2*m1, a2 + m2*a1).
__device__ __forceinline__ void deemph_block_scan(float* sm_m, float2* sm_a, int t, float m, float2 e) {
    sm_m[t] = m;
    sm_a[t] = e;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        float pm = 1.0f;
        float2 pa = make_float2(0.0f, 0.0f);
        const bool has = t >= d;
        if (has) {
            pm = sm_m[t - d];
            pa = sm_a[t - d];
        }
        __syncthreads();
        if (has) {
            const float mm = sm_m[t];
            const float2 aa = sm_a[t];
            sm_m[t] = mm * pm;
            sm_a[t] = make_float2(aa.x + mm * pa.x, aa.y + mm * pa.y);
        }
        __syncthreads();
    }
}

// First-order recurrences over a two-channel stream as a two-level scan.
//   KIND 0  Deemphasis<stereo_t>:   y[i] = alpha * x[i] + (1 - alpha) * y[i-1]                       (state = y)
//   KIND 1  DCBlocker<complex_t>:   out[i] = x[i] - off;  off += out[i] * rate   [then optional conj]  (state = off)
// Both states evolve by an affine map per sample (slope 1 - alpha / 1 - rate).
// PASS 0: segment maps from a zero state (grid: x = segment, y = job).  PASS 1: every segment composes the maps of the segments
// before it onto the carried state (a few dozen multiply-adds), then each work-item re-runs the reference's exact expression from
// its true carry-in; vfo_deemph_state_kernel stores the new state.
template <int KIND, int PASS>
__device__ __forceinline__ void vfo_deemph_body(const KIdx bid, float* smem, const DeempJob* __restrict__ jobs) {
    float* sm_m = smem;                                        // [256]
    float2* sm_a = reinterpret_cast<float2*>(smem + 256);      // [256]
    const DeempJob& job = jobs[bid.y];
    const int sg = bid.x;
    if (sg >= job.nseg) { return; }  // (the whole workgroup)
    constexpr int C = SDRPP_DEEMP_C;
    const int t = threadIdx.x;
    const float alpha = job.alpha, beta = 1.0f - alpha;
    const int i0 = sg * SDRPP_DEEMP_SEG + t * C;
    float2 x[C];
    float2 e = make_float2(0.0f, 0.0f);
    float m = 1.0f;
#pragma unroll
    for (int j = 0; j < C; j++) {
        const bool ok = i0 + j < job.n;
        x[j] = ok ? job.in[i0 + j] : make_float2(0.0f, 0.0f);
        if (ok) {
            if constexpr (KIND == 0) {
                e.x = (alpha * x[j].x) + (beta * e.x);
                e.y = (alpha * x[j].y) + (beta * e.y);
            }
            else {
                e.x += (x[j].x - e.x) * alpha;
                e.y += (x[j].y - e.y) * alpha;
            }
            m *= beta;
        }
    }
    deemph_block_scan(sm_m, sm_a, t, m, e);
    if constexpr (PASS == 0) {
        if (t == 255) { job.seg[sg] = make_float4(sm_m[255], sm_a[255].x, sm_a[255].y, 0.0f); }
    }
    else {
        float2 c0 = *job.state_in;  // carry into the push, then through the earlier segments (uniform: every work-item does the same)
        for (int q = 0; q < sg; q++) {
            const float4 g = job.seg[q];
            c0 = make_float2(g.y + g.x * c0.x, g.z + g.x * c0.y);
        }
        float2 y = c0;
        if (t > 0) { y = make_float2(sm_a[t - 1].x + sm_m[t - 1] * c0.x, sm_a[t - 1].y + sm_m[t - 1] * c0.y); }
#pragma unroll
        for (int j = 0; j < C; j++) {
            if (i0 + j < job.n) {
                if constexpr (KIND == 0) {
                    y.x = (alpha * x[j].x) + (beta * y.x);  // deephasis.h:66-69, same expression
                    y.y = (alpha * x[j].y) + (beta * y.y);
                    job.out[i0 + j] = y;
                }
                else {
                    const float2 o = make_float2(x[j].x - y.x, x[j].y - y.y);  // dc_blocker.h:56-57
                    y.x += o.x * alpha;
                    y.y += o.y * alpha;
                    job.out[i0 + j] = make_float2(o.x, job.conj ? -o.y : o.y);
                }
            }
        }
        // the state the NEXT block starts from: lastOut = out[n - 1] (deephasis.h:72-73) resp. the offset after the last sample — the work-item
        // that holds the last sample of the push has it in `y`
        if (sg == job.nseg - 1 && i0 < job.n && i0 + C >= job.n) { *job.state_out = y; }
    }
}
template <int KIND, int PASS>
__global__ __launch_bounds__(256) void vfo_deemph_kernel(const DeempJob* __restrict__ jobs) {
    __shared__ float sm[3 * 256];
    vfo_deemph_body<KIND, PASS>(kidx(blockIdx), sm, jobs);
}
// Conjugate alone (dsp/math/conjugate.h:12-15)
__global__ __launch_bounds__(256) void iq_conjugate_kernel(const float2* __restrict__ in, float2* __restrict__ out, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float2 x = in[i];
        out[i] = make_float2(x.x, -x.y);
    }
}

// =====================================================================================================================
// Polyphase resampler with many phases (the AF chain's 96/125): cycle-major.  A tile = CT whole phase cycles (CT * L outputs,
// CT * M inputs); lane j owns cycle j, a wavefront walks over phases r = w, w + 4, ...: within a wavefront the phase — hence
// the tap row — is uniform (scalar loads) and all L phases reuse ONE LDS window of CT * M + tpp input samples.
// =====================================================================================================================
// `cap2g` = LDS window in float2 (low 24 bits) | phase groups G - 1 (bits 24 ..): a tile's L phases can be dealt out over G workgroups (each loads
// the tile's window and walks phases wv + 4 g, wv + 4 g + 4 G, ...) — what a wavefront does one after the other is L / 4 phases x tpp taps, the
// whole life of the workgroup, and at the reference's block size a block's AF output is 2-3 cycles: 3 busy lanes walking 24 phases x 99 taps.
__device__ __forceinline__ void vfo_polyc_body(const KIdx bid, float2* xsc, const PolyJob* __restrict__ jobs, int cap2g) {
    const PolyJob& job = jobs[bid.y];
    const int L = job.interp, M = job.decim, tpp = job.tpp;
    const int cap2 = cap2g & 0xffffff, G = (cap2g >> 24) + 1;
    int CT = (cap2 - tpp - M) / M;  // cycles per tile: window (CT - 1) * M + o_max + tpp <= cap2, o_max <= M
    if (CT > 64) { CT = 64; }
    const int g = bid.x % G;
    const int c0 = (bid.x / G) * CT;
    if ((long long)c0 * L >= job.nout) { return; }
    const int first = job.off0 + c0 * M - (tpp - 1);
    const int nwin = CT * M + M + tpp;
    for (int s = threadIdx.x; s < nwin; s += 256) { xsc[s] = stream_load2(job.in, first + s); }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = wv + 4 * g; r < L; r += 4 * G) {
        const int A = job.phase0 + r * M, ph = A % L, o = A / L;
        const UniformF32 taps = as_uniform(job.bank + (size_t)ph * tpp);
        const float2* xp = xsc + lane * M + o;
        float2 acc = make_float2(0.0f, 0.0f);
        if (lane < CT) {
            for (int k = 0; k < tpp; k++) {
                const float h = taps[k];
                const float2 x = xp[k];
                acc.x = fmaf(h, x.x, acc.x);
                acc.y = fmaf(h, x.y, acc.y);
            }
            const long long n = (long long)(c0 + lane) * L + r;
            if (n < job.nout) { global_store_f32x2(job.out, n, acc); }
        }
    }
}
__global__ __launch_bounds__(256) void vfo_polyc_kernel(const PolyJob* __restrict__ jobs, int cap2) {
    HIP_DYNAMIC_SHARED(float2, xsc)
    vfo_polyc_body(kidx(blockIdx), xsc, jobs, cap2);
}

}  // namespace sdrpp_k
