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
    return (i >= 0) ? d[i] : h[s.hist_len + i];
}
__device__ __forceinline__ float stream_load1(const StreamIn& s, int i) {
    if (i >= s.n) { return 0.0f; }
    return (i >= 0) ? s.data[i] : s.hist[s.hist_len + i];
}
// The same without a branch, for loops that fetch several samples per lane: the load is unconditional (the address is clamped into the
// stream, the value selected afterwards), so the compiler issues all loads of the loop before the first wait — behind a per-element
// branch every load costs its own memory round trip (measured: 18 x 0.75 us for the first window of the audio filter of a 50 000-sample
// block).  Same values; needs i >= -hist_len like the functions above.
__device__ __forceinline__ float2 stream_load2_nb(const StreamIn& s, int i, bool ok = true) {  // ok false: zero (no load is ever guarded by a branch)
    const bool use = ok && i < s.n, cur = i >= 0;
    int ic = cur ? i : (s.hist_len + i);
    ic = (use && ic >= 0) ? ic : 0;
    const float2 v = global_load_f32x2(reinterpret_cast<const float2*>((cur && use) ? s.data : s.hist), ic);
    return use ? v : make_float2(0.0f, 0.0f);
}
__device__ __forceinline__ float stream_load1_nb(const StreamIn& s, int i, bool ok = true) {
    const bool use = ok && i < s.n, cur = i >= 0;
    int ic = cur ? i : (s.hist_len + i);
    ic = (use && ic >= 0) ? ic : 0;
    const float v = global_load_f32((cur && use) ? s.data : s.hist, ic);
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
template <int VT>
__global__ __launch_bounds__(256) void vfo_stage1_kernel(IqSrc src, const Stage1Job* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float2, xs)
    const Stage1Job& job = jobs[blockIdx.y];
    const int tile = blockDim.x;
    const int j0 = blockIdx.x * tile;
    if (j0 >= job.nout) { return; }
    const int K = job.ntaps, lgD = job.log2_decim, D = 1 << lgD;
    const int extra = (K - 1 + D - 1) >> lgD;
    const int pitch = tile + extra + 1;
    const int nsamp = (tile - 1) * D + K;
    const long long base = (long long)job.off0 + (long long)j0 * D - (K - 1);  // push-relative index of tile sample 0
    for (int s = threadIdx.x; s < nsamp; s += tile) {
        const long long gi = base + s;
        xs[(s & (D - 1)) * pitch + (s >> lgD)] = (gi < job.min_idx) ? make_float2(0.0f, 0.0f) : iq_load_clamped(src, gi);
    }
    __syncthreads();
    const int j = threadIdx.x;
    float2 acc[VT];
#pragma unroll
    for (int v = 0; v < VT; v++) { acc[v] = make_float2(0.0f, 0.0f); }
    stage1_accumulate<VT>(xs, pitch, lgD, K, j, as_uniform(job.ctaps), acc);  // taps are wave-uniform: scalar loads
    if (j0 + j >= job.nout) { return; }
    const double centre = (double)(base + (long long)j * D) + 0.5 * (double)(K - 1);
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

// Large first-stage decimation (D >= 32: the 61.44 MS/s plans decimate by 64 with 257..400 taps).  An LDS tile for even 64
// outputs would be ~36 KiB, leaving one wavefront per SIMD.  Consecutive outputs start D samples apart, so there is almost
// no overlap between neighbouring lanes to exploit anyway: every lane streams its own K contiguous samples straight from
// global memory (each 64-byte line is consumed over 8 iterations and stays in L1), no LDS, full occupancy.  Reuse is across
// the VT VFOs of the work-item, exactly as in the tiled kernel.
template <int VT>
__global__ __launch_bounds__(256) void vfo_stage1_direct_kernel(IqSrc src, const Stage1Job* __restrict__ jobs) {
    const Stage1Job& job = jobs[blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
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
    const long long blk_first = (long long)job.off0 + ((long long)(blockIdx.x * blockDim.x) << lgD) - (K - 1);
    const long long blk_last = (long long)job.off0 + ((long long)min((int)(blockIdx.x * blockDim.x + blockDim.x - 1), job.nout - 1) << lgD);
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
    const double centre = (double)i0 + 0.5 * (double)(K - 1);
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

// =====================================================================================================================
// Reference-rotator mode (sdrpp_set_nco_mode(ctx, 1); parity runs against the reference's CPU path).
//
// The reference's FrequencyXlator (frequency_xlator.h:43-50) calls VOLK's rotator2 once per block: out[i] = in[i] * phase;
// phase *= phaseDelta in float, phase /= |phase| after every 512 samples and at the end of a call with a remainder.  That phase
// sequence drifts from arg(phaseDelta) * i by its own rounding (1e-10 .. 2e-9 rad/sample) and its modulus saw-tooths by up to
// 512 ulp; a product detector (SSB) and the raw IF see both.  The recursion is a strictly sequential float chain, so exactly
// reproducing it costs one dependent complex multiply per input sample and VFO: here ONE LANE per VFO walks the push, all lanes of
// a wavefront share the input samples (64 at a time, one coalesced load, v_readlane broadcast) and the 64 x 64 tile of rotated
// samples goes through LDS so that the stores are coalesced rows.  The rotated stream then feeds the first decimator as a plain
// FIR (the fused translate + filter kernels cannot be used: their NCO is folded into the taps).  ~50 cycles per sample: a few
// times real time at 10 MS/s — a parity mode, not the throughput path.
// `bounds` = cumulative sample counts at which the reference's blocks end inside this push (strictly what its rotator calls saw).
// =====================================================================================================================
struct RotXJob {
    float2* out;    // rotated samples of this push
    float2* state;  // persistent phase (re, im)
    float dr, di;   // phaseDelta (frequency_xlator.h:17)
};
__device__ __forceinline__ void rotator_norm(float& pr, float& pi) {
    // hypotf as glibc evaluates it for floats: sqrt in double of the exactly representable squares' sum, rounded once to float
    const double h2 = ((double)pr * (double)pr) + ((double)pi * (double)pi);
    const float h = (float)sqrt(h2);
    pr = pr / h;
    pi = pi / h;
}
__global__ __launch_bounds__(64) void vfo_rotate_exact_kernel(IqSrc src, const RotXJob* __restrict__ jobs, int njobs, const int* __restrict__ bounds, int nb) {
    HIP_DYNAMIC_SHARED(float2, rot_tile)  // [64 samples][65]: column = VFO (lane)
    const int lane = threadIdx.x;
    const int jid = (int)blockIdx.x * 64 + lane;
    const bool live = jid < njobs;
    const RotXJob job = jobs[live ? jid : njobs - 1];
    float pr = job.state->x, pi = job.state->y;
    const float dr = job.dr, di = job.di;
    const int nrows = min(64, njobs - (int)blockIdx.x * 64);
    int b0 = 0;
    for (int blk = 0; blk < nb; blk++) {
        const int b1 = bounds[blk];
        int since = 0;  // samples since the start of this call (block)
        for (int base = b0; base < b1; base += 64) {
            const int cnt = min(64, b1 - base);
            const float2 xv = (lane < cnt) ? src.cur[base + lane] : make_float2(0.0f, 0.0f);
            for (int i = 0; i < cnt; i++) {
                const float xr = wave_bcast(xv.x, i), xi = wave_bcast(xv.y, i);
                rot_tile[i * 65 + lane] = make_float2((xr * pr) - (xi * pi), (xr * pi) + (xi * pr));
                const float nr = (pr * dr) - (pi * di);
                const float ni = (pr * di) + (pi * dr);
                pr = nr;
                pi = ni;
                since++;
                if ((since & 511) == 0) { rotator_norm(pr, pi); }
            }
            wave_sync();
            for (int r = 0; r < nrows; r++) {
                float2* o = jobs[(int)blockIdx.x * 64 + r].out;
                if (lane < cnt) { o[base + lane] = rot_tile[lane * 65 + r]; }
            }
            wave_sync();
        }
        if ((since & 511) != 0) { rotator_norm(pr, pi); }
        b0 = b1;
    }
    if (live) { *job.state = make_float2(pr, pi); }
}

// The same recursion with the work split over the four wavefronts of a workgroup (round 3): one wavefront per sample would spend ~16
// vector instructions (64 cycles) on it — two broadcasts, the rotation, the LDS write, the phase update — but only the phase update is
// sequential.  Wavefront 0 (lane = VFO) runs NOTHING but the phase chain (the two complex products' four multiplies, a subtraction and an
// addition per sample, renormalised every 512 samples and at every reference-block end exactly like the reference's calls) and leaves
// the 64 phases of a 64-sample chunk in LDS; wavefronts 1-3 (lane = sample) meanwhile apply the PREVIOUS chunk's phases to its samples,
// VFO by VFO, and store coalesced rows.  Two LDS buffers, one workgroup barrier per chunk.  Same operations in the same order on the
// same operands: bit-identical to vfo_rotate_exact_kernel (and to the reference's rotator).
struct RotChunkIt {
    int blk, base, b1, nb;
    UniformI32 bounds;  // (scalar loads: block ends are wave-uniform, and a vector load here would put a `s_waitcnt vmcnt(0)` into the chunk walk)
    __device__ __forceinline__ void settle() {
        while (blk < nb && base >= b1) {
            blk++;
            if (blk < nb) { b1 = bounds[blk]; }
        }
    }
    __device__ __forceinline__ void init(UniformI32 bnd, int n) {
        bounds = bnd;
        nb = n;
        blk = 0;
        base = 0;
        b1 = n > 0 ? bnd[0] : 0;
        settle();
    }
    __device__ __forceinline__ bool valid() const { return blk < nb; }
    __device__ __forceinline__ int cnt() const { return (b1 - base < 64) ? b1 - base : 64; }
    __device__ __forceinline__ bool ends_block() const { return base + cnt() >= b1; }
    __device__ __forceinline__ void advance() {
        base += cnt();
        settle();
    }
};
// (round 3b: the first version of this kernel ran at 63 ns per sample — 150 cycles — instead of the chain's ~16: every consumer row began
// with a load of its output pointer from the job table (a memory round trip each, 15 per chunk and wavefront), the chunk's samples were
// loaded when the chunk began (another round trip, on the critical path of every chunk) and the chain compiled to seven scalar-operand vector
// instructions + a compare-and-branch per sample.  Then: pointers in LDS, the chain as two packed multiplies + one packed add per sample —
// the same IEEE operations on the same operands — in straight-line code per full chunk, the samples requested six chunks ahead by the
// wavefronts that apply the phases: still 27 ns per sample (56 cycles), measured on cfg 4 (profiles/r03q_bench_cfg4_ssb_exact_before.json).
// The ISA said why: loads and stores share ONE in-order counter (vmcnt) on gfx9, the applying wavefronts issue a data-dependent number of
// stores between a request and its use, so the compiler can only wait with vmcnt(0) — every chunk waited for its own newest stores and
// for all six requests in flight.  Round 3c: the wavefront that runs the chain is the only one that LOADS (its waits are exact counts: it
// never stores) and hands a chunk's samples over in LDS next to the phases; the applying wavefronts only read LDS and store, and never
// wait for memory at all.)
// (and the two roles are two separate loops, each with its own barriers: in ONE loop with a branch per role the compiler's wait-count
// analysis, which does not know that a wavefront keeps its role, merges "this register has a request in flight" with "any number of stores
// have been issued since" and falls back to vmcnt(0) again)
typedef float rot_v2f __attribute__((vector_size(8)));
#define SDRPP_ROTX4_TRIP 4  // chunks per request round of the chain wavefront
// (SKIP, a template parameter: in a full chunk the chain publishes every SKIP-th phase; the applying wavefronts take the steps in between themselves)
#define SDRPP_ROTX4_LDS_BYTES ((size_t)2 * 64 * 65 * sizeof(float2) + (size_t)2 * SDRPP_ROTX4_TRIP * 64 * sizeof(float2) + 64 * sizeof(float2*) + 64 * sizeof(float2) + 2 * sizeof(int))
template <int SKIP>
__global__ __launch_bounds__(256) void vfo_rotate_exact4_kernel(IqSrc src, const RotXJob* __restrict__ jobs, int njobs, const int* __restrict__ bounds_g, int nb, int vpw) {
    HIP_DYNAMIC_SHARED(float2, ph_tile)  // [2][64 samples][65]: column = VFO; then [2 * TRIP][64] samples; then the 64 output pointers
    constexpr int TRIP = SDRPP_ROTX4_TRIP;
    float2* x_tile = ph_tile + (size_t)2 * 64 * 65;
    float2** outp = reinterpret_cast<float2**>(x_tile + 2 * TRIP * 64);
    float2* dtab = reinterpret_cast<float2*>(outp + 64);  // phaseDelta of the workgroup's VFOs
    int* sparse = reinterpret_cast<int*>(dtab + 64);       // [2]: the chunk in this buffer carries every SKIP-th phase only
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wv = wave_uniform(tid >> 6);  // (known to be uniform: the roles are scalar branches and the chunk walk stays in scalar registers)
    const int j0 = (int)blockIdx.x * vpw;  // vpw <= 64 VFOs per workgroup (the host's choice: see rot_exact_vpw)
    const int nrows = min(vpw, njobs - j0);
    const UniformI32 bounds = as_uniform_i32(bounds_g);
    RotChunkIt cit;  // the chunk every wavefront of the workgroup is at (one barrier per chunk)
    cit.init(bounds, nb);
    if (!cit.valid()) { return; }  // (uniform; no barrier has been passed)
    if (wv == 0) {
        // ---- the chain: lane = VFO ----
        const bool live = lane < nrows;
        const RotXJob job = jobs[live ? j0 + lane : njobs - 1];
        outp[lane] = job.out;
        dtab[lane] = make_float2(job.dr, job.di);
        rot_v2f p = { job.state->x, job.state->y };
        const rot_v2f d = { job.dr, job.di }, dyxn = { -job.di, job.dr };
        int since = 0;  // samples since the start of the reference block the producer is in
        RotChunkIt pit, fit;  // phases (one chunk ahead of the consumers), sample requests (one to two rounds ahead)
        pit.init(bounds, nb);
        fit.init(bounds, nb);
        auto step = [&](float2* slot) {
            *slot = make_float2(p[0], p[1]);
            const rot_v2f pxx = { p[0], p[0] }, pyy = { p[1], p[1] };
            const rot_v2f a = pxx * d;      // (pr * dr, pr * di)                  = (a0, a1)
            const rot_v2f b = pyy * dyxn;   // (pi * -di, pi * dr) = (-(pi * di), b1) = (-b0, b1): a product's sign does not touch its rounding
            p = a + b;                      // (a0 - b0, a1 + b1): x + (-y) is x - y bit for bit
        };
        auto norm = [&]() {
            float pr = p[0], pi = p[1];
            rotator_norm(pr, pi);
            p = rot_v2f{ pr, pi };
        };
        auto produce = [&](int buf) {
            float2* ph = ph_tile + (size_t)buf * 64 * 65 + lane;
            const int cnt = pit.cnt();
            if (cnt == 64 && (since & 63) == 0) {  // a full chunk between two possible renormalisation points: straight-line code
                // Handing a phase over costs the chain ~12 cycles on top of its own ~21 per sample (the LDS write's operands go through the
                // same register read port as the arithmetic, wherever the write is placed: tools/probe/chain_latency_probe.hip, variants A / E /
                // F), so only every SKIP-th phase is handed over; the applying wavefronts take the up to SKIP - 1 steps in between
                // themselves — the same two products and one sum on the same operands, so the same bits.  (No renormalisation can fall
                // inside such a chunk: it starts a multiple of 64 samples into its block.)
#pragma unroll
                for (int i = 0; i < 64; i++) {
                    if (i % SKIP == 0) { ph[i * 65] = make_float2(p[0], p[1]); }
                    const rot_v2f pxx = { p[0], p[0] }, pyy = { p[1], p[1] };
                    const rot_v2f a = pxx * d, b = pyy * dyxn;
                    p = a + b;
                }
                if (lane == 0) { sparse[buf] = 1; }
                since += 64;
                if ((since & 511) == 0) { norm(); }
            }
            else {
                for (int i = 0; i < cnt; i++) {
                    step(ph + i * 65);
                    since++;
                    if ((since & 511) == 0) { norm(); }
                }
                if (lane == 0) { sparse[buf] = 0; }
            }
            if (pit.ends_block()) {
                if ((since & 511) != 0) { norm(); }
                since = 0;
            }
            pit.advance();
        };
        // Samples are requested a ROUND of TRIP chunks at a time, one to two rounds before the consumers reach them (a chunk lasts ~0.5 us at
        // the chain's pace, a first touch of the input in HBM ~2 us), always all TRIP requests — past the end of the push with a clamped
        // address — and handed over in LDS at the top of the next round: by then they have long landed.
        auto fetch = [&]() -> float2 {
            const bool ok = fit.valid() && lane < fit.cnt();
            const float2 v = global_load_f32x2(src.cur, ok ? (long long)(fit.base + lane) : 0ll);
            if (fit.valid()) { fit.advance(); }
            return v;
        };
        float2 xr[TRIP];
#pragma unroll
        for (int k = 0; k < TRIP; k++) { xr[k] = fetch(); }  // round 0
#pragma unroll
        for (int k = 0; k < TRIP; k++) { x_tile[k * 64 + lane] = xr[k]; }
#pragma unroll
        for (int k = 0; k < TRIP; k++) { xr[k] = fetch(); }  // round 1
        produce(0);
        __syncthreads();
        int buf = 0, half = 0;
        while (cit.valid()) {
            half ^= 1;
#pragma unroll
            for (int k = 0; k < TRIP; k++) { x_tile[(half * TRIP + k) * 64 + lane] = xr[k]; }  // the round after the one being consumed
#pragma unroll
            for (int k = 0; k < TRIP; k++) { xr[k] = fetch(); }                               // the round after that
#pragma unroll
            for (int k = 0; k < TRIP; k++) {
                if (cit.valid()) {  // (uniform over the workgroup: the consumers walk the same chunks)
                    if (pit.valid()) { produce(buf ^ 1); }
                    cit.advance();
                    buf ^= 1;
                    __syncthreads();
                }
            }
        }
        if (live) { *job.state = make_float2(p[0], p[1]); }
    }
    else {
        // ---- the consumers: lane = sample of the chunk; they read LDS and store, nothing else ----
        __syncthreads();
        int buf = 0, slot = 0;
        while (cit.valid()) {
            const int cnt = cit.cnt(), base = cit.base;
            const float2 x = x_tile[slot * 64 + lane];
            if (wave_uniform(sparse[buf]) != 0) {  // a full chunk with every SKIP-th phase: lane i starts from phase i - i % SKIP and takes i % SKIP steps
                const float2* ph = ph_tile + (size_t)buf * 64 * 65 + (lane & ~(SKIP - 1)) * 65;
                const int more = lane & (SKIP - 1);
#pragma unroll 2
                for (int r = wv - 1; r < nrows; r += 3) {
                    const float2 p0 = ph[r], dv = dtab[r];
                    rot_v2f q = { p0.x, p0.y };
                    const rot_v2f d = { dv.x, dv.y }, dyxn = { -dv.y, dv.x };
#pragma unroll
                    for (int st = 0; st < SKIP - 1; st++) {
                        const rot_v2f qxx = { q[0], q[0] }, qyy = { q[1], q[1] };
                        const rot_v2f a = qxx * d, b = qyy * dyxn;
                        const rot_v2f n = a + b;
                        q = (st < more) ? n : q;
                    }
                    float2* o = outp[r];
                    global_store_f32x2(o, base + lane, make_float2((x.x * q[0]) - (x.y * q[1]), (x.x * q[1]) + (x.y * q[0])));
                }
            }
            else if (lane < cnt) {
                const float2* ph = ph_tile + (size_t)buf * 64 * 65 + lane * 65;
#pragma unroll 4
                for (int r = wv - 1; r < nrows; r += 3) {
                    const float2 ph_r = ph[r];
                    float2* o = outp[r];
                    global_store_f32x2(o, base + lane, make_float2((x.x * ph_r.x) - (x.y * ph_r.y), (x.x * ph_r.y) + (x.y * ph_r.x)));
                }
            }
            cit.advance();
            buf ^= 1;
            slot = (slot + 1) & (2 * TRIP - 1);
            __syncthreads();
        }
    }
}

// SSB's second translation (ssb.h:78, a FrequencyXlator at the IF rate) in reference-rotator mode: one wavefront per VFO, every lane
// evaluates the same (uniform) recursion, lane i keeps Re{x[i] * phase} of sample i of the 64-sample chunk.
struct SsbRotXJob {
    const float2* in;
    float* out;     // Re{} of the rotated samples (ComplexToReal, ssb.h:81-88)
    float2* state;
    float dr, di;
    const int* bounds;
    int nb;
};
__global__ __launch_bounds__(64) void vfo_ssb_rotate_exact_kernel(const SsbRotXJob* __restrict__ jobs) {
    const SsbRotXJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    float pr = job.state->x, pi = job.state->y;
    int b0 = 0;
    for (int blk = 0; blk < job.nb; blk++) {
        const int b1 = job.bounds[blk];
        int since = 0;
        for (int base = b0; base < b1; base += 64) {
            const int cnt = min(64, b1 - base);
            const float2 xv = (lane < cnt) ? job.in[base + lane] : make_float2(0.0f, 0.0f);
            float mine = 0.0f;
            for (int i = 0; i < cnt; i++) {
                const float xr = wave_bcast(xv.x, i), xi = wave_bcast(xv.y, i);
                const float re = (xr * pr) - (xi * pi);
                if (lane == i) { mine = re; }
                const float nr = (pr * job.dr) - (pi * job.di);
                const float ni = (pr * job.di) + (pi * job.dr);
                pr = nr;
                pi = ni;
                since++;
                if ((since & 511) == 0) { rotator_norm(pr, pi); }
            }
            if (lane < cnt) { job.out[base + lane] = mine; }
        }
        if ((since & 511) != 0) { rotator_norm(pr, pi); }
        b0 = b1;
    }
    if (lane == 0) { *job.state = make_float2(pr, pi); }
}

// =====================================================================================================================
// Retune hand-over of the closed-form NCO (RxVFO::setOffset, rx_vfo.h:72-77).  In the reference only phaseDelta changes: the
// samples already in the first decimator's delay line stay rotated with the OLD increment, the phase is continuous.  The fused
// translate + filter kernels rotate a whole filter window with ONE increment, so the first outputs after a retune — those whose
// window still reaches in front of the retune point — are recomputed here sample by sample with a piecewise phase
//     phase(n) = seg[s].phi + seg[s].theta * (n - seg[s].start),  seg[s].start <= n < seg[s + 1].start   (turns, push-relative n)
// and overwrite what the front kernel wrote.  A handful of outputs per retuned VFO; later stages are linear and need nothing.
// =====================================================================================================================
#define SDRPP_RETUNE_MAX_SEG 4
struct RetuneJob {
    float2* out;          // first-stage (or composite) output stream of this push
    const float* taps;    // [K] real taps of the (composite) filter
    int K, log2_decim;
    int off;              // push-relative IQ index of tap 0 of output 0
    int nfix;             // outputs 0 .. nfix-1 are recomputed
    int min_idx;          // IQ samples before this index read as zero
    int nseg;
    int start[SDRPP_RETUNE_MAX_SEG];  // ascending; segment 0 covers everything before start[1]
    double theta[SDRPP_RETUNE_MAX_SEG];
    double phi[SDRPP_RETUNE_MAX_SEG];
};
__global__ __launch_bounds__(64) void vfo_retune_fix_kernel(IqSrc src, const RetuneJob* __restrict__ jobs) {
    const RetuneJob& job = jobs[blockIdx.y];
    const int m = (int)blockIdx.x;
    if (m >= job.nfix) { return; }
    const int lane = threadIdx.x;
    const long long i0 = (long long)job.off + ((long long)m << job.log2_decim);
    float ar = 0.0f, ai = 0.0f;
    for (int k = lane; k < job.K; k += 64) {
        const long long n = i0 + k;
        if (n < job.min_idx) { continue; }
        int s = 0;
        for (int q = 1; q < job.nseg; q++) {
            if (n >= job.start[q]) { s = q; }
        }
        double ph = fma((double)(n - job.start[s]), job.theta[s], job.phi[s]);
        ph -= rint(ph);
        float sn, cs;
        sincospif(2.0f * (float)ph, &sn, &cs);
        const float2 x = iq_load_clamped(src, n);
        const float h = job.taps[k];
        const float rr = (x.x * cs) - (x.y * sn), ri = (x.x * sn) + (x.y * cs);
        ar = fmaf(h, rr, ar);
        ai = fmaf(h, ri, ai);
    }
    ar = wave_sum(ar);
    ai = wave_sum(ai);
    if (lane == 0) { job.out[m] = make_float2(ar, ai); }
}

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

__global__ __launch_bounds__(256) void vfo_poly_kernel(const PolyJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float2, xs)
    const PolyJob& job = jobs[blockIdx.y];
    const int tile = blockDim.x;
    const int n0 = blockIdx.x * tile;
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
        const float2 x = job.in[i];
        if (job.mode == 2) { job.out[i] = sqrtf((x.x * x.x) + (x.y * x.y)); }
        else {
            double ph = fma((double)i, job.theta2, job.phi2);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            job.out[i] = fmaf(x.x, cs, -(x.y * sn));
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
__device__ __forceinline__ void carry_body(const KIdx bid, const KIdx gdim, const CarryJob* __restrict__ jobs) {
    const CarryJob job = jobs[bid.y];
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
    const int nthreads = gdim.x * 256, t = bid.x * 256 + (int)threadIdx.x;
    for (int e = first + t; e < first4 && e < total; e += nthreads) {  // (the up to three elements in front of the first whole quad)
        const long long sx = nw + e;
        job.new_hist[e] = global_load_f32(e < eb ? job.old_hist : job.data, e < eb ? sx : (long long)e - ebl);
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
                for (int k = 0; k < 4 && e + k < total; k++) { job.new_hist[e + k] = w4[k]; }
            }
        }
    }
}
__global__ __launch_bounds__(256) void carry_kernel(const CarryJob* __restrict__ jobs) { carry_body(kidx(blockIdx), kidx(gridDim), jobs); }

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
__global__ __launch_bounds__(256) void vfo_fir_direct_kernel(const FirBJob* __restrict__ jobs) {
    const FirBJob& job = jobs[blockIdx.y];
    const int D = 1 << job.log2_decim, kp = job.kp_pad;
    for (int j = (int)(blockIdx.x * blockDim.x + threadIdx.x); j < job.nout; j += (int)(gridDim.x * blockDim.x)) {
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
        for (int s = threadIdx.x; s <= nvalid; s += nall) {
            const float2 x = stream_load2(job.in, base - 1 + s);
            phase[s] = fm_phase(x.y, x.x);
        }
        __syncthreads();
        for (int s = threadIdx.x; s < ncomp; s += nall) {
            const float v = (s < nvalid) ? normalize_phase(phase[s + 1] - phase[s]) * job.inv_deviation : 0.0f;
            xs[(s & (R - 1)) * P1 + (s >> 3)] = v;
        }
    }
    else {
        for (int s = threadIdx.x; s < ncomp * D; s += nall) {
            const int p = s & (D - 1), e = s >> lgD;
            T v;
            if constexpr (WIDTH == 2) { v = (s < nvalid) ? stream_load2(job.in, base + s) : make_float2(0.0f, 0.0f); }
            else { v = (s < nvalid) ? stream_load1(job.in, base + s) : 0.0f; }
            xs[p * P2 + (e & (R - 1)) * P1 + (e >> 3)] = v;
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
            if constexpr (WIDTH == 2) { reinterpret_cast<float2*>(job.out)[jo + r] = acc[r]; }
            else if constexpr (STEREO) { reinterpret_cast<float2*>(job.out)[jo + r] = make_float2(acc[r], acc[r]); }
            else { job.out[jo + r] = acc[r]; }
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

// =====================================================================================================================
// Fused front: stage 1 (translation folded into the first decimating FIR) + stage 2 (second decimating FIR) in ONE kernel.
// The stage-1 outputs of a tile never leave the CU: they go to an LDS buffer and are consumed by stage 2 right away, which
// removes the largest intermediate stream of the cascade (V * P/D1 complex samples written and read back per push).
// Each block produces T2 stage-2 outputs per VFO from TS1 = (T2-1)*D2 + K2 <= blockDim stage-1 outputs; the K2-1 overlap
// between neighbouring tiles is recomputed (a few %).  Stage-1 outputs with negative index (the delay line of stage 2) are
// recomputed from the IQ history instead of being stored, so the only state is the IQ history and the integer offsets.
// =====================================================================================================================
struct Front2Job {
    int nv;
    int ntaps1, log2_decim1, off1;   // stage 1 (decimating_fir.h:51-62 state `offset`)
    int ntaps2, log2_decim2, off2;   // stage 2
    int nout2;                       // stage-2 outputs of this push
    int t2;                          // stage-2 outputs per block
    int min_idx;                     // IQ samples before this push-relative index read as zero
    const float2* ctaps;             // [(ntaps1+1)/2][VT] modulated stage-1 tap pairs
    const float2* ptab;              // [256][VT] exp(j*2*pi*theta_v*D1*j): NCO advance inside a tile (host, double -> float)
    const float* taps2;              // [ntaps2] real taps, natural order
    double theta[SDRPP_S1_MAX_VT];
    double phi0[SDRPP_S1_MAX_VT];
    float2* out[SDRPP_S1_MAX_VT];    // stage-2 output arrays
};

// NCO bookkeeping: the phasor of stage-1 output j of a tile is P_tile * ptab[j], P_tile = exp(j*2*pi*(phi0 + theta*(base + kc)))
// evaluated once per block and VFO in double precision.  ptab[j] is applied to the stage-1 output; P_tile is constant over the
// tile, so by linearity it is applied AFTER stage 2 (T2 instead of 256 complex multiplies per VFO).
template <int VT, int K1S, int LGD1S>  // K1S > 0: stage-1 geometry known at compile time (fully unrolled)
__global__ __launch_bounds__(256, 8) void vfo_front2_kernel(IqSrc src, const Front2Job* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float2, smem2)
    const Front2Job& job = jobs[blockIdx.y];
    constexpr int tile = 256;  // stage-1 outputs computed per block (one per work-item); blockDim.x == 256
    const int T2 = job.t2;
    const int j2_0 = blockIdx.x * T2;
    if (j2_0 >= job.nout2) { return; }
    const int K1 = (K1S > 0) ? K1S : job.ntaps1, lgD1 = (K1S > 0) ? LGD1S : job.log2_decim1, D1 = 1 << lgD1;
    const int K2 = job.ntaps2, lgD2 = job.log2_decim2, D2 = 1 << lgD2;
    const int extra = (K1 - 1 + D1 - 1) >> lgD1;
    const int pitch = tile + extra + 1;
    // LDS is used twice: first as the [D1][pitch] de-interleaved IQ tile, then (after a barrier) as the [VT][s1p] stage-1 output
    // buffer — halving the footprint doubles the number of resident wavefronts that hide the tile-load latency.
    constexpr int s1p = tile + 16;         // row pitch; the slack absorbs stage-2 reads of (unused) lanes past the tile
    const int region = max(D1 * pitch, VT * s1p);
    float2* xs = smem2;
    float2* s1 = smem2;
    float2* ptile = smem2 + region;        // [VT] tile phasors
    // first stage-1 output index this block needs (relative to the push's stage-1 output sequence; may be negative)
    const int i1_0 = job.off2 + j2_0 * D2 - (K2 - 1);
    const long long base = (long long)job.off1 + (long long)i1_0 * D1 - (K1 - 1);
    const int nsamp = (tile - 1) * D1 + K1;
    if (base >= 0 && base >= job.min_idx && base + nsamp <= src.n_cur) {
        const float2* p = src.cur + base;  // whole tile inside this push: plain coalesced loads
        for (int s = threadIdx.x; s < nsamp; s += tile) { xs[(s & (D1 - 1)) * pitch + (s >> lgD1)] = p[s]; }
    }
    else {
        for (int s = threadIdx.x; s < nsamp; s += tile) {
            const long long gi = base + s;
            xs[(s & (D1 - 1)) * pitch + (s >> lgD1)] = (gi < job.min_idx) ? make_float2(0.0f, 0.0f) : iq_load_clamped(src, gi);
        }
    }
    if (threadIdx.x < VT && (int)threadIdx.x < job.nv) {
        const int v = threadIdx.x;
        double ph = fma((double)base + 0.5 * (double)(K1 - 1), job.theta[v], job.phi0[v]);
        ph -= rint(ph);
        float sn, cs;
        sincospif(2.0f * (float)ph, &sn, &cs);
        ptile[v] = make_float2(cs, sn);
    }
    __syncthreads();
    const int j = threadIdx.x;
    {
        float2 acc[VT];
#pragma unroll
        for (int v = 0; v < VT; v++) { acc[v] = make_float2(0.0f, 0.0f); }
        if constexpr (K1S > 0) { stage1_accumulate_static<VT, K1S, LGD1S>(xs, pitch, j, as_uniform(job.ctaps), acc); }
        else { stage1_accumulate<VT>(xs, pitch, lgD1, K1, j, as_uniform(job.ctaps), acc); }
        const float2* __restrict__ pt = job.ptab + (size_t)j * VT;
#pragma unroll
        for (int v = 0; v < VT; v++) {
            const float2 t = pt[v];
            acc[v] = make_float2(fmaf(acc[v].x, t.x, -(acc[v].y * t.y)), fmaf(acc[v].x, t.y, acc[v].y * t.x));
        }
        __syncthreads();  // every work-item is done reading the IQ tile: the region becomes the stage-1 buffer
#pragma unroll
        for (int v = 0; v < VT; v++) { s1[v * s1p + j] = acc[v]; }
    }
    __syncthreads();
    // stage 2: out2[j2] = P_tile * sum_k taps2[k] * s1[(j2 - j2_0) * D2 + k]; 256/VT lanes per VFO, outputs strided by that
    constexpr int LPV = tile / VT;
    constexpr int RB = (128 + LPV - 1) / LPV;  // T2 <= 123 because D2 >= 2
    const int v = threadIdx.x / LPV, l = threadIdx.x % LPV;
    int n2 = job.nout2 - j2_0;
    if (n2 > T2) { n2 = T2; }
    if (v < job.nv) {
        const UniformF32 h2 = as_uniform(job.taps2);
        const float2* sp = s1 + v * s1p + (l << lgD2);
        float2 a[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) { a[r] = make_float2(0.0f, 0.0f); }
        for (int k = 0; k < K2; k++) {
            const float h = h2[k];
#pragma unroll
            for (int r = 0; r < RB; r++) {
                const int idx = ((r * LPV) << lgD2) + k;
                const float2 x = sp[idx];  // lanes past n2 read stale LDS (inside the padded row) and are never stored
                a[r].x = fmaf(h, x.x, a[r].x);
                a[r].y = fmaf(h, x.y, a[r].y);
            }
        }
        const float2 P = ptile[v];
        float2* o = job.out[v] + j2_0;
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const int jj = l + r * LPV;
            if (jj < n2) { o[jj] = make_float2(fmaf(a[r].x, P.x, -(a[r].y * P.y)), fmaf(a[r].x, P.y, a[r].y * P.x)); }
        }
    }
}

// =====================================================================================================================
// Matrix-core front end for banks of >= 17 VFOs that share one decimation geometry.
//
// Stage 1 (translation + first decimating FIR) and stage 2 (second decimating FIR) are both linear and time invariant up to the
// NCO phasor, so their cascade is ONE decimating FIR with the composite taps h12 = h1 (*) upsample(h2, D1), K = K1 + (K2-1)*D1
// taps, decimation D = D1*D2 — again linear phase, so the tap-pair form of stage1_accumulate applies:
//     y[v][n] = sum_p  gr[v][p] * sr[p][n] - gi[v][p] * di[p][n]          (real part; sr/di = pair sums / differences of the IQ tile)
//               sum_p  gr[v][p] * si[p][n] + gi[v][p] * dr[p][n]          (imaginary part)
// which is a matrix product with M = 32 VFOs, N = 32 consecutive outputs and K = 2 per tap pair: exactly one
// v_mfma_f32_32x32x2_f32 per tap pair and component.  Evaluating the cascade at its OUTPUT rate costs ~1.3x the multiply-adds of
// the two-stage form, but they run on the otherwise idle matrix cores at 4x the VALU rate, and the intermediate stream, its LDS
// buffer, the second filter loop and two of the barriers disappear: the kernel is a pure LDS -> MFMA stream.
//   A operand (taps):  [pair][64] table in LDS; lane l supplies (l < 32 ? gr : -gi) of VFO l & 31.
//   B operand (data):  the IQ tile lives in two skewed planes (index i + i / D: a lane stride of D samples becomes the odd stride
//                      D + 1, conflict free); lanes 0-31 read the real plane where lanes 32-63 read the imaginary one, four
//                      ds_read_b32 (two ds_read2_b32 when the geometry is a template constant) per pair.
//   D (results):       lane l holds output n = l & 31 of 16 VFOs; NCO phasor = tile phasor (double precision, once per tile and
//                      VFO) x in-tile table entry (registers, loaded once per block); stores are coalesced along n.
// A block walks over `tiles_per_block` consecutive tiles and prefetches the next IQ tile into registers while the matrix cores
// work on the current one.  Rounding differs from the two-stage reference only in the order of the f32 accumulations (the
// intermediate stream is never rounded to f32) — far inside the 1e-5 RMS bar, see tests/test_parity_vfo.py.
// =====================================================================================================================
#define SDRPP_FCM_VT 32
#define SDRPP_FCM_TILE 32   // outputs per wavefront tile (the N of the matrix instruction)
struct FrontCMJob {
    int nv;
    int ntaps;        // composite K
    int log2_decim;   // log2(D1 * D2)
    int off;          // push-relative IQ index of tap 0 of output 0 (negative: history)
    int nout;         // outputs of this push (= stage-2 outputs)
    int min_idx;      // IQ samples before this push-relative index read as zero
    int tiles_per_wave;
    const float* atab;    // [npad][64]: lane l -> (l < 32 ? gr : -gi) of VFO l & 31 (0 for unused VFO slots and padding rows)
    const float2* ptab;   // [32][32] exp(j*2*pi*theta_v*D*n): NCO advance inside a tile
    double theta[SDRPP_FCM_VT];
    double phi0[SDRPP_FCM_VT];
    float2* out[SDRPP_FCM_VT];
};

// floats per skewed IQ plane; == 32 (mod 64) so that the two planes sit on complementary halves of the 64 LDS banks
__host__ __device__ inline int frontcm_plane(int nsamp, int lgD) {
    const int sk = nsamp + (nsamp >> lgD) + 1;
    return ((sk + 31) / 64) * 64 + 32;
}
// LDS map (float offsets): [4 wavefronts x (XR, XI planes) | tap operand table | 4 x 32 tile phasors | 32 output pointers]
struct FCMLayout { int pl, a_off, pt_off, out_off, total; };
__host__ __device__ inline FCMLayout frontcm_layout(int K, int lgD) {
    FCMLayout L;
    const int nsamp = (SDRPP_FCM_TILE - 1) * (1 << lgD) + K;
    const int np4 = ((((K + 1) >> 1) + 3) >> 2) << 2;
    L.pl = frontcm_plane(nsamp, lgD);
    L.a_off = 4 * 2 * L.pl;
    L.pt_off = L.a_off + np4 * 64;
    L.out_off = L.pt_off + 4 * SDRPP_FCM_VT * 2;
    L.total = L.out_off + SDRPP_FCM_VT * 2;
    return L;
}

// Every WAVEFRONT is an independent tile engine: it owns two skewed IQ planes in LDS, walks over `tiles_per_wave` consecutive
// 32-output tiles and never meets a workgroup barrier after the prologue (the four wavefronts of a block only share the tap
// table), so the matrix pipe of a SIMD always has several unsynchronised wavefronts to pick from.
// PF: IQ samples prefetched per lane (>= ceil(nsamp / 64)); KS > 0: geometry known at compile time (fully unrolled matrix loop:
// every LDS offset is an immediate, the pair reads fuse into ds_read2_b32 and no scalar index arithmetic is left)
template <int PF, int KS, int LGDS>
__device__ __forceinline__ void vfo_frontcm_body(const KIdx bid, float* smemf, const IqSrc& src, const FrontCMJob* __restrict__ jobs) {
    const FrontCMJob& job = jobs[bid.y];
    constexpr int tile = SDRPP_FCM_TILE, VT = SDRPP_FCM_VT;
    const int K = (KS > 0) ? KS : job.ntaps, lgD = (KS > 0) ? LGDS : job.log2_decim, D = 1 << lgD;
    const int NP = (K + 1) >> 1, NP4 = ((NP + 3) >> 2) << 2;
    const bool odd = (K & 1) != 0;
    const int nsamp = (tile - 1) * D + K;
    const FCMLayout L = frontcm_layout(K, lgD);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, jl = lane & 31, hi = lane >> 5;
    float* XR = smemf + wv * 2 * L.pl;
    float* XI = XR + L.pl;
    float* AL = smemf + L.a_off;
    float2* ptile = reinterpret_cast<float2*>(smemf + L.pt_off) + wv * VT;  // [VT], private to the wavefront
    float2** outp = reinterpret_cast<float2**>(smemf + L.out_off);        // [VT]

    const int tile0 = (bid.x * 4 + wv) * job.tiles_per_wave;
    const bool has_tiles = tile0 * tile < job.nout;
    int ntl = (job.nout - tile0 * tile + tile - 1) / tile;  // tiles this wavefront really has
    if (ntl > job.tiles_per_wave) { ntl = job.tiles_per_wave; }

    auto tile_base = [&](int tb) -> long long { return (long long)job.off + (long long)tb * tile * D; };
    float2 pf[PF];
    auto fetch = [&](long long base) {
        if (base >= 0 && base >= job.min_idx && base + nsamp <= src.n_cur) {
            const float2* p = src.cur + base;
#pragma unroll
            for (int q = 0; q < PF; q++) {
                const int sidx = lane + q * 64;
                pf[q] = (sidx < nsamp) ? p[sidx] : make_float2(0.0f, 0.0f);
            }
        }
        else {
#pragma unroll
            for (int q = 0; q < PF; q++) {
                const int sidx = lane + q * 64;
                const long long gi = base + sidx;
                pf[q] = iq_load_nb(src, gi, sidx < nsamp && gi >= job.min_idx);
            }
        }
    };
    float2* X2 = reinterpret_cast<float2*>(XR);  // KS > 0: ONE skewed plane of complex samples in the same 2 * pl floats
    auto planes_store = [&]() {
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const int sidx = lane + q * 64;
            if (sidx < nsamp) {
                const int idx = sidx + (sidx >> lgD);
                if constexpr (KS > 0) { X2[idx] = pf[q]; }
                else {
                    XR[idx] = pf[q].x;
                    XI[idx] = pf[q].y;
                }
            }
        }
    };
    auto tile_phasor = [&](int tb) {
        if (lane < VT && lane < job.nv) {
            double ph = fma((double)tile_base(tb) + 0.5 * (double)(K - 1), job.theta[lane], job.phi0[lane]);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            ptile[lane] = make_float2(cs, sn);
        }
    };

    // ---- wavefront prologue: this lane's slice of the in-tile NCO table, first IQ tile ----
    float2 pt[16];
    auto wave_prologue = [&]() {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int v = (r & 3) + 8 * (r >> 2) + 4 * hi;
            pt[r] = global_load_f32x2(job.ptab, v * tile + jl);
        }
        fetch(tile_base(tile0));
    };
#ifdef SDRPP_FCM_EARLY_IQ
    // measurement build: the first IQ tile is requested BEFORE the tap table (at 10^6-sample blocks ~600 workgroups start together and the first
    // tile arrived ~10 us into a front-end workgroup's life, behind everybody's table and window requests)
    if (has_tiles) { wave_prologue(); }
#endif
    // ---- block prologue: tap operand table and output pointers (the only workgroup barrier of the kernel) ----
    if constexpr (KS > 0) {
        // pair-per-half form (see the matrix loop): lane (jl, hi) wants (gr, -gi) of VFO jl and pair 2 q + hi as ONE 8-byte read — row p of the
        // host's table ([pair][64]: gr of the 32 VFOs, then -gi) goes into LDS with its two halves interleaved.  A task = 4 VFOs of one pair:
        // two 16-byte loads, two 16-byte LDS writes; all loads of a work-item in flight before its first LDS write.
        constexpr int NPS = (KS + 1) / 2, NT = NPS * 8, NB = (NT + 255) / 256;
        const float4* at4 = reinterpret_cast<const float4*>(job.atab);
        float4* AL4 = reinterpret_cast<float4*>(AL);
        float4 tr[NB], ti[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int t = min(tid + q * 256, NT - 1), pr = t >> 3, j4 = t & 7;  // (index clamped, never a guarded load)
            tr[q] = global_load_f32x4(at4, pr * 16 + j4);
            ti[q] = global_load_f32x4(at4, pr * 16 + 8 + j4);
        }
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int t = tid + q * 256;
            if (t < NT) {
                const int pr = t >> 3, j4 = t & 7;
                AL4[pr * 16 + 2 * j4] = make_float4(tr[q].x, ti[q].x, tr[q].y, ti[q].y);
                AL4[pr * 16 + 2 * j4 + 1] = make_float4(tr[q].z, ti[q].z, tr[q].w, ti[q].w);
            }
        }
    }
    else {   // (all loads of a work-item in flight before the first LDS write: a wait per load is a memory round trip each — 8 of them measured)
        constexpr int NB = 5;  // 68 pairs x 64 lanes = 17 floats per work-item: four rounds of 16-byte loads + a rest
        const int n4 = NP4 * 16;
        const float4* at4 = reinterpret_cast<const float4*>(job.atab);
        float4* AL4 = reinterpret_cast<float4*>(AL);
        for (int i0 = tid; i0 < n4; i0 += 256 * NB) {
            float4 tv[NB];
#pragma unroll
            for (int q = 0; q < NB; q++) { tv[q] = global_load_f32x4(at4, min(i0 + q * 256, n4 - 1)); }  // (index clamped, never a guarded load)
#pragma unroll
            for (int q = 0; q < NB; q++) {
                if (i0 + q * 256 < n4) { AL4[i0 + q * 256] = tv[q]; }
            }
        }
    }
    if (tid < VT) { outp[tid] = job.out[tid]; }
    __syncthreads();
    TICK_MARK(0);
    if (!has_tiles) { return; }
#ifndef SDRPP_FCM_EARLY_IQ
    wave_prologue();
#endif

    const float sgn = hi ? -1.0f : 1.0f;
    const float* P1 = hi ? XI : XR;
    const float* P2 = hi ? XR : XI;
    const int ib = jl * D + jl;  // skewed index of IQ sample jl * D
    for (int it = 0; it < ntl; it++) {
        const int tb = tile0 + it;
        planes_store();   // registers -> this wavefront's planes (the previous tile's reads are complete: wave_sync below)
        if (it == 0) { TICK_MARK(3); }
        tile_phasor(tb);
        if (it + 1 < ntl) { fetch(tile_base(tb + 1)); }  // in flight during the matrix loop
        wave_sync();
        if (it == 0) { TICK_MARK(1); }
        wave_prio_low();
        f32x16 accR = mfma_zero(), accI = mfma_zero();
        if constexpr (KS > 0) {
            // PAIR-PER-HALF form (round 5).  The 32 x 32 x 2 instruction takes k = 0 from lanes 0-31 and k = 1 from lanes 32-63.  Until round 4
            // k = 0 / 1 were the sums / differences of ONE tap pair, so every lane needed the pair's four sample components from two planes, a
            // tap, and formed its B operands with two fmaf: 3 LDS instructions + 2 vector instructions per pair of matrix instructions, and an
            // LDS read costs the issuing wavefront 12-15 cycles of matrix issue (tools/probe/mfma_operand_probe.hip).  Now k = 0 / 1 are two
            // CONSECUTIVE pairs: lane (jl, hi) owns pair 2 q + hi of output jl, reads the pair's two complex samples (a, b) and its taps
            // (gr, -gi) as 8-byte values — consecutive q merge into ds_read2_b64 — and two PACKED adds give all four B operands:
            //     s = a + b = (sr, si)        d = (b.re - a.re, a.im - b.im) = (-dr, di)
            //     accR += gr * sr  (pairs 2q, 2q+1);  accR += -gi * di;      accI += gr * si;  accI += -gi * -dr
            // 0.75 LDS + 0.5 vector instructions per matrix instruction pair instead of 3 + 2.  Every output is still ONE k-ordered fmaf chain,
            // in the order (sums 2q, sums 2q+1, differences 2q, differences 2q+1) — the order the 16 x 16 x 4 shapes below follow as well.
            constexpr int NPS = (KS + 1) / 2, NQ = NPS / 2;
            static_assert((KS & 1) == 0 && (NPS & 1) == 0, "even filters with an even number of tap pairs");
            const float2* Pa = X2 + ib + hi;   // a of pair 2 q + hi: sample jl * D + 2 q + hi  ((2q + 1) >> lgD == 2q >> lgD)
            const float2* Pb = X2 + ib - hi;   // b: sample jl * D + K - 1 - 2 q - hi  (K - 1 - 2q is odd: taking hi off never crosses a multiple of D)
            const float2* Tp = reinterpret_cast<const float2*>(AL) + lane;
            // operands travel in CHUNKS of two double pairs (the three 8-byte reads of q and q + 1 share their bases: three ds_read2_b64), a
            // chunk = eight matrix instructions ahead of its use
            constexpr int NC = (NQ + 1) / 2;
            float2 ra[2][2], rb[2][2], rg[2][2];
            auto issue = [&](int c, int slot) {
                constexpr int K1 = KS - 1;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int q = (2 * c + u < NQ) ? 2 * c + u : NQ - 1;  // (an odd number of double pairs: the last chunk reads its one double pair twice, uses it once)
                    ra[slot][u] = Pa[2 * q + ((2 * q) >> lgD)];
                    rb[slot][u] = Pb[(K1 - 2 * q) + ((K1 - 2 * q) >> lgD)];
                    rg[slot][u] = Tp[q * 64];
                }
            };
            issue(0, 0);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (c + 1 < NC) { issue(c + 1, (c + 1) & 1); }
                sched_fence();
                const int sl = c & 1;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (2 * c + u < NQ) {
                        f32x2 s, d;
                        pk_sum_diff_f32(ra[sl][u], rb[sl][u], s, d);
                        accR = mfma_32x32x2(rg[sl][u].x, s.x, accR);
                        accR = mfma_32x32x2(rg[sl][u].y, d.y, accR);
                        accI = mfma_32x32x2(rg[sl][u].x, s.y, accI);
                        accI = mfma_32x32x2(rg[sl][u].y, d.x, accI);
                    }
                }
                sched_fence();
            }
        }
        else {
            // operands of pair p: B = (sums | differences) of the two IQ samples the pair touches, A = its tap column
            auto operands = [&](int p, float& a_re, float& bre, float& bim) {
                const int pe = p < NP ? p : NP - 1;  // padding rows carry zero taps; keep their B operand finite
                const int kb = K - 1 - pe;
                const int ia = ib + pe + (pe >> lgD), ibb = ib + kb + (kb >> lgD);
                const float a1 = P1[ia], a2 = P2[ia];
                float b1 = P1[ibb], b2 = P2[ibb];
                if (odd && pe == NP - 1) { b1 = 0.0f; b2 = 0.0f; }
                bre = fmaf(sgn, b1, a1);
                bim = fmaf(sgn, a2, b2);  // lanes 32-63: -dr, so that the (gr, -gi) tap operand serves the imaginary part too
                a_re = AL[p * 64 + lane];
            };
            float a_c, br_c, bi_c;
            operands(0, a_c, br_c, bi_c);
            for (int p0 = 0; p0 < NP4; p0 += 4) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    float a_n, br_n, bi_n;
                    operands(p0 + u + 1 < NP4 ? p0 + u + 1 : NP4 - 1, a_n, br_n, bi_n);
                    accR = mfma_32x32x2(a_c, br_c, accR);
                    accI = mfma_32x32x2(a_c, bi_c, accI);
                    a_c = a_n;
                    br_c = br_n;
                    bi_c = bi_n;
                }
            }
        }
        // ---- NCO: tile phasor x in-tile advance, then coalesced stores (lanes = consecutive outputs of one VFO) ----
        if (it == 0) { TICK_MARK(2); }
        wave_prio_high();  // outside the matrix loop the wavefront's vector instructions go first (1 % on the launch: they wait ~30 cycles each behind the neighbours' v_mfma's otherwise)
        {
            const int j0 = tb * tile;
            const bool live = j0 + jl < job.nout;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int v = (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (v < job.nv && live) {
                    const float2 P = ptile[v];
                    const float qr = fmaf(P.x, pt[r].x, -(P.y * pt[r].y)), qi = fmaf(P.x, pt[r].y, P.y * pt[r].x);
                    global_store_f32x2(outp[v], j0 + jl, make_float2(fmaf(accR[r], qr, -(accI[r] * qi)), fmaf(accR[r], qi, accI[r] * qr)));
                }
            }
        }
        wave_sync();  // every lane is done with the planes and tile phasors before the next tile overwrites them
    }
}
template <int PF, int KS, int LGDS>
__global__ __launch_bounds__(256, 3) void vfo_frontcm_kernel(IqSrc src, const FrontCMJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float, smemf)
    vfo_frontcm_body<PF, KS, LGDS>(kidx(blockIdx), smemf, src, jobs);
}

// ---- the same front end in the 16 x 16 x 4 matrix shape, for SMALL blocks ---------------------------------------------------------------
// At the reference's own block size (sr/200 = 50 000 samples: 98 tiles of 32 outputs for the ratio-32 plan) every wavefront of the kernel
// above has ONE tile, and its 132 matrix instructions of 64 cycles each are 4 of the 12 us a front-end workgroup lives — the longest role of
// a 13 us tick.  Here a WORKGROUP takes one 32-output tile and its four wavefronts a quarter each: 16 VFOs x 16 outputs, two tap pairs per
// v_mfma_f32_16x16x4_f32 (k = 0, 1: gr * sums of pairs p, p + 1; k = 2, 3: -gi * differences of pairs p, p + 1), 66 instructions of 32 cycles
// instead of 132 of 64.  The matrix instruction accumulates its k in order, so every output is the same chain of fmaf's as in the 32 x 32 x 2
// form — two pairs at a time, their sums before their differences — and the NCO values come from the same tile phasor and the same in-tile table:
// bit-identical outputs (test_small_block_front_end_shape_is_bit_identical).  Same tap operand table, same job.
struct FCM16Layout { int pl, a_off, pt_off, out_off, total; };
__host__ __device__ inline FCM16Layout frontcm16_layout(int K, int lgD) {
    FCM16Layout L;
    const int nsamp = 15 * (1 << lgD) + K;
    const int np4 = ((((K + 1) >> 1) + 3) >> 2) << 2;
    L.pl = frontcm_plane(nsamp, lgD);
    L.a_off = 4 * 2 * L.pl;
    L.pt_off = L.a_off + np4 * 64;
    L.out_off = L.pt_off + SDRPP_FCM_VT * 2;
    L.total = L.out_off + SDRPP_FCM_VT * 2;
    return L;
}
template <int KS, int LGDS>
__device__ __forceinline__ void vfo_frontcm16_body(const KIdx bid, float* smemf, const IqSrc& src, const FrontCMJob* __restrict__ jobs) {
    const FrontCMJob& job = jobs[bid.y];
    constexpr int K = KS, lgD = LGDS, D = 1 << lgD, VT = SDRPP_FCM_VT, tile = SDRPP_FCM_TILE;
    constexpr int NP = (K + 1) >> 1, NP4 = ((NP + 3) >> 2) << 2, NSTEP = (NP + 1) >> 1;
    constexpr int nsamp = 15 * D + K;
    constexpr int PF = (nsamp + 63) / 64;
    const FCM16Layout L = frontcm16_layout(K, lgD);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int vh = wv & 1, nh = wv >> 1;                     // this wavefront's half of the VFOs / of the tile's outputs
    const int jj = lane & 15, kq = lane >> 4, comp = kq >> 1, po = kq & 1;  // matrix k index kq: sums of pairs p, p + 1, then their differences (the order of vfo_frontcm_body's pair-per-half form)
    float* XR = smemf + wv * 2 * L.pl;
    float* XI = XR + L.pl;
    float* AL = smemf + L.a_off;
    float2* ptile = reinterpret_cast<float2*>(smemf + L.pt_off);      // [VT] tile phasors, shared by the workgroup
    float2** outp = reinterpret_cast<float2**>(smemf + L.out_off);  // [VT]
    const int tb = bid.x;  // the 32-output tile of this workgroup
    const long long tbase = (long long)job.off + (long long)tb * tile * D;
    // this wavefront's IQ window and its slice of the in-tile NCO table: requested first
    float2 pf[PF];
    {
        const long long base = tbase + (long long)nh * 16 * D;
        if (base >= 0 && base >= job.min_idx && base + nsamp <= src.n_cur) {
            const float2* p = src.cur + base;
#pragma unroll
            for (int q = 0; q < PF; q++) {
                const int sidx = lane + q * 64;
                pf[q] = (sidx < nsamp) ? p[sidx] : make_float2(0.0f, 0.0f);
            }
        }
        else {
#pragma unroll
            for (int q = 0; q < PF; q++) {
                const int sidx = lane + q * 64;
                const long long gi = base + sidx;
                pf[q] = iq_load_nb(src, gi, sidx < nsamp && gi >= job.min_idx);
            }
        }
    }
    float2 pt[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { pt[r] = global_load_f32x2(job.ptab, (vh * 16 + 4 * kq + r) * tile + nh * 16 + jj); }
    {   // tap operand table (as in vfo_frontcm_body)
        constexpr int NB = 5;
        const int n4 = NP4 * 16;
        const float4* at4 = reinterpret_cast<const float4*>(job.atab);
        float4* AL4 = reinterpret_cast<float4*>(AL);
        for (int i0 = tid; i0 < n4; i0 += 256 * NB) {
            float4 tv[NB];
#pragma unroll
            for (int q = 0; q < NB; q++) { tv[q] = global_load_f32x4(at4, min(i0 + q * 256, n4 - 1)); }
#pragma unroll
            for (int q = 0; q < NB; q++) {
                if (i0 + q * 256 < n4) { AL4[i0 + q * 256] = tv[q]; }
            }
        }
    }
    if (tid < VT) {
        outp[tid] = job.out[tid];
        if (tid < job.nv) {  // the tile's phasor per VFO: exactly vfo_frontcm_body's tile_phasor
            double ph = fma((double)tbase + 0.5 * (double)(K - 1), job.theta[tid], job.phi0[tid]);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            ptile[tid] = make_float2(cs, sn);
        }
    }
#pragma unroll
    for (int q = 0; q < PF; q++) {
        const int sidx = lane + q * 64;
        if (sidx < nsamp) {
            const int idx = sidx + (sidx >> lgD);
            XR[idx] = pf[q].x;
            XI[idx] = pf[q].y;
        }
    }
    __syncthreads();
    TICK_MARK(0);
    if (tb * tile >= job.nout) { return; }
    const float sgn = comp ? -1.0f : 1.0f;
    const int ib = jj * D + jj;  // skewed index of IQ sample jj * D
    // pair p = 2 m + po of step m: with K even and an even number of pairs every index below is a per-lane base + a compile-time offset
    // ((2 m + po) >> lgD == 2 m >> lgD, and K - 1 - 2 m is odd, so taking po off it never crosses a multiple of D either)
    static_assert((K & 1) == 0 && (NP & 1) == 0 && NSTEP * 2 == NP, "even filters with an even number of tap pairs");
    const float* P1a = (comp ? XI : XR) + ib + po;
    const float* P2a = (comp ? XR : XI) + ib + po;
    const float* P1b = (comp ? XI : XR) + ib - po;
    const float* P2b = (comp ? XR : XI) + ib - po;
    const float* Ap = AL + po * 64 + comp * 32 + vh * 16 + jj;
    wave_prio_low();
    f32x4 accR = mfma4_zero(), accI = mfma4_zero();
#pragma unroll
    for (int m = 0; m < NSTEP; m++) {
        const int oa = 2 * m + ((2 * m) >> lgD);
        const int ob = (K - 1 - 2 * m) + ((K - 1 - 2 * m) >> lgD);
        const float a1 = P1a[oa], a2 = P2a[oa];
        const float b1 = P1b[ob], b2 = P2b[ob];
        const float bre = fmaf(sgn, b1, a1);  // sums: a.re + b.re   differences: a.im - b.im
        const float bim = fmaf(sgn, a2, b2);  // sums: a.im + b.im   differences: b.re - a.re (-dr: the (gr, -gi) operand serves both products)
        const float a_op = Ap[2 * m * 64];
        accR = mfma_16x16x4(a_op, bre, accR);
        accI = mfma_16x16x4(a_op, bim, accI);
    }
    TICK_MARK(2);
    wave_prio_high();
    {
        const int n = tb * tile + nh * 16 + jj;
        const bool live = n < job.nout;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int v = vh * 16 + 4 * kq + r;
            if (v < job.nv && live) {
                const float2 P = ptile[v];
                const float qr = fmaf(P.x, pt[r].x, -(P.y * pt[r].y)), qi = fmaf(P.x, pt[r].y, P.y * pt[r].x);
                global_store_f32x2(outp[v], n, make_float2(fmaf(accR[r], qr, -(accI[r] * qi)), fmaf(accR[r], qi, accI[r] * qr)));
            }
        }
    }
}
template <int KS, int LGDS>
__global__ __launch_bounds__(256) void vfo_frontcm16_kernel(IqSrc src, const FrontCMJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float, smemf)
    vfo_frontcm16_body<KS, LGDS>(kidx(blockIdx), smemf, src, jobs);
}

// ---- ... and the 16 x 16 x 4 shape for LARGE blocks: every wavefront an independent engine that walks the tiles of its workgroup ------------
// Why: the 32 x 32 x 2 form needs 168 registers and 41 KB of LDS per workgroup — three workgroups per CU — and in pipelined mode the whole tick
// kernel inherits that budget: every role of a tick, the Toeplitz filters and the FFT passes included, runs at three wavefronts per SIMD because
// ONE role needs the registers.  This shape needs ~70 registers and 31 KB: with it as the front end a tick kernel built for FOUR workgroups per CU
// (tick_kernel<2>) holds every role of the radio path.  Same tap operand table, same k-ordered chains, same tile phasor and in-tile NCO table as
// vfo_frontcm16_body: bit-identical outputs.  A workgroup takes `tiles_per_wave` consecutive 32-output tiles (the job field counts tiles per
// WORKGROUP here); its four wavefronts are the four quarters (VFO half x output half) of every tile, each with planes and tile phasors of its own,
// the next tile's IQ window in flight during the matrix loop, no workgroup barrier after the prologue.
struct FCM16WLayout { int pl, a_off, pt_off, out_off, total; };
__host__ __device__ inline FCM16WLayout frontcm16w_layout(int K, int lgD) {
    FCM16WLayout L;
    const int nsamp = 15 * (1 << lgD) + K;
    const int np2 = (((K + 1) >> 1) + 1) & ~1;  // tap pairs, padded to a whole number of matrix steps (two pairs each)
    L.pl = frontcm_plane(nsamp, lgD);
    L.a_off = 4 * 2 * L.pl;
    L.pt_off = L.a_off + np2 * 64;
    L.out_off = L.pt_off + 4 * 16 * 2;
    L.total = L.out_off + SDRPP_FCM_VT * 2;
    return L;
}
template <int KS, int LGDS>
__device__ __forceinline__ void vfo_frontcm16w_body(const KIdx bid, float* smemf, const IqSrc& src, const FrontCMJob* __restrict__ jobs) {
    const FrontCMJob& job = jobs[bid.y];
    constexpr int K = KS, lgD = LGDS, D = 1 << lgD, VT = SDRPP_FCM_VT, tile = SDRPP_FCM_TILE;
    constexpr int NP = (K + 1) >> 1, NSTEP = (NP + 1) >> 1;
    constexpr int nsamp = 15 * D + K;
    constexpr int PF = (nsamp + 63) / 64;
    static_assert((K & 1) == 0 && (NP & 1) == 0 && NSTEP * 2 == NP, "even filters with an even number of tap pairs");
    const FCM16WLayout L = frontcm16w_layout(K, lgD);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int vh = wv & 1, nh = wv >> 1;                     // this wavefront's half of the VFOs / of every tile's outputs
    const int jj = lane & 15, kq = lane >> 4, comp = kq >> 1, po = kq & 1;  // matrix k index kq: sums of pairs p, p + 1, then their differences (the order of vfo_frontcm_body's pair-per-half form)
    float* XR = smemf + wv * 2 * L.pl;
    float* XI = XR + L.pl;
    float* AL = smemf + L.a_off;
    float2* ptile = reinterpret_cast<float2*>(smemf + L.pt_off) + wv * 16;  // [16]: tile phasors of this wavefront's VFOs
    float2** outp = reinterpret_cast<float2**>(smemf + L.out_off);          // [VT]
    {   // tap operand table (exactly NP rows: the unrolled matrix loop reads no padding), all loads of a work-item in flight before the first LDS write
        constexpr int n4 = NP * 16, NB = (n4 + 255) / 256;
        const float4* at4 = reinterpret_cast<const float4*>(job.atab);
        float4* AL4 = reinterpret_cast<float4*>(AL);
        float4 tv[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) { tv[q] = global_load_f32x4(at4, min(tid + q * 256, n4 - 1)); }
#pragma unroll
        for (int q = 0; q < NB; q++) {
            if (tid + q * 256 < n4) { AL4[tid + q * 256] = tv[q]; }
        }
    }
    if (tid < VT) { outp[tid] = job.out[tid]; }
    __syncthreads();
    TICK_MARK(0);
    const int tile0 = bid.x * job.tiles_per_wave;
    if (tile0 * tile >= job.nout) { return; }
    int ntl = (job.nout - tile0 * tile + tile - 1) / tile;
    if (ntl > job.tiles_per_wave) { ntl = job.tiles_per_wave; }
    auto tile_base = [&](int tb) -> long long { return (long long)job.off + (long long)tb * tile * D; };
    float2 pf[PF];
    auto fetch = [&](int tb) {
        const long long base = tile_base(tb) + (long long)nh * 16 * D;
        if (base >= 0 && base >= job.min_idx && base + nsamp <= src.n_cur) {
            const float2* p = src.cur + base;
#pragma unroll
            for (int q = 0; q < PF; q++) {
                const int sidx = lane + q * 64;
                pf[q] = (sidx < nsamp) ? p[sidx] : make_float2(0.0f, 0.0f);
            }
        }
        else {
#pragma unroll
            for (int q = 0; q < PF; q++) {
                const int sidx = lane + q * 64;
                const long long gi = base + sidx;
                pf[q] = iq_load_nb(src, gi, sidx < nsamp && gi >= job.min_idx);
            }
        }
    };
    float2 pt[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { pt[r] = global_load_f32x2(job.ptab, (vh * 16 + 4 * kq + r) * tile + nh * 16 + jj); }
    fetch(tile0);
    const float sgn = comp ? -1.0f : 1.0f;
    const int ib = jj * D + jj;  // skewed index of IQ sample jj * D
    const float* P1a = (comp ? XI : XR) + ib + po;
    const float* P2a = (comp ? XR : XI) + ib + po;
    const float* P1b = (comp ? XI : XR) + ib - po;
    const float* P2b = (comp ? XR : XI) + ib - po;
    const float* Ap = AL + po * 64 + comp * 32 + vh * 16 + jj;
    for (int it = 0; it < ntl; it++) {
        const int tb = tile0 + it;
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const int sidx = lane + q * 64;
            if (sidx < nsamp) {
                const int idx = sidx + (sidx >> lgD);
                XR[idx] = pf[q].x;
                XI[idx] = pf[q].y;
            }
        }
        if (lane < 16 && vh * 16 + lane < job.nv) {  // the tile's phasor per VFO: exactly vfo_frontcm_body's tile_phasor
            const int v = vh * 16 + lane;
            double ph = fma((double)tile_base(tb) + 0.5 * (double)(K - 1), job.theta[v], job.phi0[v]);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            ptile[lane] = make_float2(cs, sn);
        }
        if (it + 1 < ntl) { fetch(tb + 1); }  // in flight during the matrix loop
        wave_sync();
        wave_prio_low();
        f32x4 accR = mfma4_zero(), accI = mfma4_zero();
#pragma unroll
        for (int m = 0; m < NSTEP; m++) {
            const int oa = 2 * m + ((2 * m) >> lgD);
            const int ob = (K - 1 - 2 * m) + ((K - 1 - 2 * m) >> lgD);
            const float a1 = P1a[oa], a2 = P2a[oa];
            const float b1 = P1b[ob], b2 = P2b[ob];
            const float bre = fmaf(sgn, b1, a1);  // sums: a.re + b.re   differences: a.im - b.im
            const float bim = fmaf(sgn, a2, b2);  // sums: a.im + b.im   differences: b.re - a.re (-dr: the (gr, -gi) operand serves both products)
            const float a_op = Ap[2 * m * 64];
            accR = mfma_16x16x4(a_op, bre, accR);
            accI = mfma_16x16x4(a_op, bim, accI);
        }
        wave_prio_high();
        {
            const int n = tb * tile + nh * 16 + jj;
            const bool live = n < job.nout;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int v = vh * 16 + 4 * kq + r;
                if (v < job.nv && live) {
                    const float2 P = ptile[4 * kq + r];
                    const float qr = fmaf(P.x, pt[r].x, -(P.y * pt[r].y)), qi = fmaf(P.x, pt[r].y, P.y * pt[r].x);
                    global_store_f32x2(outp[v], n, make_float2(fmaf(accR[r], qr, -(accI[r] * qi)), fmaf(accR[r], qi, accI[r] * qr)));
                }
            }
        }
        wave_sync();  // every lane is done with the planes and tile phasors before the next tile overwrites them
    }
}
template <int KS, int LGDS>
__global__ __launch_bounds__(256, 4) void vfo_frontcm16w_kernel(IqSrc src, const FrontCMJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float, smemf)
    vfo_frontcm16w_body<KS, LGDS>(kidx(blockIdx), smemf, src, jobs);
}

// Long first stages (decimation by 32 or 64 with 143...726 taps: the plans for narrow channels in a very wide capture, e.g. cfg 4's
// 61.44 MS/s -> 60 kS/s) use the same matrix formulation with the first stage alone as the "composite" filter, in a leaner
// shape: 2 wavefronts per block (a wavefront's two IQ planes are ~20 KB), the IQ window goes straight from global memory to the
// planes (no register staging: it would need ~70 VGPRs), and the tap operand — up to 363 pairs x 64 lanes — streams from
// global memory / L2 through a four-deep register ring instead of living in LDS.
__host__ __device__ inline int frontcl_lds_floats(int K, int lgD, int nw = 2) {
    const int nsamp = (SDRPP_FCM_TILE - 1) * (1 << lgD) + K;
    return nw * 2 * frontcm_plane(nsamp, lgD) + nw * SDRPP_FCM_VT * 2 + SDRPP_FCM_VT * 2;  // nw waves x 2 planes + tile phasors + pointers
}
// PF: IQ samples prefetched per lane into registers (covers windows of nsamp <= 64 * PF samples: PF = 38 -> first stages up to 448
// taps at /64); PF = 0: longer windows are loaded in place, unpipelined.
#define SDRPP_FCL_PF 38
// NARROW (round 5): jobs of at most 16 VFOs — cfg 4's 43 channels per mode are a job of 32 and a job of 11 — in the 16 x 16 x 4 shape: 16 VFO rows x 16
// outputs per tile, the instruction's k = 0 .. 3 (lanes 16 kq .. 16 kq + 15) are FOUR consecutive tap pairs, each lane owning the pair 4 Q + kq of
// output n = lane & 15.  Same pair-per-lane operands, same table, half the matrix cycles of a 32-row tile that would be two thirds empty.
template <int PF, bool NARROW>
// nw: tile engines (wavefronts) per workgroup — 2 for a launch of its own (128 work-items); as a role of the tick kernel, whose workgroups are
// 256 wide, 4 when four wavefronts' planes fit half a CU's LDS (the build of the tick kernel that holds this role runs two workgroups per CU:
// with two engines each only ONE wavefront per SIMD was at work, and the long first stages were three quarters of cfg 4's tick).
__device__ __forceinline__ void vfo_frontcl_impl(const KIdx bid, float* smemf, const IqSrc& src, const FrontCMJob* __restrict__ jobs, int nw) {
    const FrontCMJob& job = jobs[bid.y];
    constexpr int tile = NARROW ? 16 : SDRPP_FCM_TILE, VT = SDRPP_FCM_VT;
    const int K = job.ntaps, lgD = job.log2_decim, D = 1 << lgD;
    const int NP = (K + 1) >> 1;
    const bool odd = (K & 1) != 0;
    const int nsamp = (tile - 1) * D + K;
    const int pl = frontcm_plane(nsamp, lgD);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int jl = NARROW ? (lane & 15) : (lane & 31), hi = NARROW ? (lane >> 4) : (lane >> 5);  // output inside the tile; which pair of a step this lane owns (kq for NARROW)
    float2* X2 = reinterpret_cast<float2*>(smemf + wv * 2 * pl);  // ONE skewed plane of complex samples (pair-per-half form, as vfo_frontcm_body)
    float2* ptile = reinterpret_cast<float2*>(smemf + 2 * nw * pl) + wv * VT;
    float2** outp = reinterpret_cast<float2**>(smemf + 2 * nw * pl + nw * VT * 2);
    if (tid < VT) { outp[tid] = job.out[tid]; }
    __syncthreads();  // the only workgroup barrier
    if (wv >= nw) { return; }  // (a role of the tick kernel with two engines: the other two wavefronts of the 256-wide workgroup have nothing to do)
    const int tile0 = (bid.x * nw + wv) * job.tiles_per_wave;
    if (tile0 * tile >= job.nout) { return; }
    int ntl = (job.nout - tile0 * tile + tile - 1) / tile;
    if (ntl > job.tiles_per_wave) { ntl = job.tiles_per_wave; }
    // The window of a tile (up to 21 KB) is fetched into REGISTERS while the matrix cores work on the previous tile — only two
    // wavefronts fit a SIMD (the planes fill the LDS), so the register file has room for it and nothing else would hide the load —
    // and goes to the skewed planes between two matrix loops.
    auto tile_base = [&](int tb) -> long long { return (long long)job.off + (long long)tb * tile * D; };
    float2 pf[PF > 0 ? PF : 1];
    bool pf_valid = false;  // wave-uniform: the registers hold the window of the next tile
    auto fetch = [&](long long base) {  // windows fully inside this push only (all but the first and last tiles of a stream)
        if constexpr (PF > 0) {
            pf_valid = base >= 0 && base >= job.min_idx && base + nsamp <= src.n_cur;
            if (pf_valid) {
                const float2* p = src.cur + base + lane;
#pragma unroll
                for (int q = 0; q < PF; q++) { pf[q] = (q * 64 + lane < nsamp) ? global_load_f32x2(p, q * 64) : make_float2(0.0f, 0.0f); }
            }
        }
    };
    auto planes_store = [&](long long base) {
        if (PF > 0 && pf_valid) {
#pragma unroll
            for (int q = 0; q < (PF > 0 ? PF : 1); q++) {
                const int sidx = lane + q * 64;
                if (sidx < nsamp) {
                    X2[sidx + (sidx >> lgD)] = pf[q];
                }
            }
            return;
        }
        for (int sidx = lane; sidx < nsamp; sidx += 64) {  // history / end of the push / samples older than the VFO: in place, unpipelined
            const long long gi = base + sidx;
            const float2 v = (gi >= job.min_idx) ? iq_load_clamped(src, gi) : make_float2(0.0f, 0.0f);
            X2[sidx + (sidx >> lgD)] = v;
        }
    };
    fetch(tile_base(tile0));
    const int ib = jl * D + jl;
    // pair-per-half form (round 5, see vfo_frontcm_body): lane (jl, hi) owns tap pair 2 q + hi of output jl — its two complex samples are two
    // 8-byte LDS reads, two packed adds give (sr, si) and (-dr, di), the taps (gr, -gi) of ITS pair come from rows 2 q + hi of the table:
    // per FOUR matrix instructions 2 LDS reads + 2 packed adds + 2 tap loads, where the pair-per-instruction form had 8 + 4 + 2 and
    // twice the index arithmetic.  Accumulation order per output: sums of pairs 2q, 2q + 1, then their differences.
    constexpr int PPS = NARROW ? 4 : 2;         // tap pairs per matrix step (the k of the instruction)
    const int NQ = (NP + PPS - 1) / PPS;        // steps (a last step that is not full multiplies valid samples by zero rows of the table)
    const int c_half = (NP - 1) % PPS, c_q = (NP - 1) / PPS;  // odd filters: the centre tap is "pair" NP - 1 with itself — its b operand is zero
    const float2* Xa = X2 + ib + hi;
    const float2* Xb = X2 + ib;
    const float* tg = job.atab + hi * 64 + jl;  // gr of pair PPS q + hi: tg[q * 64 * PPS]; -gi: 32 floats behind it
    auto operands = [&](int q, float2& a, float2& b) {  // q wave-uniform
        const int p2 = PPS * q, kb = K - 1 - p2;
        const int oa = p2 + (p2 >> lgD);                                           // (p2 + hi) >> lgD == p2 >> lgD: PPS divides D
        a = Xa[oa];
        if constexpr (NARROW) {
            const int kbl = kb - hi;                                               // this lane's own b index: may cross a multiple of D inside the step
            b = Xb[kbl + (kbl >> lgD)];
        }
        else {
            const int ob0 = kb + (kb >> lgD), ob1 = (kb - 1) + ((kb - 1) >> lgD);  // scalar; the lane picks its half's
            b = Xb[hi ? ob1 : ob0];
        }
        if (odd && q == c_q && hi == c_half) { b = make_float2(0.0f, 0.0f); }
    };
    for (int it = 0; it < ntl; it++) {
        const int tb = tile0 + it;
        const long long base = tile_base(tb);
        planes_store(base);  // the previous tile's reads are complete (wave_sync at the end of the loop body)
        if (lane < VT && lane < job.nv) {
            double ph = fma((double)base + 0.5 * (double)(K - 1), job.theta[lane], job.phi0[lane]);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            ptile[lane] = make_float2(cs, sn);
        }
        pf_valid = false;
        if (it + 1 < ntl) { fetch(tile_base(tb + 1)); }  // in flight during the matrix loop (spreading these loads over the loop — vector memory
                                                         // operations retire in order, the tap loads queue behind them — measured no faster)
        wave_sync();
        typename std::conditional<NARROW, f32x4, f32x16>::type accR, accI;
        if constexpr (NARROW) { accR = mfma4_zero(); accI = mfma4_zero(); }
        else { accR = mfma_zero(); accI = mfma_zero(); }
        {
            // tap operand ring: four steps ahead, coalesced half rows of the [pair][64] table.  The loop runs over whole rings — the table is
            // zero padded to a multiple of SIXTEEN rows (plan_vfo.h), a padded step multiplies VALID samples (index clamped) by zero taps — so
            // that every ring slot is a fixed register (a uniform branch per slot made the compiler rotate the ring through moves and wait for
            // every tap load where it was issued)
            constexpr int RING = 4;
            const int NQr = ((NQ + RING - 1) / RING) * RING;
            float gq[RING], hq[RING];
#pragma unroll
            for (int u = 0; u < RING; u++) {
                gq[u] = global_load_f32(tg, u * 64 * PPS);
                hq[u] = global_load_f32(tg, u * 64 * PPS + 32);
            }
            float2 a_c, b_c;
            operands(0, a_c, b_c);
            for (int q0 = 0; q0 < NQr; q0 += RING) {
#pragma unroll
                for (int u = 0; u < RING; u++) {
                    const int q = q0 + u;
                    float2 a_n, b_n;
                    operands(q + 1 < NQ ? q + 1 : NQ - 1, a_n, b_n);  // one step = four matrix instructions ahead
                    f32x2 sm, df;
                    pk_sum_diff_f32(a_c, b_c, sm, df);
                    if constexpr (NARROW) {
                        accR = mfma_16x16x4(gq[u], sm.x, accR);
                        accI = mfma_16x16x4(gq[u], sm.y, accI);
                        accR = mfma_16x16x4(hq[u], df.y, accR);
                        accI = mfma_16x16x4(hq[u], df.x, accI);
                    }
                    else {
                        accR = mfma_32x32x2(gq[u], sm.x, accR);
                        accR = mfma_32x32x2(hq[u], df.y, accR);
                        accI = mfma_32x32x2(gq[u], sm.y, accI);
                        accI = mfma_32x32x2(hq[u], df.x, accI);
                    }
                    sched_fence();  // the slot is reloaded BEHIND the matrix instructions that read it: the same registers, no copies, no wait for a load just issued
                    const int qn = q + RING < NQr ? q + RING : q;
                    gq[u] = global_load_f32(tg, qn * 64 * PPS);
                    hq[u] = global_load_f32(tg, qn * 64 * PPS + 32);
                    a_c = a_n;
                    b_c = b_n;
                }
            }
        }
        {
            const int j0 = tb * tile;
            const bool live = j0 + jl < job.nout;
            constexpr int NR = NARROW ? 4 : 16;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int v = NARROW ? (4 * hi + r) : ((r & 3) + 8 * (r >> 2) + 4 * hi);  // the VFO row this lane holds in register r (sdrpp_gfx950.h)
                if (v < job.nv && live) {
                    const float2 P = ptile[v];
                    const float2 T = global_load_f32x2(job.ptab, v * SDRPP_FCM_TILE + jl);  // in-tile NCO advance (L2-resident table; keeping it in 32 registers would spill the prefetch)
                    const float qr = fmaf(P.x, T.x, -(P.y * T.y)), qi = fmaf(P.x, T.y, P.y * T.x);
                    global_store_f32x2(outp[v], j0 + jl, make_float2(fmaf(accR[r], qr, -(accI[r] * qi)), fmaf(accR[r], qi, accI[r] * qr)));
                }
            }
        }
        wave_sync();
    }
}
template <int PF>
__device__ __forceinline__ void vfo_frontcl_body(const KIdx bid, float* smemf, const IqSrc& src, const FrontCMJob* __restrict__ jobs, int nw = 2) {
    if (jobs[bid.y].nv <= 16) { vfo_frontcl_impl<PF, true>(bid, smemf, src, jobs, nw); }  // (wave-uniform: a job is one geometry and one row count)
    else { vfo_frontcl_impl<PF, false>(bid, smemf, src, jobs, nw); }
}
template <int PF>
__global__ __launch_bounds__(128, 2) void vfo_frontcl_kernel(IqSrc src, const FrontCMJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float, smemf)
    vfo_frontcl_body<PF>(kidx(blockIdx), smemf, src, jobs);
}

// =====================================================================================================================
// Per-stream FIR work on the matrix cores ("Toeplitz" form).  Any of the per-VFO filters behind the front end — a decimating
// FIR, the channel filter, the audio low-pass (optionally with the FM discriminator fused into the load), the polyphase
// resampler — computes   out[q * rows + m] = sum_k  B[k][m] * x[base + q * s_in + k]   for consecutive "tiles" q:
// every tile applies the same small banded matrix B (k = input offset inside the tile window, m = output inside the tile:
// B[k][m] = h[k - D * m] for a FIR decimating by D, bank[phase_m][k - shift_m] for the resampler) to a window of the stream.
// Sixteen tiles side by side are one v_mfma_f32_16x16x4_f32 chain: A[i = tile][k] = x[base + i * s_in + k] (data, one LDS read
// per lane and step), B from a zero-padded tap table through a per-lane base index (one LDS read), D[i = tile][j = m].
// Only 15 of the 16 matrix columns are used per tile (rows = 15 for FIRs): s_in = 15 * D is then odd or 2 (mod 4), so the
// 16 lanes that read 16 different tiles fall on different LDS banks without any address skew and every offset is an immediate.
// Efficiency = K / (K + (rows - 1) * D) of the matrix work (the band), at 4x the VALU FMA rate and no register-blocked tap loop.
// A WAVEFRONT is an independent engine (private LDS window, no workgroup barriers after the tap table is loaded); G groups of
// 16 tiles share the B operand.
// =====================================================================================================================
struct ToepJob {
    StreamIn in;
    float* out;
    const float* tl;       // [tl_len] zero-padded tap table
    const int* lbase;      // [64] per-lane base index into tl (includes the lane's k = lane >> 4)
    int tl_len, nsteps;    // matrix steps (4 input offsets each)
    int s_in, rows;        // input samples / outputs per tile
    int base0;             // stream index of window offset 0 of tile 0
    int nout;
    int mt_per_wave;       // macro tiles (G * 16 tiles) per wavefront
    float inv_deviation;   // QUAD only
};

// -DSDRPP_TOEP_KNOCK builds only (`make knock`, diagnostic, results are WRONG by design): g_toep_knock bit 0 drops the output stores,
// bit 1 the window loads, bit 2 the matrix loop — the timing of what is left shows what each part costs (DESIGN_HISTORY.md §4).
#ifdef SDRPP_TOEP_KNOCK
__device__ int g_toep_knock;
#endif
// -DSDRPP_TOEP_PROF builds only (`make prof`, diagnostic): shader-clock cycles every wavefront spends in the phases of a round of the
// pipelined path, summed per launch kind (0 decimator, 1 resampler, 2 channel filter, 3 discriminator + audio low-pass):
// [kind][0] matrix loop, [1] waiting for the next window's loads + registers -> LDS, [2] issuing the loads of the window after,
// [3] discriminator, [4] issuing the output stores, [5] rounds, [6] whole wavefront lifetime, [7] wavefronts.  Printed at sdrpp_destroy.
#ifdef SDRPP_TOEP_PROF
__device__ unsigned long long g_toep_prof[4][8];
#define TOEP_TICK() ((long long)__builtin_readcyclecounter())
#endif

template <int WIDTH, int G, bool QUAD>
__device__ __forceinline__ void vfo_toep_body(const KIdx bid, const KIdx gdim, float* smemt, const ToepJob* __restrict__ jobs) {
    const ToepJob job = jobs[bid.y];  // by value: the fields stay in scalar registers (a reference is re-read from memory after every store)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef SDRPP_TOEP_KNOCK
    const int knock = g_toep_knock;
#endif
    const int nsteps = job.nsteps, s_in = job.s_in, rows = job.rows;
    const int span = (G * 16 - 1) * s_in + 4 * nsteps;  // window of one macro tile
    const int pl = (span + 8) & ~3;
    constexpr int NPL = (WIDTH == 2 || QUAD) ? 2 : 1;
    const int tl_pad = (job.tl_len + 3) & ~3;
    float* TLs = smemt;
    float* XR = smemt + tl_pad + wv * NPL * pl;
    float* XI = XR + pl;  // imaginary plane, or the phase scratch of the fused discriminator
    {   // tap table -> LDS, all loads of a work-item in flight before the first LDS write (a wait per load is a memory round trip each)
        constexpr int NB = 4;
        for (int i0 = tid; i0 < job.tl_len; i0 += 256 * NB) {
            float tv[NB];
#pragma unroll
            for (int q = 0; q < NB; q++) { tv[q] = global_load_f32(job.tl, min(i0 + q * 256, job.tl_len - 1)); }  // (index clamped, never a guarded load)
#pragma unroll
            for (int q = 0; q < NB; q++) {
                if (i0 + q * 256 < job.tl_len) { TLs[i0 + q * 256] = tv[q]; }
            }
        }
    }
    __syncthreads();  // the only workgroup barrier
    TICK_MARK(0);
    const int omt = G * 16 * rows;  // outputs per macro tile
    // macro tiles are dealt out CYCLICALLY: round `it` of wavefront w works on tile w + it * (wavefronts of this job), so at any
    // moment the wavefronts of a job stream through one contiguous region of its input and output
    const int mt0 = bid.x * 4 + wv, mts = gdim.x * 4;
    const int c = lane & 15, kk = lane >> 4;
    const float* Bp = TLs + global_load_i32(job.lbase, lane);
    const float* Ar = XR + c * s_in + kk;
    const float* Ai = XI + c * s_in + kk;
    // complex streams keep the window INTERLEAVED (re, im) in the same 2 * pl floats: samples arrive two at a time (one dwordx4
    // load, one ds_write_b128) and one ds_read_b64 feeds both matrix products — every vector instruction saved here is matrix
    // issue time won back (VALU / LDS-address instructions issued between v_mfma's delay them)
    constexpr bool ILV = (WIDTH == 2) && !QUAD;
    float2* X2 = reinterpret_cast<float2*>(XR);
    const float2* A2 = X2 + c * s_in + kk;
    // window fetch: all loads of a macro tile are in flight together (registers), and the NEXT window is fetched while the matrix
    // cores work on the current one.  A window longer than PF * 64 samples (very long filters) is loaded in place, unpipelined.
    constexpr int PF = 18;
    constexpr int PF4 = 9;  // interleaved mode: sample PAIRS per lane
    constexpr bool CPLX_IN = (WIDTH == 2) || QUAD;
    const int cnt = QUAD ? span + 1 : span;  // QUAD needs one more sample in front: d[i] uses x[i - 1]
    const int npair = (cnt + 1) >> 1;
    const bool piped = ILV ? (npair <= PF4 * 64) : (cnt <= PF * 64);
    float2 pf2[(CPLX_IN && !ILV) ? PF : 1];
    float pf1[CPLX_IN ? 1 : PF];
    float4 pf4[ILV ? PF4 : 1];
    auto fetch = [&](int mt) {
#ifdef SDRPP_TOEP_KNOCK
        if (knock & 2) { return; }
#endif
        const int lo = job.base0 + mt * G * 16 * s_in - (QUAD ? 1 : 0);
        const bool inside = lo >= 0 && lo + cnt <= job.in.n;  // all but the first and last macro tiles: no history / end tests
        if constexpr (ILV) {
            if (inside) {  // (an odd window reads one sample past its end: inside the stream's allocation slack, never used)
#pragma unroll
                for (int q = 0; q < PF4; q++) {
                    const int e = q * 64 + lane;
                    pf4[q] = (e < npair) ? global_load_f32x4_unaligned(job.in.data, 2ll * (lo + 2 * e)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
            else {
#pragma unroll
                for (int q = 0; q < PF4; q++) {
                    const int e = q * 64 + lane;
                    const float2 a = stream_load2_nb(job.in, lo + 2 * e, e < npair), b = stream_load2_nb(job.in, lo + 2 * e + 1, e < npair);
                    pf4[q] = make_float4(a.x, a.y, b.x, b.y);
                }
            }
        }
        else if constexpr (CPLX_IN) {
            if (inside) {
                const float2* src2 = reinterpret_cast<const float2*>(job.in.data) + lo;
#pragma unroll
                for (int q = 0; q < PF; q++) {
                    const int s = q * 64 + lane;
                    pf2[q] = (s < cnt) ? global_load_f32x2(src2, s) : make_float2(0.0f, 0.0f);
                }
            }
            else {
#pragma unroll
                for (int q = 0; q < PF; q++) {
                    const int s = q * 64 + lane;
                    pf2[q] = stream_load2_nb(job.in, lo + s, s < cnt);
                }
            }
        }
        else {
            if (inside) {
#pragma unroll
                for (int q = 0; q < PF; q++) {
                    const int s = q * 64 + lane;
                    pf1[q] = (s < cnt) ? global_load_f32(job.in.data + lo, s) : 0.0f;
                }
            }
            else {
#pragma unroll
                for (int q = 0; q < PF; q++) {
                    const int s = q * 64 + lane;
                    pf1[q] = stream_load1_nb(job.in, lo + s, s < cnt);
                }
            }
        }
    };
    auto window_store = [&]() {
        if constexpr (ILV) {
            float4* X4 = reinterpret_cast<float4*>(XR);
#pragma unroll
            for (int q = 0; q < PF4; q++) {
                const int e = q * 64 + lane;
                if (e < npair) { X4[e] = pf4[q]; }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const int s = q * 64 + lane;
            if (s < cnt) {
                if constexpr (QUAD) { XI[s] = fm_phase(pf2[q].y, pf2[q].x); }
                else if constexpr (WIDTH == 2) {
                    XR[s] = pf2[q].x;
                    XI[s] = pf2[q].y;
                }
                else { XR[s] = pf1[q]; }
            }
        }
    };
    // quadrature.h:39-46 fused into the load: d[i] = normalizePhase(atan2f(x[i]) - atan2f(x[i-1])) * invDeviation
    auto discriminate = [&]() {
        if constexpr (QUAD) {
            wave_sync();
            for (int s = lane; s < span; s += 64) { XR[s] = normalize_phase(XI[s + 1] - XI[s]) * job.inv_deviation; }
        }
    };
    // Order of one round of the pipelined path: matrix work on window t | window t+1 from registers to LDS | loads of window t+2 |
    // stores of the outputs of t.  The only wait for global memory (in front of the LDS writes) then covers loads and stores that
    // were issued one whole round earlier, never the stores just issued.
#ifdef SDRPP_TOEP_PROF
    long long tp_acc[5] = { 0, 0, 0, 0, 0 }, tp_rounds = 0;
    const long long tp_birth = TOEP_TICK();
#endif
    if (piped && mt0 * omt < job.nout) {
        fetch(mt0);
        window_store();
        if (1 < job.mt_per_wave && (mt0 + mts) * omt < job.nout) { fetch(mt0 + mts); }
        discriminate();
        TICK_MARK(1);
    }
    for (int it = 0; it < job.mt_per_wave; it++) {
        const int mt = mt0 + it * mts;
        const int obase = mt * omt;
        if (obase >= job.nout) { break; }
        if (!piped) {
            const int lo = job.base0 + mt * G * 16 * s_in - (QUAD ? 1 : 0);
            for (int s = lane; s < cnt; s += 64) {
                if constexpr (QUAD) {
                    const float2 x = stream_load2(job.in, lo + s);
                    XI[s] = fm_phase(x.y, x.x);
                }
                else if constexpr (WIDTH == 2) { X2[s] = stream_load2(job.in, lo + s); }
                else { XR[s] = stream_load1(job.in, lo + s); }
            }
            discriminate();
        }
        wave_sync();
#ifdef SDRPP_TOEP_PROF
        const long long tp0 = TOEP_TICK();
#endif
        f32x4 accR[G], accI[G];
#pragma unroll
        for (int g = 0; g < G; g++) { accR[g] = mfma4_zero(); accI[g] = mfma4_zero(); }
        // operands of four (complex) / eight (real) steps are fetched together (20-24 independent ds_reads, one wait) in front of
        // their matrix instructions
        {
            constexpr int U = (WIDTH == 2) ? 4 : 8;
            int t0 = 0;
#ifdef SDRPP_TOEP_KNOCK
            if (knock & 4) { t0 = nsteps; }
#endif
            for (; t0 + U <= nsteps; t0 += U) {
                float b[U], xr[U][G], xi[U][G];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    b[u] = Bp[4 * (t0 + u)];
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        if constexpr (ILV) {
                            const float2 a = A2[g * 16 * s_in + 4 * (t0 + u)];
                            xr[u][g] = a.x;
                            xi[u][g] = a.y;
                        }
                        else {
                            xr[u][g] = Ar[g * 16 * s_in + 4 * (t0 + u)];
                            if constexpr (WIDTH == 2) { xi[u][g] = Ai[g * 16 * s_in + 4 * (t0 + u)]; }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        accR[g] = mfma_16x16x4(xr[u][g], b[u], accR[g]);
                        if constexpr (WIDTH == 2) { accI[g] = mfma_16x16x4(xi[u][g], b[u], accI[g]); }
                    }
                }
            }
            for (; t0 < nsteps; t0++) {
                const float b = Bp[4 * t0];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    if constexpr (ILV) {
                        const float2 a = A2[g * 16 * s_in + 4 * t0];
                        accR[g] = mfma_16x16x4(a.x, b, accR[g]);
                        accI[g] = mfma_16x16x4(a.y, b, accI[g]);
                    }
                    else {
                        accR[g] = mfma_16x16x4(Ar[g * 16 * s_in + 4 * t0], b, accR[g]);
                        if constexpr (WIDTH == 2) { accI[g] = mfma_16x16x4(Ai[g * 16 * s_in + 4 * t0], b, accI[g]); }
                    }
                }
            }
        }
#ifdef SDRPP_TOEP_PROF
        sched_fence();
        const long long tp1 = TOEP_TICK();
        long long tp2 = tp1, tp3 = tp1, tp4 = tp1;
#endif
        if (it == 0) { TICK_MARK(2); }
        if (piped && it + 1 < job.mt_per_wave && (mt + mts) * omt < job.nout) {
            wave_sync();  // every lane has read its operands of this window
            window_store();
#ifdef SDRPP_TOEP_PROF
            sched_fence();
            tp2 = TOEP_TICK();
#endif
            if (it + 2 < job.mt_per_wave && (mt + 2 * mts) * omt < job.nout) { fetch(mt + 2 * mts); }
#ifdef SDRPP_TOEP_PROF
            sched_fence();
            tp3 = TOEP_TICK();
#endif
            discriminate();
            sched_fence();
#ifdef SDRPP_TOEP_PROF
            tp4 = TOEP_TICK();
#endif
        }
#ifdef SDRPP_TOEP_KNOCK
        if ((knock & 1) && accR[0][0] != 123.456f) { continue; }
#endif
        // D[i = tile][j = m]: this lane holds output m = lane & 15 of tiles 4 * (lane >> 4) + r
        if (obase + omt <= job.nout) {
            // full macro tile (all but the last one of a stream): no per-output bound tests, one lane offset for all sixteen stores and
            // a wave-uniform base per store (scalar address arithmetic instead of ~12 vector instructions and a branch per store)
            if (c < rows) {
                float2* const ob = reinterpret_cast<float2*>(job.out) + obase;
                const int lofs = 4 * kk * rows + c;
#pragma unroll
                for (int g = 0; g < G; g++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        global_store_f32x2(ob + (g * 16 + r) * rows, lofs, make_float2(accR[g][r], (WIDTH == 2) ? accI[g][r] : accR[g][r]));
                    }
                }
            }
        }
        else if (c < rows) {
#pragma unroll
            for (int g = 0; g < G; g++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int o = obase + (g * 16 + 4 * kk + r) * rows + c;
                    if (o < job.nout) {
                        if constexpr (WIDTH == 2) { global_store_f32x2(reinterpret_cast<float2*>(job.out), o, make_float2(accR[g][r], accI[g][r])); }
                        else { global_store_f32x2(reinterpret_cast<float2*>(job.out), o, make_float2(accR[g][r], accR[g][r])); }  // mono -> stereo
                    }
                }
            }
        }
        if (!piped) { wave_sync(); }  // the next macro tile overwrites the window
#ifdef SDRPP_TOEP_PROF
        sched_fence();
        const long long tp5 = TOEP_TICK();
        tp_acc[0] += tp1 - tp0;
        tp_acc[1] += tp2 - tp1;
        tp_acc[2] += tp3 - tp2;
        tp_acc[3] += tp4 - tp3;
        tp_acc[4] += tp5 - tp4;
        tp_rounds++;
#endif
    }
#ifdef SDRPP_TOEP_PROF
    if (lane == 0) {
        const int kind = QUAD ? 3 : (s_in >= 30 ? 0 : (rows < 15 ? 1 : 2));
        for (int k = 0; k < 5; k++) { atomicAdd(&g_toep_prof[kind][k], (unsigned long long)tp_acc[k]); }
        atomicAdd(&g_toep_prof[kind][5], (unsigned long long)tp_rounds);
        atomicAdd(&g_toep_prof[kind][6], (unsigned long long)(TOEP_TICK() - tp_birth));
        atomicAdd(&g_toep_prof[kind][7], 1ull);
    }
#endif
}
template <int WIDTH, int G, bool QUAD>
__global__ __launch_bounds__(256, 5) void vfo_toep_kernel(const ToepJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float, smemt)
    vfo_toep_body<WIDTH, G, QUAD>(kidx(blockIdx), kidx(gridDim), smemt, jobs);
}

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
