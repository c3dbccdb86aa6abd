// Fused per-VFO back end on the matrix cores: last decimator -> polyphase resampler -> channel filter -> FM discriminator + audio
// low-pass in ONE kernel, the three intermediate streams never leave the CU.
//
// A workgroup = 4 wavefronts = 4 filter STAGES of one VFO working as a software pipeline over "slabs" of 960 IF samples: in
// iteration k wavefront s runs stage s on slab k - s, reading its input slab from an LDS double buffer its producer filled one
// iteration earlier and writing the consumer's next slab; one workgroup barrier per iteration.  Every stage is the banded
// Toeplitz product of vfo_toep_kernel (v_mfma_f32_16x16x4_f32, 16 tiles of 15 — resampler: 12 — outputs per chain); the four
// stages of the ratio-32 WFM chain need 250 / 280 / 280 / 252 matrix instructions per slab, so the pipeline is balanced without
// splitting any stage.  Only the complex IF (RxVFO::out) and the stereo audio go to HBM.
//
// The kernel is stateless apart from the input stream's history: a workgroup owns a time chunk of one VFO and first re-runs the
// pipeline over `warm` slabs in front of it (their outputs are not stored), which reproduces exactly what the delay lines would
// hold — for the first push of a VFO the history reads as zeros, like the reference's cleared buffers.  Output alignment: stage s
// produces, in slab q, its outputs [q * slab_out[s] + C[s], ...); the constants C[s] are chosen on the host so that every
// consumer window starts at offset (tile * s_in) of its LDS slot: no index arithmetic in the matrix loops.
#pragma once
#include "vfo_kernels.h"

namespace sdrpp_k {

#define SDRPP_CHAIN_SLAB 960
struct ChainJob {
    StreamIn in;          // input of stage 0 (complex samples of the stream in front of the last decimator), with history
    float* if_out;        // complex IF = stage-2 output (RxVFO::out)
    float* audio_out;     // stereo audio = stage-3 output
    int n_if;             // IF / audio samples of this push
    const float* tl;      // tap tables of the four stages, concatenated (vfo_toep_kernel format)
    const int* lb;        // [4][64] lane bases, relative to the stage's table
    int tl_off[4], tl_len;
    int nsteps[4], s_in[4], rows[4], groups[4];
    int hist[4];          // [1..3]: history samples kept in front of the stage's input slab
    int slab_out[4];      // outputs per slab
    int C[4];             // output alignment (see above)
    long long in0_base;   // stage 0: stream index of window offset 0 of its output 0
    int d0;               // stage 0: decimation
    int nslabs;           // slabs that cover the push
    int slabs_per_block, warm;
    float inv_deviation;
};

struct ChainLayout { int lb_off, win0, pl0, buf[4], sl[4], phase, total; };
__host__ __device__ inline ChainLayout chain_layout(const ChainJob& j) {
    ChainLayout L;
    int o = (j.tl_len + 3) & ~3;
    L.lb_off = o;
    o += 4 * 64;
    L.pl0 = (15 * j.s_in[0] + 4 * j.nsteps[0] + 8 + 3) & ~3;
    L.win0 = o;
    o += 2 * L.pl0;
    L.sl[0] = 0;
    L.buf[0] = 0;
    for (int s = 1; s < 4; s++) {
        L.sl[s] = (j.hist[s] + j.slab_out[s - 1] + 8 + 3) & ~3;  // samples per plane and slot
        L.buf[s] = o;
        o += 2 * (s < 3 ? 2 : 1) * L.sl[s];  // two slots; complex = two planes, the audio stage's input is real
    }
    L.phase = o;
    o += (j.slab_out[2] + 4 + 3) & ~3;
    L.total = o;
    return L;
}

// One Toeplitz chain: acc += sum_t A[4t] (x) B[4t].  The LDS operands of eight steps are fetched together (24 independent ds_reads,
// one wait) in front of their 16 matrix instructions: with only two wavefronts per SIMD a read -> wait -> mfma loop would spend
// most of its time in LDS latency.
template <bool CPLX>
__device__ __forceinline__ void chain_mfma(const float* Ar, const float* Ai, const float* Bp, int nsteps, f32x4& aR, f32x4& aI) {
    constexpr int U = 8;
    int t0 = 0;
    for (; t0 + U <= nsteps; t0 += U) {
        float b[U], xr[U], xi[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            b[u] = Bp[4 * (t0 + u)];
            xr[u] = Ar[4 * (t0 + u)];
            if constexpr (CPLX) { xi[u] = Ai[4 * (t0 + u)]; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            aR = mfma_16x16x4(xr[u], b[u], aR);
            if constexpr (CPLX) { aI = mfma_16x16x4(xi[u], b[u], aI); }
        }
    }
    for (; t0 < nsteps; t0++) {
        const float b = Bp[4 * t0];
        aR = mfma_16x16x4(Ar[4 * t0], b, aR);
        if constexpr (CPLX) { aI = mfma_16x16x4(Ai[4 * t0], b, aI); }
    }
}

#ifdef SDRPP_CHAIN_PROF
#define CHAIN_T() clock64()
#else
#define CHAIN_T() 0ll
#endif
__global__ __launch_bounds__(256) void vfo_chain_kernel(const ChainJob* __restrict__ jobs, long long* prof) {
    HIP_DYNAMIC_SHARED(float, smc)
    long long p_work = 0, p_wait = 0, p_a = 0, p_b = 0;
    (void)p_a; (void)p_b;
    const ChainJob& job = jobs[blockIdx.y];
    const ChainLayout L = chain_layout(job);
    const int tid = threadIdx.x, lane = tid & 63, st = tid >> 6;
    const int c = lane & 15, kk = lane >> 4;
    // ---- prologue: tap tables, lane bases, cleared buffers ----
    for (int i = tid; i < job.tl_len; i += 256) { smc[i] = global_load_f32(job.tl, i); }
    int* lbs = reinterpret_cast<int*>(smc + L.lb_off);
    for (int i = tid; i < 4 * 64; i += 256) { lbs[i] = global_load_i32(job.lb, i); }
    for (int i = L.win0 + tid; i < L.total; i += 256) { smc[i] = 0.0f; }
    __syncthreads();

    const int chunk = blockIdx.x;
    const int q_lo = chunk * job.slabs_per_block;
    if (q_lo >= job.nslabs) { return; }
    int q_hi = q_lo + job.slabs_per_block;
    const bool last_chunk = q_hi >= job.nslabs;
    if (last_chunk) { q_hi = job.nslabs; }
    const int q_start = q_lo - job.warm;
    const int nloc = q_hi - q_start;
    // what this workgroup stores of stage s's outputs: consecutive chunks tile the index range exactly
    const int lo2 = (chunk == 0) ? 0 : q_lo * job.slab_out[2] + job.C[2];
    const int hi2 = last_chunk ? job.n_if : min(job.n_if, q_hi * job.slab_out[2] + job.C[2]);
    const int lo3 = (chunk == 0) ? 0 : q_lo * job.slab_out[3] + job.C[3];
    const int hi3 = last_chunk ? job.n_if : min(job.n_if, q_hi * job.slab_out[3] + job.C[3]);

    // per-stage constants, picked once with constant indices (a run-time index into the layout struct would put it in scratch
    // memory and turn every use into a global-memory round trip)
    auto pick = [&](int a0, int a1, int a2, int a3) { return st == 0 ? a0 : (st == 1 ? a1 : (st == 2 ? a2 : a3)); };
    const int nsteps = pick(job.nsteps[0], job.nsteps[1], job.nsteps[2], job.nsteps[3]);
    const int s_in = pick(job.s_in[0], job.s_in[1], job.s_in[2], job.s_in[3]);
    const int rows = pick(job.rows[0], job.rows[1], job.rows[2], job.rows[3]);
    const int groups = pick(job.groups[0], job.groups[1], job.groups[2], job.groups[3]);
    const int my_hist = pick(0, job.hist[1], job.hist[2], job.hist[3]);                 // history in front of my input slab
    const int my_in = pick(0, job.slab_out[0], job.slab_out[1], job.slab_out[2]);       // samples per input slab
    const int my_buf = pick(0, L.buf[1], L.buf[2], L.buf[3]);                           // my input double buffer
    const int my_sl = pick(0, L.sl[1], L.sl[2], L.sl[3]);
    const int out_buf = pick(L.buf[1], L.buf[2], L.buf[3], 0);                          // my consumer's double buffer
    const int out_sl = pick(L.sl[1], L.sl[2], L.sl[3], 0);
    const int out_hist = pick(job.hist[1], job.hist[2], job.hist[3], 0);
    const int my_slab = pick(job.slab_out[0], job.slab_out[1], job.slab_out[2], job.slab_out[3]);
    const int my_C = pick(job.C[0], job.C[1], job.C[2], job.C[3]);
    const int tl_off = pick(job.tl_off[0], job.tl_off[1], job.tl_off[2], job.tl_off[3]);
    const float* Bp = smc + tl_off + lbs[st * 64 + lane];
    float* phase = smc + L.phase;  // [0] = last phase of the previous slab, [1 + i] = phase of IF sample i of this slab
    const int win0 = L.win0, pl0 = L.pl0;
    const float inv_dev = job.inv_deviation;
    const int d0 = job.d0;
    const long long in0_base = job.in0_base;
    const StreamIn sin0 = job.in;
    float2* const if_out = reinterpret_cast<float2*>(job.if_out);
    float2* const audio_out = reinterpret_cast<float2*>(job.audio_out);

    for (int k = 0; k < nloc + 3; k++) {
        const long long t_it = CHAIN_T();
        const int rel = k - st;
        if (rel >= 0 && rel < nloc) {
            const int q = q_start + rel;
            const int slot = q & 1;
            const long long o0 = (long long)q * my_slab + my_C;  // first output index of this slab
            if (st == 0) {
                // ---- stage 0: complex decimator, input from global memory through a private window, output -> stage 1's buffer ----
                float* XR = smc + win0;
                float* XI = XR + pl0;
                float* OR = smc + out_buf + slot * 2 * out_sl + out_hist;
                float* OI = OR + out_sl;
                const int span = 15 * s_in + 4 * nsteps;
                for (int g = 0; g < groups; g++) {
                    const long long t_a = CHAIN_T();
                    const long long wbase = in0_base + (long long)d0 * o0 + (long long)s_in * (g * 16);
                    const bool inside = wbase >= 0 && wbase + span <= sin0.n;
                    for (int s0 = 0; s0 < span; s0 += 64 * 8) {
                        float2 tmp[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int s = s0 + u * 64 + lane;
                            if (inside) { tmp[u] = (s < span) ? global_load_f32x2(reinterpret_cast<const float2*>(sin0.data), wbase + s) : make_float2(0.0f, 0.0f); }
                            else {
                                const long long gi = wbase + s;
                                tmp[u] = (s < span && gi >= -(long long)sin0.hist_len) ? stream_load2(sin0, (int)gi) : make_float2(0.0f, 0.0f);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int s = s0 + u * 64 + lane;
                            if (s < span) {
                                XR[s] = tmp[u].x;
                                XI[s] = tmp[u].y;
                            }
                        }
                    }
                    wave_sync();
                    const long long t_b = CHAIN_T();
                    p_a += t_b - t_a;
                    f32x4 aR = mfma4_zero(), aI = mfma4_zero();
                    chain_mfma<true>(XR + c * s_in + kk, XI + c * s_in + kk, Bp, nsteps, aR, aI);
                    if (c < rows) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int o = (g * 16 + 4 * kk + r) * rows + c;
                            OR[o] = aR[r];
                            OI[o] = aI[r];
                        }
                    }
                    wave_sync();
                    p_b += CHAIN_T() - t_b;
                }
            }
            else {
                // ---- stages 1..3: input slab in LDS (slot `slot`: [history | slab]) ----
                const bool cplx = st < 3;
                float* IR = smc + my_buf + slot * (cplx ? 2 : 1) * my_sl;
                float* II = IR + my_sl;
                float* OR = smc + out_buf + slot * 2 * out_sl + out_hist;  // stage 1 only
                for (int g = 0; g < groups; g++) {
                    f32x4 aR = mfma4_zero(), aI = mfma4_zero();
                    const float* Ar = IR + (g * 16 + c) * s_in + kk;
                    const float* Ai = II + (g * 16 + c) * s_in + kk;
                    if (cplx) { chain_mfma<true>(Ar, Ai, Bp, nsteps, aR, aI); }
                    else { chain_mfma<false>(Ar, Ai, Bp, nsteps, aR, aI); }
                    if (c < rows) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int o = (g * 16 + 4 * kk + r) * rows + c;  // slab-relative output index
                            if (st == 1) {
                                OR[o] = aR[r];
                                OR[out_sl + o] = aI[r];
                            }
                            else if (st == 2) {
                                const long long gi = o0 + o;
                                if (gi >= lo2 && gi < hi2) { global_store_f32x2(if_out, gi, make_float2(aR[r], aI[r])); }
                                phase[1 + o] = fm_phase(aI[r], aR[r]);  // quadrature.h:39-46: discriminator on the IF
                            }
                            else {
                                const long long gi = o0 + o;
                                if (gi >= lo3 && gi < hi3) { global_store_f32x2(audio_out, gi, make_float2(aR[r], aR[r])); }  // mono -> stereo
                            }
                        }
                    }
                }
                if (st == 2) {
                    // demodulated samples of this slab -> the audio stage's input slot: d[i] = normalizePhase(ph[i] - ph[i-1]) * invDeviation
                    wave_sync();
                    float* D3 = smc + out_buf + slot * out_sl + out_hist;
                    for (int i = lane; i < my_slab; i += 64) { D3[i] = normalize_phase(phase[1 + i] - phase[i]) * inv_dev; }
                    wave_sync();
                    if (lane == 0) { phase[0] = phase[my_slab]; }
                }
                // history for the NEXT slab: tail of this slab -> front of the other slot (the producer only writes that slot's data part)
                {
                    float* NR = smc + my_buf + (slot ^ 1) * (cplx ? 2 : 1) * my_sl;
                    for (int i = lane; i < my_hist; i += 64) {
                        NR[i] = IR[my_in + i];
                        if (cplx) { NR[my_sl + i] = II[my_in + i]; }
                    }
                }
            }
        }
        const long long t_w = CHAIN_T();
        __syncthreads();
        const long long t_e = CHAIN_T();
        p_work += t_w - t_it;
        p_wait += t_e - t_w;
    }
#ifdef SDRPP_CHAIN_PROF
    if (prof && blockIdx.x == 1 && blockIdx.y == 0 && lane == 0) {
        prof[st * 4 + 0] = p_work;
        prof[st * 4 + 1] = p_wait;
        prof[st * 4 + 2] = p_a;
        prof[st * 4 + 3] = p_b;
        if (st == 0) { prof[16] = nloc + 3; }
    }
#else
    (void)prof;
    (void)p_work;
    (void)p_wait;
#endif
}

}  // namespace sdrpp_k
