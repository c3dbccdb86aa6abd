// Per-stream FIR work on the matrix cores (banded-Toeplitz form) — part of vfo_kernels.h (included from there, inside namespace sdrpp_k; split out in round 5: the file had grown to 2 700 lines).
#pragma once

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

