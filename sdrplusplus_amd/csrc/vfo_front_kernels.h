// Front ends: the fused VALU form, the matrix-core composite front end in its 32 x 32 x 2 and 16 x 16 x 4 shapes, the long first stages — part of vfo_kernels.h (included from there, inside namespace sdrpp_k; split out in round 5: the file had grown to 2 700 lines).
#pragma once

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
    int anchor;                      // the index phi0 belongs to (0 = the block's first sample; a push of a launch group: where ITS samples start, so that its phases round as they do block by block)
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
// (also a role of the tick kernel — TR_F2_1, round 5: banks too small for the matrix front end stay pipelined)
template <int VT, int K1S, int LGD1S>  // K1S > 0: stage-1 geometry known at compile time (fully unrolled)
__device__ __forceinline__ void vfo_front2_body(const KIdx bid, float2* smem2, const IqSrc& src, const Front2Job* __restrict__ jobs) {
    const Front2Job& job = jobs[bid.y];
    constexpr int tile = 256;  // stage-1 outputs computed per block (one per work-item); blockDim.x == 256
    const int T2 = job.t2;
    const int j2_0 = bid.x * T2;
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
        double ph = fma((double)(base - job.anchor) + 0.5 * (double)(K1 - 1), job.theta[v], job.phi0[v]);
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
template <int VT, int K1S, int LGD1S>
__global__ __launch_bounds__(256, 8) void vfo_front2_kernel(IqSrc src, const Front2Job* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float2, smem2)
    vfo_front2_body<VT, K1S, LGD1S>(kidx(blockIdx), smem2, src, jobs);
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
    int anchor;       // the index phi0 belongs to (0 = the block's first sample; a push of a launch group: where ITS samples start — the tile phasors then round as they do block by block)
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
struct FCMLayout { int pl, a_off, pt_off, out_off, total, nco_off; };
__host__ __device__ inline FCMLayout frontcm_layout(int K, int lgD) {
    FCMLayout L;
    const int nsamp = (SDRPP_FCM_TILE - 1) * (1 << lgD) + K;
    const int np4 = ((((K + 1) >> 1) + 3) >> 2) << 2;
    L.pl = frontcm_plane(nsamp, lgD);
    L.a_off = 4 * 2 * L.pl;
    L.pt_off = L.a_off + np4 * 64;
    L.out_off = L.pt_off + 4 * SDRPP_FCM_VT * 2;
    L.total = L.out_off + SDRPP_FCM_VT * 2;
    // (round 5) the in-tile NCO table [32 VFOs][32 outputs] for the epilogue: held in 32 registers per lane until now, it pushed the kernel over its
    // 168 registers — ten 8-byte spill stores and reloads per lane and TILE (the next tile's prefetched IQ samples), 10 MB of scratch writes per
    // 10^6-sample block on top of the 16 MB the front end has to write (profiles/r05p_per_role_instruction_mix.md: WRITE_SIZE 26.5 MB)
    L.nco_off = L.total;
    L.total += SDRPP_FCM_VT * SDRPP_FCM_TILE * 2;
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
            double ph = fma((double)(tile_base(tb) - job.anchor) + 0.5 * (double)(K - 1), job.theta[lane], job.phi0[lane]);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            ptile[lane] = make_float2(cs, sn);
        }
    };

    // ---- wavefront prologue: this lane's slice of the in-tile NCO table, first IQ tile ----
    const float2* PT = reinterpret_cast<const float2*>(smemf + L.nco_off) + jl;  // [v * tile]: exp(j 2 pi theta_v D n) of this lane's output n
    auto wave_prologue = [&]() { fetch(tile_base(tile0)); };
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
    {   // in-tile NCO table -> LDS: 2 048 floats, two 16-byte loads per work-item
        const float4* pg = reinterpret_cast<const float4*>(job.ptab);
        float4* pl4 = reinterpret_cast<float4*>(smemf + L.nco_off);
        const float4 t0 = global_load_f32x4(pg, tid), t1 = global_load_f32x4(pg, tid + 256);
        pl4[tid] = t0;
        pl4[tid + 256] = t1;
    }
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
                    const float2 P = ptile[v], T = PT[v * tile];
                    const float qr = fmaf(P.x, T.x, -(P.y * T.y)), qi = fmaf(P.x, T.y, P.y * T.x);
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
            double ph = fma((double)(tbase - job.anchor) + 0.5 * (double)(K - 1), job.theta[tid], job.phi0[tid]);
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
            double ph = fma((double)(tile_base(tb) - job.anchor) + 0.5 * (double)(K - 1), job.theta[v], job.phi0[v]);
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
#ifndef SDRPP_FCL_RING_NARROW
#define SDRPP_FCL_RING_NARROW 4
#endif
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
    TICK_MARK(0);
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
            double ph = fma((double)(base - job.anchor) + 0.5 * (double)(K - 1), job.theta[lane], job.phi0[lane]);
            ph -= rint(ph);
            float sn, cs;
            sincospif(2.0f * (float)ph, &sn, &cs);
            ptile[lane] = make_float2(cs, sn);
        }
        pf_valid = false;
        if (it + 1 < ntl) { fetch(tile_base(tb + 1)); }  // in flight during the matrix loop (spreading these loads over the loop — vector memory
                                                         // operations retire in order, the tap loads queue behind them — measured no faster)
        wave_sync();
        if (it == 0) { TICK_MARK(1); }
        typename std::conditional<NARROW, f32x4, f32x16>::type accR, accI;
        if constexpr (NARROW) { accR = mfma4_zero(); accI = mfma4_zero(); }
        else { accR = mfma_zero(); accI = mfma_zero(); }
        {
            // tap operand ring: four steps ahead, coalesced half rows of the [pair][64] table.  The loop runs over whole rings — the table is
            // zero padded to a multiple of SIXTEEN rows (plan_vfo.h), a padded step multiplies VALID samples (index clamped) by zero taps — so
            // that every ring slot is a fixed register (a uniform branch per slot made the compiler rotate the ring through moves and wait for
            // every tap load where it was issued)
            constexpr int RING = NARROW ? SDRPP_FCL_RING_NARROW : 4;
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
        if (it == 0) { TICK_MARK(2); }
        {
            const int j0 = tb * tile;
            const bool live = j0 + jl < job.nout;
            constexpr int NR = NARROW ? 4 : 16;
            // in-tile NCO advance (L2-resident table; keeping it in 32 registers would spill the prefetch).  Its loads go out in batches of
            // four (eight spilled) IN FRONT of the stores they feed: a store through a pointer out of the job table may alias the table for all the compiler
            // knows, so load / store pairs written one after the other were sixteen memory round trips in a row (3-5 us per tile in cfg 4's tick).
            constexpr int EB = 4;
#pragma unroll
            for (int r0 = 0; r0 < NR; r0 += EB) {
                float2 T[EB];
#pragma unroll
                for (int e = 0; e < EB; e++) {
                    const int r = r0 + e;
                    const int v = NARROW ? (4 * hi + r) : ((r & 3) + 8 * (r >> 2) + 4 * hi);  // the VFO row this lane holds in register r (sdrpp_gfx950.h)
                    T[e] = global_load_f32x2(job.ptab, (v < job.nv ? v : 0) * SDRPP_FCM_TILE + jl);
                }
#pragma unroll
                for (int e = 0; e < EB; e++) {
                    const int r = r0 + e;
                    const int v = NARROW ? (4 * hi + r) : ((r & 3) + 8 * (r >> 2) + 4 * hi);
                    if (v < job.nv && live) {
                        const float2 P = ptile[v];
                        const float qr = fmaf(P.x, T[e].x, -(P.y * T[e].y)), qi = fmaf(P.x, T[e].y, P.y * T[e].x);
                        global_store_f32x2(outp[v], j0 + jl, make_float2(fmaf(accR[r], qr, -(accI[r] * qi)), fmaf(accR[r], qi, accI[r] * qr)));
                    }
                }
            }
        }
        wave_sync();
        if (it == 0) { TICK_MARK(3); }
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

