// One launch per block ("tick"): the per-block execution path for hosts that hand blocks over one at a time at the reference's block
// size (sample_rate / 200 samples, core/src/dsp/stream.h:9, source_modules/file_source/src/main.cpp:157).
//
// At that size every kernel of the path is latency-bound (a 50 000-sample block of the 32-VFO bank is less than one wave of work for
// 256 CUs) and a block costs the SUM of its ~8 dependent launches + 2 cross-stream joins: 65 us per block, 0.75 GS/s, 1 % of the
// roofline (DESIGN_HISTORY.md 6b).  The stages of the path form a pipeline, though, and nothing but the data flow orders them: stage s of
// block n needs stage s-1 of block n (and, as filter history, of block n-1).  So the stages are SKEWED over consecutive launches:
// tick t runs stage 0 (landing copy of the samples, upload of the job tables) of block t, stage 1 (front end, FFT pass 1, IQ history)
// of block t-1, stage 2 (first separate decimator, FFT pass 2) of block t-2, ... — all of them independent inside one tick, every
// dependency satisfied by the kernel boundary in front of it.  No flags, no waiting inside a kernel, no events between streams: a
// tick is ONE kernel whose workgroups look up their ROLE (which kernel body, which job table, which grid coordinates) in a table, and
// a block costs the host one launch (2.6 us measured, tools/probe/tick_probe.hip) and the device max(stage) instead of sum(stages).
// Results arrive `depth` ticks late; sdrpp_pipeline_flush / any observing call runs the remaining ticks without new input.
//
// The roles are the bodies of the ordinary kernels (fft_kernels.h, vfo_kernels.h): same code, same arithmetic, bit-identical results
// (tests/test_pipelined.py compares the two execution paths sample for sample).  Per-block buffers are rings (`ring depth` buffers per
// stream) so that a producer working on block n+1 does not overwrite what a consumer still reads of block n.
#pragma once
#include <cstddef>

namespace sdrpp_k {

// ---- generic copy role: landing copy of the input block (page-locked host memory or device memory -> landing ring), job-table upload
//      (page-locked host arena -> device arena), result gather (device streams -> page-locked result slot of the block) ----
struct CopyJob {
    const void* src;
    void* dst;
    long long bytes;  // multiple of 4 (kind 1: SOURCE bytes, multiple of 2)
    int kind;         // 0: verbatim; 1: interleaved int16 -> float (x / 32768: file_source/src/main.cpp:162); 2: complex conjugate (dsp/math/conjugate.h:12-15, bytes a multiple of 8); bit 8: dst is host memory
    int pad;
};
__device__ __forceinline__ void copy_one(const CopyJob& job, int bx, int gx) {
    const long long tid = (long long)bx * 256 + threadIdx.x, nth = (long long)gx * 256;
    if ((job.kind & 0xff) == 1) {
        const short* in = reinterpret_cast<const short*>(job.src);
        float* out = reinterpret_cast<float*>(job.dst);
        const float inv = 1.0f / 32768.0f;
        const long long n = job.bytes / 2;
        for (long long i = tid; i < n; i += nth) { out[i] = ((float)in[i]) * inv; }
    }
    else if ((job.kind & 0xff) == 2) {
        const float2* in = reinterpret_cast<const float2*>(job.src);
        float2* out = reinterpret_cast<float2*>(job.dst);
        const long long n = job.bytes / 8;
        for (long long i = tid; i < n; i += nth) {
            const float2 x = in[i];
            out[i] = make_float2(x.x, -x.y);
        }
    }
    else {
        const bool al16 = ((((unsigned long long)job.src) | ((unsigned long long)job.dst)) & 15ull) == 0ull;
        long long done = 0;
        // (explicit GLOBAL loads / stores: through the generic pointers of the job every access was FLAT, and the eight in-flight vectors below
        // lived in scratch memory — stored and reloaded around every store, because a flat store may alias private memory: sdrpp_gfx950.h)
        if (al16) {
            const long long n16 = job.bytes / 16;
            // eight loads in flight per work-item before the first store: a copy out of page-locked host memory is a round trip over the bus per
            // load, and FEW workgroups with many loads each disturb fewer CUs than many with one (DESIGN_HISTORY.md 4b: whoever shares a CU with such a
            // workgroup waits behind its reads)
            constexpr int U = 8;
            long long i = tid;
            for (; i + (U - 1) * nth < n16; i += U * nth) {
                uint4 v[U];
#pragma unroll
                for (int u = 0; u < U; u++) { v[u] = global_load_u32x4(job.src, i + u * nth); }
#pragma unroll
                for (int u = 0; u < U; u++) { global_store_u32x4(job.dst, i + u * nth, v[u]); }
            }
            for (; i < n16; i += nth) { global_store_u32x4(job.dst, i, global_load_u32x4(job.src, i)); }
            done = n16 * 4;
        }
        const long long n4 = job.bytes / 4;
        for (long long i = done + tid; i < n4; i += nth) { global_store_u32(job.dst, i, global_load_u32(job.src, i)); }
    }
    // (results for the host: page-locked memory is not cached on the device, the stores are complete — tick_finish waits for them — before
    // this wavefront counts itself done)
}
__device__ __forceinline__ void copy_body(const KIdx bid, const KIdx gdim, const CopyJob* __restrict__ jobs) { copy_one(jobs[bid.y], bid.x, gdim.x); }
__global__ __launch_bounds__(256) void copy_kernel(const CopyJob* __restrict__ jobs) { copy_body(kidx(blockIdx), kidx(gridDim), jobs); }

// ---- roles ----
enum TickRole : int {
    TR_NONE = 0,
    TR_COPY,       // CopyJob[gy]
    TR_CARRY,      // CarryJob[gy]; aux = njobs > 0: one wavefront per job, gy = ceil(njobs / 4)
    TR_ROT,        // RotJob[gy], p.src
    TR_FCM_132_4,  // FrontCMJob[gy], p.src: vfo_frontcm_body<10, 132, 4>
    TR_FCM_6,      // <6, 0, 0>
    TR_FCM_10,     // <10, 0, 0>
    TR_FCM_16,     // <16, 0, 0>
    TR_FCM16_132_4,  // vfo_frontcm16_body<132, 4>: the ratio-32 front end in 16 x 16 x 4 shape, one 32-output tile per WORKGROUP (small blocks)
    TR_FCL_0,      // vfo_frontcl_body<0>, aux = tile engines per workgroup (2, or 4 when their planes fit)
    TR_FCL_PF,     // vfo_frontcl_body<SDRPP_FCL_PF> (247 registers: only in the SET = 1 build of the kernel)
    TR_TOEP_C,     // ToepJob[gy]: vfo_toep_body<2, 2, false>
    TR_TOEP_R,     // <1, 2, false>
    TR_TOEP_Q,     // <1, 2, true>
    TR_FIRB_C,     // FirBJob[gy], aux = work-items per workgroup: vfo_firb_body<2, false> (register-blocked VALU FIR: decimators by 4 and 8, very long filters)
    TR_FIRB_R,     // <1, false>
    TR_FIRB_S,     // <1, true>
    TR_FIRB_Q,     // <1, true, true>
    TR_PRE,        // PreJob[gy]
    TR_SEQ,        // SeqJob[gx] (one wavefront per job), aux = njobs
    TR_FFT_S10, TR_FFT_S11, TR_FFT_S12,                              // p.fs
    TR_FFT_P1_5, TR_FFT_P1_6, TR_FFT_P1_7, TR_FFT_P1_8, TR_FFT_P1_9, TR_FFT_P1_10,  // p.p1
    TR_FFT_P2_7, TR_FFT_P2_8, TR_FFT_P2_9, TR_FFT_P2_10,             // p.p2
    TR_FFT_P2ROW, TR_FFT_TR,                                         // p.p2 (long transforms: 4096-point rows in place; transpose into bin order, aux = doZoom group size)
    TR_ZOOM_16, TR_ZOOM_4, TR_ZOOM_1,                                // p.z
    TR_FCM16W_132_4,  // vfo_frontcm16w_body<132, 4>: the ratio-32 front end in 16 x 16 x 4 shape walking `tiles_per_wave` tiles per workgroup (large blocks, SET = 2)
    TR_POLYC,      // PolyJob[gy], aux = LDS window in float2: vfo_polyc_body (many-phase resampler, cycle-major: the AF chain's 96/125)
    TR_DEEMP_P0, TR_DEEMP_P1,  // DeempJob[gy]: vfo_deemph_body<0, 0 / 1> (de-emphasis: segment maps, then the outputs one level later)
    TR_DC_P0, TR_DC_P1,        // DeempJob[gy]: vfo_deemph_body<1, 0 / 1> (the front end's DC blocker)
    TR_WF_RING, TR_WF_TRACE,   // p.wf: raw lines into the waterfall's ring; FFT trace smoothing / hold over the block's zoomed lines
    TR_PIPE,       // PipeJob[gy], gx = segments per VFO: vfo_pipe_body<1> — an FM back end (last decimator, resampler, channel filter, discriminator + audio low-pass) as ONE role
    // reference-rotator VFOs (sdrpp_vfo_desc.nco_mode = 2) inside a pipelined bank (round 5): the chain bounds the tick, every VFO's results stay pipelined
    TR_ROTX16,     // RotXHead (jobs), p.src, gx = workgroups of `vpw` VFOs: vfo_rotate_exact4_body<16> — the reference's float rotator recursion at the full rate
    TR_FIRD,       // FirBJob[gy]: vfo_fir_direct_body<false> (plain decimators whose window fits no LDS tile: the first stages behind that rotator)
    TR_SSBX,       // SsbRotXJob[aux], one wavefront per job (gx = ceil(aux / 4)): SSB's second translation as the same recursion at the IF rate
    // banks too small for the matrix front end (fewer than 17 VFOs of one geometry — what a user's session usually is), one VFO per job (round 5)
    TR_S1_1,       // Stage1Job[gy], p.src, aux = tile (work-items that compute): vfo_stage1_body<1> (first stage alone, window in LDS)
    TR_S1D_1,      // Stage1Job[gy], p.src: vfo_stage1_direct_body<1> (first stage alone, decimation >= 32: straight from memory)
    TR_F2_1,       // Front2Job[gy], p.src: vfo_front2_body<1, 0, 0> (first two stages fused, VALU form)
    TR_POLY,       // PolyJob[gy]: vfo_poly_body (a resampler neither the matrix nor the cycle-major form takes: one output per work-item)
    TR_COUNT
};
struct TickP1 { IqSrc src; FrameGeom g; const float* window; const float2* tw1; const float2* twn; float2* scratch; int lg2, ntiles; };
struct TickP2 { const float2* scratch; const float2* tw2; float* out; float* grp; int lg1, ntiles; };
struct TickFS { IqSrc src; FrameGeom g; const float* window; const float2* tw; float* out; };
struct TickZoom { const float* lines; const int32_t* zs; const int32_t* zc; float* zoomed; int32_t* index; const float* grp; int fft_size, data_width, gsz; float wf_min, wf_max; int pad; };
// waterfall display state (fft_kernels.h: wf_ring_store_body / wf_trace_body)
struct TickWf { const float* src; float* a; float* b; float* c; int n0, n1, n2, n3; float f0, f1, f2, pad; };
struct TickEntry {
    int role, gx, gy, aux;
    const void* jobs;
    union {
        IqSrc src;
        TickP1 p1;
        TickP2 p2;
        TickFS fs;
        TickZoom z;
        TickWf wf;
    } p;
};
#define SDRPP_TICK_MAX_ENTRIES 64
struct TickTable {
    int n, pad[3];
    int block_end[SDRPP_TICK_MAX_ENTRIES];  // running total of workgroups
    TickEntry e[SDRPP_TICK_MAX_ENTRIES];
};
// stage 0 of the block that arrives with this tick: its descriptors are not in device memory yet (the upload is part of this very
// tick), so they travel as kernel arguments — two copy jobs, kept small (3 KB of kernel arguments were measured to cost 6 us per launch)
struct TickL0 {
    CopyJob job[2];  // [0] landing copy, [1] job tables of this block + role table of the NEXT tick -> device arena
    int blocks[2];
    int first;       // index of the first of these workgroups in the launch grid (0: in front of the roles)
    int pad;
};
// completion without a host API call: every wavefront counts itself done, the last one publishes the tick's number in page-locked
// host memory that the host polls (7 us from launch to "host knows" against 12.5 us for hipStreamSynchronize)
struct TickDone {
    unsigned* counter;    // device, running total of finished workgroups
    unsigned* host_flag;  // page-locked, device-mapped: number of completed ticks
    unsigned target;      // value of the counter when this tick's last workgroup has finished (mod 2^32)
    unsigned value;       // what to publish
};
// No device-scope fence per wavefront: on gfx950 that is a write-back of the XCD's whole L2 (65 us per tick of 800 wavefronts, 430 us at
// 10^6-sample blocks, measured) — and what the roles write to device memory is made visible by the end of the kernel anyway, which is all
// the next tick needs.  The host only ever looks at page-locked memory (uncached on the device: its stores are complete when the
// wavefront's memory counter is down, wave_stores_done) after the flag that the LAST wavefront publishes behind a system-scope fence.
__device__ __forceinline__ void tick_finish(const TickDone& d) {
    // ONE count per workgroup: 1 800 wavefronts adding to the same word were measured to cost more than the roles themselves (atomics to
    // one address are carried out one after the other at the memory side).  Every wavefront reaches this point exactly once and past the
    // last barrier of its role (the roles' early exits all lie behind their barriers), so a workgroup barrier is safe here.
    wave_stores_done();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd(d.counter, 1u);
        if (old + 1u == d.target) {
            __threadfence_system();
            *reinterpret_cast<volatile unsigned*>(d.host_flag) = d.value;
        }
    }
}

// LDS (bytes) the FFT / zoom roles carve out of the dynamic allocation: what their stand-alone kernels declare statically
__host__ __device__ inline size_t tick_lds_fft_single(int lg, int fpw) { return ((size_t)(1 << lg) / 2 + (size_t)fpw * ((1 << lg) + (1 << lg) / 16)) * 8; }
__host__ __device__ inline size_t tick_lds_fft_p1(int lg1, int c) { return ((size_t)(1 << lg1) / 2 + (size_t)(1 << lg1) * c) * 8; }
__host__ __device__ inline size_t tick_lds_fft_p2(int lg2, int r) { return ((size_t)(1 << lg2) / 2 + (size_t)r * ((1 << lg2) + (1 << lg2) / 16)) * 8; }
__host__ __device__ inline size_t tick_lds_zoom(int tp) { return (size_t)(256 / tp) * (tp + 1) * 4; }

template <int LG, int FPW>
__device__ __forceinline__ void tick_fft_single(const KIdx bid, float* smem, const TickFS& q) {
    const IqSrc src = q.src;
    const FrameGeom g = q.g;
    float2* tw = reinterpret_cast<float2*>(smem);
    fft_single_body<LG, FPW>(bid, tw, tw + FftSingleLds<LG, FPW>::TW, src, g, q.window, q.tw, q.out);
}
template <int LG1, int C>
__device__ __forceinline__ void tick_fft_p1(const KIdx bid, const KIdx gdim, float* smem, const TickP1& q) {
    static_assert(((1 << LG1) / 16) * C == 256, "tick roles run in 256-thread workgroups");
    const IqSrc src = q.src;
    const FrameGeom g = q.g;
    float2* tw = reinterpret_cast<float2*>(smem);
    fft_pass1_body<LG1, C>(bid, tw, tw + (1 << LG1) / 2, src, g, q.window, q.tw1, q.twn, q.scratch, q.lg2, q.ntiles, gdim.x);
}
template <int LG2, int R>
__device__ __forceinline__ void tick_fft_p2(const KIdx bid, const KIdx gdim, float* smem, const TickP2& q) {
    static_assert(((1 << LG2) / 16) * R == 256, "tick roles run in 256-thread workgroups");
    float2* tw = reinterpret_cast<float2*>(smem);
    fft_pass2_body<LG2, R>(bid, tw, tw + (1 << LG2) / 2, q.scratch, q.tw2, q.out, q.lg1, q.ntiles, q.grp, gdim.x);
}
// (a zoom workgroup of the stand-alone kernel covers 256 / TP pixels of one line — a few microseconds of latency and next to no work, yet a
// resident workgroup slot for that long: in a tick, where slots are what the roles compete for, one workgroup walks `groups` such pixel groups)
template <int TP>
__device__ __forceinline__ void tick_zoom(const KIdx bid, float* smem, const TickZoom& q, int groups) {
    for (int i = 0; i < groups; i++) {
        if (i > 0) { __syncthreads(); }  // the partial maxima of the previous group have been read
        zoom_palette_body<TP>(KIdx{ bid.x * groups + i, bid.y }, smem, q.lines, q.fft_size, q.data_width, q.zs, q.zc, q.wf_min, q.wf_max, q.zoomed, q.index, q.grp, q.gsz);
    }
}

#ifdef SDRPP_TICK_TRACE
// diagnostic build (`make ticktrace`): one record per workgroup — tick, role, entry, workgroup index, start / end in 100 MHz wall-clock
// ticks, hardware id — in a device buffer the host dumps at sdrpp_destroy (SDRPP_TICK_TRACE_FILE); tools/tick_trace.py draws the timeline
struct TickTraceRec { unsigned tick; short role, entry; int block; unsigned long long t0, t1; unsigned hwid, xcc; unsigned long long m[4]; };
#define SDRPP_TICK_TRACE_CAP (1 << 20)
__device__ TickTraceRec g_tick_trace[SDRPP_TICK_TRACE_CAP];
__device__ unsigned g_tick_trace_n;
#endif

// SET 0: every role but TR_FCL_PF (168 registers: three wavefronts per SIMD); SET 1: all roles (247 registers: two); SET 2: every role that
// fits 128 registers and 40 KB of LDS — i.e. all but the 32 x 32 x 2 front ends, the long first stages and the one-pass 4096-point transform —
// FOUR wavefronts per SIMD: the build a tick runs in when its front end has the 16 x 16 x 4 shape (tick_set_for, tick_host.h).
// (SET 0 squeezed into 128 registers by the compiler — four wavefronts per SIMD, the matrix front end spilling 668 bytes per lane — was
// measured slower at every block size: 12.9 against 13.3 GS/s at 10^6 samples, 2.5 against 3.1 at 50 000; profiles/r03e)
template <int SET>
__global__ __launch_bounds__(256, SET == 1 ? 2 : (SET == 2 ? 4 : 3)) void tick_kernel(TickL0 l0, const TickTable* __restrict__ tab, TickDone done) {
    HIP_DYNAMIC_SHARED(float, smem)
    int b = (int)blockIdx.x;
    const int nb0 = l0.blocks[0] + l0.blocks[1];
    // (where in the grid the stage-0 copies stand is the host's choice — workgroups are handed out round the CUs in index order, and a
    // workgroup that shares a CU with a landing copy from host memory waits behind its reads over the bus: DESIGN_HISTORY.md 4b)
    const bool is_l0 = b >= l0.first && b < l0.first + nb0;
    if (is_l0) { b -= l0.first; }
    else if (b >= l0.first + nb0) { b -= nb0; }
#ifdef SDRPP_TICK_TRACE
    const unsigned long long tr_t0 = (unsigned long long)wall_clock64();
    int tr_role = -1, tr_entry = -1;
#endif
    if (is_l0) {
        if (b < l0.blocks[0]) { copy_one(l0.job[0], b, l0.blocks[0]); }
        else { copy_one(l0.job[1], b - l0.blocks[0], l0.blocks[1]); }
    }
    else {
        const int n = tab->n;
        int first = 0;
        // the entry's header (role, grid, aux, job table) comes with the look-up itself; only the per-role parameter block `p` still costs a
        // load of its own, and that one runs beside the load of the job table it no longer has to wait for
        static_assert(offsetof(TickEntry, jobs) == 16 && sizeof(void*) == 8, "wave_upper_bound64_rec hands out the first six dwords of an entry");
        int hdr[6];
        const int ei = wave_upper_bound64_rec(tab->block_end, n, b, &first, tab->e, (int)sizeof(TickEntry), hdr);
        if (ei < n) {
            const TickEntry& e = tab->e[ei];
            const int e_role = hdr[0], e_gy = hdr[2], e_aux = hdr[3];
            const void* e_jobs = reinterpret_cast<const void*>(((unsigned long long)(unsigned)hdr[5] << 32) | (unsigned long long)(unsigned)hdr[4]);
#ifdef SDRPP_TICK_TRACE
            tr_role = e_role;
            tr_entry = ei;
#endif
            const int lb = b - first, gx = hdr[1];
            const KIdx bid{ lb % gx, lb / gx }, gdim{ gx, e_gy };
            switch (e_role) {
            case TR_COPY: copy_body(bid, gdim, reinterpret_cast<const CopyJob*>(e_jobs)); break;
            case TR_CARRY: carry_body(bid, gdim, reinterpret_cast<const CarryJob*>(e_jobs), e_aux); break;
            case TR_ROTX16:
                if constexpr (SET != 2) {  // (71 KB of LDS: not in the four-wavefronts-per-SIMD build)
                    const IqSrc src = e.p.src;
                    const RotXHead h = *reinterpret_cast<const RotXHead*>(e_jobs);
                    vfo_rotate_exact4_body<16>(bid.x, reinterpret_cast<float2*>(smem), src, h.jobs, h.njobs, h.bounds, h.nb, h.vpw);
                }
                break;
            case TR_FIRD: vfo_fir_direct_body<false>(bid, gdim, reinterpret_cast<const FirBJob*>(e_jobs)); break;
            case TR_POLY: vfo_poly_body(bid, reinterpret_cast<float2*>(smem), reinterpret_cast<const PolyJob*>(e_jobs)); break;
            case TR_S1_1: { const IqSrc src = e.p.src; vfo_stage1_body<1>(bid, reinterpret_cast<float2*>(smem), e_aux, 256, src, reinterpret_cast<const Stage1Job*>(e_jobs)); } break;
            case TR_S1D_1: { const IqSrc src = e.p.src; vfo_stage1_direct_body<1>(bid, src, reinterpret_cast<const Stage1Job*>(e_jobs)); } break;
            case TR_F2_1: { const IqSrc src = e.p.src; vfo_front2_body<1, 0, 0>(bid, reinterpret_cast<float2*>(smem), src, reinterpret_cast<const Front2Job*>(e_jobs)); } break;
            case TR_SSBX: {
                const int j = bid.x * 4 + ((int)threadIdx.x >> 6);
                if (j < e_aux) { vfo_ssb_rotate_exact_body(j, reinterpret_cast<const SsbRotXJob*>(e_jobs)); }
            } break;
            case TR_ROT: { const IqSrc src = e.p.src; vfo_rotate_body(bid, gdim, src, reinterpret_cast<const RotJob*>(e_jobs)); } break;
            case TR_FCM_132_4:
                if constexpr (SET != 2) { const IqSrc src = e.p.src; vfo_frontcm_body<10, 132, 4>(bid, smem, src, reinterpret_cast<const FrontCMJob*>(e_jobs)); }
                break;
            // (the front ends with RUN-TIME geometry — any composite filter but the ratio-32 plan's <132 taps, / 16> — need more registers than the 168 of the
            // SET = 0 build: there they spilled 57 registers, 204 bytes of scratch per work-item for the whole kernel, VERDICT r5.  They live in the
            // 247-register build only; a tick that holds one runs as tick_kernel<1>)
            case TR_FCM_6:
                if constexpr (SET == 1) { const IqSrc src = e.p.src; vfo_frontcm_body<6, 0, 0>(bid, smem, src, reinterpret_cast<const FrontCMJob*>(e_jobs)); }
                break;
            case TR_FCM_10:
                if constexpr (SET == 1) { const IqSrc src = e.p.src; vfo_frontcm_body<10, 0, 0>(bid, smem, src, reinterpret_cast<const FrontCMJob*>(e_jobs)); }
                break;
            case TR_FCM_16:
                if constexpr (SET == 1) { const IqSrc src = e.p.src; vfo_frontcm_body<16, 0, 0>(bid, smem, src, reinterpret_cast<const FrontCMJob*>(e_jobs)); }
                break;
            case TR_FCM16_132_4: { const IqSrc src = e.p.src; vfo_frontcm16_body<132, 4>(bid, smem, src, reinterpret_cast<const FrontCMJob*>(e_jobs)); } break;
            case TR_FCL_0:
                if constexpr (SET != 2) { const IqSrc src = e.p.src; vfo_frontcl_body<0>(bid, smem, src, reinterpret_cast<const FrontCMJob*>(e_jobs), e_aux == 4 ? 4 : 2); }
                break;
            case TR_FCL_PF:
                if constexpr (SET == 1) { const IqSrc src = e.p.src; vfo_frontcl_body<SDRPP_FCL_PF>(bid, smem, src, reinterpret_cast<const FrontCMJob*>(e_jobs), e_aux == 4 ? 4 : 2); }
                break;
            case TR_TOEP_C: vfo_toep_body<2, 2, false>(bid, gdim, smem, reinterpret_cast<const ToepJob*>(e_jobs)); break;
            case TR_TOEP_R: vfo_toep_body<1, 2, false>(bid, gdim, smem, reinterpret_cast<const ToepJob*>(e_jobs)); break;
            case TR_TOEP_Q: vfo_toep_body<1, 2, true>(bid, gdim, smem, reinterpret_cast<const ToepJob*>(e_jobs)); break;
            case TR_FIRB_C:
                vfo_firb_body<2, false>(bid, smem, e_aux, reinterpret_cast<const FirBJob*>(e_jobs));
                break;
            case TR_FIRB_R:
                vfo_firb_body<1, false>(bid, smem, e_aux, reinterpret_cast<const FirBJob*>(e_jobs));
                break;
            case TR_FIRB_S:
                vfo_firb_body<1, true>(bid, smem, e_aux, reinterpret_cast<const FirBJob*>(e_jobs));
                break;
            case TR_FIRB_Q:
                vfo_firb_body<1, true, true>(bid, smem, e_aux, reinterpret_cast<const FirBJob*>(e_jobs));
                break;
            case TR_PRE: vfo_demod_pre_body(bid, gdim, reinterpret_cast<const PreJob*>(e_jobs)); break;
            case TR_SEQ:
                if (threadIdx.x < 64) { vfo_sequential_body(bid, reinterpret_cast<const SeqJob*>(e_jobs), e_aux); }
                break;
            case TR_FFT_S10: tick_fft_single<10, 4>(bid, smem, e.p.fs); break;
            case TR_FFT_S11: tick_fft_single<11, 2>(bid, smem, e.p.fs); break;
            case TR_FFT_S12:
                if constexpr (SET != 2) { tick_fft_single<12, 1>(bid, smem, e.p.fs); }
                break;
            case TR_FFT_P1_5: tick_fft_p1<5, 128>(bid, gdim, smem, e.p.p1); break;
            case TR_FFT_P1_6: tick_fft_p1<6, 64>(bid, gdim, smem, e.p.p1); break;
            case TR_FFT_P1_7: tick_fft_p1<7, 32>(bid, gdim, smem, e.p.p1); break;
            case TR_FFT_P1_8: tick_fft_p1<8, 16>(bid, gdim, smem, e.p.p1); break;
            case TR_FFT_P1_9: tick_fft_p1<9, 8>(bid, gdim, smem, e.p.p1); break;
            case TR_FFT_P1_10: tick_fft_p1<10, 4>(bid, gdim, smem, e.p.p1); break;
            case TR_FFT_P2_7: tick_fft_p2<7, 32>(bid, gdim, smem, e.p.p2); break;
            case TR_FFT_P2_8: tick_fft_p2<8, 16>(bid, gdim, smem, e.p.p2); break;
            case TR_FFT_P2_9: tick_fft_p2<9, 8>(bid, gdim, smem, e.p.p2); break;
            case TR_FFT_P2_10: tick_fft_p2<10, 4>(bid, gdim, smem, e.p.p2); break;
            case TR_FFT_P2ROW: {
                float2* tw = reinterpret_cast<float2*>(smem);
                fft_pass2row_body<12>(bid, tw, tw + (1 << 12) / 2, const_cast<float2*>(e.p.p2.scratch), e.p.p2.tw2, e.p.p2.lg1);
            } break;
            case TR_FFT_TR: fft_transpose_body(bid, smem, reinterpret_cast<const float*>(e.p.p2.scratch), e.p.p2.out, e.p.p2.grp, e.p.p2.lg1, 12, e_aux); break;
            case TR_ZOOM_16: tick_zoom<16>(bid, smem, e.p.z, e_aux > 0 ? e_aux : 1); break;
            case TR_ZOOM_4: tick_zoom<4>(bid, smem, e.p.z, e_aux > 0 ? e_aux : 1); break;
            case TR_ZOOM_1: tick_zoom<1>(bid, smem, e.p.z, e_aux > 0 ? e_aux : 1); break;
            case TR_FCM16W_132_4: { const IqSrc src = e.p.src; vfo_frontcm16w_body<132, 4>(bid, smem, src, reinterpret_cast<const FrontCMJob*>(e_jobs)); } break;
            case TR_POLYC: vfo_polyc_body(bid, reinterpret_cast<float2*>(smem), reinterpret_cast<const PolyJob*>(e_jobs), e_aux); break;
            case TR_DEEMP_P0: vfo_deemph_body<0, 0>(bid, smem, reinterpret_cast<const DeempJob*>(e_jobs)); break;
            case TR_DEEMP_P1: vfo_deemph_body<0, 1>(bid, smem, reinterpret_cast<const DeempJob*>(e_jobs)); break;
            case TR_DC_P0: vfo_deemph_body<1, 0>(bid, smem, reinterpret_cast<const DeempJob*>(e_jobs)); break;
            case TR_DC_P1: vfo_deemph_body<1, 1>(bid, smem, reinterpret_cast<const DeempJob*>(e_jobs)); break;
            case TR_WF_RING: { const TickWf q = e.p.wf; wf_ring_store_body(bid, gdim, q.src, q.n0, q.n1, q.a, q.n2, q.n3); } break;
            case TR_WF_TRACE: { const TickWf q = e.p.wf; wf_trace_body(bid, q.src, q.n0, q.n1, q.a, q.b, q.f0, q.f1, q.c, q.f2); } break;
            case TR_PIPE: vfo_pipe_body<1>(bid, gdim, smem, reinterpret_cast<const PipeJob*>(e_jobs)); break;
            default: break;
            }
        }
    }
#ifdef SDRPP_TICK_TRACE
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned k = atomicAdd(&g_tick_trace_n, 1u);
        if (k < SDRPP_TICK_TRACE_CAP) {
            unsigned hw = 0, xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            TickTraceRec rec{ done.value, (short)tr_role, (short)tr_entry, (int)blockIdx.x, tr_t0, (unsigned long long)wall_clock64(), hw, xcc, { 0, 0, 0, 0 } };
            for (int q = 0; q < 4; q++) {
                rec.m[q] = g_tick_mark[blockIdx.x & 0xffff][q];
                g_tick_mark[blockIdx.x & 0xffff][q] = 0;
            }
            g_tick_trace[k] = rec;
        }
    }
#endif
    tick_finish(done);
}

}  // namespace sdrpp_k
