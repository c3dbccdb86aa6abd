// The reference's float rotator recursion on the device (parity mode), SSB's second rotator, the retune hand-over — part of vfo_kernels.h (included from there, inside namespace sdrpp_k; split out in round 5: the file had grown to 2 700 lines).
#pragma once

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
// (a role of the tick kernel too — TR_ROTX16, round 5: a bank with a few reference-rotator VFOs keeps the results of all its VFOs pipelined; the
// tick then lasts as long as this chain, which is what bounds such a stream anyway)
struct RotXHead { const RotXJob* jobs; const int* bounds; int njobs, nb, vpw, pad; };  // what the role finds behind its entry's job pointer
template <int SKIP>
__device__ __forceinline__ void vfo_rotate_exact4_body(const int bx, float2* ph_tile, const IqSrc& src, const RotXJob* __restrict__ jobs, int njobs, const int* __restrict__ bounds_g, int nb, int vpw) {
    // ph_tile: [2][64 samples][65]: column = VFO; then [2 * TRIP][64] samples; then the 64 output pointers
    constexpr int TRIP = SDRPP_ROTX4_TRIP;
    float2* x_tile = ph_tile + (size_t)2 * 64 * 65;
    float2** outp = reinterpret_cast<float2**>(x_tile + 2 * TRIP * 64);
    float2* dtab = reinterpret_cast<float2*>(outp + 64);  // phaseDelta of the workgroup's VFOs
    int* sparse = reinterpret_cast<int*>(dtab + 64);       // [2]: the chunk in this buffer carries every SKIP-th phase only
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wv = wave_uniform(tid >> 6);  // (known to be uniform: the roles are scalar branches and the chunk walk stays in scalar registers)
    const int j0 = bx * vpw;  // vpw <= 64 VFOs per workgroup (the host's choice: see rot_exact_vpw)
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
template <int SKIP>
__global__ __launch_bounds__(256) void vfo_rotate_exact4_kernel(IqSrc src, const RotXJob* __restrict__ jobs, int njobs, const int* __restrict__ bounds_g, int nb, int vpw) {
    HIP_DYNAMIC_SHARED(float2, ph_tile)
    vfo_rotate_exact4_body<SKIP>((int)blockIdx.x, ph_tile, src, jobs, njobs, bounds_g, nb, vpw);
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
__device__ __forceinline__ void vfo_ssb_rotate_exact_body(const int jidx, const SsbRotXJob* __restrict__ jobs) {  // one WAVEFRONT per job
    const SsbRotXJob job = jobs[jidx];
    const int lane = (int)threadIdx.x & 63;
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
__global__ __launch_bounds__(64) void vfo_ssb_rotate_exact_kernel(const SsbRotXJob* __restrict__ jobs) { vfo_ssb_rotate_exact_body((int)blockIdx.x, jobs); }

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

