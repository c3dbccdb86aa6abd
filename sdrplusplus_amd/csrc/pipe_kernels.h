// The FM back end of a VFO as ONE launch: last decimator -> polyphase resampler -> channel filter -> discriminator + audio low-pass,
// the four banded-Toeplitz matrix products of vfo_toep_kernel (same tap tables, same k-ordered chains: bit-identical results) run as a
// PIPELINE inside a workgroup.  Each of the four wavefronts is one stage; the streams between the stages never leave the CU: they live
// in LDS rings (producer / consumer positions in LDS flags), only the front end's output is read from HBM and only the IF stream (the
// RxVFO output, an API-visible stream), the audio and the few tail samples the next push needs as filter history are written.
// Against four launches this removes three launches and ~0.9 GB of intermediate traffic per 2^24-sample push of the 32-VFO bank.
//
// A workgroup walks a SEGMENT of one VFO's push: a range of macro tiles of the last stage; the tile ranges of the earlier stages follow
// from the windows their consumers need (the warm-up outputs in front of a segment are computed twice, once by each neighbour: about
// one macro tile per stage and segment).  Positions in front of the push come from the streams' history buffers, exactly as in the
// separate launches.
//
// Ring of the stream feeding stage s (consumer macro-tile step W = G * 16 * s_in samples, window W + h): R = k * W samples plus a
// mirror of the first h, so every window — they start at multiples of W — is contiguous; slot(p) = (p - base0_s) mod R.
// Flags: prod = stream position below which the ring is filled, cons = start of the consumer's current window (everything below it
// may be overwritten).  A producer writes [p, p + n) once p + n - R <= cons; a consumer multiplies window m once prod covers it.
// The FM discriminator sits between stages 2 and 3 and is shared between them (a vector instruction of a wavefront whose SIMD neighbours
// are inside their matrix loops waits ~30 cycles for its issue slot, so the stage with the most of them paces the whole pipeline):
// stage 2 takes the phase of its IF outputs while they are in registers (and writes the IF samples to HBM) and hands the PHASES over in
// a plain FIFO; stage 3 turns every macro tile of them into wrapped phase differences in a ring of its own.  Each phase is taken once
// per sample (the separate launches redo the window overlap).
#pragma once

namespace sdrpp_k {

struct PipeJob {
    ToepJob st[4];      // decimator, resampler, channel filter, discriminator + audio low-pass: what the separate launches would get
    int* timeouts;      // the context's page-locked count of flag waits that gave up (lds_flag_wait_ge)
    int tl_off[4];      // LDS offsets (floats) of the four tap tables
    int win_off;        // window of stage 0 (register-staged from HBM)
    int ring_off[3];    // rings feeding stages 1 and 2 (complex: 2 floats per sample), and the FIFO of IF PHASES between stage 2 and stage 3 (real)
    int ring_len[3];    // R (samples); the FIFO: a whole number of stage-2 macro tiles
    int ring_mir[3];    // h (samples); the FIFO: 0
    int dring_off, dring_len, dring_mir;  // stage 3's private ring of discriminator outputs (real), filled by stage 3 itself from the FIFO
    int flag_off;       // prod[3], cons[3]
    int zero_lo, zero_hi;  // rings (cleared at start: a window may reach past what its producer ever writes, and 0 * NaN is NaN)
    int keep[3];        // stage s < 3 also writes its outputs with index >= keep[s] to its HBM stream (0: all of it)
    int dec_stage;      // (host) index of stage 0 in the VFO's decimation plan
    int lvl;            // (host) level of stage 0 in the block's data flow
};

#ifdef SDRPP_TOEP_PROF
// diagnostic build (`make prof`): shader-clock cycles per role in [0] waiting for input, [1] matrix loop, [2] releasing the window / staging
// the next one, [3] own epilogue work before the ring + waiting for ring space, [4] writing outputs + publishing, [5] macro tiles,
// [6] wavefront lifetime, [7] wavefronts
__device__ unsigned long long g_pipe_prof[4][8];
__device__ unsigned long long g_pipe_clock[4] = { 0ull, 0ull, ~0ull, 0ull };  // [2] min / [3] max wavefront lifetime (cycles)  // summed wavefront lifetimes in shader-clock cycles and in 100 MHz wall-clock ticks: the clock the chip held
#endif

__device__ __forceinline__ int pipe_pmod(int a, int m) {
    const int r = a % m;
    return r < 0 ? r + m : r;
}

template <int G, int ROLE>
__device__ __forceinline__ void pipe_role(float* smp, const PipeJob* __restrict__ Jp, int* flags, int tb, int te, int pb, int pe) {  // [tb, te) own macro tiles, [pb, pe) the producer's
    constexpr bool CPLX = ROLE < 3;
    const ToepJob job = Jp->st[ROLE];
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, kk = lane >> 4;
    const int nsteps = job.nsteps, s_in = job.s_in, rows = job.rows;
    const int W = G * 16 * s_in, span = W - s_in + 4 * nsteps, omt = G * 16 * rows;
    const float* Bp = smp + Jp->tl_off[ROLE] + global_load_i32(job.lbase, lane);
    // ---- input side ----
    float* const inbase = smp + (ROLE == 0 ? Jp->win_off : (ROLE == 3 ? Jp->dring_off : Jp->ring_off[ROLE > 0 ? ROLE - 1 : 0]));
    const int Rin = ROLE == 3 ? Jp->dring_len : (ROLE > 0 ? Jp->ring_len[ROLE > 0 ? ROLE - 1 : 0] : 0);
    const int hin = ROLE == 3 ? Jp->dring_mir : (ROLE > 0 ? Jp->ring_mir[ROLE > 0 ? ROLE - 1 : 0] : 0);
    int* const prod_in = flags + (ROLE > 0 ? ROLE - 1 : 0);
    int* const cons_in = flags + 3 + (ROLE > 0 ? ROLE - 1 : 0);
    // ---- output side ----
    float* const outring = smp + Jp->ring_off[ROLE < 3 ? ROLE : 0];
    const int Rout = ROLE < 3 ? Jp->ring_len[ROLE < 3 ? ROLE : 0] : 1, hout = ROLE < 3 ? Jp->ring_mir[ROLE < 3 ? ROLE : 0] : 0;
    int* const prod_out = flags + (ROLE < 3 ? ROLE : 0);
    int* const cons_out = flags + 3 + (ROLE < 3 ? ROLE : 0);
    const int bnext = ROLE < 2 ? Jp->st[ROLE < 2 ? ROLE + 1 : 0].base0 : 0;  // (the FIFO behind stage 2 starts at position 0)
    const int keep = ROLE < 3 ? Jp->keep[ROLE < 3 ? ROLE : 0] : 0;

    // ---- stage 0: window from HBM through registers, one macro tile ahead (as vfo_toep_kernel's interleaved path) ----
    constexpr int PF4 = (G == 1) ? 5 : 9;  // sample PAIRS per lane: every unused one still costs its predicated load / select / LDS write
    const int npair = (span + 1) >> 1;
    float4 pf4[ROLE == 0 ? PF4 : 1];
    auto fetch = [&](int mt) {
        if constexpr (ROLE == 0) {
            const int lo = job.base0 + mt * W;
            if (lo >= 0 && lo + span <= job.in.n) {
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
                    float2 a = make_float2(0.0f, 0.0f), b = make_float2(0.0f, 0.0f);
                    if (e < npair) {
                        a = stream_load2(job.in, lo + 2 * e);
                        b = stream_load2(job.in, lo + 2 * e + 1);
                    }
                    pf4[q] = make_float4(a.x, a.y, b.x, b.y);
                }
            }
        }
    };
    auto window_store = [&]() {
        if constexpr (ROLE == 0) {
            float4* X4 = reinterpret_cast<float4*>(inbase);
#pragma unroll
            for (int q = 0; q < PF4; q++) {
                const int e = q * 64 + lane;
                if (e < npair) { X4[e] = pf4[q]; }
            }
        }
    };

    // ---- stages 1-3: the part of the first window that lies in front of this push comes from the stream's history ----
    int ws = 0;  // slot of the current window's first sample
    if constexpr (ROLE > 0) {
        ws = (int)(((long long)tb * W) % Rin);
        const int wstart = job.base0 + tb * W;
        if (wstart < 0) {
            const int pend = (wstart + span < 0) ? wstart + span : 0;
            for (int p = wstart + lane; p < pend; p += 64) {
                int slot = ws + (p - wstart);
                slot -= (slot >= Rin) ? Rin : 0;
                if constexpr (ROLE == 3) {
                    const float2 x1 = stream_load2(job.in, p), x0 = stream_load2(job.in, p - 1);
                    const float d = normalize_phase(fm_phase(x1.y, x1.x) - fm_phase(x0.y, x0.x)) * job.inv_deviation;
                    inbase[slot] = d;
                    if (slot < hin) { inbase[slot + Rin] = d; }
                }
                else {
                    const float2 x = stream_load2(job.in, p);
                    float2* I2 = reinterpret_cast<float2*>(inbase);
                    I2[slot] = x;
                    if (slot < hin) { I2[slot + Rin] = x; }
                }
            }
        }
    }
    // ---- stage 3: IF phases of one stage-2 macro tile from the FIFO -> discriminator outputs in the private ring ----
    // d[p] = normalizePhase(phase[p] - phase[p - 1]) * invDeviation (quadrature.h:39-46)
    int dpos = pb * (G * 16 * Jp->st[2].rows);  // next IF position to convert
    int fslot = 0, dslot = 0;
    float ph_carry = 0.0f;
    const int omtc = G * 16 * Jp->st[2].rows;
    const float* const fifo = smp + Jp->ring_off[2];
    if constexpr (ROLE == 3) {
        fslot = dpos % Jp->ring_len[2];
        dslot = pipe_pmod(dpos - job.base0, Rin);
        if (dpos == 0) {
            const float2 x = stream_load2(job.in, -1);
            ph_carry = fm_phase(x.y, x.x);
        }
    }
    auto convert_tile = [&]() {
        if constexpr (ROLE == 3) {
            const bool avail = dpos < pe * omtc;  // past the producer's last macro tile: zeros
            if (avail) { lds_flag_wait_ge(prod_in, dpos + omtc, Jp->timeouts); }
            const bool plain = avail && dpos + omtc <= job.nout && dslot + omtc <= Rin && dslot >= hin;
            // lane l takes positions dpos + l + 64 i; the phase in front of position dpos is the last one of the tile before
            constexpr int NI = 4 * G;  // omtc <= 16 * G * 15 samples
            float ph_last = ph_carry;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const int q = i * 64 + lane;
                const bool act = q < omtc;
                float ph = 0.0f, prev = 0.0f;
                if (act && avail) {
                    ph = fifo[fslot + q];
                    prev = (q > 0) ? fifo[fslot + q - 1] : ph_carry;
                }
                if (i * 64 < omtc) { ph_last = wave_bcast(ph, (omtc - i * 64 >= 64) ? 63 : ((omtc - 1) & 63)); }
                float d = normalize_phase(ph - prev) * job.inv_deviation;
                if (plain) {
                    if (act) { inbase[dslot + q] = d; }
                }
                else if (act) {
                    if (dpos + q >= job.nout) { d = 0.0f; }
                    int slot = dslot + q;
                    slot -= (slot >= Rin) ? Rin : 0;
                    inbase[slot] = d;
                    if (slot < hin) { inbase[slot + Rin] = d; }
                }
            }
            ph_carry = ph_last;
            wave_sync();
            if (avail && lane == 0) { lds_flag_set(cons_in, dpos + omtc); }
            dpos += omtc;
            fslot += omtc;
            fslot -= (fslot >= Jp->ring_len[2]) ? Jp->ring_len[2] : 0;
            dslot += omtc;
            dslot -= (dslot >= Rin) ? Rin : 0;
        }
    };
    int so = 0;  // slot of this macro tile's first output in the output ring
    if constexpr (ROLE < 3) { so = pipe_pmod(tb * omt - bnext, Rout); }

    if constexpr (ROLE == 0) {
        if (tb < te) {
            fetch(tb);
            window_store();
            if (tb + 1 < te) { fetch(tb + 1); }
        }
    }
#ifdef SDRPP_TOEP_PROF
    long long tp_acc[5] = { 0, 0, 0, 0, 0 };
    const long long tp_birth = TOEP_TICK();
    const long long tp_birth_wall = (long long)wall_clock64();
#endif
    for (int m = tb; m < te; m++) {
        const int obase = m * omt;
#ifdef SDRPP_TOEP_PROF
        const long long tq0 = TOEP_TICK();
#endif
        if constexpr (ROLE == 3) {
            while (dpos < job.base0 + m * W + span) { convert_tile(); }
        }
        else if constexpr (ROLE > 0) { lds_flag_wait_ge(prod_in, job.base0 + m * W + span, Jp->timeouts); }
        wave_sync();
        wave_prio_low();
#ifdef SDRPP_TOEP_PROF
        sched_fence();
        const long long tq1 = TOEP_TICK();
#endif
        f32x4 accR[G], accI[CPLX ? G : 1];
#pragma unroll
        for (int g = 0; g < G; g++) {
            accR[g] = mfma4_zero();
            if constexpr (CPLX) { accI[g] = mfma4_zero(); }
        }
        if constexpr (CPLX) {
            const float2* A2 = reinterpret_cast<const float2*>(inbase) + ws + c * s_in + kk;
            constexpr int U = 4;
            int t0 = 0;
            for (; t0 + U <= nsteps; t0 += U) {
                float b[U];
                float2 a[U][G];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    b[u] = Bp[4 * (t0 + u)];
#pragma unroll
                    for (int g = 0; g < G; g++) { a[u][g] = A2[g * 16 * s_in + 4 * (t0 + u)]; }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        accR[g] = mfma_16x16x4(a[u][g].x, b[u], accR[g]);
                        accI[g] = mfma_16x16x4(a[u][g].y, b[u], accI[g]);
                    }
                }
            }
            for (; t0 < nsteps; t0++) {
                const float b = Bp[4 * t0];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const float2 a = A2[g * 16 * s_in + 4 * t0];
                    accR[g] = mfma_16x16x4(a.x, b, accR[g]);
                    accI[g] = mfma_16x16x4(a.y, b, accI[g]);
                }
            }
        }
        else {
            const float* Ar = inbase + ws + c * s_in + kk;
            constexpr int U = 8;
            int t0 = 0;
            for (; t0 + U <= nsteps; t0 += U) {
                float b[U], a[U][G];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    b[u] = Bp[4 * (t0 + u)];
#pragma unroll
                    for (int g = 0; g < G; g++) { a[u][g] = Ar[g * 16 * s_in + 4 * (t0 + u)]; }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
#pragma unroll
                    for (int g = 0; g < G; g++) { accR[g] = mfma_16x16x4(a[u][g], b[u], accR[g]); }
                }
            }
            for (; t0 < nsteps; t0++) {
                const float b = Bp[4 * t0];
#pragma unroll
                for (int g = 0; g < G; g++) { accR[g] = mfma_16x16x4(Ar[g * 16 * s_in + 4 * t0], b, accR[g]); }
            }
        }
        wave_prio_high();  // outside the matrix loop: the vector instructions of an epilogue should not queue behind the neighbours' v_mfma's
#ifdef SDRPP_TOEP_PROF
        sched_fence();
        const long long tq2 = TOEP_TICK();
#endif
        // ---- the window is free ----
        if constexpr (ROLE == 0) {
            if (m + 1 < te) {
                wave_sync();
                window_store();
                if (m + 2 < te) { fetch(m + 2); }
                sched_fence();
            }
        }
        else {
            if constexpr (ROLE < 3) {
                wave_sync();
                if (lane == 0) { lds_flag_set(cons_in, job.base0 + (m + 1) * W); }
            }
            ws += W;
            ws -= (ws >= Rin) ? Rin : 0;
        }
#ifdef SDRPP_TOEP_PROF
        sched_fence();
        const long long tq3 = TOEP_TICK();
        long long tq4 = tq3;
#endif
        // ---- outputs: this lane holds output c of tiles g * 16 + 4 * kk + r ----
        if constexpr (ROLE == 3) {
            if (obase + omt <= job.nout) {
                if (c < rows) {
                    float2* const ob = reinterpret_cast<float2*>(job.out) + obase;
                    const int lofs = 4 * kk * rows + c;
#pragma unroll
                    for (int g = 0; g < G; g++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) { global_store_f32x2(ob + (g * 16 + r) * rows, lofs, make_float2(accR[g][r], accR[g][r])); }
                    }
                }
            }
            else if (c < rows) {
#pragma unroll
                for (int g = 0; g < G; g++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = obase + (g * 16 + 4 * kk + r) * rows + c;
                        if (o < job.nout) { global_store_f32x2(reinterpret_cast<float2*>(job.out), o, make_float2(accR[g][r], accR[g][r])); }
                    }
                }
            }
        }
        else {
            const bool tail = obase + omt > keep;  // (wave-uniform) some of these outputs also go to the stage's HBM stream
            const bool full = obase + omt <= job.nout;
            // (wave-uniform) the common macro tile: all outputs valid, no wrap inside the ring, clear of the mirrored head — one
            // address addition per output instead of the bound tests, the wrap and the mirror
            const bool plain = full && so + omt <= Rout && so >= hout;
            const int lofs = 4 * kk * rows + c;
            if constexpr (ROLE == 2) {
                // the IF stream (the RxVFO's output) to HBM
                if (c < rows) {
                    if (full && keep <= obase) {
                        float2* const ob = reinterpret_cast<float2*>(job.out) + obase;
#pragma unroll
                        for (int g = 0; g < G; g++) {
#pragma unroll
                            for (int r = 0; r < 4; r++) { global_store_f32x2(ob + (g * 16 + r) * rows, lofs, make_float2(accR[g][r], accI[g][r])); }
                        }
                    }
                    else if (tail) {
#pragma unroll
                        for (int g = 0; g < G; g++) {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int p = obase + (g * 16 + r) * rows + lofs;
                                if (p >= keep && p < job.nout) { global_store_f32x2(reinterpret_cast<float2*>(job.out), p, make_float2(accR[g][r], accI[g][r])); }
                            }
                        }
                    }
                }
            }
            lds_flag_wait_ge(cons_out, obase + omt - Rout, Jp->timeouts);
#ifdef SDRPP_TOEP_PROF
            sched_fence();
            tq4 = TOEP_TICK();
#endif
            if (c < rows) {
                if constexpr (ROLE == 2) {
                    // (the FIFO holds whole macro tiles: no wrap inside one, no mirror; samples past the end of the stream have phase 0)
                    float* const o = outring + so + lofs;
#pragma unroll
                    for (int g = 0; g < G; g++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int idx = (g * 16 + r) * rows;
                            const float ph = fm_phase(accI[g][r], accR[g][r]);
                            o[idx] = (full || obase + idx + lofs < job.nout) ? ph : 0.0f;
                        }
                    }
                }
                else if (plain) {
                    float* const o = outring + 2 * (so + lofs);
#pragma unroll
                    for (int g = 0; g < G; g++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int idx = 2 * (g * 16 + r) * rows;
                            o[idx] = accR[g][r];
                            o[idx + 1] = accI[g][r];
                        }
                    }
                }
                else {
#pragma unroll
                    for (int g = 0; g < G; g++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int off = (g * 16 + r) * rows + lofs;
                            const int p = obase + off;
                            int slot = so + off;
                            slot -= (slot >= Rout) ? Rout : 0;
                            const float2 v = (p < job.nout) ? make_float2(accR[g][r], accI[g][r]) : make_float2(0.0f, 0.0f);
                            float2* O2 = reinterpret_cast<float2*>(outring);
                            O2[slot] = v;
                            if (slot < hout) { O2[slot + Rout] = v; }
                        }
                    }
                }
                if constexpr (ROLE < 2) {
                    if (tail) {
#pragma unroll
                        for (int g = 0; g < G; g++) {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int p = obase + (g * 16 + r) * rows + lofs;
                                if (p >= keep && p < job.nout) { global_store_f32x2(reinterpret_cast<float2*>(job.out), p, make_float2(accR[g][r], accI[g][r])); }
                            }
                        }
                    }
                }
            }
            wave_sync();
            if (lane == 0) { lds_flag_set(prod_out, obase + omt); }
            so += omt;
            so -= (so >= Rout) ? Rout : 0;
        }
#ifdef SDRPP_TOEP_PROF
        sched_fence();
        const long long tq5 = TOEP_TICK();
        tp_acc[0] += tq1 - tq0;
        tp_acc[1] += tq2 - tq1;
        tp_acc[2] += tq3 - tq2;
        tp_acc[3] += tq4 - tq3;
        tp_acc[4] += tq5 - tq4;
#endif
    }
#ifdef SDRPP_TOEP_PROF
    if (lane == 0) {
        for (int k = 0; k < 5; k++) { atomicAdd(&g_pipe_prof[ROLE][k], (unsigned long long)tp_acc[k]); }
        atomicAdd(&g_pipe_prof[ROLE][5], (unsigned long long)(te - tb));
        atomicAdd(&g_pipe_prof[ROLE][6], (unsigned long long)(TOEP_TICK() - tp_birth));
        atomicAdd(&g_pipe_prof[ROLE][7], 1ull);
        atomicAdd(&g_pipe_clock[0], (unsigned long long)(TOEP_TICK() - tp_birth));
        atomicAdd(&g_pipe_clock[1], (unsigned long long)((long long)wall_clock64() - tp_birth_wall));
        atomicMin(&g_pipe_clock[2], (unsigned long long)(TOEP_TICK() - tp_birth));
        atomicMax(&g_pipe_clock[3], (unsigned long long)(TOEP_TICK() - tp_birth));
    }
#endif
    // nobody waits for a wavefront that has left
    if (lane == 0) {
        if constexpr (ROLE < 3) { lds_flag_set(prod_out, 0x7fffffff); }
        if constexpr (ROLE > 0) { lds_flag_set(cons_in, 0x7fffffff); }
    }
}

// (body of the launch and of the tick kernel's TR_PIPE role: bid.x = segment, gdim.x = segments per VFO, bid.y = job)
template <int G>
__device__ __forceinline__ void vfo_pipe_body(const KIdx bid, const KIdx gdim, float* smp, const PipeJob* __restrict__ jobs) {
    const PipeJob* __restrict__ Jp = jobs + bid.y;
    const int tid = threadIdx.x, wv = tid >> 6;
    // macro-tile ranges of the four stages for this segment: the last stage's share, then back through the windows
    int omt[4], W[4], span[4], nmt[4], b0[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int rows = Jp->st[s].rows, s_in = Jp->st[s].s_in;
        omt[s] = G * 16 * rows;
        W[s] = G * 16 * s_in;
        span[s] = W[s] - s_in + 4 * Jp->st[s].nsteps;
        nmt[s] = (Jp->st[s].nout + omt[s] - 1) / omt[s];
        b0[s] = Jp->st[s].base0;
    }
    int t0[4], t1[4];
    t0[3] = (int)((long long)nmt[3] * (long long)bid.x / (long long)gdim.x);
    t1[3] = (int)((long long)nmt[3] * (long long)(bid.x + 1) / (long long)gdim.x);
    // The LAST segment takes every stage to the end of its stream whatever the stage behind it consumes in this push: a stage's trailing
    // outputs (a decimator's odd sample, up to M - 1 resampler inputs, everything when a tiny push gives the later stages nothing to do)
    // are the next push's filter history and must reach its stream.
    const bool last_seg = bid.x + 1 == gdim.x;
    if (t1[3] <= t0[3] && !last_seg) { return; }  // (the whole workgroup: no barrier is left waiting)
#pragma unroll
    for (int s = 3; s >= 1; s--) {
        int a = 0, b = 0;
        if (t1[s] > t0[s]) {
            const int lo = b0[s] + t0[s] * W[s] - (s == 3 ? 1 : 0);  // (the discriminator also needs the sample in front)
            const int hi = b0[s] + (t1[s] - 1) * W[s] + span[s];
            a = lo > 0 ? lo / omt[s - 1] : 0;
            b = hi > 0 ? (hi + omt[s - 1] - 1) / omt[s - 1] : 0;
        }
        else { b = nmt[s - 1]; }  // nothing to do behind (a tiny push; only the last segment gets here): the whole stage
        b = b < nmt[s - 1] ? b : nmt[s - 1];
        if (last_seg) { b = nmt[s - 1]; }
        a = a < b ? a : b;
        t0[s - 1] = a;
        t1[s - 1] = b;
    }
    for (int i = Jp->zero_lo + tid; i < Jp->zero_hi; i += 256) { smp[i] = 0.0f; }
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const float* tl = Jp->st[s].tl;
        float* dst = smp + Jp->tl_off[s];
        for (int i = tid; i < Jp->st[s].tl_len; i += 256) { dst[i] = global_load_f32(tl, i); }
    }
    int* flags = reinterpret_cast<int*>(smp + Jp->flag_off);
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            flags[i] = t0[i] * omt[i];                      // prod: nothing below the producer's first output is ever asked of it
            flags[3 + i] = (i < 2) ? b0[i + 1] + t0[i + 1] * W[i + 1] : t0[2] * omt[2];  // cons: the consumer's first window (FIFO: nothing taken yet)
        }
    }
    __syncthreads();
    // roles rotate with the segment index, so that whichever SIMD the hardware gives wavefront w does not always run the same stage
    const int role = (wv + bid.x) & 3;
    if (role == 0) { pipe_role<G, 0>(smp, Jp, flags, t0[0], t1[0], 0, 0); }
    else if (role == 1) { pipe_role<G, 1>(smp, Jp, flags, t0[1], t1[1], t0[0], t1[0]); }
    else if (role == 2) { pipe_role<G, 2>(smp, Jp, flags, t0[2], t1[2], t0[1], t1[1]); }
    else { pipe_role<G, 3>(smp, Jp, flags, t0[3], t1[3], t0[2], t1[2]); }
}
template <int G>
__global__ __launch_bounds__(256, 5) void vfo_pipe_kernel(const PipeJob* __restrict__ jobs) {
    HIP_DYNAMIC_SHARED(float, smp)
    vfo_pipe_body<G>(kidx(blockIdx), kidx(gridDim), smp, jobs);
}

}  // namespace sdrpp_k
