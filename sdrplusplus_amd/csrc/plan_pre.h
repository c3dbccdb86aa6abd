// IQFrontEnd pre-processing chain of one block (PowerDecimator -> DCBlocker -> Conjugate, iq_frontend.cpp:32-39).
// Part of the one translation unit sdrpp_gpu.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

// ---- IQFrontEnd pre-processing chain: one block -------------------------------------------------------------------------------------
// Decimator stages run on the matrix-core FIR kernel (register-blocked VALU kernel for tap counts it does not cover), the DC
// blocker as a two-level scan (vfo_deemph_body<1, *>), the conjugate inside its store (or alone).  On return *d_iq / *count
// describe the pre-processed stream; the raw and stage histories are carried for the next block.
// Like the VFO bank, every step has a LEVEL: stage s of the decimator is level 1 + s, the DC blocker's two passes the two levels behind the
// last stage.  An ordinary pass launches them in that order; in pipelined mode they are roles of the block's first ticks and the FFT branch /
// VFO bank start `plan_lvl0` levels later (reference-order arithmetic has no roles: such a block runs as an ordinary pass).
int run_preproc(sdrpp_ctx* c, const float** d_iq, int64_t* count) {
    sdrpp_ctx::Pre& P = c->pre;
    const bool ticking = c->tick_planning;
    const int n_in = (int)*count;
    std::vector<ToepJob> tj[SDRPP_MAX_DECIM_STAGES];
    std::vector<FirBJob> fj[SDRPP_MAX_DECIM_STAGES];
    Lev<CarryJob> carry;  // pipelined: a stream's history is carried at the level of its consumer; a pass: all behind the chain
    std::vector<DeempJob> dc;
    std::vector<CopyJob> conj;
    P.raw.data = const_cast<float*>(*d_iq);
    P.raw.n = n_in;
    Stream* cur = &P.raw;
    const int last_level = P.n_stages + (P.dc_rate != 0.0f ? 2 : (P.conj ? 1 : 0));
    for (int s = 0; s < P.n_stages; s++) {
        Stream* nxt = &P.st[(size_t)s];
        const int D = P.decim_s[s], K = (int)P.staps[s].size();
        const int no = decim_nout(cur->n, P.soff[s], D);
        if ((size_t)no > nxt->cap) { return fail(c, SDRPP_ERR_INVALID, "pre-processing stage %d: %d outputs exceed the capacity", s, no); }
        bounds_decim(c->vfo_bounds, P.soff[s], D);  // the reference's blocks behind this stage
        if (P.tp[s].ok && !P.ref_order) { tj[s].push_back(toep_job(P.tp[s], 0, stream_in(*cur), nxt->data, P.soff[s] - (K - 1), no, 0.0f)); }
        else { fj[s].push_back(FirBJob{ stream_in(*cur), nxt->data, P.d_staps[s], K, ilog2(D), P.soff[s], no, P.s_kp[s] }); }
        P.soff[s] = P.soff[s] + no * D - cur->n;
        nxt->n = no;
        if (cur->hist_len > 0) { carry.add(ticking ? 1 + s : std::max(1, last_level), CarryJob{ cur->data, cur->hist[cur->cur], cur->hist[cur->cur ^ 1], cur->hist_len, cur->n, 2, cur->hist_len }); }
        cur = nxt;
    }
    const int n_out = cur->n;
    const float* result = cur->data;
    if (P.dc_rate != 0.0f) {
        const int nseg = std::min(P.seg_cap, (n_out + SDRPP_DEEMP_SEG - 1) / SDRPP_DEEMP_SEG);
        dc.push_back(DeempJob{ (const float2*)cur->data, (float2*)P.out.data, n_out, P.dc_rate, P.d_off + P.state_cur, P.d_off + (P.ref_order ? P.state_cur : (P.state_cur ^ 1)),
                               P.d_seg + (size_t)P.state_cur * ((size_t)P.seg_cap + 1), nseg, P.conj });
        if (nseg > 0 && !P.ref_order) { P.state_cur ^= 1; }
        result = P.out.data;
    }
    else if (P.conj) {
        if (n_out > 0) { conj.push_back(CopyJob{ cur->data, P.out.data, (long long)n_out * 8, 2, 0 }); }
        result = P.out.data;
    }
    // job tables
    ToepPlan tp[SDRPP_MAX_DECIM_STAGES];
    ToepJob* d_tj[SDRPP_MAX_DECIM_STAGES] = {};
    FirBJob* d_fj[SDRPP_MAX_DECIM_STAGES] = {};
    for (int s = 0; s < P.n_stages; s++) {
        tp[s] = toep_plan(tj[s], 2, ticking ? c->tick_toep_blocks : 2048);
        d_tj[s] = arena_push(c, tj[s]);
        d_fj[s] = arena_push(c, fj[s]);
        if ((!tj[s].empty() && !d_tj[s]) || (!fj[s].empty() && !d_fj[s])) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
    }
    DeempJob* d_dc = arena_push(c, dc);
    CopyJob* d_conj = arena_push(c, conj);
    if ((!dc.empty() && !d_dc) || (!conj.empty() && !d_conj) || !arena_push_lev(c, carry)) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
    int rc = arena_commit(c);
    if (rc) { return rc; }
    auto emit_carries = [&](int level) {
        if (level >= carry.top || carry.at[level].empty()) { return; }
        int mx = 0;
        for (auto& k : carry.at[level]) { mx = std::max(mx, k.need * k.width); }
        emit(c, level, F_MISC, TR_CARRY, std::max(1, std::min((mx + 8191) / 8192, 64)), (int)carry.at[level].size(), 0, carry.dev[level]);
    };
    {
        FamilyTimer t(c, F_MISC);
        for (int s = 0; s < P.n_stages; s++) {
            const int level = 1 + s;
            if (!tj[s].empty() && tp[s].grid_x > 0) { emit(c, level, F_MISC, TR_TOEP_C, tp[s].grid_x, (int)tj[s].size(), tp[s].lds, d_tj[s]); }
            for (auto& jb : fj[s]) {  // register-blocked fallback (one job): largest work-group whose window fits
                if (P.ref_order) {  // parity mode: the reference's tap-ordered multiply-then-add dot product, one output per work-item
                    if (jb.nout > 0) { launch(c, vfo_fir_direct_kernel<true>, dim3((unsigned)std::min((jb.nout + 255) / 256, 4096), 1), dim3(256), 0, (const FirBJob*)d_fj[s]); }
                    continue;
                }
                const int R = SDRPP_FIR_R;
                int threads = 256;
                auto lds_for = [&](int nt) { return (size_t)(1 << jb.log2_decim) * R * (size_t)(nt + jb.kp_pad / R + 1) * 2 * 4; };
                while (ticking && threads > 64 && lds_for(threads) > (size_t)c->tick_lds_cap_fir) { threads >>= 1; }
                while (threads >= 32 && lds_for(threads) > (size_t)kMaxLds) { threads >>= 1; }
                if (threads < 32) { return fail(c, SDRPP_ERR_UNSUPPORTED, "pre-processing FIR (decim %d, %d taps) does not fit in LDS", 1 << jb.log2_decim, jb.ntaps); }
                if (jb.nout > 0) { emit(c, level, F_MISC, TR_FIRB_C, (jb.nout + threads * R - 1) / (threads * R), 1, lds_for(threads), d_fj[s], nullptr, threads); }
            }
            if (ticking) { emit_carries(level); }
        }
        if (!dc.empty() && P.ref_order) {  // parity mode: the sequential recursion itself
            if (n_out > 0) { launch(c, iq_dc_block_exact_kernel, dim3(1), dim3(64), 0, dc[0].in, dc[0].out, n_out, P.dc_rate, dc[0].state_out, P.conj); }
        }
        else if (!dc.empty() && dc[0].nseg > 0) {
            emit(c, P.n_stages + 1, F_MISC, TR_DC_P0, dc[0].nseg, 1, 3 * 256 * sizeof(float), d_dc);
            emit(c, P.n_stages + 2, F_MISC, TR_DC_P1, dc[0].nseg, 1, 3 * 256 * sizeof(float), d_dc);
        }
        else if (!conj.empty()) { emit(c, P.n_stages + 1, F_MISC, TR_COPY, std::max(1, std::min((n_out + 2047) / 2048, 256)), 1, 0, d_conj); }
        if (!ticking) { emit_carries(std::max(1, last_level)); }
    }
    if (P.raw.hist_len > 0) { P.raw.cur ^= 1; }
    for (int s = 0; s + 1 < P.n_stages; s++) {
        if (P.st[(size_t)s].hist_len > 0) { P.st[(size_t)s].cur ^= 1; }
    }
    P.last = result;
    P.last_n = n_out;
    *d_iq = result;
    *count = n_out;
    if (ticking) {
        c->plan_lvl0 = last_level;
        c->plan_top = std::max(c->plan_top, last_level + 2);
    }
    return SDRPP_OK;
}

}  // namespace
