// IQFrontEnd pre-processing chain of one block (PowerDecimator -> DCBlocker -> Conjugate, iq_frontend.cpp:32-39).
// Part of the one translation unit sdrpp_gpu.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

// ---- IQFrontEnd pre-processing chain: one push ------------------------------------------------------------------------------------
// Decimator stages run on the matrix-core FIR kernel (register-blocked VALU kernel for tap counts it does not cover), the DC
// blocker as a two-level scan (vfo_deemph_kernel<1, *>), the conjugate inside its store (or alone).  On return *d_iq / *count
// describe the pre-processed stream; the raw and stage histories are carried for the next push.
int run_preproc(sdrpp_ctx* c, const float** d_iq, int64_t* count) {
    sdrpp_ctx::Pre& P = c->pre;
    const int n_in = (int)*count;
    std::vector<ToepJob> tj[SDRPP_MAX_DECIM_STAGES];
    std::vector<FirBJob> fj[SDRPP_MAX_DECIM_STAGES];
    std::vector<CarryJob> carry;
    std::vector<DeempJob> dc;
    P.raw.data = const_cast<float*>(*d_iq);
    P.raw.n = n_in;
    Stream* cur = &P.raw;
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
        if (cur->hist_len > 0) { carry.push_back(CarryJob{ cur->data, cur->hist[cur->cur], cur->hist[cur->cur ^ 1], cur->hist_len, cur->n, 2, cur->hist_len }); }
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
    else if (P.conj) { result = P.out.data; }
    // job tables
    ToepPlan tp[SDRPP_MAX_DECIM_STAGES];
    ToepJob* d_tj[SDRPP_MAX_DECIM_STAGES] = {};
    FirBJob* d_fj[SDRPP_MAX_DECIM_STAGES] = {};
    for (int s = 0; s < P.n_stages; s++) {
        tp[s] = toep_plan(tj[s], 2);
        d_tj[s] = arena_push(c, tj[s]);
        d_fj[s] = arena_push(c, fj[s]);
        if ((!tj[s].empty() && !d_tj[s]) || (!fj[s].empty() && !d_fj[s])) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
    }
    DeempJob* d_dc = arena_push(c, dc);
    CarryJob* d_carry = arena_push(c, carry);
    if ((!dc.empty() && !d_dc) || (!carry.empty() && !d_carry)) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
    int rc = arena_commit(c);
    if (rc) { return rc; }
    {
        FamilyTimer t(c, F_MISC);
        for (int s = 0; s < P.n_stages; s++) {
            launch_toep(c, tj[s], d_tj[s], tp[s], 2, false);
            for (auto& jb : fj[s]) {  // register-blocked fallback (one job): largest work-group whose window fits
                if (P.ref_order) {  // parity mode: the reference's tap-ordered multiply-then-add dot product, one output per work-item
                    if (jb.nout > 0) { launch(c, vfo_fir_direct_kernel<true>, dim3((unsigned)std::min((jb.nout + 255) / 256, 4096), 1), dim3(256), 0, (const FirBJob*)d_fj[s]); }
                    continue;
                }
                const int R = SDRPP_FIR_R;
                int threads = 256;
                auto lds_for = [&](int nt) { return (size_t)(1 << jb.log2_decim) * R * (size_t)(nt + jb.kp_pad / R + 1) * 2 * 4; };
                while (threads >= 32 && lds_for(threads) > (size_t)kMaxLds) { threads >>= 1; }
                if (threads < 32) { return fail(c, SDRPP_ERR_UNSUPPORTED, "pre-processing FIR (decim %d, %d taps) does not fit in LDS", 1 << jb.log2_decim, jb.ntaps); }
                if (jb.nout > 0) { launch(c, vfo_firb_kernel<2, false>, dim3((unsigned)((jb.nout + threads * R - 1) / (threads * R)), 1), dim3(threads), lds_for(threads), (const FirBJob*)d_fj[s]); }
            }
        }
        if (!dc.empty() && P.ref_order) {  // parity mode: the sequential recursion itself
            if (n_out > 0) { launch(c, iq_dc_block_exact_kernel, dim3(1), dim3(64), 0, dc[0].in, dc[0].out, n_out, P.dc_rate, dc[0].state_out, P.conj); }
        }
        else if (!dc.empty() && dc[0].nseg > 0) {
            const dim3 grid((unsigned)dc[0].nseg, 1);
            launch(c, vfo_deemph_kernel<1, 0>, grid, dim3(256), 0, (const DeempJob*)d_dc);
            launch(c, vfo_deemph_kernel<1, 1>, grid, dim3(256), 0, (const DeempJob*)d_dc);
        }
        else if (dc.empty() && P.conj && n_out > 0) {
            launch(c, iq_conjugate_kernel, dim3((unsigned)std::min((n_out + 255) / 256, 4096)), dim3(256), 0, (const float2*)cur->data, (float2*)P.out.data, n_out);
        }
        if (!carry.empty()) {
            int mx = 0;
            for (auto& k : carry) { mx = std::max(mx, k.need * k.width); }
            launch(c, carry_kernel, dim3((unsigned)std::max(1, std::min((mx + 255) / 256, 64)), (unsigned)carry.size()), dim3(256), 0, (const CarryJob*)d_carry);
        }
    }
    if (P.raw.hist_len > 0) { P.raw.cur ^= 1; }
    for (int s = 0; s + 1 < P.n_stages; s++) {
        if (P.st[(size_t)s].hist_len > 0) { P.st[(size_t)s].cur ^= 1; }
    }
    P.last = result;
    P.last_n = n_out;
    *d_iq = result;
    *count = n_out;
    return SDRPP_OK;
}

}  // namespace
