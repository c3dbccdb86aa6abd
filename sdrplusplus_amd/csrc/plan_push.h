// One ordinary pass: snapshot / restore of the streaming state (a block happens completely or not at all), reference-block bounds, push_common.
// Part of the one translation unit sdrpp_gpu.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

// ---- streaming state that planning a block changes, for roll-back: a block either happens completely or not at all ----------------------
struct PlanSnapshot {
    struct V { int soff[SDRPP_MAX_DECIM_STAGES]; int pphase, poff; double phi, phi2; long long seen; int i_if, lvl_if, lvl_out, lvl_af; int n[24], cur[24]; size_t nrecs;
               int af_soff[SDRPP_MAX_DECIM_STAGES], af_pphase, af_poff, af_last, af_state; };
    std::vector<V> v;
    int64_t fft_pos, fft_next;
    int n_lines, iq_cur, wf_cur, wf_lines;
    bool wf_have;
    int pre_soff[SDRPP_MAX_DECIM_STAGES], pre_state, pre_raw_cur, pre_cur[SDRPP_MAX_DECIM_STAGES];
};
// `rotate`: pipelined mode — every stream of the VFO moves on to its next ring buffer in the same walk (stream_rotate)
void plan_snapshot(sdrpp_ctx* c, PlanSnapshot& S, bool rotate = false) {
    S.v.resize(c->vfo_list.size());
    for (size_t i = 0; i < c->vfo_list.size(); i++) {
        Vfo& v = *c->vfo_list[i];
        PlanSnapshot::V& q = S.v[i];
        for (int k = 0; k < SDRPP_MAX_DECIM_STAGES; k++) { q.soff[k] = v.soff[k]; q.af_soff[k] = v.af.soff[k]; }
        q.pphase = v.pphase; q.poff = v.poff; q.phi = v.phi; q.phi2 = v.phi2; q.seen = v.seen; q.i_if = v.i_if; q.lvl_if = v.lvl_if; q.lvl_out = v.lvl_out; q.lvl_af = v.lvl_af;
        q.nrecs = v.recs.size();
        q.af_pphase = v.af.pphase; q.af_poff = v.af.poff; q.af_last = v.af.i_last; q.af_state = v.af.state_cur;
        for (size_t k = 0; k < v.st.size() && k < 24; k++) { q.n[k] = v.st[k].n; q.cur[k] = v.st[k].cur; }
        if (rotate) {
            for (auto& st : v.st) { stream_rotate(st); }
        }
    }
    S.fft_pos = c->fft_pos; S.fft_next = c->fft_next; S.n_lines = c->n_lines; S.iq_cur = c->iq_cur;
    S.wf_cur = c->wf.cur; S.wf_lines = c->wf.lines; S.wf_have = c->wf.have_latest;
    for (int k = 0; k < SDRPP_MAX_DECIM_STAGES; k++) { S.pre_soff[k] = c->pre.soff[k]; }
    S.pre_state = c->pre.state_cur;
    S.pre_raw_cur = c->pre.raw.cur;
    for (size_t k = 0; k < c->pre.st.size() && k < SDRPP_MAX_DECIM_STAGES; k++) { S.pre_cur[k] = c->pre.st[k].cur; }
}
// (retune records a plan has dropped stay dropped: they were out of every window's reach)
void plan_restore(sdrpp_ctx* c, const PlanSnapshot& S) {
    for (size_t i = 0; i < c->vfo_list.size(); i++) {
        Vfo& v = *c->vfo_list[i];
        const PlanSnapshot::V& q = S.v[i];
        for (int k = 0; k < SDRPP_MAX_DECIM_STAGES; k++) { v.soff[k] = q.soff[k]; v.af.soff[k] = q.af_soff[k]; }
        v.pphase = q.pphase; v.poff = q.poff; v.phi = q.phi; v.phi2 = q.phi2; v.seen = q.seen; v.i_if = q.i_if; v.lvl_if = q.lvl_if; v.lvl_out = q.lvl_out; v.lvl_af = q.lvl_af;
        v.af.pphase = q.af_pphase; v.af.poff = q.af_poff; v.af.i_last = q.af_last; v.af.state_cur = q.af_state;
        for (size_t k = 0; k < v.st.size() && k < 24; k++) { v.st[k].n = q.n[k]; v.st[k].cur = q.cur[k]; }
    }
    c->fft_pos = S.fft_pos; c->fft_next = S.fft_next; c->n_lines = S.n_lines; c->iq_cur = S.iq_cur;
    c->wf.cur = S.wf_cur; c->wf.lines = S.wf_lines; c->wf.have_latest = S.wf_have;
    for (int k = 0; k < SDRPP_MAX_DECIM_STAGES; k++) { c->pre.soff[k] = S.pre_soff[k]; }
    c->pre.state_cur = S.pre_state;
    c->pre.raw.cur = S.pre_raw_cur;
    for (size_t k = 0; k < c->pre.st.size() && k < SDRPP_MAX_DECIM_STAGES; k++) { c->pre.st[k].cur = S.pre_cur[k]; }
}

void block_bounds(sdrpp_ctx* c, int64_t count, const std::vector<int>* push_ends) {
    c->arena_begins++;
    c->arena_allocs = 0;
    // the reference's blocks inside this push (sdrpp_set_reference_block): ends as cumulative sample counts
    // every push is at least one block of its own; with a reference block size it is cut further
    std::vector<int>& B = c->vfo_bounds;
    B.clear();
    const std::vector<int> whole{ (int)count };
    int64_t lo = 0;
    for (int e : (push_ends ? *push_ends : whole)) {
        if (c->ref_block > 0) {
            for (int64_t q = lo + c->ref_block; q < e; q += c->ref_block) { B.push_back((int)q); }
        }
        if (e > lo || B.empty()) { B.push_back(e); }
        lo = e;
    }
}
int iq_hist_need(sdrpp_ctx* c) {
    int need_hist = 1;
    if (c->fft_on) { need_hist = std::max(need_hist, c->nz - 1); }
    for (auto& kv : c->vfos) {
        const sdrpp_vfo_desc& d = kv.second->d;
        if (d.n_stages > 0) { need_hist = std::max(need_hist, d.stage_ntaps[0] - 1); }
        if (d.n_stages > 1) { need_hist = std::max(need_hist, d.stage_ntaps[0] - 1 + d.stage_decim[0] * (d.stage_ntaps[1] - 1)); }  // fused front
    }
    return need_hist;
}
// what the NEXT push can reach back to: the samples of the frame in progress and the deepest stage-1 (+ fused stage-2) window
CarryJob iq_carry_job(sdrpp_ctx* c, const float* d_iq, int64_t count) {
    int need = 1;
    if (c->fft_on) {
        const int64_t P = (int64_t)c->nz + c->skip;
        const int64_t partial = c->fft_pos - c->fft_next * P;  // do_fft already advanced both
        if (partial > 0) { need = std::max(need, (int)std::min<int64_t>(partial, c->nz - 1)); }
    }
    for (auto& kv : c->vfos) {
        const sdrpp_vfo_desc& d = kv.second->d;
        if (d.n_stages > 0) { need = std::max(need, d.stage_ntaps[0] - 1); }
        if (d.n_stages > 1) { need = std::max(need, d.stage_ntaps[0] - 1 + d.stage_decim[0] * (d.stage_ntaps[1] - 1)); }
    }
    need = std::min(need, c->iq_hist_cap);
    return CarryJob{ d_iq, c->iq_hist[c->iq_cur], c->iq_hist[c->iq_cur ^ 1], c->iq_hist_cap, (int)count, 2, need };
}

// `push_ends`: cumulative ends of the pushes a deferred pass combines (nullptr: the pass is one push)
int push_common(sdrpp_ctx* c, const float* d_iq, int64_t count, const std::vector<int>* push_ends = nullptr) {
    if (count == 0) {  // an empty block produces nothing (and changes no state)
        c->n_lines = 0;
        for (auto& kv : c->vfos) {
            for (auto& s : kv.second->st) { s.n = 0; }
        }
        return SDRPP_OK;
    }
    int rc0;
    {
        HostScope hs("arena_begin (backpressure)");
        rc0 = arena_begin(c);
    }
    if (rc0) { return rc0; }
    static thread_local PlanSnapshot snap;  // (re-used: with 128 VFOs a fresh one is a 38 KB allocation per block; a nested pass starts only after the outer plan has been restored)
    plan_snapshot(c, snap);
    c->plan_lvl0 = 0;
    block_bounds(c, count, push_ends);
    if (c->pre.on) {
        rc0 = run_preproc(c, &d_iq, &count);
        if (rc0) {
            plan_restore(c, snap);
            return rc0;
        }
        if (count == 0) {  // the decimator swallowed the whole block (offset carried): nothing reaches the FFT / VFOs
            c->n_lines = 0;
            for (auto& kv : c->vfos) {
                for (auto& s : kv.second->st) { s.n = 0; }
            }
            return arena_end(c);
        }
    }
    int rc = ensure_iq_hist(c, iq_hist_need(c));
    if (rc) {
        plan_restore(c, snap);
        return rc;
    }
    IqSrc src{ (const float2*)d_iq, (const float2*)c->iq_hist[c->iq_cur], c->iq_hist_cap, (long long)count };
    // fork: the FFT branch goes to its own stream and overlaps the VFO bank; both only read the IQ buffers
    // (a push that completes no frame launches nothing on the FFT branch: no fork / join either — each costs the main stream 5-9 us, and at
    // the reference's block size every fourth push of a 65 536-point waterfall is such a push)
    bool fft_work = false;
    if (c->fft_on) {
        const int64_t P = (int64_t)c->nz + c->skip;
        fft_work = (c->fft_pos + count) - c->nz - c->fft_next * P >= 0;
    }
    const bool fork = fft_work && !c->vfos.empty();
    if (fork) {
        HostScope hs("fork events");
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->fft_stream, c->ev_fork, 0));
        c->launch_stream = c->fft_stream;
    }
    {
        HostScope hs("do_fft");
        rc = do_fft(c, src, count);
    }
    c->launch_stream = c->stream;
    if (fork) { (void)hipEventRecord(c->ev_join, c->fft_stream); }
    if (!rc) {
        const CarryJob iqc = iq_carry_job(c, d_iq, count);
        if (c->vfos.empty()) {
            std::vector<CarryJob> carry{ iqc };
            CarryJob* d_carry = arena_push(c, carry);
            rc = arena_commit(c);
            if (!rc) {
                FamilyTimer t(c, F_MISC);
                const int iq_elems = carry[0].need * 2;
                launch(c, carry_kernel, dim3((unsigned)std::max(1, std::min((iq_elems + 8191) / 8192, 512)), 1), dim3(256), 0, (const CarryJob*)d_carry, 0);  // (carry_body: 8 192 floats per workgroup and round)
            }
        }
        else {
            HostScope hs("do_vfos");
            rc = do_vfos_plan(c, src, count, iqc);
        }
    }
    HostScope hs2("join + arena_end");
    if (fork) { (void)hipStreamWaitEvent(c->stream, c->ev_join, 0); }  // (also after a failure: the FFT branch's launches must not overtake what follows)
    if (rc) {
        // the block did not happen: the streaming state is what it was before the push (what the kernels launched so far wrote is never
        // looked at: counts, history sides and frame positions are the host's)
        plan_restore(c, snap);
        (void)arena_end(c);
        return rc;
    }
    c->iq_cur ^= 1;
    rc = arena_end(c);
    if (rc) { return rc; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { return fail(c, SDRPP_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e)); }
    return SDRPP_OK;
}

}  // namespace
