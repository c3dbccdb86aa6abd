// VFO bank of one block: per-VFO chains -> job lists per level -> front-end grouping -> launches / roles.
// Part of the one translation unit sdrpp_gpu.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

// ---- VFO bank: one push ------------------------------------------------------------------------------------------------------------
struct S1Member { Vfo* v; int K, lgD, off0, nout; double phi0; int min_idx; int fused, K2, lgD2, off2, nout2; unsigned long long taph;
                  int shift = 0, oshift = 0; };  // a push of a launch group: where its samples start in the group's block, where its outputs start in the group's output block

// Cache key of a front-end job's tap operand: membership (VFO ids) and NCO increments, as two independent 64-bit hashes (the job
// tables are rebuilt on every push: formatting 32 ids and doubles into a string cost more host time than the launch itself)
std::string member_key(char kind, const S1Member* m, int n) {
    unsigned long long h1 = 1469598103934665603ull ^ (unsigned char)kind, h2 = 0x9e3779b97f4a7c15ull + (unsigned char)kind;
    for (int i = 0; i < n; i++) {
        unsigned long long tb;
        memcpy(&tb, &m[i].v->theta, 8);
        const unsigned long long id = (unsigned long long)(unsigned)m[i].v->id;
        h1 = (h1 ^ id) * 1099511628211ull;
        h1 = (h1 ^ tb) * 1099511628211ull;
        h2 ^= id + 0x9e3779b97f4a7c15ull + (h2 << 6) + (h2 >> 2);
        h2 ^= tb + 0x9e3779b97f4a7c15ull + (h2 << 6) + (h2 >> 2);
    }
    char raw[16];
    memcpy(raw, &h1, 8);
    memcpy(raw + 8, &h2, 8);
    return std::string(raw, 16);
}

// Stage-2 outputs per block of the fused front kernel (0 = do not fuse: the recomputed overlap would dominate or LDS would overflow).
int front2_t2(int K1, int D1, int K2, int D2, int vt) {
    const int tile = 256;
    if (K2 >= tile) { return 0; }
    const int t2 = (tile - K2) / D2 + 1;
    if (t2 * D2 * 4 < tile * 3) { return 0; }  // more than 25 % of the stage-1 work would be recomputed overlap
    if (D2 < 2) { return 0; }
    const size_t lds = (std::max((size_t)D1 * (tile + (K1 - 1 + D1 - 1) / D1 + 1), (size_t)vt * (tile + 16)) + (size_t)vt) * sizeof(float2);
    return lds <= (size_t)kMaxLds ? t2 : 0;
}


// Matrix-core front kernel (composite stage 1 + 2 filter, one 32-output x 32-VFO tile per wavefront step): usable?  Picks the
// prefetch depth (IQ samples per lane) of the template variant.
bool frontcm_ok(int K1, int lgD1, int K2, int lgD2, int* pf) {
    const int K = K1 + (K2 - 1) * (1 << lgD1), lgD = lgD1 + lgD2;
    if (K < 9 || lgD < 1 || lgD > 5) { return false; }
    const int nsamp = (SDRPP_FCM_TILE - 1) * (1 << lgD) + K;
    if (nsamp > 16 * 64) { return false; }
    *pf = nsamp <= 6 * 64 ? 6 : (nsamp <= 10 * 64 ? 10 : 16);
    return (size_t)frontcm_layout(K, lgD).total * 4 <= (size_t)(160 * 1024 / 3);  // three blocks per CU
}

// ---- matrix-core FIR launches (vfo_toep_kernel): job construction, per-list planning (macro tiles per wavefront, grid, LDS), launch ----
ToepJob toep_job(const ToepTab& T, int var, StreamIn in, float* out, int base0, int nout, float inv_dev) {
    ToepJob j{};
    j.in = in;
    j.out = out;
    j.tl = T.d_tl;
    j.lbase = T.d_lb + (size_t)var * 64;
    j.tl_len = T.tl_len;
    j.nsteps = T.nsteps;
    j.s_in = T.s_in;
    j.rows = T.rows;
    j.base0 = base0;
    j.nout = nout;
    j.mt_per_wave = 1;
    j.inv_deviation = inv_dev;
    return j;
}

struct ToepPlan { int grid_x = 0; size_t lds = 0; };
ToepPlan toep_plan(std::vector<ToepJob>& jobs, int npl, int max_blocks = 2048) {
    ToepPlan P;
    if (jobs.empty()) { return P; }
    const int G = 2;
    int mtw = 1;
    // about two resident rounds (256 CUs x 4 blocks of four wavefronts, each job padded to whole blocks): alone the kernels do not
    // care (1 024 ... 8 192 blocks measured equal), but blocks that end let the FFT branch's blocks in — 2.5 % on the whole step
    for (; mtw < 16; mtw++) {
        size_t blocks = 0;
        for (auto& jb : jobs) { blocks += (size_t)((jb.nout + G * 16 * jb.rows - 1) / (G * 16 * jb.rows) + 4 * mtw - 1) / (size_t)(4 * mtw); }
        if (blocks <= (size_t)max_blocks) { break; }
    }
    for (auto& jb : jobs) {
        jb.mt_per_wave = mtw;
        const int nmt = (jb.nout + G * 16 * jb.rows - 1) / (G * 16 * jb.rows);
        P.grid_x = std::max(P.grid_x, (nmt + 4 * mtw - 1) / (4 * mtw));
        const int span = (G * 16 - 1) * jb.s_in + 4 * jb.nsteps, pl = (span + 8) & ~3;
        P.lds = std::max(P.lds, ((size_t)((jb.tl_len + 3) & ~3) + (size_t)4 * npl * pl) * sizeof(float));
    }
    return P;
}

void launch_toep(sdrpp_ctx* c, std::vector<ToepJob>& jobs, ToepJob* d_jobs, const ToepPlan& P, int width, bool quad) {
    if (jobs.empty() || P.grid_x == 0) { return; }
    const dim3 grid((unsigned)P.grid_x, (unsigned)jobs.size());
    if (quad) { launch(c, vfo_toep_kernel<1, 2, true>, grid, dim3(256), P.lds, (const ToepJob*)d_jobs); }
    else if (width == 2) { launch(c, vfo_toep_kernel<2, 2, false>, grid, dim3(256), P.lds, (const ToepJob*)d_jobs); }
    else { launch(c, vfo_toep_kernel<1, 2, false>, grid, dim3(256), P.lds, (const ToepJob*)d_jobs); }
}

// ---- pipelined FM back end (vfo_pipe_kernel) ----
constexpr int kPipeG = 1;    // groups of 16 tiles per macro tile: one keeps a job's LDS at ~30 KB, five workgroups per CU
constexpr int kPipeBpc = 5;     // workgroups per CU the kernel is built for (launch bounds; jobs of ~30 KB)
constexpr int kPipeBpcMin = 3;  // longer filters (NFM's 300-tap channel / audio filters: ~38 KB) run with fewer workgroups per CU
// LDS layout of one job; false = does not fit / stage 0 not register-staged -> the VFO keeps its separate launches
bool pipe_layout_try(PipeJob& J, int fifo_tiles, size_t* lds_bytes) {
    const int G = kPipeG;
    int off = 0;
    auto take = [&](int nfloats) { const int o = off; off += (nfloats + 3) & ~3; return o; };
    for (int s = 0; s < 4; s++) { J.tl_off[s] = take(J.st[s].tl_len); }
    int W[4], span[4], omt[4];
    for (int s = 0; s < 4; s++) {
        W[s] = G * 16 * J.st[s].s_in;
        span[s] = W[s] - J.st[s].s_in + 4 * J.st[s].nsteps;
        omt[s] = G * 16 * J.st[s].rows;
    }
    if (((span[0] + 1) >> 1) > (G == 1 ? 5 : 9) * 64) { return false; }  // stage 0's window is register-staged: five (G = 1) / nine sample pairs per lane
    J.win_off = take(2 * ((span[0] + 8) & ~3));
    J.zero_lo = off;
    for (int i = 0; i < 3; i++) {
        // R >= W + h + what the producer writes at a time, a multiple of W: windows start at multiples of W, no stall cycle.  (i = 2:
        // stage 3's own ring of discriminator outputs, which it fills a stage-2 macro tile at a time between two matrix loops.)
        const int s = i + 1, h = span[s] - W[s];
        if (h < 0) { return false; }
        const int R = std::max(2, (W[s] + h + omt[i] + W[s] - 1) / W[s]) * W[s];
        if (i < 2) {
            J.ring_len[i] = R;
            J.ring_mir[i] = h;
            J.ring_off[i] = take(2 * (R + h));
        }
        else {
            J.dring_len = R;
            J.dring_mir = h;
            J.dring_off = take(R + h);
        }
    }
    J.ring_len[2] = fifo_tiles * omt[2];  // IF phases on their way to the discriminator's difference: a plain FIFO of whole stage-2 macro tiles
    J.ring_mir[2] = 0;
    J.ring_off[2] = take(J.ring_len[2]);
    J.zero_hi = off;
    J.flag_off = take(8);
    *lds_bytes = (size_t)off * sizeof(float);
    return *lds_bytes <= (size_t)(160 * 1024) / kPipeBpcMin;
}
bool pipe_layout(PipeJob& J, size_t* lds_bytes) { return pipe_layout_try(J, 2, lds_bytes) || pipe_layout_try(J, 1, lds_bytes); }
// Segments per VFO, 0 = this push is better served by the separate launches.  A segment pays one warm-up macro tile per stage and the
// pipeline's fill: with fewer than ~12 last-stage macro tiles per segment of a full grid the four launches win (measured: 1 M-sample
// pushes of the 32-VFO bank, 2.6 tiles per segment, 14 % slower) — unless the push is so small that it is launch-bound anyway.
int pipe_segments(const std::vector<PipeJob>& pipes, int forced, size_t lds, int tick_blocks = 0) {
    if (pipes.empty()) { return 0; }
    const int bpc = std::max(kPipeBpcMin, std::min(kPipeBpc, (int)((size_t)(160 * 1024) / std::max<size_t>(lds, 1))));
    int max_nmt = 1;
    for (auto& pj : pipes) { max_nmt = std::max(max_nmt, (pj.st[3].nout + kPipeG * 16 * pj.st[3].rows - 1) / (kPipeG * 16 * pj.st[3].rows)); }
    if (forced >= 2) { return std::min(forced, max_nmt); }
    if (tick_blocks > 0) {
        // a role of the tick kernel (pipelined mode): ALWAYS the pipeline — whether a VFO's back end runs as one role or as four is a property of
        // its filters, never of the block, because the streams between the stages are handed from block to block differently in the two forms
        // (do_vfos_plan) — with about `tick_blocks` workgroups for the whole role (a tick's roles share the GPU) and at least two last-stage
        // macro tiles per segment (every segment pays one warm-up macro tile per stage)
        return std::max(1, std::min((tick_blocks + (int)pipes.size() - 1) / (int)pipes.size(), (max_nmt + 1) / 2));
    }
    const int s_full = (256 * bpc + (int)pipes.size() - 1) / (int)pipes.size();
    if (max_nmt >= 12 * s_full) { return s_full; }
    if (max_nmt <= 16) { return std::max(1, (max_nmt + 1) / 2); }  // two macro tiles per workgroup: the chain of hand-offs is what a small push waits for (B = 50 000: 53 us per push with one segment, 48.5 with three)
    return 0;
}

// ---- levels: the position of a launch in the data flow of one block ---------------------------------------------------------------------
constexpr int kLevels = 28;
template <class T>
struct Lev {
    std::vector<T> at[kLevels];
    T* dev[kLevels] = {};
    int top = 0;  // highest level in use + 1
    void add(int l, const T& j) {
        if (l >= kLevels) { l = kLevels - 1; }
        at[l].push_back(j);
        if (l + 1 > top) { top = l + 1; }
    }
};
template <class T>
bool arena_push_lev(sdrpp_ctx* c, Lev<T>& L) {
    for (int l = 0; l < L.top; l++) {
        if (L.at[l].empty()) { continue; }
        L.dev[l] = arena_push(c, L.at[l]);
        if (!L.dev[l]) { return false; }
    }
    return true;
}

// One block of the VFO bank, planned in steps: every VFO's chain is walked once (chain: stage by stage — outputs per stage from the integer
// streaming state, one job per stage in the list of its kind and LEVEL), the first stages are grouped into front-end jobs (group_front),
// the job tables go into the arena in one piece (upload), then the launches — or, in pipelined mode, the roles — are emitted level by level
// (emit_front, emit_levels).  Every job carries the LEVEL of its launch in the block's data flow (level L reads what level L - 1 wrote): the
// front end is level 1 (level 0 = the block's arrival), every filter behind it one more.  A pass launches level by level; in pipelined mode
// level L of this block runs L ticks from now (tick_kernels.h).
// One BankPlan lives with its context and is re-used block after block (begin() empties the lists and keeps their storage: with 128 VFOs a
// fresh plan per block was ~1 000 small allocations, a third of the 225 us of host time that had become cfg 4's limit once its tick took 135 us).
struct BankPlan {
    sdrpp_ctx* c;
    IqSrc src{};
    int n_in = 0;
    bool ticking = false;
    int L0 = 0;  // levels the pre-processing chain takes in front (pipelined mode): the front end runs at level L0 + 1
    static constexpr int carry_last = kLevels - 1;
    static constexpr int carry_wave_max = 8192;  // floats: up to four rounds of a wavefront
    const std::vector<int>& fb;  // reference-block ends of this push (at least one entry: n_in)
    bool blocks = false;
    std::vector<S1Member> s1;
    std::vector<RotJob> rot;
    Lev<FirBJob> f_dec;  // register-blocked decimators (tap counts the matrix form does not cover; stage 0 only in reference-rotator mode)
    std::vector<RotXJob> rotx;                          // reference-rotator mode: full-rate float recursion, one lane per VFO
    std::vector<RetuneJob> retune;                      // closed-form NCO: first outputs after a setOffset
    Lev<PolyJob> poly;
    Lev<PolyBJob> polyb[4];  // [0]: LMAX 4, [1]: LMAX 8 (de-interleaved tile); [2], [3]: same with odd decimation (linear tile)
    Lev<FirBJob> chan;
    Lev<SeqJob> seq;
    Lev<PreJob> pre;
    Lev<FirBJob> audio;     // AM: real stream -> low-pass -> stereo
    Lev<FirBJob> audio_fm;  // WFM/NFM: IF -> discriminator -> low-pass -> stereo, one kernel
    // the same work on the matrix cores (vfo_toep_kernel) whenever the VFO has a tap table for it
    Lev<ToepJob> t_dec, t_poly, t_chan, t_audio, t_audio_fm;
    std::vector<PipeJob> pipes;  // FM back ends that run as one pipelined launch (all at the level of their decimator)
    size_t pipe_lds = 0;
    // radio AF chain (stereo frames have the layout of complex samples, so the same kernels serve)
    Lev<ToepJob> t_af_dec, t_af_poly, t_af_hpf;
    Lev<FirBJob> af_dec, af_hpf;
    Lev<PolyJob> af_poly;
    Lev<DeempJob> af_deemp;
    Lev<SsbRotXJob> ssbx_l;
    Lev<CarryJob> carry;  // history carries at the level of the stream's consumer (a pass without pipelining: all at the last level)
    int max_rot = 0;
    // front-end jobs (group_front)
    struct S1Launch { int vt; std::vector<Stage1Job> jobs; int max_nout = 0; int tile = 256; size_t lds = 0; };
    struct F2Launch { int vt; std::vector<Front2Job> jobs; int max_blocks = 0; size_t lds = 0; };
    struct FCMLaunch { std::vector<FrontCMJob> jobs; int max_blocks = 0; size_t lds = 0; bool w16 = false; int w16_blocks = 0; };
    S1Launch s1l[4];
    F2Launch f2l[4];
    FCMLaunch fcm[3];  // PF 6 / 10 / 16
    FCMLaunch fcl;     // long first stages (vfo_frontcl_kernel)
    int fcl_nw = 2;    // tile engines per workgroup of that launch: 4 as a role of a tick when four wavefronts' planes fit half a CU's LDS
    const int vts[4] = { 8, 4, 2, 1 };
    // device addresses of the job tables (upload)
    Stage1Job* d_s1[4] = {};
    Front2Job* d_f2[4] = {};
    FrontCMJob* d_fcl = nullptr;
    FrontCMJob* d_fcm[3] = {};
    RotXJob* d_rotx = nullptr;
    RotXHead* d_rotx_head = nullptr;  // pipelined: what the TR_ROTX16 role finds behind its job pointer
    RetuneJob* d_retune = nullptr;
    RotJob* d_rot = nullptr;
    const int* d_fb = nullptr;
    struct PipeGroup { int lvl = 0; std::vector<PipeJob> jobs; PipeJob* dev = nullptr; int seg = 0; };
    std::vector<PipeGroup> pgroups;
    int pipe_top = 0;
    struct ToepList { Lev<ToepJob>* L; int npl, width; bool quad; int fam; int role; };
    static constexpr int kToepLists = 8;
    ToepList tlists[kToepLists] = { { &t_dec, 2, 2, false, F_DECIM, TR_TOEP_C },      { &t_poly, 2, 2, false, F_POLY, TR_TOEP_C },       { &t_chan, 2, 2, false, F_FIR, TR_TOEP_C },
                                    { &t_audio, 1, 1, false, F_FIR, TR_TOEP_R },      { &t_audio_fm, 2, 1, true, F_FIR, TR_TOEP_Q },     { &t_af_dec, 2, 2, false, F_AF, TR_TOEP_C },
                                    { &t_af_poly, 2, 2, false, F_AF, TR_TOEP_C },     { &t_af_hpf, 2, 2, false, F_AF, TR_TOEP_C } };
    ToepPlan tplan[kToepLists][kLevels];

    explicit BankPlan(sdrpp_ctx* c_) : c(c_), fb(c_->vfo_bounds) {
        for (int i = 0; i < 4; i++) { s1l[i].vt = vts[i]; f2l[i].vt = vts[i]; }
    }
    template <class T>
    static void lev_reset(Lev<T>& L) {
        for (int l = 0; l < L.top; l++) {
            L.at[l].clear();
            L.dev[l] = nullptr;
        }
        L.top = 0;
    }
    // a new block: every list empty (storage kept), every per-block scalar back to its initial value
    void begin(const IqSrc& src_, int64_t count, const CarryJob& iq_carry) {
        src = src_;
        n_in = (int)count;
        ticking = c->tick_planning;
        L0 = ticking ? c->plan_lvl0 : 0;
        blocks = fb.size() > 1;
        s1.clear(); rot.clear(); rotx.clear(); retune.clear(); pipes.clear(); pgroups.clear();
        lev_reset(f_dec); lev_reset(poly);
        for (auto& q : polyb) { lev_reset(q); }
        lev_reset(chan); lev_reset(seq); lev_reset(pre); lev_reset(audio); lev_reset(audio_fm);
        lev_reset(t_dec); lev_reset(t_poly); lev_reset(t_chan); lev_reset(t_audio); lev_reset(t_audio_fm);
        lev_reset(t_af_dec); lev_reset(t_af_poly); lev_reset(t_af_hpf); lev_reset(af_dec); lev_reset(af_hpf); lev_reset(af_poly); lev_reset(af_deemp);
        lev_reset(ssbx_l); lev_reset(carry);
        pipe_lds = 0;
        max_rot = 0;
        pipe_top = 0;
        fcl_nw = 2;
        for (int i = 0; i < 4; i++) {
            s1l[i].jobs.clear(); s1l[i].max_nout = 0; s1l[i].tile = 256; s1l[i].lds = 0;
            f2l[i].jobs.clear(); f2l[i].max_blocks = 0; f2l[i].lds = 0;
            d_s1[i] = nullptr;
            d_f2[i] = nullptr;
        }
        for (int i = 0; i < 3; i++) {
            fcm[i].jobs.clear(); fcm[i].max_blocks = 0; fcm[i].lds = 0; fcm[i].w16 = false; fcm[i].w16_blocks = 0;
            d_fcm[i] = nullptr;
        }
        fcl.jobs.clear(); fcl.max_blocks = 0; fcl.lds = 0; fcl.w16 = false; fcl.w16_blocks = 0;
        d_fcl = nullptr; d_rotx = nullptr; d_retune = nullptr; d_rot = nullptr; d_fb = nullptr;
        for (auto& row : tplan) {
            for (auto& q : row) { q = ToepPlan{}; }
        }
        carry.add(ticking ? L0 + 1 : carry_last, iq_carry);  // job 0 of its level: the shared IQ stream
    }
    BankPlan(const BankPlan&) = delete;
    BankPlan& operator=(const BankPlan&) = delete;

    // ---- one VFO's chain: stage by stage, a job per stage in the list of its kind and level ----
    int chain(Vfo& v) {
        Stream* cur = &v.st[(size_t)v.i_first];
        int lvl = L0 + 1;  // level at which `cur` is written
        for (auto& s : v.st) { s.clevel = 0; s.wlevel = 0; }
        cur->wlevel = lvl;
        // reference-block ends carried stage by stage down to the demodulator's rate, for the block-dependent operations there
        // (AGC look-ahead, SSB rotator calls)
        const bool agc_mode = v.d.demod == SDRPP_DEMOD_AM || (v.d.demod >= SDRPP_DEMOD_USB && v.d.demod <= SDRPP_DEMOD_DSB);
        const bool need_bnd = agc_mode && (blocks || v.nco_exact);
        std::vector<int> bnd;
        if (need_bnd) { bnd = fb; }
        // a launch group of several pushes (sdrpp_set_pipeline_group): the PUSH ends carried the same way, through every rate change down to the
        // stream the results are read from — which samples of the group's output block belong to which push (tick_results_describe)
        const bool split = c->grp_ends.size() > 1;
        std::vector<int>& tk = v.tk_if;
        tk.clear();
        v.tk_af.clear();
        if (split) { tk = c->grp_ends; }
        int first_sep = 0;  // first decimator stage that runs as its own FIR launch
        double phi_end = 0.0;
        bool have_phi_end = false;  // (a launch group: the NCO phase after its last push, advanced push by push as block-by-block processing does)
        if (v.nco_exact) {
            // the reference's own data flow: rotate at the full rate (float recursion), then every stage of the plan as a plain FIR
            Stream* tgt = (v.d.n_stages == 0) ? cur : &v.st[(size_t)v.i_rot];
            rotx.push_back(RotXJob{ (float2*)tgt->data, v.d_rot, v.d.phase_delta_re, v.d.phase_delta_im });
            tgt->n = n_in;
            tgt->wlevel = lvl;
            cur = tgt;
        }
        else if (v.d.n_stages == 0) {
            rot.push_back(RotJob{ v.theta, v.phi, (float2*)cur->data, n_in });
            cur->n = n_in;
            max_rot = std::max(max_rot, n_in);
        }
        else {
            const int D = v.d.stage_decim[0];
            const int nout = decim_nout(n_in, v.soff[0], D);
            if (v.modtaps_dirty) { build_modtaps(v); }
            const int K0 = v.d.stage_ntaps[0];
            S1Member mem{ &v, K0, ilog2(D), v.soff[0], nout, v.phi, 0, 0, 0, 0, 0, 0, v.tap_hash };
            int need = K0 - 1;
            first_sep = 1;
            if (need_bnd) { bounds_decim(bnd, v.soff[0], D); }
            if (split) { bounds_decim(tk, v.soff[0], D); }
            if (v.fused_front) {
                const int D2 = v.d.stage_decim[1];
                mem.fused = 1;
                mem.K2 = v.d.stage_ntaps[1];
                mem.lgD2 = ilog2(D2);
                mem.off2 = v.soff[1];
                mem.nout2 = decim_nout(nout, v.soff[1], D2);
                need = K0 - 1 + D * (mem.K2 - 1);
                first_sep = 2;
                if (need_bnd) { bounds_decim(bnd, v.soff[1], D2); }
                if (split) { bounds_decim(tk, v.soff[1], D2); }
            }
            mem.min_idx = (v.seen >= need) ? -need : -(int)v.seen;  // older samples predate this VFO: zero
            if (!split) { s1.push_back(mem); }
            else {
                // A launch group: ONE front-end job per push, with the offsets, the NCO phase and the history bound block-by-block processing would
                // have given that push (the closed-form NCO is anchored at the start of a job: tile phasor x in-tile table — one job over the whole
                // group would round the same samples differently), reading its samples where they lie in the group's block and writing its outputs
                // behind those of the pushes in front.  Everything behind the front end is a plain FIR over the stream, indifferent to the cuts.
                int so0 = v.soff[0], so1 = v.soff[1], prev_e = 0, oshift = 0;
                long long seen = v.seen;
                double ph = v.phi;
                for (size_t j = 0; j < c->grp_ends.size(); j++) {
                    const int nj = c->grp_ends[j] - prev_e;
                    S1Member m = mem;
                    m.off0 = so0;
                    m.nout = decim_nout(nj, so0, D);
                    m.phi0 = ph;
                    int outs = m.nout;
                    if (mem.fused) {
                        m.off2 = so1;
                        m.nout2 = decim_nout(m.nout, so1, v.d.stage_decim[1]);
                        outs = m.nout2;
                    }
                    m.min_idx = ((seen >= need) ? -need : -(int)seen) + prev_e;
                    m.shift = prev_e;
                    m.oshift = oshift;
                    if (outs > 0) { s1.push_back(m); }
                    so0 = so0 + m.nout * D - nj;
                    if (mem.fused) { so1 = so1 + m.nout2 * v.d.stage_decim[1] - m.nout; }
                    const double pj = ph + (double)nj * v.theta;
                    ph = pj - std::floor(pj);
                    seen += nj;
                    oshift += outs;
                    prev_e = c->grp_ends[j];
                }
                phi_end = ph;
                have_phi_end = true;
            }
            // setOffset hand-over: outputs whose window still reaches in front of the latest retune point are recomputed with the
            // piecewise phase (vfo_retune_fix_kernel); retune points no window can reach any more are forgotten
            {
                const int D1 = D, Kc = v.h12_K;
                const int off = mem.fused ? mem.off0 + (mem.off2 - (mem.K2 - 1)) * D1 - (K0 - 1) : mem.off0 - (K0 - 1);
                const int nout_f = mem.fused ? mem.nout2 : nout;
                while (!v.recs.empty() && v.recs.front().pos - v.seen <= (long long)off) { v.recs.erase(v.recs.begin()); }
                while (v.recs.size() > SDRPP_RETUNE_MAX_SEG - 1) { v.recs.erase(v.recs.begin()); }
                if (!v.recs.empty() && nout_f > 0) {
                    const long long r_last = v.recs.back().pos - v.seen;  // push-relative, <= 0
                    const long long Dc = 1ll << v.h12_lgD;
                    const int nfix = (int)std::min<long long>((long long)nout_f, (r_last - off + Dc - 1) / Dc);  // outputs m with off + m * Dc < r_last
                    if (nfix > 0) {
                        RetuneJob rj{};
                        rj.out = (float2*)v.st[(size_t)v.i_first + (mem.fused ? 1 : 0)].data;
                        rj.taps = v.d_h12;
                        rj.K = Kc;
                        rj.log2_decim = v.h12_lgD;
                        rj.off = off;
                        rj.nfix = nfix;
                        rj.min_idx = mem.min_idx;
                        const int nr = (int)v.recs.size();
                        rj.nseg = nr + 1;
                        // segment q >= 1 starts at retune point q - 1 and runs with the increment that was in effect from there on (the
                        // newest with the current one); segment 0 = everything in front of the oldest remembered point, anchored there.
                        // Phases are continuous across the points, evaluated backwards from the current phase.
                        double P = v.phi + v.theta * (double)r_last;  // phase at the newest point
                        for (int q = nr; q >= 1; q--) {
                            const long long Sq = v.recs[(size_t)q - 1].pos - v.seen;
                            rj.start[q] = (int)std::max<long long>(Sq, -2000000000ll);
                            rj.theta[q] = (q == nr) ? v.theta : v.recs[(size_t)q].theta_before;
                            rj.phi[q] = P - std::floor(P);
                            if (q >= 2) {  // phase at the start of the segment in front: back along ITS increment
                                const long long Sp = v.recs[(size_t)q - 2].pos - v.seen;
                                P = P + v.recs[(size_t)q - 1].theta_before * (double)(Sp - Sq);
                            }
                        }
                        rj.start[0] = rj.start[1];
                        rj.theta[0] = v.recs[0].theta_before;
                        rj.phi[0] = rj.phi[1];
                        retune.push_back(rj);
                    }
                }
            }
            v.soff[0] = v.soff[0] + nout * D - n_in;
            cur->n = nout;
            if (mem.fused) {
                Stream* nxt = &v.st[(size_t)v.i_first + 1];
                v.soff[1] = v.soff[1] + mem.nout2 * v.d.stage_decim[1] - nout;
                cur->n = 0;  // the stage-1 stream is never materialised
                nxt->n = mem.nout2;
                nxt->wlevel = lvl;
                cur = nxt;
            }
        }
        // the FM back end as one pipelined launch: last decimator, resampler, channel filter, discriminator + audio low-pass all in
        // their matrix form, and the pipeline's LDS layout fits
        const int last_dec = v.d.n_stages - 1;
        bool piped_be = c->pipe_on && (!ticking || c->tick_pipe) && (v.d.demod == SDRPP_DEMOD_WFM || v.d.demod == SDRPP_DEMOD_NFM) && last_dec >= first_sep && v.tp_stage[last_dec].ok &&
                        v.i_poly >= 0 && v.tp_poly.ok && v.i_chan >= 0 && v.chan_ntaps > 0 && v.tp_chan.ok && v.tp_audio.ok;
        // Pipelined mode: the four stages of a piped back end run in ONE tick (level L), so a stage cannot take the history of its input stream
        // from the side buffer — the carry that fills it from the previous block's tail runs one level behind that block's pipeline role, i.e. in
        // the very tick this block's role runs in.  It reads the tail where the previous block's role left it instead: the end of the previous
        // block's data buffer (the buffers are a ring of kRing, that one is not written again for three more ticks).  Needs the previous block to
        // have produced at least a history's worth of samples; a block after a shorter one runs as an ordinary pass (which waits for everything
        // queued and finds the side buffers complete).  With nothing queued the side buffers ARE complete and are used as they stand.
        auto pipe_in = [&](Stream& sx) -> StreamIn {
            StreamIn in = stream_in(sx);
            if (ticking && sx.hist_len > 0 && !c->tickq.empty()) {
                if (sx.prev_data && sx.prev_n >= sx.hist_len) { in.hist = sx.prev_data + (size_t)(sx.prev_n - sx.hist_len) * (size_t)sx.width; }
                else { c->tick_abort = true; }
            }
            return in;
        };
        const bool tick_piped = ticking;  // (all four stages at the level of the decimator)
        PipeJob pj{};
        size_t pj_lds = 0;
        if (piped_be) {
            pj.st[0] = toep_job(v.tp_stage[last_dec], 0, StreamIn{}, nullptr, 0, 0, 0.0f);
            pj.st[1] = toep_job(v.tp_poly, 0, StreamIn{}, nullptr, 0, 0, 0.0f);
            pj.st[2] = toep_job(v.tp_chan, 0, StreamIn{}, nullptr, 0, 0, 0.0f);
            pj.st[3] = toep_job(v.tp_audio, 0, StreamIn{}, nullptr, 0, 0, 0.0f);
            pj.timeouts = c->hd_tick_flag ? (int*)(c->hd_tick_flag + 8) : nullptr;
            piped_be = pipe_layout(pj, &pj_lds);
        }
        for (int s = first_sep; s < v.d.n_stages; s++) {
            Stream* nxt = &v.st[(size_t)v.i_first + s];
            const int Ds = v.d.stage_decim[s];
            const int no = decim_nout(cur->n, v.soff[s], Ds);
            if (need_bnd) { bounds_decim(bnd, v.soff[s], Ds); }
            if (split) { bounds_decim(tk, v.soff[s], Ds); }
            lvl++;
            cur->clevel = lvl;
            if (piped_be && s == last_dec) {
                pj.st[0] = toep_job(v.tp_stage[s], 0, stream_in(*cur), nxt->data, v.soff[s] - (v.d.stage_ntaps[s] - 1), no, 0.0f);
                pj.keep[0] = std::max(0, no - nxt->hist_len);
                pj.dec_stage = s;
                pj.lvl = lvl;
            }
            else if (v.tp_stage[s].ok) { t_dec.add(lvl, toep_job(v.tp_stage[s], 0, stream_in(*cur), nxt->data, v.soff[s] - (v.d.stage_ntaps[s] - 1), no, 0.0f)); }
            else { f_dec.add(lvl, FirBJob{ stream_in(*cur), nxt->data, v.d_staps[s], v.d.stage_ntaps[s], ilog2(Ds), v.soff[s], no, v.s_kp[s] }); }
            v.soff[s] = v.soff[s] + no * Ds - cur->n;
            nxt->n = no;
            nxt->wlevel = lvl;
            cur = nxt;
        }
        if (v.i_poly >= 0) {
            Stream* nxt = &v.st[(size_t)v.i_poly];
            const int no = poly_nout(cur->n, v.poff, v.pphase, v.d.interp, v.d.decim);
            if (need_bnd) { bounds_poly(bnd, v.poff, v.pphase, v.d.interp, v.d.decim); }
            if (split) { bounds_poly(tk, v.poff, v.pphase, v.d.interp, v.d.decim); }
            if (piped_be && tick_piped) { cur->clevel = lvl + 1; }  // (its tail is carried one level behind the role that writes it)
            else {
                lvl++;
                cur->clevel = lvl;
            }
            if (piped_be) {
                pj.st[1] = toep_job(v.tp_poly, v.pphase, tick_piped ? pipe_in(*cur) : stream_in(*cur), nxt->data, v.poff - (v.tpp - 1), no, 0.0f);
                pj.keep[1] = std::max(0, no - nxt->hist_len);
            }
            else if (v.tp_poly.ok) { t_poly.add(lvl, toep_job(v.tp_poly, v.pphase, stream_in(*cur), nxt->data, v.poff - (v.tpp - 1), no, 0.0f)); }
            else if (v.d_cyc) {
                polyb[(v.cyc_lmax == 4 ? 0 : 1) + ((v.d.decim & 1) ? 2 : 0)].add(lvl, PolyBJob{ stream_in(*cur), (float2*)nxt->data, v.d_cyc + (size_t)v.pphase * v.cyc_rows * v.cyc_lmax, v.d.interp, v.d.decim,
                                                                  v.tpp, v.poff, no, v.cyc_rows });
            }
            else {
                poly.add(lvl, PolyJob{ stream_in(*cur), (float2*)nxt->data, v.d_bank, v.d.interp, v.d.decim, v.tpp, v.pphase, v.poff, no });
            }
            const long long A = (long long)v.pphase + (long long)no * v.d.decim;
            v.pphase = (int)(A % v.d.interp);
            v.poff = v.poff + (int)(A / v.d.interp) - cur->n;
            nxt->n = no;
            nxt->wlevel = lvl;
            cur = nxt;
        }
        if (v.i_chan >= 0 && v.chan_ntaps > 0) {
            Stream* nxt = &v.st[(size_t)v.i_chan];
            if (piped_be && tick_piped) { cur->clevel = lvl + 1; }
            else {
                lvl++;
                cur->clevel = lvl;
            }
            if (piped_be) {
                pj.st[2] = toep_job(v.tp_chan, 0, tick_piped ? pipe_in(*cur) : stream_in(*cur), nxt->data, -(v.chan_ntaps - 1), cur->n, 0.0f);
                pj.keep[2] = 0;  // the IF stream is the RxVFO's output: all of it
            }
            else if (v.tp_chan.ok) { t_chan.add(lvl, toep_job(v.tp_chan, 0, stream_in(*cur), nxt->data, -(v.chan_ntaps - 1), cur->n, 0.0f)); }
            else { chan.add(lvl, FirBJob{ stream_in(*cur), nxt->data, v.d_chan, v.chan_ntaps, 0, 0, cur->n, v.chan_kp }); }
            nxt->n = cur->n;
            nxt->wlevel = lvl;
            cur = nxt;
        }
        v.i_if = (int)(cur - &v.st[0]);
        v.lvl_if = lvl;
        v.lvl_out = lvl;
        const int nif = cur->n;
        AgcState* agc = (AgcState*)v.d_state;
        float* dc = (float*)(v.d_state + 2 * sizeof(AgcState));
        const int* d_bnd = nullptr;
        if (need_bnd) {
            d_bnd = arena_push(c, bnd);
            if (!d_bnd) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
        }
        const int nbnd = need_bnd ? (int)bnd.size() : 0;
        if (v.d.demod == SDRPP_DEMOD_WFM || v.d.demod == SDRPP_DEMOD_NFM) {
            Stream& out = v.st[(size_t)v.i_out];
            if (piped_be && tick_piped) { cur->clevel = lvl + 1; }
            else {
                lvl++;
                cur->clevel = lvl;
            }
            if (piped_be) {
                pj.st[3] = toep_job(v.tp_audio, 0, tick_piped ? pipe_in(*cur) : stream_in(*cur), out.data, -(v.audio_ntaps - 1), nif, v.d.inv_deviation);
                pipes.push_back(pj);
                pipe_lds = std::max(pipe_lds, pj_lds);
            }
            else if (v.tp_audio.ok) { t_audio_fm.add(lvl, toep_job(v.tp_audio, 0, stream_in(*cur), out.data, -(v.audio_ntaps - 1), nif, v.d.inv_deviation)); }
            else { audio_fm.add(lvl, FirBJob{ stream_in(*cur), out.data, v.d_audio, v.audio_ntaps, 0, 0, nif, v.audio_kp, v.d.inv_deviation }); }
            out.n = nif;
            out.wlevel = lvl;
            v.lvl_out = lvl;
        }
        else if (v.d.demod == SDRPP_DEMOD_AM) {
            Stream& dem = v.st[(size_t)v.i_dem];
            Stream& out = v.st[(size_t)v.i_out];
            if (!v.d.am_carrier_agc) { pre.add(lvl + 1, PreJob{ 2, nif, (const float2*)cur->data, dem.data, 0.0, 0.0 }); }
            seq.add(lvl + 2, SeqJob{ 2, nif, (const float2*)cur->data, dem.data, nullptr, agc, agc + 1, dc, v.d.dc_block_rate, v.d.am_carrier_agc, d_bnd, nbnd });
            dem.n = nif;
            dem.wlevel = lvl + 2;
            lvl += 3;
            dem.clevel = lvl;
            if (v.tp_audio.ok) { t_audio.add(lvl, toep_job(v.tp_audio, 0, stream_in(dem), out.data, -(v.audio_ntaps - 1), nif, 0.0f)); }
            else { audio.add(lvl, FirBJob{ stream_in(dem), out.data, v.d_audio, v.audio_ntaps, 0, 0, nif, v.audio_kp }); }
            out.n = nif;
            out.wlevel = lvl;
            v.lvl_out = lvl;
        }
        else if (v.d.demod >= SDRPP_DEMOD_USB && v.d.demod <= SDRPP_DEMOD_DSB) {
            Stream& dem = v.st[(size_t)v.i_dem];
            Stream& out = v.st[(size_t)v.i_out];
            if (v.nco_exact) { ssbx_l.add(lvl + 1, SsbRotXJob{ (const float2*)cur->data, dem.data, v.d_rot + 1, v.d.ssb_phase_delta_re, v.d.ssb_phase_delta_im, d_bnd, nbnd }); }
            else if (split) {  // a launch group: the second translation anchored push by push, like the first (its phase advances as block-by-block processing advances it)
                int lo = 0;
                for (size_t j = 0; j < tk.size(); j++) {
                    const int nj = tk[j] - lo;
                    if (nj > 0) { pre.add(lvl + 1, PreJob{ v.d.demod, nj, (const float2*)cur->data + lo, dem.data + (size_t)lo * (size_t)dem.width, v.theta2, v.phi2 }); }
                    const double q2 = v.phi2 + (double)nj * v.theta2;
                    v.phi2 = q2 - std::floor(q2);
                    lo = tk[j];
                }
            }
            else { pre.add(lvl + 1, PreJob{ v.d.demod, nif, (const float2*)cur->data, dem.data, v.theta2, v.phi2 }); }
            seq.add(lvl + 2, SeqJob{ v.d.demod, nif, (const float2*)cur->data, dem.data, out.data, agc, agc + 1, dc, 0.0f, 0, d_bnd, nbnd });
            lvl += 2;
            dem.n = 0;  // scratch only
            out.n = nif;
            out.wlevel = lvl;
            v.lvl_out = lvl;
            if (!split || v.nco_exact) {
                const double p2 = v.phi2 + (double)nif * v.theta2;
                v.phi2 = p2 - std::floor(p2);
            }
        }
        if (v.af.on && v.i_out >= 0) {  // radio AF chain on the demodulator's stereo output
            Vfo::Af& a = v.af;
            Stream* acur = &v.st[(size_t)v.i_out];
            if (split) { v.tk_af = tk; }
            for (int s = 0; s < a.n_stages; s++) {
                Stream* nxt = &v.st[(size_t)a.i_stage0 + s];
                const int Ds = a.decim_s[s], K = (int)a.staps[s].size();
                const int no = decim_nout(acur->n, a.soff[s], Ds);
                if (split) { bounds_decim(v.tk_af, a.soff[s], Ds); }
                lvl++;
                acur->clevel = lvl;
                if (a.tp_stage[s].ok) { t_af_dec.add(lvl, toep_job(a.tp_stage[s], 0, stream_in(*acur), nxt->data, a.soff[s] - (K - 1), no, 0.0f)); }
                else { af_dec.add(lvl, FirBJob{ stream_in(*acur), nxt->data, a.d_staps[s], K, ilog2(Ds), a.soff[s], no, a.s_kp[s] }); }
                a.soff[s] = a.soff[s] + no * Ds - acur->n;
                nxt->n = no;
                nxt->wlevel = lvl;
                acur = nxt;
            }
            if (a.i_poly >= 0) {
                Stream* nxt = &v.st[(size_t)a.i_poly];
                const int no = poly_nout(acur->n, a.poff, a.pphase, a.interp, a.decim);
                if (split) { bounds_poly(v.tk_af, a.poff, a.pphase, a.interp, a.decim); }
                lvl++;
                acur->clevel = lvl;
                if (a.tp_poly.ok) { t_af_poly.add(lvl, toep_job(a.tp_poly, a.pphase, stream_in(*acur), nxt->data, a.poff - (a.tpp - 1), no, 0.0f)); }
                else { af_poly.add(lvl, PolyJob{ stream_in(*acur), (float2*)nxt->data, a.d_bank, a.interp, a.decim, a.tpp, a.pphase, a.poff, no }); }
                const long long A = (long long)a.pphase + (long long)no * a.decim;
                a.pphase = (int)(A % a.interp);
                a.poff = a.poff + (int)(A / a.interp) - acur->n;
                nxt->n = no;
                nxt->wlevel = lvl;
                acur = nxt;
            }
            if (a.i_hpf >= 0) {
                Stream* nxt = &v.st[(size_t)a.i_hpf];
                const int K = (int)a.htaps.size();
                lvl++;
                acur->clevel = lvl;
                if (a.tp_hpf.ok) { t_af_hpf.add(lvl, toep_job(a.tp_hpf, 0, stream_in(*acur), nxt->data, -(K - 1), acur->n, 0.0f)); }
                else { af_hpf.add(lvl, FirBJob{ stream_in(*acur), nxt->data, a.d_hpf, K, 0, 0, acur->n, a.hpf_kp }); }
                nxt->n = acur->n;
                nxt->wlevel = lvl;
                acur = nxt;
            }
            if (a.i_deemp >= 0) {
                Stream* nxt = &v.st[(size_t)a.i_deemp];
                lvl++;
                const int nseg = std::min(a.seg_cap, (acur->n + SDRPP_DEEMP_SEG - 1) / SDRPP_DEEMP_SEG);
                af_deemp.add(lvl, DeempJob{ (const float2*)acur->data, (float2*)nxt->data, acur->n, a.alpha, a.d_last + a.state_cur, a.d_last + (a.state_cur ^ 1),
                                             a.d_seg + (size_t)a.state_cur * ((size_t)a.seg_cap + 1), nseg, 0 });
                if (nseg > 0) { a.state_cur ^= 1; }  // (a block without audio leaves the state where it is)
                nxt->n = acur->n;
                lvl += 1;  // (the de-emphasis is two dependent launches: segment maps, then the outputs)
                nxt->wlevel = lvl;
                acur = nxt;
            }
            a.i_last = (int)(acur - &v.st[0]);
            v.lvl_af = lvl;
        }
        if (have_phi_end) { v.phi = phi_end; }
        else {
            const double p = v.phi + (double)n_in * v.theta;
            v.phi = p - std::floor(p);
        }
        v.seen += n_in;
        // history carries for every stream that has a consumer with memory
        const Stream* phantom = (v.fused_front && v.d.n_stages >= 2 && !v.nco_exact) ? &v.st[(size_t)v.i_first] : nullptr;  // stage-1 output of a fused front end: never written, never read
        for (auto& s : v.st) {
            if (s.hist_len > 0 && s.data && &s != phantom) {
                // pipelined: at the level of the consumer (its window of the NEXT block reads the new history one tick later, the carry of
                // the next block overwrites the old one one tick later still); a stream nobody reads with memory (a consumer may be attached
                // later: sdrpp_vfo_set_af, a taps change): one level behind the role that WRITES it — not behind the whole chain, which with
                // an AF chain is up to a dozen levels later, when the stream's ring buffer (kRing = 4) already holds a later block
                const int cl = !ticking ? carry_last : (s.clevel > 0 ? s.clevel : (s.wlevel > 0 ? s.wlevel + 1 : lvl + 1));
                carry.add(cl, CarryJob{ s.data, s.hist[s.cur], s.hist[s.cur ^ 1], s.hist_len, s.n, s.width, s.hist_len });
            }
        }
        c->plan_top = std::max(c->plan_top, lvl + 2);
        return SDRPP_OK;
    }

    // ---- stage 1 (optionally fused with stage 2): group VFOs with identical geometry, VT per job ----
    int group_front() {
        auto same = [](const S1Member& a, const S1Member& b) {
            return a.fused == b.fused && a.K == b.K && a.lgD == b.lgD && a.off0 == b.off0 && a.nout == b.nout && a.min_idx == b.min_idx && a.K2 == b.K2 &&
                   a.lgD2 == b.lgD2 && a.off2 == b.off2 && a.nout2 == b.nout2 && a.taph == b.taph && a.shift == b.shift && a.oshift == b.oshift;
        };
        std::sort(s1.begin(), s1.end(), [](const S1Member& a, const S1Member& b) {
            if (a.fused != b.fused) { return a.fused < b.fused; }
            if (a.K != b.K) { return a.K < b.K; }
            if (a.lgD != b.lgD) { return a.lgD < b.lgD; }
            if (a.off0 != b.off0) { return a.off0 < b.off0; }
            if (a.nout != b.nout) { return a.nout < b.nout; }
            if (a.min_idx != b.min_idx) { return a.min_idx < b.min_idx; }
            if (a.K2 != b.K2) { return a.K2 < b.K2; }
            if (a.lgD2 != b.lgD2) { return a.lgD2 < b.lgD2; }
            if (a.off2 != b.off2) { return a.off2 < b.off2; }
            if (a.nout2 != b.nout2) { return a.nout2 < b.nout2; }
            if (a.taph != b.taph) { return a.taph < b.taph; }
            if (a.shift != b.shift) { return a.shift < b.shift; }
            if (a.oshift != b.oshift) { return a.oshift < b.oshift; }
            return a.v->id < b.v->id;
        });
        if (ticking) {
            fcl_nw = 4;
            for (auto& m : s1) {
                if (!m.fused && m.lgD >= 5 && m.K >= 9 && (size_t)frontcl_lds_floats(m.K, m.lgD, 4) * 4 > (size_t)(160 * 1024 / 2)) { fcl_nw = 2; }
            }
        }
        size_t i = 0;
        while (i < s1.size()) {
            size_t j = i;
            while (j < s1.size() && same(s1[j], s1[i])) { j++; }
            size_t g = i;
            // ---- matrix-core path: >= 17 fused VFOs of one geometry -> jobs of up to 32 VFOs, stages 1 + 2 as one composite FIR ----
            int m_pf = 0;
            const bool m_fused = s1[i].fused && frontcm_ok(s1[i].K, s1[i].lgD, s1[i].K2, s1[i].lgD2, &m_pf);
            // ... or a long first stage on its own (decimation >= 32: no fusion, vfo_frontcl_kernel)
            const bool m_long = !m_fused && !s1[i].fused && s1[i].lgD >= 5 && s1[i].K >= 9 && (size_t)frontcl_lds_floats(s1[i].K, s1[i].lgD) * 4 <= (size_t)kMaxLds;
            const bool m_ok = m_fused || m_long;
            // worth it from 17 VFOs against the fused VALU kernel (8 VFOs per work-item); a long first stage has no good VALU form (its
            // per-VFO windows do not fit LDS), there the matrix kernel pays off from 2 VFOs on
            const size_t m_min = m_long ? 2 : 17;
            while (m_ok && j - g >= m_min) {
                const int vt = (int)std::min<size_t>(j - g, SDRPP_FCM_VT);
                S1Member h = s1[g];
                if (m_long) {  // the "composite" is the first stage alone
                    h.K2 = 1;
                    h.lgD2 = 0;
                    h.off2 = 0;
                    h.nout2 = h.nout;
                }
                const int D1 = 1 << h.lgD;
                const int K = h.K + (h.K2 - 1) * D1, lgD = h.lgD + h.lgD2;
                const int NP = (K + 1) / 2, NP4 = (NP + 31) / 32 * 32;  // rows of the tap operand table, zero padded (the kernels read whole rings: up to 32 rows — eight steps of four pairs — at a time)
                const std::string key = member_key(m_long ? 'L' : 'M', &s1[g], vt);
                float2* d_taps = nullptr;
                auto it = c->s1_tap_cache.find(key);
                if (it != c->s1_tap_cache.end()) { d_taps = it->second; }
                else {
                    // [NP4][64] floats (= NP4 * 32 float2; rows >= NP are zero padding) followed by [32][TILE] float2
                    std::vector<float2> host((size_t)NP4 * 32 + (size_t)SDRPP_FCM_VT * SDRPP_FCM_TILE, make_float2(0.0f, 0.0f));
                    float* at = reinterpret_cast<float*>(host.data());
                    std::vector<double> h12((size_t)K);
                    const double kc = 0.5 * (double)(K - 1);
                    for (int m = 0; m < vt; m++) {
                        const Vfo& vv = *s1[g + m].v;
                        // composite taps h12 = h1 (*) upsample(h2, D1) in double precision (both are linear phase, so is h12)
                        std::fill(h12.begin(), h12.end(), 0.0);
                        for (int k2 = 0; k2 < h.K2; k2++) {
                            const double w2 = m_long ? 1.0 : (double)vv.staps[1][(size_t)k2];
                            for (int k1 = 0; k1 < h.K; k1++) { h12[(size_t)k2 * D1 + k1] += w2 * (double)vv.staps[0][(size_t)k1]; }
                        }
                        for (int pz = 0; pz < NP; pz++) {
                            double t = ((double)pz - kc) * vv.theta;  // modulation centred on the filter: g[K-1-k] = conj(g[k])
                            t -= std::rint(t);
                            const double a = 2.0 * 3.14159265358979323846 * t;
                            double gr = h12[(size_t)pz] * std::cos(a), gi = h12[(size_t)pz] * std::sin(a);
                            if ((K & 1) && pz == NP - 1) { gr = h12[(size_t)pz]; gi = 0.0; }
                            at[(size_t)pz * 64 + m] = (float)gr;
                            at[(size_t)pz * 64 + 32 + m] = (float)-gi;
                        }
                    }
                    for (int m = 0; m < SDRPP_FCM_VT; m++) {
                        const double step = m < vt ? s1[g + m].v->theta * (double)(1 << lgD) : 0.0;
                        for (int jj = 0; jj < SDRPP_FCM_TILE; jj++) {
                            double tt = step * (double)jj;
                            tt -= std::rint(tt);
                            const double a = 2.0 * 3.14159265358979323846 * tt;
                            host[(size_t)NP4 * 32 + (size_t)m * SDRPP_FCM_TILE + jj] = make_float2((float)std::cos(a), (float)std::sin(a));
                        }
                    }
                    if (c->s1_tap_cache.size() > 4096) {
                        HIPCHK(c, hipStreamSynchronize(c->stream));
                        for (auto& e : c->s1_tap_cache) { (void)hipFree(e.second); }
                        c->s1_tap_cache.clear();
                    }
                    int rc = dev_alloc(c, &d_taps, host.size());
                    if (rc) { return rc; }
                    HIPCHK(c, hipMemcpyAsync(d_taps, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    c->s1_tap_cache[key] = d_taps;
                }
                FrontCMJob job{};
                job.nv = vt;
                job.ntaps = K;
                job.log2_decim = lgD;
                job.off = h.off0 + (h.off2 - (h.K2 - 1)) * D1 - (h.K - 1) + h.shift;
                job.nout = h.nout2;
                job.min_idx = h.min_idx;
                job.anchor = h.shift;
                // (a long first stage with at most 16 VFOs runs in the 16 x 16 x 4 shape: 16 outputs per tile, vfo_frontcl_impl<PF, true>)
                const int tile_n = (m_long && vt <= 16) ? 16 : SDRPP_FCM_TILE;
                const int ntiles = (h.nout2 + tile_n - 1) / tile_n;
                // one resident round: 256 CUs x 3 blocks x 4 wavefronts (a second, partly filled round would cost as much as the first);
                // the long-stage kernel runs 2 wavefronts per block, its LDS footprint decides how many blocks fit
                const int long_blocks = m_long ? std::max(1, (int)((size_t)(160 * 1024) / ((size_t)frontcl_lds_floats(K, lgD, fcl_nw) * 4))) : 0;
                // (a launch group brings one job per push: together they get the wavefronts one job of the whole block would)
                const int nsub = std::max<int>(1, (int)c->grp_ends.size());
                const int resident = std::max(64, (m_long ? 256 * long_blocks * fcl_nw : (c->tick_planning ? std::min(3072, c->tick_fcm_waves * c->plan_sparse) : 3072)) / nsub);
                job.tiles_per_wave = std::max(1, (ntiles + resident - 1) / resident);
                job.atab = reinterpret_cast<const float*>(d_taps);
                job.ptab = d_taps + (size_t)NP4 * 32;
                for (int m = 0; m < SDRPP_FCM_VT; m++) {
                    Vfo* v = s1[g + std::min(m, vt - 1)].v;
                    job.theta[m] = v->theta;
                    job.phi0[m] = s1[g + std::min(m, vt - 1)].phi0;
                    job.out[m] = (float2*)v->st[(size_t)v->i_first + (m_long ? 0 : 1)].data + h.oshift;
                }
                if (m_long) {
                    fcl.jobs.push_back(job);
                    fcl.max_blocks = std::max(fcl.max_blocks, (ntiles + fcl_nw * job.tiles_per_wave - 1) / (fcl_nw * job.tiles_per_wave));
                    fcl.lds = std::max(fcl.lds, (size_t)frontcl_lds_floats(K, lgD, fcl_nw) * 4);
                }
                else {
                    FCMLaunch& L = fcm[m_pf == 6 ? 0 : (m_pf == 10 ? 1 : 2)];
                    if (ticking && c->tick_fcm16w && K == 132 && lgD == 4 && (L.jobs.empty() || L.w16)) {
                        // the 16 x 16 x 4 shape walking its tiles: the field counts tiles per WORKGROUP there (vfo_frontcm16w_body)
                        job.tiles_per_wave = std::max(1, (ntiles + c->tick_fcm16w_blocks - 1) / c->tick_fcm16w_blocks);
                        L.w16 = true;
                        L.w16_blocks = std::max(L.w16_blocks, (ntiles + job.tiles_per_wave - 1) / job.tiles_per_wave);
                    }
                    L.jobs.push_back(job);
                    L.max_blocks = std::max(L.max_blocks, (ntiles + 4 * job.tiles_per_wave - 1) / (4 * job.tiles_per_wave));
                    L.lds = std::max(L.lds, (size_t)frontcm_layout(K, lgD).total * 4);
                }
                g += (size_t)vt;
            }
            while (g < j) {
                const size_t left = j - g;
                int li = left >= 8 ? 0 : (left >= 4 ? 1 : (left >= 2 ? 2 : 3));
                if (ticking) { li = 3; }  // (pipelined: one VFO per job — the forms that are roles of the tick kernel, TR_S1_1 / TR_S1D_1 / TR_F2_1; same sums per VFO)
                const int vt = vts[li];
                // tap array for this membership (cached on the device)
                const std::string key = member_key('V', &s1[g], vt);
                float2* d_taps = nullptr;
                auto it = c->s1_tap_cache.find(key);
                if (it != c->s1_tap_cache.end()) { d_taps = it->second; }
                else {
                    const int K = (s1[g].K + 1) / 2;  // tap pairs
                    std::vector<float2> host((size_t)K * vt + (size_t)256 * vt);
                    for (int k = 0; k < K; k++) {
                        for (int m = 0; m < vt; m++) { host[(size_t)k * vt + m] = s1[g + m].v->modtaps[(size_t)k]; }
                    }
                    // NCO advance inside a 256-output tile: exp(j*2*pi*theta*D1*j) (fused front kernel)
                    for (int m = 0; m < vt; m++) {
                        const double step = s1[g + m].v->theta * (double)(1 << s1[g].lgD);
                        for (int jj = 0; jj < 256; jj++) {
                            double tt = step * (double)jj;
                            tt -= std::rint(tt);
                            const double a = 2.0 * 3.14159265358979323846 * tt;
                            host[(size_t)K * vt + (size_t)jj * vt + m] = make_float2((float)std::cos(a), (float)std::sin(a));
                        }
                    }
                    if (c->s1_tap_cache.size() > 4096) {  // retune churn: drop everything (rare)
                        HIPCHK(c, hipStreamSynchronize(c->stream));
                        for (auto& e : c->s1_tap_cache) { (void)hipFree(e.second); }
                        c->s1_tap_cache.clear();
                    }
                    int rc = dev_alloc(c, &d_taps, host.size());
                    if (rc) { return rc; }
                    HIPCHK(c, hipMemcpyAsync(d_taps, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));  // `host` is pageable and goes out of scope
                    c->s1_tap_cache[key] = d_taps;
                }
                const S1Member& h = s1[g];
                if (h.fused) {
                    Front2Job job{};
                    job.nv = vt;
                    job.ntaps1 = h.K;
                    job.log2_decim1 = h.lgD;
                    job.off1 = h.off0 + h.shift;
                    job.ntaps2 = h.K2;
                    job.log2_decim2 = h.lgD2;
                    job.off2 = h.off2;
                    job.nout2 = h.nout2;
                    job.t2 = front2_t2(h.K, 1 << h.lgD, h.K2, 1 << h.lgD2, 8);
                    job.min_idx = h.min_idx;
                    job.anchor = h.shift;
                    job.ctaps = d_taps;
                    job.ptab = d_taps + (size_t)((h.K + 1) / 2) * vt;
                    job.taps2 = h.v->d_staps_nat[1];
                    for (int m = 0; m < vt; m++) {
                        Vfo* v = s1[g + m].v;
                        job.theta[m] = v->theta;
                        job.phi0[m] = s1[g + m].phi0;
                        job.out[m] = (float2*)v->st[(size_t)v->i_first + 1].data + h.oshift;
                    }
                    f2l[li].jobs.push_back(job);
                    f2l[li].max_blocks = std::max(f2l[li].max_blocks, (job.nout2 + job.t2 - 1) / job.t2);
                    const int D1 = 1 << h.lgD;
                    f2l[li].lds = std::max(f2l[li].lds, (std::max((size_t)D1 * (256 + (h.K - 1 + D1 - 1) / D1 + 1), (size_t)vt * 272) + (size_t)vt) * sizeof(float2));
                }
                else {
                    Stage1Job job{};
                    job.nv = vt;
                    job.ntaps = h.K;
                    job.log2_decim = h.lgD;
                    job.off0 = h.off0 + h.shift;
                    job.nout = h.nout;
                    job.min_idx = h.min_idx;
                    job.anchor = h.shift;
                    job.ctaps = d_taps;
                    for (int m = 0; m < vt; m++) {
                        Vfo* v = s1[g + m].v;
                        job.theta[m] = v->theta;
                        job.phi0[m] = s1[g + m].phi0;  // phase (turns) of push-relative sample 0
                        job.out[m] = (float2*)v->st[(size_t)v->i_first].data + h.oshift;
                    }
                    s1l[li].jobs.push_back(job);
                    s1l[li].max_nout = std::max(s1l[li].max_nout, job.nout);
                    const int D = 1 << job.log2_decim;
                    const int tile = pick_tile(D, job.ntaps, 8);
                    if (tile == 0 && job.log2_decim < 5) { return fail(c, SDRPP_ERR_UNSUPPORTED, "stage-1 filter (decim %d, %d taps) does not fit in LDS", D, job.ntaps); }
                    if (tile > 0) { s1l[li].tile = std::min(s1l[li].tile, tile); }
                }
                g += (size_t)vt;
            }
            i = j;
        }
        return SDRPP_OK;
    }

    // ---- job tables into the arena (one upload for the whole block) ----
    int upload() {
        for (int k = 0; k < 4; k++) {
            if (!s1l[k].jobs.empty()) {
                for (auto& jb : s1l[k].jobs) { s1l[k].lds = std::max(s1l[k].lds, fir_lds(s1l[k].tile, 1 << jb.log2_decim, jb.ntaps, 8)); }
                d_s1[k] = arena_push(c, s1l[k].jobs);
                if (!d_s1[k]) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
            }
        }
        for (int k = 0; k < 4; k++) {
            if (!f2l[k].jobs.empty()) {
                d_f2[k] = arena_push(c, f2l[k].jobs);
                if (!d_f2[k]) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
            }
        }
        if (!fcl.jobs.empty()) {
            // Tiles per wavefront of the long first stages, chosen over ALL their jobs (round 5, profiles/r05h_fcl_tiles_per_wave*.log): a tile is
            // mostly start-up (window fetch, tile phasor, epilogue) around a matrix loop of 3 us (16 rows) .. 12 us (32 rows), and a wavefront that
            // walks several tiles fetches the next window under the current loop — but the walks must still fill the device: the best setting
            // put ~0.7 of the resident wavefront slots to work (cfg 4 at 10^6-sample blocks: 2 tiles per 32-row, 4 per 16-row wavefront, 8.29 ->
            // 8.89 GS/s; at 307 200: 1 and 2, 5.11 -> 5.26; twice that was 15 % slower at either size).  SDRPP_GPU_FCL_TPW / _TPW16 override.
            static const int tpw32_env = getenv("SDRPP_GPU_FCL_TPW") ? atoi(getenv("SDRPP_GPU_FCL_TPW")) : 0;
            static const int tpw16_env = getenv("SDRPP_GPU_FCL_TPW16") ? atoi(getenv("SDRPP_GPU_FCL_TPW16")) : 0;
            size_t lds_max = 0;
            for (auto& jb : fcl.jobs) { lds_max = std::max(lds_max, (size_t)frontcl_lds_floats(jb.ntaps, jb.log2_decim, fcl_nw) * 4); }
            const int long_blocks = std::max(1, (int)((size_t)(160 * 1024) / std::max<size_t>(lds_max, 1)));
            const double target = 0.7 * 256.0 * (double)long_blocks * (double)fcl_nw;
            fcl.max_blocks = 0;
            // Round 6: the jobs of a tick differ in what a tile costs — 257 .. 400 taps (NFM / USB / AM at 61.44 MS/s), 32-row or 16-row tiles — and a walk of
            // the SAME number of tiles took 1.5x as long in one job as in another: the tick ended on the 400-tap jobs' workgroups (last end 337 us against a
            // mean life of 220, profiles/r06q_tick_timeline_cfg4_B1000000_group4.txt).  Walk lengths now follow a cost model, tile ~ c0 + c1 * taps (16-row
            // tiles: 0.45 of the matrix part), so that every wavefront of the role is busy about equally long; the total number of wavefronts still follows
            // the 0.7-of-the-resident-slots rule.  SDRPP_GPU_FCL_BALANCE=0: the old rule (measurements).
            static const bool balance = getenv("SDRPP_GPU_FCL_BALANCE") ? atoi(getenv("SDRPP_GPU_FCL_BALANCE")) != 0 : true;
            static const double c0 = getenv("SDRPP_GPU_FCL_C0") ? atof(getenv("SDRPP_GPU_FCL_C0")) : 1.5, c1 = getenv("SDRPP_GPU_FCL_C1") ? atof(getenv("SDRPP_GPU_FCL_C1")) : 0.031;
            auto tile_cost = [&](const FrontCMJob& jb) { return c0 + c1 * (double)jb.ntaps * (jb.nv <= 16 ? 0.45 : 1.0); };
            double total_cost = 0.0;
            for (auto& jb : fcl.jobs) {
                const int tile_n = jb.nv <= 16 ? 16 : SDRPP_FCM_TILE;
                total_cost += tile_cost(jb) * (double)((jb.nout + tile_n - 1) / tile_n);
            }
            const double per_wave = total_cost / target;  // what one wavefront should carry
            for (auto& jb : fcl.jobs) {
                const int tile_n = jb.nv <= 16 ? 16 : SDRPP_FCM_TILE;
                const int ntiles = (jb.nout + tile_n - 1) / tile_n;
                const int env = tile_n == 16 ? tpw16_env : tpw32_env;
                // (ticks: the rule above.  An ordinary pass has the device to itself and its pushes are long: one resident round per JOB as before —
                // the 0.7 rule gave walks of 34 tiles at 2^24-sample pushes and lost 17 % there, profiles/r05z_bench_default.json vs r05d)
                const int resident = 256 * long_blocks * fcl_nw;
                int tpw = env > 0 ? env : (ticking ? std::max(1, (int)((double)ntiles * (double)fcl.jobs.size() / target + 0.75)) : std::max(1, (ntiles + resident - 1) / resident));
                if (env <= 0 && ticking && balance && fcl.jobs.size() > 1) { tpw = std::max(1, (int)(per_wave / tile_cost(jb) + 0.5)); }
                jb.tiles_per_wave = tpw;
                fcl.max_blocks = std::max(fcl.max_blocks, (ntiles + fcl_nw * tpw - 1) / (fcl_nw * tpw));
            }
        }
        d_fcl = arena_push(c, fcl.jobs);
        if (!fcl.jobs.empty() && !d_fcl) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
        for (int k = 0; k < 3; k++) {
            if (!fcm[k].jobs.empty()) {
                d_fcm[k] = arena_push(c, fcm[k].jobs);
                if (!d_fcm[k]) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
            }
        }
        d_rotx = arena_push(c, rotx);
        d_retune = arena_push(c, retune);
        d_rot = arena_push(c, rot);
        d_fb = (!rotx.empty()) ? arena_push(c, fb) : nullptr;
        d_rotx_head = nullptr;
        if (!rotx.empty() && d_rotx && d_fb && c->tick_planning) {
            std::vector<RotXHead> head{ RotXHead{ d_rotx, d_fb, (int)rotx.size(), (int)fb.size(), c->rot_exact_vpw, 0 } };
            d_rotx_head = arena_push(c, head);
            if (!d_rotx_head) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
        }
        if ((!rotx.empty() && (!d_rotx || !d_fb)) || (!retune.empty() && !d_retune) || (!rot.empty() && !d_rot)) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
        // Pipelined back ends.  An ordinary pass: ONE launch, at the latest level any of its jobs starts at (levels only order the launches of a
        // pass) — or none, when this push is better served by the separate launches.  Pipelined mode: one role per LEVEL (a role that ran later
        // than its VFO's level would find the history of its first stage's input overwritten by the next block's carry), always the pipeline.
        if (!pipes.empty() && !ticking && pipe_segments(pipes, c->pipe_on, pipe_lds) == 0) {  // not this push: the same four jobs go to the separate launches
            for (auto& pj : pipes) {
                t_dec.add(pj.lvl, pj.st[0]);
                t_poly.add(pj.lvl + 1, pj.st[1]);
                t_chan.add(pj.lvl + 2, pj.st[2]);
                t_audio_fm.add(pj.lvl + 3, pj.st[3]);
            }
            pipes.clear();
        }
        for (auto& pj : pipes) {
            PipeGroup* g = nullptr;
            for (auto& q : pgroups) {
                if (!ticking || q.lvl == pj.lvl) { g = &q; }
            }
            if (!g) {
                pgroups.emplace_back();
                g = &pgroups.back();
                g->lvl = pj.lvl;
            }
            g->lvl = std::max(g->lvl, pj.lvl);
            g->jobs.push_back(pj);
        }
        for (auto& g : pgroups) {
            g.seg = pipe_segments(g.jobs, c->pipe_on, pipe_lds, ticking ? c->tick_pipe_blocks : 0);
            g.dev = arena_push(c, g.jobs);
            if (!g.dev) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
            pipe_top = std::max(pipe_top, g.lvl + 1);
        }
        for (int i = 0; i < kToepLists; i++) {
            Lev<ToepJob>& L = *tlists[i].L;
            for (int l = 0; l < L.top; l++) {
                if (L.at[l].empty()) { continue; }
                tplan[i][l] = toep_plan(L.at[l], tlists[i].npl, c->tick_planning ? std::min(2048, c->tick_toep_blocks * c->plan_sparse) : 2048);
                if (tplan[i][l].lds > (size_t)kMaxLds) { return fail(c, SDRPP_ERR_UNSUPPORTED, "matrix-core FIR window does not fit in LDS"); }
            }
            if (!arena_push_lev(c, L)) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
        }
        if (!arena_push_lev(c, f_dec) || !arena_push_lev(c, poly) || !arena_push_lev(c, polyb[0]) || !arena_push_lev(c, polyb[1]) || !arena_push_lev(c, polyb[2]) ||
            !arena_push_lev(c, polyb[3]) || !arena_push_lev(c, chan) || !arena_push_lev(c, seq) || !arena_push_lev(c, pre) || !arena_push_lev(c, audio) ||
            !arena_push_lev(c, audio_fm) || !arena_push_lev(c, af_dec) || !arena_push_lev(c, af_hpf) || !arena_push_lev(c, af_poly) || !arena_push_lev(c, af_deemp) ||
            !arena_push_lev(c, ssbx_l) || !arena_push_lev(c, carry)) {
            return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted");
        }
        int rc;
        {
            HostScope hs("arena_commit (H2D)");
            rc = arena_commit(c);
        }
        if (rc) { return rc; }
        return SDRPP_OK;
    }

    // ---- level 1: the front end ----
    int emit_front() {
        {
            FamilyTimer t(c, F_S1);
            if (!rotx.empty() && n_in > 0) {
                if (c->rot_exact_single) { launch(c, vfo_rotate_exact_kernel, dim3(((unsigned)rotx.size() + 63) / 64), dim3(64), (size_t)64 * 65 * sizeof(float2), src, (const RotXJob*)d_rotx, (int)rotx.size(), d_fb, (int)fb.size()); }
                else if (c->tick_planning && c->rot_exact_skip >= 16) {  // pipelined: the chain as a role of the tick (level 1: its VFOs' first stages follow at level 2)
                    emit(c, L0 + 1, F_S1, TR_ROTX16, ((int)rotx.size() + c->rot_exact_vpw - 1) / c->rot_exact_vpw, 1, SDRPP_ROTX4_LDS_BYTES, d_rotx_head, &src);
                }
                else {
                    const int vpw = c->rot_exact_vpw;
                    const dim3 grid(((unsigned)rotx.size() + vpw - 1) / vpw);
                    if (c->rot_exact_skip >= 16) { launch(c, vfo_rotate_exact4_kernel<16>, grid, dim3(256), SDRPP_ROTX4_LDS_BYTES, src, (const RotXJob*)d_rotx, (int)rotx.size(), d_fb, (int)fb.size(), vpw); }
                    else if (c->rot_exact_skip >= 8) { launch(c, vfo_rotate_exact4_kernel<8>, grid, dim3(256), SDRPP_ROTX4_LDS_BYTES, src, (const RotXJob*)d_rotx, (int)rotx.size(), d_fb, (int)fb.size(), vpw); }
                    else { launch(c, vfo_rotate_exact4_kernel<4>, grid, dim3(256), SDRPP_ROTX4_LDS_BYTES, src, (const RotXJob*)d_rotx, (int)rotx.size(), d_fb, (int)fb.size(), vpw); }
                }
            }
            for (int k = 0; k < 4; k++) {
                if (s1l[k].jobs.empty() || s1l[k].max_nout == 0) { continue; }
                bool direct = true;  // every job of the class decimates by >= 32: stream from global memory, no LDS tile
                for (auto& jb : s1l[k].jobs) { direct = direct && jb.log2_decim >= 5; }
                if (direct) {
                    if (c->tick_planning && s1l[k].vt == 1) {
                        emit(c, L0 + 1, F_S1, TR_S1D_1, (s1l[k].max_nout + 255) / 256, (int)s1l[k].jobs.size(), 0, d_s1[k], &src);
                        continue;
                    }
                    const dim3 grid((s1l[k].max_nout + 255) / 256, (unsigned)s1l[k].jobs.size());
                    switch (s1l[k].vt) {
                    case 8: launch(c, vfo_stage1_direct_kernel<8>, grid, dim3(256), 0, src, (const Stage1Job*)d_s1[k]); break;
                    case 4: launch(c, vfo_stage1_direct_kernel<4>, grid, dim3(256), 0, src, (const Stage1Job*)d_s1[k]); break;
                    case 2: launch(c, vfo_stage1_direct_kernel<2>, grid, dim3(256), 0, src, (const Stage1Job*)d_s1[k]); break;
                    default: launch(c, vfo_stage1_direct_kernel<1>, grid, dim3(256), 0, src, (const Stage1Job*)d_s1[k]); break;
                    }
                    continue;
                }
                if (c->tick_planning && s1l[k].vt == 1) {
                    emit(c, L0 + 1, F_S1, TR_S1_1, (s1l[k].max_nout + s1l[k].tile - 1) / s1l[k].tile, (int)s1l[k].jobs.size(), s1l[k].lds, d_s1[k], &src, s1l[k].tile);
                    continue;
                }
                const dim3 grid((s1l[k].max_nout + s1l[k].tile - 1) / s1l[k].tile, (unsigned)s1l[k].jobs.size());
                const dim3 block(s1l[k].tile);
                switch (s1l[k].vt) {
                case 8: launch(c, vfo_stage1_kernel<8>, grid, block, s1l[k].lds, src, (const Stage1Job*)d_s1[k]); break;
                case 4: launch(c, vfo_stage1_kernel<4>, grid, block, s1l[k].lds, src, (const Stage1Job*)d_s1[k]); break;
                case 2: launch(c, vfo_stage1_kernel<2>, grid, block, s1l[k].lds, src, (const Stage1Job*)d_s1[k]); break;
                default: launch(c, vfo_stage1_kernel<1>, grid, block, s1l[k].lds, src, (const Stage1Job*)d_s1[k]); break;
                }
            }
            for (int k = 0; k < 4; k++) {
                if (f2l[k].jobs.empty() || f2l[k].max_blocks == 0) { continue; }
                if (c->tick_planning && f2l[k].vt == 1) {
                    emit(c, L0 + 1, F_S1, TR_F2_1, f2l[k].max_blocks, (int)f2l[k].jobs.size(), f2l[k].lds, d_f2[k], &src);
                    continue;
                }
                const dim3 grid((unsigned)f2l[k].max_blocks, (unsigned)f2l[k].jobs.size());
                const dim3 block(256);
                // all jobs of a launch class share VT; the (44 taps, /8) first stage of the ratio-32 plan (10 MS/s -> 312.5 kS/s) has a
                // fully unrolled instance, everything else runs the generic loop
                bool all_44_3 = true;
                for (auto& jb : f2l[k].jobs) { all_44_3 = all_44_3 && jb.ntaps1 == 44 && jb.log2_decim1 == 3; }
                switch (f2l[k].vt) {
                case 8:
                    if (all_44_3) { launch(c, vfo_front2_kernel<8, 44, 3>, grid, block, f2l[k].lds, src, (const Front2Job*)d_f2[k]); }
                    else { launch(c, vfo_front2_kernel<8, 0, 0>, grid, block, f2l[k].lds, src, (const Front2Job*)d_f2[k]); }
                    break;
                case 4: launch(c, vfo_front2_kernel<4, 0, 0>, grid, block, f2l[k].lds, src, (const Front2Job*)d_f2[k]); break;
                case 2: launch(c, vfo_front2_kernel<2, 0, 0>, grid, block, f2l[k].lds, src, (const Front2Job*)d_f2[k]); break;
                default: launch(c, vfo_front2_kernel<1, 0, 0>, grid, block, f2l[k].lds, src, (const Front2Job*)d_f2[k]); break;
                }
            }
            for (int k = 0; k < 3; k++) {
                if (fcm[k].jobs.empty() || fcm[k].max_blocks == 0) { continue; }
                bool all_132_4 = true;  // ratio-32 plan: fir_32_8 (44 taps, /8) + fir_4_2 (12 taps, /2) -> 132 composite taps, /16
                for (auto& jb : fcm[k].jobs) { all_132_4 = all_132_4 && jb.ntaps == 132 && jb.log2_decim == 4; }
                const int role = (k == 1 && all_132_4) ? TR_FCM_132_4 : (k == 0 ? TR_FCM_6 : (k == 1 ? TR_FCM_10 : TR_FCM_16));
                // small blocks: every wavefront of the 32 x 32 x 2 form would have ONE tile and spend 4 us in its matrix loop alone — a workgroup
                // per tile in the 16 x 16 x 4 shape instead (same sums in the same order: bit-identical), up to fcm16_max_tiles tiles per job
                int max_tiles = 0;
                bool one_tile = true;
                for (auto& jb : fcm[k].jobs) {
                    max_tiles = std::max(max_tiles, (jb.nout + SDRPP_FCM_TILE - 1) / SDRPP_FCM_TILE);
                    one_tile = one_tile && jb.tiles_per_wave == 1;
                }
                if (fcm[k].w16 && role == TR_FCM_132_4 && !(one_tile && max_tiles <= 256 && !c->plan_block_from_host)) {
                    emit(c, L0 + 1, F_S1, TR_FCM16W_132_4, fcm[k].w16_blocks, (int)fcm[k].jobs.size(), (size_t)frontcm16w_layout(132, 4).total * 4, d_fcm[k], &src);
                    continue;
                }
                const int small_limit = c->fcm16_max_tiles >= 0 ? c->fcm16_max_tiles : ((c->tick_planning && c->plan_block_from_host) ? 0 : 256);
                if (role == TR_FCM_132_4 && one_tile && max_tiles > 0 && max_tiles <= small_limit) {
                    if (getenv("SDRPP_TICK_DEBUG")) { fprintf(stderr, "[sdrpp] front end in its small-block shape: %d tiles x %zu jobs\n", max_tiles, fcm[k].jobs.size()); }
                    emit(c, L0 + 1, F_S1, TR_FCM16_132_4, max_tiles, (int)fcm[k].jobs.size(), (size_t)frontcm16_layout(132, 4).total * 4, d_fcm[k], &src);
                    continue;
                }
                emit(c, L0 + 1, F_S1, role, fcm[k].max_blocks, (int)fcm[k].jobs.size(), fcm[k].lds, d_fcm[k], &src);
            }
            if (!fcl.jobs.empty() && fcl.max_blocks > 0) {
                bool pf_ok = true;  // every window of the launch fits the register prefetch
                for (auto& jb : fcl.jobs) { pf_ok = pf_ok && (SDRPP_FCM_TILE - 1) * (1 << jb.log2_decim) + jb.ntaps <= 64 * SDRPP_FCL_PF; }
                emit(c, L0 + 1, F_S1, pf_ok ? TR_FCL_PF : TR_FCL_0, fcl.max_blocks, (int)fcl.jobs.size(), fcl.lds, d_fcl, &src, fcl_nw);
            }
            if (!rot.empty() && max_rot > 0) { emit(c, L0 + 1, F_S1, TR_ROT, std::min((max_rot + 255) / 256, 4096), (int)rot.size(), 0, d_rot, &src); }
            if (!retune.empty()) {
                int mx = 0;
                for (auto& r : retune) { mx = std::max(mx, r.nfix); }
                launch(c, vfo_retune_fix_kernel, dim3((unsigned)mx, (unsigned)retune.size()), dim3(64), 0, src, (const RetuneJob*)d_retune);
            }
        }
        return SDRPP_OK;
    }

    int launch_fir(int level, int fam, std::vector<FirBJob>& jobs, FirBJob* d_jobs, int width, bool stereo, bool quad = false) {
            if (jobs.empty()) { return SDRPP_OK; }
            const int R = SDRPP_FIR_R;
            int max_nout = 0, threads = 256;
            auto lds_for = [&](const FirBJob& jb, int nt) {
                size_t b = (size_t)(1 << jb.log2_decim) * R * (size_t)(nt + jb.kp_pad / R + 1) * width * 4;
                if (quad) { b += ((size_t)nt * R + jb.ntaps + 2) * 4; }  // phase scratch of the fused discriminator
                return b;
            };
            for (auto& jb : jobs) {
                max_nout = std::max(max_nout, jb.nout);
                int nt = 256;
                // (a role of a tick: every workgroup of the launch gets the largest role's LDS — stay near the other roles' ~40 KB where the filter allows)
                while (ticking && nt > 64 && lds_for(jb, nt) > (size_t)c->tick_lds_cap_fir) { nt >>= 1; }
                while (nt >= 32 && lds_for(jb, nt) > (size_t)kMaxLds) { nt >>= 1; }
                if (nt < 32) {
                    if (width != 2 || quad || stereo) { return fail(c, SDRPP_ERR_UNSUPPORTED, "FIR (decim %d, %d taps) does not fit in LDS", 1 << jb.log2_decim, jb.ntaps); }
                    threads = 0;  // complex stream: the untiled kernel takes the whole list
                    break;
                }
                threads = std::min(threads, nt);
            }
            if (threads == 0) {
                for (auto& jb : jobs) { max_nout = std::max(max_nout, jb.nout); }
                if (max_nout > 0) {
                    if (c->tick_planning) { emit(c, level, fam, TR_FIRD, std::min((max_nout + 255) / 256, 1024), (int)jobs.size(), 0, d_jobs); }
                    else { launch(c, vfo_fir_direct_kernel<false>, dim3((unsigned)std::min((max_nout + 255) / 256, 1024), (unsigned)jobs.size()), dim3(256), 0, (const FirBJob*)d_jobs); }
                }
                return SDRPP_OK;
            }
            if (max_nout == 0) { return SDRPP_OK; }
            // enough blocks to load-balance 256 CUs: shrink the tile while the grid has fewer than ~8 blocks per CU
            // (also as a role of a tick: a wider tile — fewer, longer workgroups in a tick that holds the long first stages' LDS anyway — measured neutral, profiles/r05zk)
            while (threads > 64 && (size_t)((max_nout + threads * R - 1) / (threads * R)) * jobs.size() < 2048) { threads >>= 1; }
            size_t lds = 0;
            for (auto& jb : jobs) { lds = std::max(lds, lds_for(jb, threads)); }
            const int tile = threads * R;
            emit(c, level, fam, width == 2 ? TR_FIRB_C : (quad ? TR_FIRB_Q : (stereo ? TR_FIRB_S : TR_FIRB_R)), (max_nout + tile - 1) / tile, (int)jobs.size(), lds, d_jobs, nullptr, threads);
            return SDRPP_OK;
    }
    // resamplers with many phases (L > 8, e.g. 96/125): cycle-major kernel — one LDS window serves all L phases of up to 64 cycles;
    // a filter whose single cycle does not fit falls back to the per-output kernel
    int launch_polyc(int level, int fam, std::vector<PolyJob>& jobs, PolyJob* d_jobs) {
            if (jobs.empty()) { return SDRPP_OK; }
            // LDS window of a tile: all 64 KB for a launch of its own (64 cycles per tile); as a role of a tick — whose workgroups all get the
            // largest role's LDS — about 24 KB (e.g. 19 cycles of the 96 / 125 resampler) unless a single cycle needs more
            int cap2 = kMaxLds / (int)sizeof(float2);
            if (ticking) {
                int need = c->tick_lds_cap / (int)sizeof(float2);
                for (auto& jb : jobs) { need = std::max(need, jb.tpp + 2 * jb.decim + 1); }
                cap2 = std::min(cap2, need);
            }
            bool fits = true;
            int max_nout = 0, max_tiles = 0;
            for (auto& jb : jobs) {
                max_nout = std::max(max_nout, jb.nout);
                const int ct = std::min(64, (cap2 - jb.tpp - jb.decim) / jb.decim);
                if (ct < 1) { fits = false; continue; }
                const int ncyc = (jb.nout + jb.interp - 1) / jb.interp;
                max_tiles = std::max(max_tiles, (ncyc + ct - 1) / ct);
            }
            if (max_nout == 0) { return SDRPP_OK; }
            if (fits) {
                // phase groups (a tick: always — a role's workgroup life is what the tick waits for): two phases per wavefront when the launch is a
                // handful of tiles (sr/200 blocks: 13 us -> 9 us of workgroup life, 767 -> 2 081 MS/s with the AF chain on 32 VFOs), six when there are
                // many (every group loads the tile's whole window again: 1 152 workgroups of 13.5 us were a third of a 10^6-sample tick's slot time)
                const int ppw = (long long)max_tiles * (long long)jobs.size() >= 64 ? 6 : 2;
                int G = 1;
                for (auto& jb : jobs) { G = std::max(G, ((jb.interp + 3) / 4 + ppw - 1) / ppw); }
                if (!ticking && (long long)max_tiles * (long long)jobs.size() >= 1024) { G = 1; }
                G = std::max(1, std::min(G, 128));
                emit(c, level, fam, TR_POLYC, max_tiles * G, (int)jobs.size(), (size_t)cap2 * sizeof(float2), d_jobs, nullptr, cap2 | ((G - 1) << 24));
                return SDRPP_OK;
            }
            size_t lds = 0;
            const int tile = 256;
            for (auto& jb : jobs) {
                const size_t ns = (size_t)((long long)tile * jb.decim / jb.interp) + jb.tpp + 4;
                lds = std::max(lds, ns * sizeof(float2));
            }
            if (lds > (size_t)kMaxLds) { return fail(c, SDRPP_ERR_UNSUPPORTED, "polyphase tile does not fit in LDS"); }
            if (c->tick_planning) { emit(c, level, fam, TR_POLY, (max_nout + tile - 1) / tile, (int)jobs.size(), lds, d_jobs); }
            else { launch(c, vfo_poly_kernel, dim3((max_nout + tile - 1) / tile, (unsigned)jobs.size()), dim3(tile), lds, (const PolyJob*)d_jobs); }
            return SDRPP_OK;
    }
    int launch_polyb(int li, std::vector<PolyBJob>& jobs, PolyBJob* d_jobs) {
            if (jobs.empty()) { return SDRPP_OK; }
            int max_cycles = 0, threads = 256;
            size_t lds = 0;
            auto lds_for = [&](const PolyBJob& jb, int nt) { return (size_t)jb.decim * (size_t)(nt + jb.rows / jb.decim + 2) * sizeof(float2); };
            for (auto& jb : jobs) {
                max_cycles = std::max(max_cycles, (jb.nout + jb.interp - 1) / jb.interp);
                int nt = 256;
                while (nt >= 32 && lds_for(jb, nt) > (size_t)kMaxLds) { nt >>= 1; }
                if (nt < 32) { return fail(c, SDRPP_ERR_UNSUPPORTED, "polyphase tile does not fit in LDS"); }
                threads = std::min(threads, nt);
            }
            if (max_cycles == 0) { return SDRPP_OK; }
            while (threads > 64 && (size_t)((max_cycles + threads - 1) / threads) * jobs.size() < 2048) { threads >>= 1; }
            for (auto& jb : jobs) { lds = std::max(lds, lds_for(jb, threads)); }
            const dim3 grid((max_cycles + threads - 1) / threads, (unsigned)jobs.size());
            if (li == 0) { launch(c, vfo_polyb_kernel<4, false>, grid, dim3(threads), lds, (const PolyBJob*)d_jobs); }
            else if (li == 1) { launch(c, vfo_polyb_kernel<8, false>, grid, dim3(threads), lds, (const PolyBJob*)d_jobs); }
            else if (li == 2) { launch(c, vfo_polyb_kernel<4, true>, grid, dim3(threads), lds, (const PolyBJob*)d_jobs); }
            else { launch(c, vfo_polyb_kernel<8, true>, grid, dim3(threads), lds, (const PolyBJob*)d_jobs); }
            return SDRPP_OK;
    }
    void emit_toep(int i, int l) {
            Lev<ToepJob>& L = *tlists[i].L;
            if (l >= L.top || L.at[l].empty() || tplan[i][l].grid_x == 0) { return; }
            emit(c, l, tlists[i].fam, tlists[i].role, tplan[i][l].grid_x, (int)L.at[l].size(), tplan[i][l].lds, L.dev[l]);
    }
    // the history carries of one level: job 0 of the IQ stream's level is the shared IQ stream (up to a whole FFT frame long), the per-VFO
    // histories are a few hundred samples
    void launch_carry(int l) {
            std::vector<CarryJob>& cj = carry.at[l];
            if (cj.empty()) { return; }
            const bool has_iq = (l == (ticking ? L0 + 1 : carry_last));
            const int iq_elems = has_iq ? cj[0].need * cj[0].width : 0;
            int mx = 0;
            for (size_t k = has_iq ? 1 : 0; k < cj.size(); k++) { mx = std::max(mx, cj[k].need * cj[k].width); }
            // (carry_body moves 4 floats per access and up to 8 accesses per work-item: 8 192 floats per workgroup and round, 2 048 per wavefront)
            // The per-VFO histories, a few hundred samples each: one WAVEFRONT per job, four jobs per workgroup (carry_body, njw > 0) unless a
            // history is long enough to want a workgroup's worth of loads in flight; the shared IQ history: an entry of its own.
            auto per_vfo = [&](const CarryJob* dev, int njobs) {
                if (njobs <= 0) { return; }
                if (mx <= carry_wave_max) { emit(c, l, F_MISC, TR_CARRY, 1, (njobs + 3) / 4, 0, dev, nullptr, njobs); }
                else { emit(c, l, F_MISC, TR_CARRY, std::max(1, std::min((mx + 8191) / 8192, 64)), njobs, 0, dev); }
            };
            if (!ticking && iq_elems > carry_wave_max && iq_elems <= 128 * 1024 * 2 && cj.size() > 1) {
                // an ordinary pass: one launch for the IQ history (up to a 65 536-point frame: 32 workgroups stride over it) and the per-VFO histories
                // (their workgroups beyond the first find nothing to do) — a kernel and its dispatch bubble less per push
                emit(c, l, F_MISC, TR_CARRY, 32, (int)cj.size(), 0, carry.dev[l]);
            }
            else if (has_iq && (iq_elems > carry_wave_max || cj.size() == 1)) {
                emit(c, l, F_MISC, TR_CARRY, std::max(1, std::min((iq_elems + 8191) / 8192, 512)), 1, 0, carry.dev[l]);
                per_vfo(carry.dev[l] + 1, (int)cj.size() - 1);
            }
            else {  // (no IQ history at this level, or one as short as the others)
                mx = std::max(mx, iq_elems);
                per_vfo(carry.dev[l], (int)cj.size());
            }
    }

    // ---- levels 2 ...: everything behind the front end, level by level (within a level the launches are independent of each other) ----
    int emit_levels() {
        int rc = SDRPP_OK;
        int top = std::max({ t_dec.top, t_poly.top, t_chan.top, t_audio.top, t_audio_fm.top, t_af_dec.top, t_af_poly.top, t_af_hpf.top, f_dec.top, poly.top,
                             polyb[0].top, polyb[1].top, polyb[2].top, polyb[3].top, chan.top, seq.top, pre.top, audio.top, audio_fm.top, af_dec.top, af_hpf.top,
                             af_poly.top, af_deemp.top, ssbx_l.top, carry.top, pipe_top });
        for (int l = 1; l < top; l++) {
            {
                FamilyTimer t(c, F_DECIM);
                emit_toep(0, l);
                if (l < f_dec.top) {
                    rc = launch_fir(l, F_DECIM, f_dec.at[l], f_dec.dev[l], 2, false);
                    if (rc) { return rc; }
                }
            }
            for (auto& g : pgroups) {
                if (g.lvl != l || g.seg <= 0) { continue; }
                FamilyTimer t(c, F_PIPE);
                c->pipe_launched = true;
                emit(c, l, F_PIPE, TR_PIPE, g.seg, (int)g.jobs.size(), pipe_lds, g.dev);
            }
            {
                FamilyTimer t(c, F_POLY);
                emit_toep(1, l);
                if (l < poly.top) {
                    rc = launch_polyc(l, F_POLY, poly.at[l], poly.dev[l]);
                    if (rc) { return rc; }
                }
                for (int li = 0; li < 4; li++) {
                    if (l < polyb[li].top) {
                        rc = launch_polyb(li, polyb[li].at[l], polyb[li].dev[l]);
                        if (rc) { return rc; }
                    }
                }
            }
            {
                FamilyTimer t(c, F_FIR);
                emit_toep(2, l);
                if (l < chan.top) {
                    rc = launch_fir(l, F_FIR, chan.at[l], chan.dev[l], 2, false);
                    if (rc) { return rc; }
                }
            }
            if ((l < pre.top && !pre.at[l].empty()) || (l < seq.top && !seq.at[l].empty()) || (l < ssbx_l.top && !ssbx_l.at[l].empty())) {
                FamilyTimer t(c, F_DEMOD);
                if (l < ssbx_l.top && !ssbx_l.at[l].empty()) {
                    const int nj = (int)ssbx_l.at[l].size();
                    if (c->tick_planning) { emit(c, l, F_DEMOD, TR_SSBX, (nj + 3) / 4, 1, 0, ssbx_l.dev[l], nullptr, nj); }
                    else { launch(c, vfo_ssb_rotate_exact_kernel, dim3((unsigned)nj), dim3(64), 0, (const SsbRotXJob*)ssbx_l.dev[l]); }
                }
                if (l < pre.top && !pre.at[l].empty()) {
                    int mx = 0;
                    for (auto& q : pre.at[l]) { mx = std::max(mx, q.n); }
                    if (mx > 0) { emit(c, l, F_DEMOD, TR_PRE, std::min((mx + 255) / 256, 1024), (int)pre.at[l].size(), 0, pre.dev[l]); }
                }
                if (l < seq.top && !seq.at[l].empty()) { emit(c, l, F_DEMOD, TR_SEQ, (int)seq.at[l].size(), 1, 0, seq.dev[l], nullptr, (int)seq.at[l].size()); }
            }
            {
                FamilyTimer t(c, F_FIR);
                emit_toep(3, l);
                emit_toep(4, l);
                if (l < audio.top) {
                    rc = launch_fir(l, F_FIR, audio.at[l], audio.dev[l], 1, true);
                    if (rc) { return rc; }
                }
                if (l < audio_fm.top) {
                    rc = launch_fir(l, F_FIR, audio_fm.at[l], audio_fm.dev[l], 1, true, true);
                    if (rc) { return rc; }
                }
            }
            if (l < std::max({ t_af_dec.top, t_af_poly.top, t_af_hpf.top, af_dec.top, af_hpf.top, af_poly.top, af_deemp.top })) {
                FamilyTimer t(c, F_AF);
                emit_toep(5, l);
                if (l < af_dec.top) {
                    rc = launch_fir(l, F_AF, af_dec.at[l], af_dec.dev[l], 2, false);
                    if (rc) { return rc; }
                }
                emit_toep(6, l);
                if (l < af_poly.top) {
                    rc = launch_polyc(l, F_AF, af_poly.at[l], af_poly.dev[l]);
                    if (rc) { return rc; }
                }
                emit_toep(7, l);
                if (l < af_hpf.top) {
                    rc = launch_fir(l, F_AF, af_hpf.at[l], af_hpf.dev[l], 2, false);
                    if (rc) { return rc; }
                }
                if (l < af_deemp.top && !af_deemp.at[l].empty()) {
                    int max_seg = 0;
                    for (auto& jb : af_deemp.at[l]) { max_seg = std::max(max_seg, jb.nseg); }
                    if (max_seg > 0) {  // segment maps at this level, the outputs (and the state the next block starts from) one level later
                        emit(c, l, F_AF, TR_DEEMP_P0, max_seg, (int)af_deemp.at[l].size(), 3 * 256 * sizeof(float), af_deemp.dev[l]);
                        emit(c, l + 1, F_AF, TR_DEEMP_P1, max_seg, (int)af_deemp.at[l].size(), 3 * 256 * sizeof(float), af_deemp.dev[l]);
                    }
                }
            }
            if (l < carry.top && !carry.at[l].empty()) {
                FamilyTimer t(c, F_MISC);
                launch_carry(l);
            }
        }
        return SDRPP_OK;
    }
};

int do_vfos_plan(sdrpp_ctx* c, const IqSrc& src, int64_t count, const CarryJob& iq_carry) {
    if (c->vfos.empty()) { return SDRPP_OK; }
#ifdef SDRPP_TOEP_KNOCK
    {   // diagnostic build: SDRPP_TOEP_KNOCK=<mask> (1: no stores, 2: no loads, 4: no matrix loop) in vfo_toep_kernel
        static bool once = false;
        if (!once) {
            once = true;
            const int m = getenv("SDRPP_TOEP_KNOCK") ? atoi(getenv("SDRPP_TOEP_KNOCK")) : 0;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(sdrpp_k::g_toep_knock), &m, sizeof(int));
        }
    }
#endif
    HostScope hs_all("plan: vfo bank");
    if (!c->bank_plan) {
        c->bank_plan = new BankPlan(c);
        c->bank_plan_free = [](void* q) { delete static_cast<BankPlan*>(q); };
    }
    BankPlan* P = static_cast<BankPlan*>(c->bank_plan);
    P->begin(src, count, iq_carry);
    int rc = SDRPP_OK;
    {
        HostScope hs("plan: chains");
        for (size_t i = 0; i < c->vfo_list.size(); i++) {
            rc = P->chain(*c->vfo_list[i]);
            if (rc) { return rc; }
        }
    }
    {
        HostScope hs("plan: group_front");
        rc = P->group_front();
    }
    if (!rc) {
        HostScope hs("plan: upload");
        rc = P->upload();
    }
    if (!rc) {
        HostScope hs("plan: emit");
        rc = P->emit_front();
        if (!rc) { rc = P->emit_levels(); }
    }
    if (rc) { return rc; }
    // flip the ping-pong side of every carried stream
    for (size_t i = 0; i < c->vfo_list.size(); i++) {
        for (auto& s : c->vfo_list[i]->st) {
            if (s.hist_len > 0 && s.data) { s.cur ^= 1; }
        }
    }
    return SDRPP_OK;
}

}  // namespace