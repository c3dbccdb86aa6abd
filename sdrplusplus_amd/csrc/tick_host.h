// Pipelined execution, host side (tick_kernels.h): role queue, one launch per block, result slots, drain.
// Part of the one translation unit sdrpp_gpu.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

// =====================================================================================================================
// Pipelined execution (tick_kernels.h): queue, launch, drain
// =====================================================================================================================
void tick_wait_done(sdrpp_ctx* c, uint64_t nticks) {
    if (!c->h_tick_flag) { return; }
    // the flag holds the number of completed ticks modulo 2^32; at most kArenaSlots ticks are ever outstanding
    const volatile unsigned* f = c->h_tick_flag;
    long spins = 0;
    while ((int)((unsigned)nticks - *f) > 0) {
        if (++spins > 2000) { std::this_thread::yield(); }
        if (spins > 40000000) {  // ~minutes: the device is gone; let the next HIP call report it
            (void)hipStreamSynchronize(c->stream);
            if ((int)((unsigned)nticks - *f) > 0) { return; }
        }
    }
}
// An event behind everything launched so far (c->ticks), from the ring.  `pre`: only reserve the ring entry and return it — the caller hands
// the event to the launch itself (hipExtLaunchKernelGGL's stop event).
hipEvent_t tick_event_take(sdrpp_ctx* c, uint64_t behind_tick) {
    const int k = c->tick_ev_next;
    c->tick_ev_next = (k + 1) % sdrpp_ctx::kTickEvents;
    tick_ev_resolve(c, k);  // (the launch it timed is kTickEvents ticks old: long complete)
    if (!c->tick_ev[k] && hipEventCreate(&c->tick_ev[k]) != hipSuccess) { return nullptr; }
    c->tick_ev_tick[k] = behind_tick;
    c->tick_ev_last = behind_tick;
    return c->tick_ev[k];
}
// The results tick `nticks` wrote into page-locked host memory are visible to this thread: 1 yes, 0 not yet (wait = false), < 0 error.
// The flag first (cheap: a cached word the device writes once per tick), then the first event recorded at or behind that tick — a later
// tick's event orders the earlier tick's writes just as well; if there is none yet, one is recorded now behind everything launched.
int tick_results_visible(sdrpp_ctx* c, uint64_t nticks, bool wait) {
    if (nticks == 0) { return 1; }
    if (wait) { tick_wait_done(c, nticks); }
    else if (c->h_tick_flag && (int)((unsigned)nticks - *(const volatile unsigned*)c->h_tick_flag) > 0) { return 0; }
    int best = -1;
    for (int k = 0; k < sdrpp_ctx::kTickEvents; k++) {
        if (c->tick_ev[k] && c->tick_ev_tick[k] >= nticks && (best < 0 || c->tick_ev_tick[k] < c->tick_ev_tick[best])) { best = k; }
    }
    hipEvent_t ev = best >= 0 ? c->tick_ev[best] : nullptr;
    if (!ev) {
        ev = tick_event_take(c, c->ticks);
        if (!ev || hipEventRecord(ev, c->stream) != hipSuccess) { return fail(c, SDRPP_ERR_HIP, "hipEventRecord failed"); }
    }
    long spins = 0;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) { return 1; }
        if (e != hipErrorNotReady) { return fail(c, SDRPP_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(e)); }
        if (!wait) { return 0; }
        if (++spins > 200) { std::this_thread::yield(); }
    }
}
bool tick_is_done(const sdrpp_ctx* c, uint64_t nticks) { return !c->h_tick_flag || (int)((unsigned)nticks - *(const volatile unsigned*)c->h_tick_flag) <= 0; }

// One tick: level-0 work of the block that arrives with it (`land`: its landing copy, may be null; the arena slot the caller has filled
// with the block's job tables) + every queued role whose turn it is.  The role table of the NEXT tick is appended to the arena slot and
// travels with this tick's upload.
// expected lifetime of a workgroup of a role relative to the others (tools/tick_trace.py timelines), for the order inside a tick
inline int tick_role_weight(int role, bool crowded) {
    // result copies write page-locked host memory over the bus: few workgroups whose life is mostly that round trip — started last they are the
    // tail of the tick, started first they finish in its shadow (SDRPP_GPU_TICK_COPY_FIRST=0: the old order, for measurements)
    static const bool copy_first = getenv("SDRPP_GPU_TICK_COPY_FIRST") ? atoi(getenv("SDRPP_GPU_TICK_COPY_FIRST")) != 0 : true;
    if (role == TR_ROTX16) { return 120; }  // the reference's rotator recursion over the whole block: milliseconds — it IS the tick, everything else runs in its shadow
    if (role == TR_COPY && copy_first) { return 110; }
    // FFT pass 1: since its workgroups walk their tiles and take the lean loader they live ~12 us at 10^6-sample blocks, shorter than the
    // Toeplitz roles' 17-33 us.  In a CROWDED tick (more workgroups than the GPU holds at once: they are handed out in index order) they go
    // behind the filters and pass 2, so that the tick ends on short workgroups: 18.6 -> 19.6 GS/s at 10^6-sample blocks; in a tick whose
    // workgroups are all resident from the start the order only decides who gets going first, and pass 1 early is worth 2 % at 200 000-sample
    // blocks (profiles/r03z_tick_p1_weight.log; SDRPP_GPU_TICK_P1_WEIGHT: measurement switch)
    static const int p1_weight = getenv("SDRPP_GPU_TICK_P1_WEIGHT") ? atoi(getenv("SDRPP_GPU_TICK_P1_WEIGHT")) : 0;
    if (role >= TR_FFT_P1_5 && role <= TR_FFT_P1_10) { return p1_weight > 0 ? p1_weight : (crowded ? 45 : 70); }
    // the long first stages (cfg 4): far more workgroups than the GPU holds, 58 us each — behind the sequential recursions (few, long), in front
    // of everything else: with four tile engines per workgroup they are 60 % of the tick's workgroup time, and started late they ARE its tail
    // (weight 72 against 58: 10^6-sample blocks 6.02 -> 6.13 GS/s, 307 200: 3.19 -> 3.78; behind the filters, 40 / 15: 5.77 / 5.64 and 3.11 / 3.07;
    // profiles/r04w_fcl_weight.log — with two engines per workgroup, round 3, the order hardly mattered: profiles/r03zj_*;
    // SDRPP_GPU_TICK_FCL_WEIGHT: measurement switch)
    static const int fcl_weight = getenv("SDRPP_GPU_TICK_FCL_WEIGHT") ? atoi(getenv("SDRPP_GPU_TICK_FCL_WEIGHT")) : 72;
    switch (role) {
    case TR_FCL_0: case TR_FCL_PF: return fcl_weight;
    case TR_FCM_132_4: case TR_FCM_6: case TR_FCM_10: case TR_FCM_16: case TR_FCM16_132_4: case TR_S1_1: case TR_S1D_1: case TR_F2_1: return 90;
    case TR_SEQ: case TR_SSBX: return 85;
    case TR_FFT_P1_5: case TR_FFT_P1_6: case TR_FFT_P1_7: case TR_FFT_P1_8: case TR_FFT_P1_9: case TR_FFT_P1_10: case TR_FIRB_C: case TR_FIRB_R: case TR_FIRB_S: case TR_FIRB_Q: return 70;
    case TR_POLYC: return 55;
    case TR_DEEMP_P1: case TR_DC_P1: return 52;
    case TR_DEEMP_P0: case TR_DC_P0: return 48;
    case TR_PIPE: return 80;  // the longest-lived workgroups behind the front end: a whole segment of four stages
    case TR_TOEP_Q: return 65;
    case TR_TOEP_C: case TR_TOEP_R: case TR_FFT_S10: case TR_FFT_S11: case TR_FFT_S12: return 60;
    case TR_FFT_P2_7: case TR_FFT_P2_8: case TR_FFT_P2_9: case TR_FFT_P2_10: case TR_FFT_P2ROW: {
        static const int p2_weight = getenv("SDRPP_GPU_TICK_P2_WEIGHT") ? atoi(getenv("SDRPP_GPU_TICK_P2_WEIGHT")) : 50;  // (measurement switch)
        return p2_weight;
    }
    case TR_FFT_TR: return 25;
    case TR_ROT: case TR_PRE: return 30;
    case TR_ZOOM_16: case TR_ZOOM_4: case TR_ZOOM_1: return 20;
    default: return 10;  // carry, copies
    }
}
int tick_launch(sdrpp_ctx* c, const CopyJob* land) {
    std::vector<sdrpp_ctx::RoleLaunch> now;
    if (!c->tickq.empty()) {
        now.swap(c->tickq.front());
        c->tickq.pop_front();
    }
    if ((int)now.size() != c->next_tab_n) { return fail(c, SDRPP_ERR_HIP, "internal: tick table out of step (%zu roles queued, %d uploaded)", now.size(), c->next_tab_n); }
    // the table of the tick after this one
    TickTable* tab_dev_next = nullptr;
    int tab_n_next = 0;
    if (!c->tickq.empty() && !c->tickq.front().empty()) {
        std::vector<sdrpp_ctx::RoleLaunch>& nx = c->tickq.front();
        // The roles of a tick are independent of each other, so their order is free — and the hardware hands out workgroups in index
        // order: longest workgroups first (front ends, FFT pass 1, the filters; zoom / carry / copies last), so that the tick ends on
        // short ones instead of on a front end that only got its turn when everything else was through (10^6-sample blocks: the front
        // end started 46 us into an 81 us tick).  The table is final here: later blocks only add to later ticks.
        if (c->tick_order) {
            long long wgs = 0;
            for (auto& r : nx) { wgs += (long long)r.e.gx * r.e.gy; }
            const bool crowded = wgs > 3ll * c->num_cus;  // (three workgroups of the tick kernel per CU)
            std::stable_sort(nx.begin(), nx.end(), [crowded](const sdrpp_ctx::RoleLaunch& a, const sdrpp_ctx::RoleLaunch& b) { return tick_role_weight(a.e.role, crowded) > tick_role_weight(b.e.role, crowded); });
        }
        if (nx.size() > SDRPP_TICK_MAX_ENTRIES) { return fail(c, SDRPP_ERR_UNSUPPORTED, "internal: %zu roles in one tick", nx.size()); }
        const size_t off = (c->arena_off + 63) & ~(size_t)63;
        if (off + sizeof(TickTable) > kArenaBytes) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
        TickTable* T = reinterpret_cast<TickTable*>(c->arena_host[c->arena_slot] + off);
        T->n = (int)nx.size();
        int total = 0;
        for (size_t i = 0; i < nx.size(); i++) {
            total += nx[i].e.gx * nx[i].e.gy;
            T->block_end[i] = total;
            T->e[i] = nx[i].e;
        }
        c->arena_off = off + offsetof(TickTable, e) + nx.size() * sizeof(TickEntry);
        tab_dev_next = reinterpret_cast<TickTable*>(c->arena_dev + off);
        tab_n_next = (int)nx.size();
    }
    TickL0 l0{};
    if (land && land->bytes > 0) {
        l0.job[0] = *land;
        l0.blocks[0] = (int)std::max<long long>(1, std::min<long long>((land->bytes + 8191) / 8192, c->tick_land_blocks));
    }
    if (c->arena_off > 0) {
        l0.job[1] = CopyJob{ c->arena_host_dev[c->arena_slot], c->arena_dev, (long long)((c->arena_off + 15) & ~(size_t)15), 0, 0 };
        l0.blocks[1] = (int)std::max<size_t>(1, std::min<size_t>((c->arena_off + 4095) / 4096, 16));  // one 16-byte load per work-item: a round trip over the bus each
    }
    int blocks = l0.blocks[0] + l0.blocks[1];
    size_t lds = 0;
    bool set1 = false, set2 = true;  // set2: every role of this tick exists in the four-wavefronts-per-SIMD build (tick_kernels.h)
    long long role_wgs = 0;
    bool to_host = false;
    for (auto& r : now) {
        to_host = to_host || r.to_host;
        blocks += r.e.gx * r.e.gy;
        role_wgs += (long long)r.e.gx * r.e.gy;
        lds = std::max(lds, r.lds);
        set1 = set1 || r.e.role == TR_FCL_PF || r.e.role == TR_FCM_6 || r.e.role == TR_FCM_10 || r.e.role == TR_FCM_16;  // (roles that exist in the 247-register build only: tick_kernels.h)
        set2 = set2 && !(r.e.role == TR_ROTX16 || r.e.role == TR_FCL_PF || r.e.role == TR_FCL_0 || r.e.role == TR_FCM_132_4 || r.e.role == TR_FCM_6 || r.e.role == TR_FCM_10 || r.e.role == TR_FCM_16 || r.e.role == TR_FFT_S12);
        if (r.e.role >= 0 && r.e.role < 64) { c->stat_role_wgs[r.e.role] += (int64_t)r.e.gx * r.e.gy; }
    }
    if (role_wgs > 3ll * c->num_cus) { c->stat_crowded++; }
    {   // where the stage-0 copies stand among the tick's workgroups (SDRPP_GPU_TICK_L0_AT: 0 = in front (default), -1 = behind all roles, n = behind
        // the first n role workgroups): a switch for the measurement DESIGN_HISTORY.md 4b names as the next step
        const int l0_at = c->tick_l0_at;
        const int role_blocks = blocks - l0.blocks[0] - l0.blocks[1];
        l0.first = l0_at < 0 ? role_blocks : std::min(l0_at, role_blocks);
    }
    if (blocks == 0) {  // nothing to do at all: a level of the queue without roles (an idle flush; a gap between a block's levels)
        c->next_tab = tab_dev_next;
        c->next_tab_n = tab_n_next;
        // No launch, so the tick counter the device publishes does not move — but every block still in the queue has just advanced by one level:
        // a block's completion tick was computed as "ticks now + its remaining levels", one COUNTED tick per level, and would lie one launch past
        // the end of the drain for every empty level (sdrpp_result_wait: "block cannot complete").  Its remaining levels are one fewer now.
        for (auto& R : c->res) {
            if (R.ticket != 0 && R.done_tick > c->ticks) { R.done_tick--; }
        }
        return SDRPP_OK;
    }
    c->tick_target += (unsigned)blocks;
    c->ticks++;
    TickDone done{ c->d_tick_counter, c->hd_tick_flag, c->tick_target, (unsigned)c->ticks };
    hipEvent_t stop_ev = nullptr;
    const TickTable* tab = c->next_tab_n > 0 ? c->next_tab : c->empty_tab;
    {
        // Completion events ride on the launch itself (hipExtLaunchKernelGGL's start / stop events: the dispatch packet's own signal, no packet
        // of their own in the queue): the stop event orders the tick's result writes for the host (host_ctx.h), start + stop time the launch.
        const bool timed = c->timing && ((c->timing_mask >> F_TICK) & 1u);
        c->fam_launch[F_TICK]++;
        bool want_ev = timed;
        if (to_host && c->tick_ev_ext && ++c->tick_ev_skipped >= c->tick_ev_every) {
            c->tick_ev_skipped = 0;
            want_ev = true;
        }
        hipEvent_t ea = nullptr;
        if (want_ev) {
            const int k = c->tick_ev_next;
            stop_ev = tick_event_take(c, c->ticks);
            if (!stop_ev) { return fail(c, SDRPP_ERR_HIP, "hipEventCreate failed"); }
            if (timed) {
                ea = get_event(c);
                c->tick_ev_start[k] = ea;
            }
        }
        HostScope hs("launch");
        static const bool allow2 = getenv("SDRPP_GPU_TICK_SET2") ? atoi(getenv("SDRPP_GPU_TICK_SET2")) != 0 : true;  // (measurement switch)
        const bool use2 = allow2 && set2 && !set1 && c->tick_fcm16w && lds <= (size_t)40 * 1024;
        if (use2) { c->stat_set2++; }
        if (stop_ev) {
            if (set1) { hipExtLaunchKernelGGL((tick_kernel<1>), dim3((unsigned)blocks), dim3(256), (unsigned)lds, c->stream, ea, stop_ev, 0, l0, tab, done); }
            else if (use2) { hipExtLaunchKernelGGL((tick_kernel<2>), dim3((unsigned)blocks), dim3(256), (unsigned)lds, c->stream, ea, stop_ev, 0, l0, tab, done); }
            else { hipExtLaunchKernelGGL((tick_kernel<0>), dim3((unsigned)blocks), dim3(256), (unsigned)lds, c->stream, ea, stop_ev, 0, l0, tab, done); }
        }
        else if (set1) { hipLaunchKernelGGL((tick_kernel<1>), dim3((unsigned)blocks), dim3(256), lds, c->stream, l0, tab, done); }
        else if (use2) { hipLaunchKernelGGL((tick_kernel<2>), dim3((unsigned)blocks), dim3(256), lds, c->stream, l0, tab, done); }
        else { hipLaunchKernelGGL((tick_kernel<0>), dim3((unsigned)blocks), dim3(256), lds, c->stream, l0, tab, done); }
    }
    if (to_host && !c->tick_ev_ext) {  // measurement switch: the event as a packet of its own behind the launch
        if (++c->tick_ev_skipped >= c->tick_ev_every) {
            c->tick_ev_skipped = 0;
            hipEvent_t ev = tick_event_take(c, c->ticks);
            if (!ev) { return fail(c, SDRPP_ERR_HIP, "hipEventCreate failed"); }
            HIPCHK(c, hipEventRecord(ev, c->stream));
        }
    }
    c->arena_tick[c->arena_slot] = c->ticks;
    c->next_tab = tab_dev_next;
    c->next_tab_n = tab_n_next;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { return fail(c, SDRPP_ERR_HIP, "tick launch failed: %s", hipGetErrorString(e)); }
    return SDRPP_OK;
}
// run every queued role (no new input): what a caller that wants the results of the last blocks NOW pays for the skew
int tick_drain(sdrpp_ctx* c) {
    while (!c->tickq.empty()) {
        int rc = arena_begin(c);
        if (rc) { return rc; }
        rc = tick_launch(c, nullptr);
        if (rc) { return rc; }
    }
    return SDRPP_OK;
}

// Can this context's blocks run as ticks at all?  (What can only be seen while planning — a VFO group too small for the matrix front end,
// a filter without the matrix form, more frames than one scratch chunk — aborts the plan instead.)
bool tick_eligible(sdrpp_ctx* c) {
    if ((c->pre.on && c->pre.ref_order) || c->deferred) { return false; }  // (the reference-order arithmetic of the pre-processing chain has no roles)
    for (auto& kv : c->vfos) {
        const Vfo& v = *kv.second;
        // (reference-rotator VFOs have their roles since round 5 — TR_ROTX16 / TR_FIRD / TR_SSBX; the measurement forms of that rotator have not)
        if ((v.nco_exact && (c->rot_exact_single || c->rot_exact_skip < 16)) || !v.recs.empty() || v.st.size() > 24) { return false; }
    }
    return true;
}

// ---- results of a block in page-locked host memory (sdrpp_set_pipelined's result flags): gather roles one level behind the producers ----
// What result flag 1 delivers per VFO: the end of its chain — the AF chain's output where one is attached (what the radio module's audio
// stream carries, radio_module.h:98-110), else the demodulator's, else (no demodulator) the IF stream — and the level it is written at.
const Stream& result_stream(const Vfo& v, int* level = nullptr) {
    if (v.af.on && v.af.i_last >= 0 && v.d.demod != SDRPP_DEMOD_RAW) {
        if (level) { *level = v.lvl_af; }
        return v.st[(size_t)v.af.i_last];
    }
    const bool raw = v.d.demod == SDRPP_DEMOD_RAW;
    if (level) { *level = raw ? v.lvl_if : v.lvl_out; }
    return raw ? v.st[(size_t)v.i_if] : v.st[(size_t)v.i_out];
}
size_t tick_results_need(sdrpp_ctx* c) {
    size_t need = 0;
    if (c->res_flags & 1) {
        for (auto& kv : c->vfos) {
            const Vfo& v = *kv.second;
            size_t cap = std::max(result_stream(v).cap, v.st[(size_t)v.i_if].cap);  // (which stream ends the chain can change with sdrpp_vfo_set_af: room for either)
            if (v.i_out >= 0) { cap = std::max(cap, v.st[(size_t)v.i_out].cap); }
            need += ((cap + 16) * 8 + 15) & ~(size_t)15;
        }
    }
    if ((c->res_flags & 8) && c->pre.on) {
        size_t cap = c->pre.out.base ? c->pre.out.cap : 0;
        for (auto& st : c->pre.st) { cap = std::max(cap, st.cap); }
        need += ((cap + 16) * 8 + 15) & ~(size_t)15;
    }
    if (c->fft_on) {
        if ((c->res_flags & 2) && c->data_width > 0) { need += 2 * ((c->lines_cap * (size_t)c->data_width * 4 + 15) & ~(size_t)15); }
        if (c->res_flags & 4) { need += (c->lines_cap * (size_t)c->fft_size * 4 + 15) & ~(size_t)15; }
    }
    return need;
}
int tick_results_ensure(sdrpp_ctx* c) {
    const size_t need = tick_results_need(c);
    if (need <= c->res_cap && c->res_ring) { return SDRPP_OK; }
    // The ring grows (a VFO was added, a larger view / FFT configured) in the middle of a run: results that are complete or on their way
    // but not yet collected must survive — a host keeps tickets across such a change (IQFrontEnd: pendingTickets across tempStop / addVFO).
    // Everything queued runs to its end first, then the ring's content moves into the larger one at the same offsets; a ring of which the host
    // is holding blocks (handed out by sdrpp_result_wait) stays alive until they are released — the pointers the host was given stay valid.
    int rc = tick_drain(c);
    if (rc) { return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    tick_wait_done(c, c->ticks);
    const size_t cap = need + need / 8 + 4096;
    const size_t bytes = (size_t)kResSlots * ((cap + 4095) & ~(size_t)4095);
    char* nh = nullptr;
    char* nd = nullptr;
    if (hipHostMalloc((void**)&nh, bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer((void**)&nd, nh, 0) != hipSuccess) {
        if (nh) { (void)hipHostFree(nh); }
        return fail(c, SDRPP_ERR_NOMEM, "page-locked result ring of %zu bytes", bytes);
    }
    if (c->res_ring) {
        int held = 0;
        for (auto& g : c->res_live) {
            memcpy(nh + g.off, c->res_ring + g.off, g.bytes);
            held += g.held;
        }
        for (auto& R : c->res) {  // blocks the host has not asked for yet follow their bytes into the new ring (held ones keep the address they were given)
            if (R.ticket != 0 && !R.held && R.epoch == c->res_epoch && R.base) {
                R.base = nh + (R.base - c->res_ring);
                R.epoch = c->res_epoch + 1;
            }
        }
        if (held > 0) { c->res_retired.push_back(sdrpp_ctx::ResRetired{ c->res_ring, c->res_epoch, held }); }
        else { (void)hipHostFree(c->res_ring); }
    }
    c->res_ring = nh;
    c->res_ring_dev = nd;
    c->res_ring_bytes = bytes;
    c->res_cap = cap;
    c->res_epoch++;
    return SDRPP_OK;
}
// a region of the result ring for launch group `gid`; regions the ring has come round to are taken back, oldest first (their blocks' results are
// gone: sdrpp_result_wait says so), unless the host still holds one — then nothing is touched and the push fails
int tick_results_alloc(sdrpp_ctx* c, uint64_t gid, size_t bytes, size_t* off_out) {
    const size_t need = (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
    if (!c->res_ring || need > c->res_ring_bytes) { return fail(c, SDRPP_ERR_INVALID, "internal: results of %zu bytes, ring of %zu", need, c->res_ring_bytes); }
    size_t off = c->res_head;
    if (off + need > c->res_ring_bytes) { off = 0; }
    size_t npop = 0;
    for (size_t i = 0; i < c->res_live.size(); i++) {
        const sdrpp_ctx::ResRegion& g = c->res_live[i];
        if (g.off < off + need && off < g.off + g.bytes) { npop = i + 1; }
    }
    for (size_t i = 0; i < npop; i++) {
        if (c->res_live[i].held > 0) {
            return fail(c, SDRPP_ERR_INVALID, "results of launch group %llu are still held (%d blocks) and the result ring has come round to them: release results sooner", (unsigned long long)c->res_live[i].gid,
                        c->res_live[i].held);
        }
    }
    c->res_live.erase(c->res_live.begin(), c->res_live.begin() + (long)npop);
    c->res_live.push_back(sdrpp_ctx::ResRegion{ gid, off, need, 0 });
    c->res_head = off + need;
    *off_out = off;
    return SDRPP_OK;
}
sdrpp_ctx::ResRegion* tick_results_region(sdrpp_ctx* c, uint64_t gid) {
    for (auto it = c->res_live.rbegin(); it != c->res_live.rend(); ++it) {
        if (it->gid == gid) { return &*it; }
        if (it->gid < gid) { break; }
    }
    return nullptr;
}

// ---- what a block's results consist of: one push — or a launch group of k pushes (c->grp_ends: their cumulative ends), planned as ONE block ----
// Every VFO's output block of the group lies in the group's result slot in one piece; push j's share of it is the samples between the push ends
// carried down the VFO's chain (Vfo::tk_if / tk_af, plan_vfo.h), its lines the frames whose last sample arrived with it.
struct ResCopy { const void* src; size_t off; size_t bytes; int level; };
int64_t tick_frames_by(const sdrpp_ctx* c, int64_t e) {  // lines complete once the first `e` samples of the block just planned are in (do_fft's own count)
    const int64_t P = (int64_t)c->nz + c->skip;
    const int64_t a = c->plan_fft_pos0 + e - c->nz - c->plan_fft_next0 * P;
    return a >= 0 ? a / P + 1 : 0;
}
int tick_results_describe(sdrpp_ctx* c, uint64_t first_ticket, int k, std::vector<ResCopy>& copies, size_t* region_off) {
    copies.clear();
    *region_off = 0;
    if (!c->res_flags) {
        for (int j = 0; j < k; j++) {
            sdrpp_ctx::Result& R0 = c->res[(first_ticket + (uint64_t)j) % kResMeta];
            if (!R0.held) { R0 = sdrpp_ctx::Result{}; }
        }
        return SDRPP_OK;
    }
    const bool split = k > 1;
    if (split && (int)c->grp_ends.size() != k) { return fail(c, SDRPP_ERR_INVALID, "internal: %d blocks, %zu push ends", k, c->grp_ends.size()); }
    sdrpp_ctx::Result* R[kGroupMax];
    for (int j = 0; j < k; j++) {
        R[j] = &c->res[(first_ticket + (uint64_t)j) % kResMeta];
        if (R[j]->held) { return fail(c, SDRPP_ERR_INVALID, "results of block %llu are still held %d pushes later: release results sooner", (unsigned long long)R[j]->ticket, kResMeta); }
    }
    for (int j = 0; j < k; j++) {
        *R[j] = sdrpp_ctx::Result{};
        R[j]->ticket = first_ticket + (uint64_t)j;
        R[j]->group = c->groups;
        R[j]->fft_size = c->fft_size;
        R[j]->data_width = c->data_width;
        R[j]->flags = c->res_flags;
    }
    size_t off = 0;
    if (c->res_flags & 1) {
        for (auto& kv : c->vfos) {
            const Vfo& v = *kv.second;
            int lvl = 1;
            const Stream& s = result_stream(v, &lvl);
            const std::vector<int>& tk = (&s == &v.st[(size_t)v.i_if] || (v.i_out >= 0 && &s == &v.st[(size_t)v.i_out])) ? v.tk_if : v.tk_af;
            if (split && ((int)tk.size() != k || tk[(size_t)k - 1] != s.n)) { return fail(c, SDRPP_ERR_INVALID, "internal: push ends of VFO %d do not add up (%zu ends, %d samples)", v.id, tk.size(), s.n); }
            for (int j = 0; j < k; j++) {
                const int lo = (split && j > 0) ? tk[(size_t)j - 1] : 0, hi = split ? tk[(size_t)j] : s.n;
                R[j]->ids.push_back(v.id);
                R[j]->offsets.push_back((int64_t)(off / 8) + lo);
                R[j]->counts.push_back(hi - lo);
            }
            const size_t bytes = (size_t)s.n * 8;
            if (bytes) { copies.push_back(ResCopy{ s.data, off, bytes, lvl + 1 }); }
            off += (bytes + 15) & ~(size_t)15;
        }
    }
    if ((c->res_flags & 8) && c->pre.on && c->pre.last_n > 0 && !split) {  // the pre-processed stream of the block (what streams bound with bindIQStream receive); groups never form behind a pre-processing chain
        const size_t bytes = (size_t)c->pre.last_n * 8;
        R[0]->off_iq = off;
        R[0]->n_iq = c->pre.last_n;
        copies.push_back(ResCopy{ c->pre.last, off, bytes, c->plan_lvl0 + 1 });
        off += (bytes + 15) & ~(size_t)15;
    }
    const int n_lines = c->fft_on ? c->n_lines : 0;
    int64_t lo_lines[kGroupMax + 1] = { 0 };
    for (int j = 0; j < k; j++) { lo_lines[j + 1] = (split && n_lines > 0) ? std::min<int64_t>(n_lines, tick_frames_by(c, c->grp_ends[(size_t)j])) : n_lines; }
    if (split && lo_lines[k] != n_lines) { return fail(c, SDRPP_ERR_INVALID, "internal: lines of the group do not add up (%lld of %d)", (long long)lo_lines[k], n_lines); }
    for (int j = 0; j < k; j++) { R[j]->n_lines = (int)(lo_lines[j + 1] - lo_lines[j]); }
    if (n_lines > 0) {
        const int lines_level = c->plan_lvl0 + (c->fft_lg <= 12 ? 1 : (c->fft_lg <= 16 ? 2 : 3));
        if ((c->res_flags & 2) && c->data_width > 0) {
            const size_t row = (size_t)c->data_width * 4, bytes = (size_t)n_lines * row;
            for (int j = 0; j < k; j++) { R[j]->off_zoomed = off + (size_t)lo_lines[j] * row; }
            copies.push_back(ResCopy{ c->d_zoomed, off, bytes, lines_level + 2 });
            off += (bytes + 15) & ~(size_t)15;
            for (int j = 0; j < k; j++) { R[j]->off_index = off + (size_t)lo_lines[j] * row; }
            copies.push_back(ResCopy{ c->d_index, off, bytes, lines_level + 2 });
            off += (bytes + 15) & ~(size_t)15;
        }
        if (c->res_flags & 4) {
            const size_t row = (size_t)c->fft_size * 4, bytes = (size_t)n_lines * row;
            for (int j = 0; j < k; j++) { R[j]->off_raw = off + (size_t)lo_lines[j] * row; }
            copies.push_back(ResCopy{ c->d_lines, off, bytes, lines_level + 1 });
            off += (bytes + 15) & ~(size_t)15;
        }
    }
    if (off > c->res_cap) { return fail(c, SDRPP_ERR_INVALID, "internal: results of %zu bytes exceed what a launch may deliver (%zu)", off, c->res_cap); }
    int rc = tick_results_alloc(c, c->groups, off, region_off);
    if (rc) {
        for (int j = 0; j < k; j++) { R[j]->ticket = 0; }
        return rc;
    }
    for (int j = 0; j < k; j++) {
        R[j]->base = c->res_ring + *region_off;
        R[j]->epoch = c->res_epoch;
    }
    return SDRPP_OK;
}
// gather roles of the block just planned -> c->emits (one level behind the producers); fills the result entries of its pushes
int tick_results_plan(sdrpp_ctx* c, uint64_t first_ticket, int k) {
    static thread_local std::vector<ResCopy> copies;
    size_t region = 0;
    int rc = tick_results_describe(c, first_ticket, k, copies, &region);
    if (rc || copies.empty()) { return rc; }
    Lev<CopyJob> jobs;
    char* base = c->res_ring_dev + region;
    for (auto& q : copies) { jobs.add(q.level, CopyJob{ q.src, base + q.off, (long long)q.bytes, 0x100, 0 }); }
    if (!arena_push_lev(c, jobs)) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
    for (int l = 0; l < jobs.top; l++) {
        if (jobs.at[l].empty()) { continue; }
        long long mx = 0;
        for (auto& j : jobs.at[l]) { mx = std::max(mx, j.bytes); }
        emit(c, l, F_MISC, TR_COPY, (int)std::max<long long>(1, std::min<long long>((mx + 32767) / 32768, 16)), (int)jobs.at[l].size(), 0, jobs.dev[l]);
        if (!c->emits.empty()) { c->emits.back().to_host = true; }
        c->plan_top = std::max(c->plan_top, l + 1);
    }
    return SDRPP_OK;
}

// results of a block that ran as an ORDINARY pass inside a pipelined run (something the device cannot pipeline: a pre-processing chain, a
// VFO group without the matrix front end, a retune hand-over ...): the same slot layout, filled by plain copies behind the pass and waited
// for here — the slow path, but sdrpp_result_wait / _release then work for EVERY block of a pipelined run, whichever way it was processed
int tick_results_direct(sdrpp_ctx* c, uint64_t first_ticket, int k) {
    int rc = tick_results_ensure(c);
    if (rc) { return rc; }
    static thread_local std::vector<ResCopy> copies;
    size_t region = 0;
    rc = tick_results_describe(c, first_ticket, k, copies, &region);
    if (rc) { return rc; }
    char* base = c->res_ring + region;
    for (auto& q : copies) { HIPCHK(c, hipMemcpyAsync(base + q.off, q.src, q.bytes, hipMemcpyDeviceToHost, c->stream)); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int j = 0; j < k; j++) {
        sdrpp_ctx::Result& R = c->res[(first_ticket + (uint64_t)j) % kResMeta];
        if (R.ticket == first_ticket + (uint64_t)j) { R.done_tick = c->ticks; }  // nothing queued is left: complete as it stands
    }
    return SDRPP_OK;
}

// sdrpp_push_staged_when: the host is still filling the staging slot with other threads while this thread plans the block; nothing that
// reads the slot may be launched before they are through (the word counts their unfinished parts).
int stage_pending_wait(sdrpp_ctx* c) {
    if (c->stage_pend.empty()) { return SDRPP_OK; }
    std::vector<const volatile uint32_t*> words;
    words.swap(c->stage_pend);
    HostScope hs("staging wait");
    const auto t0 = std::chrono::steady_clock::now();
    for (const volatile uint32_t* w : words) {
        for (unsigned spins = 0; *w != 0u; spins++) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((spins & 0xfffu) == 0xfffu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_staged_when: the staging slot was not completed within 5 s"); }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return SDRPP_OK;
}

// One block in pipelined mode — one push, or a launch group of k pushes (`ends`: their cumulative ends, `count` samples in all; tickets
// first_ticket .. first_ticket + k - 1, already counted in c->pushes: the caller takes them back if this fails).  `d_iq`: where the samples are
// (caller's device buffer) or will be once `land` has run (landing ring).
int tick_push(sdrpp_ctx* c, const float* d_iq, int64_t count, const CopyJob* land, const std::vector<int>* ends, uint64_t first_ticket) {
    if (count == 0) { return SDRPP_OK; }
    const int k = (ends && ends->size() > 1) ? (int)ends->size() : 1;
    struct GroupScope {  // the push ends are visible to the planners (plan_vfo.h: chain) for the duration of this block only
        sdrpp_ctx* c;
        GroupScope(sdrpp_ctx* c_, const std::vector<int>* e, int k_) : c(c_) { if (k_ > 1) { c->grp_ends = *e; } else { c->grp_ends.clear(); } }
        ~GroupScope() { c->grp_ends.clear(); }
    } group_scope(c, ends, k);
    const float* const d_iq_raw = d_iq;  // (planning a pre-processing chain moves d_iq / count on to the pre-processed stream)
    const int64_t count_raw = count;
    c->plan_block_from_host = land != nullptr && land->bytes > 0;
    bool as_tick = tick_eligible(c);
    if (as_tick) {  // rings of the per-block buffers, result slots (allocated on first use / after a change of the configuration)
        for (auto& kv : c->vfos) {
            for (auto& st : kv.second->st) {
                if (st.n_extra < kRing - 1 && st.base) {
                    int rc = stream_ring_ensure(c, st);
                    if (rc) { return rc; }
                }
            }
        }
        if (c->pre.on) {  // the pre-processing chain's stage outputs are per-block buffers too
            for (auto& st : c->pre.st) {
                if (st.n_extra < kRing - 1 && st.base) {
                    int rc = stream_ring_ensure(c, st);
                    if (rc) { return rc; }
                }
            }
            if (c->pre.out.base && c->pre.out.n_extra < kRing - 1) {
                int rc = stream_ring_ensure(c, c->pre.out);
                if (rc) { return rc; }
            }
        }
        int rc = fft_ring_ensure(c);
        if (!rc && c->res_flags) { rc = tick_results_ensure(c); }
        if (rc) { return rc; }
    }
    c->groups++;
    const bool have_slot = as_tick;
    static thread_local PlanSnapshot snap;  // (re-used: with 128 VFOs a fresh one is a 38 KB allocation per block; a nested pass starts only after the outer plan has been restored)
    int rc = SDRPP_OK;
    if (as_tick) {
        HostScope hs("tick plan");
        rc = arena_begin(c);
        if (rc) {
            c->groups--;
            return rc;
        }
        {
            HostScope hs2("tick: snapshot + rotate");
            plan_snapshot(c, snap, true);  // (and every VFO stream on to its next ring buffer, in the same walk)
        }
        fft_ring_rotate(c);
        if (c->pre.on) {
            for (auto& st : c->pre.st) { stream_rotate(st); }
            stream_rotate(c->pre.out);
        }
        c->plan_sparse = 1;
        if (c->tick_sparse_boost && c->h_tick_flag) {
            const int in_flight = (int)((unsigned)c->ticks - *(const volatile unsigned*)c->h_tick_flag);
            c->plan_sparse = in_flight <= 0 ? 4 : (in_flight == 1 ? 2 : 1);
        }
        c->tick_planning = true;
        c->tick_abort = false;
        c->emits.clear();
        c->plan_top = 2;
        c->plan_lvl0 = 0;
        block_bounds(c, count, k > 1 ? ends : nullptr);
        if (c->pre.on) {  // levels 1 .. plan_lvl0 of the block: from here on `d_iq` / `count` are the pre-processed stream
            rc = run_preproc(c, &d_iq, &count);
            if (!rc && count == 0) { c->tick_abort = true; }  // (the decimator swallowed the whole block: an ordinary pass sorts that out)
        }
        if (!rc && !c->tick_abort) { rc = ensure_iq_hist(c, iq_hist_need(c)); }
        if (!rc && !c->tick_abort) {
            IqSrc src{ (const float2*)d_iq, (const float2*)c->iq_hist[c->iq_cur], c->iq_hist_cap, (long long)count };
            rc = do_fft(c, src, count);
            if (!rc && !c->tick_abort) {
                const CarryJob iqc = iq_carry_job(c, d_iq, count);
                if (c->vfos.empty()) {
                    std::vector<CarryJob> carry{ iqc };
                    CarryJob* d_carry = arena_push(c, carry);
                    if (!d_carry) { rc = fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
                    else { emit(c, 1 + c->plan_lvl0, F_MISC, TR_CARRY, std::max(1, std::min((iqc.need * 2 + 8191) / 8192, 512)), 1, 0, d_carry); }
                }
                else { rc = do_vfos_plan(c, src, count, iqc); }
            }
            if (!rc && !c->tick_abort) {
                HostScope hs2("tick: results plan");
                rc = tick_results_plan(c, first_ticket, k);
            }
        }
        c->tick_planning = false;
        // the block's depth is where its last ROLE stands, not where the planners reserved room (a bank whose outputs stay on the device has
        // nothing behind its last filter: one tick less until the block is complete, one tick less to drain)
        if (!rc && !c->tick_abort) {
            int top = 2;
            for (auto& r : c->emits) { top = std::max(top, r.level + 1); }
            c->plan_top = std::min(c->plan_top, top);
        }
        if (!rc && !c->tick_abort && c->plan_top > kTickDepth + 1) { c->tick_abort = true; }
        if (rc || c->tick_abort) {
            plan_restore(c, snap);
            c->emits.clear();
            for (int j = 0; j < k; j++) {
                sdrpp_ctx::Result& R = c->res[(first_ticket + (uint64_t)j) % kResMeta];
                if (!R.held) { R.ticket = 0; }
            }
            as_tick = false;
            if (rc) {
                c->groups--;
                return rc;
            }
            c->arena_off = 0;  // (the slot stays this tick's: only the next role table goes in)
        }
    }
    if (!as_tick) {
        c->stat_pass_blocks += k;
        // this block runs as an ordinary pass: everything queued first (the first of those ticks carries the landing copy), then the pass
        // behind them on the same stream
        if (!have_slot) {
            rc = arena_begin(c);
            if (rc) {
                c->groups--;
                return rc;
            }
        }
        rc = stage_pending_wait(c);
        if (!rc) { rc = tick_launch(c, land); }
        if (!rc) { rc = tick_drain(c); }
        if (land) { c->land_tick = c->ticks; }
        if (!rc) { rc = push_common(c, d_iq_raw, count_raw, k > 1 ? ends : nullptr); }
        // its results are where an ordinary pass leaves them (device buffers, readable after a synchronisation) and — with result flags —
        // also in the block's result slot like every other block's
        if (!rc && c->res_flags) { rc = tick_results_direct(c, first_ticket, k); }
        else {
            for (int j = 0; j < k; j++) {
                sdrpp_ctx::Result& R = c->res[(first_ticket + (uint64_t)j) % kResMeta];
                if (!R.held) { R = sdrpp_ctx::Result{}; }
            }
        }
        if (rc) { c->groups--; }
        return rc;
    }
    // queue the roles level by level and launch this block's tick
    HostScope hs3("tick: queue + launch");
    c->stat_tick_blocks += k;
    c->stat_last_depth = c->plan_top;
    c->stat_last_table_bytes = (int64_t)c->arena_off;  // job tables of this block (the next tick's role table is added at the launch)
    if ((int)c->tickq.size() < c->plan_top) { c->tickq.resize((size_t)c->plan_top); }
    for (auto& r : c->emits) { c->tickq[(size_t)r.level].push_back(r); }
    c->emits.clear();
    c->iq_cur ^= 1;
    // tickq[0] is this very tick: it holds only what earlier blocks queued (a block's own roles start at level 1)
    rc = stage_pending_wait(c);
    if (!rc) {
        HostScope hs("tick launch");
        rc = tick_launch(c, land);
    }
    if (land) { c->land_tick = c->ticks; }
    for (int j = 0; j < k; j++) {
        sdrpp_ctx::Result& R = c->res[(first_ticket + (uint64_t)j) % kResMeta];
        if (R.ticket == first_ticket + (uint64_t)j) { R.done_tick = c->ticks + (uint64_t)(c->plan_top - 1); }
    }
    return rc;
}

// ---- several blocks per launch (sdrpp_set_pipeline_group) ---------------------------------------------------------------------------------
// May pushes share a launch at all right now?  Not behind a pre-processing chain (its decimator changes the rate the push ends live at), not while a
// block cannot run as a tick, and not with a VFO that only rotates (no decimation plan: its NCO is anchored at the start of a launch, the results
// would differ from block-by-block processing in the last bit).
bool group_eligible(sdrpp_ctx* c) {
    if (c->group_max <= 1 || c->pre.on || !tick_eligible(c)) { return false; }
    for (Vfo* v : c->vfo_list) {
        if (!v->nco_exact && v->d.n_stages == 0) { return false; }
    }
    return true;
}
// Whatever is held goes out as ONE block of the tick queue (nothing held: nothing happens).  On failure the held pushes did not happen: their
// tickets are taken back.
int tick_group_launch(sdrpp_ctx* c) {
    if (c->held.kind < 0 || c->held.ends.empty()) { return SDRPP_OK; }  // (a staging slot handed out by sdrpp_push_stage and not yet pushed stays open)
    sdrpp_ctx::Held H;
    std::swap(H, c->held);
    const int k = (int)H.ends.size();
    const uint64_t first_ticket = c->pushes - (uint64_t)k + 1;
    int rc = SDRPP_OK;
    if (H.kind == 0) { rc = tick_push(c, reinterpret_cast<const float*>(H.base), H.total, nullptr, &H.ends, first_ticket); }  // read in place, one tick from now at the earliest
    else {
        const int li = (int)((c->groups + 1) % 3);
        if (!c->tick_land[li]) { rc = dev_alloc(c, &c->tick_land[li], (size_t)c->max_push * 2 + 32); }
        void* d = nullptr;
        if (!rc && (hipHostGetDevicePointer(&d, H.kind == 3 ? (void*)H.base : (void*)c->stage_host[H.stage_slot], 0) != hipSuccess || !d)) { rc = fail(c, SDRPP_ERR_HIP, "hipHostGetDevicePointer(staging) failed"); }
        if (!rc) {
            const CopyJob land{ d, c->tick_land[li], (long long)((size_t)H.total * (H.kind == 2 ? 4 : 8)), H.kind == 2 ? 1 : 0, 0 };
            rc = tick_push(c, c->tick_land[li], H.total, &land, &H.ends, first_ticket);
            if (H.kind != 3) { c->stage_tick[H.stage_slot] = c->ticks; }
        }
    }
    c->stage_pend.clear();
    if (rc) { c->pushes -= (uint64_t)k; }
    else if (k > 1) {
        c->stat_groups++;
        c->stat_group_blocks += k;
        c->stat_group_max = std::max<int64_t>(c->stat_group_max, k);
    }
    return rc;
}
// One push of a pipelined run: joins the held group, or — the group full, the push of another kind or not contiguous with it, grouping off — sends
// what is held on its way first.  kind 0: `p` = device address (read in place); 1 / 2: float / int16 samples in host memory, copied into the
// group's page-locked staging slot (p == nullptr: the host has filled the slot itself, sdrpp_push_stage); 3: the caller's page-locked memory.
int tick_hold(sdrpp_ctx* c, int kind, const void* p, int64_t count, const volatile uint32_t* pending) {
    if (count == 0) { return SDRPP_OK; }
    sdrpp_ctx::Held& H = c->held;
    const bool groupable = group_eligible(c);
    const size_t bps = kind == 2 ? 4 : 8;
    if (H.kind >= 0) {
        bool fits = groupable && H.kind == kind && (int)H.ends.size() < c->group_max && H.total + count <= c->max_push;
        if (fits && (kind == 0 || kind == 3)) { fits = reinterpret_cast<const char*>(p) == H.base + (size_t)8 * (size_t)H.total; }
        if (H.ends.empty() && H.kind == kind && (kind == 1 || kind == 2) && H.total == 0) { fits = true; }  // (a slot opened by sdrpp_push_stage, nothing in it yet)
        if (!fits) {
            if ((kind == 1 || kind == 2) && !p) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_staged: %lld samples do not fit the open staging slot", (long long)count); }
            int rc = tick_group_launch(c);
            if (rc) { return rc; }
            c->held = sdrpp_ctx::Held{};
        }
    }
    if (H.kind < 0) {
        if ((kind == 1 || kind == 2) && !p) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_staged without an open staging slot (sdrpp_push_stage)"); }
        H.kind = kind;
        H.base = reinterpret_cast<const char*>(p);
        H.total = 0;
        H.ends.clear();
        if (kind == 1 || kind == 2) {
            const int si = c->stage_cur;
            c->stage_cur = (c->stage_cur + 1) % kStageSlots;
            if (!c->stage_host[si]) {
                if (hipHostMalloc((void**)&c->stage_host[si], (size_t)c->max_push * 8 + 64, hipHostMallocMapped) != hipSuccess) {
                    H.kind = -1;
                    return fail(c, SDRPP_ERR_NOMEM, "page-locked staging buffer");
                }
            }
            if (c->stage_tick[si]) { tick_wait_done(c, c->stage_tick[si]); }  // its last landing copy has run
            H.stage_slot = si;
        }
    }
    if ((kind == 1 || kind == 2) && p) { memcpy(reinterpret_cast<char*>(c->stage_host[H.stage_slot]) + (size_t)H.total * bps, p, (size_t)count * bps); }
    H.total += count;
    H.ends.push_back((int)H.total);
    c->pushes++;
    bool go = !groupable || (int)H.ends.size() >= c->group_max;
    if (!go && c->group_adaptive && c->h_tick_flag) { go = (int)((unsigned)c->ticks - *(const volatile unsigned*)c->h_tick_flag) < 2; }  // the device would run dry: do not wait for more
    if (pending) {
        // sdrpp_push_staged_when: the word belongs to the caller and is only promised to live for the call.  The push that sends the group on its
        // way is planned while its copy threads are still at work (the wait comes just before the launch, stage_pending_wait); one that is merely
        // held has nothing to overlap with: it waits here — unless the caller has promised (sdrpp_set_pipeline_group flag 2) that its words live until
        // the launch: then the hold returns at once and the copy runs on under whatever the caller does next; the launch of the group waits for all of them.
        c->stage_pend.push_back(pending);
        if (!go && !c->stage_pend_stable) {
            int rc = stage_pending_wait(c);
            if (rc) {  // the block did not arrive: it is not part of the group
                H.ends.pop_back();
                H.total -= count;
                c->pushes--;
                return rc;
            }
        }
    }
    return go ? tick_group_launch(c) : SDRPP_OK;
}

}  // namespace
