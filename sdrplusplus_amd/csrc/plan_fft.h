// FFT branch of one block: pass launches / roles, doZoom + palette, the per-block buffer rings of pipelined mode, waterfall display state.
// Part of the one translation unit sdrpp_gpu.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

// ---- FFT launches ---------------------------------------------------------------------------------------------------------------
// Workgroups of a pass whose workgroups WALK their tiles (fft_pass1_body / fft_pass2_body: tile, tile + grid, ... with the next tile's loads
// in flight during the current tile's arithmetic): about one resident round, a multiple of the tiles per frame (so that a pass-1 workgroup
// stays on its columns and keeps their window values), never more than there are tiles.
inline int fft_walk_grid(int ntiles, int per_frame, int cap) {
    if (cap <= 0 || ntiles <= cap) { return ntiles; }
    return std::max(per_frame, (cap / per_frame) * per_frame);
}
template <int LG, int FPW>
void launch_single(sdrpp_ctx* c, const IqSrc& src, const FrameGeom& g, float* out) {
    const int blocks = (g.nframes + FPW - 1) / FPW;
    launch(c, fft_single_kernel<LG, FPW>, dim3(blocks), dim3(((1 << LG) / 16) * FPW), 0, src, g, (const float*)c->d_window, (const float2*)c->d_tw1, out);
}
template <int LG1, int C>
void launch_p1(sdrpp_ctx* c, const IqSrc& src, const FrameGeom& g, int lg2) {
    const int per_frame = (1 << lg2) / C, ntiles = g.nframes * per_frame;
    launch(c, fft_pass1_kernel<LG1, C>, dim3((unsigned)fft_walk_grid(ntiles, per_frame, c->fft_p1_grid)), dim3(((1 << LG1) / 16) * C), 0, src, g, (const float*)c->d_window, (const float2*)c->d_tw1,
           (const float2*)c->d_twn, c->d_scratch, lg2, ntiles);
}
// four-step split of a 2^m-point transform (m > 12): N1 = 2^lg1 column transforms, N2 = 2^lg2 row transforms.  Even up to 65536 points;
// above, the rows take the 4096 points one workgroup holds and the columns the rest (fft_kernels.h) — the oracle splits the same way
inline void fft_split(int m, int* lg1, int* lg2) {
    *lg1 = m <= 16 ? m / 2 : m - 12;
    *lg2 = m - *lg1;
}
constexpr int kZoomGrpLong = 16;  // bins per doZoom group of the long transforms (left by the transpose pass)
constexpr int pass2_rows(int lg2) { return lg2 == 7 ? 32 : (lg2 == 8 ? 16 : (lg2 == 9 ? 8 : 4)); }  // rows per workgroup of fft_pass2_kernel = bins per doZoom group
template <int LG2, int R>
void launch_p2(sdrpp_ctx* c, int nframes, int lg1, float* out, float* grp) {
    static_assert(R == pass2_rows(LG2), "pass2_rows out of step with the launch table");
    const int per_frame = (1 << lg1) / R, ntiles = nframes * per_frame;
    launch(c, fft_pass2_kernel<LG2, R>, dim3((unsigned)fft_walk_grid(ntiles, per_frame, c->fft_p2_grid)), dim3(((1 << LG2) / 16) * R), 0, (const float2*)c->d_scratch, (const float2*)c->d_tw2, out, lg1, ntiles, grp);
}

void launch_p2row(sdrpp_ctx* c, int nframes, int lg1) {
    launch(c, fft_pass2row_kernel<12>, dim3((unsigned)(nframes << lg1)), dim3(256), 0, c->d_scratch, (const float2*)c->d_tw2, lg1);
}
void launch_transpose(sdrpp_ctx* c, int nframes, int lg1, int lg2, float* out, float* grp) {
    const int tiles = (1 << lg2) / (SDRPP_FFT_TR_TILE >> lg1);
    launch(c, fft_transpose_kernel, dim3((unsigned)(nframes * tiles)), dim3(256), 0, (const float*)c->d_scratch, out, grp, lg1, lg2, kZoomGrpLong);
}

// pipelined mode: the FFT branch of one block as roles of the tick kernel — pass 1 (or the whole small transform) at level 1 next to the
// front end, pass 2 at level 2, doZoom + palette index behind the lines (256-thread shapes of the same bodies: bit-identical)
int plan_fft_roles(sdrpp_ctx* c, const IqSrc& src, const FrameGeom& g, float* out, float* grp) {
    const int m = c->fft_lg;
    sdrpp_ctx::RoleLaunch r{};
    r.e.gy = 1;
    r.fam = F_FFTS;
    if (m <= 12) {
        const int fpw = m == 10 ? 4 : (m == 11 ? 2 : 1);
        r.e.role = m == 10 ? TR_FFT_S10 : (m == 11 ? TR_FFT_S11 : TR_FFT_S12);
        r.e.gx = (g.nframes + fpw - 1) / fpw;
        r.e.p.fs = TickFS{ src, g, c->d_window, c->d_tw1, out };
        r.lds = tick_lds_fft_single(m, fpw);
        r.level = 1 + c->plan_lvl0;
        c->emits.push_back(r);
        return SDRPP_OK;
    }
    int lg1, lg2;
    fft_split(m, &lg1, &lg2);
    static const int p1_role[6] = { TR_FFT_P1_5, TR_FFT_P1_6, TR_FFT_P1_7, TR_FFT_P1_8, TR_FFT_P1_9, TR_FFT_P1_10 }, p1_c[6] = { 128, 64, 32, 16, 8, 4 };
    static const int p2_role[4] = { TR_FFT_P2_7, TR_FFT_P2_8, TR_FFT_P2_9, TR_FFT_P2_10 };
    if (lg1 < 5 || lg1 > 10 || lg2 < 7 || lg2 > 12 || (lg2 > 10 && lg2 != 12)) { return fail(c, SDRPP_ERR_UNSUPPORTED, "fft size 2^%d unsupported", m); }
    r.e.role = p1_role[lg1 - 5];
    {
        const int per_frame = (1 << lg2) / p1_c[lg1 - 5], ntiles = g.nframes * per_frame;
        r.e.gx = fft_walk_grid(ntiles, per_frame, c->fft_tick_grid);
        r.e.p.p1 = TickP1{ src, g, c->d_window, c->d_tw1, c->d_twn, c->d_scratch, lg2, ntiles };
    }
    r.lds = tick_lds_fft_p1(lg1, p1_c[lg1 - 5]);
    r.level = 1 + c->plan_lvl0;
    r.fam = F_FFT1;
    c->emits.push_back(r);
    if (lg2 == 12) {  // long transforms: 4096-point rows (dB in place), then the transpose into bin order one level later
        sdrpp_ctx::RoleLaunch q{};
        q.e.gy = 1;
        q.e.role = TR_FFT_P2ROW;
        q.e.gx = g.nframes << lg1;
        q.e.p.p2 = TickP2{ c->d_scratch, c->d_tw2, nullptr, nullptr, lg1, g.nframes };
        q.lds = tick_lds_fft_single(12, 1);
        q.level = 2 + c->plan_lvl0;
        q.fam = F_FFT2;
        c->emits.push_back(q);
        sdrpp_ctx::RoleLaunch t{};
        t.e.gy = 1;
        t.e.role = TR_FFT_TR;
        t.e.gx = g.nframes * ((1 << lg2) / (SDRPP_FFT_TR_TILE >> lg1));
        t.e.aux = kZoomGrpLong;
        t.e.p.p2 = TickP2{ c->d_scratch, nullptr, out, grp, lg1, g.nframes };
        t.lds = (size_t)(SDRPP_FFT_TR_TILE + 256) * sizeof(float);
        t.level = 3 + c->plan_lvl0;
        t.fam = F_FFT2;
        c->emits.push_back(t);
        return SDRPP_OK;
    }
    sdrpp_ctx::RoleLaunch q{};
    q.e.gy = 1;
    q.e.role = p2_role[lg2 - 7];
    {
        const int per_frame = (1 << lg1) / pass2_rows(lg2), ntiles = g.nframes * per_frame;
        q.e.gx = fft_walk_grid(ntiles, per_frame, c->fft_tick_grid);
        q.e.p.p2 = TickP2{ c->d_scratch, c->d_tw2, out, grp, lg1, ntiles };
    }
    q.lds = tick_lds_fft_p2(lg2, pass2_rows(lg2));
    q.level = 2 + c->plan_lvl0;
    q.fam = F_FFT2;
    c->emits.push_back(q);
    return SDRPP_OK;
}

int run_fft_chunk(sdrpp_ctx* c, const IqSrc& src, const FrameGeom& g, float* out, float* grp) {
    const int m = c->fft_lg;
    if (m <= 12) {
        FamilyTimer t(c, F_FFTS);
        switch (m) {
        case 10: launch_single<10, 4>(c, src, g, out); break;
        case 11: launch_single<11, 2>(c, src, g, out); break;
        case 12: launch_single<12, 1>(c, src, g, out); break;
        default: return fail(c, SDRPP_ERR_UNSUPPORTED, "fft size 2^%d unsupported", m);
        }
        return SDRPP_OK;
    }
    int lg1, lg2;
    fft_split(m, &lg1, &lg2);
    {
        FamilyTimer t(c, F_FFT1);
        switch (lg1) {
        case 5: launch_p1<5, 128>(c, src, g, lg2); break;
        case 6: launch_p1<6, 64>(c, src, g, lg2); break;
        case 7: launch_p1<7, 32>(c, src, g, lg2); break;
        case 8:
            // 16 columns (128-byte row segments, 256 work-items, 34 KB of LDS: four workgroups per CU) against 32 (256-byte segments, 512
            // work-items, 66 KB: two): 0.064-0.069 ms against 0.098-0.106 per 2^24 samples (round 2 had measured the wider one 7 % ahead: before
            // the window values left the load path)
            if (c->fft_p1_c32) { launch_p1<8, 32>(c, src, g, lg2); }
            else { launch_p1<8, 16>(c, src, g, lg2); }
            break;
        case 9: launch_p1<9, 8>(c, src, g, lg2); break;
        case 10: launch_p1<10, 4>(c, src, g, lg2); break;
        default: return fail(c, SDRPP_ERR_UNSUPPORTED, "fft pass-1 size 2^%d unsupported", lg1);
        }
    }
    {
        FamilyTimer t(c, F_FFT2);
        switch (lg2) {
        case 12:
            launch_p2row(c, g.nframes, lg1);
            launch_transpose(c, g.nframes, lg1, lg2, out, grp);
            break;
        case 7: launch_p2<7, 32>(c, g.nframes, lg1, out, grp); break;
        case 8: launch_p2<8, 16>(c, g.nframes, lg1, out, grp); break;
        case 9: launch_p2<9, 8>(c, g.nframes, lg1, out, grp); break;
        case 10: launch_p2<10, 4>(c, g.nframes, lg1, out, grp); break;
        default: return fail(c, SDRPP_ERR_UNSUPPORTED, "fft pass-2 size 2^%d unsupported", lg2);
        }
    }
    return SDRPP_OK;
}

// Lanes per pixel of the zoom kernel (16 / 4 / 1): from the LARGEST number of elements a pixel of the current view really walks — bins, or
// with pass 2's group maxima ragged head + whole groups + ragged tail (zoom_palette_body).  An aligned full-span view of a 65536-point
// line at 1024 pixels walks 4 group maxima per pixel: one lane per pixel, 4 workgroups per line instead of 64 and no LDS exchange (the
// estimate bins / group + group — the worst ragged case — chose 16 lanes, 12 of them idle).  max is order independent: same bits.
int zoom_lanes(sdrpp_ctx* c, bool with_grp) {
    const int gsz = with_grp ? c->zoom_grp : 0;
    if (c->zoom_tp_grp == gsz && c->zoom_tp_cache) { return c->zoom_tp_cache; }
    int worst = 1;
    for (size_t i = 0; i < c->h_zstart.size(); i++) {
        const int s = c->h_zstart[i], n = c->h_zcount[i], e = s + n;
        int el = n;
        if (gsz > 1 && n >= 2 * gsz) {
            const int a = ((s + gsz - 1) / gsz) * gsz, bnd = (e / gsz) * gsz;
            el = (a - s) + (bnd - a) / gsz + (e - bnd);
        }
        worst = std::max(worst, el);
    }
    c->zoom_tp_cache = worst >= 16 ? 16 : (worst > 4 ? 4 : 1);
    c->zoom_tp_grp = gsz;
    return c->zoom_tp_cache;
}
// may a view use the group maxima at all?  (zoom_palette_body takes them for pixels of >= 2 groups; narrower views read the bins)
inline bool zoom_uses_grp(const sdrpp_ctx* c, const float* grp, int view_bins, int data_width, int gsz) { return grp && gsz > 1 && view_bins / std::max(1, data_width) >= 2 * gsz; }

// doZoom + palette launch: lanes per pixel from the view's bins per pixel (coalesced bin reads for wide pixels, no idle lanes for narrow ones)
void launch_zoom(hipStream_t stream, const float* lines, int nlines, int fft_size, int view_bins, int data_width, const int32_t* zs, const int32_t* zc, float wf_min, float wf_max,
                 float* zoomed, int32_t* index, const float* grp = nullptr, int gsz = 0, int tp_exact = 0) {
    int bpp = view_bins / std::max(1, data_width);
    if (grp && gsz > 1 && bpp >= 2 * gsz) { bpp = bpp / gsz + gsz; }  // elements a pixel walks: whole groups + the ragged ends
    else { grp = nullptr; }
    const int tp = tp_exact ? tp_exact : ((bpp >= 16) ? 16 : ((bpp >= 4) ? 4 : 1));  // (tp_exact: zoom_lanes of the context's own view)
    const dim3 grid((unsigned)((data_width + 256 / tp - 1) / (256 / tp)), (unsigned)nlines);
    switch (tp) {
    case 16: hipLaunchKernelGGL(zoom_palette_kernel<16>, grid, dim3(256), 0, stream, lines, fft_size, data_width, zs, zc, wf_min, wf_max, zoomed, index, grp, gsz); break;
    case 4: hipLaunchKernelGGL(zoom_palette_kernel<4>, grid, dim3(256), 0, stream, lines, fft_size, data_width, zs, zc, wf_min, wf_max, zoomed, index, grp, gsz); break;
    default: hipLaunchKernelGGL(zoom_palette_kernel<1>, grid, dim3(256), 0, stream, lines, fft_size, data_width, zs, zc, wf_min, wf_max, zoomed, index, grp, gsz); break;
    }
}

// pipelined mode: kRing sets of the per-block FFT buffers.  Every call that re-sizes one of them drops the extra sets first.
void fft_ring_drop(sdrpp_ctx* c) {
    for (int i = 0; i < kRing - 1; i++) {
        sdrpp_ctx::FftBufs& b = c->fft_extra[i];
        dev_free(b.scratch);
        dev_free(b.lines);
        dev_free(b.grp);
        dev_free(b.zoomed);
        dev_free(b.index);
    }
    c->fft_extra_n = 0;
    c->fft_rot = 0;
}
int fft_ring_ensure(sdrpp_ctx* c) {
    if (!c->fft_on || c->fft_extra_n == kRing - 1) { return SDRPP_OK; }
    fft_ring_drop(c);
    const size_t per_chunk = std::max<size_t>(1, kScratchBytes / ((size_t)c->fft_size * sizeof(float2)));
    for (int i = 0; i < kRing - 1; i++) {
        sdrpp_ctx::FftBufs& b = c->fft_extra[i];
        int rc = SDRPP_OK;
        if (c->d_scratch) { rc = dev_alloc(c, &b.scratch, per_chunk * (size_t)c->fft_size); }
        if (!rc) { rc = dev_alloc(c, &b.lines, c->lines_cap * (size_t)c->fft_size); }
        if (!rc && c->zoom_grp) { rc = dev_alloc(c, &b.grp, c->lines_cap * (size_t)(c->fft_size / c->zoom_grp)); }
        if (!rc && c->zoom_cap) { rc = dev_alloc(c, &b.zoomed, c->zoom_cap); }
        if (!rc && c->zoom_cap) { rc = dev_alloc(c, &b.index, c->zoom_cap); }
        if (rc) {
            fft_ring_drop(c);
            return rc;
        }
        c->fft_extra_n = i + 1;
    }
    return SDRPP_OK;
}
void fft_ring_rotate(sdrpp_ctx* c) {
    if (c->fft_extra_n == 0) { return; }
    sdrpp_ctx::FftBufs& b = c->fft_extra[c->fft_rot];
    std::swap(c->d_scratch, b.scratch);
    std::swap(c->d_lines, b.lines);
    std::swap(c->d_lines_grp, b.grp);
    std::swap(c->d_zoomed, b.zoomed);
    std::swap(c->d_index, b.index);
    c->fft_rot = (c->fft_rot + 1) % c->fft_extra_n;
}

int ensure_zoom(sdrpp_ctx* c, size_t lines) {
    if (c->data_width <= 0) { return SDRPP_OK; }
    const size_t need = lines * (size_t)c->data_width;
    if (need <= c->zoom_cap) { return SDRPP_OK; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fft_ring_drop(c);
    dev_free(c->d_zoomed);
    dev_free(c->d_index);
    int rc = dev_alloc(c, &c->d_zoomed, need);
    if (rc) { return rc; }
    rc = dev_alloc(c, &c->d_index, need);
    if (rc) { return rc; }
    c->zoom_cap = need;
    return SDRPP_OK;
}

// latestFFT / smoothing / hold arrays follow the view's data width (WaterFall::onResize reallocates them)
int wf_ensure_trace(sdrpp_ctx* c) {
    sdrpp_ctx::Wf& W = c->wf;
    if (W.width == c->data_width && W.d_latest) { return SDRPP_OK; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const bool had_smooth = W.d_smooth != nullptr;
    dev_free(W.d_latest);
    dev_free(W.d_smooth);
    dev_free(W.d_hold);
    W.width = c->data_width;
    W.have_latest = false;
    if (W.width <= 0) { return SDRPP_OK; }
    int rc = dev_alloc(c, &W.d_latest, (size_t)W.width);
    if (rc) { return rc; }
    HIPCHK(c, hipMemset(W.d_latest, 0, (size_t)W.width * sizeof(float)));
    rc = dev_alloc(c, &W.d_hold, (size_t)W.width);
    if (rc) { return rc; }
    std::vector<float> init((size_t)W.width, -1000.0f);  // setFFTHold, waterfall.cpp:1153-1160
    HIPCHK(c, hipMemcpy(W.d_hold, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice));
    if (had_smooth) {
        rc = dev_alloc(c, &W.d_smooth, (size_t)W.width);
        if (rc) { return rc; }
        HIPCHK(c, hipMemset(W.d_smooth, 0, (size_t)W.width * sizeof(float)));
    }
    return SDRPP_OK;
}

// WaterFall display state behind the lines of one block (waterfall.cpp:875-941): the raw lines into the ring at `level`, the FFT trace's
// smoothing / hold over the block's zoomed lines one level behind the zoom (`level` + 1) — launches in an ordinary pass, roles of later
// ticks in pipelined mode (consecutive blocks' roles run in consecutive ticks: the read-modify-write of the trace arrays stays in order).
int plan_wf_state(sdrpp_ctx* c, int nframes, int level) {
    if (c->wf.height <= 0) { return SDRPP_OK; }
    if (c->data_width > 0) {  // FFT trace: latestFFT after smoothing / hold (pushFFT, waterfall.cpp:913-939)
        int rc = wf_ensure_trace(c);
        if (rc) { return rc; }
        FamilyTimer t(c, F_ZOOM);
        emit_wf(c, level + 1, F_ZOOM, TR_WF_TRACE, (c->data_width + 255) / 256, 1,
                TickWf{ c->d_zoomed, c->wf.d_latest, c->wf.d_smooth, c->wf.hold_on ? c->wf.d_hold : (float*)nullptr, nframes, c->data_width, 0, 0, c->wf.alpha, c->wf.beta, c->wf.hold_speed, 0.0f });
        c->wf.have_latest = true;
    }
    {   // raw lines into the ring (getFFTBuffer, waterfall.cpp:875-886)
        FamilyTimer t(c, F_ZOOM);
        emit_wf(c, level, F_ZOOM, TR_WF_RING, std::max(1, std::min(c->fft_size / 1024, 64)), nframes, TickWf{ c->d_lines, c->wf.d_ring, nullptr, nullptr, nframes, c->fft_size, c->wf.height, c->wf.cur, 0.0f, 0.0f, 0.0f, 0.0f });
        const long long nc = (long long)c->wf.cur - nframes;
        c->wf.cur = (int)(((nc % c->wf.height) + c->wf.height) % c->wf.height);
        c->wf.lines = (int)std::min<int64_t>((int64_t)c->wf.lines + nframes, c->wf.height);
    }
    return SDRPP_OK;
}

int do_fft(sdrpp_ctx* c, const IqSrc& src, int64_t count) {
    c->n_lines = 0;
    if (!c->fft_on) { return SDRPP_OK; }
    c->plan_fft_pos0 = c->fft_pos;
    c->plan_fft_next0 = c->fft_next;
    const int64_t P = (int64_t)c->nz + c->skip;
    const int64_t end = c->fft_pos + count;
    int64_t nframes = 0;
    if (end - c->nz - c->fft_next * P >= 0) { nframes = (end - c->nz - c->fft_next * P) / P + 1; }
    if (nframes > 0) {
        if ((size_t)nframes > c->lines_cap) { return fail(c, SDRPP_ERR_INVALID, "internal: %lld frames exceed line capacity %zu", (long long)nframes, c->lines_cap); }
        const size_t per_chunk = std::max<size_t>(1, kScratchBytes / ((size_t)c->fft_size * sizeof(float2)));
        if (c->tick_planning) {
            // one chunk only (the chunks of an ordinary pass share the scratch matrix one after the other)
            if ((size_t)nframes > per_chunk) {
                c->tick_abort = true;
                return SDRPP_OK;
            }
            FrameGeom g;
            g.nframes = (int)nframes;
            g.stride = (int)P;
            g.nz = c->nz;
            g.first_start = c->fft_next * P - c->fft_pos;
            int rc = plan_fft_roles(c, src, g, c->d_lines, c->zoom_grp ? c->d_lines_grp : nullptr);
            if (rc) { return rc; }
            const int lines_level = c->plan_lvl0 + (c->fft_lg <= 12 ? 1 : (c->fft_lg <= 16 ? 2 : 3));
            if (c->data_width > 0) {
                if ((size_t)nframes * (size_t)c->data_width > c->zoom_cap) { return fail(c, SDRPP_ERR_INVALID, "internal: zoom capacity"); }
                const float* zgrp = c->d_lines_grp;
                int gsz = c->zoom_grp;
                if (!zoom_uses_grp(c, zgrp, c->view_size, c->data_width, gsz)) { zgrp = nullptr; }
                const int tp = zoom_lanes(c, zgrp != nullptr);
                sdrpp_ctx::RoleLaunch z{};
                z.e.role = tp == 16 ? TR_ZOOM_16 : (tp == 4 ? TR_ZOOM_4 : TR_ZOOM_1);
                const int zgroups = (c->data_width + 256 / tp - 1) / (256 / tp);
                // pixel groups per workgroup (tick_kernels.h: tick_zoom): several once there are hundreds of them (10^6-sample blocks: 15 lines x 64
                // groups, 13.3 -> 13.9 GS/s with 8), one when a block completes a line or two (a workgroup's groups run one after the other)
                z.e.aux = std::max(1, std::min(std::min(c->tick_zoom_groups, zgroups), (int)nframes * zgroups / 128));
                z.e.gx = (zgroups + z.e.aux - 1) / z.e.aux;
                z.e.gy = (int)nframes;
                z.e.p.z = TickZoom{ c->d_lines, c->d_zstart, c->d_zcount, c->d_zoomed, c->d_index, zgrp, c->fft_size, c->data_width, gsz, c->wf_min, c->wf_max, 0 };
                z.lds = tick_lds_zoom(tp);
                z.level = lines_level + 1;
                z.fam = F_ZOOM;
                c->emits.push_back(z);
            }
            c->plan_top = std::max(c->plan_top, lines_level + (c->data_width > 0 ? 2 : 1));
            if (c->wf.height > 0) {
                rc = plan_wf_state(c, (int)nframes, lines_level + 1);
                if (rc) { return rc; }
                c->plan_top = std::max(c->plan_top, lines_level + (c->data_width > 0 ? 3 : 2));
            }
            c->fft_next += nframes;
            c->fft_pos = end;
            c->n_lines = (int)nframes;
            return SDRPP_OK;
        }
        for (int64_t f0 = 0; f0 < nframes; f0 += (int64_t)per_chunk) {
            FrameGeom g;
            g.nframes = (int)std::min<int64_t>((int64_t)per_chunk, nframes - f0);
            g.stride = (int)P;
            g.nz = c->nz;
            g.first_start = (c->fft_next + f0) * P - c->fft_pos;
            int rc = run_fft_chunk(c, src, g, c->d_lines + (size_t)f0 * c->fft_size, c->zoom_grp ? c->d_lines_grp + (size_t)f0 * (c->fft_size / c->zoom_grp) : nullptr);
            if (rc) { return rc; }
        }
        if (c->data_width > 0) {
            int rc = ensure_zoom(c, (size_t)nframes);
            if (rc) { return rc; }
            FamilyTimer t(c, F_ZOOM);
            launch_zoom(c->launch_stream, c->d_lines, (int)nframes, c->fft_size, c->view_size, c->data_width, c->d_zstart, c->d_zcount, c->wf_min, c->wf_max, c->d_zoomed, c->d_index,
                        c->d_lines_grp, c->zoom_grp, zoom_lanes(c, zoom_uses_grp(c, c->d_lines_grp, c->view_size, c->data_width, c->zoom_grp)));
        }
        if (c->wf.height > 0) {
            int rc = plan_wf_state(c, (int)nframes, 1);
            if (rc) { return rc; }
        }
        c->fft_next += nframes;
    }
    c->fft_pos = end;
    c->n_lines = (int)nframes;
    return SDRPP_OK;
}

}  // namespace
