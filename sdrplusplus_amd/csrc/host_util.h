// Host helpers: errors, device allocations, event timing, the job arena, streams and their histories, tap-table builders, the launch / emit pair
// (a kernel launched now — an ordinary pass — or queued as a role of a later tick).
// Part of the one translation unit sdrpp_gpu.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

int fail(sdrpp_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) { c->err = buf; }
    return code;
}

#define HIPCHK(c, expr)                                                                                           \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) { return fail((c), SDRPP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)

template <class T>
int dev_alloc(sdrpp_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16));
    if (e != hipSuccess) { return fail(c, SDRPP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e)); }
    *p = (T*)q;
    return SDRPP_OK;
}
template <class T>
void dev_free(T*& p) {
    if (p) { (void)hipFree((void*)p); p = nullptr; }
}
template <class T>
int upload(sdrpp_ctx* c, T** dst, const T* src, size_t count) {
    dev_free(*dst);
    if (count == 0) { return SDRPP_OK; }
    int rc = dev_alloc(c, dst, count);
    if (rc) { return rc; }
    HIPCHK(c, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return SDRPP_OK;
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) { l++; }
    return l;
}
bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ---- timing --------------------------------------------------------------------------------------------------------------
hipEvent_t get_event(sdrpp_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
// a tick launch timed through its own start / stop events (tick_host.h): read the pair out, give the start event back to the pool
void tick_ev_resolve(sdrpp_ctx* c, int k) {
    if (!c->tick_ev_start[k]) { return; }
    float ms = 0.0f;
    (void)hipEventSynchronize(c->tick_ev[k]);
    if (hipEventElapsedTime(&ms, c->tick_ev_start[k], c->tick_ev[k]) == hipSuccess) { c->fam_ms[F_TICK] += ms; }
    c->ev_pool.push_back(c->tick_ev_start[k]);
    c->tick_ev_start[k] = nullptr;
}
void timing_flush(sdrpp_ctx* c) {
    for (int k = 0; k < sdrpp_ctx::kTickEvents; k++) { tick_ev_resolve(c, k); }
    if (c->tpairs.empty()) { return; }
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->fft_stream);
    for (auto& p : c->tpairs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { c->fam_ms[p.family] += ms; }
        c->ev_pool.push_back(p.a);
        c->ev_pool.push_back(p.b);
    }
    c->tpairs.clear();
}
struct FamilyTimer {
    sdrpp_ctx* c;
    int fam;
    hipEvent_t a = nullptr;
    FamilyTimer(sdrpp_ctx* c_, int f) : c(c_), fam(f) {
        if (c->tick_planning) { return; }
        c->fam_launch[fam]++;
        if (c->timing && ((c->timing_mask >> fam) & 1u)) {
            a = get_event(c);
            (void)hipEventRecord(a, c->launch_stream);
        }
    }
    ~FamilyTimer() {
        if (c->timing && a) {
            hipEvent_t b = get_event(c);
            (void)hipEventRecord(b, c->launch_stream);
            c->tpairs.push_back({ a, b, fam });
            if (c->tpairs.size() > 8192) { timing_flush(c); }
        }
    }
};

// ---- the VFOs as a list (per-block loops) ----------------------------------------------------------------------------------------------
void vfo_list_rebuild(sdrpp_ctx* c) {
    c->vfo_list.clear();
    for (auto& kv : c->vfos) { c->vfo_list.push_back(kv.second.get()); }
}
// ---- job arena -------------------------------------------------------------------------------------------------------------
void tick_wait_done(sdrpp_ctx* c, uint64_t nticks);
int arena_begin(sdrpp_ctx* c) {
    c->arena_slot = (c->arena_slot + 1) % kArenaSlots;
    if (c->arena_used[c->arena_slot]) {
        HIPCHK(c, hipEventSynchronize(c->arena_ev[c->arena_slot]));
        c->arena_used[c->arena_slot] = false;
    }
    if (c->arena_tick[c->arena_slot]) {  // last used by a tick: its upload has run once that tick is complete (kArenaSlots ticks ago: normally long done)
        tick_wait_done(c, c->arena_tick[c->arena_slot]);
        c->arena_tick[c->arena_slot] = 0;
    }
    c->arena_dev = c->arena_dev_slot[c->arena_slot];
    c->arena_off = 0;
    return SDRPP_OK;
}
template <class T>
T* arena_push(sdrpp_ctx* c, const std::vector<T>& v, T** host_copy = nullptr) {
    if (v.empty()) { return nullptr; }
    size_t off = (c->arena_off + 63) & ~(size_t)63;
    size_t bytes = v.size() * sizeof(T);
    if (off + bytes > kArenaBytes) { return nullptr; }
    // test hook (SDRPP_GPU_TEST_FAIL_ARENA="pass:allocation"): the job tables of that pass "do not fit" from that allocation on — the push
    // fails half-way through its planning and must leave the stream exactly as it was (tests/test_parity_vfo.py::test_failed_push_changes_nothing)
    if (c->test_fail_pass > 0 && c->arena_begins == c->test_fail_pass && ++c->arena_allocs >= c->test_fail_alloc) { return nullptr; }  // (counted in block_bounds)
    memcpy(c->arena_host[c->arena_slot] + off, v.data(), bytes);
    if (host_copy) { *host_copy = (T*)(c->arena_host[c->arena_slot] + off); }
    c->arena_off = off + bytes;
    return (T*)(c->arena_dev + off);
}
// The job tables of one push (a few tens of KB) travel from the pinned host slot to the device arena through a tiny copy KERNEL
// that reads the pinned (device-mapped) host memory directly.  hipMemcpyAsync is avoided on purpose: above ~16 KB the runtime's
// staged copy path was measured to block the enqueuing thread for up to 8 ms every few pushes (tools/hosttime.py,
// SDRPP_GPU_HOSTPROF=1), which starved the GPU; a kernel launch costs ~7 us of host time, always.
__global__ __launch_bounds__(256) void arena_upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) { dst[i] = src[i]; }
}
int arena_commit(sdrpp_ctx* c) {
    if (c->arena_off == 0 || c->tick_planning) { return SDRPP_OK; }  // (pipelined: the upload is part of the tick)
    const int n16 = (int)((c->arena_off + 15) / 16);
    hipLaunchKernelGGL(arena_upload_kernel, dim3((unsigned)std::min((n16 + 255) / 256, 64)), dim3(256), 0, c->stream,
                       (const uint4*)c->arena_host_dev[c->arena_slot], (uint4*)c->arena_dev, n16);
    return SDRPP_OK;
}
int arena_end(sdrpp_ctx* c) {
    HIPCHK(c, hipEventRecord(c->arena_ev[c->arena_slot], c->stream));
    c->arena_used[c->arena_slot] = true;
    return SDRPP_OK;
}

// ---- streams ---------------------------------------------------------------------------------------------------------------
int stream_alloc(sdrpp_ctx* c, Stream& s, int width, int hist_len, size_t cap) {
    s.width = width;
    s.hist_len = hist_len;
    s.cap = cap;
    s.cur = 0;
    s.n = 0;
    int rc = dev_alloc(c, &s.base, (cap + 16) * width);
    if (rc) { return rc; }
    s.data = s.base;
    for (int i = 0; i < 2; i++) {
        rc = dev_alloc(c, &s.hist[i], (size_t)std::max(hist_len, 1) * width);
        if (rc) { return rc; }
        HIPCHK(c, hipMemset(s.hist[i], 0, (size_t)std::max(hist_len, 1) * width * sizeof(float)));
    }
    return SDRPP_OK;
}
void stream_free(Stream& s) {
    dev_free(s.base);
    for (int i = 0; i < kRing - 1; i++) { dev_free(s.extra[i]); }
    s.n_extra = 0;
    s.rot = 0;
    s.data = nullptr;
    dev_free(s.hist[0]);
    dev_free(s.hist[1]);
}
StreamIn stream_in(const Stream& s) { return StreamIn{ s.data, s.hist[s.cur], s.hist_len, s.n }; }
// pipelined mode: kRing data buffers per stream, used round robin block by block
int stream_ring_ensure(sdrpp_ctx* c, Stream& s) {
    while (s.n_extra < kRing - 1) {
        int rc = dev_alloc(c, &s.extra[s.n_extra], (s.cap + 16) * s.width);
        if (rc) { return rc; }
        s.n_extra++;
    }
    return SDRPP_OK;
}
void stream_rotate(Stream& s) {
    if (s.n_extra == 0 || !s.base) { return; }
    s.prev_data = s.data;
    s.prev_n = s.n;
    std::swap(s.base, s.extra[s.rot]);
    s.data = s.base;
    s.rot = (s.rot + 1) % s.n_extra;
}

// Enlarge a stream's history (a consumer got more taps): the existing samples stay the most recent ones, older entries are
// zero — exactly what fir.h:44-47 does to its delay line when the tap count grows.
int stream_grow_hist(sdrpp_ctx* c, Stream& s, int new_len) {
    if (new_len <= s.hist_len) { return SDRPP_OK; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float* nh[2] = { nullptr, nullptr };
    for (int i = 0; i < 2; i++) {
        int rc = dev_alloc(c, &nh[i], (size_t)new_len * s.width);
        if (rc) { return rc; }
        HIPCHK(c, hipMemset(nh[i], 0, (size_t)new_len * s.width * sizeof(float)));
    }
    if (s.hist_len > 0) {
        HIPCHK(c, hipMemcpy(nh[0] + (size_t)(new_len - s.hist_len) * s.width, s.hist[s.cur], (size_t)s.hist_len * s.width * sizeof(float), hipMemcpyDeviceToDevice));
    }
    dev_free(s.hist[0]);
    dev_free(s.hist[1]);
    s.hist[0] = nh[0];
    s.hist[1] = nh[1];
    s.cur = 0;
    s.hist_len = new_len;
    return SDRPP_OK;
}

int ensure_iq_hist(sdrpp_ctx* c, int need) {
    if (need <= c->iq_hist_cap) { return SDRPP_OK; }
    int cap = std::max(need, 1024);
    float* nh[2] = { nullptr, nullptr };
    for (int i = 0; i < 2; i++) {
        int rc = dev_alloc(c, &nh[i], (size_t)cap * 2);
        if (rc) { return rc; }
        HIPCHK(c, hipMemset(nh[i], 0, (size_t)cap * 2 * sizeof(float)));
    }
    if (c->iq_hist[c->iq_cur] && c->iq_hist_cap > 0) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // keep the most recent samples at the END of the larger buffer
        HIPCHK(c, hipMemcpy(nh[0] + (size_t)(cap - c->iq_hist_cap) * 2, c->iq_hist[c->iq_cur], (size_t)c->iq_hist_cap * 2 * sizeof(float), hipMemcpyDeviceToDevice));
    }
    dev_free(c->iq_hist[0]);
    dev_free(c->iq_hist[1]);
    c->iq_hist[0] = nh[0];
    c->iq_hist[1] = nh[1];
    c->iq_cur = 0;
    c->iq_hist_cap = cap;
    return SDRPP_OK;
}

// ---- VFO helpers -------------------------------------------------------------------------------------------------------------
// Phase-major, zero-padded tap layout of the register-blocked FIR kernel: t[p][q] = h[D*q + p], q < kp (multiple of R).
std::vector<float> blocked_taps(const float* h, int K, int D, int* kp_out) {
    const int R = SDRPP_FIR_R;
    const int per = (K + D - 1) / D;
    const int kp = ((per + R - 1) / R) * R;
    std::vector<float> t((size_t)D * kp, 0.0f);
    for (int k = 0; k < K; k++) { t[(size_t)(k % D) * kp + (size_t)(k / D)] = h[k]; }
    *kp_out = kp;
    return t;
}
int upload_blocked(sdrpp_ctx* c, float** dst, const float* h, int K, int D, int* kp) {
    std::vector<float> t = blocked_taps(h, K, D, kp);
    return upload(c, dst, t.data(), t.size());
}

void toep_free(ToepTab& T) {
    dev_free(T.d_tl);
    dev_free(T.d_lb);
    const int kind = T.kind;
    T = ToepTab{};
    T.kind = kind;
}
int toep_upload(sdrpp_ctx* c, ToepTab& T, const std::vector<float>& tl, const std::vector<int>& lb) {
    dev_free(T.d_tl);
    dev_free(T.d_lb);
    T.ok = false;
    int rc = upload(c, &T.d_tl, tl.data(), tl.size());
    if (rc) { return rc; }
    rc = upload(c, &T.d_lb, lb.data(), lb.size());
    if (rc) { return rc; }
    T.tl_len = (int)tl.size();
    T.nvar = (int)lb.size() / 64;
    // usable only if four wavefront windows (two planes each) + the tap table fit the block's LDS budget (1/3 of a CU) — very
    // long filters stay on the register-blocked VALU kernels
    const int span = (2 * 16 - 1) * T.s_in + 4 * T.nsteps, pl = (span + 8) & ~3;
    T.ok = ((size_t)((T.tl_len + 3) & ~3) + (size_t)4 * 2 * pl) * sizeof(float) <= (size_t)(160 * 1024 / 3);
    // A/B switch for benchmarking: SDRPP_GPU_VALU_FIR=<mask> keeps the register-blocked VALU kernels (1: decimators, 2: resampler,
    // 4: channel filter, 8: audio low-pass; 15 = all)
    if (const char* e = getenv("SDRPP_GPU_VALU_FIR")) {
        if (atoi(e) & T.kind) { T.ok = false; }
    }
    return SDRPP_OK;
}
// FIR decimating by D: tile = 15 outputs, window offset k' = D * m + k  ->  B[k'][m] = h[k' - D * m]
int toep_build_fir(sdrpp_ctx* c, ToepTab& T, const float* h, int K, int D) {
    const int rows = 15, padl = (rows - 1) * D, kp = K + padl, nsteps = (kp + 3) / 4;
    const int mainlen = padl + 4 * nsteps + 4, zb = mainlen;
    std::vector<float> tl((size_t)mainlen + 4 * (size_t)nsteps + 4, 0.0f);
    for (int k = 0; k < K; k++) { tl[(size_t)padl + k] = h[k]; }
    std::vector<int> lb(64);
    for (int lane = 0; lane < 64; lane++) {
        const int m = lane & 15, kk = lane >> 4;
        lb[(size_t)lane] = (m < rows) ? padl - D * m + kk : zb + kk;
    }
    T.nsteps = nsteps;
    T.s_in = rows * D;
    T.rows = rows;
    return toep_upload(c, T, tl, lb);
}
// Polyphase resampler L/M (bank[phase][tpp], polyphase_resampler.h:75-93): tile = CY whole phase cycles (CY * L <= 15 outputs,
// CY * M inputs); output (cy, r) uses phase (phase0 + r*M) % L at window offset cy*M + (phase0 + r*M) / L.  One lane-base set
// per carried phase0.
int toep_build_poly(sdrpp_ctx* c, ToepTab& T, const std::vector<float>& bank, int L, int M, int tpp) {
    const int cy_n = 15 / L;
    if (cy_n < 1) { return SDRPP_OK; }  // T.ok stays false: the VALU kernels handle it
    const int rows = cy_n * L, omax = ((L - 1) + (L - 1) * M) / L, shift_max = (cy_n - 1) * M + omax;
    const int kp = shift_max + tpp, nsteps = (kp + 3) / 4, padp = shift_max, row = padp + 4 * nsteps + 4, zb = L * row;
    std::vector<float> tl((size_t)zb + 4 * (size_t)nsteps + 4, 0.0f);
    for (int ph = 0; ph < L; ph++) {
        for (int q = 0; q < tpp; q++) { tl[(size_t)ph * row + (size_t)padp + q] = bank[(size_t)ph * tpp + q]; }
    }
    std::vector<int> lb((size_t)L * 64);
    for (int ph0 = 0; ph0 < L; ph0++) {
        for (int lane = 0; lane < 64; lane++) {
            const int m = lane & 15, kk = lane >> 4;
            if (m < rows) {
                const int cy = m / L, r = m % L, A = ph0 + r * M;
                lb[(size_t)ph0 * 64 + lane] = (A % L) * row + padp - (cy * M + A / L) + kk;
            }
            else { lb[(size_t)ph0 * 64 + lane] = zb + kk; }
        }
    }
    T.nsteps = nsteps;
    T.s_in = cy_n * M;
    T.rows = rows;
    return toep_upload(c, T, tl, lb);
}

void build_modtaps(Vfo& v) {
    // g[k] = h[k] * exp(j*2*pi*(k - kc)*theta), kc = (K-1)/2, for the first (K+1)/2 taps; the other half is the conjugate
    // mirror (stage1_accumulate).  An odd K has a real centre tap.
    const int K = v.d.stage_ntaps[0];
    const int npairs = (K + 1) / 2;
    v.modtaps.resize((size_t)npairs);
    const double kc = 0.5 * (double)(K - 1);
    for (int k = 0; k < npairs; k++) {
        double t = ((double)k - kc) * v.theta;
        t -= std::rint(t);
        const double a = 2.0 * 3.14159265358979323846 * t;
        const double h = (double)v.staps[0][(size_t)k];
        v.modtaps[(size_t)k] = make_float2((float)(h * std::cos(a)), (float)(h * std::sin(a)));
    }
    if (K & 1) { v.modtaps[(size_t)npairs - 1] = make_float2(v.staps[0][(size_t)npairs - 1], 0.0f); }
    v.modtaps_dirty = false;
}

void vfo_free(Vfo& v) {
    for (int i = 0; i < SDRPP_MAX_DECIM_STAGES; i++) { dev_free(v.d_staps[i]); dev_free(v.d_staps_nat[i]); }
    dev_free(v.d_bank);
    dev_free(v.d_cyc);
    dev_free(v.d_chan);
    dev_free(v.d_audio);
    dev_free(v.d_state);
    dev_free(v.d_h12);
    dev_free(v.d_rot);
    for (int i = 0; i < SDRPP_MAX_DECIM_STAGES; i++) { toep_free(v.tp_stage[i]); }
    toep_free(v.tp_poly);
    toep_free(v.tp_chan);
    toep_free(v.tp_audio);
    for (int i = 0; i < SDRPP_MAX_DECIM_STAGES; i++) {
        dev_free(v.af.d_staps[i]);
        toep_free(v.af.tp_stage[i]);
    }
    toep_free(v.af.tp_poly);
    toep_free(v.af.tp_hpf);
    dev_free(v.af.d_bank);
    dev_free(v.af.d_hpf);
    dev_free(v.af.d_last);
    dev_free(v.af.d_seg);
    for (auto& s : v.st) { stream_free(s); }
    v.st.clear();
}

int vfo_reset_state(sdrpp_ctx* c, Vfo& v) {
    v.phi = 0.0;
    v.phi2 = 0.0;
    v.seen = 0;
    v.recs.clear();
    std::fill(v.chan_stale.begin(), v.chan_stale.end(), 0.0f);  // RxVFO::reset clears the channel filter's delay line whether it is bypassed or not (rx_vfo.h:79-87)
    for (int i = 0; i < SDRPP_MAX_DECIM_STAGES; i++) { v.soff[i] = 0; }
    v.pphase = 0;
    v.poff = 0;
    for (int i = 0; i < SDRPP_MAX_DECIM_STAGES; i++) { v.af.soff[i] = 0; }
    v.af.pphase = 0;
    v.af.poff = 0;
    if (v.af.d_last) { HIPCHK(c, hipMemsetAsync(v.af.d_last, 0, 2 * sizeof(float2), c->stream)); }
    for (auto& s : v.st) {
        for (int i = 0; i < 2; i++) {
            if (s.hist[i]) { HIPCHK(c, hipMemsetAsync(s.hist[i], 0, (size_t)std::max(s.hist_len, 1) * s.width * sizeof(float), c->stream)); }
        }
        s.n = 0;
    }
    AgcState st[2];
    for (int i = 0; i < 2; i++) {
        st[i].set_point = v.d.agc_set_point;
        st[i].attack = v.d.agc_attack;
        st[i].inv_attack = 1.0f - v.d.agc_attack;
        st[i].decay = v.d.agc_decay;
        st[i].inv_decay = 1.0f - v.d.agc_decay;
        st[i].max_gain = v.d.agc_max_gain;
        st[i].max_output_amp = v.d.agc_max_output_amp;
        st[i].amp = v.d.agc_set_point / v.d.agc_init_gain;  // agc.h:25
    }
    char blob[2 * sizeof(AgcState) + sizeof(float)];
    memcpy(blob, st, sizeof(st));
    float zero = 0.0f;
    memcpy(blob + sizeof(st), &zero, sizeof(float));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(v.d_state, blob, sizeof(blob), hipMemcpyHostToDevice));
    const float2 unit[2] = { make_float2(1.0f, 0.0f), make_float2(1.0f, 0.0f) };  // frequency_xlator.h:21: phase = (1, 0)
    HIPCHK(c, hipMemcpy(v.d_rot, unit, sizeof(unit), hipMemcpyHostToDevice));
    return SDRPP_OK;
}

// Number of outputs a decimating stage produces from n inputs with carried offset `off` (decimating_fir.h:51-62).
inline int decim_nout(int n, int off, int D) { return (n > off) ? (n - off + D - 1) / D : 0; }
// Outputs of the polyphase resampler (polyphase_resampler.h:75-93): smallest m with poff + (pphase + m*M)/L >= n.
inline int poly_nout(int n, int poff, int pphase, int L, int M) {
    if (n <= poff) { return 0; }
    const long long need = (long long)L * (n - poff) - pphase;  // A_m >= L*(n - poff)
    return (int)((need + M - 1) / M);
}

// Reference-block ends (cumulative counts) carried through a stage: outputs produced once the first `b` inputs are in.  Evaluate
// with the stage's state BEFORE the push updates it.
void bounds_decim(std::vector<int>& b, int off, int D) {
    for (auto& x : b) { x = decim_nout(x, off, D); }
}
void bounds_poly(std::vector<int>& b, int poff, int pphase, int L, int M) {
    for (auto& x : b) { x = poly_nout(x, poff, pphase, L, M); }
}

// Host-side enqueue profiler (SDRPP_GPU_HOSTPROF=1): wall time spent inside named sections of the push path, printed when the
// context is destroyed.  Diagnostic only.
struct HostProf {
    struct Acc { double total = 0.0, mx = 0.0; long n = 0; };
    std::map<std::string, Acc> acc;
    bool on = getenv("SDRPP_GPU_HOSTPROF") != nullptr;
    void add(const char* name, double us) {
        Acc& a = acc[name];
        a.total += us;
        a.mx = std::max(a.mx, us);
        a.n++;
    }
    void report() {
        if (!on) { return; }
        for (auto& kv : acc) { fprintf(stderr, "[sdrpp hostprof] %-28s n=%6ld avg %8.1f us max %9.1f us\n", kv.first.c_str(), kv.second.n, kv.second.total / (double)kv.second.n, kv.second.mx); }
    }
};
HostProf g_hostprof;
struct HostScope {
    const char* name;
    std::chrono::steady_clock::time_point t0;
    explicit HostScope(const char* n) : name(n) { if (g_hostprof.on) { t0 = std::chrono::steady_clock::now(); } }
    ~HostScope() {
        if (g_hostprof.on) { g_hostprof.add(name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); }
    }
};

template <class K, class... A>
void launch(sdrpp_ctx* c, K kernel, dim3 grid, dim3 block, size_t lds, A... args) {
    if (c->tick_planning) {  // a kernel that is not a role of the tick kernel: this block runs as an ordinary pass instead
        if (!c->tick_abort && getenv("SDRPP_TICK_DEBUG")) { fprintf(stderr, "[sdrpp tick] ordinary pass because of %s\n", __PRETTY_FUNCTION__); }
        c->tick_abort = true;
        return;
    }
    HostScope hs("launch");
    hipLaunchKernelGGL(kernel, grid, block, lds, c->launch_stream, args...);
}

// ---- roles: kernels that exist both as a launch of their own and as a role of the tick kernel ------------------------------------------
void launch_role(sdrpp_ctx* c, const sdrpp_ctx::RoleLaunch& r) {
    const TickEntry& e = r.e;
    const dim3 grid((unsigned)e.gx, (unsigned)e.gy), b256(256);
    hipStream_t st = c->launch_stream;
    HostScope hs("launch");
    switch (e.role) {
    case TR_COPY: hipLaunchKernelGGL(copy_kernel, grid, b256, 0, st, (const CopyJob*)e.jobs); break;
    case TR_CARRY: hipLaunchKernelGGL(carry_kernel, grid, b256, 0, st, (const CarryJob*)e.jobs, e.aux); break;
    case TR_ROT: hipLaunchKernelGGL(vfo_rotate_kernel, grid, b256, 0, st, e.p.src, (const RotJob*)e.jobs); break;
    case TR_FCM_132_4: hipLaunchKernelGGL((vfo_frontcm_kernel<10, 132, 4>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCM_6: hipLaunchKernelGGL((vfo_frontcm_kernel<6, 0, 0>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCM_10: hipLaunchKernelGGL((vfo_frontcm_kernel<10, 0, 0>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCM_16: hipLaunchKernelGGL((vfo_frontcm_kernel<16, 0, 0>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCM16_132_4: hipLaunchKernelGGL((vfo_frontcm16_kernel<132, 4>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCM16W_132_4: hipLaunchKernelGGL((vfo_frontcm16w_kernel<132, 4>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCL_0: hipLaunchKernelGGL((vfo_frontcl_kernel<0>), grid, dim3(128), r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCL_PF: hipLaunchKernelGGL((vfo_frontcl_kernel<SDRPP_FCL_PF>), grid, dim3(128), r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_TOEP_C: hipLaunchKernelGGL((vfo_toep_kernel<2, 2, false>), grid, b256, r.lds, st, (const ToepJob*)e.jobs); break;
    case TR_TOEP_R: hipLaunchKernelGGL((vfo_toep_kernel<1, 2, false>), grid, b256, r.lds, st, (const ToepJob*)e.jobs); break;
    case TR_TOEP_Q: hipLaunchKernelGGL((vfo_toep_kernel<1, 2, true>), grid, b256, r.lds, st, (const ToepJob*)e.jobs); break;
    case TR_FIRB_C: hipLaunchKernelGGL((vfo_firb_kernel<2, false>), grid, dim3((unsigned)e.aux), r.lds, st, (const FirBJob*)e.jobs); break;
    case TR_FIRB_R: hipLaunchKernelGGL((vfo_firb_kernel<1, false>), grid, dim3((unsigned)e.aux), r.lds, st, (const FirBJob*)e.jobs); break;
    case TR_FIRB_S: hipLaunchKernelGGL((vfo_firb_kernel<1, true>), grid, dim3((unsigned)e.aux), r.lds, st, (const FirBJob*)e.jobs); break;
    case TR_FIRB_Q: hipLaunchKernelGGL((vfo_firb_kernel<1, true, true>), grid, dim3((unsigned)e.aux), r.lds, st, (const FirBJob*)e.jobs); break;
    case TR_PRE: hipLaunchKernelGGL(vfo_demod_pre_kernel, grid, b256, 0, st, (const PreJob*)e.jobs); break;
    case TR_SEQ: hipLaunchKernelGGL(vfo_sequential_kernel, grid, dim3(64), 0, st, (const SeqJob*)e.jobs, e.aux); break;
    case TR_PIPE: hipLaunchKernelGGL(vfo_pipe_kernel<1>, grid, b256, r.lds, st, (const PipeJob*)e.jobs); break;
    case TR_POLYC: hipLaunchKernelGGL(vfo_polyc_kernel, grid, b256, r.lds, st, (const PolyJob*)e.jobs, e.aux); break;
    case TR_DEEMP_P0: hipLaunchKernelGGL((vfo_deemph_kernel<0, 0>), grid, b256, 0, st, (const DeempJob*)e.jobs); break;
    case TR_DEEMP_P1: hipLaunchKernelGGL((vfo_deemph_kernel<0, 1>), grid, b256, 0, st, (const DeempJob*)e.jobs); break;
    case TR_DC_P0: hipLaunchKernelGGL((vfo_deemph_kernel<1, 0>), grid, b256, 0, st, (const DeempJob*)e.jobs); break;
    case TR_DC_P1: hipLaunchKernelGGL((vfo_deemph_kernel<1, 1>), grid, b256, 0, st, (const DeempJob*)e.jobs); break;
    case TR_WF_RING: hipLaunchKernelGGL(wf_ring_store_kernel, grid, b256, 0, st, e.p.wf.src, e.p.wf.n0, e.p.wf.n1, e.p.wf.a, e.p.wf.n2, e.p.wf.n3); break;
    case TR_WF_TRACE: hipLaunchKernelGGL(wf_trace_kernel, grid, b256, 0, st, e.p.wf.src, e.p.wf.n0, e.p.wf.n1, e.p.wf.a, e.p.wf.b, e.p.wf.f0, e.p.wf.f1, e.p.wf.c, e.p.wf.f2); break;
    default: break;  // (the FFT branch launches its kernels itself outside pipelined mode: its pass-1 workgroups are wider there)
    }
}
// A launch of the block being processed at `level` of its data flow: now (an ordinary pass), or `level` ticks from now (pipelined).
void emit(sdrpp_ctx* c, int level, int fam, int role, int gx, int gy, size_t lds, const void* jobs, const IqSrc* src = nullptr, int aux = 0) {
    if (gx <= 0 || gy <= 0) { return; }
    sdrpp_ctx::RoleLaunch r{};
    r.e.role = role;
    r.e.gx = gx;
    r.e.gy = gy;
    r.e.aux = aux;
    r.e.jobs = jobs;
    if (src) { r.e.p.src = *src; }
    r.lds = lds;
    r.level = level;
    r.fam = fam;
    if (c->tick_planning) { c->emits.push_back(r); }
    else { launch_role(c, r); }
}

// ... the same for a role whose parameters travel in the entry itself (TickWf)
void emit_wf(sdrpp_ctx* c, int level, int fam, int role, int gx, int gy, const TickWf& q) {
    if (gx <= 0 || gy <= 0) { return; }
    sdrpp_ctx::RoleLaunch r{};
    r.e.role = role;
    r.e.gx = gx;
    r.e.gy = gy;
    r.e.p.wf = q;
    r.level = level;
    r.fam = fam;
    if (c->tick_planning) { c->emits.push_back(r); }
    else { launch_role(c, r); }
}

int pick_tile(int D, int K, int width_bytes) {
    for (int tile : { 256, 128, 64 }) {
        const int extra = (K - 1 + D - 1) / D;
        const size_t lds = (size_t)D * (tile + extra + 1) * width_bytes;
        if (lds <= (size_t)kMaxLds) { return tile; }
    }
    return 0;
}
size_t fir_lds(int tile, int D, int K, int width_bytes) {
    const int extra = (K - 1 + D - 1) / D;
    return (size_t)D * (tile + extra + 1) * width_bytes;
}

}  // namespace
