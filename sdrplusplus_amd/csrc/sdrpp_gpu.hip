// C-ABI implementation (include/sdrpp_gpu.h): context, streaming state, per-push planning and kernel launches.
// One context = one IQ stream on one GPU; everything is enqueued on one HIP stream so a push is a fixed sequence of
// launches whose sizes are computed on the host from integer state (decimation offsets, polyphase phase, frame position).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdrpp_gpu.h"
#include "fft_kernels.h"
#include "host_design.h"
#include "vfo_kernels.h"
#include "pipe_kernels.h"
#include "tick_kernels.h"

using namespace sdrpp_k;

#include "host_ctx.h"
#include "host_util.h"
#include "plan_fft.h"
#include "plan_vfo.h"
#include "plan_pre.h"
#include "plan_push.h"
#include "tick_host.h"

// =====================================================================================================================
// C ABI
// =====================================================================================================================
// HIP's current device is a per-THREAD setting: a host that owns several contexts on several GPUs (StreamBank, one worker thread
// per IQFrontEnd) calls into a context from threads whose current device is some other GPU.  Every entry point that takes a context
// therefore makes the context's device current for the duration of the call (allocations, launches, copies, symbol accesses all
// follow the current device) and restores the caller's on return.
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(const sdrpp_ctx* c) {
        if (!c) { return; }
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != c->device) {
            prev = cur;
            (void)hipSetDevice(c->device);
        }
    }
    ~DeviceScope() {
        if (prev >= 0) { (void)hipSetDevice(prev); }
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
// every call that observes results or changes the configuration first processes what deferred pushes have staged
#define FLUSH_PENDING(c)                          \
    do {                                          \
        int frc_ = flush_pending(c);              \
        if (frc_) { return frc_; }                \
    } while (0)

extern "C" {

int flush_pending(sdrpp_ctx* c);  // deferred pushes -> one pass (defined with the data path below; internal, not part of the ABI header)
int flush_pending_opt(sdrpp_ctx* c, int drain);  // drain = 0: a call that only reports what the HOST knows (counts) leaves the tick queue of the pipelined mode alone

const char* sdrpp_strerror(int code) {
    switch (code) {
    case SDRPP_OK: return "ok";
    case SDRPP_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case SDRPP_ERR_INVALID: return "invalid argument or call sequence";
    case SDRPP_ERR_NOMEM: return "device memory allocation failed";
    case SDRPP_ERR_HIP: return "HIP runtime error";
    case SDRPP_ERR_UNSUPPORTED: return "unsupported parameter";
    case SDRPP_ERR_NOT_FOUND: return "no such VFO";
    default: return "unknown error";
    }
}

int sdrpp_abi_version(int* sizeof_vfo_desc) {  // 2: sdrpp_vfo_desc carries nco_mode (round 3); sdrpp_pipeline_stats
    if (sizeof_vfo_desc) { *sizeof_vfo_desc = (int)sizeof(sdrpp_vfo_desc); }
    return SDRPP_ABI_VERSION;
}

const char* sdrpp_kernel_family_name(int family) { return (family >= 0 && family < SDRPP_NUM_KERNEL_FAMILIES) ? kFamilyNames[family] : "?"; }

int sdrpp_device_count(void) {
    int n = 0;
    return (hipGetDeviceCount(&n) == hipSuccess) ? n : 0;
}

int sdrpp_create(int device, int64_t max_push, sdrpp_ctx** out) {
    if (!out || max_push <= 0 || max_push > ((int64_t)1 << 28)) { return SDRPP_ERR_INVALID; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) { return SDRPP_ERR_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { return SDRPP_ERR_NO_DEVICE; }
    sdrpp_ctx* c = new sdrpp_ctx;
    c->device = device;
    c->max_push = max_push;
    if (const char* tf = getenv("SDRPP_GPU_TEST_FAIL_ARENA")) {
        long a = 0;
        int b = 0;
        if (sscanf(tf, "%ld:%d", &a, &b) == 2) {
            c->test_fail_pass = a;
            c->test_fail_alloc = b;
        }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        char b[600];
        snprintf(b, sizeof(b), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
        c->devinfo = b;
        c->num_cus = std::max(1, prop.multiProcessorCount);
        // the long-first-stage role with four tile engines asks for up to 80 KB of dynamic LDS per workgroup (two workgroups per CU): above the
        // 64 KB a launch gets without asking
        // (tick_kernel<0> as well: a long first stage whose window does NOT fit the register prefetch — nsamp > 64 * SDRPP_FCL_PF, a custom plan with
        // a very long /64 stage — goes out as role TR_FCL_0 in the SET = 0 build with the same four-engine LDS request)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tick_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tick_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / 2);
        (void)hipGetLastError();
    }
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SDRPP_ERR_NO_DEVICE;
    }
    c->stream = c->own_stream;
    c->launch_stream = c->stream;
    {   // the FFT branch's stream gets the LOWEST queue priority: it is the filler behind the VFO bank, whose launches form the critical
        // path (measured on the headline step: 0.686-0.693 ms against 0.704-0.706 at equal priority and 0.699-0.701 with the FFT favoured)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least (numerically largest), hi = greatest
        if (hipStreamCreateWithPriority(&c->fft_stream, hipStreamNonBlocking, lo) != hipSuccess) { c->fft_stream = nullptr; }
    }
    if ((!c->fft_stream && hipStreamCreateWithFlags(&c->fft_stream, hipStreamNonBlocking) != hipSuccess) || hipEventCreate(&c->ev_fork) != hipSuccess ||
        hipEventCreate(&c->ev_join) != hipSuccess) {
        sdrpp_destroy(c);
        return SDRPP_ERR_NO_DEVICE;
    }
    for (int i = 0; i < kArenaSlots; i++) {
        if (hipHostMalloc((void**)&c->arena_host[i], kArenaBytes, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&c->arena_host_dev[i], c->arena_host[i], 0) != hipSuccess || hipEventCreate(&c->arena_ev[i]) != hipSuccess) {
            sdrpp_destroy(c);
            return SDRPP_ERR_NOMEM;
        }
    }
    if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) {
        sdrpp_destroy(c);
        return SDRPP_ERR_NO_DEVICE;
    }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev_copy) != hipSuccess ||
        hipEventCreate(&c->land_ev[0]) != hipSuccess || hipEventCreate(&c->land_ev[1]) != hipSuccess) {
        sdrpp_destroy(c);
        return SDRPP_ERR_NO_DEVICE;
    }
    for (int i = 0; i < kArenaSlots; i++) {
        if (dev_alloc(c, &c->arena_dev_slot[i], kArenaBytes) != SDRPP_OK) {
            sdrpp_destroy(c);
            return SDRPP_ERR_NOMEM;
        }
    }
    c->arena_dev = c->arena_dev_slot[0];
    if (dev_alloc(c, &c->iq_land[0], (size_t)max_push * 2 + 32) != SDRPP_OK) {
        sdrpp_destroy(c);
        return SDRPP_ERR_NOMEM;
    }
    // completion flag and counter of the pipelined mode (sdrpp_set_pipelined)
    if (dev_alloc(c, &c->d_tick_counter, 4) != SDRPP_OK || dev_alloc(c, &c->empty_tab, 1) != SDRPP_OK ||
        hipMemset(c->d_tick_counter, 0, 16) != hipSuccess || hipMemset(c->empty_tab, 0, sizeof(TickTable)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_tick_flag, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->hd_tick_flag, c->h_tick_flag, 0) != hipSuccess) {
        sdrpp_destroy(c);
        return SDRPP_ERR_NOMEM;
    }
    *c->h_tick_flag = 0;
    c->h_tick_flag[8] = 0;  // flag waits that gave up (PipeJob::timeouts)
    *out = c;
    return SDRPP_OK;
}

// ---- WaterFall display state ----------------------------------------------------------------------------------
static void wf_free(sdrpp_ctx* c) {
    dev_free(c->wf.d_ring);
    dev_free(c->wf.d_latest);
    dev_free(c->wf.d_smooth);
    dev_free(c->wf.d_hold);
    c->wf = sdrpp_ctx::Wf{};
}

int sdrpp_wf_configure(sdrpp_ctx* c, int height) {
    DeviceScope dev_scope_(c);
    if (!c || height < 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->fft_stream) { HIPCHK(c, hipStreamSynchronize(c->fft_stream)); }
    wf_free(c);
    if (height == 0) { return SDRPP_OK; }
    if (!c->fft_on) { return fail(c, SDRPP_ERR_INVALID, "configure the FFT first (sdrpp_fft_configure)"); }
    int rc = dev_alloc(c, &c->wf.d_ring, (size_t)height * c->fft_size);
    if (rc) { return rc; }
    HIPCHK(c, hipMemset(c->wf.d_ring, 0, (size_t)height * c->fft_size * sizeof(float)));
    c->wf.height = height;
    return SDRPP_OK;
}

int sdrpp_wf_set_smoothing(sdrpp_ctx* c, int enabled, float speed) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (c->wf.height <= 0) { return fail(c, SDRPP_ERR_INVALID, "no waterfall history configured (sdrpp_wf_configure)"); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->fft_stream) { HIPCHK(c, hipStreamSynchronize(c->fft_stream)); }
    int rc = wf_ensure_trace(c);
    if (rc) { return rc; }
    sdrpp_ctx::Wf& W = c->wf;
    dev_free(W.d_smooth);  // setFFTSmoothing (waterfall.cpp:1166-1188): the buffer is re-created as a copy of latestFFT
    if (enabled && W.width > 0) {
        rc = dev_alloc(c, &W.d_smooth, (size_t)W.width);
        if (rc) { return rc; }
        HIPCHK(c, hipMemcpy(W.d_smooth, W.d_latest, (size_t)W.width * sizeof(float), hipMemcpyDeviceToDevice));
    }
    W.alpha = speed;  // setFFTSmoothingSpeed (:1190-1194)
    W.beta = 1.0f - speed;
    return SDRPP_OK;
}

int sdrpp_wf_set_hold(sdrpp_ctx* c, int enabled, float speed) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (c->wf.height <= 0) { return fail(c, SDRPP_ERR_INVALID, "no waterfall history configured (sdrpp_wf_configure)"); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->fft_stream) { HIPCHK(c, hipStreamSynchronize(c->fft_stream)); }
    int rc = wf_ensure_trace(c);
    if (rc) { return rc; }
    sdrpp_ctx::Wf& W = c->wf;
    W.hold_on = enabled != 0;
    if (W.hold_on && W.width > 0) {  // setFFTHold (:1153-1160)
        std::vector<float> init((size_t)W.width, -1000.0f);
        HIPCHK(c, hipMemcpy(W.d_hold, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    W.hold_speed = speed;
    return SDRPP_OK;
}

int sdrpp_wf_latest(sdrpp_ctx* c, float* latest, float* hold) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    sdrpp_ctx::Wf& W = c->wf;
    if (W.height <= 0 || W.width <= 0 || !W.d_latest) { return fail(c, SDRPP_ERR_INVALID, "no waterfall trace yet"); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (latest) { HIPCHK(c, hipMemcpy(latest, W.d_latest, (size_t)W.width * sizeof(float), hipMemcpyDeviceToHost)); }
    if (hold) { HIPCHK(c, hipMemcpy(hold, W.d_hold, (size_t)W.width * sizeof(float), hipMemcpyDeviceToHost)); }
    return W.width;
}

// updateWaterfallFb (waterfall.cpp:600-631): every stored line re-zoomed with a NEW view, newest first
int sdrpp_wf_raster(sdrpp_ctx* c, int draw_start, int draw_size, int data_width, float wf_min, float wf_max, int32_t* dst_host, int* n_lines) {
    DeviceScope dev_scope_(c);
    if (!c || !dst_host || data_width <= 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    sdrpp_ctx::Wf& W = c->wf;
    if (W.height <= 0) { return fail(c, SDRPP_ERR_INVALID, "no waterfall history configured (sdrpp_wf_configure)"); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->fft_stream) { HIPCHK(c, hipStreamSynchronize(c->fft_stream)); }
    const int count = std::min(W.lines, W.height);
    std::vector<int32_t> zs, zc;
    sdrpp_host::zoomTable(draw_start, draw_size, c->fft_size, data_width, zs, zc);
    int32_t *d_zs = nullptr, *d_zc = nullptr, *d_idx = nullptr;
    float* d_zm = nullptr;
    int rc = upload(c, &d_zs, zs.data(), zs.size());
    if (!rc) { rc = upload(c, &d_zc, zc.data(), zc.size()); }
    if (!rc) { rc = dev_alloc(c, &d_zm, (size_t)std::max(count, 1) * data_width); }
    if (!rc) { rc = dev_alloc(c, &d_idx, (size_t)std::max(count, 1) * data_width); }
    if (!rc && count > 0) {
        // display row i = ring slot (i + cur) mod H: two contiguous runs of slots
        const int first = std::min(count, W.height - W.cur);
        launch_zoom(c->stream, W.d_ring + (size_t)W.cur * c->fft_size, first, c->fft_size, draw_size, data_width, d_zs, d_zc, wf_min, wf_max, d_zm, d_idx);
        if (count > first) {
            launch_zoom(c->stream, W.d_ring, count - first, c->fft_size, draw_size, data_width, d_zs, d_zc, wf_min, wf_max, d_zm + (size_t)first * data_width, d_idx + (size_t)first * data_width);
        }
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) { e = hipMemcpy(dst_host, d_idx, (size_t)count * data_width * sizeof(int32_t), hipMemcpyDeviceToHost); }
        if (e != hipSuccess) { rc = fail(c, SDRPP_ERR_HIP, "waterfall raster failed: %s", hipGetErrorString(e)); }
    }
    dev_free(d_zs);
    dev_free(d_zc);
    dev_free(d_zm);
    dev_free(d_idx);
    if (rc) { return rc; }
    for (size_t i = (size_t)count * data_width; i < (size_t)W.height * data_width; i++) { dst_host[i] = -1; }  // (uint32_t)255 << 24 in the reference
    if (n_lines) { *n_lines = count; }
    return SDRPP_OK;
}

static void preproc_free(sdrpp_ctx* c) {
    sdrpp_ctx::Pre& P = c->pre;
    for (int i = 0; i < SDRPP_MAX_DECIM_STAGES; i++) {
        dev_free(P.d_staps[i]);
        toep_free(P.tp[i]);
    }
    P.raw.data = nullptr;  // the caller's buffer, never owned
    stream_free(P.raw);
    for (auto& s : P.st) { stream_free(s); }
    stream_free(P.out);
    dev_free(P.d_off);
    dev_free(P.d_seg);
    P = sdrpp_ctx::Pre{};
}

int sdrpp_destroy(sdrpp_ctx* c) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_OK; }
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    g_hostprof.report();
#ifdef SDRPP_TICK_TRACE
    if (const char* path = getenv("SDRPP_TICK_TRACE_FILE")) {
        unsigned n = 0;
        if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(sdrpp_k::g_tick_trace_n), sizeof(n)) == hipSuccess && n > 0) {
            n = std::min<unsigned>(n, SDRPP_TICK_TRACE_CAP);
            std::vector<sdrpp_k::TickTraceRec> recs(n);
            if (hipMemcpyFromSymbol(recs.data(), HIP_SYMBOL(sdrpp_k::g_tick_trace), (size_t)n * sizeof(sdrpp_k::TickTraceRec)) == hipSuccess) {
                if (FILE* f = fopen(path, "ab")) {
                    fwrite(recs.data(), sizeof(sdrpp_k::TickTraceRec), n, f);
                    fclose(f);
                }
            }
            const unsigned zero = 0;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(sdrpp_k::g_tick_trace_n), &zero, sizeof(zero));
        }
    }
#endif
#ifdef SDRPP_TOEP_PROF
    {
        unsigned long long h[4][8];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(sdrpp_k::g_toep_prof), sizeof(h)) == hipSuccess) {
            const char* names[4] = { "decimator", "resampler", "channel filter", "discriminator+audio" };
            for (int k = 0; k < 4; k++) {
                if (!h[k][5]) { continue; }
                const double r = (double)h[k][5];
                fprintf(stderr, "[sdrpp toep prof] %-20s rounds %llu, cycles per round: matrix %.0f | wait loads + regs->LDS %.0f | issue loads %.0f | discriminate %.0f | issue stores %.0f ; wavefront lifetime %.0f cycles, %.2f rounds per wavefront\n",
                        names[k], h[k][5], h[k][0] / r, h[k][1] / r, h[k][2] / r, h[k][3] / r, h[k][4] / r, (double)h[k][6] / (double)h[k][7], r / (double)h[k][7]);
            }
            unsigned long long z[4][8] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(sdrpp_k::g_toep_prof), z, sizeof(z));
        }
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(sdrpp_k::g_pipe_prof), sizeof(h)) == hipSuccess) {
            const char* names[4] = { "decimator", "resampler", "channel filter", "discriminator+audio" };
            for (int k = 0; k < 4; k++) {
                if (!h[k][5]) { continue; }
                const double r = (double)h[k][5];
                fprintf(stderr, "[sdrpp pipe prof] %-20s tiles %llu, cycles per tile: wait input %.0f | matrix %.0f | release / stage next %.0f | epilogue + wait space %.0f | write + publish %.0f ; wavefront lifetime %.0f cycles, %.2f tiles per wavefront\n",
                        names[k], h[k][5], h[k][0] / r, h[k][1] / r, h[k][2] / r, h[k][3] / r, h[k][4] / r, (double)h[k][6] / (double)h[k][7], r / (double)h[k][7]);
            }
            unsigned long long z[4][8] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(sdrpp_k::g_pipe_prof), z, sizeof(z));
            unsigned long long ck[4] = { 0, 0, 0, 0 };
            if (hipMemcpyFromSymbol(ck, HIP_SYMBOL(sdrpp_k::g_pipe_clock), sizeof(ck)) == hipSuccess && ck[1]) {
                fprintf(stderr, "[sdrpp pipe prof] shader clock while the pipelined kernel ran: %.0f MHz (s_memtime cycles per 100 MHz s_memrealtime tick); wavefront lifetime min %llu max %llu cycles (all launches)\n",
                        100.0 * (double)ck[0] / (double)ck[1], ck[2], ck[3]);
            }
        }
    }
#endif
    preproc_free(c);
    wf_free(c);
    dev_free(c->d_pack);
    dev_free(c->d_gather);
    dev_free(c->d_gather_jobs);
    for (auto& kv : c->vfos) { vfo_free(*kv.second); }
    c->vfos.clear();
    for (auto& e : c->s1_tap_cache) { (void)hipFree(e.second); }
    for (auto& p : c->tpairs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : c->ev_pool) { (void)hipEventDestroy(e); }
    for (int i = 0; i < kArenaSlots; i++) {
        if (c->arena_host[i]) { (void)hipHostFree(c->arena_host[i]); }
        if (c->arena_ev[i]) { (void)hipEventDestroy(c->arena_ev[i]); }
    }
    for (int i = 0; i < kArenaSlots; i++) { dev_free(c->arena_dev_slot[i]); }
    c->arena_dev = nullptr;
    dev_free(c->d_tick_counter);
    dev_free(c->empty_tab);
    if (c->h_tick_flag) { (void)hipHostFree(c->h_tick_flag); }
    for (int i = 0; i < 3; i++) { dev_free(c->tick_land[i]); }
    for (int i = 0; i < kStageSlots; i++) {
        if (c->stage_host[i]) { (void)hipHostFree(c->stage_host[i]); }
    }
    if (c->bank_plan && c->bank_plan_free) { c->bank_plan_free(c->bank_plan); }
    c->bank_plan = nullptr;
    for (int i = 0; i < sdrpp_ctx::kTickEvents; i++) {
        if (c->tick_ev[i]) { (void)hipEventDestroy(c->tick_ev[i]); }
        if (c->tick_ev_start[i]) { (void)hipEventDestroy(c->tick_ev_start[i]); }
    }
    if (c->res_ring) { (void)hipHostFree(c->res_ring); }
    for (auto& q : c->res_retired) { (void)hipHostFree(q.ring); }
    c->res_retired.clear();
    fft_ring_drop(c);
    for (int i = 0; i < 2; i++) {
        dev_free(c->iq_land[i]);
        dev_free(c->iq_land16[i]);
        if (c->land_ev[i]) { (void)hipEventDestroy(c->land_ev[i]); }
    }
    if (c->ev_copy) { (void)hipEventDestroy(c->ev_copy); }
    if (c->copy_stream) { (void)hipStreamDestroy(c->copy_stream); }
    if (c->side_stream) { (void)hipStreamDestroy(c->side_stream); }
    dev_free(c->iq_hist[0]);
    dev_free(c->iq_hist[1]);
    dev_free(c->d_window);
    dev_free(c->d_tw1);
    dev_free(c->d_tw2);
    dev_free(c->d_twn);
    dev_free(c->d_scratch);
    dev_free(c->d_lines);
    dev_free(c->d_lines_grp);
    dev_free(c->d_zstart);
    dev_free(c->d_zcount);
    dev_free(c->d_zoomed);
    dev_free(c->d_index);
    if (c->ev_fork) { (void)hipEventDestroy(c->ev_fork); }
    if (c->ev_join) { (void)hipEventDestroy(c->ev_join); }
    if (c->fft_stream) { (void)hipStreamDestroy(c->fft_stream); }
    if (c->own_stream) { (void)hipStreamDestroy(c->own_stream); }
    delete c;
    return SDRPP_OK;
}

const char* sdrpp_last_error(const sdrpp_ctx* c) { return c ? c->err.c_str() : "null context"; }

// A pipelined back-end launch whose wavefronts gave up waiting for each other (never seen; a hang would be worse) counted that in THIS context's
// page-locked word.  Called wherever the host has just synchronised with the stream and is about to hand out results.
static int pipe_timeouts_check(sdrpp_ctx* c) {
    if (!c->pipe_launched || !c->h_tick_flag) { return SDRPP_OK; }
    c->pipe_launched = false;
    const int n = *(const volatile int*)(c->h_tick_flag + 8);
    if (n != c->timeouts_seen) {
        const int d = n - c->timeouts_seen;
        c->timeouts_seen = n;
        return fail(c, SDRPP_ERR_HIP, "pipelined back end: %d wavefront waits timed out (results of the last pushes are invalid)", d);
    }
    return SDRPP_OK;
}

int sdrpp_set_stream(sdrpp_ctx* c, void* s) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    c->launch_stream = c->stream;
    return SDRPP_OK;
}

int sdrpp_sync(sdrpp_ctx* c) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int rc = pipe_timeouts_check(c)) { return rc; }
    return SDRPP_OK;
}

int sdrpp_device_info(sdrpp_ctx* c, char* buf, int buflen) {
    DeviceScope dev_scope_(c);
    if (!c || !buf || buflen <= 0) { return SDRPP_ERR_INVALID; }
    snprintf(buf, (size_t)buflen, "%s", c->devinfo.c_str());
    return SDRPP_OK;
}

// ---- FFT ---------------------------------------------------------------------------------------------------------------------
int sdrpp_fft_configure(sdrpp_ctx* c, int fft_size, int nz, int skip, const float* window) {
    DeviceScope dev_scope_(c);
    if (!c || !window) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (!is_pow2(fft_size) || fft_size < 1024 || fft_size > (1 << 20)) { return fail(c, SDRPP_ERR_UNSUPPORTED, "fft_size %d: need a power of two in [1024, 1048576]", fft_size); }
    if (nz <= 0 || nz > fft_size || skip < 0) { return fail(c, SDRPP_ERR_INVALID, "bad framing nz=%d skip=%d", nz, skip); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->fft_stream) { HIPCHK(c, hipStreamSynchronize(c->fft_stream)); }
    if (c->wf.height > 0 && fft_size != c->fft_size) { wf_free(c); }  // the line history holds lines of the old size (setRawFFTSize reallocates rawFFTs)
    c->fft_on = false;
    const int m = ilog2(fft_size);
    int rc = upload(c, &c->d_window, window, (size_t)nz);
    if (rc) { return rc; }
    auto half_table = [](int L) {
        std::vector<float2> t((size_t)std::max(L / 2, 1));
        for (int e = 0; e < L / 2; e++) { sdrpp_host::twiddle(e, L, &t[(size_t)e].x, &t[(size_t)e].y); }
        return t;
    };
    fft_ring_drop(c);
    dev_free(c->d_tw2);
    dev_free(c->d_twn);
    dev_free(c->d_scratch);
    if (m <= 12) {
        auto t = half_table(fft_size);
        rc = upload(c, &c->d_tw1, t.data(), t.size());
        if (rc) { return rc; }
    }
    else {
        int lg1, lg2;
        fft_split(m, &lg1, &lg2);
        const int N1 = 1 << lg1, N2 = 1 << lg2;
        auto t1 = half_table(N1);
        auto t2 = half_table(N2);
        rc = upload(c, &c->d_tw1, t1.data(), t1.size());
        if (rc) { return rc; }
        rc = upload(c, &c->d_tw2, t2.data(), t2.size());
        if (rc) { return rc; }
        std::vector<float2> full((size_t)fft_size);
        for (int e = 0; e < fft_size; e++) { sdrpp_host::twiddle(e, fft_size, &full[(size_t)e].x, &full[(size_t)e].y); }
        std::vector<float2> tn((size_t)fft_size);
        for (int k1 = 0; k1 < N1; k1++) {
            for (int n2 = 0; n2 < N2; n2++) { tn[(size_t)k1 * N2 + n2] = full[(size_t)k1 * n2]; }
        }
        rc = upload(c, &c->d_twn, tn.data(), tn.size());
        if (rc) { return rc; }
        const size_t per_chunk = std::max<size_t>(1, kScratchBytes / ((size_t)fft_size * sizeof(float2)));
        rc = dev_alloc(c, &c->d_scratch, per_chunk * (size_t)fft_size);
        if (rc) { return rc; }
    }
    const size_t lines = (size_t)(c->max_push / ((int64_t)nz + skip)) + 2;
    dev_free(c->d_lines);
    rc = dev_alloc(c, &c->d_lines, lines * (size_t)fft_size);
    if (rc) { return rc; }
    dev_free(c->d_lines_grp);
    c->d_lines_grp = nullptr;
    c->zoom_grp = (m > 16) ? kZoomGrpLong : ((m > 12) ? pass2_rows(m - m / 2) : 0);
    if (c->zoom_grp) {
        rc = dev_alloc(c, &c->d_lines_grp, lines * (size_t)(fft_size / c->zoom_grp));
        if (rc) { return rc; }
    }
    c->lines_cap = lines;
    c->fft_size = fft_size;
    c->fft_lg = m;
    c->nz = nz;
    c->skip = skip;
    c->fft_pos = 0;
    c->fft_next = 0;
    c->n_lines = 0;
    c->fft_on = true;
    if (c->data_width > 0) { return sdrpp_fft_set_view(c, c->view_start, c->view_size, c->data_width, c->wf_min, c->wf_max); }
    return SDRPP_OK;
}

int sdrpp_fft_disable(sdrpp_ctx* c) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->fft_on = false;
    c->n_lines = 0;
    return SDRPP_OK;
}

int sdrpp_fft_set_view(sdrpp_ctx* c, int start, int size, int data_width, float wf_min, float wf_max) {
    DeviceScope dev_scope_(c);
    if (!c || data_width < 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->view_start = start;
    c->view_size = size;
    c->data_width = data_width;
    c->wf_min = wf_min;
    c->wf_max = wf_max;
    c->zoom_cap = 0;
    fft_ring_drop(c);
    dev_free(c->d_zoomed);
    dev_free(c->d_index);
    if (data_width == 0 || c->fft_size == 0) { return SDRPP_OK; }
    std::vector<int32_t> zs, zc;
    sdrpp_host::zoomTable(start, size, c->fft_size, data_width, zs, zc);
    int rc = upload(c, &c->d_zstart, zs.data(), zs.size());
    if (rc) { return rc; }
    rc = upload(c, &c->d_zcount, zc.data(), zc.size());
    if (rc) { return rc; }
    c->h_zstart = zs;
    c->h_zcount = zc;
    c->zoom_tp_cache = 0;
    return ensure_zoom(c, c->lines_cap);
}

int sdrpp_fft_lines(sdrpp_ctx* c) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    {
        int frc = flush_pending_opt(c, 0);
        if (frc) { return frc; }
    }
    return c->n_lines;
}

int sdrpp_fft_read(sdrpp_ctx* c, int first, int n, float* raw, float* zoomed, int32_t* index) {
    DeviceScope dev_scope_(c);
    if (!c || first < 0 || n < 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (first >= c->n_lines) { return 0; }
    n = std::min(n, c->n_lines - first);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (raw) { HIPCHK(c, hipMemcpy(raw, c->d_lines + (size_t)first * c->fft_size, (size_t)n * c->fft_size * sizeof(float), hipMemcpyDeviceToHost)); }
    if (c->data_width > 0) {
        if (zoomed) { HIPCHK(c, hipMemcpy(zoomed, c->d_zoomed + (size_t)first * c->data_width, (size_t)n * c->data_width * sizeof(float), hipMemcpyDeviceToHost)); }
        if (index) { HIPCHK(c, hipMemcpy(index, c->d_index + (size_t)first * c->data_width, (size_t)n * c->data_width * sizeof(int32_t), hipMemcpyDeviceToHost)); }
    }
    else if (zoomed || index) {
        return fail(c, SDRPP_ERR_INVALID, "no view configured (sdrpp_fft_set_view)");
    }
    return n;
}

int sdrpp_fft_copy_device(sdrpp_ctx* c, int first, int n, float* raw, float* zoomed, int32_t* index) {
    DeviceScope dev_scope_(c);
    if (!c || first < 0 || n < 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (first >= c->n_lines) { return 0; }
    n = std::min(n, c->n_lines - first);
    if (raw) { HIPCHK(c, hipMemcpyAsync(raw, c->d_lines + (size_t)first * c->fft_size, (size_t)n * c->fft_size * sizeof(float), hipMemcpyDeviceToDevice, c->stream)); }
    if ((zoomed || index) && c->data_width <= 0) { return fail(c, SDRPP_ERR_INVALID, "no view configured (sdrpp_fft_set_view)"); }
    if (zoomed) { HIPCHK(c, hipMemcpyAsync(zoomed, c->d_zoomed + (size_t)first * c->data_width, (size_t)n * c->data_width * sizeof(float), hipMemcpyDeviceToDevice, c->stream)); }
    if (index) { HIPCHK(c, hipMemcpyAsync(index, c->d_index + (size_t)first * c->data_width, (size_t)n * c->data_width * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream)); }
    return n;
}

int sdrpp_fft_device_buffers(sdrpp_ctx* c, const float** raw, const float** zoomed, const int32_t** index, int* n_lines) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (raw) { *raw = c->d_lines; }
    if (zoomed) { *zoomed = c->d_zoomed; }
    if (index) { *index = c->d_index; }
    if (n_lines) { *n_lines = c->n_lines; }
    return SDRPP_OK;
}

int sdrpp_preproc_configure(sdrpp_ctx* c, int n_stages, const int* stage_decim, const int* stage_ntaps, const float* const* stage_taps, float dc_rate, int conjugate) {
    DeviceScope dev_scope_(c);
    if (!c || n_stages < 0 || n_stages > SDRPP_MAX_DECIM_STAGES) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    for (int s = 0; s < n_stages; s++) {
        if (!stage_decim || !stage_ntaps || !stage_taps || !is_pow2(stage_decim[s]) || stage_ntaps[s] <= 0 || !stage_taps[s]) {
            return fail(c, SDRPP_ERR_UNSUPPORTED, "pre-processing stage %d: decimation must be a power of two with taps", s);
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    preproc_free(c);
    if (n_stages == 0 && dc_rate == 0.0f && !conjugate) { return SDRPP_OK; }  // chain fully disabled: pushes go straight through
    sdrpp_ctx::Pre& P = c->pre;
    P.ref_order = c->pre_ref_order;
    P.n_stages = n_stages;
    P.dc_rate = dc_rate;
    P.conj = conjugate ? 1 : 0;
    int rc;
    size_t cap = (size_t)c->max_push;
    if (n_stages > 0) {
        P.raw.width = 2;
        P.raw.hist_len = stage_ntaps[0] - 1;
        for (int i = 0; i < 2; i++) {
            rc = dev_alloc(c, &P.raw.hist[i], (size_t)std::max(P.raw.hist_len, 1) * 2);
            if (rc) { return rc; }
            HIPCHK(c, hipMemset(P.raw.hist[i], 0, (size_t)std::max(P.raw.hist_len, 1) * 2 * sizeof(float)));
        }
    }
    P.st.resize((size_t)n_stages);
    for (int s = 0; s < n_stages; s++) {
        P.decim_s[s] = stage_decim[s];
        P.staps[s].assign(stage_taps[s], stage_taps[s] + stage_ntaps[s]);
        rc = upload_blocked(c, &P.d_staps[s], P.staps[s].data(), (int)P.staps[s].size(), P.decim_s[s], &P.s_kp[s]);
        if (rc) { return rc; }
        P.tp[s].kind = 1;
        rc = toep_build_fir(c, P.tp[s], P.staps[s].data(), (int)P.staps[s].size(), P.decim_s[s]);
        if (rc) { return rc; }
        cap = cap / (size_t)P.decim_s[s] + 2;
        rc = stream_alloc(c, P.st[(size_t)s], 2, (s + 1 < n_stages) ? stage_ntaps[s + 1] - 1 : 0, cap);
        if (rc) { return rc; }
    }
    if (dc_rate != 0.0f || conjugate) {
        rc = stream_alloc(c, P.out, 2, 0, cap);
        if (rc) { return rc; }
    }
    if (dc_rate != 0.0f) {
        rc = dev_alloc(c, &P.d_off, 2);
        if (rc) { return rc; }
        HIPCHK(c, hipMemset(P.d_off, 0, 2 * sizeof(float2)));
        P.state_cur = 0;
        P.seg_cap = (int)(cap / SDRPP_DEEMP_SEG) + 2;
        rc = dev_alloc(c, &P.d_seg, 2 * ((size_t)P.seg_cap + 1));
        if (rc) { return rc; }
    }
    P.on = true;
    return SDRPP_OK;
}

// IQFrontEnd's setters re-plan the chain but its blocks live on (iq_frontend.cpp:76-130): setSampleRate / setDCBlocking / setInvertIQ do not touch
// the decimator (its delay lines stay), and the DC blocker keeps its estimate through everything — setRate only swaps the rate, a blocker that
// is switched off and on again continues from where it was (dc_blocker.h: the object is only taken out of the chain).  setDecimation creates
// new decimator stages (PowerDecimator::setRatio -> reconfigure: power_decimator.h:91-108): cleared delay lines.
//   keep & 1: the decimator's delay lines and offsets, if the new description has the same stages;   keep & 2: the DC blocker's estimate
int sdrpp_preproc_reconfigure(sdrpp_ctx* c, int n_stages, const int* stage_decim, const int* stage_ntaps, const float* const* stage_taps, float dc_rate, int conjugate, int keep) {
    DeviceScope dev_scope_(c);
    if (!c || n_stages < 0 || n_stages > SDRPP_MAX_DECIM_STAGES || keep < 0 || keep > 3) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    sdrpp_ctx::Pre& P = c->pre;
    if (P.on && P.dc_rate != 0.0f && P.d_off) { HIPCHK(c, hipMemcpy(&c->pre_dc_last, P.d_off + P.state_cur, sizeof(float2), hipMemcpyDeviceToHost)); }
    bool same = (keep & 1) && P.on && n_stages > 0 && P.n_stages == n_stages && stage_decim && stage_ntaps && stage_taps;
    for (int s = 0; same && s < n_stages; s++) {
        same = P.decim_s[s] == stage_decim[s] && (int)P.staps[s].size() == stage_ntaps[s] && stage_taps[s] && memcmp(P.staps[s].data(), stage_taps[s], sizeof(float) * (size_t)stage_ntaps[s]) == 0;
    }
    sdrpp_ctx::Pre old;
    if (same) {  // set the decimator's streams (delay lines, ring buffers) and offsets aside; the rest of the old chain is freed by the configure below
        std::swap(old.raw, P.raw);
        std::swap(old.st, P.st);
        for (int s = 0; s < SDRPP_MAX_DECIM_STAGES; s++) { old.soff[s] = P.soff[s]; }
    }
    int rc = sdrpp_preproc_configure(c, n_stages, stage_decim, stage_ntaps, stage_taps, dc_rate, conjugate);
    if (same) {
        if (!rc && P.on) {
            std::swap(old.raw, P.raw);
            std::swap(old.st, P.st);
            for (int s = 0; s < SDRPP_MAX_DECIM_STAGES; s++) { P.soff[s] = old.soff[s]; }
        }
        old.raw.data = nullptr;
        stream_free(old.raw);
        for (auto& st : old.st) { stream_free(st); }
    }
    if (rc) { return rc; }
    if ((keep & 2) && P.on && P.dc_rate != 0.0f && P.d_off) { HIPCHK(c, hipMemcpy(P.d_off + P.state_cur, &c->pre_dc_last, sizeof(float2), hipMemcpyHostToDevice)); }
    if (!(keep & 2)) { c->pre_dc_last = make_float2(0.0f, 0.0f); }
    return SDRPP_OK;
}

int sdrpp_preproc_set_reference_order(sdrpp_ctx* c, int on) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    c->pre_ref_order = on != 0;
    c->pre.ref_order = c->pre_ref_order;
    return SDRPP_OK;
}

int sdrpp_preproc_out_count(sdrpp_ctx* c) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (!c->pre.on) { return fail(c, SDRPP_ERR_INVALID, "no pre-processing chain configured"); }
    return c->pre.last_n;
}

int sdrpp_preproc_read(sdrpp_ctx* c, float* dst, int max) {
    DeviceScope dev_scope_(c);
    if (!c || !dst || max < 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (!c->pre.on) { return fail(c, SDRPP_ERR_INVALID, "no pre-processing chain configured"); }
    const int n = std::min(max, c->pre.last_n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n > 0) { HIPCHK(c, hipMemcpy(dst, c->pre.last, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost)); }
    return n;
}

int sdrpp_preproc_device_buffer(sdrpp_ctx* c, const float** iq, int* n) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (!c->pre.on) { return fail(c, SDRPP_ERR_INVALID, "no pre-processing chain configured"); }
    if (iq) { *iq = c->pre.last; }
    if (n) { *n = c->pre.last_n; }
    return SDRPP_OK;
}

// ---- VFOs ----------------------------------------------------------------------------------------------------------------------
int sdrpp_vfo_add(sdrpp_ctx* c, const sdrpp_vfo_desc* d, int* id) {
    DeviceScope dev_scope_(c);
    if (!c || !d || !id) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (d->n_stages < 0 || d->n_stages > SDRPP_MAX_DECIM_STAGES) { return fail(c, SDRPP_ERR_INVALID, "n_stages %d", d->n_stages); }
    for (int s = 0; s < d->n_stages; s++) {
        if (!is_pow2(d->stage_decim[s]) || d->stage_ntaps[s] <= 0 || !d->stage_taps[s]) { return fail(c, SDRPP_ERR_UNSUPPORTED, "stage %d: decimation must be a power of two with taps", s); }
    }
    if (d->n_stages > 0) {  // the fused translation + FIR kernel uses the linear-phase pairing (all reference plans are symmetric)
        const float* h = d->stage_taps[0];
        for (int k = 0; k < d->stage_ntaps[0] / 2; k++) {
            if (h[k] != h[d->stage_ntaps[0] - 1 - k]) { return fail(c, SDRPP_ERR_UNSUPPORTED, "first decimation stage must have symmetric (linear-phase) taps"); }
        }
    }
    const bool has_poly = (d->interp != d->decim);
    if (has_poly && (d->interp <= 0 || d->decim <= 0 || d->resamp_ntaps <= 0 || !d->resamp_taps)) { return fail(c, SDRPP_ERR_INVALID, "bad polyphase description"); }
    if (d->chan_ntaps < 0 || d->chan_ntaps > kChanHistCap + 1) { return fail(c, SDRPP_ERR_UNSUPPORTED, "channel filter of %d taps (max %d)", d->chan_ntaps, kChanHistCap + 1); }
    if (d->demod < SDRPP_DEMOD_RAW || d->demod > SDRPP_DEMOD_DSB) { return fail(c, SDRPP_ERR_INVALID, "demod %d", d->demod); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // every early return below gives the device allocations made so far back (vfo_free); only a fully built VFO is handed to the context
    struct VfoFreer {
        void operator()(Vfo* p) const {
            if (p) {
                vfo_free(*p);
                delete p;
            }
        }
    };
    std::unique_ptr<Vfo, VfoFreer> v(new Vfo);
    v->id = c->next_id++;
    v->d = *d;
    if (d->nco_mode < 0 || d->nco_mode > 2) { return fail(c, SDRPP_ERR_INVALID, "nco_mode %d: 0 (context), 1 (closed form) or 2 (reference rotator)", d->nco_mode); }
    v->nco_exact = d->nco_mode == 0 ? (c->nco_exact != 0) : (d->nco_mode == 2);
    int rc;
    // capacities
    size_t cap = (size_t)c->max_push;
    auto add_stream = [&](int width, int hist, size_t capn) -> int {
        v->st.emplace_back();
        int r = stream_alloc(c, v->st.back(), width, hist, capn);
        return r ? -1 : (int)v->st.size() - 1;
    };
    // what consumes the decimator / rotator output
    const int tpp = has_poly ? (d->resamp_ntaps + d->interp - 1) / d->interp : 0;
    const bool fm = (d->demod == SDRPP_DEMOD_WFM || d->demod == SDRPP_DEMOD_NFM);
    const bool fm_mode = (d->demod == SDRPP_DEMOD_WFM || d->demod == SDRPP_DEMOD_NFM);
    const int if_hist = fm_mode ? std::max(d->audio_ntaps, 1) + 1 : 1;  // fused discriminator + audio FIR re-reads the IF history
    const int chan_hist = ((std::max(std::max(d->chan_ntaps - 1, 1), if_hist) + 63) / 64) * 64;  // grown on demand by sdrpp_vfo_set_channel_taps
    auto hist_after_decim = [&]() -> int {
        if (has_poly) { return tpp - 1; }
        return chan_hist;  // channel filter or the discriminator
    };
    for (int s = 0; s < d->n_stages; s++) {
        v->staps[s].assign(d->stage_taps[s], d->stage_taps[s] + d->stage_ntaps[s]);
        v->d.stage_taps[s] = nullptr;
        rc = upload_blocked(c, &v->d_staps[s], v->staps[s].data(), (int)v->staps[s].size(), d->stage_decim[s], &v->s_kp[s]);
        if (rc) { return rc; }
        rc = upload(c, &v->d_staps_nat[s], v->staps[s].data(), v->staps[s].size());
        if (rc) { return rc; }
        if (s >= 1 || v->nco_exact) {  // stage 0 runs as a plain FIR only behind the reference rotator
            v->tp_stage[s].kind = 1;
            rc = toep_build_fir(c, v->tp_stage[s], v->staps[s].data(), (int)v->staps[s].size(), d->stage_decim[s]);
            if (rc) { return rc; }
        }
        cap = cap / (size_t)d->stage_decim[s] + 2;
        const int hist = (s + 1 < d->n_stages) ? d->stage_ntaps[s + 1] - 1 : hist_after_decim();
        if (add_stream(2, hist, cap) < 0) { return SDRPP_ERR_NOMEM; }
    }
    if (d->n_stages == 0) {
        if (add_stream(2, hist_after_decim(), cap) < 0) { return SDRPP_ERR_NOMEM; }
    }
    v->i_first = 0;
    {   // the front end as one filter: fusion decision (geometry only), tap identity, composite taps for the retune hand-over
        unsigned long long hsh = 1469598103934665603ull;  // FNV-1a over the taps of stages 0 and 1
        for (int s = 0; s < std::min(d->n_stages, 2); s++) {
            for (float t : v->staps[s]) {
                unsigned u;
                memcpy(&u, &t, 4);
                hsh = (hsh ^ u) * 1099511628211ull;
            }
        }
        v->tap_hash = hsh;
        if (d->n_stages >= 2) {  // the composite forms pair taps k and K-1-k: stage 1 must be linear phase as well
            const std::vector<float>& h2 = v->staps[1];
            for (size_t k = 0; k < h2.size() / 2; k++) { v->no_fuse = v->no_fuse || (h2[k] != h2[h2.size() - 1 - k]); }
        }
        v->fused_front = !v->nco_exact && !v->no_fuse && d->n_stages >= 2 && front2_t2(d->stage_ntaps[0], d->stage_decim[0], d->stage_ntaps[1], d->stage_decim[1], 8) > 0;
        if (d->n_stages >= 1 && !v->nco_exact) {
            const int K0 = d->stage_ntaps[0], D1 = d->stage_decim[0], K2 = v->fused_front ? d->stage_ntaps[1] : 1;
            const int K = K0 + (K2 - 1) * D1;
            std::vector<double> h12((size_t)K, 0.0);
            for (int k2 = 0; k2 < K2; k2++) {
                const double w2 = v->fused_front ? (double)v->staps[1][(size_t)k2] : 1.0;
                for (int k1 = 0; k1 < K0; k1++) { h12[(size_t)k2 * D1 + k1] += w2 * (double)v->staps[0][(size_t)k1]; }
            }
            std::vector<float> hf(h12.begin(), h12.end());
            rc = upload(c, &v->d_h12, hf.data(), hf.size());
            if (rc) { return rc; }
            v->h12_K = K;
            v->h12_lgD = ilog2(D1) + (v->fused_front ? ilog2(d->stage_decim[1]) : 0);
        }
        if (v->nco_exact && d->n_stages >= 1) {  // reference-rotator mode: the rotated full-rate stream feeds stage 0
            v->i_rot = add_stream(2, d->stage_ntaps[0] - 1, (size_t)c->max_push);
            if (v->i_rot < 0) { return SDRPP_ERR_NOMEM; }
        }
    }
    if (has_poly) {
        v->rtaps.assign(d->resamp_taps, d->resamp_taps + d->resamp_ntaps);
        v->d.resamp_taps = nullptr;
        v->tpp = tpp;
        std::vector<float> bank((size_t)d->interp * tpp, 0.0f);
        const int tot = d->interp * tpp;
        for (int i = 0; i < tot; i++) { bank[(size_t)((d->interp - 1) - (i % d->interp)) * tpp + (size_t)(i / d->interp)] = (i < d->resamp_ntaps) ? v->rtaps[(size_t)i] : 0.0f; }  // polyphase_bank.h:31-34
        rc = upload(c, &v->d_bank, bank.data(), bank.size());
        if (rc) { return rc; }
        v->tp_poly.kind = 2;
        rc = toep_build_poly(c, v->tp_poly, bank, d->interp, d->decim, tpp);
        if (rc) { return rc; }
        if (d->interp <= 8) {  // register-blocked kernel: per carried phase, taps of one full phase cycle
            const int L = d->interp, M = d->decim, lmax = (L <= 4) ? 4 : 8, rows = tpp + M;
            std::vector<float> cyc((size_t)L * rows * lmax, 0.0f);
            for (int ph0 = 0; ph0 < L; ph0++) {
                for (int r = 0; r < L; r++) {
                    const int A = ph0 + r * M, ph = A % L, o = A / L;
                    for (int k = 0; k < tpp; k++) { cyc[((size_t)ph0 * rows + (size_t)(k + o)) * lmax + r] = bank[(size_t)ph * tpp + k]; }
                }
            }
            rc = upload(c, &v->d_cyc, cyc.data(), cyc.size());
            if (rc) { return rc; }
            v->cyc_rows = rows;
            v->cyc_lmax = lmax;
        }
        cap = cap * (size_t)d->interp / (size_t)d->decim + 4;
        v->i_poly = add_stream(2, chan_hist, cap);
        if (v->i_poly < 0) { return SDRPP_ERR_NOMEM; }
    }
    // channel-filter output stream always exists (taps may be enabled later); its consumer is the demodulator
    v->i_chan = add_stream(2, if_hist, cap);
    if (v->i_chan < 0) { return SDRPP_ERR_NOMEM; }
    if (d->chan_ntaps > 0) {
        if (!d->chan_taps) { return fail(c, SDRPP_ERR_INVALID, "chan_taps null"); }
        v->ctaps_chan.assign(d->chan_taps, d->chan_taps + d->chan_ntaps);
        rc = upload_blocked(c, &v->d_chan, v->ctaps_chan.data(), (int)v->ctaps_chan.size(), 1, &v->chan_kp);
        if (rc) { return rc; }
        v->tp_chan.kind = 4;
        rc = toep_build_fir(c, v->tp_chan, v->ctaps_chan.data(), (int)v->ctaps_chan.size(), 1);
        if (rc) { return rc; }
        v->chan_ntaps = d->chan_ntaps;
    }
    v->d.chan_taps = nullptr;
    if (d->demod != SDRPP_DEMOD_RAW) {
        if (fm || d->demod == SDRPP_DEMOD_AM) {
            static const float unit = 1.0f;  // fm.h:165-168 loadDummyTaps: a single unit tap when the low-pass is off
            const float* at = d->audio_ntaps > 0 ? d->audio_taps : &unit;
            const int an = d->audio_ntaps > 0 ? d->audio_ntaps : 1;
            if (d->audio_ntaps > 0 && !d->audio_taps) { return fail(c, SDRPP_ERR_INVALID, "audio_taps null"); }
            v->ataps.assign(at, at + an);
            v->audio_ntaps = an;
            rc = upload_blocked(c, &v->d_audio, v->ataps.data(), (int)v->ataps.size(), 1, &v->audio_kp);
            if (rc) { return rc; }
            v->tp_audio.kind = 8;
            rc = toep_build_fir(c, v->tp_audio, v->ataps.data(), (int)v->ataps.size(), 1);
            if (rc) { return rc; }
            if (!fm) {  // AM: the sequential envelope/AGC kernel writes a real stream for the low-pass; FM demodulates inside the FIR kernel
                v->i_dem = add_stream(1, std::max(an - 1, 1), cap);
                if (v->i_dem < 0) { return SDRPP_ERR_NOMEM; }
            }
        }
        if (d->demod >= SDRPP_DEMOD_USB) {  // SSB: real scratch between the parallel translation and the sequential AGC
            v->i_dem = add_stream(1, 0, cap);
            if (v->i_dem < 0) { return SDRPP_ERR_NOMEM; }
        }
        v->i_out = add_stream(2, 0, cap);
        if (v->i_out < 0) { return SDRPP_ERR_NOMEM; }
    }
    v->d.audio_taps = nullptr;
    rc = dev_alloc(c, &v->d_state, 2 * sizeof(AgcState) + sizeof(float));
    if (rc) { return rc; }
    rc = dev_alloc(c, &v->d_rot, 2);
    if (rc) { return rc; }
    v->theta = sdrpp_host::turnsPerSample(d->phase_delta_re, d->phase_delta_im);
    v->theta2 = sdrpp_host::turnsPerSample(d->ssb_phase_delta_re, d->ssb_phase_delta_im);
    if (d->demod < SDRPP_DEMOD_USB) { v->theta2 = 0.0; }
    v->modtaps_dirty = true;
    rc = vfo_reset_state(c, *v);
    if (rc) { return rc; }
    const int vid = v->id;  // (the right-hand side of the assignment below is evaluated first)
    *id = vid;
    c->vfos[vid] = std::unique_ptr<Vfo>(v.release());
    vfo_list_rebuild(c);
    return SDRPP_OK;
}

int sdrpp_vfo_remove(sdrpp_ctx* c, int id) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    vfo_free(*it->second);
    c->vfos.erase(it);
    vfo_list_rebuild(c);
    return SDRPP_OK;
}

int sdrpp_vfo_count(sdrpp_ctx* c) { return c ? (int)c->vfos.size() : SDRPP_ERR_INVALID; }

// RxVFO::setInSamplerate / setOutSamplerate (rx_vfo.h:35-58): the channeliser is re-planned, but not everything starts over.  The reference keeps
//   * the translation's phase (FrequencyXlator::setOffset only swaps phaseDelta, frequency_xlator.h:24-30) and
//   * the channel filter's delay line (setOutSamplerate: FIR::setTaps moves it under the new tap count, fir.h:31-52; setInSamplerate does not touch the
//     filter at all; a filter that is bypassed under the new settings keeps what it held, one that wakes up continues from that)          -> keep bit 0
// while its decimator stages are new objects and the polyphase resampler is reset (power_decimator.h:91-108, polyphase_resampler.h:38-67), and
//   * the demodulator behind it is a separate block that setInSamplerate leaves alone: discriminator / audio low-pass history, AGC and DC-blocker
//     states, SSB's second translation                                                                                                     -> keep bit 1
// (a demodulator SWITCH deletes and creates it, radio_module.h:419-563: bit 1 off).  The AF chain is re-attached by the caller and starts cleared.
static int hist_tail_copy(sdrpp_ctx* c, Stream& to, const Stream& from, int max_samples) {
    if (!to.hist[to.cur] || !from.hist[from.cur] || to.width != from.width) { return SDRPP_OK; }
    const int H = std::min(std::min(from.hist_len, to.hist_len), max_samples);
    if (H <= 0) { return SDRPP_OK; }
    const size_t w = (size_t)to.width;
    HIPCHK(c, hipMemcpy(to.hist[to.cur] + (size_t)(to.hist_len - H) * w, from.hist[from.cur] + (size_t)(from.hist_len - H) * w, (size_t)H * w * sizeof(float), hipMemcpyDeviceToDevice));
    return SDRPP_OK;
}
int sdrpp_vfo_replace(sdrpp_ctx* c, int old_id, const sdrpp_vfo_desc* d, int keep, int* new_id) {
    DeviceScope dev_scope_(c);
    if (!c || !d || !new_id || keep < 0 || keep > 3) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (c->vfos.find(old_id) == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", old_id); }
    int nid = 0;
    int rc = sdrpp_vfo_add(c, d, &nid);
    if (rc) { return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    Vfo& o = *c->vfos[old_id];
    Vfo& n = *c->vfos[nid];
    auto feed_idx = [](const Vfo& v) { return (v.i_poly >= 0) ? v.i_poly : v.i_first + std::max(v.d.n_stages, 1) - 1; };  // the stream the channel filter reads
    Stream& of = o.st[(size_t)feed_idx(o)];
    Stream& nf = n.st[(size_t)feed_idx(n)];
    if (keep & 1) {
        if (o.nco_exact == n.nco_exact) {
            n.phi = o.phi;
            if (o.d_rot && n.d_rot) { HIPCHK(c, hipMemcpy(n.d_rot, o.d_rot, sizeof(float2), hipMemcpyDeviceToDevice)); }
        }
        // the channel filter's delay line as the reference's FIR object holds it now: the newest old_taps - 1 samples it was fed, or what it held
        // when it was last bypassed
        const int w = of.width;
        std::vector<float> line;
        if (o.chan_ntaps > 1 && of.hist[of.cur] && of.hist_len >= o.chan_ntaps - 1) {
            line.resize((size_t)(o.chan_ntaps - 1) * (size_t)w);
            HIPCHK(c, hipMemcpy(line.data(), of.hist[of.cur] + (size_t)(of.hist_len - (o.chan_ntaps - 1)) * w, line.size() * sizeof(float), hipMemcpyDeviceToHost));
        }
        else if (o.chan_ntaps == 0) { line = o.chan_stale; }
        if (n.chan_ntaps > 0 && nf.width == w) {  // FIR::setTaps: the newest min(old, new) - 1 samples stay, zeros in front of them
            if (nf.hist[nf.cur]) {
                HIPCHK(c, hipMemset(nf.hist[nf.cur], 0, (size_t)nf.hist_len * (size_t)w * sizeof(float)));
                const int have = (int)(line.size() / (size_t)w), m = std::min(have, std::min(n.chan_ntaps - 1, nf.hist_len));
                if (m > 0) { HIPCHK(c, hipMemcpy(nf.hist[nf.cur] + (size_t)(nf.hist_len - m) * w, line.data() + (size_t)(have - m) * w, (size_t)m * w * sizeof(float), hipMemcpyHostToDevice)); }
            }
        }
        else if (n.chan_ntaps == 0) { n.chan_stale = line; }  // bypassed under the new settings: the filter object keeps what it held
    }
    if ((keep & 2) && o.d.demod == n.d.demod) {
        // the demodulator's view of the IF stream: the discriminator's previous sample and the audio low-pass's delay line are its newest samples
        Stream& oif = (o.chan_ntaps > 0 && o.i_chan >= 0) ? o.st[(size_t)o.i_chan] : of;
        Stream& nif = (n.chan_ntaps > 0 && n.i_chan >= 0) ? n.st[(size_t)n.i_chan] : nf;
        const int if_need = (n.d.demod == SDRPP_DEMOD_WFM || n.d.demod == SDRPP_DEMOD_NFM) ? std::max(n.audio_ntaps, 1) + 1 : 1;
        rc = hist_tail_copy(c, nif, oif, if_need);
        if (rc) { return rc; }
        if (o.i_dem >= 0 && n.i_dem >= 0) {
            rc = hist_tail_copy(c, n.st[(size_t)n.i_dem], o.st[(size_t)o.i_dem], 1 << 30);
            if (rc) { return rc; }
        }
        if (o.d_state && n.d_state) { HIPCHK(c, hipMemcpy(n.d_state, o.d_state, 2 * sizeof(AgcState) + sizeof(float), hipMemcpyDeviceToDevice)); }
        n.phi2 = o.phi2;
        if (o.d_rot && n.d_rot) { HIPCHK(c, hipMemcpy(n.d_rot + 1, o.d_rot + 1, sizeof(float2), hipMemcpyDeviceToDevice)); }
    }
    rc = sdrpp_vfo_remove(c, old_id);
    if (rc) { return rc; }
    *new_id = nid;
    return SDRPP_OK;
}

int sdrpp_vfo_set_phase_delta(sdrpp_ctx* c, int id, float re, float im) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Vfo& v = *it->second;
    const double th = sdrpp_host::turnsPerSample(re, im);
    // the samples already in the first decimator's delay line stay rotated with the old increment (rx_vfo.h:72-77 only swaps
    // phaseDelta): remember where it changed so the first outputs of the next pushes can be handed over exactly (do_vfos)
    if (!v.nco_exact && v.d.n_stages > 0 && th != v.theta) {
        if (!v.recs.empty() && v.recs.back().pos == v.seen) { /* retuned twice between pushes: the older increment stays the one before */ }
        else { v.recs.push_back(Vfo::Retune{ v.seen, v.theta }); }
    }
    v.d.phase_delta_re = re;
    v.d.phase_delta_im = im;
    v.theta = th;
    v.modtaps_dirty = true;
    return SDRPP_OK;
}

int sdrpp_vfo_set_channel_taps(sdrpp_ctx* c, int id, const float* taps, int n) {
    DeviceScope dev_scope_(c);
    if (!c || n < 0 || n > kChanHistCap + 1 || (n > 0 && !taps)) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Vfo& v = *it->second;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {   // the stream that feeds the channel filter must remember n-1 samples
        const int idx = (v.i_poly >= 0) ? v.i_poly : v.i_first + std::max(v.d.n_stages, 1) - 1;
        Stream& fs = v.st[(size_t)idx];
        const int old_n = v.chan_ntaps, w = fs.width;
        const bool was_on = old_n > 0, now_on = n > 0;
        // ---- the filter's BYPASS switched (bandwidth == IF rate exactly, rx_vfo.h:60-70 / :89-100) ----
        // (a) The consumers behind the filter read "the IF stream" with memory (the demodulator's audio low-pass): their delay line holds the last
        //     samples they were FED — the filter's outputs until now, its input from now on, or the other way round.  The IF stream changes its
        //     identity here (st[i_chan] <-> the filter's input stream), so the newest history goes with it.
        // (b) The reference does not touch a bypassed filter: its delay line keeps what it held when it last ran, and a filter switched on again
        //     continues from that stale content (FIR::setTaps moves it like any other change of the tap count, fir.h:31-52).  The stream's side
        //     buffer here keeps being refreshed for consumer (a), so the filter's own last history is set aside when it goes to sleep and put
        //     back — under the new tap count — when it wakes up.
        if (was_on != now_on && v.i_chan >= 0) {
            Stream& cs = v.st[(size_t)v.i_chan];
            if (was_on) {  // going to sleep: (b) first, (a) overwrites the buffer
                // exactly the old filter's old_n - 1 samples, whatever an EARLIER bypass left here (a one-tap filter has no delay line: nothing; a
                // stream without history yet: zeros, which is what FIR's cleared buffer holds, fir.h:24-26)
                v.chan_stale.assign((size_t)std::max(old_n - 1, 0) * (size_t)w, 0.0f);
                if (old_n > 1 && fs.hist[fs.cur] && fs.hist_len >= old_n - 1) {
                    HIPCHK(c, hipMemcpy(v.chan_stale.data(), fs.hist[fs.cur] + (size_t)(fs.hist_len - (old_n - 1)) * w, v.chan_stale.size() * sizeof(float), hipMemcpyDeviceToHost));
                }
            }
            if (now_on) {  // waking up: the buffer must be long enough for the new filter before anything is put into it
                int rc = stream_grow_hist(c, fs, n - 1);
                if (rc) { return rc; }
            }
            Stream& from = was_on ? cs : fs;
            Stream& to = was_on ? fs : cs;
            const int H = std::min(from.hist_len, to.hist_len);
            if (H > 0 && from.hist[from.cur] && to.hist[to.cur] && from.width == to.width) {
                HIPCHK(c, hipMemcpy(to.hist[to.cur] + (size_t)(to.hist_len - H) * w, from.hist[from.cur] + (size_t)(from.hist_len - H) * w, (size_t)H * w * sizeof(float), hipMemcpyDeviceToDevice));
            }
            if (now_on && fs.hist[fs.cur]) {  // (b): zeros, then the newest part of what the filter held when it last ran (all zeros for one that never ran)
                HIPCHK(c, hipMemset(fs.hist[fs.cur], 0, (size_t)fs.hist_len * (size_t)w * sizeof(float)));
                const int have = (int)(v.chan_stale.size() / (size_t)w), m = std::min(have, std::min(n - 1, fs.hist_len));
                if (m > 0) {
                    HIPCHK(c, hipMemcpy(fs.hist[fs.cur] + (size_t)(fs.hist_len - m) * w, v.chan_stale.data() + (size_t)(have - m) * w, (size_t)m * w * sizeof(float), hipMemcpyHostToDevice));
                }
            }
        }
        else {
            int rc = stream_grow_hist(c, fs, n - 1);
            if (rc) { return rc; }
            // FIR::setTaps (dsp/filter/fir.h:31-52): a LONGER filter starts with zeros in front of the old delay line — its old_n - 1 samples are
            // all the reference kept, whatever the stream held before them.  The side buffer here holds the stream's true tail (newest last):
            // everything older than the old filter's reach is cleared.
            const int keep = std::max(old_n - 1, 0);
            if (n > old_n && fs.hist_len > keep && fs.hist[fs.cur]) {
                HIPCHK(c, hipMemset(fs.hist[fs.cur], 0, (size_t)(fs.hist_len - keep) * (size_t)w * sizeof(float)));
            }
        }
    }
    v.ctaps_chan.assign(taps, taps + n);
    v.chan_ntaps = n;
    v.d.chan_ntaps = n;
    if (n > 0) {
        int rc = upload_blocked(c, &v.d_chan, v.ctaps_chan.data(), n, 1, &v.chan_kp);
        if (rc) { return rc; }
        v.tp_chan.kind = 4;
        rc = toep_build_fir(c, v.tp_chan, v.ctaps_chan.data(), n, 1);
        if (rc) { return rc; }
    }
    return SDRPP_OK;
}

static void af_detach(Vfo& v) {
    Vfo::Af& a = v.af;
    for (int i = 0; i < SDRPP_MAX_DECIM_STAGES; i++) {
        dev_free(a.d_staps[i]);
        toep_free(a.tp_stage[i]);
    }
    toep_free(a.tp_poly);
    toep_free(a.tp_hpf);
    dev_free(a.d_bank);
    dev_free(a.d_hpf);
    dev_free(a.d_last);
    dev_free(a.d_seg);
    if (a.base >= 0) {
        for (size_t i = (size_t)a.base; i < v.st.size(); i++) { stream_free(v.st[i]); }
        v.st.resize((size_t)a.base);
    }
    a = Vfo::Af{};
}

int sdrpp_vfo_set_af(sdrpp_ctx* c, int id, const sdrpp_af_desc* af) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Vfo& v = *it->second;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    af_detach(v);
    if (!af) { return SDRPP_OK; }
    if (v.d.demod == SDRPP_DEMOD_RAW || v.i_out < 0) { return fail(c, SDRPP_ERR_UNSUPPORTED, "the AF chain needs a demodulating VFO"); }
    if (af->n_stages < 0 || af->n_stages > SDRPP_MAX_DECIM_STAGES) { return fail(c, SDRPP_ERR_INVALID, "af n_stages %d", af->n_stages); }
    for (int s = 0; s < af->n_stages; s++) {
        if (!is_pow2(af->stage_decim[s]) || af->stage_ntaps[s] <= 0 || !af->stage_taps[s]) { return fail(c, SDRPP_ERR_UNSUPPORTED, "af stage %d: decimation must be a power of two with taps", s); }
    }
    const bool has_poly = af->interp != af->decim;
    if (has_poly && (af->interp <= 0 || af->decim <= 0 || af->resamp_ntaps <= 0 || !af->resamp_taps)) { return fail(c, SDRPP_ERR_INVALID, "bad af polyphase description"); }
    if (af->hpf_ntaps < 0 || af->hpf_ntaps > kChanHistCap + 1 || (af->hpf_ntaps > 0 && !af->hpf_taps)) { return fail(c, SDRPP_ERR_INVALID, "bad af high-pass description"); }
    Vfo::Af& a = v.af;
    a.base = (int)v.st.size();
    a.n_stages = af->n_stages;
    a.interp = has_poly ? af->interp : 1;
    a.decim = has_poly ? af->decim : 1;
    a.tpp = has_poly ? (af->resamp_ntaps + af->interp - 1) / af->interp : 0;
    a.alpha = af->deemph_alpha;
    int rc;
    // history a stream must keep = (taps - 1) of its consumer; `stage` = first block that can be the consumer
    // (0..n_stages-1 decimators, n_stages polyphase, n_stages+1 high-pass; de-emphasis needs none)
    auto need_of = [&](int stage) -> int {
        if (stage < a.n_stages) { return af->stage_ntaps[stage] - 1; }
        if (stage <= a.n_stages && has_poly) { return a.tpp - 1; }
        if (stage <= a.n_stages + 1 && af->hpf_ntaps > 0) { return af->hpf_ntaps - 1; }
        return 0;
    };
    rc = stream_grow_hist(c, v.st[(size_t)v.i_out], need_of(0));
    if (rc) { return rc; }
    size_t cap = v.st[(size_t)v.i_out].cap;
    auto add_stream = [&](int hist, size_t capn) -> int {
        v.st.emplace_back();
        int r = stream_alloc(c, v.st.back(), 2, hist, capn);
        return r ? -1 : (int)v.st.size() - 1;
    };
    for (int s = 0; s < a.n_stages; s++) {
        a.decim_s[s] = af->stage_decim[s];
        a.staps[s].assign(af->stage_taps[s], af->stage_taps[s] + af->stage_ntaps[s]);
        rc = upload_blocked(c, &a.d_staps[s], a.staps[s].data(), (int)a.staps[s].size(), a.decim_s[s], &a.s_kp[s]);
        if (rc) { return rc; }
        a.tp_stage[s].kind = 1;
        rc = toep_build_fir(c, a.tp_stage[s], a.staps[s].data(), (int)a.staps[s].size(), a.decim_s[s]);
        if (rc) { return rc; }
        cap = cap / (size_t)a.decim_s[s] + 2;
        const int idx = add_stream(need_of(s + 1), cap);
        if (idx < 0) { return SDRPP_ERR_NOMEM; }
        if (s == 0) { a.i_stage0 = idx; }
    }
    if (has_poly) {
        a.rtaps.assign(af->resamp_taps, af->resamp_taps + af->resamp_ntaps);
        std::vector<float> bank((size_t)a.interp * a.tpp, 0.0f);
        const int tot = a.interp * a.tpp;
        for (int i = 0; i < tot; i++) { bank[(size_t)((a.interp - 1) - (i % a.interp)) * a.tpp + (size_t)(i / a.interp)] = (i < af->resamp_ntaps) ? a.rtaps[(size_t)i] : 0.0f; }  // polyphase_bank.h:31-34
        rc = upload(c, &a.d_bank, bank.data(), bank.size());
        if (rc) { return rc; }
        a.tp_poly.kind = 2;
        rc = toep_build_poly(c, a.tp_poly, bank, a.interp, a.decim, a.tpp);
        if (rc) { return rc; }
        cap = cap * (size_t)a.interp / (size_t)a.decim + 4;
        a.i_poly = add_stream(need_of(a.n_stages + 1), cap);
        if (a.i_poly < 0) { return SDRPP_ERR_NOMEM; }
    }
    if (af->hpf_ntaps > 0) {
        a.htaps.assign(af->hpf_taps, af->hpf_taps + af->hpf_ntaps);
        rc = upload_blocked(c, &a.d_hpf, a.htaps.data(), (int)a.htaps.size(), 1, &a.hpf_kp);
        if (rc) { return rc; }
        a.tp_hpf.kind = 4;
        rc = toep_build_fir(c, a.tp_hpf, a.htaps.data(), (int)a.htaps.size(), 1);
        if (rc) { return rc; }
        a.i_hpf = add_stream(0, cap);
        if (a.i_hpf < 0) { return SDRPP_ERR_NOMEM; }
    }
    if (a.alpha != 0.0f) {
        rc = dev_alloc(c, &a.d_last, 2);
        if (rc) { return rc; }
        HIPCHK(c, hipMemset(a.d_last, 0, 2 * sizeof(float2)));
        a.state_cur = 0;
        a.seg_cap = (int)(cap / SDRPP_DEEMP_SEG) + 2;
        rc = dev_alloc(c, &a.d_seg, 2 * ((size_t)a.seg_cap + 1));
        if (rc) { return rc; }
        a.i_deemp = add_stream(0, cap);
        if (a.i_deemp < 0) { return SDRPP_ERR_NOMEM; }
    }
    a.i_last = v.i_out;
    a.on = true;
    return SDRPP_OK;
}

static Stream* af_stream(Vfo& v) { return (v.af.on && v.af.i_last >= 0) ? &v.st[(size_t)v.af.i_last] : nullptr; }

int sdrpp_vfo_af_count(sdrpp_ctx* c, int id) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Stream* s = af_stream(*it->second);
    if (!s) { return fail(c, SDRPP_ERR_INVALID, "VFO %d has no AF chain", id); }
    return s->n;
}

int sdrpp_vfo_af_read(sdrpp_ctx* c, int id, float* dst, int max) {
    DeviceScope dev_scope_(c);
    if (!c || !dst || max < 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Stream* s = af_stream(*it->second);
    if (!s) { return fail(c, SDRPP_ERR_INVALID, "VFO %d has no AF chain", id); }
    const int n = std::min(max, s->n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int rc = pipe_timeouts_check(c)) { return rc; }
    if (n > 0) { HIPCHK(c, hipMemcpy(dst, s->data, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost)); }
    return n;
}

int sdrpp_vfo_af_device_buffer(sdrpp_ctx* c, int id, const float** out, int* n_out) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Stream* s = af_stream(*it->second);
    if (!s) { return fail(c, SDRPP_ERR_INVALID, "VFO %d has no AF chain", id); }
    if (out) { *out = s->data; }
    if (n_out) { *n_out = s->n; }
    return SDRPP_OK;
}

int sdrpp_abi_sizeof_af_desc(void) { return (int)sizeof(sdrpp_af_desc); }

// ---- sink-side sample packing ----------------------------------------------------------------------------------
static Stream* pick_stream(sdrpp_ctx* c, Vfo& v, int which) {
    (void)c;
    if (which == 0) { return (v.d.demod == SDRPP_DEMOD_RAW) ? &v.st[(size_t)v.i_if] : &v.st[(size_t)v.i_out]; }
    if (which == 1) { return &v.st[(size_t)v.i_if]; }
    if (which == 2) { return (v.af.on && v.af.i_last >= 0) ? &v.st[(size_t)v.af.i_last] : nullptr; }
    return nullptr;
}
static int pack_scratch(sdrpp_ctx* c, size_t bytes) {
    if (bytes <= c->pack_cap) { return SDRPP_OK; }
    dev_free(c->d_pack);
    c->pack_cap = 0;
    int rc = dev_alloc(c, &c->d_pack, bytes + 1024);
    if (rc) { return rc; }
    c->pack_cap = bytes + 1024;
    return SDRPP_OK;
}

int sdrpp_vfo_read_pcm(sdrpp_ctx* c, int id, int which, int pcm_type, float scale, void* dst_host, int max_frames) {
    DeviceScope dev_scope_(c);
    if (!c || !dst_host || max_frames < 0 || (pcm_type != 0 && pcm_type != 1)) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Stream* s = pick_stream(c, *it->second, which);
    if (!s) { return fail(c, SDRPP_ERR_INVALID, "VFO %d has no such stream (%d)", id, which); }
    const int n = std::min(max_frames, s->n);
    if (n == 0) { return 0; }
    const long long nv = (long long)n * 2;
    const size_t esz = pcm_type == 1 ? 2 : 1;
    int rc = pack_scratch(c, (size_t)nv * esz);
    if (rc) { return rc; }
    const dim3 grid((unsigned)std::min<long long>((nv + 255) / 256, 4096));
    if (pcm_type == 1) { hipLaunchKernelGGL(pack_convert_kernel<int16_t>, grid, dim3(256), 0, c->stream, (const float*)s->data, scale, nv, (int16_t*)c->d_pack); }
    else { hipLaunchKernelGGL(pack_convert_kernel<int8_t>, grid, dim3(256), 0, c->stream, (const float*)s->data, scale, nv, (int8_t*)c->d_pack); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int rc = pipe_timeouts_check(c)) { return rc; }
    HIPCHK(c, hipMemcpy(dst_host, c->d_pack, (size_t)nv * esz, hipMemcpyDeviceToHost));
    return n;
}

// the pre-processed wideband IQ of the most recent push as int16 / int8: what the recorder's baseband mode writes (bindIQStream consumer ->
// wav::Writer::write, utils/wav.cpp:158-167), converted on the device so that the copy to the host carries 4 (2) bytes per sample
int sdrpp_preproc_read_pcm(sdrpp_ctx* c, int pcm_type, float scale, void* dst_host, int max_samples) {
    DeviceScope dev_scope_(c);
    if (!c || !dst_host || max_samples < 0 || (pcm_type != 0 && pcm_type != 1)) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (!c->pre.on) { return fail(c, SDRPP_ERR_INVALID, "no pre-processing chain configured"); }
    const int n = std::min(max_samples, c->pre.last_n);
    if (n == 0) { return 0; }
    const long long nv = (long long)n * 2;
    const size_t esz = pcm_type == 1 ? 2 : 1;
    int rc = pack_scratch(c, (size_t)nv * esz);
    if (rc) { return rc; }
    const dim3 grid((unsigned)std::min<long long>((nv + 255) / 256, 4096));
    if (pcm_type == 1) { hipLaunchKernelGGL(pack_convert_kernel<int16_t>, grid, dim3(256), 0, c->stream, (const float*)c->pre.last, scale, nv, (int16_t*)c->d_pack); }
    else { hipLaunchKernelGGL(pack_convert_kernel<int8_t>, grid, dim3(256), 0, c->stream, (const float*)c->pre.last, scale, nv, (int8_t*)c->d_pack); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int rc = pipe_timeouts_check(c)) { return rc; }
    HIPCHK(c, hipMemcpy(dst_host, c->d_pack, (size_t)nv * esz, hipMemcpyDeviceToHost));
    return n;
}

int sdrpp_vfo_read_compressed(sdrpp_ctx* c, int id, int which, int pcm_type, unsigned char* dst_host, int max_bytes) {
    DeviceScope dev_scope_(c);
    if (!c || !dst_host || pcm_type < 0 || pcm_type > 2) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Stream* s = pick_stream(c, *it->second, which);
    if (!s) { return fail(c, SDRPP_ERR_INVALID, "VFO %d has no such stream (%d)", id, which); }
    const int n = s->n;
    const long long nv = (long long)n * 2;
    const size_t esz = pcm_type == 2 ? 4 : (pcm_type == 1 ? 2 : 1);
    const size_t total = 8 + (size_t)nv * esz;
    if (n == 0) { return 0; }  // the reference block does not swap an empty frame (sample_stream_compressor.h:70-73)
    if ((size_t)max_bytes < total) { return fail(c, SDRPP_ERR_INVALID, "compressed frame needs %zu bytes", total); }
    uint16_t hdr[2] = { 0, (uint16_t)pcm_type };
    memcpy(dst_host, hdr, 4);
    float scaler = 0.0f;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (pcm_type == 2) {
        memcpy(dst_host + 4, &scaler, 4);
        HIPCHK(c, hipMemcpy(dst_host + 8, s->data, (size_t)nv * 4, hipMemcpyDeviceToHost));
        return (int)total;
    }
    int rc = pack_scratch(c, (size_t)nv * esz + 256 * sizeof(float));
    if (rc) { return rc; }
    float* d_part = (float*)(c->d_pack);
    char* d_data = c->d_pack + 256 * sizeof(float);
    const int nb = (int)std::min<long long>((nv + 255) / 256, 256);
    hipLaunchKernelGGL(pack_max_kernel, dim3((unsigned)nb), dim3(256), 0, c->stream, (const float*)s->data, nv, d_part);
    float part[256];
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(part, d_part, (size_t)nb * sizeof(float), hipMemcpyDeviceToHost));
    float maxVal = part[0];
    for (int i = 1; i < nb; i++) {
        if (part[i] > maxVal) { maxVal = part[i]; }
    }
    scaler = maxVal;
    memcpy(dst_host + 4, &scaler, 4);
    const dim3 grid((unsigned)std::min<long long>((nv + 255) / 256, 4096));
    if (pcm_type == 1) { hipLaunchKernelGGL(pack_convert_kernel<int16_t>, grid, dim3(256), 0, c->stream, (const float*)s->data, 32768.0f / maxVal, nv, (int16_t*)d_data); }
    else { hipLaunchKernelGGL(pack_convert_kernel<int8_t>, grid, dim3(256), 0, c->stream, (const float*)s->data, 128.0f / maxVal, nv, (int8_t*)d_data); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst_host + 8, d_data, (size_t)nv * esz, hipMemcpyDeviceToHost));
    return (int)total;
}

// calculateVFOSignalInfo (waterfall.cpp:558-598) on the newest line of the history ring
int sdrpp_wf_signal_info(sdrpp_ctx* c, double center_offset, double bandwidth, double whole_bandwidth, float* strength, float* snr) {
    DeviceScope dev_scope_(c);
    if (!c || !strength || !snr) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    sdrpp_ctx::Wf& W = c->wf;
    if (W.height <= 0) { return fail(c, SDRPP_ERR_INVALID, "no waterfall history configured (sdrpp_wf_configure)"); }
    if (W.lines <= 0) { return 0; }  // the reference returns false: nothing to measure yet
    const int N = c->fft_size;
    const double f[4] = { center_offset - bandwidth, center_offset - (bandwidth / 2.0), center_offset + (bandwidth / 2.0), center_offset + bandwidth };
    int off[4];
    for (int i = 0; i < 4; i++) {
        const int v = (int)(((f[i] / (whole_bandwidth / 2.0)) * (double)(N / 2)) + (N / 2));
        off[i] = std::min(std::max(v, 0), N);
    }
    if (off[2] >= N) { off[2] = N - 1; }  // the reference reads fftLine[rawFFTSize] here; clamp to the last bin
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->fft_stream) { HIPCHK(c, hipStreamSynchronize(c->fft_stream)); }
    int rc = pack_scratch(c, 64);
    if (rc) { return rc; }
    hipLaunchKernelGGL(wf_signal_info_kernel, dim3(1), dim3(256), 0, c->stream, (const float*)(W.d_ring + (size_t)W.cur * N), off[0], off[1], off[2], off[3], (float*)c->d_pack);
    float out[2];
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->d_pack, sizeof(out), hipMemcpyDeviceToHost));
    *strength = out[0];
    *snr = out[1];
    return 1;
}

int sdrpp_set_reference_block(sdrpp_ctx* c, int ref_block) {
    DeviceScope dev_scope_(c);
    if (!c || ref_block < 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    c->ref_block = ref_block;
    return SDRPP_OK;
}

int sdrpp_set_backend_pipeline(sdrpp_ctx* c, int on) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    c->pipe_on = on < 0 ? 0 : on;
    return SDRPP_OK;
}
int sdrpp_set_nco_mode(sdrpp_ctx* c, int mode) {
    DeviceScope dev_scope_(c);
    if (!c || (mode != SDRPP_NCO_CLOSED_FORM && mode != SDRPP_NCO_REFERENCE_ROTATOR)) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    if (!c->vfos.empty() && mode != c->nco_exact) { return fail(c, SDRPP_ERR_INVALID, "the NCO mode can only change while no VFO exists (it decides how a VFO's front end is built)"); }
    c->nco_exact = mode;
    return SDRPP_OK;
}

int sdrpp_vfo_set_ssb_phase_delta(sdrpp_ctx* c, int id, float re, float im) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Vfo& v = *it->second;
    if (v.d.demod < SDRPP_DEMOD_USB) { return fail(c, SDRPP_ERR_INVALID, "VFO %d has no SSB demodulator", id); }
    v.d.ssb_phase_delta_re = re;
    v.d.ssb_phase_delta_im = im;
    v.theta2 = sdrpp_host::turnsPerSample(re, im);  // the translation is sample-wise: nothing to hand over, the phase stays continuous
    return SDRPP_OK;
}

int sdrpp_vfo_reset(sdrpp_ctx* c, int id) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    return vfo_reset_state(c, *it->second);
}

static Stream* out_stream(Vfo& v) { return (v.d.demod == SDRPP_DEMOD_RAW) ? &v.st[(size_t)v.i_if] : &v.st[(size_t)v.i_out]; }

int sdrpp_vfo_out_count(sdrpp_ctx* c, int id) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    {
        int frc = flush_pending_opt(c, 0);
        if (frc) { return frc; }
    }
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    return out_stream(*it->second)->n;
}

int sdrpp_vfo_read(sdrpp_ctx* c, int id, float* dst, int max) {
    DeviceScope dev_scope_(c);
    if (!c || !dst || max < 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Stream* s = out_stream(*it->second);
    const int n = std::min(max, s->n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int rc = pipe_timeouts_check(c)) { return rc; }
    if (n > 0) { HIPCHK(c, hipMemcpy(dst, s->data, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost)); }
    return n;
}

int sdrpp_vfo_read_many(sdrpp_ctx* c, int n, const int* ids, const int* which, float* dst_host, int64_t max_samples, int64_t* offsets, int* counts) {
    DeviceScope dev_scope_(c);
    if (!c || n < 0 || (n > 0 && (!ids || !offsets || !counts)) || max_samples < 0) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    std::vector<GatherJob> jobs;
    int64_t total = 0;
    int mx = 0;
    for (int i = 0; i < n; i++) {
        auto it = c->vfos.find(ids[i]);
        if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", ids[i]); }
        Stream* s = pick_stream(c, *it->second, which ? which[i] : 0);
        if (!s) { return fail(c, SDRPP_ERR_INVALID, "VFO %d has no such stream (%d)", ids[i], which ? which[i] : 0); }
        offsets[i] = total;
        counts[i] = s->n;
        if (s->n > 0) { jobs.push_back(GatherJob{ (const float2*)s->data, (long long)total, s->n }); }
        mx = std::max(mx, s->n);
        total += s->n;
    }
    if (!dst_host) { return (int)std::min<int64_t>(total, 0x7fffffff); }  // size query: offsets / counts filled, nothing copied
    if (total > max_samples) { return fail(c, SDRPP_ERR_INVALID, "the outputs need room for %lld samples", (long long)total); }
    if (total == 0) { return 0; }
    if ((size_t)total > c->gather_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dev_free(c->d_gather);
        c->gather_cap = 0;
        int rc = dev_alloc(c, &c->d_gather, (size_t)total + 4096);
        if (rc) { return rc; }
        c->gather_cap = (size_t)total + 4096;
    }
    if (jobs.size() <= SDRPP_GATHER_INLINE) {
        GatherArgs ga{};
        for (size_t k = 0; k < jobs.size(); k++) { ga.j[k] = jobs[k]; }
        hipLaunchKernelGGL(gather_inline_kernel, dim3((unsigned)std::max(1, std::min((mx + 255) / 256, 64)), (unsigned)jobs.size()), dim3(256), 0, c->stream, ga, c->d_gather);
        HIPCHK(c, hipMemcpyAsync(dst_host, c->d_gather, (size_t)total * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (int rc = pipe_timeouts_check(c)) { return rc; }
        return (int)std::min<int64_t>(total, 0x7fffffff);
    }
    if ((int)jobs.size() > c->gather_jobs_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dev_free(c->d_gather_jobs);
        c->gather_jobs_cap = 0;
        int rc = dev_alloc(c, &c->d_gather_jobs, jobs.size() + 64);
        if (rc) { return rc; }
        c->gather_jobs_cap = (int)jobs.size() + 64;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_gather_jobs, jobs.data(), jobs.size() * sizeof(GatherJob), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)std::max(1, std::min((mx + 255) / 256, 64)), (unsigned)jobs.size()), dim3(256), 0, c->stream, (const GatherJob*)c->d_gather_jobs, c->d_gather);
    HIPCHK(c, hipMemcpyAsync(dst_host, c->d_gather, (size_t)total * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // also keeps `jobs` alive until the upload has been consumed
    if (int rc = pipe_timeouts_check(c)) { return rc; }
    return (int)std::min<int64_t>(total, 0x7fffffff);
}

int sdrpp_vfo_device_buffers(sdrpp_ctx* c, int id, const float** out, int* n_out, const float** if_out, int* n_if) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    FLUSH_PENDING(c);
    auto it = c->vfos.find(id);
    if (it == c->vfos.end()) { return fail(c, SDRPP_ERR_NOT_FOUND, "no VFO %d", id); }
    Vfo& v = *it->second;
    Stream* o = out_stream(v);
    if (out) { *out = o->data; }
    if (n_out) { *n_out = o->n; }
    if (if_out) { *if_out = v.st[(size_t)v.i_if].data; }
    if (n_if) { *n_if = v.st[(size_t)v.i_if].n; }
    return SDRPP_OK;
}

// ---- data path --------------------------------------------------------------------------------------------------------------------
// Landing buffer of the pass being assembled (allocated on first use of the second one; waits until the pass that last read it is done)
static int landing_acquire(sdrpp_ctx* c, bool need16) {
    const int b = c->land_cur;
    if (!c->iq_land[b]) {
        int rc = dev_alloc(c, &c->iq_land[b], (size_t)c->max_push * 2 + 32);
        if (rc) { return rc; }
    }
    if (need16 && !c->iq_land16[b]) {
        int rc = dev_alloc(c, &c->iq_land16[b], (size_t)c->max_push * 2 + 32);
        if (rc) { return rc; }
    }
    if (c->pending == 0 && c->land_used[b]) {
        HIPCHK(c, hipEventSynchronize(c->land_ev[b]));
        c->land_used[b] = false;
    }
    return SDRPP_OK;
}
// the pass over what has been staged: kernels enqueued, landing buffer marked busy until they are done, the other one becomes current
static int landing_process(sdrpp_ctx* c, int64_t count, const std::vector<int>* ends) {
    const int b = c->land_cur;
    int rc = push_common(c, c->iq_land[b], count, ends);
    (void)hipEventRecord(c->land_ev[b], c->stream);
    c->land_used[b] = true;
    c->land_cur ^= 1;
    return rc;
}
int flush_pending(sdrpp_ctx* c) { return flush_pending_opt(c, 1); }
int flush_pending_opt(sdrpp_ctx* c, int drain) {
    if (c->pipelined && c->held.kind >= 0) {  // pushes held for a launch group: the group goes out as it stands (what the host "knows" about the last push needs its plan)
        int rc = tick_group_launch(c);
        if (rc) { return rc; }
    }
    if (drain && c->pipelined && !c->tickq.empty()) {  // pipelined mode: run what the last blocks still have queued
        int rc = tick_drain(c);
        if (rc) { return rc; }
    }
    if (c->pending == 0) { return SDRPP_OK; }
    if (c->async_staged) {  // copy kernels of sdrpp_push_pinned_async still in flight: the pass waits for them on the device
        c->async_staged = false;
        HIPCHK(c, hipEventRecord(c->ev_copy, c->copy_stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copy, 0));
    }
    const int64_t n = c->pending;
    c->pending = 0;  // cleared first: push_common's own helpers may call observing functions
    std::vector<int> ends;
    ends.swap(c->pend_ends);
    return landing_process(c, n, &ends);
}
static int push_args_ok(sdrpp_ctx* c, const void* p, int64_t count) {
    if (!c) { return SDRPP_ERR_INVALID; }
    if ((!p && count > 0) || count < 0 || count > c->max_push) { return fail(c, SDRPP_ERR_INVALID, "push of %lld samples (max %lld)", (long long)count, (long long)c->max_push); }
    if (c->deferred && c->pending + count > c->max_push) {
        return fail(c, SDRPP_ERR_INVALID, "deferred pushes hold %lld samples, %lld more exceed max_push %lld: observe the results first", (long long)c->pending, (long long)count, (long long)c->max_push);
    }
    return SDRPP_OK;
}

// The staging slot of the next block handed to the HOST to fill (pipelined mode): what tick_hold's memcpy does, in the caller's hands.  With
// several blocks per launch (sdrpp_set_pipeline_group) the slot is the launch group's: the block lands behind the ones already held.
int sdrpp_push_stage(sdrpp_ctx* c, int64_t count, float** slot) {
    DeviceScope dev_scope_(c);
    if (!c || !slot) { return SDRPP_ERR_INVALID; }
    if (!c->pipelined) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_stage: the context is not in pipelined mode"); }
    if (count <= 0 || count > c->max_push) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_stage: count %lld out of range (max_push %lld)", (long long)count, (long long)c->max_push); }
    sdrpp_ctx::Held& H = c->held;
    if (H.kind >= 0 && !H.ends.empty()) {
        const bool fits = H.kind == 1 && (int)H.ends.size() < c->group_max && H.total + count <= c->max_push && group_eligible(c);
        if (!fits) {
            int rc = tick_group_launch(c);
            if (rc) { return rc; }
        }
    }
    if (H.kind != 1) {  // open a group with a fresh slot
        H = sdrpp_ctx::Held{};
        const int si = c->stage_cur;
        if (!c->stage_host[si]) {
            if (hipHostMalloc((void**)&c->stage_host[si], (size_t)c->max_push * 8 + 64, hipHostMallocMapped) != hipSuccess) { return fail(c, SDRPP_ERR_NOMEM, "page-locked staging buffer"); }
        }
        c->stage_cur = (c->stage_cur + 1) % kStageSlots;
        if (c->stage_tick[si]) { tick_wait_done(c, c->stage_tick[si]); }  // its last landing copy has run
        H.kind = 1;
        H.stage_slot = si;
    }
    c->stage_open = H.stage_slot;
    *slot = reinterpret_cast<float*>(c->stage_host[H.stage_slot]) + (size_t)2 * (size_t)H.total;
    return SDRPP_OK;
}
static int push_staged_impl(sdrpp_ctx* c, int64_t count, const volatile uint32_t* pending) {
    if (!c->pipelined || c->stage_open < 0 || c->held.kind != 1 || c->held.stage_slot != c->stage_open) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_staged without an open staging slot (sdrpp_push_stage)"); }
    if (count <= 0 || count > c->max_push) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_staged: count out of range"); }
    c->stage_open = -1;
    return tick_hold(c, 1, nullptr, count, pending);
}
int sdrpp_push_staged(sdrpp_ctx* c, int64_t count) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    return push_staged_impl(c, count, nullptr);
}
int sdrpp_push_staged_when(sdrpp_ctx* c, int64_t count, const volatile uint32_t* pending) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    return push_staged_impl(c, count, pending);  // (a failed plan returns without having waited: the caller joins its own threads)
}

int sdrpp_push(sdrpp_ctx* c, const float* iq_host, int64_t count) {
    DeviceScope dev_scope_(c);
    int rc = push_args_ok(c, iq_host, count);
    if (rc) { return rc; }
    if (c->pipelined) { return tick_hold(c, 1, iq_host, count, nullptr); }
    if (count == 0) { return c->deferred ? SDRPP_OK : push_common(c, nullptr, 0); }
    rc = landing_acquire(c, false);
    if (rc) { return rc; }
    float* land = c->iq_land[c->land_cur] + 2 * c->pending;
    HIPCHK(c, hipMemcpyAsync(land, iq_host, (size_t)count * 2 * sizeof(float), hipMemcpyHostToDevice, c->copy_stream));
    // from here on the copy may be reading the caller's buffer: no return before the host has waited for it
    if (hipEventRecord(c->ev_copy, c->copy_stream) != hipSuccess) {
        (void)hipStreamSynchronize(c->copy_stream);
        return fail(c, SDRPP_ERR_HIP, "hipEventRecord after the landing copy");
    }
    if (c->deferred) {
        HIPCHK(c, hipEventSynchronize(c->ev_copy));  // the caller's buffer is free again; the kernels of the previous pass keep running meanwhile
        c->pending += count;
        c->pend_ends.push_back((int)c->pending);
        return SDRPP_OK;
    }
    // the pass is enqueued behind the copy ON THE DEVICE while the copy runs (planning + launches take longer than the copy of a
    // reference-sized block), and only then does the host wait for the copy: the caller's buffer is free on return, as before
    if (hipStreamWaitEvent(c->stream, c->ev_copy, 0) != hipSuccess) {
        (void)hipEventSynchronize(c->ev_copy);
        return fail(c, SDRPP_ERR_HIP, "hipStreamWaitEvent on the landing copy");
    }
    rc = landing_process(c, count, nullptr);
    HIPCHK(c, hipEventSynchronize(c->ev_copy));
    return rc;
}

// Staging without a host wait: a copy KERNEL on the copy stream reads the page-locked (device-mapped) buffer over the bus; the pass that
// follows waits for it on the device.  (hipMemcpyAsync is not used: see arena_commit.)
__global__ __launch_bounds__(256) void pinned_stage_kernel(const float2* __restrict__ src, float2* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { dst[i] = src[i]; }
}
int sdrpp_push_pinned_async(sdrpp_ctx* c, const float* iq_pinned, int64_t count) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    void* dptr = nullptr;
    if (c->pipelined) {  // the tick's landing role fetches the block from the page-locked buffer itself; sdrpp_push_wait says when it has
        int prc = push_args_ok(c, iq_pinned, count);
        if (prc) { return prc; }
        if (count == 0) { return SDRPP_OK; }
        if (hipHostGetDevicePointer(&dptr, (void*)iq_pinned, 0) != hipSuccess || !dptr) {
            (void)hipGetLastError();
            return tick_hold(c, 1, iq_pinned, count, nullptr);
        }
        return tick_hold(c, 3, iq_pinned, count, nullptr);
    }
    if (!c->deferred || count <= 0 || !iq_pinned || hipHostGetDevicePointer(&dptr, (void*)iq_pinned, 0) != hipSuccess || !dptr) {
        (void)hipGetLastError();
        return sdrpp_push(c, iq_pinned, count);
    }
    int rc = push_args_ok(c, iq_pinned, count);
    if (rc) { return rc; }
    rc = landing_acquire(c, false);
    if (rc) { return rc; }
    float* land = c->iq_land[c->land_cur] + 2 * c->pending;
    hipLaunchKernelGGL(pinned_stage_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((count + 1023) / 1024, 128))), dim3(256), 0, c->copy_stream, (const float2*)dptr, (float2*)land,
                       (long long)count);
    c->async_staged = true;
    c->async_inflight = true;
    c->pending += count;
    c->pend_ends.push_back((int)c->pending);
    return SDRPP_OK;
}

int sdrpp_push_wait(sdrpp_ctx* c) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    if (c->pipelined) {  // every landing copy so far has run once its tick is complete (what is still held goes out first)
        int rc = tick_group_launch(c);
        if (rc) { return rc; }
        if (c->land_tick) { tick_wait_done(c, c->land_tick); }
        return SDRPP_OK;
    }
    // (not `async_staged`: a flushing call that does not host-synchronise — sdrpp_fft_lines, sdrpp_vfo_out_count, a setter — clears that
    // one while the copy kernels may still be reading the caller's page-locked buffers)
    if (c->async_inflight) {
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        c->async_inflight = false;
    }
    return SDRPP_OK;
}

int sdrpp_push_device(sdrpp_ctx* c, const float* iq_dev, int64_t count) {
    DeviceScope dev_scope_(c);
    int rc = push_args_ok(c, iq_dev, count);
    if (rc) { return rc; }
    if (c->pipelined) { return tick_hold(c, 0, iq_dev, count, nullptr); }  // read in place, one tick from now at the earliest
    if (!c->deferred) { return push_common(c, iq_dev, count); }  // read in place
    if (count == 0) { return SDRPP_OK; }
    rc = landing_acquire(c, false);
    if (rc) { return rc; }
    HIPCHK(c, hipMemcpyAsync(c->iq_land[c->land_cur] + 2 * c->pending, iq_dev, (size_t)count * 2 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    c->pending += count;
    c->pend_ends.push_back((int)c->pending);
    return SDRPP_OK;
}

int sdrpp_push_int16(sdrpp_ctx* c, const int16_t* iq_host, int64_t count) {
    DeviceScope dev_scope_(c);
    int rc = push_args_ok(c, iq_host, count);
    if (rc) { return rc; }
    if (c->pipelined) { return tick_hold(c, 2, iq_host, count, nullptr); }
    if (count == 0) { return c->deferred ? SDRPP_OK : push_common(c, nullptr, 0); }
    rc = landing_acquire(c, true);
    if (rc) { return rc; }
    const int b = c->land_cur;
    int16_t* land16 = c->iq_land16[b] + 2 * c->pending;
    HIPCHK(c, hipMemcpyAsync(land16, iq_host, (size_t)count * 2 * sizeof(int16_t), hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(c, hipEventRecord(c->ev_copy, c->copy_stream));
    HIPCHK(c, hipEventSynchronize(c->ev_copy));
    {
        FamilyTimer t(c, F_MISC);
        const long long n = (long long)count * 2;
        launch(c, int16_to_float_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, (const int16_t*)land16, c->iq_land[b] + 2 * c->pending, n);
    }
    if (c->deferred) {
        c->pending += count;
        c->pend_ends.push_back((int)c->pending);
        return SDRPP_OK;
    }
    return landing_process(c, count, nullptr);
}

// page-locked host memory for buffers the caller pushes from (an H2D copy from pageable memory is staged by the runtime: ~3x slower)
void* sdrpp_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { return nullptr; }
    return p;
}
void sdrpp_host_free(void* p) {
    if (p) { (void)hipHostFree(p); }
}
void* sdrpp_device_alloc(sdrpp_ctx* c, size_t bytes) {
    if (!c) { return nullptr; }
    DeviceScope dev_scope_(c);
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { return nullptr; }
    return p;
}
void sdrpp_device_free(sdrpp_ctx* c, void* p) {
    if (!c || !p) { return; }
    DeviceScope dev_scope_(c);
    (void)hipFree(p);
}
// THREAD-SAFE, unlike the rest of a context: a several-GPU host copies a front end's newest line from its gather thread while the worker thread
// is inside sdrpp_push / sdrpp_result_wait on the same context (host/sdrpp_gpu_blocks.h: copyLatestLineDevice).  So the call touches nothing of
// the context that ever changes: a stream of its own (created with the context), its own lock, no error text (the code is all the caller gets).
// It is therefore NOT ordered with the work on the context's stream — it is for buffers whose content is complete (written by an earlier
// sdrpp_device_copy, or by sdrpp_fft_copy_device followed by sdrpp_sync) and it does not wait for the launches queued there.
int sdrpp_device_copy(sdrpp_ctx* c, void* dst, const void* src, size_t bytes, int kind) {
    if (!c) { return SDRPP_ERR_INVALID; }
    if (kind < 0 || kind > 2 || (bytes && (!dst || !src))) { return SDRPP_ERR_INVALID; }
    if (!bytes) { return SDRPP_OK; }
    DeviceScope dev_scope_(c);
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost);
    std::lock_guard<std::mutex> lck(c->side_mtx);
    if (hipMemcpyAsync(dst, src, bytes, k, c->side_stream) != hipSuccess) { return SDRPP_ERR_HIP; }
    if (hipStreamSynchronize(c->side_stream) != hipSuccess) { return SDRPP_ERR_HIP; }
    return SDRPP_OK;
}

int sdrpp_set_deferred(sdrpp_ctx* c, int on) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    if (on && c->pipelined) { return fail(c, SDRPP_ERR_INVALID, "deferred and pipelined processing exclude each other"); }
    int rc = flush_pending(c);
    c->deferred = on != 0;
    return rc;
}
int64_t sdrpp_pending(sdrpp_ctx* c) { return c ? c->pending : SDRPP_ERR_INVALID; }

// ---- pipelined execution ----------------------------------------------------------------------------------------------------------------
int sdrpp_set_pipelined(sdrpp_ctx* c, int on, int result_flags) {
    DeviceScope dev_scope_(c);
    if (!c || result_flags < 0 || result_flags > 15) { return SDRPP_ERR_INVALID; }
    if (on && c->deferred) { return fail(c, SDRPP_ERR_INVALID, "deferred and pipelined processing exclude each other"); }
    int rc = flush_pending(c);  // (drains the queue when the mode is being left)
    if (rc) { return rc; }
    for (auto& R : c->res) {
        if (R.held) { return fail(c, SDRPP_ERR_INVALID, "results of block %llu are still held", (unsigned long long)R.ticket); }
    }
    c->held = sdrpp_ctx::Held{};
    c->stage_open = -1;
    c->pipelined = on != 0;
    c->res_flags = on ? result_flags : 0;
    return SDRPP_OK;
}
int sdrpp_set_pipeline_group(sdrpp_ctx* c, int max_blocks, int adaptive) {
    DeviceScope dev_scope_(c);
    if (!c || max_blocks < 1 || max_blocks > kGroupMax) { return c ? fail(c, SDRPP_ERR_INVALID, "sdrpp_set_pipeline_group: 1 .. %d blocks per launch", kGroupMax) : SDRPP_ERR_INVALID; }
    if (c->pipelined) {
        int rc = tick_group_launch(c);  // what is held goes out under the old rule
        if (rc) { return rc; }
    }
    c->group_max = max_blocks;
    c->group_adaptive = (adaptive & 1) != 0;
    c->stage_pend_stable = (adaptive & 2) != 0;
    return SDRPP_OK;
}
int sdrpp_pipeline_launch_held(sdrpp_ctx* c) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    return c->pipelined ? tick_group_launch(c) : SDRPP_OK;
}
int sdrpp_pipeline_group_stats(sdrpp_ctx* c, int64_t* out, int max) {
    if (!c || !out || max < 0) { return SDRPP_ERR_INVALID; }
    const int64_t v[5] = { (int64_t)c->groups, c->stat_groups, c->stat_group_blocks, c->stat_group_max, (int64_t)c->held.ends.size() };
    int n = 0;
    for (; n < 5 && n < max; n++) { out[n] = v[n]; }
    return n;
}
uint64_t sdrpp_ticket(sdrpp_ctx* c) { return c ? c->pushes : 0; }
int sdrpp_pipeline_flush(sdrpp_ctx* c) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    if (!c->pipelined) { return SDRPP_OK; }
    int rc = tick_group_launch(c);
    return rc ? rc : tick_drain(c);
}
static bool ticket_is_held_back(const sdrpp_ctx* c, uint64_t ticket) {  // pushed, but its launch group has not gone out yet
    return c->held.kind >= 0 && !c->held.ends.empty() && ticket <= c->pushes && ticket + c->held.ends.size() > c->pushes;
}
static sdrpp_ctx::Result* result_of(sdrpp_ctx* c, uint64_t ticket) {
    if (!c || ticket == 0 || ticket > c->pushes) { return nullptr; }
    sdrpp_ctx::Result& R = c->res[ticket % kResMeta];
    if (R.ticket != ticket) { return nullptr; }
    if (R.held) { return &R; }  // (its bytes are where the host was told, whatever has happened to the ring since)
    return (R.epoch == c->res_epoch && tick_results_region(c, R.group)) ? &R : nullptr;  // (the ring has not come round to its group's region)
}
int sdrpp_result_ready(sdrpp_ctx* c, uint64_t ticket) {
    if (c && ticket_is_held_back(c, ticket)) { return 0; }
    sdrpp_ctx::Result* R = result_of(c, ticket);
    if (!R) { return c ? fail(c, SDRPP_ERR_NOT_FOUND, "no results for block %llu (not gathered, overwritten, or processed as an ordinary pass)", (unsigned long long)ticket) : SDRPP_ERR_INVALID; }
    if (R->done_tick > c->ticks) { return 0; }
    DeviceScope dev_scope_(c);
    return tick_results_visible(c, R->done_tick, false);
}
int sdrpp_result_wait(sdrpp_ctx* c, uint64_t ticket, sdrpp_result* out) {
    DeviceScope dev_scope_(c);
    if (!c || !out) { return SDRPP_ERR_INVALID; }
    if (ticket_is_held_back(c, ticket)) {  // nothing more is coming for its group: it goes out as it stands
        int rc = tick_group_launch(c);
        if (rc) { return rc; }
    }
    sdrpp_ctx::Result* R = result_of(c, ticket);
    if (!R) { return fail(c, SDRPP_ERR_NOT_FOUND, "no results for block %llu (not gathered, overwritten, or processed as an ordinary pass)", (unsigned long long)ticket); }
    while (R->done_tick > c->ticks) {  // its last levels have not been launched yet: nothing more is coming, run them without new input
        int rc = tick_group_launch(c);  // (younger pushes held for a group ride along: their first level shares the launch)
        if (!rc && R->done_tick <= c->ticks) { break; }
        if (!rc) { rc = arena_begin(c); }
        if (!rc) { rc = tick_launch(c, nullptr); }
        if (rc) { return rc; }
        if (c->tickq.empty() && R->done_tick > c->ticks) { return fail(c, SDRPP_ERR_HIP, "internal: block %llu cannot complete", (unsigned long long)ticket); }
    }
    {   // complete on the device AND visible here (the flag alone is not: see tick_results_visible)
        const int vis = tick_results_visible(c, R->done_tick, true);
        if (vis < 0) { return vis; }
    }
    if (!tick_is_done(c, R->done_tick)) { return fail(c, SDRPP_ERR_HIP, "tick %llu did not complete", (unsigned long long)R->done_tick); }
    if (!R->held) {
        sdrpp_ctx::ResRegion* g = tick_results_region(c, R->group);
        if (!g) { return fail(c, SDRPP_ERR_NOT_FOUND, "no results for block %llu (overwritten)", (unsigned long long)ticket); }
        R->held = true;
        g->held++;
    }
    const char* base = R->base;
    out->ticket = ticket;
    out->n_vfo = (int)R->ids.size();
    out->ids = R->ids.data();
    out->offsets = R->offsets.data();
    out->counts = R->counts.data();
    out->samples = reinterpret_cast<const float*>(base);
    out->n_lines = R->n_lines;
    // shapes and presence as the block was planned: sdrpp_fft_set_view / sdrpp_fft_configure between push and wait do not re-label the slot
    out->fft_size = R->fft_size;
    out->data_width = R->data_width;
    const bool zo = R->n_lines > 0 && (R->flags & 2) && R->data_width > 0;
    out->zoomed = zo ? reinterpret_cast<const float*>(base + R->off_zoomed) : nullptr;
    out->index = zo ? reinterpret_cast<const int32_t*>(base + R->off_index) : nullptr;
    out->raw = (R->n_lines > 0 && (R->flags & 4)) ? reinterpret_cast<const float*>(base + R->off_raw) : nullptr;
    out->n_iq = R->n_iq;
    out->iq = R->n_iq > 0 ? reinterpret_cast<const float*>(base + R->off_iq) : nullptr;
    return SDRPP_OK;
}
int sdrpp_result_release(sdrpp_ctx* c, uint64_t ticket) {
    if (!c || ticket == 0 || ticket > c->pushes) { return c ? SDRPP_ERR_NOT_FOUND : SDRPP_ERR_INVALID; }
    sdrpp_ctx::Result& R = c->res[ticket % kResMeta];
    if (R.ticket != ticket) { return SDRPP_ERR_NOT_FOUND; }
    if (R.held) {
        if (sdrpp_ctx::ResRegion* g = tick_results_region(c, R.group)) {
            if (g->held > 0) { g->held--; }
        }
        if (R.epoch != c->res_epoch) {  // a ring the results outgrew while the host held this block: freed with its last held block
            for (size_t i = 0; i < c->res_retired.size(); i++) {
                if (c->res_retired[i].epoch == R.epoch && --c->res_retired[i].held <= 0) {
                    DeviceScope dev_scope_(c);
                    (void)hipHostFree(c->res_retired[i].ring);
                    c->res_retired.erase(c->res_retired.begin() + (long)i);
                    break;
                }
            }
        }
    }
    R.held = false;
    R.ticket = 0;
    return SDRPP_OK;
}

int sdrpp_result_take_lines(sdrpp_ctx* c, uint64_t ticket, float* zoomed_dst, int32_t* index_dst, int max_lines, int* n_lines) {
    sdrpp_result r{};
    int rc = sdrpp_result_wait(c, ticket, &r);
    if (rc) { return rc; }
    if (n_lines) { *n_lines = r.n_lines; }
    if (r.n_lines > max_lines) {
        (void)sdrpp_result_release(c, ticket);
        return fail(c, SDRPP_ERR_INVALID, "block %llu completed %d lines, room for %d", (unsigned long long)ticket, r.n_lines, max_lines);
    }
    if (r.n_lines > 0 && r.zoomed) {
        const size_t bytes = (size_t)r.n_lines * (size_t)r.data_width * 4;
        if (zoomed_dst) { memcpy(zoomed_dst, r.zoomed, bytes); }
        if (index_dst) { memcpy(index_dst, r.index, bytes); }
    }
    return sdrpp_result_release(c, ticket);
}
int sdrpp_pipeline_stats(sdrpp_ctx* c, int64_t* out, int max) {
    if (!c || !out || max < 0) { return SDRPP_ERR_INVALID; }
    const int64_t head[SDRPP_PIPELINE_STATS_HEAD] = { (int64_t)c->ticks, c->stat_tick_blocks, c->stat_pass_blocks, c->stat_crowded, c->stat_last_depth, (int64_t)TR_COUNT, c->stat_set2, c->stat_last_table_bytes };
    int n = 0;
    for (; n < SDRPP_PIPELINE_STATS_HEAD && n < max; n++) { out[n] = head[n]; }
    for (int r = 0; r < TR_COUNT && n < max; r++, n++) { out[n] = c->stat_role_wgs[r]; }
    return n;
}
const char* sdrpp_pipeline_role_name(int role) {
    static const char* const names[] = { "none", "copy", "carry", "rot", "fcm_132_4", "fcm_6", "fcm_10", "fcm_16", "fcm16_132_4", "fcl_0", "fcl_pf", "toep_c", "toep_r", "toep_q",
                                         "firb_c", "firb_r", "firb_s", "firb_q", "pre", "seq", "fft_s10", "fft_s11", "fft_s12", "fft_p1_5", "fft_p1_6", "fft_p1_7", "fft_p1_8", "fft_p1_9",
                                         "fft_p1_10", "fft_p2_7", "fft_p2_8", "fft_p2_9", "fft_p2_10", "fft_p2row", "fft_tr", "zoom_16", "zoom_4", "zoom_1", "fcm16w_132_4", "polyc", "deemp_p0", "deemp_p1",
                                         "dc_p0", "dc_p1", "wf_ring", "wf_trace", "pipe", "rotx16", "fird", "ssbx", "s1_1", "s1d_1", "f2_1", "poly" };
    static_assert(sizeof(names) / sizeof(names[0]) == TR_COUNT, "role names out of step with TickRole");
    return (role >= 0 && role < TR_COUNT) ? names[role] : nullptr;
}

// ---- measurement ---------------------------------------------------------------------------------------------------------------------
int sdrpp_timing_enable(sdrpp_ctx* c, int on) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    timing_flush(c);
    c->timing = on != 0;
    c->timing_mask = (on > 1) ? (unsigned)(on >> 1) : 0xffffffffu;  // on = 1: all families; on = 1 | (mask << 1): selected ones
    for (int i = 0; i < SDRPP_NUM_KERNEL_FAMILIES; i++) {
        c->fam_ms[i] = 0.0;
        c->fam_launch[i] = 0;
    }
    return SDRPP_OK;
}

int sdrpp_timing_read(sdrpp_ctx* c, double* ms, int64_t* launches) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    timing_flush(c);
    for (int i = 0; i < SDRPP_NUM_KERNEL_FAMILIES; i++) {
        if (ms) { ms[i] = c->fam_ms[i]; }
        if (launches) { launches[i] = c->fam_launch[i]; }
    }
    return SDRPP_OK;
}

}  // extern "C"
