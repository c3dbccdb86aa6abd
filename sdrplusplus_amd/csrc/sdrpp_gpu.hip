// C-ABI implementation (include/sdrpp_gpu.h): context, streaming state, per-push planning and kernel launches.
// One context = one IQ stream on one GPU; everything is enqueued on one HIP stream so a push is a fixed sequence of
// launches whose sizes are computed on the host from integer state (decimation offsets, polyphase phase, frame position).
#include <hip/hip_runtime.h>

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

namespace {

constexpr int kArenaSlots = 16;     // >= kTickDepth + 2: a block's job tables are read for kTickDepth ticks after their upload
constexpr size_t kArenaBytes = 4u << 20;
constexpr int kRing = 4;            // pipelined mode: buffers per per-block stream (a consumer runs at most 2 ticks behind its producer; + the gather)
constexpr int kTickDepth = 10;      // pipelined mode: levels 0 .. kTickDepth of a block (see the level table at emit())
constexpr int kResSlots = 16;       // pipelined mode: page-locked result slots (blocks whose results the host has not released yet)
constexpr int kStageSlots = 4;      // pipelined mode: page-locked staging buffers for pushes from pageable host memory
constexpr int kChanHistCap = 4095;  // channel filter may be re-designed up to 4096 taps without reallocating (rx_vfo.h:60-70)
constexpr size_t kScratchBytes = 64u << 20;
constexpr int kMaxLds = 64 * 1024;

enum Family { F_FFT1 = 0, F_FFT2, F_FFTS, F_ZOOM, F_S1, F_DECIM, F_POLY, F_FIR, F_DEMOD, F_MISC, F_AF, F_PIPE, F_TICK };
const char* kFamilyNames[SDRPP_NUM_KERNEL_FAMILIES] = { "fft_pass1", "fft_pass2", "fft_single", "zoom_palette", "vfo_stage1",
                                                        "vfo_decim", "vfo_poly",  "vfo_fir",    "demod",        "carry_misc", "af_chain",  "vfo_pipe", "tick" };

struct Stream {
    int width = 2;
    int hist_len = 0;
    float* data = nullptr;
    float* base = nullptr;  // the allocation `data` lives in (data = base + skew, see stream_alloc)
    size_t cap = 0;  // samples
    float* hist[2] = { nullptr, nullptr };
    int cur = 0;
    int n = 0;
    // pipelined mode: the other kRing - 1 data buffers (base allocations); `base` / `data` rotate through them block by block so that
    // the producer of block n + 1 does not overwrite what a consumer of block n still reads (stream_rotate)
    float* extra[kRing - 1] = {};
    int n_extra = 0, rot = 0;
    int clevel = 0;  // pipelined mode: level of the role that consumes this stream with memory (its history carry runs there)
};

// Tap tables of the matrix-core FIR kernel (vfo_toep_kernel): zero-padded taps + per-lane base indices (one set per carried
// resampler phase).
struct ToepTab {
    float* d_tl = nullptr;
    int* d_lb = nullptr;  // [nvar][64]
    int tl_len = 0, nsteps = 0, s_in = 0, rows = 0, nvar = 0;
    int kind = 0;  // 1 decimator, 2 resampler, 4 channel filter, 8 audio low-pass
    bool ok = false;
};

struct Vfo {
    bool nco_exact = false;        // this VFO runs the reference's float rotator recursion (desc.nco_mode, else the context's mode)
    int id = 0;
    sdrpp_vfo_desc d{};
    std::vector<float> staps[SDRPP_MAX_DECIM_STAGES];
    std::vector<float> rtaps, ctaps_chan, ataps;
    // NCO
    double theta = 0.0, phi = 0.0;
    std::vector<float2> modtaps;  // stage-1 modulated taps
    bool modtaps_dirty = true;
    // integer streaming state
    int soff[SDRPP_MAX_DECIM_STAGES] = { 0, 0, 0, 0 };
    int tpp = 0, pphase = 0, poff = 0;
    long long seen = 0;  // input samples this VFO has consumed since it was added / reset (bounds its view of the IQ history)
    // device constants
    float* d_staps[SDRPP_MAX_DECIM_STAGES] = { nullptr, nullptr, nullptr, nullptr };  // phase-major, padded (FirBJob)
    float* d_staps_nat[SDRPP_MAX_DECIM_STAGES] = { nullptr, nullptr, nullptr, nullptr };  // natural order (fused front kernel)
    int s_kp[SDRPP_MAX_DECIM_STAGES] = { 0, 0, 0, 0 };
    float* d_bank = nullptr;
    float* d_cyc = nullptr;  // blocked polyphase: [interp][rows][lmax] cycle tap tables (one per carried phase)
    int cyc_rows = 0, cyc_lmax = 0;
    int chan_kp = 0, audio_kp = 0;
    float* d_chan = nullptr;
    int chan_ntaps = 0;
    float* d_audio = nullptr;
    int audio_ntaps = 0;
    // device loop state: [AgcState agc][AgcState carrier][float dc]
    char* d_state = nullptr;
    double theta2 = 0.0, phi2 = 0.0;
    // streams: 0..nstages-1 decimator outputs (index 0 also used by the rotate-only path), then poly, chan, dem, out
    std::vector<Stream> st;
    int i_first = 0, i_poly = -1, i_chan = -1, i_dem = -1, i_out = -1, i_if = 0;
    int lvl_if = 1, lvl_out = 1;  // levels (do_vfos_plan) at which the IF stream / the output of the most recent block are written
    ToepTab tp_stage[SDRPP_MAX_DECIM_STAGES], tp_poly, tp_chan, tp_audio;
    // front end as one filter (what the fused translate + filter kernels evaluate): stages 0 (+ 1) of the plan
    bool fused_front = false;      // stages 0 and 1 run as one composite filter (front2_t2 > 0)
    bool no_fuse = false;          // stage-1 taps are not linear phase: the composite forms do not apply
    unsigned long long tap_hash = 0;  // of the stage-0 / stage-1 taps: only VFOs with identical taps share a front-end job
    float* d_h12 = nullptr;        // composite (or stage-0) taps, real, natural order — used by the retune hand-over kernel
    int h12_K = 0, h12_lgD = 0;
    // RxVFO::setOffset hand-over (closed-form NCO): retune points whose old-increment samples a filter window can still reach
    struct Retune { long long pos; double theta_before; };  // pos: input samples consumed (Vfo::seen) when the increment changed
    std::vector<Retune> recs;
    // reference-rotator mode (sdrpp_set_nco_mode): rotated full-rate stream + persistent float phases (main xlator, SSB xlator)
    int i_rot = -1;
    float2* d_rot = nullptr;
    // radio AF chain (sdrpp_vfo_set_af): RationalResampler<stereo_t> -> high-pass -> de-emphasis, fed by st[i_out]
    struct Af {
        bool on = false;
        int n_stages = 0, decim_s[SDRPP_MAX_DECIM_STAGES] = { 1, 1, 1, 1 };
        std::vector<float> staps[SDRPP_MAX_DECIM_STAGES], rtaps, htaps;
        float* d_staps[SDRPP_MAX_DECIM_STAGES] = { nullptr, nullptr, nullptr, nullptr };
        int s_kp[SDRPP_MAX_DECIM_STAGES] = { 0, 0, 0, 0 };
        ToepTab tp_stage[SDRPP_MAX_DECIM_STAGES], tp_poly, tp_hpf;
        int interp = 1, decim = 1, tpp = 0;
        float* d_bank = nullptr;
        float* d_hpf = nullptr;
        int hpf_kp = 0;
        float alpha = 0.0f;
        float2* d_last = nullptr;  // Deemphasis::lastOut
        float4* d_seg = nullptr;   // per-segment affine maps of the de-emphasis scan
        int seg_cap = 0;
        int soff[SDRPP_MAX_DECIM_STAGES] = { 0, 0, 0, 0 };
        int pphase = 0, poff = 0;
        int i_stage0 = -1, i_poly = -1, i_hpf = -1, i_deemp = -1, i_last = -1;  // indices into st (i_last: where the AF output is)
        int base = -1;  // first AF stream in st (they are appended behind the VFO's own streams)
    } af;
};

struct TimingPair { hipEvent_t a, b; int family; };

}  // namespace

struct sdrpp_ctx {
    int device = 0;
    int64_t max_push = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;         // main stream (VFO bank, copies); may be the caller's
    hipStream_t fft_stream = nullptr;     // FFT branch runs here, concurrently with the VFO bank (HBM-bound vs VALU-bound)
    hipStream_t launch_stream = nullptr;  // stream the next launches / timers go to
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string err;
    std::string devinfo;

    // input: landing buffers of host pushes / of a deferred pass (max_push complex each), ping-pong per pass so that the copies of the
    // next pass overlap the kernels of the current one; copies run on their own stream and are host-synchronised (the caller's
    // buffer is free again when sdrpp_push returns, like a dsp::stream read buffer after flush())
    float* iq_land[2] = { nullptr, nullptr };
    int16_t* iq_land16[2] = { nullptr, nullptr };
    hipEvent_t land_ev[2] = { nullptr, nullptr };   // recorded behind the pass that read the buffer
    bool land_used[2] = { false, false };
    int land_cur = 0;
    hipStream_t copy_stream = nullptr;
    bool async_staged = false;     // sdrpp_push_pinned_async copies enqueued since the last pass (the pass waits for them on the device)
    bool async_inflight = false;   // ... and not yet known to have landed: cleared only by a HOST synchronisation of the copy stream (sdrpp_push_wait)
    hipEvent_t ev_copy = nullptr;
    // deferred processing (sdrpp_set_deferred): pushes are only staged; the next observing call processes them as ONE pass
    bool deferred = false;
    int64_t pending = 0;
    std::vector<int> pend_ends;       // cumulative end of every staged push
    float* iq_hist[2] = { nullptr, nullptr };
    int iq_hist_cap = 0;              // samples of history kept
    int iq_cur = 0;

    // IQFrontEnd pre-processing chain (sdrpp_preproc_configure): PowerDecimator -> DCBlocker -> Conjugate on the wideband stream,
    // in front of the FFT branch and the VFO bank (iq_frontend.cpp:32-39)
    struct Pre {
        bool ref_order = false;            // sdrpp_preproc_set_reference_order: the reference's own summation order / sequential DC blocker
        bool on = false;
        int n_stages = 0, decim_s[SDRPP_MAX_DECIM_STAGES] = { 1, 1, 1, 1 };
        std::vector<float> staps[SDRPP_MAX_DECIM_STAGES];
        ToepTab tp[SDRPP_MAX_DECIM_STAGES];
        float* d_staps[SDRPP_MAX_DECIM_STAGES] = { nullptr, nullptr, nullptr, nullptr };
        int s_kp[SDRPP_MAX_DECIM_STAGES] = { 0, 0, 0, 0 };
        int soff[SDRPP_MAX_DECIM_STAGES] = { 0, 0, 0, 0 };
        float dc_rate = 0.0f;
        int conj = 0;
        Stream raw;               // history of the caller's buffer (data stays the caller's)
        std::vector<Stream> st;   // decimator stage outputs
        Stream out;               // DC blocker / conjugate output
        float2* d_off = nullptr;  // DCBlocker::offset
        float4* d_seg = nullptr;
        int seg_cap = 0;
        const float* last = nullptr;  // what the chain handed on for the most recent push
        int last_n = 0;
    } pre;

    // WaterFall display state (sdrpp_wf_*): raw-line ring in HBM, FFT trace smoothing / hold
    struct Wf {
        int height = 0;
        float* d_ring = nullptr;  // [height][fft_size]
        int cur = 0, lines = 0;   // currentFFTLine, fftLines (waterfall.cpp:879-882)
        int width = 0;            // data_width the trace arrays were sized for
        float* d_latest = nullptr;
        float* d_smooth = nullptr;  // nullptr = smoothing off
        float* d_hold = nullptr;
        bool hold_on = false, have_latest = false;
        float alpha = 0.0f, beta = 1.0f, hold_speed = 0.0f;
    } wf;

    char* d_pack = nullptr;  // scratch of the packed-sample reads (sdrpp_vfo_read_pcm / _compressed)
    size_t pack_cap = 0;
    float2* d_gather = nullptr;     // sdrpp_vfo_read_many: packed outputs + job table
    size_t gather_cap = 0;          // samples
    GatherJob* d_gather_jobs = nullptr;
    int gather_jobs_cap = 0;

    // job arena
    char* arena_host[kArenaSlots] = {};
    char* arena_host_dev[kArenaSlots] = {};  // device-side address of the same pinned memory
    hipEvent_t arena_ev[kArenaSlots] = {};
    bool arena_used[kArenaSlots] = {};
    char* arena_dev_slot[kArenaSlots] = {};  // a device arena per slot: the job tables of a block outlive its first launch in pipelined mode
    char* arena_dev = nullptr;               // = arena_dev_slot[arena_slot]
    int arena_slot = 0;
    size_t arena_off = 0;

    // FFT
    bool fft_on = false;
    int fft_size = 0, fft_lg = 0, nz = 0, skip = 0;
    float* d_window = nullptr;
    float2* d_tw1 = nullptr;   // tw(e, N1) / tw(e, N) for single pass, e < L/2
    float2* d_tw2 = nullptr;
    float2* d_twn = nullptr;   // [k1][n2] tw(n2*k1, N)
    float2* d_scratch = nullptr;
    float* d_lines = nullptr;
    float* d_lines_grp = nullptr;  // per line: maxima of aligned groups of zoom_grp bins (pass 2 writes them for the zoom kernel); N > 4096 only
    int zoom_grp = 0;
    size_t lines_cap = 0;
    int64_t fft_pos = 0, fft_next = 0;
    int n_lines = 0;
    // view
    int view_start = 0, view_size = 0, data_width = 0;
    float wf_min = -120.0f, wf_max = 0.0f;
    int32_t* d_zstart = nullptr;
    std::vector<int32_t> h_zstart, h_zcount;  // host copies of the view's pixel ranges (zoom_lanes)
    int zoom_tp_cache = 0, zoom_tp_grp = -1;     // lanes per pixel for (the view, zoom_grp == zoom_tp_grp)
    int32_t* d_zcount = nullptr;
    float* d_zoomed = nullptr;
    int32_t* d_index = nullptr;
    size_t zoom_cap = 0;
    // pipelined mode: the other kRing - 1 sets of the per-block FFT buffers (scratch, lines, group maxima, zoomed, index); the members
    // above rotate through them block by block (fft_ring_rotate), so they always name the buffers of the most recent block
    struct FftBufs { float2* scratch = nullptr; float* lines = nullptr; float* grp = nullptr; float* zoomed = nullptr; int32_t* index = nullptr; };
    FftBufs fft_extra[kRing - 1];
    int fft_extra_n = 0, fft_rot = 0;

    // reference block structure / NCO flavour (sdrpp_set_reference_block, sdrpp_set_nco_mode)
    int ref_block = 0;             // 0: one push = one reference block
    int nco_exact = 0;             // 1: the reference's float rotator recursion instead of the closed-form NCO
    int pipe_on = 1;               // FM back ends as one pipelined launch where that pays (sdrpp_set_backend_pipeline)
    bool pipe_launched = false;    // since the last host synchronisation that looked at the kernels' timeout counter (pipe_timeouts_check)
    int timeouts_seen = 0;         // value of the counter (h_tick_flag[8]) at that look
    std::vector<int> vfo_bounds;   // reference-block ends (cumulative sample counts) of the current push at the VFO bank's input

    // VFOs
    std::map<int, std::unique_ptr<Vfo>> vfos;
    int next_id = 1;
    // cached stage-1 job tap arrays, keyed by membership signature
    std::map<std::string, float2*> s1_tap_cache;  // key = 16 raw bytes: two independent 64-bit hashes of (kind, member ids, increments)

    // ---- pipelined ("tick") execution: one launch per block, the stages of consecutive blocks skewed over consecutive launches
    //      (tick_kernels.h; sdrpp_set_pipelined) ----
    struct RoleLaunch { TickEntry e; size_t lds; int level; int fam; };
    struct Result {                       // what the host knows about the block in a result slot
        uint64_t ticket = 0;              // 0: slot free
        uint64_t done_tick = 0;           // its last level has run when this many ticks have completed
        bool held = false;                // handed out by sdrpp_result_wait, not yet released
        std::vector<int> ids, counts;
        std::vector<int64_t> offsets;
        int n_lines = 0;
        size_t off_zoomed = 0, off_index = 0, off_raw = 0;  // byte offsets in the slot
    };
    bool pipelined = false;
    int res_flags = 0;                    // bit 0: gather every VFO's output, bit 1: zoomed lines + palette indices, bit 2: raw dB lines
    int num_cus = 256;
    int tick_l0_at = getenv("SDRPP_GPU_TICK_L0_AT") ? atoi(getenv("SDRPP_GPU_TICK_L0_AT")) : 0;  // (read when the context is created)
    bool tick_order = getenv("SDRPP_GPU_TICK_ORDER") ? atoi(getenv("SDRPP_GPU_TICK_ORDER")) != 0 : true;  // longest roles first inside a tick (diagnostic switch)
    // grid rules of the roles inside a tick (the stand-alone kernels size their grids for a GPU of their own; in a tick ~8 roles share it, and
    // fewer, longer workgroups amortise the per-workgroup prologues): environment overrides are for measurements
    // workgroups of a pass-1 / pass-2 launch (fft_walk_grid; 0: one tile per workgroup).  Measured on 65536-point frames, 2^24 samples per pass
    // (profiles/r03o_fft16_sweeps.log): pass 1 with 16 columns per workgroup 0.069 ms at one tile each, 0.064 walking from 1024 workgroups (four
    // per CU), 0.106 from 512; pass 2 0.0565 at one tile each, 0.057-0.061 walking (its tiles are contiguous 32 KB reads: nothing to hide)
    int fft_p1_grid = getenv("SDRPP_GPU_FFT_P1_GRID") ? atoi(getenv("SDRPP_GPU_FFT_P1_GRID")) : 1024;
    int fft_p2_grid = getenv("SDRPP_GPU_FFT_P2_GRID") ? atoi(getenv("SDRPP_GPU_FFT_P2_GRID")) : 0;
    int fft_tick_grid = getenv("SDRPP_GPU_FFT_TICK_GRID") ? atoi(getenv("SDRPP_GPU_FFT_TICK_GRID")) : 0;  // ... of a pass-1 / pass-2 role inside a tick (a shared GPU: 18.4 / 16.2 / 18.8 / 18.5 GS/s at 0 / 64 / 128 / 256; cfg 2: 61.7 / - / 57.0)
    bool fft_p1_c32 = getenv("SDRPP_GPU_FFT_P1_C32") != nullptr;  // measurement switch: 32 instead of 16 columns per pass-1 workgroup of a 65536-point transform
    int tick_zoom_groups = getenv("SDRPP_GPU_TICK_ZOOM_GROUPS") ? atoi(getenv("SDRPP_GPU_TICK_ZOOM_GROUPS")) : 8;
    int tick_fcm_waves = getenv("SDRPP_GPU_TICK_FCM_WAVES") ? atoi(getenv("SDRPP_GPU_TICK_FCM_WAVES")) : 768;
    int tick_toep_blocks = getenv("SDRPP_GPU_TICK_TOEP_BLOCKS") ? atoi(getenv("SDRPP_GPU_TICK_TOEP_BLOCKS")) : 256;
    long arena_begins = 0;                // blocks planned so far (block_bounds: one per ordinary pass / per block of a pipelined run)
    int arena_allocs = 0;
    long test_fail_pass = 0;              // SDRPP_GPU_TEST_FAIL_ARENA (see arena_push)
    int test_fail_alloc = 0;
    bool pre_ref_order = false;           // survives sdrpp_preproc_configure (which rebuilds `pre`)
    const volatile uint32_t* stage_pending = nullptr;  // sdrpp_push_staged_when: the block's first launch waits (on the host) for this word to reach 0
    bool rot_exact_single = getenv("SDRPP_GPU_ROT_EXACT_SINGLE") != nullptr;  // measurement switch: the one-wavefront form of the reference rotator
    // VFOs per workgroup of vfo_rotate_exact4_kernel (1 .. 64).  The chain wavefront costs the same for 1 or 64 VFOs (a lane each); the three
    // wavefronts that apply the phases take ~100 cycles per VFO and chunk: beyond ~16 VFOs they, not the chain, set the pace of the workgroup
    // and the input is 8 bytes per sample however often it is read.
    // 32-output tiles per front-end job up to which the ratio-32 front end runs in its small-block shape (vfo_frontcm16_body); 0: never.
    // Unset: 256 for ordinary passes (sr/200 pushes 764 -> 814 MS/s) and for pipelined blocks that are read where they lie in device memory
    // (3.48 -> 3.96 GS/s), never for blocks the tick's landing copy fetches from host memory — workgroups that share a CU with a landing-copy
    // workgroup start 8 us late, which the longer front end hides and the short one does not (profiles/r03zl-r03zn).
    int fcm16_max_tiles = getenv("SDRPP_GPU_FCM16_MAX_TILES") ? atoi(getenv("SDRPP_GPU_FCM16_MAX_TILES")) : -1;
    bool plan_block_from_host = false;    // the block being planned reaches the device through a landing copy
    // phases handed over per full chunk: every 4th / 8th / 16th (cfg 4's 43 SSB channels, the family's time per 2^20 samples: 14.2 / 13.4 / 13.0 ms,
    // profiles/r03x_*; the applying wavefronts take up to SKIP - 1 steps per sample themselves, so fewer VFOs per workgroup go with a larger stride)
    int rot_exact_skip = getenv("SDRPP_GPU_ROTX_SKIP") ? atoi(getenv("SDRPP_GPU_ROTX_SKIP")) : 16;
    int rot_exact_vpw = [] { const char* e = getenv("SDRPP_GPU_ROTX_VPW"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    bool tick_planning = false;           // a block is being planned for the tick queue: emit() queues, plain launches abort the plan
    bool tick_abort = false;              // ... and met a launch that has no role in the tick kernel: the block runs as an ordinary pass
    int plan_top = 0;                     // highest level + 1 the block being planned uses
    std::vector<RoleLaunch> emits;        // roles of the block being planned
    std::deque<std::vector<RoleLaunch>> tickq;  // [0]: roles of the next tick to launch, [1]: of the one after, ...
    uint64_t ticks = 0;                   // ticks launched so far
    uint64_t pushes = 0;                  // blocks accepted so far in pipelined mode (= ticket of the most recent one)
    uint64_t land_tick = 0;               // landing copies of every push so far have run when this many ticks have completed
    TickTable* next_tab = nullptr;        // device address of the role table of the next tick (uploaded by the tick before)
    int next_tab_n = 0;
    TickTable* empty_tab = nullptr;       // device: a table without roles
    unsigned* d_tick_counter = nullptr;   // device: finished workgroups, running total
    unsigned tick_target = 0;             // its value when every tick launched so far has finished
    unsigned* h_tick_flag = nullptr;      // page-locked: completed ticks (written by the last wavefront of each tick)
    unsigned* hd_tick_flag = nullptr;     // the same, device address
    uint64_t arena_tick[kArenaSlots] = {};  // pipelined: the tick that uploaded from this arena slot (+1; 0 = never)
    float* tick_land[3] = {};             // landing ring of host pushes (max_push complex each; allocated on first use)
    float* stage_host[kStageSlots] = {};  // page-locked staging of pushes from pageable memory (max_push complex each; allocated on first use)
    uint64_t stage_tick[kStageSlots] = {};  // the tick whose landing copy reads the slot (+1)
    int stage_cur = 0;
    int stage_open = -1;                  // slot handed out by sdrpp_push_stage and not yet pushed
    char* res_host[kResSlots] = {};       // page-locked result slots
    char* res_dev[kResSlots] = {};        // their device addresses
    size_t res_cap = 0;                   // bytes per slot
    Result res[kResSlots];

    // timing
    bool timing = false;
    unsigned timing_mask = 0xffffffffu;  // families whose launches are bracketed by events while timing is on
    std::vector<TimingPair> tpairs;
    std::vector<hipEvent_t> ev_pool;
    double fam_ms[SDRPP_NUM_KERNEL_FAMILIES] = {};
    int64_t fam_launch[SDRPP_NUM_KERNEL_FAMILIES] = {};
};

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
void timing_flush(sdrpp_ctx* c) {
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
    for (int i = 0; i < SDRPP_MAX_DECIM_STAGES; i++) { v.soff[i] = 0; }
    v.pphase = 0;
    v.poff = 0;
    for (int i = 0; i < SDRPP_MAX_DECIM_STAGES; i++) { v.af.soff[i] = 0; }
    v.af.pphase = 0;
    v.af.poff = 0;
    if (v.af.d_last) { HIPCHK(c, hipMemsetAsync(v.af.d_last, 0, sizeof(float2), c->stream)); }
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
    case TR_CARRY: hipLaunchKernelGGL(carry_kernel, grid, b256, 0, st, (const CarryJob*)e.jobs); break;
    case TR_ROT: hipLaunchKernelGGL(vfo_rotate_kernel, grid, b256, 0, st, e.p.src, (const RotJob*)e.jobs); break;
    case TR_FCM_132_4: hipLaunchKernelGGL((vfo_frontcm_kernel<10, 132, 4>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCM_6: hipLaunchKernelGGL((vfo_frontcm_kernel<6, 0, 0>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCM_10: hipLaunchKernelGGL((vfo_frontcm_kernel<10, 0, 0>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCM_16: hipLaunchKernelGGL((vfo_frontcm_kernel<16, 0, 0>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
    case TR_FCM16_132_4: hipLaunchKernelGGL((vfo_frontcm16_kernel<132, 4>), grid, b256, r.lds, st, e.p.src, (const FrontCMJob*)e.jobs); break;
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
        r.level = 1;
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
    r.level = 1;
    r.fam = F_FFT1;
    c->emits.push_back(r);
    if (lg2 == 12) {  // long transforms: 4096-point rows (dB in place), then the transpose into bin order one level later
        sdrpp_ctx::RoleLaunch q{};
        q.e.gy = 1;
        q.e.role = TR_FFT_P2ROW;
        q.e.gx = g.nframes << lg1;
        q.e.p.p2 = TickP2{ c->d_scratch, c->d_tw2, nullptr, nullptr, lg1, g.nframes };
        q.lds = tick_lds_fft_single(12, 1);
        q.level = 2;
        q.fam = F_FFT2;
        c->emits.push_back(q);
        sdrpp_ctx::RoleLaunch t{};
        t.e.gy = 1;
        t.e.role = TR_FFT_TR;
        t.e.gx = g.nframes * ((1 << lg2) / (SDRPP_FFT_TR_TILE >> lg1));
        t.e.aux = kZoomGrpLong;
        t.e.p.p2 = TickP2{ c->d_scratch, nullptr, out, grp, lg1, g.nframes };
        t.lds = (size_t)(SDRPP_FFT_TR_TILE + 256) * sizeof(float);
        t.level = 3;
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
    q.level = 2;
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

int do_fft(sdrpp_ctx* c, const IqSrc& src, int64_t count) {
    c->n_lines = 0;
    if (!c->fft_on) { return SDRPP_OK; }
    const int64_t P = (int64_t)c->nz + c->skip;
    const int64_t end = c->fft_pos + count;
    int64_t nframes = 0;
    if (end - c->nz - c->fft_next * P >= 0) { nframes = (end - c->nz - c->fft_next * P) / P + 1; }
    if (nframes > 0) {
        if ((size_t)nframes > c->lines_cap) { return fail(c, SDRPP_ERR_INVALID, "internal: %lld frames exceed line capacity %zu", (long long)nframes, c->lines_cap); }
        const size_t per_chunk = std::max<size_t>(1, kScratchBytes / ((size_t)c->fft_size * sizeof(float2)));
        if (c->tick_planning) {
            // one chunk only (the chunks of an ordinary pass share the scratch matrix one after the other), no display state
            if ((size_t)nframes > per_chunk || c->wf.height > 0) {
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
            const int lines_level = c->fft_lg <= 12 ? 1 : (c->fft_lg <= 16 ? 2 : 3);
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
            if (c->wf.height > 0) {  // FFT trace: latestFFT after smoothing / hold (pushFFT, waterfall.cpp:913-939)
                int rc2 = wf_ensure_trace(c);
                if (rc2) { return rc2; }
                launch(c, wf_trace_kernel, dim3((unsigned)(c->data_width + 255) / 256), dim3(256), 0, (const float*)c->d_zoomed, (int)nframes, c->data_width, c->wf.d_latest, c->wf.d_smooth,
                       c->wf.alpha, c->wf.beta, c->wf.hold_on ? c->wf.d_hold : (float*)nullptr, c->wf.hold_speed);
                c->wf.have_latest = true;
            }
        }
        if (c->wf.height > 0) {  // raw lines into the ring (getFFTBuffer, waterfall.cpp:875-886)
            FamilyTimer t(c, F_ZOOM);
            launch(c, wf_ring_store_kernel, dim3((unsigned)std::max(1, std::min(c->fft_size / 1024, 64)), (unsigned)nframes), dim3(256), 0, (const float*)c->d_lines, (int)nframes, c->fft_size,
                   c->wf.d_ring, c->wf.height, c->wf.cur);
            const long long nc = (long long)c->wf.cur - nframes;
            c->wf.cur = (int)(((nc % c->wf.height) + c->wf.height) % c->wf.height);
            c->wf.lines = (int)std::min<int64_t>((int64_t)c->wf.lines + nframes, c->wf.height);
        }
        c->fft_next += nframes;
    }
    c->fft_pos = end;
    c->n_lines = (int)nframes;
    return SDRPP_OK;
}

// ---- VFO bank: one push ------------------------------------------------------------------------------------------------------------
struct S1Member { Vfo* v; int K, lgD, off0, nout; double phi0; int min_idx; int fused, K2, lgD2, off2, nout2; unsigned long long taph; };

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
int pipe_segments(const std::vector<PipeJob>& pipes, int forced, size_t lds) {
    if (pipes.empty()) { return 0; }
    const int bpc = std::max(kPipeBpcMin, std::min(kPipeBpc, (int)((size_t)(160 * 1024) / std::max<size_t>(lds, 1))));
    int max_nmt = 1;
    for (auto& pj : pipes) { max_nmt = std::max(max_nmt, (pj.st[3].nout + kPipeG * 16 * pj.st[3].rows - 1) / (kPipeG * 16 * pj.st[3].rows)); }
    if (forced >= 2) { return std::min(forced, max_nmt); }
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
    const int n_in = (int)count;
    std::vector<S1Member> s1;
    std::vector<RotJob> rot;
    Lev<FirBJob> f_dec;  // register-blocked decimators (tap counts the matrix form does not cover; stage 0 only in reference-rotator mode)
    std::vector<RotXJob> rotx;                          // reference-rotator mode: full-rate float recursion, one lane per VFO
    std::vector<RetuneJob> retune;                      // closed-form NCO: first outputs after a setOffset
    const std::vector<int>& fb = c->vfo_bounds;         // reference-block ends of this push (at least one entry: n_in)
    const bool blocks = fb.size() > 1;
    // Every job carries the LEVEL of its launch in the block's data flow (level L reads what level L - 1 wrote): the front end is
    // level 1 (level 0 = the block's arrival), every filter behind it one more.  A pass launches level by level; in pipelined mode
    // level L of this block runs L ticks from now (tick_kernels.h).
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
    const bool ticking = c->tick_planning;
    const int carry_last = kLevels - 1;
    carry.add(ticking ? 1 : carry_last, iq_carry);  // job 0 of its level: the shared IQ stream
    int max_rot = 0;

    for (auto& kv : c->vfos) {
        Vfo& v = *kv.second;
        Stream* cur = &v.st[(size_t)v.i_first];
        int lvl = 1;  // level at which `cur` is written
        for (auto& s : v.st) { s.clevel = 0; }
        // reference-block ends carried stage by stage down to the demodulator's rate, for the block-dependent operations there
        // (AGC look-ahead, SSB rotator calls)
        const bool agc_mode = v.d.demod == SDRPP_DEMOD_AM || (v.d.demod >= SDRPP_DEMOD_USB && v.d.demod <= SDRPP_DEMOD_DSB);
        const bool need_bnd = agc_mode && (blocks || v.nco_exact);
        std::vector<int> bnd;
        if (need_bnd) { bnd = fb; }
        int first_sep = 0;  // first decimator stage that runs as its own FIR launch
        if (v.nco_exact) {
            // the reference's own data flow: rotate at the full rate (float recursion), then every stage of the plan as a plain FIR
            Stream* tgt = (v.d.n_stages == 0) ? cur : &v.st[(size_t)v.i_rot];
            rotx.push_back(RotXJob{ (float2*)tgt->data, v.d_rot, v.d.phase_delta_re, v.d.phase_delta_im });
            tgt->n = n_in;
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
            }
            mem.min_idx = (v.seen >= need) ? -need : -(int)v.seen;  // older samples predate this VFO: zero
            s1.push_back(mem);
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
                cur = nxt;
            }
        }
        // the FM back end as one pipelined launch: last decimator, resampler, channel filter, discriminator + audio low-pass all in
        // their matrix form, and the pipeline's LDS layout fits
        const int last_dec = v.d.n_stages - 1;
        bool piped_be = c->pipe_on && !ticking && (v.d.demod == SDRPP_DEMOD_WFM || v.d.demod == SDRPP_DEMOD_NFM) && last_dec >= first_sep && v.tp_stage[last_dec].ok &&
                        v.i_poly >= 0 && v.tp_poly.ok && v.i_chan >= 0 && v.chan_ntaps > 0 && v.tp_chan.ok && v.tp_audio.ok;
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
            cur = nxt;
        }
        if (v.i_poly >= 0) {
            Stream* nxt = &v.st[(size_t)v.i_poly];
            const int no = poly_nout(cur->n, v.poff, v.pphase, v.d.interp, v.d.decim);
            if (need_bnd) { bounds_poly(bnd, v.poff, v.pphase, v.d.interp, v.d.decim); }
            lvl++;
            cur->clevel = lvl;
            if (piped_be) {
                pj.st[1] = toep_job(v.tp_poly, v.pphase, stream_in(*cur), nxt->data, v.poff - (v.tpp - 1), no, 0.0f);
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
            cur = nxt;
        }
        if (v.i_chan >= 0 && v.chan_ntaps > 0) {
            Stream* nxt = &v.st[(size_t)v.i_chan];
            lvl++;
            cur->clevel = lvl;
            if (piped_be) {
                pj.st[2] = toep_job(v.tp_chan, 0, stream_in(*cur), nxt->data, -(v.chan_ntaps - 1), cur->n, 0.0f);
                pj.keep[2] = 0;  // the IF stream is the RxVFO's output: all of it
            }
            else if (v.tp_chan.ok) { t_chan.add(lvl, toep_job(v.tp_chan, 0, stream_in(*cur), nxt->data, -(v.chan_ntaps - 1), cur->n, 0.0f)); }
            else { chan.add(lvl, FirBJob{ stream_in(*cur), nxt->data, v.d_chan, v.chan_ntaps, 0, 0, cur->n, v.chan_kp }); }
            nxt->n = cur->n;
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
            lvl++;
            cur->clevel = lvl;
            if (piped_be) {
                pj.st[3] = toep_job(v.tp_audio, 0, stream_in(*cur), out.data, -(v.audio_ntaps - 1), nif, v.d.inv_deviation);
                pipes.push_back(pj);
                pipe_lds = std::max(pipe_lds, pj_lds);
            }
            else if (v.tp_audio.ok) { t_audio_fm.add(lvl, toep_job(v.tp_audio, 0, stream_in(*cur), out.data, -(v.audio_ntaps - 1), nif, v.d.inv_deviation)); }
            else { audio_fm.add(lvl, FirBJob{ stream_in(*cur), out.data, v.d_audio, v.audio_ntaps, 0, 0, nif, v.audio_kp, v.d.inv_deviation }); }
            out.n = nif;
            v.lvl_out = lvl;
        }
        else if (v.d.demod == SDRPP_DEMOD_AM) {
            Stream& dem = v.st[(size_t)v.i_dem];
            Stream& out = v.st[(size_t)v.i_out];
            if (!v.d.am_carrier_agc) { pre.add(lvl + 1, PreJob{ 2, nif, (const float2*)cur->data, dem.data, 0.0, 0.0 }); }
            seq.add(lvl + 2, SeqJob{ 2, nif, (const float2*)cur->data, dem.data, nullptr, agc, agc + 1, dc, v.d.dc_block_rate, v.d.am_carrier_agc, d_bnd, nbnd });
            dem.n = nif;
            lvl += 3;
            dem.clevel = lvl;
            if (v.tp_audio.ok) { t_audio.add(lvl, toep_job(v.tp_audio, 0, stream_in(dem), out.data, -(v.audio_ntaps - 1), nif, 0.0f)); }
            else { audio.add(lvl, FirBJob{ stream_in(dem), out.data, v.d_audio, v.audio_ntaps, 0, 0, nif, v.audio_kp }); }
            out.n = nif;
            v.lvl_out = lvl;
        }
        else if (v.d.demod >= SDRPP_DEMOD_USB && v.d.demod <= SDRPP_DEMOD_DSB) {
            Stream& dem = v.st[(size_t)v.i_dem];
            Stream& out = v.st[(size_t)v.i_out];
            if (v.nco_exact) { ssbx_l.add(lvl + 1, SsbRotXJob{ (const float2*)cur->data, dem.data, v.d_rot + 1, v.d.ssb_phase_delta_re, v.d.ssb_phase_delta_im, d_bnd, nbnd }); }
            else { pre.add(lvl + 1, PreJob{ v.d.demod, nif, (const float2*)cur->data, dem.data, v.theta2, v.phi2 }); }
            seq.add(lvl + 2, SeqJob{ v.d.demod, nif, (const float2*)cur->data, dem.data, out.data, agc, agc + 1, dc, 0.0f, 0, d_bnd, nbnd });
            lvl += 2;
            dem.n = 0;  // scratch only
            out.n = nif;
            v.lvl_out = lvl;
            double p2 = v.phi2 + (double)nif * v.theta2;
            v.phi2 = p2 - std::floor(p2);
        }
        if (v.af.on && v.i_out >= 0) {  // radio AF chain on the demodulator's stereo output
            Vfo::Af& a = v.af;
            Stream* acur = &v.st[(size_t)v.i_out];
            for (int s = 0; s < a.n_stages; s++) {
                Stream* nxt = &v.st[(size_t)a.i_stage0 + s];
                const int Ds = a.decim_s[s], K = (int)a.staps[s].size();
                const int no = decim_nout(acur->n, a.soff[s], Ds);
                lvl++;
                acur->clevel = lvl;
                if (a.tp_stage[s].ok) { t_af_dec.add(lvl, toep_job(a.tp_stage[s], 0, stream_in(*acur), nxt->data, a.soff[s] - (K - 1), no, 0.0f)); }
                else { af_dec.add(lvl, FirBJob{ stream_in(*acur), nxt->data, a.d_staps[s], K, ilog2(Ds), a.soff[s], no, a.s_kp[s] }); }
                a.soff[s] = a.soff[s] + no * Ds - acur->n;
                nxt->n = no;
                acur = nxt;
            }
            if (a.i_poly >= 0) {
                Stream* nxt = &v.st[(size_t)a.i_poly];
                const int no = poly_nout(acur->n, a.poff, a.pphase, a.interp, a.decim);
                lvl++;
                acur->clevel = lvl;
                if (a.tp_poly.ok) { t_af_poly.add(lvl, toep_job(a.tp_poly, a.pphase, stream_in(*acur), nxt->data, a.poff - (a.tpp - 1), no, 0.0f)); }
                else { af_poly.add(lvl, PolyJob{ stream_in(*acur), (float2*)nxt->data, a.d_bank, a.interp, a.decim, a.tpp, a.pphase, a.poff, no }); }
                const long long A = (long long)a.pphase + (long long)no * a.decim;
                a.pphase = (int)(A % a.interp);
                a.poff = a.poff + (int)(A / a.interp) - acur->n;
                nxt->n = no;
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
                acur = nxt;
            }
            if (a.i_deemp >= 0) {
                Stream* nxt = &v.st[(size_t)a.i_deemp];
                lvl++;
                af_deemp.add(lvl, DeempJob{ (const float2*)acur->data, (float2*)nxt->data, acur->n, a.alpha, a.d_last, a.d_seg,
                                             std::min(a.seg_cap, (acur->n + SDRPP_DEEMP_SEG - 1) / SDRPP_DEEMP_SEG), 0 });
                nxt->n = acur->n;
                acur = nxt;
                lvl += 2;  // (the de-emphasis is three dependent launches)
            }
            a.i_last = (int)(acur - &v.st[0]);
        }
        double p = v.phi + (double)n_in * v.theta;
        v.phi = p - std::floor(p);
        v.seen += n_in;
        // history carries for every stream that has a consumer with memory
        for (auto& s : v.st) {
            if (s.hist_len > 0 && s.data) {
                // pipelined: at the level of the consumer (its window of the NEXT block reads the new history one tick later, the carry of
                // the next block overwrites the old one one tick later still); a stream nobody reads with memory: behind the whole chain
                const int cl = !ticking ? carry_last : (s.clevel > 0 ? s.clevel : lvl + 1);
                carry.add(cl, CarryJob{ s.data, s.hist[s.cur], s.hist[s.cur ^ 1], s.hist_len, s.n, s.width, s.hist_len });
            }
        }
        c->plan_top = std::max(c->plan_top, lvl + 2);
    }

    // ---- stage 1 (optionally fused with stage 2): group VFOs with identical geometry, VT per job ----
    auto same = [](const S1Member& a, const S1Member& b) {
        return a.fused == b.fused && a.K == b.K && a.lgD == b.lgD && a.off0 == b.off0 && a.nout == b.nout && a.min_idx == b.min_idx && a.K2 == b.K2 &&
               a.lgD2 == b.lgD2 && a.off2 == b.off2 && a.nout2 == b.nout2 && a.taph == b.taph;
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
        return a.v->id < b.v->id;
    });
    struct S1Launch { int vt; std::vector<Stage1Job> jobs; int max_nout = 0; int tile = 256; size_t lds = 0; };
    struct F2Launch { int vt; std::vector<Front2Job> jobs; int max_blocks = 0; size_t lds = 0; };
    struct FCMLaunch { std::vector<FrontCMJob> jobs; int max_blocks = 0; size_t lds = 0; };
    S1Launch s1l[4];
    F2Launch f2l[4];
    FCMLaunch fcm[3];  // PF 6 / 10 / 16
    FCMLaunch fcl;     // long first stages (vfo_frontcl_kernel)
    const int vts[4] = { 8, 4, 2, 1 };
    for (int i = 0; i < 4; i++) { s1l[i].vt = vts[i]; f2l[i].vt = vts[i]; }
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
            const int NP = (K + 1) / 2, NP4 = (NP + 7) / 8 * 8;  // rows of the tap operand table, zero padded (the kernels read 4 / 8 rows at a time)
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
            job.off = h.off0 + (h.off2 - (h.K2 - 1)) * D1 - (h.K - 1);
            job.nout = h.nout2;
            job.min_idx = h.min_idx;
            const int ntiles = (h.nout2 + SDRPP_FCM_TILE - 1) / SDRPP_FCM_TILE;
            // one resident round: 256 CUs x 3 blocks x 4 wavefronts (a second, partly filled round would cost as much as the first);
            // the long-stage kernel runs 2 wavefronts per block, its LDS footprint decides how many blocks fit
            const int long_blocks = m_long ? std::max(1, (int)((size_t)(160 * 1024) / ((size_t)frontcl_lds_floats(K, lgD) * 4))) : 0;
            const int resident = m_long ? 256 * long_blocks * 2 : (c->tick_planning ? c->tick_fcm_waves : 3072);
            job.tiles_per_wave = std::max(1, (ntiles + resident - 1) / resident);
            job.atab = reinterpret_cast<const float*>(d_taps);
            job.ptab = d_taps + (size_t)NP4 * 32;
            for (int m = 0; m < SDRPP_FCM_VT; m++) {
                Vfo* v = s1[g + std::min(m, vt - 1)].v;
                job.theta[m] = v->theta;
                job.phi0[m] = s1[g + std::min(m, vt - 1)].phi0;
                job.out[m] = (float2*)v->st[(size_t)v->i_first + (m_long ? 0 : 1)].data;
            }
            if (m_long) {
                fcl.jobs.push_back(job);
                fcl.max_blocks = std::max(fcl.max_blocks, (ntiles + 2 * job.tiles_per_wave - 1) / (2 * job.tiles_per_wave));
                fcl.lds = std::max(fcl.lds, (size_t)frontcl_lds_floats(K, lgD) * 4);
            }
            else {
                FCMLaunch& L = fcm[m_pf == 6 ? 0 : (m_pf == 10 ? 1 : 2)];
                L.jobs.push_back(job);
                L.max_blocks = std::max(L.max_blocks, (ntiles + 4 * job.tiles_per_wave - 1) / (4 * job.tiles_per_wave));
                L.lds = std::max(L.lds, (size_t)frontcm_layout(K, lgD).total * 4);
            }
            g += (size_t)vt;
        }
        while (g < j) {
            const size_t left = j - g;
            int li = left >= 8 ? 0 : (left >= 4 ? 1 : (left >= 2 ? 2 : 3));
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
                job.off1 = h.off0;
                job.ntaps2 = h.K2;
                job.log2_decim2 = h.lgD2;
                job.off2 = h.off2;
                job.nout2 = h.nout2;
                job.t2 = front2_t2(h.K, 1 << h.lgD, h.K2, 1 << h.lgD2, 8);
                job.min_idx = h.min_idx;
                job.ctaps = d_taps;
                job.ptab = d_taps + (size_t)((h.K + 1) / 2) * vt;
                job.taps2 = h.v->d_staps_nat[1];
                for (int m = 0; m < vt; m++) {
                    Vfo* v = s1[g + m].v;
                    job.theta[m] = v->theta;
                    job.phi0[m] = s1[g + m].phi0;
                    job.out[m] = (float2*)v->st[(size_t)v->i_first + 1].data;
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
                job.off0 = h.off0;
                job.nout = h.nout;
                job.min_idx = h.min_idx;
                job.ctaps = d_taps;
                for (int m = 0; m < vt; m++) {
                    Vfo* v = s1[g + m].v;
                    job.theta[m] = v->theta;
                    job.phi0[m] = s1[g + m].phi0;  // phase (turns) of push-relative sample 0
                    job.out[m] = (float2*)v->st[(size_t)v->i_first].data;
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

    // ---- job tables into the arena (one upload for the whole block) ----
    Stage1Job* d_s1[4] = {};
    for (int k = 0; k < 4; k++) {
        if (!s1l[k].jobs.empty()) {
            for (auto& jb : s1l[k].jobs) { s1l[k].lds = std::max(s1l[k].lds, fir_lds(s1l[k].tile, 1 << jb.log2_decim, jb.ntaps, 8)); }
            d_s1[k] = arena_push(c, s1l[k].jobs);
            if (!d_s1[k]) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
        }
    }
    Front2Job* d_f2[4] = {};
    for (int k = 0; k < 4; k++) {
        if (!f2l[k].jobs.empty()) {
            d_f2[k] = arena_push(c, f2l[k].jobs);
            if (!d_f2[k]) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
        }
    }
    FrontCMJob* d_fcl = arena_push(c, fcl.jobs);
    if (!fcl.jobs.empty() && !d_fcl) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
    FrontCMJob* d_fcm[3] = {};
    for (int k = 0; k < 3; k++) {
        if (!fcm[k].jobs.empty()) {
            d_fcm[k] = arena_push(c, fcm[k].jobs);
            if (!d_fcm[k]) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
        }
    }
    RotXJob* d_rotx = arena_push(c, rotx);
    RetuneJob* d_retune = arena_push(c, retune);
    RotJob* d_rot = arena_push(c, rot);
    const int* d_fb = (!rotx.empty()) ? arena_push(c, fb) : nullptr;
    if ((!rotx.empty() && (!d_rotx || !d_fb)) || (!retune.empty() && !d_retune) || (!rot.empty() && !d_rot)) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
    const int pipe_seg = pipe_segments(pipes, c->pipe_on, pipe_lds);
    int pipe_lvl = 0;
    if (!pipes.empty() && pipe_seg == 0) {  // not this push: the same four jobs go to the separate launches
        for (auto& pj : pipes) {
            t_dec.add(pj.lvl, pj.st[0]);
            t_poly.add(pj.lvl + 1, pj.st[1]);
            t_chan.add(pj.lvl + 2, pj.st[2]);
            t_audio_fm.add(pj.lvl + 3, pj.st[3]);
        }
        pipes.clear();
    }
    for (auto& pj : pipes) { pipe_lvl = std::max(pipe_lvl, pj.lvl); }  // (one launch: at the latest level any of its jobs starts at)
    PipeJob* d_pipes = arena_push(c, pipes);
    if (!pipes.empty() && !d_pipes) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
    // matrix-core FIR launches: macro tiles per wavefront, grid and LDS size per job list (before the job tables are uploaded)
    struct ToepList { Lev<ToepJob>* L; int npl, width; bool quad; int fam; int role; };
    ToepList tlists[] = { { &t_dec, 2, 2, false, F_DECIM, TR_TOEP_C },      { &t_poly, 2, 2, false, F_POLY, TR_TOEP_C },       { &t_chan, 2, 2, false, F_FIR, TR_TOEP_C },
                          { &t_audio, 1, 1, false, F_FIR, TR_TOEP_R },      { &t_audio_fm, 2, 1, true, F_FIR, TR_TOEP_Q },     { &t_af_dec, 2, 2, false, F_AF, TR_TOEP_C },
                          { &t_af_poly, 2, 2, false, F_AF, TR_TOEP_C },     { &t_af_hpf, 2, 2, false, F_AF, TR_TOEP_C } };
    constexpr int kToepLists = (int)(sizeof(tlists) / sizeof(tlists[0]));
    ToepPlan tplan[kToepLists][kLevels];
    for (int i = 0; i < kToepLists; i++) {
        Lev<ToepJob>& L = *tlists[i].L;
        for (int l = 0; l < L.top; l++) {
            if (L.at[l].empty()) { continue; }
            tplan[i][l] = toep_plan(L.at[l], tlists[i].npl, c->tick_planning ? c->tick_toep_blocks : 2048);
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

    // ---- level 1: the front end ----
    {
        FamilyTimer t(c, F_S1);
        if (!rotx.empty() && n_in > 0) {
            if (c->rot_exact_single) { launch(c, vfo_rotate_exact_kernel, dim3(((unsigned)rotx.size() + 63) / 64), dim3(64), (size_t)64 * 65 * sizeof(float2), src, (const RotXJob*)d_rotx, (int)rotx.size(), d_fb, (int)fb.size()); }
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
                const dim3 grid((s1l[k].max_nout + 255) / 256, (unsigned)s1l[k].jobs.size());
                switch (s1l[k].vt) {
                case 8: launch(c, vfo_stage1_direct_kernel<8>, grid, dim3(256), 0, src, (const Stage1Job*)d_s1[k]); break;
                case 4: launch(c, vfo_stage1_direct_kernel<4>, grid, dim3(256), 0, src, (const Stage1Job*)d_s1[k]); break;
                case 2: launch(c, vfo_stage1_direct_kernel<2>, grid, dim3(256), 0, src, (const Stage1Job*)d_s1[k]); break;
                default: launch(c, vfo_stage1_direct_kernel<1>, grid, dim3(256), 0, src, (const Stage1Job*)d_s1[k]); break;
                }
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
            const int small_limit = c->fcm16_max_tiles >= 0 ? c->fcm16_max_tiles : ((c->tick_planning && c->plan_block_from_host) ? 0 : 256);
            if (role == TR_FCM_132_4 && one_tile && max_tiles > 0 && max_tiles <= small_limit) {
                if (getenv("SDRPP_TICK_DEBUG")) { fprintf(stderr, "[sdrpp] front end in its small-block shape: %d tiles x %zu jobs\n", max_tiles, fcm[k].jobs.size()); }
                emit(c, 1, F_S1, TR_FCM16_132_4, max_tiles, (int)fcm[k].jobs.size(), (size_t)frontcm16_layout(132, 4).total * 4, d_fcm[k], &src);
                continue;
            }
            emit(c, 1, F_S1, role, fcm[k].max_blocks, (int)fcm[k].jobs.size(), fcm[k].lds, d_fcm[k], &src);
        }
        if (!fcl.jobs.empty() && fcl.max_blocks > 0) {
            bool pf_ok = true;  // every window of the launch fits the register prefetch
            for (auto& jb : fcl.jobs) { pf_ok = pf_ok && (SDRPP_FCM_TILE - 1) * (1 << jb.log2_decim) + jb.ntaps <= 64 * SDRPP_FCL_PF; }
            emit(c, 1, F_S1, pf_ok ? TR_FCL_PF : TR_FCL_0, fcl.max_blocks, (int)fcl.jobs.size(), fcl.lds, d_fcl, &src);
        }
        if (!rot.empty() && max_rot > 0) { emit(c, 1, F_S1, TR_ROT, std::min((max_rot + 255) / 256, 4096), (int)rot.size(), 0, d_rot, &src); }
        if (!retune.empty()) {
            int mx = 0;
            for (auto& r : retune) { mx = std::max(mx, r.nfix); }
            launch(c, vfo_retune_fix_kernel, dim3((unsigned)mx, (unsigned)retune.size()), dim3(64), 0, src, (const RetuneJob*)d_retune);
        }
    }
    auto launch_fir = [&](int level, int fam, std::vector<FirBJob>& jobs, FirBJob* d_jobs, int width, bool stereo, bool quad = false) -> int {
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
            if (max_nout > 0) { launch(c, vfo_fir_direct_kernel<false>, dim3((unsigned)std::min((max_nout + 255) / 256, 1024), (unsigned)jobs.size()), dim3(256), 0, (const FirBJob*)d_jobs); }
            return SDRPP_OK;
        }
        if (max_nout == 0) { return SDRPP_OK; }
        // enough blocks to load-balance 256 CUs: shrink the tile while the grid has fewer than ~8 blocks per CU
        while (threads > 64 && (size_t)((max_nout + threads * R - 1) / (threads * R)) * jobs.size() < 2048) { threads >>= 1; }
        size_t lds = 0;
        for (auto& jb : jobs) { lds = std::max(lds, lds_for(jb, threads)); }
        const int tile = threads * R;
        emit(c, level, fam, width == 2 ? TR_FIRB_C : (quad ? TR_FIRB_Q : (stereo ? TR_FIRB_S : TR_FIRB_R)), (max_nout + tile - 1) / tile, (int)jobs.size(), lds, d_jobs, nullptr, threads);
        return SDRPP_OK;
    };
    // resamplers with many phases (L > 8, e.g. 96/125): cycle-major kernel — one LDS window serves all L phases of up to 64 cycles;
    // a filter whose single cycle does not fit falls back to the per-output kernel
    auto launch_polyc = [&](std::vector<PolyJob>& jobs, PolyJob* d_jobs) -> int {
        if (jobs.empty()) { return SDRPP_OK; }
        const int cap2 = kMaxLds / (int)sizeof(float2);
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
            launch(c, vfo_polyc_kernel, dim3((unsigned)max_tiles, (unsigned)jobs.size()), dim3(256), (size_t)kMaxLds, (const PolyJob*)d_jobs, cap2);
            return SDRPP_OK;
        }
        size_t lds = 0;
        const int tile = 256;
        for (auto& jb : jobs) {
            const size_t ns = (size_t)((long long)tile * jb.decim / jb.interp) + jb.tpp + 4;
            lds = std::max(lds, ns * sizeof(float2));
        }
        if (lds > (size_t)kMaxLds) { return fail(c, SDRPP_ERR_UNSUPPORTED, "polyphase tile does not fit in LDS"); }
        launch(c, vfo_poly_kernel, dim3((max_nout + tile - 1) / tile, (unsigned)jobs.size()), dim3(tile), lds, (const PolyJob*)d_jobs);
        return SDRPP_OK;
    };
    auto launch_polyb = [&](int li, std::vector<PolyBJob>& jobs, PolyBJob* d_jobs) -> int {
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
    };
    auto emit_toep = [&](int i, int l) {
        Lev<ToepJob>& L = *tlists[i].L;
        if (l >= L.top || L.at[l].empty() || tplan[i][l].grid_x == 0) { return; }
        emit(c, l, tlists[i].fam, tlists[i].role, tplan[i][l].grid_x, (int)L.at[l].size(), tplan[i][l].lds, L.dev[l]);
    };
    // the history carries of one level: job 0 of the IQ stream's level is the shared IQ stream (up to a whole FFT frame long), the per-VFO
    // histories are a few hundred samples
    auto launch_carry = [&](int l) {
        std::vector<CarryJob>& cj = carry.at[l];
        if (cj.empty()) { return; }
        const bool has_iq = (l == (ticking ? 1 : carry_last));
        const int iq_elems = has_iq ? cj[0].need * cj[0].width : 0;
        int mx = 0;
        for (size_t k = has_iq ? 1 : 0; k < cj.size(); k++) { mx = std::max(mx, cj[k].need * cj[k].width); }
        if (iq_elems > 128 * 1024 * 2 && cj.size() > 1) {  // a very long IQ carry (FFT frames of 2^18 points and more): its own wide grid
            emit(c, l, F_MISC, TR_CARRY, std::max(1, std::min((iq_elems + 1023) / 1024, 2048)), 1, 0, carry.dev[l]);
            emit(c, l, F_MISC, TR_CARRY, std::max(1, std::min((mx + 255) / 256, 64)), (int)cj.size() - 1, 0, carry.dev[l] + 1);
        }
        else if (iq_elems > 16384 && cj.size() > 1) {
            // one launch for the IQ history (up to a 65 536-point frame: 128 workgroups stride over it) and the per-VFO histories (their
            // workgroups beyond the first find nothing to do): one kernel and one dispatch bubble less per push
            emit(c, l, F_MISC, TR_CARRY, 128, (int)cj.size(), 0, carry.dev[l]);
        }
        else {
            mx = std::max(mx, iq_elems);
            emit(c, l, F_MISC, TR_CARRY, std::max(1, std::min((mx + 1023) / 1024, 2048)), (int)cj.size(), 0, carry.dev[l]);
        }
    };

    // ---- levels 2 ...: everything behind the front end, level by level (within a level the launches are independent of each other) ----
    int top = std::max({ t_dec.top, t_poly.top, t_chan.top, t_audio.top, t_audio_fm.top, t_af_dec.top, t_af_poly.top, t_af_hpf.top, f_dec.top, poly.top,
                         polyb[0].top, polyb[1].top, polyb[2].top, polyb[3].top, chan.top, seq.top, pre.top, audio.top, audio_fm.top, af_dec.top, af_hpf.top,
                         af_poly.top, af_deemp.top, ssbx_l.top, carry.top, pipe_lvl + 1 });
    for (int l = 1; l < top; l++) {
        {
            FamilyTimer t(c, F_DECIM);
            emit_toep(0, l);
            if (l < f_dec.top) {
                rc = launch_fir(l, F_DECIM, f_dec.at[l], f_dec.dev[l], 2, false);
                if (rc) { return rc; }
            }
        }
        if (!pipes.empty() && l == pipe_lvl) {
            FamilyTimer t(c, F_PIPE);
            c->pipe_launched = true;
            launch(c, vfo_pipe_kernel<kPipeG>, dim3((unsigned)pipe_seg, (unsigned)pipes.size()), dim3(256), pipe_lds, (const PipeJob*)d_pipes);
        }
        {
            FamilyTimer t(c, F_POLY);
            emit_toep(1, l);
            if (l < poly.top) {
                rc = launch_polyc(poly.at[l], poly.dev[l]);
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
            if (l < ssbx_l.top && !ssbx_l.at[l].empty()) { launch(c, vfo_ssb_rotate_exact_kernel, dim3((unsigned)ssbx_l.at[l].size()), dim3(64), 0, (const SsbRotXJob*)ssbx_l.dev[l]); }
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
                rc = launch_polyc(af_poly.at[l], af_poly.dev[l]);
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
                if (max_seg > 0) {
                    const dim3 grid((unsigned)max_seg, (unsigned)af_deemp.at[l].size());
                    launch(c, vfo_deemph_kernel<0, 0>, grid, dim3(256), 0, (const DeempJob*)af_deemp.dev[l]);
                    launch(c, vfo_deemph_kernel<0, 1>, grid, dim3(256), 0, (const DeempJob*)af_deemp.dev[l]);
                    launch(c, vfo_deemph_state_kernel<0>, dim3(((unsigned)af_deemp.at[l].size() + 63) / 64), dim3(64), 0, (const DeempJob*)af_deemp.dev[l], (int)af_deemp.at[l].size());
                }
            }
        }
        if (l < carry.top && !carry.at[l].empty()) {
            FamilyTimer t(c, F_MISC);
            launch_carry(l);
        }
    }
    // flip the ping-pong side of every carried stream
    for (auto& kv : c->vfos) {
        for (auto& s : kv.second->st) {
            if (s.hist_len > 0 && s.data) { s.cur ^= 1; }
        }
    }
    return SDRPP_OK;
}

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
        dc.push_back(DeempJob{ (const float2*)cur->data, (float2*)P.out.data, n_out, P.dc_rate, P.d_off, P.d_seg, nseg, P.conj });
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
            if (n_out > 0) { launch(c, iq_dc_block_exact_kernel, dim3(1), dim3(64), 0, dc[0].in, dc[0].out, n_out, P.dc_rate, (float2*)P.d_off, P.conj); }
        }
        else if (!dc.empty() && dc[0].nseg > 0) {
            const dim3 grid((unsigned)dc[0].nseg, 1);
            launch(c, vfo_deemph_kernel<1, 0>, grid, dim3(256), 0, (const DeempJob*)d_dc);
            launch(c, vfo_deemph_kernel<1, 1>, grid, dim3(256), 0, (const DeempJob*)d_dc);
            launch(c, vfo_deemph_state_kernel<1>, dim3(1), dim3(64), 0, (const DeempJob*)d_dc, 1);
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

// ---- streaming state that planning a block changes, for roll-back: a block either happens completely or not at all ----------------------
struct PlanSnapshot {
    struct V { int soff[SDRPP_MAX_DECIM_STAGES]; int pphase, poff; double phi, phi2; long long seen; int i_if, lvl_if, lvl_out; int n[24], cur[24]; size_t nrecs;
               int af_soff[SDRPP_MAX_DECIM_STAGES], af_pphase, af_poff, af_last; };
    std::vector<V> v;
    int64_t fft_pos, fft_next;
    int n_lines, iq_cur, wf_cur, wf_lines;
    bool wf_have;
    int pre_soff[SDRPP_MAX_DECIM_STAGES];
};
void plan_snapshot(sdrpp_ctx* c, PlanSnapshot& S) {
    S.v.resize(c->vfos.size());
    size_t i = 0;
    for (auto& kv : c->vfos) {
        Vfo& v = *kv.second;
        PlanSnapshot::V& q = S.v[i++];
        for (int k = 0; k < SDRPP_MAX_DECIM_STAGES; k++) { q.soff[k] = v.soff[k]; q.af_soff[k] = v.af.soff[k]; }
        q.pphase = v.pphase; q.poff = v.poff; q.phi = v.phi; q.phi2 = v.phi2; q.seen = v.seen; q.i_if = v.i_if; q.lvl_if = v.lvl_if; q.lvl_out = v.lvl_out;
        q.nrecs = v.recs.size();
        q.af_pphase = v.af.pphase; q.af_poff = v.af.poff; q.af_last = v.af.i_last;
        for (size_t k = 0; k < v.st.size() && k < 24; k++) { q.n[k] = v.st[k].n; q.cur[k] = v.st[k].cur; }
    }
    S.fft_pos = c->fft_pos; S.fft_next = c->fft_next; S.n_lines = c->n_lines; S.iq_cur = c->iq_cur;
    S.wf_cur = c->wf.cur; S.wf_lines = c->wf.lines; S.wf_have = c->wf.have_latest;
    for (int k = 0; k < SDRPP_MAX_DECIM_STAGES; k++) { S.pre_soff[k] = c->pre.soff[k]; }
}
// (retune records a plan has dropped stay dropped: they were out of every window's reach)
void plan_restore(sdrpp_ctx* c, const PlanSnapshot& S) {
    size_t i = 0;
    for (auto& kv : c->vfos) {
        Vfo& v = *kv.second;
        const PlanSnapshot::V& q = S.v[i++];
        for (int k = 0; k < SDRPP_MAX_DECIM_STAGES; k++) { v.soff[k] = q.soff[k]; v.af.soff[k] = q.af_soff[k]; }
        v.pphase = q.pphase; v.poff = q.poff; v.phi = q.phi; v.phi2 = q.phi2; v.seen = q.seen; v.i_if = q.i_if; v.lvl_if = q.lvl_if; v.lvl_out = q.lvl_out;
        v.af.pphase = q.af_pphase; v.af.poff = q.af_poff; v.af.i_last = q.af_last;
        for (size_t k = 0; k < v.st.size() && k < 24; k++) { v.st[k].n = q.n[k]; v.st[k].cur = q.cur[k]; }
    }
    c->fft_pos = S.fft_pos; c->fft_next = S.fft_next; c->n_lines = S.n_lines; c->iq_cur = S.iq_cur;
    c->wf.cur = S.wf_cur; c->wf.lines = S.wf_lines; c->wf.have_latest = S.wf_have;
    for (int k = 0; k < SDRPP_MAX_DECIM_STAGES; k++) { c->pre.soff[k] = S.pre_soff[k]; }
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
    PlanSnapshot snap;
    plan_snapshot(c, snap);
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
                launch(c, carry_kernel, dim3((unsigned)std::max(1, std::min((iq_elems + 1023) / 1024, 2048)), 1), dim3(256), 0, (const CarryJob*)d_carry);
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
bool tick_is_done(const sdrpp_ctx* c, uint64_t nticks) { return !c->h_tick_flag || (int)((unsigned)nticks - *(const volatile unsigned*)c->h_tick_flag) <= 0; }

// One tick: level-0 work of the block that arrives with it (`land`: its landing copy, may be null; the arena slot the caller has filled
// with the block's job tables) + every queued role whose turn it is.  The role table of the NEXT tick is appended to the arena slot and
// travels with this tick's upload.
// expected lifetime of a workgroup of a role relative to the others (tools/tick_trace.py timelines), for the order inside a tick
inline int tick_role_weight(int role, bool crowded) {
    // result copies write page-locked host memory over the bus: few workgroups whose life is mostly that round trip — started last they are the
    // tail of the tick, started first they finish in its shadow (SDRPP_GPU_TICK_COPY_FIRST=0: the old order, for measurements)
    static const bool copy_first = getenv("SDRPP_GPU_TICK_COPY_FIRST") ? atoi(getenv("SDRPP_GPU_TICK_COPY_FIRST")) != 0 : true;
    if (role == TR_COPY && copy_first) { return 110; }
    // FFT pass 1: since its workgroups walk their tiles and take the lean loader they live ~12 us at 10^6-sample blocks, shorter than the
    // Toeplitz roles' 17-33 us.  In a CROWDED tick (more workgroups than the GPU holds at once: they are handed out in index order) they go
    // behind the filters and pass 2, so that the tick ends on short workgroups: 18.6 -> 19.6 GS/s at 10^6-sample blocks; in a tick whose
    // workgroups are all resident from the start the order only decides who gets going first, and pass 1 early is worth 2 % at 200 000-sample
    // blocks (profiles/r03z_tick_p1_weight.log; SDRPP_GPU_TICK_P1_WEIGHT: measurement switch)
    static const int p1_weight = getenv("SDRPP_GPU_TICK_P1_WEIGHT") ? atoi(getenv("SDRPP_GPU_TICK_P1_WEIGHT")) : 0;
    if (role >= TR_FFT_P1_5 && role <= TR_FFT_P1_10) { return p1_weight > 0 ? p1_weight : (crowded ? 45 : 70); }
    // the long first stages (cfg 4): far more workgroups than the GPU holds — behind the sequential recursions and the filters, whose few long
    // workgroups then run beside them instead of after them (10^6-sample blocks 4.25 -> 4.33 GS/s, 307 200: 3.08 -> 3.24;
    // profiles/r03zj_tick_fcl_weight.log; SDRPP_GPU_TICK_FCL_WEIGHT: measurement switch)
    static const int fcl_weight = getenv("SDRPP_GPU_TICK_FCL_WEIGHT") ? atoi(getenv("SDRPP_GPU_TICK_FCL_WEIGHT")) : 58;
    switch (role) {
    case TR_FCL_0: case TR_FCL_PF: return fcl_weight;
    case TR_FCM_132_4: case TR_FCM_6: case TR_FCM_10: case TR_FCM_16: case TR_FCM16_132_4: return 90;
    case TR_SEQ: return 85;
    case TR_FFT_P1_5: case TR_FFT_P1_6: case TR_FFT_P1_7: case TR_FFT_P1_8: case TR_FFT_P1_9: case TR_FFT_P1_10: case TR_FIRB_C: case TR_FIRB_R: case TR_FIRB_S: case TR_FIRB_Q: return 70;
    case TR_TOEP_Q: return 65;
    case TR_TOEP_C: case TR_TOEP_R: case TR_FFT_S10: case TR_FFT_S11: case TR_FFT_S12: return 60;
    case TR_FFT_P2_7: case TR_FFT_P2_8: case TR_FFT_P2_9: case TR_FFT_P2_10: case TR_FFT_P2ROW: return 50;
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
        l0.blocks[0] = (int)std::max<long long>(1, std::min<long long>((land->bytes + 8191) / 8192, 64));
    }
    if (c->arena_off > 0) {
        l0.job[1] = CopyJob{ c->arena_host_dev[c->arena_slot], c->arena_dev, (long long)((c->arena_off + 15) & ~(size_t)15), 0, 0 };
        l0.blocks[1] = (int)std::max<size_t>(1, std::min<size_t>((c->arena_off + 4095) / 4096, 16));  // one 16-byte load per work-item: a round trip over the bus each
    }
    int blocks = l0.blocks[0] + l0.blocks[1];
    size_t lds = 0;
    bool set1 = false;
    for (auto& r : now) {
        blocks += r.e.gx * r.e.gy;
        lds = std::max(lds, r.lds);
        set1 = set1 || r.e.role == TR_FCL_PF;
    }
    {   // where the stage-0 copies stand among the tick's workgroups (SDRPP_GPU_TICK_L0_AT: 0 = in front (default), -1 = behind all roles, n = behind
        // the first n role workgroups): a switch for the measurement DESIGN.md 4b names as the next step
        const int l0_at = c->tick_l0_at;
        const int role_blocks = blocks - l0.blocks[0] - l0.blocks[1];
        l0.first = l0_at < 0 ? role_blocks : std::min(l0_at, role_blocks);
    }
    if (blocks == 0) {  // nothing to do at all (an idle flush)
        c->next_tab = tab_dev_next;
        c->next_tab_n = tab_n_next;
        return SDRPP_OK;
    }
    c->tick_target += (unsigned)blocks;
    c->ticks++;
    TickDone done{ c->d_tick_counter, c->hd_tick_flag, c->tick_target, (unsigned)c->ticks };
    const TickTable* tab = c->next_tab_n > 0 ? c->next_tab : c->empty_tab;
    {
        hipEvent_t ea = nullptr;
        const bool timed = c->timing && ((c->timing_mask >> F_TICK) & 1u);
        c->fam_launch[F_TICK]++;
        if (timed) {
            ea = get_event(c);
            (void)hipEventRecord(ea, c->stream);
        }
        HostScope hs("launch");
        if (set1) { hipLaunchKernelGGL((tick_kernel<1>), dim3((unsigned)blocks), dim3(256), lds, c->stream, l0, tab, done); }
        else { hipLaunchKernelGGL((tick_kernel<0>), dim3((unsigned)blocks), dim3(256), lds, c->stream, l0, tab, done); }
        if (timed) {
            hipEvent_t eb = get_event(c);
            (void)hipEventRecord(eb, c->stream);
            c->tpairs.push_back({ ea, eb, F_TICK });
            if (c->tpairs.size() > 8192) { timing_flush(c); }
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
    if (c->pre.on || c->wf.height > 0 || c->deferred) { return false; }
    for (auto& kv : c->vfos) {
        const Vfo& v = *kv.second;
        if (v.af.on || v.nco_exact || !v.recs.empty() || v.st.size() > 24) { return false; }
    }
    return true;
}

// ---- results of a block in page-locked host memory (sdrpp_set_pipelined's result flags): gather roles one level behind the producers ----
size_t tick_results_need(sdrpp_ctx* c) {
    size_t need = 0;
    if (c->res_flags & 1) {
        for (auto& kv : c->vfos) {
            const Vfo& v = *kv.second;
            const Stream& s = (v.d.demod == SDRPP_DEMOD_RAW) ? v.st[(size_t)v.i_if] : v.st[(size_t)v.i_out];
            need += ((s.cap + 16) * 8 + 15) & ~(size_t)15;
        }
    }
    if (c->fft_on) {
        if ((c->res_flags & 2) && c->data_width > 0) { need += 2 * ((c->lines_cap * (size_t)c->data_width * 4 + 15) & ~(size_t)15); }
        if (c->res_flags & 4) { need += (c->lines_cap * (size_t)c->fft_size * 4 + 15) & ~(size_t)15; }
    }
    return need;
}
int tick_results_ensure(sdrpp_ctx* c) {
    const size_t need = tick_results_need(c);
    if (need <= c->res_cap) { return SDRPP_OK; }
    for (int i = 0; i < kResSlots; i++) {
        if (c->res[i].held) { return fail(c, SDRPP_ERR_INVALID, "results of block %llu are still held: release them before the outputs grow", (unsigned long long)c->res[i].ticket); }
    }
    int rc = tick_drain(c);
    if (rc) { return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < kResSlots; i++) {
        if (c->res_host[i]) { (void)hipHostFree(c->res_host[i]); }
        c->res_host[i] = nullptr;
        c->res_dev[i] = nullptr;
        c->res[i] = sdrpp_ctx::Result{};
    }
    c->res_cap = 0;
    const size_t cap = need + need / 8 + 4096;
    for (int i = 0; i < kResSlots; i++) {
        if (hipHostMalloc((void**)&c->res_host[i], cap, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer((void**)&c->res_dev[i], c->res_host[i], 0) != hipSuccess) {
            return fail(c, SDRPP_ERR_NOMEM, "page-locked result slots of %zu bytes", cap);
        }
    }
    c->res_cap = cap;
    return SDRPP_OK;
}
// gather roles of the block just planned -> c->emits; fills its result slot's description
int tick_results_plan(sdrpp_ctx* c) {
    const int slot = (int)(c->pushes % kResSlots);
    sdrpp_ctx::Result& R = c->res[slot];
    if (!c->res_flags) {
        R = sdrpp_ctx::Result{};
        return SDRPP_OK;
    }
    if (R.held) { return fail(c, SDRPP_ERR_INVALID, "result slot of block %llu is still held (release results before %d more blocks are pushed)", (unsigned long long)R.ticket, kResSlots); }
    R = sdrpp_ctx::Result{};
    R.ticket = c->pushes;
    Lev<CopyJob> jobs;
    size_t off = 0;
    char* base = c->res_dev[slot];
    if (c->res_flags & 1) {
        for (auto& kv : c->vfos) {
            const Vfo& v = *kv.second;
            const bool raw = v.d.demod == SDRPP_DEMOD_RAW;
            const Stream& s = raw ? v.st[(size_t)v.i_if] : v.st[(size_t)v.i_out];
            R.ids.push_back(v.id);
            R.offsets.push_back((int64_t)(off / 8));
            R.counts.push_back(s.n);
            const size_t bytes = (size_t)s.n * 8;
            if (bytes) { jobs.add((raw ? v.lvl_if : v.lvl_out) + 1, CopyJob{ s.data, base + off, (long long)bytes, 0x100, 0 }); }
            off += (bytes + 15) & ~(size_t)15;
        }
    }
    R.n_lines = c->fft_on ? c->n_lines : 0;
    if (R.n_lines > 0) {
        const int lines_level = c->fft_lg <= 12 ? 1 : (c->fft_lg <= 16 ? 2 : 3);
        if ((c->res_flags & 2) && c->data_width > 0) {
            const size_t bytes = (size_t)R.n_lines * c->data_width * 4;
            R.off_zoomed = off;
            jobs.add(lines_level + 2, CopyJob{ c->d_zoomed, base + off, (long long)bytes, 0x100, 0 });
            off += (bytes + 15) & ~(size_t)15;
            R.off_index = off;
            jobs.add(lines_level + 2, CopyJob{ c->d_index, base + off, (long long)bytes, 0x100, 0 });
            off += (bytes + 15) & ~(size_t)15;
        }
        if (c->res_flags & 4) {
            const size_t bytes = (size_t)R.n_lines * c->fft_size * 4;
            R.off_raw = off;
            jobs.add(lines_level + 1, CopyJob{ c->d_lines, base + off, (long long)bytes, 0x100, 0 });
            off += (bytes + 15) & ~(size_t)15;
        }
    }
    if (off > c->res_cap) { return fail(c, SDRPP_ERR_INVALID, "internal: results of %zu bytes exceed the slot (%zu)", off, c->res_cap); }
    if (!arena_push_lev(c, jobs)) { return fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
    for (int l = 0; l < jobs.top; l++) {
        if (jobs.at[l].empty()) { continue; }
        long long mx = 0;
        for (auto& j : jobs.at[l]) { mx = std::max(mx, j.bytes); }
        emit(c, l, F_MISC, TR_COPY, (int)std::max<long long>(1, std::min<long long>((mx + 32767) / 32768, 16)), (int)jobs.at[l].size(), 0, jobs.dev[l]);
        c->plan_top = std::max(c->plan_top, l + 1);
    }
    return SDRPP_OK;
}

// results of a block that ran as an ORDINARY pass inside a pipelined run (something the device cannot pipeline: a pre-processing chain, a
// VFO group without the matrix front end, a retune hand-over ...): the same slot layout, filled by plain copies behind the pass and waited
// for here — the slow path, but sdrpp_result_wait / _release then work for EVERY block of a pipelined run, whichever way it was processed
int tick_results_direct(sdrpp_ctx* c) {
    int rc = tick_results_ensure(c);
    if (rc) { return rc; }
    const int slot = (int)(c->pushes % kResSlots);
    sdrpp_ctx::Result& R = c->res[slot];
    if (R.held) { return fail(c, SDRPP_ERR_INVALID, "result slot of block %llu is still held (release results before %d more blocks are pushed)", (unsigned long long)R.ticket, kResSlots); }
    R = sdrpp_ctx::Result{};
    R.ticket = c->pushes;
    size_t off = 0;
    char* base = c->res_host[slot];
    if (c->res_flags & 1) {
        for (auto& kv : c->vfos) {
            const Vfo& v = *kv.second;
            const Stream& s = (v.d.demod == SDRPP_DEMOD_RAW) ? v.st[(size_t)v.i_if] : v.st[(size_t)v.i_out];
            R.ids.push_back(v.id);
            R.offsets.push_back((int64_t)(off / 8));
            R.counts.push_back(s.n);
            const size_t bytes = (size_t)s.n * 8;
            if (off + bytes > c->res_cap) { return fail(c, SDRPP_ERR_INVALID, "internal: results exceed the slot"); }
            if (bytes) { HIPCHK(c, hipMemcpyAsync(base + off, s.data, bytes, hipMemcpyDeviceToHost, c->stream)); }
            off += (bytes + 15) & ~(size_t)15;
        }
    }
    R.n_lines = c->fft_on ? c->n_lines : 0;
    if (R.n_lines > 0) {
        if ((c->res_flags & 2) && c->data_width > 0) {
            const size_t bytes = (size_t)R.n_lines * c->data_width * 4;
            if (off + 2 * ((bytes + 15) & ~(size_t)15) > c->res_cap) { return fail(c, SDRPP_ERR_INVALID, "internal: results exceed the slot"); }
            R.off_zoomed = off;
            HIPCHK(c, hipMemcpyAsync(base + off, c->d_zoomed, bytes, hipMemcpyDeviceToHost, c->stream));
            off += (bytes + 15) & ~(size_t)15;
            R.off_index = off;
            HIPCHK(c, hipMemcpyAsync(base + off, c->d_index, bytes, hipMemcpyDeviceToHost, c->stream));
            off += (bytes + 15) & ~(size_t)15;
        }
        if (c->res_flags & 4) {
            const size_t bytes = (size_t)R.n_lines * c->fft_size * 4;
            if (off + bytes > c->res_cap) { return fail(c, SDRPP_ERR_INVALID, "internal: results exceed the slot"); }
            R.off_raw = off;
            HIPCHK(c, hipMemcpyAsync(base + off, c->d_lines, bytes, hipMemcpyDeviceToHost, c->stream));
            off += (bytes + 15) & ~(size_t)15;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    R.done_tick = c->ticks;  // nothing queued is left: complete as it stands
    return SDRPP_OK;
}

// sdrpp_push_staged_when: the host is still filling the staging slot with other threads while this thread plans the block; nothing that
// reads the slot may be launched before they are through (the word counts their unfinished parts).
int stage_pending_wait(sdrpp_ctx* c) {
    const volatile uint32_t* w = c->stage_pending;
    if (!w) { return SDRPP_OK; }
    c->stage_pending = nullptr;
    HostScope hs("staging wait");
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; *w != 0u; spins++) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0xfffu) == 0xfffu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_staged_when: the staging slot was not completed within 5 s"); }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return SDRPP_OK;
}

// One block in pipelined mode.  `d_iq`: where the samples are (caller's device buffer) or will be once `land` has run (landing ring).
int tick_push(sdrpp_ctx* c, const float* d_iq, int64_t count, const CopyJob* land) {
    if (count == 0) { return SDRPP_OK; }
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
        int rc = fft_ring_ensure(c);
        if (!rc && c->res_flags) { rc = tick_results_ensure(c); }
        if (rc) { return rc; }
    }
    c->pushes++;
    const bool have_slot = as_tick;
    PlanSnapshot snap;
    int rc = SDRPP_OK;
    if (as_tick) {
        HostScope hs("tick plan");
        rc = arena_begin(c);
        if (rc) {
            c->pushes--;
            return rc;
        }
        plan_snapshot(c, snap);
        for (auto& kv : c->vfos) {
            for (auto& s : kv.second->st) { stream_rotate(s); }
        }
        fft_ring_rotate(c);
        c->tick_planning = true;
        c->tick_abort = false;
        c->emits.clear();
        c->plan_top = 2;
        block_bounds(c, count, nullptr);
        rc = ensure_iq_hist(c, iq_hist_need(c));
        if (!rc) {
            IqSrc src{ (const float2*)d_iq, (const float2*)c->iq_hist[c->iq_cur], c->iq_hist_cap, (long long)count };
            rc = do_fft(c, src, count);
            if (!rc && !c->tick_abort) {
                const CarryJob iqc = iq_carry_job(c, d_iq, count);
                if (c->vfos.empty()) {
                    std::vector<CarryJob> carry{ iqc };
                    CarryJob* d_carry = arena_push(c, carry);
                    if (!d_carry) { rc = fail(c, SDRPP_ERR_UNSUPPORTED, "job arena exhausted"); }
                    else { emit(c, 1, F_MISC, TR_CARRY, std::max(1, std::min((iqc.need * 2 + 1023) / 1024, 2048)), 1, 0, d_carry); }
                }
                else { rc = do_vfos_plan(c, src, count, iqc); }
            }
            if (!rc && !c->tick_abort) { rc = tick_results_plan(c); }
        }
        c->tick_planning = false;
        if (!rc && !c->tick_abort && c->plan_top > kTickDepth + 1) { c->tick_abort = true; }
        if (rc || c->tick_abort) {
            plan_restore(c, snap);
            c->emits.clear();
            if (!c->res[c->pushes % kResSlots].held) { c->res[c->pushes % kResSlots].ticket = 0; }
            as_tick = false;
            if (rc) {
                c->pushes--;
                return rc;
            }
            c->arena_off = 0;  // (the slot stays this tick's: only the next role table goes in)
        }
    }
    if (!as_tick) {
        // this block runs as an ordinary pass: everything queued first (the first of those ticks carries the landing copy), then the pass
        // behind them on the same stream
        if (!have_slot) {
            rc = arena_begin(c);
            if (rc) { return rc; }
        }
        rc = stage_pending_wait(c);
        if (!rc) { rc = tick_launch(c, land); }
        if (!rc) { rc = tick_drain(c); }
        if (land) { c->land_tick = c->ticks; }
        if (!rc) { rc = push_common(c, d_iq, count, nullptr); }
        // its results are where an ordinary pass leaves them (device buffers, readable after a synchronisation) and — with result flags —
        // also in the block's result slot like every other block's
        if (!rc && c->res_flags) { rc = tick_results_direct(c); }
        else {
            sdrpp_ctx::Result& R = c->res[c->pushes % kResSlots];
            if (!R.held) { R = sdrpp_ctx::Result{}; }
        }
        return rc;
    }
    // queue the roles level by level and launch this block's tick
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
    sdrpp_ctx::Result& R = c->res[c->pushes % kResSlots];
    if (R.ticket == c->pushes) { R.done_tick = c->ticks + (uint64_t)(c->plan_top - 1); }
    return rc;
}

}  // namespace

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

int sdrpp_abi_version(int* sizeof_vfo_desc) {
    if (sizeof_vfo_desc) { *sizeof_vfo_desc = (int)sizeof(sdrpp_vfo_desc); }
    return 1;
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
    for (int i = 0; i < kResSlots; i++) {
        if (c->res_host[i]) { (void)hipHostFree(c->res_host[i]); }
    }
    fft_ring_drop(c);
    for (int i = 0; i < 2; i++) {
        dev_free(c->iq_land[i]);
        dev_free(c->iq_land16[i]);
        if (c->land_ev[i]) { (void)hipEventDestroy(c->land_ev[i]); }
    }
    if (c->ev_copy) { (void)hipEventDestroy(c->ev_copy); }
    if (c->copy_stream) { (void)hipStreamDestroy(c->copy_stream); }
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
        rc = dev_alloc(c, &P.d_off, 1);
        if (rc) { return rc; }
        HIPCHK(c, hipMemset(P.d_off, 0, sizeof(float2)));
        P.seg_cap = (int)(cap / SDRPP_DEEMP_SEG) + 2;
        rc = dev_alloc(c, &P.d_seg, (size_t)P.seg_cap + 1);
        if (rc) { return rc; }
    }
    P.on = true;
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
    return SDRPP_OK;
}

int sdrpp_vfo_count(sdrpp_ctx* c) { return c ? (int)c->vfos.size() : SDRPP_ERR_INVALID; }

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
        int rc = stream_grow_hist(c, v.st[(size_t)idx], n - 1);
        if (rc) { return rc; }
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
        rc = dev_alloc(c, &a.d_last, 1);
        if (rc) { return rc; }
        HIPCHK(c, hipMemset(a.d_last, 0, sizeof(float2)));
        a.seg_cap = (int)(cap / SDRPP_DEEMP_SEG) + 2;
        rc = dev_alloc(c, &a.d_seg, (size_t)a.seg_cap + 1);
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

// pipelined mode: a block from host memory.  `src_dev`: device address of page-locked memory the samples can be fetched from in place
// (sdrpp_push_pinned_async); nullptr: `src_host` is copied into a page-locked staging slot first (the caller's buffer is free on return).
// `bytes_per_sample`: 8 (complex float) or 4 (interleaved int16).
static int tick_push_host(sdrpp_ctx* c, const void* src_host, const void* src_dev, int64_t count, int bytes_per_sample) {
    const int li = (int)((c->pushes + 1) % 3);
    if (!c->tick_land[li]) {
        int rc = dev_alloc(c, &c->tick_land[li], (size_t)c->max_push * 2 + 32);
        if (rc) { return rc; }
    }
    const size_t bytes = (size_t)count * (size_t)bytes_per_sample;
    if (!src_dev) {
        const int si = c->stage_cur;
        c->stage_cur = (c->stage_cur + 1) % kStageSlots;
        if (!c->stage_host[si]) {
            if (hipHostMalloc((void**)&c->stage_host[si], (size_t)c->max_push * 8 + 64, hipHostMallocMapped) != hipSuccess) { return fail(c, SDRPP_ERR_NOMEM, "page-locked staging buffer"); }
        }
        if (c->stage_tick[si]) { tick_wait_done(c, c->stage_tick[si]); }  // its last landing copy has run
        memcpy(c->stage_host[si], src_host, bytes);
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, c->stage_host[si], 0) != hipSuccess || !d) { return fail(c, SDRPP_ERR_HIP, "hipHostGetDevicePointer(staging) failed"); }
        src_dev = d;
        const CopyJob land{ src_dev, c->tick_land[li], (long long)bytes, bytes_per_sample == 4 ? 1 : 0, 0 };
        int rc = tick_push(c, c->tick_land[li], count, &land);
        c->stage_tick[si] = c->ticks;
        return rc;
    }
    const CopyJob land{ src_dev, c->tick_land[li], (long long)bytes, bytes_per_sample == 4 ? 1 : 0, 0 };
    return tick_push(c, c->tick_land[li], count, &land);
}

// The staging slot of the next block handed to the HOST to fill (pipelined mode): what tick_push_host's memcpy does, in the caller's hands
int sdrpp_push_stage(sdrpp_ctx* c, int64_t count, float** slot) {
    DeviceScope dev_scope_(c);
    if (!c || !slot) { return SDRPP_ERR_INVALID; }
    if (!c->pipelined) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_stage: the context is not in pipelined mode"); }
    if (count <= 0 || count > c->max_push) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_stage: count %lld out of range (max_push %lld)", (long long)count, (long long)c->max_push); }
    const int si = c->stage_cur;
    if (!c->stage_host[si]) {
        if (hipHostMalloc((void**)&c->stage_host[si], (size_t)c->max_push * 8 + 64, hipHostMallocMapped) != hipSuccess) { return fail(c, SDRPP_ERR_NOMEM, "page-locked staging buffer"); }
    }
    if (c->stage_tick[si]) { tick_wait_done(c, c->stage_tick[si]); }  // its last landing copy has run
    c->stage_open = si;
    *slot = reinterpret_cast<float*>(c->stage_host[si]);
    return SDRPP_OK;
}
static int push_staged_impl(sdrpp_ctx* c, int64_t count);
int sdrpp_push_staged(sdrpp_ctx* c, int64_t count) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    c->stage_pending = nullptr;
    return push_staged_impl(c, count);
}
int sdrpp_push_staged_when(sdrpp_ctx* c, int64_t count, const volatile uint32_t* pending) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    c->stage_pending = pending;
    const int rc = push_staged_impl(c, count);
    c->stage_pending = nullptr;  // (a failed plan returns without having waited: the caller joins its own threads)
    return rc;
}
static int push_staged_impl(sdrpp_ctx* c, int64_t count) {
    if (!c->pipelined || c->stage_open < 0 || c->stage_open != c->stage_cur) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_staged without an open staging slot (sdrpp_push_stage)"); }
    if (count <= 0 || count > c->max_push) { return fail(c, SDRPP_ERR_INVALID, "sdrpp_push_staged: count out of range"); }
    const int si = c->stage_open;
    c->stage_open = -1;
    c->stage_cur = (c->stage_cur + 1) % kStageSlots;
    const int li = (int)((c->pushes + 1) % 3);
    if (!c->tick_land[li]) {
        int rc = dev_alloc(c, &c->tick_land[li], (size_t)c->max_push * 2 + 32);
        if (rc) { return rc; }
    }
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, c->stage_host[si], 0) != hipSuccess || !d) { return fail(c, SDRPP_ERR_HIP, "hipHostGetDevicePointer(staging) failed"); }
    const CopyJob land{ d, c->tick_land[li], (long long)((size_t)count * 8), 0, 0 };
    int rc = tick_push(c, c->tick_land[li], count, &land);
    c->stage_tick[si] = c->ticks;
    return rc;
}

int sdrpp_push(sdrpp_ctx* c, const float* iq_host, int64_t count) {
    DeviceScope dev_scope_(c);
    int rc = push_args_ok(c, iq_host, count);
    if (rc) { return rc; }
    if (c->pipelined) { return count == 0 ? SDRPP_OK : tick_push_host(c, iq_host, nullptr, count, 8); }
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
            return tick_push_host(c, iq_pinned, nullptr, count, 8);
        }
        return tick_push_host(c, iq_pinned, dptr, count, 8);
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
    if (c->pipelined && c->land_tick) {  // every landing copy so far has run once its tick is complete
        tick_wait_done(c, c->land_tick);
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
    if (c->pipelined) { return tick_push(c, iq_dev, count, nullptr); }  // read in place, one tick from now at the earliest
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
    if (c->pipelined) { return count == 0 ? SDRPP_OK : tick_push_host(c, iq_host, nullptr, count, 4); }
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
    if (!c || result_flags < 0 || result_flags > 7) { return SDRPP_ERR_INVALID; }
    if (on && c->deferred) { return fail(c, SDRPP_ERR_INVALID, "deferred and pipelined processing exclude each other"); }
    int rc = flush_pending(c);  // (drains the queue when the mode is being left)
    if (rc) { return rc; }
    for (int i = 0; i < kResSlots; i++) {
        if (c->res[i].held) { return fail(c, SDRPP_ERR_INVALID, "results of block %llu are still held", (unsigned long long)c->res[i].ticket); }
    }
    c->pipelined = on != 0;
    c->res_flags = on ? result_flags : 0;
    return SDRPP_OK;
}
uint64_t sdrpp_ticket(sdrpp_ctx* c) { return c ? c->pushes : 0; }
int sdrpp_pipeline_flush(sdrpp_ctx* c) {
    DeviceScope dev_scope_(c);
    if (!c) { return SDRPP_ERR_INVALID; }
    return c->pipelined ? tick_drain(c) : SDRPP_OK;
}
static sdrpp_ctx::Result* result_of(sdrpp_ctx* c, uint64_t ticket) {
    if (!c || ticket == 0 || ticket > c->pushes) { return nullptr; }
    sdrpp_ctx::Result& R = c->res[ticket % kResSlots];
    return R.ticket == ticket ? &R : nullptr;
}
int sdrpp_result_ready(sdrpp_ctx* c, uint64_t ticket) {
    sdrpp_ctx::Result* R = result_of(c, ticket);
    if (!R) { return c ? fail(c, SDRPP_ERR_NOT_FOUND, "no results for block %llu (not gathered, overwritten, or processed as an ordinary pass)", (unsigned long long)ticket) : SDRPP_ERR_INVALID; }
    return (R->done_tick <= c->ticks && tick_is_done(c, R->done_tick)) ? 1 : 0;
}
int sdrpp_result_wait(sdrpp_ctx* c, uint64_t ticket, sdrpp_result* out) {
    DeviceScope dev_scope_(c);
    if (!c || !out) { return SDRPP_ERR_INVALID; }
    sdrpp_ctx::Result* R = result_of(c, ticket);
    if (!R) { return fail(c, SDRPP_ERR_NOT_FOUND, "no results for block %llu (not gathered, overwritten, or processed as an ordinary pass)", (unsigned long long)ticket); }
    while (R->done_tick > c->ticks) {  // its last levels have not been launched yet: nothing more is coming, run them without new input
        int rc = arena_begin(c);
        if (!rc) { rc = tick_launch(c, nullptr); }
        if (rc) { return rc; }
        if (c->tickq.empty() && R->done_tick > c->ticks) { return fail(c, SDRPP_ERR_HIP, "internal: block %llu cannot complete", (unsigned long long)ticket); }
    }
    tick_wait_done(c, R->done_tick);
    if (!tick_is_done(c, R->done_tick)) { return fail(c, SDRPP_ERR_HIP, "tick %llu did not complete", (unsigned long long)R->done_tick); }
    R->held = true;
    const char* base = c->res_host[ticket % kResSlots];
    out->ticket = ticket;
    out->n_vfo = (int)R->ids.size();
    out->ids = R->ids.data();
    out->offsets = R->offsets.data();
    out->counts = R->counts.data();
    out->samples = reinterpret_cast<const float*>(base);
    out->n_lines = R->n_lines;
    out->fft_size = c->fft_size;
    out->data_width = c->data_width;
    const bool zo = R->n_lines > 0 && (c->res_flags & 2) && c->data_width > 0;
    out->zoomed = zo ? reinterpret_cast<const float*>(base + R->off_zoomed) : nullptr;
    out->index = zo ? reinterpret_cast<const int32_t*>(base + R->off_index) : nullptr;
    out->raw = (R->n_lines > 0 && (c->res_flags & 4)) ? reinterpret_cast<const float*>(base + R->off_raw) : nullptr;
    return SDRPP_OK;
}
int sdrpp_result_release(sdrpp_ctx* c, uint64_t ticket) {
    sdrpp_ctx::Result* R = result_of(c, ticket);
    if (!R) { return c ? SDRPP_ERR_NOT_FOUND : SDRPP_ERR_INVALID; }
    R->held = false;
    R->ticket = 0;
    return SDRPP_OK;
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
