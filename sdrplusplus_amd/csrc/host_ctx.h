// The context: constants, per-stream / per-VFO state, struct sdrpp_ctx (what one IQ stream on one GPU owns).
// Part of the one translation unit sdrpp_gpu.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

constexpr int kArenaSlots = 24;     // >= kTickDepth + 2: a block's job tables are read for kTickDepth ticks after their upload
constexpr size_t kArenaBytes = 4u << 20;
constexpr int kRing = 4;            // pipelined mode: buffers per per-block stream (a consumer runs at most 2 ticks behind its producer; + the gather)
constexpr int kTickDepth = 18;      // pipelined mode: levels 0 .. kTickDepth of a block (see the level table at emit()): pre-processing chain 3 + VFO chain 5 + AF chain 5 + result copy
constexpr int kResSlots = SDRPP_RESULT_SLOTS;       // pipelined mode: page-locked result slots, one per LAUNCH GROUP (sdrpp_set_pipeline_group: 1 .. kGroupMax blocks) whose results the host has not released yet
constexpr int kGroupMax = SDRPP_GROUP_MAX;          // pipelined mode: blocks one launch may carry
constexpr int kResMeta = kResSlots * kGroupMax;     // ... and what the host knows about every block of those groups (ring by ticket)
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
    int wlevel = 0;  // pipelined mode: level of the role that WRITES this stream in the current block (0: not written by a role of this plan)
    float* prev_data = nullptr;  // pipelined mode: the data buffer of the block before (stream_rotate) and its sample count — where a piped back end
    int prev_n = 0;              // finds the tail of the previous block (do_vfos_plan: pipe_in)
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
    std::vector<float> chan_stale;  // what the channel filter's delay line held when the filter was last bypassed (sdrpp_vfo_set_channel_taps)
    float* d_audio = nullptr;
    int audio_ntaps = 0;
    // device loop state: [AgcState agc][AgcState carrier][float dc]
    char* d_state = nullptr;
    double theta2 = 0.0, phi2 = 0.0;
    // streams: 0..nstages-1 decimator outputs (index 0 also used by the rotate-only path), then poly, chan, dem, out
    std::vector<Stream> st;
    int i_first = 0, i_poly = -1, i_chan = -1, i_dem = -1, i_out = -1, i_if = 0;
    int lvl_if = 1, lvl_out = 1, lvl_af = 1;  // levels (do_vfos_plan) at which the IF stream / the demodulator's output / the AF chain's output of the most recent block are written
    std::vector<int> tk_if, tk_af;  // a launch group of several pushes: cumulative sample counts of the IF / demodulator stream and of the AF chain's output at every push end
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
        float2* d_last = nullptr;  // Deemphasis::lastOut: two slots, read from [state_cur], written to the other (DeempJob)
        int state_cur = 0;
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
    hipStream_t side_stream = nullptr;   // sdrpp_device_copy: copies that may come from ANOTHER thread than the one driving the context (created with the context, never changed)
    std::mutex side_mtx;                  // ... one at a time
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
        float2* d_off = nullptr;  // DCBlocker::offset: two slots like Vfo::Af::d_last
        int state_cur = 0;
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
    std::vector<Vfo*> vfo_list;           // the same VFOs in id order (rebuilt on add / remove): what the per-block loops walk
    int next_id = 1;
    // cached stage-1 job tap arrays, keyed by membership signature
    std::map<std::string, float2*> s1_tap_cache;  // key = 16 raw bytes: two independent 64-bit hashes of (kind, member ids, increments)

    // ---- pipelined ("tick") execution: one launch per block, the stages of consecutive blocks skewed over consecutive launches
    //      (tick_kernels.h; sdrpp_set_pipelined) ----
    struct RoleLaunch { TickEntry e; size_t lds; int level; int fam; bool to_host; };  // to_host: a copy into a page-locked result slot
    struct Result {                       // what the host knows about one block's results (ring by ticket); the bytes live in its group's slot
        uint64_t ticket = 0;              // 0: entry free
        uint64_t done_tick = 0;           // its last level has run when this many ticks have completed
        bool held = false;                // handed out by sdrpp_result_wait, not yet released
        uint64_t group = 0;               // its launch group (alive while res_live still holds a region of that id)
        const char* base = nullptr;       // host address of the group's region in the result ring (stays valid for a held block when the ring is replaced by a larger one)
        int epoch = 0;                    // which ring `base` points into (res_epoch when the results were planned)
        std::vector<int> ids, counts;
        std::vector<int64_t> offsets;
        int n_lines = 0;
        int fft_size = 0, data_width = 0, flags = 0;        // what the block was PLANNED with (the view / FFT size / result flags may change before it is collected)
        size_t off_zoomed = 0, off_index = 0, off_raw = 0, off_iq = 0;  // byte offsets in the slot
        int n_iq = 0;                                        // pre-processed IQ samples delivered (result flag 8)
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
    // FM back ends as ONE role of the tick (last decimator -> resampler -> channel filter -> discriminator + audio low-pass in a workgroup, the
    // streams between them in LDS: pipe_kernels.h) instead of four roles on four ticks.  Bit-identical, two levels shallower, half the
    // inter-stage traffic — and SLOWER wherever it was measured (profiles/r04j_pipe_role_sweep.log): a segment's four coupled stages are one
    // long dependent walk (10^6-sample blocks: tick 82.6 us with 256 workgroups, 70.8 with 512, against 50.0 for the four roles; sr/200 blocks
    // 32 us against 12), and blocks shorter than a filter history per VFO (cfg 4's NFM channels at sr/200) would fall back to ordinary passes.
    // Off by default; SDRPP_GPU_TICK_PIPE=1 for measurements (tests/test_pipelined.py keeps it bit-identical).
    // Large blocks in pipelined mode: the ratio-32 front end in its 16 x 16 x 4 shape walking its tiles (vfo_frontcm16w_body: 84 registers, 31 KB of
    // LDS), so that the whole tick runs in the FOUR-wavefronts-per-SIMD build of the tick kernel (tick_kernel<2>) instead of the three the
    // 32 x 32 x 2 front end's 168 registers / 41 KB allow.  Built to test the reading that the SIMDs idle behind latency with three wavefronts
    // each — and measured (10^6-sample blocks, 200 steps, profiles/r04n_*, r04o_*): the SAME roles at three and at four wavefronts per SIMD take
    // 55.8 / 55.9 us per tick (768 front-end workgroups; 57.5 / 55.6 with 1 024): occupancy is NOT what limits the tick, and the 16 x 16 x 4 front
    // end itself costs 5.5 us more than the 32 x 32 x 2 one (49.9 us: twice the LDS reads and twice the vector instructions per matrix cycle).
    // Bit-identical (tests/test_pipelined.py), off by default: a measurement switch.
    bool tick_fcm16w = getenv("SDRPP_GPU_TICK_FCM16W") ? atoi(getenv("SDRPP_GPU_TICK_FCM16W")) != 0 : false;
    int tick_fcm16w_blocks = getenv("SDRPP_GPU_TICK_FCM16W_BLOCKS") ? std::max(1, atoi(getenv("SDRPP_GPU_TICK_FCM16W_BLOCKS"))) : 512;
    int tick_land_blocks = getenv("SDRPP_GPU_TICK_LAND_BLOCKS") ? std::max(1, atoi(getenv("SDRPP_GPU_TICK_LAND_BLOCKS"))) : 64;  // workgroups of a tick's landing copy (host-fed blocks), at most
    int tick_lds_cap = 24 * 1024;       // LDS window of the many-phase resampler as a role of a tick (launch_polyc)
    int tick_lds_cap_fir = 40 * 1024;   // ... of the register-blocked FIR roles (launch_fir)
    bool tick_pipe = getenv("SDRPP_GPU_TICK_PIPE") ? atoi(getenv("SDRPP_GPU_TICK_PIPE")) != 0 : false;
    int tick_pipe_blocks = getenv("SDRPP_GPU_TICK_PIPE_BLOCKS") ? std::max(1, atoi(getenv("SDRPP_GPU_TICK_PIPE_BLOCKS"))) : 256;
    long arena_begins = 0;                // blocks planned so far (block_bounds: one per ordinary pass / per block of a pipelined run)
    int arena_allocs = 0;
    long test_fail_pass = 0;              // SDRPP_GPU_TEST_FAIL_ARENA (see arena_push)
    int test_fail_alloc = 0;
    bool pre_ref_order = false;           // survives sdrpp_preproc_configure (which rebuilds `pre`)
    float2 pre_dc_last = { 0.0f, 0.0f };  // the DC blocker's estimate when it was last looked at (sdrpp_preproc_reconfigure: the reference's block object lives on through every re-plan of the chain)
    std::vector<const volatile uint32_t*> stage_pend;  // sdrpp_push_staged_when: the block's first launch waits (on the host) for these words to reach 0
    bool rot_exact_single = getenv("SDRPP_GPU_ROT_EXACT_SINGLE") != nullptr;  // measurement switch: the one-wavefront form of the reference rotator
    // VFOs per workgroup of vfo_rotate_exact4_kernel (1 .. 64).  The chain wavefront costs the same for 1 or 64 VFOs (a lane each); the three
    // wavefronts that apply the phases take ~100 cycles per VFO and chunk: beyond ~16 VFOs they, not the chain, set the pace of the workgroup
    // and the input is 8 bytes per sample however often it is read.
    // 32-output tiles per front-end job up to which the ratio-32 front end runs in its small-block shape (vfo_frontcm16_body); 0: never.
    // Unset: 256 for ordinary passes (sr/200 pushes 764 -> 814 MS/s) and for pipelined blocks that are read where they lie in device memory
    // (3.48 -> 3.96 GS/s), never for blocks the tick's landing copy fetches from host memory — workgroups that share a CU with a landing-copy
    // workgroup start 8 us late, which the longer front end hides and the short one does not (DESIGN_HISTORY.md 4b, profiles/r03zl-r03zn).
    int fcm16_max_tiles = getenv("SDRPP_GPU_FCM16_MAX_TILES") ? atoi(getenv("SDRPP_GPU_FCM16_MAX_TILES")) : -1;
    bool plan_block_from_host = false;    // the block being planned reaches the device through a landing copy
    // phases handed over per full chunk: every 4th / 8th / 16th (cfg 4's 43 SSB channels, the family's time per 2^20 samples: 14.2 / 13.4 / 13.0 ms,
    // profiles/r03x_*; the applying wavefronts take up to SKIP - 1 steps per sample themselves, so fewer VFOs per workgroup go with a larger stride)
    int rot_exact_skip = getenv("SDRPP_GPU_ROTX_SKIP") ? atoi(getenv("SDRPP_GPU_ROTX_SKIP")) : 16;
    int rot_exact_vpw = [] { const char* e = getenv("SDRPP_GPU_ROTX_VPW"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    // Grids of a block that is planned while the device has nothing (factor 4) or one launch (2) in flight: its roles will run in ticks that hold little
    // else — the fill of the pipeline after a pause, a host slower than the device — where the grid rules above (sized for a tick that ~8 roles of
    // as many blocks share) leave most of the device idle.  The results do not depend on the grids.  SDRPP_GPU_TICK_SPARSE=0: off (measurements).
    int plan_sparse = 1;
    bool tick_sparse_boost = getenv("SDRPP_GPU_TICK_SPARSE") ? atoi(getenv("SDRPP_GPU_TICK_SPARSE")) != 0 : false;  // (measured: no gain on the 20-step line, profiles/r06j_sparse_ab.log — off)
    bool tick_planning = false;           // a block is being planned for the tick queue: emit() queues, plain launches abort the plan
    bool tick_abort = false;              // ... and met a launch that has no role in the tick kernel: the block runs as an ordinary pass
    int plan_top = 0;                     // highest level + 1 the block being planned uses
    int plan_lvl0 = 0;                    // levels the pre-processing chain of the block being planned takes in front of the FFT branch / VFO bank (pipelined mode; 0 in a pass)
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
    // Results in page-locked host memory: ONE ring of kResSlots x (what a launch of max_push samples can deliver), handed out in allocation order, a
    // region per launch group sized for what that group really delivers — 24 launches at the stream cap, ten times as many at the reference's block
    // size (a host that collects a fixed number of BLOCKS behind its pushes must not lose results because the groups happened to be small).  A region
    // is taken back, oldest first, when the ring comes round to it; one the host still holds fails the push that needs it.
    char* res_ring = nullptr;             // host address
    char* res_ring_dev = nullptr;         // ... and as the device sees it
    size_t res_ring_bytes = 0;
    size_t res_cap = 0;                   // what the largest possible launch delivers (tick_results_need)
    size_t res_head = 0;                  // next free byte
    int res_epoch = 0;                    // counts the rings (tick_results_ensure replaces the ring by a larger one)
    struct ResRegion { uint64_t gid; size_t off, bytes; int held; };
    std::deque<ResRegion> res_live;       // regions in use, in allocation order (front = oldest)
    struct ResRetired { char* ring; int epoch; int held; };
    std::vector<ResRetired> res_retired;  // smaller rings that held blocks the host had not released when the ring grew: its pointers stay valid until it does
    Result res[kResMeta];
    // ---- several blocks per launch (sdrpp_set_pipeline_group): pushes are HELD — nothing planned, nothing launched — until the group goes out as ONE
    //      block of the tick queue whose reference-block ends are the push ends (what a deferred pass does with its staged pushes, plan_push.h) ----
    int group_max = 1;                    // blocks per launch, at most
    int group_adaptive = 0;               // 1: a group goes out as soon as the device has fewer than two launches in flight (a host slower than the device: one block per launch)
    bool stage_pend_stable = false;       // sdrpp_set_pipeline_group flag 2: the words handed to sdrpp_push_staged_when stay valid until their block has been LAUNCHED — a held push does not wait for its copy
    struct Held {
        int kind = -1;                    // -1: nothing held; 0: device memory read in place; 1 / 2: float / int16 samples in a page-locked staging slot; 3: the caller's page-locked memory
        const char* base = nullptr;       // kind 0 / 3: address of the first block (the following ones are contiguous with it)
        int stage_slot = -1;              // kind 1 / 2
        int64_t total = 0;                // samples held
        std::vector<int> ends;            // cumulative end of every held push
    } held;
    std::vector<int> grp_ends;            // push ends of the group being planned (empty / one entry: a single block)
    int64_t plan_fft_pos0 = 0, plan_fft_next0 = 0;  // frame position in front of the block being planned (do_fft): which push of a group completes which line
    uint64_t groups = 0;                  // launch groups planned so far (a group's result slot: groups % kResSlots)
    int64_t stat_groups = 0, stat_group_blocks = 0, stat_group_max = 0;  // groups of more than one block, the blocks in them, the largest
    // Completion of a tick whose roles wrote RESULTS into page-locked host memory, as the host may rely on it: an event recorded behind the
    // launch.  The flag the kernel itself publishes (h_tick_flag) says that every workgroup has finished and its stores are acknowledged — but
    // a result block is megabytes of posted writes over the bus from every XCD, and the four bytes of the flag were measured to overtake them by
    // up to ~100 us (round 4: a host that copied a VFO block within microseconds of the flag met the slot's previous content).  The command
    // processor's end-of-kernel release behind an event is the ordering HIP guarantees for kernel writes to host memory.
    static constexpr int kTickEvents = 32;
    hipEvent_t tick_ev[kTickEvents] = {};
    uint64_t tick_ev_tick[kTickEvents] = {};   // the tick the event was last recorded behind (0: never)
    hipEvent_t tick_ev_start[kTickEvents] = {}; // timing on: the start event of that tick's launch (the pair is read out when the entry comes round again, or at timing_flush)
    int tick_ev_next = 0;
    uint64_t tick_ev_last = 0;                 // the newest tick an event stands behind
    // measurement switches: an event behind every n-th tick with results (a waiter takes the first event at or behind its tick, recording one on
    // demand); the event as the tick launch's own completion signal (hipExtLaunchKernelGGL's stop event) instead of a packet of its own
    int tick_ev_every = getenv("SDRPP_GPU_TICK_EVENT_EVERY") ? std::max(1, atoi(getenv("SDRPP_GPU_TICK_EVENT_EVERY"))) : 1;
    bool tick_ev_ext = getenv("SDRPP_GPU_TICK_EVENT_EXT") ? atoi(getenv("SDRPP_GPU_TICK_EVENT_EXT")) != 0 : true;  // (15.8 against 15.2 GS/s with the event as a packet of its own: profiles/r04h_*)
    int tick_ev_skipped = 0;
    void* bank_plan = nullptr;            // the context's BankPlan (plan_vfo.h), re-used block after block
    void (*bank_plan_free)(void*) = nullptr;
    // how the blocks of a pipelined run were executed (sdrpp_pipeline_stats: tests and bench.py assert the mode they mean to measure)
    int64_t stat_tick_blocks = 0, stat_pass_blocks = 0, stat_crowded = 0, stat_last_depth = 0, stat_set2 = 0, stat_last_table_bytes = 0;
    int64_t stat_role_wgs[64] = {};

    // timing
    bool timing = false;
    unsigned timing_mask = 0xffffffffu;  // families whose launches are bracketed by events while timing is on
    std::vector<TimingPair> tpairs;
    std::vector<hipEvent_t> ev_pool;
    double fam_ms[SDRPP_NUM_KERNEL_FAMILIES] = {};
    int64_t fam_launch[SDRPP_NUM_KERNEL_FAMILIES] = {};
};
