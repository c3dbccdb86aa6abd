/*
 * sdrpp_gpu.h — C-ABI of the MI355X (gfx950) implementation of SDR++'s streaming-DSP hot path.
 *
 * One `sdrpp_ctx` = one wideband IQ stream on one GPU = what one `IQFrontEnd` owns in the reference
 * (core/src/signal_path/iq_frontend.h:12-109): the FFT -> log-power -> waterfall-line branch and N per-VFO channelisers
 * (dsp::channel::RxVFO, core/src/dsp/channel/rx_vfo.h) followed by the radio module's demodulators
 * (decoder_modules/radio/src/demodulators/{wfm,nfm,am,usb,lsb,dsb}.h).  All arithmetic runs in hand-written HIP
 * kernels; there is NO CPU fallback — every entry point returns SDRPP_ERR_NO_DEVICE when no gfx950 device exists.
 *
 * Conventions: plain C, opaque context, `int` return (0 = OK, negative = error, see sdrpp_strerror), no exceptions,
 * no C++ or torch types.  Complex IQ is interleaved float32 {re, im} (dsp::complex_t, core/src/dsp/types.h:6-92);
 * audio is interleaved float32 {l, r} (dsp::stereo_t, types.h:94-127).  A context is thread-compatible: one caller
 * thread at a time (the reference serialises reconfiguration with block::ctrlMtx + tempStop/tempStart,
 * core/src/dsp/block.h:46-62; the host block wrappers in sdrplusplus_amd/host keep that rule).
 *
 * The taps / window / NCO constants are passed IN by the caller, designed with the reference's own double-precision
 * host maths (dsp/taps, dsp/window, dsp/multirate/decim/plans.h) so that they stay bit-identical to what an
 * SDR++ build computes; sdrpp_design_* below restate that maths for callers that do not link SDR++'s headers.
 */
#ifndef SDRPP_GPU_H
#define SDRPP_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRPP_OK 0
#define SDRPP_ERR_NO_DEVICE (-1)   /* no HIP device / not gfx950 / HIP runtime error at create */
#define SDRPP_ERR_INVALID (-2)     /* bad argument or call sequence */
#define SDRPP_ERR_NOMEM (-3)       /* hipMalloc failed */
#define SDRPP_ERR_HIP (-4)         /* HIP runtime error (message in sdrpp_last_error) */
#define SDRPP_ERR_UNSUPPORTED (-5) /* parameter outside what the kernels implement */
#define SDRPP_ERR_NOT_FOUND (-6)   /* unknown VFO id (IQFrontEnd::removeVFO logs the same condition, iq_frontend.cpp:164-167) */

typedef struct sdrpp_ctx sdrpp_ctx;

/* ---- lifecycle ----------------------------------------------------------------------------------------------------- */
/* Replaces IQFrontEnd::init's allocation half (iq_frontend.cpp:17-71).  `max_push` = largest sample count one
 * sdrpp_push* call will carry (the reference's streams carry <= 1 000 000, core/src/dsp/stream.h:9; device-resident
 * callers may use larger batches). */
int sdrpp_device_count(void);   /* usable devices (0: none — nothing here runs without one) */
int sdrpp_create(int device, int64_t max_push, sdrpp_ctx** ctx);
int sdrpp_destroy(sdrpp_ctx* ctx);
const char* sdrpp_strerror(int code);
const char* sdrpp_last_error(const sdrpp_ctx* ctx);
/* Run all work of this context on the caller's HIP stream (hipStream_t cast to void*), e.g. torch's current stream;
 * NULL restores the context-owned stream. */
int sdrpp_set_stream(sdrpp_ctx* ctx, void* hip_stream);
int sdrpp_sync(sdrpp_ctx* ctx);
/* ABI self-description for foreign-function bindings: returns the ABI version and, if non-NULL, sizeof(sdrpp_vfo_desc). */
#define SDRPP_ABI_VERSION 2   /* 2: sdrpp_vfo_desc.nco_mode, sdrpp_pipeline_stats */
int sdrpp_abi_version(int* sizeof_vfo_desc);
/* Human-readable device name / arch into buf (for logs and bench records). */
int sdrpp_device_info(sdrpp_ctx* ctx, char* buf, int buflen);

/* ---- host-side design maths (pure CPU, double precision; restates the reference's header maths) ---------------------- */
/* dsp/taps/low_pass.h:7-11, high_pass.h:7-15 (Nuttall-windowed sinc, tap count int(3.8*sr/tw)).  Return the tap count;
 * write min(count, max) taps.  Call with max = 0 to size the buffer. */
int sdrpp_design_low_pass(double cutoff, double trans_width, double sample_rate, int odd_tap_count, float* taps, int max);
int sdrpp_design_high_pass(double cutoff, double trans_width, double sample_rate, int odd_tap_count, float* taps, int max);
/* iq_frontend.cpp:280-291: kind 0 RECTANGULAR, 1 BLACKMAN, 2 NUTTALL; includes the (-1)^i fftshift factor. */
int sdrpp_design_fft_window(int kind, int nz, float* window);
/* iq_frontend.h:59-63 */
void sdrpp_design_reshape_params(double sample_rate, int fft_size, double fft_rate, int* skip, int* nz);
/* frequency_xlator.h:17,28: phaseDelta = (float)cos(w), (float)sin(w), w = 2*pi*(offset_hz/sample_rate). */
void sdrpp_design_phase_delta(double offset_hz, double sample_rate, float* re, float* im);
/* rational_resampler.h:120-165: picks the power-of-two pre-decimation, the polyphase L/M and designs the polyphase taps
 * (already multiplied by L).  mode: 0 BOTH, 1 DECIM_ONLY, 2 RESAMP_ONLY, 3 NONE.  `max_ratio` = 1 << plans_len
 * (power_decimator.h:29-31; 8192 for the reference's tables).  Returns tap count (0 if no polyphase stage). */
int sdrpp_design_resampler(double in_sr, double out_sr, int max_ratio, int* mode, int* predec_ratio, int* interp, int* decim,
                           float* taps, int max);
/* filter/deephasis.h:90-93: alpha = dt / (tau + dt), dt = 1.0f / samplerate (float arithmetic as in the reference). */
float sdrpp_design_deemphasis_alpha(double tau, double sample_rate);
/* gui/widgets/waterfall.cpp:891-894 (view -> bin window) */
void sdrpp_design_waterfall_view(double view_offset, double view_bandwidth, double whole_bandwidth, int raw_fft_size,
                                 int* draw_data_start, int* draw_data_size);

/* ---- what in the reference depends on how its stream is cut into blocks ---------------------------------------------------------------
 * Two operations of the path are block-size dependent in the reference: loop::AGC's look-ahead on clipping scans to the end of the
 * CURRENT block (agc.h:91-104; AM and SSB demodulators), and VOLK's rotator renormalises its phase at the end of every call
 * (frequency_xlator.h:43-50).  ref_block > 0: every push is treated as consecutive reference blocks of `ref_block` input samples (the
 * last one may be shorter), whatever the push size — e.g. sample_rate / 200 for a file source (file_source/src/main.cpp:157) — and the
 * block ends are carried through the decimators / resampler down to the demodulator's rate.  0 (default): one push = one block. */
int sdrpp_set_reference_block(sdrpp_ctx* ctx, int ref_block);
/* NCO of the frequency translations (RxVFO's xlator, SSB's second xlator).
 *   SDRPP_NCO_CLOSED_FORM (default): phase = arg(phaseDelta) * n evaluated in float64, folded into the first filter's taps; exact
 *     frequency, no drift.  FM / AM outputs agree with the reference to ~1e-7 for any run length; the raw IF and an SSB product detector
 *     additionally see the reference rotator's own rounding drift (1e-10 .. 2e-9 rad per sample, linear in time), which this mode does not have.
 *     VALIDITY WINDOW against the reference for SSB / DSB audio and the raw IF — MEASURED by tests/test_bench_geometry_gpu.py::
 *     test_closed_form_nco_validity_window_vs_pinned_oracle (BASELINE cfg 4, 61.44 MS/s, 42 USB channels, pipelined, against the oracle pinned to
 *     the reference's own rotator; windows of 10^5 INPUT samples since the VFO was added / reset, errors relative to the RMS of the reference
 *     stream): audio inside BASELINE.json's 1e-5 over the first 2e5 input samples for every channel; once the channel filter has filled
 *     (10^6 samples) the difference grows by 2.8e-8 .. 1.05e-5 per 10^5 samples (median channel 1.5e-6), the raw IF by 1.0e-7 .. 3.1e-5
 *     (median 4.5e-6): the worst channel leaves 1e-5 in the third window, the median channel after ~1.2e6 input samples (20 ms of the stream);
 *     against the reference with its rotator replaced by an exact NCO it stays at
 *     3e-7 for any length.  A host that needs SSB / raw-IF parity with the reference's OWN phase sequence beyond that window selects
 *     SDRPP_NCO_REFERENCE_ROTATOR for those channels (sdrpp_vfo_desc.nco_mode = 2: per VFO, the FM / AM channels of the bank stay on the fast path).
 *   SDRPP_NCO_REFERENCE_ROTATOR: the reference's float recursion itself (VOLK generic rotator2: phase *= phaseDelta in float, renormalised
 *     every 512 samples and at the end of every block), one lane per VFO at the full input rate, then the plan's stages as plain FIRs.
 *     Reproduces the reference's phase sequence — IF and SSB parity ~1e-7 for any run length — at a few times real time instead of
 *     thousands: a parity mode.  Set sdrpp_set_reference_block as well: the renormalisation points are the reference's block ends.
 *     Why there is no third, "re-seeded" mode (closed form inside a reference block, phase re-seeded from the recursion's value at every block
 *     end): the block-end phases ARE the recursion — N dependent float multiply-adds per block and VFO whatever is done with the samples, and a
 *     dependent packed multiply + add is 21 cycles on this SIMD (tools/probe/chain_latency_probe.hip), i.e. <= 115 MS/s for ANY scheme that
 *     follows the reference's phase sequence; the reference-rotator kernel runs at 26 cycles per sample (cfg 4: 67-69 MS/s with every SSB channel
 *     exact), so the re-seeded form could gain at most 1.2x while giving up exactness inside the block (its error would saw-tooth to ~1.7e-5 per
 *     307 200-sample block unless the drift were interpolated as well).  Not built; the numbers are in DESIGN.md 5.
 * Can only be changed while the context has no VFO. */
#define SDRPP_NCO_CLOSED_FORM 0
#define SDRPP_NCO_REFERENCE_ROTATOR 1
int sdrpp_set_nco_mode(sdrpp_ctx* ctx, int mode);
/* How the per-VFO filters behind the front end are launched for FM VFOs (WFM / NFM: last decimator -> resampler -> channel filter ->
 * discriminator + audio low-pass; the chain of rx_vfo.h:20-66 + demod/broadcast_fm.h, demod/fm.h).
 *   on (default): as ONE launch where that pays — the four stages run as a pipeline inside a workgroup, the streams between them stay in
 *     LDS (csrc/pipe_kernels.h); the library decides push by push (large pushes and launch-bound small ones).
 *   off: one launch per stage, intermediate streams in HBM.
 * Results are bit-identical either way (same tap tables, same summation order); a switch for measurements and for the test that says so
 * (on >= 2, a test hook: always pipelined, `on` segments per VFO). */
int sdrpp_set_backend_pipeline(sdrpp_ctx* ctx, int on);

/* ---- FFT -> log-power -> waterfall line (replaces Reshaper + Handler + IQFrontEnd::handler, iq_frontend.cpp:248-309,
 *      and WaterFall::pushFFT's doZoom + palette index, waterfall.cpp:65-90, 889-906) --------------------------------------- */
/* Numerics: the reference links libfftw3f / libvolk, neither of which is part of its tree (unpinned distro packages), so "bit-exact" for
 * this branch is defined against a fully specified float32 FFT + log2 that the test oracle and these kernels
 * share operation for operation; against a float64 DFT it is accurate to < 4e-7 relative (tests/test_oracle_kat.py), i.e. palette
 * indices agree with ANY correct FFT to +-1 of 10^6 levels.
 * fft_size: power of two 1024..1048576.  Frame k covers stream samples [k*(nz+skip), k*(nz+skip)+nz) since the last
 * configure/reset (reshaper.h:101-128); fftIn[nz:fft_size] is zero (iq_frontend.cpp:301).  `window` has nz floats.
 * Reconfiguring drops the partial frame, like updateFFTPath's tempStop/tempStart. */
int sdrpp_fft_configure(sdrpp_ctx* ctx, int fft_size, int nz, int skip, const float* window);
int sdrpp_fft_disable(sdrpp_ctx* ctx);
/* doZoom(draw_data_start, draw_data_size, fft_size, data_width) then palette index
 * (int)((clamp(v, wf_min, wf_max) - wf_min) / (wf_max - wf_min) * 999999).  data_width = 0 disables the zoomed outputs. */
int sdrpp_fft_set_view(sdrpp_ctx* ctx, int draw_data_start, int draw_data_size, int data_width, float wf_min, float wf_max);
/* Lines produced by the most recent push (each overwrites the previous push's). */
int sdrpp_fft_lines(sdrpp_ctx* ctx);
/* Copy line `first`..`first+n-1` of the last push to host memory: raw dB lines (fft_size floats each, DC-centred — what
 * acquireFFTBuffer/releaseFFTBuffer hand to the waterfall), zoomed lines (data_width floats) and palette indices.
 * Any destination may be NULL.  Returns the number of lines copied. */
int sdrpp_fft_read(sdrpp_ctx* ctx, int first, int n, float* raw_host, float* zoomed_host, int32_t* index_host);
/* Same, but into DEVICE memory of the caller (asynchronous D2D copies on the context's stream) — e.g. torch tensors that
 * are then handed to an RCCL gather of waterfall lines. */
int sdrpp_fft_copy_device(sdrpp_ctx* ctx, int first, int n, float* raw_dev, float* zoomed_dev, int32_t* index_dev);
/* Device pointers to the same buffers (valid until the next push). */
int sdrpp_fft_device_buffers(sdrpp_ctx* ctx, const float** raw, const float** zoomed, const int32_t** index, int* n_lines);

/* ---- VFO bank (replaces Splitter fan-out + N x RxVFO + radio demodulators) --------------------------------------------------- */
#define SDRPP_MAX_DECIM_STAGES 4

enum sdrpp_demod_mode {
    SDRPP_DEMOD_RAW = -1, /* no demodulator: output = RxVFO::out (complex IF)                                        */
    SDRPP_DEMOD_WFM = 0,  /* dsp::demod::BroadcastFM mono branch (broadcast_fm.h:146,205-211)                        */
    SDRPP_DEMOD_NFM = 1,  /* dsp::demod::FM<stereo_t> (fm.h:79-96)                                                   */
    SDRPP_DEMOD_AM = 2,   /* dsp::demod::AM<stereo_t> (am.h:101-133)                                                 */
    SDRPP_DEMOD_USB = 3,  /* dsp::demod::SSB<stereo_t> (ssb.h:77-92)                                                 */
    SDRPP_DEMOD_LSB = 4,
    SDRPP_DEMOD_DSB = 5
};

typedef struct sdrpp_vfo_desc {
    /* FrequencyXlator (frequency_xlator.h:14-30): phaseDelta for -offset at the input rate */
    float phase_delta_re, phase_delta_im;
    /* PowerDecimator (power_decimator.h:93-111): stages of the chosen plan, in order; n_stages = 0 for ratio 1 */
    int n_stages;
    int stage_decim[SDRPP_MAX_DECIM_STAGES];
    int stage_ntaps[SDRPP_MAX_DECIM_STAGES];
    const float* stage_taps[SDRPP_MAX_DECIM_STAGES];
    /* PolyphaseResampler (polyphase_resampler.h:69-99): interp == decim -> stage absent; taps already scaled by interp */
    int interp, decim;
    int resamp_ntaps;
    const float* resamp_taps;
    /* RxVFO channel filter (rx_vfo.h:95-98, 117-121); chan_ntaps = 0 when bandwidth == out rate (filterNeeded false) */
    int chan_ntaps;
    const float* chan_taps;
    /* demodulator */
    int demod;                /* enum sdrpp_demod_mode */
    float inv_deviation;      /* Quadrature::_invDeviation = 1/hzToRads(deviation, if_rate) (quadrature.h:19-26)         */
    int audio_ntaps;          /* WFM alFir / NFM fir / AM lpf real taps; 0 = filter bypassed (fm.h:88, broadcast_fm.h:205) */
    const float* audio_taps;
    /* loop::AGC (agc.h:15-27), used by AM (audio or carrier AGC) and SSB */
    float agc_set_point, agc_attack, agc_decay, agc_max_gain, agc_max_output_amp, agc_init_gain;
    int am_carrier_agc;       /* AM: 1 = AGCMode::CARRIER, 0 = AGCMode::AUDIO (am.h:14-17)                               */
    float dc_block_rate;      /* AM: DCBlocker rate (am.h:32 -> dc_blocker.h:54-60)                                       */
    float ssb_phase_delta_re, ssb_phase_delta_im; /* SSB second xlator (ssb.h:24,106-117)                                */
    /* NCO of THIS VFO: 0 = the context's mode (sdrpp_set_nco_mode), 1 = closed form, 2 = the reference's float rotator recursion.
     * FM and AM do not see the difference (no output depends on the absolute phase); a product detector (SSB) and the raw IF follow
     * the reference's own rounding drift only with the recursion, which costs a sequential pass over the full-rate stream — so a bank
     * can keep its FM / AM channels on the fast path and pay for exactness where it shows. */
    int nco_mode;
} sdrpp_vfo_desc;

/* IQFrontEnd::addVFO / removeVFO (iq_frontend.cpp:140-183).  Arrays in `desc` are copied.  *id receives a handle. */
int sdrpp_vfo_add(sdrpp_ctx* ctx, const sdrpp_vfo_desc* desc, int* id);
int sdrpp_vfo_remove(sdrpp_ctx* ctx, int id);
int sdrpp_vfo_count(sdrpp_ctx* ctx);
/* RxVFO::setInSamplerate / setOutSamplerate (rx_vfo.h:35-58) — and with them IQFrontEnd::setSampleRate / setDecimation (iq_frontend.cpp:76-123) and
 * the radio module's demodulator switch (vfo_manager.cpp:52, radio_module.h:419-563): the VFO is described anew (`desc`: new plan, new taps), the
 * old handle dies, *new_id takes its place — and what the reference's objects carry across such a change is carried here:
 *   keep & 1: the RxVFO's own state — the translation's phase (FrequencyXlator::setOffset swaps phaseDelta only) and the channel filter's delay line
 *             (FIR::setTaps moves it under the new tap count, fir.h:31-52; a filter bypassed under the new settings keeps it for when it wakes up).
 *             The decimator stages and the polyphase resampler start from cleared state, as the reference re-creates / resets them
 *             (power_decimator.h:91-108, polyphase_resampler.h:38-67).
 *   keep & 2: the demodulator behind it lives on (setInSamplerate: the radio's demodulator block is not touched): discriminator and audio low-pass
 *             history, AGC / DC-blocker state, SSB's second translation.  Only where the new description has the same demodulator; a demodulator
 *             SWITCH creates a new demodulator object in the reference too: leave the bit off.
 * An AF chain (sdrpp_vfo_set_af) is not carried over: attach it to the new handle (it starts cleared, as afChain's blocks do after a restart). */
int sdrpp_vfo_replace(sdrpp_ctx* ctx, int old_id, const sdrpp_vfo_desc* desc, int keep, int* new_id);
/* RxVFO::setOffset (rx_vfo.h:72-77): only phaseDelta changes, phase stays continuous; the samples already inside the first
 * decimator's delay line keep their old rotation (the first outputs after the change are handed over sample-exactly). */
int sdrpp_vfo_set_phase_delta(sdrpp_ctx* ctx, int id, float re, float im);
/* SSB::setBandwidth / setMode (ssb.h:44-62, 106-117): the second translation's increment follows bandwidth / 2; phase continuous. */
int sdrpp_vfo_set_ssb_phase_delta(sdrpp_ctx* ctx, int id, float re, float im);
/* RxVFO::setBandwidth (rx_vfo.h:60-70): swap the channel-filter taps, history kept (fir.h:31-52). n = 0 bypasses. */
int sdrpp_vfo_set_channel_taps(sdrpp_ctx* ctx, int id, const float* taps, int n);
/* RxVFO::reset + demod reset: clears histories, NCO phase and loop states. */
int sdrpp_vfo_reset(sdrpp_ctx* ctx, int id);
/* Output of the most recent push: number of samples (stereo frames, or complex IF samples in RAW mode). */
int sdrpp_vfo_out_count(sdrpp_ctx* ctx, int id);
/* Copy up to max samples (2 floats each) to host memory; returns the count copied (what RxVFO::run/demod swap()). */
int sdrpp_vfo_read(sdrpp_ctx* ctx, int id, float* dst_host, int max);
/* The same for many VFOs with ONE device-to-host copy (a host block that delivers every VFO stream after each block would otherwise
 * pay one small copy + synchronisation per VFO).  which[i]: 0 = what sdrpp_vfo_read returns, 1 = the complex IF, 2 = the AF chain
 * output (NULL: all 0).  The blocks are packed back to back into dst_host (2 floats per sample); offsets[i] / counts[i] (in samples)
 * locate VFO ids[i]'s block.  Returns the total number of samples, SDRPP_ERR_INVALID if max_samples is too small.  dst_host = NULL is
 * a size query: offsets / counts are filled and the total returned, nothing is copied. */
int sdrpp_vfo_read_many(sdrpp_ctx* ctx, int n, const int* ids, const int* which, float* dst_host, int64_t max_samples, int64_t* offsets, int* counts);
/* Device pointers: demodulated audio (or IF in RAW mode) and the complex IF stream (RxVFO::out) of the last push. */
int sdrpp_vfo_device_buffers(sdrpp_ctx* ctx, int id, const float** out, int* n_out, const float** if_out, int* n_if);

/* ---- radio AF chain behind a demodulating VFO (decoder_modules/radio/src/radio_module.h:98-110, 540-547) ------
 * RationalResampler<stereo_t> (demodulator AF rate -> audio rate: power-of-two pre-decimation plan + L/M polyphase,
 * rational_resampler.h:80-165) -> FIR<stereo_t,float> high-pass (radio_module.h:103,597; optional) -> Deemphasis<stereo_t>
 * (filter/deephasis.h:58-77; optional).  Attached to an existing VFO; its output replaces nothing: the demodulator output
 * (sdrpp_vfo_read) stays available, the AF output has its own accessors. */
typedef struct sdrpp_af_desc {
    int n_stages;                                     /* PowerDecimator stages of the pre-decimation plan (0: ratio 1)        */
    int stage_decim[SDRPP_MAX_DECIM_STAGES];
    int stage_ntaps[SDRPP_MAX_DECIM_STAGES];
    const float* stage_taps[SDRPP_MAX_DECIM_STAGES];
    int interp, decim;                                /* polyphase L/M; interp == decim -> stage absent                         */
    int resamp_ntaps;
    const float* resamp_taps;                         /* already scaled by interp (sdrpp_design_resampler)                      */
    int hpf_ntaps;                                    /* 0 = high-pass block disabled                                           */
    const float* hpf_taps;
    float deemph_alpha;                               /* 0 = de-emphasis block disabled (sdrpp_design_deemphasis_alpha)         */
} sdrpp_af_desc;
/* afChain.enableBlock / setAudioSampleRate (radio_module.h:540-547, 585-600).  af == NULL detaches.  Arrays are copied; the chain
 * starts from cleared state.  Only for VFOs with a demodulator (demod != RAW). */
int sdrpp_vfo_set_af(sdrpp_ctx* ctx, int id, const sdrpp_af_desc* af);
/* AF output of the most recent push: stereo frames at the audio rate (what afChain.out swap()s to the sink stream). */
int sdrpp_vfo_af_count(sdrpp_ctx* ctx, int id);
int sdrpp_vfo_af_read(sdrpp_ctx* ctx, int id, float* dst_host, int max);
int sdrpp_vfo_af_device_buffer(sdrpp_ctx* ctx, int id, const float** out, int* n_out);
int sdrpp_abi_sizeof_af_desc(void);

/* ---- sink-side sample packing: float -> int16 / int8 on the device, before the copy to the host -----------------
 * `which`: 0 = what sdrpp_vfo_read returns (demodulator output, or the IF in RAW mode), 1 = the complex IF (RxVFO::out), 2 = the AF
 * chain output.  pcm_type follows dsp/compression/pcm_type.h: 0 = I8, 1 = I16, 2 = F32.
 * sdrpp_vfo_read_pcm: VOLK's volk_32f_s32f_convert_16i / _8i with the caller's scale (the recorder writes int16 WAV with 32767,
 *   utils/wav.cpp:166); returns frames.  sdrpp_vfo_read_compressed: one SDR++-server frame exactly as
 *   SampleStreamCompressor::process builds it (sample_stream_compressor.h:30-62): [u16 0][u16 pcm_type][f32 scaler][data], scaler =
 *   the largest VALUE of the block, data scaled by 128 / scaler or 32768 / scaler; returns bytes (0 for an empty block). */
int sdrpp_vfo_read_pcm(sdrpp_ctx* ctx, int id, int which, int pcm_type, float scale, void* dst_host, int max_frames);
/* The same conversion for the pre-processed wideband IQ of the most recent push (sdrpp_preproc_read's samples): what the recorder's baseband
 * mode writes from its bindIQStream consumer (recorder/src/main.cpp:209, 528-530 -> wav::Writer::write, utils/wav.cpp:158-167); returns
 * complex samples (2 values each). */
int sdrpp_preproc_read_pcm(sdrpp_ctx* ctx, int pcm_type, float scale, void* dst_host, int max_samples);
int sdrpp_vfo_read_compressed(sdrpp_ctx* ctx, int id, int which, int pcm_type, unsigned char* dst_host, int max_bytes);

/* ---- WaterFall display state around the raw-line history (core/src/gui/widgets/waterfall.cpp) ---------------
 * The last `height` raw dB lines stay resident in HBM in the reference's ring order (getFFTBuffer :875-886), so a zoom / pan /
 * level change re-renders the whole waterfall on the device (updateWaterfallFb :600-631) instead of re-reading host memory, and
 * the FFT trace's smoothing and peak hold (pushFFT :913-939) run per new line next to the zoom that already does.
 * Needs sdrpp_fft_configure first (and sdrpp_fft_set_view for the trace); re-configuring the FFT size invalidates it. */
int sdrpp_wf_configure(sdrpp_ctx* ctx, int height);                              /* waterfallHeight lines kept; 0 removes          */
int sdrpp_wf_set_smoothing(sdrpp_ctx* ctx, int enabled, float speed);            /* setFFTSmoothing + setFFTSmoothingSpeed :1166-1194 */
int sdrpp_wf_set_hold(sdrpp_ctx* ctx, int enabled, float speed);                 /* setFFTHold + setFFTHoldSpeed :1153-1164          */
/* latestFFT (after smoothing) and latestFFTHold of the current view: data_width floats each (NULL to skip); returns data_width */
int sdrpp_wf_latest(sdrpp_ctx* ctx, float* latest, float* hold);
/* calculateVFOSignalInfo (:558-598) on the newest stored line: strength = max dB inside the VFO, snr = strength - mean of the two
 * half-bandwidth side bands.  Returns 1, or 0 while no line is stored. */
int sdrpp_wf_signal_info(sdrpp_ctx* ctx, double center_offset, double bandwidth, double whole_bandwidth, float* strength, float* snr);
/* updateWaterfallFb: palette indices [height][data_width], newest line first, rows beyond the stored lines = -1 (opaque black). */
int sdrpp_wf_raster(sdrpp_ctx* ctx, int draw_data_start, int draw_data_size, int data_width, float wf_min, float wf_max, int32_t* dst_host, int* n_lines);

/* ---- IQFrontEnd pre-processing chain (core/src/signal_path/iq_frontend.cpp:32-39, setDecimation /
 *      setDCBlocking / setInvertIQ :105-130): PowerDecimator<complex_t> (stages of the plan for the ratio, power_decimator.h:93-111)
 *      -> DCBlocker<complex_t> (dc_rate = genDCBlockRate(effectiveSr) = 50 / effectiveSr, iq_frontend.h:55-57; 0 = block disabled)
 *      -> Conjugate.  Runs on the device in front of the FFT branch and the VFO bank: every later stage sees the pre-processed
 *      stream at the effective sample rate (FFT framing, VFO descriptors are the caller's, designed at that rate).  max_push of
 *      sdrpp_create counts RAW samples.  n_stages = 0, dc_rate = 0, conjugate = 0 removes the chain.  State starts cleared. */
/* Numerics of the chain.  Default: the decimator stages run on the matrix cores (k-ordered fused multiply-adds) and the DC blocker as a
 * parallel scan — the pre-processed stream then agrees with the reference to ~1e-7 (decimation only) resp. ~5e-5 (DC blocker on: the
 * reference's sequential float32 integrator carries a rounding drift of its own that a parallel sum does not reproduce), and waterfall
 * lines computed from it are NOT bit-exact any more (up to 5e-3 dB with decimation, 0.05 dB with DC blocking).  With
 * sdrpp_preproc_set_reference_order(ctx, 1) the chain evaluates the reference's own arithmetic — VOLK's generic tap-ordered
 * multiply-then-add dot product (decimating_fir.h:51-61) and the sequential DC-blocker recursion (dc_blocker.h:54-60), one wavefront,
 * ~40 cycles per sample — and the pre-processed stream, and every waterfall line behind it, is bit-identical to the compiled reference
 * (tests/test_parity_fft.py::test_preproc_chain_reference_order).  A parity mode: a few times real time at 10 MS/s, not thousands.
 * The switch survives sdrpp_preproc_configure; streaming state (history, offsets, the DC estimate) is shared by both modes. */
int sdrpp_preproc_set_reference_order(sdrpp_ctx* ctx, int on);
int sdrpp_preproc_configure(sdrpp_ctx* ctx, int n_stages, const int* stage_decim, const int* stage_ntaps, const float* const* stage_taps,
                            float dc_rate, int conjugate);
/* The same for a chain that is RE-planned while the stream runs — what IQFrontEnd's setters do to their blocks (iq_frontend.cpp:76-130):
 *   keep & 1: the decimator's delay lines and offsets stay if the new description has the same stages (setSampleRate / setDCBlocking / setInvertIQ
 *             do not touch the decimator; setDecimation creates new stages — PowerDecimator::setRatio, power_decimator.h:91-108 — so it passes 0 here);
 *   keep & 2: the DC blocker continues from its estimate (DCBlocker::setRate swaps the rate only; a blocker switched off and on again is the same
 *             object, dc_blocker.h:54-60).
 * sdrpp_preproc_configure = keep 0: everything cleared. */
int sdrpp_preproc_reconfigure(sdrpp_ctx* ctx, int n_stages, const int* stage_decim, const int* stage_ntaps, const float* const* stage_taps,
                              float dc_rate, int conjugate, int keep);
/* The pre-processed samples of the most recent push — what Splitter hands to streams bound with bindIQStream (iq_frontend.cpp:132-138). */
int sdrpp_preproc_out_count(sdrpp_ctx* ctx);
int sdrpp_preproc_read(sdrpp_ctx* ctx, float* dst_host, int max);
int sdrpp_preproc_device_buffer(sdrpp_ctx* ctx, const float** iq, int* n);

/* ---- data path ------------------------------------------------------------------------------------------------------------ */
/* One block of IQ, as Splitter::run hands to every bound stream (splitter.h:46-61).  Host pointer: copied H2D first.
 * Device pointer: read in place (must stay valid until the next sdrpp_sync / stream synchronisation).  Runs the FFT
 * branch and every VFO; asynchronous on the context's stream — outputs are readable after sdrpp_sync (the read calls
 * synchronise themselves). */
int sdrpp_push(sdrpp_ctx* ctx, const float* iq_host, int64_t count);
int sdrpp_push_device(sdrpp_ctx* ctx, const float* iq_dev, int64_t count);
/* file_source path (source_modules/file_source/src/main.cpp:154-167): interleaved int16 IQ converted on the device
 * (x / 32768), halving the PCIe bytes.  Host pointer. */
int sdrpp_push_int16(sdrpp_ctx* ctx, const int16_t* iq_host, int64_t count);

/* Page-locked host memory for buffers that are pushed from (the copy out of pageable memory is staged by the runtime and about three
 * times slower); NULL on failure.  The host blocks allocate their frame-buffer slots with it. */
void* sdrpp_host_alloc(size_t bytes);
void sdrpp_host_free(void* p);
/* Device memory on the context's GPU for buffers the host hands to sdrpp_fft_copy_device (e.g. the line a several-GPU host keeps per stream for
 * its RCCL gather, host/sdrpp_gpu_rccl.h) without linking the HIP runtime itself; NULL on failure. */
void* sdrpp_device_alloc(sdrpp_ctx* ctx, size_t bytes);
void sdrpp_device_free(sdrpp_ctx* ctx, void* p);
/* A synchronous copy on the context's GPU for the same kind of host: kind 0 host -> device, 1 device -> device, 2 device -> host; returns when the
 * bytes have arrived.  THE ONE CALL OF A CONTEXT THAT IS THREAD-SAFE: it may come from another thread while the context's own thread is inside any
 * other call (a several-GPU host's gather thread copies a front end's newest line while its worker pushes blocks).  It runs on a stream of its own
 * and is therefore NOT ordered with the work queued on the context's stream: use it for buffers whose content is complete (filled by an earlier
 * sdrpp_device_copy, or by sdrpp_fft_copy_device followed by sdrpp_sync), not to read what a push has just been asked to compute.  It never waits
 * for queued launches and sets no error text (sdrpp_last_error belongs to the context's thread): the return code is all there is. */
int sdrpp_device_copy(sdrpp_ctx* ctx, void* dst, const void* src, size_t bytes, int kind);
/* Deferred processing: sdrpp_push* only stage the samples (the H2D copy runs, the caller's buffer is free on return) and the next call
 * that observes results — sdrpp_sync, any *_lines / *_read* / *_count / *_device_buffer(s) / sdrpp_wf_* call — or changes the
 * configuration processes everything staged since the previous one as ONE pass over the device.  The results then cover ALL those
 * pushes, concatenated (lines, VFO blocks); every staged push still counts as a reference block of its own for the block-dependent
 * operations (sdrpp_set_reference_block), so the output is what pushing and reading block by block gives — only the launch sequence is
 * paid once per pass instead of once per block.  This is how a host that drains a queue of blocks (IQFrontEnd with buffering on) keeps
 * up at the reference's block size (sample_rate / 200), where a pass is launch-bound.  At most max_push samples can be staged:
 * a push beyond that returns SDRPP_ERR_INVALID (observe first).  Off (default): every push is processed at once. */
int sdrpp_set_deferred(sdrpp_ctx* ctx, int on);
/* Deferred mode, page-locked source (sdrpp_host_alloc): stage WITHOUT waiting for the copy — the device fetches the samples itself, the
 * call returns at once.  The buffer must stay untouched until sdrpp_push_wait (all such copies have landed) or the next call that
 * returns results of the pass (any *_read*, sdrpp_sync) has returned.  How a frame-buffer worker stages a whole backlog of blocks for the
 * price of a few kernel launches and ONE wait (host/sdrpp_gpu_blocks.h: SampleFrameBuffer::worker, frame_buffer.h:76-98).  Outside
 * deferred mode, or with memory that is not page-locked, it is sdrpp_push. */
int sdrpp_push_pinned_async(sdrpp_ctx* ctx, const float* iq_pinned, int64_t count);
int sdrpp_push_wait(sdrpp_ctx* ctx);
int64_t sdrpp_pending(sdrpp_ctx* ctx);   /* samples staged and not yet processed */

/* ---- pipelined execution: one launch per block, results a few blocks late ----------------------------------------------------------
 * At the reference's block size (sample_rate / 200 samples per swap(): core/src/dsp/stream.h:9, source_modules/file_source/src/main.cpp:157)
 * a block is far less than one wave of work for the GPU and an ordinary pass costs the SUM of its ~8 dependent launches.  The reference
 * itself is a pipeline at that grain — one thread per dsp::block, every stream<T> a hand-over between two of them (block.h:46-73,
 * stream.h:43-92): while the demodulator works on block n the VFO already has block n + 1.  Pipelined mode does the same on the device:
 * every sdrpp_push* launches ONE kernel ("tick") that runs the arrival of block n (its copy into device memory, the upload of its job
 * tables) next to the front end / FFT pass 1 of block n - 1, the first decimator / FFT pass 2 of block n - 2, ... — every stage of every
 * block exactly as in an ordinary pass (same kernels' bodies, same arithmetic: results are bit-identical), only not one after the
 * other.  A block's results are complete `depth` launches later (depth <= 18: where the block's last role stands; 6 for a WFM bank + 65536-point FFT whose outputs stay on the device, 7 with its VFO blocks delivered, 12 with the AF chain): with the next blocks, or at once when the
 * caller asks (sdrpp_pipeline_flush, sdrpp_result_wait, any observing call such as sdrpp_vfo_read / sdrpp_sync — these run the queued
 * stages without new input; sdrpp_fft_lines and sdrpp_vfo_out_count only report what the host already knows and do not).
 *   sdrpp_push_device        reads the caller's buffer IN PLACE one launch later at the earliest: it must stay valid until sdrpp_sync.
 *   sdrpp_push / _push_int16 copy into a page-locked staging slot (the caller's buffer is free on return), fetched by the next launch.
 *   sdrpp_push_pinned_async  page-locked memory is fetched by the launch itself; sdrpp_push_wait returns when all such fetches have run.
 * The pre-processing chain (sdrpp_preproc_configure, default arithmetic), the radio's AF chain (sdrpp_vfo_set_af) and the waterfall display
 * state (sdrpp_wf_configure) run that way too — their stages are further levels of the block; so do VFOs on the reference-rotator NCO
 * (nco_mode = 2: the recursion over the whole block is one of the launch's work items — such a stream is bound by that chain, ~26 cycles per
 * sample, but every VFO's results stay pipelined) and banks of any size, a single VFO included (the vector-unit front ends, one VFO per work
 * item of the launch).  What cannot (the
 * reference-order arithmetic of the pre-processing chain, a resampler in its register-blocked vector form, a retune hand-over in progress, more
 * FFT frames than one scratch chunk, a block the pre-processing decimator swallows whole) is processed as an ordinary
 * pass behind everything queued: always correct, pipelined where possible — and its results are delivered into the block's result slot
 * like any other block's (by plain copies and a wait inside the push: the slow path).
 * result_flags (sdrpp_set_pipelined): which results every block also delivers into page-locked host memory, ready for sdrpp_result_wait
 * without any copy call: 1 = every VFO's output block (the end of its chain: the AF chain's output where one is attached, else what
 * sdrpp_vfo_read returns), 2 = zoomed lines + palette indices, 4 = raw dB lines, 8 = the pre-processed IQ stream of the block (only with a
 * pre-processing chain configured: without one it is the input block itself).
 * At most SDRPP_RESULT_SLOTS (24) launches' results exist at a time (one block per launch unless sdrpp_set_pipeline_group says otherwise): release
 * them (a block whose slot is still held 24 launches later fails the push).
 * HOW FAR BEHIND TO ASK: a streaming host takes block t - lag when it has pushed block t.  With lag >= depth + 1 (sdrpp_pipeline_stats
 * out[4]: 6-7 levels for a radio bank + FFT, 10 for cfg 4's NFM / AM / SSB chains, 12 with AF chains, + the levels of a pre-processing chain)
 * sdrpp_result_wait finds the block complete.  With a smaller lag it must run the queued stages and WAIT for the device at every call: host
 * and device then take turns instead of overlapping (measured: cfg 4, lag 8 against 10 levels, 213 instead of 165 us per block).
 * sdrpp_gpu::IQFrontEnd and bench.py follow the reported depth (up to 22, leaving two of the 24 slots to the blocks in the making). */
/* Host blocks in pipelined mode without the library's own copy: sdrpp_push_stage hands out the page-locked staging slot the next block is
 * fetched from (room for max_push samples); the host fills it — with several threads if it likes: a 400 KB memcpy is the largest single
 * item of a 50 000-sample block's host time — and its own buffer is free as soon as it has; sdrpp_push_staged then launches the block
 * exactly like sdrpp_push.  One slot is open at a time. */
int sdrpp_push_stage(sdrpp_ctx* ctx, int64_t count, float** slot);
int sdrpp_push_staged(sdrpp_ctx* ctx, int64_t count);
/* The same with the host's copy threads still at work: the call plans the block at once (the job tables do not depend on the samples:
 * ~8 us of a 50 000-sample block's ~50 us of host time) and waits, on the host, for *pending to reach 0 before the first launch that
 * reads the slot.  `pending` = the number of unfinished parts of the caller's copy, decremented (release order) by the copying threads;
 * a word that does not reach 0 within 5 s fails the push.  sdrpp_gpu::IQFrontEnd's pipelined worker stages its blocks this way. */
int sdrpp_push_staged_when(sdrpp_ctx* ctx, int64_t count, const volatile uint32_t* pending);
#define SDRPP_RESULT_SLOTS 24
#define SDRPP_GROUP_MAX 32
typedef struct sdrpp_result {
    uint64_t ticket;          /* the block: 1 for the first push in pipelined mode, counted by sdrpp_ticket                          */
    int n_vfo;                /* VFO blocks delivered (0 without result flag 1), in sdrpp_vfo_add order                              */
    const int* ids;           /* [n_vfo] VFO handles                                                                                 */
    const int64_t* offsets;   /* [n_vfo] first sample (2 floats each) of the VFO's block in `samples`                                */
    const int* counts;        /* [n_vfo] samples                                                                                     */
    const float* samples;     /* page-locked host memory of the library, valid until sdrpp_result_release                            */
    int n_lines, fft_size, data_width;
    const float* zoomed;      /* [n_lines][data_width] (flag 2), else NULL                                                           */
    const int32_t* index;     /* [n_lines][data_width] (flag 2)                                                                      */
    const float* raw;         /* [n_lines][fft_size] (flag 4)                                                                        */
    int n_iq;                 /* pre-processed IQ samples of the block (flag 8 with a pre-processing chain configured), else 0       */
    const float* iq;          /* [n_iq] complex: what streams bound with bindIQStream receive (iq_frontend.cpp:32-39 -> Splitter)    */
} sdrpp_result;
int sdrpp_set_pipelined(sdrpp_ctx* ctx, int on, int result_flags);
/* SEVERAL BLOCKS PER LAUNCH.  A launch costs the device a start ramp, a tail and the gap to the next one whatever the block holds, and the host one
 * plan: at the reference's block size (sample_rate / 200) that is most of a block's time, and a host that pushes faster than the device works
 * only makes the launch queue longer.  With max_blocks > 1 a push is HELD — nothing planned, nothing launched — until max_blocks pushes have come
 * together (or the next push cannot join: another kind of push, device / page-locked memory that does not continue where the last block ended,
 * more than max_push samples in all; or any call that observes results or changes the configuration, sdrpp_pipeline_flush, sdrpp_result_wait
 * for one of them).  The group then goes out as ONE launch: its blocks are planned as one block of the stream whose reference-block ends
 * (sdrpp_set_reference_block: AGC look-ahead, rotator calls) are the ends of the pushes — exactly what a deferred pass does with its staged
 * pushes — so every sample of every output is the one block-by-block processing gives (bit-identical: tests/test_pipelined.py::
 * test_grouped_launches_equal_block_by_block), and EVERY PUSH KEEPS ITS OWN TICKET AND RESULTS: sdrpp_result_wait(ticket) hands out that push's
 * share of every VFO's output block and the lines its samples completed (views into the group's result slot; SDRPP_RESULT_SLOTS now counts launch
 * groups, of up to SDRPP_GROUP_MAX blocks each).  Host pushes of a group land back to back in ONE page-locked staging slot and are fetched by
 * one landing copy; device / page-locked blocks must be contiguous to share a launch (a ring of blocks in one allocation is; the wrap-around
 * starts a new group).  The sum of a group's blocks is limited by max_push (sdrpp_create): size it for max_blocks blocks.
 * adaptive = 1: a push also goes out at once — with whatever is held — when the device has fewer than two launches in flight, i.e. the group size
 * follows what is queued: 1 while the host is the slower side (no added latency), max_blocks when the device is.  adaptive = 0: always wait for
 * max_blocks (deterministic: tests, benchmarks).  What it costs: a held block waits for its group, and a block's results are complete `depth`
 * LAUNCHES after its group went out — a streaming host asks (depth + 1) * max_blocks blocks behind (sdrpp_pipeline_stats out[4] is still the
 * depth in launches).  max_blocks = 1 (default): one launch per push, as before.  Groups do not form behind a pre-processing chain
 * (sdrpp_preproc_configure) or while a block cannot run as a launch of the pipeline at all: such pushes go out one by one.
 * `adaptive` is a set of flags: 1 as above; 2: the words handed to sdrpp_push_staged_when stay valid until their block has been LAUNCHED (not just for the
 * call) — a push that is merely held then returns at once instead of waiting for its copy threads, and the copy runs on under whatever the host does next
 * (the launch of the group waits for every word).  sdrpp_gpu::IQFrontEnd uses 2 with a ring of words and decides on the HOST side when a group goes out: full,
 * or no new block for a few microseconds (sdrpp_pipeline_launch_held) — at the stream seam the device is never the slower side, so rule 1 would never group. */
int sdrpp_set_pipeline_group(sdrpp_ctx* ctx, int max_blocks, int adaptive);
/* Launches what is held (nothing held: no-op).  Unlike sdrpp_pipeline_flush it does not run the queued stages of earlier blocks to completion. */
int sdrpp_pipeline_launch_held(sdrpp_ctx* ctx);
/* out[0] launch groups so far (single blocks included), [1] groups of more than one block, [2] the blocks in those, [3] the largest group,
 * [4] pushes held right now.  Returns the number of entries written. */
int sdrpp_pipeline_group_stats(sdrpp_ctx* ctx, int64_t* out, int max);
uint64_t sdrpp_ticket(sdrpp_ctx* ctx);                       /* ticket of the most recent push (pushes so far in pipelined mode)    */
int sdrpp_pipeline_flush(sdrpp_ctx* ctx);                    /* launch what is queued, no new input; does not wait                  */
int sdrpp_result_ready(sdrpp_ctx* ctx, uint64_t ticket);     /* 1 / 0 without blocking or flushing; SDRPP_ERR_NOT_FOUND: no slot    */
int sdrpp_result_wait(sdrpp_ctx* ctx, uint64_t ticket, sdrpp_result* out);   /* flushes if needed, waits, hands the slot out        */
int sdrpp_result_release(sdrpp_ctx* ctx, uint64_t ticket);
/* wait + copy + release in one call for a host that only wants a block's zoomed lines / palette indices (result flag 2): up to max_lines lines
 * of data_width values each go to zoomed_dst / index_dst (either may be NULL); *n_lines = lines the block completed.  SDRPP_ERR_INVALID if it
 * completed more than max_lines (the slot is released all the same). */
int sdrpp_result_take_lines(sdrpp_ctx* ctx, uint64_t ticket, float* zoomed_dst, int32_t* index_dst, int max_lines, int* n_lines);
/* How the blocks of a pipelined run were executed — for tests and bench.py, which assert the mode they mean to measure.
 * out[0] launches ("ticks") so far, [1] blocks that ran as ticks, [2] blocks that fell back to an ordinary pass, [3] ticks with more role
 * workgroups than the device holds at once (3 per CU: "crowded" order of the roles, tick_host.h), [4] levels of the most recent block (its
 * results are complete that many launches after its push), [5] number of roles R, [6] ticks launched in the four-wavefronts-per-SIMD build of
 * the tick kernel, [7] bytes of job tables the most recent block uploaded; then out[8 + r], r < R: workgroups
 * launched so far in role r (sdrpp_pipeline_role_name(r); e.g. "fcm16_132_4" = the front end in its small-block shape).
 * Returns the number of entries written (<= max).  Counters start at sdrpp_create. */
#define SDRPP_PIPELINE_STATS_HEAD 8
int sdrpp_pipeline_stats(sdrpp_ctx* ctx, int64_t* out, int max);
const char* sdrpp_pipeline_role_name(int role);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------------------------------------- */
/* Cumulative per-kernel-family device time measured with HIP events on the context's stream while timing is enabled.
 * family: 0 fft_pass1, 1 fft_pass2, 2 fft_single, 3 zoom, 4 vfo_stage1, 5 vfo_decim, 6 vfo_poly, 7 vfo_fir, 8 demod, 9 carry/misc,
 * 10 af_chain, 11 vfo_pipe (FM back ends as one launch), 12 tick (pipelined mode: one launch per block) */
#define SDRPP_NUM_KERNEL_FAMILIES 13
/* on = 0: off; 1: every family; 1 | (family_bitmask << 1): only the selected families (each timed launch costs two event
 * records on its stream, so a throughput run instruments just the kernel it reports). */
int sdrpp_timing_enable(sdrpp_ctx* ctx, int on);
int sdrpp_timing_read(sdrpp_ctx* ctx, double* ms_per_family, int64_t* launches_per_family);
const char* sdrpp_kernel_family_name(int family);

#ifdef __cplusplus
}
#endif
#endif /* SDRPP_GPU_H */
