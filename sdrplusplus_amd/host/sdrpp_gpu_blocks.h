// Host-side C++ mirror of the reference's plugin/operator surface for the hot path, on top of the C-ABI
// (include/sdrpp_gpu.h).  Header-only, like the reference's DSP library, and written against SDR++'s own block / stream headers:
// compile with SDR++'s core/src on the include path — the blocks derive from the real dsp::block and speak the real dsp::stream<T>
// (core/src/dsp/block.h, stream.h, types.h).  (tests/host_cpp/standalone holds a small test double of those three headers so the
// mirror can also be built and run on a machine without the SDR++ tree.)
//
// Blocks:
//   sdrpp_gpu::IQFrontEnd  — the public API of the reference's IQFrontEnd (core/src/signal_path/iq_frontend.h:12-49), call for call:
//       init(in, sampleRate, buffering, decimRatio, dcBlocking, fftSize, fftRate, fftWindow, acquireFFTBuffer, releaseFFTBuffer,
//       fftCtx), setInput, setSampleRate / getSampleRate, setBuffering, setDecimation, setInvertIQ, setDCBlocking,
//       bindIQStream / unbindIQStream, addVFO / removeVFO, setFFTSize / Rate / Window, flushInputBuffer, start / stop,
//       getEffectiveSamplerate.  One worker replaces inBuf + preproc + Splitter + Reshaper + Handler + every RxVFO thread: it takes
//       a block from the input stream (directly, or through a 32-slot frame buffer when buffering is on — frame_buffer.h:51-98),
//       hands it to sdrpp_push (one H2D copy, all kernels), delivers the finished dB lines through the acquire / release callback
//       pair (iq_frontend.cpp:258-266), the pre-processed IQ to every stream bound with bindIQStream (Splitter::run,
//       routing/splitter.h:46-61) and one block on every VFO's output stream (all VFO outputs arrive in ONE D2H copy).
//   sdrpp_gpu::RxVFO — what addVFO returns: public `out` stream + setInSamplerate / setOutSamplerate / setBandwidth / setOffset /
//       reset with the reference's meaning (core/src/dsp/channel/rx_vfo.h:38-87).  With a demodulator attached (attachDemod) the
//       demodulated audio is delivered on `audio` (dsp::stream<stereo_t>), which is what radio's Demodulator::getOutput()
//       returns (decoder_modules/radio/src/demod.h:60); sdrpp_gpu_radio.h wraps that as a demod::Demodulator.
#pragma once
#include <atomic>
#include <climits>
#include <cassert>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdrpp_gpu.h"

#include <dsp/block.h>
#include <dsp/stream.h>
#include <dsp/types.h>
#if defined(__has_include)
#if __has_include(<dsp/multirate/decim/plans.h>)
#include <dsp/multirate/decim/plans.h>
#define SDRPP_GPU_HAVE_SDRPP_PLANS 1
#endif
#endif

#ifdef SDRPP_GPU_BLOCKS_PROF
#define SDRPP_PIPE_TICK(slot) { const auto _now = std::chrono::steady_clock::now(); pipeUs[slot] += std::chrono::duration<double, std::micro>(_now - pipeT).count(); pipeT = _now; }
#else
#define SDRPP_PIPE_TICK(slot)
#endif
namespace sdrpp_gpu {

// The reference's power-of-two decimation plans (dsp/multirate/decim/plans.h).  Inside an SDR++ tree they come straight
// from its headers; stand-alone they are read from the numeric fixture sdrplusplus_amd/data/decim_plans.bin.
struct DecimStage { int decimation; std::vector<float> taps; };
class DecimPlans {
public:
    std::map<int, std::vector<DecimStage>> plans;
    int maxRatio = 1;
#ifdef SDRPP_GPU_HAVE_SDRPP_PLANS
    DecimPlans() {
        using namespace dsp::multirate::decim;
        for (unsigned i = 0; i < plans_len; i++) {
            std::vector<DecimStage> st;
            for (unsigned s = 0; s < dsp::multirate::decim::plans[i].stageCount; s++) {
                const auto& g = dsp::multirate::decim::plans[i].stages[s];
                st.push_back({ (int)g.decimation, std::vector<float>(g.taps, g.taps + g.tapcount) });
            }
            this->plans[2 << i] = st;
        }
        maxRatio = 1 << plans_len;
    }
#endif
    bool load(const std::string& path) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) { return false; }
        char magic[4];
        uint32_t ver = 0, n = 0;
        bool ok = fread(magic, 1, 4, f) == 4 && !memcmp(magic, "SDPL", 4) && fread(&ver, 4, 1, f) == 1 && fread(&n, 4, 1, f) == 1;
        for (uint32_t i = 0; ok && i < n; i++) {
            uint32_t ratio, ns;
            ok = fread(&ratio, 4, 1, f) == 1 && fread(&ns, 4, 1, f) == 1;
            std::vector<DecimStage> st;
            for (uint32_t s = 0; ok && s < ns; s++) {
                uint32_t d, nt;
                ok = fread(&d, 4, 1, f) == 1 && fread(&nt, 4, 1, f) == 1;
                std::vector<float> t(nt);
                ok = ok && fread(t.data(), 4, nt, f) == nt;
                st.push_back({ (int)d, t });
            }
            plans[(int)ratio] = st;
        }
        fclose(f);
        if (ok) { maxRatio = 1 << n; }
        return ok;
    }
};

enum class Demod { RAW = SDRPP_DEMOD_RAW, WFM = SDRPP_DEMOD_WFM, NFM = SDRPP_DEMOD_NFM, AM = SDRPP_DEMOD_AM, USB = SDRPP_DEMOD_USB, LSB = SDRPP_DEMOD_LSB, DSB = SDRPP_DEMOD_DSB };

class IQFrontEnd;

// dsp::channel::RxVFO look-alike (rx_vfo.h): created/destroyed through IQFrontEnd::addVFO/removeVFO only.
class RxVFO {
public:
    dsp::stream<dsp::complex_t> out;   // RxVFO::out (IF); delivered when no demodulator is attached
    dsp::stream<dsp::stereo_t> audio;  // demodulator output (radio's Demodulator::getOutput()) when attached

    void setInSamplerate(double inSamplerate);                // rx_vfo.h:38-43
    void setOffset(double offset);                            // rx_vfo.h:72-77
    void setBandwidth(double bandwidth);                      // rx_vfo.h:60-70
    void setOutSamplerate(double outSamplerate, double bandwidth);  // rx_vfo.h:45-58
    void reset();                                             // rx_vfo.h:79-87
    // Radio-module demodulator fused behind this VFO (decoder_modules/radio/src/demodulators/*.h defaults).
    void attachDemod(Demod mode, bool lowPass = true, double agcAttack = 50.0, double agcDecay = 5.0, bool carrierAgc = false);
    // demod::Demodulator::setBandwidth of the fused demodulator (FM deviation = bw / 2, NFM / AM audio low-pass, SSB's second
    // translation: fm.h:47-58, am.h:62-72, ssb.h:44-62); RxVFO::setBandwidth stays the channel filter only, as in the reference
    void setDemodBandwidth(double bandwidth);
    // The radio module's AF chain behind the demodulator (radio_module.h:98-110, 540-547): RationalResampler<stereo_t> to
    // `audioSamplerate`, optional highPass(300, 100) FIR, optional Deemphasis(tau) (tau <= 0: off).  With the chain attached `audio`
    // carries its output (what afChain.out hands to the sink stream); afRate = the demodulator's getAFSampleRate() (= IF rate for
    // every analog demodulator of the radio module).  detachAF() puts the demodulator output back on `audio`.
    void attachAF(double audioSamplerate = 48000.0, double deempTau = 50e-6, bool highPass = false);
    void detachAF();
    // Parity switch for THIS channel (sdrpp_vfo_desc.nco_mode): true = the reference's float rotator recursion on the device (what an SSB
    // product detector or a raw-IF consumer needs to follow a CPU build's own rounding drift), false = closed-form NCO, the fast path
    void setReferenceRotator(bool enabled);

    double inSamplerate = 0, outSamplerate = 0, bandwidth = 0, offset = 0;
    double demodBandwidth = 0;  // 0: follows `bandwidth`
    Demod demod = Demod::RAW;
    bool lowPass = true, carrierAgc = false;
    double agcAttack = 50.0, agcDecay = 5.0;
    int ncoMode = 0;  // 0: the front end's mode (IQFrontEnd::setReferenceRotator), 1: closed form, 2: reference rotator
    bool afOn = false, afHighPass = false;
    double afAudioRate = 48000.0, afDeempTau = 50e-6;

private:
    friend class IQFrontEnd;
    IQFrontEnd* fe = nullptr;
    int id = -1;
    std::vector<int> prevIds;  // handles this VFO had before its last re-plans (setInSamplerate / setOutSamplerate / a new demodulator): blocks pushed under them are still delivered
    std::string name;
};

class IQFrontEnd : public dsp::block {
public:
    enum FFTWindow { RECTANGULAR, BLACKMAN, NUTTALL };  // iq_frontend.h:18-22

    ~IQFrontEnd() override {
        if (_block_init) {
            stop();
            _block_init = false;  // dsp::block's destructor has nothing left to stop
        }
        for (auto& kv : vfos) { delete kv.second; }
        for (int k = 0; k < 2; k++) { sdrpp_device_free(ctx, devLine[k]); }
        if (ctx) { sdrpp_destroy(ctx); }
        for (int i = 0; i < FRAME_SLOTS; i++) { sdrpp_host_free(frames[i]); }
        sdrpp_host_free(gatherPin);
        sdrpp_host_free(linesPin);
    }

    // iq_frontend.h:23.  decimRatio / dcBlocking configure the pre-processing chain (PowerDecimator -> DCBlocker -> Conjugate,
    // iq_frontend.cpp:32-39), which runs on the device in front of the FFT branch and the VFO bank; `buffering` puts the 32-slot
    // SampleFrameBuffer in front of it (iq_frontend.cpp:29-30, frame_buffer.h).  `device` / `plans` are additions with defaults.
    void init(dsp::stream<dsp::complex_t>* in, double sampleRate, bool buffering, int decimRatio, bool dcBlocking, int fftSize, double fftRate,
              FFTWindow fftWindow, float* (*acquireFFTBuffer)(void* ctx), void (*releaseFFTBuffer)(void* ctx), void* fftCtx, int device = 0,
              const DecimPlans* plans = nullptr) {
        _in = in;
        _sampleRate = sampleRate;
        _buffering = buffering;
        _decimRatio = decimRatio;
        _dcBlocking = dcBlocking;
        _fftSize = fftSize;
        _fftRate = fftRate;
        _fftWindow = fftWindow;
        _acquire = acquireFFTBuffer;
        _release = releaseFFTBuffer;
        _fftCtx = fftCtx;
        if (plans) { _plans = *plans; }
        _device = device;
        int rc = sdrpp_create(device, SDRPP_GPU_MAX_BLOCK, &ctx);
        if (rc) { throw std::runtime_error(std::string("[sdrpp_gpu::IQFrontEnd] ") + sdrpp_strerror(rc)); }
        sdrpp_set_deferred(ctx, 1);  // blocks are staged by stage() and processed by the first read of deliver(): one block or a whole backlog
        registerInput(_in);
        updatePreproc();
        updateFFTPath();
        _block_init = true;
    }

    // iq_frontend.cpp:72-74 -> SampleFrameBuffer::setInput (frame_buffer.h:36-44)
    void setInput(dsp::stream<dsp::complex_t>* in) {
        assert(_block_init);
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        tempStop();
        unregisterInput(_in);
        _in = in;
        registerInput(_in);
        tempStart();
    }

    // iq_frontend.cpp:101-103: inBuf.bypass = !enabled — takes effect with the next block, no restart
    void setBuffering(bool enabled) { _buffering = enabled; }
    // Addition (not in the reference): pipelined execution of the bypass path (sdrpp_set_pipelined, include/sdrpp_gpu.h).  Every block is
    // ONE kernel launch; its dB lines and VFO blocks are handed to the callback pair / output streams `lagBlocks` blocks later (the
    // reference's own graph is a pipeline of one thread per block with a stream hand-over between each pair, a few blocks deep as well).
    // Same results, bit for bit; at the reference's block size the front end takes blocks 3-4x faster than with one pass per block.
    // What the device cannot pipeline (the reference-order arithmetic of the pre-processing chain, a retune hand-over in progress: include/sdrpp_gpu.h)
    // falls back to one pass per block by itself; banks of any size — a single VFO too — run pipelined.
    // lagBlocks should not be smaller than the depth of the device pipeline (6-8 launches for a radio bank + FFT): asking for a block's
    // results earlier makes the library run the missing stages without new input, launch by launch.
    // Takes effect with the next block.  The tail still in flight when the input stops is handed out by drainPipeline() (worker stopped)
    // or with the next blocks.
    // groupBlocks > 1 (round 6): up to that many consecutive blocks share ONE launch while the device is the slower side
    // (sdrpp_set_pipeline_group, adaptive: a block goes out at once while the device has fewer than two launches in flight — a real-time stream
    // keeps one launch per block and its latency).  Results are handed out as soon as they are complete (polled once per block), at the latest
    // when (depth + 1) x groupBlocks blocks are outstanding.
    void setPipelining(bool enabled, int lagBlocks = 8, int groupBlocks = 1) {
        _pipeLag = lagBlocks < 1 ? 1 : (lagBlocks > 12 ? 12 : lagBlocks);
        _pipeGroup = groupBlocks < 1 ? 1 : (groupBlocks > SDRPP_GROUP_MAX ? SDRPP_GROUP_MAX : groupBlocks);
        _pipelining = enabled;
    }
    // Hands out everything still in flight (call with the block stopped, or from a control operation): -1 if a stream was stopped
    int drainPipeline() {
        int rc = finishDelivery();
        while (!pendingTickets.empty()) {  // (a block whose hand-over failed — a stopped stream — is gone, like a block the reference had in flight)
            const uint64_t t = pendingTickets.front();
            pendingTickets.erase(pendingTickets.begin());
            if (startDelivery(t) < 0 || finishDelivery() < 0) { rc = -1; }
        }
        tapPending.clear();
        return rc;
    }
    // iq_frontend.cpp:200-202 -> SampleFrameBuffer::flush (frame_buffer.h:46-49): drop what is queued
    void flushInputBuffer() {
        std::unique_lock<std::mutex> lck(frameMtx);
        frameRead = frameWrite;
    }

    // iq_frontend.cpp:105-130: the effective sample rate changes with the decimation; every VFO and the FFT framing follow it
    void setDecimation(int ratio) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        tempStop();
        _decimRatio = ratio;
        updatePreproc(2);
        for (auto& kv : vfos) { kv.second->setInSamplerate(getEffectiveSamplerate()); }
        updateFFTPath();
        tempStart();
    }
    void setDCBlocking(bool enabled) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _dcBlocking = enabled; updatePreproc(3); tempStart(); }
    void setInvertIQ(bool enabled) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _invertIQ = enabled; updatePreproc(3); tempStart(); }

    void setSampleRate(double sampleRate) {  // iq_frontend.cpp:76-99
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        tempStop();
        _sampleRate = sampleRate;
        updatePreproc(3);  // the DC blocker's rate follows the effective sample rate (iq_frontend.cpp:85-86)
        for (auto& kv : vfos) { kv.second->setInSamplerate(getEffectiveSamplerate()); }
        updateFFTPath();
        tempStart();
    }
    double getSampleRate() { return _sampleRate / (double)_decimRatio; }           // iq_frontend.h:27
    double getEffectiveSamplerate() { return _sampleRate / (double)_decimRatio; }  // iq_frontend.cpp:244-246
    void setFFTSize(int size) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _fftSize = size; updateFFTPath(); tempStart(); }
    void setFFTRate(double rate) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _fftRate = rate; updateFFTPath(); tempStart(); }
    void setFFTWindow(FFTWindow w) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _fftWindow = w; updateFFTPath(); tempStart(); }

    // iq_frontend.cpp:132-138 -> Splitter::bindStream / unbindStream (routing/splitter.h:14-44): every bound stream receives each
    // block of the PRE-PROCESSED IQ (recorder, iq_exporter).  Same exceptions as the Splitter.
    void bindIQStream(dsp::stream<dsp::complex_t>* stream) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        if (std::find(iqStreams.begin(), iqStreams.end(), stream) != iqStreams.end()) { throw std::runtime_error("[Splitter] Tried to bind stream to that is already bound"); }
        tempStop();
        registerOutput(stream);
        iqStreams.push_back(stream);
        tempStart();
    }
    void unbindIQStream(dsp::stream<dsp::complex_t>* stream) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        auto sit = std::find(iqStreams.begin(), iqStreams.end(), stream);
        if (sit == iqStreams.end()) { throw std::runtime_error("[Splitter] Tried to unbind stream to that isn't bound"); }
        tempStop();
        iqStreams.erase(sit);
        unregisterOutput(stream);
        tempStart();
    }

    // iq_frontend.cpp:140-160: duplicate names are rejected with NULL
    RxVFO* addVFO(std::string name, double sampleRate, double bandwidth, double offset) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        if (vfos.find(name) != vfos.end()) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] Tried to add VFO with existing name.\n");
            return nullptr;
        }
        tempStop();
        RxVFO* v = new RxVFO;
        v->fe = this;
        v->name = name;
        v->inSamplerate = getEffectiveSamplerate();
        v->outSamplerate = sampleRate;
        v->bandwidth = bandwidth;
        v->offset = offset;
        vfos[name] = v;
        rebuild(*v);
        registerOutput(&v->out);
        registerOutput(&v->audio);
        tempStart();
        return v;
    }
    void removeVFO(std::string name) {  // iq_frontend.cpp:162-183
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        auto it = vfos.find(name);
        if (it == vfos.end()) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] Tried to remove a VFO that doesn't exist.\n");
            return;
        }
        tempStop();
        if (it->second->id >= 0) { sdrpp_vfo_remove(ctx, it->second->id); }
        unregisterOutput(&it->second->out);
        unregisterOutput(&it->second->audio);
        delete it->second;
        vfos.erase(it);
        tempStart();
    }
    // A several-GPU host gathers the waterfall lines of its streams on the display GPU over RCCL (sdrpp_gpu_rccl.h: BankLineGather): for that the
    // newest raw dB line of this stream is kept in DEVICE memory of this front end's GPU, refreshed behind every block that completes one — in
    // EVERY mode: bypass and buffered blocks copy it on the device, pipelined blocks (setPipelining) take it out of the block's result slot —
    // and nothing of it crosses the bus again until the gather's grouped send / receive.  The buffers belong to the worker (they are re-allocated
    // when the FFT size changes); another thread never gets their address: it asks for a COPY, made under the lock the worker takes to flip or
    // re-allocate them, so a gather can neither read a freed buffer nor a half-written line.
    void keepDeviceLine(bool enabled) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        tempStop();
        _keepDevLine = enabled;
        tempStart();
    }
    // Newest raw dB line -> `dstDevice` (memory of THIS front end's device, room for `capacity` floats).  Returns the line's length (= the FFT
    // size it was computed with), 0 if no line has been kept yet, -(length) if it does not fit `capacity` (nothing copied), INT_MIN on a copy error.
    int copyLatestLineDevice(float* dstDevice, int capacity) {
        std::lock_guard<std::mutex> lck(devLineMtx);
        if (devLineCur < 0 || devLineSize <= 0) { return 0; }
        if (devLineSize > capacity) { return -devLineSize; }
        if (sdrpp_device_copy(ctx, dstDevice, devLine[devLineCur], (size_t)devLineSize * sizeof(float), 1)) { return INT_MIN; }
        return devLineSize;
    }
    int deviceLineSize() {  // length of the line copyLatestLineDevice would hand out now (0: none yet)
        std::lock_guard<std::mutex> lck(devLineMtx);
        return devLineCur >= 0 ? devLineSize : 0;
    }
    int device() const { return _device; }

    // The VFO whose `out` stream this is (NULL if none): how a demod::Demodulator-shaped adaptor (sdrpp_gpu_radio.h), which is handed
    // `&vfo->out` as its input stream exactly like a CPU demodulator, finds the channel it is fused into.
    RxVFO* vfoOfStream(dsp::stream<dsp::complex_t>* stream) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        for (auto& kv : vfos) {
            if (&kv.second->out == stream) { return kv.second; }
        }
        return nullptr;
    }
    // Control calls that do not restart the worker in the reference (RxVFO::setOffset, SSB's translation) reach the context through the
    // worker: the C-ABI context belongs to one thread at a time.  While the worker runs they are queued and applied in front of the next
    // block (where the reference's mutexes let them in as well); otherwise they are applied at once.  Call with ctrlMtx held.
    void control(std::function<void()> op) {
        if (running && !tempStopped) {
            std::lock_guard<std::mutex> lck(ctlMtx);
            ctlOps.push_back(std::move(op));
        }
        else { op(); }
    }

    // Parity switch (not in the reference): run the reference's float rotator recursion on the device instead of the closed-form NCO
    // (sdrpp_set_nco_mode).  Rebuilds every VFO.
    void setReferenceRotator(bool enabled) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        tempStop();
        for (auto& kv : vfos) {
            if (kv.second->id >= 0) { sdrpp_vfo_remove(ctx, kv.second->id); kv.second->id = -1; }
        }
        sdrpp_set_nco_mode(ctx, enabled ? SDRPP_NCO_REFERENCE_ROTATOR : SDRPP_NCO_CLOSED_FORM);
        for (auto& kv : vfos) { rebuild(*kv.second); }
        tempStart();
    }

    // Parity switch (not in the reference): the pre-processing chain (decimation / DC blocking) in the reference's own arithmetic — VOLK's
    // generic tap-ordered dot product and the sequential DC-blocker recursion — so that the pre-processed stream and the waterfall lines
    // behind it are bit-identical to a CPU build (sdrpp_preproc_set_reference_order).  A few times real time instead of thousands.
    void setReferenceArithmetic(bool enabled) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        tempStop();
        sdrpp_preproc_set_reference_order(ctx, enabled ? 1 : 0);
        tempStart();
    }

    // Worker of the block: one block from the input stream -> processed at once (bypass, the file source's setting:
    // file_source/src/main.cpp:74) or queued in the 32-slot frame buffer and processed by the second worker
    // (SampleFrameBuffer::run / worker, frame_buffer.h:51-98; same index arithmetic, so an overrun drops a whole lap like the reference).
    int run() override {
        const int rc = runBlock();
        if (rc >= 0) { _blocksTaken.fetch_add(1, std::memory_order_release); }
        return rc;
    }
    // Addition (not in the reference): blocks the worker has taken from its input stream and finished handling (processed, pushed or queued in the
    // frame buffer).  A control thread that has handed over block k and sees k + 1 here knows that a setter it calls now takes effect from
    // block k + 1 on — how tests/host_cpp/test_reconfig.cpp makes a reconfiguration schedule deterministic.
    uint64_t blocksTaken() const { return _blocksTaken.load(std::memory_order_acquire); }
    // How long a stop (every setter's tempStop) lets the block in hand and the open hand-over finish before it stops the writers (doStop);
    // default 250 ms — only a sink that does not read ever waits that long.
    void setStopGrace(int milliseconds) { _stopGraceMs = milliseconds < 0 ? 0 : milliseconds; }

    // Addition (not in the reference): how long the worker LOOKS for the next block before it goes to sleep in stream::read().  The reference's hand-over
    // is a condition variable: a reader that sleeps is woken through the kernel, 10-30 us later on a server part (100 us out of a deep idle state) —
    // at sample_rate / 200 blocks that wake-up, not the device, is most of a block's time (profiles/r06f_seam_prof.log: 14-30 us of a 36-50 us cycle).
    // swap() exchanges the stream's PUBLIC buffer pointers before it raises dataReady (dsp/stream.h:43-67), so a changed readBuf says "a block is
    // there" without touching the stream's private state; read() then returns at once.  0 = always sleep (the reference's behaviour).
    void setSpinWait(int microseconds) { _spinUs = microseconds < 0 ? 0 : microseconds; }

#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define SDRPP_GPU_NO_TSAN __attribute__((no_sanitize("thread")))
#endif
#endif
#if !defined(SDRPP_GPU_NO_TSAN) && defined(__SANITIZE_THREAD__)
#define SDRPP_GPU_NO_TSAN __attribute__((no_sanitize("thread")))
#endif
#ifndef SDRPP_GPU_NO_TSAN
#define SDRPP_GPU_NO_TSAN
#endif
    // (a peek at a pointer another thread exchanges under its lock: only ever a hint — read() is what takes the block)
    // The input stream's buffer is read by THIS side's threads and filled again by the source's thread two blocks later.  Lines that other cores have read
    // cost the writer an ownership request each: at the seam a SpeedTester-style source (speed_tester.h:75-81: one memcpy + swap() per block) needed
    // 38-44 us for its 400 KB memcpy — ~10 GB/s, the whole cycle of a block — while this side's copy took 8.  So the copy threads write the lines back and
    // drop them from every cache (clflushopt) once they are in the staging slot: the source's fill 41 -> 15 us, the seam 1 030-1 190 -> 1 590-1 650 MS/s
    // (same box, alternating runs: profiles/r06v_seam_evict.log).  SDRPP_GPU_STAGE_EVICT=0 turns it off (measurements); not x86-64 / no clflushopt: off.
    static bool cpuHasClflushopt() {
#if defined(__x86_64__)
        unsigned a = 7, b = 0, c = 0, d = 0;
        __asm__ volatile("cpuid" : "+a"(a), "=b"(b), "+c"(c), "=d"(d));
        return ((b >> 23) & 1u) != 0;
#else
        return false;
#endif
    }
    static void evictLines(const void* ptr, size_t n) {
#if defined(__x86_64__)
        const char* p = (const char*)ptr;
        const char* e = p + n;
        for (const char* q = (const char*)((uintptr_t)p & ~(uintptr_t)63); q < e; q += 64) { __asm__ volatile(".byte 0x66; clflush %0" : "+m"(*(volatile char*)q)); }  // (= clflushopt)
        __asm__ volatile("sfence" ::: "memory");
#else
        (void)ptr; (void)n;
#endif
    }
    bool _stageEvict = cpuHasClflushopt() && !(getenv("SDRPP_GPU_STAGE_EVICT") && atoi(getenv("SDRPP_GPU_STAGE_EVICT")) == 0);
    SDRPP_GPU_NO_TSAN static const void* peekReadBuf(dsp::stream<dsp::complex_t>* st) { return __atomic_load_n(reinterpret_cast<void* const*>(&st->readBuf), __ATOMIC_RELAXED); }

    int runBlock() {
#ifdef SDRPP_GPU_BLOCKS_PROF
        pipeT = std::chrono::steady_clock::now();
#endif
        // The staging copy of a HELD block may still be running (its push returned without waiting for it): the copy's last part is what flush()es the input
        // stream, and read() must not be called before that — the stream would hand out the same block again (dsp/stream.h: dataReady stays up until
        // flush()).  Nothing is lost by waiting here: the source cannot swap() the next block in before the flush either.
        stagers.finish();
        // Blocks held for a launch group (setPipelining's groupBlocks > 1) go out when the group is full — or HERE, when no new block has shown up for
        // _kickUs: the host side of "adaptive".  (The library's own rule — send a push at once while the device has fewer than two launches in flight —
        // never groups at this seam: the device is always ahead of a dsp::stream.  A real-time stream pays _kickUs of latency, nothing else.)
        if (spinStream == _in && spinLast && (_spinUs > 0 || heldBlocks > 0)) {
            const auto t0 = std::chrono::steady_clock::now();
            // (how long "no new block" is follows the stream: twice what the last blocks took to arrive after the flush — a source that needs 22 us per block
            // must not have its groups sent off after 20 —, at least _kickUs, at most 200 us / the spin window)
            static const bool kickFixed = getenv("SDRPP_GPU_KICK_FIXED") != nullptr;  // (measurement switch)
            const long kickUs = std::min<long>(kickFixed ? (long)_kickUs : std::max<long>(_kickUs, std::min<long>(200, (long)(2.0 * arriveEmaUs) + 5)), _spinUs);
            unsigned n = 0;
            bool arrived = true;
            while (peekReadBuf(_in) == spinLast) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
                if ((++n & 15u) == 0) {
                    const long us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                    if (heldBlocks > 0 && us >= kickUs && launchHeld() < 0) { return -1; }
                    if (us >= _spinUs) { arrived = false; break; }
                }
            }
            if (arrived) {
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                arriveEmaUs += 0.125 * (us - arriveEmaUs);
            }
        }
        else if (heldBlocks > 0 && launchHeld() < 0) { return -1; }
        int count = _in->read();
        spinStream = _in;
        spinLast = peekReadBuf(_in);
        if (count < 0) { return -1; }
        SDRPP_PIPE_TICK(0)
        if (!_buffering) {
            drainControl();
            if (_pipelining && pipelineEligible()) {
                if (pipeOn && (pipeFlags != pipelineFlags() || pipeGroupOn != _pipeGroup) && leavePipelined() < 0) { return -1; }  // (a stream was bound / a chain configured / the group size changed since)
                if (!pipeOn && enterPipelined() < 0) { return -1; }
                if (!iqStreams.empty() && !(pipeFlags & 8)) {  // bound streams, no chain in front: they receive this very block (Splitter::run) — when its turn comes
                    tapPending.emplace_back(0, std::vector<dsp::complex_t>(_in->readBuf, _in->readBuf + count));
                }
                // The block goes into the library's page-locked staging slot — the largest single host cost of a block (400 KB at sr/200 of
                // 10 MS/s) — and is fetched from there by the launch.  Round 3b, in the order that frees the source soonest: the copy starts at
                // once on threads of its own (`stagers`), the one that finishes last frees the stream buffer; meanwhile this thread plans the new block inside
                // sdrpp_push_staged_when — the job tables do not depend on the samples — and the library holds the block's launch back until the
                // last part has landed.  (Per block the reference's stream costs two futex wake-ups, source -> worker -> source, ~10 us each:
                // the time from read() to flush() is the one part of that cycle this side controls; it went from ~19 us to the copy's ~7.)
                float* slot = nullptr;
                int prc = sdrpp_push_stage(ctx, count, &slot);
                if (!prc) {
                    const char* srcb = (const char*)_in->readBuf;
                    char* dstb = (char*)slot;
                    const size_t bytes = (size_t)count * sizeof(dsp::complex_t), parts = bytes >= (size_t)(64 << 10) ? (size_t)kStagers : 1, per = ((bytes / parts) + 63) & ~(size_t)63;
                    std::vector<std::function<void()>> cj;
                    size_t njobs = 0;
                    for (size_t q = 0; q < parts; q++) { njobs += (q * per < bytes) ? 1 : 0; }
                    // (one pair of words per block, a ring longer than the largest launch group: the library keeps the word of a HELD block until
                    // the group's launch — sdrpp_set_pipeline_group flag 2 — and the copy of a held block runs on while this thread is back in read())
                    stageWord = (stageWord + 1) % kStageWords;
                    std::atomic<uint32_t>& stageLeft = stageLeftRing[stageWord];
                    std::atomic<uint32_t>& stagePending = stagePendingRing[stageWord];
                    stageLeft.store((uint32_t)njobs, std::memory_order_relaxed);
                    stagePending.store((uint32_t)njobs, std::memory_order_release);
                    for (size_t q = 0; q < parts; q++) {
                        const size_t o = q * per, n = o >= bytes ? 0 : std::min(per, bytes - o);
                        if (!n) { continue; }
                        std::atomic<uint32_t>* left = &stageLeft;
                        std::atomic<uint32_t>* pend = &stagePending;
                        cj.emplace_back([this, srcb, dstb, o, n, left, pend]() {
                            memcpy(dstb + o, srcb + o, n);
                            if (_stageEvict) { evictLines(srcb + o, n); }
                            if (left->fetch_sub(1, std::memory_order_acq_rel) == 1) { _in->flush(); }  // the stream buffer is free (BEFORE the word reaches 0: the worker's next read() follows it)
                            pend->fetch_sub(1, std::memory_order_release);
                        });
                    }
                    stagers.begin(std::move(cj));
                    // The hand-over of the block delivered last (32 swap()s = 32 futex wake-ups on the helpers, ~16-18 us of wall time) is NOT joined here
                    // any more: it runs while this thread plans and launches the new block, startDelivery joins it below.  (Round 6: with the source no
                    // longer the slow side — see evictLines — the join here had become 15.7 us of a 34 us cycle, profiles/r06v_seam_prof_evict.log.)
                    SDRPP_PIPE_TICK(1)
                    static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "the pending word is handed to the C ABI as a plain uint32_t");
                    prc = sdrpp_push_staged_when(ctx, count, reinterpret_cast<const volatile uint32_t*>(&stagePending));
                    if (prc) { stagers.finish(); }  // (a failed plan returns without waiting for the copy: join it)
                    else if (pipeGroupOn > 1) {
                        int64_t gst[5] = {};
                        heldBlocks = sdrpp_pipeline_group_stats(ctx, gst, 5) >= 5 ? (int)gst[4] : 0;
                    }
                }
                else {
                    (void)finishDelivery();
                    _in->flush();
                }
                if (prc) {
                    fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] push failed: %s\n", sdrpp_last_error(ctx));
                    return -1;
                }
                pendingTickets.push_back(sdrpp_ticket(ctx));
                if (!tapPending.empty() && tapPending.back().first == 0) { tapPending.back().first = sdrpp_ticket(ctx); }
                SDRPP_PIPE_TICK(2)
                // (a block's results are complete `depth` launches after its push — 6-7 levels for a WFM bank + FFT, 12 with the AF chain, 3 more behind a
                // pre-processing chain: asking for them sooner makes the library run the missing stages as launches without new input)
                int64_t pst[SDRPP_PIPELINE_STATS_HEAD] = {};
                int lagNow = (sdrpp_pipeline_stats(ctx, pst, SDRPP_PIPELINE_STATS_HEAD) >= 5) ? std::min(SDRPP_RESULT_SLOTS - 2, std::max(_pipeLag, (int)pst[4] + 1)) : _pipeLag;
                lagNow *= pipeGroupOn;  // (`depth` counts LAUNCHES: with launch groups a block is complete depth launches of up to pipeGroupOn blocks later)
                // one block out per block in, its hand-over on the helpers while the next block arrives — the oldest block when more than lagNow are
                // outstanding (the device is the slower side: waiting for it is what back-pressure means), else the oldest one only if it is
                // complete already (a real-time stream: results leave as soon as they exist, not a fixed number of blocks late)
                if (!pendingTickets.empty() && ((int)pendingTickets.size() > lagNow || (pipeGroupOn > 1 && sdrpp_result_ready(ctx, pendingTickets.front()) == 1))) {
                    const uint64_t t = pendingTickets.front();
                    pendingTickets.erase(pendingTickets.begin());
                    if (startDelivery(t) < 0) { return -1; }
                }
                SDRPP_PIPE_TICK(3)
#ifdef SDRPP_GPU_BLOCKS_PROF
                pipeBlocks++;
#endif
                return count;
            }
            if (pipeOn && leavePipelined() < 0) { return -1; }
            int rc = stage((const dsp::complex_t*)_in->readBuf, count);  // the H2D copy is complete on return: the stream buffer is free
            if (_stageEvict) { evictLines(_in->readBuf, (size_t)count * sizeof(dsp::complex_t)); }
            _in->flush();
            if (rc >= 0) { rc = deliver((const dsp::complex_t*)nullptr, count); }
            return rc < 0 ? -1 : count;
        }
        if (pipeOn && leavePipelined() < 0) { return -1; }  // (the frame worker only touches the context once a frame is queued)
        {
            std::lock_guard<std::mutex> lck(frameMtx);
            if (!frames[frameWrite]) {  // page-locked: the copy to the device runs straight out of the slot
                frames[frameWrite] = (dsp::complex_t*)sdrpp_host_alloc(sizeof(dsp::complex_t) * STREAM_BUFFER_SIZE);
                if (!frames[frameWrite]) { throw std::runtime_error("[sdrpp_gpu::IQFrontEnd] cannot allocate a frame-buffer slot"); }
            }
            memcpy(frames[frameWrite], _in->readBuf, (size_t)count * sizeof(dsp::complex_t));
            if (_stageEvict) { evictLines(_in->readBuf, (size_t)count * sizeof(dsp::complex_t)); }
            frameSizes[frameWrite] = count;
            frameWrite = (frameWrite + 1) % FRAME_SLOTS;
        }
        frameCnd.notify_all();
        _in->flush();
        return count;
    }

    // Direct ingest of one block by a source that drives the front end itself (sdrpp_gpu::WavSource::start(IQFrontEnd*)): the block is
    // processed and its results are handed out before the call returns, exactly like a block taken from the input stream in bypass
    // mode.  ingestInt16 takes interleaved 16-bit IQ as file_source reads it from a WAV (main.cpp:160-162): the conversion x / 32768 runs on
    // the device (sdrpp_push_int16) and the bus carries half the bytes.  The front end's own worker must not be running (no start()):
    // returns -1 if it is, or when an output stream has been stopped.
    int ingestInt16(const int16_t* iq, int count) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        if (running || count <= 0 || count > SDRPP_GPU_MAX_BLOCK) { return -1; }
        if (pipeOn && leavePipelined() < 0) { return -1; }
        if (sdrpp_push_int16(ctx, iq, count)) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] push failed: %s\n", sdrpp_last_error(ctx));
            return -1;
        }
        if (!iqStreams.empty()) {  // the bound consumers receive floats (what the Splitter would have copied)
            tapCopy.resize((size_t)count);
            for (int i = 0; i < count; i++) { tapCopy[(size_t)i] = dsp::complex_t{ (float)iq[2 * i] / 32768.0f, (float)iq[2 * i + 1] / 32768.0f }; }
        }
        return deliver((const dsp::complex_t*)nullptr, count);
    }
    int ingestFloat(const dsp::complex_t* iq, int count) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        if (running || count <= 0 || count > SDRPP_GPU_MAX_BLOCK) { return -1; }
        if (pipeOn && leavePipelined() < 0) { return -1; }
        if (stage(iq, count) < 0) { return -1; }
        return deliver((const dsp::complex_t*)nullptr, count);
    }

    sdrpp_ctx* context() { return ctx; }
#ifdef SDRPP_GPU_BLOCKS_PROF
    // diagnostic build of a host program: where the worker's wall time goes (microseconds, passes)
    double profUs[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    long profPasses = 0;
    double pipeUs[5] = { 0, 0, 0, 0, 0 };  // pipelined bypass, per block: wait for the block | join the previous hand-over | stage + plan + launch | result wait + start of the hand-over
    long pipeBlocks = 0;
    std::chrono::steady_clock::time_point pipeT;
    std::chrono::steady_clock::time_point profT;
    void profReport() {
        const double n = (double)std::max(1L, profPasses);
        fprintf(stderr, "[sdrpp_gpu blocks prof] passes %ld, us per pass: wait for frames %.1f | stage (H2D) %.1f | process + line count %.1f | FFT lines out %.1f | bound IQ streams %.1f | VFO read (D2H) %.1f | VFO hand-off (memcpy + swap) %.1f\n",
                profPasses, profUs[0] / n, profUs[1] / n, profUs[2] / n, profUs[3] / n, profUs[4] / n, profUs[5] / n, profUs[6] / n);
        const double nb = (double)std::max(1L, pipeBlocks);
        fprintf(stderr, "[sdrpp_gpu blocks prof] pipelined blocks %ld, us per block: wait for the block %.1f | join previous hand-over %.1f | stage + plan + launch %.1f | result wait + hand-over start %.1f\n",
                pipeBlocks, pipeUs[0] / nb, pipeUs[1] / nb, pipeUs[2] / nb, pipeUs[3] / nb);
    }
#define SDRPP_BLOCKS_TICK(slot) { const auto _now = std::chrono::steady_clock::now(); profUs[slot] += std::chrono::duration<double, std::micro>(_now - profT).count(); profT = _now; }
#define SDRPP_BLOCKS_TICK0() { profT = std::chrono::steady_clock::now(); }
#else
#define SDRPP_BLOCKS_TICK(slot)
#define SDRPP_BLOCKS_TICK0()
#endif
    static constexpr int64_t SDRPP_GPU_MAX_BLOCK = 1000000;
#ifndef SDRPP_GPU_HELPERS
#define SDRPP_GPU_HELPERS 8
#endif
    int kStagers = [] { const char* e = getenv("SDRPP_GPU_STAGERS"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : (v > 16 ? 16 : v); }();  // threads of the staging copy (measurement switch)
    static constexpr int kHelpers = SDRPP_GPU_HELPERS;  // threads that hand a block's outputs to the streams (round 6: 8 — with 6, the hand-over of 32 streams took as long as planning the next block: profiles/r06d_seam_prof.log; 12 / 16 / 32 are progressively SLOWER, with or without a back-off or a yield in the idle loop: r06w_seam_helpers.log, r06w_seam_pauses.log)

protected:
    // dsp::block hooks: the frame-buffer worker lives and dies with the block's own worker (SampleFrameBuffer::doStart / doStop,
    // frame_buffer.h:105-124)
    void doStart() override {
        stopFrameWorker = false;
        helpers.start(kHelpers);  // hand-overs: 32 stream swaps per block are 32 futex wake-ups (~3 us each for the waker)
        stagers.start(kStagers);
        workerDone.store(false, std::memory_order_relaxed);
        workerThread = std::thread([this]() {
            workerLoop();
            workerDone.store(true, std::memory_order_release);
        });
        frameThread = std::thread(&IQFrontEnd::frameWorker, this);
    }
    void doStop() override {
        for (auto& in : inputs) { in->stopReader(); }
        // The reference stops readers and writers together: whatever a worker has in flight towards a stream is dropped (its swap() returns
        // false).  There that is one block of one stage; here the worker's current block AND the hand-over batch on the helpers (30+ swaps: a
        // whole block's outputs of every VFO) would go — at EVERY setter a GUI calls while blocks flow.  So the writers get a grace period first:
        // the worker finishes the block it holds (its read() fails next) and the open hand-over completes — microseconds when the sinks read —
        // and only a sink that does not read (the case stopWriter() exists for) runs into the time-out and loses the block, as in the reference.
        {
            // (every setter goes through here with ctrlMtx held — a sink that has stopped reading must not cost EVERY setter the whole grace
            // period: once a stop has run into the time-out, the following ones wait 5 ms at most until one completes in time again)
            const auto t0 = std::chrono::steady_clock::now();
            const auto limit = std::chrono::milliseconds(_lastStopTimedOut ? std::min(_stopGraceMs, 5) : _stopGraceMs);
            bool pending = true;
            while ((pending = (!workerDone.load(std::memory_order_acquire) || helpers.busy())) && std::chrono::steady_clock::now() - t0 < limit) {
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            _lastStopTimedOut = pending;
        }
        for (auto& out : outputs) { out->stopWriter(); }
        {
            std::lock_guard<std::mutex> lck(frameMtx);
            stopFrameWorker = true;
        }
        frameCnd.notify_all();
        if (workerThread.joinable()) { workerThread.join(); }
        if (frameThread.joinable()) { frameThread.join(); }
        // close an open hand-over batch before the helpers go: its unclaimed jobs hold RxVFO pointers (inflightOrder) that a removeVFO after
        // this stop would leave dangling for the next start's helpers.  The writers are stopped, so the swaps fail fast.
        (void)finishDelivery();
        inflightOrder.clear();
        helpers.stop();
        stagers.stop();
        for (auto& in : inputs) { in->clearReadStop(); }
        for (auto& out : outputs) { out->clearWriteStop(); }
    }

private:
    friend class RxVFO;
    static constexpr int FRAME_SLOTS = 32;  // TEST_BUFFER_SIZE, frame_buffer.h:3

    // Second worker (SampleFrameBuffer::worker, frame_buffer.h:76-98).  The reference hands the queued frames on one by one; here
    // EVERYTHING that is queued goes to the device as one deferred pass (every frame still a block of its own for the block-dependent
    // operations), so a backlog is worked off at large-batch speed instead of launch-bound frame by frame.  With consumers of the
    // wideband IQ bound (bindIQStream) the frames are handed on one by one, as those consumers receive them block-wise.
    void frameWorker() {
        while (true) {
            int staged = 0, last = 0;
            SDRPP_BLOCKS_TICK0()
            {
                std::unique_lock<std::mutex> lck(frameMtx);
                frameCnd.wait(lck, [this]() { return (((frameWrite - frameRead + FRAME_SLOTS) % FRAME_SLOTS) > 0) || stopFrameWorker; });
                if (stopFrameWorker) { break; }
                SDRPP_BLOCKS_TICK(0)
                lck.unlock();
                drainControl();  // may run a pass (a retune first processes what is staged): not under the producer's lock
                lck.lock();
                // under the lock, like the reference's copy into out.writeBuf.  The slots are page-locked: the device fetches the frames
                // itself (sdrpp_push_pinned_async: a kernel launch per frame, no wait), ONE wait covers the whole backlog
                while (((frameWrite - frameRead + FRAME_SLOTS) % FRAME_SLOTS) > 0) {
                    const int count = frameSizes[frameRead];
                    if (staged > 0 && ((int64_t)staged + count > SDRPP_GPU_MAX_BLOCK || !iqStreams.empty())) { break; }
                    if (!iqStreams.empty()) {  // the bound consumers need this block's samples after the lock is gone
                        tapCopy.assign(frames[frameRead], frames[frameRead] + count);
                    }
                    if (sdrpp_push_pinned_async(ctx, (const float*)frames[frameRead], count)) {
                        fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] push failed: %s\n", sdrpp_last_error(ctx));
                        return;
                    }
                    frameRead = (frameRead + 1) % FRAME_SLOTS;
                    staged += count;
                    last = count;
                }
                if (sdrpp_push_wait(ctx)) { return; }
            }
            SDRPP_BLOCKS_TICK(1)
            if (deliver(iqStreams.empty() ? nullptr : tapCopy.data(), last) < 0) { break; }
        }
    }

    // ---- pipelined bypass (setPipelining) ----
    // (round 4: the pre-processing chain, the radio's AF chain and bound IQ streams no longer keep a front end out of pipelined mode — their
    // stages are levels of the block on the device like everything else, and what they deliver comes out of the block's result slot)
    bool pipelineEligible() { return true; }
    // what every block delivers into its result slot: every VFO's block (the AF chain's output where one is attached), the raw dB lines, and —
    // with streams bound AND a pre-processing chain in front — the pre-processed IQ (without a chain the bound streams get the input block itself)
    int pipelineFlags() const {
        const bool pre = _decimRatio > 1 || _dcBlocking || _invertIQ;
        return 1 | 4 | ((pre && !iqStreams.empty()) ? 8 : 0);
    }
    int enterPipelined() {
        if (sdrpp_sync(ctx)) { return -1; }  // (nothing is staged in bypass mode; a deferred pass left over from buffered mode runs here)
        sdrpp_set_deferred(ctx, 0);
        pipeFlags = pipelineFlags();
        if (sdrpp_set_pipelined(ctx, 1, pipeFlags)) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] pipelined mode refused: %s\n", sdrpp_last_error(ctx));
            sdrpp_set_deferred(ctx, 1);
            _pipelining = false;
            return 0;
        }
        pipeGroupOn = _pipeGroup;
        // fixed groups + words that live until the launch (flag 2): WHEN a group goes out is decided in runBlock (full, or no new block for _kickUs)
        sdrpp_set_pipeline_group(ctx, pipeGroupOn, 2);
        heldBlocks = 0;
        pipeOn = true;
        return 0;
    }
    int launchHeld() {
        heldBlocks = 0;
        if (sdrpp_pipeline_launch_held(ctx)) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] launch of the held blocks failed: %s\n", sdrpp_last_error(ctx));
            return -1;
        }
        return 0;
    }
    int leavePipelined() {
        heldBlocks = 0;
        const int rc = drainPipeline();
        sdrpp_set_pipeline_group(ctx, 1, 0);
        sdrpp_set_pipelined(ctx, 0, 0);
        sdrpp_set_deferred(ctx, 1);
        pipeOn = false;
        return rc;
    }
    // one block's results: lines through acquire / release, one block out on every VFO stream (the VFOs the block was processed with:
    // a VFO removed or rebuilt since has a new handle and is skipped)
    // (startDelivery hands the copies + swaps to the helpers and returns; finishDelivery joins them and releases the slot)
    int startDelivery(uint64_t ticket) {
        (void)finishDelivery();
        sdrpp_result& r = inflight;
        const int wrc = sdrpp_result_wait(ctx, ticket, &r);
        if (wrc == SDRPP_ERR_NOT_FOUND) {  // no slot for this block any more (the context was rebuilt around it): the block is gone, like one the reference had in flight across a tempStop
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] block %llu skipped: %s\n", (unsigned long long)ticket, sdrpp_last_error(ctx));
            return 0;
        }
        if (wrc) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] result of block %llu: %s\n", (unsigned long long)ticket, sdrpp_last_error(ctx));
            return -1;
        }
        std::vector<std::function<void()>> jobs;
        // Lines of a block that was pushed BEFORE a setFFTSize and is handed out after it have the old size; the display's buffers (acquire) have
        // the new one — the reference resizes the display first (gui::waterfall.setRawFFTSize, then IQFrontEnd::setFFTSize).  Such lines are
        // dropped, like the frame the reference's Reshaper has in progress when it restarts; the block's VFO outputs are delivered as always.
        const bool linesFit = r.fft_size == _fftSize;
        if (_keepDevLine && linesFit && r.n_lines > 0 && r.raw) { refreshDeviceLine(r.raw + (size_t)(r.n_lines - 1) * (size_t)r.fft_size, 0, r.fft_size); }
        if (linesFit && r.n_lines > 0 && r.raw) {
            jobs.emplace_back([this, r]() {
                for (int i = 0; i < r.n_lines; i++) {
                    float* buf = _acquire ? _acquire(_fftCtx) : nullptr;
                    if (buf) { memcpy(buf, r.raw + (size_t)i * (size_t)r.fft_size, (size_t)r.fft_size * sizeof(float)); }
                    if (_release) { _release(_fftCtx); }
                }
            });
        }
        deliveryFailed = false;
        // Splitter::run for the streams bound with bindIQStream: the pre-processed block out of the result slot, or the input block kept at the push
        while (!tapPending.empty() && tapPending.front().first < ticket) { tapPending.pop_front(); }  // (blocks that were never delivered)
        if (!iqStreams.empty()) {
            const dsp::complex_t* src = nullptr;
            int n = 0;
            if (r.n_iq > 0 && r.iq) {
                src = (const dsp::complex_t*)r.iq;
                n = r.n_iq;
            }
            else if (!tapPending.empty() && tapPending.front().first == ticket) {
                inflightTap.swap(tapPending.front().second);
                tapPending.pop_front();
                src = inflightTap.data();
                n = (int)inflightTap.size();
            }
            if (src && n > 0) {
                jobs.emplace_back([this, src, n]() {
                    for (auto* st : iqStreams) {
                        memcpy(st->writeBuf, src, (size_t)n * sizeof(dsp::complex_t));
                        if (!st->swap(n)) { deliveryFailed = true; }
                    }
                });
            }
        }
        std::vector<std::pair<RxVFO*, int>>& order = inflightOrder;
        order.clear();
        for (int k = 0; k < r.n_vfo; k++) {
            if (r.counts[k] <= 0) { continue; }
            for (auto& kv : vfos) {
                const RxVFO& cand = *kv.second;
                if (cand.id == r.ids[k] || std::find(cand.prevIds.begin(), cand.prevIds.end(), r.ids[k]) != cand.prevIds.end()) {
                    order.emplace_back(kv.second, k);
                    break;
                }
            }
        }
        const int groups = kHelpers;  // (one per helper)
        for (int g = 0; g < groups && !order.empty(); g++) {
            jobs.emplace_back([this, g, groups, &order, &r]() {
                std::atomic<bool>& failed = deliveryFailed;
                for (size_t q = (size_t)g; q < order.size(); q += (size_t)groups) {
                    RxVFO* v = order[q].first;
                    const int k = order[q].second, n = r.counts[k];
                    const float* src = r.samples + 2 * (size_t)r.offsets[k];
                    if (v->demod == Demod::RAW) {
                        memcpy(v->out.writeBuf, src, (size_t)n * sizeof(dsp::complex_t));
                        if (!v->out.swap(n)) { failed = true; }
                    }
                    else {
                        memcpy(v->audio.writeBuf, src, (size_t)n * sizeof(dsp::stereo_t));
                        if (!v->audio.swap(n)) { failed = true; }
                    }
                }
            });
        }
        inflightTicket = ticket;
        helpers.begin(std::move(jobs));
        return 0;
    }
    int finishDelivery() {
        if (!inflightTicket) { return 0; }
        helpers.finish();
        sdrpp_result_release(ctx, inflightTicket);
        inflightTicket = 0;
        return deliveryFailed ? -1 : 0;
    }

    void drainControl() {
        std::vector<std::function<void()>> ops;
        {
            std::lock_guard<std::mutex> lck(ctlMtx);
            ops.swap(ctlOps);
        }
        for (auto& op : ops) { op(); }
    }

    // keepDeviceLine: the newest raw line into the buffer that is NOT handed out, then the flip — all under devLineMtx (see copyLatestLineDevice).
    // hostLine != NULL: the line lies in host memory (a pipelined block's result slot); else line `index` of the ordinary pass's lines on the device.
    void refreshDeviceLine(const float* hostLine, int index, int fftSize) {
        std::lock_guard<std::mutex> lck(devLineMtx);
        if (devLineSize != fftSize) {
            for (int k = 0; k < 2; k++) {
                sdrpp_device_free(ctx, devLine[k]);
                devLine[k] = (float*)sdrpp_device_alloc(ctx, (size_t)fftSize * sizeof(float));
            }
            devLineSize = fftSize;
            devLineCur = -1;
        }
        const int nxt = devLineCur == 0 ? 1 : 0;
        if (!devLine[nxt]) { return; }
        const bool ok = hostLine ? sdrpp_device_copy(ctx, devLine[nxt], hostLine, (size_t)fftSize * sizeof(float), 0) == 0
                                 : (sdrpp_fft_copy_device(ctx, index, 1, devLine[nxt], nullptr, nullptr) >= 0 && sdrpp_sync(ctx) == 0);
        if (ok) { devLineCur = nxt; }
    }

    // one block to the device (copy only; the kernels run with the first read of deliver())
    int stage(const dsp::complex_t* data, int count) {
        int rc = sdrpp_push(ctx, (const float*)data, count);
        if (rc) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] push failed: %s\n", sdrpp_last_error(ctx));
            return -1;
        }
        if (!_buffering && !iqStreams.empty()) { tapCopy.assign(data, data + count); }
        return count;
    }

    // everything staged -> FFT lines through acquire/release, the pre-processed IQ on every bound stream, one block out on every VFO stream
    int deliver(const dsp::complex_t* data, int count) {
        if (!data && !iqStreams.empty()) { data = tapCopy.data(); }
        const int nlines = sdrpp_fft_lines(ctx);  // first observing call: processes what is staged
        if (nlines < 0) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] processing failed: %s\n", sdrpp_last_error(ctx));
            return -1;
        }
        SDRPP_BLOCKS_TICK(2)
        if (_keepDevLine && nlines > 0) { refreshDeviceLine(nullptr, nlines - 1, _fftSize); }  // the newest line stays on the device as well (keepDeviceLine)
        // all new lines with ONE device-to-host copy into page-locked staging; they are handed out below, next to the VFO blocks
        if (nlines > 0 && _acquire) {
            const size_t need = (size_t)nlines * (size_t)_fftSize;
            if (need > linesCap) {
                sdrpp_host_free(linesPin);
                linesPin = (float*)sdrpp_host_alloc((need + (size_t)4 * _fftSize) * sizeof(float));
                linesCap = linesPin ? need + (size_t)4 * _fftSize : 0;
                if (!linesPin) { return -1; }
            }
            if (sdrpp_fft_read(ctx, 0, nlines, linesPin, nullptr, nullptr) < 0) { return -1; }
        }
        std::vector<std::function<void()>> jobs;
        jobs.emplace_back([this, nlines]() {  // the acquire / release protocol is sequential (one buffer in flight): one job for all lines
            for (int i = 0; i < nlines; i++) {
                float* buf = _acquire ? _acquire(_fftCtx) : nullptr;  // may be NULL: still paired with release (iq_frontend.cpp:258-266)
                if (buf) { memcpy(buf, linesPin + (size_t)i * (size_t)_fftSize, (size_t)_fftSize * sizeof(float)); }
                if (_release) { _release(_fftCtx); }
            }
        });
        SDRPP_BLOCKS_TICK(3)
        if (!iqStreams.empty()) {  // Splitter::run: copy to every bound stream, then swap (blocking on the slowest consumer)
            const bool pre = _decimRatio > 1 || _dcBlocking || _invertIQ;
            for (auto* st : iqStreams) {
                int n = count;
                if (pre) { n = sdrpp_preproc_read(ctx, (float*)st->writeBuf, STREAM_BUFFER_SIZE); }
                else { memcpy(st->writeBuf, data, (size_t)count * sizeof(dsp::complex_t)); }
                if (n < 0) { return -1; }
                if (n > 0 && !st->swap(n)) { return -1; }
            }
        }
        SDRPP_BLOCKS_TICK(4)
        // every VFO's block with ONE device-to-host copy, then the per-stream hand-offs
        const int nv = (int)vfos.size();
        if (nv == 0) {
            helpers.run(std::move(jobs));
            return count;
        }
        ids.resize((size_t)nv);
        which.resize((size_t)nv);
        offs.resize((size_t)nv);
        cnts.resize((size_t)nv);
        int k = 0;
        for (auto& kv : vfos) {
            ids[(size_t)k] = kv.second->id;
            which[(size_t)k] = (kv.second->demod != Demod::RAW && kv.second->afOn) ? 2 : 0;
            k++;
        }
        auto gatherRoom = [this](size_t floats) {
            if (floats <= gatherCap) { return true; }
            sdrpp_host_free(gatherPin);
            gatherPin = (float*)sdrpp_host_alloc(floats * sizeof(float));
            gatherCap = gatherPin ? floats : 0;
            return gatherPin != nullptr;
        };
        if (!gatherRoom((size_t)2 * STREAM_BUFFER_SIZE)) { return -1; }
        int total = sdrpp_vfo_read_many(ctx, nv, ids.data(), which.data(), gatherPin, (int64_t)(gatherCap / 2), offs.data(), cnts.data());
        if (total == SDRPP_ERR_INVALID) {  // more than the staging holds (very many VFOs at a high output rate): size it and retry once
            long long need = 0;
            for (auto& kv : vfos) { need += std::max(sdrpp_vfo_out_count(ctx, kv.second->id), kv.second->afOn ? sdrpp_vfo_af_count(ctx, kv.second->id) : 0); }
            if (!gatherRoom((size_t)2 * (size_t)(need + 1024))) { return -1; }
            total = sdrpp_vfo_read_many(ctx, nv, ids.data(), which.data(), gatherPin, (int64_t)(gatherCap / 2), offs.data(), cnts.data());
        }
        if (total < 0) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] reading the VFO outputs failed: %s\n", sdrpp_last_error(ctx));
            return -1;
        }
        SDRPP_BLOCKS_TICK(5)
        // per-stream hand-offs (memcpy into the stream's write buffer + swap, which waits for the stream's reader), in three groups
        std::atomic<bool> failed{ false };
        {
            std::vector<std::pair<RxVFO*, int>> order;
            k = 0;
            for (auto& kv : vfos) { order.emplace_back(kv.second, k++); }
            const int groups = kHelpers;
            for (int g = 0; g < groups; g++) {
                jobs.emplace_back([this, g, groups, order, &failed]() {
                    for (size_t q = (size_t)g; q < order.size(); q += (size_t)groups) {
                        RxVFO* v = order[q].first;
                        const int n = cnts[(size_t)order[q].second];
                        const float* src = gatherPin + 2 * (size_t)offs[(size_t)order[q].second];
                        if (n <= 0) { continue; }
                        if (v->demod == Demod::RAW) {
                            memcpy(v->out.writeBuf, src, (size_t)n * sizeof(dsp::complex_t));
                            if (!v->out.swap(n)) { failed = true; }
                        }
                        else {
                            memcpy(v->audio.writeBuf, src, (size_t)n * sizeof(dsp::stereo_t));
                            if (!v->audio.swap(n)) { failed = true; }
                        }
                    }
                });
            }
        }
        helpers.run(std::move(jobs));
        if (failed) { return -1; }
        SDRPP_BLOCKS_TICK(6)
#ifdef SDRPP_GPU_BLOCKS_PROF
        profPasses++;
#endif
        return count;
    }

    // keep: what the reference's blocks carry across the setter that calls this — 3 for setSampleRate / setDCBlocking / setInvertIQ (decimator untouched, the
    // DC blocker's estimate lives on), 2 for setDecimation (new decimator stages), 0 at init
    void updatePreproc(int keep = 0) {  // iq_frontend.cpp:32-39: decim enabled for ratio > 1, dcBlock rate genDCBlockRate(effectiveSr), conjugate
        int dec[SDRPP_MAX_DECIM_STAGES] = { 0 }, nt[SDRPP_MAX_DECIM_STAGES] = { 0 };
        const float* tp[SDRPP_MAX_DECIM_STAGES] = { nullptr };
        int n = 0;
        if (_decimRatio > 1) {
            auto it = _plans.plans.find(_decimRatio);
            if (it == _plans.plans.end()) { throw std::runtime_error("[sdrpp_gpu::IQFrontEnd] no decimation plan for ratio " + std::to_string(_decimRatio)); }
            for (auto& st : it->second) {
                dec[n] = st.decimation;
                nt[n] = (int)st.taps.size();
                tp[n] = st.taps.data();
                n++;
            }
        }
        const float rate = _dcBlocking ? (float)(50.0 / getEffectiveSamplerate()) : 0.0f;  // iq_frontend.h:55-57
        int rc = sdrpp_preproc_reconfigure(ctx, n, dec, nt, tp, rate, _invertIQ ? 1 : 0, keep);
        if (rc) { throw std::runtime_error(std::string("[sdrpp_gpu::IQFrontEnd] ") + sdrpp_last_error(ctx)); }
    }

    void updateFFTPath() {  // iq_frontend.cpp:269-309
        int skip = 0, nz = 0;
        sdrpp_design_reshape_params(getEffectiveSamplerate(), _fftSize, _fftRate, &skip, &nz);
        std::vector<float> w((size_t)nz);
        sdrpp_design_fft_window((int)_fftWindow, nz, w.data());
        int rc = sdrpp_fft_configure(ctx, _fftSize, nz, skip, w.data());
        if (rc) { throw std::runtime_error(std::string("[sdrpp_gpu::IQFrontEnd] ") + sdrpp_last_error(ctx)); }
    }

    // (Re)creates the device-side VFO from the host-side description — what RxVFO::init + RationalResampler::reconfigure +
    // the radio demodulator constructors compute (rx_vfo.h:19-36, rational_resampler.h:120-165, demodulators/*.h).
    // keep (an existing VFO): what survives the change in the reference — bit 0 the RxVFO's own state (translation phase, channel filter's delay line:
    // setInSamplerate / setOutSamplerate, rx_vfo.h:35-58), bit 1 the demodulator behind it (setInSamplerate leaves that block alone); 0 = everything new
    void rebuild(RxVFO& v, int keep = 0) {
        const int oldId = (keep && v.id >= 0) ? v.id : -1;
        if (v.id >= 0 && oldId < 0) { sdrpp_vfo_remove(ctx, v.id); }
        // blocks pushed under the old handle may still be on their way (pipelined mode: results a few blocks behind their pushes): they are THIS VFO's
        // blocks and go out on its stream, in order, in front of the first block of the new description
        if (v.id >= 0) {
            v.prevIds.push_back(v.id);
            if (v.prevIds.size() > 8) { v.prevIds.erase(v.prevIds.begin()); }
        }
        v.id = -1;
        sdrpp_vfo_desc d;
        memset(&d, 0, sizeof(d));
        sdrpp_design_phase_delta(-v.offset, v.inSamplerate, &d.phase_delta_re, &d.phase_delta_im);
        int mode = 0, predec = 1, interp = 1, decim = 1;
        int nt = sdrpp_design_resampler(v.inSamplerate, v.outSamplerate, _plans.maxRatio, &mode, &predec, &interp, &decim, nullptr, 0);
        std::vector<float> rtaps((size_t)(nt > 0 ? nt : 1));
        if (nt > 0) { sdrpp_design_resampler(v.inSamplerate, v.outSamplerate, _plans.maxRatio, &mode, &predec, &interp, &decim, rtaps.data(), nt); }
        const std::vector<DecimStage>* st = nullptr;
        if ((mode == 0 || mode == 1) && predec > 1) {
            auto it = _plans.plans.find(predec);
            if (it == _plans.plans.end()) { throw std::runtime_error("[sdrpp_gpu::IQFrontEnd] no decimation plan for ratio " + std::to_string(predec)); }
            st = &it->second;
        }
        d.n_stages = st ? (int)st->size() : 0;
        for (int i = 0; i < d.n_stages; i++) {
            d.stage_decim[i] = (*st)[(size_t)i].decimation;
            d.stage_ntaps[i] = (int)(*st)[(size_t)i].taps.size();
            d.stage_taps[i] = (*st)[(size_t)i].taps.data();
        }
        d.interp = (mode == 0 || mode == 2) ? interp : 1;
        d.decim = (mode == 0 || mode == 2) ? decim : 1;
        d.resamp_ntaps = (mode == 0 || mode == 2) ? nt : 0;
        d.resamp_taps = rtaps.data();
        std::vector<float> ctaps;
        if (v.bandwidth != v.outSamplerate) {  // rx_vfo.h:24
            const double fw = v.bandwidth / 2.0;
            int n = sdrpp_design_low_pass(fw, fw * 0.1, v.outSamplerate, 0, nullptr, 0);
            ctaps.resize((size_t)n);
            sdrpp_design_low_pass(fw, fw * 0.1, v.outSamplerate, 0, ctaps.data(), n);
            d.chan_ntaps = n;
            d.chan_taps = ctaps.data();
        }
        d.demod = (int)v.demod;
        d.nco_mode = v.ncoMode;
        d.agc_set_point = 1.0f;
        d.agc_max_gain = 10e6;
        d.agc_max_output_amp = 10.0f;
        d.agc_init_gain = INFINITY;
        d.agc_attack = (float)(v.agcAttack / v.outSamplerate);
        d.agc_decay = (float)(v.agcDecay / v.outSamplerate);
        d.am_carrier_agc = v.carrierAgc ? 1 : 0;
        d.dc_block_rate = (float)(100.0 / v.outSamplerate);
        d.ssb_phase_delta_re = 1.0f;
        std::vector<float> ataps;
        const double twoPi = 2.0 * 3.14159265358979323846;
        auto lp = [&](double cutoff, double tw) {
            int n = sdrpp_design_low_pass(cutoff, tw, v.outSamplerate, 0, nullptr, 0);
            ataps.resize((size_t)n);
            sdrpp_design_low_pass(cutoff, tw, v.outSamplerate, 0, ataps.data(), n);
            d.audio_ntaps = n;
            d.audio_taps = ataps.data();
        };
        const double dbw = v.demodBandwidth > 0.0 ? v.demodBandwidth : v.bandwidth;  // the demodulator's own bandwidth (Demodulator::setBandwidth)
        if (v.demod == Demod::WFM) {
            d.inv_deviation = (float)(1.0 / (twoPi * ((dbw / 2.0) / v.outSamplerate)));
            if (v.lowPass) { lp(15000.0, 4000.0); }
        }
        else if (v.demod == Demod::NFM) {
            d.inv_deviation = (float)(1.0 / (twoPi * ((dbw / 2.0) / v.outSamplerate)));
            if (v.lowPass) { lp(dbw / 2.0, (dbw / 2.0) * 0.1); }
        }
        else if (v.demod == Demod::AM) {
            lp(dbw / 2.0, (dbw / 2.0) * 0.1);
        }
        else if (v.demod == Demod::USB || v.demod == Demod::LSB || v.demod == Demod::DSB) {
            const double tr = v.demod == Demod::USB ? dbw / 2.0 : (v.demod == Demod::LSB ? -dbw / 2.0 : 0.0);
            sdrpp_design_phase_delta(tr, v.outSamplerate, &d.ssb_phase_delta_re, &d.ssb_phase_delta_im);
        }
        int rc = oldId >= 0 ? sdrpp_vfo_replace(ctx, oldId, &d, keep, &v.id) : sdrpp_vfo_add(ctx, &d, &v.id);
        if (rc) { throw std::runtime_error(std::string("[sdrpp_gpu::IQFrontEnd] vfo_add: ") + sdrpp_last_error(ctx)); }
        if (v.afOn && v.demod != Demod::RAW) { applyAF(v); }
    }

    // radio_module.h:98-110: resamp.init(NULL, afRate, audioRate); hpTaps = highPass(300, 100, audioRate); deemp.init(NULL, tau, audioRate)
    void applyAF(RxVFO& v) {
        sdrpp_af_desc a;
        memset(&a, 0, sizeof(a));
        int mode = 0, predec = 1, interp = 1, decim = 1;
        int nt = sdrpp_design_resampler(v.outSamplerate, v.afAudioRate, _plans.maxRatio, &mode, &predec, &interp, &decim, nullptr, 0);
        std::vector<float> rtaps((size_t)(nt > 0 ? nt : 1));
        if (nt > 0) { sdrpp_design_resampler(v.outSamplerate, v.afAudioRate, _plans.maxRatio, &mode, &predec, &interp, &decim, rtaps.data(), nt); }
        const std::vector<DecimStage>* st = nullptr;
        if ((mode == 0 || mode == 1) && predec > 1) {
            auto it = _plans.plans.find(predec);
            if (it == _plans.plans.end()) { throw std::runtime_error("[sdrpp_gpu::IQFrontEnd] no decimation plan for ratio " + std::to_string(predec)); }
            st = &it->second;
        }
        a.n_stages = st ? (int)st->size() : 0;
        for (int i = 0; i < a.n_stages; i++) {
            a.stage_decim[i] = (*st)[(size_t)i].decimation;
            a.stage_ntaps[i] = (int)(*st)[(size_t)i].taps.size();
            a.stage_taps[i] = (*st)[(size_t)i].taps.data();
        }
        a.interp = (mode == 0 || mode == 2) ? interp : 1;
        a.decim = (mode == 0 || mode == 2) ? decim : 1;
        a.resamp_ntaps = (mode == 0 || mode == 2) ? nt : 0;
        a.resamp_taps = rtaps.data();
        std::vector<float> htaps;
        if (v.afHighPass) {
            int n = sdrpp_design_high_pass(300.0, 100.0, v.afAudioRate, 0, nullptr, 0);
            htaps.resize((size_t)n);
            sdrpp_design_high_pass(300.0, 100.0, v.afAudioRate, 0, htaps.data(), n);
            a.hpf_ntaps = n;
            a.hpf_taps = htaps.data();
        }
        a.deemph_alpha = v.afDeempTau > 0.0 ? sdrpp_design_deemphasis_alpha(v.afDeempTau, v.afAudioRate) : 0.0f;
        int rc = sdrpp_vfo_set_af(ctx, v.id, &a);
        if (rc) { throw std::runtime_error(std::string("[sdrpp_gpu::IQFrontEnd] vfo_set_af: ") + sdrpp_last_error(ctx)); }
    }

    dsp::stream<dsp::complex_t>* _in = nullptr;
    sdrpp_ctx* ctx = nullptr;
    DecimPlans _plans;
    std::map<std::string, RxVFO*> vfos;
    double _sampleRate = 0, _fftRate = 20.0;
    int _fftSize = 65536;
    int _decimRatio = 1;
    bool _dcBlocking = false, _invertIQ = false;
    std::atomic<bool> _buffering{ false };
    std::atomic<bool> _pipelining{ false };
    bool _lastStopTimedOut = false;         // doStop: the previous stop ran into its grace period (a sink is not reading)
    int _spinUs = 100;                      // setSpinWait
    dsp::stream<dsp::complex_t>* spinStream = nullptr;
    const void* spinLast = nullptr;         // the input stream's readBuf at the last read(): swap() exchanges it (worker)
    int _pipeLag = 8;
    int _pipeGroup = 1;                     // setPipelining: blocks one launch may carry
    int pipeGroupOn = 1;                    // ... as the context has it now (worker)
    bool pipeOn = false;                    // the context is in pipelined mode (owned by the worker)
    std::atomic<uint64_t> _blocksTaken{ 0 };
    std::atomic<bool> workerDone{ true };   // the worker thread has left its loop (doStop's grace period)
    int _stopGraceMs = 250;                 // how long doStop lets the block in hand and the open hand-over finish before it stops the writers
    std::vector<uint64_t> pendingTickets;   // blocks launched whose results have not been handed out yet
    static constexpr int kStageWords = SDRPP_GROUP_MAX + 2;
    std::atomic<uint32_t> stageLeftRing[kStageWords] = {}, stagePendingRing[kStageWords] = {};  // staging copy of a block: parts not yet copied / not yet accounted for
    int stageWord = 0;
    int heldBlocks = 0;                     // blocks the library holds for a launch group (worker)
    double arriveEmaUs = 0.0;               // how long the last blocks took to arrive after the flush of the one before (worker; only arrivals inside the spin window count)
    int _kickUs = [] { const char* e = getenv("SDRPP_GPU_KICK_US"); const int v = e ? atoi(e) : 20; return v < 0 ? 0 : v; }();  // a held group goes out after this long without a new block
    uint64_t inflightTicket = 0;            // the block whose hand-over is running on the helpers (its result slot is held)
    sdrpp_result inflight{};
    std::vector<std::pair<RxVFO*, int>> inflightOrder;
    std::atomic<bool> deliveryFailed{ false };
    // SampleFrameBuffer state (frame_buffer.h:100-134)
    dsp::complex_t* frames[FRAME_SLOTS] = {};
    int frameSizes[FRAME_SLOTS] = {};
    int frameWrite = 0, frameRead = 0;
    std::vector<dsp::complex_t> tapCopy;  // the block the bound IQ consumers are about to receive
    std::deque<std::pair<uint64_t, std::vector<dsp::complex_t>>> tapPending;  // pipelined: (ticket, input block) kept for the bound streams until the block's turn
    std::vector<dsp::complex_t> inflightTap;
    int pipeFlags = 0;                      // result flags the context was put into pipelined mode with
    bool _keepDevLine = false;              // keepDeviceLine
    float* devLine[2] = { nullptr, nullptr };  // (devLineMtx)
    int devLineSize = 0;                       // floats per buffer = the FFT size of the line kept
    int devLineCur = -1;                       // the complete buffer, -1: none
    std::mutex devLineMtx;
    int _device = 0;
    std::mutex ctlMtx;
    std::vector<std::function<void()>> ctlOps;
    std::mutex frameMtx;
    std::condition_variable frameCnd;
    std::thread frameThread;
    bool stopFrameWorker = false;
    // bound IQ consumers (Splitter::streams) and the staging of the batched VFO read
    std::vector<dsp::stream<dsp::complex_t>*> iqStreams;
    std::vector<int> ids, which, cnts;
    std::vector<int64_t> offs;
    // page-locked staging of the two device-to-host copies of a pass (all VFO blocks; all new FFT lines)
    float* gatherPin = nullptr;
    size_t gatherCap = 0;  // floats
    float* linesPin = nullptr;
    size_t linesCap = 0;   // floats
    // helpers of deliver(): the hand-offs of a pass (one memcpy + swap per VFO stream, one acquire / memcpy / release per FFT line) are
    // host copies of several MB at the frame buffer's batch sizes; they run side by side
    struct Helpers {
        // A batch of jobs is claimed job by job through ONE atomic word (total << 32 | next) by the helpers and by the thread that waits
        // for the batch.  A helper that runs out of work keeps looking for ~100 us before it goes to sleep on the condition variable: at
        // hundreds of thousands of blocks per second a block's host budget is tens of microseconds, and a futex wake-up alone is 5-20 us
        // (measured: with sleeping helpers the pipelined bypass path was SLOWER with the copies spread over four threads than with one).
        std::vector<std::thread> th;
        std::mutex m;
        std::condition_variable cv;
        std::vector<std::function<void()>> jobs;
        std::atomic<uint64_t> word{ 0 };   // (total << 32) | next
        std::atomic<uint32_t> finished{ 0 };
        std::atomic<int> sleepers{ 0 };
        std::atomic<bool> quit{ false };
        static void relax() {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        }
        bool claim_and_run() {
            uint64_t w = word.load(std::memory_order_acquire);
            while ((uint32_t)w < (uint32_t)(w >> 32)) {
                if (word.compare_exchange_weak(w, w + 1, std::memory_order_acq_rel)) {
                    jobs[(size_t)(uint32_t)w]();
                    finished.fetch_add(1, std::memory_order_release);
                    return true;
                }
            }
            return false;
        }
        void start(int n) {
            quit = false;
            for (int i = 0; i < n; i++) {
                th.emplace_back([this]() {
                    auto idle_since = std::chrono::steady_clock::now();
                    while (!quit.load(std::memory_order_relaxed)) {
                        if (claim_and_run()) {
                            idle_since = std::chrono::steady_clock::now();
                            continue;
                        }
                        if (std::chrono::steady_clock::now() - idle_since < std::chrono::microseconds(100)) {
                            relax();
                            continue;
                        }
                        std::unique_lock<std::mutex> lck(m);
                        sleepers.fetch_add(1, std::memory_order_seq_cst);
                        // (bounded wait: begin()'s store of the word and its look at `sleepers` against this thread's increment and its look at
                        // the word are a store -> load pair on either side; both are sequentially consistent, the time-out is the belt to that brace)
                        cv.wait_for(lck, std::chrono::milliseconds(2), [this]() {
                            const uint64_t w = word.load(std::memory_order_seq_cst);
                            return quit.load() || (uint32_t)w < (uint32_t)(w >> 32);
                        });
                        sleepers.fetch_sub(1, std::memory_order_seq_cst);
                        idle_since = std::chrono::steady_clock::now();
                    }
                });
            }
        }
        void stop() {
            {
                std::lock_guard<std::mutex> lck(m);
                quit = true;
            }
            cv.notify_all();
            for (auto& t : th) { if (t.joinable()) { t.join(); } }
            th.clear();
        }
        // hand the jobs to the helpers and return; finish() joins in and waits for the batch (one batch at a time)
        void begin(std::vector<std::function<void()>>&& js) {
            jobs = std::move(js);  // (no batch is open: every job of the previous one has been claimed AND finished, see finish())
            finished.store(0, std::memory_order_relaxed);
            word.store((uint64_t)jobs.size() << 32, std::memory_order_seq_cst);
            if (sleepers.load(std::memory_order_seq_cst) > 0) {
                std::lock_guard<std::mutex> lck(m);
                cv.notify_all();
            }
        }
        // a batch is open and not all of its jobs have finished (any thread may ask)
        bool busy() const {
            const uint32_t total = (uint32_t)(word.load(std::memory_order_acquire) >> 32);
            return total != 0 && finished.load(std::memory_order_acquire) < total;
        }
        void finish() {
            while (claim_and_run()) {}
            const uint32_t total = (uint32_t)(word.load(std::memory_order_acquire) >> 32);
            int spins = 0;
            while (finished.load(std::memory_order_acquire) < total) {
                if (++spins < 2000) { relax(); }
                else { std::this_thread::yield(); }
            }
            word.store(0, std::memory_order_release);
            jobs.clear();
        }
        // run the jobs on the helpers AND the calling thread; returns when all are done
        void run(std::vector<std::function<void()>>&& js) {
            begin(std::move(js));
            finish();
        }
    } helpers, stagers;  // (stagers: the staging copy of the pipelined bypass path, which must not queue behind a hand-over)
    FFTWindow _fftWindow = NUTTALL;
    float* (*_acquire)(void*) = nullptr;
    void (*_release)(void*) = nullptr;
    void* _fftCtx = nullptr;
};

inline void RxVFO::setReferenceRotator(bool enabled) {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    fe->tempStop();
    ncoMode = enabled ? 2 : 1;
    fe->rebuild(*this);
    fe->tempStart();
}
inline void RxVFO::setInSamplerate(double sr) {  // rx_vfo.h:38-43: xlator offset and resampler follow the new input rate
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    fe->tempStop();
    inSamplerate = sr;
    fe->rebuild(*this, 3);  // (the translation's phase, the channel filter's delay line and the demodulator live on; decimators and resampler are new)
    fe->tempStart();
}
inline void RxVFO::setDemodBandwidth(double bw) {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    demodBandwidth = bw;
    if (demod == Demod::RAW) { return; }
    if (demod == Demod::USB || demod == Demod::LSB) {  // ssb.h:44-50, 106-117: only the second translation follows the bandwidth — no restart
        float re, im;
        sdrpp_design_phase_delta(demod == Demod::USB ? bw / 2.0 : -bw / 2.0, outSamplerate, &re, &im);
        IQFrontEnd* f = fe;
        RxVFO* self = this;
        fe->control([f, self, re, im]() { sdrpp_vfo_set_ssb_phase_delta(f->ctx, self->id, re, im); });
        return;
    }
    if (demod == Demod::DSB) { return; }
    fe->tempStop();
    fe->rebuild(*this, 1);  // (the RxVFO in front of the demodulator is not involved)
    fe->tempStart();
}
inline void RxVFO::setOffset(double off) {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    offset = off;
    float re, im;
    sdrpp_design_phase_delta(-offset, inSamplerate, &re, &im);
    IQFrontEnd* f = fe;
    RxVFO* self = this;
    fe->control([f, self, re, im]() { sdrpp_vfo_set_phase_delta(f->ctx, self->id, re, im); });  // phase stays continuous, no restart (rx_vfo.h:72-77)
}
inline void RxVFO::setBandwidth(double bw) {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    bandwidth = bw;
    fe->tempStop();
    if (bandwidth != outSamplerate) {
        const double fw = bandwidth / 2.0;
        int n = sdrpp_design_low_pass(fw, fw * 0.1, outSamplerate, 0, nullptr, 0);
        std::vector<float> t((size_t)n);
        sdrpp_design_low_pass(fw, fw * 0.1, outSamplerate, 0, t.data(), n);
        sdrpp_vfo_set_channel_taps(fe->ctx, id, t.data(), n);
    }
    else {
        sdrpp_vfo_set_channel_taps(fe->ctx, id, nullptr, 0);
    }
    fe->tempStart();
}
inline void RxVFO::setOutSamplerate(double sr, double bw) {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    fe->tempStop();
    outSamplerate = sr;
    bandwidth = bw;
    // (rx_vfo.h:45-58: phase and the channel filter's delay line stay; the demodulator is told its new rate by whoever owns it — the radio module
    // creates a new one on a switch, and one that stays is re-initialised by its own setters — so it starts from cleared state here)
    fe->rebuild(*this, 1);
    fe->tempStart();
}
inline void RxVFO::reset() {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    fe->tempStop();
    sdrpp_vfo_reset(fe->ctx, id);
    fe->tempStart();
}
inline void RxVFO::attachDemod(Demod mode, bool lp, double att, double dec, bool carrier) {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    fe->tempStop();
    demod = mode;
    lowPass = lp;
    agcAttack = att;
    agcDecay = dec;
    carrierAgc = carrier;
    fe->rebuild(*this, 1);  // a new demodulator object behind the SAME RxVFO (radio_module.h:419-563)
    fe->tempStart();
}

inline void RxVFO::attachAF(double audioSamplerate, double deempTau, bool highPass) {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    fe->tempStop();  // radio_module.h:589 afChain.stop() ... :600 afChain.start()
    afOn = true;
    afAudioRate = audioSamplerate;
    afDeempTau = deempTau;
    afHighPass = highPass;
    if (demod != Demod::RAW && id >= 0) { fe->applyAF(*this); }
    fe->tempStart();
}
inline void RxVFO::detachAF() {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    fe->tempStop();
    afOn = false;
    if (id >= 0) { sdrpp_vfo_set_af(fe->ctx, id, nullptr); }
    fe->tempStart();
}

}  // namespace sdrpp_gpu
