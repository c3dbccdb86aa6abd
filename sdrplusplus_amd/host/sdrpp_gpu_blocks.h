// Host-side C++ mirror of the reference's plugin/operator surface for the hot path, on top of the C-ABI
// (include/sdrpp_gpu.h).  Header-only, like the reference's DSP library.
//
// Two ways to use it:
//   * inside an SDR++ tree: compile with -DSDRPP_GPU_USE_SDRPP_DSP and SDR++'s core/src on the include path — the blocks
//     then derive from the real dsp::block and speak the real dsp::stream<T> (core/src/dsp/block.h, stream.h);
//   * stand-alone (tests, other hosts): the minimal `dsp::stream<T>` / `dsp::block` below reproduce the reference's
//     contract — double-buffered swap()/read()/flush(), stopWriter/stopReader, one worker thread per block running
//     `while (run() >= 0)`, tempStop/tempStart around reconfiguration (block.h:46-94, stream.h:43-116).
//
// Blocks:
//   sdrpp_gpu::IQFrontEnd  — same public API as the reference's IQFrontEnd (core/src/signal_path/iq_frontend.h:12-49):
//       init(in, sampleRate, buffering, decimRatio, dcBlocking, fftSize, fftRate, fftWindow, acquireFFTBuffer,
//       releaseFFTBuffer, fftCtx), setFFTSize/Rate/Window, setSampleRate, addVFO/removeVFO, start/stop,
//       getEffectiveSamplerate.  One worker thread replaces inBuf + Splitter + Reshaper + Handler + every RxVFO thread:
//       it reads a block from the input stream, hands it to sdrpp_push (one H2D copy, all kernels), then delivers the
//       finished dB lines through the same acquire/release callback pair (iq_frontend.cpp:258-266) and swaps every VFO's
//       output stream.
//   sdrpp_gpu::RxVFO — what addVFO returns: public `out` stream + setOffset / setBandwidth / setOutSamplerate / reset with
//       the reference's meaning (core/src/dsp/channel/rx_vfo.h:38-87).  With a demodulator attached (attachDemod) the
//       demodulated audio is delivered on `audio` (dsp::stream<stereo_t>), which is what radio's Demodulator::getOutput()
//       returns (decoder_modules/radio/src/demod.h:60).
#pragma once
#include <atomic>
#include <cassert>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdrpp_gpu.h"

#ifdef SDRPP_GPU_USE_SDRPP_DSP
#include <dsp/block.h>
#include <dsp/stream.h>
#include <dsp/types.h>
#include <dsp/multirate/decim/plans.h>
#define SDRPP_GPU_STREAM_BUFFER_SIZE STREAM_BUFFER_SIZE
#else
#define SDRPP_GPU_STREAM_BUFFER_SIZE 1000000  // STREAM_BUFFER_SIZE, core/src/dsp/stream.h:9
namespace dsp {
struct complex_t { float re, im; };
struct stereo_t { float l, r; };

class untyped_stream {
public:
    virtual ~untyped_stream() {}
    virtual void stopWriter() {}
    virtual void clearWriteStop() {}
    virtual void stopReader() {}
    virtual void clearReadStop() {}
};

// Same hand-off protocol as the reference's stream<T>: the producer fills writeBuf and calls swap(n) (blocks until the
// consumer has flushed the previous block; false when stopped); the consumer calls read() (-1 when stopped), uses
// readBuf[0..n) and calls flush().  The two pointers are exchanged on every swap.
template <class T>
class stream : public untyped_stream {
public:
    stream() {
        writeBuf = (T*)aligned_alloc(64, sizeof(T) * SDRPP_GPU_STREAM_BUFFER_SIZE);
        readBuf = (T*)aligned_alloc(64, sizeof(T) * SDRPP_GPU_STREAM_BUFFER_SIZE);
    }
    ~stream() override { free(writeBuf); free(readBuf); }
    bool swap(int size) {
        {
            std::unique_lock<std::mutex> lck(swapMtx);
            swapCV.wait(lck, [this] { return canSwap || writerStop; });
            if (writerStop) { return false; }
            dataSize = size;
            std::swap(writeBuf, readBuf);
            canSwap = false;
        }
        { std::lock_guard<std::mutex> lck(rdyMtx); dataReady = true; }
        rdyCV.notify_all();
        return true;
    }
    int read() {
        std::unique_lock<std::mutex> lck(rdyMtx);
        rdyCV.wait(lck, [this] { return dataReady || readerStop; });
        return readerStop ? -1 : dataSize;
    }
    void flush() {
        { std::lock_guard<std::mutex> lck(rdyMtx); dataReady = false; }
        { std::lock_guard<std::mutex> lck(swapMtx); canSwap = true; }
        swapCV.notify_all();
    }
    void stopWriter() override { { std::lock_guard<std::mutex> lck(swapMtx); writerStop = true; } swapCV.notify_all(); }
    void clearWriteStop() override { writerStop = false; }
    void stopReader() override { { std::lock_guard<std::mutex> lck(rdyMtx); readerStop = true; } rdyCV.notify_all(); }
    void clearReadStop() override { readerStop = false; }
    T* writeBuf;
    T* readBuf;
private:
    std::mutex swapMtx, rdyMtx;
    std::condition_variable swapCV, rdyCV;
    bool canSwap = true, dataReady = false, readerStop = false, writerStop = false;
    int dataSize = 0;
};

class block {
public:
    virtual ~block() {}
    virtual void start() {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        if (running) { return; }
        running = true;
        doStart();
    }
    virtual void stop() {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        if (!running) { return; }
        doStop();
        running = false;
    }
    void tempStart() {
        if (!tempStopDepth || --tempStopDepth) { return; }
        if (tempStopped) { doStart(); tempStopped = false; }
    }
    void tempStop() {
        if (tempStopDepth++) { return; }
        if (running && !tempStopped) { doStop(); tempStopped = true; }
    }
    virtual int run() = 0;
protected:
    void doStart() { workerThread = std::thread([this] { while (run() >= 0) {} }); }
    void doStop() {
        for (auto* in : inputs) { in->stopReader(); }
        for (auto* out : outputs) { out->stopWriter(); }
        if (workerThread.joinable()) { workerThread.join(); }
        for (auto* in : inputs) { in->clearReadStop(); }
        for (auto* out : outputs) { out->clearWriteStop(); }
    }
    void registerInput(untyped_stream* s) { inputs.push_back(s); }
    void registerOutput(untyped_stream* s) { outputs.push_back(s); }
    void unregisterOutput(untyped_stream* s) {
        for (size_t i = 0; i < outputs.size(); i++) { if (outputs[i] == s) { outputs.erase(outputs.begin() + (long)i); break; } }
    }
    std::recursive_mutex ctrlMtx;
    std::vector<untyped_stream*> inputs, outputs;
    bool running = false, tempStopped = false;
    int tempStopDepth = 0;
    std::thread workerThread;
};
}  // namespace dsp
#endif

namespace sdrpp_gpu {

// The reference's power-of-two decimation plans (dsp/multirate/decim/plans.h).  Inside an SDR++ tree they come straight
// from its headers; stand-alone they are read from the numeric fixture sdrplusplus_amd/data/decim_plans.bin.
struct DecimStage { int decimation; std::vector<float> taps; };
class DecimPlans {
public:
    std::map<int, std::vector<DecimStage>> plans;
    int maxRatio = 1;
#ifdef SDRPP_GPU_USE_SDRPP_DSP
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

    void setOffset(double offset);                            // rx_vfo.h:72-77
    void setBandwidth(double bandwidth);                      // rx_vfo.h:60-70
    void setOutSamplerate(double outSamplerate, double bandwidth);  // rx_vfo.h:45-58
    void reset();                                             // rx_vfo.h:79-87
    // Radio-module demodulator fused behind this VFO (decoder_modules/radio/src/demodulators/*.h defaults).
    void attachDemod(Demod mode, bool lowPass = true, double agcAttack = 50.0, double agcDecay = 5.0, bool carrierAgc = false);
    // The radio module's AF chain behind the demodulator (radio_module.h:98-110, 540-547): RationalResampler<stereo_t> to
    // `audioSamplerate`, optional highPass(300, 100) FIR, optional Deemphasis(tau) (tau <= 0: off).  With the chain attached `audio`
    // carries its output (what afChain.out hands to the sink stream); afRate = the demodulator's getAFSampleRate() (= IF rate for
    // every analog demodulator of the radio module).  detachAF() puts the demodulator output back on `audio`.
    void attachAF(double audioSamplerate = 48000.0, double deempTau = 50e-6, bool highPass = false);
    void detachAF();

    double inSamplerate = 0, outSamplerate = 0, bandwidth = 0, offset = 0;
    Demod demod = Demod::RAW;
    bool lowPass = true, carrierAgc = false;
    double agcAttack = 50.0, agcDecay = 5.0;
    bool afOn = false, afHighPass = false;
    double afAudioRate = 48000.0, afDeempTau = 50e-6;

private:
    friend class IQFrontEnd;
    IQFrontEnd* fe = nullptr;
    int id = -1;
    std::string name;
};

class IQFrontEnd : public dsp::block {
public:
    enum FFTWindow { RECTANGULAR, BLACKMAN, NUTTALL };  // iq_frontend.h:18-22

    ~IQFrontEnd() override {
        if (_init) { stop(); }
        for (auto& kv : vfos) { delete kv.second; }
        if (ctx) { sdrpp_destroy(ctx); }
    }

    // iq_frontend.h:23 — decimRatio / dcBlocking configure the pre-processing chain (PowerDecimator -> DCBlocker -> Conjugate,
    // iq_frontend.cpp:32-39), which runs on the device in front of the FFT branch and the VFO bank; `buffering` (the
    // SampleFrameBuffer in front of it) is a host-side hand-off and has no device counterpart.
    void init(dsp::stream<dsp::complex_t>* in, double sampleRate, bool buffering, int decimRatio, bool dcBlocking, int fftSize, double fftRate,
              FFTWindow fftWindow, float* (*acquireFFTBuffer)(void* ctx), void (*releaseFFTBuffer)(void* ctx), void* fftCtx, int device = 0,
              const DecimPlans* plans = nullptr) {
        (void)buffering;
        _in = in;
        _sampleRate = sampleRate;
        _decimRatio = decimRatio;
        _dcBlocking = dcBlocking;
        _fftSize = fftSize;
        _fftRate = fftRate;
        _fftWindow = fftWindow;
        _acquire = acquireFFTBuffer;
        _release = releaseFFTBuffer;
        _fftCtx = fftCtx;
        if (plans) { _plans = *plans; }
        int rc = sdrpp_create(device, SDRPP_GPU_MAX_BLOCK, &ctx);
        if (rc) { throw std::runtime_error(std::string("[sdrpp_gpu::IQFrontEnd] ") + sdrpp_strerror(rc)); }
        registerInput(_in);
        updatePreproc();
        updateFFTPath();
        _init = true;
    }

    // iq_frontend.cpp:105-130: the effective sample rate changes with the decimation; every VFO and the FFT framing follow it
    void setDecimation(int ratio) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        tempStop();
        _decimRatio = ratio;
        updatePreproc();
        for (auto& kv : vfos) {
            kv.second->inSamplerate = getEffectiveSamplerate();
            rebuild(*kv.second);
        }
        updateFFTPath();
        tempStart();
    }
    void setDCBlocking(bool enabled) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _dcBlocking = enabled; updatePreproc(); tempStart(); }
    void setInvertIQ(bool enabled) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _invertIQ = enabled; updatePreproc(); tempStart(); }

    void setSampleRate(double sampleRate) {
        std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
        tempStop();
        _sampleRate = sampleRate;
        updatePreproc();  // the DC blocker's rate follows the effective sample rate (iq_frontend.cpp:85-86)
        for (auto& kv : vfos) {
            kv.second->inSamplerate = getEffectiveSamplerate();
            rebuild(*kv.second);
        }
        updateFFTPath();
        tempStart();
    }
    double getSampleRate() { return _sampleRate; }
    double getEffectiveSamplerate() { return _sampleRate / (double)_decimRatio; }
    void setFFTSize(int size) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _fftSize = size; updateFFTPath(); tempStart(); }
    void setFFTRate(double rate) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _fftRate = rate; updateFFTPath(); tempStart(); }
    void setFFTWindow(FFTWindow w) { std::lock_guard<std::recursive_mutex> lck(ctrlMtx); tempStop(); _fftWindow = w; updateFFTPath(); tempStart(); }

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

    // One block in -> FFT lines through acquire/release, one block out on every VFO stream.
    int run() override {
        int count = _in->read();
        if (count < 0) { return -1; }
        int rc = sdrpp_push(ctx, (const float*)_in->readBuf, count);
        _in->flush();
        if (rc) {
            fprintf(stderr, "[sdrpp_gpu::IQFrontEnd] push failed: %s\n", sdrpp_last_error(ctx));
            return -1;
        }
        const int nlines = sdrpp_fft_lines(ctx);
        for (int i = 0; i < nlines; i++) {
            float* buf = _acquire ? _acquire(_fftCtx) : nullptr;  // may be NULL: still paired with release (iq_frontend.cpp:258-266)
            if (buf) { sdrpp_fft_read(ctx, i, 1, buf, nullptr, nullptr); }
            if (_release) { _release(_fftCtx); }
        }
        for (auto& kv : vfos) {
            RxVFO* v = kv.second;
            if (v->demod == Demod::RAW) {
                int n = sdrpp_vfo_read(ctx, v->id, (float*)v->out.writeBuf, SDRPP_GPU_STREAM_BUFFER_SIZE);
                if (n > 0 && !v->out.swap(n)) { return -1; }
            }
            else {
                int n = v->afOn ? sdrpp_vfo_af_read(ctx, v->id, (float*)v->audio.writeBuf, SDRPP_GPU_STREAM_BUFFER_SIZE)
                                : sdrpp_vfo_read(ctx, v->id, (float*)v->audio.writeBuf, SDRPP_GPU_STREAM_BUFFER_SIZE);
                if (n > 0 && !v->audio.swap(n)) { return -1; }
            }
        }
        return count;
    }

    sdrpp_ctx* context() { return ctx; }
    static constexpr int64_t SDRPP_GPU_MAX_BLOCK = 1000000;

private:
    friend class RxVFO;

    void updatePreproc() {  // iq_frontend.cpp:32-39: decim enabled for ratio > 1, dcBlock rate genDCBlockRate(effectiveSr), conjugate
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
        int rc = sdrpp_preproc_configure(ctx, n, dec, nt, tp, rate, _invertIQ ? 1 : 0);
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
    void rebuild(RxVFO& v) {
        if (v.id >= 0) { sdrpp_vfo_remove(ctx, v.id); v.id = -1; }
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
        if (v.demod == Demod::WFM) {
            d.inv_deviation = (float)(1.0 / (twoPi * ((v.bandwidth / 2.0) / v.outSamplerate)));
            if (v.lowPass) { lp(15000.0, 4000.0); }
        }
        else if (v.demod == Demod::NFM) {
            d.inv_deviation = (float)(1.0 / (twoPi * ((v.bandwidth / 2.0) / v.outSamplerate)));
            if (v.lowPass) { lp(v.bandwidth / 2.0, (v.bandwidth / 2.0) * 0.1); }
        }
        else if (v.demod == Demod::AM) {
            lp(v.bandwidth / 2.0, (v.bandwidth / 2.0) * 0.1);
        }
        else if (v.demod == Demod::USB || v.demod == Demod::LSB || v.demod == Demod::DSB) {
            const double tr = v.demod == Demod::USB ? v.bandwidth / 2.0 : (v.demod == Demod::LSB ? -v.bandwidth / 2.0 : 0.0);
            sdrpp_design_phase_delta(tr, v.outSamplerate, &d.ssb_phase_delta_re, &d.ssb_phase_delta_im);
        }
        int rc = sdrpp_vfo_add(ctx, &d, &v.id);
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
    FFTWindow _fftWindow = NUTTALL;
    float* (*_acquire)(void*) = nullptr;
    void (*_release)(void*) = nullptr;
    void* _fftCtx = nullptr;
    bool _init = false;
};

inline void RxVFO::setOffset(double off) {
    std::lock_guard<std::recursive_mutex> lck(fe->ctrlMtx);
    offset = off;
    float re, im;
    sdrpp_design_phase_delta(-offset, inSamplerate, &re, &im);
    sdrpp_vfo_set_phase_delta(fe->ctx, id, re, im);  // phase stays continuous, no restart (rx_vfo.h:72-77)
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
    fe->rebuild(*this);
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
    fe->rebuild(*this);
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
