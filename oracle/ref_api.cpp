// ORACLE / TEST INFRASTRUCTURE ONLY — builds oracle/_ref/libsdrpp_ref*.so (git-ignored, shipped to the GPU box).
//
// Thin extern "C" wrapper around the REFERENCE'S OWN code, compiled where it lies under /root/reference:
//   * core/src/dsp/**            header-only DSP library (RxVFO, RationalResampler, FIR, demodulators, ...)
//   * core/src/signal_path/iq_frontend.cpp   included verbatim (FFT branch + Reshaper + Splitter + threads)
// against the restated third-party arithmetic in oracle/shim (volk/volk.h, fftw3.h) and three stub headers
// (gui/gui.h, core.h, utils/flog.h).  No reference source is copied into this repository.
//
// The wrapper exists so that tests can (a) validate oracle/oracle.c bit-for-bit against the reference's classes and
// (b) generate tests/golden/*.npz, and so that bench.py can time the reference's process() code as the CPU baseline
// (cpu_baseline.kind = "reference").
#include <cstring>
#include <cstdio>
#include <thread>
#include <vector>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cmath>
#include <string>
#include <algorithm>

// RationalResampler::reconfigure (rational_resampler.h:162) printf()s a status line on every retune; inside a test / benchmark
// process that stdout noise lands after bench.py's JSON line.  The reference is compiled unmodified, so the call is routed to a
// no-op here (all standard headers that declare printf are included above this point).
static inline int sdrpp_ref_quiet_printf(const char*, ...) { return 0; }
#define printf(...) sdrpp_ref_quiet_printf(__VA_ARGS__)

#include <dsp/channel/rx_vfo.h>
#include <dsp/demod/broadcast_fm.h>
#include <dsp/demod/fm.h>
#include <dsp/demod/am.h>
#include <dsp/demod/ssb.h>
#include <dsp/filter/deephasis.h>
#include <dsp/multirate/rational_resampler.h>
#include <dsp/multirate/power_decimator.h>
#include <dsp/correction/dc_blocker.h>
#include <dsp/math/conjugate.h>
#include <dsp/compression/sample_stream_compressor.h>
#include <dsp/taps/low_pass.h>
#include <dsp/taps/high_pass.h>
#include <dsp/window/nuttall.h>
#include <dsp/window/blackman.h>

// The reference translation unit itself (its relative includes resolve next to it).
#include <signal_path/iq_frontend.cpp>

using dsp::complex_t;
using dsp::stereo_t;

#undef printf

extern "C" {

// ---- host maths ------------------------------------------------------------------------------------------------------
int ref_low_pass(double cutoff, double transWidth, double sampleRate, int odd, float* out, int max) {
    dsp::tap<float> t = dsp::taps::lowPass(cutoff, transWidth, sampleRate, odd != 0);
    int n = t.size;
    memcpy(out, t.taps, sizeof(float) * (size_t)(n < max ? n : max));
    dsp::taps::free(t);
    return n;
}
int ref_high_pass(double cutoff, double transWidth, double sampleRate, int odd, float* out, int max) {
    dsp::tap<float> t = dsp::taps::highPass(cutoff, transWidth, sampleRate, odd != 0);
    int n = t.size;
    memcpy(out, t.taps, sizeof(float) * (size_t)(n < max ? n : max));
    dsp::taps::free(t);
    return n;
}
double ref_nuttall(double n, double N) { return dsp::window::nuttall(n, N); }
double ref_blackman(double n, double N) { return dsp::window::blackman(n, N); }

// ---- RxVFO -------------------------------------------------------------------------------------------------------------
struct RefRxVFO {
    dsp::channel::RxVFO vfo;
    complex_t* work;
};
void* ref_rxvfo_create(double inSR, double outSR, double bandwidth, double offset) {
    RefRxVFO* r = new RefRxVFO;
    r->vfo.init(NULL, inSR, outSR, bandwidth, offset);
    r->work = dsp::buffer::alloc<complex_t>(STREAM_BUFFER_SIZE);
    return r;
}
void ref_rxvfo_destroy(void* h) {
    RefRxVFO* r = (RefRxVFO*)h;
    dsp::buffer::free(r->work);
    // RxVFO's destructor asserts/joins only if initialised with a stream thread; never started here.
    delete r;
}
void ref_rxvfo_set_offset(void* h, double offset) { ((RefRxVFO*)h)->vfo.setOffset(offset); }
void ref_rxvfo_set_bandwidth(void* h, double bandwidth) { ((RefRxVFO*)h)->vfo.setBandwidth(bandwidth); }  // rx_vfo.h:60-70 (FIR::setTaps keeps the delay line)
void ref_rxvfo_set_in_samplerate(void* h, double sr) { ((RefRxVFO*)h)->vfo.setInSamplerate(sr); }                              // rx_vfo.h:35-43
void ref_rxvfo_set_out_samplerate(void* h, double sr, double bandwidth) { ((RefRxVFO*)h)->vfo.setOutSamplerate(sr, bandwidth); }  // rx_vfo.h:45-58
// count <= 1 000 000 (STREAM_BUFFER_SIZE, dsp/stream.h:9); out must hold `count` complex samples.
int ref_rxvfo_process(void* h, int count, const float* in, float* out) {
    RefRxVFO* r = (RefRxVFO*)h;
    return r->vfo.process(count, (const complex_t*)in, (complex_t*)out);
}

// ---- demodulators ---------------------------------------------------------------------------------------------------------
enum { REF_WFM = 0, REF_NFM = 1, REF_AM = 2, REF_USB = 3, REF_LSB = 4, REF_DSB = 5 };
struct RefDemod {
    int mode;
    dsp::demod::BroadcastFM* wfm = NULL;
    dsp::demod::FM<stereo_t>* nfm = NULL;
    dsp::demod::AM<stereo_t>* am = NULL;
    dsp::demod::SSB<stereo_t>* ssb = NULL;
};
void* ref_demod_create(int mode, double bandwidth, double ifSR, int lowPass, double agcAttack, double agcDecay, int carrierAgc) {
    RefDemod* d = new RefDemod;
    d->mode = mode;
    if (mode == REF_WFM) {
        d->wfm = new dsp::demod::BroadcastFM;
        d->wfm->init(NULL, bandwidth / 2.0f, ifSR, false, lowPass != 0, false); // wfm.h:78 with stereo/RDS off (wfm.h:363-365)
    }
    else if (mode == REF_NFM) {
        d->nfm = new dsp::demod::FM<stereo_t>;
        d->nfm->init(NULL, ifSR, bandwidth, lowPass != 0); // nfm.h:29
    }
    else if (mode == REF_AM) {
        d->am = new dsp::demod::AM<stereo_t>;
        d->am->init(NULL, carrierAgc ? dsp::demod::AM<stereo_t>::AGCMode::CARRIER : dsp::demod::AM<stereo_t>::AGCMode::AUDIO, bandwidth,
                    agcAttack / ifSR, agcDecay / ifSR, 100.0 / ifSR, ifSR); // radio am.h:34
    }
    else {
        d->ssb = new dsp::demod::SSB<stereo_t>;
        auto m = (mode == REF_USB) ? dsp::demod::SSB<stereo_t>::Mode::USB : ((mode == REF_LSB) ? dsp::demod::SSB<stereo_t>::Mode::LSB : dsp::demod::SSB<stereo_t>::Mode::DSB);
        d->ssb->init(NULL, m, bandwidth, ifSR, agcAttack / ifSR, agcDecay / ifSR); // radio usb.h:34
    }
    return d;
}
void ref_demod_destroy(void* h) {
    RefDemod* d = (RefDemod*)h;
    delete d->wfm; delete d->nfm; delete d->am; delete d->ssb;
    delete d;
}
int ref_demod_process(void* h, int count, const float* in, float* out) {
    RefDemod* d = (RefDemod*)h;
    complex_t* cin = (complex_t*)in; // the reference's process() signatures are non-const
    if (d->wfm) { int rds = 0; return d->wfm->process(count, cin, (stereo_t*)out, rds, NULL); }
    if (d->nfm) { return d->nfm->process(count, cin, (stereo_t*)out); }
    if (d->am) { return d->am->process(count, cin, (stereo_t*)out); }
    return d->ssb->process(count, cin, (stereo_t*)out);
}

// ---- server wire format: SampleStreamCompressor::process (sample_stream_compressor.h:30-62) ---------------------------------------------
int ref_compress(int count, int pcmType, const float* in, unsigned char* out) {
    return dsp::compression::SampleStreamCompressor::process(count, (dsp::compression::PCMType)pcmType, (const complex_t*)in, (uint8_t*)out);
}

// ---- IQFrontEnd pre-processing chain (iq_frontend.cpp:32-39): PowerDecimator -> DCBlocker -> Conjugate, process() level -------------
struct RefPreproc {
    dsp::multirate::PowerDecimator<complex_t> decim;
    dsp::correction::DCBlocker<complex_t> dcBlock;
    int ratio, dc, conj;
    complex_t* work;
};
void* ref_preproc_create(int ratio, int dcBlocking, double dcRate, int conjugate) {
    RefPreproc* p = new RefPreproc;
    p->decim.init(NULL, ratio);
    p->dcBlock.init(NULL, dcRate);
    p->ratio = ratio;
    p->dc = dcBlocking;
    p->conj = conjugate;
    p->work = dsp::buffer::alloc<complex_t>(STREAM_BUFFER_SIZE);
    return p;
}
void ref_preproc_destroy(void* h) {
    RefPreproc* p = (RefPreproc*)h;
    dsp::buffer::free(p->work);
    delete p;
}
// what IQFrontEnd's setters do to the chain's block objects while the stream runs (iq_frontend.cpp:76-130): setDecimation -> decim.setRatio (new stages)
// only for ratio > 1; setSampleRate -> dcBlock.setRate (the estimate stays); setDCBlocking / setInvertIQ only take a block in or out of the chain
void ref_preproc_set(void* h, int ratio, int dcBlocking, double dcRate, int conjugate, int newDecimator) {
    RefPreproc* p = (RefPreproc*)h;
    p->ratio = ratio;
    if (newDecimator && ratio > 1) { p->decim.setRatio(ratio); }
    p->dcBlock.setRate(dcRate);
    p->dc = dcBlocking;
    p->conj = conjugate;
}
// count <= STREAM_BUFFER_SIZE; the chain enables the decimator only for ratio > 1 (iq_frontend.cpp:37)
int ref_preproc_process(void* h, int count, const float* in, float* out) {
    RefPreproc* p = (RefPreproc*)h;
    int n = count;
    if (p->ratio > 1) { n = p->decim.process(count, (const complex_t*)in, (complex_t*)out); }
    else { memcpy(out, in, sizeof(complex_t) * (size_t)count); }
    if (p->dc) { p->dcBlock.process(n, (complex_t*)out, (complex_t*)out); }
    if (p->conj) { dsp::math::Conjugate::process(n, (complex_t*)out, (complex_t*)out); }
    return n;
}

// ---- RationalResampler<stereo_t/complex_t> and Deemphasis (AF chain, radio_module.h:102-110) ----------------------------------
void* ref_resampler_create(double inSR, double outSR) {
    auto* r = new dsp::multirate::RationalResampler<complex_t>;
    r->init(NULL, inSR, outSR);
    return r;
}
void ref_resampler_destroy(void* h) { delete (dsp::multirate::RationalResampler<complex_t>*)h; }
int ref_resampler_process(void* h, int count, const float* in, float* out) {
    return ((dsp::multirate::RationalResampler<complex_t>*)h)->process(count, (const complex_t*)in, (complex_t*)out);
}
void* ref_deemp_create(double tau, double sr) {
    auto* d = new dsp::filter::Deemphasis<stereo_t>;
    d->init(NULL, tau, sr);
    return d;
}
void ref_deemp_destroy(void* h) { delete (dsp::filter::Deemphasis<stereo_t>*)h; }
void ref_deemp_process(void* h, int count, const float* in, float* out) {
    ((dsp::filter::Deemphasis<stereo_t>*)h)->process(count, (const stereo_t*)in, (stereo_t*)out);
}

// ---- IQFrontEnd: the real threaded graph (inBuf -> preproc -> Splitter -> Reshaper -> Handler -> handler()) --------------------
struct RefFrontEnd {
    IQFrontEnd fe;
    dsp::stream<complex_t> in;
    int fftSize;
    std::mutex mtx;
    std::vector<float> lines; // appended, fftSize floats per line
    std::vector<float> cur;
    std::atomic<int> nlines{ 0 };
};
static float* ref_fe_acquire(void* ctx) {
    RefFrontEnd* f = (RefFrontEnd*)ctx;
    f->cur.assign((size_t)f->fftSize, 0.0f);
    return f->cur.data();
}
static void ref_fe_release(void* ctx) {
    RefFrontEnd* f = (RefFrontEnd*)ctx;
    std::lock_guard<std::mutex> lck(f->mtx);
    f->lines.insert(f->lines.end(), f->cur.begin(), f->cur.end());
    f->nlines++;
}
// window: 0 RECTANGULAR, 1 BLACKMAN, 2 NUTTALL (iq_frontend.h:18-22)
void* ref_frontend_create(double sampleRate, int fftSize, double fftRate, int window) {
    RefFrontEnd* f = new RefFrontEnd;
    f->fftSize = fftSize;
    f->fe.init(&f->in, sampleRate, /*buffering*/ false, /*decim*/ 1, /*dcBlocking*/ false, fftSize, fftRate, (IQFrontEnd::FFTWindow)window,
               ref_fe_acquire, ref_fe_release, f);
    // The application always re-applies the FFT settings after init (display.cpp:86-92); that is what installs the
    // fftshift-ed window.
    f->fe.setFFTWindow((IQFrontEnd::FFTWindow)window);
    f->fe.start();
    return f;
}
// Feeds `count` samples as blocks of `blockSize` (<= 1e6) and waits until `expectLines` lines have been produced.
int ref_frontend_feed(void* h, const float* iq, long long count, int blockSize, int expectLines, int timeoutMs) {
    RefFrontEnd* f = (RefFrontEnd*)h;
    long long done = 0;
    while (done < count) {
        int n = (int)((count - done) < blockSize ? (count - done) : blockSize);
        memcpy(f->in.writeBuf, iq + 2 * done, sizeof(complex_t) * (size_t)n);
        if (!f->in.swap(n)) { return -1; }
        done += n;
    }
    auto t0 = std::chrono::steady_clock::now();
    while (f->nlines.load() < expectLines) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeoutMs) { break; }
    }
    return f->nlines.load();
}
int ref_frontend_lines(void* h, float* out, int maxLines) {
    RefFrontEnd* f = (RefFrontEnd*)h;
    std::lock_guard<std::mutex> lck(f->mtx);
    int n = f->nlines.load();
    if (n > maxLines) { n = maxLines; }
    memcpy(out, f->lines.data(), sizeof(float) * (size_t)n * f->fftSize);
    return n;
}
void ref_frontend_destroy(void* h) {
    RefFrontEnd* f = (RefFrontEnd*)h;
    f->fe.stop();
    delete f;
}

// ---- CPU baseline: the reference's per-VFO process() chain + FFT handler, `nthreads` worker threads ------------------------------
// Workload = BASELINE cfg 3: nVfo x (RxVFO(inSR -> 250 kHz, bw 150 kHz) + BroadcastFM mono) and one windowed FFT + log-power
// every `fftInterval` samples, input handed over in blocks of `blockSize` (= sr/200, file_source/main.cpp:157).  VFOs are dealt
// round-robin to the worker threads (the reference itself runs one thread per block; this partition has less hand-off overhead, i.e.
// it flatters the CPU).  The buffer is streamed `repeat` times (filter state carries on).  Returns seconds of wall time for
// repeat * totalSamples input samples.
double ref_bench_cfg3(const float* iq, long long totalSamples, int blockSize, double inSR, int nVfo, const double* offsets, int fftSize, int nthreads, int repeat) {
    std::vector<dsp::channel::RxVFO*> vfos((size_t)nVfo);
    std::vector<dsp::demod::BroadcastFM*> dem((size_t)nVfo);
    for (int v = 0; v < nVfo; v++) {
        vfos[v] = new dsp::channel::RxVFO;
        vfos[v]->init(NULL, inSR, 250000.0, 150000.0, offsets[v]);
        dem[v] = new dsp::demod::BroadcastFM;
        dem[v]->init(NULL, 75000.0, 250000.0, false, true, false);
    }
    // FFT branch state (what IQFrontEnd::handler does per frame)
    std::vector<float> window((size_t)fftSize);
    for (int i = 0; i < fftSize; i++) { window[i] = dsp::window::nuttall(i, fftSize) * ((i % 2) ? -1.0f : 1.0f); }
    fftwf_complex* fin = (fftwf_complex*)fftwf_malloc(sizeof(fftwf_complex) * (size_t)fftSize);
    fftwf_complex* fout = (fftwf_complex*)fftwf_malloc(sizeof(fftwf_complex) * (size_t)fftSize);
    fftwf_plan plan = fftwf_plan_dft_1d(fftSize, fin, fout, FFTW_FORWARD, FFTW_ESTIMATE);
    std::vector<float> db((size_t)fftSize);

    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) {
        th.emplace_back([&, t]() {
            complex_t* work = dsp::buffer::alloc<complex_t>(STREAM_BUFFER_SIZE);
            stereo_t* audio = dsp::buffer::alloc<stereo_t>(STREAM_BUFFER_SIZE);
            for (int rep = 0; rep < repeat; rep++)
            for (long long pos = 0; pos + blockSize <= totalSamples; pos += blockSize) {
                const complex_t* blk = (const complex_t*)(iq + 2 * pos);
                for (int v = t; v < nVfo; v += nthreads) {
                    int n = vfos[v]->process(blockSize, blk, work);
                    int rds = 0;
                    dem[v]->process(n, work, audio, rds, NULL);
                }
            }
            if (t == 0) {
                // dense framing: every fftSize samples one frame (skip = 0)
                for (int rep = 0; rep < repeat; rep++)
                for (long long pos = 0; pos + fftSize <= totalSamples; pos += fftSize) {
                    volk_32fc_32f_multiply_32fc((lv_32fc_t*)fin, (const lv_32fc_t*)(iq + 2 * pos), window.data(), fftSize);
                    fftwf_execute(plan);
                    volk_32fc_s32f_power_spectrum_32f(db.data(), (lv_32fc_t*)fout, fftSize, fftSize);
                }
            }
            dsp::buffer::free(work);
            dsp::buffer::free(audio);
        });
    }
    for (auto& x : th) { x.join(); }
    auto t1 = std::chrono::steady_clock::now();
    for (int v = 0; v < nVfo; v++) { delete vfos[v]; delete dem[v]; }
    fftwf_destroy_plan(plan);
    fftwf_free(fin);
    fftwf_free(fout);
    return std::chrono::duration<double>(t1 - t0).count();
}

// Any BASELINE configuration: per VFO its own IF rate / bandwidth / offset / radio demodulator (REF_* mode; < 0: none, RxVFO only),
// plus (fftSize > 0) one windowed FFT + log-power per `fftSize` samples.  cfg 2 = nVfo 0; cfg 4 = 128 VFOs NFM / AM / USB at 61.44 MS/s
// with the 2^20-point FFT evaluated directly (the reference's own Reshaper cannot carry a 2^20-sample frame).
// Same threading as ref_bench_cfg3.  Returns seconds of wall time for repeat * totalSamples input samples.
double ref_bench_cfg(const float* iq, long long totalSamples, int blockSize, double inSR, int nVfo, const double* offsets, const double* ifRates,
                     const double* bandwidths, const int* modes, int fftSize, int nthreads, int repeat) {
    std::vector<dsp::channel::RxVFO*> vfos((size_t)nVfo);
    std::vector<void*> dem((size_t)nVfo, (void*)NULL);
    for (int v = 0; v < nVfo; v++) {
        vfos[v] = new dsp::channel::RxVFO;
        vfos[v]->init(NULL, inSR, ifRates[v], bandwidths[v], offsets[v]);
        if (modes[v] >= 0) { dem[v] = ref_demod_create(modes[v], bandwidths[v], ifRates[v], 1, 50.0, 5.0, 0); }
    }
    std::vector<float> window((size_t)(fftSize > 0 ? fftSize : 1));
    fftwf_complex *fin = NULL, *fout = NULL;
    fftwf_plan plan = NULL;
    std::vector<float> db((size_t)(fftSize > 0 ? fftSize : 1));
    if (fftSize > 0) {
        for (int i = 0; i < fftSize; i++) { window[i] = dsp::window::nuttall(i, fftSize) * ((i % 2) ? -1.0f : 1.0f); }
        fin = (fftwf_complex*)fftwf_malloc(sizeof(fftwf_complex) * (size_t)fftSize);
        fout = (fftwf_complex*)fftwf_malloc(sizeof(fftwf_complex) * (size_t)fftSize);
        plan = fftwf_plan_dft_1d(fftSize, fin, fout, FFTW_FORWARD, FFTW_ESTIMATE);
    }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) {
        th.emplace_back([&, t]() {
            complex_t* work = dsp::buffer::alloc<complex_t>(STREAM_BUFFER_SIZE);
            stereo_t* audio = dsp::buffer::alloc<stereo_t>(STREAM_BUFFER_SIZE);
            for (int rep = 0; rep < repeat; rep++)
            for (long long pos = 0; pos + blockSize <= totalSamples; pos += blockSize) {
                const complex_t* blk = (const complex_t*)(iq + 2 * pos);
                for (int v = t; v < nVfo; v += nthreads) {
                    int n = vfos[v]->process(blockSize, blk, work);
                    if (dem[v]) { ref_demod_process(dem[v], n, (const float*)work, (float*)audio); }
                }
            }
            dsp::buffer::free(work);
            dsp::buffer::free(audio);
        });
    }
    if (fftSize > 0) {  // the FFT branch on a thread of its own (the reference runs it on the Handler sink's thread, iq_frontend.cpp:41-44) — never behind a VFO worker's blocks
        th.emplace_back([&]() {
            for (int rep = 0; rep < repeat; rep++)
            for (long long pos = 0; pos + fftSize <= totalSamples; pos += fftSize) {
                volk_32fc_32f_multiply_32fc((lv_32fc_t*)fin, (const lv_32fc_t*)(iq + 2 * pos), window.data(), fftSize);
                fftwf_execute(plan);
                volk_32fc_s32f_power_spectrum_32f(db.data(), (lv_32fc_t*)fout, fftSize, fftSize);
            }
        });
    }
    for (auto& x : th) { x.join(); }
    auto t1 = std::chrono::steady_clock::now();
    for (int v = 0; v < nVfo; v++) {
        delete vfos[v];
        if (dem[v]) { ref_demod_destroy(dem[v]); }
    }
    if (plan) {
        fftwf_destroy_plan(plan);
        fftwf_free(fin);
        fftwf_free(fout);
    }
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- CPU baseline as SURVEY.md 8(d) defines it: the reference's own THREADED graph, timed the way dsp::bench::SpeedTester times a block
//      (speed_tester.h:31-56, 78-92) -------------------------------------------------------------------------------------------------------
// IQFrontEnd (iq_frontend.cpp verbatim: inBuf -> preproc -> Splitter -> Reshaper -> Handler -> handler(), its own threads) + per VFO the RxVFO
// that IQFrontEnd::addVFO creates on the Splitter and starts (one thread), the radio demodulator as a started dsp::block behind RxVFO::out (one
// thread) and a reader thread that read()s / flush()es the audio stream like SpeedTester::readWorker — ~3 threads per VFO + the front end's seven,
// every hand-over a dsp::stream swap, exactly the reference's threading model (block.h:71-73, splitter.h:46-61).  An unthrottled source thread
// memcpy()s fixed blocks into the input stream and swap()s (SpeedTester::writeWorker).  After `warmupS` seconds the samples the source got rid
// of during `runS` seconds are counted — back-pressured by the slowest branch — `nRuns` times; out_rates[i] = samples per second of run i.
// Returns the number of runs measured (0 on failure).  WFM / NFM / AM / SSB demodulators as ref_demod_create initialises them.
int ref_bench_graph(const float* iq, long long totalSamples, int blockSize, double inSR, int nVfo, const double* offsets, const double* ifRates, const double* bandwidths, const int* modes,
                    int fftSize, double fftRate, double warmupS, double runS, int nRuns, double* out_rates, long long* out_audio_frames, long long* out_lines) {
    struct Graph {
        IQFrontEnd fe;
        dsp::stream<complex_t> in;
        std::vector<float> line;
        std::atomic<long long> nlines{ 0 };
    };
    Graph* g = new Graph;
    g->line.assign((size_t)(fftSize > 0 ? fftSize : 1), 0.0f);
    auto acq = [](void* c) -> float* { return ((Graph*)c)->line.data(); };
    auto rel = [](void* c) { ((Graph*)c)->nlines++; };
    g->fe.init(&g->in, inSR, /*buffering*/ false, /*decim*/ 1, /*dcBlocking*/ false, fftSize > 0 ? fftSize : 1024, fftSize > 0 ? fftRate : 1.0, IQFrontEnd::FFTWindow::NUTTALL, acq, rel, g);
    g->fe.setFFTWindow(IQFrontEnd::FFTWindow::NUTTALL);
    std::vector<dsp::channel::RxVFO*> vfos;
    std::vector<RefDemod*> dem;
    std::vector<std::thread> readers;
    std::atomic<long long> audio{ 0 };
    std::vector<dsp::stream<stereo_t>*> outs;
    for (int v = 0; v < nVfo; v++) {
        dsp::channel::RxVFO* x = g->fe.addVFO("v" + std::to_string(v), ifRates[v], bandwidths[v], offsets[v]);  // (starts the RxVFO's thread)
        if (!x) { return 0; }
        vfos.push_back(x);
        RefDemod* d = (RefDemod*)ref_demod_create(modes[v], bandwidths[v], ifRates[v], 1, 50.0, 5.0, 0);
        dsp::stream<stereo_t>* o = NULL;
        if (d->wfm) { d->wfm->setInput(&x->out); d->wfm->start(); o = &d->wfm->out; }
        else if (d->nfm) { d->nfm->setInput(&x->out); d->nfm->start(); o = &d->nfm->out; }
        else if (d->am) { d->am->setInput(&x->out); d->am->start(); o = &d->am->out; }
        else { d->ssb->setInput(&x->out); d->ssb->start(); o = &d->ssb->out; }
        dem.push_back(d);
        outs.push_back(o);
        readers.emplace_back([o, &audio]() {
            while (true) {
                int n = o->read();
                o->flush();
                if (n < 0) { return; }
                audio += n;
            }
        });
    }
    g->fe.start();
    std::atomic<long long> fed{ 0 };
    std::atomic<bool> stop{ false };
    std::thread source([&]() {
        long long pos = 0;
        while (!stop) {
            if (pos + blockSize > totalSamples) { pos = 0; }
            memcpy(g->in.writeBuf, iq + 2 * pos, sizeof(complex_t) * (size_t)blockSize);
            if (!g->in.swap(blockSize)) { return; }
            fed += blockSize;
            pos += blockSize;
        }
    });
    std::this_thread::sleep_for(std::chrono::duration<double>(warmupS));
    int done = 0;
    long long a0 = audio, l0 = g->nlines;
    for (int r = 0; r < nRuns; r++) {
        const long long f0 = fed;
        const auto t0 = std::chrono::steady_clock::now();
        std::this_thread::sleep_for(std::chrono::duration<double>(runS));
        const long long f1 = fed;
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        out_rates[done++] = (double)(f1 - f0) / dt;
    }
    if (out_audio_frames) { *out_audio_frames = audio - a0; }
    if (out_lines) { *out_lines = g->nlines - l0; }
    stop = true;
    g->in.stopWriter();
    source.join();
    g->fe.stop();
    for (size_t v = 0; v < vfos.size(); v++) {
        RefDemod* d = dem[v];
        if (d->wfm) { d->wfm->stop(); }
        if (d->nfm) { d->nfm->stop(); }
        if (d->am) { d->am->stop(); }
        if (d->ssb) { d->ssb->stop(); }
        outs[v]->stopReader();
    }
    for (auto& t : readers) { t.join(); }
    for (size_t v = 0; v < vfos.size(); v++) {
        g->fe.removeVFO("v" + std::to_string(v));
        ref_demod_destroy(dem[v]);
    }
    delete g;
    return done;
}

// single-thread process()-only rate of ONE stage of a cfg-3-style VFO chain, for the per-stage table SURVEY.md 8(d) asks for: stage 0 = the whole
// RxVFO (translation + decimators + resampler + channel filter), 1 = the demodulator alone (fed the RxVFO's output), 2 = one windowed FFT + log-power
// per fftSize samples.  Returns INPUT samples per second of that stage (stage 1: IF samples per second).
double ref_bench_stage(int stage, const float* iq, long long totalSamples, int blockSize, double inSR, double ifRate, double bandwidth, double offset, int mode, int fftSize, double seconds) {
    if (stage == 2) {
        std::vector<float> window((size_t)fftSize), db((size_t)fftSize);
        for (int i = 0; i < fftSize; i++) { window[i] = dsp::window::nuttall(i, fftSize) * ((i % 2) ? -1.0f : 1.0f); }
        fftwf_complex* fin = (fftwf_complex*)fftwf_malloc(sizeof(fftwf_complex) * (size_t)fftSize);
        fftwf_complex* fout = (fftwf_complex*)fftwf_malloc(sizeof(fftwf_complex) * (size_t)fftSize);
        fftwf_plan plan = fftwf_plan_dft_1d(fftSize, fin, fout, FFTW_FORWARD, FFTW_ESTIMATE);
        long long n = 0;
        const auto t0 = std::chrono::steady_clock::now();
        double dt = 0.0;
        do {
            for (long long pos = 0; pos + fftSize <= totalSamples; pos += fftSize) {
                volk_32fc_32f_multiply_32fc((lv_32fc_t*)fin, (const lv_32fc_t*)(iq + 2 * pos), window.data(), fftSize);
                fftwf_execute(plan);
                volk_32fc_s32f_power_spectrum_32f(db.data(), (lv_32fc_t*)fout, fftSize, fftSize);
                n += fftSize;
            }
            dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        } while (dt < seconds);
        fftwf_destroy_plan(plan);
        fftwf_free(fin);
        fftwf_free(fout);
        return (double)n / dt;
    }
    dsp::channel::RxVFO vfo;
    vfo.init(NULL, inSR, ifRate, bandwidth, offset);
    void* d = ref_demod_create(mode, bandwidth, ifRate, 1, 50.0, 5.0, 0);
    complex_t* work = dsp::buffer::alloc<complex_t>(STREAM_BUFFER_SIZE);
    stereo_t* audio = dsp::buffer::alloc<stereo_t>(STREAM_BUFFER_SIZE);
    int nif = vfo.process(blockSize, (const complex_t*)iq, work);  // (a block of IF for stage 1)
    long long n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    double dt = 0.0;
    do {
        for (long long pos = 0; pos + blockSize <= totalSamples; pos += blockSize) {
            if (stage == 0) { n += blockSize; (void)vfo.process(blockSize, (const complex_t*)(iq + 2 * pos), work); }
            else { n += nif; (void)ref_demod_process(d, nif, (const float*)work, (float*)audio); }
        }
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (dt < seconds);
    ref_demod_destroy(d);
    dsp::buffer::free(work);
    dsp::buffer::free(audio);
    return (double)n / dt;
}

} // extern "C"
