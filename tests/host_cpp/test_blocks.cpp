// Drives the C++ host mirror (sdrplusplus_amd/host/sdrpp_gpu_blocks.h) the way SDR++ drives IQFrontEnd: a source thread
// swap()s IQ blocks into a dsp::stream, the front end delivers dB lines through acquire/release callbacks and per-VFO blocks
// on dsp::streams read by sink threads.  Outputs are written to files and compared with the oracle by tests/test_host_cpp.py.
//   usage: test_blocks <plans.bin> <iq.f32> <sample_rate> <block> <fft_size> <fft_rate> <outdir> [buffered|bypass] [drain_ms]
// Built three ways by tests/test_host_cpp.py: against the test double of dsp::block / dsp::stream (tests/host_cpp/standalone), against
// the reference's REAL headers (-I<sdrpp>/core/src, oracle/_ref/test_blocks_ref), and with -DSDRPP_GPU_TEST_DEMOD_IFACE against the
// radio module's demod::Demodulator interface (extracted from the reference at build time, oracle/_ref/demod_iface.h).
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "../../sdrplusplus_amd/host/sdrpp_gpu_blocks.h"
#ifdef SDRPP_GPU_TEST_DEMOD_IFACE
#include "demod_iface.h"  // namespace demod { class Demodulator { ... }; } as declared by decoder_modules/radio/src/demod.h
#include "../../sdrplusplus_amd/host/sdrpp_gpu_radio.h"
#endif

struct LineSink {
    int fftSize;
    std::vector<float> cur;
    std::vector<float> all;
    int acquired = 0, released = 0;
};
static float* acquire(void* c) { LineSink* s = (LineSink*)c; s->acquired++; s->cur.assign((size_t)s->fftSize, 0.0f); return s->cur.data(); }
static void release(void* c) { LineSink* s = (LineSink*)c; s->released++; s->all.insert(s->all.end(), s->cur.begin(), s->cur.end()); }

template <class T>
static void drain(dsp::stream<T>* st, std::vector<float>* dst) {
    while (true) {
        int n = st->read();
        if (n < 0) { break; }
        const float* p = (const float*)st->readBuf;
        dst->insert(dst->end(), p, p + 2 * (size_t)n);
        st->flush();
    }
}

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage\n"); return 2; }
    sdrpp_gpu::DecimPlans plans;
    if (!plans.load(argv[1])) { fprintf(stderr, "cannot load plans\n"); return 1; }
    std::ifstream f(argv[2], std::ios::binary | std::ios::ate);
    const size_t bytes = (size_t)f.tellg();
    f.seekg(0);
    std::vector<float> iq(bytes / 4);
    f.read((char*)iq.data(), (std::streamsize)bytes);
    const double sr = atof(argv[3]);
    const int block = atoi(argv[4]), fftSize = atoi(argv[5]);
    const double fftRate = atof(argv[6]);
    const std::string outdir = argv[7];
    const bool buffered = argc > 8 && std::string(argv[8]) == "buffered";
    // "pipelined": the bypass path as one launch per block (IQFrontEnd::setPipelining), results handed out a few blocks late and the tail by
    // drainPipeline() after stop() — with the IQ tap bound and the AF chain attached like in the other modes (round 4: both stay pipelined).
    // "pipelined_groups": the same with up to four blocks per launch (setPipelining(true, lag, 4): sdrpp_set_pipeline_group, adaptive) — the source here is
    // far slower than a launch, so groups form only when the worker falls behind; the results must not depend on whether they do.
    const bool groups = argc > 8 && std::string(argv[8]) == "pipelined_groups";
    const bool pipelined = groups || (argc > 8 && std::string(argv[8]) == "pipelined");
    const int drainMs = argc > 9 ? atoi(argv[9]) : (buffered ? 1500 : 300);  // time to let handed-over blocks drain (the CPU emulator needs seconds)
    const size_t nsamp = iq.size() / 2;

    dsp::stream<dsp::complex_t> src, src2;  // the second one takes over half way (IQFrontEnd::setInput, source.cpp:29,57)
    LineSink lines{ fftSize };
    sdrpp_gpu::IQFrontEnd fe;
    fe.init(&src, sr, buffered, 1, false, fftSize, fftRate, sdrpp_gpu::IQFrontEnd::NUTTALL, acquire, release, &lines, 0, &plans);
    if (fe.getSampleRate() != sr || fe.getEffectiveSamplerate() != sr) { fprintf(stderr, "getSampleRate\n"); return 1; }
    sdrpp_gpu::RxVFO* raw = fe.addVFO("raw", 250000.0, 150000.0, sr / 8);
    sdrpp_gpu::RxVFO* wfm = fe.addVFO("radio", 250000.0, 150000.0, 300000.0);
    if (!raw || !wfm) { return 1; }
#ifdef SDRPP_GPU_TEST_DEMOD_IFACE
    // the radio module's view: a demod::Demodulator that was handed the VFO's output stream (radio_module.h:455-473)
    sdrpp_gpu::FusedDemodulator<demod::Demodulator, sdrpp_gpu::Demod::WFM> gpuWfm(&fe);
    demod::Demodulator* dm = &gpuWfm;
    dm->init("radio", nullptr, &wfm->out, 150000.0, 48000.0);
    dm->start();
    if (dm->getOutput() != &wfm->audio || dm->getIFSampleRate() != 250000.0 || std::string(dm->getName()) != "WFM" || dm->getVFOReference() != 1) {
        fprintf(stderr, "Demodulator adaptor\n");
        return 1;
    }
    {   // an input that is no VFO output cannot be fused
        dsp::stream<dsp::complex_t> stray;
        bool threw = false;
        sdrpp_gpu::FusedDemodulator<demod::Demodulator, sdrpp_gpu::Demod::NFM> bad(&fe);
        try { bad.init("x", nullptr, &stray, 12500.0, 48000.0); } catch (const std::runtime_error&) { threw = true; }
        if (!threw) { fprintf(stderr, "stray input accepted\n"); return 1; }
    }
#else
    wfm->attachDemod(sdrpp_gpu::Demod::WFM);
#endif
    // a consumer of the (pre-processed) wideband IQ, like the recorder's baseband tap (recorder/src/main.cpp:209,229)
    dsp::stream<dsp::complex_t> iqTap;
    if (pipelined) { fe.setPipelining(true, 4, groups ? 4 : 1); }
    fe.bindIQStream(&iqTap);
    {
        bool threw = false;
        try { fe.bindIQStream(&iqTap); } catch (const std::runtime_error&) { threw = true; }  // Splitter::bindStream, splitter.h:18-20
        dsp::stream<dsp::complex_t> other;
        bool threw2 = false;
        try { fe.unbindIQStream(&other); } catch (const std::runtime_error&) { threw2 = true; }
        if (!threw || !threw2) { fprintf(stderr, "bind/unbind error behaviour\n"); return 1; }
    }
    // a second radio instance with the AF chain (48 kHz, 50 us de-emphasis) behind its demodulator (radio_module.h:98-110)
    sdrpp_gpu::RxVFO* wfmAf = fe.addVFO("radio_af", 250000.0, 150000.0, 300000.0);
    if (!wfmAf) { return 1; }
    wfmAf->attachDemod(sdrpp_gpu::Demod::WFM);
    wfmAf->attachAF(48000.0, 50e-6, false);
    if (fe.addVFO("raw", 1.0, 1.0, 0.0) != nullptr) { fprintf(stderr, "duplicate VFO name accepted\n"); return 1; }
    fe.removeVFO("nope");  // logs, like the reference

    // pipelined mode: 17 more radios so that the bank has the matrix-core front end (>= 17 VFOs of one geometry) and the blocks really
    // run as ticks; their audio is read and dropped
    std::vector<sdrpp_gpu::RxVFO*> extra;
    std::vector<std::vector<float>> extraOut(17);
    std::vector<std::thread> extraThreads;
    if (pipelined) {
        for (int k = 0; k < 17; k++) {
            sdrpp_gpu::RxVFO* v = fe.addVFO("x" + std::to_string(k), 250000.0, 150000.0, -1.0e6 + 50e3 * k);
            if (!v) { return 1; }
            v->attachDemod(sdrpp_gpu::Demod::WFM);
            extra.push_back(v);
        }
        for (int k = 0; k < 17; k++) { extraThreads.emplace_back(drain<dsp::stereo_t>, &extra[(size_t)k]->audio, &extraOut[(size_t)k]); }
    }
    std::vector<float> ifOut, audioOut, afOut, tapOut;
    std::thread t0(drain<dsp::complex_t>, &iqTap, &tapOut);
    std::thread t1(drain<dsp::complex_t>, &raw->out, &ifOut);
    std::thread t2(drain<dsp::stereo_t>, &wfm->audio, &audioOut);
    std::thread t3(drain<dsp::stereo_t>, &wfmAf->audio, &afOut);
    fe.start();
    fe.flushInputBuffer();  // nothing queued yet: a no-op that must not disturb the indices (main_window.cpp:679,689)
    size_t pos = 0;
    int blk = 0;
    const int nblocks = (int)(nsamp / (size_t)block);
    dsp::stream<dsp::complex_t>* cur = &src;
    while (pos + (size_t)block <= nsamp) {
        memcpy(cur->writeBuf, &iq[2 * pos], sizeof(float) * 2 * (size_t)block);
        if (!cur->swap(block)) { break; }
        pos += (size_t)block;
        blk++;
        if (blk == 3) { raw->setOffset(-sr / 4); }  // retune while running (phase-continuous)
        if (blk == nblocks / 2) {  // change of source: everything handed over so far must have been consumed first
            std::this_thread::sleep_for(std::chrono::milliseconds(drainMs));
            fe.setInput(&src2);
            cur = &src2;
        }
    }
    // let the last block drain: a final empty swap is not part of the reference protocol, so wait on the line/audio counts instead
    std::this_thread::sleep_for(std::chrono::milliseconds(drainMs));
    fe.stop();
    if (pipelined) {  // the blocks still in flight: handed out now (the sinks are still reading)
        if (fe.drainPipeline() < 0) { fprintf(stderr, "drainPipeline\n"); return 1; }
        std::this_thread::sleep_for(std::chrono::milliseconds(drainMs));
    }
    fe.unbindIQStream(&iqTap);
    iqTap.stopReader();
    raw->out.stopReader();
    wfm->audio.stopReader();
    wfmAf->audio.stopReader();
    for (auto* v : extra) { v->audio.stopReader(); }
    for (auto& t : extraThreads) { t.join(); }
    t0.join();
    t1.join();
    t2.join();
    t3.join();
    for (auto& o : extraOut) {
        if (pipelined && o.size() != audioOut.size()) { fprintf(stderr, "extra radio delivered %zu values, expected %zu\n", o.size(), audioOut.size()); return 1; }
    }
    auto dump = [&](const char* name, const std::vector<float>& v) {
        std::ofstream o(outdir + "/" + name, std::ios::binary);
        o.write((const char*)v.data(), (std::streamsize)(v.size() * sizeof(float)));
    };
    dump("lines.f32", lines.all);
    dump("if.f32", ifOut);
    dump("audio.f32", audioOut);
    dump("af.f32", afOut);
    dump("iq_tap.f32", tapOut);
    printf("blocks %d lines %d (acquire %d release %d) if %zu audio %zu\n", blk, (int)(lines.all.size() / (size_t)fftSize), lines.acquired, lines.released, ifOut.size() / 2, audioOut.size() / 2);
    return (lines.acquired == lines.released) ? 0 : 1;
}
