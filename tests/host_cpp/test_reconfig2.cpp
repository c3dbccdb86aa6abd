// The REST of the C++ control surface while a pipelined graph runs (VERDICT r5 missing #2): IQFrontEnd::setFFTWindow / setFFTRate / setInvertIQ /
// setBuffering / setDecimation / setSampleRate / setDCBlocking (iq_frontend.cpp:76-130, 190-198) and RxVFO::setOutSamplerate as the radio module's
// demodulator switch WFM -> NFM -> USB -> WFM (vfo_manager.cpp:52, radio_module.h:419-563: setOutSamplerate on the SAME RxVFO, then a new demodulator),
// RxVFO::setInSamplerate through the front end's rate changes (rx_vfo.h:35-58).  Same harness as test_reconfig.cpp: a source thread hands blocks over,
// sink threads read every stream, the setters are called between blocks once the worker has taken the block just handed over, results several blocks
// behind their pushes:
//     after block 1: setFFTWindow(BLACKMAN)          after block 2: setFFTRate(50)               after block 3: "radio" -> NFM
//     after block 4: setInvertIQ(true)                after block 5: "radio" -> USB               after block 6: setBuffering(true)
//     after block 7: setBuffering(false)              after block 8: "radio" -> WFM               after block 9: setDecimation(2)
//     after block 10: setSampleRate(2.0e6)            after block 11: setDCBlocking(true)         after block 12: setSampleRate(2.4e6)
// tests/test_host_cpp.py replays the schedule on the COMPILED REFERENCE's RxVFO / demodulator / pre-processing objects (oracle/_ref) — their state across
// these calls is the specification — and on the oracle's spectrum: every block of "radio" and "steady" from its first sample, every line.
//   usage: test_reconfig2 <plans.bin> <iq.f32> <sample_rate> <block> <outdir> <extra_vfos> [wait_ms]
#include <atomic>
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "../../sdrplusplus_amd/host/sdrpp_gpu_blocks.h"

struct LineSink {
    std::atomic<int> fftSize;
    std::vector<float> cur;
    std::vector<float> all;
    int acquired = 0, released = 0;
};
static float* acquire(void* c) { LineSink* s = (LineSink*)c; s->acquired++; s->cur.assign((size_t)s->fftSize.load(), 0.0f); return s->cur.data(); }
static void release(void* c) {
    LineSink* s = (LineSink*)c;
    s->released++;
    s->all.insert(s->all.end(), s->cur.begin(), s->cur.end());
}

template <class T>
static void drain(dsp::stream<T>* st, std::vector<float>* dst, std::vector<int>* counts, std::atomic<int>* nblocks) {
    while (true) {
        int n = st->read();
        if (n < 0) { break; }
        const float* p = (const float*)st->readBuf;
        dst->insert(dst->end(), p, p + 2 * (size_t)n);
        counts->push_back(n);
        st->flush();
        if (nblocks) { nblocks->fetch_add(1); }
    }
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage\n"); return 2; }
    sdrpp_gpu::DecimPlans plans;
    if (!plans.load(argv[1])) { fprintf(stderr, "cannot load plans\n"); return 1; }
    std::ifstream f(argv[2], std::ios::binary | std::ios::ate);
    const size_t bytes = (size_t)f.tellg();
    f.seekg(0);
    std::vector<float> iq(bytes / 4);
    f.read((char*)iq.data(), (std::streamsize)bytes);
    const double sr = atof(argv[3]);
    const int block = atoi(argv[4]);
    const std::string outdir = argv[5];
    const int nextra = atoi(argv[6]);
    const int waitMs = argc > 7 ? atoi(argv[7]) : 20000;
    const size_t nsamp = iq.size() / 2;
    const int nblocks = (int)(nsamp / (size_t)block);

    dsp::stream<dsp::complex_t> src;
    LineSink lines;
    lines.fftSize = 4096;
    sdrpp_gpu::IQFrontEnd fe;
    fe.init(&src, sr, false, 1, false, 4096, 100.0, sdrpp_gpu::IQFrontEnd::NUTTALL, acquire, release, &lines, 0, &plans);
    sdrpp_gpu::RxVFO* radio = fe.addVFO("radio", 250000.0, 150000.0, 300000.0);
    sdrpp_gpu::RxVFO* steady = fe.addVFO("steady", 250000.0, 150000.0, -200000.0);
    if (!radio || !steady) { return 1; }
    radio->attachDemod(sdrpp_gpu::Demod::WFM);
    steady->attachDemod(sdrpp_gpu::Demod::WFM);
    std::vector<sdrpp_gpu::RxVFO*> extra;
    std::vector<std::vector<float>> extraOut((size_t)nextra);
    std::vector<std::vector<int>> extraCnt((size_t)nextra);
    for (int k = 0; k < nextra; k++) {  // more radios of the same geometry: the matrix-core front end from 17 VFOs on (device runs)
        sdrpp_gpu::RxVFO* v = fe.addVFO("x" + std::to_string(k), 250000.0, 150000.0, -4.0e5 + 15e3 * k);
        if (!v) { return 1; }
        v->attachDemod(sdrpp_gpu::Demod::WFM);
        extra.push_back(v);
    }
    fe.setPipelining(true, 4);
    fe.setStopGrace(waitMs);
    std::vector<float> radioOut, steadyOut;
    std::vector<int> radioCnt, steadyCnt;
    std::atomic<int> radioN{ 0 }, steadyN{ 0 };
    std::thread tRadio(drain<dsp::stereo_t>, &radio->audio, &radioOut, &radioCnt, &radioN);
    std::thread tSteady(drain<dsp::stereo_t>, &steady->audio, &steadyOut, &steadyCnt, &steadyN);
    std::vector<std::thread> tExtra;
    for (int k = 0; k < nextra; k++) { tExtra.emplace_back(drain<dsp::stereo_t>, &extra[(size_t)k]->audio, &extraOut[(size_t)k], &extraCnt[(size_t)k], (std::atomic<int>*)nullptr); }
    fe.start();

    auto settled = [&](int k) -> bool {  // the worker has finished block k: a setter called now takes effect from block k + 1 on
        const auto t0 = std::chrono::steady_clock::now();
        while (fe.blocksTaken() < (uint64_t)(k + 1)) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(waitMs)) { return false; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        return true;
    };
    // the radio module's demodulator switch: VFOManager::VFO::setSampleRate -> RxVFO::setOutSamplerate(sr, bw), then the new demodulator behind the same RxVFO
    auto switchDemod = [&](sdrpp_gpu::Demod mode, double ifRate, double bw) {
        radio->setOutSamplerate(ifRate, bw);
        radio->attachDemod(mode);
    };
    size_t pos = 0;
    for (int k = 0; k < nblocks; k++) {
        memcpy(src.writeBuf, &iq[2 * pos], sizeof(float) * 2 * (size_t)block);
        if (!src.swap(block)) { fprintf(stderr, "source stream stopped at block %d\n", k); return 1; }
        pos += (size_t)block;
        if (k >= 1 && k <= 12) {
            if (!settled(k)) { fprintf(stderr, "block %d was not taken in time\n", k); return 1; }
        }
        if (k == 1) { fe.setFFTWindow(sdrpp_gpu::IQFrontEnd::BLACKMAN); }
        if (k == 2) { fe.setFFTRate(50.0); }
        if (k == 3) { switchDemod(sdrpp_gpu::Demod::NFM, 50000.0, 12500.0); }
        if (k == 4) { fe.setInvertIQ(true); }
        if (k == 5) { switchDemod(sdrpp_gpu::Demod::USB, 24000.0, 2800.0); }
        if (k == 6) { fe.setBuffering(true); }
        if (k == 7) {
            // the frame buffer's own worker hands block 7 out; the reference's SampleFrameBuffer has the same two writers on one stream when `bypass` flips with
            // frames queued (frame_buffer.h:51-98) — a GUI toggles it at rest: wait until everything queued has left before the flip
            const auto t0 = std::chrono::steady_clock::now();
            while ((radioN.load() < 8 || steadyN.load() < 8) && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(waitMs)) { std::this_thread::sleep_for(std::chrono::microseconds(200)); }
            if (radioN.load() < 8 || steadyN.load() < 8) { fprintf(stderr, "the buffered block was not handed out in time\n"); return 1; }
            fe.setBuffering(false);
        }
        if (k == 8) { switchDemod(sdrpp_gpu::Demod::WFM, 250000.0, 150000.0); }
        if (k == 9) { fe.setDecimation(2); }
        if (k == 10) { fe.setSampleRate(2.0e6); }
        if (k == 11) { fe.setDCBlocking(true); }
        if (k == 12) { fe.setSampleRate(2.4e6); }
    }
    if (!settled(nblocks - 1)) { fprintf(stderr, "the last block was not taken in time\n"); return 1; }
    fe.stop();
    if (fe.drainPipeline() < 0) { fprintf(stderr, "drainPipeline\n"); return 1; }
    {
        const auto t0 = std::chrono::steady_clock::now();
        while ((radioN.load() < nblocks || steadyN.load() < nblocks) && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(3000)) {
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    radio->audio.stopReader();
    steady->audio.stopReader();
    for (auto* v : extra) { v->audio.stopReader(); }
    tRadio.join();
    tSteady.join();
    for (auto& t : tExtra) { t.join(); }
    auto dump = [&](const char* name, const void* p, size_t n) {
        std::ofstream o(outdir + "/" + name, std::ios::binary);
        o.write((const char*)p, (std::streamsize)n);
    };
    dump("lines.f32", lines.all.data(), lines.all.size() * 4);
    dump("radio.f32", radioOut.data(), radioOut.size() * 4);
    dump("radio_counts.i32", radioCnt.data(), radioCnt.size() * 4);
    dump("steady.f32", steadyOut.data(), steadyOut.size() * 4);
    dump("steady_counts.i32", steadyCnt.data(), steadyCnt.size() * 4);
    for (int k = 0; k < nextra; k++) {
        if (extraOut[(size_t)k].size() != steadyOut.size()) { fprintf(stderr, "extra radio %d delivered %zu values, steady %zu\n", k, extraOut[(size_t)k].size(), steadyOut.size()); return 1; }
    }
    printf("blocks %d lines %zu (acquire %d release %d) radio %zu in %zu blocks, steady %zu in %zu\n", nblocks, lines.all.size() / 4096, lines.acquired, lines.released, radioOut.size() / 2, radioCnt.size(),
           steadyOut.size() / 2, steadyCnt.size());
    return (lines.acquired == lines.released) ? 0 : 1;
}
