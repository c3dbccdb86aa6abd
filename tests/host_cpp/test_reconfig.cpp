// Reconfiguration WHILE a pipelined graph runs, through the C++ mirror — what the GUI does all day (iq_frontend.cpp:105-183: every setter
// brackets its change with tempStop / tempStart; dsp/block.h:46-94).  A source thread hands blocks over like SDR++'s sources do, sink threads
// read every VFO's stream, and between blocks a control thread (here: the source thread itself, once the worker has finished the block just
// handed over — IQFrontEnd::blocksTaken — so that the schedule is deterministic) calls, with setPipelining(true) and results several blocks
// behind their pushes:
//     after block 2: addVFO("late") + attachDemod          after block 4: RxVFO::setBandwidth on "radio"
//     after block 5: removeVFO("victim") with its blocks still in flight (its sink is stopped first, as the radio module does)
//     after block 6: setFFTSize(2048)                       after block 8: setPipelining(false)      after block 9: setPipelining(true)
//     after block 10: stop(); start()
// Every stream is dumped; tests/test_host_cpp.py replays the same schedule on the oracle: every delivered block matches, none is lost or
// duplicated ("radio" and "late" complete, "victim" a prefix), lines bit-exact across the change of the FFT size.
//   usage: test_reconfig <plans.bin> <iq.f32> <sample_rate> <block> <outdir> <extra_vfos> [wait_ms]
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
    std::vector<float> all;      // lines of every size, back to back
    std::vector<int> sizes;      // the size of each line
    int acquired = 0, released = 0;
};
static float* acquire(void* c) { LineSink* s = (LineSink*)c; s->acquired++; s->cur.assign((size_t)s->fftSize.load(), 0.0f); return s->cur.data(); }
static void release(void* c) {
    LineSink* s = (LineSink*)c;
    s->released++;
    s->all.insert(s->all.end(), s->cur.begin(), s->cur.end());
    s->sizes.push_back((int)s->cur.size());
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
    sdrpp_gpu::RxVFO* victim = fe.addVFO("victim", 250000.0, 150000.0, -200000.0);
    if (!radio || !victim) { return 1; }
    radio->attachDemod(sdrpp_gpu::Demod::WFM);
    victim->attachDemod(sdrpp_gpu::Demod::WFM);
    // more radios of the same geometry: from 17 VFOs on the bank has the matrix-core front end and its blocks run as ticks (device runs)
    std::vector<sdrpp_gpu::RxVFO*> extra;
    std::vector<std::vector<float>> extraOut((size_t)nextra);
    std::vector<std::vector<int>> extraCnt((size_t)nextra);
    for (int k = 0; k < nextra; k++) {
        sdrpp_gpu::RxVFO* v = fe.addVFO("x" + std::to_string(k), 250000.0, 150000.0, -1.0e6 + 40e3 * k);
        if (!v) { return 1; }
        v->attachDemod(sdrpp_gpu::Demod::WFM);
        extra.push_back(v);
    }
    fe.setPipelining(true, 4);
    fe.setStopGrace(waitMs);  // (a CPU emulator under a sanitizer or a loaded box needs more than the default 250 ms per block: no block may be dropped for that here)
    std::vector<float> radioOut, victimOut, lateOut;
    std::vector<int> radioCnt, victimCnt, lateCnt;
    std::atomic<int> radioN{ 0 }, lateN{ 0 };
    std::thread tRadio(drain<dsp::stereo_t>, &radio->audio, &radioOut, &radioCnt, &radioN);
    std::thread tVictim(drain<dsp::stereo_t>, &victim->audio, &victimOut, &victimCnt, (std::atomic<int>*)nullptr);
    std::vector<std::thread> tExtra;
    for (int k = 0; k < nextra; k++) { tExtra.emplace_back(drain<dsp::stereo_t>, &extra[(size_t)k]->audio, &extraOut[(size_t)k], &extraCnt[(size_t)k], (std::atomic<int>*)nullptr); }
    std::thread tLate;
    sdrpp_gpu::RxVFO* late = nullptr;
    fe.start();

    // the worker has finished block `k` (taken, processed / pushed): a setter called now takes effect from block k + 1 on
    auto settled = [&](int k) -> bool {
        const auto t0 = std::chrono::steady_clock::now();
        while (fe.blocksTaken() < (uint64_t)(k + 1)) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(waitMs)) { return false; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        return true;
    };
    size_t pos = 0;
    for (int k = 0; k < nblocks; k++) {
        memcpy(src.writeBuf, &iq[2 * pos], sizeof(float) * 2 * (size_t)block);
        if (!src.swap(block)) { fprintf(stderr, "source stream stopped at block %d\n", k); return 1; }
        pos += (size_t)block;
        if (k == 2 || k == 4 || k == 5 || k == 6 || k == 8 || k == 9 || k == 10) {
            if (!settled(k)) { fprintf(stderr, "block %d was not taken in time\n", k); return 1; }
        }
        if (k == 2) {
            late = fe.addVFO("late", 250000.0, 150000.0, 500000.0);
            if (!late) { return 1; }
            late->attachDemod(sdrpp_gpu::Demod::WFM);
            tLate = std::thread(drain<dsp::stereo_t>, &late->audio, &lateOut, &lateCnt, &lateN);
        }
        if (k == 4) { radio->setBandwidth(120000.0); }
        if (k == 5) {  // the consumer goes first (the radio module stops its chain before it deletes the VFO), then the VFO — with its newest blocks still in flight
            victim->audio.stopReader();
            tVictim.join();
            fe.removeVFO("victim");
            victim = nullptr;
        }
        if (k == 6) {
            lines.fftSize = 2048;  // (acquire hands out buffers of the size the display was told: gui::waterfall.setRawFFTSize precedes setFFTSize in the reference)
            fe.setFFTSize(2048);
        }
        if (k == 8) { fe.setPipelining(false); }
        if (k == 9) { fe.setPipelining(true, 4); }
        if (k == 10) {
            fe.stop();
            fe.start();
        }
    }
    if (!settled(nblocks - 1)) { fprintf(stderr, "the last block was not taken in time\n"); return 1; }
    fe.stop();
    if (fe.drainPipeline() < 0) { fprintf(stderr, "drainPipeline\n"); return 1; }
    // swap() returns when a block has been HANDED to a stream, not when its reader has taken it — and a stopped reader does not take it any
    // more: the sinks catch up (bounded) before they are stopped, or the last block of every stream would be lost in this test's own shutdown
    {
        const auto t0 = std::chrono::steady_clock::now();
        while ((radioN.load() < nblocks || lateN.load() < nblocks - 3) && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2000)) {
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));  // (the other radios' sinks)
    }
    radio->audio.stopReader();
    if (late) { late->audio.stopReader(); }
    for (auto* v : extra) { v->audio.stopReader(); }
    tRadio.join();
    if (tLate.joinable()) { tLate.join(); }
    for (auto& t : tExtra) { t.join(); }
    auto dump = [&](const char* name, const void* p, size_t n) {
        std::ofstream o(outdir + "/" + name, std::ios::binary);
        o.write((const char*)p, (std::streamsize)n);
    };
    dump("lines.f32", lines.all.data(), lines.all.size() * 4);
    dump("line_sizes.i32", lines.sizes.data(), lines.sizes.size() * 4);
    dump("radio.f32", radioOut.data(), radioOut.size() * 4);
    dump("radio_counts.i32", radioCnt.data(), radioCnt.size() * 4);
    dump("victim.f32", victimOut.data(), victimOut.size() * 4);
    dump("victim_counts.i32", victimCnt.data(), victimCnt.size() * 4);
    dump("late.f32", lateOut.data(), lateOut.size() * 4);
    dump("late_counts.i32", lateCnt.data(), lateCnt.size() * 4);
    for (int k = 0; k < nextra; k++) {
        if (extraOut[(size_t)k].size() != radioOut.size()) { fprintf(stderr, "extra radio %d delivered %zu values, radio %zu\n", k, extraOut[(size_t)k].size(), radioOut.size()); return 1; }
    }
    printf("blocks %d lines %zu (acquire %d release %d) radio %zu in %zu blocks, late %zu in %zu, victim %zu in %zu\n", nblocks, lines.sizes.size(), lines.acquired, lines.released,
           radioOut.size() / 2, radioCnt.size(), lateOut.size() / 2, lateCnt.size(), victimOut.size() / 2, victimCnt.size());
    return (lines.acquired == lines.released) ? 0 : 1;
}
