// cfg 1 from real WAV bytes through the C++ host: sdrpp_gpu::WavSource (44-byte header, int16 or float32 payload, blocks of sr / 200,
// loop at the end of the file) -> IQFrontEnd (4096-point FFT at 20 lines/s, Nuttall) + one WFM radio.
//   usage: test_wav <plans.bin> <file.wav> <outdir> <mode: stream | direct | dump> <nblocks> [float32 0|1]
//   stream: the source feeds a dsp::stream like FileSourceModule, the front end's worker reads it (what the application does)
//   direct: the source hands its int16 blocks to the front end itself (conversion on the device)
//   dump  : no device — the blocks the source produces on its stream, as floats, to blocks.f32 (compared with the reference's reader)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "../../sdrplusplus_amd/host/sdrpp_gpu_wav.h"

struct LineSink {
    int fftSize;
    std::vector<float> cur, all;
    int acquired = 0, released = 0;
};
static float* acquire(void* c) { LineSink* s = (LineSink*)c; s->acquired++; s->cur.assign((size_t)s->fftSize, 0.0f); return s->cur.data(); }
static void release(void* c) { LineSink* s = (LineSink*)c; s->released++; s->all.insert(s->all.end(), s->cur.begin(), s->cur.end()); }

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
    const std::string mode = argv[4], outdir = argv[3];
    const long long nblocks = atoll(argv[5]);
    const bool f32 = argc > 6 && atoi(argv[6]) != 0;
    auto dump = [&](const char* name, const std::vector<float>& v) {
        std::ofstream o(outdir + "/" + name, std::ios::binary);
        o.write((const char*)v.data(), (std::streamsize)(v.size() * sizeof(float)));
    };
    sdrpp_gpu::WavSource src;
    {   // what the file selector rejects: not a RIFF/WAVE file, zero sample rate
        sdrpp_gpu::WavSource bad;
        if (bad.open(argv[1])) { fprintf(stderr, "plans.bin accepted as a WAV file\n"); return 1; }
    }
    if (!src.open(argv[2])) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    src.setFloat32Mode(f32);
    src.setBlockLimit(nblocks);
    const double sr = src.getSampleRate();
    const int B = src.blockSize();
    if (mode == "dump") {
        std::vector<float> all;
        src.start();
        while (true) {
            int n = src.stream.read();
            if (n < 0) { break; }
            const float* p = (const float*)src.stream.readBuf;
            all.insert(all.end(), p, p + 2 * (size_t)n);
            src.stream.flush();
            if ((long long)(all.size() / 2) >= nblocks * B) { break; }
        }
        src.stop();
        dump("blocks.f32", all);
        printf("sr %.0f block %d blocks %lld\n", sr, B, (long long)(all.size() / 2 / (size_t)B));
        return 0;
    }
    sdrpp_gpu::DecimPlans plans;
    if (!plans.load(argv[1])) { fprintf(stderr, "cannot load plans\n"); return 1; }
    const int fftSize = 4096;
    LineSink lines{ fftSize };
    sdrpp_gpu::IQFrontEnd fe;
    fe.init(&src.stream, sr, false, 1, false, fftSize, 20.0, sdrpp_gpu::IQFrontEnd::NUTTALL, acquire, release, &lines, 0, &plans);
    sdrpp_gpu::RxVFO* wfm = fe.addVFO("radio", 250000.0, 150000.0, 300000.0);
    if (!wfm) { return 1; }
    wfm->attachDemod(sdrpp_gpu::Demod::WFM);
    std::vector<float> audio;
    std::atomic<bool> sinkDone{ false };
    std::thread sink([&]() {
        while (true) {
            int n = wfm->audio.read();
            if (n < 0) { break; }
            const float* p = (const float*)wfm->audio.readBuf;
            audio.insert(audio.end(), p, p + 2 * (size_t)n);
            wfm->audio.flush();
        }
        sinkDone = true;
    });
    if (mode == "stream") {
        fe.start();
        src.start();
    }
    else { src.start(&fe); }
    while (!src.finished()) { std::this_thread::sleep_for(std::chrono::milliseconds(5)); }
    std::this_thread::sleep_for(std::chrono::milliseconds(argc > 7 ? atoi(argv[7]) : 300));  // the last block drains
    src.stop();
    if (mode == "stream") { fe.stop(); }
    wfm->audio.stopReader();
    sink.join();
    dump("lines.f32", lines.all);
    dump("audio.f32", audio);
    printf("sr %.0f block %d blocks %lld lines %d audio %zu\n", sr, B, src.blocksDone(), (int)(lines.all.size() / (size_t)fftSize), audio.size() / 2);
    return (lines.acquired == lines.released) ? 0 : 1;
}
