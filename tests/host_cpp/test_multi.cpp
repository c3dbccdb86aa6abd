// sdrpp_gpu::StreamBank (sdrplusplus_amd/host/sdrpp_gpu_multi.h): two independent IQ streams, each with its own front end / context /
// worker (on a one-device machine both land on device 0), lines of both arriving through the one handler with the right stream index.
//   usage: test_multi <plans.bin> <outdir> [drain_ms]      writes lines_0.f32 / lines_1.f32 and audio_0.f32 / audio_1.f32
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "../../sdrplusplus_amd/host/sdrpp_gpu_multi.h"

int main(int argc, char** argv) {
    if (argc < 3) { return 2; }
    sdrpp_gpu::DecimPlans plans;
    if (!plans.load(argv[1])) { return 1; }
    const std::string outdir = argv[2];
    const int drainMs = argc > 3 ? atoi(argv[3]) : 500;
    const double sr = 2.4e6;
    const int block = 12000, N = 4096, nblk = 6;
    dsp::stream<dsp::complex_t> src[2];
    std::vector<float> lines[2], audio[2];
    sdrpp_gpu::StreamBank bank;
    bank.init({ &src[0], &src[1] }, sr, false, 1, false, N, 100.0, sdrpp_gpu::IQFrontEnd::NUTTALL,
              [&](int idx, const float* line, int n) { lines[idx].insert(lines[idx].end(), line, line + n); }, &plans);
    if (bank.size() != 2) { return 1; }
    sdrpp_gpu::RxVFO* v[2];
    std::vector<std::thread> sinks;
    for (int s = 0; s < 2; s++) {
        v[s] = bank[s].addVFO("radio", 250000.0, 150000.0, 300000.0);
        v[s]->attachDemod(sdrpp_gpu::Demod::WFM);
        sinks.emplace_back([&, s]() {
            while (true) {
                int n = v[s]->audio.read();
                if (n < 0) { break; }
                const float* p = (const float*)v[s]->audio.readBuf;
                audio[s].insert(audio[s].end(), p, p + 2 * (size_t)n);
                v[s]->audio.flush();
            }
        });
    }
    bank.start();
    std::vector<std::thread> feeders;
    for (int s = 0; s < 2; s++) {
        feeders.emplace_back([&, s]() {
            for (int b = 0; b < nblk; b++) {
                for (int i = 0; i < block; i++) {  // stream 0: FM carrier at +300 kHz; stream 1: a different tone frequency and level
                    const double t = (double)(b * block + i) / sr;
                    const double ph = 2.0 * M_PI * 300e3 * t + (s ? 20.0 : 60.0) * std::sin(2.0 * M_PI * (s ? 1700.0 : 1000.0) * t);
                    src[s].writeBuf[i] = { (float)((s ? 0.2 : 0.4) * std::cos(ph)), (float)((s ? 0.2 : 0.4) * std::sin(ph)) };
                }
                if (!src[s].swap(block)) { break; }
            }
        });
    }
    for (auto& f : feeders) { f.join(); }
    std::this_thread::sleep_for(std::chrono::milliseconds(drainMs));
    bank.stop();
    for (int s = 0; s < 2; s++) { v[s]->audio.stopReader(); }
    for (auto& t : sinks) { t.join(); }
    for (int s = 0; s < 2; s++) {
        std::ofstream a(outdir + "/lines_" + std::to_string(s) + ".f32", std::ios::binary), b(outdir + "/audio_" + std::to_string(s) + ".f32", std::ios::binary);
        a.write((const char*)lines[s].data(), (std::streamsize)(lines[s].size() * 4));
        b.write((const char*)audio[s].data(), (std::streamsize)(audio[s].size() * 4));
    }
    printf("lines %zu / %zu, audio %zu / %zu\n", lines[0].size() / N, lines[1].size() / N, audio[0].size() / 2, audio[1].size() / 2);
    return 0;
}
