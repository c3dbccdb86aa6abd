// sdrpp_gpu::BankLineGather (host/sdrpp_gpu_rccl.h): a StreamBank of two streams whose newest waterfall lines are kept on their devices and
// gathered over RCCL on the display device — against the lines the bank's handler received through host memory (the same line of the same
// stream, bit for bit).  On a one-GPU box both streams share device 0: one RCCL rank, both lines in its block, the collective still runs.
//   usage: test_bank_gather <plans.bin>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../../sdrplusplus_amd/host/sdrpp_gpu_rccl.h"

int main(int argc, char** argv) {
    if (argc < 2) { return 2; }
    sdrpp_gpu::DecimPlans plans;
    if (!plans.load(argv[1])) { return 1; }
    const double sr = 2.4e6;
    const int block = 12000, N = 4096, nblk = 8;
    dsp::stream<dsp::complex_t> src[2];
    std::vector<float> last[2];
    int nlines[2] = { 0, 0 };
    sdrpp_gpu::StreamBank bank;
    bank.init({ &src[0], &src[1] }, sr, false, 1, false, N, 100.0, sdrpp_gpu::IQFrontEnd::NUTTALL, [&](int idx, const float* line, int n) {
        last[idx].assign(line, line + n);
        nlines[idx]++;
    }, &plans);
    try {
        sdrpp_gpu::BankLineGather bg;
        bg.init(bank, 0);
        bank.start();
        std::vector<std::thread> feeders;
        for (int s = 0; s < 2; s++) {
            feeders.emplace_back([&, s]() {
                for (int b = 0; b < nblk; b++) {
                    for (int i = 0; i < block; i++) {
                        const double t = (double)(b * block + i) / sr;
                        const double ph = 2.0 * M_PI * (s ? -250e3 : 300e3) * t;
                        src[s].writeBuf[i] = { (float)((s ? 0.2 : 0.4) * std::cos(ph)), (float)((s ? 0.2 : 0.4) * std::sin(ph)) };
                    }
                    if (!src[s].swap(block)) { break; }
                }
            });
        }
        for (auto& f : feeders) { f.join(); }
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
        bank.stop();  // (workers stopped: the kept lines and the handler's copies are final)
        std::vector<float> got;
        const int have = bg.gather(got);
        if (have != 2 || nlines[0] < 2 || nlines[1] < 2) { fprintf(stderr, "lines kept %d, delivered %d / %d\n", have, nlines[0], nlines[1]); return 1; }
        for (int s = 0; s < 2; s++) {
            if ((int)last[s].size() != N || memcmp(&got[(size_t)s * N], last[s].data(), (size_t)N * sizeof(float)) != 0) {
                fprintf(stderr, "stream %d: the gathered line differs from the delivered one\n", s);
                return 1;
            }
        }
        if (memcmp(&got[0], &got[(size_t)N], (size_t)N * sizeof(float)) == 0) { fprintf(stderr, "both streams gathered the same line\n"); return 1; }
        printf("ranks %d streams 2 lines %d / %d ok\n", bg.ranks(), nlines[0], nlines[1]);

        // ---- round 5: the same with the front ends PIPELINED (the kept line comes out of the block's result slot), a second thread gathering
        // WHILE the workers flip their buffers, and an FFT-size change under a live gatherer (stale / freed / half-written lines) ----
        auto feed = [&](int blocks, int phase0) {
            std::vector<std::thread> fs;
            for (int s = 0; s < 2; s++) {
                fs.emplace_back([&, s]() {
                    for (int b = 0; b < blocks; b++) {
                        for (int i = 0; i < block; i++) {
                            const double t = (double)((phase0 + b) * block + i) / sr;
                            const double ph = 2.0 * M_PI * (s ? -410e3 : 150e3) * t;
                            src[s].writeBuf[i] = { (float)((s ? 0.3 : 0.25) * std::cos(ph)), (float)((s ? 0.3 : 0.25) * std::sin(ph)) };
                        }
                        if (!src[s].swap(block)) { break; }
                    }
                });
            }
            for (auto& f : fs) { f.join(); }
        };
        for (int s = 0; s < 2; s++) { bank[s].setPipelining(true); }
        nlines[0] = nlines[1] = 0;
        bank.start();
        std::atomic<bool> quit{ false };
        std::atomic<int> gathers{ 0 }, bad{ 0 };
        std::thread gatherer([&]() {
            std::vector<float> g2;
            while (!quit.load()) {
                const int h = bg.gather(g2);
                gathers++;
                // a line that is handed out is a COMPLETE line of ONE block: its peak bin is where this stream's tone is (never a mix of two buffers)
                const int n2 = (int)(g2.size() / 2);
                for (int s = 0; s < 2 && h > 0; s++) {
                    const float* l = &g2[(size_t)s * n2];
                    int arg = 0;
                    bool any = false;
                    for (int k = 0; k < n2; k++) {
                        if (l[k] != 0.0f) { any = true; }
                        if (l[k] > l[arg]) { arg = k; }
                    }
                    if (!any) { continue; }  // "no line (at this size) yet"
                    const double f1 = (s ? -250e3 : 300e3), f2 = (s ? -410e3 : 150e3);
                    const int b1 = (int)std::lround(n2 / 2 + f1 / sr * n2), b2 = (int)std::lround(n2 / 2 + f2 / sr * n2);
                    if (std::abs(arg - b1) > 2 && std::abs(arg - b2) > 2) { bad++; }
                }
            }
        });
        feed(10, nblk);
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        bank.setFFTSize(2048);  // buffers of the front ends are re-allocated under the gatherer; the gather follows the bank's size
        feed(10, nblk + 10);
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
        bank.stop();
        quit = true;
        gatherer.join();
        const int have2 = bg.gather(got);
        if (have2 != 2 || (int)got.size() != 2 * 2048 || nlines[0] < 2 || nlines[1] < 2) {
            fprintf(stderr, "pipelined: lines kept %d, gathered %zu floats, delivered %d / %d\n", have2, got.size(), nlines[0], nlines[1]);
            return 1;
        }
        for (int s = 0; s < 2; s++) {
            if ((int)last[s].size() != 2048 || memcmp(&got[(size_t)s * 2048], last[s].data(), (size_t)2048 * sizeof(float)) != 0) {
                fprintf(stderr, "pipelined stream %d: the gathered line differs from the delivered one\n", s);
                return 1;
            }
        }
        if (bad.load() != 0 || gathers.load() < 3) { fprintf(stderr, "concurrent gathers: %d, with a torn / foreign line: %d\n", gathers.load(), bad.load()); return 1; }
        printf("pipelined + concurrent gathers %d (none torn) + FFT size 4096 -> 2048 ok\n", gathers.load());
    }
    catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
