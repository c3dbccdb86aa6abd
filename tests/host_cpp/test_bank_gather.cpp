// sdrpp_gpu::BankLineGather (host/sdrpp_gpu_rccl.h): a StreamBank of two streams whose newest waterfall lines are kept on their devices and
// gathered over RCCL on the display device — against the lines the bank's handler received through host memory (the same line of the same
// stream, bit for bit).  On a one-GPU box both streams share device 0: one RCCL rank, both lines in its block, the collective still runs.
//   usage: test_bank_gather <plans.bin>
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
    }
    catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
