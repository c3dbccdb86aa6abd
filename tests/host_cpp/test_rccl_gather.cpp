// sdrpp_gpu::LineGather (host/sdrpp_gpu_rccl.h): one context per visible device, one block of the cfg 2 signal (a different level per
// stream) through each, the zoomed lines of every stream gathered on device 0 with RCCL and compared with what sdrpp_fft_read returns.
// On a one-GPU machine the communicator has one rank: the RCCL calls still run (send / receive to itself).
//   usage: test_rccl_gather            prints "ranks N lines L ok" and exits 0
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/sdrpp_gpu.h"
#include "../../sdrplusplus_amd/host/sdrpp_gpu_rccl.h"

int main() {
    const int ndev = sdrpp_device_count();
    if (ndev <= 0) { fprintf(stderr, "no device\n"); return 2; }
    const int N = 4096, W = 512, nlines = 6;
    const int64_t count = (int64_t)N * nlines;
    std::vector<sdrpp_ctx*> ctx((size_t)ndev, nullptr);
    std::vector<float> win((size_t)N);
    sdrpp_design_fft_window(2, N, win.data());
    std::vector<std::vector<float>> expect((size_t)ndev);
    std::vector<const float*> send((size_t)ndev);
    std::vector<int> devs;
    for (int d = 0; d < ndev; d++) {
        devs.push_back(d);
        if (sdrpp_create(d, count, &ctx[(size_t)d])) { fprintf(stderr, "create on device %d failed\n", d); return 1; }
        sdrpp_fft_configure(ctx[(size_t)d], N, N, 0, win.data());
        int start = 0, size = 0;
        sdrpp_design_waterfall_view(0.0, 10e6, 10e6, N, &start, &size);
        sdrpp_fft_set_view(ctx[(size_t)d], start, size, W, -120.0f, 0.0f);
        std::vector<float> x((size_t)count * 2);
        for (int64_t i = 0; i < count; i++) {
            const double ph = 2.0 * M_PI * (0.1 + 0.03 * d) * (double)i;
            x[(size_t)(2 * i)] = (float)((0.5 / (d + 1)) * std::cos(ph));
            x[(size_t)(2 * i + 1)] = (float)((0.5 / (d + 1)) * std::sin(ph));
        }
        if (sdrpp_push(ctx[(size_t)d], x.data(), count)) { fprintf(stderr, "push: %s\n", sdrpp_last_error(ctx[(size_t)d])); return 1; }
        if (sdrpp_fft_lines(ctx[(size_t)d]) != nlines) { fprintf(stderr, "lines\n"); return 1; }
        expect[(size_t)d].resize((size_t)nlines * W);
        if (sdrpp_fft_read(ctx[(size_t)d], 0, nlines, nullptr, expect[(size_t)d].data(), nullptr) != nlines) { return 1; }
        const float *raw = nullptr, *zo = nullptr;
        const int32_t* ix = nullptr;
        int nl = 0;
        if (sdrpp_fft_device_buffers(ctx[(size_t)d], &raw, &zo, &ix, &nl) || nl != nlines || !zo) { fprintf(stderr, "device buffers\n"); return 1; }
        sdrpp_sync(ctx[(size_t)d]);
        send[(size_t)d] = zo;
    }
    try {
        sdrpp_gpu::LineGather g;
        g.init(devs, (size_t)nlines * W, 0);
        g.gather(send, (size_t)nlines * W);
        std::vector<float> got((size_t)ndev * nlines * W);
        g.toHost(got.data(), (size_t)nlines * W);
        for (int d = 0; d < ndev; d++) {
            if (memcmp(&got[(size_t)d * nlines * W], expect[(size_t)d].data(), (size_t)nlines * W * sizeof(float)) != 0) {
                fprintf(stderr, "stream %d: gathered lines differ\n", d);
                return 1;
            }
        }
        printf("ranks %d lines %d backend %s ok\n", g.ranks(), nlines, g.backend());
    }
    catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    for (auto* c : ctx) { sdrpp_destroy(c); }
    return 0;
}
