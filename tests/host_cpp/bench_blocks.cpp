// Throughput of the hot path THROUGH the C++ host mirror, measured the way SDR++'s SpeedTester measures a graph
// (core/src/dsp/bench/speed_tester.h:31-92): an unthrottled source thread swap()s fixed-size IQ blocks into the front end's input
// stream (back-pressured by the stream hand-off), sink threads read() / flush() every output stream, and the rate is samples
// ingested per wall second.  Workload = BASELINE cfg 3 (65536-point FFT, nvfo WFM VFOs 300 kHz apart) at the block size given.
//   usage: bench_blocks <plans.bin> <sample_rate> <block> <fft_size> <nvfo> <seconds> <buffered 0|1> [pipelined 0|1 [blocks per launch]]
// Prints one JSON object.  Built by oracle/Makefile against the reference's own dsp/stream.h + dsp/block.h (oracle/_ref/bench_blocks_ref: what
// bench.py runs) and, where the reference tree is absent, by bench.py against tests/host_cpp/standalone (the test double of those headers).
#include <fcntl.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include "../../sdrplusplus_amd/host/sdrpp_gpu_blocks.h"

static std::atomic<long long> g_lines{ 0 };
static std::vector<float> g_line;
static float* acquire(void*) { return g_line.data(); }
static void release(void*) { g_lines++; }

int main(int argc, char** argv) {
    if (argc < 8) {
        fprintf(stderr, "usage: bench_blocks <plans.bin> <sample_rate> <block> <fft_size> <nvfo> <seconds> <buffered>\n");
        return 2;
    }
    // SDRPP_BENCH_CPU_DMA_LATENCY=<us>: hold /dev/cpu_dma_latency at that value for the run (PM QoS: keeps the cores out of the idle states whose exit
    // latency is longer) — the stream hand-overs of the graph are futex wake-ups of threads that sleep for tens of microseconds at a time, and on a
    // server part the wake-up out of a deep idle state is longer than the block's whole device time (what real-time audio applications set too)
    int qosFd = -1;
    if (const char* q = getenv("SDRPP_BENCH_CPU_DMA_LATENCY")) {
        qosFd = open("/dev/cpu_dma_latency", O_WRONLY);
        const int32_t us = atoi(q);
        if (qosFd < 0 || write(qosFd, &us, sizeof(us)) != (ssize_t)sizeof(us)) { fprintf(stderr, "cpu_dma_latency not available\n"); }
    }
    sdrpp_gpu::DecimPlans plans;
    if (!plans.load(argv[1])) {
        fprintf(stderr, "cannot load plans\n");
        return 1;
    }
    const double sr = atof(argv[2]);
    const int block = atoi(argv[3]), fftSize = atoi(argv[4]), nvfo = atoi(argv[5]);
    const double seconds = atof(argv[6]);
    const bool buffered = atoi(argv[7]) != 0;
    const bool pipelined = argc > 8 && atoi(argv[8]) != 0;  // bypass path as one launch per block (IQFrontEnd::setPipelining)
    const int group = argc > 9 ? atoi(argv[9]) : 1;         // ... or up to `group` blocks per launch while the device is the slower side
    g_line.assign((size_t)fftSize, 0.0f);

    // four distinct blocks of tones + FM carriers (content does not change the work)
    std::vector<std::vector<dsp::complex_t>> blocks(4, std::vector<dsp::complex_t>((size_t)block));
    for (int b = 0; b < 4; b++) {
        for (int i = 0; i < block; i++) {
            const double t = (double)(b * block + i) / sr;
            double re = 0.0, im = 0.0;
            for (int k = 0; k < nvfo; k += 4) {
                const double f = (k - (nvfo - 1) / 2.0) * 300e3;
                const double ph = 2.0 * M_PI * f * t + 50.0 * std::sin(2.0 * M_PI * 1000.0 * t);
                re += 0.05 * std::cos(ph);
                im += 0.05 * std::sin(ph);
            }
            blocks[(size_t)b][(size_t)i] = { (float)re, (float)im };
        }
    }

    dsp::stream<dsp::complex_t> src;
    sdrpp_gpu::IQFrontEnd fe;
    fe.init(&src, sr, buffered, 1, false, fftSize, sr / (double)fftSize /* dense framing */, sdrpp_gpu::IQFrontEnd::NUTTALL, acquire, release, nullptr, 0, &plans);
    fe.setPipelining(pipelined, 8, group);
    std::vector<sdrpp_gpu::RxVFO*> vfos;
    for (int k = 0; k < nvfo; k++) {
        sdrpp_gpu::RxVFO* v = fe.addVFO("vfo" + std::to_string(k), 250000.0, 150000.0, (k - (nvfo - 1) / 2.0) * 300e3);
        if (!v) { return 1; }
        v->attachDemod(sdrpp_gpu::Demod::WFM);
        vfos.push_back(v);
    }
    std::atomic<long long> audioFrames{ 0 };
    // SOAK MODE (SDRPP_BENCH_BLOCKS=N): the source hands over exactly N blocks, the run ends when every stream has received every frame of them (stop() +
    // drainPipeline() hand out what is in flight), the JSON carries the frames and a digest of every byte the sinks saw — two runs with different launch
    // grouping (whose group sizes follow the timing of the threads) must print the same digest
    const long long soakBlocks = getenv("SDRPP_BENCH_BLOCKS") ? atoll(getenv("SDRPP_BENCH_BLOCKS")) : 0;
    std::vector<uint64_t> digests((size_t)std::max(1, nvfo), 14695981039346656037ull);
    const int sinkSpinUs = getenv("SDRPP_BENCH_SINK_SPIN_US") ? atoi(getenv("SDRPP_BENCH_SINK_SPIN_US")) : 0;
    std::vector<std::thread> sinks;
    for (auto* v : vfos) {
        const int k = (int)sinks.size();
        sinks.emplace_back([v, k, soakBlocks, &digests, &audioFrames, sinkSpinUs]() {
            const void* last = nullptr;
            while (true) {
                // DIAGNOSTIC (SDRPP_BENCH_SINK_SPIN_US, default 0 = the reference's sleeping reader): the sink looks for its next block before it sleeps —
                // shows how much of a block's cycle is the wake-up of 32 sleeping sink threads (one futex wake-up per stream and block)
                if (sinkSpinUs > 0 && last) {
                    const auto t0 = std::chrono::steady_clock::now();
                    unsigned q = 0;
                    while (__atomic_load_n((void* const*)&v->audio.readBuf, __ATOMIC_RELAXED) == last) {
                        if ((++q & 63u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(sinkSpinUs)) { break; }
                    }
                }
                int n = v->audio.read();
                last = __atomic_load_n((void* const*)&v->audio.readBuf, __ATOMIC_RELAXED);
                if (n < 0) { break; }
                audioFrames += n;
                if (soakBlocks > 0) {  // (soak mode: every byte of every stream goes into a per-stream FNV-1a digest, in stream order)
                    const unsigned char* b = (const unsigned char*)v->audio.readBuf;
                    uint64_t h = digests[(size_t)k];
                    for (size_t q = 0; q < (size_t)n * sizeof(dsp::stereo_t); q++) { h = (h ^ b[q]) * 1099511628211ull; }
                    digests[(size_t)k] = h;
                }
                v->audio.flush();
            }
        });
    }
    fe.start();
    std::atomic<bool> stop{ false };
    std::atomic<long long> fed{ 0 };
    // where the source thread's time goes (two clock reads per block): filling writeBuf (the SpeedTester's memcpy — the lines of that buffer were last read
    // by OTHER cores, the front end's copy helpers) and swap() (waiting for the reader's flush() of the block before)
    std::atomic<long long> srcFillNs{ 0 }, srcSwapNs{ 0 }, srcBlocks{ 0 };
    std::thread source([&]() {
        int b = 0;
        while (!stop && (soakBlocks <= 0 || b < soakBlocks)) {
            const auto a0 = std::chrono::steady_clock::now();
            memcpy(src.writeBuf, blocks[(size_t)(b++ & 3)].data(), sizeof(dsp::complex_t) * (size_t)block);
            const auto a1 = std::chrono::steady_clock::now();
            if (!src.swap(block)) { break; }
            const auto a2 = std::chrono::steady_clock::now();
            srcFillNs += std::chrono::duration_cast<std::chrono::nanoseconds>(a1 - a0).count();
            srcSwapNs += std::chrono::duration_cast<std::chrono::nanoseconds>(a2 - a1).count();
            srcBlocks++;
            fed += block;
        }
    });
    if (soakBlocks <= 0) { std::this_thread::sleep_for(std::chrono::milliseconds(500)); }  // warm-up
    const long long f0 = soakBlocks > 0 ? 0 : (long long)fed, a0 = soakBlocks > 0 ? 0 : (long long)audioFrames, l0 = soakBlocks > 0 ? 0 : (long long)g_lines;
    const auto t0 = std::chrono::steady_clock::now();
    if (soakBlocks > 0) { while (fed.load() < soakBlocks * (long long)block) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); } }
    else { std::this_thread::sleep_for(std::chrono::duration<double>(seconds)); }
    const long long f1 = fed, a1 = audioFrames, l1 = g_lines;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop = true;
    src.stopWriter();
    source.join();
    fe.stop();
    if (soakBlocks > 0 && fe.drainPipeline() < 0) { fprintf(stderr, "drainPipeline failed\n"); }
    if (soakBlocks > 0) {  // (swap() returns when the block lies in the stream: give the sinks a moment to take the last ones)
        const long long want = soakBlocks * (long long)block / (long long)(sr / 250000.0 + 0.5) * (long long)nvfo;
        for (int w = 0; w < 2000 && audioFrames.load() < want; w++) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
    }
    for (auto* v : vfos) { v->audio.stopReader(); }
    for (auto& t : sinks) { t.join(); }
#ifdef SDRPP_GPU_BLOCKS_PROF
    fe.profReport();
#endif
    uint64_t soakDigest = 0;
    for (int q = 0; q < nvfo; q++) { soakDigest = soakDigest * 1099511628211ull + digests[(size_t)q]; }  // (per stream in order, streams in VFO order)
    // the frame buffer does not back-pressure its producer (an overrun drops a lap, like the reference's): count what came OUT
    const double processed = nvfo > 0 ? ((double)(a1 - a0) / nvfo) * (sr / 250000.0) : (double)(l1 - l0) * fftSize;
    const double nb = (double)std::max<long long>(1, srcBlocks);
    printf("{\"block\": %d, \"buffered\": %s, \"pipelined\": %s, \"blocks_per_launch_max\": %d, \"nvfo\": %d, \"msps\": %.2f, \"msps_fed\": %.2f, \"audio_frames_per_s\": %.0f, \"lines_per_s\": %.1f, \"seconds\": %.2f, "
           "\"source_us_per_block_fill\": %.1f, \"source_us_per_block_swap\": %.1f, \"soak_blocks\": %lld, \"soak_audio_frames\": %lld, \"soak_digest\": \"%016llx\"}\n", block,
           buffered ? "true" : "false", pipelined ? "true" : "false", pipelined ? group : 1, nvfo, processed / dt / 1e6, (double)(f1 - f0) / dt / 1e6, (double)(a1 - a0) / dt, (double)(l1 - l0) / dt, dt,
           (double)srcFillNs / nb / 1e3, (double)srcSwapNs / nb / 1e3, soakBlocks, soakBlocks > 0 ? (long long)audioFrames : 0ll, (unsigned long long)soakDigest);
    return 0;
}
