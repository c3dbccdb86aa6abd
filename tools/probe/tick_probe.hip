// Stand-alone micro-benchmark behind the per-block ("tick") execution path (DESIGN_HISTORY.md section 6c): what does the runtime /
// hardware charge for (A) a chain of small dependent launches on one stream, (B) a kernel that fetches its input from page-locked
// host memory itself / writes results there, (C) completion signalled through a flag in host memory instead of an event wait,
// (D) the same chain as a hipGraph with a fork / join.  Build: hipcc --offload-arch=gfx950 -O2 -o tick_probe tick_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct BigArgs { int v[768]; };  // 3 KB of kernel arguments

__global__ __launch_bounds__(256) void work_kernel(float* buf, int iters) {
    float a = (float)threadIdx.x, b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = fmaf(a, b, 0.5f); }
    if (a == 12345.678f) { buf[blockIdx.x] = a; }
}
__global__ __launch_bounds__(256) void work_big_kernel(float* buf, int iters, BigArgs args) {
    float a = (float)threadIdx.x + (float)args.v[threadIdx.x], b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = fmaf(a, b, 0.5f); }
    if (a == 12345.678f) { buf[blockIdx.x] = a; }
}
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n16) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) { dst[i] = src[i]; }
}
// last block to finish publishes `seq` in host memory
__global__ __launch_bounds__(256) void flag_kernel(float* buf, int iters, unsigned* counter, volatile unsigned* host_flag, unsigned seq) {
    float a = (float)threadIdx.x, b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = fmaf(a, b, 0.5f); }
    if (a == 12345.678f) { buf[blockIdx.x] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned done = atomicAdd(counter, 1u) + 1u;
        if (done == gridDim.x * seq) {
            __threadfence_system();
            *host_flag = seq;
        }
    }
}

int main() {
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    if (ndev == 0) { printf("no device\n"); return 1; }
    CK(hipSetDevice(0));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float* d_buf;
    CK(hipMalloc(&d_buf, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    BigArgs big;
    memset(&big, 0, sizeof(big));

    // ---- A: chain of dependent launches on one stream ----
    printf("A: chains of N=2000 launches on one stream (us per launch: host enqueue / device incl. gaps)\n");
    for (int blocks : { 1, 128, 1024 }) {
        for (int iters : { 0, 2000, 8000 }) {
            for (int bigargs = 0; bigargs < 2; bigargs++) {
                const int N = 2000;
                for (int w = 0; w < 20; w++) { hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, s, d_buf, iters); }
                CK(hipStreamSynchronize(s));
                const double t0 = now_us();
                for (int i = 0; i < N; i++) {
                    if (bigargs) { hipLaunchKernelGGL(work_big_kernel, dim3(blocks), dim3(256), 0, s, d_buf, iters, big); }
                    else { hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, s, d_buf, iters); }
                }
                const double t1 = now_us();
                CK(hipStreamSynchronize(s));
                const double t2 = now_us();
                // one launch alone (event timed)
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, s, d_buf, iters);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("  blocks %4d iters %5d kernarg %4s: host %6.2f  total %6.2f  (one launch between events %6.2f)\n", blocks, iters, bigargs ? "3KB" : "16B", (t1 - t0) / N, (t2 - t0) / N,
                       ms * 1000.0);
            }
        }
    }

    // ---- B: kernel-side PCIe copies ----
    printf("B: copy kernel pinned host <-> device (GB/s; us)\n");
    const size_t maxb = 16u << 20;
    void *h_src, *h_dst, *d_land;
    CK(hipHostMalloc(&h_src, maxb, hipHostMallocMapped));
    CK(hipHostMalloc(&h_dst, maxb, hipHostMallocMapped));
    CK(hipMalloc(&d_land, maxb));
    memset(h_src, 1, maxb);
    void *hd_src, *hd_dst;
    CK(hipHostGetDevicePointer(&hd_src, h_src, 0));
    CK(hipHostGetDevicePointer(&hd_dst, h_dst, 0));
    for (size_t bytes : { (size_t)96000, (size_t)400000, (size_t)2457600, (size_t)8000000 }) {
        for (int blocks : { 4, 16, 64, 256 }) {
            for (int dir = 0; dir < 3; dir++) {  // 0: host -> device by kernel, 1: device -> host by kernel, 2: hipMemcpyAsync H2D
                const int R = 20;
                float best = 1e9f, sum = 0;
                for (int r = 0; r < R + 3; r++) {
                    CK(hipEventRecord(e0, s));
                    if (dir == 0) { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)hd_src, (uint4*)d_land, (long long)(bytes / 16)); }
                    else if (dir == 1) { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)d_land, (uint4*)hd_dst, (long long)(bytes / 16)); }
                    else { CK(hipMemcpyAsync(d_land, h_src, bytes, hipMemcpyHostToDevice, s)); }
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
                }
                if (dir == 2 && blocks != 4) { continue; }
                printf("  %8zu B  blocks %3d  %s: avg %7.2f us  best %7.2f us  (%6.1f GB/s best)\n", bytes, blocks, dir == 0 ? "kernel H->D" : (dir == 1 ? "kernel D->H" : "memcpyAsync H->D"),
                       sum / R * 1000.0, best * 1000.0, (double)bytes / (best * 1e-3) * 1e-9);
            }
        }
    }
    // both directions at once inside ONE launch is what a tick does: approximate with two kernels on two streams
    {
        const size_t bytes = 400000;
        const int R = 50;
        const double t0 = now_us();
        for (int r = 0; r < R; r++) {
            hipLaunchKernelGGL(copy_kernel, dim3(32), dim3(256), 0, s, (const uint4*)hd_src, (uint4*)d_land, (long long)(bytes / 16));
            hipLaunchKernelGGL(copy_kernel, dim3(32), dim3(256), 0, s2, (const uint4*)d_buf, (uint4*)hd_dst, (long long)(320000 / 16));
        }
        CK(hipStreamSynchronize(s));
        CK(hipStreamSynchronize(s2));
        printf("  400 KB in + 320 KB out concurrently on two streams: %.2f us per pair\n", (now_us() - t0) / R);
    }

    // ---- C: completion through a flag in host memory ----
    printf("C: completion latency, launch -> host knows (us): flag in mapped host memory vs hipStreamSynchronize vs hipEventSynchronize\n");
    unsigned* d_counter;
    CK(hipMalloc(&d_counter, 4));
    volatile unsigned* h_flag;
    CK(hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped));
    void* hd_flag;
    CK(hipHostGetDevicePointer(&hd_flag, (void*)h_flag, 0));
    for (int blocks : { 1, 128 }) {
        for (int iters : { 0, 4000 }) {
            double acc[3] = { 0, 0, 0 };
            const int R = 200;
            for (int mode = 0; mode < 3; mode++) {
                CK(hipMemset(d_counter, 0, 4));
                *h_flag = 0;
                CK(hipDeviceSynchronize());
                for (int r = 1; r <= R; r++) {
                    const double t0 = now_us();
                    hipLaunchKernelGGL(flag_kernel, dim3(blocks), dim3(256), 0, s, d_buf, iters, d_counter, (volatile unsigned*)hd_flag, (unsigned)r);
                    if (mode == 0) {
                        while (*h_flag != (unsigned)r) { }
                    }
                    else if (mode == 1) { CK(hipStreamSynchronize(s)); }
                    else {
                        CK(hipEventRecord(e1, s));
                        CK(hipEventSynchronize(e1));
                    }
                    acc[mode] += now_us() - t0;
                    if (mode == 0) { CK(hipStreamSynchronize(s)); }
                }
            }
            printf("  blocks %3d iters %4d: flag %6.2f  streamSync %6.2f  eventSync %6.2f\n", blocks, iters, acc[0] / R, acc[1] / R, acc[2] / R);
        }
    }
    // pipelined: enqueue launches back to back, a second thread-less consumer polls the flag: tick rate with completion flags
    {
        const int N = 2000, blocks = 128, iters = 2000;
        CK(hipMemset(d_counter, 0, 4));
        *h_flag = 0;
        CK(hipDeviceSynchronize());
        const double t0 = now_us();
        for (int r = 1; r <= N; r++) {
            hipLaunchKernelGGL(flag_kernel, dim3(blocks), dim3(256), 0, s, d_buf, iters, d_counter, (volatile unsigned*)hd_flag, (unsigned)r);
            if (r > 4) {
                while (*h_flag < (unsigned)(r - 4)) { }  // never more than 4 in flight
            }
        }
        while (*h_flag != (unsigned)N) { }
        printf("  pipelined ticks (<= 4 in flight, completion by flag): %.2f us per tick\n", (now_us() - t0) / N);
        CK(hipStreamSynchronize(s));
    }

    // ---- D: the same chain as a hipGraph (6 kernels on the main branch, 3 on a forked branch) ----
    printf("D: hipGraph of 6 + 3 small kernels with a fork/join vs the same launched directly (us per iteration)\n");
    for (int blocks : { 32, 256 }) {
        for (int iters : { 0, 2000 }) {
            hipEvent_t ef, ej;
            CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
            auto enqueue = [&]() {
                CK(hipEventRecord(ef, s));
                CK(hipStreamWaitEvent(s2, ef, 0));
                for (int k = 0; k < 3; k++) { hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, s2, d_buf, iters); }
                CK(hipEventRecord(ej, s2));
                for (int k = 0; k < 6; k++) { hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, s, d_buf, iters); }
                CK(hipStreamWaitEvent(s, ej, 0));
            };
            const int N = 500;
            for (int w = 0; w < 10; w++) { enqueue(); }
            CK(hipStreamSynchronize(s));
            double t0 = now_us();
            for (int i = 0; i < N; i++) { enqueue(); }
            double t1 = now_us();
            CK(hipStreamSynchronize(s));
            double t2 = now_us();
            printf("  blocks %3d iters %4d direct: host %6.2f total %6.2f", blocks, iters, (t1 - t0) / N, (t2 - t0) / N);
            hipGraph_t graph;
            hipGraphExec_t exec;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            enqueue();
            CK(hipStreamEndCapture(s, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            for (int w = 0; w < 10; w++) { CK(hipGraphLaunch(exec, s)); }
            CK(hipStreamSynchronize(s));
            t0 = now_us();
            for (int i = 0; i < N; i++) { CK(hipGraphLaunch(exec, s)); }
            t1 = now_us();
            CK(hipStreamSynchronize(s));
            t2 = now_us();
            printf("   graph: host %6.2f total %6.2f\n", (t1 - t0) / N, (t2 - t0) / N);
            // single-stream chain of 9 as a graph
            hipGraph_t g2;
            hipGraphExec_t x2;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < 9; k++) { hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, s, d_buf, iters); }
            CK(hipStreamEndCapture(s, &g2));
            CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
            for (int w = 0; w < 10; w++) { CK(hipGraphLaunch(x2, s)); }
            CK(hipStreamSynchronize(s));
            t0 = now_us();
            for (int i = 0; i < N; i++) { CK(hipGraphLaunch(x2, s)); }
            t1 = now_us();
            CK(hipStreamSynchronize(s));
            t2 = now_us();
            printf("      linear chain of 9 as a graph: host %6.2f total %6.2f\n", (t1 - t0) / N, (t2 - t0) / N);
            CK(hipGraphExecDestroy(exec));
            CK(hipGraphDestroy(graph));
            CK(hipGraphExecDestroy(x2));
            CK(hipGraphDestroy(g2));
        }
    }
    printf("done\n");
    return 0;
}
