// Stand-alone probe: can two kernels enqueued on ONE stream overlap when the second is launched with hipExtAnyOrderLaunch (no barrier
// bit in its dispatch packet)?  And does an ordinary launch behind them still wait for both?  (hip_ext.h says the flag is "not supported
// on AMD GFX9xx boards" for the module-launch API; this measures what the runtime of this image really does on gfx950.)
// Build: hipcc --offload-arch=gfx950 -O2 -o anyorder_probe anyorder_probe.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// spins for about `ticks` of the 100 MHz wall clock, then leaves a stamp
__global__ __launch_bounds__(256) void spin_kernel(unsigned long long* stamps, int slot, long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[2 * slot] = t0;
        stamps[2 * slot + 1] = wall_clock64();
    }
}
// checks that the two kernels in front of it have both finished (their stamps are there)
__global__ void check_kernel(const unsigned long long* stamps, int* ok) {
    *ok = (stamps[1] != 0 && stamps[3] != 0) ? 1 : 0;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned long long* stamps;
    int* ok;
    CK(hipMalloc(&stamps, 64));
    CK(hipMalloc(&ok, 4));
    unsigned long long h[4];
    int hok = 0;
    const long long ticks = 5000;  // 50 us
    for (int mode = 0; mode < 3; mode++) {
        // 0: two ordinary launches; 1: second with hipExtAnyOrderLaunch; 2: both with the flag
        double best = 1e30;
        double overlap = 0;
        int oks = 0;
        for (int rep = 0; rep < 20; rep++) {
            CK(hipMemsetAsync(stamps, 0, 64, s));
            CK(hipStreamSynchronize(s));
            const double t0 = now_us();
            if (mode == 2) { hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, stamps, 0, ticks); }
            else { hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, stamps, 0, ticks); }
            if (mode >= 1) { hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, stamps, 1, ticks); }
            else { hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, stamps, 1, ticks); }
            hipLaunchKernelGGL(check_kernel, dim3(1), dim3(1), 0, s, stamps, ok);
            CK(hipStreamSynchronize(s));
            const double t1 = now_us();
            CK(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&hok, ok, 4, hipMemcpyDeviceToHost));
            oks += hok;
            if (t1 - t0 < best) {
                best = t1 - t0;
                overlap = ((double)h[1] - (double)h[2]) / 100.0;  // end of the first minus start of the second: > 0 means they overlapped
            }
        }
        printf("mode %d (%s): best wall %.1f us for two 50 us kernels + check; first end - second start = %+.1f us; check saw both done in %d/20\n", mode,
               mode == 0 ? "ordinary, ordinary" : (mode == 1 ? "ordinary, any-order" : "any-order, any-order"), best, overlap, oks);
    }
    // host cost of the extended launch
    for (int mode = 0; mode < 2; mode++) {
        CK(hipStreamSynchronize(s));
        const int n = 2000;
        const double t0 = now_us();
        for (int i = 0; i < n; i++) {
            if (mode) { hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, stamps, 2, 0LL); }
            else { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, stamps, 2, 0LL); }
        }
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        const double t2 = now_us();
        printf("%s: host %.2f us per launch, total %.2f us per launch\n", mode ? "hipExtLaunchKernelGGL(any-order)" : "hipLaunchKernelGGL", (t1 - t0) / n, (t2 - t0) / n);
    }
    return 0;
}
