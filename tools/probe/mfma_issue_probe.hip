// Micro-benchmark (diagnostic, not part of the product): how much FP32 matrix-core issue rate survives when VALU / LDS / SALU work is
// interleaved with v_mfma_f32_16x16x4_f32, for 1..4 wavefronts per SIMD.  Prints SIMD cycles per loop iteration (4 MFMAs = 128 cycles
// when the matrix pipe is the only limit).   hipcc --offload-arch=gfx950 -O3 -o mfma_issue_probe mfma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, int NM, int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    lds[lane] = seed * lane;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; i++) { acc[i] = f32x4{ 0, 0, 0, 0 }; }
    float a = seed + lane, b = seed - lane;
    float v[8];
    for (int i = 0; i < 8; i++) { v[i] = seed * i; }
    float l[8];
    for (int i = 0; i < 8; i++) { l[i] = 0.0f; }
    for (int it = 0; it < iters; it++) {
        constexpr int SL = NM ? NM : 1;  // slots per iteration
#pragma unroll
        for (int m = 0; m < SL; m++) {
            if constexpr (NM > 0) { acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0); }
            if constexpr (MODE == 2) { asm volatile("s_nop 7"); }
            if constexpr (MODE == 3) { asm volatile("s_sleep 1"); }
            if constexpr (MODE == 4) { if ((m & 1) == 1) { asm volatile("s_sleep 1"); } }
            // NV VALU and NL LDS instructions spread behind the matrix instructions
#pragma unroll
            for (int k = 0; k < (NV + SL - 1 - m) / SL; k++) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[(m + k) & 7]) : "v"(a), "v"(b)); }
#pragma unroll
            for (int k = 0; k < (NL + SL - 1 - m) / SL; k++) { asm volatile("ds_read_b32 %0, %1" : "=v"(l[(m + k) & 7]) : "v"((lane & 63) << 2)); }
        }
        if (NL && MODE != 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    }
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - w0;
    }
    float s = 0.0f;
    for (int i = 0; i < 4; i++) { s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; }
    for (int i = 0; i < 8; i++) { s += v[i] + l[i]; }
    if (s == 12345.678f) { out[lane] = s; }
}

template <int NV, int NL, int NM, int MODE = 0>
void run(const char* name, float* d_out, int waves_per_simd) {
    static long long* d_clk = nullptr;
    if (!d_clk) { hipMalloc(&d_clk, 16); }
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 160 * 1024 / waves_per_simd - 512;  // exactly `waves_per_simd` blocks of 4 wavefronts per CU
    hipFuncSetAttribute((const void*)probe<NV, NL, NM, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 grid(256 * waves_per_simd);
    hipLaunchKernelGGL((probe<NV, NL, NM, MODE>), grid, dim3(256), lds, 0, d_out, 100, 1.0f, (long long*)nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NV, NL, NM, MODE>), grid, dim3(256), lds, 0, d_out, iters, 1.0f, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = (double)ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd);
    long long h[2] = { 0, 0 };
    hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost);
    int wfreq = 0;
    hipDeviceGetAttribute(&wfreq, hipDeviceAttributeWallClockRate, 0);  // kHz
    const double secs = (double)h[1] / ((double)wfreq * 1e3);
    printf("[%4.0f MHz, %6.1f shader cycles per iteration] ", (double)h[0] / secs * 1e-6, (double)h[0] / iters);
    printf("%-34s waves/SIMD %d  %8.1f SIMD-cycles per iteration and wave (matrix pipe alone: %d)  util %.2f\n", name, waves_per_simd, cyc, NM * 32, NM * 32.0 / cyc);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 4096);
    for (int w : { 1, 4 }) {
        run<0, 0, 4>("4 mfma", d_out, w);
        run<0, 0, 4, 2>("4 x (mfma, s_nop 7)", d_out, w);
        run<0, 0, 4, 3>("4 x (mfma, s_sleep 1)", d_out, w);
        run<0, 0, 4, 4>("2 x (mfma, mfma, s_sleep 1)", d_out, w);
        run<0, 2, 2>("2 mfma + 2 ds_read + wait", d_out, w);
        run<0, 4, 4>("4 mfma + 4 ds_read + wait", d_out, w);
        run<0, 3, 8>("8 mfma + 3 ds_read + wait", d_out, w);
        run<0, 6, 16>("16 mfma + 6 ds_read + wait", d_out, w);
        run<0, 4, 4, 1>("4 mfma + 4 ds_read, no wait", d_out, w);
        run<0, 3, 8, 1>("8 mfma + 3 ds_read, no wait", d_out, w);
        run<8, 3, 8>("8 mfma + 8 v_fma + 3 ds_read + wait", d_out, w);
        run<16, 3, 8>("8 mfma + 16 v_fma + 3 ds_read + wait", d_out, w);
    }
    return 0;
}
