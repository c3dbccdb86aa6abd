// Micro-benchmark (diagnostic, not part of the product): can FP32 matrix work and streaming global loads overlap on one CU?
// Every wavefront runs `iters` iterations of NM x v_mfma_f32_16x16x4_f32 (4 independent accumulators) + NLD x global_load_dwordx4
// (1 KiB per wavefront and load, streaming through a large buffer, each consumed eight loads later).
// Compared: matrix only, loads only, both.   hipcc --offload-arch=gfx950 -O3 -o mfma_mem_overlap_probe mfma_mem_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ACC: 0 = the compiler's choice for the accumulators (in THIS small kernel: accumulation registers a[...]; in the product kernels, whose
// accumulators are read by vector code, architectural VGPRs), 1 = pinned to a[...] (the upper half of gfx950's unified file), 2 = pinned to
// v[...] through an asm constraint — round 5's question: does it matter to the overlap of matrix work and returning loads WHICH half of the
// register file the accumulators live in (the product kernels use v[...], this probe had always measured a[...])?
template <int NM, int NLD, int ACC = 0>
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ buf, size_t nvec, float* out, int iters, float seed) {
    f32x4 acc[4];
    for (int i = 0; i < 4; i++) { acc[i] = f32x4{ 0, 0, 0, 0 }; }
    const float a = seed + threadIdx.x, b = seed - threadIdx.x;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    float sink[8];
    for (int i = 0; i < 8; i++) { sink[i] = 0.0f; }
    size_t pos = wave * 64 + lane;  // wavefront w reads vectors [w * 64, w * 64 + 64), then strides by all wavefronts
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int l = 0; l < NLD; l++) {
            const float4 v = buf[pos];  // consumed eight loads later: eight 1 KiB loads in flight per wavefront
            sink[(it * NLD + l) & 7] += v.x + v.w;
            pos += nwaves * 64;
            if (pos >= nvec) { pos -= nvec; }
        }
#pragma unroll
        for (int m = 0; m < NM; m++) {
            if constexpr (ACC == 1) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b)); }
            else if constexpr (ACC == 2) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b)); }
            else { acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0); }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 4; i++) { s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; }
    for (int i = 0; i < 8; i++) { s += sink[i]; }
    if (s == 12345.678f) { out[threadIdx.x] = s; }
}

template <int NM, int NLD, int ACC = 0>
double run(const char* name, const float4* d_buf, size_t nvec, float* d_out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid(256 * 4);  // 4 wavefronts per SIMD
    hipLaunchKernelGGL((probe<NM, NLD, ACC>), grid, dim3(256), 0, 0, d_buf, nvec, d_out, 50, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NM, NLD, ACC>), grid, dim3(256), 0, 0, d_buf, nvec, d_out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)iters * NLD * 1024.0 * 4096.0, flops = (double)iters * NM * 2048.0 * 4096.0;
    printf("%-28s %8.3f ms   %7.2f TB/s   %7.1f TFLOP/s (of 157.3)\n", name, ms, bytes / ms * 1e-9, flops / ms * 1e-9);
    return ms;
}

int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB: far larger than the 256 MiB Infinity Cache
    float4* d_buf;
    float* d_out;
    if (hipMalloc(&d_buf, bytes) != hipSuccess || hipMemset(d_buf, 0, bytes) != hipSuccess || hipMalloc(&d_out, 4096) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    const size_t nvec = bytes / 16;
    const int iters = 4000;
    run<16, 0>("16 mfma", d_buf, nvec, d_out, iters);
    run<0, 1>("1 load", d_buf, nvec, d_out, iters);
    run<16, 1>("16 mfma + 1 load", d_buf, nvec, d_out, iters);
    run<0, 2>("2 loads", d_buf, nvec, d_out, iters);
    run<16, 2>("16 mfma + 2 loads", d_buf, nvec, d_out, iters);
    run<0, 4>("4 loads", d_buf, nvec, d_out, iters);
    run<16, 4>("16 mfma + 4 loads", d_buf, nvec, d_out, iters);
    run<8, 4>("8 mfma + 4 loads", d_buf, nvec, d_out, iters);
    run<32, 4>("32 mfma + 4 loads", d_buf, nvec, d_out, iters);
    printf("-- accumulators in AGPRs --\n");
    run<16, 0, 1>("16 mfma (agpr)", d_buf, nvec, d_out, iters);
    run<16, 1, 1>("16 mfma (agpr) + 1 load", d_buf, nvec, d_out, iters);
    run<16, 2, 1>("16 mfma (agpr) + 2 loads", d_buf, nvec, d_out, iters);
    run<16, 4, 1>("16 mfma (agpr) + 4 loads", d_buf, nvec, d_out, iters);
    run<32, 4, 1>("32 mfma (agpr) + 4 loads", d_buf, nvec, d_out, iters);
    printf("-- accumulators in architectural VGPRs (what the product kernels use) --\n");
    run<16, 0, 2>("16 mfma (vgpr)", d_buf, nvec, d_out, iters);
    run<16, 1, 2>("16 mfma (vgpr) + 1 load", d_buf, nvec, d_out, iters);
    run<16, 2, 2>("16 mfma (vgpr) + 2 loads", d_buf, nvec, d_out, iters);
    run<16, 4, 2>("16 mfma (vgpr) + 4 loads", d_buf, nvec, d_out, iters);
    run<32, 4, 2>("32 mfma (vgpr) + 4 loads", d_buf, nvec, d_out, iters);
    return 0;
}
