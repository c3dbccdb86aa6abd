// Micro-benchmark (diagnostic, not part of the product): what a VECTOR-instruction wavefront sees on a SIMD whose other wavefronts sit
// in matrix loops, and what the matrix wavefronts lose to it.  Workgroups of four wavefronts (one per SIMD); every fourth workgroup
// runs a chain of v_fma_f32, the others loops of v_mfma_f32_16x16x4_f32 with nothing / s_nop N between the matrix instructions.
// Prints, per variant: shader cycles per vector instruction of the VALU wavefronts and per matrix instruction of the others.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix_probe mfma_valu_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int GAP, int VALU_EVERY>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed, unsigned long long* acc_cycles) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const bool valu_block = VALU_EVERY > 0 && (blockIdx.x % VALU_EVERY) == 0;
    float a = seed + lane, b = seed - lane, s = 0.0f;
    const long long c0 = clock64();
    if (valu_block) {
        float v0 = seed, v1 = seed * 2, v2 = seed * 3, v3 = seed * 4;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {  // 32 vector instructions, four independent chains
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v3) : "v"(a), "v"(b));
            }
        }
        s = v0 + v1 + v2 + v3;
        if (lane == 0) { atomicAdd(&acc_cycles[0], (unsigned long long)(clock64() - c0)); atomicAdd(&acc_cycles[1], 1ull); }
    }
    else {
        f32x4 acc[4];
        for (int i = 0; i < 4; i++) { acc[i] = f32x4{ 0, 0, 0, 0 }; }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
                if constexpr (GAP == 1) { asm volatile("s_nop 0"); }
                if constexpr (GAP == 2) { asm volatile("s_nop 3"); }
                if constexpr (GAP == 3) { asm volatile("s_nop 7"); }
                if constexpr (GAP == 4) { asm volatile("s_nop 15"); }
                if constexpr (GAP == 5) { asm volatile("s_sleep 1"); }
            }
        }
        for (int i = 0; i < 4; i++) { s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; }
        if (lane == 0) { atomicAdd(&acc_cycles[2], (unsigned long long)(clock64() - c0)); atomicAdd(&acc_cycles[3], 1ull); }
    }
    if (s == 12345.678f) { out[lane] = s + lds[lane]; }
}

template <int GAP, int VALU_EVERY>
void run(const char* name, float* d_out, unsigned long long* d_acc, int blocks_per_cu) {
    const int iters_m = 4000, iters_v = 4000;
    (void)iters_v;
    const size_t lds = 160 * 1024 / blocks_per_cu - 512;
    (void)hipFuncSetAttribute((const void*)probe<GAP, VALU_EVERY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipMemset(d_acc, 0, 32);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<GAP, VALU_EVERY>), dim3(256 * blocks_per_cu), dim3(256), lds, 0, d_out, iters_m, 1.0f, d_acc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4];
    (void)hipMemcpy(h, d_acc, 32, hipMemcpyDeviceToHost);
    const double valu = h[1] ? (double)h[0] / (double)h[1] / (iters_m * 32.0) : 0.0;
    const double mfma = h[3] ? (double)h[2] / (double)h[3] / (iters_m * 8.0) : 0.0;
    printf("%-44s %d wg/CU: %7.1f cycles per vector instruction | %6.1f cycles per v_mfma per wavefront | %.3f ms\n", name, blocks_per_cu, valu, mfma, ms);
}

int main() {
    float* d_out;
    unsigned long long* d_acc;
    (void)hipMalloc(&d_out, 4096);
    (void)hipMalloc(&d_acc, 64);
    for (int bpc : { 4, 5 }) {
        run<0, 0>("matrix only, back to back", d_out, d_acc, bpc);
        run<3, 0>("matrix only, s_nop 7 between", d_out, d_acc, bpc);
        run<0, 4>("mixed, matrix back to back", d_out, d_acc, bpc);
        run<1, 4>("mixed, s_nop 0 between", d_out, d_acc, bpc);
        run<2, 4>("mixed, s_nop 3 between", d_out, d_acc, bpc);
        run<3, 4>("mixed, s_nop 7 between", d_out, d_acc, bpc);
        run<4, 4>("mixed, s_nop 15 between", d_out, d_acc, bpc);
        run<5, 4>("mixed, s_sleep 1 between", d_out, d_acc, bpc);
    }
    return 0;
}
