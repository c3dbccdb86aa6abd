// Micro-benchmark (diagnostic, not part of the product): what a v_mfma_f32_16x16x4_f32 / 32x32x2 costs ONE wavefront per SIMD (and four)
// depending on where its operands come from: constant registers, rotating registers, registers an LDS read has just written (the shape of
// the Toeplitz / front-end loops: operands of the next batch are read while this batch's matrix instructions issue).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_operand_probe mfma_operand_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: 16 x 16x16x4, operands from 8 + 8 constant registers (rotating), no LDS
// MODE 1: + 6 ds_read_b64 per 16 matrix instructions, results unused
// MODE 2: operands of iteration i + 1 are the results of the reads issued during iteration i (software pipeline, 12 ds_read_b32 + ... as the kernel)
// MODE 3: as 2, but the reads all issue BEFORE the matrix instructions of the iteration (the kernel's old order)
// MODE 4: 8 x 32x32x2 with constant operands; MODE 5: 32x32x2 with operands from LDS reads one iteration ahead
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) { lds[i] = seed * (float)(i & 31); }
    __syncthreads();
    const float* p = lds + lane * 2;
    float s = 0.0f;
    if constexpr (MODE <= 3) {
        f32x4 acc[4];
        for (int i = 0; i < 4; i++) { acc[i] = f32x4{ 0, 0, 0, 0 }; }
        float a[8], b[4], na[8], nb[4];
        for (int i = 0; i < 8; i++) { a[i] = seed + i + lane; na[i] = a[i]; }
        for (int i = 0; i < 4; i++) { b[i] = seed - i; nb[i] = b[i]; }
        for (int it = 0; it < iters; it++) {
            if constexpr (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; i++) { na[i] = p[(i * 128 + (it & 3) * 16) & 2047]; }
#pragma unroll
                for (int i = 0; i < 4; i++) { nb[i] = p[(i * 192 + 64) & 2047]; }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; i++) { na[i] = p[(i * 128 + (it & 3) * 16) & 2047]; }
#pragma unroll
                for (int i = 0; i < 4; i++) { nb[i] = p[(i * 192 + 64) & 2047]; }
            }
            float unused[6];
            if constexpr (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 6; i++) { unused[i] = p[(i * 160 + (it & 7) * 8) & 2047]; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * u], b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * u + 1], b[u], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * u] , b[(u + 1) & 3], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * u + 1], b[(u + 1) & 3], acc[3], 0, 0, 0);
            }
            if constexpr (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            if constexpr (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                for (int i = 0; i < 6; i++) { s += unused[i] * 1e-30f; }
            }
            if constexpr (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 8; i++) { a[i] = na[i]; }
#pragma unroll
                for (int i = 0; i < 4; i++) { b[i] = nb[i]; }
            }
        }
        for (int i = 0; i < 4; i++) { s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; }
    }
    else {
        f32x16 accR, accI;
        for (int i = 0; i < 16; i++) { accR[i] = 0.0f; accI[i] = 0.0f; }
        float a[4], br[4], bi[4], na[4], nbr[4], nbi[4];
        for (int i = 0; i < 4; i++) { a[i] = seed + i; br[i] = seed - i + lane; bi[i] = seed * i; na[i] = a[i]; nbr[i] = br[i]; nbi[i] = bi[i]; }
        for (int it = 0; it < iters; it++) {
            if constexpr (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    na[i] = p[(i * 128 + 32) & 2047];
                    nbr[i] = p[(i * 96 + (it & 3) * 16) & 2047];
                    nbi[i] = p[(i * 96 + 1024 + (it & 3) * 16) & 2047];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                accR = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], br[u], accR, 0, 0, 0);
                accI = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bi[u], accI, 0, 0, 0);
            }
            if constexpr (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) { a[i] = na[i]; br[i] = nbr[i]; bi[i] = nbi[i]; }
            }
        }
        for (int i = 0; i < 16; i++) { s += accR[i] + accI[i]; }
    }
    if (s == 12345.678f) { out[threadIdx.x] = s; }
}

template <int MODE>
void run(const char* name, float* d_out, int waves_per_simd) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 160 * 1024 / waves_per_simd - 512;
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 grid(256 * waves_per_simd);
    hipLaunchKernelGGL((probe<MODE>), grid, dim3(256), lds, 0, d_out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE>), grid, dim3(256), lds, 0, d_out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int nm = MODE <= 3 ? 16 : 8, cyc1 = MODE <= 3 ? 32 : 64;
    const double cyc = (double)ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd * nm);
    printf("%-78s waves/SIMD %d  %6.1f SIMD-cycles (at 2.4 GHz) per matrix instruction and wave (pipe alone: %d)  util %.2f\n", name, waves_per_simd, cyc, cyc1, cyc1 / cyc);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 4096);
    for (int w : { 1, 2, 4 }) {
        run<0>("16x16x4: 16 back to back, rotating constant operands", d_out, w);
        run<1>("16x16x4: + 6 LDS reads between them, results unused", d_out, w);
        run<2>("16x16x4: operands = LDS reads issued between the previous batch's instructions", d_out, w);
        run<3>("16x16x4: operands = LDS reads issued in front of the batch (one batch ahead)", d_out, w);
        run<4>("32x32x2: 8 back to back, constant operands", d_out, w);
        run<5>("32x32x2: operands = LDS reads issued between the previous batch's instructions", d_out, w);
    }
    return 0;
}
