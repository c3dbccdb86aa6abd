// Micro-benchmark (diagnostic, not part of the product): what ONE wavefront alone on its SIMD pays per step of the reference rotator's
// phase chain  p <- p * d  (complex, float, products rounded then added), in the forms the kernel could use:
//   A  packed:      a = (pr, pr) * (dr, di);  b = (pi, pi) * (-di, dr);  p = a + b            (2 v_pk_mul_f32 + 1 v_pk_add_f32: the round-3b kernel)
//   B  plain:       four v_mul_f32, one v_sub_f32, one v_add_f32
//   C  lane split:  real part in the even lane, imaginary part in the odd lane of a pair:  m1 = p * c1;  m2 = swap(p) * c2 (DPP);  p = m1 + m2
//   D  like C, with the LDS write of every phase (ds_write_b32) in the loop
//   E  like A, with the LDS write (ds_write_b64)
//   F  like E, two steps per round, a pair of phases written one round late between the products and the sum of the next step
// Prints shader-clock cycles per step (s_memtime).     hipcc --offload-arch=gfx950 -O3 -o chain_latency_probe chain_latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(64) void probe(float* out, int iters, float dr, float di, unsigned long long* cyc) {
    __shared__ float lds[64 * 130];
    const int lane = threadIdx.x;
    float pr = 1.0f, pi = 0.0f;
    const long long c0 = __builtin_amdgcn_s_memtime();
    if constexpr (V == 0 || V == 4) {
        v2f p = { 1.0f, 0.0f };
        const v2f d = { dr, di }, dn = { -di, dr };
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 64; k++) {
                if constexpr (V == 4) { *reinterpret_cast<v2f*>(&lds[(k * 65 + lane) * 2]) = p; }
                const v2f pxx = { p.x, p.x }, pyy = { p.y, p.y };
                const v2f a = pxx * d, b = pyy * dn;
                p = a + b;
            }
            asm volatile("" : "+v"(p));
        }
        pr = p.x;
        pi = p.y;
    }
    else if constexpr (V == 5) {  // F: like E, two steps per round, the write of a pair one round late between the products and the sum
        v2f p = { 1.0f, 0.0f }, w0 = p, w1 = p;
        const v2f d = { dr, di }, dn = { -di, dr };
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 64; k += 2) {
                const v2f s0 = p;
                v2f pxx = { p.x, p.x }, pyy = { p.y, p.y };
                v2f a = pxx * d, b = pyy * dn;
                __builtin_amdgcn_sched_barrier(0);
                *reinterpret_cast<v2f*>(&lds[(((k + 62) & 63) * 65 + lane) * 2]) = w0;
                *reinterpret_cast<v2f*>(&lds[(((k + 63) & 63) * 65 + lane) * 2]) = w1;
                __builtin_amdgcn_sched_barrier(0);
                const v2f s1 = a + b;
                pxx = v2f{ s1.x, s1.x };
                pyy = v2f{ s1.y, s1.y };
                a = pxx * d;
                b = pyy * dn;
                p = a + b;
                w0 = s0;
                w1 = s1;
            }
            asm volatile("" : "+v"(p));
        }
        pr = p.x;
        pi = p.y;
    }
    else if constexpr (V == 1) {
        float a0, b0, a1, b1;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 64; k++) {
                asm volatile("v_mul_f32 %0, %4, %6\n\tv_mul_f32 %1, %5, %7\n\tv_mul_f32 %2, %4, %7\n\tv_mul_f32 %3, %5, %6" : "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1) : "v"(pr), "v"(pi), "v"(dr), "v"(di));
                asm volatile("v_sub_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5" : "=&v"(pr), "=&v"(pi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
            }
        }
    }
    else {
        float p = (lane & 1) ? 0.0f : 1.0f, m1, m2;
        const float c1 = dr, c2 = (lane & 1) ? di : -di;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 64; k++) {
                if constexpr (V == 3) { lds[(k * 65 + (lane >> 1)) * 2 + (lane & 1)] = p; }
                // (p was written by the v_add two instructions + one wait state before the DPP read: the 2 wait states gfx9 asks for)
                asm volatile("v_mul_f32 %0, %2, %3\n\ts_nop 0\n\tv_mul_f32_dpp %1, %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(m1), "=&v"(m2) : "v"(p), "v"(c1), "v"(c2));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(p) : "v"(m1), "v"(m2));
            }
        }
        pr = p;
    }
    const long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { cyc[blockIdx.x] = (unsigned long long)(c1 - c0); }
    out[blockIdx.x * 64 + lane] = pr + pi + lds[lane];
}

template <int V>
void run(const char* name) {
    float* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, 64 * 64 * 4);
    (void)hipMalloc(&cyc, 64 * 8);
    const int iters = 2000;
    const float th = 0.123f;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(probe<V>, dim3(8), dim3(64), 0, 0, out, iters, cosf(th), sinf(th), cyc); }
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime counts at the constant 100 MHz reference on gfx950?  print both the raw count per step and, with clock64, the shader cycles
    printf("%-44s %8.2f s_memtime ticks per step (x %d steps)\n", name, (double)h[0] / ((double)iters * 64), iters * 64);
    (void)hipFree(out);
    (void)hipFree(cyc);
}


// The chain inside the product kernel's structure: a workgroup of four wavefronts, wavefront 0 runs the chain for a 64-sample chunk (phases
// written pairwise, ds_write2_b64) while wavefronts 1-3 optionally read the previous chunk's phases from LDS (ROWS rows each) and optionally
// store products to global memory; one workgroup barrier per chunk.   MODE bit 0: barrier, bit 1: consumers read LDS, bit 2: consumers store
template <int MODE, int ROWS>
__global__ __launch_bounds__(256) void probe4(float* out, v2f* gout, int chunks, float dr, float di, unsigned long long* cyc) {
    __shared__ v2f ph[2][64 * 65];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2f p = { 1.0f, 0.0f };
    const v2f d = { dr, di }, dn = { -di, dr };
    float acc = 0.0f;
    const long long c0 = __builtin_amdgcn_s_memtime();
    if (wv == 0) {
        for (int c = 0; c < chunks; c++) {
            v2f* dst = &ph[c & 1][lane];
#pragma unroll
            for (int k = 0; k < 64; k++) {
                dst[k * 65] = p;
                const v2f pxx = { p.x, p.x }, pyy = { p.y, p.y };
                const v2f a = pxx * d, b = pyy * dn;
                p = a + b;
            }
            if (MODE & 1) { __syncthreads(); }
        }
    }
    else {
        for (int c = 0; c < chunks; c++) {
            if (MODE & 2) {
                const v2f* src = &ph[(c & 1) ^ 1][lane * 65];
#pragma unroll 4
                for (int r = wv - 1; r < 3 * ROWS; r += 3) {
                    const v2f q = src[r];
                    const v2f o = { q.x * 0.5f - q.y * 0.25f, q.x * 0.25f + q.y * 0.5f };
                    if (MODE & 4) { gout[((size_t)r * 4096 + (c & 63) * 64 + lane)] = o; }
                    else { acc += o.x + o.y; }
                }
            }
            if (MODE & 1) { __syncthreads(); }
        }
    }
    const long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = (unsigned long long)(c1 - c0); }
    out[blockIdx.x * 256 + threadIdx.x] = p.x + p.y + acc;
}
template <int MODE, int ROWS>
void run4(const char* name) {
    float* out;
    v2f* gout;
    unsigned long long* cyc;
    (void)hipMalloc(&out, 64 * 256 * 4);
    (void)hipMalloc(&gout, (size_t)64 * 4096 * 8);
    (void)hipMalloc(&cyc, 64 * 8);
    const int chunks = 4000;
    const float th = 0.123f;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((probe4<MODE, ROWS>), dim3(2), dim3(256), 0, 0, out, gout, chunks, cosf(th), sinf(th), cyc); }
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-74s %8.2f cycles per sample\n", name, (double)h[0] / ((double)chunks * 64));
    (void)hipFree(out);
    (void)hipFree(gout);
    (void)hipFree(cyc);
}

__global__ void clk(unsigned long long* o) {
    const long long a = __builtin_amdgcn_s_memtime();
    const long long b = clock64();
    const long long w = wall_clock64();
    for (int i = 0; i < 200000; i++) { asm volatile("s_nop 15"); }
    o[0] = (unsigned long long)(__builtin_amdgcn_s_memtime() - a);
    o[1] = (unsigned long long)(clock64() - b);
    o[2] = (unsigned long long)(wall_clock64() - w);
}

int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 32);
    hipLaunchKernelGGL(clk, dim3(1), dim3(64), 0, 0, d);
    unsigned long long h[3];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("calibration: s_memtime %llu, clock64 %llu, wall_clock64 (100 MHz) %llu -> s_memtime runs at %.1f MHz\n", h[0], h[1], h[2], 100.0 * (double)h[0] / (double)h[2]);
    run<0>("A packed (2 pk_mul + pk_add)");
    run<1>("B plain (4 mul + sub + add)");
    run<2>("C lane split (mul, mul dpp, add)");
    run<3>("D lane split + ds_write_b32 per step");
    run<4>("E packed + ds_write_b64 per step");
    run<5>("F packed, pair written one round late");
    run4<0, 5>("G four wavefronts, chain alone (pair writes), no barrier");
    run4<1, 5>("H + one workgroup barrier per chunk");
    run4<3, 5>("I + consumers read 5 rows each from LDS");
    run4<7, 5>("J + consumers store them (5 rows each)");
    run4<7, 15>("K the same with 15 rows each (43 VFOs in one workgroup)");
    return 0;
}
