// TEST-EMULATOR version of <sdrpp_gfx950.h>: same interface, plain memory reads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace sdrpp_k {
struct UniformF32 {
    const float* p;
    float operator[](int i) const { return p[i]; }
};
static inline UniformF32 as_uniform(const void* ptr) { return UniformF32{ (const float*)ptr }; }
struct UniformI32 {
    const int* p;
    int operator[](int i) const { return p[i]; }
};
static inline UniformI32 as_uniform_i32(const void* ptr) { return UniformI32{ (const int*)ptr }; }
static inline int wave_uniform(int v) { return v; }
static inline float global_load_f32(const float* p, long long i) { return p[i]; }
static inline int global_load_i32(const int* p, long long i) { return p[i]; }
static inline float2 global_load_f32x2(const float2* p, long long i) { return p[i]; }
static inline float4 global_load_f32x4(const float4* p, long long i) { return p[i]; }
static inline float4 global_load_f32x4_unaligned(const float* p, long long i) { return make_float4(p[i], p[i + 1], p[i + 2], p[i + 3]); }
static inline void global_store_f32x2(float2* p, long long i, float2 v) { p[i] = v; }
static inline float2 global_load_f32x2_boff(const void* base, unsigned byte_off) { return *(const float2*)((const char*)base + byte_off); }
static inline void global_store_f32x2_boff(void* base, unsigned byte_off, float2 v) { *(float2*)((char*)base + byte_off) = v; }
static inline void global_store_f32_boff(void* base, unsigned byte_off, float v) { *(float*)((char*)base + byte_off) = v; }
static inline void global_store_f32x4(float* p, long long i, float4 v) { p[i] = v.x; p[i + 1] = v.y; p[i + 2] = v.z; p[i + 3] = v.w; }
static inline void global_store_f32x4_unaligned(float* p, long long i, float4 v) { p[i] = v.x; p[i + 1] = v.y; p[i + 2] = v.z; p[i + 3] = v.w; }
static inline uint4 global_load_u32x4(const void* p, long long i) { return ((const uint4*)p)[i]; }
static inline void global_store_u32x4(void* p, long long i, uint4 v) { ((uint4*)p)[i] = v; }
static inline unsigned global_load_u32(const void* p, long long i) { return ((const unsigned*)p)[i]; }
static inline void global_store_u32(void* p, long long i, unsigned v) { ((unsigned*)p)[i] = v; }
static inline float fast_rcp(float x) { return 1.0f / x; }
static inline void sched_fence() {}
static inline int opaque(int v) {
    asm volatile("" : "+r"(v));
    return v;
}
#define TICK_MARK(k) do { } while (0)
static inline void wave_stores_done() {}
static inline int wave_upper_bound(const int* ends, int n, int b, int* prev_end) {
    int idx = 0;
    while (idx < n && ends[idx] <= b) { idx++; }
    *prev_end = idx > 0 ? ends[idx - 1] : 0;
    return idx;
}
static inline int wave_upper_bound64_rec(const int* ends, int n, int b, int* prev_end, const void* recs, int stride_bytes, int (&rec)[6]) {
    const int idx = wave_upper_bound(ends, n, b, prev_end);
    const int* r = reinterpret_cast<const int*>(reinterpret_cast<const char*>(recs) + (size_t)(idx < 64 ? idx : 63) * (size_t)stride_bytes);
    for (int k = 0; k < 6; k++) { rec[k] = r[k]; }
    return idx;
}
static inline void lds_flag_set(int* f, int v) { *(volatile int*)f = v; }
static inline void lds_flag_wait_ge(int* f, int need, int* /*timeouts*/) { while (*(volatile int*)f < need) { hipemu::yield(); } }  // the other wavefronts' fibers run meanwhile
static inline void wave_prio_high() {}
static inline void wave_prio_low() {}
static inline float wave_bcast(float v, int src) { return hipemu::wave_exchange(v, 0.0f)[src * 2]; }
static inline float wave_shr1(float v, float first) {
    const float* ab = hipemu::wave_exchange(v, 0.0f);
    const int l = hipemu::lane_id();
    return l == 0 ? first : ab[(l - 1) * 2];
}
static inline int wave_first(bool pred) {
    const float* ab = hipemu::wave_exchange(pred ? 1.0f : 0.0f, 0.0f);
    for (int l = 0; l < 64; l++) {
        if (ab[l * 2] != 0.0f) { return l; }
    }
    return 64;
}
static inline float wave_sum(float v) {  // same butterfly order as the device's shfl_xor reduction
    float cur = v;
    for (int d = 32; d >= 1; d >>= 1) {
        const float* ab = hipemu::wave_exchange(cur, 0.0f);
        cur = cur + ab[(hipemu::lane_id() ^ d) * 2];
    }
    return cur;
}
static inline float wave_max(float v) {
    const float* ab = hipemu::wave_exchange(v, 0.0f);
    float m = ab[0];
    for (int l = 1; l < 64; l++) { m = (ab[l * 2] > m) ? ab[l * 2] : m; }
    return m;
}
static inline void wave_sync() { hipemu::wave_exchange(0.0f, 0.0f); }  // fibers run lane by lane: a real 64-lane rendezvous
// v_mfma_f32_32x32x2_f32 emulated with a 64-lane rendezvous (hipemu::wave_exchange): same lane <-> element maps, same fmaf chain.
struct f32x16 {
    float v[16];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
static inline f32x16 mfma_zero() { f32x16 z; for (int i = 0; i < 16; i++) { z.v[i] = 0.0f; } return z; }
static inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    const float* ab = hipemu::wave_exchange(a, b);  // ab[lane * 2 + {0: a, 1: b}] of all 64 lanes of this wavefront
    const int lane = hipemu::lane_id(), j = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float d = c.v[r];
        d = fmaf(ab[(i) * 2], ab[(j) * 2 + 1], d);            // k = 0: A from lane i, B from lane j
        d = fmaf(ab[(i + 32) * 2], ab[(j + 32) * 2 + 1], d);  // k = 1: lanes i + 32 / j + 32
        c.v[r] = d;
    }
    return c;
}
struct f32x2 { float x, y; };
static inline void pk_sum_diff_f32(float2 a, float2 b, f32x2& s, f32x2& d) {
    s = f32x2{ a.x + b.x, a.y + b.y };
    d = f32x2{ b.x - a.x, a.y - b.y };
}
struct f32x4 {
    float v[4];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
static inline f32x4 mfma4_zero() { f32x4 z; for (int i = 0; i < 4; i++) { z.v[i] = 0.0f; } return z; }
static inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    const float* ab = hipemu::wave_exchange(a, b);
    const int lane = hipemu::lane_id(), j = lane & 15, g = lane >> 4;
    for (int r = 0; r < 4; r++) {
        const int i = 4 * g + r;
        float d = c.v[r];
        for (int k = 0; k < 4; k++) { d = fmaf(ab[(i + 16 * k) * 2], ab[(j + 16 * k) * 2 + 1], d); }  // A from lane i + 16k, B from lane j + 16k
        c.v[r] = d;
    }
    return c;
}
}
