// gfx950-specific device helpers (included as <sdrpp_gfx950.h>; the test emulator supplies its own plain-C++ version).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrpp_k {

// Wave-uniform read-only float array.  Reading it through the constant address space makes the compiler fetch the
// values with scalar loads (s_load_dwordxN into SGPRs) instead of 64 identical vector loads: filter taps are the same
// for every lane of a wavefront, and an SGPR can feed v_fma_f32 directly, so tap fetches cost no VALU, LDS or TA cycles.
// Only valid for memory that is not written while the kernel runs and for indices that are uniform across the wave.
struct UniformF32 {
    const float __attribute__((address_space(4))) * p;
    __device__ __forceinline__ float operator[](int i) const { return p[i]; }
};
__device__ __forceinline__ UniformF32 as_uniform(const void* ptr) {
    UniformF32 u;
    u.p = (const float __attribute__((address_space(4)))*)(uintptr_t)ptr;
    return u;
}

// The same for a wave-uniform int table (indices uniform across the wave): s_load_dword, counted by lgkmcnt — not by the vector-memory
// counter the kernel's loads and stores share.
struct UniformI32 {
    const int __attribute__((address_space(4))) * p;
    __device__ __forceinline__ int operator[](int i) const { return p[i]; }
};
__device__ __forceinline__ UniformI32 as_uniform_i32(const void* ptr) {
    UniformI32 u;
    u.p = (const int __attribute__((address_space(4)))*)(uintptr_t)ptr;
    return u;
}

// A value that is the same in every lane of the wavefront, stated so that the compiler keeps it (and what is computed from it) in scalar registers
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Pointers fetched from a job table in memory have no known address space, so the compiler falls back to FLAT accesses — which
// count against BOTH the vector-memory and the LDS wait counters (every LDS wait then also waits for L2/HBM).  These helpers
// state that the pointer is device global memory, which restores global_load / global_store.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float global_load_f32(const float* p, long long i) { return ((const float __attribute__((address_space(1)))*)(uintptr_t)p)[i]; }
__device__ __forceinline__ int global_load_i32(const int* p, long long i) { return ((const int __attribute__((address_space(1)))*)(uintptr_t)p)[i]; }
__device__ __forceinline__ float2 global_load_f32x2(const float2* p, long long i) {
    const f32x2 v = ((const f32x2 __attribute__((address_space(1)))*)(uintptr_t)p)[i];
    return make_float2(v.x, v.y);
}
// four consecutive floats, 16-byte aligned: one global_load_dwordx4
__device__ __forceinline__ float4 global_load_f32x4(const float4* p, long long i) {
    typedef float f32x4_ld __attribute__((ext_vector_type(4)));
    const f32x4_ld v = ((const f32x4_ld __attribute__((address_space(1)))*)(uintptr_t)p)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
// four consecutive floats (two complex samples) that are only 8-byte aligned: one global_load_dwordx4 (global memory accesses
// need dword alignment only)
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 global_load_f32x4_unaligned(const float* p, long long i) {
    const f32x4_a4 v = *((const f32x4_a4 __attribute__((address_space(1)))*)(uintptr_t)(p + i));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void global_store_f32x2(float2* p, long long i, float2 v) {
    f32x2 t;
    t.x = v.x;
    t.y = v.y;
    ((f32x2 __attribute__((address_space(1)))*)(uintptr_t)p)[i] = t;
}

// Wave-uniform base pointer + a 32-bit BYTE offset per lane: the global_load / global_store form with the base in scalar registers and one
// 32-bit offset register (no 64-bit address arithmetic per access).  The offset must stay below 4 GiB.
__device__ __forceinline__ float2 global_load_f32x2_boff(const void* base, unsigned byte_off) {
    const f32x2 v = *((const f32x2 __attribute__((address_space(1)))*)((const char __attribute__((address_space(1)))*)(uintptr_t)base + byte_off));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void global_store_f32x2_boff(void* base, unsigned byte_off, float2 v) {
    f32x2 t;
    t.x = v.x;
    t.y = v.y;
    *((f32x2 __attribute__((address_space(1)))*)((char __attribute__((address_space(1)))*)(uintptr_t)base + byte_off)) = t;
}

__device__ __forceinline__ void global_store_f32_boff(void* base, unsigned byte_off, float v) {
    *((float __attribute__((address_space(1)))*)((char __attribute__((address_space(1)))*)(uintptr_t)base + byte_off)) = v;
}

// 1 / x to 1 ulp (v_rcp_f32); x must be a normal number
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// four consecutive floats, 16-byte aligned: one global_store_dwordx4
typedef float f32x4_a16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void global_store_f32x4(float* p, long long i, float4 v) {
    f32x4_a16 t;
    t.x = v.x;
    t.y = v.y;
    t.z = v.z;
    t.w = v.w;
    *((f32x4_a16 __attribute__((address_space(1)))*)(uintptr_t)(p + i)) = t;
}

// 16 bytes as four dwords through the GLOBAL address space (copies: tick_kernels.h copy_one).  A generic `const void*` makes every access FLAT, and a
// flat store may alias private memory — so the compiler kept a copy loop's in-flight values in SCRATCH and reloaded each one behind the store before
// it (round 5: 8 scratch stores + 8 scratch loads per 8 copied vectors in the landing copy, the job-table upload and the result copies of every tick).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 global_load_u32x4(const void* p, long long i) {
    const u32x4_t v = ((const u32x4_t __attribute__((address_space(1)))*)(uintptr_t)p)[i];
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void global_store_u32x4(void* p, long long i, uint4 v) {
    u32x4_t t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    ((u32x4_t __attribute__((address_space(1)))*)(uintptr_t)p)[i] = t;
}
__device__ __forceinline__ unsigned global_load_u32(const void* p, long long i) { return ((const unsigned __attribute__((address_space(1)))*)(uintptr_t)p)[i]; }
__device__ __forceinline__ void global_store_u32(void* p, long long i, unsigned v) { ((unsigned __attribute__((address_space(1)))*)(uintptr_t)p)[i] = v; }

// ... and only 4-byte aligned (one global_store_dwordx4 all the same)
__device__ __forceinline__ void global_store_f32x4_unaligned(float* p, long long i, float4 v) {
    f32x4_a4 t;
    t.x = v.x;
    t.y = v.y;
    t.z = v.z;
    t.w = v.w;
    *((f32x4_a4 __attribute__((address_space(1)))*)(uintptr_t)(p + i)) = t;
}

// Scheduling fence: the instruction scheduler does not move anything across it.  Used to keep LDS reads issued two steps ahead
// of the matrix instructions that consume them (left alone, the scheduler sinks them next to their use and the wave then
// waits out the full LDS latency in front of every v_mfma).
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// A value the optimiser cannot see through.  Used on thread indices at the top of a loop body whose address arithmetic is loop invariant:
// hoisted out of the loop those addresses (LDS exchange positions, store offsets) stay alive across it — 60-90 registers in the FFT tile
// walks — for the price of recomputing a handful of integer instructions per tile.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// Wavefront-level synchronisation of LDS traffic: the 64 lanes of a wavefront run in lock step and its LDS operations complete in
// order, so no hardware barrier is needed for one lane to read what another lane of the SAME wavefront wrote — only the compiler
// must be kept from moving LDS accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// diagnostic build (`make ticktrace`): time marks inside a kernel body (100 MHz wall clock), one set per workgroup of the launch
#ifdef SDRPP_TICK_TRACE
__device__ unsigned long long g_tick_mark[1 << 16][4];
#define TICK_MARK(k)                                                                                      \
    do {                                                                                                  \
        if (threadIdx.x == 0) { g_tick_mark[blockIdx.x & 0xffff][k] = (unsigned long long)wall_clock64(); } \
    } while (0)
#else
#define TICK_MARK(k) \
    do {             \
    } while (0)
#endif

// Every memory operation this wavefront has issued is complete — no cache write-back, no invalidate.  The wait is spelled out: a
// workgroup-scope release fence (`__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")`) compiles to NOTHING for global stores on gfx950
// outside threadgroup-split mode (the waves of a workgroup share their L1), and a workgroup was then counted "done" — and the tick's completion
// flag published to the host — with the result stores of three of its four wavefronts still in flight: a host that read a result slot
// within microseconds of the flag met pieces of the previous content (found by tests/test_bench_geometry_gpu.py, round 4).
__device__ __forceinline__ void wave_stores_done() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// Index of the first entry of the ascending table `ends[0 .. n)` (n <= 64) that is greater than b, and the entry in front of it (0 for the
// first): every lane fetches one entry — ONE memory round trip instead of a chain of dependent scalar loads — and a ballot does the search.
__device__ __forceinline__ int wave_upper_bound(const int* ends, int n, int b, int* prev_end) {
    const int lane = threadIdx.x & 63;
    const int mine = global_load_i32(ends, lane < n ? lane : (n > 0 ? n - 1 : 0));
    const unsigned long long below = __ballot(lane < n && mine <= b);
    const int idx = (int)__builtin_popcountll(below);
    const int prev = __builtin_amdgcn_readlane(mine, idx > 0 ? idx - 1 : 0);
    *prev_end = idx > 0 ? prev : 0;
    return idx;
}

// Flags in LDS between the wavefronts of a workgroup (producer / consumer rings): a wavefront's LDS operations are carried out in
// program order and the LDS serves one CU, so a flag written after the data (release) and read before it (acquire) is all the
// ordering there is to it; a waiting wavefront sleeps 64 cycles between looks so that it does not take issue slots from the others.
__device__ __forceinline__ void lds_flag_set(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// A wait that cannot be satisfied (a bug in the tile bookkeeping) must not hang the device: after ~2^20 looks (a quarter of a second) the
// wavefront gives up, counts the event in its context's page-locked word (PipeJob::timeouts) — the host turns that into an error at the next
// sdrpp_sync or host-synchronising read of that context — and goes on.
#ifndef SDRPP_FLAG_SLEEP
#define SDRPP_FLAG_SLEEP 8  // 512 cycles between looks (64 -> 512: 2 % on the pipelined launch: every look is a vector compare and an LDS read taken from the matrix loops)
#endif
__device__ __forceinline__ void lds_flag_wait_ge(int* f, int need, int* timeouts) {
    int looks = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
        __builtin_amdgcn_s_sleep(SDRPP_FLAG_SLEEP);
        if (++looks > (1 << 20)) {
            if ((threadIdx.x & 63) == 0 && timeouts) { __hip_atomic_fetch_add(timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            break;
        }
    }
}

// The same search over a table of exactly 64 slots, fetching in the SAME round trip the first six dwords of the record that belongs to every
// slot (records `stride_bytes` apart, dword aligned, all 64 readable): lane l asks for end l and record l, the ballot finds the slot and
// v_readlane hands out its record — where a look-up followed by a load of the record it found is two trips to memory, one after the other
// (a tick's workgroups start on a cold cache: every dependent load is ~1.5 us of a 13 us tick).
__device__ __forceinline__ int wave_upper_bound64_rec(const int* ends, int n, int b, int* prev_end, const void* recs, int stride_bytes, int (&rec)[6]) {
    const int lane = threadIdx.x & 63;
    const int mine = global_load_i32(ends, lane);
    const float* rp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(recs) + (size_t)lane * (size_t)stride_bytes);
    const float4 r0 = global_load_f32x4_unaligned(rp, 0);
    const float r4 = global_load_f32(rp, 4), r5 = global_load_f32(rp, 5);
    const unsigned long long below = __ballot(lane < n && mine <= b);
    const int idx = __popcll(below);
    const int prev = __builtin_amdgcn_readlane(mine, idx > 0 ? idx - 1 : 0);
    *prev_end = idx > 0 ? prev : 0;
    const int src = idx < 64 ? idx : 63;
    rec[0] = __builtin_amdgcn_readlane(__float_as_int(r0.x), src);
    rec[1] = __builtin_amdgcn_readlane(__float_as_int(r0.y), src);
    rec[2] = __builtin_amdgcn_readlane(__float_as_int(r0.z), src);
    rec[3] = __builtin_amdgcn_readlane(__float_as_int(r0.w), src);
    rec[4] = __builtin_amdgcn_readlane(__float_as_int(r4), src);
    rec[5] = __builtin_amdgcn_readlane(__float_as_int(r5), src);
    return idx;
}

// Issue priority of this wavefront among the wavefronts of its SIMD (s_setprio 0..3)
__device__ __forceinline__ void wave_prio_high() { __builtin_amdgcn_s_setprio(3); }
__device__ __forceinline__ void wave_prio_low() { __builtin_amdgcn_s_setprio(0); }

// Wave-uniform helpers: value of lane `src` (src uniform across the wavefront: v_readlane_b32), and the maximum over all 64 lanes.
__device__ __forceinline__ float wave_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
// value of the lane to the left (lane 0: `first`)
__device__ __forceinline__ float wave_shr1(float v, float first) {
    const float r = __shfl_up(v, 1, 64);
    return ((threadIdx.x & 63) == 0) ? first : r;
}
// index of the first lane whose predicate holds, 64 if none (v_cmp + s_ff1)
__device__ __forceinline__ int wave_first(bool pred) {
    const unsigned long long m = __ballot(pred);
    return m ? (int)__builtin_ctzll(m) : 64;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { v += __shfl_xor(v, d, 64); }
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float o = __shfl_xor(v, d, 64);
        v = (o > v) ? o : v;
    }
    return v;
}

// FP32 matrix core: D(32x32) += A(32x2) * B(2x32), v_mfma_f32_32x32x2_f32, 64 cycles per instruction.  Lane l supplies
// A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; it receives, in register r, D[i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][j = l & 31].
// Numerically the result is the k-ordered fmaf chain d = fmaf(A[i][1], B[1][j], fmaf(A[i][0], B[0][j], c)) — plain f32, one
// rounding per product-accumulate, which is what keeps the CPU test emulator and the hardware bit-identical.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma_zero() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) { z[i] = 0.0f; }
    return z;
}
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// Packed FP32 sum and "crossed" difference of two (re, im) register pairs, the B operands of the pair-per-half front ends:
//     s = (a.x + b.x,  a.y + b.y)            d = (b.x - a.x,  a.y - b.y)     (neg_lo on the first, neg_hi on the second operand)
// TWO vector instructions for four values (v_pk_add_f32; the compiler splits a plain two-element vector add into two v_add_f32 as soon as the
// halves go to different consumers, hence the spelled-out instructions).  Each component is the correctly rounded sum / difference — bitwise
// what fmaf(+-1, b, a) or a plain add gives.
// The trailing s_nop 1 is REQUIRED: a matrix instruction that reads a register a vector instruction has just written needs two wait states in
// between (the compiler inserts them for instructions it schedules itself — `v_fmac; v_mfma; s_nop 0; v_mfma` in the round-4 loop — but cannot
// classify the text of an asm statement: with one wait state the v_mfma behind these adds read the OLD register contents on the device, which
// the CPU emulator — plain C++ — cannot show; found by the round-5 device run, 23 parity tests).
__device__ __forceinline__ void pk_sum_diff_f32(float2 a, float2 b, f32x2& s, f32x2& d) {
    f32x2 av, bv;
    av.x = a.x; av.y = a.y; bv.x = b.x; bv.y = b.y;
    asm("v_pk_add_f32 %0, %2, %3\n\tv_pk_add_f32 %1, %2, %3 neg_lo:[1,0] neg_hi:[0,1]\n\ts_nop 1" : "=&v"(s), "=&v"(d) : "v"(av), "v"(bv));
}

// D(16x16) += A(16x4) * B(4x16), v_mfma_f32_16x16x4_f32, 32 cycles per instruction (same FLOP rate as the 32x32x2 form).  Lane l
// supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; register r receives D[i = 4 * (l >> 4) + r][j = l & 15];
// numerically d = fmaf(A[i][3], B[3][j], fmaf(A[i][2], B[2][j], fmaf(A[i][1], B[1][j], fmaf(A[i][0], B[0][j], c)))).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4_zero() {
    f32x4 z;
    z[0] = 0.0f; z[1] = 0.0f; z[2] = 0.0f; z[3] = 0.0f;
    return z;
}
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

}  // namespace sdrpp_k
