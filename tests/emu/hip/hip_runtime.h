// TEST INFRASTRUCTURE ONLY — a tiny single-threaded HIP *emulator* so the product's kernel sources
// (sdrplusplus_amd/csrc/*.hip) can be compiled with g++ and executed on a machine without a GPU, purely to check
// indexing / streaming-state logic in `-m "not gpu"` tests (tests/emu/Makefile builds tests/emu/libsdrpp_gpu_emu.so).
// It is NOT a fallback: the product library is always the hipcc build, the Python binding refuses to load this file
// unless a test asks for it explicitly, and no `-m gpu` test, smoke() or bench.py ever touches it.
//
// Model: one workgroup at a time; each work-item is a fiber (a stack of its own and a register switch in user space: glibc's swapcontext
// makes two signal-mask system calls per switch, which was a third of the CPU suite's run time); __syncthreads() yields round-robin until
// all fibers of the block have arrived.  Only the HIP subset the product uses is provided.
#pragma once
#if !defined(__x86_64__)
#include <ucontext.h>
#endif
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{ x, y }; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{ x, y, z, w }; }
static inline int2 make_int2(int x, int y) { return int2{ x, y }; }

namespace hipemu {
#if defined(__x86_64__)
    // switch(from, to): callee-saved registers + the floating-point control words go onto the current stack, its pointer into *from; then
    // the same in reverse from the stack `to` points at.  A fresh fiber's stack is laid out as if it had once called this function.
    __attribute__((naked, noinline)) static void fiber_switch(void** /*from_sp*/, void* /*to_sp*/) {
        asm volatile(
            "pushq %rbp\n\t" "pushq %rbx\n\t" "pushq %r12\n\t" "pushq %r13\n\t" "pushq %r14\n\t" "pushq %r15\n\t"
            "subq $8, %rsp\n\t" "stmxcsr (%rsp)\n\t" "fnstcw 4(%rsp)\n\t"
            "movq %rsp, (%rdi)\n\t"
            "movq %rsi, %rsp\n\t"
            "ldmxcsr (%rsp)\n\t" "fldcw 4(%rsp)\n\t" "addq $8, %rsp\n\t"
            "popq %r15\n\t" "popq %r14\n\t" "popq %r13\n\t" "popq %r12\n\t" "popq %rbx\n\t" "popq %rbp\n\t"
            "ret\n\t");
    }
#endif
    struct State {
        dim3 tIdx, bIdx, bDim, gDim;
#if defined(__x86_64__)
        std::vector<void*> sp;
        void* sched_sp = nullptr;
#else
        std::vector<ucontext_t> ctx;
        ucontext_t sched;
#endif
        std::vector<char*> stacks;
        std::vector<int> done;
        int cur = -1;
        std::function<void()> body;
        alignas(64) char dynshared[160 * 1024];
        // barrier / wavefront rendezvous bookkeeping (reset per block)
        int nthreads = 0, ndone = 0, bar_arrived = 0;
        unsigned bar_gen = 0;
        std::vector<unsigned> wx_gen;              // per work-item: number of wave exchanges done
        std::vector<unsigned long long> wx_count;  // [wave][parity]: deposits so far
        std::vector<float> wx_buf;                 // [wave][parity][64][2]
    };
    inline State& S() { static State s; return s; }
    inline std::mutex& launch_mutex() { static std::mutex m; return m; }  // (a function-local static: ONE for all kernel templates)
    inline void to_scheduler() {
        State& s = S();
#if defined(__x86_64__)
        fiber_switch(&s.sp[s.cur], s.sched_sp);
#else
        swapcontext(&s.ctx[s.cur], &s.sched);
#endif
    }
    inline void trampoline() {
        State& s = S();
        s.body();
        s.done[s.cur] = 1;
        s.ndone++;
        to_scheduler();  // (never resumed)
        abort();
    }
    inline void set_tid(int i) {
        State& s = S();
        s.tIdx.x = i % s.bDim.x;
        s.tIdx.y = (i / s.bDim.x) % s.bDim.y;
        s.tIdx.z = i / (s.bDim.x * s.bDim.y);
    }
    inline void run_block(int nthreads) {
        State& s = S();
        const size_t STK = 256 * 1024;
        if ((int)s.stacks.size() < nthreads) {
            size_t old = s.stacks.size();
            s.stacks.resize(nthreads, nullptr);
            for (size_t i = old; i < (size_t)nthreads; i++) { s.stacks[i] = (char*)malloc(STK); }
        }
#if defined(__x86_64__)
        s.sp.resize(s.stacks.size());
#else
        s.ctx.resize(s.stacks.size());
#endif
        s.done.assign(nthreads, 0);
        s.nthreads = nthreads;
        s.ndone = 0;
        s.bar_arrived = 0;
        s.bar_gen = 0;
        const int nwaves = (nthreads + 63) / 64;
        s.wx_gen.assign(nthreads, 0u);
        s.wx_count.assign((size_t)nwaves * 2, 0ull);
        s.wx_buf.assign((size_t)nwaves * 2 * 128, 0.0f);
        for (int i = 0; i < nthreads; i++) {
#if defined(__x86_64__)
            // [mxcsr | x87 control word] r15 r14 r13 r12 rbx rbp, return address = trampoline; at the `ret` the stack pointer is 8 past a
            // 16-byte boundary, as at every function entry
            uintptr_t top = (reinterpret_cast<uintptr_t>(s.stacks[i]) + STK) & ~(uintptr_t)15;
            uint64_t* q = reinterpret_cast<uint64_t*>(top) - 2;  // q[0]: return address (16-byte aligned slot), q[1]: padding
            q[0] = reinterpret_cast<uint64_t>(reinterpret_cast<void*>(&trampoline));
            q[1] = 0;
            for (int k = 1; k <= 6; k++) { q[-k] = 0; }
            uint32_t* cw = reinterpret_cast<uint32_t*>(q - 7);
            cw[0] = 0x1F80u;  // mxcsr: round to nearest, all exceptions masked
            cw[1] = 0x037Fu;  // x87 control word
            s.sp[i] = q - 7;
#else
            getcontext(&s.ctx[i]);
            s.ctx[i].uc_stack.ss_sp = s.stacks[i];
            s.ctx[i].uc_stack.ss_size = STK;
            s.ctx[i].uc_link = &s.sched;
            makecontext(&s.ctx[i], (void (*)())trampoline, 0);
#endif
        }
        int remaining = nthreads;
        while (remaining > 0) {
            remaining = 0;
            for (int i = 0; i < nthreads; i++) {
                if (s.done[i]) { continue; }
                s.cur = i;
                set_tid(i);
#if defined(__x86_64__)
                fiber_switch(&s.sched_sp, s.sp[i]);
#else
                swapcontext(&s.sched, &s.ctx[i]);
#endif
                if (!s.done[i]) { remaining++; }
            }
        }
    }
    inline void yield() {
        State& s = S();
        int me = s.cur;
        to_scheduler();
        s.cur = me;
        set_tid(me);
    }
    // workgroup barrier: released when every work-item that has not exited has arrived (exited wavefronts do not take part,
    // as on the hardware)
    inline void syncthreads() {
        State& s = S();
        const unsigned gen = s.bar_gen;
        s.bar_arrived++;
        while (s.bar_gen == gen) {
            if (s.bar_arrived >= s.nthreads - s.ndone) {
                s.bar_gen++;
                s.bar_arrived = 0;
                break;
            }
            yield();
        }
    }
    inline int lane_id() { State& s = S(); return s.cur & 63; }
    // all 64 lanes of a wavefront deposit (a, b) and get a view of everybody's values (double buffered by parity: a lane can be at
    // most one exchange ahead of the slowest lane of its wavefront)
    inline const float* wave_exchange(float a, float b) {
        State& s = S();
        const int me = s.cur, w = me >> 6, lane = me & 63;
        const unsigned g = s.wx_gen[me]++;
        const int par = (int)(g & 1u);
        float* buf = &s.wx_buf[((size_t)w * 2 + par) * 128];
        buf[lane * 2] = a;
        buf[lane * 2 + 1] = b;
        unsigned long long& cnt = s.wx_count[(size_t)w * 2 + par];
        cnt++;
        const unsigned long long want = 64ull * ((unsigned long long)(g >> 1) + 1ull);
        while (cnt < want) { yield(); }
        return buf;
    }
}
#define threadIdx (hipemu::S().tIdx)
#define blockIdx (hipemu::S().bIdx)
#define blockDim (hipemu::S().bDim)
#define gridDim (hipemu::S().gDim)
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::S().dynshared;
static inline void __syncthreads() { hipemu::syncthreads(); }

typedef int hipError_t;
typedef struct hipemuStream* hipStream_t;
typedef struct hipemuEvent { double t; }* hipEvent_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
#define hipErrorNoDevice 100
#define hipErrorNotReady 600
#define hipEventDisableTiming 2
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipStreamNonBlocking 1
#define HIP_SYMBOL(x) (x)
template <class T> static inline int hipMemcpyFromSymbol(void* dst, const T& sym, size_t n) { memcpy(dst, &sym, n); return 0; }
template <class T> static inline int hipMemcpyToSymbol(T& sym, const void* src, size_t n) { memcpy(&sym, src, n); return 0; }
#define hipHostMallocDefault 0
#define hipHostMallocMapped 2
#define hipHostMallocCoherent 0x40000000
static inline int hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return 0; }

static inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "hipSuccess(emu)" : "hipError(emu)"; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? 0 : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
static inline hipError_t hipHostFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent{ 0 }; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventQuery(hipEvent_t) { return 0; }
static inline hipError_t hipStreamQuery(hipStream_t) { return 0; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return 0; }
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "hipemu (CPU fiber emulator, tests only)");
    strcpy(p->gcnArchName, "emu");
    p->multiProcessorCount = 1;
    return 0;
}

namespace hipemu {
// SDRPP_EMU_NO_EXEC: an argument that looks like the tick kernel's completion record gets its flag published, nothing else happens
template <class T> static inline auto publish_done(const T& d, int) -> decltype((void)d.host_flag, (void)d.value, void()) { *d.host_flag = d.value; }
template <class T> static inline void publish_done(const T&, long) {}
}
template <typename... KArgs, typename... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*shmem*/, hipStream_t, Args... args) {
    static const bool no_exec = getenv("SDRPP_EMU_NO_EXEC") != nullptr;  // host-planner timing runs (tools/plan_time_emu.py): nothing is computed
    if (no_exec) {
        (hipemu::publish_done(args, 0), ...);
        return;
    }
    std::lock_guard<std::mutex> launch_lck(hipemu::launch_mutex());  // one workgroup at a time PROCESS-wide: host threads take turns
    hipemu::State& s = hipemu::S();
    if (getenv("SDRPP_EMU_TRACE")) { fprintf(stderr, "[hipemu] launch grid=(%u,%u,%u) block=(%u,%u,%u)\n", grid.x, grid.y, grid.z, block.x, block.y, block.z); }
    s.gDim = grid;
    s.bDim = block;
    const int nthreads = (int)(block.x * block.y * block.z);
    for (unsigned bz = 0; bz < grid.z; bz++) {
        for (unsigned by = 0; by < grid.y; by++) {
            for (unsigned bx = 0; bx < grid.x; bx++) {
                s.bIdx = dim3(bx, by, bz);
                s.body = [&]() { kernel(static_cast<KArgs>(args)...); };
                hipemu::run_block(nthreads);
            }
        }
    }
}

// device intrinsics used by the kernels
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline void sincospif(float x, float* s, float* c) {
    const double a = 3.14159265358979323846 * (double)x;
    *s = (float)sin(a);
    *c = (float)cos(a);
}
static inline void __threadfence() {}
static inline void __threadfence_system() {}
static inline int atomicAdd(int* p, int v) { int o = *p; *p += v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p += v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) { *p = v; } return o; }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) { r = (r << 1) | (v & 1u); v >>= 1; } return r; }
using std::min;
using std::max;
