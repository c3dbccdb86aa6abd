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

}  // namespace sdrpp_k
