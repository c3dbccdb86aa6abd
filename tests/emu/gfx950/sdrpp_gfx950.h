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
}
