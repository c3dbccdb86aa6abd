// TEST INFRASTRUCTURE (fiber emulator): the one hip_ext.h entry point the product uses — a launch with start / stop events.
#pragma once
#include "hip_runtime.h"
template <class... KArgs, class... Args>
static inline void hipExtLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, hipEvent_t, hipEvent_t, unsigned, Args... args) {
    hipLaunchKernelGGL(kernel, grid, block, (size_t)shmem, s, args...);
}
