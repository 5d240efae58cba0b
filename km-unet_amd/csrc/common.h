// Shared host/device helpers for libkmunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/kmunet_hip.h"

namespace kmu {

void set_error(const char* fmt, ...);

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

typedef float floatx4 __attribute__((ext_vector_type(4)));

// wave64 all-reduce helpers (DPP/bpermute via __shfl_xor)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace kmu

// Raise a kernel's dynamic-LDS limit once per call site / template instantiation (never during a later
// stream capture: the first, un-captured warm-up call does it).
#define KMU_MAX_LDS(kern, bytes)                                                                                   \
    do {                                                                                                           \
        static size_t kmu_lds_cfg_ = 48 * 1024;                                                                    \
        if ((size_t)(bytes) > kmu_lds_cfg_) {                                                                      \
            (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            kmu_lds_cfg_ = (size_t)(bytes);                                                                        \
        }                                                                                                          \
    } while (0)

#define KMU_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            kmu::set_error(__VA_ARGS__); \
            return KMU_ERR_ARG;         \
        }                               \
    } while (0)
