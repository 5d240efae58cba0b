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

// Tied-operand fp32 MFMA (vdst == srcC by construction).  With the builtin, hipcc (ROCm 7.2) sometimes rotates
// loop-carried accumulators through v_accvgpr_write/mov placed right behind an MFMA that still reads them as
// SrcC -- a WAR hazard it does not always pad (tools/check_mfma_overlap.py; it corrupted K2's d_w_bcdt at C=64).
// Hand-placed wait states (cdna_hip_programming.md 5.7): `s_nop 1` ahead of each MFMA covers a VALU-written A/B
// operand; mfma_drain() before the first non-MFMA reader covers the 8-pass result latency.
__device__ __forceinline__ void mfma_tied(floatx4& acc, float a, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain(floatx4& acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc)); }

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
