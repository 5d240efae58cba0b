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

// wave64 reductions on DPP (VALU-rate row shifts + row broadcasts; __shfl_xor lowers to ds_bpermute = an LDS
// round trip per step).  Sequence: row_shr 1,2,4,8 leaves each 16-lane row's total in its lane 15; row_bcast:15 adds
// it into the next row (lanes 31 / 63 then hold their 32-lane half's total); row_bcast:31 completes lane 63.
// Out-of-row sources read `identity` (old operand, bound_ctrl off).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {  // the builtin is int -> int: bit-cast, never convert
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
struct OpSum {
    static __device__ __forceinline__ float id() { return 0.f; }
    static __device__ __forceinline__ float f(float a, float b) { return a + b; }
};
struct OpMax {
    static __device__ __forceinline__ float id() { return -INFINITY; }
    static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); }
};
// WIDTH = 64: total over the wave; WIDTH = 32: total over each 32-lane half; WIDTH = 16: over each 16-lane row.  Result broadcast to every lane of the group.
template <class Op, int WIDTH>
__device__ __forceinline__ float wave_reduce(float v) {
    v = Op::f(v, dpp_mov<0x111, 0xf>(Op::id(), v));  // row_shr:1
    v = Op::f(v, dpp_mov<0x112, 0xf>(Op::id(), v));  // row_shr:2
    v = Op::f(v, dpp_mov<0x114, 0xf>(Op::id(), v));  // row_shr:4
    v = Op::f(v, dpp_mov<0x118, 0xf>(Op::id(), v));  // row_shr:8
    if (WIDTH == 16)                                  // lane 15 of each DPP row holds the row's total
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)((threadIdx.x & 63) | 15) << 2, __builtin_bit_cast(int, v)));
    v = Op::f(v, dpp_mov<0x142, 0xa>(Op::id(), v));  // row_bcast:15 -> rows 1, 3
    if (WIDTH == 64) {
        v = Op::f(v, dpp_mov<0x143, 0xc>(Op::id(), v));  // row_bcast:31 -> rows 2, 3
        return readlane_f(v, 63);
    }
    const float lo = readlane_f(v, 31), hi = readlane_f(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce<OpSum, 64>(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce<OpMax, 64>(v); }

// Tied-operand fp32 MFMA (vdst == srcC by construction).  With the builtin, hipcc (ROCm 7.2) sometimes rotates
// loop-carried accumulators through v_accvgpr_write/mov placed right behind an MFMA that still reads them as
// SrcC -- a WAR hazard it does not always pad (tools/check_mfma_overlap.py; it corrupted K2's d_w_bcdt at C=64).
// Hand-placed wait states (cdna_hip_programming.md 5.7): `s_nop 1` ahead of each MFMA covers a VALU-written A/B
// operand; mfma_drain() before the first non-MFMA reader covers the 8-pass result latency.
__device__ __forceinline__ void mfma_tied(floatx4& acc, float a, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain(floatx4& acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc)); }

// The same for the bf16 matrix core (v_mfma_f32_16x16x32_bf16: 8 bf16 per lane for A and B).  Used where the ISA screen
// (tools/check_mfma_overlap.py) catches hipcc emitting a partially overlapping vdst / SrcC pair for the builtin.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma_bf16_tied(floatx4& acc, bf16x8_t a, bf16x8_t b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
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
