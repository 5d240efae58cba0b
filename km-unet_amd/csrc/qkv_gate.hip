// DirectionAttention's local gate  attn = sigmoid(q * k) * v  (KM_UNetV3_SH.py:258-261) as one streaming kernel
// over the packed qkv tensor [B,3C,HW] (q, k, v = chunks of the 1x1 qkv conv output), forward and backward.
// HBM-bound: forward reads 3 and writes 1 tensor-equivalents; ATen runs it as mul, sigmoid, mul (+ chunk views).
#include "common.h"

using kmu::floatx4;

namespace {

__global__ __launch_bounds__(256) void qkv_gate_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C,
                                                           int HW, size_t total4) {
    // total4 = B*C*HW/4 (HW % 4 == 0) ; one thread = 4 consecutive positions of one (b, c) plane
    const size_t chw = (size_t)C * HW;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t * 4, b = e / chw, r = e - b * chw;
        const float* base = qkv + b * 3 * chw + r;
        const floatx4 q = *reinterpret_cast<const floatx4*>(base), k = *reinterpret_cast<const floatx4*>(base + chw),
                      v = *reinterpret_cast<const floatx4*>(base + 2 * chw);
        floatx4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = v[i] / (1.f + __expf(-q[i] * k[i]));
        *reinterpret_cast<floatx4*>(out + e) = o;
    }
}

__global__ __launch_bounds__(256) void qkv_gate_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ g,
                                                           float* __restrict__ dqkv, int C, int HW, size_t total4) {
    const size_t chw = (size_t)C * HW;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t * 4, b = e / chw, r = e - b * chw;
        const float* base = qkv + b * 3 * chw + r;
        float* dbase = dqkv + b * 3 * chw + r;
        const floatx4 q = *reinterpret_cast<const floatx4*>(base), k = *reinterpret_cast<const floatx4*>(base + chw),
                      v = *reinterpret_cast<const floatx4*>(base + 2 * chw), go = *reinterpret_cast<const floatx4*>(g + e);
        floatx4 dq, dk, dv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = 1.f / (1.f + __expf(-q[i] * k[i]));
            const float ds = go[i] * v[i] * s * (1.f - s);
            dq[i] = ds * k[i];
            dk[i] = ds * q[i];
            dv[i] = go[i] * s;
        }
        *reinterpret_cast<floatx4*>(dbase) = dq;
        *reinterpret_cast<floatx4*>(dbase + chw) = dk;
        *reinterpret_cast<floatx4*>(dbase + 2 * chw) = dv;
    }
}

inline unsigned grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b ? b : 1));
}

}  // namespace

extern "C" int kmu_qkv_gate_fwd(const float* qkv, float* out, int B, int C, int HW, kmu_stream_t stream) {
    KMU_REQUIRE(qkv && out, "qkv_gate_fwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && HW > 0 && HW % 4 == 0, "qkv_gate_fwd: HW=%d must be a positive multiple of 4", HW);
    const size_t total4 = (size_t)B * C * HW / 4;
    hipLaunchKernelGGL(qkv_gate_fwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, qkv, out, C, HW, total4);
    return kmu::launch_status("qkv_gate_fwd");
}

extern "C" int kmu_qkv_gate_bwd(const float* qkv, const float* gout, float* dqkv, int B, int C, int HW,
                                kmu_stream_t stream) {
    KMU_REQUIRE(qkv && gout && dqkv, "qkv_gate_bwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && HW > 0 && HW % 4 == 0, "qkv_gate_bwd: HW=%d must be a positive multiple of 4", HW);
    const size_t total4 = (size_t)B * C * HW / 4;
    hipLaunchKernelGGL(qkv_gate_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, qkv, gout, dqkv, C, HW,
                       total4);
    return kmu::launch_status("qkv_gate_bwd");
}
