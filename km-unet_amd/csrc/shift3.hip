// Tap stacking for DirectionViM's (3,1) / (1,3) projections (KM_UNetV3_SH.py:170-172): a 3-tap convolution along one
// spatial axis is a pointwise convolution of the three shifted copies,
//     conv_{3 taps}(x)[co] = sum_t sum_ci W[co,ci,t] x[ci](. + (t-1) e_axis)  =  pwconv(stack3(x), W')[co],
//     stack3(x)[3*c + t](y, x) = x[c]((y, x) + (t-1) e_axis)    (zero outside the image: padding 1 on that axis).
// Channel-major tap order: W'[co][3*ci + t] = W[co][ci][t] is then the convolution's own weight tensor read as [Co, 3*Ci] -- a
// view, no permuted copy forward and none of its gradient backward (22 small ATen copies per step with the tap-major order).
// One streaming launch each way (forward writes the 3C-channel stack, backward gathers the three shifted gradient
// slices back into dx), after which csrc/pwconv.hip does the contraction on the matrix cores -- instead of MIOpen's
// NCHW->NHWC transpose + implicit-GEMM + transpose back (7-9 launches per convolution and direction).
#include "common.h"

namespace {

// axis 0: shift along H, axis 1: along W.  One thread per element of x / dx; out[3c + t][p] = x[c][p + (t-1) e_axis] or 0.
__global__ __launch_bounds__(256) void shift3_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W,
                                                         int axis, size_t total) {
    const size_t hw = (size_t)H * W, chw = (size_t)C * hw;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t b = e / chw, r = e - b * chw;      // r = c*hw + y*W + xx
        const int p = (int)(r % hw), y = p / W, xx = p - y * W;
        float lo, hi;
        if (axis == 0) {
            lo = y > 0 ? x[e - W] : 0.f;
            hi = y + 1 < H ? x[e + W] : 0.f;
        } else {
            lo = xx > 0 ? x[e - 1] : 0.f;
            hi = xx + 1 < W ? x[e + 1] : 0.f;
        }
        float* ob = out + b * 3 * chw + 3 * (r - p) + p;      // channel 3c of sample b
        ob[0] = lo;
        ob[hw] = x[e];
        ob[2 * hw] = hi;
    }
}

__global__ __launch_bounds__(256) void shift3_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int C, int H, int W,
                                                         int axis, size_t total) {
    const size_t hw = (size_t)H * W, chw = (size_t)C * hw;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t b = e / chw, r = e - b * chw;
        const int p = (int)(r % hw), y = p / W, xx = p - y * W;
        const float* gb = g + b * 3 * chw + 3 * (r - p) + p;
        float s = gb[hw];
        if (axis == 0) {
            if (y + 1 < H) s += gb[W];
            if (y > 0) s += gb[2 * hw - W];
        } else {
            if (xx + 1 < W) s += gb[1];
            if (xx > 0) s += gb[2 * hw - 1];
        }
        dx[e] = s;
    }
}

inline unsigned grid_for(size_t n) {
    const size_t b = (n + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b ? b : 1));
}

}  // namespace

extern "C" int kmu_shift3_fwd(const float* x, float* out, int B, int C, int H, int W, int axis, kmu_stream_t stream) {
    KMU_REQUIRE(x && out, "shift3_fwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (axis == 0 || axis == 1), "shift3_fwd: bad dims / axis");
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(shift3_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, C, H, W, axis, total);
    return kmu::launch_status("shift3_fwd");
}

extern "C" int kmu_shift3_bwd(const float* gout, float* dx, int B, int C, int H, int W, int axis, kmu_stream_t stream) {
    KMU_REQUIRE(gout && dx, "shift3_bwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (axis == 0 || axis == 1), "shift3_bwd: bad dims / axis");
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(shift3_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, gout, dx, C, H, W, axis, total);
    return kmu::launch_status("shift3_bwd");
}
