// Front half of IntelligentWaveletPoolingModule (WPL/iwp.py:124-130) as one 2x2 stencil each way:
//   Haar DWT_2D (iwp.py:47-113): with a b / c d the 2x2 block of x,
//     LL = (a+b+c+d)/2   LH = (a-b+c-d)/2   HL = (a+b-c-d)/2   HH = (a-b-c+d)/2
//   the reference's high-pass matrix has an all-zero last row (iwp.py:79): the last column of LH / HH and the last row
//   of HL / HH of the half-resolution maps are zero;
//   high_freq_attention = Softmax2d(conv1x1(high)) is a softmax over ONE channel == 1 exactly, so
//   enhanced_high_freq == cat[LH, HL, HH] and its gradient w.r.t. high_freq_conv is exactly zero;
//   out = cat[LL (C channels), mean over the 3C high-band channels (1 channel), zero channels up to Ct]  -> fusion_conv stays
//   a 1x1 conv; Ct = C + 1 rounded up to a multiple of 16 (with zero weight columns) lets it run on csrc/pwconv.hip.
// Stock ATen: ~35 launches forward / ~70 backward of strided-slice arithmetic per module; HBM-bound, reads x once.
#include "common.h"

namespace {

typedef float floatx2 __attribute__((ext_vector_type(2)));

// one thread per half-resolution pixel (b, y, x), looping over channels (the high-band mean needs all of them)
__global__ __launch_bounds__(256) void iwp_front_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int Ct, int H,
                                                            int W, int total) {
    const int h2 = H / 2, w2 = W / 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int xo = t % w2, yo = (t / w2) % h2, b = t / (w2 * h2);
    const float cm = xo == w2 - 1 ? 0.f : 1.f, rm = yo == h2 - 1 ? 0.f : 1.f;
    const float* src = x + ((size_t)b * C * H + 2 * yo) * W + 2 * xo;
    float* dst = out + ((size_t)b * Ct * h2 + yo) * w2 + xo;
    float hsum = 0.f;
    for (int c = 0; c < C; ++c) {
        const floatx2 r0 = *reinterpret_cast<const floatx2*>(src + (size_t)c * H * W);
        const floatx2 r1 = *reinterpret_cast<const floatx2*>(src + (size_t)c * H * W + W);
        const float a = r0[0], bb = r0[1], cc = r1[0], d = r1[1];
        dst[(size_t)c * h2 * w2] = (a + bb + cc + d) * 0.5f;
        const float lh = (a - bb + cc - d) * 0.5f * cm, hl = (a + bb - cc - d) * 0.5f * rm, hh = (a - bb - cc + d) * 0.5f * (rm * cm);
        hsum += (lh + hl) + hh;
    }
    dst[(size_t)C * h2 * w2] = hsum / (3.f * C);
    for (int c = C + 1; c < Ct; ++c) dst[(size_t)c * h2 * w2] = 0.f;   // channel padding for the pointwise-conv kernels
}

__global__ __launch_bounds__(256) void iwp_front_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int C, int Ct, int H,
                                                            int W, int total) {
    const int h2 = H / 2, w2 = W / 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int xo = t % w2, yo = (t / w2) % h2, b = t / (w2 * h2);
    const float cm = xo == w2 - 1 ? 0.f : 1.f, rm = yo == h2 - 1 ? 0.f : 1.f;
    const float* gs = g + ((size_t)b * Ct * h2 + yo) * w2 + xo;
    float* dst = dx + ((size_t)b * C * H + 2 * yo) * W + 2 * xo;
    const float gm = gs[(size_t)C * h2 * w2] / (3.f * C);
    const float dlh = gm * cm, dhl = gm * rm, dhh = gm * (rm * cm);
    const float ha = (dlh + dhl + dhh) * 0.5f, hb = (-dlh + dhl - dhh) * 0.5f, hc = (dlh - dhl - dhh) * 0.5f,
                hd = (-dlh - dhl + dhh) * 0.5f;
    for (int c = 0; c < C; ++c) {
        const float gl = gs[(size_t)c * h2 * w2] * 0.5f;
        *reinterpret_cast<floatx2*>(dst + (size_t)c * H * W) = floatx2{gl + ha, gl + hb};
        *reinterpret_cast<floatx2*>(dst + (size_t)c * H * W + W) = floatx2{gl + hc, gl + hd};
    }
}

int check(const char* what, int B, int C, int H, int W) {
    KMU_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "%s: need even H, W (got %dx%d)", what, H, W);
    KMU_REQUIRE((long)B * (H / 2) * (W / 2) < (1L << 31), "%s: problem too large", what);
    return 0;
}

}  // namespace

extern "C" int kmu_iwp_front_fwd(const float* x, float* out, int B, int C, int Ct, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && out, "iwp_front_fwd: null pointer");
    KMU_REQUIRE(Ct >= C + 1, "iwp_front_fwd: %d output channels cannot hold C + 1 = %d", Ct, C + 1);
    if (int rc = check("iwp_front_fwd", B, C, H, W)) return rc;
    const int total = B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(iwp_front_fwd_kernel, dim3(kmu::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, C, Ct, H, W, total);
    return kmu::launch_status("iwp_front_fwd");
}

extern "C" int kmu_iwp_front_bwd(const float* gout, float* dx, int B, int C, int Ct, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(gout && dx, "iwp_front_bwd: null pointer");
    KMU_REQUIRE(Ct >= C + 1, "iwp_front_bwd: %d gradient channels cannot hold C + 1 = %d", Ct, C + 1);
    if (int rc = check("iwp_front_bwd", B, C, H, W)) return rc;
    const int total = B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(iwp_front_bwd_kernel, dim3(kmu::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gout, dx, C, Ct, H, W, total);
    return kmu::launch_status("iwp_front_bwd");
}
