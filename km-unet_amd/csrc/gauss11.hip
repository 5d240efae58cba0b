// Separable 11-tap Gaussian window of the SSIM term of HybridLoss (train_shanghai.py:298-325; torchmetrics'
// StructuralSimilarityIndexMeasure: gaussian_kernel=True, kernel_size 11, sigma 1.5, applied as a depthwise conv).
//     out[n][y][x] = sum_{i,j} g[i] g[j] in[n][y + i - pad][x + j - pad]        (zero outside the input)
//   pad = 0  : the 'valid' forward filter,  out = (H-10) x (W-10)
//   pad = 10 : its adjoint (the taps are symmetric), out = (H+10) x (W+10) -- the input gradient.
// MIOpen serves this depthwise fp32 shape with its naive kernels (150 us forward, 320 us backward on [40,5,138,138]);
// this is an HBM-bound stencil: a 32x32 output tile stages its 42x42 input window in LDS, filters rows, then columns.
#include "common.h"

namespace {

constexpr int KT = 11, TILE = 32, WIN = TILE + KT - 1;   // 42

__global__ __launch_bounds__(256) void gauss11_kernel(const float* __restrict__ in, const float* __restrict__ taps, float* __restrict__ out,
                                                      int H, int W, int Ho, int Wo, int pad) {
    __shared__ float win[WIN][WIN + 1];
    __shared__ float rowf[WIN][TILE + 1];
    __shared__ float g[KT];
    const int tid = threadIdx.x, n = blockIdx.z;
    const int y0 = blockIdx.y * TILE, x0 = blockIdx.x * TILE;
    if (tid < KT) g[tid] = taps[tid];
    const float* src = in + (size_t)n * H * W;
    for (int e = tid; e < WIN * WIN; e += 256) {
        const int r = e / WIN, c = e - r * WIN;
        const int yy = y0 + r - pad, xx = x0 + c - pad;
        win[r][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[(size_t)yy * W + xx] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < WIN * TILE; e += 256) {          // rows: rowf[r][x] = sum_j g[j] win[r][x + j]
        const int r = e / TILE, x = e - r * TILE;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j) s += g[j] * win[r][x + j];
        rowf[r][x] = s;
    }
    __syncthreads();
    float* dst = out + (size_t)n * Ho * Wo;
    for (int e = tid; e < TILE * TILE; e += 256) {         // columns
        const int y = e / TILE, x = e - y * TILE;
        if (y0 + y >= Ho || x0 + x >= Wo) continue;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < KT; ++i) s += g[i] * rowf[y + i][x];
        dst[(size_t)(y0 + y) * Wo + x0 + x] = s;
    }
}

}  // namespace

extern "C" int kmu_gauss11_filter(const float* in, const float* taps, float* out, int N, int H, int W, int adjoint, kmu_stream_t stream) {
    KMU_REQUIRE(in && taps && out, "gauss11_filter: null pointer");
    KMU_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0, "gauss11_filter: bad dims (N=%d, %dx%d)", N, H, W);
    const int pad = adjoint ? KT - 1 : 0;
    const int Ho = H + 2 * pad - (KT - 1), Wo = W + 2 * pad - (KT - 1);
    KMU_REQUIRE(Ho > 0 && Wo > 0, "gauss11_filter: %dx%d is smaller than the 11x11 window", H, W);
    hipLaunchKernelGGL(gauss11_kernel, dim3(kmu::cdiv(Wo, TILE), kmu::cdiv(Ho, TILE), N), dim3(256), 0, (hipStream_t)stream, in, taps, out, H,
                       W, Ho, Wo, pad);
    return kmu::launch_status("gauss11_filter");
}
