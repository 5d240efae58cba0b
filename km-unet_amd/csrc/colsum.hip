// Second stage of the deterministic two-stage reductions: every backward kernel of this library leaves its
// parameter gradients as per-workgroup partial rows [rows][cols]; this kernel column-sums up to 64 such arrays in ONE
// launch (fixed summation order, no atomics).  Replaces one ATen reduce launch per array -- e.g. five per
// HSMSSD backward (d_w_bcdt, d_w_dw, d_w_hz, d_w_out, d_D).
#include "common.h"

namespace {

constexpr int MAXA = 64, COLS = 32, PARTS = 8;     // 64 arrays: 1.8 KB of kernel arguments

struct ColsumArgs {
    const float* src[MAXA];
    float* dst[MAXA];
    int rows[MAXA], cols[MAXA], blk0[MAXA + 1];
    int n;
};

// 256 threads = 32 columns x 8 row partitions; partition sums meet in LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumArgs a) {
    __shared__ float part[PARTS][COLS];
    int k = 0;
    while (k + 1 < a.n && (int)blockIdx.x >= a.blk0[k + 1]) ++k;
    const int c = (blockIdx.x - a.blk0[k]) * COLS + (threadIdx.x & (COLS - 1)), pt = threadIdx.x / COLS;
    const int rows = a.rows[k], cols = a.cols[k];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        const float* p = a.src[k] + c;
        int r = pt;
        for (; r + 3 * PARTS < rows; r += 4 * PARTS) {
            s0 += p[(size_t)r * cols];
            s1 += p[(size_t)(r + PARTS) * cols];
            s2 += p[(size_t)(r + 2 * PARTS) * cols];
            s3 += p[(size_t)(r + 3 * PARTS) * cols];
        }
        for (; r < rows; r += PARTS) s0 += p[(size_t)r * cols];
    }
    part[pt][threadIdx.x & (COLS - 1)] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (pt == 0 && c < cols) {
        float s = part[0][threadIdx.x];
#pragma unroll
        for (int j = 1; j < PARTS; ++j) s += part[j][threadIdx.x];
        a.dst[k][c] = s;
    }
}

}  // namespace

extern "C" int kmu_colsum_multi(int n, const float* const* srcs, float* const* dsts, const int* rows, const int* cols,
                                kmu_stream_t stream) {
    KMU_REQUIRE(n > 0 && n <= MAXA, "colsum_multi: %d arrays (1..%d supported)", n, MAXA);
    KMU_REQUIRE(srcs && dsts && rows && cols, "colsum_multi: null pointer");
    ColsumArgs a;
    a.n = n;
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        KMU_REQUIRE(srcs[k] && dsts[k] && rows[k] > 0 && cols[k] > 0, "colsum_multi: array %d is empty or null", k);
        a.src[k] = srcs[k];
        a.dst[k] = dsts[k];
        a.rows[k] = rows[k];
        a.cols[k] = cols[k];
        a.blk0[k] = blocks;
        blocks += kmu::cdiv(cols[k], COLS);
    }
    a.blk0[n] = blocks;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return kmu::launch_status("colsum_multi");
}
