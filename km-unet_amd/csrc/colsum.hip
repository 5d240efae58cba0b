// Second stage of the deterministic two-stage reductions: every backward kernel of this library leaves its
// parameter gradients as per-workgroup partial rows [rows][cols]; this kernel column-sums up to 64 such arrays in ONE
// launch (fixed summation order, no atomics).  Replaces one ATen reduce launch per array -- e.g. five per
// HSMSSD backward (d_w_bcdt, d_w_dw, d_w_hz, d_w_out, d_D).
#include "common.h"

namespace {

constexpr int MAXA = 64, COLS = 32, PARTS = 8;     // 64 arrays: 2.8 KB of kernel arguments

struct ColsumArgs {
    const float* src[MAXA];
    float* dst[MAXA];
    // row r of array k starts at (r / inner) * ostride + (r % inner) * stride: packed rows (stride = cols, inner = rows), a column
    // range of a wider array (stride > cols), or the rows of ONE weight group of a [samples / G][G][tiles] partial array
    int rows[MAXA], cols[MAXA], stride[MAXA], inner[MAXA], blk0[MAXA + 1];
    long long ostride[MAXA];
    int n;
};

// 256 threads = 32 columns x 8 row partitions; partition sums meet in LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumArgs a) {
    __shared__ float part[PARTS][COLS];
    int k = 0;
    while (k + 1 < a.n && (int)blockIdx.x >= a.blk0[k + 1]) ++k;
    const int c = (blockIdx.x - a.blk0[k]) * COLS + (threadIdx.x & (COLS - 1)), pt = threadIdx.x / COLS;
    const int rows = a.rows[k], cols = a.cols[k], rs = a.stride[k], inner = a.inner[k];
    const long long os = a.ostride[k];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        const float* p = a.src[k] + c;
        int r = pt;
        if (inner >= rows) {
            for (; r + 3 * PARTS < rows; r += 4 * PARTS) {
                s0 += p[(size_t)r * rs];
                s1 += p[(size_t)(r + PARTS) * rs];
                s2 += p[(size_t)(r + 2 * PARTS) * rs];
                s3 += p[(size_t)(r + 3 * PARTS) * rs];
            }
            for (; r < rows; r += PARTS) s0 += p[(size_t)r * rs];
        } else {
            for (; r < rows; r += PARTS) s0 += p[(size_t)(r / inner) * os + (size_t)(r % inner) * rs];
        }
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

// ---- gather copy: the step's parameter gradients (664 small tensors) into their slots of the flat gradient bucket.  ATen's
// _foreach_copy_ needs 11 launches / ~150 us for them at the very end of the step (nothing overlaps it); here the (src, dst, n)
// triples ride in the kernel arguments, 160 per launch, one 256-thread block per 4096-element chunk.
constexpr int CPY_MAX = 160, CPY_CHUNK = 4096;
struct CopyArgs {
    const float* src[CPY_MAX];
    float* dst[CPY_MAX];
    int n[CPY_MAX], blk0[CPY_MAX + 1];
    int count;
};
static_assert(sizeof(CopyArgs) <= 4096, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(256) void copy_multi_kernel(CopyArgs a) {
    int lo = 0, hi = a.count;                       // blk0[lo] <= blockIdx.x < blk0[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= a.blk0[mid]) lo = mid;
        else hi = mid;
    }
    const float* __restrict__ s = a.src[lo];
    float* __restrict__ d = a.dst[lo];
    const int n = a.n[lo], e0 = (blockIdx.x - a.blk0[lo]) * CPY_CHUNK;
    const int e1 = e0 + CPY_CHUNK < n ? e0 + CPY_CHUNK : n;
    if ((((size_t)s | (size_t)d) & 15) == 0) {
        const int v1 = e0 + ((e1 - e0) & ~3);
        for (int e = e0 + 4 * threadIdx.x; e < v1; e += 1024) *reinterpret_cast<kmu::floatx4*>(d + e) = *reinterpret_cast<const kmu::floatx4*>(s + e);
        for (int e = v1 + threadIdx.x; e < e1; e += 256) d[e] = s[e];
    } else {
        for (int e = e0 + threadIdx.x; e < e1; e += 256) d[e] = s[e];
    }
}

// out = ((a + b) + c) + d, elementwise, c / d optional: the fan-in of a tensor with up to four consumers in one launch (autograd
// accumulates pairwise: three adds for EnhancedViMBlock's x, which feeds the three direction branches and the residual)
__global__ __launch_bounds__(256) void add_n_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                    const float* __restrict__ d, float* __restrict__ out, size_t n4, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        kmu::floatx4 v = reinterpret_cast<const kmu::floatx4*>(a)[i] + reinterpret_cast<const kmu::floatx4*>(b)[i];
        if (c) v += reinterpret_cast<const kmu::floatx4*>(c)[i];
        if (d) v += reinterpret_cast<const kmu::floatx4*>(d)[i];
        reinterpret_cast<kmu::floatx4*>(out)[i] = v;
    }
    if (blockIdx.x == 0)
        for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 256) {
            float v = a[i] + b[i];
            if (c) v += c[i];
            if (d) v += d[i];
            out[i] = v;
        }
}

// bias gradients of the plain convolutions: out[c] = sum over (b, pixel) of dy[b][c][pixel] for up to 32 tensors in one launch (ATen
// ran one reduce_kernel per convolution in the backward's tail: 27 launches, 0.28 ms of kernel time per step).  One workgroup per
// (tensor, channel), fixed summation order.
constexpr int BS_MAX = 32;
struct BiasSumArgs {
    const float* src[BS_MAX];
    float* dst[BS_MAX];
    int B[BS_MAX], C[BS_MAX], HW[BS_MAX], blk0[BS_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void bias_sum_multi_kernel(BiasSumArgs a) {
    __shared__ float red[4];
    int k = 0;
    while (k + 1 < a.n && (int)blockIdx.x >= a.blk0[k + 1]) ++k;
    const int c = blockIdx.x - a.blk0[k], B = a.B[k], C = a.C[k], HW = a.HW[k];
    const float* __restrict__ s = a.src[k];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if ((HW & 3) == 0 && (((size_t)s) & 15) == 0) {
        const int q = HW >> 2;
        for (int b = 0; b < B; ++b) {
            const kmu::floatx4* p = reinterpret_cast<const kmu::floatx4*>(s + ((size_t)b * C + c) * HW);
            for (int i = threadIdx.x; i < q; i += 256) {
                const kmu::floatx4 v = p[i];
                s0 += v[0], s1 += v[1], s2 += v[2], s3 += v[3];
            }
        }
    } else {
        for (int b = 0; b < B; ++b) {
            const float* p = s + ((size_t)b * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += 256) s0 += p[i];
        }
    }
    float v = kmu::wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) a.dst[k][c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out = y > 0 ? dy : 0 -- the ReLU behind StableHybridKANConv's residual add (KM_UNetV3_SH.py:91-94) in the backward: ATen ran `gt` + `mul`
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                                                        size_t n4, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const kmu::floatx4 g = reinterpret_cast<const kmu::floatx4*>(dy)[i], v = reinterpret_cast<const kmu::floatx4*>(y)[i];
        reinterpret_cast<kmu::floatx4*>(out)[i] = kmu::floatx4{v[0] > 0.f ? g[0] : 0.f, v[1] > 0.f ? g[1] : 0.f, v[2] > 0.f ? g[2] : 0.f,
                                                              v[3] > 0.f ? g[3] : 0.f};
    }
    if (blockIdx.x == 0)
        for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 256) out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

}  // namespace

extern "C" int kmu_relu_mask(const float* dy, const float* y, float* out, long long numel, kmu_stream_t stream) {
    KMU_REQUIRE(dy && y && out && numel > 0, "relu_mask: bad arguments");
    const bool vec = ((((size_t)dy | (size_t)y | (size_t)out) & 15) == 0);
    const size_t n4 = vec ? (size_t)numel / 4 : 0;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y, out, n4, (size_t)numel);
    return kmu::launch_status("relu_mask");
}

extern "C" int kmu_bias_sum_multi(int n, const float* const* srcs, float* const* dsts, const int* B, const int* C, const int* HW,
                                  kmu_stream_t stream) {
    KMU_REQUIRE(n > 0 && n <= BS_MAX && srcs && dsts && B && C && HW, "bias_sum_multi: %d tensors (1..%d supported)", n, BS_MAX);
    BiasSumArgs a;
    a.n = n;
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        KMU_REQUIRE(srcs[k] && dsts[k] && B[k] > 0 && C[k] > 0 && HW[k] > 0, "bias_sum_multi: tensor %d is empty or null", k);
        a.src[k] = srcs[k], a.dst[k] = dsts[k], a.B[k] = B[k], a.C[k] = C[k], a.HW[k] = HW[k];
        a.blk0[k] = blocks;
        blocks += C[k];
    }
    a.blk0[n] = blocks;
    hipLaunchKernelGGL(bias_sum_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return kmu::launch_status("bias_sum_multi");
}

extern "C" int kmu_add_n(const float* a, const float* b, const float* c, const float* d, float* out, long long numel, kmu_stream_t stream) {
    KMU_REQUIRE(a && b && out && numel > 0 && (c || !d), "add_n: needs a, b, out (c before d)");
    const bool vec = ((((size_t)a | (size_t)b | (size_t)c | (size_t)d | (size_t)out) & 15) == 0);
    const size_t n4 = vec ? (size_t)numel / 4 : 0;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(add_n_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, d, out, n4, (size_t)numel);
    return kmu::launch_status("add_n");
}

extern "C" int kmu_copy_multi(int count, const float* const* srcs, float* const* dsts, const long long* numel, kmu_stream_t stream) {
    KMU_REQUIRE(count > 0 && srcs && dsts && numel, "copy_multi: bad arguments");
    for (int first = 0; first < count; first += CPY_MAX) {
        CopyArgs a;
        a.count = count - first < CPY_MAX ? count - first : CPY_MAX;
        int blocks = 0;
        for (int k = 0; k < a.count; ++k) {
            const long long n = numel[first + k];
            KMU_REQUIRE(srcs[first + k] && dsts[first + k] && n > 0 && n < (1ll << 31), "copy_multi: tensor %d is empty, null or too large", first + k);
            a.src[k] = srcs[first + k];
            a.dst[k] = dsts[first + k];
            a.n[k] = (int)n;
            a.blk0[k] = blocks;
            blocks += (int)((n + CPY_CHUNK - 1) / CPY_CHUNK);
        }
        a.blk0[a.count] = blocks;
        hipLaunchKernelGGL(copy_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    return kmu::launch_status("copy_multi");
}

// strides[k] = elements between two partial rows of array k (>= cols[k]; NULL: every array is packed, stride = cols): column sums of a
// column RANGE of a wider partial array without a copy.  inner[k] / ostrides[k] (NULL: one run): the rows come in runs of inner[k],
// run i starting i * ostrides[k] elements in -- the rows of one weight group of a grouped launch's [samples / G][G][tiles] partials.
extern "C" int kmu_colsum_multi_strided(int n, const float* const* srcs, float* const* dsts, const int* rows, const int* cols,
                                        const int* strides, const int* inner, const long long* ostrides, kmu_stream_t stream) {
    KMU_REQUIRE(n > 0 && n <= MAXA, "colsum_multi: %d arrays (1..%d supported)", n, MAXA);
    KMU_REQUIRE(srcs && dsts && rows && cols && ((inner == nullptr) == (ostrides == nullptr)), "colsum_multi: null pointer");
    ColsumArgs a;
    a.n = n;
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        KMU_REQUIRE(srcs[k] && dsts[k] && rows[k] > 0 && cols[k] > 0, "colsum_multi: array %d is empty or null", k);
        KMU_REQUIRE(!strides || strides[k] >= cols[k], "colsum_multi: array %d has row stride %d < %d columns", k, strides ? strides[k] : 0, cols[k]);
        KMU_REQUIRE(!inner || inner[k] > 0, "colsum_multi: array %d has runs of %d rows", k, inner ? inner[k] : 0);
        a.src[k] = srcs[k];
        a.dst[k] = dsts[k];
        a.rows[k] = rows[k];
        a.cols[k] = cols[k];
        a.stride[k] = strides ? strides[k] : cols[k];
        a.inner[k] = inner ? inner[k] : rows[k];
        a.ostride[k] = ostrides ? ostrides[k] : 0;
        a.blk0[k] = blocks;
        blocks += kmu::cdiv(cols[k], COLS);
    }
    a.blk0[n] = blocks;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return kmu::launch_status("colsum_multi");
}
extern "C" int kmu_colsum_multi(int n, const float* const* srcs, float* const* dsts, const int* rows, const int* cols,
                                kmu_stream_t stream) {
    return kmu_colsum_multi_strided(n, srcs, dsts, rows, cols, nullptr, nullptr, nullptr, stream);
}
