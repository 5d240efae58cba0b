// EnhancedViMBlock's branch fusion (KM_UNetV3_SH.py:141-146):
//     out[b] = x[b] + s[b] * (g[b,0] f0[b] + g[b,1] f1[b] + g[b,2] f2[b])
// g = softmax fusion weights [B,3], s = DropPath's per-sample mask / keep_prob (NULL = 1).  Stock ATen: 3 mul + 2 add
// + mul + add forward, ~12 launches backward; here one streaming kernel each way (backward also leaves the per-block
// partial dot products for d g).  HBM-bound: forward reads 4 and writes 1 tensor, backward reads 4 and writes 3.
#include "common.h"

using kmu::floatx4;

namespace {

constexpr int VPT = 4;   // float4s per thread => 4096 floats per 256-thread block

__global__ __launch_bounds__(256) void mix3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ f0,
                                                       const float* __restrict__ f1, const float* __restrict__ f2,
                                                       const float* __restrict__ g, const float* __restrict__ s, float* __restrict__ out,
                                                       int n4, int fs4) {
    // fs4: per-sample stride of f0 / f1 / f2 in float4s (n4 for separate tensors, 3*n4 for channel slices of one [B,3C,H,W])
    const int b = blockIdx.y;
    const float sc = s ? s[b] : 1.f;
    const float k0 = sc * g[b * 3], k1 = sc * g[b * 3 + 1], k2 = sc * g[b * 3 + 2];
    const size_t base = (size_t)b * n4, fb = (size_t)b * fs4;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int i = (blockIdx.x * VPT + v) * 256 + threadIdx.x;
        if (i >= n4) break;
        const floatx4 xv = reinterpret_cast<const floatx4*>(x)[base + i], a = reinterpret_cast<const floatx4*>(f0)[fb + i],
                      bb = reinterpret_cast<const floatx4*>(f1)[fb + i], c = reinterpret_cast<const floatx4*>(f2)[fb + i];
        reinterpret_cast<floatx4*>(out)[base + i] = xv + k0 * a + k1 * bb + k2 * c;
    }
}

// pooled[b][t*C + c] = mean_{hw} f_t[b][c][hw] for up to three tensors in one launch (the fusion gate pools the channel
// concat of the three branches: KM_UNetV3_SH.py:111-117 -- AdaptiveAvgPool2d(1) commutes with the concat); one workgroup per
// (t, b, c) row, fixed-order tree => deterministic.  Also serves the single-tensor squeeze-excite pools (:231, :320, :342).
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                        const float* __restrict__ f2, float* __restrict__ pooled, int C, int HW, int NT) {
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y, t = blockIdx.z;
    const float* src = (t == 0 ? f0 : (t == 1 ? f1 : f2)) + ((size_t)b * C + c) * HW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if ((HW & 3) == 0) {
        const floatx4* s4 = reinterpret_cast<const floatx4*>(src);
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            const floatx4 v = s4[i];
            a0 += v[0], a1 += v[1], a2 += v[2], a3 += v[3];
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) a0 += src[i];
    }
    const float w = kmu::wave_sum((a0 + a1) + (a2 + a3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) pooled[(size_t)b * NT * C + t * C + c] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)HW;
}

// second half of the composite node's backward:  d f_t[b][c][hw] = (s g_t)[b] dy[b][c][hw] + d pooled[b][t*C + c] / HW
__global__ __launch_bounds__(256) void mix3_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ g,
                                                             const float* __restrict__ s, const float* __restrict__ dpool,
                                                             float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2,
                                                             int n4, int C, int HW4, float inv_hw, int fs4) {
    const int b = blockIdx.y;
    const float sc = s ? s[b] : 1.f;
    const float k0 = sc * g[b * 3], k1 = sc * g[b * 3 + 1], k2 = sc * g[b * 3 + 2];
    const size_t base = (size_t)b * n4, fb = (size_t)b * fs4;
    const float* dp = dpool + (size_t)b * 3 * C;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int i = (blockIdx.x * VPT + v) * 256 + threadIdx.x;
        if (i >= n4) break;
        const int c = i / HW4;
        const floatx4 gy = reinterpret_cast<const floatx4*>(dy)[base + i];
        reinterpret_cast<floatx4*>(d0)[fb + i] = k0 * gy + dp[c] * inv_hw;
        reinterpret_cast<floatx4*>(d1)[fb + i] = k1 * gy + dp[C + c] * inv_hw;
        reinterpret_cast<floatx4*>(d2)[fb + i] = k2 * gy + dp[2 * C + c] * inv_hw;
    }
}

// df_i = (s g_i) dy ;  part[blockIdx.x][b*3 + i] = s * sum_block dy . f_i        (WRITE = false: the partial sums only)
template <bool WRITE>
__global__ __launch_bounds__(256) void mix3_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ f0,
                                                       const float* __restrict__ f1, const float* __restrict__ f2,
                                                       const float* __restrict__ g, const float* __restrict__ s, float* __restrict__ d0,
                                                       float* __restrict__ d1, float* __restrict__ d2, float* __restrict__ part, int n4,
                                                       int B, int fs4) {
    __shared__ float red[4][3];
    const int b = blockIdx.y;
    const float sc = s ? s[b] : 1.f;
    const float k0 = sc * g[b * 3], k1 = sc * g[b * 3 + 1], k2 = sc * g[b * 3 + 2];
    const size_t base = (size_t)b * n4, fb = (size_t)b * fs4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int i = (blockIdx.x * VPT + v) * 256 + threadIdx.x;
        if (i >= n4) break;
        const floatx4 gy = reinterpret_cast<const floatx4*>(dy)[base + i], a = reinterpret_cast<const floatx4*>(f0)[fb + i],
                      bb = reinterpret_cast<const floatx4*>(f1)[fb + i], c = reinterpret_cast<const floatx4*>(f2)[fb + i];
        if (WRITE) {
            reinterpret_cast<floatx4*>(d0)[base + i] = k0 * gy;
            reinterpret_cast<floatx4*>(d1)[base + i] = k1 * gy;
            reinterpret_cast<floatx4*>(d2)[base + i] = k2 * gy;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0 += gy[q] * a[q];
            a1 += gy[q] * bb[q];
            a2 += gy[q] * c[q];
        }
    }
    a0 = kmu::wave_sum(a0), a1 = kmu::wave_sum(a1), a2 = kmu::wave_sum(a2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave][0] = a0, red[wave][1] = a1, red[wave][2] = a2;
    __syncthreads();
    if (threadIdx.x < 3)
        part[(size_t)blockIdx.x * B * 3 + b * 3 + threadIdx.x] =
            sc * ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// LocalContrastAttention's output (KM_UNetV3_SH.py:366-368): x * (1 - g) + g with g [B,C] the sigmoid gate of the pooled channel
// groups (torch.lerp(x, 1, g)).  One workgroup per (b, c) plane; the backward writes dx = dy (1 - g) and reduces
// dg[b,c] = sum_p dy (1 - x) in the same pass (deterministic).
__global__ __launch_bounds__(256) void lca_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ y, int HW) {
    const size_t base = (size_t)blockIdx.x * HW;
    const float gv = g[blockIdx.x], a = 1.f - gv;
    if ((HW & 3) == 0) {
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            const floatx4 v = reinterpret_cast<const floatx4*>(x + base)[i];
            reinterpret_cast<floatx4*>(y + base)[i] = v * a + gv;
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) y[base + i] = x[base + i] * a + gv;
    }
}

__global__ __launch_bounds__(256) void lca_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ dy,
                                                      float* __restrict__ dx, float* __restrict__ dg, int HW) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * HW;
    const float a = 1.f - g[blockIdx.x];
    float s = 0.f;
    if ((HW & 3) == 0) {
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            const floatx4 v = reinterpret_cast<const floatx4*>(x + base)[i], d = reinterpret_cast<const floatx4*>(dy + base)[i];
            reinterpret_cast<floatx4*>(dx + base)[i] = d * a;
            s += (d[0] * (1.f - v[0]) + d[1] * (1.f - v[1])) + (d[2] * (1.f - v[2]) + d[3] * (1.f - v[3]));
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) {
            const float d = dy[base + i];
            dx[base + i] = d * a;
            s += d * (1.f - x[base + i]);
        }
    }
    s = kmu::wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dg[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" int kmu_mix3_blocks(int n_per_sample) { return n_per_sample > 0 ? kmu::cdiv(n_per_sample / 4, 256 * VPT) : 0; }

extern "C" int kmu_mix3_fwd(const float* x, const float* f0, const float* f1, const float* f2, const float* g, const float* s, float* out,
                            int B, int n_per_sample, kmu_stream_t stream) {
    KMU_REQUIRE(x && f0 && f1 && f2 && g && out, "mix3_fwd: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && n_per_sample > 0 && n_per_sample % 4 == 0, "mix3_fwd: C*H*W = %d must be a positive multiple of 4",
                n_per_sample);
    hipLaunchKernelGGL(mix3_fwd_kernel, dim3(kmu_mix3_blocks(n_per_sample), B), dim3(256), 0, (hipStream_t)stream, x, f0, f1, f2, g, s, out,
                       n_per_sample / 4, n_per_sample / 4);
    return kmu::launch_status("mix3_fwd");
}

extern "C" int kmu_mix3_bwd(const float* dy, const float* f0, const float* f1, const float* f2, const float* g, const float* s, float* d_f0,
                            float* d_f1, float* d_f2, float* d_g_partial, int B, int n_per_sample, kmu_stream_t stream) {
    KMU_REQUIRE(dy && f0 && f1 && f2 && g && d_f0 && d_f1 && d_f2 && d_g_partial, "mix3_bwd: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && n_per_sample > 0 && n_per_sample % 4 == 0, "mix3_bwd: C*H*W = %d must be a positive multiple of 4",
                n_per_sample);
    hipLaunchKernelGGL(mix3_bwd_kernel<true>, dim3(kmu_mix3_blocks(n_per_sample), B), dim3(256), 0, (hipStream_t)stream, dy, f0, f1, f2, g, s, d_f0,
                       d_f1, d_f2, d_g_partial, n_per_sample / 4, B, n_per_sample / 4);
    return kmu::launch_status("mix3_bwd");
}

// ---- the fusion gate + branch mix as ONE autograd node (KM_UNetV3_SH.py:111-117 + :141-146) -------------------------------
//   forward : pooled = mean_hw(cat(f0, f1, f2))  ->  [gate MLP: kmu_gate_mlp_fwd]  ->  kmu_mix3_fwd
//   backward: kmu_mix3_bwd_dg (partials of d g)  ->  [kmu_gate_mlp_bwd: d pooled]  ->  kmu_mix3_bwd_apply
// instead of 3 mean + cat forward and 3 (grad / HW).expand + 3 fan-in adds backward as separate full-tensor launches.
extern "C" int kmu_mean_rows(const float* f0, const float* f1, const float* f2, float* pooled, int B, int C, int HW, int n_tensors,
                             kmu_stream_t stream) {
    KMU_REQUIRE(f0 && pooled && (n_tensors < 2 || f1) && (n_tensors < 3 || f2), "mean_rows: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && C > 0 && HW > 0 && n_tensors >= 1 && n_tensors <= 3, "mean_rows: bad dims B=%d C=%d HW=%d n=%d", B, C,
                HW, n_tensors);
    hipLaunchKernelGGL(mean_rows_kernel, dim3(C, B, n_tensors), dim3(256), 0, (hipStream_t)stream, f0, f1, f2, pooled, C, HW, n_tensors);
    return kmu::launch_status("mean_rows");
}

extern "C" int kmu_mix3_bwd_dg(const float* dy, const float* f0, const float* f1, const float* f2, const float* g, const float* s,
                               float* d_g_partial, int B, int n_per_sample, kmu_stream_t stream) {
    KMU_REQUIRE(dy && f0 && f1 && f2 && g && d_g_partial, "mix3_bwd_dg: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && n_per_sample > 0 && n_per_sample % 4 == 0, "mix3_bwd_dg: C*H*W = %d must be a positive multiple of 4",
                n_per_sample);
    hipLaunchKernelGGL(mix3_bwd_kernel<false>, dim3(kmu_mix3_blocks(n_per_sample), B), dim3(256), 0, (hipStream_t)stream, dy, f0, f1, f2, g, s,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, d_g_partial, n_per_sample / 4, B, n_per_sample / 4);
    return kmu::launch_status("mix3_bwd_dg");
}

extern "C" int kmu_mix3_bwd_apply(const float* dy, const float* g, const float* s, const float* d_pooled, float* d_f0, float* d_f1,
                                  float* d_f2, int B, int C, int HW, kmu_stream_t stream) {
    KMU_REQUIRE(dy && g && d_pooled && d_f0 && d_f1 && d_f2, "mix3_bwd_apply: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && C > 0 && HW > 0 && HW % 4 == 0, "mix3_bwd_apply: H*W = %d must be a positive multiple of 4", HW);
    const int n = C * HW;
    hipLaunchKernelGGL(mix3_bwd_apply_kernel, dim3(kmu_mix3_blocks(n), B), dim3(256), 0, (hipStream_t)stream, dy, g, s, d_pooled, d_f0, d_f1,
                       d_f2, n / 4, C, HW / 4, 1.0f / (float)HW, n / 4);
    return kmu::launch_status("mix3_bwd_apply");
}

// ---- the same three kernels on channel slices of ONE stacked tensor F [B, 3C, H, W] (the grouped direction branches write
// their outputs side by side): f_t = F + t*C*HW with a per-sample stride of 3*C*HW; d_f_t likewise into one stacked gradient.
extern "C" int kmu_mix3_fwd_stacked(const float* x, const float* F, const float* g, const float* s, float* out, int B, int n_per_sample,
                                    kmu_stream_t stream) {
    KMU_REQUIRE(x && F && g && out, "mix3_fwd_stacked: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && n_per_sample > 0 && n_per_sample % 4 == 0, "mix3_fwd_stacked: C*H*W = %d must be a positive multiple of 4",
                n_per_sample);
    const int n = n_per_sample;
    hipLaunchKernelGGL(mix3_fwd_kernel, dim3(kmu_mix3_blocks(n), B), dim3(256), 0, (hipStream_t)stream, x, F, F + n, F + 2 * (size_t)n, g, s,
                       out, n / 4, 3 * (n / 4));
    return kmu::launch_status("mix3_fwd_stacked");
}

extern "C" int kmu_mix3_bwd_dg_stacked(const float* dy, const float* F, const float* g, const float* s, float* d_g_partial, int B,
                                       int n_per_sample, kmu_stream_t stream) {
    KMU_REQUIRE(dy && F && g && d_g_partial, "mix3_bwd_dg_stacked: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && n_per_sample > 0 && n_per_sample % 4 == 0, "mix3_bwd_dg_stacked: C*H*W = %d must be a positive multiple of 4",
                n_per_sample);
    const int n = n_per_sample;
    hipLaunchKernelGGL(mix3_bwd_kernel<false>, dim3(kmu_mix3_blocks(n), B), dim3(256), 0, (hipStream_t)stream, dy, F, F + n, F + 2 * (size_t)n,
                       g, s, (float*)nullptr, (float*)nullptr, (float*)nullptr, d_g_partial, n / 4, B, 3 * (n / 4));
    return kmu::launch_status("mix3_bwd_dg_stacked");
}

extern "C" int kmu_mix3_bwd_apply_stacked(const float* dy, const float* g, const float* s, const float* d_pooled, float* dF, int B, int C,
                                          int HW, kmu_stream_t stream) {
    KMU_REQUIRE(dy && g && d_pooled && dF, "mix3_bwd_apply_stacked: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && C > 0 && HW > 0 && HW % 4 == 0, "mix3_bwd_apply_stacked: H*W = %d must be a positive multiple of 4", HW);
    const int n = C * HW;
    hipLaunchKernelGGL(mix3_bwd_apply_kernel, dim3(kmu_mix3_blocks(n), B), dim3(256), 0, (hipStream_t)stream, dy, g, s, d_pooled, dF, dF + n,
                       dF + 2 * (size_t)n, n / 4, C, HW / 4, 1.0f / (float)HW, 3 * (n / 4));
    return kmu::launch_status("mix3_bwd_apply_stacked");
}

extern "C" int kmu_lca_fwd(const float* x, const float* g, float* y, int B, int C, int HW, kmu_stream_t stream) {
    KMU_REQUIRE(x && g && y, "lca_fwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && HW > 0, "lca_fwd: bad dims");
    hipLaunchKernelGGL(lca_fwd_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, x, g, y, HW);
    return kmu::launch_status("lca_fwd");
}

extern "C" int kmu_lca_bwd(const float* x, const float* g, const float* dy, float* dx, float* dg, int B, int C, int HW,
                           kmu_stream_t stream) {
    KMU_REQUIRE(x && g && dy && dx && dg, "lca_bwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && HW > 0, "lca_bwd: bad dims");
    hipLaunchKernelGGL(lca_bwd_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, x, g, dy, dx, dg, HW);
    return kmu::launch_status("lca_bwd");
}
