// GroupNorm(G, C) over NCHW for gfx950, forward + backward.
//
// Glue-side norm of the reference graph: StableHybridKANConv.pre_norm = GroupNorm(4, C) (KM_UNetV3_SH.py:57,73),
// TripleNorm.norm_h / norm_w = GroupNorm(1, C) (:271-273), MultiScaleFusion GroupNorm(1, C) (:294),
// KM_UNetV3.output_norm = GroupNorm(1, num_classes) (:448,516).  At B = 8 these are 8..32 rows of up to 262144
// contiguous elements; ATen reduces one row per workgroup (59 us per call on MI355X).  Here every (b, c) plane is
// split over up to 8 workgroups; all kernels are streaming, HBM-bound:
//   forward : plane partial sums -> apply (each block folds its row's partials: (C/G)*S pairs)
//   backward: plane partial sums of g and g*xhat -> apply; d_gamma / d_beta leave as per-(b) partials [B,C].
#include "common.h"

using kmu::floatx4;

namespace {

constexpr int MAXS = 8;

inline int splits_for(int HW) {
    int S = (HW + 2047) / 2048;
    return S < 1 ? 1 : (S > MAXS ? MAXS : S);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = kmu::wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// part[(plane*S + s)*2 + {0,1}] = sum x, sum x^2 over this block's slice of plane (b,c)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int HW) {
    __shared__ float red[4];
    const int plane = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const int chunk = ((HW + S - 1) / S + 3) / 4 * 4, p0 = s * chunk, p1 = min(HW, p0 + chunk);
    const float* xp = x + (size_t)plane * HW;
    float a = 0.f, q = 0.f;
    if ((HW & 3) == 0) {
        for (int p = p0 + threadIdx.x * 4; p < p1; p += 1024) {
            const floatx4 v = *reinterpret_cast<const floatx4*>(xp + p);
            a += (v[0] + v[1]) + (v[2] + v[3]);
            q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            const float v = xp[p];
            a += v;
            q += v * v;
        }
    }
    a = block_sum(a, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        part[((size_t)plane * S + s) * 2] = a;
        part[((size_t)plane * S + s) * 2 + 1] = q;
    }
}

// fold the row's partials (double), normalise this block's slice; block (plane, 0) of the row's first channel
// publishes (mean, rstd) for the backward
// ACT = 1: y = SiLU(GroupNorm(x)) (MultiScaleFusion's blocks, KM_UNetV3_SH.py:300-306: conv -> GroupNorm -> SiLU); the backward kernels
// then take g * SiLU'(u), u re-derived from x and the saved statistics with the forward's expression
__device__ __forceinline__ float silu_f(float u) { return u / (1.f + __expf(-u)); }
__device__ __forceinline__ float silu_grad(float u) {
    const float s = 1.f / (1.f + __expf(-u));
    return s * (1.f + u * (1.f - s));
}
// ACT = 2: y = sigmoid(GroupNorm(x)) (the model's output head, KM_UNetV3_SH.py:516-517)
template <int ACT>
__device__ __forceinline__ float act_f(float u) { return ACT == 1 ? silu_f(u) : (ACT == 2 ? 1.f / (1.f + __expf(-u)) : u); }
template <int ACT>
__device__ __forceinline__ float act_grad(float u) {
    if (ACT == 1) return silu_grad(u);
    if (ACT == 2) {
        const float s = 1.f / (1.f + __expf(-u));
        return s * (1.f - s);
    }
    return 1.f;
}
template <int ACT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ part,
                                                       float* __restrict__ y, float* __restrict__ stats, int C, int G,
                                                       int HW, float eps) {
    const int plane = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const int b = plane / C, c = plane % C, cpg = C / G, grp = c / cpg;
    const float* rp = part + ((size_t)(b * C + grp * cpg) * S) * 2;
    double sa = 0.0, sq = 0.0;
    for (int i = 0; i < cpg * S; ++i) {
        sa += rp[i * 2];
        sq += rp[i * 2 + 1];
    }
    const double n = (double)cpg * HW, m = sa / n;
    double var = sq / n - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (s == 0 && c == grp * cpg && threadIdx.x == 0) {
        stats[(b * G + grp) * 2] = mean;
        stats[(b * G + grp) * 2 + 1] = rstd;
    }
    const float scale = gamma[c] * rstd, shift = beta[c] - mean * scale;
    const int chunk = ((HW + S - 1) / S + 3) / 4 * 4, p0 = s * chunk, p1 = min(HW, p0 + chunk);
    const float* xp = x + (size_t)plane * HW;
    float* yp = y + (size_t)plane * HW;
    if ((HW & 3) == 0) {
        for (int p = p0 + threadIdx.x * 4; p < p1; p += 1024) {
            floatx4 v = *reinterpret_cast<const floatx4*>(xp + p);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = v[k] * scale + shift;
                v[k] = act_f<ACT>(u);
            }
            *reinterpret_cast<floatx4*>(yp + p) = v;
        }
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            const float u = xp[p] * scale + shift;
            yp[p] = act_f<ACT>(u);
        }
    }
}

// part[(plane*S+s)*2 + {0,1}] = sum g, sum g*xhat
template <int ACT>
__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ stats, float* __restrict__ part,
                                                          int C, int G, int HW) {
    __shared__ float red[4];
    const int plane = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const int b = plane / C, c = plane % C, grp = c / (C / G);
    const float mean = stats[(b * G + grp) * 2], rstd = stats[(b * G + grp) * 2 + 1];
    const float scale = ACT ? gamma[c] * rstd : 0.f, shift = ACT ? beta[c] - mean * scale : 0.f;
    const int chunk = ((HW + S - 1) / S + 3) / 4 * 4, p0 = s * chunk, p1 = min(HW, p0 + chunk);
    const float* xp = x + (size_t)plane * HW;
    const float* gp = g + (size_t)plane * HW;
    float a = 0.f, q = 0.f;
    if ((HW & 3) == 0) {
        for (int p = p0 + threadIdx.x * 4; p < p1; p += 1024) {
            const floatx4 xv = *reinterpret_cast<const floatx4*>(xp + p), gv = *reinterpret_cast<const floatx4*>(gp + p);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ge = ACT ? gv[k] * act_grad<ACT>(xv[k] * scale + shift) : gv[k];
                a += ge;
                q += ge * ((xv[k] - mean) * rstd);
            }
        }
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            const float ge = ACT ? gp[p] * act_grad<ACT>(xp[p] * scale + shift) : gp[p];
            a += ge;
            q += ge * ((xp[p] - mean) * rstd);
        }
    }
    a = block_sum(a, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        part[((size_t)plane * S + s) * 2] = a;
        part[((size_t)plane * S + s) * 2 + 1] = q;
    }
}

// dx = rstd * (gamma_c*g - mean_row(gamma*g) - xhat * mean_row(gamma*g*xhat)); d_gamma/d_beta partial per (b, c)
template <int ACT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ part, float* __restrict__ dx,
                                                           float* __restrict__ dgamma_part,
                                                           float* __restrict__ dbeta_part, int C, int G, int HW) {
    const int plane = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const int b = plane / C, c = plane % C, cpg = C / G, grp = c / cpg;
    const float mean = stats[(b * G + grp) * 2], rstd = stats[(b * G + grp) * 2 + 1];
    double r0 = 0.0, r1 = 0.0, own0 = 0.0, own1 = 0.0;
    for (int cc = 0; cc < cpg; ++cc) {
        const int ch = grp * cpg + cc;
        const float* pp = part + ((size_t)(b * C + ch) * S) * 2;
        double a = 0.0, q = 0.0;
        for (int i = 0; i < S; ++i) {
            a += pp[i * 2];
            q += pp[i * 2 + 1];
        }
        r0 += gamma[ch] * a;
        r1 += gamma[ch] * q;
        if (ch == c) {
            own0 = a;
            own1 = q;
        }
    }
    if (s == 0 && threadIdx.x == 0) {
        dgamma_part[plane] = (float)own1;
        dbeta_part[plane] = (float)own0;
    }
    const double n = (double)cpg * HW;
    const float m0 = (float)(r0 / n), m1 = (float)(r1 / n), gc = gamma[c];
    const float scale = ACT ? gc * rstd : 0.f, shift = ACT ? beta[c] - mean * scale : 0.f;
    const int chunk = ((HW + S - 1) / S + 3) / 4 * 4, p0 = s * chunk, p1 = min(HW, p0 + chunk);
    const float* xp = x + (size_t)plane * HW;
    const float* gp = g + (size_t)plane * HW;
    float* dp = dx + (size_t)plane * HW;
    if ((HW & 3) == 0) {
        for (int p = p0 + threadIdx.x * 4; p < p1; p += 1024) {
            const floatx4 xv = *reinterpret_cast<const floatx4*>(xp + p), gv = *reinterpret_cast<const floatx4*>(gp + p);
            floatx4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ge = ACT ? gv[k] * act_grad<ACT>(xv[k] * scale + shift) : gv[k];
                o[k] = rstd * (gc * ge - m0 - ((xv[k] - mean) * rstd) * m1);
            }
            *reinterpret_cast<floatx4*>(dp + p) = o;
        }
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            const float ge = ACT ? gp[p] * act_grad<ACT>(xp[p] * scale + shift) : gp[p];
            dp[p] = rstd * (gc * ge - m0 - ((xp[p] - mean) * rstd) * m1);
        }
    }
}

}  // namespace

extern "C" int kmu_group_norm_splits(int HW) { return splits_for(HW); }

// act: 0 = GroupNorm, 1 = SiLU(GroupNorm(x)), 2 = sigmoid(GroupNorm(x)).  beta is needed by the backward when act != 0 (u is re-derived from x).
extern "C" int kmu_group_norm_act_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* ws, int B, int C,
                                      int G, int HW, float eps, int act, kmu_stream_t stream) {
    KMU_REQUIRE(x && gamma && beta && y && stats && ws, "group_norm_fwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && HW > 0, "group_norm_fwd: bad dims (C=%d, G=%d)", C, G);
    KMU_REQUIRE(act >= 0 && act <= 2, "group_norm_fwd: act must be 0 (none), 1 (SiLU) or 2 (sigmoid)");
    hipStream_t st = (hipStream_t)stream;
    const int S = splits_for(HW);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * C, S), dim3(256), 0, st, x, ws, HW);
    int rc = kmu::launch_status("group_norm_fwd stats");
    if (rc) return rc;
    if (act == 1) hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(B * C, S), dim3(256), 0, st, x, gamma, beta, ws, y, stats, C, G, HW, eps);
    else if (act == 2) hipLaunchKernelGGL(gn_apply_kernel<2>, dim3(B * C, S), dim3(256), 0, st, x, gamma, beta, ws, y, stats, C, G, HW, eps);
    else hipLaunchKernelGGL(gn_apply_kernel<0>, dim3(B * C, S), dim3(256), 0, st, x, gamma, beta, ws, y, stats, C, G, HW, eps);
    return kmu::launch_status("group_norm_fwd apply");
}
extern "C" int kmu_group_norm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                  float* ws, int B, int C, int G, int HW, float eps, kmu_stream_t stream) {
    return kmu_group_norm_act_fwd(x, gamma, beta, y, stats, ws, B, C, G, HW, eps, 0, stream);
}

extern "C" int kmu_group_norm_act_bwd(const float* x, const float* gout, const float* gamma, const float* beta, const float* stats, float* dx,
                                      float* d_gamma_partial, float* d_beta_partial, float* ws, int B, int C, int G, int HW, int act,
                                      kmu_stream_t stream) {
    KMU_REQUIRE(x && gout && gamma && stats && dx && d_gamma_partial && d_beta_partial && ws, "group_norm_bwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && HW > 0, "group_norm_bwd: bad dims");
    KMU_REQUIRE(act == 0 || ((act == 1 || act == 2) && beta), "group_norm_bwd: act must be 0, 1 (SiLU) or 2 (sigmoid); 1 / 2 need beta");
    hipStream_t st = (hipStream_t)stream;
    const int S = splits_for(HW);
    if (act == 1) hipLaunchKernelGGL(gn_bwd_sums_kernel<1>, dim3(B * C, S), dim3(256), 0, st, x, gout, gamma, beta, stats, ws, C, G, HW);
    else if (act == 2) hipLaunchKernelGGL(gn_bwd_sums_kernel<2>, dim3(B * C, S), dim3(256), 0, st, x, gout, gamma, beta, stats, ws, C, G, HW);
    else hipLaunchKernelGGL(gn_bwd_sums_kernel<0>, dim3(B * C, S), dim3(256), 0, st, x, gout, gamma, beta, stats, ws, C, G, HW);
    int rc = kmu::launch_status("group_norm_bwd sums");
    if (rc) return rc;
    if (act == 1)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<1>, dim3(B * C, S), dim3(256), 0, st, x, gout, gamma, beta, stats, ws, dx, d_gamma_partial,
                           d_beta_partial, C, G, HW);
    else if (act == 2)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<2>, dim3(B * C, S), dim3(256), 0, st, x, gout, gamma, beta, stats, ws, dx, d_gamma_partial,
                           d_beta_partial, C, G, HW);
    else
        hipLaunchKernelGGL(gn_bwd_apply_kernel<0>, dim3(B * C, S), dim3(256), 0, st, x, gout, gamma, beta, stats, ws, dx, d_gamma_partial,
                           d_beta_partial, C, G, HW);
    return kmu::launch_status("group_norm_bwd apply");
}
extern "C" int kmu_group_norm_bwd(const float* x, const float* gout, const float* gamma, const float* stats, float* dx,
                                  float* d_gamma_partial, float* d_beta_partial, float* ws, int B, int C, int G, int HW,
                                  kmu_stream_t stream) {
    return kmu_group_norm_act_bwd(x, gout, gamma, nullptr, stats, dx, d_gamma_partial, d_beta_partial, ws, B, C, G, HW, 0, stream);
}
