// TripleNorm of EnhancedViMBlock (KM_UNetV3_SH.py:266-284) as one operator:
//     y = ( GroupNorm_h(x) + GroupNorm_w(x) + LayerNorm_c(x) ) / 3
// norm_h / norm_w = nn.GroupNorm(1, C): the 'height' branch normalises the H/W-transposed tensor and transposes back, but the
// statistics of ONE group over (C, H, W) are permutation invariant, so both use the same xg = (x - mu_b) r_b and only their affine
// parameters differ; norm_c = nn.LayerNorm(C) on the channels-last view = per-pixel statistics over C (xl = (x - m_p) s_p):
//     y[b,c,p] = ( xg (gh[c] + gw[c]) + (bh[c] + bw[c]) + xl gc[c] + bc[c] ) / 3
// As separate nodes this was 7 launches forward (LayerNorm, two parameter adds, GroupNorm statistics + apply, add, divide) and
// ~10 backward, all on the step's main dependent chain (no side stream is busy while EnhancedViMBlock's tail runs).  Here:
//   forward : per-plane sums of x (gn-style partials)  ->  one per-pixel kernel (thread = V pixels, all C channels in registers)
//   backward: per-plane sums of g and g*xg             ->  one per-pixel kernel: dx (+ an optional addend: the residual branch's
//             gradient, saving autograd's fan-in add), per-block partials of d gc; d(gh+gw), d(bh+bw+bc) leave as per-(b,c) sums
// HBM-bound streaming; C in {16, 32, 64} (the channel counts KM-UNet builds).
#include "common.h"

using kmu::floatx4;

namespace {

typedef float floatx2 __attribute__((ext_vector_type(2)));
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

template <int V>
struct VecOf;
template <>
struct VecOf<1> { typedef float T; };
template <>
struct VecOf<2> { typedef floatx2 T; };
template <>
struct VecOf<4> { typedef floatx4 T; };
template <int V>
__device__ __forceinline__ void ldv(float (&dst)[V], const float* p) {
    typename VecOf<V>::T t = *reinterpret_cast<const typename VecOf<V>::T*>(p);
    if constexpr (V == 1) dst[0] = t;
    else
#pragma unroll
        for (int i = 0; i < V; ++i) dst[i] = t[i];
}
template <int V>
__device__ __forceinline__ void stv(float* p, const float (&src)[V]) {
    typename VecOf<V>::T t;
    if constexpr (V == 1) t = src[0];
    else
#pragma unroll
        for (int i = 0; i < V; ++i) t[i] = src[i];
    *reinterpret_cast<typename VecOf<V>::T*>(p) = t;
}

// part[(plane*S + s)*2 + {0,1}] = sum a, sum a*b' over this block's slice of plane (b,c);  MODE 0: (x, x^2);
// MODE 1: (g, g * xg) with xg = (x - mean_b) rstd_b
template <int MODE>
__global__ __launch_bounds__(256) void tn_plane_sums_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ stats, float* __restrict__ part, int C, int HW) {
    __shared__ float red[4];
    const int plane = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const int chunk = ((HW + S - 1) / S + 3) / 4 * 4, p0 = s * chunk, p1 = min(HW, p0 + chunk);
    const float* xp = x + (size_t)plane * HW;
    const float* gp = MODE ? g + (size_t)plane * HW : nullptr;
    const float mean = MODE ? stats[(plane / C) * 2] : 0.f, rstd = MODE ? stats[(plane / C) * 2 + 1] : 1.f;
    float a = 0.f, q = 0.f;
    if ((HW & 3) == 0) {
        for (int p = p0 + threadIdx.x * 4; p < p1; p += 1024) {
            const floatx4 v = *reinterpret_cast<const floatx4*>(xp + p);
            if (MODE == 0) {
                a += (v[0] + v[1]) + (v[2] + v[3]);
                q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            } else {
                const floatx4 gv = *reinterpret_cast<const floatx4*>(gp + p);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a += gv[k];
                    q += gv[k] * ((v[k] - mean) * rstd);
                }
            }
        }
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            if (MODE == 0) {
                a += xp[p];
                q += xp[p] * xp[p];
            } else {
                a += gp[p];
                q += gp[p] * ((xp[p] - mean) * rstd);
            }
        }
    }
    a = block_sum(a, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        part[((size_t)plane * S + s) * 2] = a;
        part[((size_t)plane * S + s) * 2 + 1] = q;
    }
}

// forward apply: grid (ceil(HW / (256 V)), B); part = MODE-0 sums of sample b: [C][S][2]
template <int C, int V>
__global__ __launch_bounds__(256) void tn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gh,
                                                       const float* __restrict__ bh, const float* __restrict__ gw,
                                                       const float* __restrict__ bw, const float* __restrict__ gc,
                                                       const float* __restrict__ bc, const float* __restrict__ part,
                                                       float* __restrict__ y, float* __restrict__ stats, int HW, int S, float eps_gn,
                                                       float eps_ln) {
    __shared__ float red[4];
    __shared__ float aff[3 * C];      // (gh+gw)/3 * rstd_b | ((bh+bw+bc) - mean_b*...)/3 | gc/3
    const int b = blockIdx.y, tid = threadIdx.x;
    // fold this sample's C*S partial pairs (<= 512): two block reductions
    float a = 0.f, q = 0.f;
    const float* pp = part + (size_t)b * C * S * 2;
    for (int i = tid; i < C * S; i += 256) {
        a += pp[i * 2];
        q += pp[i * 2 + 1];
    }
    a = block_sum(a, red);
    q = block_sum(q, red);
    const double n = (double)C * (double)HW, md = (double)a / n;
    double vd = (double)q / n - md * md;
    if (vd < 0.0) vd = 0.0;
    const float mean = (float)md, rstd = (float)(1.0 / sqrt(vd + (double)eps_gn));
    if (blockIdx.x == 0 && tid == 0) {
        stats[b * 2] = mean;
        stats[b * 2 + 1] = rstd;
    }
    if (tid < C) {
        const float G = (gh[tid] + gw[tid]) * rstd * (1.f / 3.f);
        aff[tid] = G;
        aff[C + tid] = (bh[tid] + bw[tid] + bc[tid]) * (1.f / 3.f) - mean * G;
        aff[2 * C + tid] = gc[tid] * (1.f / 3.f);
    }
    __syncthreads();
    const int l0 = (blockIdx.x * 256 + tid) * V;
    if (l0 >= HW) return;
    const size_t base = (size_t)b * C * HW + l0;
    float v[C][V], mu[V], rs[V];
#pragma unroll
    for (int c = 0; c < C; ++c) ldv<V>(v[c], x + base + (size_t)c * HW);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) m += v[c][i];
        m /= (float)C;
        float vr = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float d = v[c][i] - m;
            vr += d * d;
        }
        mu[i] = m;
        rs[i] = 1.f / sqrtf(vr / (float)C + eps_ln);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float G = aff[c], Bv = aff[C + c], L = aff[2 * C + c];
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = v[c][i] * G + Bv + (v[c][i] - mu[i]) * rs[i] * L;
        stv<V>(y + base + (size_t)c * HW, o);
    }
}

// backward apply: part = MODE-1 sums of sample b ([C][S][2]: sum g, sum g*xg, g = dy as it arrives);
//   dx = addend + (1/3) [ r_b (Gm g - m0 - xg m1) + s_p (gc g - mean_c(gc g) - xl mean_c(gc g xl)) ],  Gm = gh + gw,
//   m0 = sum_c Gm[c] sum_p g / n, m1 = sum_c Gm[c] sum_p g xg / n;  dgc_part[row][c] = (1/3) sum_{block} g xl
//   d(gh+gw)[b][c] = (1/3) sum_p g xg,  d(bh+bw+bc)[b][c] = (1/3) sum_p g   (written by block 0 of the sample)
template <int C, int V>
__global__ __launch_bounds__(256) void tn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ gh, const float* __restrict__ gw,
                                                           const float* __restrict__ gc, const float* __restrict__ stats,
                                                           const float* __restrict__ part, const float* __restrict__ addend,
                                                           float* __restrict__ dx, float* __restrict__ dG_part,
                                                           float* __restrict__ dB_part, float* __restrict__ dgc_part, int HW, int S,
                                                           float eps_ln) {
    __shared__ float red[4];
    __shared__ float prm[2 * C];      // Gm | gc
    __shared__ float wred[C][4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float mean = stats[b * 2], rstd = stats[b * 2 + 1];
    if (tid < C) {
        prm[tid] = gh[tid] + gw[tid];
        prm[C + tid] = gc[tid];
    }
    __syncthreads();
    const float* pp = part + (size_t)b * C * S * 2;
    float a = 0.f, q = 0.f;
    for (int i = tid; i < C * S; i += 256) {
        const float Gm = prm[i / S];
        a += Gm * pp[i * 2];
        q += Gm * pp[i * 2 + 1];
    }
    a = block_sum(a, red);
    q = block_sum(q, red);
    const float n = (float)C * (float)HW;
    const float m0 = a / n, m1 = q / n;
    if (blockIdx.x == 0 && tid < C) {
        float s0 = 0.f, s1 = 0.f;
        for (int i = 0; i < S; ++i) {
            s0 += pp[(tid * S + i) * 2];
            s1 += pp[(tid * S + i) * 2 + 1];
        }
        dB_part[b * C + tid] = s0 * (1.f / 3.f);
        dG_part[b * C + tid] = s1 * (1.f / 3.f);
    }
    const int l0 = (blockIdx.x * 256 + tid) * V;
    const bool ok = l0 < HW;
    const size_t base = (size_t)b * C * HW + (ok ? l0 : 0);
    float gdw[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gdw[c] = 0.f;
    if (ok) {
        float xv[C][V], g[C][V], mu[V], rs[V], s1[V], s2[V];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            ldv<V>(xv[c], x + base + (size_t)c * HW);
            ldv<V>(g[c], dy + base + (size_t)c * HW);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float m = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) m += xv[c][i];
            m /= (float)C;
            float vr = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d = xv[c][i] - m;
                vr += d * d;
            }
            mu[i] = m;
            rs[i] = 1.f / sqrtf(vr / (float)C + eps_ln);
            s1[i] = s2[i] = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float xl = (xv[c][i] - m) * rs[i];
                const float gw_ = g[c][i] * prm[C + c];
                gdw[c] += g[c][i] * xl;
                s1[i] += gw_;
                s2[i] += gw_ * xl;
            }
            s1[i] /= (float)C;
            s2[i] /= (float)C;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float Gm = prm[c], L = prm[C + c];
            float o[V], ad[V];
            if (addend) ldv<V>(ad, addend + base + (size_t)c * HW);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float xg = (xv[c][i] - mean) * rstd, xl = (xv[c][i] - mu[i]) * rs[i];
                const float dgn = rstd * (Gm * g[c][i] - m0 - xg * m1);
                const float dln = rs[i] * (L * g[c][i] - s1[i] - xl * s2[i]);
                o[i] = (dgn + dln) * (1.f / 3.f) + (addend ? ad[i] : 0.f);
            }
            stv<V>(dx + base + (size_t)c * HW, o);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float w = kmu::wave_sum(gdw[c]);
        if (lane == 0) wred[c][wave] = w;
    }
    __syncthreads();
    if (tid < C) {
        const size_t prow = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        dgc_part[prow * C + tid] = ((wred[tid][0] + wred[tid][1]) + (wred[tid][2] + wred[tid][3])) * (1.f / 3.f);
    }
}

inline int tn_v(int C, int HW) { return (C == 16 && HW % 2 == 0) ? 2 : ((C == 32 || C == 64) ? 1 : 0); }

}  // namespace

extern "C" int kmu_triple_norm_supported(int C, int HW) { return tn_v(C, HW) != 0; }
extern "C" int kmu_triple_norm_splits(int HW) { return splits_for(HW); }
// rows of the d gc partials (one per block of the per-pixel kernels)
extern "C" int kmu_triple_norm_partials(int B, int C, int HW) {
    const int V = tn_v(C, HW);
    return V ? B * kmu::cdiv(HW, 256 * V) : 0;
}

extern "C" int kmu_triple_norm_fwd(const float* x, const float* gh, const float* bh, const float* gw, const float* bw, const float* gc,
                                   const float* bc, float* y, float* stats, float* ws, int B, int C, int HW, float eps_gn, float eps_ln,
                                   kmu_stream_t stream) {
    KMU_REQUIRE(x && gh && bh && gw && bw && gc && bc && y && stats && ws, "triple_norm_fwd: null pointer");
    const int V = tn_v(C, HW);
    KMU_REQUIRE(B > 0 && B <= 65535 && HW > 0 && V, "triple_norm_fwd: C=%d (16/32/64), H*W=%d unsupported", C, HW);
    hipStream_t st = (hipStream_t)stream;
    const int S = splits_for(HW);
    hipLaunchKernelGGL(tn_plane_sums_kernel<0>, dim3(B * C, S), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr, ws, C, HW);
    int rc = kmu::launch_status("triple_norm_fwd sums");
    if (rc) return rc;
    const dim3 grid(kmu::cdiv(HW, 256 * V), B);
    if (C == 16)
        hipLaunchKernelGGL((tn_apply_kernel<16, 2>), grid, dim3(256), 0, st, x, gh, bh, gw, bw, gc, bc, ws, y, stats, HW, S, eps_gn, eps_ln);
    else if (C == 32)
        hipLaunchKernelGGL((tn_apply_kernel<32, 1>), grid, dim3(256), 0, st, x, gh, bh, gw, bw, gc, bc, ws, y, stats, HW, S, eps_gn, eps_ln);
    else
        hipLaunchKernelGGL((tn_apply_kernel<64, 1>), grid, dim3(256), 0, st, x, gh, bh, gw, bw, gc, bc, ws, y, stats, HW, S, eps_gn, eps_ln);
    return kmu::launch_status("triple_norm_fwd apply");
}

extern "C" int kmu_triple_norm_bwd(const float* x, const float* dy, const float* gh, const float* gw, const float* gc, const float* stats,
                                   const float* addend, float* dx, float* d_gsum_partial, float* d_bsum_partial, float* d_gc_partial,
                                   float* ws, int B, int C, int HW, float eps_ln, kmu_stream_t stream) {
    KMU_REQUIRE(x && dy && gh && gw && gc && stats && dx && d_gsum_partial && d_bsum_partial && d_gc_partial && ws,
                "triple_norm_bwd: null pointer");
    const int V = tn_v(C, HW);
    KMU_REQUIRE(B > 0 && B <= 65535 && HW > 0 && V, "triple_norm_bwd: C=%d (16/32/64), H*W=%d unsupported", C, HW);
    hipStream_t st = (hipStream_t)stream;
    const int S = splits_for(HW);
    hipLaunchKernelGGL(tn_plane_sums_kernel<1>, dim3(B * C, S), dim3(256), 0, st, x, dy, stats, ws, C, HW);
    int rc = kmu::launch_status("triple_norm_bwd sums");
    if (rc) return rc;
    const dim3 grid(kmu::cdiv(HW, 256 * V), B);
    if (C == 16)
        hipLaunchKernelGGL((tn_bwd_apply_kernel<16, 2>), grid, dim3(256), 0, st, x, dy, gh, gw, gc, stats, ws, addend, dx, d_gsum_partial,
                           d_bsum_partial, d_gc_partial, HW, S, eps_ln);
    else if (C == 32)
        hipLaunchKernelGGL((tn_bwd_apply_kernel<32, 1>), grid, dim3(256), 0, st, x, dy, gh, gw, gc, stats, ws, addend, dx, d_gsum_partial,
                           d_bsum_partial, d_gc_partial, HW, S, eps_ln);
    else
        hipLaunchKernelGGL((tn_bwd_apply_kernel<64, 1>), grid, dim3(256), 0, st, x, dy, gh, gw, gc, stats, ws, addend, dx, d_gsum_partial,
                           d_bsum_partial, d_gc_partial, HW, S, eps_ln);
    return kmu::launch_status("triple_norm_bwd apply");
}
