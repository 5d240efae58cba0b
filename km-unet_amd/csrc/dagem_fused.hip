// DAGEM (DAGEM_md.py:56-111) without its deformable convolution, as ONE launch per BatchNorm boundary.
//
// The bridge works on [8, 64, 16, 16] tensors (0.5 MB): round 3 ran it as ~90 launches of 3-10 us (Linear -> BatchNorm1d statistics ->
// apply + ReLU -> cat -> ...), all of them latency.  A BatchNorm in training mode needs the complete batch before it can normalise, so a
// launch boundary per BatchNorm is the floor; everything between two boundaries is fused here:
//
//   forward   F0  edges e = x . roll(x) (:56-62), a_pre = Linear(4,1)(e) (:65), u_pre = Linear(2C, C/2)([x | e]) (:74-81)
//                                                                                            + partial sums of BN_a, BN_e
//             F1  agg = ReLU(BN_a(a_pre)); v_pre = Linear(2C, C/2)([x | agg]) (:68-72); u = ReLU(BN_e(u_pre));
//                 r_pre = Linear(4,1)(u) (:82)                                               + partial sums of BN_v, BN_r
//             F2  f = ReLU(BN_v(v_pre)) . ReLU(BN_r(r_pre)) (:85); z = Conv1x1([deform(x) + x | f]) (:97-103)   + partial sums of BN_f
//             F3  out = ReLU(BN_f(z)) (:104)
//   backward  B0  BN_f sums;  B1  dz, d[deform + x], df -> BN_v / BN_r sums, d W_f;  B2  dv_pre, dr_pre -> d W_v, dx, dagg -> BN_a sums,
//             BN_e sums, d w_r;  B3  da_pre, du_pre -> d w_a, d W_e, de, dx;  B4  dx = edge adjoint (gather) + the other dx pieces.
//
// Workgroup = TP = 16 consecutive pixels of one sample (all channels), 256 threads (128 workgroups at [8,64,16,16]; 32-pixel tiles: 64
// workgroups, every kernel ~40 % slower).  Every kernel leaves its BatchNorm sums as per-workgroup
// partial rows and every workgroup of the NEXT kernel folds them in a fixed order in double: deterministic, no atomics, no separate
// statistics launch.  The BatchNorm1d layers act on rows whose layout differs from NCHW only by a permutation, so they are per-channel
// (BN_v, BN_e) or single-feature (BN_a, BN_r) statistics over the same elements.  ReLU branches are decided by ONE expression,
// relu_bn() below, in the forward and in every backward recomputation.  fp32 throughout (VALU: the contractions are 64..128 deep).
#include "common.h"

using kmu::floatx4;

namespace {

#ifndef KMU_DAGEM_TP
#define KMU_DAGEM_TP 16
#endif
constexpr int TP = KMU_DAGEM_TP;  // pixels per workgroup (16 or 32)
constexpr int TPS = TP == 32 ? 5 : 4, NGR = 256 / TP;      // thread = (pixel p = tid % TP, group g = tid / TP)
static_assert(TP == 16 || TP == 32, "tile of 16 or 32 pixels");
enum { BN_A = 0, BN_V = 1, BN_E = 2, BN_R = 3, BN_F = 4 };

__device__ __forceinline__ float relu_bn(float pre, float scale, float shift) { return fmaxf(fmaf(pre, scale, shift), 0.f); }

// per-workgroup partial layout (floats): [A 2 | E 2*C2 | V 2*C2 | R 2 | F 2*C]
__host__ __device__ inline int part_stride(int C) { return 4 + 4 * C; }
__host__ __device__ inline int part_off(int which, int C) {
    const int C2 = C / 2;
    return which == BN_A ? 0 : (which == BN_E ? 2 : (which == BN_V ? 2 + 2 * C2 : (which == BN_R ? 2 + 4 * C2 : 4 + 4 * C2)));
}

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    int2 p = __builtin_bit_cast(int2, v);
    p.x = __shfl_xor(p.x, m, 64);
    p.y = __shfl_xor(p.y, m, 64);
    return __builtin_bit_cast(double, p);
}

// Sum the NWG partial pairs of NF features: 8 threads per feature (32 features per round), each a strided slice in double, joined by
// three xor steps; NF == 1: all 256 threads + an LDS tree.  Results (s0, s1) per feature into LDS tmp[NF][2] as doubles.
template <int NF>
__device__ __forceinline__ void fold_pairs(const float* __restrict__ part, int stride, int NWG, double* tmp, double* red) {
    const int tid = threadIdx.x;
    if (NF == 1) {
        double a = 0.0, q = 0.0;
        for (int i = tid; i < NWG; i += 256) {
            a += (double)part[(size_t)i * stride];
            q += (double)part[(size_t)i * stride + 1];
        }
        red[tid] = a;
        red[256 + tid] = q;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                red[tid] += red[tid + s];
                red[256 + tid] += red[256 + tid + s];
            }
            __syncthreads();
        }
        if (tid == 0) {
            tmp[0] = red[0];
            tmp[1] = red[256];
        }
    } else {
        const int j = tid & 7;
        for (int f0 = 0; f0 < NF; f0 += 32) {
            const int f = f0 + (tid >> 3);
            double a = 0.0, q = 0.0;
            if (f < NF) {
                int i = j;
                for (; i + 24 < NWG; i += 32) {          // four independent loads in flight, summed in index order
                    const float2 v0 = *reinterpret_cast<const float2*>(part + (size_t)i * stride + 2 * f);
                    const float2 v1 = *reinterpret_cast<const float2*>(part + (size_t)(i + 8) * stride + 2 * f);
                    const float2 v2 = *reinterpret_cast<const float2*>(part + (size_t)(i + 16) * stride + 2 * f);
                    const float2 v3 = *reinterpret_cast<const float2*>(part + (size_t)(i + 24) * stride + 2 * f);
                    a += (double)v0.x, q += (double)v0.y;
                    a += (double)v1.x, q += (double)v1.y;
                    a += (double)v2.x, q += (double)v2.y;
                    a += (double)v3.x, q += (double)v3.y;
                }
                for (; i < NWG; i += 8) {
                    a += (double)part[(size_t)i * stride + 2 * f];
                    q += (double)part[(size_t)i * stride + 2 * f + 1];
                }
            }
#pragma unroll
            for (int m = 4; m > 0; m >>= 1) {
                a += shfl_xor_d(a, m);
                q += shfl_xor_d(q, m);
            }
            if (j == 0 && f < NF) {
                tmp[2 * f] = a;
                tmp[2 * f + 1] = q;
            }
        }
    }
    __syncthreads();
}

// forward BatchNorm coefficients of NF features from (sum, sum of squares) partials: sc = gamma rstd, sh = beta - mean sc into LDS,
// (mean, rstd) to `stat` and the running statistics / batch counter by the `writer` workgroup.
template <int NF>
__device__ __forceinline__ void bn_coeffs(const kmu_dagem_args& a, int which, const float* __restrict__ part, int NWG, double n, bool writer,
                                          float* sc, float* sh, double* tmp, double* red) {
    const int tid = threadIdx.x, C = a.C;
    if (a.training) fold_pairs<NF>(part + part_off(which, C), part_stride(C), NWG, tmp, red);
    for (int f = tid; f < NF; f += 256) {
        float mean, rstd;
        if (a.training) {
            const double m = tmp[2 * f] / n;
            double var = tmp[2 * f + 1] / n - m * m;
            if (var < 0.0) var = 0.0;
            mean = (float)m;
            rstd = (float)(1.0 / sqrt(var + (double)a.eps[which]));
            if (writer) {
                const float mom = a.momentum[which];
                a.running_mean[which][f] = (1.f - mom) * a.running_mean[which][f] + mom * mean;
                const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                a.running_var[which][f] = (1.f - mom) * a.running_var[which][f] + mom * (float)unb;
                if (f == 0 && a.num_batches_tracked[which]) *a.num_batches_tracked[which] += 1;
            }
        } else {
            mean = a.running_mean[which][f];
            rstd = 1.f / sqrtf(a.running_var[which][f] + a.eps[which]);
        }
        const float s = a.gamma[which][f] * rstd;
        sc[f] = s;
        sh[f] = a.beta[which][f] - mean * s;
        if (writer) {
            a.bnstat[(which * C + f) * 2] = mean;
            a.bnstat[(which * C + f) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
}

// backward: m1 = mean(g'), m2 = mean(g' xhat) of NF features from the (sum g', sum g' xhat) partials (zeros in eval mode), the
// parameter gradients d beta = sum g', d gamma = sum g' xhat by the writer; sc / sh / mean / rstd reloaded from bnstat.
template <int NF>
__device__ __forceinline__ void bn_bwd_coeffs(const kmu_dagem_args& a, int which, const float* __restrict__ part, int NWG, double n, bool writer,
                                              float* m1, float* m2, float* mean, float* rstd, float* sc, float* sh, double* tmp, double* red) {
    const int tid = threadIdx.x, C = a.C;
    fold_pairs<NF>(part + part_off(which, C), part_stride(C), NWG, tmp, red);
    for (int f = tid; f < NF; f += 256) {
        m1[f] = a.training ? (float)(tmp[2 * f] / n) : 0.f;
        m2[f] = a.training ? (float)(tmp[2 * f + 1] / n) : 0.f;
        const float mu = a.bnstat[(which * C + f) * 2], rs = a.bnstat[(which * C + f) * 2 + 1];
        mean[f] = mu;
        rstd[f] = rs;
        const float s = a.gamma[which][f] * rs;
        sc[f] = s;
        sh[f] = a.beta[which][f] - mu * s;
        if (writer) {
            a.d_beta[which][f] = (float)tmp[2 * f];
            a.d_gamma[which][f] = (float)tmp[2 * f + 1];
        }
    }
    __syncthreads();
}

// sum of v over the TP pixel lanes of a thread group (lanes p = tid % TP); every lane gets the total
__device__ __forceinline__ float sum32(float v) {
#pragma unroll
    for (int m = TP / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// block total of two values (256 threads); result valid in thread 0
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red8) {
    a = kmu::wave_sum(a);
    b = kmu::wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red8[threadIdx.x >> 6] = a;
        red8[4 + (threadIdx.x >> 6)] = b;
    }
    __syncthreads();
    a = (red8[0] + red8[1]) + (red8[2] + red8[3]);
    b = (red8[4] + red8[5]) + (red8[6] + red8[7]);
}

struct Tile {
    int b, gp0, np;      // sample, first pixel, valid pixels (<= 32)
    size_t wg;
};
__device__ __forceinline__ Tile tile_of(const kmu_dagem_args& a) {
    const int P = a.H * a.W, NT = (P + TP - 1) / TP;
    Tile t;
    t.b = blockIdx.x / NT;
    t.gp0 = (blockIdx.x % NT) * TP;
    t.np = min(TP, P - t.gp0);
    t.wg = blockIdx.x;
    return t;
}
// The loaders below issue LB global loads before the first LDS store: a plain load -> store loop waits out one memory round trip per
// iteration (the first version of these kernels spent 15 of their 20 us there).
constexpr int LB = 8;
// dst[c][32] <- src[b][c][gp0 + p] (zeros past the sample's last pixel)
__device__ __forceinline__ void load_tile(float* dst, const float* __restrict__ src, int nch, int P, const Tile& t) {
    const float* s = src + (size_t)t.b * nch * P + t.gp0;
    const int n = nch * TP;
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * LB) {
        float v[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int i = i0 + u * 256, c = i >> TPS, p = i & (TP - 1);
            v[u] = (i < n && p < t.np) ? s[(size_t)c * P + p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < LB; ++u)
            if (i0 + u * 256 < n) dst[i0 + u * 256] = v[u];
    }
}
// the pixel's own value and its four cyclic neighbours (DAGEM_md.py:57-60: rows -1 / +1, columns -1 / +1): xs[k5][c][TP]
__device__ __forceinline__ void load_tile5(float* xs, const float* __restrict__ x, int C, int H, int W, const Tile& t) {
    const int P = H * W, p = threadIdx.x & (TP - 1);
    const float* s = x + (size_t)t.b * C * P;
    int q[5];
    {
        const int g = min(t.gp0 + p, P - 1), h = g / W, w = g - h * W;
        q[0] = g;
        q[1] = (h == 0 ? H - 1 : h - 1) * W + w;
        q[2] = (h == H - 1 ? 0 : h + 1) * W + w;
        q[3] = h * W + (w == 0 ? W - 1 : w - 1);
        q[4] = h * W + (w == W - 1 ? 0 : w + 1);
    }
    const bool ok = p < t.np;
    // thread = (pixel p, channel phase tid >> TPS): channels c = (tid >> TPS) + 8 j, all five positions of LB / ... channels in flight
    for (int c0 = threadIdx.x >> TPS; c0 < C; c0 += NGR * 2) {
        float v[2][5];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int c = c0 + NGR * u;
                v[u][k] = (ok && c < C) ? s[(size_t)c * P + q[k]] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int c = c0 + NGR * u;
                if (c < C) xs[(k * C + c) * TP + p] = v[u][k];
            }
    }
}
// dst[c][p] = op(c, p, {src_0[b][c][pixel], ..}) over an nch-channel tile, the loads of four elements in flight before the first use
template <int NS, class OP>
__device__ __forceinline__ void tile_apply(float* dst, int nch, int P, const Tile& t, const float* const (&src)[NS], OP op) {
    constexpr int UB = 4;
    const int n = nch * TP;
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * UB) {
        float v[UB][NS];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + u * 256, c = i >> TPS, p = i & (TP - 1);
            const bool ok = i < n && p < t.np;
#pragma unroll
            for (int k = 0; k < NS; ++k) v[u][k] = ok ? src[k][((size_t)t.b * nch + c) * P + t.gp0 + p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + u * 256;
            if (i < n) dst[i] = (i & (TP - 1)) < t.np ? op(i >> TPS, i & (TP - 1), v[u]) : 0.f;
        }
    }
}
__device__ __forceinline__ void load_vec(float* dst, const float* __restrict__ src, int n) {
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * LB) {
        float v[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) v[u] = i0 + u * 256 < n ? src[i0 + u * 256] : 0.f;
#pragma unroll
        for (int u = 0; u < LB; ++u)
            if (i0 + u * 256 < n) dst[i0 + u * 256] = v[u];
    }
}

// ===================================================================================================================== forward
// F0: a_pre [B,C,P], u_pre [B,C2,P,4]; partial sums of BN_a (1 feature) and BN_e (C2 features)
template <int C>
__global__ __launch_bounds__(256) void dagem_f0(kmu_dagem_args a) {
    constexpr int C2 = C / 2, OPT = C2 / NGR;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                       // [5][C][32]
    float* we = xs + 5 * C * TP;          // [C2][2C]
    float* red8 = we + C2 * 2 * C;        // [8]
    const int tid = threadIdx.x, P = a.H * a.W;
    const Tile t = tile_of(a);
    load_tile5(xs, a.x, C, a.H, a.W, t);
    load_vec(we, a.we, C2 * 2 * C);
    __syncthreads();
    float* part = a.part + t.wg * part_stride(C);
    {   // edge aggregation pre-activation
        const float w0 = a.wa[0], w1 = a.wa[1], w2 = a.wa[2], w3 = a.wa[3], b0 = a.ba[0];
        float s = 0.f, q = 0.f;
        for (int i = tid; i < C * TP; i += 256) {
            const int c = i >> TPS, p = i & (TP - 1);
            if (p < t.np) {
                const float x0 = xs[i];
                const float v = b0 + w0 * (x0 * xs[C * TP + i]) + w1 * (x0 * xs[2 * C * TP + i]) + w2 * (x0 * xs[3 * C * TP + i]) +
                                w3 * (x0 * xs[4 * C * TP + i]);
                a.a_pre[((size_t)t.b * C + c) * P + t.gp0 + p] = v;
                s += v;
                q += v * v;
            }
        }
        block_sum2(s, q, red8);
        if (tid == 0) {
            part[0] = s;
            part[1] = q;
        }
    }
    {   // edge update pre-activation: thread = (pixel p, output group g)
        const int p = tid & (TP - 1), g = tid >> TPS;
        float base[OPT], u[OPT][4];
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            base[j] = a.be[g * OPT + j];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[j][k] = 0.f;
        }
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float x0 = xs[c * TP + p];
            float e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = x0 * xs[((k + 1) * C + c) * TP + p];
#pragma unroll
            for (int j = 0; j < OPT; ++j) {
                const float* wr = we + (g * OPT + j) * 2 * C;
                base[j] += wr[c] * x0;
                const float w2 = wr[C + c];
#pragma unroll
                for (int k = 0; k < 4; ++k) u[j][k] += w2 * e[k];
            }
        }
        const bool ok = p < t.np;
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int o = g * OPT + j;
            floatx4 v = {base[j] + u[j][0], base[j] + u[j][1], base[j] + u[j][2], base[j] + u[j][3]};
            if (!ok) v = floatx4{0.f, 0.f, 0.f, 0.f};
            if (ok) *reinterpret_cast<floatx4*>(a.u_pre + (((size_t)t.b * C2 + o) * P + t.gp0 + p) * 4) = v;
            const float s = sum32((v[0] + v[1]) + (v[2] + v[3]));
            const float q = sum32((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
            if (p == 0) {
                part[part_off(BN_E, C) + 2 * o] = s;
                part[part_off(BN_E, C) + 2 * o + 1] = q;
            }
        }
    }
}

// F1: v_pre, r_pre [B,C2,P]; partial sums of BN_v (C2) and BN_r (1)
template <int C>
__global__ __launch_bounds__(256) void dagem_f1(kmu_dagem_args a, int NWG) {
    constexpr int C2 = C / 2, OPT = C2 / NGR;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    double* tmp = reinterpret_cast<double*>(sm);           // [2 * C2]
    double* red = tmp + 2 * C;                              // [512]
    float* xs = reinterpret_cast<float*>(red + 512);        // [C][32]
    float* ag = xs + C * TP;                                // [C][32]
    float* wv = ag + C * TP;                                // [C2][2C]
    float* scE = wv + C2 * 2 * C;                           // [C2] x 2
    float* shE = scE + C2;
    float* scA = shE + C2;                                  // [1] x 2
    float* red8 = scA + 2;
    const int tid = threadIdx.x, P = a.H * a.W;
    const Tile t = tile_of(a);
    const bool writer = blockIdx.x == 0;
    bn_coeffs<1>(a, BN_A, a.part, NWG, (double)a.B * C * P, writer, scA, scA + 1, tmp, red);
    bn_coeffs<C2>(a, BN_E, a.part, NWG, (double)a.B * P * 4, writer, scE, shE, tmp, red);
    load_tile(xs, a.x, C, P, t);
    load_vec(wv, a.wv, C2 * 2 * C);
    {
        const float s = scA[0], h = scA[1];
        const float* const src[1] = {a.a_pre};
        tile_apply<1>(ag, C, P, t, src, [&](int c, int p, const float (&v)[1]) {
            const float r = relu_bn(v[0], s, h);
            if (a.agg_out) a.agg_out[((size_t)t.b * C + c) * P + t.gp0 + p] = r;
            return r;
        });
    }
    __syncthreads();
    float* part = a.part + t.wg * part_stride(C);
    const int p = tid & (TP - 1), g = tid >> TPS;
    const bool ok = p < t.np;
    {   // vertex update pre-activation
        float acc[OPT];
#pragma unroll
        for (int j = 0; j < OPT; ++j) acc[j] = a.bv[g * OPT + j];
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float x0 = xs[c * TP + p], a0 = ag[c * TP + p];
#pragma unroll
            for (int j = 0; j < OPT; ++j) {
                const float* wr = wv + (g * OPT + j) * 2 * C;
                acc[j] += wr[c] * x0 + wr[C + c] * a0;
            }
        }
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int o = g * OPT + j;
            const float v = ok ? acc[j] : 0.f;
            if (ok) a.v_pre[((size_t)t.b * C2 + o) * P + t.gp0 + p] = v;
            const float s = sum32(v), q = sum32(v * v);
            if (p == 0) {
                part[part_off(BN_V, C) + 2 * o] = s;
                part[part_off(BN_V, C) + 2 * o + 1] = q;
            }
        }
    }
    {   // edge reduce pre-activation
        const float w0 = a.wr[0], w1 = a.wr[1], w2 = a.wr[2], w3 = a.wr[3], b0 = a.br[0];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int o = g * OPT + j;
            if (ok) {
                const size_t idx = ((size_t)t.b * C2 + o) * P + t.gp0 + p;
                const floatx4 up = *reinterpret_cast<const floatx4*>(a.u_pre + idx * 4);
                const floatx4 u = {relu_bn(up[0], scE[o], shE[o]), relu_bn(up[1], scE[o], shE[o]), relu_bn(up[2], scE[o], shE[o]),
                                   relu_bn(up[3], scE[o], shE[o])};
                if (a.u_out) *reinterpret_cast<floatx4*>(a.u_out + idx * 4) = u;
                const float v = b0 + w0 * u[0] + w1 * u[1] + w2 * u[2] + w3 * u[3];
                a.r_pre[idx] = v;
                s += v;
                q += v * v;
            }
        }
        block_sum2(s, q, red8);
        if (tid == 0) {
            part[part_off(BN_R, C)] = s;
            part[part_off(BN_R, C) + 1] = q;
        }
    }
}

// F2: z [B,C,P] = W_f . [dconv + x | ReLU(BN_v(v_pre)) ReLU(BN_r(r_pre))]; partial sums of BN_f (C)
template <int C>
__global__ __launch_bounds__(256) void dagem_f2(kmu_dagem_args a, int NWG) {
    constexpr int C2 = C / 2, KF = C + C2, OPT = C / NGR;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    double* tmp = reinterpret_cast<double*>(sm);
    double* red = tmp + 2 * C;
    float* in = reinterpret_cast<float*>(red + 512);        // [C + C2][32]
    float* wf = in + KF * TP;                               // [C][C + C2]
    float* scV = wf + C * KF;
    float* shV = scV + C2;
    float* scR = shV + C2;
    const int tid = threadIdx.x, P = a.H * a.W;
    const Tile t = tile_of(a);
    const bool writer = blockIdx.x == 0;
    bn_coeffs<C2>(a, BN_V, a.part, NWG, (double)a.B * P, writer, scV, shV, tmp, red);
    bn_coeffs<1>(a, BN_R, a.part, NWG, (double)a.B * C2 * P, writer, scR, scR + 1, tmp, red);
    load_vec(wf, a.wf, C * KF);
    {
        const float* const s2[2] = {a.dconv, a.x};
        tile_apply<2>(in, C, P, t, s2, [](int, int, const float (&v)[2]) { return v[0] + v[1]; });
        const float sr = scR[0], hr = scR[1];
        const float* const s3[2] = {a.v_pre, a.r_pre};
        tile_apply<2>(in + C * TP, C2, P, t, s3, [&](int o, int p, const float (&v)[2]) {
            const float vert = relu_bn(v[0], scV[o], shV[o]), ue = relu_bn(v[1], sr, hr);
            const size_t idx = ((size_t)t.b * C2 + o) * P + t.gp0 + p;
            if (a.vert_out) a.vert_out[idx] = vert;
            if (a.ue_out) a.ue_out[idx] = ue;
            return vert * ue;
        });
    }
    __syncthreads();
    float* part = a.part + t.wg * part_stride(C);
    const int p = tid & (TP - 1), g = tid >> TPS;
    const bool ok = p < t.np;
    float acc[OPT];
#pragma unroll
    for (int j = 0; j < OPT; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int k = 0; k < KF; ++k) {
        const float v = in[k * TP + p];
#pragma unroll
        for (int j = 0; j < OPT; ++j) acc[j] += wf[(g * OPT + j) * KF + k] * v;
    }
#pragma unroll
    for (int j = 0; j < OPT; ++j) {
        const int o = g * OPT + j;
        const float v = ok ? acc[j] : 0.f;
        if (ok) a.z[((size_t)t.b * C + o) * P + t.gp0 + p] = v;
        const float s = sum32(v), q = sum32(v * v);
        if (p == 0) {
            part[part_off(BN_F, C) + 2 * o] = s;
            part[part_off(BN_F, C) + 2 * o + 1] = q;
        }
    }
}

// F3: out = ReLU(BN_f(z))
template <int C>
__global__ __launch_bounds__(256) void dagem_f3(kmu_dagem_args a, int NWG) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    double* tmp = reinterpret_cast<double*>(sm);
    double* red = tmp + 2 * C;
    float* sc = reinterpret_cast<float*>(red + 512);
    float* sh = sc + C;
    const int P = a.H * a.W;
    const Tile t = tile_of(a);
    bn_coeffs<C>(a, BN_F, a.part, NWG, (double)a.B * P, blockIdx.x == 0, sc, sh, tmp, red);
    for (int i = threadIdx.x; i < C * TP; i += 256) {
        const int c = i >> TPS, p = i & (TP - 1);
        if (p < t.np) {
            const size_t idx = ((size_t)t.b * C + c) * P + t.gp0 + p;
            a.out[idx] = relu_bn(a.z[idx], sc[c], sh[c]);
        }
    }
}

// ===================================================================================================================== backward
// B0: partial (sum g', sum g' zhat) of BN_f, g' = g . [out > 0]
template <int C>
__global__ __launch_bounds__(256) void dagem_b0(kmu_dagem_args a) {
    const int P = a.H * a.W, p = threadIdx.x & (TP - 1), g = threadIdx.x >> TPS;
    const Tile t = tile_of(a);
    float* part = a.part_bwd + t.wg * part_stride(C);
    for (int o = g; o < C; o += NGR) {
        const float mu = a.bnstat[(BN_F * C + o) * 2], rs = a.bnstat[(BN_F * C + o) * 2 + 1];
        const float sc = a.gamma[BN_F][o] * rs, sh = a.beta[BN_F][o] - mu * sc;
        float s1 = 0.f, s2 = 0.f;
        if (p < t.np) {
            const size_t idx = ((size_t)t.b * C + o) * P + t.gp0 + p;
            const float zv = a.z[idx];
            const float gp = relu_bn(zv, sc, sh) > 0.f ? a.g_out[idx] : 0.f;
            s1 = gp;
            s2 = gp * ((zv - mu) * rs);
        }
        s1 = sum32(s1);
        s2 = sum32(s2);
        if (p == 0) {
            part[part_off(BN_F, C) + 2 * o] = s1;
            part[part_off(BN_F, C) + 2 * o + 1] = s2;
        }
    }
}

// B1: dz -> g_dd [B,C,P] (gradient of deform(x) + x), df -> gv, gr [B,C2,P] (through the product and both ReLUs) + BN_v / BN_r sums,
// d W_f partial rows
template <int C>
__global__ __launch_bounds__(256) void dagem_b1(kmu_dagem_args a, int NWG) {
    constexpr int C2 = C / 2, KF = C + C2;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    double* tmp = reinterpret_cast<double*>(sm);
    double* red = tmp + 2 * C;
    float* dz = reinterpret_cast<float*>(red + 512);        // [C][32]
    float* in = dz + C * TP;                                // [C + C2][32]: dconv + x | f
    float* wf = in + KF * TP;                               // [C][KF]
    float* m1 = wf + C * KF;                                // 6 x [C]
    float* m2 = m1 + C;
    float* mean = m2 + C;
    float* rstd = mean + C;
    float* sc = rstd + C;
    float* sh = sc + C;
    float* red8 = sh + C;
    const int tid = threadIdx.x, P = a.H * a.W;
    const Tile t = tile_of(a);
    const bool writer = blockIdx.x == 0;
    bn_bwd_coeffs<C>(a, BN_F, a.part_bwd, NWG, (double)a.B * P, writer, m1, m2, mean, rstd, sc, sh, tmp, red);
    load_vec(wf, a.wf, C * KF);
    {
        const float* const s2[2] = {a.dconv, a.x};
        tile_apply<2>(in, C, P, t, s2, [](int, int, const float (&v)[2]) { return v[0] + v[1]; });
        const float* const s3[2] = {a.z, a.g_out};
        tile_apply<2>(dz, C, P, t, s3, [&](int c, int, const float (&v)[2]) {
            const float gp = relu_bn(v[0], sc[c], sh[c]) > 0.f ? v[1] : 0.f;
            return sc[c] * (gp - m1[c] - ((v[0] - mean[c]) * rstd[c]) * m2[c]);
        });
    }
    // vert / ue of this tile (registers of the threads that own (o, p) below) and f into `in`
    const int p = tid & (TP - 1), g = tid >> TPS;
    const bool ok = p < t.np;
    constexpr int OPT = C2 / NGR;
    float vert[OPT], ue[OPT], vhat[OPT], rhat[OPT];
    const float muR = a.bnstat[(BN_R * C) * 2], rsR = a.bnstat[(BN_R * C) * 2 + 1];
    const float scR = a.gamma[BN_R][0] * rsR, shR = a.beta[BN_R][0] - muR * scR;
#pragma unroll
    for (int j = 0; j < OPT; ++j) {
        const int o = g * OPT + j;
        const float muV = a.bnstat[(BN_V * C + o) * 2], rsV = a.bnstat[(BN_V * C + o) * 2 + 1];
        const float scV = a.gamma[BN_V][o] * rsV, shV = a.beta[BN_V][o] - muV * scV;
        float vp = 0.f, rp = 0.f;
        if (ok) {
            const size_t idx = ((size_t)t.b * C2 + o) * P + t.gp0 + p;
            vp = a.v_pre[idx];
            rp = a.r_pre[idx];
        }
        vert[j] = ok ? relu_bn(vp, scV, shV) : 0.f;
        ue[j] = ok ? relu_bn(rp, scR, shR) : 0.f;
        vhat[j] = (vp - muV) * rsV;
        rhat[j] = (rp - muR) * rsR;
        in[(C + o) * TP + p] = vert[j] * ue[j];
    }
    __syncthreads();
    // d(dconv + x)[c][p] = sum_o W_f[o][c] dz[o][p]: thread = (p, 8 channel groups)
    {
        constexpr int CPT = C / NGR;
        float acc[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
#pragma unroll 4
        for (int o = 0; o < C; ++o) {
            const float d = dz[o * TP + p];
#pragma unroll
            for (int j = 0; j < CPT; ++j) acc[j] += wf[o * KF + g * CPT + j] * d;
        }
        if (ok)
#pragma unroll
            for (int j = 0; j < CPT; ++j) a.g_dd[((size_t)t.b * C + g * CPT + j) * P + t.gp0 + p] = acc[j];
    }
    // df[j][p] = sum_o W_f[o][C + j] dz[o][p] -> gv', gr' and the BN_v / BN_r sums
    float* part = a.part_bwd + t.wg * part_stride(C);
    {
        float df[OPT];
#pragma unroll
        for (int j = 0; j < OPT; ++j) df[j] = 0.f;
#pragma unroll 4
        for (int o = 0; o < C; ++o) {
            const float d = dz[o * TP + p];
#pragma unroll
            for (int j = 0; j < OPT; ++j) df[j] += wf[o * KF + C + g * OPT + j] * d;
        }
        float r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int o = g * OPT + j;
            const float gv = (ok && vert[j] > 0.f) ? df[j] * ue[j] : 0.f;
            const float gr = (ok && ue[j] > 0.f) ? df[j] * vert[j] : 0.f;
            if (ok) {
                const size_t idx = ((size_t)t.b * C2 + o) * P + t.gp0 + p;
                a.gv[idx] = gv;
                a.gr[idx] = gr;
            }
            const float s1 = sum32(gv), s2 = sum32(gv * vhat[j]);
            if (p == 0) {
                part[part_off(BN_V, C) + 2 * o] = s1;
                part[part_off(BN_V, C) + 2 * o + 1] = s2;
            }
            r1 += gr;
            r2 += gr * rhat[j];
        }
        block_sum2(r1, r2, red8);
        if (tid == 0) {
            part[part_off(BN_R, C)] = r1;
            part[part_off(BN_R, C) + 1] = r2;
        }
    }
    // d W_f[o][i] partial = sum_p dz[o][p] in[i][p]: thread = (o = tid % C, i-group)
    {
        constexpr int NIG = 256 / C, IPT = KF / NIG;
        static_assert(256 % C == 0 && KF % NIG == 0, "d W_f split over the threads");
        const int o = tid % C, ig = tid / C;
        float acc[IPT];
#pragma unroll
        for (int i = 0; i < IPT; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int q = 0; q < TP; ++q) {
            const float d = dz[o * TP + q];
#pragma unroll
            for (int i = 0; i < IPT; ++i) acc[i] += d * in[(ig * IPT + i) * TP + q];
        }
        float* dst = a.p_wf + (t.wg * C + o) * KF + ig * IPT;
#pragma unroll
        for (int i = 0; i < IPT; ++i) dst[i] = acc[i];
    }
}

// B2: dv_pre -> d W_v / d b_v partial rows, dx piece, dagg -> ga [B,C,P] + BN_a sums; dr_pre [B,C2,P] -> d w_r / d b_r partials, BN_e sums
template <int C>
__global__ __launch_bounds__(256) void dagem_b2(kmu_dagem_args a, int NWG) {
    constexpr int C2 = C / 2, OPT = C2 / NGR;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    double* tmp = reinterpret_cast<double*>(sm);
    double* red = tmp + 2 * C;
    float* in = reinterpret_cast<float*>(red + 512);        // [2C][32]: x | agg
    float* dv = in + 2 * C * TP;                            // [C2][32]
    float* wv = dv + C2 * TP;                               // [C2][2C]
    float* cf = wv + C2 * 2 * C;                            // 6 x [C2] (BN_v), then 6 (BN_r)
    float* red8 = cf + 6 * C2 + 6;
    float *m1 = cf, *m2 = cf + C2, *mean = cf + 2 * C2, *rstd = cf + 3 * C2, *sc = cf + 4 * C2, *sh = cf + 5 * C2, *cr = cf + 6 * C2;
    const int tid = threadIdx.x, P = a.H * a.W;
    const Tile t = tile_of(a);
    const bool writer = blockIdx.x == 0;
    bn_bwd_coeffs<C2>(a, BN_V, a.part_bwd, NWG, (double)a.B * P, writer, m1, m2, mean, rstd, sc, sh, tmp, red);
    bn_bwd_coeffs<1>(a, BN_R, a.part_bwd, NWG, (double)a.B * C2 * P, writer, cr, cr + 1, cr + 2, cr + 3, cr + 4, cr + 5, tmp, red);
    load_tile(in, a.x, C, P, t);
    load_vec(wv, a.wv, C2 * 2 * C);
    const float muA = a.bnstat[(BN_A * C) * 2], rsA = a.bnstat[(BN_A * C) * 2 + 1];
    const float scA = a.gamma[BN_A][0] * rsA, shA = a.beta[BN_A][0] - muA * scA;
    {
        const float* const s1[1] = {a.a_pre};
        tile_apply<1>(in + C * TP, C, P, t, s1, [&](int, int, const float (&v)[1]) { return relu_bn(v[0], scA, shA); });
        const float* const s2[2] = {a.gv, a.v_pre};
        tile_apply<2>(dv, C2, P, t, s2, [&](int o, int, const float (&v)[2]) {
            return sc[o] * (v[0] - m1[o] - ((v[1] - mean[o]) * rstd[o]) * m2[o]);
        });
    }
    __syncthreads();
    float* part = a.part_bwd + t.wg * part_stride(C);
    const int p = tid & (TP - 1), g = tid >> TPS;
    const bool ok = p < t.np;
    {   // dx piece and dagg: [c][p] = sum_o W_v[o][c (+C)] dv[o][p]; thread = (p, 8 channel groups)
        constexpr int CPT = C / NGR;
        float ax[CPT], ag[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) ax[j] = ag[j] = 0.f;
#pragma unroll 4
        for (int o = 0; o < C2; ++o) {
            const float d = dv[o * TP + p];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                ax[j] += wv[o * 2 * C + g * CPT + j] * d;
                ag[j] += wv[o * 2 * C + C + g * CPT + j] * d;
            }
        }
        float s1 = 0.f, s2 = 0.f, apre[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) apre[j] = ok ? a.a_pre[((size_t)t.b * C + g * CPT + j) * P + t.gp0 + p] : 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = g * CPT + j;
            if (ok) {
                const size_t idx = ((size_t)t.b * C + c) * P + t.gp0 + p;
                const float gav = in[(C + c) * TP + p] > 0.f ? ag[j] : 0.f;
                a.dxb[idx] = ax[j];
                a.ga[idx] = gav;
                s1 += gav;
                s2 += gav * ((apre[j] - muA) * rsA);
            }
        }
        block_sum2(s1, s2, red8);
        if (tid == 0) {
            part[part_off(BN_A, C)] = s1;
            part[part_off(BN_A, C) + 1] = s2;
        }
    }
    {   // d W_v[o][i] partial = sum_p dv[o][p] in[i][p], d b_v[o] = sum_p dv[o][p]: thread = (o = tid % C2, i-group)
        constexpr int NIG = 256 / C2, IPT = 2 * C / NIG;
        const int o = tid % C2, ig = tid / C2;
        float acc[IPT], sb = 0.f;
#pragma unroll
        for (int i = 0; i < IPT; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int q = 0; q < TP; ++q) {
            const float d = dv[o * TP + q];
            sb += d;
#pragma unroll
            for (int i = 0; i < IPT; ++i) acc[i] += d * in[(ig * IPT + i) * TP + q];
        }
        float* dst = a.p_wv + (t.wg * C2 + o) * 2 * C + ig * IPT;
#pragma unroll
        for (int i = 0; i < IPT; ++i) dst[i] = acc[i];
        if (ig == 0) a.p_bv[t.wg * C2 + o] = sb;
    }
    {   // dr_pre, the reduce layer's parameter gradients and the BN_e sums
        const float mR1 = cr[0], mR2 = cr[1], muR = cr[2], rsR = cr[3], scR = cr[4];
        float wsum[4] = {0.f, 0.f, 0.f, 0.f}, bsum = 0.f;
        const float w0 = a.wr[0], w1 = a.wr[1], w2 = a.wr[2], w3 = a.wr[3];
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int o = g * OPT + j;
            const float muE = a.bnstat[(BN_E * C + o) * 2], rsE = a.bnstat[(BN_E * C + o) * 2 + 1];
            const float scE = a.gamma[BN_E][o] * rsE, shE = a.beta[BN_E][o] - muE * scE;
            float s1 = 0.f, s2 = 0.f;
            if (ok) {
                const size_t idx = ((size_t)t.b * C2 + o) * P + t.gp0 + p;
                const float dr = scR * (a.gr[idx] - mR1 - ((a.r_pre[idx] - muR) * rsR) * mR2);
                a.dr_pre[idx] = dr;
                bsum += dr;
                const floatx4 up = *reinterpret_cast<const floatx4*>(a.u_pre + idx * 4);
                const float wk[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float u = relu_bn(up[k], scE, shE);
                    wsum[k] += dr * u;
                    const float gu = u > 0.f ? wk[k] * dr : 0.f;
                    s1 += gu;
                    s2 += gu * ((up[k] - muE) * rsE);
                }
            }
            s1 = sum32(s1);
            s2 = sum32(s2);
            if (p == 0) {
                part[part_off(BN_E, C) + 2 * o] = s1;
                part[part_off(BN_E, C) + 2 * o + 1] = s2;
            }
        }
        block_sum2(wsum[0], wsum[1], red8);
        block_sum2(wsum[2], wsum[3], red8);
        float dummy = 0.f;
        block_sum2(bsum, dummy, red8);
        if (tid == 0) {
            float* d = a.p_wr + t.wg * 5;
            d[0] = wsum[0], d[1] = wsum[1], d[2] = wsum[2], d[3] = wsum[3], d[4] = bsum;
        }
    }
}

// B3: da_pre, du_pre -> d w_a / d b_a, d W_e / d b_e partial rows, de [B,C,P,4], dx piece added into dxb
template <int C>
__global__ __launch_bounds__(256) void dagem_b3(kmu_dagem_args a, int NWG) {
    constexpr int C2 = C / 2, OPT = C2 / NGR;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    double* tmp = reinterpret_cast<double*>(sm);
    double* red = tmp + 2 * C;
    float* xs = reinterpret_cast<float*>(red + 512);        // [5][C][32]
    float* du = xs + 5 * C * TP;                            // [C2][32][4]
    float* we = du + C2 * TP * 4;                           // [C2][2C]
    float* cf = we + C2 * 2 * C;                            // 6 x [C2] (BN_e), then 6 (BN_a)
    float* red8 = cf + 6 * C2 + 6;
    float *m1 = cf, *m2 = cf + C2, *mean = cf + 2 * C2, *rstd = cf + 3 * C2, *sc = cf + 4 * C2, *sh = cf + 5 * C2, *ca = cf + 6 * C2;
    const int tid = threadIdx.x, P = a.H * a.W;
    const Tile t = tile_of(a);
    const bool writer = blockIdx.x == 0;
    bn_bwd_coeffs<C2>(a, BN_E, a.part_bwd, NWG, (double)a.B * P * 4, writer, m1, m2, mean, rstd, sc, sh, tmp, red);
    bn_bwd_coeffs<1>(a, BN_A, a.part_bwd, NWG, (double)a.B * C * P, writer, ca, ca + 1, ca + 2, ca + 3, ca + 4, ca + 5, tmp, red);
    load_tile5(xs, a.x, C, a.H, a.W, t);
    load_vec(we, a.we, C2 * 2 * C);
    const int p = tid & (TP - 1), g = tid >> TPS;
    const bool ok = p < t.np;
    {   // du_pre[o][p][k] into LDS
        const float w0 = a.wr[0], w1 = a.wr[1], w2 = a.wr[2], w3 = a.wr[3];
        const float wk[4] = {w0, w1, w2, w3};
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int o = g * OPT + j;
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const size_t idx = ((size_t)t.b * C2 + o) * P + t.gp0 + p;
                const float dr = a.dr_pre[idx];
                const floatx4 up = *reinterpret_cast<const floatx4*>(a.u_pre + idx * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gu = relu_bn(up[k], sc[o], sh[o]) > 0.f ? wk[k] * dr : 0.f;
                    v[k] = sc[o] * (gu - m1[o] - ((up[k] - mean[o]) * rstd[o]) * m2[o]);
                }
            }
            *reinterpret_cast<floatx4*>(du + (o * TP + p) * 4) = v;
        }
    }
    __syncthreads();
    {   // per (c, p): da_pre, de = w_a da_pre + sum_o W_e[o][C + c] du[o][p][.], dx += sum_o W_e[o][c] sum_k du; thread = (p, 8 channel groups)
        constexpr int CPT = C / NGR;
        const float mA1 = ca[0], mA2 = ca[1], muA = ca[2], rsA = ca[3], scA = ca[4];
        const float wa0 = a.wa[0], wa1 = a.wa[1], wa2 = a.wa[2], wa3 = a.wa[3];
        float was[4] = {0.f, 0.f, 0.f, 0.f}, bas = 0.f;
        float gav[CPT], apre[CPT], dxv[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const size_t idx = ((size_t)t.b * C + g * CPT + j) * P + t.gp0 + p;
            gav[j] = ok ? a.ga[idx] : 0.f;
            apre[j] = ok ? a.a_pre[idx] : 0.f;
            dxv[j] = ok ? a.dxb[idx] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = g * CPT + j;
            floatx4 de = {0.f, 0.f, 0.f, 0.f};
            float dx = 0.f;
    #pragma unroll 4
        for (int o = 0; o < C2; ++o) {
                const floatx4 d = *reinterpret_cast<const floatx4*>(du + (o * TP + p) * 4);
                const float w2 = we[o * 2 * C + C + c];
                de += w2 * d;
                dx += we[o * 2 * C + c] * ((d[0] + d[1]) + (d[2] + d[3]));
            }
            if (ok) {
                const size_t idx = ((size_t)t.b * C + c) * P + t.gp0 + p;
                const float da = scA * (gav[j] - mA1 - ((apre[j] - muA) * rsA) * mA2);
                const float x0 = xs[c * TP + p];
                const float e0 = x0 * xs[(C + c) * TP + p], e1 = x0 * xs[(2 * C + c) * TP + p], e2 = x0 * xs[(3 * C + c) * TP + p],
                            e3 = x0 * xs[(4 * C + c) * TP + p];
                was[0] += da * e0, was[1] += da * e1, was[2] += da * e2, was[3] += da * e3;
                bas += da;
                de += floatx4{wa0 * da, wa1 * da, wa2 * da, wa3 * da};
                *reinterpret_cast<floatx4*>(a.de + idx * 4) = de;
                a.dxb[idx] = dxv[j] + dx;
            }
        }
        block_sum2(was[0], was[1], red8);
        block_sum2(was[2], was[3], red8);
        float dummy = 0.f;
        block_sum2(bas, dummy, red8);
        if (tid == 0) {
            float* d = a.p_wa + t.wg * 5;
            d[0] = was[0], d[1] = was[1], d[2] = was[2], d[3] = was[3], d[4] = bas;
        }
    }
    {   // d W_e[o][i] partial: i < C: sum_p (sum_k du[o][p][k]) x[i][p];  i >= C: sum_{p,k} du[o][p][k] e[i - C][p][k];  d b_e[o]
        constexpr int NIG = 256 / C2, IPT = C / NIG;       // every thread: IPT x-columns and the same IPT edge columns
        const int o = tid % C2, ig = tid / C2;
        float ax[IPT], ae[IPT], sb = 0.f;
#pragma unroll
        for (int i = 0; i < IPT; ++i) ax[i] = ae[i] = 0.f;
#pragma unroll 4
        for (int q = 0; q < TP; ++q) {
            const floatx4 d = *reinterpret_cast<const floatx4*>(du + (o * TP + q) * 4);
            const float ds = (d[0] + d[1]) + (d[2] + d[3]);
            sb += ds;
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int c = ig * IPT + i;
                const float x0 = xs[c * TP + q];
                ax[i] += ds * x0;
                ae[i] += x0 * (d[0] * xs[(C + c) * TP + q] + d[1] * xs[(2 * C + c) * TP + q] + d[2] * xs[(3 * C + c) * TP + q] +
                               d[3] * xs[(4 * C + c) * TP + q]);
            }
        }
        float* dst = a.p_we + (t.wg * C2 + o) * 2 * C;
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            dst[ig * IPT + i] = ax[i];
            dst[C + ig * IPT + i] = ae[i];
        }
        if (ig == 0) a.p_be[t.wg * C2 + o] = sb;
    }
}

// B4: dx = adjoint of the edge products (gather form, as csrc/dagem.hip) + g_dd (the residual of deform(x) + x) + dxb
__global__ __launch_bounds__(256) void dagem_b4(kmu_dagem_args a) {
    const int H = a.H, W = a.W, HW = H * W;
    const size_t total = (size_t)a.B * a.C * HW, t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int p = (int)(t % HW), h = p / W, w = p - h * W;
    const float* xp = a.x + (t - p);
    const floatx4* gp = reinterpret_cast<const floatx4*>(a.de) + (t - p);
    const int hm = h == 0 ? H - 1 : h - 1, hp = h == H - 1 ? 0 : h + 1, wm = w == 0 ? W - 1 : w - 1, wp = w == W - 1 ? 0 : w + 1;
    const int q0 = hm * W + w, q1 = hp * W + w, q2 = h * W + wm, q3 = h * W + wp;
    const floatx4 own = gp[p];
    float s = own[0] * xp[q0] + own[1] * xp[q1] + own[2] * xp[q2] + own[3] * xp[q3];
    s += gp[q1][0] * xp[q1] + gp[q0][1] * xp[q0] + gp[q3][2] * xp[q3] + gp[q2][3] * xp[q2];
    a.dx[t] = s + a.g_dd[t] + a.dxb[t];
}

template <int C>
size_t lds_of(int stage) {
    constexpr int C2 = C / 2, KF = C + C2;
    const size_t fold = (size_t)(2 * C + 512) * sizeof(double);
    switch (stage) {
        case 0: return (size_t)(5 * C * TP + C2 * 2 * C + 8) * 4;
        case 1: return fold + (size_t)(2 * C * TP + C2 * 2 * C + 2 * C2 + 2 + 8) * 4;
        case 2: return fold + (size_t)(KF * TP + C * KF + 2 * C2 + 2) * 4;
        case 3: return fold + (size_t)(2 * C) * 4;
        case 4: return 0;
        case 5: return fold + (size_t)(C * TP + KF * TP + C * KF + 6 * C + 8) * 4;
        case 6: return fold + (size_t)(2 * C * TP + C2 * TP + C2 * 2 * C + 6 * C2 + 6 + 8) * 4;
        case 7: return fold + (size_t)(5 * C * TP + C2 * TP * 4 + C2 * 2 * C + 6 * C2 + 6 + 8) * 4;
        default: return 0;
    }
}

template <int C>
int run_stage(const kmu_dagem_args& a, int stage, hipStream_t st) {
    const int P = a.H * a.W, NWG = a.B * ((P + TP - 1) / TP);
    const size_t lds = lds_of<C>(stage);
    const dim3 grid(NWG), blk(256);
    switch (stage) {
        case 0: KMU_MAX_LDS(dagem_f0<C>, lds); hipLaunchKernelGGL(dagem_f0<C>, grid, blk, lds, st, a); break;
        case 1: KMU_MAX_LDS(dagem_f1<C>, lds); hipLaunchKernelGGL(dagem_f1<C>, grid, blk, lds, st, a, NWG); break;
        case 2: KMU_MAX_LDS(dagem_f2<C>, lds); hipLaunchKernelGGL(dagem_f2<C>, grid, blk, lds, st, a, NWG); break;
        case 3: hipLaunchKernelGGL(dagem_f3<C>, grid, blk, lds, st, a, NWG); break;
        case 4: hipLaunchKernelGGL(dagem_b0<C>, grid, blk, 0, st, a); break;
        case 5: KMU_MAX_LDS(dagem_b1<C>, lds); hipLaunchKernelGGL(dagem_b1<C>, grid, blk, lds, st, a, NWG); break;
        case 6: KMU_MAX_LDS(dagem_b2<C>, lds); hipLaunchKernelGGL(dagem_b2<C>, grid, blk, lds, st, a, NWG); break;
        case 7: KMU_MAX_LDS(dagem_b3<C>, lds); hipLaunchKernelGGL(dagem_b3<C>, grid, blk, lds, st, a, NWG); break;
        default: {
            const size_t total = (size_t)a.B * C * P;
            hipLaunchKernelGGL(dagem_b4, dim3((unsigned)((total + 255) / 256)), blk, 0, st, a);
        }
    }
    return kmu::launch_status("dagem_stage");
}

}  // namespace

extern "C" size_t kmu_dagem_args_bytes(void) { return sizeof(kmu_dagem_args); }
extern "C" int kmu_dagem_supported(int C) { return C == 32 || C == 64; }
extern "C" int kmu_dagem_tiles(int B, int H, int W) { return B * ((H * W + TP - 1) / TP); }
extern "C" size_t kmu_dagem_part_floats(int B, int C, int H, int W) { return (size_t)kmu_dagem_tiles(B, H, W) * part_stride(C); }

extern "C" int kmu_dagem_stage(const kmu_dagem_args* args, int stage, kmu_stream_t stream) {
    KMU_REQUIRE(args, "dagem_stage: null argument block");
    const kmu_dagem_args& a = *args;
    KMU_REQUIRE(stage >= 0 && stage <= 8, "dagem_stage: stage %d (0..3 forward, 4..8 backward)", stage);
    KMU_REQUIRE(kmu_dagem_supported(a.C) && a.B > 0 && a.H > 0 && a.W > 0, "dagem_stage: C=%d (32 / 64), B=%d, H=%d, W=%d", a.C, a.B, a.H, a.W);
    KMU_REQUIRE(a.x && a.wa && a.ba && a.wv && a.bv && a.we && a.be && a.wr && a.br && a.wf && a.a_pre && a.u_pre && a.v_pre && a.r_pre && a.z &&
                    a.bnstat && a.part,
                "dagem_stage: null pointer");
    for (int i = 0; i < 5; ++i)
        KMU_REQUIRE(a.gamma[i] && a.beta[i] && a.running_mean[i] && a.running_var[i], "dagem_stage: BatchNorm %d lacks parameters / running statistics", i);
    if (stage <= 3) KMU_REQUIRE(stage != 2 || a.dconv, "dagem_stage: stage 2 needs the deformable convolution's output");
    if (stage == 3) KMU_REQUIRE(a.out, "dagem_stage: null output");
    if (stage >= 4) {
        KMU_REQUIRE(a.g_out && a.part_bwd && a.dconv && a.g_dd && a.gv && a.gr && a.ga && a.dr_pre && a.dxb && a.de && a.dx && a.p_wf && a.p_wv &&
                        a.p_bv && a.p_we && a.p_be && a.p_wa && a.p_wr,
                    "dagem_stage: null backward pointer");
        for (int i = 0; i < 5; ++i) KMU_REQUIRE(a.d_gamma[i] && a.d_beta[i], "dagem_stage: BatchNorm %d lacks gradient outputs", i);
    }
    hipStream_t st = (hipStream_t)stream;
    if (a.C == 32) return run_stage<32>(a, stage, st);
    return run_stage<64>(a, stage, st);
}
