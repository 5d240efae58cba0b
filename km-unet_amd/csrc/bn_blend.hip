// Fused BatchNorm2d (+ReLU) (+ sigmoid-alpha blend) for gfx950 -- the elementwise chain that wraps every
// convolution of EfficientViMBlock (vim_block_init/efficient_vim_init.py:81-97):
//       x <- (1 - a) * x + a * f(t) ,   a = sigmoid(alpha[k]) per channel           (:82,:85,:90,:93,:96)
//       f(t) = BatchNorm2d(t) [then ReLU]   (ConvLayer2D, vim_utils_init.py:62-89)   or   f(t) = t   (:90)
// Train mode uses batch statistics (biased variance for normalisation, unbiased for running_var, as
// nn.BatchNorm2d); eval mode uses the running statistics.  Everything here is HBM-bound streaming:
//   forward : stats kernel  (grid C x S: per-block sum / sum-of-squares partials)
//             apply kernel  (every block re-reduces its channel's <= 32 partials, then streams)
//   backward: reduce kernel (per-block partials of sum dz, sum dz*that, sum g*(f - x))
//             apply kernel  (dt, dx; block 0 of each channel finalises d_gamma, d_beta, d_alpha)
// versus 3 launches forward and ~9 backward through MIOpen BatchNorm + ATen lerp/ReLU autograd.
#include "common.h"

using kmu::floatx4;

namespace {

#ifndef KMU_BN_MAXS
#define KMU_BN_MAXS 32
#endif
#ifndef KMU_BN_CHUNK
#define KMU_BN_CHUNK 4096
#endif
constexpr int MAXS = KMU_BN_MAXS;

struct Split {
    int S, chunk;  // S blocks per channel, each `chunk` (multiple of 4) flattened (b,p) positions
};
inline Split split_for(int B, int HW) {
    const long n = (long)B * HW;
    int S = (int)((n + KMU_BN_CHUNK - 1) / KMU_BN_CHUNK);
    if (S < 1) S = 1;
    if (S > MAXS) S = MAXS;
    long chunk = (n + S - 1) / S;
    chunk = (chunk + 3) / 4 * 4;
    Split r;
    r.S = (int)((n + chunk - 1) / chunk);
    r.chunk = (int)chunk;
    return r;
}

__device__ __forceinline__ float block_sum(float v, float* red) {  // red: [4] floats of LDS, result in all threads
    v = kmu::wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// position n in [0, B*HW) of channel c  ->  element offset
__device__ __forceinline__ size_t off_of(long n, int c, int C, int HW) {
    const long b = n / HW, p = n - b * HW;
    return ((size_t)b * C + c) * HW + p;
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ t, float* __restrict__ part, int B,
                                                       int C, int HW, int chunk) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const long n0 = (long)s * chunk, n1 = min((long)B * HW, n0 + chunk);
    float a = 0.f, q = 0.f;
    if ((HW & 3) == 0) {
        for (long n = n0 + threadIdx.x * 4; n < n1; n += 1024) {
            const floatx4 v = *reinterpret_cast<const floatx4*>(t + off_of(n, c, C, HW));
            a += (v[0] + v[1]) + (v[2] + v[3]);
            q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    } else {
        for (long n = n0 + threadIdx.x; n < n1; n += 256) {
            const float v = t[off_of(n, c, C, HW)];
            a += v;
            q += v * v;
        }
    }
    a = block_sum(a, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        part[((size_t)c * S + s) * 2] = a;
        part[((size_t)c * S + s) * 2 + 1] = q;
    }
}

// out = blend(x, act(bn(t)), sigmoid(alpha)) ; stats_out[c] = {mean, rstd}
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ t, const float* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ alpha, const float* __restrict__ part,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       float momentum, float eps, int relu, int training,
                                                       float* __restrict__ out, float* __restrict__ stats_out,
                                                       long long* __restrict__ num_batches_tracked, int B, int C, int HW,
                                                       int chunk, int SP) {
    // SP: partial pairs per channel in `part` (= gridDim.y when bn_stats_kernel made them; the producing convolution's workgroups
    // per channel when IT made them: kmu_bn_blend_fwd_pre)
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    if (num_batches_tracked && training && gamma && c == 0 && s == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const long N = (long)B * HW;
    float mean = 0.f, rstd = 1.f, g = 1.f, bt = 0.f;
    if (gamma) {
        if (training) {
            double da = 0.0, dq = 0.0;
            for (int i = threadIdx.x; i < SP; i += 256) {
                da += part[((size_t)c * SP + i) * 2];
                dq += part[((size_t)c * SP + i) * 2 + 1];
            }
            const double sa = (double)block_sum((float)da, red), sq = (double)block_sum((float)dq, red);
            const double m = sa / (double)N;
            double var = sq / (double)N - m * m;
            if (var < 0.0) var = 0.0;
            mean = (float)m;
            rstd = (float)(1.0 / sqrt(var + (double)eps));
            if (s == 0 && threadIdx.x == 0) {
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                const double unbiased = N > 1 ? var * (double)N / (double)(N - 1) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        } else {
            mean = running_mean[c];
            rstd = 1.f / sqrtf(running_var[c] + eps);
        }
        g = gamma[c];
        bt = beta[c];
        if (s == 0 && threadIdx.x == 0) {
            stats_out[c * 2] = mean;
            stats_out[c * 2 + 1] = rstd;
        }
    }
    const float scale = g * rstd, shift = bt - mean * scale;
    const float a = alpha ? 1.f / (1.f + __expf(-alpha[c])) : 1.f;
    const long n0 = (long)s * chunk, n1 = min(N, n0 + chunk);
    if ((HW & 3) == 0) {
        for (long n = n0 + threadIdx.x * 4; n < n1; n += 1024) {
            const size_t o = off_of(n, c, C, HW);
            floatx4 v = *reinterpret_cast<const floatx4*>(t + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = v[k] * scale + shift;
                if (relu) v[k] = fmaxf(v[k], 0.f);
            }
            if (x) {
                const floatx4 xv = *reinterpret_cast<const floatx4*>(x + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = xv[k] + a * (v[k] - xv[k]);
            }
            *reinterpret_cast<floatx4*>(out + o) = v;
        }
    } else {
        for (long n = n0 + threadIdx.x; n < n1; n += 256) {
            const size_t o = off_of(n, c, C, HW);
            float v = t[o] * scale + shift;
            if (relu) v = fmaxf(v, 0.f);
            if (x) v = x[o] + a * (v - x[o]);
            out[o] = v;
        }
    }
}

// partial sums per (c, s): [0] sum dz, [1] sum dz*that, [2] sum g*(f - x)
//   f = act(bn(t)); dz = d loss / d bn-output (masked by the ReLU); that = (t - mean) * rstd
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ gout, const float* __restrict__ t,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ alpha,
                                                            const float* __restrict__ stats, int relu,
                                                            float* __restrict__ part, int B, int C, int HW, int chunk) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const long N = (long)B * HW, n0 = (long)s * chunk, n1 = min(N, n0 + chunk);
    const float mean = gamma ? stats[c * 2] : 0.f, rstd = gamma ? stats[c * 2 + 1] : 1.f;
    const float g = gamma ? gamma[c] : 1.f, bt = gamma ? beta[c] : 0.f;
    const float a = alpha ? 1.f / (1.f + __expf(-alpha[c])) : 1.f;
    const float scale = g * rstd, shift = bt - mean * scale;  // the forward's exact arithmetic => identical ReLU mask
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    auto body = [&](float tv, float go, float xv) {
        const float that = (tv - mean) * rstd;
        float f = tv * scale + shift;
        const bool on = !relu || f > 0.f;
        f = on ? f : 0.f;
        const float dz = on ? a * go : 0.f;
        s0 += dz;
        s1 += dz * that;
        s2 += go * (f - xv);
    };
    if ((HW & 3) == 0) {
        for (long n = n0 + threadIdx.x * 4; n < n1; n += 1024) {
            const size_t o = off_of(n, c, C, HW);
            const floatx4 tv = *reinterpret_cast<const floatx4*>(t + o), gv = *reinterpret_cast<const floatx4*>(gout + o);
            floatx4 xv = {0.f, 0.f, 0.f, 0.f};
            if (x) xv = *reinterpret_cast<const floatx4*>(x + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) body(tv[k], gv[k], xv[k]);   // s2 is only consumed when x != NULL
        }
    } else {
        for (long n = n0 + threadIdx.x; n < n1; n += 256) {
            const size_t o = off_of(n, c, C, HW);
            body(t[o], gout[o], x ? x[o] : 0.f);
        }
    }
    s0 = block_sum(s0, red);
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        float* p = part + ((size_t)c * S + s) * 3;
        p[0] = s0;
        p[1] = s1;
        p[2] = s2;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ gout, const float* __restrict__ t,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ alpha,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ part, int relu, int training,
                                                           float* __restrict__ dt, float* __restrict__ dx,
                                                           float* __restrict__ d_gamma, float* __restrict__ d_beta,
                                                           float* __restrict__ d_alpha, int B, int C, int HW,
                                                           int chunk) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const long N = (long)B * HW, n0 = (long)s * chunk, n1 = min(N, n0 + chunk);
    double q0 = 0.0, q1 = 0.0, q2 = 0.0;
    for (int i = threadIdx.x; i < S; i += 256) {
        const float* p = part + ((size_t)c * S + i) * 3;
        q0 += p[0];
        q1 += p[1];
        q2 += p[2];
    }
    const double p0 = (double)block_sum((float)q0, red), p1 = (double)block_sum((float)q1, red), p2 = (double)block_sum((float)q2, red);
    const float mean = gamma ? stats[c * 2] : 0.f, rstd = gamma ? stats[c * 2 + 1] : 1.f;
    const float g = gamma ? gamma[c] : 1.f, bt = gamma ? beta[c] : 0.f;
    const float a = alpha ? 1.f / (1.f + __expf(-alpha[c])) : 1.f;
    if (s == 0 && threadIdx.x == 0) {
        if (d_gamma) {
            d_gamma[c] = (float)p1;
            d_beta[c] = (float)p0;
        }
        if (d_alpha) d_alpha[c] = (float)p2 * a * (1.f - a);
    }
    // train: dt = g*rstd*(dz - mean(dz) - that*mean(dz*that)) ; eval or no BN: dt = g*rstd*dz
    const float m0 = (gamma && training) ? (float)(p0 / (double)N) : 0.f;
    const float m1 = (gamma && training) ? (float)(p1 / (double)N) : 0.f;
    const float k = g * rstd, shift = bt - mean * k;
    auto one = [&](float tv, float go, float& dtv, float& dxv) {
        const float that = (tv - mean) * rstd;
        const bool on = !relu || (tv * k + shift) > 0.f;  // same expression as bn_apply_kernel
        const float dz = on ? a * go : 0.f;
        dtv = k * (dz - m0 - that * m1);
        dxv = (1.f - a) * go;
    };
    if ((HW & 3) == 0) {
        for (long n = n0 + threadIdx.x * 4; n < n1; n += 1024) {
            const size_t o = off_of(n, c, C, HW);
            const floatx4 tv = *reinterpret_cast<const floatx4*>(t + o), gv = *reinterpret_cast<const floatx4*>(gout + o);
            float dta[4], dxa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) one(tv[i], gv[i], dta[i], dxa[i]);
            *reinterpret_cast<floatx4*>(dt + o) = floatx4{dta[0], dta[1], dta[2], dta[3]};
            if (dx) *reinterpret_cast<floatx4*>(dx + o) = floatx4{dxa[0], dxa[1], dxa[2], dxa[3]};
        }
    } else {
        for (long n = n0 + threadIdx.x; n < n1; n += 256) {
            const size_t o = off_of(n, c, C, HW);
            float dtv, dxv;
            one(t[o], gout[o], dtv, dxv);
            dt[o] = dtv;
            if (dx) dx[o] = dxv;
        }
    }
}

}  // namespace

extern "C" int kmu_bn_blend_splits(int B, int HW) { return split_for(B, HW).S; }

extern "C" int kmu_bn_blend_fwd(const float* t, const float* x, const float* gamma, const float* beta,
                                const float* alpha, float* running_mean, float* running_var, float momentum, float eps,
                                int relu, int training, float* out, float* stats, float* ws, long long* num_batches_tracked,
                                int B, int C, int HW, kmu_stream_t stream) {
    KMU_REQUIRE(t && out, "bn_blend_fwd: null pointer");
    KMU_REQUIRE(!gamma || (beta && running_mean && running_var && stats && ws), "bn_blend_fwd: BatchNorm needs beta, running stats, stats, ws");
    KMU_REQUIRE(!alpha || x, "bn_blend_fwd: a blend needs x");
    KMU_REQUIRE(B > 0 && C > 0 && C <= 65535 && HW > 0, "bn_blend_fwd: bad dims");
    hipStream_t st = (hipStream_t)stream;
    const Split sp = split_for(B, HW);
    if (gamma && training) {
        hipLaunchKernelGGL(bn_stats_kernel, dim3(C, sp.S), dim3(256), 0, st, t, ws, B, C, HW, sp.chunk);
        int rc = kmu::launch_status("bn_blend_fwd stats");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(C, sp.S), dim3(256), 0, st, t, x, gamma, beta, alpha, ws, running_mean,
                       running_var, momentum, eps, relu, training, out, stats, num_batches_tracked, B, C, HW, sp.chunk, sp.S);
    return kmu::launch_status("bn_blend_fwd apply");
}

// Train-mode forward with the statistics partials ALREADY made by the kernel that produced t (kmu_dwconv3x3_fwd_stats,
// kmu_pwconv_fwd_stats): stat_part [C][S_part][2] = per-workgroup (sum, sum of squares).  One launch instead of two.
extern "C" int kmu_bn_blend_fwd_pre(const float* t, const float* x, const float* gamma, const float* beta, const float* alpha,
                                    float* running_mean, float* running_var, float momentum, float eps, int relu, float* out,
                                    float* stats, const float* stat_part, int S_part, long long* num_batches_tracked, int B, int C,
                                    int HW, kmu_stream_t stream) {
    KMU_REQUIRE(t && out && gamma && beta && running_mean && running_var && stats && stat_part && S_part > 0,
                "bn_blend_fwd_pre: null pointer / no partials");
    KMU_REQUIRE(!alpha || x, "bn_blend_fwd_pre: a blend needs x");
    KMU_REQUIRE(B > 0 && C > 0 && C <= 65535 && HW > 0, "bn_blend_fwd_pre: bad dims");
    const Split sp = split_for(B, HW);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(C, sp.S), dim3(256), 0, (hipStream_t)stream, t, x, gamma, beta, alpha, stat_part,
                       running_mean, running_var, momentum, eps, relu, 1, out, stats, num_batches_tracked, B, C, HW, sp.chunk, S_part);
    return kmu::launch_status("bn_blend_fwd_pre");
}

extern "C" int kmu_bn_blend_bwd(const float* gout, const float* t, const float* x, const float* gamma, const float* beta,
                                const float* alpha, const float* stats, int relu, int training, float* dt, float* dx,
                                float* d_gamma, float* d_beta, float* d_alpha, float* ws, int B, int C, int HW,
                                kmu_stream_t stream) {
    KMU_REQUIRE(gout && t && dt && ws, "bn_blend_bwd: null pointer");
    KMU_REQUIRE(!gamma || (beta && stats && d_gamma && d_beta), "bn_blend_bwd: BatchNorm needs beta, stats, d_gamma, d_beta");
    KMU_REQUIRE(!alpha || (x && dx && d_alpha), "bn_blend_bwd: a blend needs x, dx, d_alpha");
    KMU_REQUIRE(B > 0 && C > 0 && C <= 65535 && HW > 0, "bn_blend_bwd: bad dims");
    hipStream_t st = (hipStream_t)stream;
    const Split sp = split_for(B, HW);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, sp.S), dim3(256), 0, st, gout, t, x, gamma, beta, alpha, stats, relu, ws,
                       B, C, HW, sp.chunk);
    int rc = kmu::launch_status("bn_blend_bwd reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(C, sp.S), dim3(256), 0, st, gout, t, gamma, beta, alpha, stats, ws, relu,
                       training, dt, dx, d_gamma, d_beta, d_alpha, B, C, HW, sp.chunk);
    return kmu::launch_status("bn_blend_bwd apply");
}

// The reduction half of kmu_bn_blend_bwd alone: partials [C][S][3] (S = kmu_bn_blend_splits) of (sum dz, sum dz that, sum g (f - x)).
// For callers that fold the apply half into their own kernel (kmu_dwconv3x3_bn_bwd_data).
extern "C" int kmu_bn_blend_bwd_partials(const float* gout, const float* t, const float* x, const float* gamma, const float* beta,
                                         const float* alpha, const float* stats, int relu, float* part, int B, int C, int HW,
                                         kmu_stream_t stream) {
    KMU_REQUIRE(gout && t && part, "bn_blend_bwd_partials: null pointer");
    KMU_REQUIRE(!gamma || (beta && stats), "bn_blend_bwd_partials: BatchNorm needs beta, stats");
    KMU_REQUIRE(!alpha || x, "bn_blend_bwd_partials: a blend needs x");
    KMU_REQUIRE(B > 0 && C > 0 && C <= 65535 && HW > 0, "bn_blend_bwd_partials: bad dims");
    const Split sp = split_for(B, HW);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, sp.S), dim3(256), 0, (hipStream_t)stream, gout, t, x, gamma, beta, alpha, stats, relu,
                       part, B, C, HW, sp.chunk);
    return kmu::launch_status("bn_blend_bwd_partials");
}
