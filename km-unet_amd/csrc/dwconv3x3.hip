// Depthwise 3x3 / stride 1 / pad 1 convolution for gfx950 (forward, input gradient, weight gradient).
//
// Used 45x per forward by the reference graph: EfficientViMBlock.dwconv1/dwconv2
// (vim_block_init/efficient_vim_init.py:74-75,85,93 -> ConvLayer2D, vim_utils_init.py:62-89, groups=dim,
// bias-free) and DirectionAttention.conv (KM_UNetV3_SH.py:222,263, groups=dim, with bias).  MIOpen serves
// this shape with its naive fallback kernel (~50 us for [8,16,128,128]); it is a pure HBM-bound stencil:
// algorithmic traffic = read x once + write y once.
//   forward / input gradient: one thread per 4 consecutive output pixels (16-B store), 3 rows x 6 taps read
//       straight from global (neighbouring threads share lines through L1/L2); the input gradient is the same
//       stencil with the taps flipped.
//   weight gradient: grid (C, B, row-splits); 9 tap sums + the bias sum per thread, wave shuffle + LDS block
//       reduce, per-block partials (deterministic; the caller sums the small partial tensor).
#include "common.h"

using kmu::floatx4;

namespace {

constexpr int WSPLIT = 4;  // row splits per (b, c) plane in the weight-gradient kernel

// STATS: additionally leave per-workgroup (sum, sum of squares) of the outputs for the BatchNorm that follows (EfficientViMBlock's
// dwconv1 / dwconv2, efficient_vim_init.py:85,93): the host only uses it when a workgroup's 1024 outputs lie in ONE (b, c) plane
// (H*W % 1024 == 0, no grid-stride wrap), so stat_part[(c*S + b*(HW/1024) + k)*2 + {0,1}] is this block's pair -- the layout
// kmu_bn_blend_fwd_pre folds.  Saves the separate statistics pass (one launch and one read of the tensor).
template <bool STATS>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        const float* __restrict__ addend, float* __restrict__ out, int C, int H,
                                                        int W, int flip, size_t total, float* __restrict__ stat_part, int S) {
    const int W4 = (W + 3) >> 2;
    const bool vec_ok = (W & 3) == 0;
    float st_a = 0.f, st_q = 0.f;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int x0 = (int)(t % W4) * 4;
        size_t r = t / W4;
        const int y = (int)(r % H);
        r /= H;
        const int c = (int)(r % C);
        const float* plane = in + r * (size_t)H * W;  // r == b*C + c
        float wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = w[c * 9 + (flip ? 8 - k : k)];
        const float b0 = bias ? bias[c] : 0.f;
        float acc[4] = {b0, b0, b0, b0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            if (yy < 0 || yy >= H) continue;
            const float* row = plane + (size_t)yy * W;
            float v[6];
            if (vec_ok) {
                const floatx4 m = *reinterpret_cast<const floatx4*>(row + x0);
                v[1] = m[0], v[2] = m[1], v[3] = m[2], v[4] = m[3];
                v[0] = x0 > 0 ? row[x0 - 1] : 0.f;
                v[5] = x0 + 4 < W ? row[x0 + 4] : 0.f;
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int xx = x0 - 1 + k;
                    v[k] = (xx >= 0 && xx < W) ? row[xx] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[q] += wv[dy * 3 + dx] * v[q + dx];
        }
        if (scale) {                        // per-(b, c) plane gate, r == b*C + c
            const float sc = scale[r];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] *= sc;
        }
        if (addend) {                       // e.g. the blend partner's gradient: saves autograd's separate fan-in add
            const float* ap = addend + (r * H + y) * (size_t)W + x0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (x0 + q < W) acc[q] += ap[q];
        }
        if (STATS) {
            st_a += (acc[0] + acc[1]) + (acc[2] + acc[3]);
            st_q += (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]);
        }
        float* dst = out + (r * H + y) * (size_t)W + x0;
        if (vec_ok) {
            *reinterpret_cast<floatx4*>(dst) = floatx4{acc[0], acc[1], acc[2], acc[3]};
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (x0 + q < W) dst[q] = acc[q];
        }
    }
    if (STATS) {
        __shared__ float red[2][4];
        st_a = kmu::wave_sum(st_a);
        st_q = kmu::wave_sum(st_q);
        if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = st_a, red[1][threadIdx.x >> 6] = st_q;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int bpp = H * W / 1024;                         // blocks per plane
            const int plane = blockIdx.x / bpp, k = blockIdx.x - plane * bpp, b = plane / C, c = plane - b * C;
            float* p = stat_part + ((size_t)c * S + b * bpp + k) * 2;
            p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}

// dw[c][tap] partial = sum over this block's rows of dy[p] * x[p + tap - centre] ; db partial = sum dy
__global__ __launch_bounds__(256) void dwconv3x3_bwd_weight_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ dy,
                                                                   float* __restrict__ dw_part,
                                                                   float* __restrict__ db_part, int C, int H, int W) {
    __shared__ float red[4][10];
    const int c = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    const int rows = (H + WSPLIT - 1) / WSPLIT, y0 = sp * rows, y1 = min(H, y0 + rows);
    const float* xp = x + ((size_t)b * C + c) * H * W;
    const float* gp = dy + ((size_t)b * C + c) * H * W;
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    const int W4 = (W + 3) >> 2, nstrip = max(0, y1 - y0) * W4;
    const bool vec_ok = (W & 3) == 0;
    for (int sidx = threadIdx.x; sidx < nstrip; sidx += 256) {  // one strip = 4 consecutive pixels of a row
        const int y = y0 + sidx / W4, x0 = (sidx % W4) * 4;
        float g4[4];
        if (vec_ok) {
            const floatx4 gv = *reinterpret_cast<const floatx4*>(gp + (size_t)y * W + x0);
            g4[0] = gv[0], g4[1] = gv[1], g4[2] = gv[2], g4[3] = gv[3];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) g4[q] = (x0 + q < W) ? gp[(size_t)y * W + x0 + q] : 0.f;
        }
        acc[9] += (g4[0] + g4[1]) + (g4[2] + g4[3]);
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy) {
            const int yy = y + dyy - 1;
            if (yy < 0 || yy >= H) continue;
            const float* row = xp + (size_t)yy * W;
            float v[6];
            if (vec_ok) {
                const floatx4 m = *reinterpret_cast<const floatx4*>(row + x0);
                v[1] = m[0], v[2] = m[1], v[3] = m[2], v[4] = m[3];
                v[0] = x0 > 0 ? row[x0 - 1] : 0.f;
                v[5] = x0 + 4 < W ? row[x0 + 4] : 0.f;
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int xx = x0 - 1 + k;
                    v[k] = (xx >= 0 && xx < W) ? row[xx] : 0.f;
                }
            }
#pragma unroll
            for (int dxx = 0; dxx < 3; ++dxx)
                acc[dyy * 3 + dxx] += (g4[0] * v[dxx] + g4[1] * v[dxx + 1]) + (g4[2] * v[dxx + 2] + g4[3] * v[dxx + 3]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float s = kmu::wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const size_t prow = (size_t)b * WSPLIT + sp;
        if (threadIdx.x < 9)
            dw_part[(prow * C + c) * 9 + threadIdx.x] = s;
        else if (db_part)
            db_part[prow * C + c] = s;
    }
}

// With y = s[b,c] * (conv(x) + bias) the per-(b, row-split, c) partials A[.,t] = sum dy*x_shift_t, G = sum dy of the
// weight-gradient kernel (dy unscaled) give everything:   dw[c,t] = sum_b s[b,c] A[b,c,t],   db[c] = sum_b s[b,c] G[b,c],
// ds[b,c] = sum_t w[c,t] A[b,c,t] + bias[c] G[b,c]   (fixed summation order).
__global__ __launch_bounds__(256) void dwconv3x3_scaled_finish_kernel(const float* __restrict__ dw_part, const float* __restrict__ db_part,
                                                                      const float* __restrict__ scale, const float* __restrict__ w,
                                                                      const float* __restrict__ bias, float* __restrict__ dw,
                                                                      float* __restrict__ db, float* __restrict__ dscale, int B, int C) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < C * 9) {
        const int c = t / 9;
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            float a = 0.f;
            for (int sp = 0; sp < WSPLIT; ++sp) a += dw_part[((size_t)(b * WSPLIT + sp) * C) * 9 + t];
            s += scale[b * C + c] * a;
        }
        dw[t] = s;
    } else if (t < C * 10) {
        const int c = t - C * 9;
        if (db) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) {
                float g = 0.f;
                for (int sp = 0; sp < WSPLIT; ++sp) g += db_part[(size_t)(b * WSPLIT + sp) * C + c];
                s += scale[b * C + c] * g;
            }
            db[c] = s;
        }
    } else if (t < C * 10 + B * C) {
        const int e = t - C * 10, b = e / C, c = e - b * C;
        float s = 0.f;
        for (int sp = 0; sp < WSPLIT; ++sp) {
            const float* a = dw_part + ((size_t)(b * WSPLIT + sp) * C + c) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) s += w[c * 9 + k] * a[k];
            if (bias) s += bias[c] * db_part[(size_t)(b * WSPLIT + sp) * C + c];
        }
        dscale[e] = s;
    }
}


// ---- EfficientViMBlock's dwconv stage backward, BatchNorm folded in (efficient_vim_init.py:85,93: x + a (BN(dwconv(x)) - x)) -----------
// g = d loss / d out.  With t = dwconv(x) saved and (sum dz, sum dz that, sum g (BN(t) - x)) per channel left as partials by
// bn_bwd_reduce_kernel (csrc/bn_blend.hip), the gradient of t is an affine function of (g, t) per channel,
//        dt = k (a g - m0 - that m1) = A g + B t + D,        that = (t - mean) rstd,  k = gamma rstd,
// so the separate BatchNorm-backward apply pass (read g, t; write dt, dxb) disappears: the transposed stencil evaluates dt at its 18
// taps on the fly (0 outside the image) and adds the blend partner's gradient (1 - a) g at the centre:
//        dx = dwconv^T(dt) + (1 - a) g.
// Every workgroup folds the <= 32 partial triples of all C channels itself (fixed order, double, as bn_bwd_apply_kernel does);
// workgroup 0 writes d_gamma / d_beta / d_alpha and the constants table cst [C][4] = {A, B, D, 1 - a} that the weight-gradient
// kernel (off the activation-gradient chain) reads to form the same dt.
__global__ __launch_bounds__(256) void dwconv3x3_bn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ t,
                                                               const float* __restrict__ w, const float* __restrict__ gamma,
                                                               const float* __restrict__ alpha, const float* __restrict__ stats,
                                                               const float* __restrict__ part, int S, int training, double N,
                                                               float* __restrict__ dx, float* __restrict__ d_gamma,
                                                               float* __restrict__ d_beta, float* __restrict__ d_alpha,
                                                               float* __restrict__ cst_out, int C, int H, int W, size_t total) {
    extern __shared__ __attribute__((aligned(16))) float cst[];      // [C][4]
    for (int c = threadIdx.x; c < C; c += 256) {
        double q0 = 0.0, q1 = 0.0, q2 = 0.0;
        for (int i = 0; i < S; ++i) {
            const float* p = part + ((size_t)c * S + i) * 3;
            q0 += p[0], q1 += p[1], q2 += p[2];
        }
        const float mean = stats[2 * c], rstd = stats[2 * c + 1], a = 1.f / (1.f + __expf(-alpha[c]));
        const float k = gamma[c] * rstd;
        const float m0 = training ? (float)(q0 / N) : 0.f, m1 = training ? (float)(q1 / N) : 0.f;
        const floatx4 v = {k * a, -k * m1 * rstd, k * (m1 * rstd * mean - m0), 1.f - a};
        reinterpret_cast<floatx4*>(cst)[c] = v;
        if (blockIdx.x == 0) {
            reinterpret_cast<floatx4*>(cst_out)[c] = v;
            d_gamma[c] = (float)q1;
            d_beta[c] = (float)q0;
            d_alpha[c] = (float)q2 * a * (1.f - a);
        }
    }
    __syncthreads();
    const int W4 = W >> 2;          // host: W % 4 == 0
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x0 = (int)(e % W4) * 4;
        size_t r = e / W4;
        const int y = (int)(r % H);
        r /= H;
        const int c = (int)(r % C);
        const floatx4 k = reinterpret_cast<const floatx4*>(cst)[c];
        const size_t plane = r * (size_t)H * W;
        float wv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wv[i] = w[c * 9 + 8 - i];       // transposed stencil: flipped taps
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        floatx4 gc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            if (yy < 0 || yy >= H) continue;
            const size_t o = plane + (size_t)yy * W + x0;
            const floatx4 gm = *reinterpret_cast<const floatx4*>(g + o), tm = *reinterpret_cast<const floatx4*>(t + o);
            if (dy == 1) gc = gm;
            float v[6];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[1 + i] = fmaf(k[0], gm[i], fmaf(k[1], tm[i], k[2]));
            v[0] = x0 > 0 ? fmaf(k[0], g[o - 1], fmaf(k[1], t[o - 1], k[2])) : 0.f;
            v[5] = x0 + 4 < W ? fmaf(k[0], g[o + 4], fmaf(k[1], t[o + 4], k[2])) : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int dxx = 0; dxx < 3; ++dxx) acc[q] += wv[dy * 3 + dxx] * v[q + dxx];
        }
        floatx4 o4;
#pragma unroll
        for (int q = 0; q < 4; ++q) o4[q] = fmaf(k[3], gc[q], acc[q]);
        *reinterpret_cast<floatx4*>(dx + plane + (size_t)y * W + x0) = o4;
    }
}

// the weight gradient of the same stage: dw[c][tap] partial = sum dt[p] x[p + tap - centre] with dt = A g + B t + D formed on the fly
__global__ __launch_bounds__(256) void dwconv3x3_bn_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                      const float* __restrict__ t, const float* __restrict__ cst,
                                                                      float* __restrict__ dw_part, int C, int H, int W) {
    __shared__ float red[4][9];
    const int c = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    const int rows = (H + WSPLIT - 1) / WSPLIT, y0 = sp * rows, y1 = min(H, y0 + rows);
    const size_t plane = ((size_t)b * C + c) * H * W;
    const float* xp = x + plane;
    const floatx4 k = reinterpret_cast<const floatx4*>(cst)[c];
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    const int W4 = W >> 2, nstrip = max(0, y1 - y0) * W4;
    for (int sidx = threadIdx.x; sidx < nstrip; sidx += 256) {
        const int y = y0 + sidx / W4, x0 = (sidx % W4) * 4;
        const floatx4 gv = *reinterpret_cast<const floatx4*>(g + plane + (size_t)y * W + x0), tv = *reinterpret_cast<const floatx4*>(t + plane + (size_t)y * W + x0);
        float g4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g4[q] = fmaf(k[0], gv[q], fmaf(k[1], tv[q], k[2]));
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy) {
            const int yy = y + dyy - 1;
            if (yy < 0 || yy >= H) continue;
            const float* row = xp + (size_t)yy * W;
            const floatx4 m = *reinterpret_cast<const floatx4*>(row + x0);
            const float v[6] = {x0 > 0 ? row[x0 - 1] : 0.f, m[0], m[1], m[2], m[3], x0 + 4 < W ? row[x0 + 4] : 0.f};
#pragma unroll
            for (int dxx = 0; dxx < 3; ++dxx)
                acc[dyy * 3 + dxx] += (g4[0] * v[dxx] + g4[1] * v[dxx + 1]) + (g4[2] * v[dxx + 2] + g4[3] * v[dxx + 3]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float s = kmu::wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 9)
        dw_part[(((size_t)b * WSPLIT + sp) * C + c) * 9 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Both of the above in one pass: a workgroup owns the rows [y0, y1) of one (b, c) plane (the weight kernel's decomposition), forms dt
// once per strip, applies the transposed stencil for dx AND accumulates the 9 weight-gradient taps against x.  The separate
// weight-gradient launch re-read g and t to rebuild dt (31 launches and ~0.3 ms of kernel time per step in the backward's tail);
// here it costs the x rows on top of the data-gradient pass.  Same arithmetic, same order: dx and the partials are bit-identical
// to the two-kernel path.
__global__ __launch_bounds__(256) void dwconv3x3_bn_bwd_all_kernel(const float* __restrict__ g, const float* __restrict__ t,
                                                                   const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ gamma, const float* __restrict__ alpha,
                                                                   const float* __restrict__ stats, const float* __restrict__ part, int S,
                                                                   int training, double N, float* __restrict__ dx,
                                                                   float* __restrict__ d_gamma, float* __restrict__ d_beta,
                                                                   float* __restrict__ d_alpha, float* __restrict__ dw_part, int C, int H,
                                                                   int W) {
    __shared__ float red[4][9];
    const int c = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    // the channel's folded BatchNorm-backward partials (uniform over the workgroup: scalar loads, fixed order, double)
    double q0 = 0.0, q1 = 0.0, q2 = 0.0;
    for (int i = 0; i < S; ++i) {
        const float* p = part + ((size_t)c * S + i) * 3;
        q0 += p[0], q1 += p[1], q2 += p[2];
    }
    floatx4 k;
    {
        const float mean = stats[2 * c], rstd = stats[2 * c + 1], a = 1.f / (1.f + __expf(-alpha[c]));
        const float kk = gamma[c] * rstd;
        const float m0 = training ? (float)(q0 / N) : 0.f, m1 = training ? (float)(q1 / N) : 0.f;
        k = floatx4{kk * a, -kk * m1 * rstd, kk * (m1 * rstd * mean - m0), 1.f - a};
        if (b == 0 && sp == 0 && threadIdx.x == 0) {
            d_gamma[c] = (float)q1;
            d_beta[c] = (float)q0;
            d_alpha[c] = (float)q2 * a * (1.f - a);
        }
    }
    float wv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wv[i] = w[c * 9 + 8 - i];       // transposed stencil: flipped taps
    const int rows = (H + WSPLIT - 1) / WSPLIT, y0 = sp * rows, y1 = min(H, y0 + rows);
    const size_t plane = ((size_t)b * C + c) * H * W;
    const float* xp = x + plane;
    float wacc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wacc[i] = 0.f;
    const int W4 = W >> 2, nstrip = max(0, y1 - y0) * W4;
    for (int sidx = threadIdx.x; sidx < nstrip; sidx += 256) {
        const int y = y0 + sidx / W4, x0 = (sidx % W4) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        floatx4 gc = {0.f, 0.f, 0.f, 0.f};
        float g4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            if (yy < 0 || yy >= H) continue;
            const size_t o = plane + (size_t)yy * W + x0;
            const floatx4 gm = *reinterpret_cast<const floatx4*>(g + o), tm = *reinterpret_cast<const floatx4*>(t + o);
            float v[6];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[1 + i] = fmaf(k[0], gm[i], fmaf(k[1], tm[i], k[2]));
            v[0] = x0 > 0 ? fmaf(k[0], g[o - 1], fmaf(k[1], t[o - 1], k[2])) : 0.f;
            v[5] = x0 + 4 < W ? fmaf(k[0], g[o + 4], fmaf(k[1], t[o + 4], k[2])) : 0.f;
            if (dy == 1) {
                gc = gm;
#pragma unroll
                for (int i = 0; i < 4; ++i) g4[i] = v[1 + i];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int dxx = 0; dxx < 3; ++dxx) acc[q] += wv[dy * 3 + dxx] * v[q + dxx];
        }
        // weight gradient: dt of this strip's 4 pixels (the centre row's, g4) against x on the three rows around it
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy) {
            const int yy = y + dyy - 1;
            if (yy < 0 || yy >= H) continue;
            const float* row = xp + (size_t)yy * W;
            const floatx4 m = *reinterpret_cast<const floatx4*>(row + x0);
            const float v[6] = {x0 > 0 ? row[x0 - 1] : 0.f, m[0], m[1], m[2], m[3], x0 + 4 < W ? row[x0 + 4] : 0.f};
#pragma unroll
            for (int dxx = 0; dxx < 3; ++dxx)
                wacc[dyy * 3 + dxx] += (g4[0] * v[dxx] + g4[1] * v[dxx + 1]) + (g4[2] * v[dxx + 2] + g4[3] * v[dxx + 3]);
        }
        floatx4 o4;
#pragma unroll
        for (int q = 0; q < 4; ++q) o4[q] = fmaf(k[3], gc[q], acc[q]);
        *reinterpret_cast<floatx4*>(dx + plane + (size_t)y * W + x0) = o4;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float s_ = kmu::wave_sum(wacc[i]);
        if (lane == 0) red[wave][i] = s_;
    }
    __syncthreads();
    if (threadIdx.x < 9)
        dw_part[(((size_t)b * WSPLIT + sp) * C + c) * 9 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// DirectionAttention's scaled stencil backward (KM_UNetV3_SH.py:262-263) in one pass: dx = s[b,c] dwconv^T(dy) and the (b, row-split,
// c) partials of the weight-gradient kernel (A[t] = sum dy x_shift_t, G = sum dy) -- the two launches read dy twice and sat back to
// back on the branch's backward chain (the gate gradient d s needs the partials before the chain can go on).  W % 4 == 0.
__global__ __launch_bounds__(256) void dwconv3x3_scaled_bwd_all_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                       const float* __restrict__ w, const float* __restrict__ scale,
                                                                       float* __restrict__ dx, float* __restrict__ dw_part,
                                                                       float* __restrict__ db_part, int C, int H, int W) {
    __shared__ float red[4][10];
    const int c = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    const int rows = (H + WSPLIT - 1) / WSPLIT, y0 = sp * rows, y1 = min(H, y0 + rows);
    const size_t plane = ((size_t)b * C + c) * H * W;
    const float* xp = x + plane;
    const float* gp = dy + plane;
    const float sc = scale[b * C + c];
    float wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = w[c * 9 + 8 - k];
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    const int W4 = W >> 2, nstrip = max(0, y1 - y0) * W4;
    for (int sidx = threadIdx.x; sidx < nstrip; sidx += 256) {
        const int y = y0 + sidx / W4, x0 = (sidx % W4) * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f}, g4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy) {
            const int yy = y + dyy - 1;
            if (yy < 0 || yy >= H) continue;
            const float* row = gp + (size_t)yy * W;
            const floatx4 m = *reinterpret_cast<const floatx4*>(row + x0);
            const float v[6] = {x0 > 0 ? row[x0 - 1] : 0.f, m[0], m[1], m[2], m[3], x0 + 4 < W ? row[x0 + 4] : 0.f};
            if (dyy == 1) g4[0] = m[0], g4[1] = m[1], g4[2] = m[2], g4[3] = m[3];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int dxx = 0; dxx < 3; ++dxx) o[q] += wv[dyy * 3 + dxx] * v[q + dxx];
        }
        *reinterpret_cast<floatx4*>(dx + plane + (size_t)y * W + x0) = floatx4{o[0] * sc, o[1] * sc, o[2] * sc, o[3] * sc};
        acc[9] += (g4[0] + g4[1]) + (g4[2] + g4[3]);
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy) {
            const int yy = y + dyy - 1;
            if (yy < 0 || yy >= H) continue;
            const float* row = xp + (size_t)yy * W;
            const floatx4 m = *reinterpret_cast<const floatx4*>(row + x0);
            const float v[6] = {x0 > 0 ? row[x0 - 1] : 0.f, m[0], m[1], m[2], m[3], x0 + 4 < W ? row[x0 + 4] : 0.f};
#pragma unroll
            for (int dxx = 0; dxx < 3; ++dxx)
                acc[dyy * 3 + dxx] += (g4[0] * v[dxx] + g4[1] * v[dxx + 1]) + (g4[2] * v[dxx + 2] + g4[3] * v[dxx + 3]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float s_ = kmu::wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s_;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float s_ = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const size_t prow = (size_t)b * WSPLIT + sp;
        if (threadIdx.x < 9) dw_part[(prow * C + c) * 9 + threadIdx.x] = s_;
        else db_part[prow * C + c] = s_;
    }
}

// ---- DirectionAttention's local gate folded into its stencil (KM_UNetV3_SH.py:258-263):
//        attn = sigmoid(q k) v     (q, k, v = the three C-channel chunks of the packed qkv tensor [B,3C,H,W])
//        out  = s[b,c] (dwconv3x3(attn) + bias)
// attn is formed on the fly at the stencil's 18 taps -- the [B,C,H,W] attn tensor is neither written nor re-read (forward: 4 passes
// instead of 6; backward: 7 instead of 10), and two launches per DirectionAttention and direction leave the branch chains.
__device__ __forceinline__ float attn_of(float q, float k, float v) { return v / (1.f + __expf(-q * k)); }     // as qkv_gate_fwd_kernel

// A workgroup owns the rows [y0, y1) of one (b, c) plane (the weight-gradient kernels' decomposition) and first builds the attn tile
// of those rows + one halo row either side in LDS -- every attn value is evaluated ONCE (one exp + one division), not once per tap
// (the first version recomputed it at all 18 taps of a strip and was no faster than the two kernels it replaced).
// Tile layout: LA[(row + 1) * S + 4 + x], S = W + 8: four zero columns either side so that a strip's x0 - 1 and x0 + 4 need no test.
__device__ __forceinline__ void stage_attn(float* LA, const float* __restrict__ qp, size_t chw, int y0, int y1, int H, int W) {
    const int S = W + 8, W4 = W >> 2, nrow = y1 - y0 + 2;
    for (int e = threadIdx.x; e < nrow * 2; e += 256) {          // the pad columns
        float* p = LA + (e >> 1) * S + ((e & 1) ? 4 + W : 0);
        *reinterpret_cast<floatx4*>(p) = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    for (int e = threadIdx.x; e < nrow * W4; e += 256) {
        const int rr = e / W4, x0 = (e - rr * W4) * 4, yy = y0 - 1 + rr;
        floatx4 a = {0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < H) {
            const float* row = qp + (size_t)yy * W + x0;
            const floatx4 q = *reinterpret_cast<const floatx4*>(row), k = *reinterpret_cast<const floatx4*>(row + chw),
                          v = *reinterpret_cast<const floatx4*>(row + 2 * chw);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = attn_of(q[i], k[i], v[i]);
        }
        *reinterpret_cast<floatx4*>(LA + rr * S + 4 + x0) = a;
    }
}

__global__ __launch_bounds__(256) void qkv_dw_scaled_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const float* __restrict__ scale,
                                                                float* __restrict__ out, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float LA[];
    const int c = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    const int rows = (H + WSPLIT - 1) / WSPLIT, y0 = sp * rows, y1 = min(H, y0 + rows);
    if (y1 <= y0) return;
    const size_t hw = (size_t)H * W, chw = (size_t)C * hw;
    stage_attn(LA, qkv + (size_t)b * 3 * chw + (size_t)c * hw, chw, y0, y1, H, W);
    float wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = w[c * 9 + k];
    const float b0 = bias ? bias[c] : 0.f, sc = scale[b * C + c];
    float* op = out + ((size_t)b * C + c) * hw;
    __syncthreads();
    const int S = W + 8, W4 = W >> 2, nstrip = (y1 - y0) * W4;
    for (int sidx = threadIdx.x; sidx < nstrip; sidx += 256) {
        const int ry = sidx / W4, x0 = (sidx - ry * W4) * 4;
        float acc[4] = {b0, b0, b0, b0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {                       // tile row ry + dy = image row y - 1 + dy (zero rows outside the image)
            const float* row = LA + (ry + dy) * S + 4 + x0;
            const floatx4 m = *reinterpret_cast<const floatx4*>(row);
            const float v[6] = {row[-1], m[0], m[1], m[2], m[3], row[4]};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[q] += wv[dy * 3 + dx] * v[q + dx];
        }
        *reinterpret_cast<floatx4*>(op + (size_t)(y0 + ry) * W + x0) = floatx4{acc[0] * sc, acc[1] * sc, acc[2] * sc, acc[3] * sc};
    }
}

// backward of the above in one pass: d attn = s dwconv^T(dy) at the strip, chained through the gate into d q / d k / d v; the
// (b, row-split, c) weight / bias partials of the scaled stencil (A[t] = sum dy attn_shift_t, G = sum dy: kmu_dwconv3x3_scaled_finish
// turns them into d weight, d bias, d s) against the attn tile in LDS
__global__ __launch_bounds__(256) void qkv_dw_scaled_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ qkv,
                                                                const float* __restrict__ w, const float* __restrict__ scale,
                                                                float* __restrict__ dqkv, float* __restrict__ dw_part,
                                                                float* __restrict__ db_part, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float LA[];
    __shared__ float red[4][10];
    const int c = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    const int rows = (H + WSPLIT - 1) / WSPLIT, y0 = sp * rows, y1 = min(H, y0 + rows);
    const size_t hw = (size_t)H * W, chw = (size_t)C * hw;
    const float* gp = dy + ((size_t)b * C + c) * hw;
    const float* qp = qkv + (size_t)b * 3 * chw + (size_t)c * hw;
    float* dqp = dqkv + (size_t)b * 3 * chw + (size_t)c * hw;
    if (y1 > y0) stage_attn(LA, qp, chw, y0, y1, H, W);
    const float sc = scale[b * C + c];
    float wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = w[c * 9 + 8 - k];
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    __syncthreads();
    const int S = W + 8, W4 = W >> 2, nstrip = max(0, y1 - y0) * W4;
    for (int sidx = threadIdx.x; sidx < nstrip; sidx += 256) {
        const int ry = sidx / W4, y = y0 + ry, x0 = (sidx - ry * W4) * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f}, g4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy) {
            const int yy = y + dyy - 1;
            if (yy < 0 || yy >= H) continue;
            const float* row = gp + (size_t)yy * W;
            const floatx4 m = *reinterpret_cast<const floatx4*>(row + x0);
            const float v[6] = {x0 > 0 ? row[x0 - 1] : 0.f, m[0], m[1], m[2], m[3], x0 + 4 < W ? row[x0 + 4] : 0.f};
            if (dyy == 1) g4[0] = m[0], g4[1] = m[1], g4[2] = m[2], g4[3] = m[3];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int dxx = 0; dxx < 3; ++dxx) o[q] += wv[dyy * 3 + dxx] * v[q + dxx];
        }
        {   // chain rule through attn = sigmoid(q k) v at the strip's 4 pixels (as qkv_gate_bwd_kernel)
            const size_t e = (size_t)y * W + x0;
            const floatx4 q = *reinterpret_cast<const floatx4*>(qp + e), k = *reinterpret_cast<const floatx4*>(qp + chw + e),
                          v = *reinterpret_cast<const floatx4*>(qp + 2 * chw + e);
            floatx4 dq, dk, dv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float go = o[i] * sc;
                const float s_ = 1.f / (1.f + __expf(-q[i] * k[i]));
                const float ds = go * v[i] * s_ * (1.f - s_);
                dq[i] = ds * k[i];
                dk[i] = ds * q[i];
                dv[i] = go * s_;
            }
            *reinterpret_cast<floatx4*>(dqp + e) = dq;
            *reinterpret_cast<floatx4*>(dqp + chw + e) = dk;
            *reinterpret_cast<floatx4*>(dqp + 2 * chw + e) = dv;
        }
        acc[9] += (g4[0] + g4[1]) + (g4[2] + g4[3]);
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy) {
            const float* row = LA + (ry + dyy) * S + 4 + x0;
            const floatx4 m = *reinterpret_cast<const floatx4*>(row);
            const float v[6] = {row[-1], m[0], m[1], m[2], m[3], row[4]};
#pragma unroll
            for (int dxx = 0; dxx < 3; ++dxx)
                acc[dyy * 3 + dxx] += (g4[0] * v[dxx] + g4[1] * v[dxx + 1]) + (g4[2] * v[dxx + 2] + g4[3] * v[dxx + 3]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float s_ = kmu::wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s_;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float s_ = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const size_t prow = (size_t)b * WSPLIT + sp;
        if (threadIdx.x < 9) dw_part[(prow * C + c) * 9 + threadIdx.x] = s_;
        else db_part[prow * C + c] = s_;
    }
}

inline size_t qkv_dw_lds(int H, int W) { return (size_t)((H + WSPLIT - 1) / WSPLIT + 2) * (W + 8) * sizeof(float); }

int launch_stencil(const float* in, const float* w, const float* bias, const float* scale, float* out, int B, int C, int H, int W,
                   int flip, hipStream_t st, const char* what, const float* addend = nullptr) {
    const size_t total = (size_t)B * C * H * ((W + 3) / 4);
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dwconv3x3_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, in, w, bias, scale, addend, out, C, H, W, flip,
                       total, (float*)nullptr, 0);
    return kmu::launch_status(what);
}

}  // namespace

extern "C" int kmu_dwconv3x3_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int C, int H,
                                 int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && weight && y, "dwconv3x3_fwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dwconv3x3_fwd: bad dims");
    return launch_stencil(x, weight, bias, nullptr, y, B, C, H, W, 0, (hipStream_t)stream, "dwconv3x3_fwd");
}

// forward + the BatchNorm statistics partials of y: returns the number of partial pairs per channel S (stat_part [C][S][2], for
// kmu_bn_blend_fwd_pre), or 0 when the shape does not allow it (then nothing was launched: use kmu_dwconv3x3_fwd)
extern "C" int kmu_dwconv3x3_stats_partials(int B, int C, int H, int W) {
    const long hw = (long)H * W;
    return (W % 4 == 0 && hw % 1024 == 0 && (long)B * C * (hw / 1024) <= 16384) ? (int)(B * (hw / 1024)) : 0;
}
extern "C" int kmu_dwconv3x3_fwd_stats(const float* x, const float* weight, const float* bias, float* y, float* stat_part, int B, int C,
                                       int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && weight && y && stat_part, "dwconv3x3_fwd_stats: null pointer");
    const int S = kmu_dwconv3x3_stats_partials(B, C, H, W);
    KMU_REQUIRE(B > 0 && C > 0 && S > 0, "dwconv3x3_fwd_stats: [%d,%d,%d,%d] is not covered (H*W %% 1024, W %% 4)", B, C, H, W);
    const size_t total = (size_t)B * C * H * (W / 4);
    hipLaunchKernelGGL(dwconv3x3_kernel<true>, dim3((unsigned)(total / 256)), dim3(256), 0, (hipStream_t)stream, x, weight, bias,
                       (const float*)nullptr, (const float*)nullptr, y, C, H, W, 0, total, stat_part, S);
    return kmu::launch_status("dwconv3x3_fwd_stats");
}

extern "C" int kmu_dwconv3x3_bwd_data(const float* dy, const float* weight, float* dx, int B, int C, int H, int W,
                                      kmu_stream_t stream) {
    KMU_REQUIRE(dy && weight && dx, "dwconv3x3_bwd_data: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dwconv3x3_bwd_data: bad dims");
    return launch_stencil(dy, weight, nullptr, nullptr, dx, B, C, H, W, 1, (hipStream_t)stream, "dwconv3x3_bwd_data");
}

extern "C" int kmu_dwconv3x3_partials(int B) { return B * WSPLIT; }

extern "C" int kmu_dwconv3x3_bwd_weight(const float* x, const float* dy, float* d_weight_partial,
                                        float* d_bias_partial, int B, int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && dy && d_weight_partial, "dwconv3x3_bwd_weight: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0, "dwconv3x3_bwd_weight: bad dims");
    hipLaunchKernelGGL(dwconv3x3_bwd_weight_kernel, dim3(C, B, WSPLIT), dim3(256), 0, (hipStream_t)stream, x, dy,
                       d_weight_partial, d_bias_partial, C, H, W);
    return kmu::launch_status("dwconv3x3_bwd_weight");
}

extern "C" int kmu_dwconv3x3_scaled_fwd(const float* x, const float* weight, const float* bias, const float* scale, float* y, int B,
                                        int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && weight && scale && y, "dwconv3x3_scaled_fwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dwconv3x3_scaled_fwd: bad dims");
    return launch_stencil(x, weight, bias, scale, y, B, C, H, W, 0, (hipStream_t)stream, "dwconv3x3_scaled_fwd");
}

extern "C" int kmu_dwconv3x3_scaled_bwd_data(const float* dy, const float* weight, const float* scale, float* dx, int B, int C, int H,
                                             int W, kmu_stream_t stream) {
    KMU_REQUIRE(dy && weight && scale && dx, "dwconv3x3_scaled_bwd_data: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dwconv3x3_scaled_bwd_data: bad dims");
    return launch_stencil(dy, weight, nullptr, scale, dx, B, C, H, W, 1, (hipStream_t)stream, "dwconv3x3_scaled_bwd_data");
}

// kmu_dwconv3x3_scaled_bwd_data + kmu_dwconv3x3_bwd_weight in one launch (W % 4 == 0); partials as the weight entry point leaves them
extern "C" int kmu_dwconv3x3_scaled_bwd_all(const float* dy, const float* x, const float* weight, const float* scale, float* dx,
                                            float* d_weight_partial, float* d_bias_partial, int B, int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(dy && x && weight && scale && dx && d_weight_partial && d_bias_partial, "dwconv3x3_scaled_bwd_all: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && W % 4 == 0, "dwconv3x3_scaled_bwd_all: bad dims (W %% 4 == 0 required)");
    hipLaunchKernelGGL(dwconv3x3_scaled_bwd_all_kernel, dim3(C, B, WSPLIT), dim3(256), 0, (hipStream_t)stream, dy, x, weight, scale, dx,
                       d_weight_partial, d_bias_partial, C, H, W);
    return kmu::launch_status("dwconv3x3_scaled_bwd_all");
}

// DirectionAttention's gate + stencil as one operator (see qkv_dw_scaled_fwd_kernel); qkv [B,3C,H,W], W % 4 == 0
extern "C" int kmu_qkv_dw_scaled_supported(int B, int C, int H, int W) {
    return B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && W % 4 == 0 && qkv_dw_lds(H, W) <= 64 * 1024;
}

extern "C" int kmu_qkv_dw_scaled_fwd(const float* qkv, const float* weight, const float* bias, const float* scale, float* out, int B, int C,
                                     int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(qkv && weight && scale && out, "qkv_dw_scaled_fwd: null pointer");
    KMU_REQUIRE(kmu_qkv_dw_scaled_supported(B, C, H, W), "qkv_dw_scaled_fwd: unsupported dims B=%d C=%d %dx%d (W %% 4 == 0, attn tile <= 64 KB)", B,
                C, H, W);
    hipLaunchKernelGGL(qkv_dw_scaled_fwd_kernel, dim3(C, B, WSPLIT), dim3(256), qkv_dw_lds(H, W), (hipStream_t)stream, qkv, weight, bias, scale,
                       out, C, H, W);
    return kmu::launch_status("qkv_dw_scaled_fwd");
}

extern "C" int kmu_qkv_dw_scaled_bwd(const float* dy, const float* qkv, const float* weight, const float* scale, float* dqkv,
                                     float* d_weight_partial, float* d_bias_partial, int B, int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(dy && qkv && weight && scale && dqkv && d_weight_partial && d_bias_partial, "qkv_dw_scaled_bwd: null pointer");
    KMU_REQUIRE(kmu_qkv_dw_scaled_supported(B, C, H, W), "qkv_dw_scaled_bwd: unsupported dims B=%d C=%d %dx%d (W %% 4 == 0, attn tile <= 64 KB)", B,
                C, H, W);
    hipLaunchKernelGGL(qkv_dw_scaled_bwd_kernel, dim3(C, B, WSPLIT), dim3(256), qkv_dw_lds(H, W), (hipStream_t)stream, dy, qkv, weight, scale,
                       dqkv, d_weight_partial, d_bias_partial, C, H, W);
    return kmu::launch_status("qkv_dw_scaled_bwd");
}

extern "C" int kmu_dwconv3x3_scaled_finish(const float* d_weight_partial, const float* d_bias_partial, const float* scale,
                                           const float* weight, const float* bias, float* d_weight, float* d_bias, float* d_scale,
                                           int B, int C, kmu_stream_t stream) {
    KMU_REQUIRE(d_weight_partial && d_bias_partial && scale && weight && d_weight && d_scale, "dwconv3x3_scaled_finish: null pointer");
    KMU_REQUIRE(B > 0 && C > 0, "dwconv3x3_scaled_finish: bad dims");
    KMU_REQUIRE((bias == nullptr) == (d_bias == nullptr), "dwconv3x3_scaled_finish: bias and d_bias must both be given or both be NULL");
    const int threads = C * 10 + B * C;
    hipLaunchKernelGGL(dwconv3x3_scaled_finish_kernel, dim3(kmu::cdiv(threads, 256)), dim3(256), 0, (hipStream_t)stream, d_weight_partial,
                       d_bias_partial, scale, weight, bias, d_weight, d_bias, d_scale, B, C);
    return kmu::launch_status("dwconv3x3_scaled_finish");
}

extern "C" int kmu_dwconv3x3_bwd_data_add(const float* dy, const float* weight, const float* addend, float* dx, int B, int C, int H, int W,
                                          kmu_stream_t stream) {
    KMU_REQUIRE(dy && weight && addend && dx, "dwconv3x3_bwd_data_add: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dwconv3x3_bwd_data_add: bad dims");
    return launch_stencil(dy, weight, nullptr, nullptr, dx, B, C, H, W, 1, (hipStream_t)stream, "dwconv3x3_bwd_data_add", addend);
}

// EfficientViMBlock's dwconv stage backward behind kmu_bn_blend_bwd_partials (csrc/bn_blend.hip): dx = dwconv^T(dt) + (1 - a) g with
// dt = BatchNorm2d's input gradient formed on the fly from (g, t) and the folded partials (efficient_vim_init.py:85,93;
// vim_utils_init.py:62-89).  part [C][S][3] and S as kmu_bn_blend_bwd_partials left them; stats [C][2] = (mean, rstd) of the forward;
// cst [C][4]: written here, read by kmu_dwconv3x3_bn_bwd_weight (d weight partials [B * 4][C][9], kmu_dwconv3x3_partials rows).
extern "C" int kmu_dwconv3x3_bn_bwd_data(const float* g, const float* t, const float* weight, const float* gamma, const float* alpha,
                                         const float* stats, const float* part, int S, int training, float* dx, float* d_gamma,
                                         float* d_beta, float* d_alpha, float* cst, int B, int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(g && t && weight && gamma && alpha && stats && part && dx && d_gamma && d_beta && d_alpha && cst, "dwconv3x3_bn_bwd_data: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && C <= 4096 && H > 0 && W > 0 && W % 4 == 0 && S > 0, "dwconv3x3_bn_bwd_data: bad dims (W %% 4 == 0 required)");
    const size_t total = (size_t)B * C * H * (W / 4);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dwconv3x3_bn_bwd_kernel, dim3((unsigned)blocks), dim3(256), (size_t)C * 16, (hipStream_t)stream, g, t, weight, gamma, alpha,
                       stats, part, S, training, (double)B * H * W, dx, d_gamma, d_beta, d_alpha, cst, C, H, W, total);
    return kmu::launch_status("dwconv3x3_bn_bwd_data");
}

// kmu_dwconv3x3_bn_bwd_data and kmu_dwconv3x3_bn_bwd_weight in ONE launch (x: the stage's input; d_weight_partial as the weight
// entry point leaves it): the weight-gradient job behind it shrinks to the column sum of the partials
extern "C" int kmu_dwconv3x3_bn_bwd_all(const float* g, const float* t, const float* x, const float* weight, const float* gamma,
                                        const float* alpha, const float* stats, const float* part, int S, int training, float* dx,
                                        float* d_gamma, float* d_beta, float* d_alpha, float* d_weight_partial, int B, int C, int H, int W,
                                        kmu_stream_t stream) {
    KMU_REQUIRE(g && t && x && weight && gamma && alpha && stats && part && dx && d_gamma && d_beta && d_alpha && d_weight_partial,
                "dwconv3x3_bn_bwd_all: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && W % 4 == 0 && S > 0, "dwconv3x3_bn_bwd_all: bad dims (W %% 4 == 0 required)");
    hipLaunchKernelGGL(dwconv3x3_bn_bwd_all_kernel, dim3(C, B, WSPLIT), dim3(256), 0, (hipStream_t)stream, g, t, x, weight, gamma, alpha, stats,
                       part, S, training, (double)B * H * W, dx, d_gamma, d_beta, d_alpha, d_weight_partial, C, H, W);
    return kmu::launch_status("dwconv3x3_bn_bwd_all");
}

extern "C" int kmu_dwconv3x3_bn_bwd_weight(const float* x, const float* g, const float* t, const float* cst, float* d_weight_partial, int B,
                                           int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && g && t && cst && d_weight_partial, "dwconv3x3_bn_bwd_weight: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && W % 4 == 0, "dwconv3x3_bn_bwd_weight: bad dims (W %% 4 == 0 required)");
    hipLaunchKernelGGL(dwconv3x3_bn_bwd_weight_kernel, dim3(C, B, WSPLIT), dim3(256), 0, (hipStream_t)stream, x, g, t, cst, d_weight_partial, C, H, W);
    return kmu::launch_status("dwconv3x3_bn_bwd_weight");
}
