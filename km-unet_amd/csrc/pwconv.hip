// Pointwise (1x1) convolution over NCHW fp32 -- the FFNs (KM_UNetV3_SH.py:120-124, vim_block_init/
// efficient_vim_init.py FFN = two ConvLayer2D 1x1, vim_utils_init.py:62-89), the qkv projection (:221), the
// 'channel' direction projection (:174) and StableHybridKANConv.residual (:59).
//
//   forward      y[b,co,p]  = sum_ci W[co,ci] * act(x[b,ci,p]) + bias[co]            act = identity | exact GELU
//   bwd_input    dx[b,ci,p] = (sum_co W[co,ci] * gy[b,co,p]) * act'(x[b,ci,p])
//   bwd_weight   dW[co,ci]  = sum_{b,p} gy[b,co,p] * act(x[b,ci,p]),   dbias[co] = sum_{b,p} gy[b,co,p]
//
// With 16..256 channels these are HBM-bound (Ci*Co/(2(Ci+Co)) = 6..26 FLOP/B against a ridge of 19.7): the job
// is to read x / gy once with wide loads and write y once, with bias, GELU, GELU' and the bias gradient folded in
// (stock path: bmm + bias add + gelu, and for backward 2 bmm + a batch sum + a bias reduction, 11 launches).
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, exact fp32).  Both the 16 "M" rows and the 4 "K" slots of an MFMA are
// free permutations as long as A, B and D agree, which lets every lane issue 16-byte loads straight from NCHW:
//   fwd / bwd_input : M = pixels, N = out channels, K = in channels.  Lane (m = l%16, q = l/16) loads ONE float4
//     x[ch(c,s,q)][p0 + 4 f(m) .. +3], ch = 16c + 4q + s, f(m) = m/4 + 4 (m%4): its 4 components are the A operands
//     of 4 M-tiles t (pixel(m,t) = p0 + 4 f(m) + t).  D row 4q + i is then pixel p0 + 16i + 4q + t: for each i the 4
//     tiles side by side give every lane one float4 and the 4 lanes q of an output channel 64 contiguous bytes per
//     store instruction (f = identity would scatter 16-byte pieces at a 64-byte stride).
//   bwd_weight      : M = out channels, N = in channels, K = pixels.  Lane (r = l%16, q) loads 2 float4 = pixels
//     p0 + 8q .. 8q+7 of row r of each gy / x tile; K-step s pairs component s of both (pixel p0 + 8q + s).
// Weights are staged once per workgroup in LDS as wl[k][n] (row stride = 16 mod 32 words: conflict-free
// ds_read_b32 B fragments).  bwd_weight streams straight from HBM into MFMA operands; LDS only adds the 4 waves'
// accumulators, each workgroup dumps one slab and pw_wgrad_reduce_kernel adds the slabs in a fixed order
// (deterministic, no atomics).
#include "common.h"

using kmu::floatx4;

namespace {

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
    return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.39894228040143268f * __expf(-0.5f * v * v);
}

constexpr int KT_MAX = 256;      // contraction channels per LDS weight tile

__host__ __device__ inline int lds_stride(int nt) { return (16 * nt) % 32 == 16 ? 16 * nt : 16 * nt + 16; }

// ---------------------------------------------------------------------------------------- fwd / bwd_input
// grid.x = B * ceil(P / 256), grid.y = N / (16 NT); 4 waves x 64 pixels.  w(n, k) = w[n * w_sn + k * w_sk].
template <int NT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const float* __restrict__ x, const float* __restrict__ w, long w_sn,
                                                      long w_sk, const float* __restrict__ bias,
                                                      const float* __restrict__ mul_pre, const float* __restrict__ addend,
                                                      float* __restrict__ y, int K, int N, int P, int act_in, int G, long w_sg,
                                                      const float* __restrict__ bscale, float* __restrict__ stat_part, long bias_sb,
                                                      float bias_mul) {
    // bias_sb: per-sample stride of `bias` (0 = one vector for the whole batch), bias_mul its factor: the input gradient of a conv
    // whose input ALSO feeds a spatial mean adds d_mean[b][n] / HW to every pixel here instead of in an ATen broadcast add
    // G > 1: grouped convolution (block-diagonal weights): x has G*K channels, y has N = G*Ng channels, tile n0 belongs to group
    // n0 / Ng and contracts that group's K input channels with w + g*w_sg indexed by the LOCAL output channel
    extern __shared__ float wl[];
    constexpr int S = (16 * NT) % 32 == 16 ? 16 * NT : 16 * NT + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int bpp = (P + 255) / 256;
    const int b = blockIdx.x / bpp, p0 = (blockIdx.x % bpp) * 256 + wave * 64;
    const int n0 = blockIdx.y * 16 * NT;
    const int Ng = N / G, g = n0 / Ng, nl0 = n0 - g * Ng;
    w += (long)g * w_sg;

    // this wave's first x chunk is requested BEFORE the weights are staged: the two round trips (x from HBM, W from L2 + the
    // barrier) then overlap instead of queueing behind each other in every (short-lived) workgroup
    const float* xb = x + ((size_t)b * G + g) * K * P + p0 + 4 * ((m >> 2) + 4 * (m & 3));
    const bool live = p0 < P;           // a wave past the end of the sample still takes part in the barriers below
    floatx4 xv[4], xn[4];
    if (live) {
#pragma unroll
        for (int s = 0; s < 4; ++s) xv[s] = *reinterpret_cast<const floatx4*>(xb + (size_t)(4 * q + s) * P);
    }
    floatx4 acc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[nt][t] = floatx4{0.f, 0.f, 0.f, 0.f};

    // the contraction runs in tiles of <= KT_MAX channels: the weight tile W[n0 .. n0+16NT) x [k0, k0 + kt) sits in LDS as wl[k][n]
    // (K > KT_MAX: DAGEM's deformable-conv contraction over Cin * 9 = 576 sampled columns, DAGEM_md.py:98-101)
    const int nchunk = K / 16;
    for (int k0 = 0; k0 < K; k0 += KT_MAX) {
        const int kt = min(KT_MAX, K - k0);
        if (k0) __syncthreads();        // every wave is done with the previous weight tile
        if (w_sk == 1) {
            for (int e = tid; e < 16 * NT * kt; e += 256) {
                const int n = e / kt, k = e - n * kt;
                wl[k * S + n] = w[(long)(nl0 + n) * w_sn + k0 + k];
            }
        } else {
            for (int e = tid; e < 16 * NT * kt; e += 256) {
                const int k = e / (16 * NT), n = e - k * 16 * NT;
                wl[k * S + n] = w[(long)(nl0 + n) * w_sn + (long)(k0 + k) * w_sk];
            }
        }
        __syncthreads();
        if (!live) continue;
        for (int c = k0 / 16; c < (k0 + kt) / 16; ++c) {
            if (c + 1 < nchunk) {
#pragma unroll
                for (int s = 0; s < 4; ++s) xn[s] = *reinterpret_cast<const floatx4*>(xb + (size_t)(16 * (c + 1) + 4 * q + s) * P);
            }
            if (act_in) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < 4; ++t) xv[s][t] = gelu_f(xv[s][t]);
            }
            const float* wrow = wl + (16 * c - k0 + 4 * q) * S + m;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float bf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[nt] = wrow[s * S + 16 * nt];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s][t], bf[nt], acc[nt][t], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) xv[s] = xn[s];
        }
    }
    if (!live) {
        if (stat_part) __syncthreads(), __syncthreads();     // the statistics epilogue's two barriers (host: P % 256 == 0 there, so never taken)
        return;
    }

    // epilogue: lane owns channel n0 + 16nt + m, pixels p0 + 16i + 4q + t
    float sta[NT], stq[NT];     // stat_part: (sum, sum of squares) of this lane's 16 outputs per channel tile, for the BatchNorm behind
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + 16 * nt + m;
        const float bv = bias ? bias[(size_t)b * bias_sb + n] * bias_mul : 0.f;
        sta[nt] = stq[nt] = 0.f;
        const float sc = bscale ? bscale[b] : 1.f;       // per-sample factor (DropPath's mask / keep_prob of a residual branch)
        const size_t off = ((size_t)b * N + n) * P + p0 + 4 * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            floatx4 o = {acc[nt][0][i] + bv, acc[nt][1][i] + bv, acc[nt][2][i] + bv, acc[nt][3][i] + bv};
            if (mul_pre) {
                const floatx4 z = *reinterpret_cast<const floatx4*>(mul_pre + off + 16 * i);
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] *= gelu_grad_f(z[t]);
            }
            if (bscale) o *= sc;
            if (addend) o += *reinterpret_cast<const floatx4*>(addend + off + 16 * i);
            sta[nt] += (o[0] + o[1]) + (o[2] + o[3]);
            stq[nt] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            *reinterpret_cast<floatx4*>(y + off + 16 * i) = o;
        }
    }
    if (stat_part) {            // host guarantees P % 256 == 0 (no wave left early): one (sum, sumsq) pair per workgroup and channel
        __syncthreads();        // every wave is done with the weight tile: wl is scratch now
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float a = sta[nt], qq = stq[nt];
            a += __shfl_xor(a, 16);
            a += __shfl_xor(a, 32);
            qq += __shfl_xor(qq, 16);
            qq += __shfl_xor(qq, 32);
            if (q == 0) {
                wl[((wave * NT + nt) * 16 + m) * 2] = a;
                wl[((wave * NT + nt) * 16 + m) * 2 + 1] = qq;
            }
        }
        __syncthreads();
        if (tid < NT * 16) {
            const int nt = tid >> 4, mm = tid & 15;
            float a = 0.f, qq = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                a += wl[((w4 * NT + nt) * 16 + mm) * 2];
                qq += wl[((w4 * NT + nt) * 16 + mm) * 2 + 1];
            }
            float* pp = stat_part + ((size_t)(n0 + 16 * nt + mm) * gridDim.x + blockIdx.x) * 2;
            pp[0] = a;
            pp[1] = qq;
        }
    }
}

// ---------------------------------------------------------------------------------------- bwd_weight
// grid.x = G pixel groups, grid.y = (Co/16MT) * (Ci/16NT) slice pairs; 4 waves, each walks 32-pixel chunks
// (chunk id = (g*4 + wave) + j * 4G, next chunk prefetched into registers), the 4 waves' accumulators are added
// through LDS (fixed order) and the workgroup dumps slab[sp][g][mt][nt][lane] (float4 per lane).
template <int MT, int NT>
__device__ __forceinline__ void wgrad_load(floatx4 (&a)[MT][2], floatx4 (&bx)[NT][2], const float* __restrict__ x,
                                           const float* __restrict__ gy, int ch, int cpp, int XC, int GC, int P, int co0, int ci0,
                                           int r, int q) {
    // XC / GC: channel counts of the TENSORS x / gy (their batch strides; > Ci / Co when one group of a grouped conv is contracted)
    const int b = ch / cpp, p0 = (ch - b * cpp) * 32 + 8 * q;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float* g = gy + ((size_t)b * GC + co0 + 16 * mt + r) * P + p0;
        a[mt][0] = *reinterpret_cast<const floatx4*>(g);
        a[mt][1] = *reinterpret_cast<const floatx4*>(g + 4);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float* xp = x + ((size_t)b * XC + ci0 + 16 * nt + r) * P + p0;
        bx[nt][0] = *reinterpret_cast<const floatx4*>(xp);
        bx[nt][1] = *reinterpret_cast<const floatx4*>(xp + 4);
    }
}

// blockIdx.z selects one of up to WMAX problems of IDENTICAL dimensions (the same layer of several modules -- the three direction
// branches, the two blocks of a level): one launch with z times the workgroups instead of z launches
constexpr int WMAX = 8;
struct WgradMulti {
    const float* x[WMAX];
    const float* gy[WMAX];
    float* slab[WMAX];
    float* bslab[WMAX];
};
template <int MT, int NT>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(WgradMulti pm, int Ci, int Co, int P, int nchunks, int act_in, int XC, int GC) {
    const float* __restrict__ x = pm.x[blockIdx.z];
    const float* __restrict__ gy = pm.gy[blockIdx.z];
    float* __restrict__ slab = pm.slab[blockIdx.z];
    float* __restrict__ bslab = pm.bslab[blockIdx.z];
    __shared__ floatx4 red[3][MT * NT][64];
    __shared__ float bred[3][MT][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nsl_i = Ci / (16 * NT);
    const int sp = blockIdx.y, co0 = (sp / nsl_i) * 16 * MT, ci0 = (sp % nsl_i) * 16 * NT;
    const int nw = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    const int cpp = P / 32;   // chunks per sample

    floatx4 acc[MT][NT];
    float bs[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        bs[mt] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    floatx4 a[MT][2], bx[NT][2], an[MT][2], bn[NT][2];
    if (gw < nchunks) wgrad_load<MT, NT>(a, bx, x, gy, gw, cpp, XC, GC, P, co0, ci0, r, q);
    for (int ch = gw; ch < nchunks; ch += nw) {
        const bool more = ch + nw < nchunks;
        if (more) wgrad_load<MT, NT>(an, bn, x, gy, ch + nw, cpp, XC, GC, P, co0, ci0, r, q);
        if (act_in) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s = 0; s < 8; ++s) bx[nt][s >> 2][s & 3] = gelu_f(bx[nt][s >> 2][s & 3]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < 8; ++s) bs[mt] += a[mt][s >> 2][s & 3];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][s >> 2][s & 3], bx[nt][s >> 2][s & 3], acc[mt][nt], 0, 0, 0);
        if (more) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt][0] = an[mt][0], a[mt][1] = an[mt][1];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bx[nt][0] = bn[nt][0], bx[nt][1] = bn[nt][1];
        }
    }
    const bool want_bias = bslab && sp % nsl_i == 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        bs[mt] += __shfl_xor(bs[mt], 16);
        bs[mt] += __shfl_xor(bs[mt], 32);
    }
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) red[wave - 1][mt * NT + nt][lane] = acc[mt][nt];
            if (want_bias && q == 0) bred[wave - 1][mt][r] = bs[mt];
        }
    }
    __syncthreads();
    if (wave == 0) {
        float* out = slab + (((size_t)sp * gridDim.x + blockIdx.x) * MT * NT) * 256 + lane * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                floatx4 v = acc[mt][nt];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += red[w][mt * NT + nt][lane];
                *reinterpret_cast<floatx4*>(out + (mt * NT + nt) * 256) = v;
            }
            if (want_bias && q == 0)
                bslab[((size_t)(sp / nsl_i) * gridDim.x + blockIdx.x) * MT * 16 + mt * 16 + r] =
                    bs[mt] + bred[0][mt][r] + bred[1][mt][r] + bred[2][mt][r];
        }
    }
}

// 256 threads = 16 accumulator float4s x 16 partitions of the G slabs; partition sums meet in LDS in a fixed order.
// float4 (sp, mt, nt, lane) -> rows co = co0 + 16mt + 4q + i, column ci = ci0 + 16nt + r.  Bias: blocks past the
// weight range, 16 channels x 16 partitions each.
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                              float* __restrict__ dw, float* __restrict__ dbias, int Ci, int Co,
                                                              int MT, int NT, int G, int wblocks) {
    __shared__ floatx4 part[16][16];
    const int o = threadIdx.x & 15, pt = threadIdx.x >> 4;
    const int per_sp = MT * NT * 64, nsl_i = Ci / (16 * NT);
    if ((int)blockIdx.x < wblocks) {
        const int t = blockIdx.x * 16 + o;                 // float4 id; per_sp % 16 == 0 so a block stays inside one tile
        const int sp = t / per_sp, e = t - sp * per_sp, tile = e >> 6, lane = e & 63;
        const float* src = slab + ((size_t)sp * G * MT * NT + tile) * 256 + lane * 4;
        floatx4 s = {0.f, 0.f, 0.f, 0.f};
        for (int g = pt; g < G; g += 16) s += *reinterpret_cast<const floatx4*>(src + (size_t)g * MT * NT * 256);
        part[pt][o] = s;
        __syncthreads();
        if (pt == 0) {
#pragma unroll
            for (int k = 1; k < 16; ++k) s += part[k][o];
            const int mt = tile / NT, nt = tile - mt * NT, r = lane & 15, q = lane >> 4;
            const int co = (sp / nsl_i) * 16 * MT + 16 * mt + 4 * q, ci = (sp % nsl_i) * 16 * NT + 16 * nt + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) dw[(size_t)(co + i) * Ci + ci] = s[i];
        }
    } else {
        const int t = (blockIdx.x - wblocks) * 16 + o;     // output channel (Co % 16 == 0)
        const int so = t / (16 * MT), rem = t - so * 16 * MT;
        const float* src = bslab + (size_t)so * G * MT * 16 + rem;
        float s = 0.f;
        for (int g = pt; g < G; g += 16) s += src[(size_t)g * MT * 16];
        part[pt][o][0] = s;
        __syncthreads();
        if (pt == 0) {
#pragma unroll
            for (int k = 1; k < 16; ++k) s += part[k][o][0];
            dbias[t] = s;
        }
    }
}

// the same reduction for up to 32 weight gradients in ONE launch (the weight-gradient tail of a train step is bound by its
// launch COUNT: ~500 launches of ~5 us on three streams take 1.45 ms)
constexpr int RMAX = 32;
struct ReduceMultiArgs {
    const float* slab[RMAX];
    const float* bslab[RMAX];
    float* dw[RMAX];
    float* dbias[RMAX];
    int Ci[RMAX], Co[RMAX], MT[RMAX], NT[RMAX], G[RMAX], wblocks[RMAX], blk0[RMAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void pw_wgrad_reduce_multi_kernel(ReduceMultiArgs a) {
    __shared__ floatx4 part[16][16];
    int k = 0;
    while (k + 1 < a.n && (int)blockIdx.x >= a.blk0[k + 1]) ++k;
    const int bid = blockIdx.x - a.blk0[k];
    const float* __restrict__ slab = a.slab[k];
    const float* __restrict__ bslab = a.bslab[k];
    float* __restrict__ dw = a.dw[k];
    float* __restrict__ dbias = a.dbias[k];
    const int Ci = a.Ci[k], MT = a.MT[k], NT = a.NT[k], G = a.G[k], wblocks = a.wblocks[k];
    const int o = threadIdx.x & 15, pt = threadIdx.x >> 4;
    const int per_sp = MT * NT * 64, nsl_i = Ci / (16 * NT);
    if (bid < wblocks) {
        const int t = bid * 16 + o;
        const int sp = t / per_sp, e = t - sp * per_sp, tile = e >> 6, lane = e & 63;
        const float* src = slab + ((size_t)sp * G * MT * NT + tile) * 256 + lane * 4;
        floatx4 s = {0.f, 0.f, 0.f, 0.f};
        for (int g = pt; g < G; g += 16) s += *reinterpret_cast<const floatx4*>(src + (size_t)g * MT * NT * 256);
        part[pt][o] = s;
        __syncthreads();
        if (pt == 0) {
#pragma unroll
            for (int j = 1; j < 16; ++j) s += part[j][o];
            const int mt = tile / NT, nt = tile - mt * NT, r = lane & 15, q = lane >> 4;
            const int co = (sp / nsl_i) * 16 * MT + 16 * mt + 4 * q, ci = (sp % nsl_i) * 16 * NT + 16 * nt + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) dw[(size_t)(co + i) * Ci + ci] = s[i];
        }
    } else {
        const int t = (bid - wblocks) * 16 + o;
        const int so = t / (16 * MT), rem = t - so * 16 * MT;
        const float* src = bslab + (size_t)so * G * MT * 16 + rem;
        float s = 0.f;
        for (int g = pt; g < G; g += 16) s += src[(size_t)g * MT * 16];
        part[pt][o][0] = s;
        __syncthreads();
        if (pt == 0) {
#pragma unroll
            for (int j = 1; j < 16; ++j) s += part[j][o][0];
            dbias[t] = s;
        }
    }
}

inline int pick_tiles(int n16) { return n16 % 4 == 0 ? 4 : (n16 % 3 == 0 ? 3 : (n16 % 2 == 0 ? 2 : 1)); }

struct WgradPlan {
    int MT, NT, nsp, G, nw, nchunks;
    size_t slab_floats, bslab_floats;
};

#ifndef KMU_PW_GMAX
#define KMU_PW_GMAX 256   // workgroups (= slabs) per slice pair; measured at B=8, 128x128: 512 -> 25 us, 256 -> 20 us, 128 -> 22 us per wgrad
#endif
inline WgradPlan wgrad_plan(int B, int Ci, int Co, int P) {
    WgradPlan pl;
    pl.MT = pick_tiles(Co / 16);
    pl.NT = pick_tiles(Ci / 16);
    pl.nsp = (Co / (16 * pl.MT)) * (Ci / (16 * pl.NT));
    pl.nchunks = B * (P / 32);
    int G = pl.nchunks / 8;                        // >= 2 chunks per wave
    G = G < 1 ? 1 : (G > KMU_PW_GMAX ? KMU_PW_GMAX : G);
    while (G > 32 && (size_t)G * pl.MT * pl.NT * 1024 * pl.nsp > ((size_t)4 << 20)) G /= 2;   // keep the slabs <= 4 MB
    pl.G = G;
    pl.nw = 4 * G;
    pl.slab_floats = (size_t)pl.nsp * G * pl.MT * pl.NT * 256;
    pl.bslab_floats = (size_t)(Co / (16 * pl.MT)) * G * pl.MT * 16;
    return pl;
}

template <int NT>
int launch_gemm(const float* x, const float* w, long w_sn, long w_sk, const float* bias, const float* mul_pre, const float* addend,
                float* y, int B, int K, int N, int P, int act_in, hipStream_t st, int G, long w_sg, const float* bscale,
                float* stat_part, long bias_sb, float bias_mul) {
    const size_t lds = (size_t)(K < KT_MAX ? K : KT_MAX) * lds_stride(NT) * sizeof(float);
    KMU_MAX_LDS((pw_gemm_kernel<NT>), lds);
    hipLaunchKernelGGL((pw_gemm_kernel<NT>), dim3(B * ((P + 255) / 256), N / (16 * NT)), dim3(256), lds, st, x, w, w_sn, w_sk, bias,
                       mul_pre, addend, y, K, N, P, act_in, G, w_sg, bscale, stat_part, bias_sb, bias_mul);
    return 0;
}

int gemm(const char* what, const float* x, const float* w, long w_sn, long w_sk, const float* bias, const float* mul_pre, float* y,
         int B, int K, int N, int P, int act_in, hipStream_t st, const float* addend = nullptr, int G = 1, long w_sg = 0,
         const float* bscale = nullptr, float* stat_part = nullptr, long bias_sb = 0, float bias_mul = 1.f) {
    // K = contraction channels PER GROUP, N = output channels in total (G groups of N / G)
    KMU_REQUIRE(B > 0 && K > 0 && N > 0 && K % 16 == 0 && N % 16 == 0, "%s: channels (%d -> %d) must be positive multiples of 16",
                what, K, N);
    KMU_REQUIRE(G >= 1 && N % (16 * G) == 0, "%s: %d output channels do not split into %d groups of whole 16-channel tiles", what, N, G);
    KMU_REQUIRE(P > 0 && P % 64 == 0, "%s: H*W = %d must be a positive multiple of 64", what, P);
    // With few pixel blocks (the 32x32 level: B*P/256 = 32) a 4-tile-wide workgroup leaves most of the 256 CUs idle and walks all K
    // channels on its own: 27.6 us for 256 -> 64 at [8, ., 32, 32] (0.38 TB/s).  Narrower channel tiles there give grid.y more
    // workgroups; x is re-read from L2 once per channel tile, which at these sizes (<= 8 MB) is cheaper than an idle device.
    int nt = pick_tiles(N / G / 16);
    const long blocks_x = (long)B * ((P + 255) / 256);
    while (nt > 1 && blocks_x * (N / (16 * nt)) < 192) nt = (nt == 4) ? 2 : 1;
    switch (nt) {
        case 4: launch_gemm<4>(x, w, w_sn, w_sk, bias, mul_pre, addend, y, B, K, N, P, act_in, st, G, w_sg, bscale, stat_part, bias_sb, bias_mul); break;
        case 3: launch_gemm<3>(x, w, w_sn, w_sk, bias, mul_pre, addend, y, B, K, N, P, act_in, st, G, w_sg, bscale, stat_part, bias_sb, bias_mul); break;
        case 2: launch_gemm<2>(x, w, w_sn, w_sk, bias, mul_pre, addend, y, B, K, N, P, act_in, st, G, w_sg, bscale, stat_part, bias_sb, bias_mul); break;
        default: launch_gemm<1>(x, w, w_sn, w_sk, bias, mul_pre, addend, y, B, K, N, P, act_in, st, G, w_sg, bscale, stat_part, bias_sb, bias_mul); break;
    }
    return kmu::launch_status(what);
}

template <int MT, int NT>
void launch_wgrad(const WgradPlan& pl, const WgradMulti& pm, int nprob, int Ci, int Co, int P, int act_in, hipStream_t st, int XC, int GC) {
    hipLaunchKernelGGL((pw_wgrad_kernel<MT, NT>), dim3(pl.G, pl.nsp, nprob), dim3(256), 0, st, pm, Ci, Co, P, pl.nchunks, act_in, XC, GC);
}

template <int MT>
void launch_wgrad_nt(const WgradPlan& pl, const WgradMulti& pm, int nprob, int Ci, int Co, int P, int act_in, hipStream_t st, int XC,
                     int GC) {
    switch (pl.NT) {
        case 4: launch_wgrad<MT, 4>(pl, pm, nprob, Ci, Co, P, act_in, st, XC, GC); break;
        case 3: launch_wgrad<MT, 3>(pl, pm, nprob, Ci, Co, P, act_in, st, XC, GC); break;
        case 2: launch_wgrad<MT, 2>(pl, pm, nprob, Ci, Co, P, act_in, st, XC, GC); break;
        default: launch_wgrad<MT, 1>(pl, pm, nprob, Ci, Co, P, act_in, st, XC, GC); break;
    }
}

}  // namespace

extern "C" int kmu_pwconv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co, int P, int act_in,
                              kmu_stream_t stream) {
    KMU_REQUIRE(x && w && y, "pwconv_fwd: null pointer");
    return gemm("pwconv_fwd", x, w, Ci, 1, bias, nullptr, y, B, Ci, Co, P, act_in, (hipStream_t)stream);
}

extern "C" int kmu_pwconv_bwd_input(const float* gy, const float* w, const float* x_pre, float* dx, int B, int Ci, int Co, int P,
                                    int act_in, kmu_stream_t stream) {
    KMU_REQUIRE(gy && w && dx, "pwconv_bwd_input: null pointer");
    KMU_REQUIRE(!act_in || x_pre, "pwconv_bwd_input: act_in needs the pre-activation input");
    // dx[ci] = sum_co W[co][ci] gy[co]: the same contraction with K = Co, N = Ci and w(n = ci, k = co) = W[k*Ci + n]
    return gemm("pwconv_bwd_input", gy, w, 1, Ci, nullptr, act_in ? x_pre : nullptr, dx, B, Co, Ci, P, 0, (hipStream_t)stream);
}

// dx = W^T gy + row_add[b][ci] * row_mul: the conv's input also fed a spatial mean whose gradient is d_mean = row_add, row_mul = 1 / HW
// (DirectionAttention.forward, KM_UNetV3_SH.py:231 + :258: AdaptiveAvgPool2d(1)(x) and qkv(x) read the same x)
extern "C" int kmu_pwconv_bwd_input_rowadd(const float* gy, const float* w, const float* row_add, float row_mul, float* dx, int B, int Ci,
                                           int Co, int P, kmu_stream_t stream) {
    KMU_REQUIRE(gy && w && row_add && dx, "pwconv_bwd_input_rowadd: null pointer");
    return gemm("pwconv_bwd_input_rowadd", gy, w, 1, Ci, row_add, nullptr, dx, B, Co, Ci, P, 0, (hipStream_t)stream, nullptr, 1, 0, nullptr,
                nullptr, Ci, row_mul);
}

extern "C" int kmu_pwconv_bwd_input_add(const float* gy, const float* w, const float* addend, float* dx, int B, int Ci, int Co, int P,
                                        kmu_stream_t stream) {
    KMU_REQUIRE(gy && w && addend && dx, "pwconv_bwd_input_add: null pointer");
    return gemm("pwconv_bwd_input_add", gy, w, 1, Ci, nullptr, nullptr, dx, B, Co, Ci, P, 0, (hipStream_t)stream, addend);
}

// Residual forms for EnhancedViMBlock's tail, out = x + s[b] * ffn(...) (KM_UNetV3_SH.py:147-150; s = DropPath's per-sample factor or
// NULL): forward y = addend + s[b] (W act(x) + bias); input gradient dx = s[b] (W^T gy) act'(x_pre).
extern "C" int kmu_pwconv_fwd_res(const float* x, const float* w, const float* bias, const float* addend, const float* bscale, float* y,
                                  int B, int Ci, int Co, int P, int act_in, kmu_stream_t stream) {
    KMU_REQUIRE(x && w && addend && y, "pwconv_fwd_res: null pointer");
    return gemm("pwconv_fwd_res", x, w, Ci, 1, bias, nullptr, y, B, Ci, Co, P, act_in, (hipStream_t)stream, addend, 1, 0, bscale);
}
extern "C" int kmu_pwconv_bwd_input_s(const float* gy, const float* w, const float* x_pre, const float* bscale, float* dx, int B, int Ci,
                                      int Co, int P, int act_in, kmu_stream_t stream) {
    KMU_REQUIRE(gy && w && dx, "pwconv_bwd_input_s: null pointer");
    KMU_REQUIRE(!act_in || x_pre, "pwconv_bwd_input_s: act_in needs the pre-activation input");
    return gemm("pwconv_bwd_input_s", gy, w, 1, Ci, nullptr, act_in ? x_pre : nullptr, dx, B, Co, Ci, P, 0, (hipStream_t)stream, nullptr, 1, 0,
                bscale);
}

// forward + the BatchNorm statistics partials of y (EfficientViMBlock's FFN: ConvLayer2D = 1x1 conv + BatchNorm2d,
// vim_utils_init.py:62-89): stat_part [Co][S][2] with S = kmu_pwconv_stats_partials(B, P) workgroups per channel, the layout
// kmu_bn_blend_fwd_pre folds.  0 partials = shape not covered (P % 256 != 0).
extern "C" int kmu_pwconv_stats_partials(int B, int P) { return (P > 0 && P % 256 == 0) ? B * (P / 256) : 0; }
extern "C" int kmu_pwconv_fwd_stats(const float* x, const float* w, const float* bias, float* y, float* stat_part, int B, int Ci, int Co,
                                    int P, int act_in, kmu_stream_t stream) {
    KMU_REQUIRE(x && w && y && stat_part, "pwconv_fwd_stats: null pointer");
    KMU_REQUIRE(kmu_pwconv_stats_partials(B, P) > 0, "pwconv_fwd_stats: H*W = %d must be a multiple of 256", P);
    return gemm("pwconv_fwd_stats", x, w, Ci, 1, bias, nullptr, y, B, Ci, Co, P, act_in, (hipStream_t)stream, nullptr, 1, 0, nullptr,
                stat_part);
}

extern "C" size_t kmu_pwconv_bwd_weight_ws_bytes(int B, int Ci, int Co, int P) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || P <= 0 || Ci % 16 || Co % 16 || P % 32) return 0;
    const WgradPlan pl = wgrad_plan(B, Ci, Co, P);
    return (pl.slab_floats + pl.bslab_floats) * sizeof(float);
}

static int pw_bwd_weight(const char* what, const float* x, const float* gy, float* dw, float* dbias, void* ws, size_t ws_bytes, int B,
                        int Ci, int Co, int P, int act_in, hipStream_t st, int XC, int GC, bool reduce = true) {
    KMU_REQUIRE(x && gy && (dw || !reduce) && ws, "%s: null pointer", what);
    KMU_REQUIRE(B > 0 && Ci > 0 && Co > 0 && Ci % 16 == 0 && Co % 16 == 0, "%s: channels (%d -> %d) must be positive multiples of 16", what,
                Ci, Co);
    KMU_REQUIRE(P > 0 && P % 32 == 0, "%s: H*W = %d must be a positive multiple of 32", what, P);
    const WgradPlan pl = wgrad_plan(B, Ci, Co, P);
    KMU_REQUIRE(ws_bytes >= (pl.slab_floats + pl.bslab_floats) * sizeof(float), "%s: workspace too small (%zu < %zu)", what, ws_bytes,
                (pl.slab_floats + pl.bslab_floats) * sizeof(float));
    float* slab = (float*)ws;
    float* bslab = dbias ? slab + pl.slab_floats : nullptr;
    WgradMulti pm;
    pm.x[0] = x, pm.gy[0] = gy, pm.slab[0] = slab, pm.bslab[0] = bslab;
    switch (pl.MT) {
        case 4: launch_wgrad_nt<4>(pl, pm, 1, Ci, Co, P, act_in, st, XC, GC); break;
        case 3: launch_wgrad_nt<3>(pl, pm, 1, Ci, Co, P, act_in, st, XC, GC); break;
        case 2: launch_wgrad_nt<2>(pl, pm, 1, Ci, Co, P, act_in, st, XC, GC); break;
        default: launch_wgrad_nt<1>(pl, pm, 1, Ci, Co, P, act_in, st, XC, GC); break;
    }
    int rc = kmu::launch_status(what);
    if (rc || !reduce) return rc;
    const int wblocks = pl.nsp * pl.MT * pl.NT * 64 / 16, bblocks = dbias ? Co / 16 : 0;
    hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3(wblocks + bblocks), dim3(256), 0, st, slab, bslab, dw, dbias, Ci, Co, pl.MT, pl.NT,
                       pl.G, wblocks);
    return kmu::launch_status(what);
}

// The two halves of kmu_pwconv_bwd_weight separately: the slab pass of ONE weight gradient (with_bias: also the bias slabs), and the
// slab reduction of up to 32 of them in one launch (ws[i] / dw[i] / dbias[i] / dims of problem i; dbias[i] NULL = no bias).
extern "C" int kmu_pwconv_bwd_weight_partial(const float* x, const float* gy, void* ws, size_t ws_bytes, int with_bias, int B, int Ci,
                                             int Co, int P, int act_in, kmu_stream_t stream) {
    float dummy;
    return pw_bwd_weight("pwconv_bwd_weight_partial", x, gy, nullptr, with_bias ? &dummy : nullptr, ws, ws_bytes, B, Ci, Co, P, act_in,
                         (hipStream_t)stream, Ci, Co, false);
}

// the slab pass of n <= 8 weight gradients of IDENTICAL dimensions in one launch (problem i: x[i], gy[i] -> ws[i])
extern "C" int kmu_pwconv_bwd_weight_partial_multi(int n, const float* const* x, const float* const* gy, void* const* ws, size_t ws_bytes,
                                                   int with_bias, int B, int Ci, int Co, int P, int act_in, kmu_stream_t stream) {
    KMU_REQUIRE(n > 0 && n <= WMAX && x && gy && ws, "pwconv_bwd_weight_partial_multi: %d problems (1..%d)", n, WMAX);
    KMU_REQUIRE(B > 0 && Ci > 0 && Co > 0 && Ci % 16 == 0 && Co % 16 == 0 && P > 0 && P % 32 == 0,
                "pwconv_bwd_weight_partial_multi: bad dims (%d -> %d, H*W = %d)", Ci, Co, P);
    const WgradPlan pl = wgrad_plan(B, Ci, Co, P);
    KMU_REQUIRE(ws_bytes >= (pl.slab_floats + pl.bslab_floats) * sizeof(float), "pwconv_bwd_weight_partial_multi: workspace too small");
    WgradMulti pm;
    for (int i = 0; i < n; ++i) {
        KMU_REQUIRE(x[i] && gy[i] && ws[i], "pwconv_bwd_weight_partial_multi: problem %d has a null pointer", i);
        pm.x[i] = x[i], pm.gy[i] = gy[i], pm.slab[i] = (float*)ws[i];
        pm.bslab[i] = with_bias ? (float*)ws[i] + pl.slab_floats : nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (pl.MT) {
        case 4: launch_wgrad_nt<4>(pl, pm, n, Ci, Co, P, act_in, st, Ci, Co); break;
        case 3: launch_wgrad_nt<3>(pl, pm, n, Ci, Co, P, act_in, st, Ci, Co); break;
        case 2: launch_wgrad_nt<2>(pl, pm, n, Ci, Co, P, act_in, st, Ci, Co); break;
        default: launch_wgrad_nt<1>(pl, pm, n, Ci, Co, P, act_in, st, Ci, Co); break;
    }
    return kmu::launch_status("pwconv_bwd_weight_partial_multi");
}

extern "C" int kmu_pwconv_bwd_weight_reduce_multi(int n, const void* const* ws, float* const* dw, float* const* dbias, const int* B,
                                                  const int* Ci, const int* Co, const int* P, kmu_stream_t stream) {
    KMU_REQUIRE(n > 0 && n <= RMAX && ws && dw && dbias && B && Ci && Co && P, "pwconv_bwd_weight_reduce_multi: %d problems (1..%d)", n, RMAX);
    ReduceMultiArgs a;
    a.n = n;
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        KMU_REQUIRE(ws[k] && dw[k], "pwconv_bwd_weight_reduce_multi: problem %d has a null pointer", k);
        const WgradPlan pl = wgrad_plan(B[k], Ci[k], Co[k], P[k]);
        a.slab[k] = (const float*)ws[k];
        a.bslab[k] = dbias[k] ? (const float*)ws[k] + pl.slab_floats : nullptr;
        a.dw[k] = dw[k];
        a.dbias[k] = dbias[k];
        a.Ci[k] = Ci[k], a.Co[k] = Co[k], a.MT[k] = pl.MT, a.NT[k] = pl.NT, a.G[k] = pl.G;
        a.wblocks[k] = pl.nsp * pl.MT * pl.NT * 64 / 16;
        a.blk0[k] = blocks;
        blocks += a.wblocks[k] + (dbias[k] ? Co[k] / 16 : 0);
    }
    a.blk0[n] = blocks;
    hipLaunchKernelGGL(pw_wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return kmu::launch_status("pwconv_bwd_weight_reduce_multi");
}

extern "C" int kmu_pwconv_bwd_weight(const float* x, const float* gy, float* dw, float* dbias, void* ws, size_t ws_bytes, int B, int Ci,
                                     int Co, int P, int act_in, kmu_stream_t stream) {
    return pw_bwd_weight("pwconv_bwd_weight", x, gy, dw, dbias, ws, ws_bytes, B, Ci, Co, P, act_in, (hipStream_t)stream, Ci, Co);
}

// ---- grouped 1x1 convolution (block-diagonal weights): x [B, G*Ci, P] -> y [B, G*Co, P], w [G*Co, Ci], bias [G*Co] ----------
// The three direction branches of EnhancedViMBlock (KM_UNetV3_SH.py:99-101,137-139) run the SAME layer sequence on tensors of the
// same shape with their own weights: stacked along the channel axis they are one launch per layer instead of three -- at the
// 64x64 / 32x32 levels every launch is latency-bound, so three times the workgroups cost almost nothing.
extern "C" int kmu_pwconv_fwd_g(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co, int P, int act_in,
                                int groups, kmu_stream_t stream) {
    KMU_REQUIRE(x && w && y && groups >= 1, "pwconv_fwd_g: null pointer / groups < 1");
    return gemm("pwconv_fwd_g", x, w, Ci, 1, bias, nullptr, y, B, Ci, groups * Co, P, act_in, (hipStream_t)stream, nullptr, groups,
                (long)Co * Ci);
}

extern "C" int kmu_pwconv_bwd_input_g(const float* gy, const float* w, const float* x_pre, const float* addend, float* dx, int B, int Ci,
                                      int Co, int P, int act_in, int groups, kmu_stream_t stream) {
    KMU_REQUIRE(gy && w && dx && groups >= 1, "pwconv_bwd_input_g: null pointer / groups < 1");
    KMU_REQUIRE(!act_in || x_pre, "pwconv_bwd_input_g: act_in needs the pre-activation input");
    return gemm("pwconv_bwd_input_g", gy, w, 1, Ci, nullptr, act_in ? x_pre : nullptr, dx, B, Co, groups * Ci, P, 0, (hipStream_t)stream,
                addend, groups, (long)Co * Ci);
}

// weight gradient of group g of the grouped conv (dw [Co, Ci] and dbias [Co] of THAT group); one call per group -- these kernels
// run off the activation-gradient chain (ops._wgrad), where their launch count does not matter
extern "C" int kmu_pwconv_bwd_weight_g(const float* x, const float* gy, float* dw, float* dbias, void* ws, size_t ws_bytes, int B, int Ci,
                                       int Co, int P, int act_in, int groups, int g, kmu_stream_t stream) {
    KMU_REQUIRE(groups >= 1 && g >= 0 && g < groups, "pwconv_bwd_weight_g: group %d of %d", g, groups);
    return pw_bwd_weight("pwconv_bwd_weight_g", x + (size_t)g * Ci * P, gy + (size_t)g * Co * P, dw, dbias, ws, ws_bytes, B, Ci, Co, P,
                         act_in, (hipStream_t)stream, groups * Ci, groups * Co);
}
