// K2: EfficientViM HSM-SSD mixer for gfx950 (forward + backward) and its LayerNorm1D prologue.
//
// Replaces vim_block_init/efficient_vim_init.py:33-61 (HSMSSD.forward) and
// vim_block_init/vim_utils_init.py:50-59 (LayerNorm1D.forward) of the reference.  There is no
// recurrence in the reference's "SSM": it is softmax over the L = Hs*Hs token axis plus two batched
// contractions, so the parallel primitive is an associative online-softmax reduction
// (m, s, acc[C,N]) over token tiles, not a prefix scan.
//
// The [B,3N,L] BCdt tensor (12x the input for C=16) never touches HBM: every kernel re-derives the
// rows it needs for one 2-D token tile (+halo) in LDS:
//     P   = W_bcdt[rows] . x            1x1 projection   -> fp32 MFMA (v_mfma_f32_16x16x4_f32)
//     BCdt= depthwise3x3(P)             LDS stencil, 4-token strips, ds_read_b64
// forward  pass 1: rows {B, dt}  -> per-tile (m[n], s[n], acc[n][c] = sum_l e^{dt-m} B x)   (MFMA)
//          gate  : combine tiles, h = acc/s ; hz_proj, SiLU gate, out_proj on the [C,N] state
//          pass 2: rows {C}      -> y[c][l] = sum_n h2[c][n] Cm[n][l]                        (MFMA)
// backward pass A: rows {C}      -> dh2[c][n] partials = sum_l dy[c][l] Cm[n][l]
//          gate  : back through out_proj / gate / hz_proj -> dh[c][n], delta[n] = sum_c dh h
//          pass B: all rows on a 2-deep halo -> dBCdt -> transposed stencil -> dP -> dx, dW partials
//
// Wave-level (64-lane) reductions use DPP/bpermute shuffles; cross-workgroup combines go through
// small partial buffers reduced by the gate kernels (deterministic, no float atomics).
// Supported: C in {16,32,64}, N = 64 (KM-UNet hard-wires state_dim=64, KM_UNetV3_SH.py:166).
#include "common.h"
#include <stdlib.h>
#include <string.h>

using kmu::floatx4;

namespace {

constexpr int NS = 64;  // state_dim
#ifndef KMU_TY16_MAXC
#define KMU_TY16_MAXC 16   // channel counts up to this use 16x16 token tiles in pass 1 / pass 2 / pass A, larger ones 8x16:
                           // measured on MI355X (B=8): C=32 @64x64 pass1 56->38 us, passA 50->33, pass2 35->25 with 8x16 tiles
                           // (35 KB LDS => 4 workgroups per CU instead of 2); C=16 @128x128 loses (twice the tiles to combine)
#endif

__host__ __device__ constexpr int pad_mod32(int n, int want) { return n + ((want - (n % 32)) + 32) % 32; }

typedef float floatx2 __attribute__((ext_vector_type(2)));

// Token-tile geometry: TY x 16 interior tokens, 1-deep halo, flat halo index pos = hy*18 + hx.
template <int TY>
struct Geo {
    static constexpr int TX = 16, NTOK = TY * TX, NSTRIP = NTOK / 4;
    static constexpr int HR = TY + 2, HW = TX + 2, HX = HR * HW;
    static constexpr int MT = (HX + 15) / 16;           // 16-position MFMA row tiles covering the halo tile
    static constexpr int XS = pad_mod32(MT * 16, 16);   // per-row stride of xs / Ps, == 16 (mod 32)
    static constexpr int CMS = NTOK + 16;               // Cm tile stride, == 16 (mod 32)
    static constexpr int WBS = NTOK + 2;                // wB tile stride, == 2 (mod 32)
};

template <int C>
struct TileFor {
    static constexpr int TY = (C <= KMU_TY16_MAXC) ? 16 : 8;
};

// ---- stage the halo tile of x[b] : xs[c][pos], zero outside the image --------------------------
template <int C, int TY>
__device__ __forceinline__ void stage_x(float* xs, const float* __restrict__ xb, int ty0, int tx0, int Hs) {
    using G = Geo<TY>;
    for (int e = threadIdx.x; e < C * G::XS; e += 256) {
        const int c = e / G::XS, pos = e - c * G::XS;
        float v = 0.f;
        if (pos < G::HX) {
            const int hy = pos / G::HW, hx = pos - hy * G::HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            if (gy >= 0 && gy < Hs && gx >= 0 && gx < Hs) v = xb[(size_t)c * Hs * Hs + gy * Hs + gx];
        }
        xs[e] = v;
    }
}

// ---- P[16 rows][pos] = W[rows] . xs  for all halo positions (one 16-row chunk) ------------------
// rows: kout j < 8 -> row0 + j ; j >= 8 -> row1 + (j-8)   (lets one chunk hold 8 B-rows + 8 dt-rows)
template <int C, int TY>
__device__ __forceinline__ void proj_chunk(float* Ps, const float* xs, const float* __restrict__ w, int row0, int row1,
                                           int wave, int li, int lq) {
    using G = Geo<TY>;
    const int row = (li < 8) ? row0 + li : row1 + (li - 8);
    float wf[C / 4];
#pragma unroll
    for (int ks = 0; ks < C / 4; ++ks) wf[ks] = w[(size_t)row * C + ks * 4 + lq];
    for (int mt = wave; mt < G::MT; mt += 4) {
        floatx4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C / 4; ++ks)
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[(ks * 4 + lq) * G::XS + mt * 16 + li], wf[ks], d, 0, 0, 0);
        *reinterpret_cast<floatx4*>(Ps + li * G::XS + mt * 16 + lq * 4) = d;  // D[pos][kout]: col = li, rows lq*4+r
    }
}

// ---- depthwise 3x3 of one P row on a 4-token strip: out[q] for tokens (ty, sx+q) ----------------
template <int TY>
__device__ __forceinline__ void dw_strip(const float* Prow, const float* __restrict__ wk, int ty, int sx,
                                         float (&out)[4]) {
    using G = Geo<TY>;
    float wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = wk[t];
    out[0] = out[1] = out[2] = out[3] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const float* p = Prow + (ty + dy) * G::HW + sx;  // halo row ty+dy, halo cols sx .. sx+5
        const floatx2 a = *reinterpret_cast<const floatx2*>(p);
        const floatx2 b = *reinterpret_cast<const floatx2*>(p + 2);
        const floatx2 c = *reinterpret_cast<const floatx2*>(p + 4);
        const float v[6] = {a[0], a[1], b[0], b[1], c[0], c[1]};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) out[q] += wv[dy * 3 + dx] * v[q + dx];
    }
}

template <int WIDTH>
__device__ __forceinline__ float grp_max(float v) {
    static_assert(WIDTH == 64 || WIDTH == 32, "strip groups are a wave or a half wave");
    return kmu::wave_reduce<kmu::OpMax, WIDTH>(v);
}
template <int WIDTH>
__device__ __forceinline__ float grp_sum(float v) {
    static_assert(WIDTH == 64 || WIDTH == 32 || WIDTH == 16, "strip groups are a wave, a half wave or a DPP row");
    return kmu::wave_reduce<kmu::OpSum, WIDTH>(v);
}

// =================================================================================================
// forward pass 1
//   part_ms [B][T][2][64]   (m, s per tile)      part_acc [B][T][64 n][C]
// =================================================================================================
template <int C>
__global__ __launch_bounds__(256) void hsm_fwd_pass1(const float* __restrict__ x, const float* __restrict__ w_bcdt,
                                                     const float* __restrict__ w_dw, float* __restrict__ part_ms,
                                                     float* __restrict__ part_acc, int Hs, int tilesX) {
    constexpr int TY = TileFor<C>::TY;
    using G = Geo<TY>;
    constexpr int NPT = G::NSTRIP / 32;  // n's per thread per 8-n chunk (2 for TY=16, 1 for TY=8)
    constexpr int CT = C / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                       // [C][XS]
    float* Ps = xs + C * G::XS;             // [16][XS]
    float* wBs = Ps + 16 * G::XS;           // [16][WBS]
    float* red = smem;                      // aliases xs|Ps after the main loop: [4 waves][64][C]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int tile = blockIdx.x, b = blockIdx.y, T = gridDim.x;
    const int ty0 = (tile / tilesX) * TY, tx0 = (tile % tilesX) * G::TX;
    const int L = Hs * Hs;

    stage_x<C, TY>(xs, x + (size_t)b * C * L, ty0, tx0, Hs);

    // strip owned by this thread in the stencil phase
    const int strip = (TY == 16) ? lane : (lane & 31);
    const int sty = strip >> 2, sx = (strip & 3) * 4;
    bool valid[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) valid[q] = (ty0 + sty < Hs) && (tx0 + sx + q < Hs);

    floatx4 acc[4][CT];  // [n-tile of 16][c-tile]; this wave's share of the token (K) range
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[a][c] = floatx4{0.f, 0.f, 0.f, 0.f};

    float* ms_out = part_ms + ((size_t)b * T + tile) * 2 * NS;

    // The 8 chunks produce disjoint state columns n (their partials never meet), so at the coarse levels, where the tile
    // count alone cannot fill 256 CUs, gridDim.z = 2 or 4 workgroups share a tile's chunks -- no extra reduction.
    const int cpz = 8 / gridDim.z, ch0 = blockIdx.z * cpz;
#pragma unroll 1
    for (int ch = ch0; ch < ch0 + cpz; ++ch) {  // chunks of 8 states: P rows {B: n0..n0+7, dt: 128+n0..}
        const int n0 = ch * 8;
        __syncthreads();  // xs staged (first trip) / previous Ps + wBs consumers done
        proj_chunk<C, TY>(Ps, xs, w_bcdt, n0, 2 * NS + n0, wave, li, lq);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const int nl = (TY == 16) ? wave * 2 + t : wave * 2 + (lane >> 5);  // 0..7 within the chunk
            const int n = n0 + nl;
            float bm[4], dt[4];
            dw_strip<TY>(Ps + nl * G::XS, w_dw + (size_t)n * 9, sty, sx, bm);
            dw_strip<TY>(Ps + (8 + nl) * G::XS, w_dw + (size_t)(2 * NS + n) * 9, sty, sx, dt);
            float m = -INFINITY;
#pragma unroll
            for (int q = 0; q < 4; ++q) m = fmaxf(m, valid[q] ? dt[q] : -INFINITY);
            m = grp_max<G::NSTRIP>(m);
            float s = 0.f, wb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float e = valid[q] ? __expf(dt[q] - m) : 0.f;
                s += e;
                wb[q] = e * bm[q];
            }
            s = grp_sum<G::NSTRIP>(s);
            float* dst = wBs + ((ch & 1) * 8 + nl) * G::WBS + strip * 4;
            *reinterpret_cast<floatx2*>(dst) = floatx2{wb[0], wb[1]};
            *reinterpret_cast<floatx2*>(dst + 2) = floatx2{wb[2], wb[3]};
            if (strip == 0) {
                ms_out[n] = m;
                ms_out[NS + n] = s;
            }
        }
        if (ch & 1) {
            __syncthreads();  // wBs for 16 states complete
            // acc[n][c] += sum_tok wB[n][tok] * x[c][tok];  this wave covers token quads [16*wave, 16*wave+16)
            const int a = (ch - ch0) >> 1;
#pragma unroll 4
            for (int kq = 0; kq < G::NSTRIP / 4; ++kq) {
                const int sq = wave * (G::NSTRIP / 4) + kq;   // strip index = token quad
                const int tok = sq * 4 + lq;                  // this lane's k-slice token
                const int ipos = ((tok >> 4) + 1) * G::HW + (tok & 15) + 1;
                const float av = wBs[li * G::WBS + tok];      // A[i = n][k = tok]
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xs[(c * 16 + li) * G::XS + ipos], acc[a][c],
                                                                     0, 0, 0);  // B[k = tok][j = c]
            }
        }
    }
    __syncthreads();
    // cross-wave reduction of acc through LDS (aliases xs|Ps)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * NS + a * 16 + lq * 4 + r) * C + c * 16 + li] = acc[a][c][r];  // D[n][c]: col c=li
    __syncthreads();
    float* acc_out = part_acc + ((size_t)b * T + tile) * NS * C + (size_t)ch0 * 8 * C;   // this workgroup's rows n
    for (int e = tid; e < cpz * 8 * C; e += 256)
        acc_out[e] = red[e] + red[NS * C + e] + red[2 * NS * C + e] + red[3 * NS * C + e];
}

// =================================================================================================
// forward gate stage: one workgroup per batch element.
// state layout per batch: [ M(64) | S(64) | hpre(C*64) | hz(2C*64) | h2(C*64) ]   ([c][n] row-major)
// =================================================================================================
// per-sample gate state: M[64] S[64] hpre[C][64] hz[2C][64] h2[C][64], then (written by the round-4 forward only) the dense 3x3 weights
// M_b[C][9][C] of  y = conv3x3(x; M_b)  in fp32 for the backward pass (hsmssd_bwdc.inc)
__host__ __device__ inline size_t state_stride(int C) { return (size_t)2 * NS + (size_t)4 * C * NS + (size_t)9 * C * C; }

__device__ __forceinline__ float silu(float z) { return z / (1.f + __expf(-z)); }

constexpr int GN = 8;        // state columns per gate workgroup
constexpr int NGRP = NS / GN;  // gate workgroups per batch element

// All gate arithmetic is independent across the state index n (hz_proj / out_proj contract over channels), so
// a batch element is split over NGRP workgroups of GN columns each: grid (B, NGRP).
__global__ __launch_bounds__(256) void hsm_fwd_gate(const float* __restrict__ part_ms,
                                                    const float* __restrict__ part_acc,
                                                    const float* __restrict__ w_hz, const float* __restrict__ w_out,
                                                    const float* __restrict__ Dp, float* __restrict__ state,
                                                    float* __restrict__ h_out, int C, int T, int NG) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    w_hz += (size_t)(blockIdx.x % NG) * 2 * C * C;      // sample b = blockIdx.x uses weight group b % NG
    w_out += (size_t)(blockIdx.x % NG) * C * C;
    Dp += blockIdx.x % NG;
    float* Ms = smem;             // [GN]
    float* Ss = Ms + GN;          // [GN]
    float* hp = Ss + GN;          // [C][GN]
    float* hz = hp + C * GN;      // [2C][GN]
    float* gg = hz + 2 * C * GN;  // [C][GN]
    const int tid = threadIdx.x, b = blockIdx.x, nb = blockIdx.y * GN;
    const float* pms = part_ms + (size_t)b * T * 2 * NS;
    const float* pac = part_acc + (size_t)b * T * NS * C;
    float* st = state + (size_t)b * state_stride(C);
    {   // M, S of this workgroup's GN columns over the T tile partials: 32 partitions of tiles per column in parallel (as 8
        // threads walking 2 T dependent loads this phase alone took ~10 us of the kernel's 15)
        __shared__ float pr[32][GN];
        const int nl = tid & (GN - 1), tp = tid / GN, n = nb + nl;
        float m = -INFINITY;
        for (int t = tp; t < T; t += 32) m = fmaxf(m, pms[(size_t)t * 2 * NS + n]);
        pr[tp][nl] = m;
        __syncthreads();
        if (tid < GN) {
            float M = -INFINITY;
#pragma unroll
            for (int k = 0; k < 32; ++k) M = fmaxf(M, pr[k][tid]);
            Ms[tid] = M;
        }
        __syncthreads();
        const float M = Ms[nl];
        float sacc = 0.f;
        for (int t = tp; t < T; t += 32) sacc += pms[(size_t)t * 2 * NS + NS + n] * __expf(pms[(size_t)t * 2 * NS + n] - M);
        pr[tp][nl] = sacc;
        __syncthreads();
        if (tid < GN) {
            float S = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) S += pr[k][tid];          // fixed order
            Ss[tid] = S;
            st[nb + tid] = Ms[tid];
            st[NS + nb + tid] = S;
        }
    }
    __syncthreads();
    for (int e = tid; e < C * GN; e += 256) {  // e = nl*C + c (the partial layout is [n][C])
        const int nl = e / C, c = e - nl * C, n = nb + nl;
        float a = 0.f;
#pragma unroll 8
        for (int t = 0; t < T; ++t) a += pac[(size_t)t * NS * C + n * C + c] * __expf(pms[(size_t)t * 2 * NS + n] - Ms[nl]);
        a /= Ss[nl];
        hp[c * GN + nl] = a;
        st[2 * NS + c * NS + n] = a;
    }
    __syncthreads();
    for (int e = tid; e < 2 * C * GN; e += 256) {  // hz[k][n] = sum_c W_hz[k][c] hpre[c][n]   (:52)
        const int k = e / GN, nl = e - k * GN;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a += w_hz[k * C + c] * hp[c * GN + nl];
        hz[e] = a;
        st[2 * NS + C * NS + k * NS + nb + nl] = a;
    }
    __syncthreads();
    const float Dv = Dp[0];
    for (int e = tid; e < C * GN; e += 256) {  // g = h1*SiLU(z) + h1*D   (:55)
        const float h1 = hz[e], z = hz[C * GN + e];
        gg[e] = h1 * silu(z) + h1 * Dv;
    }
    __syncthreads();
    for (int e = tid; e < C * GN; e += 256) {  // h2[c'][n] = sum_c W_out[c'][c] g[c][n]
        const int co = e / GN, nl = e - co * GN;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a += w_out[co * C + c] * gg[c * GN + nl];
        st[2 * NS + 3 * C * NS + co * NS + nb + nl] = a;
        h_out[(size_t)b * C * NS + co * NS + nb + nl] = a;
    }
}

// =================================================================================================
// forward pass 2:  y[c][tok] = sum_n h2[c][n] * Cm[n][tok]       (:57)
// =================================================================================================
template <int C>
__global__ __launch_bounds__(256) void hsm_fwd_pass2(const float* __restrict__ x, const float* __restrict__ w_bcdt,
                                                     const float* __restrict__ w_dw, const float* __restrict__ state,
                                                     float* __restrict__ y, int Hs, int tilesX) {
    constexpr int TY = TileFor<C>::TY;
    using G = Geo<TY>;
    constexpr int CT = C / 16, RPW = TY / 4;  // tile rows per wave
    constexpr int IPT = G::NSTRIP * 16 / 256;  // (strip, row) stencil items per thread per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                 // [C][XS]
    float* Ps = xs + C * G::XS;       // [16][XS]
    float* Cms = Ps + 16 * G::XS;     // [16][CMS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty0 = (tile / tilesX) * TY, tx0 = (tile % tilesX) * G::TX;
    const int L = Hs * Hs;
    const float* h2 = state + (size_t)b * state_stride(C) + 2 * NS + 3 * C * NS;  // [C][64]

    stage_x<C, TY>(xs, x + (size_t)b * C * L, ty0, tx0, Hs);

    floatx4 acc[RPW][CT];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = floatx4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {  // 4 chunks of 16 states: P rows 64 + n0 ..
        const int n0 = ch * 16;
        __syncthreads();
        proj_chunk<C, TY>(Ps, xs, w_bcdt, NS + n0, NS + n0 + 8, wave, li, lq);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < IPT; ++t) {
            const int item = t * 256 + tid;
            const int strip = item % G::NSTRIP, nl = item / G::NSTRIP;
            float cm[4];
            dw_strip<TY>(Ps + nl * G::XS, w_dw + (size_t)(NS + n0 + nl) * 9, strip >> 2, (strip & 3) * 4, cm);
            *reinterpret_cast<floatx4*>(Cms + nl * G::CMS + strip * 4) = floatx4{cm[0], cm[1], cm[2], cm[3]};
        }
        __syncthreads();
        float hf[CT][4];  // B[k = n][j = c] = h2[c][n0 + ks*4 + lq]
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) hf[c][ks] = h2[(c * 16 + li) * NS + n0 + ks * 4 + lq];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = wave * RPW + r;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float av = Cms[(ks * 4 + lq) * G::CMS + row * 16 + li];  // A[i = tok][k = n]
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, hf[c][ks], acc[r][c], 0, 0, 0);
            }
        }
    }
    // D[tok][c]: lane col = channel li, rows = 4 consecutive tokens lq*4 + r of tile row `row`
    float* yb = y + (size_t)b * C * L;
    const bool vec_ok = (Hs & 3) == 0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int gy = ty0 + wave * RPW + r, gx = tx0 + lq * 4;
        if (gy >= Hs || gx >= Hs) continue;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            float* dst = yb + (size_t)(c * 16 + li) * L + gy * Hs + gx;
            if (vec_ok && gx + 3 < Hs) {
                *reinterpret_cast<floatx4*>(dst) = acc[r][c];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (gx + q < Hs) dst[q] = acc[r][c][q];
            }
        }
    }
}

// =================================================================================================
// LayerNorm1D over the channel axis of [B,C,L]  (vim_utils_init.py:50-59), one thread per token
// =================================================================================================
__global__ __launch_bounds__(256) void ln1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ stats, int C, int L, float eps, int G) {
    const int b = blockIdx.y, l = blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    w += (b % G) * C;          // G > 1: [B, G*C, L] seen as B*G samples, sample b uses the affine parameters of group b % G
    bias += (b % G) * C;
    const float* xp = x + (size_t)b * C * L + l;
    float mu = 0.f;
    for (int c = 0; c < C; ++c) mu += xp[(size_t)c * L];
    mu /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = xp[(size_t)c * L] - mu;
        var += d * d;
    }
    var /= (float)C;
    const float rstd = 1.f / sqrtf(var + eps);
    float* yp = y + (size_t)b * C * L + l;
    for (int c = 0; c < C; ++c) yp[(size_t)c * L] = (xp[(size_t)c * L] - mu) * rstd * w[c] + bias[c];
    stats[((size_t)b * L + l) * 2] = rstd;
    stats[((size_t)b * L + l) * 2 + 1] = mu;
}

// dx = rstd * (g - mean_c(g) - xhat * mean_c(g*xhat)), g = dy*w ; per-block partials of dw, db
__global__ __launch_bounds__(256) void ln1d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ stats, const float* __restrict__ dy,
                                                       float* __restrict__ dx, float* __restrict__ dw_part,
                                                       float* __restrict__ db_part, int C, int L, int G,
                                                       const float* __restrict__ addend) {
    __shared__ float red[2][4];
    const int b = blockIdx.y, l = blockIdx.x * 256 + threadIdx.x;
    w += (b % G) * C;
    const bool ok = l < L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)b * C * L + (ok ? l : 0);
    float rstd = 0.f, mu = 0.f, s1 = 0.f, s2 = 0.f;
    if (ok) {
        rstd = stats[((size_t)b * L + l) * 2];
        mu = stats[((size_t)b * L + l) * 2 + 1];
        for (int c = 0; c < C; ++c) {
            const float g = dy[base + (size_t)c * L] * w[c];
            const float xh = (x[base + (size_t)c * L] - mu) * rstd;
            s1 += g;
            s2 += g * xh;
        }
        s1 /= (float)C;
        s2 /= (float)C;
    }
    const int prow = blockIdx.y * gridDim.x + blockIdx.x;
    for (int c = 0; c < C; ++c) {
        float gdw = 0.f, gdb = 0.f;
        if (ok) {
            const float d = dy[base + (size_t)c * L];
            const float xh = (x[base + (size_t)c * L] - mu) * rstd;
            dx[base + (size_t)c * L] = rstd * (d * w[c] - s1 - xh * s2) + (addend ? addend[base + (size_t)c * L] : 0.f);
            gdw = d * xh;
            gdb = d;
        }
        gdw = kmu::wave_sum(gdw);
        gdb = kmu::wave_sum(gdb);
        __syncthreads();
        if (lane == 0) {
            red[0][wave] = gdw;
            red[1][wave] = gdb;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            dw_part[(size_t)prow * C + c] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
            db_part[(size_t)prow * C + c] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        }
    }
}

// Register-resident variants for the channel counts KM-UNet uses: a thread owns V consecutive tokens (V-wide loads,
// every element read exactly once), x / dy stay in registers between the statistics and the output pass, and the
// parameter-gradient partials need one block reduction per kernel instead of one per channel.
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

template <int C, int V>
__global__ __launch_bounds__(256) void ln1d_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           float* __restrict__ stats, int L, float eps, int G) {
    const int b = blockIdx.y, l0 = (blockIdx.x * 256 + threadIdx.x) * V;
    if (l0 >= L) return;
    w += (b % G) * C;
    bias += (b % G) * C;
    const size_t base = (size_t)b * C * L + l0;
    float v[C][V], mu[V], rstd[V];
#pragma unroll
    for (int c = 0; c < C; ++c) ldv<V>(v[c], x + base + (size_t)c * L);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) m += v[c][i];
        m /= (float)C;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float d = v[c][i] - m;
            var += d * d;
        }
        var /= (float)C;
        mu[i] = m;
        rstd[i] = 1.f / sqrtf(var + eps);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float wc = w[c], bc = bias[c];
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = (v[c][i] - mu[i]) * rstd[i] * wc + bc;
        stv<V>(y + base + (size_t)c * L, o);
    }
    float* sp = stats + ((size_t)b * L + l0) * 2;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        sp[2 * i] = rstd[i];
        sp[2 * i + 1] = mu[i];
    }
}

template <int C, int V>
__global__ __launch_bounds__(256) void ln1d_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ stats, const float* __restrict__ dy,
                                                           float* __restrict__ dx, float* __restrict__ dw_part,
                                                           float* __restrict__ db_part, int L, int G,
                                                           const float* __restrict__ addend) {
    // addend: gradient of a second consumer of x (EfficientViMBlock blends the mixer's output with the x it normalised,
    // efficient_vim_init.py:88-90) added here instead of in an ATen fan-in launch
    __shared__ float red[2 * C][4];
    const int b = blockIdx.y, l0 = (blockIdx.x * 256 + threadIdx.x) * V;
    w += (b % G) * C;
    const bool ok = l0 < L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)b * C * L + (ok ? l0 : 0);
    float gdw[C], gdb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gdw[c] = gdb[c] = 0.f;
    if (ok) {
        float xh[C][V], g[C][V], rstd[V], s1[V], s2[V];
        const float* sp = stats + ((size_t)b * L + l0) * 2;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            ldv<V>(xh[c], x + base + (size_t)c * L);
            ldv<V>(g[c], dy + base + (size_t)c * L);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            rstd[i] = sp[2 * i];
            const float mu = sp[2 * i + 1];
            s1[i] = s2[i] = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                xh[c][i] = (xh[c][i] - mu) * rstd[i];
                gdw[c] += g[c][i] * xh[c][i];
                gdb[c] += g[c][i];
                const float gw = g[c][i] * w[c];
                s1[i] += gw;
                s2[i] += gw * xh[c][i];
            }
            s1[i] /= (float)C;
            s2[i] /= (float)C;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float wc = w[c];
            float o[V];
#pragma unroll
            for (int i = 0; i < V; ++i) o[i] = rstd[i] * (g[c][i] * wc - s1[i] - xh[c][i] * s2[i]);
            if (addend) {
                float ad[V];
                ldv<V>(ad, addend + base + (size_t)c * L);
#pragma unroll
                for (int i = 0; i < V; ++i) o[i] += ad[i];
            }
            stv<V>(dx + base + (size_t)c * L, o);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float a = kmu::wave_sum(gdw[c]), d = kmu::wave_sum(gdb[c]);
        if (lane == 0) {
            red[c][wave] = a;
            red[C + c][wave] = d;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        const int c = threadIdx.x % C;
        const float s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
        const size_t prow = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        (threadIdx.x < C ? dw_part : db_part)[prow * C + c] = s;
    }
}

// tokens per thread of the register variants (0 = generic kernels): forward keeps C*V = 64 values, backward 2 x 32
inline int ln1d_vf(int C, int L) { return (C == 16 && L % 4 == 0) ? 4 : (C == 32 && L % 2 == 0) ? 2 : (C == 64 ? 1 : 0); }
inline int ln1d_vb(int C, int L) { return (C == 16 && L % 2 == 0) ? 2 : ((C == 32 || C == 64) ? 1 : 0); }

// ---------------------------------------------------------------------------------------------
template <int C>
size_t lds_pass1() {
    using G = Geo<TileFor<C>::TY>;
    size_t a = (size_t)C * G::XS + 16 * G::XS + 16 * G::WBS, r = (size_t)4 * NS * C;
    return (a > r ? a : r) * sizeof(float);
}
template <int C>
size_t lds_pass2() {
    using G = Geo<TileFor<C>::TY>;
    return ((size_t)C * G::XS + 16 * G::XS + 16 * G::CMS) * sizeof(float);
}
// workgroups per tile in pass 1 / pass A (their chunks own disjoint state columns): enough to give every CU work
inline int chunk_split(int tile_workgroups) { return tile_workgroups < 128 ? 4 : (tile_workgroups < 512 ? 2 : 1); }

inline int tiles_for(int C, int Hs, int* tilesX) {
    const int TY = (C <= KMU_TY16_MAXC) ? 16 : 8;
    *tilesX = kmu::cdiv(Hs, 16);
    return *tilesX * kmu::cdiv(Hs, TY);
}

// stages: bit 0 = pass 1, bit 1 = gate, bit 2 = pass 2 (7 = the whole forward)
template <int C>
int fwd_impl(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz, const float* w_out,
             const float* D, float* y, float* h, float* state, float* ws, int B, int Hs, int stages, hipStream_t st) {
    int tilesX;
    const int T = tiles_for(C, Hs, &tilesX);
    float* part_ms = ws;
    float* part_acc = ws + (size_t)B * T * 2 * NS;
    const size_t l1 = lds_pass1<C>(), l2 = lds_pass2<C>();
    int rc = 0;
    if (stages & 1) {
        KMU_MAX_LDS(hsm_fwd_pass1<C>, l1);
        hipLaunchKernelGGL(hsm_fwd_pass1<C>, dim3(T, B, chunk_split(T * B)), dim3(256), l1, st, x, w_bcdt, w_dw, part_ms, part_acc, Hs,
                           tilesX);
        rc = kmu::launch_status("hsmssd_fwd pass1");
        if (rc) return rc;
    }
    if (stages & 2) {
        const size_t lg = ((size_t)2 * GN + 4 * C * GN) * sizeof(float);
        hipLaunchKernelGGL(hsm_fwd_gate, dim3(B, NGRP), dim3(256), lg, st, part_ms, part_acc, w_hz, w_out, D, state, h, C, T, 1);
        rc = kmu::launch_status("hsmssd_fwd gate");
        if (rc) return rc;
    }
    if (stages & 4) {
        KMU_MAX_LDS(hsm_fwd_pass2<C>, l2);
        hipLaunchKernelGGL(hsm_fwd_pass2<C>, dim3(T, B), dim3(256), l2, st, x, w_bcdt, w_dw, state, y, Hs, tilesX);
        rc = kmu::launch_status("hsmssd_fwd pass2");
    }
    return rc;
}

#include "hsmssd_x3.inc"
#include "hsmssd_v2.inc"
#include "hsmssd_bwd.inc"
#include "hsmssd_bwdb.inc"
#include "hsmssd_bwdc.inc"

// backward on the matrix core: ws = [partA | dhpre | delta | composite weights (fragment order) | transposed composite weights]
template <int C>
int bwd_impl_x3(const float* x, const float* dy, const float* dh, const float* w_bcdt, const float* w_dw, const float* w_hz,
                const float* w_out, const float* D, const float* state, float* dx, float* p_bcdt, float* p_dw, float* p_hz,
                float* p_out, float* p_D, float* ws, int B, int Hs, int stages, hipStream_t st, int NG = 1,
                const void* wpk_ext = nullptr) {
    int txA;
    const int TA = tiles_x3<C>(Hs, &txA);
    float* partA = ws;
    float* dhp = partA + (size_t)B * TA * C * NS;
    float* delta = dhp + (size_t)B * C * NS;
    unsigned short* wpk = wpk_ext ? (unsigned short*)wpk_ext : reinterpret_cast<unsigned short*>(delta + (size_t)B * NS);
    const size_t la = lds_passA_x3<C>(), lg = (size_t)7 * C * GN * sizeof(float);
    // Pass B: the exact-fp32 kernel (hsmssd_bwd.inc).
    const bool split_dx = passB_split(C, Hs) == 2;      // fp32 pass B at C = 64: two workgroups per tile add into dx
    int rc = 0;
    if (stages & 1) {
        if (!wpk_ext) {
            hipLaunchKernelGGL(hsm_pack_x3_kernel<C>, dim3(48, NG), dim3(256), 0, st, w_bcdt, w_dw, wpk);
            rc = kmu::launch_status("hsmssd_bwd pack");
            if (rc) return rc;
        }
        KMU_MAX_LDS(hsm_bwd_passA_x3<C>, la);
        hipLaunchKernelGGL(hsm_bwd_passA_x3<C>, dim3(TA, B, chunk_split(TA * B)), dim3(256), la, st, x, dy, (const bf16x8*)wpk, partA, Hs, txA, NG);
        rc = kmu::launch_status("hsmssd_bwd passA (bf16x3)");
        if (rc) return rc;
    }
    if (stages & 2) {
        hipLaunchKernelGGL(hsm_bwd_gate, dim3(B, NGRP), dim3(256), lg, st, partA, dh, w_hz, w_out, D, state, dhp, delta, p_hz, p_out,
                           p_D, split_dx ? dx : (float*)nullptr, C * Hs * Hs / 4, C, TA, NG);
        rc = kmu::launch_status("hsmssd_bwd gate");
        if (rc) return rc;
    }
    if (stages & 4) {
        {
            int txF;
            const int TBF = tilesB_for(C, Hs, &txF);
            const size_t lbf = lds_passB<C>();
            KMU_MAX_LDS(hsm_bwd_passB<C>, lbf);
            hipLaunchKernelGGL(hsm_bwd_passB<C>, dim3(TBF, B, passB_split(C, Hs)), dim3(TileForB<C>::NW * 64), lbf, st, x, dy, w_bcdt, w_dw,
                               state, dhp, delta, dx, p_bcdt, p_dw, Hs, txF, NG);
            rc = kmu::launch_status("hsmssd_bwd passB");
        }
    }
    return rc;
}

}  // namespace

extern "C" size_t kmu_hsmssd_state_elems(int B, int C, int N) { return (size_t)B * ((size_t)2 * N + (size_t)4 * C * N + (size_t)9 * C * C); }

extern "C" size_t kmu_hsmssd_fwd_ws_bytes(int B, int C, int N, int Hs) {
    int tx;
    int T = tiles_for(C, Hs, &tx);
    const int T3 = C == 16 ? tiles_x3<16>(Hs, &tx) : (C == 32 ? tiles_x3<32>(Hs, &tx) : tiles_x3<64>(Hs, &tx));
    if (T3 > T) T = T3;     // one size serves both the exact-fp32 and the bf16x3 tilings; the latter appends its packed weights
    return (size_t)B * T * ((size_t)2 * N + (size_t)N * C) * sizeof(float) + pack_x3_elems(C) * 2;
}

static int hsmssd_fwd_stages(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz,
                             const float* w_out, const float* D, float* y, float* h, float* state, void* ws,
                             size_t ws_bytes, int B, int C, int N, int Hs, int stages, kmu_stream_t stream, bool x3 = false,
                             int groups = 1, const void* wpk = nullptr) {
    KMU_REQUIRE(x && w_bcdt && w_dw && w_hz && w_out && D && y && h && state && ws, "hsmssd_fwd: null pointer");
    KMU_REQUIRE(groups >= 1 && (groups == 1 || x3) && B % groups == 0, "hsmssd_fwd: %d weight groups need the bf16x3 path and B %% groups == 0",
                groups);
    KMU_REQUIRE(N == NS, "hsmssd_fwd: state_dim=%d unsupported (kernels are built for 64)", N);
    KMU_REQUIRE(C == 16 || C == 32 || C == 64, "hsmssd_fwd: C=%d unsupported (16/32/64)", C);
    KMU_REQUIRE(B > 0 && B <= 65535 && Hs > 0, "hsmssd_fwd: bad dims");
    KMU_REQUIRE(ws_bytes >= kmu_hsmssd_fwd_ws_bytes(B, C, N, Hs) + (size_t)(groups - 1) * pack_x3_elems(C) * 2, "hsmssd_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (x3) {
        if (C == 16) return fwd_impl_x3<16>(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, (float*)ws, B, Hs, stages, st, groups, wpk);
        if (C == 32) return fwd_impl_x3<32>(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, (float*)ws, B, Hs, stages, st, groups, wpk);
        return fwd_impl_x3<64>(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, (float*)ws, B, Hs, stages, st, groups, wpk);
    }
    if (C == 16) return fwd_impl<16>(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, (float*)ws, B, Hs, stages, st);
    if (C == 32) return fwd_impl<32>(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, (float*)ws, B, Hs, stages, st);
    return fwd_impl<64>(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, (float*)ws, B, Hs, stages, st);
}

extern "C" int kmu_hsmssd_fwd(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz,
                              const float* w_out, const float* D, float* y, float* h, float* state, void* ws,
                              size_t ws_bytes, int B, int C, int N, int Hs, kmu_stream_t stream) {
    return hsmssd_fwd_stages(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, ws, ws_bytes, B, C, N, Hs, 7, stream);
}
extern "C" int kmu_hsmssd_fwd_stage(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz,
                                    const float* w_out, const float* D, float* y, float* h, float* state, void* ws,
                                    size_t ws_bytes, int B, int C, int N, int Hs, int stage, kmu_stream_t stream) {
    KMU_REQUIRE(stage >= 0 && stage <= 2, "hsmssd_fwd_stage: stage must be 0 (pass 1), 1 (gate) or 2 (pass 2)");
    return hsmssd_fwd_stages(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, ws, ws_bytes, B, C, N, Hs, 1 << stage, stream);
}

extern "C" int kmu_layernorm1d_partials(int B, int C, int L) {
    const int V = ln1d_vb(C, L);
    return B * kmu::cdiv(L, 256 * (V ? V : 1));
}

// groups > 1: x is [B/groups, groups*C, L] (= B samples of [C, L]); sample b is normalised with weight / bias rows b % groups of
// the stacked [groups, C] parameters.  The weight-gradient partials keep their per-sample row order (kmu_layernorm1d_partials).
extern "C" int kmu_layernorm1d_fwd_g(const float* x, const float* weight, const float* bias, float* y, float* rstd_mean,
                                     int B, int C, int L, float eps, int groups, kmu_stream_t stream) {
    KMU_REQUIRE(x && weight && bias && y && rstd_mean, "layernorm1d_fwd: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && C > 0 && L > 0 && groups >= 1 && B % groups == 0, "layernorm1d_fwd: bad dims");
    const int G = groups;
    hipStream_t st = (hipStream_t)stream;
    const int V = ln1d_vf(C, L);
    const dim3 grid(kmu::cdiv(L, 256 * (V ? V : 1)), B);
    if (C == 16 && V == 4)
        hipLaunchKernelGGL((ln1d_fwd_reg_kernel<16, 4>), grid, dim3(256), 0, st, x, weight, bias, y, rstd_mean, L, eps, G);
    else if (C == 32 && V == 2)
        hipLaunchKernelGGL((ln1d_fwd_reg_kernel<32, 2>), grid, dim3(256), 0, st, x, weight, bias, y, rstd_mean, L, eps, G);
    else if (C == 64 && V == 1)
        hipLaunchKernelGGL((ln1d_fwd_reg_kernel<64, 1>), grid, dim3(256), 0, st, x, weight, bias, y, rstd_mean, L, eps, G);
    else
        hipLaunchKernelGGL(ln1d_fwd_kernel, grid, dim3(256), 0, st, x, weight, bias, y, rstd_mean, C, L, eps, G);
    return kmu::launch_status("layernorm1d_fwd");
}

extern "C" int kmu_layernorm1d_fwd(const float* x, const float* weight, const float* bias, float* y, float* rstd_mean,
                                   int B, int C, int L, float eps, kmu_stream_t stream) {
    return kmu_layernorm1d_fwd_g(x, weight, bias, y, rstd_mean, B, C, L, eps, 1, stream);
}

extern "C" int kmu_layernorm1d_bwd_add(const float* x, const float* weight, const float* rstd_mean, const float* dy, const float* addend,
                                       float* dx, float* d_weight_partial, float* d_bias_partial, int B, int C, int L,
                                       int groups, kmu_stream_t stream) {
    KMU_REQUIRE(x && weight && rstd_mean && dy && dx && d_weight_partial && d_bias_partial, "layernorm1d_bwd: null pointer");
    KMU_REQUIRE(B > 0 && B <= 65535 && C > 0 && L > 0 && groups >= 1 && B % groups == 0, "layernorm1d_bwd: bad dims");
    const int G = groups;
    hipStream_t st = (hipStream_t)stream;
    const int V = ln1d_vb(C, L);
    const dim3 grid(kmu::cdiv(L, 256 * (V ? V : 1)), B);
    if (C == 16 && V == 2)
        hipLaunchKernelGGL((ln1d_bwd_reg_kernel<16, 2>), grid, dim3(256), 0, st, x, weight, rstd_mean, dy, dx, d_weight_partial,
                           d_bias_partial, L, G, addend);
    else if (C == 32 && V == 1)
        hipLaunchKernelGGL((ln1d_bwd_reg_kernel<32, 1>), grid, dim3(256), 0, st, x, weight, rstd_mean, dy, dx, d_weight_partial,
                           d_bias_partial, L, G, addend);
    else if (C == 64 && V == 1)
        hipLaunchKernelGGL((ln1d_bwd_reg_kernel<64, 1>), grid, dim3(256), 0, st, x, weight, rstd_mean, dy, dx, d_weight_partial,
                           d_bias_partial, L, G, addend);
    else
        hipLaunchKernelGGL(ln1d_bwd_kernel, grid, dim3(256), 0, st, x, weight, rstd_mean, dy, dx, d_weight_partial, d_bias_partial, C,
                           L, G, addend);
    return kmu::launch_status("layernorm1d_bwd");
}

extern "C" int kmu_layernorm1d_bwd_g(const float* x, const float* weight, const float* rstd_mean, const float* dy,
                                     float* dx, float* d_weight_partial, float* d_bias_partial, int B, int C, int L,
                                     int groups, kmu_stream_t stream) {
    return kmu_layernorm1d_bwd_add(x, weight, rstd_mean, dy, nullptr, dx, d_weight_partial, d_bias_partial, B, C, L, groups, stream);
}

extern "C" int kmu_layernorm1d_bwd(const float* x, const float* weight, const float* rstd_mean, const float* dy,
                                   float* dx, float* d_weight_partial, float* d_bias_partial, int B, int C, int L,
                                   kmu_stream_t stream) {
    return kmu_layernorm1d_bwd_g(x, weight, rstd_mean, dy, dx, d_weight_partial, d_bias_partial, B, C, L, 1, stream);
}

extern "C" int kmu_hsmssd_fwd_stage_x3(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz,
                                       const float* w_out, const float* D, float* y, float* h, float* state, void* ws,
                                       size_t ws_bytes, int B, int C, int N, int Hs, int stage, kmu_stream_t stream) {
    KMU_REQUIRE(stage >= 0 && stage <= 2, "hsmssd_fwd_stage_x3: stage must be 0 (pack + pass 1), 1 (gate) or 2 (pass 2)");
    return hsmssd_fwd_stages(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, ws, ws_bytes, B, C, N, Hs, 1 << stage, stream, true);
}

// Grouped HSMSSD: x is [B, C, L] with B = samples x groups (a [B/groups, groups*C, L] tensor, the three direction branches of
// EnhancedViMBlock stacked along the channel axis); sample b uses weight set b % groups of the stacked
// w_bcdt [groups, 3N, C], w_dw [groups, 3N, 9], w_hz [groups, 2C, C], w_out [groups, C, C], D [groups].
extern "C" size_t kmu_hsmssd_fwd_ws_bytes_g(int B, int C, int N, int Hs, int groups) {
    return kmu_hsmssd_fwd_ws_bytes(B, C, N, Hs) + (size_t)(groups > 1 ? groups - 1 : 0) * pack_x3_elems(C) * 2;
}
extern "C" int kmu_hsmssd_fwd_stage_x3_g(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz,
                                         const float* w_out, const float* D, float* y, float* h, float* state, void* ws,
                                         size_t ws_bytes, int B, int C, int N, int Hs, int stage, int groups, kmu_stream_t stream) {
    KMU_REQUIRE(stage >= 0 && stage <= 2, "hsmssd_fwd_stage_x3_g: stage must be 0 (pack + pass 1), 1 (gate) or 2 (pass 2)");
    return hsmssd_fwd_stages(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, ws, ws_bytes, B, C, N, Hs, 1 << stage, stream, true, groups);
}

// ---- backward entry points (kernels in hsmssd_bwd.inc) -------------------------------------------
extern "C" size_t kmu_hsmssd_bwd_ws_bytes(int B, int C, int N, int Hs) {
    int tx;
    const int TA = tiles_for(C, Hs, &tx);
    return ((size_t)B * TA * C * N + (size_t)B * C * N + (size_t)B * N) * sizeof(float);
}
extern "C" int kmu_hsmssd_gate_partials(int B) { return B * NGRP; }
extern "C" int kmu_hsmssd_bwd_partials(int B, int C, int Hs) {
    int tx;
    return B * tilesB_for(C, Hs, &tx);
}
extern "C" size_t kmu_hsmssd_bwd_ws_bytes_x3(int B, int C, int N, int Hs) {
    int tx;
    const int TA = C == 16 ? tiles_x3<16>(Hs, &tx) : (C == 32 ? tiles_x3<32>(Hs, &tx) : tiles_x3<64>(Hs, &tx));
    return ((size_t)B * TA * C * N + (size_t)B * C * N + (size_t)B * N) * sizeof(float) + pack_x3_elems(C) * 2;
}
extern "C" int kmu_hsmssd_bwd_partials_x3(int B, int C, int Hs) {
    int tx;
    return B * tilesB_for(C, Hs, &tx);       // pass B runs on the exact-fp32 kernel: its tiling
}

static int hsmssd_bwd_stages(const float* x, const float* dy, const float* dh, const float* w_bcdt, const float* w_dw,
                             const float* w_hz, const float* w_out, const float* D, const float* state, float* dx,
                             float* d_w_bcdt_partial, float* d_w_dw_partial, float* d_w_hz_partial,
                             float* d_w_out_partial, float* d_D_partial, void* ws, size_t ws_bytes, int B, int C, int N,
                             int Hs, int stages, kmu_stream_t stream, bool x3 = false, int groups = 1, const void* wpk = nullptr) {
    KMU_REQUIRE(groups >= 1 && (groups == 1 || x3) && B % groups == 0, "hsmssd_bwd: %d weight groups need the bf16x3 path and B %% groups == 0",
                groups);
    KMU_REQUIRE(x && dy && w_bcdt && w_dw && w_hz && w_out && D && state && dx && d_w_bcdt_partial && d_w_dw_partial &&
                    d_w_hz_partial && d_w_out_partial && d_D_partial && ws,
                "hsmssd_bwd: null pointer");
    KMU_REQUIRE(N == NS, "hsmssd_bwd: state_dim=%d unsupported (kernels are built for 64)", N);
    KMU_REQUIRE(C == 16 || C == 32 || C == 64, "hsmssd_bwd: C=%d unsupported (16/32/64)", C);
    KMU_REQUIRE(B > 0 && B <= 65535 && Hs > 0, "hsmssd_bwd: bad dims");
    KMU_REQUIRE(ws_bytes >= (x3 ? kmu_hsmssd_bwd_ws_bytes_x3(B, C, N, Hs) + (size_t)(groups - 1) * pack_x3_elems(C) * 2
                                : kmu_hsmssd_bwd_ws_bytes(B, C, N, Hs)),
                "hsmssd_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
    if (x3) {
        if (C == 16)
            return bwd_impl_x3<16>(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                                   d_w_hz_partial, d_w_out_partial, d_D_partial, w, B, Hs, stages, st, groups, wpk);
        if (C == 32)
            return bwd_impl_x3<32>(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                                   d_w_hz_partial, d_w_out_partial, d_D_partial, w, B, Hs, stages, st, groups, wpk);
        return bwd_impl_x3<64>(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                               d_w_hz_partial, d_w_out_partial, d_D_partial, w, B, Hs, stages, st, groups, wpk);
    }
    if (C == 16)
        return bwd_impl<16>(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                            d_w_hz_partial, d_w_out_partial, d_D_partial, w, B, Hs, stages, st);
    if (C == 32)
        return bwd_impl<32>(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                            d_w_hz_partial, d_w_out_partial, d_D_partial, w, B, Hs, stages, st);
    return bwd_impl<64>(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                        d_w_hz_partial, d_w_out_partial, d_D_partial, w, B, Hs, stages, st);
}

extern "C" int kmu_hsmssd_bwd(const float* x, const float* dy, const float* dh, const float* w_bcdt, const float* w_dw,
                              const float* w_hz, const float* w_out, const float* D, const float* state, float* dx,
                              float* d_w_bcdt_partial, float* d_w_dw_partial, float* d_w_hz_partial,
                              float* d_w_out_partial, float* d_D_partial, void* ws, size_t ws_bytes, int B, int C, int N,
                              int Hs, kmu_stream_t stream) {
    return hsmssd_bwd_stages(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                             d_w_hz_partial, d_w_out_partial, d_D_partial, ws, ws_bytes, B, C, N, Hs, 7, stream);
}
extern "C" int kmu_hsmssd_bwd_stage_x3(const float* x, const float* dy, const float* dh, const float* w_bcdt,
                                       const float* w_dw, const float* w_hz, const float* w_out, const float* D,
                                       const float* state, float* dx, float* d_w_bcdt_partial, float* d_w_dw_partial,
                                       float* d_w_hz_partial, float* d_w_out_partial, float* d_D_partial, void* ws,
                                       size_t ws_bytes, int B, int C, int N, int Hs, int stage, kmu_stream_t stream) {
    KMU_REQUIRE(stage >= 0 && stage <= 2, "hsmssd_bwd_stage_x3: stage must be 0 (pack + pass A), 1 (gate) or 2 (pass B)");
    return hsmssd_bwd_stages(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                             d_w_hz_partial, d_w_out_partial, d_D_partial, ws, ws_bytes, B, C, N, Hs, 1 << stage, stream, true);
}
// grouped backward (see kmu_hsmssd_fwd_stage_x3_g).  The per-tile / per-sample weight-gradient partials keep their sample-major row
// order: rows of sample b belong to weight group b % groups (the caller sums them per group).
extern "C" size_t kmu_hsmssd_bwd_ws_bytes_x3_g(int B, int C, int N, int Hs, int groups) {
    return kmu_hsmssd_bwd_ws_bytes_x3(B, C, N, Hs) + (size_t)(groups > 1 ? groups - 1 : 0) * pack_x3_elems(C) * 2;
}
extern "C" int kmu_hsmssd_bwd_stage_x3_g(const float* x, const float* dy, const float* dh, const float* w_bcdt,
                                         const float* w_dw, const float* w_hz, const float* w_out, const float* D,
                                         const float* state, float* dx, float* d_w_bcdt_partial, float* d_w_dw_partial,
                                         float* d_w_hz_partial, float* d_w_out_partial, float* d_D_partial, void* ws,
                                         size_t ws_bytes, int B, int C, int N, int Hs, int stage, int groups, kmu_stream_t stream) {
    KMU_REQUIRE(stage >= 0 && stage <= 2, "hsmssd_bwd_stage_x3_g: stage must be 0 (pack + pass A), 1 (gate) or 2 (pass B)");
    return hsmssd_bwd_stages(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                             d_w_hz_partial, d_w_out_partial, d_D_partial, ws, ws_bytes, B, C, N, Hs, 1 << stage, stream, true, groups);
}
extern "C" int kmu_hsmssd_bwd_stage(const float* x, const float* dy, const float* dh, const float* w_bcdt,
                                    const float* w_dw, const float* w_hz, const float* w_out, const float* D,
                                    const float* state, float* dx, float* d_w_bcdt_partial, float* d_w_dw_partial,
                                    float* d_w_hz_partial, float* d_w_out_partial, float* d_D_partial, void* ws,
                                    size_t ws_bytes, int B, int C, int N, int Hs, int stage, kmu_stream_t stream) {
    KMU_REQUIRE(stage >= 0 && stage <= 2, "hsmssd_bwd_stage: stage must be 0 (pass A), 1 (gate) or 2 (pass B)");
    return hsmssd_bwd_stages(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                             d_w_hz_partial, d_w_out_partial, d_D_partial, ws, ws_bytes, B, C, N, Hs, 1 << stage, stream);
}

// ---- packs made once per step (ops.PackCache): the composite weights packed by the CALLER, shared by forward and backward --------
extern "C" size_t kmu_hsmssd_pack_elems(int C, int groups) { return pack_x3_elems(C) * (size_t)(groups > 0 ? groups : 1); }

extern "C" int kmu_hsmssd_pack_x3(const float* w_bcdt, const float* w_dw, void* wpk, int C, int groups, kmu_stream_t stream) {
    KMU_REQUIRE(w_bcdt && w_dw && wpk, "hsmssd_pack_x3: null pointer");
    KMU_REQUIRE((C == 16 || C == 32 || C == 64) && groups >= 1, "hsmssd_pack_x3: C=%d (16/32/64), groups=%d", C, groups);
    hipStream_t st = (hipStream_t)stream;
    if (C == 16) hipLaunchKernelGGL(hsm_pack_x3_kernel<16>, dim3(48, groups), dim3(256), 0, st, w_bcdt, w_dw, (unsigned short*)wpk);
    else if (C == 32) hipLaunchKernelGGL(hsm_pack_x3_kernel<32>, dim3(48, groups), dim3(256), 0, st, w_bcdt, w_dw, (unsigned short*)wpk);
    else hipLaunchKernelGGL(hsm_pack_x3_kernel<64>, dim3(48, groups), dim3(256), 0, st, w_bcdt, w_dw, (unsigned short*)wpk);
    return kmu::launch_status("hsmssd_pack_x3");
}

// job table as kmu_conv_pack_job: `table` is host memory, kmu_pack_job_bytes() per record
extern "C" int kmu_hsm_pack_job(void* table, int index, const float* w_bcdt, const float* w_dw, void* wpk, int C, int groups) {
    KMU_REQUIRE(table && index >= 0 && w_bcdt && w_dw && wpk, "hsm_pack_job: bad arguments");
    KMU_REQUIRE((C == 16 || C == 32 || C == 64) && groups >= 1, "hsm_pack_job: C=%d (16/32/64), groups=%d", C, groups);
    HsmPackJob j{};
    j.w_bcdt = w_bcdt, j.w_dw = w_dw, j.wpk = (unsigned short*)wpk, j.C = C, j.groups = groups;
    reinterpret_cast<HsmPackJob*>(table)[index] = j;
    return 0;
}

extern "C" int kmu_hsm_pack_multi(const void* device_table, int njobs, kmu_stream_t stream) {
    KMU_REQUIRE(device_table && njobs > 0 && njobs <= 65535, "hsm_pack_multi: bad arguments");
    hipLaunchKernelGGL(hsm_pack_multi_kernel, dim3(48, njobs), dim3(256), 0, (hipStream_t)stream, (const HsmPackJob*)device_table);
    return kmu::launch_status("hsm_pack_multi");
}

// the *_g stage entry points with the pack handed in (wpk from kmu_hsmssd_pack_x3 / kmu_hsm_pack_multi; NULL = pack here)
extern "C" int kmu_hsmssd_fwd_stage_x3_pk(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz,
                                          const float* w_out, const float* D, float* y, float* h, float* state, void* ws,
                                          size_t ws_bytes, int B, int C, int N, int Hs, int stage, int groups, const void* wpk,
                                          kmu_stream_t stream) {
    KMU_REQUIRE(stage >= 0 && stage <= 2, "hsmssd_fwd_stage_x3_pk: stage must be 0 (pass 1), 1 (gate) or 2 (pass 2)");
    return hsmssd_fwd_stages(x, w_bcdt, w_dw, w_hz, w_out, D, y, h, state, ws, ws_bytes, B, C, N, Hs, 1 << stage, stream, true, groups, wpk);
}
extern "C" int kmu_hsmssd_bwd_stage_x3_pk(const float* x, const float* dy, const float* dh, const float* w_bcdt,
                                          const float* w_dw, const float* w_hz, const float* w_out, const float* D,
                                          const float* state, float* dx, float* d_w_bcdt_partial, float* d_w_dw_partial,
                                          float* d_w_hz_partial, float* d_w_out_partial, float* d_D_partial, void* ws,
                                          size_t ws_bytes, int B, int C, int N, int Hs, int stage, int groups, const void* wpk,
                                          kmu_stream_t stream) {
    KMU_REQUIRE(stage >= 0 && stage <= 2, "hsmssd_bwd_stage_x3_pk: stage must be 0 (pass A), 1 (gate) or 2 (pass B)");
    return hsmssd_bwd_stages(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial,
                             d_w_hz_partial, d_w_out_partial, d_D_partial, ws, ws_bytes, B, C, N, Hs, 1 << stage, stream, true, groups, wpk);
}

// ---- round 4: the K2 backward behind kmu_mixer_fwd_stage (csrc/hsmssd_bwdc.inc) -------------------------------------------------
// The C rows handled as a per-sample dense convolution with the M_b the forward left in `state`.  stage 0: G = dy (*) x per sample
// (replaces pass A), 1: the C rows' contractions of G (dh2, d W_C, d w_dw[C rows]), 2: gate, 3: pass B ({B, dt} rows + dx of the C rows).
extern "C" size_t kmu_mixer_bwd_ws_bytes(int B, int C, int N, int Hs) {
    (void)N;
    return C == 16 ? bwdc_ws_bytes<16>(B, Hs) : (C == 32 ? bwdc_ws_bytes<32>(B, Hs) : bwdc_ws_bytes<64>(B, Hs));
}
extern "C" int kmu_mixer_bwd_partials(int B, int C) { return B * (C / (C == 16 ? 2 : 4)); }
// tools / tests: pass B of kmu_mixer_bwd_stage as hsm_bwd_passB(cmode) (0) or hsm_bwd_passB2 (1: contractions on the bf16 matrix core;
// needs the weight pack and C in {16, 32}); -1 restores the default choice
extern "C" void kmu_mixer_debug_passb(int mode) { g_passb2 = mode; }

extern "C" int kmu_mixer_bwd_stage(const float* x, const float* dy, const float* dh, const float* w_bcdt, const float* w_dw, const float* w_hz,
                                   const float* w_out, const float* D, const float* state, float* dx, float* d_w_bcdt_partial,
                                   float* d_w_dw_partial, float* d_w_hz_partial, float* d_w_out_partial, float* d_D_partial,
                                   float* d_wc_partial, float* d_dwc_partial, void* ws, size_t ws_bytes, int B, int C, int N, int Hs, int stage,
                                   int groups, const void* wpk, kmu_stream_t stream) {
    KMU_REQUIRE(stage >= 0 && stage <= 3, "mixer_bwd_stage: stage must be 0 (correlation), 1 (C rows), 2 (gate) or 3 (pass B)");
    KMU_REQUIRE(x && dy && w_bcdt && w_dw && w_hz && w_out && D && state && dx && d_w_bcdt_partial && d_w_dw_partial && d_w_hz_partial &&
                    d_w_out_partial && d_D_partial && d_wc_partial && d_dwc_partial && ws,
                "mixer_bwd_stage: null pointer");
    KMU_REQUIRE(N == NS && (C == 16 || C == 32 || C == 64), "mixer_bwd_stage: C=%d, N=%d not instantiated (C in {16,32,64}, N = 64)", C, N);
    KMU_REQUIRE(B >= 1 && Hs >= 1 && groups >= 1 && B % groups == 0, "mixer_bwd_stage: B=%d, Hs=%d, groups=%d", B, Hs, groups);
    KMU_REQUIRE(ws_bytes >= kmu_mixer_bwd_ws_bytes(B, C, N, Hs), "mixer_bwd_stage: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
#define KMU_BWDC_GO(CC)                                                                                                        \
    return bwdc_impl<CC>(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dx, d_w_bcdt_partial, d_w_dw_partial, d_w_hz_partial, \
                         d_w_out_partial, d_D_partial, d_wc_partial, d_dwc_partial, w, B, Hs, 1 << stage, st, groups, wpk)
    if (C == 16) KMU_BWDC_GO(16);
    if (C == 32) KMU_BWDC_GO(32);
    KMU_BWDC_GO(64);
#undef KMU_BWDC_GO
}

// ---- round 4: LayerNorm1D + HSMSSD forward in two launches (csrc/hsmssd_v2.inc) ---------------------------------------------
extern "C" size_t kmu_mixer_fwd_ws_bytes(int B, int C, int N, int Hs) {
    (void)N;
    return v2_ws_bytes(B, C, Hs);
}

extern "C" int kmu_mixer_fwd_stage(const float* x, const float* ln_weight, const float* ln_bias, float eps, const float* w_dw,
                                   const float* w_hz, const float* w_out, const float* D, const void* wpk, float* y, float* h, float* state,
                                   float* xn, float* rstd_mean, void* ws, size_t ws_bytes, unsigned int* tickets, int B, int C, int N,
                                   int Hs, int stage, int groups, kmu_stream_t stream) {
    KMU_REQUIRE(x && w_dw && w_hz && w_out && D && wpk && y && h && state && ws && tickets, "mixer_fwd: null pointer");
    KMU_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), "mixer_fwd: LayerNorm weight and bias come together (both NULL: x is used as it is)");
    KMU_REQUIRE(!xn || rstd_mean, "mixer_fwd: xn comes with rstd_mean");
    KMU_REQUIRE(ln_weight || (!xn && !rstd_mean), "mixer_fwd: xn / rstd_mean are outputs of the LayerNorm prologue only");
    KMU_REQUIRE(N == NS, "mixer_fwd: state_dim=%d unsupported (kernels are built for 64)", N);
    KMU_REQUIRE(C == 16 || C == 32 || C == 64, "mixer_fwd: C=%d unsupported (16/32/64)", C);
    KMU_REQUIRE(B > 0 && B <= 65535 && Hs > 0 && groups >= 1 && B % groups == 0, "mixer_fwd: bad dims");
    KMU_REQUIRE(stage == 0 || stage == 1, "mixer_fwd: stage must be 0 (pass 1 + gate) or 1 (pass 2)");
    KMU_REQUIRE(ws_bytes >= v2_ws_bytes(B, C, Hs), "mixer_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (C == 16)
        return v2_fwd_impl<16>(x, ln_weight, ln_bias, eps, wpk, w_dw, w_hz, w_out, D, y, h, state, xn, rstd_mean, (float*)ws, tickets, B, Hs,
                               1 << stage, groups, st);
    if (C == 32)
        return v2_fwd_impl<32>(x, ln_weight, ln_bias, eps, wpk, w_dw, w_hz, w_out, D, y, h, state, xn, rstd_mean, (float*)ws, tickets, B, Hs,
                               1 << stage, groups, st);
    return v2_fwd_impl<64>(x, ln_weight, ln_bias, eps, wpk, w_dw, w_hz, w_out, D, y, h, state, xn, rstd_mean, (float*)ws, tickets, B, Hs,
                           1 << stage, groups, st);
}

// tools only: force the tile height of pass 1 (rows per lane group: 1, 2 or 4; 0 = choose by grid size)
extern "C" void kmu_mixer_debug_rows(int rows) {
    g_v2_dbg = rows >> 8;                      // bit 8: skip the gate tail; bit 9: skip pass 1's main loop; bit 10: phase stamps into `state`
                                               // (timing experiments: wrong results)
    rows &= 255;                               // H | 16 (8-wave workgroups) | 32 (wide tiles: 4H x 32, needs 16)
    const int H = rows & 15;
    const bool ok = (rows & 32) ? ((rows & 16) && (H == 1 || H == 2 || H == 4)) : ((rows & 16) ? (H == 1 || H == 2) : (H == 1 || H == 2 || H == 4));
    g_v2_cfg_override = ok ? rows : 0;
}
