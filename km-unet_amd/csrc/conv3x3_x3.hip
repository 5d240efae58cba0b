// 3x3 / stride 1 / pad 1 convolutions of gfx950's bf16 matrix core with SPLIT-bf16 ("bf16x3") operands:
//
//   K1  KANConv2d(x) = conv3x3(Phi(x), W')            convKAN/KANConv2Dlayers.py:15-37 + KANlayers.py:577-660
//   plain nn.Conv2d(Cin, Cout, 3, padding=1)          KM_UNetV3_SH.py:375 (conv_f), :430-446 (dec2 / dec3), :287-311
//                                                     (MultiScaleFusion), DAGEM_md.py:43 (offset_conv)
//
// Why: the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, csrc/kan_conv2d.hip) runs at the fp32 vector rate, 1/16 of the bf16
// matrix rate.  Every fp32 operand is split  v = hi + lo,  hi = bf16(v), lo = bf16(v - hi)  (16 significant bits
// together) and each product is accumulated in fp32 as  lo.hi + hi.lo + hi.hi  on v_mfma_f32_16x16x32_bf16: three MFMAs
// at 16x the rate = 5.3x the fp32-MFMA ceiling, relative error of a product ~2^-16 (the dropped lo.lo term and the
// truncation of lo), i.e. ~1e-5 of the result after a K = 1296..5184 accumulation -- two orders inside the 1e-3 parity
// bound; the exact-fp32 kernels stay as the tested reference for these.
//
// Structure (one 256-thread workgroup = TH x TW output pixels, all Cout):
//   for every chunk of 32 features (KAN: 4 input channels x 8 spline bases, then 32 channels of SiLU(x); plain: 32 input
//   channels):   fill  F[halo position][32 features] as (hi, lo) bf16 in LDS   (Phi evaluated ONCE per input element,
//                                                                                out-of-image positions hold Phi(0))
//                for the 9 taps: A = 16 pixels x 32 features straight from F (one ds_read_b128 per lane and half),
//                                B = packed weights (fragment order, L2-resident), 3 MFMAs per (pixel tile, channel tile).
// LDS position stride = 160 B = 32 x 5: with an odd multiple of 32 B the four 16-lane groups of a ds_read_b128 fragment read
// (16 positions x 16 B per k-group) cover all 64 banks exactly once (MI355X_MICROARCH.md, LDS table).
#include "common.h"

using kmu::floatx4;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

constexpr int PSTR = 160;        // bytes per halo position: hi[32] | lo[32] | 32 B pad
constexpr int TABW = 12;         // per-span table: u[i-2..i+3], then the six knot-difference reciprocals of the de Boor triangle

enum Mode { MODE_KAN = 0, MODE_PLAIN = 1, MODE_PLAIN_DGRAD = 2 };   // the last one is a weight-pack mode only

__host__ __device__ inline int n_chunks(int mode, int Cin) {
    return mode == MODE_KAN ? Cin / 4 + (Cin + 31) / 32 : (Cin + 31) / 32;
}

// ---- split one fp32 into (hi, lo) bf16 bit patterns -------------------------------------------------------------------
__device__ __forceinline__ void split(float v, unsigned& hi, unsigned& lo) {
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    hi = (unsigned)__builtin_bit_cast(unsigned short, h);
    lo = (unsigned)__builtin_bit_cast(unsigned short, l);
}

// ---- weight packs: wp[((chunk*9 + tap)*NT + nt)*2 + {hi,lo}][lane][8]; lane l holds B[k = 8(l>>4)+j][col = l&15] -------------
__device__ __forceinline__ float kan_wprime(const float* bw, const float* sw, const float* sc, int Cin, int Cout, int o, int c,
                                            int j, int tap) {   // j = 0: SiLU branch, 1..8: spline bases (as csrc/kan_conv2d.hip)
    if (o >= Cout || c >= Cin) return 0.f;
    const size_t f = (size_t)o * (Cin * 9) + c * 9 + tap;
    return j == 0 ? bw[f] : sw[f * 8 + (j - 1)] * sc[f];
}

__device__ __forceinline__ void pack_x3_body(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2,
                                             unsigned short* __restrict__ wp, int mode, int Cin, int Cout, int NT, int NCH, int T,
                                             size_t first, size_t stride) {
    const size_t total = (size_t)NCH * T * NT * 64 * 8;
    const int nspl = mode == MODE_KAN ? Cin / 4 : 0;
    for (size_t e = first; e < total; e += stride) {
        const int j = e & 7, lane = (e >> 3) & 63;
        size_t t = e >> 9;
        const int nt = t % NT;
        t /= NT;
        const int tap = t % T, chunk = (int)(t / T);
        const int k = 8 * (lane >> 4) + j, o = nt * 16 + (lane & 15);
        float v;
        if (mode == MODE_KAN) {
            if (chunk < nspl) v = kan_wprime(w0, w1, w2, Cin, Cout, o, chunk * 4 + (k >> 3), 1 + (k & 7), tap);
            else v = kan_wprime(w0, w1, w2, Cin, Cout, o, (chunk - nspl) * 32 + k, 0, tap);
        } else if (mode == MODE_PLAIN) {
            const int c = chunk * 32 + k;
            v = (o < Cout && c < Cin) ? w0[((size_t)o * Cin + c) * T + tap] : 0.f;
        } else {
            // input-gradient pack of a plain conv: dx = conv3x3(dy, W2), W2[c][o][tap'] = W[o][c][8 - tap'] -- here the kernel's
            // "Cin" is the layer's Cout (k = layer output channel) and its "Cout" the layer's Cin (n = layer input channel)
            const int lo = chunk * 32 + k, lc = o;
            v = (lo < Cin && lc < Cout) ? w0[((size_t)lo * Cout + lc) * T + (T - 1 - tap)] : 0.f;
        }
        unsigned hi, lo;
        split(v, hi, lo);
        const size_t base = ((((size_t)chunk * T + tap) * NT + nt) * 2) * 512 + lane * 8 + j;
        wp[base] = (unsigned short)hi;
        wp[base + 512] = (unsigned short)lo;
    }
}
__global__ void pack_x3_kernel(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2,
                               unsigned short* __restrict__ wp, int mode, int Cin, int Cout, int NT, int NCH, int T) {
    pack_x3_body(w0, w1, w2, wp, mode, Cin, Cout, NT, NCH, T, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

// ---- per-span table of the layer's knot vector (KANlayers.py:526-535), spans i = 0..10: U[i] <= x < U[i+1] ------------------
__device__ __forceinline__ float ext_knot(const float* knots, int idx) {      // idx in [-3, 14]: 3 virtual knots either side
    if (idx < 0) return knots[0] + (float)idx * (knots[1] - knots[0]);
    if (idx > 11) return knots[11] + (float)(idx - 11) * (knots[11] - knots[10]);
    return knots[idx];
}
__device__ __forceinline__ void load_span_table(const float* __restrict__ knots, float* tab, float* kn, int tid) {
    if (tid < 18) kn[tid] = ext_knot(knots, tid - 3);
    if (tid < 11) {
        float u[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) u[d] = ext_knot(knots, tid - 2 + d);     // u[2] = U[i], u[3] = U[i+1]
        float* t = tab + tid * TABW;
#pragma unroll
        for (int d = 0; d < 6; ++d) t[d] = u[d];
        t[6] = 1.f / (u[3] - u[2]);
        t[7] = 1.f / (u[3] - u[1]);
        t[8] = 1.f / (u[4] - u[2]);
        t[9] = 1.f / (u[3] - u[0]);
        t[10] = 1.f / (u[4] - u[1]);
        t[11] = 1.f / (u[5] - u[2]);
    }
}

// The 8 cubic B-spline values of x as two 128-bit vectors of bf16 (hi parts, lo parts).  Same half-open span test as
// KANlayers.py:593 on the layer's real knots; the triangle is csrc/kan_conv2d.hip's with the knot-difference divisions
// replaced by multiplications with their tabulated reciprocals (<= 1 ulp apart per step).
__device__ __forceinline__ void spline_bf16x8(float x, const float* tab, const float* kn, float u0, float inv_h, uintx4& vhi,
                                              uintx4& vlo) {
    int i = (int)floorf((x - u0) * inv_h);
    i = min(max(i, -1), 11);
    if (x < kn[i + 3]) --i;                    // kn[idx + 3] = U[idx]
    else if (x >= kn[i + 4]) ++i;
    const bool valid = (i >= 0) && (i <= 10) && (x >= kn[3]) && (x < kn[14]);
    const int ic = min(max(i, 0), 10);
    const floatx4 ta = *reinterpret_cast<const floatx4*>(tab + ic * TABW);
    const floatx4 tb = *reinterpret_cast<const floatx4*>(tab + ic * TABW + 4);
    const floatx4 tc = *reinterpret_cast<const floatx4*>(tab + ic * TABW + 8);
    const float l1 = x - ta[2], r1 = ta[3] - x, l2 = x - ta[1], r2 = tb[0] - x, l3 = x - ta[0], r3 = tb[1] - x;
    const float n0 = r1 * tb[2], n1 = l1 * tb[2];
    float t = n0 * tb[3];
    const float m0 = r1 * t;
    float sv = l2 * t;
    t = n1 * tc[0];
    const float m1 = sv + r2 * t, m2 = l1 * t;
    const float t0 = m0 * tc[1], t1 = m1 * tc[2], t2 = m2 * tc[3];
    const float c0 = r1 * t0, c1 = l3 * t0 + r2 * t1, c2 = l2 * t1 + r3 * t2, c3 = l1 * t2;     // N_{i-3..i,3}(x)
    unsigned h0, h1, h2, h3, q0, q1, q2, q3;
    split(c0, h0, q0);
    split(c1, h1, q1);
    split(c2, h2, q2);
    split(c3, h3, q3);
    unsigned long long qh = (unsigned long long)(h0 | (h1 << 16)) | ((unsigned long long)(h2 | (h3 << 16)) << 32);
    unsigned long long ql = (unsigned long long)(q0 | (q1 << 16)) | ((unsigned long long)(q2 | (q3 << 16)) << 32);
    if (!valid) qh = ql = 0ull;
    // place the quad at basis slots a = i-3 .. i of the 8-slot vector (slots outside 0..7 fall off either end)
    const int s = ic - 3;                      // -3 .. 7, in 16-bit slots
    unsigned long long hl, hh, ll, lh;
    if (s < 0) {
        hl = qh >> (16 * -s), hh = 0ull, ll = ql >> (16 * -s), lh = 0ull;
    } else if (s == 0) {
        hl = qh, hh = 0ull, ll = ql, lh = 0ull;
    } else if (s < 4) {
        hl = qh << (16 * s), hh = qh >> (64 - 16 * s), ll = ql << (16 * s), lh = ql >> (64 - 16 * s);
    } else {
        hl = 0ull, hh = qh << (16 * (s - 4)), ll = 0ull, lh = ql << (16 * (s - 4));
    }
    vhi = uintx4{(unsigned)hl, (unsigned)(hl >> 32), (unsigned)hh, (unsigned)(hh >> 32)};
    vlo = uintx4{(unsigned)ll, (unsigned)(ll >> 32), (unsigned)lh, (unsigned)(lh >> 32)};
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

template <int TH, int TW, int KSZ = 3>
struct Geo {
    static constexpr int R = KSZ / 2, RS = TW + 2 * R, HT = (TH + 2 * R) * RS;
    static constexpr int FBYTES = HT * PSTR;
    static constexpr int LDS_BYTES = FBYTES + (11 * TABW + 32) * 4;
};

// =====================================================================================================================
// forward
// =====================================================================================================================
template <int MODE, int KSZ, int TH, int TW, int WM, int WN, int NREP>
__global__ __launch_bounds__(256) void conv3x3_x3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ knots,
                                                             const bf16x8* __restrict__ wp, const float* __restrict__ bias,
                                                             const float* __restrict__ residual, float* __restrict__ y, int Cin,
                                                             int Cout, int H, int W, int NT, int tilesX, int relu) {
    using G = Geo<TH, TW, KSZ>;
    constexpr int RS = G::RS, HT = G::HT, MF = TH * TW / 16, MREP = MF / WM, SPR = TW / 16, R = G::R, T = KSZ * KSZ;
    static_assert(MODE != MODE_KAN || KSZ == 3, "KANConv2d is 3x3");
    static_assert(WM * WN == 4 && MF % WM == 0, "4 waves over the pixel fragments / channel tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* F = smem;
    float* tab = reinterpret_cast<float*>(smem + G::FBYTES);
    float* kn = tab + 11 * TABW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ty0 = (blockIdx.x / tilesX) * TH, tx0 = (blockIdx.x % tilesX) * TW;
    const int b = blockIdx.z;
    const int nt0 = (blockIdx.y * WN + wn) * NREP;
    const int li = lane & 15, lg = lane >> 4;

    float u0 = 0.f, inv_h = 0.f;
    if (MODE == MODE_KAN) {
        load_span_table(knots, tab, kn, tid);
        u0 = knots[0];
        inv_h = 11.f / (knots[11] - knots[0]);
    }

    floatx4 acc[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[m][n] = floatx4{0.f, 0.f, 0.f, 0.f};

    int aoff[MREP];   // byte offset of this lane's A row (tap (0,0)) for each of its pixel fragments
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int seg = wm * MREP + m;
        aoff[m] = ((seg / SPR) * RS + (seg % SPR) * 16 + li) * PSTR + lg * 16;
    }

    const float* xb = x + (size_t)b * Cin * H * W;
    const int nspl = MODE == MODE_KAN ? Cin / 4 : 0;
    const int NCH = nspl + (Cin + 31) / 32;
    for (int ch = 0; ch < NCH; ++ch) {
        __syncthreads();   // previous chunk's fragment reads done (and the span table visible on the first trip)
        // this chunk's packed weights (L2-resident): requested before the feature tile is computed, so that their latency
        // hides under the fill instead of stalling the first MFMA of every tap
        const bf16x8* wpc = wp + (((size_t)ch * T) * NT + nt0) * 128 + lane;
        constexpr int TPRE = KSZ == 3 ? 9 : 1;          // 3x3: all taps' fragments ahead of the fill; 5x5 / 7x7: per tap below
        bf16x8 bh[TPRE][NREP], bl[TPRE][NREP];
        if (KSZ == 3) {
#pragma unroll
            for (int tap = 0; tap < TPRE; ++tap)
#pragma unroll
                for (int n = 0; n < NREP; ++n) {
                    bh[tap][n] = wpc[((size_t)tap * NT + n) * 128];
                    bl[tap][n] = wpc[((size_t)tap * NT + n) * 128 + 64];
                }
        }
        constexpr int ITEMS = (4 * HT + 255) / 256;
        if (MODE == MODE_KAN && ch < nspl) {
            float xv[ITEMS];
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {     // all loads first, then the arithmetic
                const int e = it * 256 + tid;
                const int q = e / HT, pos = e - q * HT;
                const int hy = pos / RS, hx = pos - hy * RS;
                const int gy = ty0 + hy - R, gx = tx0 + hx - R, c = ch * 4 + q;
                xv[it] = 0.f;   // out-of-image taps see x = 0 -> Phi(0) (reference: F.unfold zero padding)
                if (e < 4 * HT && gy >= 0 && gy < H && gx >= 0 && gx < W) xv[it] = xb[((size_t)c * H + gy) * W + gx];
            }
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int e = it * 256 + tid;
                if (e < 4 * HT) {
                    const int q = e / HT, pos = e - q * HT;
                    uintx4 vh, vl;
                    spline_bf16x8(xv[it], tab, kn, u0, inv_h, vh, vl);
                    unsigned char* dst = F + pos * PSTR + q * 16;
                    *reinterpret_cast<uintx4*>(dst) = vh;
                    *reinterpret_cast<uintx4*>(dst + 64) = vl;
                }
            }
        } else {
            const int c0 = (ch - nspl) * 32;
#pragma unroll 2
            for (int e = tid; e < 4 * HT; e += 256) {
                const int q = e / HT, pos = e - q * HT;
                const int hy = pos / RS, hx = pos - hy * RS;
                const int gy = ty0 + hy - R, gx = tx0 + hx - R;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = c0 + q * 8 + j;
                    v[j] = (in && c < Cin) ? xb[((size_t)c * H + gy) * W + gx] : 0.f;
                }
                unsigned hv[8], lv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) split(MODE == MODE_KAN ? silu_f(v[j]) : v[j], hv[j], lv[j]);
                unsigned char* dst = F + pos * PSTR + q * 16;
                *reinterpret_cast<uintx4*>(dst) = uintx4{hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), hv[4] | (hv[5] << 16), hv[6] | (hv[7] << 16)};
                *reinterpret_cast<uintx4*>(dst + 64) = uintx4{lv[0] | (lv[1] << 16), lv[2] | (lv[3] << 16), lv[4] | (lv[5] << 16), lv[6] | (lv[7] << 16)};
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            const int toff = ((tap / KSZ) * RS + (tap % KSZ)) * PSTR;
            const int tb = KSZ == 3 ? tap : 0;
            if (KSZ != 3) {
#pragma unroll
                for (int n = 0; n < NREP; ++n) {
                    bh[0][n] = wpc[((size_t)tap * NT + n) * 128];
                    bl[0][n] = wpc[((size_t)tap * NT + n) * 128 + 64];
                }
            }
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(F + aoff[m] + toff);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(F + aoff[m] + toff + 64);
#pragma unroll
                for (int n = 0; n < NREP; ++n) {
                    // (the empty asm keeps the fragments live across the MFMA: vdst must never land on srcA / srcB -- see
                    //  csrc/hsmssd_x3.inc and tools/check_mfma_overlap.py)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[tb][n], acc[m][n], 0, 0, 0);
                    asm volatile("" ::"v"(al), "v"(bh[tb][n]));
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[tb][n], acc[m][n], 0, 0, 0);
                    asm volatile("" ::"v"(ah), "v"(bl[tb][n]));
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[tb][n], acc[m][n], 0, 0, 0);
                    asm volatile("" ::"v"(ah), "v"(bh[tb][n]));
                }
            }
        }
    }

    // epilogue: lane holds 4 consecutive pixels (rows lg*4 + r of the C tile) of output channel nt*16 + li
    const bool vec_ok = (W & 3) == 0;
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int seg = wm * MREP + m;
        const int gy = ty0 + seg / SPR, px0 = tx0 + (seg % SPR) * 16 + lg * 4;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
            const int o = (nt0 + n) * 16 + li;
            if (o >= Cout || gy >= H || px0 >= W) continue;
            const size_t idx = (((size_t)b * Cout + o) * H + gy) * W + px0;
            floatx4 v = acc[m][n];
            if (bias) v += bias[o];
            if (vec_ok && px0 + 3 < W) {
                if (residual) v += *reinterpret_cast<const floatx4*>(residual + idx);
                if (relu) v = floatx4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                *reinterpret_cast<floatx4*>(y + idx) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (px0 + r < W) {
                        float s = v[r];
                        if (residual) s += residual[idx + r];
                        if (relu) s = fmaxf(s, 0.f);
                        y[idx + r] = s;
                    }
            }
        }
    }
}

template <int MODE, int TH, int TW, int WM, int WN, int NREP, int KSZ = 3>
int launch_fwd(const float* x, const float* knots, const void* wp, const float* bias, const float* residual, float* y, int B,
               int Cin, int Cout, int H, int W, int relu, hipStream_t st) {
    using G = Geo<TH, TW, KSZ>;
    const int NT = kmu::cdiv(Cout, 16);
    const int tilesX = kmu::cdiv(W, TW), tilesY = kmu::cdiv(H, TH);
    dim3 grid(tilesX * tilesY, NT / (WN * NREP), B);
    auto kern = conv3x3_x3_fwd_kernel<MODE, KSZ, TH, TW, WM, WN, NREP>;
    KMU_MAX_LDS(kern, G::LDS_BYTES);
    hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, st, x, knots, (const bf16x8*)wp, bias, residual, y, Cin, Cout, H, W, NT,
                       tilesX, relu);
    return kmu::launch_status(MODE == MODE_KAN ? "kan_conv2d_fwd_x3" : "conv3x3_fwd_x3");
}

// tile / wave layout: all Cout channel tiles in one workgroup (the feature tile is evaluated once), enough workgroups to fill
// 256 CUs where the image allows it
static int g_conv_split = 0;
template <int MODE>
int dispatch_fwd(const float* x, const float* knots, const void* wp, const float* bias, const float* residual, float* y, int B,
                 int Cin, int Cout, int H, int W, int relu, hipStream_t st) {
    const int NT = kmu::cdiv(Cout, 16);
    const long px = (long)B * H * W;
    const bool big = px >= 65536 && W >= 32;
    // small images (32 x 32 and below at B = 8: 128 tiles of 4 x 16): the output-channel tiles go to separate workgroups (grid.y),
    // each re-evaluating the feature tile, when that is what it takes to give every CU one (g_split: 0 = automatic, 1 = never, 2 = always)
    const bool split = !big && NT >= 2 && (g_conv_split == 2 || (g_conv_split == 0 && px / 64 * 2 <= 384));
    if (split) {
        if (NT % 2 == 0 && px / 64 * (NT / 2) >= 256 && g_conv_split != 2)
            return launch_fwd<MODE, 4, 16, 2, 2, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);                // grid.y = NT / 2
        return launch_fwd<MODE, 4, 16, 4, 1, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);                    // grid.y = NT
    }
    if (NT == 1) {
        if (big) return launch_fwd<MODE, 8, 32, 4, 1, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);
        return launch_fwd<MODE, 4, 16, 4, 1, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);
    }
    if (NT == 2) {
        if (big) return launch_fwd<MODE, 8, 32, 2, 2, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);
        return launch_fwd<MODE, 4, 16, 2, 2, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);
    }
    if (NT == 4) {
        if (big) return launch_fwd<MODE, 8, 16, 1, 4, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);
        return launch_fwd<MODE, 4, 16, 1, 4, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);
    }
    if (NT % 4 == 0) return launch_fwd<MODE, 4, 16, 1, 4, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);  // grid.y = NT/4
    if (NT % 2 == 0) return launch_fwd<MODE, 4, 16, 2, 2, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);
    return launch_fwd<MODE, 4, 16, 4, 1, 1>(x, knots, wp, bias, residual, y, B, Cin, Cout, H, W, relu, st);                    // grid.y = NT
}


// =====================================================================================================================
// K1 input gradient:  G[pix][c][j] = sum_{o,tap} dY[o][pix - tap + 1] W'[o][c][j][tap]   (matrix core, as the forward on dY)
//                     dX[pix][c]   = sum_j dPhi_j(x[pix][c]) G[pix][c][j]               (epilogue)
// N tiles are (basis j, 16 input channels): a lane holds all 9 G_j of ITS channel for 4 pixels, so the epilogue is lane-local.
// Pack: wpd[(((chunk*9 + tap)*9 + j)*CT + ct)*2 + {hi,lo}][lane][8], k = output channel 32 chunk + 8 (l>>4) + jj, col = channel
// ct*16 + (l&15), value W'[o][c][j][8 - tap] (the flipped tap turns the transposed convolution into a plain one on dY).
// =====================================================================================================================
__device__ __forceinline__ void pack_kan_dgrad_x3_body(const float* __restrict__ bw, const float* __restrict__ sw,
                                                       const float* __restrict__ sc, unsigned short* __restrict__ wp, int Cin, int Cout,
                                                       int CT, int NCH, size_t first, size_t stride) {
    const size_t total = (size_t)NCH * 9 * 9 * CT * 64 * 8;
    for (size_t e = first; e < total; e += stride) {
        const int jj = e & 7, lane = (e >> 3) & 63;
        size_t t = e >> 9;
        const int ct = t % CT;
        t /= CT;
        const int j = t % 9;
        t /= 9;
        const int tap = t % 9, chunk = (int)(t / 9);
        const int o = chunk * 32 + 8 * (lane >> 4) + jj, c = ct * 16 + (lane & 15);
        const float v = kan_wprime(bw, sw, sc, Cin, Cout, o, c, j, 8 - tap);
        unsigned hi, lo;
        split(v, hi, lo);
        const size_t base = (((((size_t)chunk * 9 + tap) * 9 + j) * CT + ct) * 2) * 512 + lane * 8 + jj;
        wp[base] = (unsigned short)hi;
        wp[base + 512] = (unsigned short)lo;
    }
}
__global__ void pack_kan_dgrad_x3_kernel(const float* __restrict__ bw, const float* __restrict__ sw, const float* __restrict__ sc,
                                         unsigned short* __restrict__ wp, int Cin, int Cout, int CT, int NCH) {
    pack_kan_dgrad_x3_body(bw, sw, sc, wp, Cin, Cout, CT, NCH, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

// ---- every pack of a training step in ONE launch: blockIdx.y walks a table of jobs that lives in device memory (built once per
// model by kmu_conv_pack_job, see ops.PackCache); the jobs' sources are the parameters themselves, so a captured hipGraph re-packs
// the current weights at every replay.  63 pack launches per step (29 conv + 4 KAN dgrad here, 30 HSMSSD) were each the head of
// a dependent chain.
struct ConvPackJob {
    const float* w0;
    const float* w1;
    const float* w2;
    unsigned short* wp;
    int kind, mode, Cin, Cout, NT, NCH, T, pad;      // kind 0: pack_x3 (mode, Cin, Cout as the kernel sees them), 1: KAN dgrad (NT = CT)
};
static_assert(sizeof(ConvPackJob) == 64, "job records are 64 bytes (kmu_pack_job_bytes)");
__global__ __launch_bounds__(256) void conv_pack_multi_kernel(const ConvPackJob* __restrict__ jobs) {
    const ConvPackJob j = jobs[blockIdx.y];
    const size_t first = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (j.kind == 0) pack_x3_body(j.w0, j.w1, j.w2, j.wp, j.mode, j.Cin, j.Cout, j.NT, j.NCH, j.T, first, stride);
    else pack_kan_dgrad_x3_body(j.w0, j.w1, j.w2, j.wp, j.Cin, j.Cout, j.NT, j.NCH, first, stride);
}

// dPhi/dx of the 9 features (SiLU', 8 cubic B-spline derivatives): the same span search and triangle as spline_bf16x8;
// N'_{a,3} = 3 [ N_{a,2}/(U_{a+3}-U_a) - N_{a+1,2}/(U_{a+4}-U_{a+1}) ] (csrc/kan_conv2d.hip)
__device__ __forceinline__ void kan_dphi(float x, const float* tab, const float* kn, float u0, float inv_h, float (&d)[9]) {
    const float sg = 1.f / (1.f + __expf(-x));
    d[0] = sg * (1.f + x * (1.f - sg));
    int i = (int)floorf((x - u0) * inv_h);
    i = min(max(i, -1), 11);
    if (x < kn[i + 3]) --i;
    else if (x >= kn[i + 4]) ++i;
    const bool valid = (i >= 0) && (i <= 10) && (x >= kn[3]) && (x < kn[14]);
    const int ic = min(max(i, 0), 10);
    const floatx4 ta = *reinterpret_cast<const floatx4*>(tab + ic * TABW);
    const floatx4 tb = *reinterpret_cast<const floatx4*>(tab + ic * TABW + 4);
    const floatx4 tc = *reinterpret_cast<const floatx4*>(tab + ic * TABW + 8);
    const float l1 = x - ta[2], r1 = ta[3] - x, l2 = x - ta[1], r2 = tb[0] - x;
    const float n0 = r1 * tb[2], n1 = l1 * tb[2];
    float t = n0 * tb[3];
    const float m0 = r1 * t;
    const float sv = l2 * t;
    t = n1 * tc[0];
    const float m1 = sv + r2 * t, m2 = l1 * t;
    const float t0 = m0 * tc[1], t1 = m1 * tc[2], t2 = m2 * tc[3];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int r = a - ic + 3;
        const float v = (r == 0) ? -t0 : (r == 1) ? (t0 - t1) : (r == 2) ? (t1 - t2) : (r == 3) ? t2 : 0.f;
        d[1 + a] = valid ? 3.f * v : 0.f;
    }
}

template <int TH, int TW>
__global__ __launch_bounds__(256) void kan_dgrad_x3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ knots, const bf16x8* __restrict__ wpd,
                                                           float* __restrict__ dx, int Cin, int Cout, int H, int W, int CT,
                                                           int tilesX) {
    using G = Geo<TH, TW>;
    constexpr int RS = G::RS, HT = G::HT, MF = TH * TW / 16, MREP = MF / 4, SPR = TW / 16;
    static_assert(MF % 4 == 0, "pixel fragments over 4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* F = smem;
    float* tab = reinterpret_cast<float*>(smem + G::FBYTES);
    float* kn = tab + 11 * TABW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty0 = (blockIdx.x / tilesX) * TH, tx0 = (blockIdx.x % tilesX) * TW;
    const int ct = blockIdx.y, b = blockIdx.z;
    const int li = lane & 15, lg = lane >> 4;

    load_span_table(knots, tab, kn, tid);
    const float u0 = knots[0], inv_h = 11.f / (knots[11] - knots[0]);

    floatx4 acc[MREP][9];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[m][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    int aoff[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int seg = wave * MREP + m;
        aoff[m] = ((seg / SPR) * RS + (seg % SPR) * 16 + li) * PSTR + lg * 16;
    }

    const float* dyb = dy + (size_t)b * Cout * H * W;
    const int NCH = (Cout + 31) / 32;
    for (int ch = 0; ch < NCH; ++ch) {
        __syncthreads();
        for (int e = tid; e < 4 * HT; e += 256) {       // dY halo tile, 32 output channels, (hi, lo) bf16
            const int q = e / HT, pos = e - q * HT;
            const int hy = pos / RS, hx = pos - hy * RS;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            unsigned hv[8], lv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int o = ch * 32 + q * 8 + j;
                split((in && o < Cout) ? dyb[((size_t)o * H + gy) * W + gx] : 0.f, hv[j], lv[j]);
            }
            unsigned char* dst = F + pos * PSTR + q * 16;
            *reinterpret_cast<uintx4*>(dst) = uintx4{hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), hv[4] | (hv[5] << 16), hv[6] | (hv[7] << 16)};
            *reinterpret_cast<uintx4*>(dst + 64) = uintx4{lv[0] | (lv[1] << 16), lv[2] | (lv[3] << 16), lv[4] | (lv[5] << 16), lv[6] | (lv[7] << 16)};
        }
        __syncthreads();
        const bf16x8* wpc = wpd + ((size_t)ch * 81 * CT + ct) * 128 + lane;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3) * RS + (tap % 3)) * PSTR;
            bf16x8 ah[MREP], al[MREP];
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(F + aoff[m] + toff);
                al[m] = *reinterpret_cast<const bf16x8*>(F + aoff[m] + toff + 64);
            }
            bf16x8 bh[9], bl[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                bh[j] = wpc[((size_t)(tap * 9 + j) * CT) * 128];
                bl[j] = wpc[((size_t)(tap * 9 + j) * CT) * 128 + 64];
            }
#pragma unroll
            for (int j = 0; j < 9; ++j)
#pragma unroll
                for (int m = 0; m < MREP; ++m) {
                    acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh[j], acc[m][j], 0, 0, 0);
                    asm volatile("" ::"v"(al[m]), "v"(bh[j]));
                    acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl[j], acc[m][j], 0, 0, 0);
                    asm volatile("" ::"v"(ah[m]), "v"(bl[j]));
                    acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh[j], acc[m][j], 0, 0, 0);
                    asm volatile("" ::"v"(ah[m]), "v"(bh[j]));
                }
        }
    }

    // epilogue: lane = input channel ct*16 + li, registers = 4 consecutive pixels
    const int c = ct * 16 + li;
    const bool vec_ok = (W & 3) == 0;
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int seg = wave * MREP + m;
        const int gy = ty0 + seg / SPR, px0 = tx0 + (seg % SPR) * 16 + lg * 4;
        if (c >= Cin || gy >= H || px0 >= W) continue;
        const size_t idx = (((size_t)b * Cin + c) * H + gy) * W + px0;
        float xv[4], out[4];
        if (vec_ok && px0 + 3 < W) {
            const floatx4 t4 = *reinterpret_cast<const floatx4*>(x + idx);
            xv[0] = t4[0], xv[1] = t4[1], xv[2] = t4[2], xv[3] = t4[3];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[r] = (px0 + r < W) ? x[idx + r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float d[9];
            kan_dphi(xv[r], tab, kn, u0, inv_h, d);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 9; ++j) sum += d[j] * acc[m][j][r];
            out[r] = sum;
        }
        if (vec_ok && px0 + 3 < W) {
            *reinterpret_cast<floatx4*>(dx + idx) = floatx4{out[0], out[1], out[2], out[3]};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (px0 + r < W) dx[idx + r] = out[r];
        }
    }
}


// =====================================================================================================================
// weight gradient:  dW'[o][f][tap] = sum_{b,pix} dY[o][pix] * F[pix + tap - 1][f]      (F = Phi(x) for K1, x for a plain conv)
// MFMA: rows = 16 output channels, columns = 16 features, K = 32 pixels; BOTH operands are read transposed
// (ds_read_b64_tr_b16) from (hi, lo) images -- dY as [pixel][16 channels], F as the forward kernel's [halo position][32 features]
// -- so every tap is just a row offset in the F image (the contraction validated in hsm_bwd_passB_x3's phase 2b).
// grid = (S pixel-tile splits, feature chunks, 16-channel output tiles); a workgroup walks its 4x32-pixel tiles (wave w = the
// 32-pixel k-step of tile row w) with 2 feature tiles x 9 taps of accumulators in registers, then the 4 waves' sums meet in LDS
// in a fixed order and ONE slab[(chunk*OT + ot)*S + s][ft][tap][f_local][o_local] goes to HBM (deterministic two-stage sum).
// =====================================================================================================================
constexpr int WG_TH = 4, WG_TW = 32, WG_DPS = 96;      // dY image: 16 channels hi (32 B) | lo (32 B) | 32 B pad

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p0, const unsigned char* p1) {
    typedef short shortx4 __attribute__((ext_vector_type(4)));
    typedef short shortx8 __attribute__((ext_vector_type(8)));
    const shortx4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) shortx4*)(p0));
    const shortx4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) shortx4*)(p1));
    return __builtin_bit_cast(bf16x8, shortx8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

template <int MODE, int KSZ>
__global__ __launch_bounds__(256) void conv3x3_x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ knots, float* __restrict__ slab, int B,
                                                               int Cin, int Cout, int H, int W, int OT, int S, int tilesX,
                                                               int tilesY) {
    using G = Geo<WG_TH, WG_TW, KSZ>;
    constexpr int RS = G::RS, HT = G::HT, R = G::R, T = KSZ * KSZ, TG = (T + 8) / 9;
    static_assert(MODE != MODE_KAN || KSZ == 3, "KANConv2d is 3x3");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* F = smem;
    float* tab = reinterpret_cast<float*>(smem + G::FBYTES);
    float* kn = tab + 11 * TABW;
    unsigned char* DY = smem + G::LDS_BYTES;              // [NPIX][WG_DPS]
    float* red = reinterpret_cast<float*>(smem);          // aliases F after the main loop: [4 waves][6 tiles][256]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // blockIdx.y = (feature chunk, group of 9 taps): a KxK kernel's K*K tap accumulators are dealt 9 per workgroup
    const int s = blockIdx.x, ch = blockIdx.y / TG, tg = blockIdx.y % TG, ot = blockIdx.z;

    float u0 = 0.f, inv_h = 0.f;
    if (MODE == MODE_KAN) {
        load_span_table(knots, tab, kn, tid);
        u0 = knots[0];
        inv_h = 11.f / (knots[11] - knots[0]);
    }
    const int nspl = MODE == MODE_KAN ? Cin / 4 : 0;

    floatx4 acc[2][9];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[ft][t] = floatx4{0.f, 0.f, 0.f, 0.f};

    // Mechanism addresses of the transposed reads (block row (li >> 2), columns 4 (li & 3)..): this wave's 32 pixels = tile row
    // `wave`, lane group lg covers pixels 8 lg + {0..3} and {4..7}
    const int px0 = 8 * lg + (li >> 2);
    const int a0 = (wave * WG_TW + px0) * WG_DPS + 8 * (li & 3), a1 = a0 + 4 * WG_DPS;
    const int f0 = (wave * RS + px0) * PSTR + 8 * (li & 3), f1 = f0 + 4 * PSTR;       // tap (0,0): halo position (wave, px0)

    const int ntiles = B * tilesY * tilesX;
    for (int tile = s; tile < ntiles; tile += S) {
        const int b = tile / (tilesY * tilesX), tr = tile % (tilesY * tilesX);
        const int ty0 = (tr / tilesX) * WG_TH, tx0 = (tr % tilesX) * WG_TW;
        const float* xb = x + (size_t)b * Cin * H * W;
        const float* dyb = dy + (size_t)b * Cout * H * W;
        __syncthreads();       // previous tile's reads done (and the span table visible on the first trip)
        if (MODE == MODE_KAN && ch < nspl) {
            for (int e = tid; e < 4 * HT; e += 256) {
                const int q = e / HT, pos = e - q * HT;
                const int hy = pos / RS, hx = pos - hy * RS;
                const int gy = ty0 + hy - R, gx = tx0 + hx - R, c = ch * 4 + q;
                float xv = 0.f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) xv = xb[((size_t)c * H + gy) * W + gx];
                uintx4 vh, vl;
                spline_bf16x8(xv, tab, kn, u0, inv_h, vh, vl);
                unsigned char* dst = F + pos * PSTR + q * 16;
                *reinterpret_cast<uintx4*>(dst) = vh;
                *reinterpret_cast<uintx4*>(dst + 64) = vl;
            }
        } else {
            const int c0 = (ch - nspl) * 32;
            for (int e = tid; e < 4 * HT; e += 256) {
                const int q = e / HT, pos = e - q * HT;
                const int hy = pos / RS, hx = pos - hy * RS;
                const int gy = ty0 + hy - R, gx = tx0 + hx - R;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                unsigned hv[8], lv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = c0 + q * 8 + j;
                    float v = (in && c < Cin) ? xb[((size_t)c * H + gy) * W + gx] : 0.f;
                    if (MODE == MODE_KAN) v = silu_f(v);
                    split(v, hv[j], lv[j]);
                }
                unsigned char* dst = F + pos * PSTR + q * 16;
                *reinterpret_cast<uintx4*>(dst) = uintx4{hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), hv[4] | (hv[5] << 16), hv[6] | (hv[7] << 16)};
                *reinterpret_cast<uintx4*>(dst + 64) = uintx4{lv[0] | (lv[1] << 16), lv[2] | (lv[3] << 16), lv[4] | (lv[5] << 16), lv[6] | (lv[7] << 16)};
            }
        }
        {   // dY tile: 2 x 8 channels of output tile ot at the 128 pixels (zero outside the image)
            const int q = tid >> 7, pix = tid & 127;
            const int gy = ty0 + pix / WG_TW, gx = tx0 + pix % WG_TW;
            const bool in = gy < H && gx < W;
            unsigned hv[8], lv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int o = ot * 16 + q * 8 + j;
                split((in && o < Cout) ? dyb[((size_t)o * H + gy) * W + gx] : 0.f, hv[j], lv[j]);
            }
            unsigned char* dst = DY + pix * WG_DPS + q * 16;
            *reinterpret_cast<uintx4*>(dst) = uintx4{hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), hv[4] | (hv[5] << 16), hv[6] | (hv[7] << 16)};
            *reinterpret_cast<uintx4*>(dst + 32) = uintx4{lv[0] | (lv[1] << 16), lv[2] | (lv[3] << 16), lv[4] | (lv[5] << 16), lv[6] | (lv[7] << 16)};
        }
        __syncthreads();
        const bf16x8 ah = tr_pair(DY + a0, DY + a1), al = tr_pair(DY + a0 + 32, DY + a1 + 32);
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int tap = tg * 9 + t;
                if (tap >= T) continue;          // (workgroup-uniform) the last tap group of a 5x5 / 7x7 kernel is partial
                const int to = ((tap / KSZ) * RS + (tap % KSZ)) * PSTR + ft * 32;
                const bf16x8 bh = tr_pair(F + f0 + to, F + f1 + to), bl = tr_pair(F + f0 + to + 64, F + f1 + to + 64);
                acc[ft][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[ft][t], 0, 0, 0);
                asm volatile("" ::"v"(al), "v"(bh));
                acc[ft][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[ft][t], 0, 0, 0);
                asm volatile("" ::"v"(ah), "v"(bl));
                acc[ft][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[ft][t], 0, 0, 0);
                asm volatile("" ::"v"(ah), "v"(bh));
            }
    }

    // cross-wave sums, 6 accumulator tiles per round through LDS; D[o][f]: lane = feature li, registers = channels 4 lg + r
    float* out = slab + (((((size_t)ch * TG + tg) * OT + ot) * S + s) * 18) * 256;
#pragma unroll
    for (int round = 0; round < 3; ++round) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int id = round * 6 + k;
            *reinterpret_cast<floatx4*>(red + ((wave * 6 + k) * 256) + li * 16 + 4 * lg) = acc[id / 9][id % 9];
        }
        __syncthreads();
        for (int e = tid; e < 6 * 64; e += 256) {
            const int k = e >> 6, v4 = e & 63;
            floatx4 sum = *reinterpret_cast<const floatx4*>(red + (k * 256) + v4 * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) sum += *reinterpret_cast<const floatx4*>(red + ((w * 6 + k) * 256) + v4 * 4);
            *reinterpret_cast<floatx4*>(out + (size_t)(round * 6 + k) * 256 + v4 * 4) = sum;
        }
    }
}

// stage 1 of the slab reduction: out[g][18][256] = sum_s slab[g][s][18][256] (fixed order, coalesced 1-KB rows; one workgroup
// per accumulator tile).  The unpack kernels below then run on the summed tiles (S = 1): as one kernel each they took 96 / 22 us
// (every thread walking 9 x S strided values), in two stages ~10.
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ out, int S) {
    const int g = blockIdx.x / 18, t = blockIdx.x % 18;
    const float* p = slab + (((size_t)g * S) * 18 + t) * 256 + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int sp = 0;
    for (; sp + 3 < S; sp += 4) {
        a0 += p[(size_t)sp * 18 * 256];
        a1 += p[(size_t)(sp + 1) * 18 * 256];
        a2 += p[(size_t)(sp + 2) * 18 * 256];
        a3 += p[(size_t)(sp + 3) * 18 * 256];
    }
    for (; sp < S; ++sp) a0 += p[(size_t)sp * 18 * 256];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = (a0 + a1) + (a2 + a3);
}

// slab -> parameter gradients.  One thread per (o, c, tap); sums the S splits in a fixed order.
__global__ void kan_wgrad_x3_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ sw, const float* __restrict__ sc,
                                           float* __restrict__ d_bw, float* __restrict__ d_sw, float* __restrict__ d_sc, int Cin,
                                           int Cout, int OT, int S) {
    const int total = Cout * Cin * 9, nspl = Cin / 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int tap = e % 9, c = (e / 9) % Cin, o = e / (9 * Cin);
        const int ot = o >> 4, ol = o & 15;
        auto fetch = [&](int chunk, int fl) {
            const float* p = slab + ((((size_t)chunk * OT + ot) * S) * 18 + (fl >> 4) * 9 + tap) * 256 + (fl & 15) * 16 + ol;
            float a = 0.f;
            for (int sp = 0; sp < S; ++sp) a += p[(size_t)sp * 18 * 256];
            return a;
        };
        const size_t f = (size_t)o * (Cin * 9) + c * 9 + tap;
        d_bw[f] = fetch(nspl + c / 32, c % 32);
        const float scale = sc[f];
        float dsc = 0.f;
#pragma unroll 1
        for (int a = 0; a < 8; ++a) {
            const float g = fetch(c / 4, (c % 4) * 8 + a);
            d_sw[f * 8 + a] = g * scale;
            dsc += g * sw[f * 8 + a];
        }
        d_sc[f] = dsc;
    }
}

__global__ void conv3x3_wgrad_x3_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cin, int Cout, int OT,
                                               int S, int T) {
    const int total = Cout * Cin * T, TG = (T + 8) / 9;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int tap = e % T, c = (e / T) % Cin, o = e / (T * Cin);
        const int chunk = c / 32, fl = c % 32;
        const float* p = slab + (((((size_t)chunk * TG + tap / 9) * OT + (o >> 4)) * S) * 18 + (fl >> 4) * 9 + tap % 9) * 256 + (fl & 15) * 16 + (o & 15);
        float a = 0.f;
        for (int sp = 0; sp < S; ++sp) a += p[(size_t)sp * 18 * 256];
        dw[e] = a;       // [Cout][Cin][K][K]
    }
}

inline int wgrad_splits(int mode, int B, int Cin, int Cout, int H, int W, int ksize = 3) {
    const int nch = n_chunks(mode, Cin) * ((ksize * ksize + 8) / 9), ot = kmu::cdiv(Cout, 16);
    const int ntiles = B * kmu::cdiv(H, WG_TH) * kmu::cdiv(W, WG_TW);
    int S = 768 / (nch * ot);
    if (S > 64) S = 64;
    if (S > ntiles) S = ntiles;
    if (S < 1) S = 1;
    return S;
}

template <int MODE, int KSZ = 3>
int launch_wgrad(const float* x, const float* dy, const float* knots, float* slab, int B, int Cin, int Cout, int H, int W,
                 hipStream_t st) {
    using G = Geo<WG_TH, WG_TW, KSZ>;
    const int NCH = n_chunks(MODE, Cin) * ((KSZ * KSZ + 8) / 9), OT = kmu::cdiv(Cout, 16), S = wgrad_splits(MODE, B, Cin, Cout, H, W, KSZ);
    const int tilesX = kmu::cdiv(W, WG_TW), tilesY = kmu::cdiv(H, WG_TH);
    const size_t lds = (size_t)G::LDS_BYTES + WG_TH * WG_TW * WG_DPS;
    auto kern = conv3x3_x3_wgrad_kernel<MODE, KSZ>;
    KMU_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(S, NCH, OT), dim3(256), lds, st, x, dy, knots, slab, B, Cin, Cout, H, W, OT, S, tilesX, tilesY);
    return kmu::launch_status("conv3x3_x3 wgrad");
}

template <int TH, int TW>
int launch_kan_dgrad(const float* x, const float* dy, const float* knots, const void* wpd, float* dx, int B, int Cin, int Cout,
                     int H, int W, hipStream_t st) {
    using G = Geo<TH, TW>;
    const int CT = kmu::cdiv(Cin, 16);
    const int tilesX = kmu::cdiv(W, TW), tilesY = kmu::cdiv(H, TH);
    auto kern = kan_dgrad_x3_kernel<TH, TW>;
    KMU_MAX_LDS(kern, G::LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(tilesX * tilesY, CT, B), dim3(256), G::LDS_BYTES, st, x, dy, knots, (const bf16x8*)wpd, dx, Cin, Cout,
                       H, W, CT, tilesX);
    return kmu::launch_status("kan_conv2d_bwd_input_x3");
}

// 5x5 / 7x7 (MultiScaleFusion, KM_UNetV3_SH.py:300-306): small tiles (the halo grows), per-tap weight fragments
template <int KSZ>
int dispatch_fwd_k(const float* x, const void* wp, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                   hipStream_t st) {
    const int NT = kmu::cdiv(Cout, 16);
    if (NT % 4 == 0) return launch_fwd<MODE_PLAIN, 4, 16, 1, 4, 1, KSZ>(x, nullptr, wp, bias, nullptr, y, B, Cin, Cout, H, W, 0, st);
    if (NT % 2 == 0) return launch_fwd<MODE_PLAIN, 4, 16, 2, 2, 1, KSZ>(x, nullptr, wp, bias, nullptr, y, B, Cin, Cout, H, W, 0, st);
    return launch_fwd<MODE_PLAIN, 4, 16, 4, 1, 1, KSZ>(x, nullptr, wp, bias, nullptr, y, B, Cin, Cout, H, W, 0, st);
}

}  // namespace

extern "C" size_t kmu_conv3x3_x3_pack_elems(int kan, int Cin, int Cout) {
    return (size_t)n_chunks(kan ? MODE_KAN : MODE_PLAIN, Cin) * 9 * kmu::cdiv(Cout, 16) * 2 * 512;
}

extern "C" int kmu_kan_pack_weights_x3(const float* base_weight, const float* spline_weight, const float* spline_scaler,
                                       void* wp, int Cin, int Cout, kmu_stream_t stream) {
    KMU_REQUIRE(base_weight && spline_weight && spline_scaler && wp, "kan_pack_weights_x3: null pointer");
    KMU_REQUIRE(Cin > 0 && Cout > 0 && Cin % 4 == 0, "kan_pack_weights_x3: Cin=%d must be a positive multiple of 4", Cin);
    const int NT = kmu::cdiv(Cout, 16), NCH = n_chunks(MODE_KAN, Cin);
    const size_t n = (size_t)NCH * 9 * NT * 512;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, base_weight, spline_weight, spline_scaler,
                       (unsigned short*)wp, (int)MODE_KAN, Cin, Cout, NT, NCH, 9);
    return kmu::launch_status("kan_pack_weights_x3");
}

static bool ksize_ok(int k) { return k == 3 || k == 5 || k == 7; }

extern "C" size_t kmu_conv2d_x3_pack_elems(int Cin, int Cout, int ksize) {
    return (size_t)n_chunks(MODE_PLAIN, Cin) * ksize * ksize * kmu::cdiv(Cout, 16) * 2 * 512;
}

// dgrad = 0: forward pack of weight [Cout][Cin][K][K]; dgrad = 1: the flipped / transposed pack whose "forward" is the input
// gradient (run kmu_conv2d_fwd_x3(dy, wp, NULL, dx, B, Cout, Cin, H, W, ksize) with it; kmu_conv2d_x3_pack_elems(Cout, Cin, ksize))
extern "C" int kmu_conv2d_pack_weights_x3(const float* weight, void* wp, int Cin, int Cout, int ksize, int dgrad, kmu_stream_t stream) {
    KMU_REQUIRE(weight && wp, "conv2d_pack_weights_x3: null pointer");
    KMU_REQUIRE(Cin > 0 && Cout > 0 && ksize_ok(ksize), "conv2d_pack_weights_x3: bad dims Cin=%d Cout=%d k=%d (3/5/7)", Cin, Cout, ksize);
    const int T = ksize * ksize;
    const int kin = dgrad ? Cout : Cin, kout = dgrad ? Cin : Cout;       // the kernel's view
    const int NT = kmu::cdiv(kout, 16), NCH = n_chunks(MODE_PLAIN, kin);
    const size_t n = (size_t)NCH * T * NT * 512;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, (const float*)nullptr,
                       (const float*)nullptr, (unsigned short*)wp, (int)(dgrad ? MODE_PLAIN_DGRAD : MODE_PLAIN), kin, kout, NT, NCH, T);
    return kmu::launch_status("conv2d_pack_weights_x3");
}

extern "C" int kmu_conv3x3_pack_weights_x3(const float* weight, void* wp, int Cin, int Cout, kmu_stream_t stream) {
    return kmu_conv2d_pack_weights_x3(weight, wp, Cin, Cout, 3, 0, stream);
}

extern "C" int kmu_conv3x3_pack_weights_dgrad_x3(const float* weight, void* wp, int Cin, int Cout, kmu_stream_t stream) {
    return kmu_conv2d_pack_weights_x3(weight, wp, Cin, Cout, 3, 1, stream);
}

// ---- job table for conv_pack_multi_kernel.  `table` is HOST memory of njobs * kmu_pack_job_bytes() bytes that the caller then
// copies to the device; which = 0 KAN forward pack, 1 KAN input-gradient pack, 2 plain forward pack, 3 plain input-gradient pack
extern "C" size_t kmu_pack_job_bytes(void) { return sizeof(ConvPackJob); }

extern "C" int kmu_conv_pack_job(void* table, int index, int which, const float* w0, const float* w1, const float* w2, void* wp, int Cin,
                                 int Cout, int ksize) {
    KMU_REQUIRE(table && index >= 0 && w0 && wp && which >= 0 && which <= 3, "conv_pack_job: bad arguments");
    KMU_REQUIRE(Cin > 0 && Cout > 0 && ksize_ok(ksize), "conv_pack_job: bad dims Cin=%d Cout=%d k=%d", Cin, Cout, ksize);
    ConvPackJob j{};
    j.w0 = w0, j.w1 = w1, j.w2 = w2, j.wp = (unsigned short*)wp;
    if (which == 0) {
        KMU_REQUIRE(w1 && w2 && ksize == 3 && Cin % 4 == 0, "conv_pack_job: KAN pack needs spline weights, a 3x3 kernel and Cin %% 4 == 0");
        j.kind = 0, j.mode = MODE_KAN, j.Cin = Cin, j.Cout = Cout, j.NT = kmu::cdiv(Cout, 16), j.NCH = n_chunks(MODE_KAN, Cin), j.T = 9;
    } else if (which == 1) {
        KMU_REQUIRE(w1 && w2 && ksize == 3, "conv_pack_job: KAN input-gradient pack needs spline weights and a 3x3 kernel");
        j.kind = 1, j.Cin = Cin, j.Cout = Cout, j.NT = kmu::cdiv(Cin, 16), j.NCH = (Cout + 31) / 32, j.T = 9;
    } else {
        const int dgrad = which == 3, kin = dgrad ? Cout : Cin, kout = dgrad ? Cin : Cout;
        j.kind = 0, j.mode = dgrad ? MODE_PLAIN_DGRAD : MODE_PLAIN, j.Cin = kin, j.Cout = kout, j.NT = kmu::cdiv(kout, 16);
        j.NCH = n_chunks(MODE_PLAIN, kin), j.T = ksize * ksize;
    }
    reinterpret_cast<ConvPackJob*>(table)[index] = j;
    return 0;
}

extern "C" int kmu_conv_pack_multi(const void* device_table, int njobs, kmu_stream_t stream) {
    KMU_REQUIRE(device_table && njobs > 0 && njobs <= 65535, "conv_pack_multi: bad arguments");
    hipLaunchKernelGGL(conv_pack_multi_kernel, dim3(96, njobs), dim3(256), 0, (hipStream_t)stream, (const ConvPackJob*)device_table);
    return kmu::launch_status("conv_pack_multi");
}

extern "C" int kmu_kan_conv2d_fwd_x3(const float* x, const float* knots, const void* wp, const float* residual, float* y, int B,
                                     int Cin, int Cout, int H, int W, int relu, kmu_stream_t stream) {
    KMU_REQUIRE(x && knots && wp && y, "kan_conv2d_fwd_x3: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && B <= 65535 && Cin % 4 == 0, "kan_conv2d_fwd_x3: bad dims");
    return dispatch_fwd<MODE_KAN>(x, knots, wp, nullptr, residual, y, B, Cin, Cout, H, W, relu, (hipStream_t)stream);
}

extern "C" int kmu_conv2d_fwd_x3(const float* x, const void* wp, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                                 int ksize, kmu_stream_t stream) {
    KMU_REQUIRE(x && wp && y, "conv2d_fwd_x3: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && B <= 65535 && ksize_ok(ksize), "conv2d_fwd_x3: bad dims");
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 3) return dispatch_fwd<MODE_PLAIN>(x, nullptr, wp, bias, nullptr, y, B, Cin, Cout, H, W, 0, st);
    if (ksize == 5) return dispatch_fwd_k<5>(x, wp, bias, y, B, Cin, Cout, H, W, st);
    return dispatch_fwd_k<7>(x, wp, bias, y, B, Cin, Cout, H, W, st);
}

extern "C" int kmu_conv3x3_fwd_x3(const float* x, const void* wp, const float* bias, float* y, int B, int Cin, int Cout, int H,
                                  int W, kmu_stream_t stream) {
    return kmu_conv2d_fwd_x3(x, wp, bias, y, B, Cin, Cout, H, W, 3, stream);
}

extern "C" size_t kmu_kan_dgrad_x3_pack_elems(int Cin, int Cout) {
    return (size_t)((Cout + 31) / 32) * 81 * kmu::cdiv(Cin, 16) * 2 * 512;
}

extern "C" int kmu_kan_pack_weights_dgrad_x3(const float* base_weight, const float* spline_weight, const float* spline_scaler,
                                             void* wpd, int Cin, int Cout, kmu_stream_t stream) {
    KMU_REQUIRE(base_weight && spline_weight && spline_scaler && wpd, "kan_pack_weights_dgrad_x3: null pointer");
    KMU_REQUIRE(Cin > 0 && Cout > 0, "kan_pack_weights_dgrad_x3: bad dims Cin=%d Cout=%d", Cin, Cout);
    const int CT = kmu::cdiv(Cin, 16), NCH = (Cout + 31) / 32;
    const size_t n = (size_t)NCH * 81 * CT * 512;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_kan_dgrad_x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, base_weight, spline_weight,
                       spline_scaler, (unsigned short*)wpd, Cin, Cout, CT, NCH);
    return kmu::launch_status("kan_pack_weights_dgrad_x3");
}

extern "C" int kmu_kan_conv2d_bwd_input_x3(const float* x, const float* dy, const float* knots, const void* wpd, float* dx, int B,
                                           int Cin, int Cout, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && dy && knots && wpd && dx, "kan_conv2d_bwd_input_x3: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && B <= 65535, "kan_conv2d_bwd_input_x3: bad dims");
    hipStream_t st = (hipStream_t)stream;
    if ((long)B * H * W >= 32768 && W >= 32) return launch_kan_dgrad<4, 32>(x, dy, knots, wpd, dx, B, Cin, Cout, H, W, st);
    return launch_kan_dgrad<4, 16>(x, dy, knots, wpd, dx, B, Cin, Cout, H, W, st);
}

extern "C" size_t kmu_conv3x3_x3_wgrad_ws_bytes(int kan, int B, int Cin, int Cout, int H, int W) {
    const int mode = kan ? MODE_KAN : MODE_PLAIN;
    return (size_t)n_chunks(mode, Cin) * kmu::cdiv(Cout, 16) * (wgrad_splits(mode, B, Cin, Cout, H, W) + 1) * 18 * 256 * sizeof(float);
}

extern "C" int kmu_kan_conv2d_bwd_weights_x3(const float* x, const float* dy, const float* knots, const float* spline_weight,
                                             const float* spline_scaler, float* d_base_weight, float* d_spline_weight,
                                             float* d_spline_scaler, void* ws, size_t ws_bytes, int B, int Cin, int Cout, int H, int W,
                                             kmu_stream_t stream) {
    KMU_REQUIRE(x && dy && knots && spline_weight && spline_scaler && d_base_weight && d_spline_weight && d_spline_scaler && ws,
                "kan_conv2d_bwd_weights_x3: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && Cin % 4 == 0, "kan_conv2d_bwd_weights_x3: bad dims");
    KMU_REQUIRE(ws_bytes >= kmu_conv3x3_x3_wgrad_ws_bytes(1, B, Cin, Cout, H, W), "kan_conv2d_bwd_weights_x3: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_wgrad<MODE_KAN>(x, dy, knots, (float*)ws, B, Cin, Cout, H, W, st);
    if (rc) return rc;
    const int total = Cout * Cin * 9, S = wgrad_splits(MODE_KAN, B, Cin, Cout, H, W), NG = n_chunks(MODE_KAN, Cin) * kmu::cdiv(Cout, 16);
    float* sum = (float*)ws + (size_t)NG * S * 18 * 256;
    hipLaunchKernelGGL(slab_sum_kernel, dim3(NG * 18), dim3(256), 0, st, (const float*)ws, sum, S);
    hipLaunchKernelGGL(kan_wgrad_x3_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)sum, spline_weight,
                       spline_scaler, d_base_weight, d_spline_weight, d_spline_scaler, Cin, Cout, kmu::cdiv(Cout, 16), 1);
    return kmu::launch_status("kan_conv2d_bwd_weights_x3 reduce");
}

extern "C" size_t kmu_conv2d_x3_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int ksize) {
    return (size_t)n_chunks(MODE_PLAIN, Cin) * ((ksize * ksize + 8) / 9) * kmu::cdiv(Cout, 16) *
           (wgrad_splits(MODE_PLAIN, B, Cin, Cout, H, W, ksize) + 1) * 18 * 256 * sizeof(float);
}

extern "C" int kmu_conv2d_bwd_weight_x3(const float* x, const float* dy, float* d_weight, void* ws, size_t ws_bytes, int B, int Cin,
                                        int Cout, int H, int W, int ksize, kmu_stream_t stream) {
    KMU_REQUIRE(x && dy && d_weight && ws, "conv2d_bwd_weight_x3: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && ksize_ok(ksize), "conv2d_bwd_weight_x3: bad dims");
    KMU_REQUIRE(ws_bytes >= kmu_conv2d_x3_wgrad_ws_bytes(B, Cin, Cout, H, W, ksize), "conv2d_bwd_weight_x3: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rc = ksize == 3   ? launch_wgrad<MODE_PLAIN, 3>(x, dy, nullptr, (float*)ws, B, Cin, Cout, H, W, st)
             : ksize == 5 ? launch_wgrad<MODE_PLAIN, 5>(x, dy, nullptr, (float*)ws, B, Cin, Cout, H, W, st)
                          : launch_wgrad<MODE_PLAIN, 7>(x, dy, nullptr, (float*)ws, B, Cin, Cout, H, W, st);
    if (rc) return rc;
    const int total = Cout * Cin * ksize * ksize, S = wgrad_splits(MODE_PLAIN, B, Cin, Cout, H, W, ksize);
    const int NG = n_chunks(MODE_PLAIN, Cin) * ((ksize * ksize + 8) / 9) * kmu::cdiv(Cout, 16);
    float* sum = (float*)ws + (size_t)NG * S * 18 * 256;
    hipLaunchKernelGGL(slab_sum_kernel, dim3(NG * 18), dim3(256), 0, st, (const float*)ws, sum, S);
    hipLaunchKernelGGL(conv3x3_wgrad_x3_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)sum, d_weight, Cin,
                       Cout, kmu::cdiv(Cout, 16), 1, ksize * ksize);
    return kmu::launch_status("conv2d_bwd_weight_x3 reduce");
}

extern "C" int kmu_conv3x3_bwd_weight_x3(const float* x, const float* dy, float* d_weight, void* ws, size_t ws_bytes, int B, int Cin,
                                         int Cout, int H, int W, kmu_stream_t stream) {
    return kmu_conv2d_bwd_weight_x3(x, dy, d_weight, ws, ws_bytes, B, Cin, Cout, H, W, 3, stream);
}

// tools only: how the forward splits output-channel tiles over workgroups at small images (0 automatic, 1 never, 2 always)
extern "C" void kmu_conv_debug_split(int mode) { g_conv_split = (mode >= 0 && mode <= 2) ? mode : 0; }
