// K1: KANConv2d 3x3/s1/p1 for gfx950 -- forward, input-gradient and weight-gradient kernels.
//
// Replaces convKAN/KANConv2Dlayers.py:15-37 + convKAN/KANlayers.py:577-660 of the reference
// (F.unfold -> Cox-de Boor b_splines -> two F.linear).  Formulation:
//
//     KANConv2d(x) == conv3x3( Phi(x), W' ),  Phi(x)[c*9+j] = {SiLU(x_c), B_0(x_c) .. B_7(x_c)},
//     W'[o,c,j,tap] = base_weight[o,c*9+tap]                      (j = 0)
//                   = spline_weight[o,c*9+tap,j-1]*scaler[o,c*9+tap]   (j = 1..8)
//
// with the image border padded by Phi(0) (NOT zero: F.unfold zero-pads x and B(0) != 0).
// Phi is evaluated ONCE per input element into an LDS halo tile (the reference evaluates it
// per unfolded element = 9x more), then contracted on the exact-fp32 matrix core
// (v_mfma_f32_16x16x4_f32): M = pixels (16 consecutive along W per fragment), N = 16 output
// channels, K-step = 4 input channels at one (basis j, tap).  Weights are pre-packed in
// fragment order so each B fragment is one coalesced 256-B wave load served from L2.
//
// LDS bank rule used throughout (MI355X_MICROARCH.md, LDS): ds_read_b32 is serviced per
// 32-lane half; a fragment read touches 16 consecutive words for each of 2 k-slices per half,
// so the k-slice stride is padded to == 16 (mod 32) words (or == 2 for the transposed read of
// the weight-gradient kernel) to stay conflict-free.
#include "common.h"

using kmu::floatx4;

namespace {

constexpr int NBASIS = 9;  // SiLU + 8 cubic B-spline bases (grid_size 5 + order 3)

__host__ __device__ constexpr int pad_mod32(int n, int want) { return n + ((want - (n % 32)) + 32) % 32; }

// ---------------------------------------------------------------------------------------------
// Phi(x) and dPhi/dx.  kn = 18 extended knots in LDS: kn[idx+3] = U[idx], idx in [-3,14]; U[0..11]
// are the layer's real knots (KANlayers.py:526-535), the 3 virtual ones on each side only feed
// bases that do not exist in the reference's 8-basis output and are discarded.
//
// Span search uses the same half-open test as the reference's order-0 indicator
// (x >= t_k) & (x < t_k+1) (KANlayers.py:593); the cubic values come from the local
// triangular scheme (NURBS Book A2.2), which is Cox-de Boor restricted to the 4 non-zero bases.
// ---------------------------------------------------------------------------------------------
template <bool DERIV>
__device__ __forceinline__ void kan_phi(float x, const float* kn, float (&phi)[NBASIS], float (&dphi)[NBASIS]) {
    const float s = 1.f / (1.f + __expf(-x));
    phi[0] = x * s;
    if (DERIV) dphi[0] = s * (1.f + x * (1.f - s));
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) cnt += (x >= kn[k + 3]) ? 1 : 0;
    const bool valid = (cnt >= 1) && (cnt <= 11);  // t_0 <= x < t_11
    const int i = min(max(cnt - 1, 0), 10);
    const float* u = kn + 3 + i;  // u[d] = U[i+d]
    const float l1 = x - u[0], r1 = u[1] - x;
    const float l2 = x - u[-1], r2 = u[2] - x;
    const float l3 = x - u[-2], r3 = u[3] - x;
    float t = 1.f / (r1 + l1);
    const float n0 = r1 * t, n1 = l1 * t;  // degree 1
    t = n0 / (r1 + l2);
    const float m0 = r1 * t;
    float sv = l2 * t;
    t = n1 / (r2 + l1);
    const float m1 = sv + r2 * t, m2 = l1 * t;  // degree 2
    const float t0 = m0 / (r1 + l3);
    const float c0 = r1 * t0;
    sv = l3 * t0;
    const float t1 = m1 / (r2 + l2);
    const float c1 = sv + r2 * t1;
    sv = l2 * t1;
    const float t2 = m2 / (r3 + l1);
    const float c2 = sv + r3 * t2, c3 = l1 * t2;  // degree 3: N_{i-3..i}
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int r = a - i + 3;
        const float v = (r == 0) ? c0 : (r == 1) ? c1 : (r == 2) ? c2 : (r == 3) ? c3 : 0.f;
        phi[1 + a] = valid ? v : 0.f;
        if (DERIV) {
            // N'_{a,3} = 3 [ N_{a,2}/(U_{a+3}-U_a) - N_{a+1,2}/(U_{a+4}-U_{a+1}) ]
            const float d = (r == 0) ? -t0 : (r == 1) ? (t0 - t1) : (r == 2) ? (t1 - t2) : (r == 3) ? t2 : 0.f;
            dphi[1 + a] = valid ? 3.f * d : 0.f;
        }
    }
}

__device__ __forceinline__ void load_knots(const float* __restrict__ knots, float* kn, int tid) {
    if (tid < 18) {
        const int idx = tid - 3;
        float v;
        if (idx < 0)
            v = knots[0] + (float)idx * (knots[1] - knots[0]);
        else if (idx > 11)
            v = knots[11] + (float)(idx - 11) * (knots[11] - knots[10]);
        else
            v = knots[idx];
        kn[tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// weight packs
//   fwd : wp[cg][j][tap][nt][q*16+r] = W'[o = nt*16+r][c = cg*4+q][j][tap]
//   bwd : wq[tap][og][j][ct][q*16+r] = W'[o = og*4+q][c = ct*16+r][j][tap]
// (zero where o >= Cout or c >= Cin)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wprime(const float* bw, const float* sw, const float* sc, int Cin, int Cout, int o,
                                        int c, int j, int tap) {
    if (o >= Cout || c >= Cin) return 0.f;
    const size_t f = (size_t)o * (Cin * 9) + c * 9 + tap;
    return j == 0 ? bw[f] : sw[f * 8 + (j - 1)] * sc[f];
}

__global__ void kan_pack_kernel(const float* __restrict__ bw, const float* __restrict__ sw,
                                const float* __restrict__ sc, float* __restrict__ wp, float* __restrict__ wq, int Cin,
                                int Cout, int CG, int NT, int OG, int CT) {
    const size_t nf = wp ? (size_t)CG * 81 * NT * 64 : 0, nb = (size_t)OG * 81 * CT * 64;   // wp == nullptr: backward pack only
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nf + (wq ? nb : 0);
         e += (size_t)gridDim.x * blockDim.x) {
        if (e < nf) {
            int l = e & 63;
            size_t t = e >> 6;
            const int nt = t % NT;
            t /= NT;
            const int tap = t % 9;
            t /= 9;
            const int j = t % 9;
            const int cg = (int)(t / 9);
            wp[e] = wprime(bw, sw, sc, Cin, Cout, nt * 16 + (l & 15), cg * 4 + (l >> 4), j, tap);
        } else {
            const size_t e2 = e - nf;
            int l = e2 & 63;
            size_t t = e2 >> 6;
            const int ct = t % CT;
            t /= CT;
            const int j = t % 9;
            t /= 9;
            const int og = t % OG;
            const int tap = (int)(t / OG);
            wq[e2] = wprime(bw, sw, sc, Cin, Cout, og * 4 + (l >> 4), ct * 16 + (l & 15), j, tap);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int TH, int TW>
struct FwdGeom {
    static constexpr int RS = TW + 2;
    static constexpr int HT = (TH + 2) * RS;            // halo-tile elements per (channel, basis) plane
    static constexpr int CS = pad_mod32(NBASIS * HT, 16);  // channel stride, == 16 (mod 32)
    static constexpr int LDS_FLOATS = 32 + 4 * CS;
};

template <int TH, int TW, int WM, int WN, int MREP, int NREP>
__global__ __launch_bounds__(256) void kan_fwd_kernel(const float* __restrict__ x, const float* __restrict__ knots,
                                                      const float* __restrict__ wp,
                                                      const float* __restrict__ residual, float* __restrict__ y,
                                                      int Cin, int Cout, int H, int W, int CG, int NT, int tilesX,
                                                      int relu) {
    static_assert(WM * WN == 4, "4 waves");
    static_assert(WM * MREP == TH * TW / 16, "segments");
    using G = FwdGeom<TH, TW>;
    constexpr int RS = G::RS, HT = G::HT, CS = G::CS, SPR = TW / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* kn = smem;
    float* phi = smem + 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ty0 = (blockIdx.x / tilesX) * TH, tx0 = (blockIdx.x % tilesX) * TW;
    const int b = blockIdx.z;
    const int nt0 = (blockIdx.y * WN + wn) * NREP;
    const int li = lane & 15, lq = lane >> 4;

    load_knots(knots, kn, tid);

    floatx4 acc[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[m][n] = floatx4{0.f, 0.f, 0.f, 0.f};

    int aoff[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int seg = wm * MREP + m;
        aoff[m] = lq * CS + (seg / SPR) * RS + (seg % SPR) * 16 + li;
    }

    const float* xb = x + (size_t)b * Cin * H * W;
    for (int cg = 0; cg < CG; ++cg) {
        __syncthreads();  // previous chunk's fragment reads done (and knots visible on the first trip)
        for (int e = tid; e < 4 * HT; e += 256) {
            const int q = e / HT, rem = e - q * HT;
            const int hy = rem / RS, hx = rem - hy * RS;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1, c = cg * 4 + q;
            float xv = 0.f;  // out-of-image taps see x = 0 -> Phi(0) (reference: F.unfold zero padding)
            if (c < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) xv = xb[((size_t)c * H + gy) * W + gx];
            float p[NBASIS], d[NBASIS];
            kan_phi<false>(xv, kn, p, d);
            float* dst = phi + q * CS + rem;
#pragma unroll
            for (int j = 0; j < NBASIS; ++j) dst[j * HT] = p[j];
        }
        __syncthreads();
        const float* wpc = wp + ((size_t)cg * 81 * NT + nt0) * 64 + lane;
        // B fragments (packed weights, L2-resident) are double-buffered in registers: the 9 taps of basis j+1
        // are requested before the 9x MREP x NREP MFMAs of basis j issue, so their L2 latency hides under them.
        float bcur[9][NREP], bnxt[9][NREP];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int n = 0; n < NREP; ++n) bcur[tap][n] = wpc[((size_t)tap * NT + n) * 64];
#pragma unroll
        for (int j = 0; j < NBASIS; ++j) {
            if (j + 1 < NBASIS) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int n = 0; n < NREP; ++n) bnxt[tap][n] = wpc[((size_t)((j + 1) * 9 + tap) * NT + n) * 64];
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
#pragma unroll
                for (int m = 0; m < MREP; ++m) {
                    const float a = phi[aoff[m] + j * HT + ky * RS + kx];
#pragma unroll
                    for (int n = 0; n < NREP; ++n) kmu::mfma_tied(acc[m][n], a, bcur[tap][n]);
                }
            }
            if (j + 1 < NBASIS) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int n = 0; n < NREP; ++n) bcur[tap][n] = bnxt[tap][n];
            }
        }
    }

    // epilogue: lane holds 4 consecutive pixels (rows of the C tile) of output channel nt*16 + li
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n) kmu::mfma_drain(acc[m][n]);
    const bool vec_ok = (W & 3) == 0;
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int seg = wm * MREP + m;
        const int gy = ty0 + seg / SPR, px0 = tx0 + (seg % SPR) * 16 + lq * 4;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
            const int o = (nt0 + n) * 16 + li;
            if (o >= Cout || gy >= H || px0 >= W) continue;
            const size_t idx = (((size_t)b * Cout + o) * H + gy) * W + px0;
            floatx4 v = acc[m][n];
            if (vec_ok && px0 + 3 < W) {
                if (residual) {
                    const floatx4 r = *reinterpret_cast<const floatx4*>(residual + idx);
                    v += r;
                }
                if (relu) {
                    v[0] = fmaxf(v[0], 0.f);
                    v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f);
                    v[3] = fmaxf(v[3], 0.f);
                }
                *reinterpret_cast<floatx4*>(y + idx) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
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
}

// ---------------------------------------------------------------------------------------------
// input gradient:  G[pixel][c][j] = sum_{o,tap} dY[o][pixel - tap + 1] * W'[o][c][j][tap]   (MFMA)
//                  dX[pixel][c]   = sum_j dPhi_j(x[pixel][c]) * G[pixel][c][j]           (epilogue)
// M = pixels, N = 16 input channels per (basis j) accumulator, K-step = 4 output channels at one tap.
// ---------------------------------------------------------------------------------------------
template <int TH, int TW>
struct BwdInGeom {
    static constexpr int RS = TW + 2;
    static constexpr int HT = (TH + 2) * RS;
    static constexpr int OS = pad_mod32(HT, 16);  // output-channel stride, == 16 (mod 32)
};

template <int TH, int TW, int WM, int WN, int MREP, int CREP>
__global__ __launch_bounds__(256) void kan_bwd_input_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ knots,
                                                            const float* __restrict__ wq, float* __restrict__ dx,
                                                            int Cin, int Cout, int H, int W, int OG, int CT,
                                                            int tilesX) {
    static_assert(WM * WN == 4, "4 waves");
    static_assert(WM * MREP == TH * TW / 16, "segments");
    using G = BwdInGeom<TH, TW>;
    constexpr int RS = G::RS, HT = G::HT, OS = G::OS, SPR = TW / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* kn = smem;
    float* dyl = smem + 32;  // [OG*4][OS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ty0 = (blockIdx.x / tilesX) * TH, tx0 = (blockIdx.x % tilesX) * TW;
    const int b = blockIdx.z;
    const int ct0 = (blockIdx.y * WN + wn) * CREP;
    const int li = lane & 15, lq = lane >> 4;

    load_knots(knots, kn, tid);
    const float* dyb = dy + (size_t)b * Cout * H * W;
    for (int e = tid; e < OG * 4 * HT; e += 256) {
        const int o = e / HT, rem = e - o * HT;
        const int hy = rem / RS, hx = rem - hy * RS;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        float v = 0.f;
        if (o < Cout && gy >= 0 && gy < H && gx >= 0 && gx < W) v = dyb[((size_t)o * H + gy) * W + gx];
        dyl[o * OS + rem] = v;
    }
    __syncthreads();

    floatx4 acc[MREP][NBASIS][CREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int j = 0; j < NBASIS; ++j)
#pragma unroll
            for (int c = 0; c < CREP; ++c) acc[m][j][c] = floatx4{0.f, 0.f, 0.f, 0.f};

    int aoff[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int seg = wm * MREP + m;
        aoff[m] = lq * OS + (seg / SPR) * RS + (seg % SPR) * 16 + li;
    }

#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        // source dY pixel of output pixel (r,c) under tap (ky,kx) is (r-ky+1, c-kx+1) -> halo (r-ky+2, c-kx+2)
        const int toff = (2 - ky) * RS + (2 - kx);
#pragma unroll 1
        for (int og = 0; og < OG; ++og) {
            const float* wqc = wq + (((size_t)tap * OG + og) * 9 * CT + ct0) * 64 + lane;
            float a[MREP];
#pragma unroll
            for (int m = 0; m < MREP; ++m) a[m] = dyl[aoff[m] + og * 4 * OS + toff];
#pragma unroll
            for (int j = 0; j < NBASIS; ++j) {
#pragma unroll
                for (int c = 0; c < CREP; ++c) {
                    const float bf = wqc[((size_t)j * CT + c) * 64];
#pragma unroll
                    for (int m = 0; m < MREP; ++m)
                        acc[m][j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bf, acc[m][j][c], 0, 0, 0);
                }
            }
        }
    }

    const bool vec_ok = (W & 3) == 0;
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int seg = wm * MREP + m;
        const int gy = ty0 + seg / SPR, px0 = tx0 + (seg % SPR) * 16 + lq * 4;
#pragma unroll
        for (int c = 0; c < CREP; ++c) {
            const int ch = (ct0 + c) * 16 + li;
            if (ch >= Cin || gy >= H || px0 >= W) continue;
            const size_t idx = (((size_t)b * Cin + ch) * H + gy) * W + px0;
            float xv[4], out[4];
            if (vec_ok && px0 + 3 < W) {
                const floatx4 t = *reinterpret_cast<const floatx4*>(x + idx);
                xv[0] = t[0], xv[1] = t[1], xv[2] = t[2], xv[3] = t[3];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[r] = (px0 + r < W) ? x[idx + r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p[NBASIS], d[NBASIS];
                kan_phi<true>(xv[r], kn, p, d);
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NBASIS; ++j) s += d[j] * acc[m][j][c][r];
                out[r] = s;
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
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dW'[o][c][j][tap] = sum_pixels dY[o][pixel] * Phi_j(xpad[c][pixel + tap - 1])
// One workgroup = (16 input channels, 16 output channels, a strided subset of 4x16-pixel tiles).
// MFMA: rows = o, cols = c, K-step = 4 consecutive pixels; the 81 (j,tap) accumulators are dealt
// round-robin to the 4 waves (21/20/20/20).  Partial slabs -> kan_bwd_weight_reduce_kernel
// (deterministic: no float atomics).
// ---------------------------------------------------------------------------------------------
struct BwdWGeom {
    static constexpr int TH = 4, TW = 16, RS = TW + 2, HT = (TH + 2) * RS;  // 108
    static constexpr int CS = pad_mod32(NBASIS * HT, 2);                     // 994, == 2 (mod 32)
    static constexpr int DS = pad_mod32(TH * TW, 2);                         // 66
    static constexpr int LDS_FLOATS = 32 + 16 * CS + 16 * DS;
    static constexpr int NACC = 21;
};

__global__ __launch_bounds__(256) void kan_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ knots, float* __restrict__ slab,
                                                             int B, int Cin, int Cout, int H, int W, int OT, int S,
                                                             int tilesX, int tilesY) {
    using G = BwdWGeom;
    constexpr int TH = G::TH, TW = G::TW, RS = G::RS, HT = G::HT, CS = G::CS, DS = G::DS, NACC = G::NACC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* kn = smem;
    float* phi = smem + 32;         // [16 c][9 j][HT]
    float* dyl = phi + 16 * CS;     // [16 o][64 px]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.x, ot = blockIdx.y, ct = blockIdx.z;
    const int li = lane & 15, lq = lane >> 4;
    load_knots(knots, kn, tid);

    floatx4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = B * tilesY * tilesX;
    for (int tile = s; tile < ntiles; tile += S) {
        const int b = tile / (tilesY * tilesX), tr = tile % (tilesY * tilesX);
        const int ty0 = (tr / tilesX) * TH, tx0 = (tr % tilesX) * TW;
        __syncthreads();
        for (int e = tid; e < 16 * HT; e += 256) {
            const int cl = e / HT, rem = e - cl * HT;
            const int hy = rem / RS, hx = rem - hy * RS;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1, c = ct * 16 + cl;
            float xv = 0.f;
            if (c < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) xv = x[(((size_t)b * Cin + c) * H + gy) * W + gx];
            float p[NBASIS], d[NBASIS];
            kan_phi<false>(xv, kn, p, d);
            float* dst = phi + cl * CS + rem;
#pragma unroll
            for (int j = 0; j < NBASIS; ++j) dst[j * HT] = p[j];
        }
        for (int e = tid; e < 16 * TH * TW; e += 256) {
            const int ol = e / (TH * TW), px = e % (TH * TW);
            const int gy = ty0 + px / TW, gx = tx0 + px % TW, o = ot * 16 + ol;
            float v = 0.f;
            if (o < Cout && gy < H && gx < W) v = dy[(((size_t)b * Cout + o) * H + gy) * W + gx];
            dyl[ol * DS + px] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int quad = 0; quad < TH * TW / 4; ++quad) {
            const int row = quad / (TW / 4), x4 = (quad % (TW / 4)) * 4;
            const float a = dyl[li * DS + quad * 4 + lq];              // A[o = li][k = pixel lq]
            const float* pb = phi + li * CS + row * RS + x4 + lq;      // B[k = pixel lq][c = li]
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                const int f = wave + 4 * t;  // wave-uniform (j,tap) id
                if (f < 81) {
                    const int j = f / 9, tap = f - j * 9;
                    const float bv = pb[j * HT + (tap / 3) * RS + (tap % 3)];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // slab layout [(ct*OT+ot)*S+s][f][c_local][o_local]; lane: c_local = li, o_local = lq*4 + r
    float* out = slab + ((((size_t)ct * OT + ot) * S + s) * 81) * 256 + li * 16 + lq * 4;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        const int f = wave + 4 * t;
        if (f < 81) *reinterpret_cast<floatx4*>(out + (size_t)f * 256) = acc[t];
    }
}

// stage 1 of the slab reduction: every 256-thread block sums up to RG splits of one (ct,ot,f) slab row
// (fully coalesced 1-KiB reads); stage 2 (below) sums the <= ceil(S/RG) group partials and unpacks.
constexpr int RG = 16;
__global__ __launch_bounds__(256) void kan_bwd_weight_reduce1_kernel(const float* __restrict__ slab,
                                                                     float* __restrict__ part, int S, int SG) {
    const int f = blockIdx.x % 81, pair = blockIdx.x / 81, sg = blockIdx.y;
    const float* src = slab + (((size_t)pair * S + (size_t)sg * RG) * 81 + f) * 256 + threadIdx.x;
    const int n = min(RG, S - sg * RG);
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += src[(size_t)i * 81 * 256];
    part[(((size_t)pair * SG + sg) * 81 + f) * 256 + threadIdx.x] = acc;
}

// sum slabs over splits and unpack dW' into the three parameter gradients (KANlayers.py:644-660 autograd)
__global__ void kan_bwd_weight_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ sw,
                                             const float* __restrict__ sc, float* __restrict__ d_bw,
                                             float* __restrict__ d_sw, float* __restrict__ d_sc, int Cin, int Cout,
                                             int OT, int CT, int S) {
    const size_t total = (size_t)CT * OT * 9 * 256;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ol = e & 15, cl = (e >> 4) & 15;
        size_t t = e >> 8;
        const int tap = t % 9;
        t /= 9;
        const int ot = t % OT, ct = (int)(t / OT);
        const int o = ot * 16 + ol, c = ct * 16 + cl;
        if (o >= Cout || c >= Cin) continue;
        const float* base = slab + (((size_t)ct * OT + ot) * S * 81) * 256 + cl * 16 + ol;
        const size_t f = (size_t)o * (Cin * 9) + c * 9 + tap;
        const float scale = sc[f];
        float dsc = 0.f;
#pragma unroll 1
        for (int j = 0; j < NBASIS; ++j) {
            float acc = 0.f;
            for (int sp = 0; sp < S; ++sp) acc += base[((size_t)sp * 81 + j * 9 + tap) * 256];
            if (j == 0) {
                d_bw[f] = acc;
            } else {
                d_sw[f * 8 + (j - 1)] = acc * scale;
                dsc += acc * sw[f * 8 + (j - 1)];
            }
        }
        d_sc[f] = dsc;
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------------------------
template <int TH, int TW, int WM, int WN, int MREP, int NREP>
int launch_fwd(const float* x, const float* knots, const float* wp, const float* residual, float* y, int B, int Cin,
               int Cout, int H, int W, int relu, hipStream_t st) {
    using G = FwdGeom<TH, TW>;
    const int CG = kmu::cdiv(Cin, 4), NT = kmu::cdiv(Cout, 16);
    const int tilesX = kmu::cdiv(W, TW), tilesY = kmu::cdiv(H, TH);
    dim3 grid(tilesX * tilesY, NT / (WN * NREP), B);
    const size_t lds = G::LDS_FLOATS * sizeof(float);
    auto kern = kan_fwd_kernel<TH, TW, WM, WN, MREP, NREP>;
    KMU_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, knots, wp, residual, y, Cin, Cout, H, W, CG, NT, tilesX,
                       relu);
    return kmu::launch_status("kan_conv2d_fwd");
}

template <int TH, int TW, int WM, int WN, int MREP, int CREP>
int launch_bwd_input(const float* x, const float* dy, const float* knots, const float* wq, float* dx, int B, int Cin,
                     int Cout, int H, int W, hipStream_t st) {
    using G = BwdInGeom<TH, TW>;
    const int OG = kmu::cdiv(Cout, 4), CT = kmu::cdiv(Cin, 16);
    const int tilesX = kmu::cdiv(W, TW), tilesY = kmu::cdiv(H, TH);
    dim3 grid(tilesX * tilesY, CT / (WN * CREP), B);
    const size_t lds = (32 + (size_t)OG * 4 * G::OS) * sizeof(float);
    auto kern = kan_bwd_input_kernel<TH, TW, WM, WN, MREP, CREP>;
    KMU_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, dy, knots, wq, dx, Cin, Cout, H, W, OG, CT, tilesX);
    return kmu::launch_status("kan_conv2d_bwd_input");
}

#ifndef KMU_KAN_WSLABS
#define KMU_KAN_WSLABS 512   // weight-gradient workgroups (= 83 KB slabs) per layer; measured at site 1 (B=8): 256 -> 281 us, 512 -> 238 us, 1024 -> 323 us
#endif
int bwd_weight_splits(int B, int Cin, int Cout, int H, int W) {
    const int CT = kmu::cdiv(Cin, 16), OT = kmu::cdiv(Cout, 16);
    const int ntiles = B * kmu::cdiv(H, BwdWGeom::TH) * kmu::cdiv(W, BwdWGeom::TW);
    int S = KMU_KAN_WSLABS / (CT * OT);
    if (S < 1) S = 1;
    if (S > ntiles) S = ntiles;
    return S;
}

}  // namespace

extern "C" size_t kmu_kan_pack_fwd_elems(int Cin, int Cout) {
    return (size_t)kmu::cdiv(Cin, 4) * 81 * kmu::cdiv(Cout, 16) * 64;
}
extern "C" size_t kmu_kan_pack_bwd_elems(int Cin, int Cout) {
    return (size_t)kmu::cdiv(Cout, 4) * 81 * kmu::cdiv(Cin, 16) * 64;
}

extern "C" int kmu_kan_pack_weights(const float* base_weight, const float* spline_weight, const float* spline_scaler,
                                    float* wp_fwd, float* wp_bwd, int Cin, int Cout, kmu_stream_t stream) {
    KMU_REQUIRE(base_weight && spline_weight && spline_scaler && (wp_fwd || wp_bwd), "kan_pack_weights: null pointer");
    KMU_REQUIRE(Cin > 0 && Cout > 0, "kan_pack_weights: bad dims Cin=%d Cout=%d", Cin, Cout);
    const int CG = kmu::cdiv(Cin, 4), NT = kmu::cdiv(Cout, 16), OG = kmu::cdiv(Cout, 4), CT = kmu::cdiv(Cin, 16);
    const size_t n = (wp_fwd ? kmu_kan_pack_fwd_elems(Cin, Cout) : 0) + (wp_bwd ? kmu_kan_pack_bwd_elems(Cin, Cout) : 0);
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(kan_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, base_weight, spline_weight,
                       spline_scaler, wp_fwd, wp_bwd, Cin, Cout, CG, NT, OG, CT);
    return kmu::launch_status("kan_pack_weights");
}

extern "C" int kmu_kan_conv2d_fwd(const float* x, const float* knots, const float* wp_fwd, const float* residual,
                                  float* y, int B, int Cin, int Cout, int H, int W, int relu, kmu_stream_t stream) {
    KMU_REQUIRE(x && knots && wp_fwd && y, "kan_conv2d_fwd: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && B <= 65535, "kan_conv2d_fwd: bad dims");
    hipStream_t st = (hipStream_t)stream;
    const int NT = kmu::cdiv(Cout, 16);
    const long px = (long)B * H * W;
    // tile / wave-layout choice: keep >= ~256 workgroups in flight where the image allows it
    if (NT % 4 == 0) return launch_fwd<4, 16, 1, 4, 4, 1>(x, knots, wp_fwd, residual, y, B, Cin, Cout, H, W, relu, st);
    if (NT % 2 == 0) {
        if (px >= 32768) return launch_fwd<8, 16, 2, 2, 4, 1>(x, knots, wp_fwd, residual, y, B, Cin, Cout, H, W, relu, st);
        return launch_fwd<4, 16, 2, 2, 2, 1>(x, knots, wp_fwd, residual, y, B, Cin, Cout, H, W, relu, st);
    }
    if (px >= 65536) return launch_fwd<8, 32, 4, 1, 4, 1>(x, knots, wp_fwd, residual, y, B, Cin, Cout, H, W, relu, st);
    return launch_fwd<4, 16, 4, 1, 1, 1>(x, knots, wp_fwd, residual, y, B, Cin, Cout, H, W, relu, st);
}

extern "C" int kmu_kan_conv2d_bwd_input(const float* x, const float* dy, const float* knots, const float* wp_bwd,
                                        float* dx, int B, int Cin, int Cout, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && dy && knots && wp_bwd && dx, "kan_conv2d_bwd_input: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && B <= 65535, "kan_conv2d_bwd_input: bad dims");
    KMU_REQUIRE(Cout <= 256, "kan_conv2d_bwd_input: Cout=%d > 256 exceeds the LDS dY tile", Cout);
    hipStream_t st = (hipStream_t)stream;
    const int CT = kmu::cdiv(Cin, 16);
    const long px = (long)B * H * W;
    if (CT % 4 == 0) return launch_bwd_input<4, 16, 1, 4, 4, 1>(x, dy, knots, wp_bwd, dx, B, Cin, Cout, H, W, st);
    if (CT % 2 == 0) return launch_bwd_input<4, 16, 2, 2, 2, 1>(x, dy, knots, wp_bwd, dx, B, Cin, Cout, H, W, st);
    if (px >= 65536 && Cout <= 32)
        return launch_bwd_input<8, 16, 4, 1, 2, 1>(x, dy, knots, wp_bwd, dx, B, Cin, Cout, H, W, st);
    return launch_bwd_input<4, 16, 4, 1, 1, 1>(x, dy, knots, wp_bwd, dx, B, Cin, Cout, H, W, st);
}

extern "C" size_t kmu_kan_bwd_ws_bytes(int B, int Cin, int Cout, int H, int W) {
    const int CT = kmu::cdiv(Cin, 16), OT = kmu::cdiv(Cout, 16);
    const int S = bwd_weight_splits(B, Cin, Cout, H, W);
    return (size_t)CT * OT * ((size_t)S + kmu::cdiv(S, RG)) * 81 * 256 * sizeof(float);
}

extern "C" int kmu_kan_conv2d_bwd_weights(const float* x, const float* dy, const float* knots,
                                          const float* spline_weight, const float* spline_scaler,
                                          float* d_base_weight, float* d_spline_weight, float* d_spline_scaler,
                                          void* ws, size_t ws_bytes, int B, int Cin, int Cout, int H, int W,
                                          kmu_stream_t stream) {
    KMU_REQUIRE(x && dy && knots && spline_weight && spline_scaler && d_base_weight && d_spline_weight &&
                    d_spline_scaler && ws,
                "kan_conv2d_bwd_weights: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "kan_conv2d_bwd_weights: bad dims");
    KMU_REQUIRE(ws_bytes >= kmu_kan_bwd_ws_bytes(B, Cin, Cout, H, W), "kan_conv2d_bwd_weights: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    using G = BwdWGeom;
    const int CT = kmu::cdiv(Cin, 16), OT = kmu::cdiv(Cout, 16);
    const int S = bwd_weight_splits(B, Cin, Cout, H, W);
    const int tilesX = kmu::cdiv(W, G::TW), tilesY = kmu::cdiv(H, G::TH);
    const size_t lds = G::LDS_FLOATS * sizeof(float);
    KMU_MAX_LDS(kan_bwd_weight_kernel, lds);
    hipLaunchKernelGGL(kan_bwd_weight_kernel, dim3(S, OT, CT), dim3(256), lds, st, x, dy, knots, (float*)ws, B, Cin,
                       Cout, H, W, OT, S, tilesX, tilesY);
    int rc = kmu::launch_status("kan_conv2d_bwd_weights");
    if (rc) return rc;
    const int SG = kmu::cdiv(S, RG);
    float* part = (float*)ws + (size_t)CT * OT * S * 81 * 256;
    hipLaunchKernelGGL(kan_bwd_weight_reduce1_kernel, dim3(CT * OT * 81, SG), dim3(256), 0, st, (const float*)ws, part, S, SG);
    rc = kmu::launch_status("kan_conv2d_bwd_weights_reduce1");
    if (rc) return rc;
    const size_t total = (size_t)CT * OT * 9 * 256;
    hipLaunchKernelGGL(kan_bwd_weight_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)part, spline_weight, spline_scaler, d_base_weight, d_spline_weight,
                       d_spline_scaler, Cin, Cout, OT, CT, SG);
    return kmu::launch_status("kan_conv2d_bwd_weights_reduce");
}
