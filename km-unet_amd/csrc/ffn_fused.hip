// EfficientViMBlock's FFN stage as recompute kernels (efficient_vim_init.py:96; FFN = two bias-free 1x1 ConvLayer2D with BatchNorm2d,
// vim_utils_init.py:62-89,122-130):
//
//       out = x + sigmoid(alpha) * ( BN2( W2 relu( BN1( W1 x ) ) ) - x )                 x [B,C,P], W1 [4C,C], W2 [C,4C]
//
// The hidden tensor is 4C wide (33.5 MB at [8,16,128,128]); as separate pointwise-conv / BatchNorm kernels it crossed HBM ~12 times
// per block and direction (z1, h forward; dh, dz1 backward; both weight gradients re-reading h / dz1).  Here nothing 4C wide ever
// reaches memory: every pass re-derives z1 = W1 x from the C-wide input on the matrix core (v_mfma_f32_16x16x4_f32, exact fp32) and
// keeps it in registers.  Train-mode BatchNorm needs its batch statistics before it can be applied, hence the passes:
//
//   forward   F1  x           -> partial (sum, sum of squares) of z1 per hidden channel
//             F2  x           -> z2 = W2 relu(BN1(z1))  (C wide) + partial statistics of z2            [eval: BN1 from running stats, no F1]
//             F3  z2, x       -> out
//   backward  B1  g, z2, x    -> partial sums of BN2's backward (sum dz, sum dz*z2hat) and of d alpha
//             B2  g, z2, x    -> dz2 -> dh = W2^T dz2 -> dr = dh * [h > 0]: partial sums of BN1's backward; dW2 = sum dz2 h^T (slabs)
//             B3  g, z2, x    -> dz1 -> dx = W1^T dz1 + (1 - a) g; dW1 = sum dz1 x^T (slabs)
//
// A kernel that needs channel statistics finds the PARTIAL rows of the kernel before it and every workgroup reduces them itself in a
// fixed order (double) -- L2 reads at its start instead of a finalising launch (or an in-kernel last-arriver with its agent-scope
// fences) on the dependent chain; workgroup 0 also writes the results out (saved statistics, running statistics, parameter
// gradients).  z1 is accumulated in the same k order in F2, B2 and B3, so the ReLU branch taken by the backward is bit-for-bit the
// forward's.
//
// Work units.  One wave = one pixel tile (64 pixels; 32 at C = 64) x one SLICE of the hidden channels (all 64 at C = 16, 32 of 128 /
// 256 at C = 32 / 64), 8 waves per workgroup, and at the bench shapes exactly one unit per wave (256 workgroups).
//   tile-major  (F2, B3): the slices of a tile sit in ONE workgroup, because z2 = W2 h and dx = W1^T dz1 sum over all hidden channels
//               (partial sums / the dz1 tile meet in LDS);
//   slice-major (F1, B2): a workgroup is 8 tiles of ONE slice.  Nothing crosses slices there, the 8 waves share their weights, and the
//               partial rows they leave are 1/JS as wide -- the consumer's fold of the hidden-channel statistics reads JS times fewer bytes.
// Orientations (lane = (n = l % 16, q = l / 16), TT = float4 / float2 components = MFMA tiles per 16 rows):
//   D'  (F1, B2, B3)  M = pixels, N = hidden j, K = channels: A = x loaded [c = 16c'+4q+s][p0 + TT f(n) .. ], f(n) = n/4 + 4(n%4)
//       (component t = tile t), B = W1[j][c].  D[t] register i <-> pixel p0 + 4TT i + TT q + t, column j on the lane: per-channel constants
//       are ONE register per 16 channels, and dz (hidden) is directly the A operand of the weight-gradient product (rows j, k = pixel),
//       whose B operand is x / dz2 loaded [c = lane%16][p0 + 4TT i + TT q .. ].
//       dx = dz1 W1 contracts over j (the lane index): dz1 takes one trip through LDS, 16 hidden channels at a time.
//   N   (F2)          M = hidden j, N = pixels, K = channels: the result D1 (rows j in registers, pixel on the lane) is directly the
//       B operand of z2 = W2 h (K = j) -- no LDS between the two products.
// The hidden tiles (16 channels) of a wave are processed one after the other in B2 / B3 (z1, dh, the element-wise step, the weight
// gradient and the dx contribution of 16 channels at a time): ~190 registers instead of ~300, two waves per SIMD.
#include "common.h"

using kmu::floatx4;

namespace {

typedef float floatx2 __attribute__((ext_vector_type(2)));

#ifndef KMU_FFN_GMAX
#define KMU_FFN_GMAX 256   // workgroups per kernel (tile-major: = partial rows; slice-major: rows = GMAX / JS)
#endif
#ifndef KMU_FFN_DBG
#define KMU_FFN_DBG 0      // timing experiments only (tools/build_variant.py): 1 = skip the partial-row reduction (wrong results)
#endif

template <int TT> struct Vec;
template <> struct Vec<4> { typedef floatx4 type; };
template <> struct Vec<2> { typedef floatx2 type; };

template <int C>
struct Cfg {
    static constexpr int HID = 4 * C;
    static constexpr int CC = C / 16;                  // 16-channel chunks of C
    static constexpr int JS = C == 16 ? 1 : C / 8;     // hidden-channel slices per tile
    static constexpr int NTW = HID / JS / 16;          // 16-channel hidden tiles per wave: 4, 2, 2
    static constexpr int SW = 16 * NTW;                // hidden channels per slice
    static constexpr int TT = C == 64 ? 2 : 4;         // MFMA tiles (of 16 pixels) per unit
    static constexpr int TP = 16 * TT;                 // pixels per tile
    static constexpr int TS = TP + 4;                  // floats per LDS pixel row: vector-aligned, +4 banks per row
    static constexpr int NW = 8;                       // waves per workgroup
    static constexpr int NTHR = NW * 64;
    static constexpr int TG = NW / JS;                 // tile-major: tiles per workgroup iteration
    typedef typename Vec<TT>::type VT;
    // column of (hidden channel j, kind) in a partial row of hidden-channel sums: [slice][kind][SW]
    static __device__ __forceinline__ int col(int j, int kind) { return (j / SW) * 2 * SW + kind * SW + (j % SW); }
};

// Every MFMA here is the tied-operand asm form (common.h): with the builtin, hipcc (ROCm 7.2) rotates these fully unrolled accumulator
// sets through D != C register pairs and then re-writes the old C registers 4 wait states behind the MFMA that still reads them, and under
// pressure emits partially overlapping D / C / A ranges (tools/check_mfma_overlap.py refuses both).  The price: hipcc pads nothing around
// an asm statement, so every product loop is closed by drain() -- wait states for the 8-pass result, then an empty asm per accumulator
// that makes it opaque at that point -- before anything but the next accumulating MFMA touches the accumulators.
__device__ __forceinline__ void mfma(floatx4& acc, float a, float b) { kmu::mfma_tied(acc, a, b); }
template <int N>
__device__ __forceinline__ void drain(floatx4 (&x)[N]) {
    asm volatile("s_nop 15");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(x[i]));
}
template <int N, int M>
__device__ __forceinline__ void drain(floatx4 (&x)[N][M]) {
    asm volatile("s_nop 15");
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) asm volatile("" : "+v"(x[i][j]));
}
template <int N>
__device__ __forceinline__ void zero(floatx4 (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = floatx4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + __expf(-v)); }
__device__ __forceinline__ int fperm(int n) { return (n >> 2) + 4 * (n & 3); }

// Column sums of part[G][W] (fp32 rows of the previous kernel) in double and in a fixed order -> tot[W].  scratch: R * W doubles.
template <int W, int NTHR>
struct RowReduce {
    static constexpr int NCG = W / 4;
    static constexpr int R = NTHR / NCG >= 1 ? NTHR / NCG : 1;
    static constexpr int SCRATCH = R * W;   // doubles
    static __device__ __forceinline__ void run(const float* __restrict__ part, int G, double* scratch, double* tot) {
        static_assert(NCG <= NTHR, "row wider than the workgroup");
        if (KMU_FFN_DBG & 1) G = 1;
        const int tid = threadIdx.x, cg = tid % NCG, rg = tid / NCG;
        if (rg < R) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const float* p = part + 4 * cg;
            int r = rg;
            for (; r + 3 * R < G; r += 4 * R) {          // four independent loads in flight
                const floatx4 v0 = *reinterpret_cast<const floatx4*>(p + (size_t)r * W);
                const floatx4 v1 = *reinterpret_cast<const floatx4*>(p + (size_t)(r + R) * W);
                const floatx4 v2 = *reinterpret_cast<const floatx4*>(p + (size_t)(r + 2 * R) * W);
                const floatx4 v3 = *reinterpret_cast<const floatx4*>(p + (size_t)(r + 3 * R) * W);
                a0 += (double)v0[0]; a1 += (double)v0[1]; a2 += (double)v0[2]; a3 += (double)v0[3];
                a0 += (double)v1[0]; a1 += (double)v1[1]; a2 += (double)v1[2]; a3 += (double)v1[3];
                a0 += (double)v2[0]; a1 += (double)v2[1]; a2 += (double)v2[2]; a3 += (double)v2[3];
                a0 += (double)v3[0]; a1 += (double)v3[1]; a2 += (double)v3[2]; a3 += (double)v3[3];
            }
            for (; r < G; r += R) {
                const floatx4 v = *reinterpret_cast<const floatx4*>(p + (size_t)r * W);
                a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
            }
            double* s = scratch + (size_t)rg * W + 4 * cg;
            s[0] = a0, s[1] = a1, s[2] = a2, s[3] = a3;
        }
        __syncthreads();
        for (int t = tid; t < W; t += NTHR) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) s += scratch[(size_t)k * W + t];
            tot[t] = s;
        }
        __syncthreads();
    }
};

// BatchNorm as one fma per element: y = x * sc + sh.  F2, B2 and B3 must form sc / sh with the SAME roundings (the backward recomputes
// the forward's ReLU branch), hence one helper with an explicit fma instead of three expressions the compiler may contract differently.
__device__ __forceinline__ void bn_affine(float gamma, float beta, float mean, float rstd, float& sc, float& sh) {
    sc = gamma * rstd;
    sh = fmaf(-mean, sc, beta);
}

// mean / rstd from (sum, sum of squares) as csrc/bn_blend.hip does it (biased variance, double)
__device__ __forceinline__ void mean_rstd(double sa, double sq, double N, float eps, float& mean, float& rstd, double& var) {
    const double m = sa / N;
    var = sq / N - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// W1 [HID][C] rows j0+16nt+n, columns 16c'+4q+s -> w1r[nt][4c'+s]: the B operand of D' / the A operand of N (same registers)
template <int C>
__device__ __forceinline__ void load_w1r(const float* __restrict__ w1, int j0, int n, int q, float (&w1r)[Cfg<C>::NTW][4 * Cfg<C>::CC]) {
#pragma unroll
    for (int nt = 0; nt < Cfg<C>::NTW; ++nt)
#pragma unroll
        for (int cc = 0; cc < Cfg<C>::CC; ++cc) {
            const floatx4 v = *reinterpret_cast<const floatx4*>(w1 + (size_t)(j0 + 16 * nt + n) * C + 16 * cc + 4 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s) w1r[nt][4 * cc + s] = v[s];
        }
}
// W2 [C][HID] rows 16c'+4q+s, columns j0+16nt+n -> w2r[nt][4c'+s]: the B operand of dh' = dz2^T W2
template <int C>
__device__ __forceinline__ void load_w2r(const float* __restrict__ w2, int j0, int n, int q, float (&w2r)[Cfg<C>::NTW][4 * Cfg<C>::CC]) {
#pragma unroll
    for (int nt = 0; nt < Cfg<C>::NTW; ++nt)
#pragma unroll
        for (int k = 0; k < 4 * Cfg<C>::CC; ++k)
            w2r[nt][k] = w2[(size_t)(16 * (k >> 2) + 4 * q + (k & 3)) * Cfg<C>::HID + j0 + 16 * nt + n];
}

// Layout-A load of one tile: v[cc][s] = t[c = 16cc + 4q + s][p0 + TT f(n) ..]  (tt = t + (b*C)*P + p0)
template <int C>
__device__ __forceinline__ void load_a(const float* __restrict__ tt, int P, typename Cfg<C>::VT (&v)[Cfg<C>::CC][4], int n, int q) {
    typedef typename Cfg<C>::VT VT;
    const int poff = Cfg<C>::TT * fperm(n);
#pragma unroll
    for (int cc = 0; cc < Cfg<C>::CC; ++cc)
#pragma unroll
        for (int s = 0; s < 4; ++s) v[cc][s] = *reinterpret_cast<const VT*>(tt + (size_t)(16 * cc + 4 * q + s) * P + poff);
}

// z1'[pixel][j] = sum_c x[c][pixel] W1[j][c] for one tile (xa: its layout-A image in registers) and ONE 16-channel hidden tile
// (orientation D').  Accumulation order: chunks c' outer, s inner, the MFMA's own k = q innermost -- the SAME order as F2's.
// Also serves dh'[pixel][j] = sum_c dz2[c][pixel] W2[c][j] (xa = dz2, w1row = this tile's W2 fragment).
template <int C>
__device__ __forceinline__ void fc1_dp(const typename Cfg<C>::VT (&xa)[Cfg<C>::CC][4], const float (&w1row)[4 * Cfg<C>::CC],
                                       floatx4 (&z1)[Cfg<C>::TT]) {
    using K = Cfg<C>;
    zero(z1);
#pragma unroll
    for (int cc = 0; cc < K::CC; ++cc)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < K::TT; ++t) mfma(z1[t], xa[cc][s][t], w1row[4 * cc + s]);
    drain(z1);
}

template <int C>
__device__ __forceinline__ void tile_of(int tile, int P, int& b, int& p0) {
    const int tpp = P / Cfg<C>::TP;
    b = tile / tpp;
    p0 = (tile - b * tpp) * Cfg<C>::TP;
}


// Slice-major workgroup -> (slice, row group r).  The JS workgroups of a row group read the SAME tiles; with b -> r = b%8 + 8 (b / (8 JS)),
// slice = (b / 8) % JS they share blockIdx % 8, i.e. one XCD (and its L2) under the dispatcher's round-robin placement -- a speed
// matter only (measured without it: 4x the fabric traffic at C = 32).  Grids that are not a multiple of 8 JS use the plain mapping.
template <int JS>
__device__ __forceinline__ void slice_major(int& slice, int& r, int& R) {
    const int b = blockIdx.x, G = gridDim.x;
    R = G / JS;
    if (JS > 1 && G % (8 * JS) == 0) {
        r = (b & 7) + 8 * (b / (8 * JS));
        slice = (b >> 3) % JS;
    } else {
        slice = b % JS;
        r = b / JS;
    }
}

// ------------------------------------------------------------------------------------------------------------ F1 (slice-major)
// grid = JS * R workgroups: slice = blockIdx % JS, r = blockIdx / JS; the 8 waves take 8 different tiles.  part1 [R][JS][2][SW].
template <int C>
__global__ __launch_bounds__(Cfg<C>::NTHR) void ffn_stats1_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                                  float* __restrict__ part1, int P, int ntiles) {
    using K = Cfg<C>;
    __shared__ float comb[K::NW * 2 * K::SW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    int slice, r, R;
    slice_major<K::JS>(slice, r, R);
    const int j0 = slice * K::SW;
    float w1r[K::NTW][4 * K::CC];
    load_w1r<C>(w1, j0, n, q, w1r);
    float sa[K::NTW], sq[K::NTW];
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) sa[nt] = sq[nt] = 0.f;
    for (int tile = r * K::NW + wave; tile < ntiles; tile += R * K::NW) {
        int b, p0;
        tile_of<C>(tile, P, b, p0);
        typename K::VT xa[K::CC][4];
        load_a<C>(x + (size_t)b * C * P + p0, P, xa, n, q);
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt) {
            floatx4 z1[K::TT];
            fc1_dp<C>(xa, w1r[nt], z1);
#pragma unroll
            for (int t = 0; t < K::TT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = z1[t][i];
                    sa[nt] += v;
                    sq[nt] = fmaf(v, v, sq[nt]);
                }
        }
    }
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) {
        sa[nt] += __shfl_xor(sa[nt], 16);
        sa[nt] += __shfl_xor(sa[nt], 32);
        sq[nt] += __shfl_xor(sq[nt], 16);
        sq[nt] += __shfl_xor(sq[nt], 32);
        if (q == 0) {
            comb[wave * 2 * K::SW + 16 * nt + n] = sa[nt];
            comb[wave * 2 * K::SW + K::SW + 16 * nt + n] = sq[nt];
        }
    }
    __syncthreads();
    float* row = part1 + ((size_t)r * K::JS + slice) * 2 * K::SW;
    for (int t = threadIdx.x; t < 2 * K::SW; t += K::NTHR) {
        float s = comb[t];
#pragma unroll
        for (int w = 1; w < K::NW; ++w) s += comb[w * 2 * K::SW + t];
        row[t] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------ F2 (tile-major)
struct BnArgs {            // one BatchNorm2d's parameters / state
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    long long* nbt;
    float momentum, eps;
};

template <int C>
struct F2Lds {
    using K = Cfg<C>;
    static constexpr int RED_D = RowReduce<2 * K::HID, K::NTHR>::SCRATCH + 2 * K::HID;              // doubles (prologue)
    static constexpr int RED_F = K::JS == 1 ? 4 : K::NW * C * K::TS;                                // z2 partial tiles of the slices
    static constexpr int COMB_F = K::TG * 2 * C;                                                    // statistics rows of the tile groups
    static constexpr int MAIN_F = RED_F > COMB_F ? RED_F : COMB_F;
    static constexpr size_t UNION_B = (size_t)RED_D * 8 > (size_t)MAIN_F * 4 ? (size_t)RED_D * 8 : (size_t)MAIN_F * 4;
    static constexpr size_t BYTES = 2 * K::HID * 4 + UNION_B;
};

template <int C>
__global__ __launch_bounds__(Cfg<C>::NTHR) void ffn_fwd_main_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                                    const float* __restrict__ w2, BnArgs bn1,
                                                                    const float* __restrict__ part1, int R1, int training,
                                                                    float* __restrict__ stats1, float* __restrict__ z2,
                                                                    float* __restrict__ part2, float* __restrict__ h_tap, int P,
                                                                    int ntiles, double Ntot) {
    using K = Cfg<C>;
    typedef typename K::VT VT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc1 = reinterpret_cast<float*>(smem);
    float* sh1 = sc1 + K::HID;
    char* un = smem + 2 * K::HID * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int grp = wave / K::JS, slice = wave % K::JS, j0 = slice * K::SW;

    // this wave's weights first: their round trip overlaps the statistics prologue
    float w1r[K::NTW][4 * K::CC];
    load_w1r<C>(w1, j0, m, q, w1r);
    floatx4 w2r[K::CC][K::NTW];          // W2[16ct + m][j0 + 16nt + 4q + i]: A[row = c][k = q] of k-step (nt, i)
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt)
            w2r[ct][nt] = *reinterpret_cast<const floatx4*>(w2 + (size_t)(16 * ct + m) * K::HID + j0 + 16 * nt + 4 * q);
    // the first tile's x is requested BEFORE the prologue: its HBM round trip overlaps the fold of the partial rows.
    // xn[cc][s]: B[k = q][n = m] of the first product: channel 16cc + 4q + s, pixel p0 + TT m + t
    const int per_it = gridDim.x * K::TG, unit0 = blockIdx.x * K::TG + grp;
    VT xn[K::CC][4];
    if (unit0 < ntiles) {
        int b, p0;
        tile_of<C>(unit0, P, b, p0);
#pragma unroll
        for (int cc = 0; cc < K::CC; ++cc)
#pragma unroll
            for (int s = 0; s < 4; ++s) xn[cc][s] = *reinterpret_cast<const VT*>(x + ((size_t)b * C + 16 * cc + 4 * q + s) * P + p0 + K::TT * m);
    }

    if (training) {
        double* scratch = reinterpret_cast<double*>(un);
        double* tot = scratch + RowReduce<2 * K::HID, K::NTHR>::SCRATCH;
        RowReduce<2 * K::HID, K::NTHR>::run(part1, R1, scratch, tot);
        for (int j = tid; j < K::HID; j += K::NTHR) {
            float mean, rstd;
            double var;
            mean_rstd(tot[K::col(j, 0)], tot[K::col(j, 1)], Ntot, bn1.eps, mean, rstd, var);
            if (blockIdx.x == 0) {
                stats1[2 * j] = mean, stats1[2 * j + 1] = rstd;
                bn1.running_mean[j] = (1.f - bn1.momentum) * bn1.running_mean[j] + bn1.momentum * mean;
                const double unb = Ntot > 1.0 ? var * Ntot / (Ntot - 1.0) : var;
                bn1.running_var[j] = (1.f - bn1.momentum) * bn1.running_var[j] + bn1.momentum * (float)unb;
            }
            bn_affine(bn1.gamma[j], bn1.beta[j], mean, rstd, sc1[j], sh1[j]);
        }
        if (blockIdx.x == 0 && tid == 0 && bn1.nbt) *bn1.nbt += 1;
    } else {
        for (int j = tid; j < K::HID; j += K::NTHR) {
            const float mean = bn1.running_mean[j], rstd = 1.f / sqrtf(bn1.running_var[j] + bn1.eps);
            if (blockIdx.x == 0) stats1[2 * j] = mean, stats1[2 * j + 1] = rstd;
            bn_affine(bn1.gamma[j], bn1.beta[j], mean, rstd, sc1[j], sh1[j]);
        }
    }
    __syncthreads();
    floatx4 scv[K::NTW], shv[K::NTW];    // hidden channel j0 + 16nt + 4q + i
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) {
        scv[nt] = *reinterpret_cast<const floatx4*>(sc1 + j0 + 16 * nt + 4 * q);
        shv[nt] = *reinterpret_cast<const floatx4*>(sh1 + j0 + 16 * nt + 4 * q);
    }
    __syncthreads();                     // the union region turns from reduction scratch into tile scratch

    float* redz = reinterpret_cast<float*>(un);
    // statistics of z2: JS == 1: lane (m, q) owns channels 16ct + 4q + i of its pixels; JS > 1: thread e / e + JS*64 of the group own channel e>>4
    constexpr int NS = K::JS == 1 ? 4 * K::CC : 2;
    float s2a[NS], s2q[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) s2a[k] = s2q[k] = 0.f;
    const int gtid = tid - grp * K::JS * 64;             // thread index inside the tile group

    const int iters = (ntiles + per_it - 1) / per_it;
    for (int it = 0; it < iters; ++it) {
        const int tile = unit0 + it * per_it;
        const bool valid = tile < ntiles;        // uniform over the tile group
        int b = 0, p0 = 0;
        if (valid) tile_of<C>(tile, P, b, p0);
        floatx4 d2[K::CC][K::TT];
        if (valid) {
            floatx4 d1[K::NTW][K::TT];
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt) zero(d1[nt]);
#pragma unroll
            for (int cc = 0; cc < K::CC; ++cc)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) mfma(d1[nt][t], w1r[nt][4 * cc + s], xn[cc][s][t]);
            drain(d1);
            if (it + 1 < iters && tile + per_it < ntiles) {         // the next tile's x, behind this tile's last use of xn
                int bn, pn;
                tile_of<C>(tile + per_it, P, bn, pn);
#pragma unroll
                for (int cc = 0; cc < K::CC; ++cc)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        xn[cc][s] = *reinterpret_cast<const VT*>(x + ((size_t)bn * C + 16 * cc + 4 * q + s) * P + pn + K::TT * m);
            }
            // BatchNorm + ReLU on the registers; d1[nt][t][i]: hidden j0 + 16nt + 4q + i, pixel p0 + TT m + t
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int t = 0; t < K::TT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) d1[nt][t][i] = fmaxf(fmaf(d1[nt][t][i], scv[nt][i], shv[nt][i]), 0.f);
            if (h_tap) {                 // tests only: the hidden activation (its sign pattern is the ReLU branch the kernels take)
#pragma unroll
                for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        VT v;
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) v[t] = d1[nt][t][i];
                        *reinterpret_cast<VT*>(h_tap + ((size_t)b * K::HID + j0 + 16 * nt + 4 * q + i) * P + p0 + K::TT * m) = v;
                    }
            }
#pragma unroll
            for (int ct = 0; ct < K::CC; ++ct) zero(d2[ct]);
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) mfma(d2[ct][t], w2r[ct][nt][i], d1[nt][t][i]);
            drain(d2);
        }
        // d2[ct][t][i]: channel 16ct + 4q + i, pixel p0 + TT m + t
        if constexpr (K::JS == 1) {
            if (valid) {
#pragma unroll
                for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        VT v;
                        float a = 0.f, qq = 0.f;
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) {
                            v[t] = d2[ct][t][i];
                            a += v[t];
                            qq = fmaf(v[t], v[t], qq);
                        }
                        *reinterpret_cast<VT*>(z2 + ((size_t)b * C + 16 * ct + 4 * q + i) * P + p0 + K::TT * m) = v;
                        s2a[4 * ct + i] += a;
                        s2q[4 * ct + i] += qq;
                    }
            }
        } else {
            if (valid) {
#pragma unroll
                for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        VT v;
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) v[t] = d2[ct][t][i];
                        *reinterpret_cast<VT*>(redz + (size_t)(wave * C + 16 * ct + 4 * q + i) * K::TS + K::TT * m) = v;
                    }
            }
            __syncthreads();
            if (valid) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {                  // C * 16 vector outputs over the group's JS * 64 threads = 2 each
                    const int e = gtid + k * K::JS * 64, c = e >> 4, pg = e & 15;
                    const float* src = redz + (size_t)(grp * K::JS * C + c) * K::TS + K::TT * pg;
                    VT v = *reinterpret_cast<const VT*>(src);
#pragma unroll
                    for (int w = 1; w < K::JS; ++w) v += *reinterpret_cast<const VT*>(src + (size_t)w * C * K::TS);
                    *reinterpret_cast<VT*>(z2 + ((size_t)b * C + c) * P + p0 + K::TT * pg) = v;
#pragma unroll
                    for (int t = 0; t < K::TT; ++t) {
                        s2a[k] += v[t];
                        s2q[k] = fmaf(v[t], v[t], s2q[k]);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!part2) return;
    // per tile group one row of (sum, sum of squares) per channel, then the groups in a fixed order
    float* comb = reinterpret_cast<float*>(un);
    __syncthreads();
    if constexpr (K::JS == 1) {
#pragma unroll
        for (int k = 0; k < 4 * K::CC; ++k) {
            const float a = kmu::wave_reduce<kmu::OpSum, 16>(s2a[k]), qq = kmu::wave_reduce<kmu::OpSum, 16>(s2q[k]);
            if (m == 0) {
                const int c = 16 * (k >> 2) + 4 * q + (k & 3);
                comb[grp * 2 * C + c] = a;
                comb[grp * 2 * C + C + c] = qq;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float a = kmu::wave_reduce<kmu::OpSum, 16>(s2a[k]), qq = kmu::wave_reduce<kmu::OpSum, 16>(s2q[k]);
            const int e = gtid + k * K::JS * 64;
            if ((e & 15) == 0) comb[grp * 2 * C + (e >> 4)] = a, comb[grp * 2 * C + C + (e >> 4)] = qq;
        }
    }
    __syncthreads();
    float* row = part2 + (size_t)blockIdx.x * 2 * C;
    for (int t = tid; t < 2 * C; t += K::NTHR) {
        float s = comb[t];
#pragma unroll
        for (int g = 1; g < K::TG; ++g) s += comb[g * 2 * C + t];
        row[t] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------ F3
// out = x + a (z2 sc2 + sh2 - x): grid-stride over float4s; every workgroup folds the (small) partial rows of z2's statistics.
template <int C>
__global__ __launch_bounds__(256) void ffn_apply_kernel(const float* __restrict__ z2, const float* __restrict__ x, BnArgs bn2,
                                                        const float* __restrict__ alpha, const float* __restrict__ part2, int G2,
                                                        int training, float* __restrict__ stats2, float* __restrict__ out, int P4,
                                                        long total4, double Ntot) {
    __shared__ double red[RowReduce<2 * C, 256>::SCRATCH + 2 * C];
    __shared__ float sc2[C], sh2[C], av[C];
    const int tid = threadIdx.x;
    if (training) {
        double* tot = red + RowReduce<2 * C, 256>::SCRATCH;
        RowReduce<2 * C, 256>::run(part2, G2, red, tot);
        if (tid < C) {
            float mean, rstd;
            double var;
            mean_rstd(tot[tid], tot[C + tid], Ntot, bn2.eps, mean, rstd, var);
            if (blockIdx.x == 0) {
                stats2[2 * tid] = mean, stats2[2 * tid + 1] = rstd;
                bn2.running_mean[tid] = (1.f - bn2.momentum) * bn2.running_mean[tid] + bn2.momentum * mean;
                const double unb = Ntot > 1.0 ? var * Ntot / (Ntot - 1.0) : var;
                bn2.running_var[tid] = (1.f - bn2.momentum) * bn2.running_var[tid] + bn2.momentum * (float)unb;
            }
            bn_affine(bn2.gamma[tid], bn2.beta[tid], mean, rstd, sc2[tid], sh2[tid]);
            av[tid] = sigmoid_f(alpha[tid]);
        }
        if (blockIdx.x == 0 && tid == 0 && bn2.nbt) *bn2.nbt += 1;
    } else if (tid < C) {
        const float mean = bn2.running_mean[tid], rstd = 1.f / sqrtf(bn2.running_var[tid] + bn2.eps);
        if (blockIdx.x == 0) stats2[2 * tid] = mean, stats2[2 * tid + 1] = rstd;
        bn_affine(bn2.gamma[tid], bn2.beta[tid], mean, rstd, sc2[tid], sh2[tid]);
        av[tid] = sigmoid_f(alpha[tid]);
    }
    __syncthreads();
    for (long e = (long)blockIdx.x * 256 + tid; e < total4; e += (long)gridDim.x * 256) {
        const int c = (int)((e / P4) % C);
        const floatx4 zv = reinterpret_cast<const floatx4*>(z2)[e], xv = reinterpret_cast<const floatx4*>(x)[e];
        const float sc = sc2[c], sh = sh2[c], a = av[c];
        floatx4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = xv[k] + a * (fmaf(zv[k], sc, sh) - xv[k]);
        reinterpret_cast<floatx4*>(out)[e] = o;
    }
}

// ------------------------------------------------------------------------------------------------------------ B1
// per channel over all pixels: s0 = sum a g, s1 = sum a g z2hat, s2 = sum g (BN2(z2) - x)      -> q1 [G][3C]; 64-pixel tiles, 8 per iteration
template <int C>
__global__ __launch_bounds__(512) void ffn_bwd_red2_kernel(const float* __restrict__ g, const float* __restrict__ z2,
                                                           const float* __restrict__ x, const float* __restrict__ gamma2,
                                                           const float* __restrict__ beta2, const float* __restrict__ stats2,
                                                           const float* __restrict__ alpha, float* __restrict__ q1, int P, int ntiles64) {
    constexpr int CC = C / 16;
    __shared__ float comb[8 * 3 * C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    float mean[CC], rstd[CC], sc[CC], sh[CC], a[CC], s0[CC], s1[CC], s2[CC];
#pragma unroll
    for (int ct = 0; ct < CC; ++ct) {
        const int c = 16 * ct + r;
        mean[ct] = stats2[2 * c], rstd[ct] = stats2[2 * c + 1];
        bn_affine(gamma2[c], beta2[c], mean[ct], rstd[ct], sc[ct], sh[ct]);
        a[ct] = sigmoid_f(alpha[c]);
        s0[ct] = s1[ct] = s2[ct] = 0.f;
    }
    const int tpp = P >> 6;
    for (int tile = blockIdx.x * 8 + wave; tile < ntiles64; tile += gridDim.x * 8) {
        const int b = tile / tpp, p0 = (tile - b * tpp) << 6;
#pragma unroll
        for (int ct = 0; ct < CC; ++ct) {
            const size_t o = ((size_t)b * C + 16 * ct + r) * P + p0 + 4 * q;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const floatx4 gv = *reinterpret_cast<const floatx4*>(g + o + 16 * i), zv = *reinterpret_cast<const floatx4*>(z2 + o + 16 * i),
                              xv = *reinterpret_cast<const floatx4*>(x + o + 16 * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float that = (zv[k] - mean[ct]) * rstd[ct], f = fmaf(zv[k], sc[ct], sh[ct]), dz = a[ct] * gv[k];
                    s0[ct] += dz;
                    s1[ct] = fmaf(dz, that, s1[ct]);
                    s2[ct] = fmaf(gv[k], f - xv[k], s2[ct]);
                }
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CC; ++ct) {
        s0[ct] += __shfl_xor(s0[ct], 16), s0[ct] += __shfl_xor(s0[ct], 32);
        s1[ct] += __shfl_xor(s1[ct], 16), s1[ct] += __shfl_xor(s1[ct], 32);
        s2[ct] += __shfl_xor(s2[ct], 16), s2[ct] += __shfl_xor(s2[ct], 32);
        if (q == 0) {
            comb[wave * 3 * C + 16 * ct + r] = s0[ct];
            comb[wave * 3 * C + C + 16 * ct + r] = s1[ct];
            comb[wave * 3 * C + 2 * C + 16 * ct + r] = s2[ct];
        }
    }
    __syncthreads();
    float* row = q1 + (size_t)blockIdx.x * 3 * C;
    for (int t = threadIdx.x; t < 3 * C; t += 512) {
        float s = comb[t];
#pragma unroll
        for (int w = 1; w < 8; ++w) s += comb[w * 3 * C + t];
        row[t] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------ B2 / B3
// PASS 2 (slice-major): partial sums of BN1's backward + dW2 slabs.  PASS 3 (tile-major): dx + dW1 slabs.
template <int C, int PASS>
struct BLds {
    using K = Cfg<C>;
    static constexpr int W_IN = PASS == 2 ? 3 * C : 2 * K::HID;                                   // width of the partial rows read
    static constexpr int RED_D = RowReduce<W_IN, K::NTHR>::SCRATCH + W_IN;                        // doubles (prologue)
    static constexpr int T_F = K::NW * 16 * K::TS;                                                // PASS 3: dz1 of 16 hidden channels per wave
    static constexpr int COMB2_F = K::NW * C * K::SW > K::NW * 2 * K::SW ? K::NW * C * K::SW : K::NW * 2 * K::SW;   // PASS 2 epilogue
    static constexpr int COMB3_F = K::TG > 1 ? K::TG * C * K::HID : 4;                             // PASS 3 epilogue: slabs of the tile groups
    static constexpr int MAIN_F3 = T_F > COMB3_F ? T_F : COMB3_F;
    static constexpr int MAIN_F = PASS == 3 ? MAIN_F3 : COMB2_F;
    static constexpr size_t UNION_B = (size_t)RED_D * 8 > (size_t)MAIN_F * 4 ? (size_t)RED_D * 8 : (size_t)MAIN_F * 4;
    static constexpr int WL_F = PASS == 3 ? K::HID * (C + 4) : 0;                                 // W1 [j][c], row stride C + 4
    static constexpr int CONST_F = 4 * C + 5 * K::HID;                                            // cst2 [C][4]; per hidden channel: 5 arrays
    static constexpr size_t BYTES = (size_t)(CONST_F + WL_F) * 4 + UNION_B;
};

struct BArgs {
    const float *g, *x, *z2, *w1, *w2;
    const float *gamma1, *beta1, *stats1, *gamma2, *stats2, *alpha;
    const float* part_in;     // PASS 2: q1 [Gin][3C]; PASS 3: q2 [Gin][2 HID]
    int Gin, training;
    float* cst2g;             // [C][4] {A, B, D, 1 - a}: written by PASS 2 (workgroup 0), read by PASS 3
    float *d_gamma2, *d_beta2, *d_alpha;    // PASS 2 (workgroup 0)
    float *d_gamma1, *d_beta1;              // PASS 3 (workgroup 0)
    float* q2;                // PASS 2 out [R][JS][2][SW]
    float* slab;              // PASS 2: dW2 slabs [R][C][HID]; PASS 3: dW1 slabs [G][HID][C]
    float* dx;                // PASS 3
    int P, ntiles;
    double Ntot;
};

template <int C, int PASS>
__global__ __launch_bounds__(Cfg<C>::NTHR) void ffn_bwd_kernel(BArgs a) {
    using K = Cfg<C>;
    using L = BLds<C, PASS>;
    typedef typename K::VT VT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    floatx4* cst2 = reinterpret_cast<floatx4*>(smem);                 // [C]
    float* hc = reinterpret_cast<float*>(smem) + 4 * C;               // [5][HID]: sc1, sh1, then (mean1, rstd1, -) or (k1, E, F)
    float* wl = hc + 5 * K::HID;                                      // PASS 3: W1 [HID][C + 4]
    char* un = smem + (size_t)(L::CONST_F + L::WL_F) * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    // PASS 2: slice = blockIdx % JS, the waves take different tiles; PASS 3: the waves of a tile group take its slices
    int sm_slice = 0, r = 0, R = 0;
    if constexpr (PASS == 2) slice_major<K::JS>(sm_slice, r, R);
    const int slice = PASS == 2 ? sm_slice : wave % K::JS;
    const int grp = PASS == 2 ? wave : wave / K::JS;                  // which of this workgroup's concurrent tiles
    const int j0 = slice * K::SW;
    const int P = a.P;

    float w1r[K::NTW][4 * K::CC], w2r[K::NTW][4 * K::CC];
    load_w1r<C>(a.w1, j0, n, q, w1r);
    load_w2r<C>(a.w2, j0, n, q, w2r);
    const int per_it = PASS == 2 ? R * K::NW : gridDim.x * K::TG;
    const int unit0 = PASS == 2 ? r * K::NW + wave : blockIdx.x * K::TG + grp;
    const int iters = (a.ntiles + per_it - 1) / per_it;
    if constexpr (PASS == 3) {
        for (int e = tid; e < K::HID * C; e += K::NTHR) {
            const int j = e / C, c = e - j * C;
            wl[j * (C + 4) + c] = a.w1[e];
        }
    }

    // ---- prologue: fold the partial rows of the kernel before this one into per-channel constants
    {
        double* scratch = reinterpret_cast<double*>(un);
        double* tot = scratch + RowReduce<L::W_IN, K::NTHR>::SCRATCH;
        RowReduce<L::W_IN, K::NTHR>::run(a.part_in, a.Gin, scratch, tot);
        if constexpr (PASS == 2) {
            for (int c = tid; c < C; c += K::NTHR) {
                const float mean2 = a.stats2[2 * c], rstd2 = a.stats2[2 * c + 1], av = sigmoid_f(a.alpha[c]);
                const float k2 = a.gamma2[c] * rstd2;
                const float m0 = a.training ? (float)(tot[c] / a.Ntot) : 0.f, m1 = a.training ? (float)(tot[C + c] / a.Ntot) : 0.f;
                // dz2 = k2 (a g - m0 - z2hat m1),  z2hat = (z2 - mean2) rstd2
                const floatx4 k = {k2 * av, -k2 * m1 * rstd2, k2 * (m1 * rstd2 * mean2 - m0), 1.f - av};
                cst2[c] = k;
                if (blockIdx.x == 0) {
                    reinterpret_cast<floatx4*>(a.cst2g)[c] = k;
                    a.d_gamma2[c] = (float)tot[C + c];
                    a.d_beta2[c] = (float)tot[c];
                    a.d_alpha[c] = (float)tot[2 * C + c] * av * (1.f - av);
                }
            }
            for (int j = tid; j < K::HID; j += K::NTHR) {
                const float mean = a.stats1[2 * j], rstd = a.stats1[2 * j + 1];
                bn_affine(a.gamma1[j], a.beta1[j], mean, rstd, hc[j], hc[K::HID + j]);
                hc[2 * K::HID + j] = mean, hc[3 * K::HID + j] = rstd;
            }
        } else {
            for (int c = tid; c < C; c += K::NTHR) cst2[c] = reinterpret_cast<const floatx4*>(a.cst2g)[c];
            for (int j = tid; j < K::HID; j += K::NTHR) {
                const float mean = a.stats1[2 * j], rstd = a.stats1[2 * j + 1];
                float k1, sh;
                bn_affine(a.gamma1[j], a.beta1[j], mean, rstd, k1, sh);
                const float n0 = a.training ? (float)(tot[K::col(j, 0)] / a.Ntot) : 0.f, n1 = a.training ? (float)(tot[K::col(j, 1)] / a.Ntot) : 0.f;
                // dz1 = k1 (dr - n0 - z1hat n1),  z1hat = (z1 - mean) rstd
                hc[j] = k1, hc[K::HID + j] = sh;
                hc[2 * K::HID + j] = k1, hc[3 * K::HID + j] = -k1 * n1 * rstd, hc[4 * K::HID + j] = k1 * (n1 * rstd * mean - n0);
                if (blockIdx.x == 0) a.d_gamma1[j] = (float)tot[K::col(j, 1)], a.d_beta1[j] = (float)tot[K::col(j, 0)];
            }
        }
    }
    __syncthreads();
    // per-lane constants of hidden channel j0 + 16nt + n
    float sc1v[K::NTW], sh1v[K::NTW], c2v[K::NTW], c3v[K::NTW], c4v[K::NTW];
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) {
        const int j = j0 + 16 * nt + n;
        sc1v[nt] = hc[j], sh1v[nt] = hc[K::HID + j], c2v[nt] = hc[2 * K::HID + j], c3v[nt] = hc[3 * K::HID + j];
        c4v[nt] = PASS == 3 ? hc[4 * K::HID + j] : 0.f;
    }
    __syncthreads();                     // the union region turns from reduction scratch into tile scratch

    float* T = reinterpret_cast<float*>(un);
    floatx4 acc[K::CC][K::NTW];          // PASS 2: dW2 tile (rows c = 16ct + 4q + i, column j); PASS 3: dW1 tile (rows j, column c = 16ct + n)
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct) zero(acc[ct]);
    float t0[K::NTW], t1[K::NTW];
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) t0[nt] = t1[nt] = 0.f;
    const int poff = K::TT * fperm(n);
    // PASS 3: this wave's share of the dx tile of its group: DN pixel-tiles (own_t0 ..) of channel tile own_ct
    constexpr int DN = K::JS == 1 ? K::TT : 2;
    typedef typename Vec<DN>::type DV;
    const int own_ct = K::JS == 1 ? 0 : slice % K::CC;
    const int own_t0 = (K::JS == 1 || C == 64) ? 0 : 2 * (slice / K::CC);
    const bool own = K::JS == 1 || C == 32 || slice < K::CC;          // C = 64: four waves (one per SIMD) carry the four channel tiles
    float* Tg = T + (size_t)grp * K::JS * 16 * K::TS;                 // this tile group's 16 JS rows

    for (int it = 0; it < iters; ++it) {
        const int tile = unit0 + it * per_it;
        const bool valid = tile < a.ntiles;              // PASS 3: uniform over the tile group (barriers below are workgroup-wide)
        if (PASS == 2 && !valid) continue;
        int b = 0, p0 = 0;
        if (valid) tile_of<C>(tile, P, b, p0);
        const size_t base = (size_t)b * C * P + p0;
        floatx4 d[DN];
        zero(d);
        // LDS reads of per-channel constants and of W1 are loop invariant; hoisted out of the tile loop they would pin 64+ registers
        // (and spill), so their offsets are made opaque once per tile
        int cofs = 4 * q, wofs = 4 * q * (C + 4) + n;
        asm volatile("" : "+v"(cofs), "+v"(wofs));
        // phase A: z1 = W1 x and dh = W2^T dz2 of every hidden tile of this wave, from ONE layout-A load of x, g, z2
        // (dz2 = A_c g + (B_c z2 + D_c)), then the element-wise step; hz[nt][t][i]: hidden j0 + 16nt + n, pixel p0 + 4TT i + TT q + t
        floatx4 hz[K::NTW][K::TT];
        if (valid) {
            VT xa[K::CC][4], za[K::CC][4];
            load_a<C>(a.x + base, P, xa, n, q);
#pragma unroll
            for (int cc = 0; cc < K::CC; ++cc) {
                // one 16-channel chunk of g and z2 at a time: a "memory" fence keeps hipcc from issuing every load of the tile at
                // once (it would, and then spills: all of x, g, z2 in two layouts do not fit beside the accumulators)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const size_t o = base + (size_t)(16 * cc + 4 * q + s) * P + poff;
                    const VT gv = *reinterpret_cast<const VT*>(a.g + o), zv = *reinterpret_cast<const VT*>(a.z2 + o);
                    const floatx4 k = cst2[16 * cc + s + cofs];
#pragma unroll
                    for (int t = 0; t < K::TT; ++t) za[cc][s][t] = fmaf(k[0], gv[t], fmaf(k[1], zv[t], k[2]));   // dz2
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt) {
                floatx4 dh[K::TT];
                fc1_dp<C>(xa, w1r[nt], hz[nt]);
                fc1_dp<C>(za, w2r[nt], dh);
#pragma unroll
                for (int t = 0; t < K::TT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float z = hz[nt][t][i], pre = fmaf(z, sc1v[nt], sh1v[nt]);
                        const bool on = pre > 0.f;
                        const float dr = on ? dh[t][i] : 0.f;
                        if constexpr (PASS == 2) {
                            const float that = (z - c2v[nt]) * c3v[nt];
                            t0[nt] += dr;
                            t1[nt] = fmaf(dr, that, t1[nt]);
                            hz[nt][t][i] = on ? pre : 0.f;                                   // h
                        } else {
                            hz[nt][t][i] = fmaf(c2v[nt], dr, fmaf(c3v[nt], z, c4v[nt]));     // dz1
                        }
                    }
            }
        }
        // phase B: the weight-gradient product over this tile's pixels -- k-step (t, i) <-> pixels p0 + 4TT i + TT q + t, q = the MFMA's
        // k; its other operand is dz2 (PASS 2) / x (PASS 3) in layout B -- and, PASS 3, dx through LDS, 16 hidden channels per round
        asm volatile("" ::: "memory");           // phase B's loads stay below phase A's
        VT vb[K::CC][4];
        if (valid) {
#pragma unroll
            for (int ct = 0; ct < K::CC; ++ct) {
                const size_t o = base + (size_t)(16 * ct + n) * P + K::TT * q;
                if constexpr (PASS == 2) {
                    const floatx4 cb = cst2[16 * ct + n];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const VT gv = *reinterpret_cast<const VT*>(a.g + o + 4 * K::TT * i), zv = *reinterpret_cast<const VT*>(a.z2 + o + 4 * K::TT * i);
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) vb[ct][i][t] = fmaf(cb[0], gv[t], fmaf(cb[1], zv[t], cb[2]));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) vb[ct][i] = *reinterpret_cast<const VT*>(a.x + o + 4 * K::TT * i);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt) {
            if (valid) {
#pragma unroll
                for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                    for (int t = 0; t < K::TT; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if constexpr (PASS == 2) mfma(acc[ct][nt], vb[ct][i][t], hz[nt][t][i]);     // dW2[c][j] += dz2 h
                            else mfma(acc[ct][nt], hz[nt][t][i], vb[ct][i][t]);                         // dW1[j][c] += dz1 x
                        }
            }
            if constexpr (PASS == 3) {
                // dz1 of these 16 hidden channels -> T[slice rows][pixel]: its transpose-through-LDS for dx[pixel][c] += dz1[pixel][j] W1[j][c]
                if (valid) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        VT v;
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) v[t] = hz[nt][t][i];
                        *reinterpret_cast<VT*>(Tg + (size_t)(slice * 16 + n) * K::TS + 4 * K::TT * i + K::TT * q) = v;
                    }
                }
                if constexpr (K::JS == 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                } else {
                    __syncthreads();
                }
                if (valid && own) {
#pragma unroll
                    for (int sl = 0; sl < K::JS; ++sl) {
                        if (sl & 1) asm volatile("" ::: "memory");           // at most two slices' operand reads in flight
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int row = sl * 16 + 4 * q + s;                                  // hidden channel sl*SW + 16nt + 4q + s
                            const DV av = *reinterpret_cast<const DV*>(Tg + (size_t)row * K::TS + poff + own_t0);
                            const float bv = wl[(sl * K::SW + 16 * nt + s) * (C + 4) + 16 * own_ct + wofs];
#pragma unroll
                            for (int k = 0; k < DN; ++k) mfma(d[k], av[k], bv);
                        }
                    }
                }
                if constexpr (K::JS == 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else {
                    __syncthreads();             // T is rewritten by the next round
                }
            }
        }
        if constexpr (PASS == 3) {
            if (valid && own) {
                drain(d);
                // d[k][i]: pixel p0 + 4TT i + TT q + own_t0 + k, channel 16 own_ct + n
                const size_t o = base + (size_t)(16 * own_ct + n) * P + K::TT * q + own_t0;
                const float oma = cst2[16 * own_ct + n][3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const DV gv = *reinterpret_cast<const DV*>(a.g + o + 4 * K::TT * i);
                    DV v;
#pragma unroll
                    for (int k = 0; k < DN; ++k) v[k] = fmaf(oma, gv[k], d[k][i]);
                    *reinterpret_cast<DV*>(a.dx + o + 4 * K::TT * i) = v;
                }
            }
        }
    }

    // ---- epilogue: partial row of BN1's backward sums (PASS 2) and this workgroup's weight-gradient slab
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct) drain(acc[ct]);
    __syncthreads();
    if constexpr (PASS == 2) {
        // the 8 waves hold sums over different tiles of the SAME slice: add them in a fixed order
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt) {
            t0[nt] += __shfl_xor(t0[nt], 16), t0[nt] += __shfl_xor(t0[nt], 32);
            t1[nt] += __shfl_xor(t1[nt], 16), t1[nt] += __shfl_xor(t1[nt], 32);
            if (q == 0) T[wave * 2 * K::SW + 16 * nt + n] = t0[nt], T[wave * 2 * K::SW + K::SW + 16 * nt + n] = t1[nt];
        }
        __syncthreads();
        float* row = a.q2 + ((size_t)r * K::JS + slice) * 2 * K::SW;
        for (int t = tid; t < 2 * K::SW; t += K::NTHR) {
            float s = T[t];
#pragma unroll
            for (int w = 1; w < K::NW; ++w) s += T[w * 2 * K::SW + t];
            row[t] = s;
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) T[(size_t)wave * C * K::SW + (16 * ct + 4 * q + i) * K::SW + 16 * nt + n] = acc[ct][nt][i];
        __syncthreads();
        float* slab = a.slab + (size_t)r * C * K::HID;
        for (int e = tid; e < C * K::SW; e += K::NTHR) {
            float s = T[e];
#pragma unroll
            for (int w = 1; w < K::NW; ++w) s += T[(size_t)w * C * K::SW + e];
            slab[(e / K::SW) * K::HID + j0 + (e % K::SW)] = s;
        }
    } else {
        float* slab = a.slab + (size_t)blockIdx.x * C * K::HID;
        if constexpr (K::TG > 1) {       // the tile groups hold sums of different tiles for the same (j, c): fixed-order add
#pragma unroll
            for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) T[(size_t)grp * C * K::HID + (j0 + 16 * nt + 4 * q + i) * C + 16 * ct + n] = acc[ct][nt][i];
            __syncthreads();
            for (int e = tid; e < C * K::HID; e += K::NTHR) {
                float s = T[e];
#pragma unroll
                for (int g2 = 1; g2 < K::TG; ++g2) s += T[(size_t)g2 * C * K::HID + e];
                slab[e] = s;
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) slab[(j0 + 16 * nt + 4 * q + i) * C + 16 * ct + n] = acc[ct][nt][i];
        }
    }
}

// ============================================================================================================ EnhancedViMBlock tail
// out = x + s[b] * ( W2 gelu(W0 n + b0) + b2 ),  n = TripleNorm(x)  (KM_UNetV3_SH.py:120-124 ffn, :147-150; s = DropPath's per-sample
// factor or NULL).  The same recompute scheme without the BatchNorms: the 4C-wide hidden tensor is never stored -- ONE launch forward
// (orientation N: W0 n -> GELU on the registers -> W2, bias / DropPath factor / residual in the epilogue) and ONE backward (D'):
//   pre = W0 n + b0,  df = s g,  dh = (W2^T df) gelu'(pre),  dn = W0^T dh,  dW2 = sum df gelu(pre)^T,  dW0 = sum dh n^T,  db0 = sum dh,
//   db2 = sum df  -- no statistics stand between the products, so nothing needs a second pass.  TripleNorm keeps its own kernels
//   (csrc/triple_norm.hip) in front of / behind these.
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
    return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.39894228040143268f * __expf(-0.5f * v * v);
}

template <int C>
struct TailFwdLds {
    using K = Cfg<C>;
    static constexpr size_t BYTES = K::JS == 1 ? 16 : (size_t)K::NW * C * K::TS * 4;
};

template <int C>
__global__ __launch_bounds__(Cfg<C>::NTHR) void tail_ffn_fwd_kernel(const float* __restrict__ nrm, const float* __restrict__ x,
                                                                    const float* __restrict__ w0, const float* __restrict__ b0,
                                                                    const float* __restrict__ w2, const float* __restrict__ b2,
                                                                    const float* __restrict__ sdp, float* __restrict__ out, int P,
                                                                    int ntiles) {
    using K = Cfg<C>;
    typedef typename K::VT VT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* redz = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int grp = wave / K::JS, slice = wave % K::JS, j0 = slice * K::SW;
    float w1r[K::NTW][4 * K::CC];
    load_w1r<C>(w0, j0, m, q, w1r);
    floatx4 w2r[K::CC][K::NTW], b0v[K::NTW];
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) {
        b0v[nt] = *reinterpret_cast<const floatx4*>(b0 + j0 + 16 * nt + 4 * q);
#pragma unroll
        for (int ct = 0; ct < K::CC; ++ct)
            w2r[ct][nt] = *reinterpret_cast<const floatx4*>(w2 + (size_t)(16 * ct + m) * K::HID + j0 + 16 * nt + 4 * q);
    }
    const int gtid = tid - grp * K::JS * 64;
    const int per_it = gridDim.x * K::TG, unit0 = blockIdx.x * K::TG + grp;
    const int iters = (ntiles + per_it - 1) / per_it;
    for (int it = 0; it < iters; ++it) {
        const int tile = unit0 + it * per_it;
        const bool valid = tile < ntiles;
        int b = 0, p0 = 0;
        if (valid) tile_of<C>(tile, P, b, p0);
        const float sb = (valid && sdp) ? sdp[b] : 1.f;
        floatx4 d2[K::CC][K::TT];
        if (valid) {
            const float* nt_ = nrm + (size_t)b * C * P + p0;
            floatx4 d1[K::NTW][K::TT];
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt) zero(d1[nt]);
#pragma unroll
            for (int cc = 0; cc < K::CC; ++cc) {
                VT xa[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) xa[s] = *reinterpret_cast<const VT*>(nt_ + (size_t)(16 * cc + 4 * q + s) * P + K::TT * m);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) mfma(d1[nt][t], w1r[nt][4 * cc + s], xa[s][t]);
            }
            drain(d1);
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int t = 0; t < K::TT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) d1[nt][t][i] = gelu_f(d1[nt][t][i] + b0v[nt][i]);
#pragma unroll
            for (int ct = 0; ct < K::CC; ++ct) zero(d2[ct]);
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) mfma(d2[ct][t], w2r[ct][nt][i], d1[nt][t][i]);
            drain(d2);
        }
        if constexpr (K::JS == 1) {
            if (valid) {
#pragma unroll
                for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = 16 * ct + 4 * q + i;
                        const size_t o = ((size_t)b * C + c) * P + p0 + K::TT * m;
                        const VT xv = *reinterpret_cast<const VT*>(x + o);
                        const float bb = b2[c];
                        VT v;
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) v[t] = fmaf(sb, d2[ct][t][i] + bb, xv[t]);
                        *reinterpret_cast<VT*>(out + o) = v;
                    }
            }
        } else {
            if (valid) {
#pragma unroll
                for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        VT v;
#pragma unroll
                        for (int t = 0; t < K::TT; ++t) v[t] = d2[ct][t][i];
                        *reinterpret_cast<VT*>(redz + (size_t)(wave * C + 16 * ct + 4 * q + i) * K::TS + K::TT * m) = v;
                    }
            }
            __syncthreads();
            if (valid) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = gtid + k * K::JS * 64, c = e >> 4, pg = e & 15;
                    const float* src = redz + (size_t)(grp * K::JS * C + c) * K::TS + K::TT * pg;
                    VT v = *reinterpret_cast<const VT*>(src);
#pragma unroll
                    for (int w = 1; w < K::JS; ++w) v += *reinterpret_cast<const VT*>(src + (size_t)w * C * K::TS);
                    const size_t o = ((size_t)b * C + c) * P + p0 + K::TT * pg;
                    const VT xv = *reinterpret_cast<const VT*>(x + o);
                    const float bb = b2[c];
#pragma unroll
                    for (int t = 0; t < K::TT; ++t) v[t] = fmaf(sb, v[t] + bb, xv[t]);
                    *reinterpret_cast<VT*>(out + o) = v;
                }
            }
            __syncthreads();
        }
    }
}

template <int C>
struct TailBwdLds {
    using K = Cfg<C>;
    static constexpr int T_F = K::NW * 16 * K::TS;
    static constexpr int COMB_F = K::TG > 1 ? K::TG * (C * K::HID + K::HID + C) : 4;
    static constexpr int MAIN_F = T_F > COMB_F ? T_F : COMB_F;
    static constexpr int WL_F = K::HID * (C + 4);
    static constexpr size_t BYTES = (size_t)(WL_F + MAIN_F) * 4;
};

struct TailBArgs {
    const float *nrm, *g, *w0, *b0, *w2, *sdp;
    float *dn, *slab_w0, *slab_w2, *rows_b;     // rows_b [G][HID + C]: db0 | db2 partial sums
    int P, ntiles;
};

template <int C>
__global__ __launch_bounds__(Cfg<C>::NTHR) void tail_ffn_bwd_kernel(TailBArgs a) {
    using K = Cfg<C>;
    typedef typename K::VT VT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);                       // W0 [HID][C + 4]
    float* T = wl + TailBwdLds<C>::WL_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    const int slice = wave % K::JS, grp = wave / K::JS, j0 = slice * K::SW;
    const int P = a.P;
    float w1r[K::NTW][4 * K::CC], w2r[K::NTW][4 * K::CC], b0v[K::NTW];
    load_w1r<C>(a.w0, j0, n, q, w1r);
    load_w2r<C>(a.w2, j0, n, q, w2r);
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) b0v[nt] = a.b0[j0 + 16 * nt + n];
    for (int e = tid; e < K::HID * C; e += K::NTHR) {
        const int j = e / C, c = e - j * C;
        wl[j * (C + 4) + c] = a.w0[e];
    }
    __syncthreads();
    floatx4 acc0[K::CC][K::NTW], acc2[K::CC][K::NTW];     // dW0 tile (rows j, column c = 16ct + n); dW2 tile (rows c = 16ct + 4q + i, column j)
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct) zero(acc0[ct]), zero(acc2[ct]);
    float sb0[K::NTW], sb2[K::CC];
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) sb0[nt] = 0.f;
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct) sb2[ct] = 0.f;
    const int poff = K::TT * fperm(n);
    constexpr int DN = K::JS == 1 ? K::TT : 2;
    typedef typename Vec<DN>::type DV;
    const int own_ct = K::JS == 1 ? 0 : slice % K::CC;
    const int own_t0 = (K::JS == 1 || C == 64) ? 0 : 2 * (slice / K::CC);
    const bool own = K::JS == 1 || C == 32 || slice < K::CC;
    float* Tg = T + (size_t)grp * K::JS * 16 * K::TS;
    const int per_it = gridDim.x * K::TG, unit0 = blockIdx.x * K::TG + grp;
    const int iters = (a.ntiles + per_it - 1) / per_it;
    for (int it = 0; it < iters; ++it) {
        const int tile = unit0 + it * per_it;
        const bool valid = tile < a.ntiles;
        int b = 0, p0 = 0;
        if (valid) tile_of<C>(tile, P, b, p0);
        const size_t base = (size_t)b * C * P + p0;
        const float sb = (valid && a.sdp) ? a.sdp[b] : 1.f;
        int wofs = 4 * q * (C + 4) + n;
        asm volatile("" : "+v"(wofs));
        floatx4 d[DN];
        zero(d);
        VT na[K::CC][4], fa[K::CC][4];       // n and df = s g in layout A (operands of the two products that make pre / dh)
        if (valid) {
            load_a<C>(a.nrm + base, P, na, n, q);
            load_a<C>(a.g + base, P, fa, n, q);
#pragma unroll
            for (int cc = 0; cc < K::CC; ++cc)
#pragma unroll
                for (int s = 0; s < 4; ++s) fa[cc][s] *= sb;
        }
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt) {
            floatx4 pre[K::TT], dh[K::TT];
            if (valid) {
                fc1_dp<C>(na, w1r[nt], pre);
                fc1_dp<C>(fa, w2r[nt], dh);
                // pre / dh [t][i]: hidden j0 + 16nt + n, pixel p0 + 4TT i + TT q + t
#pragma unroll
                for (int t = 0; t < K::TT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float z = pre[t][i] + b0v[nt];
                        const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752f));     // one erf serves gelu and gelu'
                        const float dz = dh[t][i] * fmaf(z * 0.39894228040143268f, __expf(-0.5f * z * z), cdf);
                        pre[t][i] = z * cdf;         // a = gelu(pre)
                        dh[t][i] = dz;               // d loss / d pre
                        sb0[nt] += dz;
                    }
                // weight gradients: the layout-B operands (n, df) are re-read per hidden tile, one 16-channel tile at a time (L1 / L2
                // hits): holding them across the hidden tiles costs 32 CC registers and spills
#pragma unroll
                for (int ct = 0; ct < K::CC; ++ct) {
                    const size_t o = base + (size_t)(16 * ct + n) * P + K::TT * q;
                    asm volatile("" ::: "memory");
                    VT nb[4], fb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        nb[i] = *reinterpret_cast<const VT*>(a.nrm + o + 4 * K::TT * i);
                        fb[i] = *reinterpret_cast<const VT*>(a.g + o + 4 * K::TT * i) * sb;
                        if (nt == 0) {
#pragma unroll
                            for (int t = 0; t < K::TT; ++t) sb2[ct] += fb[i][t];
                        }
                    }
#pragma unroll
                    for (int t = 0; t < K::TT; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            mfma(acc2[ct][nt], fb[i][t], pre[t][i]);        // dW2[c][j] += df a
                            mfma(acc0[ct][nt], dh[t][i], nb[i][t]);         // dW0[j][c] += dh n
                        }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    VT v;
#pragma unroll
                    for (int t = 0; t < K::TT; ++t) v[t] = dh[t][i];
                    *reinterpret_cast<VT*>(Tg + (size_t)(slice * 16 + n) * K::TS + 4 * K::TT * i + K::TT * q) = v;
                }
            }
            if constexpr (K::JS == 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
                __syncthreads();
            }
            if (valid && own) {
#pragma unroll
                for (int sl = 0; sl < K::JS; ++sl) {
                    if (sl & 1) asm volatile("" ::: "memory");
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int row = sl * 16 + 4 * q + s;
                        const DV av = *reinterpret_cast<const DV*>(Tg + (size_t)row * K::TS + poff + own_t0);
                        const float bv = wl[(sl * K::SW + 16 * nt + s) * (C + 4) + 16 * own_ct + wofs];
#pragma unroll
                        for (int k = 0; k < DN; ++k) mfma(d[k], av[k], bv);
                    }
                }
            }
            if constexpr (K::JS == 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                __syncthreads();
            }
        }
        if (valid && own) {
            drain(d);
            const size_t o = base + (size_t)(16 * own_ct + n) * P + K::TT * q + own_t0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                DV v;
#pragma unroll
                for (int k = 0; k < DN; ++k) v[k] = d[k][i];
                *reinterpret_cast<DV*>(a.dn + o + 4 * K::TT * i) = v;
            }
        }
    }
    // ---- epilogue: slabs of both weight gradients and the bias-gradient row of this workgroup
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct) drain(acc0[ct]), drain(acc2[ct]);
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) sb0[nt] += __shfl_xor(sb0[nt], 16), sb0[nt] += __shfl_xor(sb0[nt], 32);
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct) sb2[ct] += __shfl_xor(sb2[ct], 16), sb2[ct] += __shfl_xor(sb2[ct], 32);
    __syncthreads();
    constexpr int SL = C * K::HID, ROWF = K::HID + C;
    float* s0 = a.slab_w0 + (size_t)blockIdx.x * SL;
    float* s2 = a.slab_w2 + (size_t)blockIdx.x * SL;
    float* rb = a.rows_b + (size_t)blockIdx.x * ROWF;
    if constexpr (K::TG > 1) {
        // the tile groups hold sums over different tiles: dW0 first, then dW2 and the bias sums, through the same LDS region
        float* cg = T + (size_t)grp * (SL + ROWF);
#pragma unroll
        for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) cg[(j0 + 16 * nt + 4 * q + i) * C + 16 * ct + n] = acc0[ct][nt][i];
        if (q == 0) {
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt) cg[SL + j0 + 16 * nt + n] = sb0[nt];
            if (slice == 0) {
#pragma unroll
                for (int ct = 0; ct < K::CC; ++ct) cg[SL + K::HID + 16 * ct + n] = sb2[ct];
            }
        }
        __syncthreads();
        for (int e = tid; e < SL + ROWF; e += K::NTHR) {
            float v = T[e];
#pragma unroll
            for (int g2 = 1; g2 < K::TG; ++g2) v += T[(size_t)g2 * (SL + ROWF) + e];
            if (e < SL) s0[e] = v;
            else rb[e - SL] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) cg[(16 * ct + 4 * q + i) * K::HID + j0 + 16 * nt + n] = acc2[ct][nt][i];
        __syncthreads();
        for (int e = tid; e < SL; e += K::NTHR) {
            float v = T[e];
#pragma unroll
            for (int g2 = 1; g2 < K::TG; ++g2) v += T[(size_t)g2 * (SL + ROWF) + e];
            s2[e] = v;
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s0[(j0 + 16 * nt + 4 * q + i) * C + 16 * ct + n] = acc0[ct][nt][i];
                    s2[(16 * ct + 4 * q + i) * K::HID + j0 + 16 * nt + n] = acc2[ct][nt][i];
                }
        if (q == 0) {
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt) rb[j0 + 16 * nt + n] = sb0[nt];
            if (slice == 0) {
#pragma unroll
                for (int ct = 0; ct < K::CC; ++ct) rb[K::HID + 16 * ct + n] = sb2[ct];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ host side
inline int rows_for(int iters, int gmax) {      // largest workgroup count <= gmax that splits `iters` iterations evenly
    if (iters <= gmax) return iters;
    const int per = (iters + gmax - 1) / gmax;
    return (iters + per - 1) / per;
}
template <int C>
struct Plan {
    int ntiles, G, R;      // tiles; tile-major workgroups (= rows of part2 / slab_w1); slice-major rows (part1, q2, slab_w2), grid = JS * R
    Plan(int B, int P) {
        using K = Cfg<C>;
        ntiles = B * (P / K::TP);
        G = rows_for((ntiles + K::TG - 1) / K::TG, KMU_FFN_GMAX);
        R = rows_for((ntiles + K::NW - 1) / K::NW, KMU_FFN_GMAX / K::JS);
    }
};
inline int red2_rows(int B, int P) { return rows_for((B * (P / 64) + 7) / 8, 256); }
inline int apply_blocks(long total4) {
    long g = (total4 + 256 * 8 - 1) / (256 * 8);
    return (int)(g < 1 ? 1 : (g > 512 ? 512 : g));
}

struct FwdPtrs {
    const float *x, *w1, *w2, *alpha;
    BnArgs bn1, bn2;
    float *z2, *out, *stats1, *stats2, *h_tap;
};

template <int C>
int ffn_fwd_t(const FwdPtrs& p, int training, float* ws, int B, int P, int stage, hipStream_t st) {
    using K = Cfg<C>;
    const Plan<C> pl(B, P);
    float* part1 = ws;
    float* part2 = ws + (size_t)pl.R * 2 * K::HID;
    const double Ntot = (double)B * (double)P;
    int rc = 0;
    if ((stage < 0 || stage == 0) && training) {
        hipLaunchKernelGGL((ffn_stats1_kernel<C>), dim3(K::JS * pl.R), dim3(K::NTHR), 0, st, p.x, p.w1, part1, P, pl.ntiles);
        if ((rc = kmu::launch_status("ffn_fused_fwd stats"))) return rc;
    }
    if (stage < 0 || stage == 1) {
        const size_t lds = F2Lds<C>::BYTES;
        KMU_MAX_LDS((ffn_fwd_main_kernel<C>), lds);
        hipLaunchKernelGGL((ffn_fwd_main_kernel<C>), dim3(pl.G), dim3(K::NTHR), lds, st, p.x, p.w1, p.w2, p.bn1, part1, pl.R, training,
                           p.stats1, p.z2, training ? part2 : (float*)nullptr, p.h_tap, P, pl.ntiles, Ntot);
        if ((rc = kmu::launch_status("ffn_fused_fwd main"))) return rc;
    }
    if (stage < 0 || stage == 2) {
        const long total4 = (long)B * C * P / 4;
        hipLaunchKernelGGL((ffn_apply_kernel<C>), dim3(apply_blocks(total4)), dim3(256), 0, st, p.z2, p.x, p.bn2, p.alpha, part2, pl.G,
                           training, p.stats2, p.out, P / 4, total4, Ntot);
        rc = kmu::launch_status("ffn_fused_fwd apply");
    }
    return rc;
}

template <int C>
int ffn_bwd_t(BArgs a, const float* beta2, float* slab_w1, float* slab_w2, float* ws, int B, int P, int stage, hipStream_t st) {
    using K = Cfg<C>;
    const Plan<C> pl(B, P);
    const int G1 = red2_rows(B, P);
    float* q1 = ws;
    float* q2 = q1 + (size_t)G1 * 3 * C;
    float* cst2g = q2 + (size_t)pl.R * 2 * K::HID;
    a.P = P, a.ntiles = pl.ntiles, a.Ntot = (double)B * (double)P, a.cst2g = cst2g, a.q2 = q2;
    int rc = 0;
    if (stage < 0 || stage == 0) {
        hipLaunchKernelGGL((ffn_bwd_red2_kernel<C>), dim3(G1), dim3(512), 0, st, a.g, a.z2, a.x, a.gamma2, beta2, a.stats2, a.alpha, q1, P,
                           B * (P / 64));
        if ((rc = kmu::launch_status("ffn_fused_bwd reduce"))) return rc;
    }
    if (stage < 0 || stage == 1) {
        a.part_in = q1, a.Gin = G1, a.slab = slab_w2;
        const size_t lds = BLds<C, 2>::BYTES;
        KMU_MAX_LDS((ffn_bwd_kernel<C, 2>), lds);
        hipLaunchKernelGGL((ffn_bwd_kernel<C, 2>), dim3(K::JS * pl.R), dim3(K::NTHR), lds, st, a);
        if ((rc = kmu::launch_status("ffn_fused_bwd mid"))) return rc;
    }
    if (stage < 0 || stage == 2) {
        a.part_in = q2, a.Gin = pl.R, a.slab = slab_w1;
        const size_t lds = BLds<C, 3>::BYTES;
        KMU_MAX_LDS((ffn_bwd_kernel<C, 3>), lds);
        hipLaunchKernelGGL((ffn_bwd_kernel<C, 3>), dim3(pl.G), dim3(K::NTHR), lds, st, a);
        rc = kmu::launch_status("ffn_fused_bwd input");
    }
    return rc;
}

inline bool ffn_ok(int C, int hid, int P) { return (C == 16 || C == 32 || C == 64) && hid == 4 * C && P > 0 && P % 64 == 0; }

template <int C>
size_t fwd_ws_floats(int B, int P) {
    const Plan<C> pl(B, P);
    return (size_t)pl.R * 2 * Cfg<C>::HID + (size_t)pl.G * 2 * C;
}
template <int C>
size_t bwd_ws_floats(int B, int P) {
    const Plan<C> pl(B, P);
    return (size_t)red2_rows(B, P) * 3 * C + (size_t)pl.R * 2 * Cfg<C>::HID + 4 * C;
}


template <int C>
int tail_fwd_t(const float* nrm, const float* x, const float* w0, const float* b0, const float* w2, const float* b2, const float* sdp,
               float* out, int B, int P, hipStream_t st) {
    using K = Cfg<C>;
    const Plan<C> pl(B, P);
    const size_t lds = TailFwdLds<C>::BYTES;
    KMU_MAX_LDS((tail_ffn_fwd_kernel<C>), lds);
    hipLaunchKernelGGL((tail_ffn_fwd_kernel<C>), dim3(pl.G), dim3(K::NTHR), lds, st, nrm, x, w0, b0, w2, b2, sdp, out, P, pl.ntiles);
    return kmu::launch_status("tail_ffn_fwd");
}
template <int C>
int tail_bwd_t(TailBArgs a, int B, int P, hipStream_t st) {
    using K = Cfg<C>;
    const Plan<C> pl(B, P);
    a.P = P, a.ntiles = pl.ntiles;
    const size_t lds = TailBwdLds<C>::BYTES;
    KMU_MAX_LDS((tail_ffn_bwd_kernel<C>), lds);
    hipLaunchKernelGGL((tail_ffn_bwd_kernel<C>), dim3(pl.G), dim3(K::NTHR), lds, st, a);
    return kmu::launch_status("tail_ffn_bwd");
}

}  // namespace

// ---- EnhancedViMBlock's FFN tail (KM_UNetV3_SH.py:120-124, :147-150): out = x + s[b] (W2 gelu(W0 nrm + b0) + b2), nrm = TripleNorm(x)
// computed by the caller (kmu_triple_norm_fwd).  Same shape support as kmu_ffn_fused_*; rows = kmu_ffn_fused_rows(B, C, P, 1).
// bwd: dn = d loss / d nrm; slab_w0 [rows][4C][C], slab_w2 [rows][C][4C], rows_b [rows][4C + C] (db0 | db2): column sums = gradients.
extern "C" int kmu_tail_ffn_fwd(const float* nrm, const float* x, const float* w0, const float* b0, const float* w2, const float* b2,
                                const float* sdp, float* out, int B, int C, int P, kmu_stream_t stream) {
    KMU_REQUIRE(nrm && x && w0 && b0 && w2 && b2 && out, "tail_ffn_fwd: null pointer");
    KMU_REQUIRE(B > 0 && ffn_ok(C, 4 * C, P), "tail_ffn_fwd: C = %d (16 / 32 / 64) with 4C hidden channels and H*W = %d (multiple of 64) only", C, P);
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 16: return tail_fwd_t<16>(nrm, x, w0, b0, w2, b2, sdp, out, B, P, st);
        case 32: return tail_fwd_t<32>(nrm, x, w0, b0, w2, b2, sdp, out, B, P, st);
        default: return tail_fwd_t<64>(nrm, x, w0, b0, w2, b2, sdp, out, B, P, st);
    }
}

extern "C" int kmu_tail_ffn_bwd(const float* nrm, const float* g, const float* w0, const float* b0, const float* w2, const float* sdp,
                                float* dn, float* slab_w0, float* slab_w2, float* rows_b, int B, int C, int P, kmu_stream_t stream) {
    KMU_REQUIRE(nrm && g && w0 && b0 && w2 && dn && slab_w0 && slab_w2 && rows_b, "tail_ffn_bwd: null pointer");
    KMU_REQUIRE(B > 0 && ffn_ok(C, 4 * C, P), "tail_ffn_bwd: C = %d (16 / 32 / 64) with 4C hidden channels and H*W = %d (multiple of 64) only", C, P);
    TailBArgs a = {};
    a.nrm = nrm, a.g = g, a.w0 = w0, a.b0 = b0, a.w2 = w2, a.sdp = sdp, a.dn = dn, a.slab_w0 = slab_w0, a.slab_w2 = slab_w2, a.rows_b = rows_b;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 16: return tail_bwd_t<16>(a, B, P, st);
        case 32: return tail_bwd_t<32>(a, B, P, st);
        default: return tail_bwd_t<64>(a, B, P, st);
    }
}

namespace {
}  // namespace

extern "C" int kmu_ffn_fused_supported(int C, int hid, int P) { return ffn_ok(C, hid, P) ? 1 : 0; }

// rows of the weight-gradient slabs: which = 1 -> slab_w1 [rows][4C][C], which = 2 -> slab_w2 [rows][C][4C]
extern "C" int kmu_ffn_fused_rows(int B, int C, int P, int which) {
    if (B <= 0 || !ffn_ok(C, 4 * C, P) || (which != 1 && which != 2)) return 0;
    switch (C) {
        case 16: return which == 1 ? Plan<16>(B, P).G : Plan<16>(B, P).R;
        case 32: return which == 1 ? Plan<32>(B, P).G : Plan<32>(B, P).R;
        default: return which == 1 ? Plan<64>(B, P).G : Plan<64>(B, P).R;
    }
}

extern "C" size_t kmu_ffn_fused_fwd_ws_bytes(int B, int C, int P) {
    if (B <= 0 || !ffn_ok(C, 4 * C, P)) return 0;
    return sizeof(float) * (C == 16 ? fwd_ws_floats<16>(B, P) : (C == 32 ? fwd_ws_floats<32>(B, P) : fwd_ws_floats<64>(B, P)));
}

extern "C" size_t kmu_ffn_fused_bwd_ws_bytes(int B, int C, int P) {
    if (B <= 0 || !ffn_ok(C, 4 * C, P)) return 0;
    return sizeof(float) * (C == 16 ? bwd_ws_floats<16>(B, P) : (C == 32 ? bwd_ws_floats<32>(B, P) : bwd_ws_floats<64>(B, P)));
}

extern "C" int kmu_ffn_fused_fwd(const float* x, const float* w1, const float* gamma1, const float* beta1, float* running_mean1,
                                 float* running_var1, long long* nbt1, float momentum1, float eps1, const float* w2, const float* gamma2,
                                 const float* beta2, float* running_mean2, float* running_var2, long long* nbt2, float momentum2, float eps2,
                                 const float* alpha, int training, float* z2, float* out, float* stats1, float* stats2, float* h_tap,
                                 void* ws, size_t ws_bytes, int B, int C, int P, int stage, kmu_stream_t stream) {
    KMU_REQUIRE(x && w1 && gamma1 && beta1 && running_mean1 && running_var1 && w2 && gamma2 && beta2 && running_mean2 && running_var2 &&
                    alpha && z2 && out && stats1 && stats2 && ws,
                "ffn_fused_fwd: null pointer");
    KMU_REQUIRE(B > 0 && ffn_ok(C, 4 * C, P), "ffn_fused_fwd: C = %d (16 / 32 / 64) with 4C hidden channels and H*W = %d (multiple of 64) only", C, P);
    KMU_REQUIRE(ws_bytes >= kmu_ffn_fused_fwd_ws_bytes(B, C, P), "ffn_fused_fwd: workspace too small");
    KMU_REQUIRE(stage >= -1 && stage <= 2, "ffn_fused_fwd: stage %d", stage);
    FwdPtrs p;
    p.x = x, p.w1 = w1, p.w2 = w2, p.alpha = alpha, p.z2 = z2, p.out = out, p.stats1 = stats1, p.stats2 = stats2, p.h_tap = h_tap;
    p.bn1 = BnArgs{gamma1, beta1, running_mean1, running_var1, nbt1, momentum1, eps1};
    p.bn2 = BnArgs{gamma2, beta2, running_mean2, running_var2, nbt2, momentum2, eps2};
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 16: return ffn_fwd_t<16>(p, training, (float*)ws, B, P, stage, st);
        case 32: return ffn_fwd_t<32>(p, training, (float*)ws, B, P, stage, st);
        default: return ffn_fwd_t<64>(p, training, (float*)ws, B, P, stage, st);
    }
}

extern "C" int kmu_ffn_fused_bwd(const float* g, const float* x, const float* z2, const float* w1, const float* gamma1, const float* beta1,
                                 const float* stats1, const float* w2, const float* gamma2, const float* beta2, const float* stats2,
                                 const float* alpha, int training, float* dx, float* d_gamma1, float* d_beta1, float* d_gamma2,
                                 float* d_beta2, float* d_alpha, float* slab_w1, float* slab_w2, void* ws, size_t ws_bytes, int B, int C,
                                 int P, int stage, kmu_stream_t stream) {
    KMU_REQUIRE(g && x && z2 && w1 && gamma1 && beta1 && stats1 && w2 && gamma2 && beta2 && stats2 && alpha && dx && d_gamma1 && d_beta1 &&
                    d_gamma2 && d_beta2 && d_alpha && slab_w1 && slab_w2 && ws,
                "ffn_fused_bwd: null pointer");
    KMU_REQUIRE(B > 0 && ffn_ok(C, 4 * C, P), "ffn_fused_bwd: C = %d (16 / 32 / 64) with 4C hidden channels and H*W = %d (multiple of 64) only", C, P);
    KMU_REQUIRE(ws_bytes >= kmu_ffn_fused_bwd_ws_bytes(B, C, P), "ffn_fused_bwd: workspace too small");
    KMU_REQUIRE(stage >= -1 && stage <= 2, "ffn_fused_bwd: stage %d", stage);
    BArgs a = {};
    a.g = g, a.x = x, a.z2 = z2, a.w1 = w1, a.w2 = w2, a.gamma1 = gamma1, a.beta1 = beta1, a.stats1 = stats1, a.gamma2 = gamma2;
    a.stats2 = stats2, a.alpha = alpha, a.training = training, a.d_gamma2 = d_gamma2, a.d_beta2 = d_beta2, a.d_alpha = d_alpha;
    a.d_gamma1 = d_gamma1, a.d_beta1 = d_beta1, a.dx = dx;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 16: return ffn_bwd_t<16>(a, beta2, slab_w1, slab_w2, (float*)ws, B, P, stage, st);
        case 32: return ffn_bwd_t<32>(a, beta2, slab_w1, slab_w2, (float*)ws, B, P, stage, st);
        default: return ffn_bwd_t<64>(a, beta2, slab_w1, slab_w2, (float*)ws, B, P, stage, st);
    }
}
