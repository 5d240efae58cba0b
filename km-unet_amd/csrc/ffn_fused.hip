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
// A kernel that needs channel statistics finds the PARTIAL rows of the kernel before it (one row per workgroup, <= 256 rows) and every
// workgroup reduces them itself in a fixed order (double) -- ~1-2 us of L2 reads at its start instead of a finalising launch (or an
// in-kernel last-arriver with its agent-scope fences) on the dependent chain; workgroup 0 also writes the results out (saved
// statistics, running statistics, parameter gradients).  z1 is accumulated in the same k order in F2, B2 and B3, so the ReLU branch
// taken by the backward is bit-for-bit the forward's.
//
// Orientations (one wave = 64 pixels x a slice of the hidden channels; lane = (n = l % 16, q = l / 16)):
//   D'  (F1, B2, B3)  M = pixels, N = hidden j, K = channels: A = x loaded as float4 [c = 16c'+4q+s][p0 + 4f(n) .. +3], f(n) = n/4 + 4(n%4)
//       (component t = tile t), B = W1[j][c].  D[t] register i <-> pixel p0 + 16i + 4q + t, column j on the lane: per-channel constants are
//       ONE register per 16 channels, and dz (hidden) is directly the A operand of the weight-gradient product (rows j, k = pixel),
//       whose B operand is x / dz2 loaded as float4 [c = lane%16][p0 + 16i + 4q .. +3].
//       dx = dz1 W1 contracts over j (the lane index): dz1 takes one trip through LDS (T[j][pixel], 64-pixel rows + 16 B pad).
//   N   (F2)          M = hidden j, N = pixels, K = channels: the result D1 (rows j in registers, pixel on the lane) is directly the
//       B operand of z2 = W2 h (K = j) -- no LDS between the two products.
// C = 16: 4 waves x their own 64-pixel tile, all 64 hidden channels per wave, no cross-wave traffic in the tile loop.
// C = 32 / 64: ONE tile per workgroup iteration, the hidden channels split over 4 / 8 waves (32 each): 8 x fewer pixels per level
// would otherwise leave most SIMDs idle; z2 partial sums (F2) and the dz1 tile (B3) meet in LDS.
#include "common.h"

using kmu::floatx4;

namespace {

typedef float floatx2 __attribute__((ext_vector_type(2)));

template <int C>
struct Cfg {
    static constexpr int HID = 4 * C;
    static constexpr int JS = C == 16 ? 1 : C / 8;     // hidden-channel slices per tile (= waves sharing a tile)
    static constexpr int NW = C == 16 ? 4 : JS;        // waves per workgroup
    static constexpr int NTW = HID / JS / 16;          // 16-channel hidden tiles per wave: 4, 2, 2
    static constexpr int CC = C / 16;                  // 16-channel chunks of C
    static constexpr int TPI = C == 16 ? 4 : 1;        // tiles per workgroup iteration
    static constexpr int NTHR = NW * 64;
    static constexpr int GMAX = C == 16 ? 256 : 128;   // workgroups (= partial rows): rows x 2 HID floats stays <= 256 KB per reader
};
constexpr int TS = 68;   // floats per 64-pixel LDS row: 16-byte aligned, +4 banks per row

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
__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + __expf(-v)); }

// Column sums of part[G][W] (fp32 rows of the previous kernel) in double and in a fixed order -> tot[W].  scratch: R * W doubles.
template <int W, int NTHR>
struct RowReduce {
    static constexpr int NCG = W / 4;
    static constexpr int R = NTHR / NCG >= 1 ? NTHR / NCG : 1;
    static constexpr int SCRATCH = R * W;   // doubles
    static __device__ __forceinline__ void run(const float* __restrict__ part, int G, double* scratch, double* tot) {
        static_assert(NCG <= NTHR, "row wider than the workgroup");
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

// z1'[pixel][j] = sum_c x[c][pixel] W1[j][c] for one 64-pixel tile and this wave's hidden slice (orientation D').
// xt = x + (b*C)*P + p0.  Accumulation order: chunks c' outer, s inner, the MFMA's own k = q innermost -- the SAME order as F2's.
template <int C>
__device__ __forceinline__ void fc1_dp(const float* __restrict__ xt, int P, const float (&w1r)[Cfg<C>::NTW][4 * Cfg<C>::CC],
                                       floatx4 (&z1)[Cfg<C>::NTW][4], int n, int q) {
    const int poff = 4 * ((n >> 2) + 4 * (n & 3));
#pragma unroll
    for (int nt = 0; nt < Cfg<C>::NTW; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) z1[nt][t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < Cfg<C>::CC; ++cc) {
        floatx4 xa[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xa[s] = *reinterpret_cast<const floatx4*>(xt + (size_t)(16 * cc + 4 * q + s) * P + poff);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < Cfg<C>::NTW; ++nt)
#pragma unroll
                for (int t = 0; t < 4; ++t) mfma(z1[nt][t], xa[s][t], w1r[nt][4 * cc + s]);
    }
    drain(z1);
}

// dh'[pixel][j] = sum_c dz2[c][pixel] W2[c][j] with dz2 = A_c g + (B_c z2 + D_c) formed on the fly (cst2[c] = {A, B, D, 1 - a} in LDS)
template <int C>
__device__ __forceinline__ void dh_dp(const float* __restrict__ gt, const float* __restrict__ zt, int P, const floatx4* cst2,
                                      const float (&w2r)[Cfg<C>::NTW][4 * Cfg<C>::CC], floatx4 (&dh)[Cfg<C>::NTW][4], int n, int q) {
    const int poff = 4 * ((n >> 2) + 4 * (n & 3));
#pragma unroll
    for (int nt = 0; nt < Cfg<C>::NTW; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) dh[nt][t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < Cfg<C>::CC; ++cc) {
        floatx4 dz[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const size_t o = (size_t)(16 * cc + 4 * q + s) * P + poff;
            const floatx4 gv = *reinterpret_cast<const floatx4*>(gt + o), zv = *reinterpret_cast<const floatx4*>(zt + o);
            const floatx4 k = cst2[16 * cc + 4 * q + s];
#pragma unroll
            for (int t = 0; t < 4; ++t) dz[s][t] = fmaf(k[0], gv[t], fmaf(k[1], zv[t], k[2]));
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < Cfg<C>::NTW; ++nt)
#pragma unroll
                for (int t = 0; t < 4; ++t) mfma(dh[nt][t], dz[s][t], w2r[nt][4 * cc + s]);
    }
    drain(dh);
}

template <int C>
__device__ __forceinline__ void tile_of(int tile, int P, int& b, int& p0) {
    const int tpp = P >> 6;
    b = tile / tpp;
    p0 = (tile - b * tpp) << 6;
}

// ------------------------------------------------------------------------------------------------------------ F1
template <int C>
__global__ __launch_bounds__(Cfg<C>::NTHR) void ffn_stats1_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                                  float* __restrict__ part1, int P, int ntiles) {
    using K = Cfg<C>;
    __shared__ float comb[K::JS == 1 ? 4 * 2 * K::HID : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const int j0 = K::JS == 1 ? 0 : wave * 16 * K::NTW;
    float w1r[K::NTW][4 * K::CC];
    load_w1r<C>(w1, j0, n, q, w1r);
    float sa[K::NTW], sq[K::NTW];
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) sa[nt] = sq[nt] = 0.f;
    for (int tile = blockIdx.x * K::TPI + (K::JS == 1 ? wave : 0); tile < ntiles; tile += gridDim.x * K::TPI) {
        int b, p0;
        tile_of<C>(tile, P, b, p0);
        floatx4 z1[K::NTW][4];
        fc1_dp<C>(x + (size_t)b * C * P + p0, P, w1r, z1, n, q);
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = z1[nt][t][i];
                    sa[nt] += v;
                    sq[nt] = fmaf(v, v, sq[nt]);
                }
    }
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) {
        sa[nt] += __shfl_xor(sa[nt], 16);
        sa[nt] += __shfl_xor(sa[nt], 32);
        sq[nt] += __shfl_xor(sq[nt], 16);
        sq[nt] += __shfl_xor(sq[nt], 32);
    }
    float* row = part1 + (size_t)blockIdx.x * 2 * K::HID;
    if constexpr (K::JS == 1) {
        if (q == 0) {
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt) {
                comb[wave * 2 * K::HID + 16 * nt + n] = sa[nt];
                comb[wave * 2 * K::HID + K::HID + 16 * nt + n] = sq[nt];
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * K::HID; t += K::NTHR)
            row[t] = ((comb[t] + comb[2 * K::HID + t]) + comb[4 * K::HID + t]) + comb[6 * K::HID + t];
    } else if (q == 0) {
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt) {
            row[j0 + 16 * nt + n] = sa[nt];
            row[K::HID + j0 + 16 * nt + n] = sq[nt];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ F2
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
    static constexpr int MAIN_F = K::JS == 1 ? 4 * 2 * C : K::JS * C * TS;                          // floats (tile loop / epilogue)
    static constexpr size_t UNION_B = (size_t)RED_D * 8 > (size_t)MAIN_F * 4 ? (size_t)RED_D * 8 : (size_t)MAIN_F * 4;
    static constexpr size_t BYTES = 2 * K::HID * 4 + UNION_B;
};

template <int C>
__global__ __launch_bounds__(Cfg<C>::NTHR) void ffn_fwd_main_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                                    const float* __restrict__ w2, BnArgs bn1,
                                                                    const float* __restrict__ part1, int G1, int training,
                                                                    float* __restrict__ stats1, float* __restrict__ z2,
                                                                    float* __restrict__ part2, float* __restrict__ h_tap, int P,
                                                                    int ntiles, double Ntot) {
    using K = Cfg<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc1 = reinterpret_cast<float*>(smem);
    float* sh1 = sc1 + K::HID;
    char* un = smem + 2 * K::HID * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int j0 = K::JS == 1 ? 0 : wave * 16 * K::NTW;

    // this wave's weights first: their round trip overlaps the statistics prologue
    float w1r[K::NTW][4 * K::CC];
    load_w1r<C>(w1, j0, m, q, w1r);
    floatx4 w2r[K::CC][K::NTW];          // W2[16ct + m][j0 + 16nt + 4q + i]: A[row = c][k = q] of k-step (nt, i)
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt)
            w2r[ct][nt] = *reinterpret_cast<const floatx4*>(w2 + (size_t)(16 * ct + m) * K::HID + j0 + 16 * nt + 4 * q);

    if (training) {
        double* scratch = reinterpret_cast<double*>(un);
        double* tot = scratch + RowReduce<2 * K::HID, K::NTHR>::SCRATCH;
        RowReduce<2 * K::HID, K::NTHR>::run(part1, G1, scratch, tot);
        for (int j = tid; j < K::HID; j += K::NTHR) {
            float mean, rstd;
            double var;
            mean_rstd(tot[j], tot[K::HID + j], Ntot, bn1.eps, mean, rstd, var);
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
    // statistics of z2: JS == 1: lane (m, q) owns channels 16ct + 4q + i of its pixels; JS > 1: thread e / e + NTHR own channel e>>4
    float s2a[K::JS == 1 ? 4 * K::CC : 2], s2q[K::JS == 1 ? 4 * K::CC : 2];
#pragma unroll
    for (int k = 0; k < (K::JS == 1 ? 4 * K::CC : 2); ++k) s2a[k] = s2q[k] = 0.f;

    for (int tile = blockIdx.x * K::TPI + (K::JS == 1 ? wave : 0); tile < ntiles; tile += gridDim.x * K::TPI) {
        int b, p0;
        tile_of<C>(tile, P, b, p0);
        const float* xt = x + (size_t)b * C * P + p0;
        floatx4 d1[K::NTW][4];
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
            for (int t = 0; t < 4; ++t) d1[nt][t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < K::CC; ++cc) {
            floatx4 xa[4];           // B[k = q][n = m]: channel 16cc + 4q + s, pixel p0 + 4m + t
#pragma unroll
            for (int s = 0; s < 4; ++s) xa[s] = *reinterpret_cast<const floatx4*>(xt + (size_t)(16 * cc + 4 * q + s) * P + 4 * m);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                    for (int t = 0; t < 4; ++t) mfma(d1[nt][t], w1r[nt][4 * cc + s], xa[s][t]);
        }
        drain(d1);
        // BatchNorm + ReLU on the registers; d1[nt][t][i]: hidden j0 + 16nt + 4q + i, pixel p0 + 4m + t
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) d1[nt][t][i] = fmaxf(fmaf(d1[nt][t][i], scv[nt][i], shv[nt][i]), 0.f);
        if (h_tap) {                 // tests only: the hidden activation (its sign pattern is the ReLU branch the kernels take)
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<floatx4*>(h_tap + ((size_t)b * K::HID + j0 + 16 * nt + 4 * q + i) * P + p0 + 4 * m) =
                        floatx4{d1[nt][0][i], d1[nt][1][i], d1[nt][2][i], d1[nt][3][i]};
        }
        floatx4 d2[K::CC][4];
#pragma unroll
        for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
            for (int t = 0; t < 4; ++t) d2[ct][t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                    for (int t = 0; t < 4; ++t) mfma(d2[ct][t], w2r[ct][nt][i], d1[nt][t][i]);
        drain(d2);
        // d2[ct][t][i]: channel 16ct + 4q + i, pixel p0 + 4m + t
        if constexpr (K::JS == 1) {
#pragma unroll
            for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const floatx4 v = {d2[ct][0][i], d2[ct][1][i], d2[ct][2][i], d2[ct][3][i]};
                    *reinterpret_cast<floatx4*>(z2 + ((size_t)b * C + 16 * ct + 4 * q + i) * P + p0 + 4 * m) = v;
                    s2a[4 * ct + i] += (v[0] + v[1]) + (v[2] + v[3]);
                    s2q[4 * ct + i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
        } else {
#pragma unroll
            for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<floatx4*>(redz + (size_t)(wave * C + 16 * ct + 4 * q + i) * TS + 4 * m) =
                        floatx4{d2[ct][0][i], d2[ct][1][i], d2[ct][2][i], d2[ct][3][i]};
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 2; ++k) {                  // C * 16 float4 outputs over NTHR threads = 2 each
                const int e = tid + k * K::NTHR, c = e >> 4, pg = e & 15;
                floatx4 v = *reinterpret_cast<const floatx4*>(redz + (size_t)c * TS + 4 * pg);
#pragma unroll
                for (int w = 1; w < K::JS; ++w) v += *reinterpret_cast<const floatx4*>(redz + (size_t)(w * C + c) * TS + 4 * pg);
                *reinterpret_cast<floatx4*>(z2 + ((size_t)b * C + c) * P + p0 + 4 * pg) = v;
                s2a[k] += (v[0] + v[1]) + (v[2] + v[3]);
                s2q[k] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
            __syncthreads();
        }
    }
    if (!part2) return;
    float* row = part2 + (size_t)blockIdx.x * 2 * C;
    if constexpr (K::JS == 1) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4 * K::CC; ++k) {
            const float a = kmu::wave_reduce<kmu::OpSum, 16>(s2a[k]), qq = kmu::wave_reduce<kmu::OpSum, 16>(s2q[k]);
            if (m == 0) {
                const int c = 16 * (k >> 2) + 4 * q + (k & 3);
                redz[wave * 2 * C + c] = a;
                redz[wave * 2 * C + C + c] = qq;
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * C; t += K::NTHR) row[t] = ((redz[t] + redz[2 * C + t]) + redz[4 * C + t]) + redz[6 * C + t];
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float a = kmu::wave_reduce<kmu::OpSum, 16>(s2a[k]), qq = kmu::wave_reduce<kmu::OpSum, 16>(s2q[k]);
            const int e = tid + k * K::NTHR;
            if ((e & 15) == 0) row[e >> 4] = a, row[C + (e >> 4)] = qq;
        }
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
            const float sc = bn2.gamma[tid] * rstd;
            sc2[tid] = sc, sh2[tid] = bn2.beta[tid] - mean * sc;
            av[tid] = sigmoid_f(alpha[tid]);
        }
        if (blockIdx.x == 0 && tid == 0 && bn2.nbt) *bn2.nbt += 1;
    } else if (tid < C) {
        const float mean = bn2.running_mean[tid], rstd = 1.f / sqrtf(bn2.running_var[tid] + bn2.eps);
        if (blockIdx.x == 0) stats2[2 * tid] = mean, stats2[2 * tid + 1] = rstd;
        const float sc = bn2.gamma[tid] * rstd;
        sc2[tid] = sc, sh2[tid] = bn2.beta[tid] - mean * sc;
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
// per channel over all pixels: s0 = sum a g, s1 = sum a g z2hat, s2 = sum g (BN2(z2) - x)      -> q1 [G][3C]
template <int C>
__global__ __launch_bounds__(256) void ffn_bwd_red2_kernel(const float* __restrict__ g, const float* __restrict__ z2,
                                                           const float* __restrict__ x, const float* __restrict__ gamma2,
                                                           const float* __restrict__ beta2, const float* __restrict__ stats2,
                                                           const float* __restrict__ alpha, float* __restrict__ q1, int P, int ntiles) {
    constexpr int CC = C / 16;
    __shared__ float comb[4 * 3 * C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    float mean[CC], rstd[CC], sc[CC], sh[CC], a[CC], s0[CC], s1[CC], s2[CC];
#pragma unroll
    for (int ct = 0; ct < CC; ++ct) {
        const int c = 16 * ct + r;
        mean[ct] = stats2[2 * c], rstd[ct] = stats2[2 * c + 1];
        sc[ct] = gamma2[c] * rstd[ct], sh[ct] = beta2[c] - mean[ct] * sc[ct];
        a[ct] = sigmoid_f(alpha[c]);
        s0[ct] = s1[ct] = s2[ct] = 0.f;
    }
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        int b, p0;
        tile_of<C>(tile, P, b, p0);
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
    for (int t = threadIdx.x; t < 3 * C; t += 256) row[t] = ((comb[t] + comb[3 * C + t]) + comb[6 * C + t]) + comb[9 * C + t];
}

// ------------------------------------------------------------------------------------------------------------ B2 / B3
// PASS 2: partial sums of BN1's backward + dW2 slabs.  PASS 3: dx + dW1 slabs.
template <int C, int PASS>
struct BLds {
    using K = Cfg<C>;
    static constexpr int W_IN = PASS == 2 ? 3 * C : 2 * K::HID;                                   // width of the partial rows read
    static constexpr int RED_D = RowReduce<W_IN, K::NTHR>::SCRATCH + W_IN;                        // doubles (prologue)
    static constexpr int T_ROWS = K::JS == 1 ? 4 * K::HID : K::HID;                               // dz1 tile(s), PASS 3
    static constexpr int COMB_F = K::JS == 1 ? 4 * C * K::HID : 4;                                // cross-wave slab combine (JS == 1)
    static constexpr int MAIN_F3 = T_ROWS * TS > COMB_F ? T_ROWS * TS : COMB_F;
    static constexpr int MAIN_F2 = COMB_F > 4 * 2 * K::HID ? COMB_F : 4 * 2 * K::HID;
    static constexpr int MAIN_F = PASS == 3 ? MAIN_F3 : MAIN_F2;
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
    float* q2;                // PASS 2 out [G][2 HID]
    float* slab;              // PASS 2: dW2 slabs [G][C][HID]; PASS 3: dW1 slabs [G][HID][C]
    float* dx;                // PASS 3
    int P, ntiles;
    double Ntot;
};

template <int C, int PASS>
__global__ __launch_bounds__(Cfg<C>::NTHR) void ffn_bwd_kernel(BArgs a) {
    using K = Cfg<C>;
    using L = BLds<C, PASS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    floatx4* cst2 = reinterpret_cast<floatx4*>(smem);                 // [C]
    float* hc = reinterpret_cast<float*>(smem) + 4 * C;               // [5][HID]: sc1, sh1, then (mean1, rstd1, -) or (k1, E, F)
    float* wl = hc + 5 * K::HID;                                      // PASS 3: W1 [HID][C + 4]
    char* un = smem + (size_t)(L::CONST_F + L::WL_F) * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    const int j0 = K::JS == 1 ? 0 : wave * 16 * K::NTW;
    const int P = a.P;

    float w1r[K::NTW][4 * K::CC], w2r[K::NTW][4 * K::CC];
    load_w1r<C>(a.w1, j0, n, q, w1r);
    load_w2r<C>(a.w2, j0, n, q, w2r);
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
                const float n0 = a.training ? (float)(tot[j] / a.Ntot) : 0.f, n1 = a.training ? (float)(tot[K::HID + j] / a.Ntot) : 0.f;
                // dz1 = k1 (dr - n0 - z1hat n1),  z1hat = (z1 - mean) rstd
                hc[j] = k1, hc[K::HID + j] = sh;
                hc[2 * K::HID + j] = k1, hc[3 * K::HID + j] = -k1 * n1 * rstd, hc[4 * K::HID + j] = k1 * (n1 * rstd * mean - n0);
                if (blockIdx.x == 0) a.d_gamma1[j] = (float)tot[K::HID + j], a.d_beta1[j] = (float)tot[j];
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
    floatx4 cb[K::CC];                   // constants of channel 16ct + n (operand layout of the weight-gradient / output side)
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct) cb[ct] = cst2[16 * ct + n];
    __syncthreads();                     // the union region turns from reduction scratch into tile scratch

    float* T = reinterpret_cast<float*>(un);
    floatx4 acc[K::CC][K::NTW];          // PASS 2: dW2 tile (rows c = 16ct + 4q + i, column j); PASS 3: dW1 tile (rows j, column c = 16ct + n)
#pragma unroll
    for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt) acc[ct][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float t0[K::NTW], t1[K::NTW];
#pragma unroll
    for (int nt = 0; nt < K::NTW; ++nt) t0[nt] = t1[nt] = 0.f;
    const int poff = 4 * ((n >> 2) + 4 * (n & 3));
    // PASS 3, JS > 1: this wave's share of the dx tile: channel tile ctw, pixel-tile pair (2th, 2th + 1)
    const int ctw = K::JS == 1 ? 0 : wave % K::CC, th = K::JS == 1 ? 0 : wave / K::CC;

    for (int tile = blockIdx.x * K::TPI + (K::JS == 1 ? wave : 0); tile < a.ntiles; tile += gridDim.x * K::TPI) {
        int b, p0;
        tile_of<C>(tile, P, b, p0);
        const size_t base = (size_t)b * C * P + p0;
        floatx4 z1[K::NTW][4], dh[K::NTW][4];
        fc1_dp<C>(a.x + base, P, w1r, z1, n, q);
        dh_dp<C>(a.g + base, a.z2 + base, P, cst2, w2r, dh, n, q);
        // z1 / dh [nt][t][i]: hidden j0 + 16nt + n, pixel p0 + 16i + 4q + t
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z = z1[nt][t][i], pre = fmaf(z, sc1v[nt], sh1v[nt]);
                    const bool on = pre > 0.f;
                    const float dr = on ? dh[nt][t][i] : 0.f;
                    if constexpr (PASS == 2) {
                        const float that = (z - c2v[nt]) * c3v[nt];
                        t0[nt] += dr;
                        t1[nt] = fmaf(dr, that, t1[nt]);
                        z1[nt][t][i] = on ? pre : 0.f;                                   // h
                    } else {
                        z1[nt][t][i] = fmaf(c2v[nt], dr, fmaf(c3v[nt], z, c4v[nt]));     // dz1
                    }
                }
        // weight-gradient product over this tile's pixels: k-step (t, i) <-> pixels p0 + 16i + 4q + t, q = the MFMA's k
#pragma unroll
        for (int ct = 0; ct < K::CC; ++ct) {
            const size_t o = base + (size_t)(16 * ct + n) * P + 4 * q;
            floatx4 v[4];
            if constexpr (PASS == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const floatx4 gv = *reinterpret_cast<const floatx4*>(a.g + o + 16 * i), zv = *reinterpret_cast<const floatx4*>(a.z2 + o + 16 * i);
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[i][t] = fmaf(cb[ct][0], gv[t], fmaf(cb[ct][1], zv[t], cb[ct][2]));   // dz2
                }
#pragma unroll
                for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) mfma(acc[ct][nt], v[i][t], z1[nt][t][i]);     // dW2[c][j] += dz2 h
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const floatx4*>(a.x + o + 16 * i);
#pragma unroll
                for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) mfma(acc[ct][nt], z1[nt][t][i], v[i][t]);     // dW1[j][c] += dz1 x
            }
        }
        if constexpr (PASS == 3) {
            // dz1 -> T[j][pixel]: its transpose-through-LDS for dx[pixel][c] = sum_j dz1[pixel][j] W1[j][c]
            float* Tw = T + (K::JS == 1 ? (size_t)wave * K::HID * TS : 0);
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<floatx4*>(Tw + (size_t)(j0 + 16 * nt + n) * TS + 16 * i + 4 * q) =
                        floatx4{z1[nt][0][i], z1[nt][1][i], z1[nt][2][i], z1[nt][3][i]};
            if constexpr (K::JS == 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
                __syncthreads();
            }
            if constexpr (K::JS == 1) {
                floatx4 d[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) d[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int jt = 0; jt < K::HID / 16; ++jt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int j = 16 * jt + 4 * q + s;
                        const floatx4 av = *reinterpret_cast<const floatx4*>(Tw + (size_t)j * TS + poff);
                        const float bv = wl[j * (C + 4) + n];
#pragma unroll
                        for (int t = 0; t < 4; ++t) mfma(d[t], av[t], bv);
                    }
                drain(d);
                // d[t][i]: pixel p0 + 16i + 4q + t, channel n
                const size_t o = base + (size_t)n * P + 4 * q;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const floatx4 gv = *reinterpret_cast<const floatx4*>(a.g + o + 16 * i);
                    floatx4 v;
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = fmaf(cb[0][3], gv[t], d[t][i]);
                    *reinterpret_cast<floatx4*>(a.dx + o + 16 * i) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                floatx4 d[2];
                d[0] = d[1] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int jt = 0; jt < K::HID / 16; ++jt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int j = 16 * jt + 4 * q + s;
                        const floatx2 av = *reinterpret_cast<const floatx2*>(T + (size_t)j * TS + poff + 2 * th);
                        const float bv = wl[j * (C + 4) + 16 * ctw + n];
                        mfma(d[0], av[0], bv);
                        mfma(d[1], av[1], bv);
                    }
                drain(d);
                // d[tt][i]: pixel p0 + 16i + 4q + 2th + tt, channel 16ctw + n
                const size_t o = base + (size_t)(16 * ctw + n) * P + 4 * q + 2 * th;
                const float oma = cst2[16 * ctw + n][3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const floatx2 gv = *reinterpret_cast<const floatx2*>(a.g + o + 16 * i);
                    floatx2 v = {fmaf(oma, gv[0], d[0][i]), fmaf(oma, gv[1], d[1][i])};
                    *reinterpret_cast<floatx2*>(a.dx + o + 16 * i) = v;
                }
                __syncthreads();             // T is rewritten by the next tile
            }
        }
    }

    // ---- epilogue: partial row of BN1's backward sums (PASS 2) and this workgroup's weight-gradient slab
    if constexpr (PASS == 2) {
#pragma unroll
        for (int nt = 0; nt < K::NTW; ++nt) {
            t0[nt] += __shfl_xor(t0[nt], 16), t0[nt] += __shfl_xor(t0[nt], 32);
            t1[nt] += __shfl_xor(t1[nt], 16), t1[nt] += __shfl_xor(t1[nt], 32);
        }
        float* row = a.q2 + (size_t)blockIdx.x * 2 * K::HID;
        if constexpr (K::JS == 1) {
            __syncthreads();
            if (q == 0) {
#pragma unroll
                for (int nt = 0; nt < K::NTW; ++nt) {
                    T[wave * 2 * K::HID + 16 * nt + n] = t0[nt];
                    T[wave * 2 * K::HID + K::HID + 16 * nt + n] = t1[nt];
                }
            }
            __syncthreads();
            for (int t = tid; t < 2 * K::HID; t += K::NTHR) row[t] = ((T[t] + T[2 * K::HID + t]) + T[4 * K::HID + t]) + T[6 * K::HID + t];
        } else if (q == 0) {
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt) row[j0 + 16 * nt + n] = t0[nt], row[K::HID + j0 + 16 * nt + n] = t1[nt];
        }
    }
    drain(acc);
    float* slab = a.slab + (size_t)blockIdx.x * C * K::HID;
    if constexpr (K::JS == 1) {                    // the four waves hold partial sums of the SAME tile: add them in a fixed order
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = PASS == 2 ? (16 * ct + 4 * q + i) * K::HID + 16 * nt + n : (16 * nt + 4 * q + i) * C + 16 * ct + n;
                    T[(size_t)wave * C * K::HID + e] = acc[ct][nt][i];
                }
        __syncthreads();
        for (int e = tid; e < C * K::HID; e += K::NTHR)
            slab[e] = ((T[e] + T[C * K::HID + e]) + T[2 * C * K::HID + e]) + T[3 * C * K::HID + e];
    } else {
#pragma unroll
        for (int ct = 0; ct < K::CC; ++ct)
#pragma unroll
            for (int nt = 0; nt < K::NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = PASS == 2 ? (16 * ct + 4 * q + i) * K::HID + j0 + 16 * nt + n : (j0 + 16 * nt + 4 * q + i) * C + 16 * ct + n;
                    slab[e] = acc[ct][nt][i];
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
int ffn_rows(int B, int P) {
    const int ntiles = B * (P / 64);
    return rows_for((ntiles + Cfg<C>::TPI - 1) / Cfg<C>::TPI, Cfg<C>::GMAX);
}
inline int red2_rows(int B, int P) { return rows_for((B * (P / 64) + 3) / 4, 256); }
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
    const int ntiles = B * (P / 64), G = ffn_rows<C>(B, P);
    float* part1 = ws;
    float* part2 = ws + (size_t)G * 2 * K::HID;
    const double Ntot = (double)B * (double)P;
    int rc = 0;
    if ((stage < 0 || stage == 0) && training) {
        hipLaunchKernelGGL((ffn_stats1_kernel<C>), dim3(G), dim3(K::NTHR), 0, st, p.x, p.w1, part1, P, ntiles);
        if ((rc = kmu::launch_status("ffn_fused_fwd stats"))) return rc;
    }
    if (stage < 0 || stage == 1) {
        const size_t lds = F2Lds<C>::BYTES;
        KMU_MAX_LDS((ffn_fwd_main_kernel<C>), lds);
        hipLaunchKernelGGL((ffn_fwd_main_kernel<C>), dim3(G), dim3(K::NTHR), lds, st, p.x, p.w1, p.w2, p.bn1, part1, G, training, p.stats1,
                           p.z2, training ? part2 : (float*)nullptr, p.h_tap, P, ntiles, Ntot);
        if ((rc = kmu::launch_status("ffn_fused_fwd main"))) return rc;
    }
    if (stage < 0 || stage == 2) {
        const long total4 = (long)B * C * P / 4;
        hipLaunchKernelGGL((ffn_apply_kernel<C>), dim3(apply_blocks(total4)), dim3(256), 0, st, p.z2, p.x, p.bn2, p.alpha, part2, G, training,
                           p.stats2, p.out, P / 4, total4, Ntot);
        rc = kmu::launch_status("ffn_fused_fwd apply");
    }
    return rc;
}

template <int C>
int ffn_bwd_t(BArgs a, const float* beta2, float* slab_w1, float* slab_w2, float* ws, int B, int P, int stage, hipStream_t st) {
    using K = Cfg<C>;
    const int ntiles = B * (P / 64), G = ffn_rows<C>(B, P), G1 = red2_rows(B, P);
    float* q1 = ws;
    float* q2 = q1 + (size_t)G1 * 3 * C;
    float* cst2g = q2 + (size_t)G * 2 * K::HID;
    a.P = P, a.ntiles = ntiles, a.Ntot = (double)B * (double)P, a.cst2g = cst2g, a.q2 = q2;
    int rc = 0;
    if (stage < 0 || stage == 0) {
        hipLaunchKernelGGL((ffn_bwd_red2_kernel<C>), dim3(G1), dim3(256), 0, st, a.g, a.z2, a.x, a.gamma2, beta2, a.stats2, a.alpha, q1, P, ntiles);
        if ((rc = kmu::launch_status("ffn_fused_bwd reduce"))) return rc;
    }
    if (stage < 0 || stage == 1) {
        a.part_in = q1, a.Gin = G1, a.slab = slab_w2;
        const size_t lds = BLds<C, 2>::BYTES;
        KMU_MAX_LDS((ffn_bwd_kernel<C, 2>), lds);
        hipLaunchKernelGGL((ffn_bwd_kernel<C, 2>), dim3(G), dim3(K::NTHR), lds, st, a);
        if ((rc = kmu::launch_status("ffn_fused_bwd mid"))) return rc;
    }
    if (stage < 0 || stage == 2) {
        a.part_in = q2, a.Gin = G, a.slab = slab_w1;
        const size_t lds = BLds<C, 3>::BYTES;
        KMU_MAX_LDS((ffn_bwd_kernel<C, 3>), lds);
        hipLaunchKernelGGL((ffn_bwd_kernel<C, 3>), dim3(G), dim3(K::NTHR), lds, st, a);
        rc = kmu::launch_status("ffn_fused_bwd input");
    }
    return rc;
}

inline bool ffn_ok(int C, int hid, int P) { return (C == 16 || C == 32 || C == 64) && hid == 4 * C && P > 0 && P % 64 == 0; }

}  // namespace

extern "C" int kmu_ffn_fused_supported(int C, int hid, int P) { return ffn_ok(C, hid, P) ? 1 : 0; }

extern "C" int kmu_ffn_fused_rows(int B, int C, int P) {
    if (B <= 0 || !ffn_ok(C, 4 * C, P)) return 0;
    return C == 16 ? ffn_rows<16>(B, P) : (C == 32 ? ffn_rows<32>(B, P) : ffn_rows<64>(B, P));
}

extern "C" size_t kmu_ffn_fused_fwd_ws_bytes(int B, int C, int P) {
    const int G = kmu_ffn_fused_rows(B, C, P);
    return (size_t)G * (2 * 4 * C + 2 * C) * sizeof(float);
}

extern "C" size_t kmu_ffn_fused_bwd_ws_bytes(int B, int C, int P) {
    const int G = kmu_ffn_fused_rows(B, C, P);
    if (!G) return 0;
    return ((size_t)red2_rows(B, P) * 3 * C + (size_t)G * 2 * 4 * C + 4 * C) * sizeof(float);
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
