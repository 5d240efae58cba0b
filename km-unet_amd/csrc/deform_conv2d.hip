// K4: deformable 3x3 conv (stride 1, pad 1, one offset group, no mask) for gfx950.
//
// Replaces DAGEM_md.py:98-101 (torchvision.ops.DeformConv2d, torchvision 0.14.0 -- third-party,
// restated from its published semantics in oracle/deform.py: PARITY UNPINNED).  The op is tiny in
// KM-UNet ([B,64,16,16], 0.5 % of the reference's forward) so these kernels are plain VALU:
// 16 output pixels per workgroup, sampled columns staged in LDS ([Cin*9][16]), then a dense
// [Cout x Cin*9] contraction per pixel.  Sampling rule: bilinear with zeros outside the image; the
// whole sample is zero when y <= -1, y >= H, x <= -1 or x >= W.
#include "common.h"

namespace {

constexpr int PXB = 16;  // output pixels per workgroup

struct Bil {
    int y0, x0;
    float ly, lx;
    bool ok00, ok01, ok10, ok11, inside;
};

__device__ __forceinline__ Bil bil_setup(float y, float x, int H, int W) {
    Bil s;
    s.inside = (y > -1.f) && (y < (float)H) && (x > -1.f) && (x < (float)W);
    const float fy = floorf(y), fx = floorf(x);
    s.y0 = (int)fy;
    s.x0 = (int)fx;
    s.ly = y - fy;
    s.lx = x - fx;
    const bool y0ok = s.y0 >= 0 && s.y0 <= H - 1, y1ok = s.y0 + 1 >= 0 && s.y0 + 1 <= H - 1;
    const bool x0ok = s.x0 >= 0 && s.x0 <= W - 1, x1ok = s.x0 + 1 >= 0 && s.x0 + 1 <= W - 1;
    s.ok00 = s.inside && y0ok && x0ok;
    s.ok01 = s.inside && y0ok && x1ok;
    s.ok10 = s.inside && y1ok && x0ok;
    s.ok11 = s.inside && y1ok && x1ok;
    return s;
}

// sample position of tap t at output pixel (b,h,w): y = h - 1 + t/3 + off[2t], x = w - 1 + t%3 + off[2t+1]
__device__ __forceinline__ Bil tap_setup(const float* __restrict__ offset, int b, int h, int w, int t, int H, int W) {
    const size_t hw = (size_t)H * W, pix = (size_t)h * W + w;
    const float oy = offset[((size_t)b * 18 + 2 * t) * hw + pix];
    const float ox = offset[((size_t)b * 18 + 2 * t + 1) * hw + pix];
    return bil_setup((float)(h - 1 + t / 3) + oy, (float)(w - 1 + t % 3) + ox, H, W);
}

__device__ __forceinline__ void stage_columns(const float* __restrict__ x, const float* __restrict__ offset,
                                              float* cols, int p0, int npix, int Cin, int H, int W) {
    const size_t hw = (size_t)H * W;
    for (int tp = threadIdx.x; tp < 9 * PXB; tp += blockDim.x) {
        const int t = tp / PXB, pl = tp % PXB, p = p0 + pl;
        if (p >= npix) {
            for (int c = 0; c < Cin; ++c) cols[(c * 9 + t) * PXB + pl] = 0.f;
            continue;
        }
        const int b = p / (H * W), h = (p / W) % H, w = p % W;
        const Bil s = tap_setup(offset, b, h, w, t, H, W);
        const float w00 = (1.f - s.ly) * (1.f - s.lx), w01 = (1.f - s.ly) * s.lx, w10 = s.ly * (1.f - s.lx),
                    w11 = s.ly * s.lx;
        const int o00 = s.y0 * W + s.x0;
        const float* xb = x + (size_t)b * Cin * hw;
        for (int c = 0; c < Cin; ++c) {
            const float* xc = xb + (size_t)c * hw;
            float v = 0.f;
            if (s.ok00) v += w00 * xc[o00];
            if (s.ok01) v += w01 * xc[o00 + 1];
            if (s.ok10) v += w10 * xc[o00 + W];
            if (s.ok11) v += w11 * xc[o00 + W + 1];
            cols[(c * 9 + t) * PXB + pl] = v;
        }
    }
}

__global__ __launch_bounds__(256) void deform_fwd_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                         const float* __restrict__ weight,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B,
                                                         int Cin, int Cout, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cols = smem;  // [Cin*9][PXB]
    const int npix = B * H * W, p0 = blockIdx.x * PXB, K = Cin * 9;
    stage_columns(x, offset, cols, p0, npix, Cin, H, W);
    __syncthreads();
    for (int e = threadIdx.x; e < Cout * PXB; e += blockDim.x) {
        const int o = e / PXB, pl = e % PXB, p = p0 + pl;
        if (p >= npix) continue;
        float acc = bias ? bias[o] : 0.f;
        const float* wr = weight + (size_t)o * K;
        for (int k = 0; k < K; ++k) acc += wr[k] * cols[k * PXB + pl];
        const int b = p / (H * W), rem = p % (H * W);
        y[((size_t)b * Cout + o) * H * W + rem] = acc;
    }
}

__global__ __launch_bounds__(256) void deform_bwd_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                         const float* __restrict__ weight,
                                                         const float* __restrict__ dy, float* __restrict__ dx,
                                                         float* __restrict__ d_offset, float* __restrict__ d_weight,
                                                         float* __restrict__ d_bias, int B, int Cin, int Cout, int H,
                                                         int W) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = Cin * 9;
    float* cols = smem;               // [K][PXB]
    float* dcols = cols + K * PXB;    // [K][PXB]
    float* dyl = dcols + K * PXB;     // [Cout][PXB]
    const int npix = B * H * W, p0 = blockIdx.x * PXB;
    const size_t hw = (size_t)H * W;
    stage_columns(x, offset, cols, p0, npix, Cin, H, W);
    for (int e = threadIdx.x; e < Cout * PXB; e += blockDim.x) {
        const int o = e / PXB, pl = e % PXB, p = p0 + pl;
        float v = 0.f;
        if (p < npix) v = dy[((size_t)(p / (H * W)) * Cout + o) * hw + p % (H * W)];
        dyl[e] = v;
    }
    __syncthreads();
    // d_weight[o][k] += sum_pl dy[o][pl] * cols[k][pl] ; d_bias[o] += sum_pl dy[o][pl]
    for (int e = threadIdx.x; e < Cout * K; e += blockDim.x) {
        const int o = e / K, k = e % K;
        float acc = 0.f;
#pragma unroll
        for (int pl = 0; pl < PXB; ++pl) acc += dyl[o * PXB + pl] * cols[k * PXB + pl];
        atomicAdd(d_weight + e, acc);
    }
    for (int o = threadIdx.x; o < Cout; o += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int pl = 0; pl < PXB; ++pl) acc += dyl[o * PXB + pl];
        atomicAdd(d_bias + o, acc);
    }
    // dcols[k][pl] = sum_o weight[o][k] * dy[o][pl]
    for (int e = threadIdx.x; e < K * PXB; e += blockDim.x) {
        const int k = e / PXB, pl = e % PXB;
        float acc = 0.f;
        for (int o = 0; o < Cout; ++o) acc += weight[(size_t)o * K + k] * dyl[o * PXB + pl];
        dcols[e] = acc;
    }
    __syncthreads();
    // scatter to dx, and offset gradients (d sample / d y, d sample / d x)
    for (int tp = threadIdx.x; tp < 9 * PXB; tp += blockDim.x) {
        const int t = tp / PXB, pl = tp % PXB, p = p0 + pl;
        if (p >= npix) continue;
        const int b = p / (H * W), h = (p / W) % H, w = p % W;
        const Bil s = tap_setup(offset, b, h, w, t, H, W);
        const float hy = 1.f - s.ly, hx = 1.f - s.lx;
        const int o00 = s.y0 * W + s.x0;
        const float* xb = x + (size_t)b * Cin * hw;
        float* dxb = dx + (size_t)b * Cin * hw;
        float gy = 0.f, gx = 0.f;
        for (int c = 0; c < Cin; ++c) {
            const float g = dcols[(c * 9 + t) * PXB + pl];
            const float* xc = xb + (size_t)c * hw;
            float* dxc = dxb + (size_t)c * hw;
            const float v00 = s.ok00 ? xc[o00] : 0.f, v01 = s.ok01 ? xc[o00 + 1] : 0.f;
            const float v10 = s.ok10 ? xc[o00 + W] : 0.f, v11 = s.ok11 ? xc[o00 + W + 1] : 0.f;
            if (s.ok00) atomicAdd(dxc + o00, g * hy * hx);
            if (s.ok01) atomicAdd(dxc + o00 + 1, g * hy * s.lx);
            if (s.ok10) atomicAdd(dxc + o00 + W, g * s.ly * hx);
            if (s.ok11) atomicAdd(dxc + o00 + W + 1, g * s.ly * s.lx);
            gy += g * (hx * (v10 - v00) + s.lx * (v11 - v01));
            gx += g * (hy * (v01 - v00) + s.ly * (v11 - v10));
        }
        const size_t pix = (size_t)h * W + w;
        d_offset[((size_t)b * 18 + 2 * t) * hw + pix] = gy;
        d_offset[((size_t)b * 18 + 2 * t + 1) * hw + pix] = gx;
    }
}

// ---- sampling / scatter halves for the im2col + GEMM formulation -------------------------------------------
// cols[b][c*9 + t][pix] = bilinear sample of x[b][c] at tap t of output pixel pix.  One thread per (b, t, pix); the
// channel loop issues its gathers 8 channels at a time (a dependent-latency chain per channel made the fused kernel
// above latency-bound: 82 us forward / 340 us backward on [8,64,16,16]).
constexpr int CH = 8;

__global__ __launch_bounds__(256) void deform_sample_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                            float* __restrict__ cols, int B, int Cin, int H, int W) {
    const int hw = H * W, total = B * 9 * hw;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int pix = e % hw, t = (e / hw) % 9, b = e / (9 * hw);
    const Bil s = tap_setup(offset, b, pix / W, pix % W, t, H, W);
    const float w00 = (1.f - s.ly) * (1.f - s.lx), w01 = (1.f - s.ly) * s.lx, w10 = s.ly * (1.f - s.lx), w11 = s.ly * s.lx;
    const int o00 = s.y0 * W + s.x0;
    const float* xb = x + (size_t)b * Cin * hw;
    float* cb = cols + ((size_t)b * Cin * 9 + t) * hw + pix;
    for (int c0 = 0; c0 < Cin; c0 += CH) {
        float v[CH][4];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const float* xc = xb + (size_t)(c0 + k) * hw;
            const bool on = c0 + k < Cin;
            v[k][0] = on && s.ok00 ? xc[o00] : 0.f;
            v[k][1] = on && s.ok01 ? xc[o00 + 1] : 0.f;
            v[k][2] = on && s.ok10 ? xc[o00 + W] : 0.f;
            v[k][3] = on && s.ok11 ? xc[o00 + W + 1] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (c0 + k < Cin) cb[(size_t)(c0 + k) * 9 * hw] = (w00 * v[k][0] + w01 * v[k][1]) + (w10 * v[k][2] + w11 * v[k][3]);
    }
}

// adjoint of the sampling: dx (atomicAdd scatter, dx pre-zeroed) and d_offset from dcols[b][c*9 + t][pix]
__global__ __launch_bounds__(256) void deform_scatter_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                             const float* __restrict__ dcols, float* __restrict__ dx,
                                                             float* __restrict__ d_offset, int B, int Cin, int H, int W) {
    const int hw = H * W, total = B * 9 * hw;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int pix = e % hw, t = (e / hw) % 9, b = e / (9 * hw);
    const Bil s = tap_setup(offset, b, pix / W, pix % W, t, H, W);
    const float hy = 1.f - s.ly, hx = 1.f - s.lx;
    const int o00 = s.y0 * W + s.x0;
    const float* xb = x + (size_t)b * Cin * hw;
    float* dxb = dx + (size_t)b * Cin * hw;
    const float* gb = dcols + ((size_t)b * Cin * 9 + t) * hw + pix;
    float gy = 0.f, gx = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += CH) {
        float v[CH][4], g[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const bool on = c0 + k < Cin;
            const float* xc = xb + (size_t)(c0 + k) * hw;
            g[k] = on ? gb[(size_t)(c0 + k) * 9 * hw] : 0.f;
            v[k][0] = on && s.ok00 ? xc[o00] : 0.f;
            v[k][1] = on && s.ok01 ? xc[o00 + 1] : 0.f;
            v[k][2] = on && s.ok10 ? xc[o00 + W] : 0.f;
            v[k][3] = on && s.ok11 ? xc[o00 + W + 1] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (c0 + k >= Cin) continue;
            float* dxc = dxb + (size_t)(c0 + k) * hw;
            if (s.ok00) atomicAdd(dxc + o00, g[k] * hy * hx);
            if (s.ok01) atomicAdd(dxc + o00 + 1, g[k] * hy * s.lx);
            if (s.ok10) atomicAdd(dxc + o00 + W, g[k] * s.ly * hx);
            if (s.ok11) atomicAdd(dxc + o00 + W + 1, g[k] * s.ly * s.lx);
            gy += g[k] * (hx * (v[k][2] - v[k][0]) + s.lx * (v[k][3] - v[k][1]));
            gx += g[k] * (hy * (v[k][1] - v[k][0]) + s.ly * (v[k][3] - v[k][2]));
        }
    }
    d_offset[((size_t)b * 18 + 2 * t) * hw + pix] = gy;
    d_offset[((size_t)b * 18 + 2 * t + 1) * hw + pix] = gx;
}

// The same adjoint without float atomics (round 3).  The kernel above issues 4.7 M global float atomics onto 131 k cells at the
// bridge ([8,64,16,16]: 36 addends per cell on average) and took 77 us alone on the main chain.  Measured on the way: LDS float
// atomics are no way out (ds_add_f32 ~190 cycles per wave instruction here: 92 us for the same scatter into LDS planes), and the
// d_offset loop -- one thread walking all channels -- was 16 us of it.  What is cheap is that the scatter's DESTINATIONS do not
// depend on the channel: every workgroup (`chunks` per sample) builds, per sample, the cell -> (item, weight) lists of all 4 * 9 * HW corner
// contributions once (integer LDS atomics only: a count pass, a prefix sum, a slot pass) and then GATHERS its 2 channels' dx from
// them: dx[c][cell] = sum over the cell's list of weight * dcols[c][item].  dx is written, not accumulated (no zero fill).  The
// order inside a list follows the slot atomics, so sums of > 2 addends can differ in the last bit between runs, as before.
// The same workgroups then compute d_offset (gathers only, 16 lanes per (tap, pixel) item).
constexpr int SC_CH = 2;
__global__ __launch_bounds__(256) void deform_scatter_lds_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                                 const float* __restrict__ dcols, float* __restrict__ dx,
                                                                 float* __restrict__ d_offset, int Cin, int H, int W, int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char scl[];
    const int hw = H * W, b = blockIdx.y, NI = 9 * hw;
    const float* xb = x + (size_t)b * Cin * hw;
    {
        // LDS: start[hw + 1] | cursor[256] (also the 256-thread scan scratch: hw <= 256) | wgt[4 NI] | itm[4 NI] (u16)
        int* start = reinterpret_cast<int*>(scl);
        int* cursor = start + hw + 1;
        float* wgt = reinterpret_cast<float*>(cursor + 256);
        unsigned short* itm = reinterpret_cast<unsigned short*>(wgt + 4 * NI);
        const int c0 = blockIdx.x * SC_CH, nch = min(SC_CH, Cin - c0);
        for (int e = threadIdx.x; e < 2 * hw + 1; e += 256) start[e] = 0;       // start[] doubles as the count array (shifted by one)
        // this thread's items (item = threadIdx.x + 256 i): all sample positions are set up first, so their offset loads are in
        // flight together instead of one dependent round trip per item; the set-ups stay in registers for the slot pass
        constexpr int MAXI = 9;                                                  // host: 9 * hw <= 256 * MAXI, i.e. hw <= 256
        Bil sv[MAXI];
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int item = threadIdx.x + 256 * i, it = item < NI ? item : 0, t = it / hw, pix = it - t * hw;
            sv[i] = tap_setup(offset, b, pix / W, pix % W, t, H, W);
            if (item >= NI) sv[i].ok00 = sv[i].ok01 = sv[i].ok10 = sv[i].ok11 = false;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {                                         // pass 1: how many contributions per cell
            const Bil& s = sv[i];
            const int o00 = s.y0 * W + s.x0;
            if (s.ok00) atomicAdd(start + 1 + o00, 1);
            if (s.ok01) atomicAdd(start + 1 + o00 + 1, 1);
            if (s.ok10) atomicAdd(start + 1 + o00 + W, 1);
            if (s.ok11) atomicAdd(start + 1 + o00 + W + 1, 1);
        }
        __syncthreads();
        {   // inclusive scan of the counts (hw <= 256: one cell per thread; Hillis-Steele through cursor[], cleared afterwards)
            int v = (int)threadIdx.x < hw ? start[threadIdx.x + 1] : 0;
            for (int d = 1; d < 256; d <<= 1) {
                cursor[threadIdx.x] = v;
                __syncthreads();
                if ((int)threadIdx.x >= d) v += cursor[threadIdx.x - d];
                __syncthreads();
            }
            if ((int)threadIdx.x < hw) start[threadIdx.x + 1] = v;
            cursor[threadIdx.x] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {                                         // pass 2: slots
            const Bil& s = sv[i];
            const int item = threadIdx.x + 256 * i;
            const float hy = 1.f - s.ly, hx = 1.f - s.lx;
            const int o00 = s.y0 * W + s.x0;
            auto put = [&](int cell, float wv) {
                const int slot = start[cell] + atomicAdd(cursor + cell, 1);
                wgt[slot] = wv;
                itm[slot] = (unsigned short)item;
            };
            if (s.ok00) put(o00, hy * hx);
            if (s.ok01) put(o00 + 1, hy * s.lx);
            if (s.ok10) put(o00 + W, s.ly * hx);
            if (s.ok11) put(o00 + W + 1, s.ly * s.lx);
        }
        __syncthreads();
        const float* gc = dcols + ((size_t)b * Cin + c0) * 9 * hw;              // [k][t][pix]: element (k, item) at k * 9 hw + item
        float* dxb = dx + ((size_t)b * Cin + c0) * hw;
        // gather: 4 lanes share a cell and take every 4th entry of its list (independent loads in flight), then a 4-lane butterfly
        for (int cq = threadIdx.x; cq < 4 * hw; cq += 256) {
            const int cell = cq >> 2, sub = cq & 3;
            float acc[SC_CH];
#pragma unroll
            for (int k = 0; k < SC_CH; ++k) acc[k] = 0.f;
            const int e1 = start[cell + 1];
            for (int e = start[cell] + sub; e < e1; e += 4) {
                const float wv = wgt[e];
                const float* gp = gc + itm[e];
#pragma unroll
                for (int k = 0; k < SC_CH; ++k)
                    if (k < nch) acc[k] = fmaf(wv, gp[(size_t)k * NI], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < SC_CH; ++k) {
                acc[k] += __shfl_xor(acc[k], 1, 64);
                acc[k] += __shfl_xor(acc[k], 2, 64);
            }
            if (sub == 0) {
#pragma unroll
                for (int k = 0; k < SC_CH; ++k)
                    if (k < nch) dxb[(size_t)k * hw + cell] = acc[k];
            }
        }
    }
    // d_offset: gathers only.  16 lanes share one (tap, pixel) item and split the channels (lane k takes k, k + 16, ...): every lane's
    // gathers are independent and in flight together, then a 16-lane butterfly sums them.  (One thread per item walking all Cin
    // channels in rounds of 8 -- the first scatter kernel's loop -- cost 16 us of its 77.)  The sample's 9 HW items are dealt over its
    // `chunks` workgroups (role B blocks of their own would each reserve the 57 KB list tile for nothing).
    const int k = threadIdx.x & 15;
    for (int item = blockIdx.x * 16 + (threadIdx.x >> 4); item < NI; item += 16 * chunks) {
        const int t = item / hw, pix = item - t * hw;
        const Bil s = tap_setup(offset, b, pix / W, pix % W, t, H, W);
        const float hy = 1.f - s.ly, hx = 1.f - s.lx;
        const int o00 = s.y0 * W + s.x0;
        const float* gb = dcols + ((size_t)b * Cin * 9 + t) * hw + pix;
        float gy = 0.f, gx = 0.f;
        for (int c = k; c < Cin; c += 16) {
            const float* xc = xb + (size_t)c * hw;
            const float g = gb[(size_t)c * 9 * hw];
            const float v0 = s.ok00 ? xc[o00] : 0.f, v1 = s.ok01 ? xc[o00 + 1] : 0.f, v2 = s.ok10 ? xc[o00 + W] : 0.f,
                        v3 = s.ok11 ? xc[o00 + W + 1] : 0.f;
            gy += g * (hx * (v2 - v0) + s.lx * (v3 - v1));
            gx += g * (hy * (v1 - v0) + s.ly * (v3 - v2));
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            gy += __shfl_xor(gy, m, 64);
            gx += __shfl_xor(gx, m, 64);
        }
        if (k == 0) {
            d_offset[((size_t)b * 18 + 2 * t) * hw + pix] = gy;
            d_offset[((size_t)b * 18 + 2 * t + 1) * hw + pix] = gx;
        }
    }
}

inline size_t scatter_lds_bytes(int H, int W) {
    const size_t hw = (size_t)H * W;
    return (hw + 1 + 256) * sizeof(int) + 4 * 9 * hw * (sizeof(float) + sizeof(unsigned short)) + 16;
}

}  // namespace

extern "C" int kmu_deform_conv2d_fwd(const float* x, const float* offset, const float* weight, const float* bias,
                                     float* y, int B, int Cin, int Cout, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && offset && weight && y, "deform_conv2d_fwd: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "deform_conv2d_fwd: bad dims");
    const size_t lds = (size_t)Cin * 9 * PXB * sizeof(float);
    KMU_REQUIRE(lds <= 160 * 1024, "deform_conv2d_fwd: Cin=%d too large for the LDS column tile", Cin);
    KMU_MAX_LDS(deform_fwd_kernel, lds);
    const int blocks = kmu::cdiv(B * H * W, PXB);
    hipLaunchKernelGGL(deform_fwd_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, x, offset, weight, bias, y,
                       B, Cin, Cout, H, W);
    return kmu::launch_status("deform_conv2d_fwd");
}

extern "C" int kmu_deform_conv2d_bwd(const float* x, const float* offset, const float* weight, const float* dy,
                                     float* dx, float* d_offset, float* d_weight, float* d_bias, int B, int Cin,
                                     int Cout, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && offset && weight && dy && dx && d_offset && d_weight && d_bias, "deform_conv2d_bwd: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "deform_conv2d_bwd: bad dims");
    const size_t lds = ((size_t)2 * Cin * 9 + Cout) * PXB * sizeof(float);
    KMU_REQUIRE(lds <= 160 * 1024, "deform_conv2d_bwd: channels too large for the LDS tiles");
    KMU_MAX_LDS(deform_bwd_kernel, lds);
    const int blocks = kmu::cdiv(B * H * W, PXB);
    hipLaunchKernelGGL(deform_bwd_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, x, offset, weight, dy, dx,
                       d_offset, d_weight, d_bias, B, Cin, Cout, H, W);
    return kmu::launch_status("deform_conv2d_bwd");
}

extern "C" int kmu_deform_sample_fwd(const float* x, const float* offset, float* cols, int B, int Cin, int H, int W,
                                     kmu_stream_t stream) {
    KMU_REQUIRE(x && offset && cols, "deform_sample_fwd: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && (long)B * 9 * H * W < (1L << 31), "deform_sample_fwd: bad dims");
    hipLaunchKernelGGL(deform_sample_kernel, dim3(kmu::cdiv(B * 9 * H * W, 256)), dim3(256), 0, (hipStream_t)stream, x, offset, cols, B, Cin,
                       H, W);
    return kmu::launch_status("deform_sample_fwd");
}

extern "C" int kmu_deform_sample_bwd(const float* x, const float* offset, const float* dcols, float* dx, float* d_offset, int B, int Cin,
                                     int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && offset && dcols && dx && d_offset, "deform_sample_bwd: null pointer");
    KMU_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && (long)B * 9 * H * W < (1L << 31), "deform_sample_bwd: bad dims");
    hipLaunchKernelGGL(deform_scatter_kernel, dim3(kmu::cdiv(B * 9 * H * W, 256)), dim3(256), 0, (hipStream_t)stream, x, offset, dcols, dx,
                       d_offset, B, Cin, H, W);
    return kmu::launch_status("deform_sample_bwd");
}

// the list form: dx is WRITTEN (no zero fill needed).  The per-sample cell lists live in LDS: kmu_deform_sample_bwd_lds_supported
extern "C" int kmu_deform_sample_bwd_lds_supported(int B, int Cin, int H, int W) {
    return B > 0 && B <= 65535 && Cin > 0 && H > 0 && W > 0 && H * W <= 256 && scatter_lds_bytes(H, W) <= 64 * 1024;
}
extern "C" int kmu_deform_sample_bwd_lds(const float* x, const float* offset, const float* dcols, float* dx, float* d_offset, int B, int Cin,
                                         int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && offset && dcols && dx && d_offset, "deform_sample_bwd_lds: null pointer");
    KMU_REQUIRE(kmu_deform_sample_bwd_lds_supported(B, Cin, H, W), "deform_sample_bwd_lds: %dx%d planes do not fit the LDS tile (use kmu_deform_sample_bwd)",
                H, W);
    const int chunks = kmu::cdiv(Cin, SC_CH);
    hipLaunchKernelGGL(deform_scatter_lds_kernel, dim3(chunks, B), dim3(256), scatter_lds_bytes(H, W),
                       (hipStream_t)stream, x, offset, dcols, dx, d_offset, Cin, H, W, chunks);
    return kmu::launch_status("deform_sample_bwd_lds");
}
