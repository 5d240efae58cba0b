// K3: DySample x2 ('lp', 4 groups) gather kernels for gfx950.  COMPILE WITH -ffp-contract=off.
//
// Replaces DySample_md.py:49-68 of the reference downstream of the 1x1 offset conv:
//   offset = conv_out*0.25 + init_pos                                   (:67)
//   coords = 2*((w+.5, h+.5) + offset)/(W,H) - 1 ; pixel_shuffle(2)     (:50-59)
//   F.grid_sample(x.view(B*4,16,H,W), coords, bilinear, border, align_corners=False)   (:60-61)
// Offset channel layout (from the view/pixel_shuffle chain): ch = xy*16 + g*4 + i*2 + j; output
// pixel (2h+i, 2w+j) of group g samples around input (h, w).
//
// Index generation is integer work and must be bit-exact against oracle/dysample.py
// (sample_indices): every fp32 operation below is written in the oracle's order and this file is
// built without FMA contraction, so floor() sees bit-identical coordinates.
// Pure HBM gather: one thread per (b, group, oy, ox) handles the group's 16 channels; consecutive
// threads walk ox, so the 4-corner reads of a wave fall in a handful of 128-B lines per channel
// and the writes are fully coalesced.
#include "common.h"

namespace {

struct Samp {
    int x0, y0, x1, y1;
    float fx, fy;
    bool in_x, in_y;  // un-clipped coordinate strictly inside (0, size-1): gradient flows
};

__device__ __forceinline__ Samp dys_coords(const float* __restrict__ conv_out, const float* __restrict__ init_pos,
                                           int b, int g, int oy, int ox, int H, int W) {
    const int h = oy >> 1, i = oy & 1, w = ox >> 1, j = ox & 1;
    const int chx = g * 4 + i * 2 + j, chy = 16 + chx;
    const size_t hw = (size_t)H * W, pix = (size_t)h * W + w;
    const float offx = conv_out[((size_t)b * 32 + chx) * hw + pix] * 0.25f + init_pos[chx];
    const float offy = conv_out[((size_t)b * 32 + chy) * hw + pix] * 0.25f + init_pos[chy];
    // normalised grid coordinate, then grid_sample's un-normalisation (align_corners=False)
    float gx = ((float)w + 0.5f) + offx;
    gx = 2.f * gx;
    gx = gx / (float)W;
    gx = gx - 1.f;
    float gy = ((float)h + 0.5f) + offy;
    gy = 2.f * gy;
    gy = gy / (float)H;
    gy = gy - 1.f;
    float px = gx + 1.f;
    px = px * (float)W;
    px = px - 1.f;
    px = px / 2.f;
    float py = gy + 1.f;
    py = py * (float)H;
    py = py - 1.f;
    py = py / 2.f;
    Samp s;
    s.in_x = (px > 0.f) && (px < (float)(W - 1));
    s.in_y = (py > 0.f) && (py < (float)(H - 1));
    px = fminf(fmaxf(px, 0.f), (float)(W - 1));  // padding_mode='border'
    py = fminf(fmaxf(py, 0.f), (float)(H - 1));
    const float fx0 = floorf(px), fy0 = floorf(py);
    s.x0 = (int)fx0;
    s.y0 = (int)fy0;
    s.x1 = min(s.x0 + 1, W - 1);
    s.y1 = min(s.y0 + 1, H - 1);
    s.fx = px - fx0;
    s.fy = py - fy0;
    return s;
}

__global__ __launch_bounds__(256) void dysample_fwd_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ conv_out,
                                                           const float* __restrict__ init_pos, float* __restrict__ y,
                                                           int32_t* __restrict__ ix0, int32_t* __restrict__ iy0, int B,
                                                           int C, int H, int W) {
    const int OH = 2 * H, OW = 2 * W, cg = C / 4;
    const size_t total = (size_t)B * 4 * OH * OW;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int ox = t % OW;
        size_t r = t / OW;
        const int oy = r % OH;
        r /= OH;
        const int g = r % 4, b = (int)(r / 4);
        const Samp s = dys_coords(conv_out, init_pos, b, g, oy, ox, H, W);
        if (ix0) {
            ix0[t] = s.x0;
            iy0[t] = s.y0;
        }
        const float w00 = (1.f - s.fx) * (1.f - s.fy), w01 = s.fx * (1.f - s.fy), w10 = (1.f - s.fx) * s.fy,
                    w11 = s.fx * s.fy;
        const size_t hw = (size_t)H * W;
        const float* xp = x + ((size_t)b * C + (size_t)g * cg) * hw;
        float* yp = y + (((size_t)b * C + (size_t)g * cg) * OH + oy) * OW + ox;
        const int o00 = s.y0 * W + s.x0, o01 = s.y0 * W + s.x1, o10 = s.y1 * W + s.x0, o11 = s.y1 * W + s.x1;
        for (int c = 0; c < cg; ++c) {
            const float* xc = xp + (size_t)c * hw;
            yp[(size_t)c * OH * OW] = xc[o00] * w00 + xc[o01] * w01 + xc[o10] * w10 + xc[o11] * w11;
        }
    }
}

constexpr int DCH = 8;   // channels whose gathers are in flight together

__global__ __launch_bounds__(256) void dysample_bwd_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ conv_out,
                                                           const float* __restrict__ init_pos,
                                                           const float* __restrict__ dy, float* __restrict__ dx,
                                                           float* __restrict__ d_conv_out, int B, int C, int H, int W) {
    const int OH = 2 * H, OW = 2 * W, cg = C / 4;
    const size_t total = (size_t)B * 4 * OH * OW;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int ox = t % OW;
        size_t r = t / OW;
        const int oy = r % OH;
        r /= OH;
        const int g = r % 4, b = (int)(r / 4);
        const Samp s = dys_coords(conv_out, init_pos, b, g, oy, ox, H, W);
        const float w00 = (1.f - s.fx) * (1.f - s.fy), w01 = s.fx * (1.f - s.fy), w10 = (1.f - s.fx) * s.fy,
                    w11 = s.fx * s.fy;
        const size_t hw = (size_t)H * W;
        const size_t cbase = ((size_t)b * C + (size_t)g * cg) * hw;
        const float* dyp = dy + (((size_t)b * C + (size_t)g * cg) * OH + oy) * OW + ox;
        const int o00 = s.y0 * W + s.x0, o01 = s.y0 * W + s.x1, o10 = s.y1 * W + s.x0, o11 = s.y1 * W + s.x1;
        float gpx = 0.f, gpy = 0.f;
        for (int c0 = 0; c0 < cg; c0 += DCH) {       // gathers issued DCH channels at a time (not one latency chain per channel)
            float go[DCH], v[DCH][4];
#pragma unroll
            for (int k = 0; k < DCH; ++k) {
                const bool on = c0 + k < cg;
                const float* xc = x + cbase + (size_t)(c0 + k) * hw;
                go[k] = on ? dyp[(size_t)(c0 + k) * OH * OW] : 0.f;
                v[k][0] = on ? xc[o00] : 0.f, v[k][1] = on ? xc[o01] : 0.f, v[k][2] = on ? xc[o10] : 0.f, v[k][3] = on ? xc[o11] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < DCH; ++k) {
                if (c0 + k >= cg) continue;
                float* dxc = dx + cbase + (size_t)(c0 + k) * hw;
                atomicAdd(dxc + o00, go[k] * w00);
                atomicAdd(dxc + o01, go[k] * w01);
                atomicAdd(dxc + o10, go[k] * w10);
                atomicAdd(dxc + o11, go[k] * w11);
                gpx += go[k] * ((v[k][1] - v[k][0]) * (1.f - s.fy) + (v[k][3] - v[k][2]) * s.fy);
                gpy += go[k] * ((v[k][2] - v[k][0]) * (1.f - s.fx) + (v[k][3] - v[k][1]) * s.fx);
            }
        }
        // d px / d offset = (W/2)*(2/W) = 1 where the coordinate is not clipped (ATen
        // clip_coordinates_set_grad: zero at and beyond the borders); offset = 0.25 * conv_out + init_pos
        const int h = oy >> 1, i = oy & 1, w = ox >> 1, j = ox & 1;
        const int chx = g * 4 + i * 2 + j, chy = 16 + chx;
        const size_t pix = (size_t)h * W + w;
        d_conv_out[((size_t)b * 32 + chx) * hw + pix] = s.in_x ? 0.25f * gpx : 0.f;
        d_conv_out[((size_t)b * 32 + chy) * hw + pix] = s.in_y ? 0.25f * gpy : 0.f;
    }
}

// Tiled backward: one workgroup = one 32x32 output tile of one (b, group).  Its samples land (for |offset| < ~2)
// inside the 20x20 input window around the tile's 16x16 source pixels, so the scatter-add goes to an LDS window
// (ds_add_f32) and is flushed to HBM once -- 16*400 global atomics per workgroup instead of 16*4*1024.
// Samples outside the window (large learned offsets) fall back to direct global atomics: correct for any offset.
constexpr int BT = 32, BWIN = BT / 2 + 4;  // output tile edge, input window edge

__global__ __launch_bounds__(256) void dysample_bwd_tiled_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ conv_out,
                                                                 const float* __restrict__ init_pos,
                                                                 const float* __restrict__ dy, float* __restrict__ dx,
                                                                 float* __restrict__ d_conv_out, int C, int H, int W,
                                                                 int tilesX) {
    extern __shared__ __attribute__((aligned(16))) float win[];  // [cg][BWIN*BWIN]
    const int OH = 2 * H, OW = 2 * W, cg = C / 4;
    const int b = blockIdx.z, g = blockIdx.y;
    const int oy0 = (blockIdx.x / tilesX) * BT, ox0 = (blockIdx.x % tilesX) * BT;
    const int wy0 = oy0 / 2 - 2, wx0 = ox0 / 2 - 2;
    for (int e = threadIdx.x; e < cg * BWIN * BWIN; e += 256) win[e] = 0.f;
    __syncthreads();
    const size_t hw = (size_t)H * W;
    const size_t cbase = ((size_t)b * C + (size_t)g * cg) * hw;
    for (int p = threadIdx.x; p < BT * BT; p += 256) {
        const int oy = oy0 + p / BT, ox = ox0 + p % BT;
        if (oy >= OH || ox >= OW) continue;
        const Samp s = dys_coords(conv_out, init_pos, b, g, oy, ox, H, W);
        const float w00 = (1.f - s.fx) * (1.f - s.fy), w01 = s.fx * (1.f - s.fy), w10 = (1.f - s.fx) * s.fy,
                    w11 = s.fx * s.fy;
        const bool inwin = s.y0 >= wy0 && s.y1 < wy0 + BWIN && s.x0 >= wx0 && s.x1 < wx0 + BWIN;
        const int o00 = s.y0 * W + s.x0, o01 = s.y0 * W + s.x1, o10 = s.y1 * W + s.x0, o11 = s.y1 * W + s.x1;
        const int l00 = (s.y0 - wy0) * BWIN + (s.x0 - wx0), l01 = (s.y0 - wy0) * BWIN + (s.x1 - wx0),
                  l10 = (s.y1 - wy0) * BWIN + (s.x0 - wx0), l11 = (s.y1 - wy0) * BWIN + (s.x1 - wx0);
        const float* dyp = dy + (((size_t)b * C + (size_t)g * cg) * OH + oy) * OW + ox;
        float gpx = 0.f, gpy = 0.f;
        for (int c0 = 0; c0 < cg; c0 += DCH) {
            float go[DCH], v[DCH][4];
#pragma unroll
            for (int k = 0; k < DCH; ++k) {
                const bool on = c0 + k < cg;
                const float* xc = x + cbase + (size_t)(c0 + k) * hw;
                go[k] = on ? dyp[(size_t)(c0 + k) * OH * OW] : 0.f;
                v[k][0] = on ? xc[o00] : 0.f, v[k][1] = on ? xc[o01] : 0.f, v[k][2] = on ? xc[o10] : 0.f, v[k][3] = on ? xc[o11] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < DCH; ++k) {
                if (c0 + k >= cg) continue;
                if (inwin) {
                    float* wc = win + (c0 + k) * BWIN * BWIN;
                    atomicAdd(wc + l00, go[k] * w00);
                    atomicAdd(wc + l01, go[k] * w01);
                    atomicAdd(wc + l10, go[k] * w10);
                    atomicAdd(wc + l11, go[k] * w11);
                } else {
                    float* dxc = dx + cbase + (size_t)(c0 + k) * hw;
                    atomicAdd(dxc + o00, go[k] * w00);
                    atomicAdd(dxc + o01, go[k] * w01);
                    atomicAdd(dxc + o10, go[k] * w10);
                    atomicAdd(dxc + o11, go[k] * w11);
                }
                gpx += go[k] * ((v[k][1] - v[k][0]) * (1.f - s.fy) + (v[k][3] - v[k][2]) * s.fy);
                gpy += go[k] * ((v[k][2] - v[k][0]) * (1.f - s.fx) + (v[k][3] - v[k][1]) * s.fx);
            }
        }
        const int h = oy >> 1, i = oy & 1, w = ox >> 1, j = ox & 1;
        const int chx = g * 4 + i * 2 + j, chy = 16 + chx;
        const size_t pix = (size_t)h * W + w;
        d_conv_out[((size_t)b * 32 + chx) * hw + pix] = s.in_x ? 0.25f * gpx : 0.f;
        d_conv_out[((size_t)b * 32 + chy) * hw + pix] = s.in_y ? 0.25f * gpy : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cg * BWIN * BWIN; e += 256) {
        const float v = win[e];
        if (v == 0.f) continue;
        const int c = e / (BWIN * BWIN), r = e % (BWIN * BWIN);
        const int yy = wy0 + r / BWIN, xx = wx0 + r % BWIN;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) atomicAdd(dx + cbase + (size_t)c * hw + (size_t)yy * W + xx, v);
    }
}

// Channel-major variant for cg = C/4 = 16 (KM-UNet: C = 64): lane = (channel c = tid & 15, pixel slot = tid >> 4).  The 64
// lanes of a wave then scatter into 16 different channel planes of the LDS window (plane stride 401 words => the 16
// lanes of a DPP row hit 16 distinct banks) instead of 64 neighbouring pixels of ONE plane, whose 2x-upsampling footprints
// collide on the same few addresses and serialise the ds_add_f32; the per-pixel offset gradient (a sum over the group's
// channels) is a 16-lane row reduction.  Same arithmetic per (pixel, channel) as the tiled kernel above.

template <int TB>   // output tile edge: 32 (window 20x20) or 16 (window 12x12: 4x the workgroups, for the small levels)
__global__ __launch_bounds__(256) void dysample_bwd_chan_kernel(const float* __restrict__ x, const float* __restrict__ conv_out,
                                                                const float* __restrict__ init_pos, const float* __restrict__ dy,
                                                                float* __restrict__ dx, float* __restrict__ d_conv_out, int C, int H,
                                                                int W, int tilesX) {
    constexpr int WN = TB / 2 + 4, WS = WN * WN + 1;   // window edge; plane stride (odd => the 16 channel lanes hit 16 banks)
    extern __shared__ __attribute__((aligned(16))) float win[];  // [16][WS]
    const int OH = 2 * H, OW = 2 * W;
    constexpr int cg = 16;
    const int b = blockIdx.z, g = blockIdx.y;
    const int oy0 = (blockIdx.x / tilesX) * TB, ox0 = (blockIdx.x % tilesX) * TB;
    const int wy0 = oy0 / 2 - 2, wx0 = ox0 / 2 - 2;
    for (int e = threadIdx.x; e < cg * WS; e += 256) win[e] = 0.f;
    __syncthreads();
    const size_t hw = (size_t)H * W;
    const int c = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const size_t cbase = ((size_t)b * C + (size_t)g * cg + c) * hw;
    const float* xc = x + cbase;
    float* dxc = dx + cbase;
    float* wc = win + c * WS;
    const float* dyc = dy + ((size_t)b * C + (size_t)g * cg + c) * OH * OW;
    for (int it = 0; it < TB * TB / 16; ++it) {
        const int p = it * 16 + slot, oy = oy0 + p / TB, ox = ox0 + p % TB;
        const bool live = oy < OH && ox < OW;
        float gpx = 0.f, gpy = 0.f;
        Samp s;
        if (live) {
            s = dys_coords(conv_out, init_pos, b, g, oy, ox, H, W);
            const float w00 = (1.f - s.fx) * (1.f - s.fy), w01 = s.fx * (1.f - s.fy), w10 = (1.f - s.fx) * s.fy, w11 = s.fx * s.fy;
            const int o00 = s.y0 * W + s.x0, o01 = s.y0 * W + s.x1, o10 = s.y1 * W + s.x0, o11 = s.y1 * W + s.x1;
            const float go = dyc[(size_t)oy * OW + ox];
            const float v00 = xc[o00], v01 = xc[o01], v10 = xc[o10], v11 = xc[o11];
            if (s.y0 >= wy0 && s.y1 < wy0 + WN && s.x0 >= wx0 && s.x1 < wx0 + WN) {
                const int ly0 = (s.y0 - wy0) * WN, ly1 = (s.y1 - wy0) * WN, lx0 = s.x0 - wx0, lx1 = s.x1 - wx0;
                atomicAdd(wc + ly0 + lx0, go * w00);
                atomicAdd(wc + ly0 + lx1, go * w01);
                atomicAdd(wc + ly1 + lx0, go * w10);
                atomicAdd(wc + ly1 + lx1, go * w11);
            } else {
                atomicAdd(dxc + o00, go * w00);
                atomicAdd(dxc + o01, go * w01);
                atomicAdd(dxc + o10, go * w10);
                atomicAdd(dxc + o11, go * w11);
            }
            gpx = go * ((v01 - v00) * (1.f - s.fy) + (v11 - v10) * s.fy);
            gpy = go * ((v10 - v00) * (1.f - s.fx) + (v11 - v01) * s.fx);
        }
        // sum over the 16 channel lanes of this pixel (one DPP row): row_shr 1, 2, 4, 8 leaves the total in lane 15
        gpx += kmu::dpp_mov<0x111, 0xf>(0.f, gpx), gpy += kmu::dpp_mov<0x111, 0xf>(0.f, gpy);
        gpx += kmu::dpp_mov<0x112, 0xf>(0.f, gpx), gpy += kmu::dpp_mov<0x112, 0xf>(0.f, gpy);
        gpx += kmu::dpp_mov<0x114, 0xf>(0.f, gpx), gpy += kmu::dpp_mov<0x114, 0xf>(0.f, gpy);
        gpx += kmu::dpp_mov<0x118, 0xf>(0.f, gpx), gpy += kmu::dpp_mov<0x118, 0xf>(0.f, gpy);
        if (live && c == 15) {
            const int h = oy >> 1, i = oy & 1, w = ox >> 1, j = ox & 1;
            const int chx = g * 4 + i * 2 + j, chy = 16 + chx;
            const size_t pix = (size_t)h * W + w;
            d_conv_out[((size_t)b * 32 + chx) * hw + pix] = s.in_x ? 0.25f * gpx : 0.f;
            d_conv_out[((size_t)b * 32 + chy) * hw + pix] = s.in_y ? 0.25f * gpy : 0.f;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cg * WN * WN; e += 256) {
        const int cc = e / (WN * WN), r = e % (WN * WN);
        const float v = win[cc * WS + r];
        if (v == 0.f) continue;
        const int yy = wy0 + r / WN, xx = wx0 + r % WN;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) atomicAdd(dx + ((size_t)b * C + (size_t)g * cg + cc) * hw + (size_t)yy * W + xx, v);
    }
}

int grid_for(size_t total) {
    size_t blocks = (total + 255) / 256;
    return (int)(blocks > 8192 ? 8192 : (blocks ? blocks : 1));
}

}  // namespace

extern "C" int kmu_dysample_lp_fwd(const float* x, const float* conv_out, const float* init_pos, float* y,
                                   int32_t* ix0, int32_t* iy0, int B, int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && conv_out && init_pos && y, "dysample_lp_fwd: null pointer");
    KMU_REQUIRE((ix0 == nullptr) == (iy0 == nullptr), "dysample_lp_fwd: pass both or neither index outputs");
    KMU_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, "dysample_lp_fwd: bad dims (C must be a multiple of 4 groups)");
    const size_t total = (size_t)B * 4 * 2 * H * 2 * W;
    hipLaunchKernelGGL(dysample_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, conv_out,
                       init_pos, y, ix0, iy0, B, C, H, W);
    return kmu::launch_status("dysample_lp_fwd");
}

extern "C" int kmu_dysample_lp_bwd(const float* x, const float* conv_out, const float* init_pos, const float* dy,
                                   float* dx, float* d_conv_out, int B, int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && conv_out && init_pos && dy && dx && d_conv_out, "dysample_lp_bwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, "dysample_lp_bwd: bad dims");
    const size_t lds = (size_t)(C / 4) * BWIN * BWIN * sizeof(float);
    if (C == 64 && B <= 65535) {
        // 16x16 output tiles (12x12 input window): the per-workgroup loop is a serial chain of gathers and LDS atomics, so
        // more, shorter workgroups win at every level -- measured (B=8) against 32x32 tiles: 16x16 input 94 -> 28 us,
        // 32x32 94 -> 53 us, 64x64 183 -> 177 us
        constexpr int tb = 16;
        const int tilesX = kmu::cdiv(2 * W, tb), tilesY = kmu::cdiv(2 * H, tb);
        const size_t ldsc = (size_t)16 * ((tb / 2 + 4) * (tb / 2 + 4) + 1) * sizeof(float);
        hipLaunchKernelGGL(dysample_bwd_chan_kernel<tb>, dim3(tilesX * tilesY, 4, B), dim3(256), ldsc, (hipStream_t)stream, x, conv_out,
                           init_pos, dy, dx, d_conv_out, C, H, W, tilesX);
        return kmu::launch_status("dysample_lp_bwd");
    }
    if (lds <= 64 * 1024 && B <= 65535) {
        const int tilesX = kmu::cdiv(2 * W, BT), tilesY = kmu::cdiv(2 * H, BT);
        KMU_MAX_LDS(dysample_bwd_tiled_kernel, lds);
        hipLaunchKernelGGL(dysample_bwd_tiled_kernel, dim3(tilesX * tilesY, 4, B), dim3(256), lds, (hipStream_t)stream, x,
                           conv_out, init_pos, dy, dx, d_conv_out, C, H, W, tilesX);
        return kmu::launch_status("dysample_lp_bwd");
    }
    const size_t total = (size_t)B * 4 * 2 * H * 2 * W;
    hipLaunchKernelGGL(dysample_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, conv_out,
                       init_pos, dy, dx, d_conv_out, B, C, H, W);
    return kmu::launch_status("dysample_lp_bwd");
}
