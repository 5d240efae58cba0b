// Bilinear resampling with align_corners=True between the encoder pyramid levels (KM_UNetV3_SH.py:487-492, 503-507:
// F.interpolate(e1 / e2, size=d.shape[2:], mode="bilinear", align_corners=True)).  ATen's upsample_bilinear2d_out_frame takes
// ~100 us per call on these 1-2 MB tensors; this is a plain coalesced gather (one thread per output pixel and channel).
//   src coordinate of output (oy, ox): sy = oy * (Hi-1)/(Ho-1), sx = ox * (Wi-1)/(Wo-1)   (0 when the output extent is 1)
// Backward is the exact adjoint in GATHER form: every input pixel sums the (at most (ceil(1/scale)+1)^2) outputs whose 2x2
// footprint contains it -- deterministic, no float atomics.
#include "common.h"

namespace {

__device__ __forceinline__ void src_coord(int o, float scale, int n_in, int& i0, int& i1, float& w1) {
    const float s = (float)o * scale;
    i0 = min((int)s, n_in - 1);
    i1 = min(i0 + 1, n_in - 1);
    w1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void resize_bilinear_ac_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int BC, int Hi,
                                                                     int Wi, int Ho, int Wo, float sy, float sx) {
    const size_t n = (size_t)BC * Ho * Wo;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int ox = e % Wo, oy = (e / Wo) % Ho;
        const size_t bc = e / ((size_t)Wo * Ho);
        int y0, y1, x0, x1;
        float wy, wx;
        src_coord(oy, sy, Hi, y0, y1, wy);
        src_coord(ox, sx, Wi, x0, x1, wx);
        const float* p = x + bc * Hi * Wi;
        const float a = p[y0 * Wi + x0], b = p[y0 * Wi + x1], c = p[y1 * Wi + x0], d = p[y1 * Wi + x1];
        // ATen's order: (1-wy) * ((1-wx) a + wx b) + wy * ((1-wx) c + wx d)
        y[e] = (1.f - wy) * ((1.f - wx) * a + wx * b) + wy * ((1.f - wx) * c + wx * d);
    }
}

// adjoint, gather form: dx[iy][ix] = sum over outputs (oy, ox) with iy in {y0(oy), y1(oy)} and ix in {x0(ox), x1(ox)}
__global__ __launch_bounds__(256) void resize_bilinear_ac_bwd_kernel(const float* __restrict__ gy, float* __restrict__ dx, int BC, int Hi,
                                                                     int Wi, int Ho, int Wo, float sy, float sx, float inv_sy,
                                                                     float inv_sx) {
    const size_t n = (size_t)BC * Hi * Wi;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int ix = e % Wi, iy = (e / Wi) % Hi;
        const size_t bc = e / ((size_t)Wi * Hi);
        // outputs whose source coordinate lies in (iy - 1, iy + 1): oy in [ceil((iy-1)/sy), floor((iy+1)/sy)], widened by one
        const int oy_lo = max(0, (int)floorf((float)(iy - 1) * inv_sy) - 1), oy_hi = min(Ho - 1, (int)ceilf((float)(iy + 1) * inv_sy) + 1);
        const int ox_lo = max(0, (int)floorf((float)(ix - 1) * inv_sx) - 1), ox_hi = min(Wo - 1, (int)ceilf((float)(ix + 1) * inv_sx) + 1);
        const float* g = gy + bc * Ho * Wo;
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1;
            float wy;
            src_coord(oy, sy, Hi, y0, y1, wy);
            float cy = 0.f;
            if (y0 == iy) cy += 1.f - wy;
            if (y1 == iy) cy += wy;
            if (cy == 0.f) continue;
            float row = 0.f;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1;
                float wx;
                src_coord(ox, sx, Wi, x0, x1, wx);
                float cx = 0.f;
                if (x0 == ix) cx += 1.f - wx;
                if (x1 == ix) cx += wx;
                if (cx != 0.f) row += cx * g[oy * Wo + ox];
            }
            acc += cy * row;
        }
        dx[e] = acc;
    }
}

}  // namespace

extern "C" int kmu_resize_bilinear_ac_fwd(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, kmu_stream_t stream) {
    KMU_REQUIRE(x && y, "resize_bilinear_ac_fwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "resize_bilinear_ac_fwd: bad dims");
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const size_t n = (size_t)B * C * Ho * Wo;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(resize_bilinear_ac_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, B * C, Hi, Wi, Ho, Wo, sy, sx);
    return kmu::launch_status("resize_bilinear_ac_fwd");
}

extern "C" int kmu_resize_bilinear_ac_bwd(const float* gy, float* dx, int B, int C, int Hi, int Wi, int Ho, int Wo, kmu_stream_t stream) {
    KMU_REQUIRE(gy && dx, "resize_bilinear_ac_bwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "resize_bilinear_ac_bwd: bad dims");
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    // sy == 0 (one output row): every output reads input row 0 -> the search window is the whole output
    const float inv_sy = sy > 0.f ? 1.f / sy : (float)Ho, inv_sx = sx > 0.f ? 1.f / sx : (float)Wo;
    const size_t n = (size_t)B * C * Hi * Wi;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(resize_bilinear_ac_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gy, dx, B * C, Hi, Wi, Ho, Wo, sy,
                       sx, inv_sy, inv_sx);
    return kmu::launch_status("resize_bilinear_ac_bwd");
}
