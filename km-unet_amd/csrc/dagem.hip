// DAGEM's edge features (DAGEM_md.py:56-62): for every (b, c) plane of x [B,C,H,W]
//     edge[b,c,h,w,k] = x[h,w] * x[nbr_k(h,w)],   nbr = (h-1,w), (h+1,w), (h,w-1), (h,w+1)  cyclic  (torch.roll by +1 / -1)
// The reference builds them with four rolls, a stack and a product (6 launches, ~12 backward); here one gather kernel each way
// (the backward is written as a gather too: pixel p collects its own four products and the four in which it is the neighbour).
#include "common.h"

using kmu::floatx4;

namespace {

__global__ __launch_bounds__(256) void edge_fwd_kernel(const float* __restrict__ x, float* __restrict__ edge, int H, int W, size_t total) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int HW = H * W, p = (int)(t % HW), h = p / W, w = p - h * W;
    const float* xp = x + (t - p);
    const float v = xp[p];
    const int hm = h == 0 ? H - 1 : h - 1, hp = h == H - 1 ? 0 : h + 1, wm = w == 0 ? W - 1 : w - 1, wp = w == W - 1 ? 0 : w + 1;
    reinterpret_cast<floatx4*>(edge)[t] = floatx4{v * xp[hm * W + w], v * xp[hp * W + w], v * xp[h * W + wm], v * xp[h * W + wp]};
}

__global__ __launch_bounds__(256) void edge_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dx,
                                                       int H, int W, size_t total) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int HW = H * W, p = (int)(t % HW), h = p / W, w = p - h * W;
    const float* xp = x + (t - p);
    const floatx4* gp = reinterpret_cast<const floatx4*>(g) + (t - p);
    const int hm = h == 0 ? H - 1 : h - 1, hp = h == H - 1 ? 0 : h + 1, wm = w == 0 ? W - 1 : w - 1, wp = w == W - 1 ? 0 : w + 1;
    const int q0 = hm * W + w, q1 = hp * W + w, q2 = h * W + wm, q3 = h * W + wp;
    const floatx4 own = gp[p];
    // own products: d/dx[p] of x[p] x[nbr_k(p)];  as neighbour: pixel q with nbr_k(q) = p is the OPPOSITE neighbour of p
    float s = own[0] * xp[q0] + own[1] * xp[q1] + own[2] * xp[q2] + own[3] * xp[q3];
    s += gp[q1][0] * xp[q1] + gp[q0][1] * xp[q0] + gp[q3][2] * xp[q3] + gp[q2][3] * xp[q2];
    dx[t] = s;
}

}  // namespace

extern "C" int kmu_dagem_edges_fwd(const float* x, float* edge, int B, int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && edge, "dagem_edges_fwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dagem_edges_fwd: bad dims");
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(edge_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, edge, H, W, total);
    return kmu::launch_status("dagem_edges_fwd");
}

extern "C" int kmu_dagem_edges_bwd(const float* x, const float* d_edge, float* dx, int B, int C, int H, int W, kmu_stream_t stream) {
    KMU_REQUIRE(x && d_edge && dx, "dagem_edges_bwd: null pointer");
    KMU_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dagem_edges_bwd: bad dims");
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(edge_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, d_edge, dx, H, W, total);
    return kmu::launch_status("dagem_edges_bwd");
}
