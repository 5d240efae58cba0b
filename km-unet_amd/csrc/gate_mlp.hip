// The squeeze-excite style gates of KM-UNet on pooled [B, I] vectors, one launch forward and one backward:
//     g = act2(W2 . act1(W1 . p + b1) + b2)
//   DirectionAttention.fc   (KM_UNetV3_SH.py:231-236)  Linear, GELU, Linear, Sigmoid
//   EnhancedViMBlock.fusion_gate (:111-117)            1x1 conv, GELU, 1x1 conv, Softmax(dim=1)   (on the pooled means)
//   ChannelAttention.fc     (:320-325)                 Linear, SiLU, Linear, Sigmoid
//   LocalContrastAttention.fc (:343-348)               Linear, ReLU, Linear, Sigmoid
// Everything (B <= 64 rows, I,H,O <= 256) fits one workgroup's LDS; the work is a few kFLOP, the point is launch
// count: stock PyTorch runs addmm, act, addmm, act forward and ~10 kernels backward per gate, 25 gates per step.
// Backward also reduces the weight gradients over the batch in-block (fixed order, deterministic).
#include "common.h"

namespace {

enum { ACT_GELU = 0, ACT_SILU = 1, ACT_RELU = 2 };
enum { OUT_SIGMOID = 0, OUT_SOFTMAX = 1 };

__device__ __forceinline__ float act1_f(float z, int kind) {
    if (kind == ACT_GELU) return 0.5f * z * (1.f + erff(z * 0.70710678118654752f));
    if (kind == ACT_SILU) return z / (1.f + __expf(-z));
    return fmaxf(z, 0.f);
}
__device__ __forceinline__ float act1_grad(float z, int kind) {
    if (kind == ACT_GELU) return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * __expf(-0.5f * z * z);
    if (kind == ACT_SILU) {
        const float s = 1.f / (1.f + __expf(-z));
        return s * (1.f + z * (1.f - s));
    }
    return z > 0.f ? 1.f : 0.f;
}

// LDS: p[B][I] | a1[B][H] | z2[B][O]
__global__ __launch_bounds__(256) void gate_mlp_fwd_kernel(const float* __restrict__ p, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, float* __restrict__ z1_out,
                                                           float* __restrict__ g_out, int B, int I, int H, int O, int act1,
                                                           int act2) {
    // gridDim.x = G gate groups with their own weights (stacked [G, ...]); row b of group g is row b*G + g of p / z1 / g
    const int G = gridDim.x, gi = blockIdx.x;
    w1 += (size_t)gi * H * I;
    w2 += (size_t)gi * O * H;
    if (b1) b1 += gi * H;
    if (b2) b2 += gi * O;
    extern __shared__ float sm[];
    float* ps = sm;
    float* a1 = ps + B * I;
    float* z2 = a1 + B * H;
    const int tid = threadIdx.x;
    for (int e = tid; e < B * I; e += 256) ps[e] = p[((size_t)(e / I) * G + gi) * I + e % I];
    __syncthreads();
    for (int e = tid; e < B * H; e += 256) {
        const int b = e / H, h = e - b * H;
        float s = b1 ? b1[h] : 0.f;
        const float* wr = w1 + (size_t)h * I;
        for (int i = 0; i < I; ++i) s += wr[i] * ps[b * I + i];
        z1_out[((size_t)b * G + gi) * H + h] = s;
        a1[e] = act1_f(s, act1);
    }
    __syncthreads();
    for (int e = tid; e < B * O; e += 256) {
        const int b = e / O, o = e - b * O;
        float s = b2 ? b2[o] : 0.f;
        const float* wr = w2 + (size_t)o * H;
        for (int h = 0; h < H; ++h) s += wr[h] * a1[b * H + h];
        z2[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < B * O; e += 256) {
        const int b = e / O;
        float g;
        if (act2 == OUT_SIGMOID) {
            g = 1.f / (1.f + __expf(-z2[e]));
        } else {
            float m = -INFINITY, s = 0.f;
            for (int o = 0; o < O; ++o) m = fmaxf(m, z2[b * O + o]);
            for (int o = 0; o < O; ++o) s += __expf(z2[b * O + o] - m);
            g = __expf(z2[e] - m) / s;
        }
        g_out[((size_t)b * G + gi) * O + (e - b * O)] = g;
    }
}

// LDS: p[B][I] | a1[B][H] | dz1[B][H] | dz2[B][O]
__global__ __launch_bounds__(256) void gate_mlp_bwd_kernel(const float* __restrict__ p, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, const float* __restrict__ z1,
                                                           const float* __restrict__ g, const float* __restrict__ dg,
                                                           float* __restrict__ dp, float* __restrict__ dw1, float* __restrict__ db1,
                                                           float* __restrict__ dw2, float* __restrict__ db2, int B, int I, int H,
                                                           int O, int act1, int act2) {
    const int G = gridDim.x, gi = blockIdx.x;           // gate groups as in the forward kernel; parameter gradients per group
    w1 += (size_t)gi * H * I;
    w2 += (size_t)gi * O * H;
    dw1 += (size_t)gi * H * I;
    dw2 += (size_t)gi * O * H;
    if (db1) db1 += gi * H;
    if (db2) db2 += gi * O;
    extern __shared__ float sm[];
    float* ps = sm;
    float* a1 = ps + B * I;
    float* dz1 = a1 + B * H;
    float* dz2 = dz1 + B * H;
    float* z1s = dz2 + B * O;                           // [B][H] this group's rows of z1
    const int tid = threadIdx.x;
    for (int e = tid; e < B * I; e += 256) ps[e] = p[((size_t)(e / I) * G + gi) * I + e % I];
    for (int e = tid; e < B * H; e += 256) {
        z1s[e] = z1[((size_t)(e / H) * G + gi) * H + e % H];
        a1[e] = act1_f(z1s[e], act1);
    }
    for (int e = tid; e < B * O; e += 256) {
        const int b = e / O;
        const size_t r = ((size_t)b * G + gi) * O;
        const float gv = g[r + (e - b * O)], dgv = dg[r + (e - b * O)];
        if (act2 == OUT_SIGMOID) {
            dz2[e] = dgv * gv * (1.f - gv);
        } else {
            float dot = 0.f;
            for (int o = 0; o < O; ++o) dot += g[r + o] * dg[r + o];
            dz2[e] = gv * (dgv - dot);
        }
    }
    __syncthreads();
    for (int e = tid; e < B * H; e += 256) {            // dz1 = (W2^T dz2) . act1'(z1)
        const int b = e / H, h = e - b * H;
        float s = 0.f;
        for (int o = 0; o < O; ++o) s += w2[(size_t)o * H + h] * dz2[b * O + o];
        dz1[e] = s * act1_grad(z1s[e], act1);
    }
    for (int e = tid; e < O * H; e += 256) {            // dW2[o][h] = sum_b dz2[b][o] a1[b][h]
        const int o = e / H, h = e - o * H;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dz2[b * O + o] * a1[b * H + h];
        dw2[e] = s;
    }
    if (db2)
        for (int o = tid; o < O; o += 256) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += dz2[b * O + o];
            db2[o] = s;
        }
    __syncthreads();
    for (int e = tid; e < H * I; e += 256) {            // dW1[h][i] = sum_b dz1[b][h] p[b][i]
        const int h = e / I, i = e - h * I;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dz1[b * H + h] * ps[b * I + i];
        dw1[e] = s;
    }
    if (db1)
        for (int h = tid; h < H; h += 256) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += dz1[b * H + h];
            db1[h] = s;
        }
    if (dp)
        for (int e = tid; e < B * I; e += 256) {        // dp[b][i] = sum_h W1[h][i] dz1[b][h]
            const int b = e / I, i = e - b * I;
            float s = 0.f;
            for (int h = 0; h < H; ++h) s += w1[(size_t)h * I + i] * dz1[b * H + h];
            dp[((size_t)b * G + gi) * I + i] = s;
        }
}

int check_dims(const char* what, int B, int I, int H, int O, int act1, int act2, size_t* lds) {
    KMU_REQUIRE(B > 0 && I > 0 && H > 0 && O > 0, "%s: empty problem", what);
    KMU_REQUIRE(act1 >= 0 && act1 <= 2 && act2 >= 0 && act2 <= 1, "%s: unknown activation code (%d, %d)", what, act1, act2);
    *lds = (size_t)B * (I + 3 * H + O) * sizeof(float);
    KMU_REQUIRE(*lds <= 144 * 1024, "%s: B=%d rows of %d+2*%d+%d floats do not fit one workgroup's LDS", what, B, I, H, O);
    return 0;
}

}  // namespace

// groups > 1: `groups` independent gates with stacked weights ([groups, H, I] ...) on p [B, groups, I] -> g [B, groups, O]
// (DirectionAttention.fc of the three direction branches, KM_UNetV3_SH.py:231-236): one workgroup per group
extern "C" int kmu_gate_mlp_fwd_g(const float* p, const float* w1, const float* b1, const float* w2, const float* b2, float* z1,
                                  float* g, int B, int I, int H, int O, int act1, int act2, int groups, kmu_stream_t stream) {
    KMU_REQUIRE(p && w1 && w2 && z1 && g && groups >= 1, "gate_mlp_fwd: null pointer / groups < 1");
    size_t lds;
    if (int rc = check_dims("gate_mlp_fwd", B, I, H, O, act1, act2, &lds)) return rc;
    KMU_MAX_LDS(gate_mlp_fwd_kernel, lds);
    hipLaunchKernelGGL(gate_mlp_fwd_kernel, dim3(groups), dim3(256), lds, (hipStream_t)stream, p, w1, b1, w2, b2, z1, g, B, I, H, O, act1,
                       act2);
    return kmu::launch_status("gate_mlp_fwd");
}
extern "C" int kmu_gate_mlp_fwd(const float* p, const float* w1, const float* b1, const float* w2, const float* b2, float* z1,
                                float* g, int B, int I, int H, int O, int act1, int act2, kmu_stream_t stream) {
    return kmu_gate_mlp_fwd_g(p, w1, b1, w2, b2, z1, g, B, I, H, O, act1, act2, 1, stream);
}

extern "C" int kmu_gate_mlp_bwd_g(const float* p, const float* w1, const float* w2, const float* z1, const float* g, const float* dg,
                                  float* dp, float* dw1, float* db1, float* dw2, float* db2, int B, int I, int H, int O, int act1,
                                  int act2, int groups, kmu_stream_t stream) {
    KMU_REQUIRE(p && w1 && w2 && z1 && g && dg && dw1 && dw2 && groups >= 1, "gate_mlp_bwd: null pointer / groups < 1");
    size_t lds;
    if (int rc = check_dims("gate_mlp_bwd", B, I, H, O, act1, act2, &lds)) return rc;
    KMU_MAX_LDS(gate_mlp_bwd_kernel, lds);
    hipLaunchKernelGGL(gate_mlp_bwd_kernel, dim3(groups), dim3(256), lds, (hipStream_t)stream, p, w1, w2, z1, g, dg, dp, dw1, db1, dw2, db2,
                       B, I, H, O, act1, act2);
    return kmu::launch_status("gate_mlp_bwd");
}
extern "C" int kmu_gate_mlp_bwd(const float* p, const float* w1, const float* w2, const float* z1, const float* g, const float* dg,
                                float* dp, float* dw1, float* db1, float* dw2, float* db2, int B, int I, int H, int O, int act1,
                                int act2, kmu_stream_t stream) {
    return kmu_gate_mlp_bwd_g(p, w1, w2, z1, g, dg, dp, dw1, db1, dw2, db2, B, I, H, O, act1, act2, 1, stream);
}
