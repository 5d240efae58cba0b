// CSI / POD / FAR / HSS contingency counts of the reference evaluator as ONE device reduction
// (metrics.py:45-47 float2int: clip(x,0,1)*value_scale -> uint16; :105-114 _cal_frame: TP/FN/FP/TN per threshold;
//  :220-288 pools the counts over frames before forming the scores).
// One streaming pass over pred / target: per thread 3 x NT register counters, LDS integer atomics per workgroup,
// one 64-bit integer atomic per counter per workgroup.  Integer adds commute: the result is exact and deterministic.
#include "common.h"

namespace {

constexpr int MAXT = 8;

struct Thresholds {
    int v[MAXT];
    int n;
};

__device__ __forceinline__ int float2int(float x, float scale) {
    // numpy: x.clip(0,1) * scale in float32, then astype(uint16) truncates toward zero
    return (int)(fminf(fmaxf(x, 0.f), 1.f) * scale);
}

__global__ __launch_bounds__(256) void contingency_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                          unsigned long long* __restrict__ counts, size_t n, float scale,
                                                          Thresholds th) {
    __shared__ unsigned int acc[MAXT * 3];
    if (threadIdx.x < MAXT * 3) acc[threadIdx.x] = 0u;
    __syncthreads();
    unsigned int c[MAXT][3];
#pragma unroll
    for (int k = 0; k < MAXT; ++k) c[k][0] = c[k][1] = c[k][2] = 0u;
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
    const kmu::floatx4* p4 = reinterpret_cast<const kmu::floatx4*>(pred);
    const kmu::floatx4* t4 = reinterpret_cast<const kmu::floatx4*>(target);
    auto tally = [&](float pv, float tv) {
        const int pi = float2int(pv, scale), ti = float2int(tv, scale);
#pragma unroll
        for (int k = 0; k < MAXT; ++k) {
            if (k < th.n) {
                const bool pb = pi >= th.v[k], tb = ti >= th.v[k];
                c[k][0] += (pb && tb) ? 1u : 0u;    // TP (hit)
                c[k][1] += (!pb && tb) ? 1u : 0u;   // FN (miss)
                c[k][2] += (pb && !tb) ? 1u : 0u;   // FP (false alarm)
            }
        }
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const kmu::floatx4 a = p4[i], b = t4[i];
        tally(a[0], b[0]);
        tally(a[1], b[1]);
        tally(a[2], b[2]);
        tally(a[3], b[3]);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) tally(pred[i], target[i]);
#pragma unroll
    for (int k = 0; k < MAXT; ++k)
        if (k < th.n) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (c[k][j]) atomicAdd(&acc[k * 3 + j], c[k][j]);
        }
    __syncthreads();
    if (threadIdx.x < th.n * 3 && acc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)acc[threadIdx.x]);
}

}  // namespace

extern "C" int kmu_contingency_counts(const float* pred, const float* target, unsigned long long* counts, size_t n,
                                      const int* thresholds, int n_thresholds, float scale, kmu_stream_t stream) {
    KMU_REQUIRE(pred && target && counts && thresholds, "contingency_counts: null pointer");
    KMU_REQUIRE(n_thresholds > 0 && n_thresholds <= MAXT, "contingency_counts: %d thresholds (1..%d supported)", n_thresholds, MAXT);
    KMU_REQUIRE(n > 0 && n < ((size_t)1 << 40), "contingency_counts: bad element count");
    KMU_REQUIRE(((uintptr_t)pred & 15) == 0 && ((uintptr_t)target & 15) == 0, "contingency_counts: inputs must be 16-byte aligned");
    Thresholds th;
    th.n = n_thresholds;
    for (int k = 0; k < MAXT; ++k) th.v[k] = k < n_thresholds ? thresholds[k] : 0;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;   // 8 workgroups per CU: each thread keeps its counters in registers across the stride loop
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(contingency_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, target, counts, n, scale, th);
    return kmu::launch_status("contingency_counts");
}
