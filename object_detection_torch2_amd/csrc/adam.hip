// Fused multi-tensor Adam for the 52 trainable tensors (train.py:97,122: torch.optim.Adam(lr, weight_decay), i.e.
// L2 decay added to the gradient, bias-corrected moments, no amsgrad).  HBM-bound: 16 B/lane loads of p, g, m, v and
// stores of p, m, v = 28 B per parameter.  Step count and bias corrections live in device memory (adam_tick) so the
// whole train step can be replayed from a hipGraph without changing kernel arguments.
#include "common.h"

namespace {

// hyper: [0] lr [1] beta1 [2] beta2 [3] eps [4] weight_decay [5] 1-beta1^t [6] sqrt(1-beta2^t) [7] grad_scale
__global__ void adam_tick_kernel(int* step, float* hyper) {
    const int t = step[0] + 1;
    step[0] = t;
    hyper[5] = (float)(1.0 - pow((double)hyper[1], (double)t));
    hyper[6] = (float)sqrt(1.0 - pow((double)hyper[2], (double)t));
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const float* __restrict__ h) {
    g = g * h[7] + h[4] * p;                 // grad = grad + weight_decay * param
    m = m + (g - m) * (1.0f - h[1]);         // exp_avg.lerp_(grad, 1 - beta1)
    v = v * h[2] + (1.0f - h[2]) * g * g;    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / h[6] + h[3];
    p = p - (h[0] / h[5]) * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(const void* const* __restrict__ table, const int64_t* __restrict__ sizes,
                                                   const int32_t* __restrict__ blockmap, int chunk,
                                                   const float* __restrict__ hyper) {
    const int tensor = blockmap[2 * blockIdx.x], ch = blockmap[2 * blockIdx.x + 1];
    float* p = (float*)table[4 * tensor];
    const float* g = (const float*)table[4 * tensor + 1];
    float* m = (float*)table[4 * tensor + 2];
    float* v = (float*)table[4 * tensor + 3];
    const int64_t n = sizes[tensor];
    const int64_t lo = (int64_t)ch * chunk, hi = min(n, lo + chunk);
    float h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = hyper[i];
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && (lo % 4 == 0);
    if (vec) {
        const int64_t hi4 = lo + ((hi - lo) / 4) * 4;
        for (int64_t i = lo + threadIdx.x * 4; i < hi4; i += 256 * 4) {
            float4 pp = *reinterpret_cast<float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            adam1(pp.x, gg.x, mm.x, vv.x, h); adam1(pp.y, gg.y, mm.y, vv.y, h);
            adam1(pp.z, gg.z, mm.z, vv.z, h); adam1(pp.w, gg.w, mm.w, vv.w, h);
            *reinterpret_cast<float4*>(p + i) = pp; *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv;
        }
        for (int64_t i = hi4 + threadIdx.x; i < hi; i += 256) adam1(p[i], g[i], m[i], v[i], h);
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) adam1(p[i], g[i], m[i], v[i], h);
    }
}

}  // namespace

extern "C" {

int ssdhip_adam_tick(int32_t* step, float* hyper, void* stream) {
    SSDHIP_REQUIRE(step && hyper, "adam_tick: null pointer");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, hyper);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_adam_multi_tensor(const void* const* table, const int64_t* sizes, const int32_t* blockmap, int n_blocks, int chunk,
                             const float* hyper, void* stream) {
    SSDHIP_REQUIRE(table && sizes && blockmap && hyper && n_blocks > 0 && chunk > 0 && chunk % 4 == 0, "adam: bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, table, sizes, blockmap, chunk, hyper);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
