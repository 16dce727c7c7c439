// Glue of the VGG16 classification head (reference: model/vgg16.py:42-61 classifier stacks, :96-101 flatten,
// :117-130 loss).  The three Linear layers run on the implicit-GEMM kernel as 1x1 convolutions (conv_igemm.hip /
// conv_wgrad.hip); what is left is HBM-bound element-wise work on (N, <= 25,088) tensors:
//   nhwc_flatten        the backbone's NHWC map <-> the reference's `x.view(N, -1)` of an NCHW tensor (index c*H*W + h*W + w), so the
//                       first Linear's weight keeps the reference's column order (weights files interchange) without a permuted copy
//   relu_dropout        nn.ReLU + nn.Dropout(p) of vgg16.py:45-46,48-49 in one pass, mask from a counter hash (object_detection_torch2_amd/
//                       synth.py: the oracle regenerates it; torch's Philox stream is not reproduced -- which units drop is unpinned
//                       by nature), scaled by 1 / (1 - p); the backward pass recomputes the mask
//   log_softmax_dot     sign * mean_n sum_c t[n][c] * log_softmax(z[n])[c] and its gradient: sign = +1 is the reference's literal
//                       VGG16.loss (vgg16.py:127-129, which lacks the minus sign of a negative log-likelihood), sign = -1 the NLL
#include "common.h"

namespace {

__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// synth.hash_u32(seed, stream, ., offset = i) for i < 2^32, with key = (seed * 0x9E3779B1 + stream * 0x85EBCA77 + 0x165667B1) mod 2^32
__device__ __forceinline__ unsigned hash_u32(unsigned key, unsigned i) { return mix32(mix32(i ^ key) + key); }

__global__ void nhwc_flatten_kernel(const float* __restrict__ src, float* __restrict__ dst, int HW, int C, int64_t total, int inverse) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // index in the NHWC tensor
    if (i >= total) return;
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int p = (int)(r % HW);
    const int64_t n = r / HW;
    const int64_t j = (n * C + c) * HW + p;                                 // index in the NCHW-flattened tensor
    if (inverse) dst[i] = src[j];
    else dst[j] = src[i];
}

__global__ void relu_dropout_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int64_t n, unsigned thr,
                                    float scale, unsigned key, int training) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const bool keep = !training || (hash_u32(key, (unsigned)i) >> 8) >= thr;
    const float m = (v > 0.f && keep) ? (training ? scale : 1.f) : 0.f;
    out[i] = dy ? dy[i] * m : v * m;                                         // backward: dx = dy * mask; forward: y = relu(x) * mask
}

// one wave per row
__global__ __launch_bounds__(64) void log_softmax_dot_kernel(const float* __restrict__ z, const float* __restrict__ t, int C, float* __restrict__ rowval,
                                                             const float* __restrict__ gscale, float* __restrict__ dz) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* zr = z + (size_t)n * C;
    const float* tr = t + (size_t)n * C;
    float mx = -__builtin_inff();
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, zr[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.f, st = 0.f, dot = 0.f;
    for (int c = lane; c < C; c += 64) { se += expf(zr[c] - mx); st += tr[c]; dot += tr[c] * (zr[c] - mx); }
    se = wave_sum(se); st = wave_sum(st); dot = wave_sum(dot);
    const float lse = logf(se);
    if (dz == nullptr) {
        if (lane == 0) rowval[n] = dot - st * lse;                          // sum_c t_c * (z_c - max - log sum exp)
    } else {
        const float g = gscale[0];
        for (int c = lane; c < C; c += 64) dz[(size_t)n * C + c] = g * (tr[c] - st * expf(zr[c] - mx - lse));
    }
}

__global__ void mean_scale_kernel(const float* __restrict__ rowval, int N, float sign, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += rowval[n];                         // fixed order
        loss[0] = sign * (s / (float)N);
    }
}

__global__ void grad_scale_kernel(const float* __restrict__ grad_out, float f, float* __restrict__ g) {
    if (threadIdx.x == 0 && blockIdx.x == 0) g[0] = grad_out[0] * f;
}

}  // namespace

extern "C" {

int ssdhip_nhwc_flatten(const float* src, float* dst, int N, int H, int W, int C, int inverse, void* stream) {
    SSDHIP_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && C > 0, "nhwc_flatten: bad arguments");
    const int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(nhwc_flatten_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, H * W, C, total, inverse);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_relu_dropout(const float* x, const float* dy, float* out, int64_t n, float p, uint32_t key, int training, void* stream) {
    SSDHIP_REQUIRE(x && out && n > 0 && n < (1ll << 32) && p >= 0.f && p < 1.f, "relu_dropout: bad arguments (0 <= p < 1, n < 2^32)");
    const unsigned thr = (unsigned)((double)p * 16777216.0);               // keep where the 24-bit uniform >= p
    hipLaunchKernelGGL(relu_dropout_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, out, n, thr, 1.f / (1.f - p), key,
                       training);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_log_softmax_dot_fwd(const float* logits, const float* targets, int N, int C, float sign, float* rowval, float* loss, void* stream) {
    SSDHIP_REQUIRE(logits && targets && rowval && loss && N > 0 && C > 0, "log_softmax_dot_fwd: bad arguments");
    hipLaunchKernelGGL(log_softmax_dot_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, logits, targets, C, rowval, nullptr, nullptr);
    hipLaunchKernelGGL(mean_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rowval, N, sign, loss);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_log_softmax_dot_bwd(const float* logits, const float* targets, int N, int C, float sign, const float* grad_out, float* scratch1,
                               float* dlogits, void* stream) {
    SSDHIP_REQUIRE(logits && targets && grad_out && scratch1 && dlogits && N > 0 && C > 0, "log_softmax_dot_bwd: bad arguments");
    hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, grad_out, sign / (float)N, scratch1);
    hipLaunchKernelGGL(log_softmax_dot_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, logits, targets, C, nullptr, scratch1, dlogits);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
