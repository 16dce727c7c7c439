// HBM-bound layer glue around the convolutions (all NHWC fp32, 16 B per lane, channel-contiguous => coalesced):
//   normalize_kernel        VGG16.normalize (model/vgg16.py:103-115) fused with NCHW -> NHWC4
//   bn_stats_stage1/2       nn.BatchNorm2d(training=True) statistics from the conv epilogue's per-block partials
//                           (model/vgg16.py:34, model/ssd.py:62): fp64 tree, fixed order (bit-reproducible)
//   bn_relu_pool_kernel     BN-apply + ReLU + MaxPool2d(2,2,pad) in one pass (model/vgg16.py:28-35)
//   bn_relu_bwd_*           backward of BN(train)+ReLU for the trainable tail (autograd of model/ssd.py:61-63)
#include "common.h"

namespace {

__global__ void normalize_kernel(const float* __restrict__ x, float4* __restrict__ y, int HW, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t n = i / HW;
    int64_t p = i - n * HW;
    const float* b = x + n * 3 * HW + p;
    float4 v;
    // x.sub(mean).div(std): true division, constants are the fp32 roundings of the python literals (vgg16.py:19-20)
    v.x = (b[0] - 0.485f) / 0.229f;
    v.y = (b[HW] - 0.456f) / 0.224f;
    v.z = (b[2 * (int64_t)HW] - 0.406f) / 0.225f;
    v.w = 0.f;
    y[i] = v;
}

// stage 1: partial (MB, C, 2) fp32 -> ws (SPLIT, C, 2) fp64.  block = 32 channels x 8 row-lanes
__global__ __launch_bounds__(256) void bn_stats_stage1(const float* __restrict__ partial, int MB, int C, int split,
                                                        double* __restrict__ ws) {
    __shared__ double red[8][32][2];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int sp = blockIdx.y;
    const int per = (MB + split - 1) / split;
    const int lo = sp * per, hi = min(MB, lo + per);
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        for (int r = lo + rl; r < hi; r += 8) {
            float2 v = *reinterpret_cast<const float2*>(partial + ((size_t)r * C + c) * 2);
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
    }
    red[rl][cl][0] = s1; red[rl][cl][1] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
        ws[((size_t)sp * C + c) * 2] = s1;
        ws[((size_t)sp * C + c) * 2 + 1] = s2;
    }
}

// scale/shift/running-stat update of one channel from its fp64 sums (nn.BatchNorm2d training semantics)
__device__ __forceinline__ void bn_finalize_channel(double s1, double s2, int c, double count, const float* gamma, const float* beta,
                                                    float* running_mean, float* running_var, float momentum, float eps, float* scale,
                                                    float* shift, float* save_mean, float* save_invstd) {
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;   // biased variance used for normalisation
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma[c], b = beta[c];
    scale[c] = (float)((double)g * invstd);
    shift[c] = (float)((double)b - mean * (double)g * invstd);
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = (float)invstd;
    if (running_mean) {
        // nn.BatchNorm2d: running = (1 - momentum) * running + momentum * stat, unbiased variance for running_var
        const double unbiased = var * count / (count - 1.0);
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

// one launch for small partial tables: block = 32 channels x 32 row-lanes (1,024 threads), loops over all MB rows, finalizes in place
__global__ __launch_bounds__(1024) void bn_stats_fused(const float* __restrict__ partial, int MB, int C, double count,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                                        float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                        float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    __shared__ double red[32][32][2];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int r = rl; r < MB; r += 32) {
            float2 v = *reinterpret_cast<const float2*>(partial + ((size_t)r * C + c) * 2);
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
    red[rl][cl][0] = s1; red[rl][cl][1] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < 32; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
        bn_finalize_channel(s1, s2, c, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd);
    }
}

// stage 2: block = 32 channels x 8 split-lanes (the former one-thread-per-channel loop walked up to 64 dependent fp64 loads: 17 us on the
// critical path of every backbone layer); fixed summation order: lane k takes splits k, k + 8, ..., lanes are added 0..7
__global__ __launch_bounds__(256) void bn_stats_stage2(const double* __restrict__ ws, int split, int C, double count, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                        float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    __shared__ double red[8][32][2];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int s = rl; s < split; s += 8) {
            const double2 v = *reinterpret_cast<const double2*>(ws + ((size_t)s * C + c) * 2);
            s1 += v.x; s2 += v.y;
        }
    red[rl][cl][0] = s1; red[rl][cl][1] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
        bn_finalize_channel(s1, s2, c, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd);
    }
}

// pooled activation from the window maxima / minima the producing convolution stored (ssdhip_wino_fused_conv_pool):
// max over a window of relu(z * scale + shift) = relu(bn(max z)) where scale >= 0, relu(bn(min z)) where scale < 0 -- the same fma and
// max on the same element the pool pass would have picked, so the result is the pool pass's bit for bit
__global__ __launch_bounds__(256) void bn_relu_select_kernel(const float4* __restrict__ zmax, const float4* __restrict__ zmin,
                                                             const float4* __restrict__ scale, const float4* __restrict__ shift,
                                                             float4* __restrict__ y, int C4, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const float4 sc = scale[c4], sh = shift[c4], a = zmax[i], b = zmin[i];
    float4 v;
    v.x = fmaxf(fmaf(sc.x >= 0.f ? a.x : b.x, sc.x, sh.x), 0.f);
    v.y = fmaxf(fmaf(sc.y >= 0.f ? a.y : b.y, sc.y, sh.y), 0.f);
    v.z = fmaxf(fmaf(sc.z >= 0.f ? a.z : b.z, sc.z, sh.z), 0.f);
    v.w = fmaxf(fmaf(sc.w >= 0.f ? a.w : b.w, sc.w, sh.w), 0.f);
    y[i] = v;
}

__global__ void bn_fold_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                    const float* __restrict__ rv, float eps, int C, float* __restrict__ scale,
                                    float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double invstd = 1.0 / sqrt((double)rv[c] + (double)eps);
    scale[c] = (float)((double)gamma[c] * invstd);
    shift[c] = (float)((double)beta[c] - (double)rm[c] * (double)gamma[c] * invstd);
}

__device__ __forceinline__ float4 bn_relu4(float4 v, float4 sc, float4 sh) {
    v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
    v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    return v;
}

// one thread = one output pixel x 4 channels
template <bool POOL>
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const float4* __restrict__ x, const float4* __restrict__ scale,
                                                           const float4* __restrict__ shift, float4* __restrict__ y, int H, int W,
                                                           int C4, int Ho, int Wo, int pad, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    int64_t pix = i / C4;
    const float4 sc = scale[c4], sh = shift[c4];
    if (!POOL) {
        y[i] = bn_relu4(x[i], sc, sh);
        return;
    }
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int64_t n = pix / Ho;
    // ReLU output >= 0, so 0 is a valid identity for max over windows that hold at least one real pixel
    // (F.max_pool2d pads with -inf; every 2x2/stride-2/pad<=1 window contains a real pixel)
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int h = ho * 2 - pad + dy;
        if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int w = wo * 2 - pad + dx;
            if ((unsigned)w >= (unsigned)W) continue;
            float4 v = bn_relu4(x[((n * H + h) * W + w) * C4 + c4], sc, sh);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    y[i] = m;
}

// MaxPool2d(2,2,pad) backward fused with the recomputation of its input a = relu(z*scale+shift): one thread = one 2x2 window x
// 4 channels; the gradient goes to the first maximum in (h, w) scan order (ATen's rule: a later element wins only if strictly
// greater); every input pixel lies in exactly one window, pixels past the last window (odd size, no padding) get 0.
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_kernel(const float4* __restrict__ z, const float4* __restrict__ scale,
                                                               const float4* __restrict__ shift, const float4* __restrict__ dpooled,
                                                               float4* __restrict__ da, int H, int W, int C4, int Ho, int Wo, int HoE,
                                                               int WoE, int pad, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    int64_t pix = i / C4;
    const int wo = (int)(pix % WoE);
    pix /= WoE;
    const int ho = (int)(pix % HoE);
    const int64_t n = pix / HoE;
    const bool real = ho < Ho && wo < Wo;
    const float4 sc = scale[c4], sh = shift[c4];
    const float4 g = real ? dpooled[((n * Ho + ho) * Wo + wo) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int bx = -1, by = -1, bz = -1, bw = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = ho * 2 - pad + (k >> 1), w = wo * 2 - pad + (k & 1);
        if ((unsigned)h >= (unsigned)H || (unsigned)w >= (unsigned)W) continue;
        const float4 v = bn_relu4(z[((n * H + h) * W + w) * C4 + c4], sc, sh);
        if (v.x > best.x) { best.x = v.x; bx = k; }
        if (v.y > best.y) { best.y = v.y; by = k; }
        if (v.z > best.z) { best.z = v.z; bz = k; }
        if (v.w > best.w) { best.w = v.w; bw = k; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = ho * 2 - pad + (k >> 1), w = wo * 2 - pad + (k & 1);
        if ((unsigned)h >= (unsigned)H || (unsigned)w >= (unsigned)W) continue;
        da[((n * H + h) * W + w) * C4 + c4] = make_float4(bx == k ? g.x : 0.f, by == k ? g.y : 0.f, bz == k ? g.z : 0.f, bw == k ? g.w : 0.f);
    }
}

// ---- BN(train) + ReLU backward -----------------------------------------------------------------------------------
// y = z*scale+shift, a = relu(y).  dy = da * (y > 0);  xhat = (z - mean) * invstd
// pass 1: per row-block partial sums of dy and dy*xhat, layout (MBrows, C, 2)
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const float* __restrict__ z, const float* __restrict__ da,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 int64_t M, int C, int rows_per_block, float* __restrict__ partial) {
    // block = 64 channels x 4 row-lanes; grid = (C/64 ceil, MBrows)
    __shared__ float red[4][64][2];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t lo = (int64_t)blockIdx.y * rows_per_block;
    const int64_t hi = min(M, lo + rows_per_block);
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        for (int64_t r = lo + rl; r < hi; r += 4) {
            const float zv = z[r * C + c];
            const float g = (fmaf(zv, sc, sh) > 0.f) ? da[r * C + c] : 0.f;
            s1 += g;
            s2 += g * ((zv - mu) * is);
        }
    }
    red[rl][cl][0] = s1; red[rl][cl][1] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
        s1 += red[1][cl][0] + red[2][cl][0] + red[3][cl][0];
        s2 += red[1][cl][1] + red[2][cl][1] + red[3][cl][1];
        partial[((size_t)blockIdx.y * C + c) * 2] = s1;
        partial[((size_t)blockIdx.y * C + c) * 2 + 1] = s2;
    }
}

__global__ void bn_relu_bwd_stage2(const double* __restrict__ ws, int split, int C, double count,
                                   float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < split; ++s) { s1 += ws[((size_t)s * C + c) * 2]; s2 += ws[((size_t)s * C + c) * 2 + 1]; }
    if (dbeta) dbeta[c] = (float)s1;           // d beta  = sum(dy)
    if (dgamma) dgamma[c] = (float)s2;         // d gamma = sum(dy * xhat)
    coef[2 * c] = (float)(s1 / count);
    coef[2 * c + 1] = (float)(s2 / count);
}

// pass 2: dz = scale * (dy - mean(dy) - xhat * mean(dy*xhat))      [scale = gamma * invstd]
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const float4* __restrict__ z, const float4* __restrict__ da,
                                                                const float4* __restrict__ scale, const float4* __restrict__ shift,
                                                                const float4* __restrict__ mean, const float4* __restrict__ invstd,
                                                                const float* __restrict__ coef, int C4, int64_t total,
                                                                float4* __restrict__ dz) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const float4 zv = z[i], g = da[i], sc = scale[c4], sh = shift[c4], mu = mean[c4], is = invstd[c4];
    const float4 k0 = *reinterpret_cast<const float4*>(coef + 8 * c4);      // (m_dy, m_dyx) x 2 channels
    const float4 k1 = *reinterpret_cast<const float4*>(coef + 8 * c4 + 4);
    float4 o;
    {
        float dy = (fmaf(zv.x, sc.x, sh.x) > 0.f) ? g.x : 0.f;
        o.x = sc.x * (dy - k0.x - (zv.x - mu.x) * is.x * k0.y);
        dy = (fmaf(zv.y, sc.y, sh.y) > 0.f) ? g.y : 0.f;
        o.y = sc.y * (dy - k0.z - (zv.y - mu.y) * is.y * k0.w);
        dy = (fmaf(zv.z, sc.z, sh.z) > 0.f) ? g.z : 0.f;
        o.z = sc.z * (dy - k1.x - (zv.z - mu.z) * is.z * k1.y);
        dy = (fmaf(zv.w, sc.w, sh.w) > 0.f) ? g.w : 0.f;
        o.w = sc.w * (dy - k1.z - (zv.w - mu.w) * is.w * k1.w);
    }
    dz[i] = o;
}

// dense (N, P, row) loss gradient -> per-head channel-padded NHWC gradient (N, HW, Cpad), pad channels zero
__device__ __forceinline__ void scatter_head_grad_body(const float* __restrict__ grad, float* __restrict__ out, int P, int row, int off,
                                                       int HW, int Cout, int Cpad, int64_t total, int64_t i) {
    if (i >= total) return;
    const int c = (int)(i % Cpad);
    int64_t pix = i / Cpad;
    const int hw = (int)(pix % HW);
    const int64_t n = pix / HW;
    out[i] = c < Cout ? grad[(n * P + off) * row + (int64_t)hw * Cout + c] : 0.f;
}
__global__ void scatter_head_grad_kernel(const float* __restrict__ grad, float* __restrict__ out, int P, int row, int off,
                                         int HW, int Cout, int Cpad, int64_t total) {
    scatter_head_grad_body(grad, out, P, row, off, HW, Cout, Cpad, total, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// all heads in one launch: blockIdx.y = head
__global__ void scatter_head_grads_kernel(const float* __restrict__ grad, const ssdhip_scatter_job* __restrict__ jobs, int N, int P, int row) {
    const ssdhip_scatter_job j = jobs[blockIdx.y];
    const int64_t total = (int64_t)N * j.HW * j.Cpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        scatter_head_grad_body(grad, j.out, P, row, j.prior_offset, j.HW, j.Cout, j.Cpad, total, i);
}

int stats_split(int MB, int C) {
    int cb = cdiv(C, 32);
    int split = 512 / cb;
    if (split > 64) split = 64;
    if (split > cdiv(MB, 8)) split = cdiv(MB, 8);
    if (split < 1) split = 1;
    return split;
}

}  // namespace

extern "C" {

int ssdhip_normalize_nchw_to_nhwc4(const float* x, float* y, int N, int H, int W, void* stream) {
    SSDHIP_REQUIRE(x && y && N > 0 && H > 0 && W > 0, "normalize: bad arguments");
    int64_t total = (int64_t)N * H * W;
    hipLaunchKernelGGL(normalize_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<float4*>(y),
                       H * W, total);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_bn_finalize(const float* partial, int MB, int C, int64_t count, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, float* scale, float* shift, float* save_mean,
                       float* save_invstd, double* ws, void* stream) {
    SSDHIP_REQUIRE(partial && gamma && beta && scale && shift && ws && MB > 0 && C > 0, "bn_finalize: bad arguments");
    SSDHIP_REQUIRE(count > 1, "Expected more than 1 value per channel when training");
    SSDHIP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats go together");
    hipStream_t st = (hipStream_t)stream;
    if ((int64_t)MB * 8 <= 4096) {      // <= 512 rows per row-lane: one launch, same fp64 summation order class (fixed order)
        hipLaunchKernelGGL(bn_stats_fused, dim3(cdiv(C, 32)), dim3(1024), 0, st, partial, MB, C, (double)count, gamma, beta, running_mean,
                           running_var, momentum, eps, scale, shift, save_mean, save_invstd);
        SSDHIP_CHECK_LAUNCH();
        return SSDHIP_OK;
    }
    const int split = stats_split(MB, C);
    hipLaunchKernelGGL(bn_stats_stage1, dim3(cdiv(C, 32), split), dim3(256), 0, st, partial, MB, C, split, ws);
    SSDHIP_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_stats_stage2, dim3(cdiv(C, 32)), dim3(256), 0, st, ws, split, C, (double)count, gamma, beta, running_mean,
                       running_var, momentum, eps, scale, shift, save_mean, save_invstd);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_bn_relu_select(const float* zmax, const float* zmin, const float* scale, const float* shift, float* y, int64_t pixels, int C,
                          void* stream) {
    SSDHIP_REQUIRE(zmax && zmin && scale && shift && y && pixels > 0 && C > 0 && C % 4 == 0, "bn_relu_select: bad arguments (C % 4 == 0)");
    const int64_t total = pixels * (C / 4);
    hipLaunchKernelGGL(bn_relu_select_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(zmax),
                       reinterpret_cast<const float4*>(zmin), reinterpret_cast<const float4*>(scale), reinterpret_cast<const float4*>(shift),
                       reinterpret_cast<float4*>(y), C / 4, total);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                        float* scale, float* shift, void* stream) {
    SSDHIP_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_fold_eval: bad arguments");
    hipLaunchKernelGGL(bn_fold_eval_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var,
                       eps, C, scale, shift);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_bn_relu_pool_bwd(const float* z, const float* scale, const float* shift, const float* dpooled, float* da, int N, int H, int W,
                            int C, int pool_pad, void* stream) {
    SSDHIP_REQUIRE(z && scale && shift && dpooled && da && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0,
                   "bn_relu_pool_bwd: bad arguments (C % 4 == 0)");
    SSDHIP_REQUIRE(pool_pad == 0 || pool_pad == 1, "bn_relu_pool_bwd: pool padding 0 or 1");
    const int C4 = C / 4;
    const int Ho = (H + 2 * pool_pad - 2) / 2 + 1, Wo = (W + 2 * pool_pad - 2) / 2 + 1;
    const int HoE = (H + pool_pad + 1) / 2, WoE = (W + pool_pad + 1) / 2;        // windows needed to touch every input pixel
    int64_t total = (int64_t)N * HoE * WoE * C4;
    hipLaunchKernelGGL(bn_relu_pool_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(z),
                       reinterpret_cast<const float4*>(scale), reinterpret_cast<const float4*>(shift),
                       reinterpret_cast<const float4*>(dpooled), reinterpret_cast<float4*>(da), H, W, C4, Ho, Wo, HoE, WoE, pool_pad, total);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_bn_relu_pool(const float* x, const float* scale, const float* shift, float* y, int N, int H, int W, int C, int pool,
                        int pool_pad, void* stream) {
    SSDHIP_REQUIRE(x && scale && shift && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bn_relu_pool: bad arguments (C % 4 == 0)");
    SSDHIP_REQUIRE(pool_pad == 0 || pool_pad == 1, "bn_relu_pool: pool padding 0 or 1");
    hipStream_t st = (hipStream_t)stream;
    const int C4 = C / 4;
    if (pool) {
        const int Ho = (H + 2 * pool_pad - 2) / 2 + 1, Wo = (W + 2 * pool_pad - 2) / 2 + 1;
        int64_t total = (int64_t)N * Ho * Wo * C4;
        hipLaunchKernelGGL(bn_relu_pool_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const float4*>(scale), reinterpret_cast<const float4*>(shift), reinterpret_cast<float4*>(y),
                           H, W, C4, Ho, Wo, pool_pad, total);
    } else {
        int64_t total = (int64_t)N * H * W * C4;
        hipLaunchKernelGGL(bn_relu_pool_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const float4*>(scale), reinterpret_cast<const float4*>(shift), reinterpret_cast<float4*>(y),
                           H, W, C4, H, W, 0, total);
    }
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_bn_relu_bwd_reduce(const float* z, const float* da, const float* scale, const float* shift, const float* save_mean,
                              const float* save_invstd, int64_t M, int C, float* partial, int MB_rows, void* stream) {
    SSDHIP_REQUIRE(z && da && scale && shift && save_mean && save_invstd && partial && M > 0 && C > 0 && MB_rows > 0,
                   "bn_relu_bwd_reduce: bad arguments");
    const int rows_per_block = (int)((M + MB_rows - 1) / MB_rows);
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(cdiv(C, 64), MB_rows), dim3(256), 0, (hipStream_t)stream, z, da, scale, shift,
                       save_mean, save_invstd, M, C, rows_per_block, partial);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_bn_relu_bwd_finalize(const float* partial, int MB, int C, int64_t M, float* dgamma, float* dbeta, float* coef, double* ws,
                                void* stream) {
    SSDHIP_REQUIRE(partial && coef && ws && MB > 0 && C > 0 && M > 0, "bn_relu_bwd_finalize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int split = stats_split(MB, C);
    hipLaunchKernelGGL(bn_stats_stage1, dim3(cdiv(C, 32), split), dim3(256), 0, st, partial, MB, C, split, ws);
    SSDHIP_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_relu_bwd_stage2, dim3(cdiv(C, 64)), dim3(64), 0, st, ws, split, C, (double)M, dgamma, dbeta, coef);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_bn_relu_bwd_apply(const float* z, const float* da, const float* scale, const float* shift, const float* save_mean,
                             const float* save_invstd, const float* coef, int64_t M, int C, float* dz, void* stream) {
    SSDHIP_REQUIRE(z && da && scale && shift && save_mean && save_invstd && coef && dz && M > 0 && C > 0 && C % 4 == 0,
                   "bn_relu_bwd_apply: bad arguments");
    int64_t total = M * (C / 4);
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(z), reinterpret_cast<const float4*>(da), reinterpret_cast<const float4*>(scale),
                       reinterpret_cast<const float4*>(shift), reinterpret_cast<const float4*>(save_mean),
                       reinterpret_cast<const float4*>(save_invstd), coef, C / 4, total, reinterpret_cast<float4*>(dz));
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_scatter_head_grad(const float* grad, float* out, int N, int P, int row, int prior_offset, int HW, int Cout, int Cpad,
                             void* stream) {
    SSDHIP_REQUIRE(grad && out && N > 0 && P > 0 && HW > 0 && Cout > 0 && Cpad >= Cout && Cout % row == 0,
                   "scatter_head_grad: bad arguments");
    int64_t total = (int64_t)N * HW * Cpad;
    hipLaunchKernelGGL(scatter_head_grad_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, grad, out, P, row,
                       prior_offset, HW, Cout, Cpad, total);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_scatter_head_grads(const float* grad, const ssdhip_scatter_job* jobs_device, int n_jobs, int N, int P, int row, int64_t max_elements,
                              void* stream) {
    SSDHIP_REQUIRE(grad && jobs_device && n_jobs > 0 && n_jobs <= 65535 && N > 0 && P > 0 && row > 0 && max_elements > 0,
                   "scatter_head_grads: bad arguments");
    int blocks = (int)((max_elements + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scatter_head_grads_kernel, dim3(blocks, n_jobs), dim3(256), 0, (hipStream_t)stream, grad, jobs_device, N, P, row);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
