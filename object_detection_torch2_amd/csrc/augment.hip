// Batch augmentation on the device (reference: src/augmentation/random.py:6-42 applied per sample by
// dataset.py:43-52 inside DataLoader workers, train.py:32-36): colour jitter (four ops in a sampled order), horizontal
// flip, uint8 -> [0,1] float (ToTensor) and up to three erased rectangles, as two HBM-bound passes over the batch.
// The host only samples the per-image parameters (augmentation.py); pixels are touched here, once.
//
// Pixel math follows torchvision's tensor path, which the reference's ColorJitter/RandomErasing subclasses inherit
// (torchvision is not importable in this image, so this row is "parity unpinned"; the oracle restates the same formulas):
//   blend(a, b, f) = clamp(f a + (1 - f) b, 0, 1);  gray = 0.2989 r + 0.587 g + 0.114 b
//   brightness: blend(x, 0, f)   contrast: blend(x, mean(gray(x)), f)   saturation: blend(x, gray(x), f)
//   hue: rgb -> hsv, h = (h + f) mod 1, hsv -> rgb
// Parameter row per image (AUG_ROW floats): [0] number of colour ops (0 or 4), [1..4] op ids in application order
// (0 brightness, 1 contrast, 2 saturation, 3 hue), [5..8] factor of op id 0..3, [9] flip, [10] number of erase boxes (<= 3),
// [11 + 4k ..] box k as (top, left, height, width) in pixels of the (flipped) output image.
#include "common.h"

namespace {

constexpr int AUG_ROW = 24, MEAN_BLOCKS = 64;

struct RGB { float r, g, b; };

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(RGB p) { return 0.2989f * p.r + 0.587f * p.g + 0.114f * p.b; }
__device__ __forceinline__ RGB blend(RGB a, float br, float bg, float bb, float f) {
    return {clamp01(f * a.r + (1.f - f) * br), clamp01(f * a.g + (1.f - f) * bg), clamp01(f * a.b + (1.f - f) * bb)};
}

__device__ __forceinline__ RGB shift_hue(RGB p, float hf) {
    const float maxc = fmaxf(p.r, fmaxf(p.g, p.b)), minc = fminf(p.r, fminf(p.g, p.b));
    const bool eq = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eq ? 1.f : maxc);
    const float div = eq ? 1.f : cr;
    const float rc = (maxc - p.r) / div, gc = (maxc - p.g) / div, bc = (maxc - p.b) / div;
    float h;
    if (maxc == p.r) h = bc - gc;
    else if (maxc == p.g) h = 2.f + rc - bc;
    else h = 4.f + gc - rc;
    h = fmodf(h / 6.f + 1.f, 1.f);
    h = h + hf;
    h = h - floorf(h);                                   // python-style mod 1
    const float v = maxc;
    const float h6 = h * 6.f, fl = floorf(h6), f = h6 - fl;
    int i = (int)fl % 6;
    const float pp = clamp01(v * (1.f - s)), q = clamp01(v * (1.f - s * f)), t = clamp01(v * (1.f - s * (1.f - f)));
    switch (i) {
        case 0: return {v, t, pp};
        case 1: return {q, v, pp};
        case 2: return {pp, v, t};
        case 3: return {pp, q, v};
        case 4: return {t, pp, v};
        default: return {v, pp, q};
    }
}

// colour ops [0, upto) of the image's order; `mean` is only read by a contrast op inside that range
__device__ __forceinline__ RGB colour_chain(RGB p, const float* __restrict__ prm, int upto, float mean) {
    for (int k = 0; k < upto; ++k) {
        const int op = (int)prm[1 + k] & 3;
        const float f = prm[5 + op];
        if (op == 0) p = blend(p, 0.f, 0.f, 0.f, f);
        else if (op == 1) p = blend(p, mean, mean, mean, f);
        else if (op == 2) { const float g = gray_of(p); p = blend(p, g, g, g, f); }
        else p = shift_hue(p, f);
    }
    return p;
}

__device__ __forceinline__ RGB load_u8(const uint8_t* __restrict__ src, size_t pix) {
    const uint8_t* q = src + pix * 3;
    return {q[0] / 255.f, q[1] / 255.f, q[2] / 255.f};
}

// pass 1: per-image partial sums of gray(x after the ops that precede contrast); fixed slices, fp64, deterministic
__global__ __launch_bounds__(256) void augment_mean_kernel(const uint8_t* __restrict__ src, const float* __restrict__ params, int HW,
                                                           double* __restrict__ partial) {
    const int n = blockIdx.y;
    const float* prm = params + (size_t)n * AUG_ROW;
    const int nops = min((int)prm[0], 4);
    int cpos = -1;
    for (int k = 0; k < nops; ++k)
        if (((int)prm[1 + k] & 3) == 1) cpos = k;
    double acc = 0.0;
    if (cpos >= 0) {
        const int per = (HW + MEAN_BLOCKS - 1) / MEAN_BLOCKS, lo = blockIdx.x * per, hi = min(HW, lo + per);
        for (int i = lo + threadIdx.x; i < hi; i += 256) acc += (double)gray_of(colour_chain(load_u8(src, (size_t)n * HW + i), prm, cpos, 0.f));
    }
    __shared__ double red[4];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)n * MEAN_BLOCKS + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// pass 2: everything, one thread per output pixel; writes the (N,3,H,W) fp32 tensor the model takes (train.py:118)
__global__ __launch_bounds__(256) void augment_apply_kernel(const uint8_t* __restrict__ src, const float* __restrict__ params,
                                                            const double* __restrict__ partial, int H, int W, float* __restrict__ out) {
    const int n = blockIdx.y, HW = H * W;
    const float* prm = params + (size_t)n * AUG_ROW;
    __shared__ float s_mean;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < MEAN_BLOCKS; ++i) s += partial[(size_t)n * MEAN_BLOCKS + i];
        s_mean = (float)(s / (double)HW);
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const int y = i / W, x = i - y * W;
    const bool flip = prm[9] != 0.f;
    const int xs = flip ? W - 1 - x : x;
    RGB p = colour_chain(load_u8(src, (size_t)n * HW + (size_t)y * W + xs), prm, min((int)prm[0], 4), s_mean);
    const int ne = min((int)prm[10], 3);
    for (int k = 0; k < ne; ++k) {
        const int top = (int)prm[11 + 4 * k], left = (int)prm[12 + 4 * k], h = (int)prm[13 + 4 * k], w = (int)prm[14 + 4 * k];
        if (y >= top && y < top + h && x >= left && x < left + w) p = {0.f, 0.f, 0.f};
    }
    float* o = out + (size_t)n * 3 * HW + i;
    o[0] = p.r; o[HW] = p.g; o[2 * (size_t)HW] = p.b;
}

}  // namespace

extern "C" {

int ssdhip_augment_row_floats(void) { return AUG_ROW; }
int64_t ssdhip_augment_ws_bytes(int N) { return (int64_t)N * MEAN_BLOCKS * sizeof(double); }

int ssdhip_augment_batch(const uint8_t* src_nhwc_u8, const float* params, int N, int H, int W, float* out_nchw, void* ws, void* stream) {
    SSDHIP_REQUIRE(src_nhwc_u8 && params && out_nchw && ws, "augment: null pointer");
    SSDHIP_REQUIRE(N > 0 && H > 0 && W > 0 && (int64_t)H * W < (1 << 30), "augment: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(augment_mean_kernel, dim3(MEAN_BLOCKS, N), dim3(256), 0, st, src_nhwc_u8, params, H * W, (double*)ws);
    hipLaunchKernelGGL(augment_apply_kernel, dim3(cdiv((int64_t)H * W, 256), N), dim3(256), 0, st, src_nhwc_u8, params, (const double*)ws, H, W,
                       out_nchw);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
