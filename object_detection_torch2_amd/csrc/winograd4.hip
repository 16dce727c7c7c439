// Winograd F(4x4, 3x3) path for the deep 3x3 / stride 1 / pad 1 layers of the FROZEN backbone (conv_3_2 .. conv_5_3, forward only;
// reference op: nn.Conv2d forward, model/vgg16.py:33): 36 multiplications per 4x4 output tile instead of 144, i.e. 4x fewer MFMA
// flops than the direct form and 1.78x fewer than the F(2x2, 3x3) path of winograd.hip, and 2.25x (not 4x) the activation bytes
// each way through the Winograd domain.  Same three-pass structure around the same plane GEMM (gemm_planes.hip, 36 planes):
//
//   winograd4_input_kernel    V[xi][t][c]  = (B^T d B)[xi]   d = 6x6 input patch of tile t, producer's BN scale/shift + ReLU applied
//                                            on load, zero padding after it                                     (HBM-bound)
//   gemm_planes_dma_kernel    M[xi][t][co] = sum_c V[xi][t][c] * U[xi][co][c]      36 GEMMs, one launch       (MFMA-bound)
//   winograd4_output_kernel   y tile       = A^T m A + bias, NHWC store, per-block BatchNorm partial sums      (HBM-bound)
//   winograd4_weight_kernel   U[xi][co][c] = (G g G^T)[xi]   in fp64, rounded once (frozen layers: once per weight version)
//
// Numerics.  F(4x4) in fp32 is less exact than F(2x2): the products M are larger than the outputs they cancel to, so the fp32
// accumulation error of the GEMM is amplified by the output transform (the transforms themselves contribute little: done in fp64
// they leave the error unchanged).  The interpolation points decide the amplification.  Measured on conv_4_2-shaped data against an
// fp64 convolution (rms error / rms output; max error / max |output|):
//      direct fp32 (torch CPU)            1.9e-7   2.6e-7
//      F(2x2), points 0, +-1              4.2e-7   5.0e-7
//      F(4x4), points 0, +-1, +-2         2.4e-6   8.1e-6      (the textbook choice)
//      F(4x4), points 0, +-1, 2, -1/2     1.6e-6   3.8e-6
//      F(4x4), points 0, +-3/2, +-2/3     1.1e-6   2.3e-6      <- used here (best of 495 rational point sets tried)
// i.e. within the 2e-5 per-layer bound the tests put on every convolution kernel, 2.7x the F(2x2) error.  SSD_WINOGRAD=f2 (engine)
// keeps every layer on the F(2x2) kernels.
//
// Matrices for the points (0, 3/2, -3/2, 2/3, -2/3, inf) (Cook-Toom; checked in exact rational arithmetic by
// tests/test_host_cpu.py::test_winograd4_matrices_are_exact against the copies in ops.WINOGRAD4_*):
//   B^T = [ 1    0   -97/36   0     1  0 ]     G = [   1        0       0    ]     A^T = [ 1   1     1     1     1    0 ]
//         [ 0  -2/3   -4/9   3/2    1  0 ]         [  8/65    12/65   18/65  ]           [ 0  3/2  -3/2   2/3  -2/3  0 ]
//         [ 0   2/3   -4/9  -3/2    1  0 ]         [  8/65   -12/65   18/65  ]           [ 0  9/4   9/4   4/9   4/9  0 ]
//         [ 0  -3/2   -9/4   2/3    1  0 ]         [-81/130  -27/65  -18/65  ]           [ 0 27/8 -27/8  8/27 -8/27  1 ]
//         [ 0   3/2   -9/4  -2/3    1  0 ]         [-81/130   27/65  -18/65  ]
//         [ 0    1      0  -97/36   0  1 ]         [   0        0       1    ]
// The +- point pairs make rows 1/2 and 3/4 an even part +- an odd part: 14 operations per 6-vector instead of 36.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr float kC97_36 = 97.0f / 36.0f, kC4_9 = 4.0f / 9.0f, kC9_4 = 2.25f, kC3_2 = 1.5f, kC2_3 = 2.0f / 3.0f;
constexpr float kC27_8 = 3.375f, kC8_27 = 8.0f / 27.0f;

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4mul(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
// a + s * b
__device__ __forceinline__ float4 f4fma(float s, float4 b, float4 a) {
    return make_float4(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y), fmaf(s, b.z, a.z), fmaf(s, b.w, a.w));
}

// v = B^T d for one 6-vector (in place)
__device__ __forceinline__ void bt6(float4& d0, float4& d1, float4& d2, float4& d3, float4& d4, float4& d5) {
    const float4 v0 = f4fma(-kC97_36, d2, f4add(d0, d4));
    const float4 e1 = f4fma(-kC4_9, d2, d4), o1 = f4fma(-kC2_3, d1, f4mul(kC3_2, d3));
    const float4 e2 = f4fma(-kC9_4, d2, d4), o2 = f4fma(-kC3_2, d1, f4mul(kC2_3, d3));
    const float4 v5 = f4fma(-kC97_36, d3, f4add(d1, d5));
    d0 = v0; d1 = f4add(e1, o1); d2 = f4sub(e1, o1); d3 = f4add(e2, o2); d4 = f4sub(e2, o2); d5 = v5;
}
__device__ __forceinline__ void bt6(float2& d0, float2& d1, float2& d2, float2& d3, float2& d4, float2& d5) {
#define SSD_BT6(c)                                                                                              \
    {                                                                                                           \
        const float v0 = fmaf(-kC97_36, d2.c, d0.c + d4.c);                                                     \
        const float e1 = fmaf(-kC4_9, d2.c, d4.c), o1 = fmaf(-kC2_3, d1.c, kC3_2 * d3.c);                       \
        const float e2 = fmaf(-kC9_4, d2.c, d4.c), o2 = fmaf(-kC3_2, d1.c, kC2_3 * d3.c);                       \
        const float v5 = fmaf(-kC97_36, d3.c, d1.c + d5.c);                                                     \
        d0.c = v0; d1.c = e1 + o1; d2.c = e1 - o1; d3.c = e2 + o2; d4.c = e2 - o2; d5.c = v5;                   \
    }
    SSD_BT6(x) SSD_BT6(y)
#undef SSD_BT6
}
__device__ __forceinline__ float4 act4(float4 v, float4 sc, float4 sh) {
    return make_float4(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f), fmaxf(fmaf(v.z, sc.z, sh.z), 0.f),
                       fmaxf(fmaf(v.w, sc.w, sh.w), 0.f));
}
__device__ __forceinline__ float2 act4(float2 v, float2 sc, float2 sh) {
    return make_float2(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f));
}
template <typename VT> __device__ __forceinline__ VT vzero();
template <> __device__ __forceinline__ float4 vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float2 vzero<float2>() { return make_float2(0.f, 0.f); }

// one thread = one tile x 4 (VT = float4) or 2 (float2) channels; CV = C / channels per thread.  The float2 form needs ~100 registers
// instead of ~175 (36 patch values live at once), i.e. 4-5 waves per SIMD instead of 2: with the producer's BN.ReLU fused the float4 form
// ran at 3.3-3.5 TB/s on the 38 / 75 px maps, the form without it (168 registers, 3 waves) at 5.3 TB/s.
template <bool XFORM, typename VT>
__global__ __launch_bounds__(256) void winograd4_input_kernel(const VT* __restrict__ x, const VT* __restrict__ scale,
                                                              const VT* __restrict__ shift, VT* __restrict__ V, int H, int W,
                                                              int CV, int TH, int TW, int T, int Tp) {
    // consecutive blocks land on different XCDs (round-robin), each with its own L2, while neighbouring tiles share a third of their
    // 6x6 patches: give every XCD a contiguous eighth of the tiles, so that the 2.25x patch overlap is served by ONE L2 instead of being
    // fetched through the fabric by each of them (conv_4_2: 3.7 -> see scripts/w4_bench.py)
    int bid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int64_t idx = (int64_t)bid * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * CV) return;
    const int cv = (int)(idx % CV);
    const int t = (int)(idx / CV);
    const int tw = t % TW;
    const int q = t / TW;
    const int th = q % TH;
    const int n = q / TH;
    VT sc, sh;
    if (XFORM) { sc = scale[cv]; sh = shift[cv]; }
    // all 36 loads are issued unconditionally (addresses clamped into the image) and back to back; activation and the zero padding
    // AFTER it are applied to the registers afterwards.  (Loads inside `if (inside)` blocks were compiled as 36 exec-masked
    // load / wait / activate sequences: one memory latency each, 3.5 TB/s.)
    VT d[6][6];
    const VT* xn = x + (size_t)n * H * W * CV + cv;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int h = min(max(4 * th - 1 + i, 0), H - 1);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int w = min(max(4 * tw - 1 + j, 0), W - 1);
            d[i][j] = xn[(size_t)(h * W + w) * CV];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const bool hin = (unsigned)(4 * th - 1 + i) < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const bool in = hin && (unsigned)(4 * tw - 1 + j) < (unsigned)W;
            VT v = d[i][j];
            if (XFORM) v = act4(v, sc, sh);
            d[i][j] = in ? v : vzero<VT>();
        }
    }
    // d B: every patch row as a 6-vector; then B^T (.) on every column
#pragma unroll
    for (int i = 0; i < 6; ++i) bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    const size_t plane = (size_t)Tp * CV;
    VT* o = V + (size_t)t * CV + cv;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
        for (int i = 0; i < 6; ++i) o[(size_t)(6 * i + j) * plane] = d[i][j];
    }
}

// y = A^T m for one 6-vector
__device__ __forceinline__ void at6(float4 m0, float4 m1, float4 m2, float4 m3, float4 m4, float4 m5, float4& y0, float4& y1, float4& y2,
                                    float4& y3) {
    const float4 s12 = f4add(m1, m2), d12 = f4sub(m1, m2), s34 = f4add(m3, m4), d34 = f4sub(m3, m4);
    y0 = f4add(f4add(m0, s12), s34);
    y1 = f4fma(kC3_2, d12, f4mul(kC2_3, d34));
    y2 = f4fma(kC9_4, s12, f4mul(kC4_9, s34));
    y3 = f4add(f4fma(kC27_8, d12, f4mul(kC8_27, d34)), m5);
}

// block = (256 / C4o) tile lanes x C4o channel quads; each block walks `tb` tiles and emits one row of BN partial sums
__global__ __launch_bounds__(256) void winograd4_output_kernel(const float4* __restrict__ Mo, const float4* __restrict__ bias,
                                                               float4* __restrict__ y, float* __restrict__ stats, int H, int W, int C4o,
                                                               int TH, int TW, int T, int Tp, int tb, int accumulate) {
    __shared__ float red[256][8];
    const int lanesT = 256 / C4o;
    const int c4 = threadIdx.x % C4o, tl = threadIdx.x / C4o;
    const float4 bv = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t plane = (size_t)Tp * C4o;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const int t_end = min(T, (int)(blockIdx.x + 1) * tb);
    for (int t = blockIdx.x * tb + tl; t < t_end; t += lanesT) {
        const int tw = t % TW;
        const int q = t / TW;
        const int th = q % TH;
        const int n = q / TH;
        const float4* mp = Mo + (size_t)t * C4o + c4;
        float4 r[4][6];                 // A^T m, column by column as the planes arrive
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = mp[(size_t)(6 * i + j) * plane];
            at6(m[0], m[1], m[2], m[3], m[4], m[5], r[0][j], r[1][j], r[2][j], r[3][j]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int h = 4 * th + a;
            float4 o[4];
            at6(r[a][0], r[a][1], r[a][2], r[a][3], r[a][4], r[a][5], o[0], o[1], o[2], o[3]);
            if (h < H) {
                const int w0 = 4 * tw;
                float4* dst = y + ((size_t)(n * H + h) * W + w0) * C4o + c4;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (w0 + b < W) {
                        float4 v = f4add(o[b], bv);
                        if (accumulate) v = f4add(dst[(size_t)b * C4o], v);      // data gradients: y += (two producers of one activation gradient)
                        dst[(size_t)b * C4o] = v;
                        s1 = f4add(s1, v);
                        s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
                    }
                }
            }
        }
    }
    if (stats != nullptr) {
        float* rr = red[threadIdx.x];
        rr[0] = s1.x; rr[1] = s1.y; rr[2] = s1.z; rr[3] = s1.w; rr[4] = s2.x; rr[5] = s2.y; rr[6] = s2.z; rr[7] = s2.w;
        __syncthreads();
        if (tl == 0) {
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = rr[k];
            for (int l = 1; l < lanesT; ++l)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += red[l * C4o + c4][k];
            float* o = stats + ((size_t)blockIdx.x * C4o * 4 + c4 * 4) * 2;      // (rows, C, 2)
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[2 * k] = acc[k]; o[2 * k + 1] = acc[4 + k]; }
        }
    }
}

// Detector-head output transform: M (36, Tp, Cp) with Cp = Cout padded to 32 -> y[n*y_batch_stride + (h*W + w)*Cout + co]
// (the (N, P, 4+C) prediction tensor: zero-copy concat of model/ssd.py:103-104), + bias.  One thread = one tile x 4 channels.
__global__ __launch_bounds__(256) void winograd4_output_head_kernel(const float4* __restrict__ Mo, const float* __restrict__ bias,
                                                                    float* __restrict__ y, long long y_batch_stride, int H, int W,
                                                                    int Cout, int Cp4, int TH, int TW, int T, int Tp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * Cp4) return;
    const int c4 = (int)(idx % Cp4);
    const int t = (int)(idx / Cp4);
    if (c4 * 4 >= Cout) return;
    const int tw = t % TW;
    const int q = t / TW;
    const int th = q % TH;
    const int n = q / TH;
    const size_t plane = (size_t)Tp * Cp4;
    const float4* mp = Mo + (size_t)t * Cp4 + c4;
    float4 r[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float4 m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = mp[(size_t)(6 * i + j) * plane];
        at6(m[0], m[1], m[2], m[3], m[4], m[5], r[0][j], r[1][j], r[2][j], r[3][j]);
    }
    float bv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bv[k] = (bias && c4 * 4 + k < Cout) ? bias[c4 * 4 + k] : 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int h = 4 * th + a;
        float4 o[4];
        at6(r[a][0], r[a][1], r[a][2], r[a][3], r[a][4], r[a][5], o[0], o[1], o[2], o[3]);
        if (h >= H) continue;
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) {
            const int w = 4 * tw + b2;
            if (w >= W) continue;
            float* dst = y + (size_t)n * y_batch_stride + (size_t)(h * W + w) * Cout + c4 * 4;
            const float ov[4] = {o[b2].x + bv[0], o[b2].y + bv[1], o[b2].z + bv[2], o[b2].w + bv[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c4 * 4 + k < Cout) dst[k] = ov[k];
        }
    }
}

// U[xi][co][ci] = (G g G^T)[xi] in fp64, rounded once
__global__ void winograd4_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int rows_pad, int KP) {
    const int64_t total = (int64_t)rows_pad * KP;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i / KP), ci = (int)(i - (int64_t)co * KP);
    double g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = (co < Cout && ci < Cin) ? (double)w[(((size_t)co * Cin + ci) * 3 + r) * 3 + s] : 0.0;
    const double G[6][3] = {{1.0, 0.0, 0.0},
                            {8.0 / 65.0, 12.0 / 65.0, 18.0 / 65.0},
                            {8.0 / 65.0, -12.0 / 65.0, 18.0 / 65.0},
                            {-81.0 / 130.0, -27.0 / 65.0, -18.0 / 65.0},
                            {-81.0 / 130.0, 27.0 / 65.0, -18.0 / 65.0},
                            {0.0, 0.0, 1.0}};
    double t[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s) t[a][s] = G[a][0] * g[0][s] + G[a][1] * g[1][s] + G[a][2] * g[2][s];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
            U[(size_t)(6 * a + b) * total + i] = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
}

// U of the ADJOINT convolution (data gradient of a 3x3 / s1 / p1 layer = the same convolution of dy with the filter flipped and its
// channel roles swapped): U[xi][ci][co] = (G g' G^T)[xi], g'[r][s] = w[co][ci][2 - r][2 - s]; rows = ci, K = co (zero beyond Cout)
__global__ void winograd4_weight_adjoint_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int rows_pad, int KP) {
    const int64_t total = (int64_t)rows_pad * KP;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i / KP), co = (int)(i - (int64_t)ci * KP);
    double g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = (co < Cout && ci < Cin) ? (double)w[(((size_t)co * Cin + ci) * 3 + (2 - r)) * 3 + (2 - s)] : 0.0;
    const double G[6][3] = {{1.0, 0.0, 0.0},
                            {8.0 / 65.0, 12.0 / 65.0, 18.0 / 65.0},
                            {8.0 / 65.0, -12.0 / 65.0, 18.0 / 65.0},
                            {-81.0 / 130.0, -27.0 / 65.0, -18.0 / 65.0},
                            {-81.0 / 130.0, 27.0 / 65.0, -18.0 / 65.0},
                            {0.0, 0.0, 1.0}};
    double t[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s) t[a][s] = G[a][0] * g[0][s] + G[a][1] * g[1][s] + G[a][2] * g[2][s];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
            U[(size_t)(6 * a + b) * total + i] = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
}

int output_tb(int T, int C4o) {
    // tiles per block of the output transform: 256 / C4o tiles are in flight per pass; small maps (19x19 at batch 32: 800 tiles) get one
    // pass per block so that the grid still covers the chip, large ones up to four
    const int lanesT = 256 / C4o;
    int passes = (int)((int64_t)T / lanesT / 1024);
    if (passes < 1) passes = 1;
    if (passes > 4) passes = 4;
    return lanesT * passes;
}

}  // namespace

extern "C" {

int ssdhip_winograd4_tiles(int N, int H, int W, int row_pad, int* T, int* Tp) {
    if (N <= 0 || H <= 0 || W <= 0 || !T || !Tp || (row_pad != 64 && row_pad != 128)) return SSDHIP_EINVAL;
    const int64_t t = (int64_t)N * ((H + 3) / 4) * ((W + 3) / 4);
    if (t * 36 >= (1ll << 31)) return SSDHIP_EINVAL;
    *T = (int)t;
    *Tp = (int)((t + row_pad - 1) / row_pad * row_pad);      // GEMM m-tiles (row_pad rows) never straddle two xi planes
    return SSDHIP_OK;
}

int ssdhip_winograd4_weight(const float* w_oihw, float* U, int Cout, int Cin, int rows_pad, int KP, void* stream) {
    SSDHIP_REQUIRE(w_oihw && U && Cout > 0 && Cin > 0 && rows_pad >= Cout && KP >= Cin && KP % 32 == 0, "winograd4_weight: bad arguments");
    const int64_t total = (int64_t)rows_pad * KP;
    hipLaunchKernelGGL(winograd4_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, U, Cout, Cin, rows_pad, KP);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_winograd4_weight_adjoint(const float* w_oihw, float* U, int Cout, int Cin, int rows_pad, int KP, void* stream) {
    SSDHIP_REQUIRE(w_oihw && U && Cout > 0 && Cin > 0 && rows_pad >= Cin && KP >= Cout && KP % 32 == 0, "winograd4_weight_adjoint: bad arguments");
    const int64_t total = (int64_t)rows_pad * KP;
    hipLaunchKernelGGL(winograd4_weight_adjoint_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, U, Cout, Cin, rows_pad, KP);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_winograd4_input(const float* x, const float* in_scale, const float* in_shift, float* V, int N, int H, int W, int C, int Tp,
                           void* stream) {
    SSDHIP_REQUIRE(x && V && C > 0 && C % 4 == 0, "winograd4_input: bad arguments (C % 4 == 0)");
    SSDHIP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "winograd4_input: in_scale and in_shift go together");
    int T, Tp128;
    SSDHIP_REQUIRE(ssdhip_winograd4_tiles(N, H, W, 128, &T, &Tp128) == SSDHIP_OK && Tp >= T, "winograd4_input: bad shape");
    SSDHIP_REQUIRE(elems_below(1ll << 31, N, H, W, C) && elems_below(1ll << 33, 36, Tp, C), "winograd4_input: tensor too large");
    const int TH = (H + 3) / 4, TW = (W + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
    int vec = 2;
#ifdef SSDHIP_PROBES
    if (const char* e = getenv("SSDHIP_W4_VEC")) vec = atoi(e) == 4 ? 4 : 2;      // A/B: float4 per thread
#endif
    const int CV = C / vec;
    const int64_t total = (int64_t)T * CV;
    dim3 grid(cdiv(total, 256)), block(256);
    if (vec == 4) {
        if (in_scale)
            hipLaunchKernelGGL((winograd4_input_kernel<true, float4>), grid, block, 0, st, reinterpret_cast<const float4*>(x),
                               reinterpret_cast<const float4*>(in_scale), reinterpret_cast<const float4*>(in_shift),
                               reinterpret_cast<float4*>(V), H, W, CV, TH, TW, T, Tp);
        else
            hipLaunchKernelGGL((winograd4_input_kernel<false, float4>), grid, block, 0, st, reinterpret_cast<const float4*>(x), nullptr, nullptr,
                               reinterpret_cast<float4*>(V), H, W, CV, TH, TW, T, Tp);
    } else {
        if (in_scale)
            hipLaunchKernelGGL((winograd4_input_kernel<true, float2>), grid, block, 0, st, reinterpret_cast<const float2*>(x),
                               reinterpret_cast<const float2*>(in_scale), reinterpret_cast<const float2*>(in_shift),
                               reinterpret_cast<float2*>(V), H, W, CV, TH, TW, T, Tp);
        else
            hipLaunchKernelGGL((winograd4_input_kernel<false, float2>), grid, block, 0, st, reinterpret_cast<const float2*>(x), nullptr, nullptr,
                               reinterpret_cast<float2*>(V), H, W, CV, TH, TW, T, Tp);
    }
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_winograd4_stat_rows(int N, int H, int W, int Cout) {
    int T, Tp;
    if (Cout <= 0 || Cout % 4 || 256 % (Cout / 4) || ssdhip_winograd4_tiles(N, H, W, 128, &T, &Tp) != SSDHIP_OK) return SSDHIP_EINVAL;
    return cdiv(T, output_tb(T, Cout / 4));
}

int ssdhip_winograd4_output(const float* Mo, const float* bias, float* y, float* stats, int accumulate, int N, int H, int W, int Cout,
                            int Tp, void* stream) {
    SSDHIP_REQUIRE(Mo && y && Cout > 0, "winograd4_output: bad arguments");
    SSDHIP_REQUIRE(Cout % 4 == 0 && 256 % (Cout / 4) == 0, "winograd4_output: Cout must be 4 * a divisor of 256 (<= 1024)");
    int T, Tp128;
    SSDHIP_REQUIRE(ssdhip_winograd4_tiles(N, H, W, 128, &T, &Tp128) == SSDHIP_OK && Tp >= T, "winograd4_output: bad shape");
    SSDHIP_REQUIRE(elems_below(1ll << 31, N, H, W, Cout), "winograd4_output: tensor too large");
    const int TH = (H + 3) / 4, TW = (W + 3) / 4, tb = output_tb(T, Cout / 4);
    hipLaunchKernelGGL(winograd4_output_kernel, dim3(cdiv(T, tb)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(Mo), reinterpret_cast<const float4*>(bias), reinterpret_cast<float4*>(y), stats,
                       H, W, Cout / 4, TH, TW, T, Tp, tb, accumulate);
    SSDHIP_CHECK_LAUNCH();
    return cdiv(T, tb);      // rows of the BN partial table
}

int ssdhip_winograd4_output_head(const float* Mo, const float* bias, float* y, int64_t y_batch_stride, int N, int H, int W, int Cout,
                                 int Cp, int Tp, void* stream) {
    SSDHIP_REQUIRE(Mo && y && Cout > 0 && Cp >= Cout && Cp % 4 == 0, "winograd4_output_head: bad arguments");
    int T, Tp128;
    SSDHIP_REQUIRE(ssdhip_winograd4_tiles(N, H, W, 128, &T, &Tp128) == SSDHIP_OK && Tp >= T, "winograd4_output_head: bad shape");
    const int TH = (H + 3) / 4, TW = (W + 3) / 4, Cp4 = Cp / 4;
    hipLaunchKernelGGL(winograd4_output_head_kernel, dim3(cdiv((int64_t)T * Cp4, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(Mo), bias, y, (long long)y_batch_stride, H, W, Cout, Cp4, TH, TW, T, Tp);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
