// Winograd F(2x2, 3x3) path for the deep 3x3 / stride 1 / pad 1 layers (conv_3_2 .. conv_6_1): 16 multiplications per 2x2
// output tile instead of 36, i.e. 2.25x fewer MFMA flops than the direct implicit GEMM of conv_igemm.hip.  fp32 throughout;
// the transform matrices hold only 0, +-1 and +-1/2 (exact in fp32).  Same reference op: nn.Conv2d forward
// (model/vgg16.py:33, model/ssd.py:61).  Non-fused form, three passes around the existing GEMM kernel:
//
//   winograd_input_kernel    V[xi][t][c]   = (B^T d B)[xi]   d = 4x4 input patch of tile t, with the producer's BN scale/shift
//                                            + ReLU applied on load and zero padding after it            (HBM-bound)
//   conv_igemm_kernel (1x1)  M[xi][t][co]  = sum_c V[xi][t][c] * U[xi][co][c]        16 GEMMs, one launch (MFMA-bound)
//   winograd_output_kernel   y tile        = A^T m A + bias, NHWC store, per-block BatchNorm partial sums   (HBM-bound)
//   winograd_weight_kernel   U[xi][co][c]  = (G g G^T)[xi]   (once for frozen layers)
//
// It pays where K = Cin is large against the tile-domain traffic (16/4 = 4x the activation bytes each way): measured
// gains for Cin >= 256 on 75x75 .. 19x19 maps; conv_1_x / conv_2_x stay on the direct kernel.
#include "common.h"

namespace {

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// one thread = one tile x 4 channels
template <bool XFORM>
__global__ __launch_bounds__(256) void winograd_input_kernel(const float4* __restrict__ x, const float4* __restrict__ scale,
                                                             const float4* __restrict__ shift, float4* __restrict__ V, int H, int W,
                                                             int C4, int TH, int TW, int T, int Tp) {
    // every XCD (own L2) takes a contiguous eighth of the tiles: neighbouring tiles share half of their 4x4 patches
    int bid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q8 = nb >> 3, r8 = nb & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    }
    const int64_t idx = (int64_t)bid * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * C4) return;
    const int c4 = (int)(idx % C4);
    const int t = (int)(idx / C4);
    const int tw = t % TW;
    const int q = t / TW;
    const int th = q % TH;
    const int n = q / TH;
    float4 sc, sh;
    if (XFORM) { sc = scale[c4]; sh = shift[c4]; }
    // 16 unconditional loads (addresses clamped into the image) issued back to back, then activation and the zero padding after it
    // on the registers: loads inside `if (inside)` blocks compile to one exec-masked load / wait / activate sequence per position
    float4 d[4][4];
    const float4* xn = x + (size_t)n * H * W * C4 + c4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = min(max(2 * th - 1 + i, 0), H - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = min(max(2 * tw - 1 + j, 0), W - 1);
            d[i][j] = xn[(size_t)(h * W + w) * C4];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool hin = (unsigned)(2 * th - 1 + i) < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = hin && (unsigned)(2 * tw - 1 + j) < (unsigned)W;
            float4 v = d[i][j];
            if (XFORM) {
                v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
            }
            d[i][j] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // B^T d : rows (d0-d2, d1+d2, d2-d1, d1-d3), then the same on columns
    float4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0][j] = f4sub(d[0][j], d[2][j]);
        r[1][j] = f4add(d[1][j], d[2][j]);
        r[2][j] = f4sub(d[2][j], d[1][j]);
        r[3][j] = f4sub(d[1][j], d[3][j]);
    }
    const size_t plane = (size_t)Tp * C4;
    float4* o = V + (size_t)t * C4 + c4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[(size_t)(4 * i + 0) * plane] = f4sub(r[i][0], r[i][2]);
        o[(size_t)(4 * i + 1) * plane] = f4add(r[i][1], r[i][2]);
        o[(size_t)(4 * i + 2) * plane] = f4sub(r[i][2], r[i][1]);
        o[(size_t)(4 * i + 3) * plane] = f4sub(r[i][1], r[i][3]);
    }
}

// block = (256 / C4o) tile lanes x C4o channel quads; each block walks TB tiles and emits one row of BN partial sums
constexpr int WINO_TB = 8;     // tiles per block: 19x19 maps at batch 32 have only 3,200 tiles, so small blocks keep all 256 CUs busy

__global__ __launch_bounds__(256) void winograd_output_kernel(const float4* __restrict__ Mo, const float4* __restrict__ bias,
                                                              float4* __restrict__ y, float* __restrict__ stats, int H, int W, int C4o,
                                                              int TH, int TW, int T, int Tp) {
    __shared__ float red[256][8];
    const int lanesT = 256 / C4o;
    const int c4 = threadIdx.x % C4o, tl = threadIdx.x / C4o;
    const float4 bv = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t plane = (size_t)Tp * C4o;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const int t_end = min(T, (int)(blockIdx.x + 1) * WINO_TB);
    for (int t = blockIdx.x * WINO_TB + tl; t < t_end; t += lanesT) {
        const int tw = t % TW;
        const int q = t / TW;
        const int th = q % TH;
        const int n = q / TH;
        const float4* mp = Mo + (size_t)t * C4o + c4;
        float4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = mp[(size_t)(4 * i + j) * plane];
        // A^T m : rows (m0+m1+m2, m1-m2-m3), then the same on columns
        float4 r[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[0][j] = f4add(f4add(m[0][j], m[1][j]), m[2][j]);
            r[1][j] = f4sub(f4sub(m[1][j], m[2][j]), m[3][j]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int h = 2 * th + a;
            float4 o0 = f4add(f4add(f4add(r[a][0], r[a][1]), r[a][2]), bv);
            float4 o1 = f4add(f4sub(f4sub(r[a][1], r[a][2]), r[a][3]), bv);
            if (h < H) {
                const int w0 = 2 * tw;
                float4* dst = y + ((size_t)(n * H + h) * W + w0) * C4o + c4;
                dst[0] = o0;
                s1 = f4add(s1, o0);
                s2.x += o0.x * o0.x; s2.y += o0.y * o0.y; s2.z += o0.z * o0.z; s2.w += o0.w * o0.w;
                if (w0 + 1 < W) {
                    dst[C4o] = o1;
                    s1 = f4add(s1, o1);
                    s2.x += o1.x * o1.x; s2.y += o1.y * o1.y; s2.z += o1.z * o1.z; s2.w += o1.w * o1.w;
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
            float* o = stats + ((size_t)blockIdx.x * C4o * 4 + c4 * 4) * 2;      // (MB, C, 2)
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[2 * k] = acc[k]; o[2 * k + 1] = acc[4 + k]; }
        }
    }
}

// Detector-head output transform: M (16, Tp, Cp) with Cp = Cout padded to 32 -> y[n*y_batch_stride + (h*W + w)*Cout + co]
// (the (N, P, 4+C) prediction tensor: zero-copy concat of model/ssd.py:103-104), + bias.  One thread = one tile x 4 channels.
__global__ __launch_bounds__(256) void winograd_output_head_kernel(const float4* __restrict__ Mo, const float* __restrict__ bias,
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
    float4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = mp[(size_t)(4 * i + j) * plane];
    float4 r[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0][j] = f4add(f4add(m[0][j], m[1][j]), m[2][j]);
        r[1][j] = f4sub(f4sub(m[1][j], m[2][j]), m[3][j]);
    }
    float bv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bv[k] = (bias && c4 * 4 + k < Cout) ? bias[c4 * 4 + k] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int h = 2 * th + a;
        if (h >= H) continue;
        const float4 o0 = f4add(f4add(r[a][0], r[a][1]), r[a][2]);
        const float4 o1 = f4sub(f4sub(r[a][1], r[a][2]), r[a][3]);
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
            const int w = 2 * tw + b2;
            if (w >= W) continue;
            const float4 o = b2 == 0 ? o0 : o1;
            float* dst = y + (size_t)n * y_batch_stride + (size_t)(h * W + w) * Cout + c4 * 4;
            const float ov[4] = {o.x + bv[0], o.y + bv[1], o.z + bv[2], o.w + bv[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c4 * 4 + k < Cout) dst[k] = ov[k];
        }
    }
}

// U[xi][co][ci] = (G g G^T)[xi],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void winograd_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int rows_pad, int KP) {
    const int64_t total = (int64_t)rows_pad * KP;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i / KP), ci = (int)(i - (int64_t)co * KP);
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = (co < Cout && ci < Cin) ? w[(((size_t)co * Cin + ci) * 3 + r) * 3 + s] : 0.f;
    float t[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[0][c] = g[0][c];
        t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
        t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
        t[3][c] = g[2][c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float u0 = t[r][0];
        const float u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
        const float u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
        const float u3 = t[r][2];
        U[(size_t)(4 * r + 0) * total + i] = u0;
        U[(size_t)(4 * r + 1) * total + i] = u1;
        U[(size_t)(4 * r + 2) * total + i] = u2;
        U[(size_t)(4 * r + 3) * total + i] = u3;
    }
}

}  // namespace

extern "C" {

int ssdhip_winograd_tiles(int N, int H, int W, int* T, int* Tp) {
    if (N <= 0 || H <= 0 || W <= 0 || !T || !Tp) return SSDHIP_EINVAL;
    const int64_t t = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2);
    if (t * 16 >= (1ll << 31)) return SSDHIP_EINVAL;
    *T = (int)t;
    *Tp = (int)((t + 127) / 128 * 128);      // GEMM m-tiles (<= 128 rows) never straddle two xi planes
    return SSDHIP_OK;
}

int ssdhip_winograd_weight(const float* w_oihw, float* U, int Cout, int Cin, int rows_pad, int KP, void* stream) {
    SSDHIP_REQUIRE(w_oihw && U && Cout > 0 && Cin > 0 && rows_pad >= Cout && KP >= Cin && KP % 32 == 0, "winograd_weight: bad arguments");
    const int64_t total = (int64_t)rows_pad * KP;
    hipLaunchKernelGGL(winograd_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, U, Cout, Cin, rows_pad, KP);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_winograd_input(const float* x, const float* in_scale, const float* in_shift, float* V, int N, int H, int W, int C,
                          void* stream) {
    SSDHIP_REQUIRE(x && V && C > 0 && C % 4 == 0, "winograd_input: bad arguments (C % 4 == 0)");
    SSDHIP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "winograd_input: in_scale and in_shift go together");
    int T, Tp;
    SSDHIP_REQUIRE(ssdhip_winograd_tiles(N, H, W, &T, &Tp) == SSDHIP_OK, "winograd_input: bad shape");
    const int TH = (H + 1) / 2, TW = (W + 1) / 2, C4 = C / 4;
    const int64_t total = (int64_t)T * C4;
    dim3 grid(cdiv(total, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (in_scale)
        hipLaunchKernelGGL(winograd_input_kernel<true>, grid, block, 0, st, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const float4*>(in_scale), reinterpret_cast<const float4*>(in_shift),
                           reinterpret_cast<float4*>(V), H, W, C4, TH, TW, T, Tp);
    else
        hipLaunchKernelGGL(winograd_input_kernel<false>, grid, block, 0, st, reinterpret_cast<const float4*>(x), nullptr, nullptr,
                           reinterpret_cast<float4*>(V), H, W, C4, TH, TW, T, Tp);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_winograd_output(const float* Mo, const float* bias, float* y, float* stats, int N, int H, int W, int Cout, void* stream) {
    SSDHIP_REQUIRE(Mo && y && Cout > 0, "winograd_output: bad arguments");
    SSDHIP_REQUIRE(Cout % 4 == 0 && 256 % (Cout / 4) == 0, "winograd_output: Cout must be 4 * a divisor of 256 (<= 1024)");
    int T, Tp;
    SSDHIP_REQUIRE(ssdhip_winograd_tiles(N, H, W, &T, &Tp) == SSDHIP_OK, "winograd_output: bad shape");
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    hipLaunchKernelGGL(winograd_output_kernel, dim3(cdiv(T, WINO_TB)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(Mo), reinterpret_cast<const float4*>(bias), reinterpret_cast<float4*>(y), stats,
                       H, W, Cout / 4, TH, TW, T, Tp);
    SSDHIP_CHECK_LAUNCH();
    return cdiv(T, WINO_TB);      // rows of the BN partial table
}

int ssdhip_winograd_output_head(const float* Mo, const float* bias, float* y, int64_t y_batch_stride, int N, int H, int W, int Cout,
                                int Cp, void* stream) {
    SSDHIP_REQUIRE(Mo && y && Cout > 0 && Cp >= Cout && Cp % 4 == 0, "winograd_output_head: bad arguments");
    int T, Tp;
    SSDHIP_REQUIRE(ssdhip_winograd_tiles(N, H, W, &T, &Tp) == SSDHIP_OK, "winograd_output_head: bad shape");
    const int TH = (H + 1) / 2, TW = (W + 1) / 2, Cp4 = Cp / 4;
    hipLaunchKernelGGL(winograd_output_head_kernel, dim3(cdiv((int64_t)T * Cp4, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(Mo), bias, y, (long long)y_batch_stride, H, W, Cout, Cp4, TH, TW, T, Tp);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
