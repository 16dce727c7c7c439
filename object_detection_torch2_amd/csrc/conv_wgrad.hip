// Weight gradient of an NHWC convolution on v_mfma_f32_32x32x2_f32 (autograd of nn.Conv2d, model/ssd.py:61,70-77;
// reached from loss.backward(), train.py:121).
//
//   dW[co][tap][ci] = sum_m dy[m][co] * a[pix(m, tap)][ci]        GEMM: rows = co, cols = ci (per tap), K = pixels m
//   block tile  128 co x 128 ci x 32 pixels for one (tap, k-split); 4 waves (2 x 2), each 64 x 64 = 2 x 2 MFMA tiles
//   staging     both operands are "k-major" in memory (pixel rows, channel-contiguous), exactly the order the MFMA
//               wants with k on the lane halves: lane (i, h) reads float2 {row 2kk+h, channels 2i, 2i+1} with one
//               conflict-free ds_read_b64 and feeds 2 tiles; tile t of a wave owns channels {2i+t}.  No transposes,
//               no LDS padding.  Input a = relu(bn(z)) is recomputed while staging (same fused transform as forward).
//   split-K     pixel range split over blocks; partial tiles go to a slab ws[split][co][tap][ci] with plain stores and a
//               second kernel sums the slabs in fixed order into the OIHW state_dict layout (bit-reproducible, no atomics)
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TM = 128, TN = 128, TK = 32;

struct FastDiv {   // exact n / d for 0 <= n < 2^31 (round-up magic number, 64-bit add): replaces ~40-instruction integer division
    unsigned mul, shift;
};
static FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    unsigned l = 0;
    while ((1u << l) < d) ++l;
    f.shift = l;
    f.mul = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << l) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ int fastdiv(int n, FastDiv f) {
    return (int)(((unsigned long long)__umulhi((unsigned)n, f.mul) + (unsigned)n) >> f.shift);
}

struct WgradArgs {
    const float* x; const float* dy; const float* in_scale; const float* in_shift; float* ws;
    long long dy_batch_stride;
    int M, Hi, Wi, Cin, Ho, Wo, HoWo, Cout, CoutRow, KS, taps, stride, pad;
    int CB, IB, splitK, stages_per_split, stages, CoutT, CinT;
    FastDiv dHoWo, dWo;
    // Winograd-domain use: `tap` indexes one of 16 planes; plane p's operands start p * stride floats further and the rows
    // are plain (no spatial gather): the geometry fields are set to a 1 x Tp image by the host
    long long x_plane_stride, dy_plane_stride;
    int plane_mode;
};

// PLANE: the Winograd-domain use (a.plane_mode): rows are plain -- no (n, ho, wo) decode, no spatial bounds; the generic index arithmetic was
// ~160 of the 222 VALU instructions of a K-step, on the lanes the fp32 MFMAs run on
template <bool XFORM, bool PLANE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TK * TM];
    float* As = lds;                 // [2][TK][TM]  dy rows
    float* Bs = lds + 2 * TK * TM;   // [2][TK][TN]  gathered input rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int b = blockIdx.x;
    const int ib = b % a.IB; b /= a.IB;
    const int cb = b % a.CB; b /= a.CB;
    const int tap = b % a.taps; b /= a.taps;
    const int split = b;
    const int co0 = cb * TM, ci0 = ib * TN;
    const int r = a.plane_mode ? 0 : tap / a.KS, s = a.plane_mode ? 0 : tap - r * a.KS;
    const float* xbase = a.x + (a.plane_mode ? (size_t)tap * a.x_plane_stride : (size_t)0);
    const float* dybase = a.dy + (a.plane_mode ? (size_t)tap * a.dy_plane_stride : (size_t)0);

    const int lrow = tid >> 5, col4 = tid & 31;     // 8 rows x 32 float4 per pass, 4 passes
    const int st0 = split * a.stages_per_split;
    const int st1 = min(a.stages, st0 + a.stages_per_split);

    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
    const int ci = ci0 + col4 * 4;
    const bool ci_ok = ci < a.Cin;
    if (XFORM && ci_ok) {
        sc = *reinterpret_cast<const float4*>(a.in_scale + ci);
        sh = *reinterpret_cast<const float4*>(a.in_shift + ci);
    }
    const int co = co0 + col4 * 4;
    const bool co_ok = co < a.CoutRow;

    // Same staging discipline as conv_igemm.hip: named registers, unconditional loads (masked at LDS-store time),
    // loads of tile st+1 pinned ahead of the MFMA block of tile st, store phase at raised priority.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    unsigned amask, bmask;
#define WG_LOAD(I)                                                                                                    \
    {                                                                                                                 \
        const int m = st_ * TK + lrow + 8 * I;                                                                        \
        const bool mok = m < a.M;                                                                                     \
        bool aok, bok;                                                                                                \
        size_t aoff, boff;                                                                                            \
        if (PLANE) {                                                                                                  \
            aok = mok && co_ok; bok = mok && ci_ok;                                                                   \
            aoff = aok ? (size_t)m * a.CoutRow + co : (size_t)0;                                                      \
            boff = bok ? (size_t)m * a.Cin + ci : (size_t)0;                                                          \
        } else {                                                                                                      \
            const int n = fastdiv(m, a.dHoWo);                                                                        \
            const int rem = m - n * a.HoWo;                                                                           \
            const int ho = fastdiv(rem, a.dWo), wo = rem - ho * a.Wo;                                                 \
            const int hi = ho * a.stride - a.pad + r, wi = wo * a.stride - a.pad + s;                                 \
            aok = mok && co_ok;                                                                                       \
            bok = mok && ci_ok && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;                     \
            aoff = aok ? (size_t)n * a.dy_batch_stride + (size_t)rem * a.CoutRow + co : (size_t)0;                    \
            boff = bok ? ((size_t)(n * a.Hi + hi) * a.Wi + wi) * a.Cin + ci : (size_t)0;                              \
        }                                                                                                             \
        ra##I = *reinterpret_cast<const float4*>(dybase + aoff);                                                      \
        rb##I = *reinterpret_cast<const float4*>(xbase + boff);                                                       \
        amask |= (aok ? 1u : 0u) << I;                                                                                \
        bmask |= (bok ? 1u : 0u) << I;                                                                                \
    }
#define WG_LOAD_STAGE(ST_) { const int st_ = (ST_); amask = 0; bmask = 0; WG_LOAD(0) WG_LOAD(1) WG_LOAD(2) WG_LOAD(3) }
#define WG_STORE(I)                                                                                                   \
    {                                                                                                                 \
        float4 va = ra##I, vb = rb##I;                                                                                \
        if (XFORM) {                                                                                                  \
            vb.x = fmaxf(fmaf(vb.x, sc.x, sh.x), 0.f); vb.y = fmaxf(fmaf(vb.y, sc.y, sh.y), 0.f);                     \
            vb.z = fmaxf(fmaf(vb.z, sc.z, sh.z), 0.f); vb.w = fmaxf(fmaf(vb.w, sc.w, sh.w), 0.f);                     \
        }                                                                                                             \
        if (!((amask >> I) & 1u)) va = make_float4(0.f, 0.f, 0.f, 0.f);                                               \
        if (!((bmask >> I) & 1u)) vb = make_float4(0.f, 0.f, 0.f, 0.f);                                               \
        *reinterpret_cast<float4*>(Ap_ + 8 * I * TM) = va;                                                            \
        *reinterpret_cast<float4*>(Bp_ + 8 * I * TN) = vb;                                                            \
    }
#define WG_STORE_STAGE(BUF_)                                                                                          \
    {                                                                                                                 \
        float* Ap_ = As + ((BUF_) * TK + lrow) * TM + col4 * 4;                                                       \
        float* Bp_ = Bs + ((BUF_) * TK + lrow) * TN + col4 * 4;                                                       \
        WG_STORE(0) WG_STORE(1) WG_STORE(2) WG_STORE(3)                                                               \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

    // two-ahead software pipeline (same scheme as conv_igemm.hip): LDS holds tile st, registers hold tile st+1; the LDS
    // writes of st+1 and the address math + global loads of st+2 are sliced between the 16 groups of four MFMAs.
    if (st0 < st1) {
        WG_LOAD_STAGE(st0)
        WG_STORE_STAGE(0)
        WG_LOAD_STAGE(st0 + 1 < st1 ? st0 + 1 : st0)
    }
    __syncthreads();
    const int fi = (lane & 31) * 2, fh = lane >> 5;
    for (int st = st0; st < st1; ++st) {
        const int buf = (st - st0) & 1;
        const int st_ = st + 2 < st1 ? st + 2 : st1 - 1;       // tile loaded during this step
        float* Ap_ = As + ((buf ^ 1) * TK + lrow) * TM + col4 * 4;
        float* Bp_ = Bs + ((buf ^ 1) * TK + lrow) * TN + col4 * 4;
        const float* Ab = As + (buf * TK + fh) * TM + wm * 64 + fi;
        const float* Bb = Bs + (buf * TK + fh) * TN + wn * 64 + fi;
        float2 fa[2], fb[2];
        fa[0] = *reinterpret_cast<const float2*>(Ab);
        fb[0] = *reinterpret_cast<const float2*>(Bb);
#pragma unroll
        for (int kk = 0; kk < TK / 2; ++kk) {
            const int cur = kk & 1;
            if (kk + 1 < TK / 2) {
                fa[cur ^ 1] = *reinterpret_cast<const float2*>(Ab + (kk + 1) * 2 * TM);
                fb[cur ^ 1] = *reinterpret_cast<const float2*>(Bb + (kk + 1) * 2 * TN);
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].y, acc[1][1], 0, 0, 0);
            switch (kk) {          // constant after unrolling
                case 0: WG_STORE(0) break;  case 1: WG_STORE(1) break;  case 2: WG_STORE(2) break;  case 3: WG_STORE(3) break;
                case 5: amask = 0; bmask = 0; WG_LOAD(0) break;
                case 7: WG_LOAD(1) break;  case 9: WG_LOAD(2) break;  case 11: WG_LOAD(3) break;
                default: break;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#undef WG_LOAD
#undef WG_LOAD_STAGE
#undef WG_STORE
#undef WG_STORE_STAGE
    // slab store: ws[split][co][tap][ci]; tile (t,u) element e: co = co0+wm*64+2*row+t, ci = ci0+wn*64+2*(lane&31)+u
    float* slab = a.ws + (size_t)split * a.CoutT * a.taps * a.CinT;
    const int cj = ci0 + wn * 64 + (lane & 31) * 2;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const int cco = co0 + wm * 64 + 2 * row + t;
            float2 v = make_float2(acc[t][0][e], acc[t][1][e]);
            *reinterpret_cast<float2*>(slab + ((size_t)cco * a.taps + tap) * a.CinT + cj) = v;
        }
}

// BASELINE configs[3] (bf16 forward / backward, fp32 master weights): the same weight gradient with both operands rounded to bf16
// on the way into LDS and the products on v_mfma_f32_32x32x16_bf16 (fp32 accumulation, fp32 slabs, same split-K and reduce pass).
// Both operands are pixel-major in memory while the MFMA wants 8 consecutive k = pixels per lane: the LDS images stay pixel-major
// ([32 pixel rows][128 channels] bf16, 256-B rows, 16-B chunk ch of row r at ch ^ (((r & 3) << 2) | ((r >> 2) & 3))) and the
// fragments are read TRANSPOSED with ds_read_b64_tr_b16 (4 pixel rows x 16 channels per 16 lanes; conflict-free with that XOR).
// The matrix pipe is 16x faster than on the fp32 kernel, so the launch is bound by its staging (global loads, BN.ReLU, rounding).
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x8 __attribute__((ext_vector_type(8)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ws16x4* lds_s16x4_t;

template <bool XFORM>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(WgradArgs a) {
    constexpr int IMG = TK * 256;                                    // one operand image: 32 pixel rows x 128 channels bf16
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * IMG];   // [buffer][dy, a]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int b = blockIdx.x;
    const int ib = b % a.IB; b /= a.IB;
    const int cb = b % a.CB; b /= a.CB;
    const int tap = b % a.taps; b /= a.taps;
    const int split = b;
    const int co0 = cb * TM, ci0 = ib * TN;
    const int r = tap / a.KS, s = tap - r * a.KS;
    const float* xbase = a.x;
    const float* dybase = a.dy;

    const int lrow = tid >> 5, col4 = tid & 31;
    const int st0 = split * a.stages_per_split;
    const int st1 = min(a.stages, st0 + a.stages_per_split);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
    const int ci = ci0 + col4 * 4;
    const bool ci_ok = ci < a.Cin;
    if (XFORM && ci_ok) {
        sc = *reinterpret_cast<const float4*>(a.in_scale + ci);
        sh = *reinterpret_cast<const float4*>(a.in_shift + ci);
    }
    const int co = co0 + col4 * 4;
    const bool co_ok = co < a.CoutRow;

    float4 ra[4], rb[4];
    unsigned amask = 0, bmask = 0;
    auto load_stage = [&](int st_) {
        amask = 0; bmask = 0;
#pragma unroll
        for (int I = 0; I < 4; ++I) {
            const int m = st_ * TK + lrow + 8 * I;
            const bool mok = m < a.M;
            const int n = fastdiv(m, a.dHoWo);
            const int rem = m - n * a.HoWo;
            const int ho = fastdiv(rem, a.dWo), wo = rem - ho * a.Wo;
            const int hi = ho * a.stride - a.pad + r, wi = wo * a.stride - a.pad + s;
            const bool aok = mok && co_ok;
            const bool bok = mok && ci_ok && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
            const size_t aoff = aok ? (size_t)n * a.dy_batch_stride + (size_t)rem * a.CoutRow + co : (size_t)0;
            const size_t boff = bok ? ((size_t)(n * a.Hi + hi) * a.Wi + wi) * a.Cin + ci : (size_t)0;
            ra[I] = *reinterpret_cast<const float4*>(dybase + aoff);
            rb[I] = *reinterpret_cast<const float4*>(xbase + boff);
            amask |= (aok ? 1u : 0u) << I;
            bmask |= (bok ? 1u : 0u) << I;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int I = 0; I < 4; ++I) {
            float4 va = ra[I], vb = rb[I];
            if (XFORM) {
                vb.x = fmaxf(fmaf(vb.x, sc.x, sh.x), 0.f); vb.y = fmaxf(fmaf(vb.y, sc.y, sh.y), 0.f);
                vb.z = fmaxf(fmaf(vb.z, sc.z, sh.z), 0.f); vb.w = fmaxf(fmaf(vb.w, sc.w, sh.w), 0.f);
            }
            if (!((amask >> I) & 1u)) va = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!((bmask >> I) & 1u)) vb = make_float4(0.f, 0.f, 0.f, 0.f);
            const int row = lrow + 8 * I;
            const int off = 256 * row + 16 * ((col4 >> 1) ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (col4 & 1);
            const wf32x4 fa = {va.x, va.y, va.z, va.w}, fb = {vb.x, vb.y, vb.z, vb.w};
            *reinterpret_cast<wbf16x4*>(lds + (buf * 2 + 0) * IMG + off) = __builtin_convertvector(fa, wbf16x4);
            *reinterpret_cast<wbf16x4*>(lds + (buf * 2 + 1) * IMG + off) = __builtin_convertvector(fb, wbf16x4);
        }
    };

    // transposed fragment reads: 16-lane group g = lane >> 4 takes pixel rows r0 .. r0 + 3 (r0 = 16 kc + 8 (g >> 1) + 4 jj) and the 16
    // channels wX * 64 + 32 t + 16 (g & 1) ..; lane 4 q + p of the group supplies row r0 + q, chunk c0 + (p >> 1), byte 8 (p & 1)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int fo[2][2][2];      // [operand][tile][jj]: byte offset inside an image for kc = 0 (kc adds 16 rows = 4,096 B: same XOR term)
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int row = 8 * (g >> 1) + 4 * jj + q;
                const int ch = (op ? wn : wm) * 8 + 4 * t + 2 * (g & 1) + (pp >> 1);
                fo[op][t][jj] = 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (pp & 1);
            }
    auto frag = [&](int buf, int op, int t, int kc) -> wbf16x8 {
        const char* base = lds + (buf * 2 + op) * IMG + kc * 4096;
        const ws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + fo[op][t][0]));
        const ws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + fo[op][t][1]));
        return __builtin_bit_cast(wbf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

    if (st0 < st1) {
        load_stage(st0);
        store_stage(0);
        load_stage(st0 + 1 < st1 ? st0 + 1 : st0);
    }
    __syncthreads();
    for (int st = st0; st < st1; ++st) {
        const int buf = (st - st0) & 1;
        store_stage(buf ^ 1);                                   // registers hold tile st + 1
        load_stage(st + 2 < st1 ? st + 2 : st1 - 1);            // in flight during the MFMAs below
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            wbf16x8 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { fa[t] = frag(buf, 0, t, kc); fb[t] = frag(buf, 1, t, kc); }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t], fb[u], acc[t][u], 0, 0, 0);
        }
        __syncthreads();
    }
    // slab store: ws[split][co][tap][ci]; tile (t, u) element e: co = co0 + wm*64 + 32 t + row, ci = ci0 + wn*64 + 32 u + (lane & 31)
    float* slab = a.ws + (size_t)split * a.CoutT * a.taps * a.CinT;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const int cco = co0 + wm * 64 + 32 * t + row;
#pragma unroll
            for (int u = 0; u < 2; ++u) slab[((size_t)cco * a.taps + tap) * a.CinT + ci0 + wn * 64 + 32 * u + (lane & 31)] = acc[t][u][e];
        }
}

// dw_oihw[co][ci][tap] = sum_split ws[split][co][tap][ci]
// (Cin = channels of dw's OIHW layout: the padded RGB input has 4 in x but 3 in the parameter)
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin, int taps, int CoutT,
                                    int CinT, int splitK) {
    const int64_t total = (int64_t)Cout * taps * Cin;
    const size_t slab = (size_t)CoutT * taps * CinT;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        int64_t q = i / Cin;
        const int tap = (int)(q % taps);
        const int co = (int)(q / taps);
        const size_t src = ((size_t)co * taps + tap) * CinT + ci;
        float v = 0.f;
        for (int s = 0; s < splitK; ++s) v += ws[s * slab + src];
        dw[((size_t)co * Cin + ci) * taps + tap] = v;
    }
}

// column sums of a (M, row) matrix for the first C columns: partial (MB, C, 2) = [sum, 0] (consumed by bn_stats_stage1)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, long long batch_stride, int HoWo,
                                                             int64_t M, int C, int row, int rows_per_block,
                                                             float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t lo = (int64_t)blockIdx.y * rows_per_block, hi = min(M, lo + rows_per_block);
    float s = 0.f;
    if (c < C)
        for (int64_t m = lo + rl; m < hi; m += 4) {
            const int64_t n = m / HoWo, rem = m - n * HoWo;
            s += dy[n * batch_stride + rem * row + c];
        }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        s += red[1][cl] + red[2][cl] + red[3][cl];
        partial[((size_t)blockIdx.y * C + c) * 2] = s;
        partial[((size_t)blockIdx.y * C + c) * 2 + 1] = 0.f;
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int MB, int C, float* __restrict__ out) {
    // 64 channels x 4 row lanes per block; fixed order: lane k sums rows k, k + 4, .. in fp64, then lanes 0 .. 3 in order
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < C)
        for (int r = rl; r < MB; r += 4) s += (double)partial[((size_t)r * C + c) * 2];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) out[c] = (float)(((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl]);
}

// ---- Winograd F(2x2,3x3) weight gradient (3x3 / stride 1 / pad 1 layers): dU[xi] = sum_t dM[xi][t]^T V[xi][t], 2.25x fewer flops ----
// dM[xi][t][co] = (A dY A^T)[xi] of every 2x2 output-gradient tile, A = [[1,0],[1,1],[1,-1],[0,-1]]; zero outside the image
__global__ __launch_bounds__(256) void winograd_outgrad_kernel(const float4* __restrict__ dy, float4* __restrict__ dM, int H, int W,
                                                               int C4, int TH, int TW, int T, int Tp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * C4) return;
    const int c4 = (int)(idx % C4);
    const int t = (int)(idx / C4);
    const int tw = t % TW;
    const int q = t / TW;
    const int th = q % TH;
    const int n = q / TH;
    float4 y[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int h = 2 * th + i, w = 2 * tw + j;
            y[i][j] = (h < H && w < W) ? dy[((size_t)(n * H + h) * W + w) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    auto add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    auto sub = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
    auto neg = [](float4 a) { return make_float4(-a.x, -a.y, -a.z, -a.w); };
    float4 r[4][2];                                   // A dY
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        r[0][j] = y[0][j]; r[1][j] = add(y[0][j], y[1][j]); r[2][j] = sub(y[0][j], y[1][j]); r[3][j] = neg(y[1][j]);
    }
    const size_t plane = (size_t)Tp * C4;
    float4* o = dM + (size_t)t * C4 + c4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                     // (A dY) A^T
        o[(size_t)(4 * i + 0) * plane] = r[i][0];
        o[(size_t)(4 * i + 1) * plane] = add(r[i][0], r[i][1]);
        o[(size_t)(4 * i + 2) * plane] = sub(r[i][0], r[i][1]);
        o[(size_t)(4 * i + 3) * plane] = neg(r[i][1]);
    }
}

// dW[co][ci][3][3] = G^T (sum_split dU) G,  G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]];  slabs ws[split][co][xi][ci]
__global__ void winograd_wgrad_output_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin, int CoutT, int CinT,
                                             int splitK) {
    const int64_t total = (int64_t)Cout * Cin;
    const size_t slab = (size_t)CoutT * 16 * CinT;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin), co = (int)(i / Cin);
        float u[4][4];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const size_t src = ((size_t)co * 16 + xi) * CinT + ci;
            float v = 0.f;
            for (int s = 0; s < splitK; ++s) v += ws[s * slab + src];
            u[xi >> 2][xi & 3] = v;
        }
        float t[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[0][c] = u[0][c] + 0.5f * (u[1][c] + u[2][c]);
            t[1][c] = 0.5f * (u[1][c] - u[2][c]);
            t[2][c] = 0.5f * (u[1][c] + u[2][c]) + u[3][c];
        }
        float* o = dw + ((size_t)co * Cin + ci) * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o[r * 3 + 0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
            o[r * 3 + 1] = 0.5f * (t[r][1] - t[r][2]);
            o[r * 3 + 2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
        }
    }
}

// ---- Winograd F(4x4,3x3) weight gradient (same layers, points 0, +-3/2, +-2/3 of winograd4.hip): dU[xi] = sum_t dM[xi][t]^T V[xi][t] over
// 36 planes of a quarter as many tiles: 1.78x fewer MFMA flops than the F(2x2) form.  dM = A dY A^T with A = (A^T)^T of winograd4.hip,
// dW = G^T dU G computed in fp64 from the fp32 slabs.
// dM[xi][t][co] of every 4x4 output-gradient tile; zero outside the image
__global__ __launch_bounds__(256) void winograd4_outgrad_kernel(const float4* __restrict__ dy, float4* __restrict__ dM, int H, int W,
                                                                int C4, int TH, int TW, int T, int Tp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * C4) return;
    const int c4 = (int)(idx % C4);
    const int t = (int)(idx / C4);
    const int tw = t % TW;
    const int q = t / TW;
    const int th = q % TH;
    const int n = q / TH;
    constexpr float k3_2 = 1.5f, k9_4 = 2.25f, k27_8 = 3.375f, k2_3 = 2.0f / 3.0f, k4_9 = 4.0f / 9.0f, k8_27 = 8.0f / 27.0f;
    float4 y[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = min(4 * th + i, H - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = min(4 * tw + j, W - 1);
            y[i][j] = dy[((size_t)(n * H + h) * W + w) * C4 + c4];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * th + i >= H || 4 * tw + j >= W) y[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fma4 = [](float s, float4 b, float4 a) { return make_float4(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y), fmaf(s, b.z, a.z), fmaf(s, b.w, a.w)); };
    auto mul4 = [](float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); };
    auto add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    auto sub = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
    // m = A y for a 4-vector: (y0, e1 + o1, e1 - o1, e2 + o2, e2 - o2, y3)
    auto a6 = [&](float4 y0, float4 y1, float4 y2, float4 y3, float4* m) {
        const float4 e1 = fma4(k9_4, y2, y0), o1 = fma4(k27_8, y3, mul4(k3_2, y1));
        const float4 e2 = fma4(k4_9, y2, y0), o2 = fma4(k8_27, y3, mul4(k2_3, y1));
        m[0] = y0; m[1] = add(e1, o1); m[2] = sub(e1, o1); m[3] = add(e2, o2); m[4] = sub(e2, o2); m[5] = y3;
    };
    float4 r[6][4];                                   // A dY
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 m[6];
        a6(y[0][j], y[1][j], y[2][j], y[3][j], m);
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = m[i];
    }
    const size_t plane = (size_t)Tp * C4;
    float4* o = dM + (size_t)t * C4 + c4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {                     // (A dY) A^T
        float4 m[6];
        a6(r[i][0], r[i][1], r[i][2], r[i][3], m);
#pragma unroll
        for (int j = 0; j < 6; ++j) o[(size_t)(6 * i + j) * plane] = m[j];
    }
}

// dW[co][ci][3][3] = G^T (sum_split dU) G in fp64; slabs ws[split][co][xi][ci].  One thread = one output channel x 2 input channels:
// the 36 * splitK slab reads are independent 8-B loads (the one-channel form with the split loop innermost ran at 2 TB/s: 75 us for
// conv_6_1's 150 MB of slabs)
__global__ __launch_bounds__(256) void winograd4_wgrad_output_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin,
                                                                     int CoutT, int CinT, int splitK) {
    const int Cin4 = (Cin + 1) / 2;
    const int64_t total = (int64_t)Cout * Cin4;
    const size_t slab = (size_t)CoutT * 36 * CinT;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % Cin4), co = (int)(i / Cin4);
        const int ci = c4 * 2;
        double u[36][2];
#pragma unroll
        for (int xi = 0; xi < 36; ++xi)
#pragma unroll
            for (int k = 0; k < 2; ++k) u[xi][k] = 0.0;
        for (int s = 0; s < splitK; ++s) {
            const float* base = ws + s * slab + (size_t)co * 36 * CinT + ci;          // CinT is a multiple of 2: 8-B aligned
#pragma unroll
            for (int xi = 0; xi < 36; ++xi) {
                const float2 v = *reinterpret_cast<const float2*>(base + (size_t)xi * CinT);
                u[xi][0] += (double)v.x; u[xi][1] += (double)v.y;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (ci + k >= Cin) break;
            double t[3][6];                               // G^T u over the first plane index, then over the second
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const double s12 = u[6 + b][k] + u[12 + b][k], d12 = u[6 + b][k] - u[12 + b][k];
                const double s34 = u[18 + b][k] + u[24 + b][k], d34 = u[18 + b][k] - u[24 + b][k];
                t[0][b] = u[b][k] + (8.0 / 65.0) * s12 - (81.0 / 130.0) * s34;
                t[1][b] = (12.0 / 65.0) * d12 - (27.0 / 65.0) * d34;
                t[2][b] = (18.0 / 65.0) * (s12 - s34) + u[30 + b][k];
            }
            float* o = dw + ((size_t)co * Cin + ci + k) * 9;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double s12 = t[r][1] + t[r][2], d12 = t[r][1] - t[r][2], s34 = t[r][3] + t[r][4], d34 = t[r][3] - t[r][4];
                o[r * 3 + 0] = (float)(t[r][0] + (8.0 / 65.0) * s12 - (81.0 / 130.0) * s34);
                o[r * 3 + 1] = (float)((12.0 / 65.0) * d12 - (27.0 / 65.0) * d34);
                o[r * 3 + 2] = (float)((18.0 / 65.0) * (s12 - s34) + t[r][5]);
            }
        }
    }
}

// split-K slabs of the F(4x4) weight gradient summed in fixed order (fp64, one rounding) into slab 0: fully parallel and coalesced.  The heads
// have few outputs and many splits (det_4_3: 51,200 weights, 12 slabs of 9.4 MB): summed inside the transform kernel that was 100 blocks
// walking 432 loads each (70 us); as a pass of its own it is bandwidth-bound
__global__ __launch_bounds__(256) void slab_sum_kernel(float4* __restrict__ ws, int64_t slab4, int splitK) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slab4; i += (int64_t)gridDim.x * blockDim.x) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int s = 0; s < splitK; ++s) {
            const float4 v = ws[s * slab4 + i];
            a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
        }
        ws[i] = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
    }
}

void choose_split(WgradArgs& a) {
    // split-K: the grid runs in rounds of 512 resident blocks (256 CUs x 2); pick the split that minimises
    // rounds x K-steps per block, plus ~3 K-steps per block for its slab store and share of the reduce pass
    const int tiles = a.CB * a.IB * a.taps;
    int best_sk = 1;
    long best_cost = -1;
    for (int sk = 1; sk <= 32; ++sk) {
        const int per = cdiv(a.stages, sk);
        if (sk > 1 && per < 4) break;
        const int splits = cdiv(a.stages, per);
        const long rounds = cdiv((long)tiles * splits, 512);
        const long cost = rounds * (per + 3);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_sk = sk; }
    }
    a.stages_per_split = cdiv(a.stages, best_sk);
    a.splitK = cdiv(a.stages, a.stages_per_split);
}

// plane-mode plan: 16 planes of (T x CoutRow)^T (T x Cin) products over Tp-row planes
void plan_winograd(const ssdhip_wgrad_params* p, WgradArgs& a, int T, int Tp, float* V, float* dM, float* slabs, int planes = 16) {
    a.x = V; a.dy = dM; a.in_scale = nullptr; a.in_shift = nullptr; a.ws = slabs;
    a.dy_batch_stride = 0;
    a.M = T; a.Hi = 1; a.Wi = Tp; a.Cin = p->Cin; a.Ho = 1; a.Wo = Tp; a.HoWo = Tp;
    a.Cout = p->Cout; a.CoutRow = p->dy_row; a.KS = 1; a.taps = planes; a.stride = 1; a.pad = 0;
    a.CB = cdiv(p->Cout, TM); a.IB = cdiv(p->Cin, TN);
    a.CoutT = a.CB * TM; a.CinT = a.IB * TN;
    a.dHoWo = make_fastdiv((unsigned)a.HoWo); a.dWo = make_fastdiv((unsigned)a.Wo);
    a.stages = cdiv(a.M, TK);
    a.plane_mode = 1; a.x_plane_stride = (long long)Tp * p->Cin; a.dy_plane_stride = (long long)Tp * p->dy_row;
    choose_split(a);
}

void plan(const ssdhip_wgrad_params* p, WgradArgs& a) {
    a.x = p->x; a.dy = p->dy; a.in_scale = p->in_scale; a.in_shift = p->in_shift; a.ws = p->ws;
    a.dy_batch_stride = p->dy_batch_stride;
    a.M = p->N * p->Ho * p->Wo; a.Hi = p->Hi; a.Wi = p->Wi; a.Cin = p->Cin; a.Ho = p->Ho; a.Wo = p->Wo; a.HoWo = p->Ho * p->Wo;
    a.Cout = p->Cout; a.CoutRow = p->dy_row; a.KS = p->S; a.taps = p->R * p->S; a.stride = p->stride; a.pad = p->pad;
    a.CB = cdiv(p->Cout, TM); a.IB = cdiv(p->Cin, TN);
    a.CoutT = a.CB * TM; a.CinT = a.IB * TN;
    a.dHoWo = make_fastdiv((unsigned)a.HoWo); a.dWo = make_fastdiv((unsigned)a.Wo);
    a.stages = cdiv(a.M, TK);
    a.plane_mode = 0; a.x_plane_stride = 0; a.dy_plane_stride = 0;
    choose_split(a);
}

}  // namespace

extern "C" {

int64_t ssdhip_conv2d_wgrad_ws_bytes(const ssdhip_wgrad_params* p) {
    if (!p || p->N <= 0 || p->Cin <= 0 || p->Cout <= 0) return SSDHIP_EINVAL;
    WgradArgs a;
    plan(p, a);
    int64_t slabs = (int64_t)a.splitK * a.CoutT * a.taps * a.CinT * 4;
    int64_t bias = (int64_t)256 * a.Cout * 2 * 4;
    return slabs + bias;
}

int ssdhip_conv2d_wgrad(const ssdhip_wgrad_params* p, void* stream) {
    SSDHIP_REQUIRE(p && p->x && p->dy && p->dw_oihw && p->ws, "wgrad: null pointer");
    SSDHIP_REQUIRE(p->Cin % 4 == 0 && p->dy_row % 4 == 0 && p->dy_row >= p->Cout, "wgrad: Cin and dy_row must be multiples of 4");
    SSDHIP_REQUIRE(p->R == p->S && (p->R == 1 || p->R == 3), "wgrad: 1x1 or 3x3 kernels");
    SSDHIP_REQUIRE(p->compute == 0 || p->compute == 1, "wgrad: compute must be 0 (fp32) or 1 (bf16 operands)");
    SSDHIP_REQUIRE((p->in_scale == nullptr) == (p->in_shift == nullptr), "wgrad: in_scale and in_shift go together");
    SSDHIP_REQUIRE(elems_below(1ll << 31, p->N, p->Hi, p->Wi), "wgrad: tensor too large");
    WgradArgs a;
    plan(p, a);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(a.CB * a.IB * a.taps * a.splitK), block(256);
    if (p->compute == 1) {
        if (p->in_scale) hipLaunchKernelGGL(wgrad_bf16_kernel<true>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(wgrad_bf16_kernel<false>, grid, block, 0, st, a);
    } else {
        // 1x1 / stride 1 / pad 0 on dense gradients (conv_7 and the 1x1 extras): output pixel m reads input pixel m -- the plain-row form
        const bool flat = a.KS == 1 && a.stride == 1 && a.pad == 0 && a.Hi == a.Ho && a.Wi == a.Wo && a.dy_batch_stride == (long long)a.HoWo * a.CoutRow;
        if (p->in_scale) {
            if (flat) hipLaunchKernelGGL((wgrad_kernel<true, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL(wgrad_kernel<true>, grid, block, 0, st, a);
        } else {
            if (flat) hipLaunchKernelGGL((wgrad_kernel<false, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL(wgrad_kernel<false>, grid, block, 0, st, a);
        }
    }
    SSDHIP_CHECK_LAUNCH();
    const int dw_cin = p->dw_cin > 0 ? p->dw_cin : p->Cin;
    SSDHIP_REQUIRE(dw_cin <= p->Cin, "wgrad: dw_cin must not exceed Cin");
    int64_t total = (int64_t)p->Cout * a.taps * dw_cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, a.ws, p->dw_oihw, p->Cout, dw_cin, a.taps, a.CoutT, a.CinT,
                       a.splitK);
    SSDHIP_CHECK_LAUNCH();
    if (p->dbias) {
        float* partial = a.ws + (size_t)a.splitK * a.CoutT * a.taps * a.CinT;
        int MB = cdiv(a.M, 64);
        if (MB > 256) MB = 256;
        const int rows_per_block = cdiv(a.M, MB);
        MB = cdiv(a.M, rows_per_block);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(p->Cout, 64), MB), dim3(256), 0, st, p->dy, (long long)p->dy_batch_stride,
                           a.HoWo, (int64_t)a.M, p->Cout, p->dy_row, rows_per_block, partial);
        SSDHIP_CHECK_LAUNCH();
        hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(p->Cout, 64)), dim3(256), 0, st, partial, MB, p->Cout, p->dbias);
        SSDHIP_CHECK_LAUNCH();
    }
    return SSDHIP_OK;
}

static int wino_geometry(const ssdhip_wgrad_params* p, int* T, int* Tp) {
    if (!(p->R == 3 && p->S == 3 && p->stride == 1 && p->pad == 1 && p->Ho == p->Hi && p->Wo == p->Wi)) return SSDHIP_EINVAL;
    return ssdhip_winograd_tiles(p->N, p->Hi, p->Wi, T, Tp);
}

int64_t ssdhip_conv2d_wgrad_winograd_ws_bytes(const ssdhip_wgrad_params* p) {
    int T, Tp;
    if (!p || wino_geometry(p, &T, &Tp) != SSDHIP_OK) return SSDHIP_EINVAL;
    WgradArgs a;
    plan_winograd(p, a, T, Tp, nullptr, nullptr, nullptr);
    int64_t v = (int64_t)16 * Tp * p->Cin * 4, dm = (int64_t)16 * Tp * p->dy_row * 4;
    int64_t slabs = (int64_t)a.splitK * a.CoutT * 16 * a.CinT * 4, bias = (int64_t)256 * p->Cout * 2 * 4;
    return v + dm + slabs + bias;
}

int ssdhip_conv2d_wgrad_winograd(const ssdhip_wgrad_params* p, void* stream) {
    SSDHIP_REQUIRE(p && p->x && p->dy && p->dw_oihw && p->ws, "wgrad_winograd: null pointer");
    SSDHIP_REQUIRE(p->Cin % 4 == 0 && p->dy_row % 4 == 0 && p->dy_row >= p->Cout, "wgrad_winograd: Cin and dy_row must be multiples of 4");
    SSDHIP_REQUIRE(p->dy_batch_stride == (int64_t)p->Ho * p->Wo * p->dy_row, "wgrad_winograd: dy must be dense (N,H,W,dy_row)");
    int T, Tp;
    SSDHIP_REQUIRE(wino_geometry(p, &T, &Tp) == SSDHIP_OK, "wgrad_winograd: 3x3 / stride 1 / pad 1 layers only");
    hipStream_t st = (hipStream_t)stream;
    // V may be handed in (the forward's input transform of the same activation); the workspace keeps its V region either way
    float* V = p->wino_v ? const_cast<float*>(p->wino_v) : p->ws;
    float* dM = p->ws + (size_t)16 * Tp * p->Cin;
    float* slabs = dM + (size_t)16 * Tp * p->dy_row;
    WgradArgs a;
    plan_winograd(p, a, T, Tp, V, dM, slabs);
    // 1. V = B^T d B of the layer input (producer BN scale/shift + ReLU fused on load)
    if (!p->wino_v) {
        int rc = ssdhip_winograd_input(p->x, p->in_scale, p->in_shift, V, p->N, p->Hi, p->Wi, p->Cin, stream);
        if (rc != SSDHIP_OK) return rc;
    }
    // 2. dM = A dY A^T of the output gradient
    const int TH = (p->Hi + 1) / 2, TW = (p->Wi + 1) / 2, C4 = p->dy_row / 4;
    hipLaunchKernelGGL(winograd_outgrad_kernel, dim3(cdiv((int64_t)T * C4, 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(p->dy),
                       reinterpret_cast<float4*>(dM), p->Hi, p->Wi, C4, TH, TW, T, Tp);
    SSDHIP_CHECK_LAUNCH();
    // 3. 16 plane products on the wgrad MFMA kernel (split-K slabs)
    hipLaunchKernelGGL((wgrad_kernel<false, true>), dim3(a.CB * a.IB * a.taps * a.splitK), dim3(256), 0, st, a);
    SSDHIP_CHECK_LAUNCH();
    // 4. dW = G^T (sum of slabs) G, OIHW
    int64_t total = (int64_t)p->Cout * p->Cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(winograd_wgrad_output_kernel, dim3(blocks), dim3(256), 0, st, slabs, p->dw_oihw, p->Cout, p->Cin, a.CoutT, a.CinT,
                       a.splitK);
    SSDHIP_CHECK_LAUNCH();
    if (p->dbias) {
        float* partial = slabs + (size_t)a.splitK * a.CoutT * 16 * a.CinT;
        const int M = p->N * p->Ho * p->Wo;
        int MB = cdiv(M, 64);
        if (MB > 256) MB = 256;
        const int rows_per_block = cdiv(M, MB);
        MB = cdiv(M, rows_per_block);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(p->Cout, 64), MB), dim3(256), 0, st, p->dy, (long long)p->dy_batch_stride,
                           p->Ho * p->Wo, (int64_t)M, p->Cout, p->dy_row, rows_per_block, partial);
        SSDHIP_CHECK_LAUNCH();
        hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(p->Cout, 64)), dim3(256), 0, st, partial, MB, p->Cout, p->dbias);
        SSDHIP_CHECK_LAUNCH();
    }
    return SSDHIP_OK;
}

static int wino4_geometry(const ssdhip_wgrad_params* p, int* T, int* Tp) {
    if (!(p->R == 3 && p->S == 3 && p->stride == 1 && p->pad == 1 && p->Ho == p->Hi && p->Wo == p->Wi)) return SSDHIP_EINVAL;
    return ssdhip_winograd4_tiles(p->N, p->Hi, p->Wi, 128, T, Tp);
}

int64_t ssdhip_conv2d_wgrad_winograd4_ws_bytes(const ssdhip_wgrad_params* p) {
    int T, Tp;
    if (!p || wino4_geometry(p, &T, &Tp) != SSDHIP_OK) return SSDHIP_EINVAL;
    WgradArgs a;
    plan_winograd(p, a, T, Tp, nullptr, nullptr, nullptr, 36);
    int64_t v = (int64_t)36 * Tp * p->Cin * 4, dm = (int64_t)36 * Tp * p->dy_row * 4;
    int64_t slabs = (int64_t)a.splitK * a.CoutT * 36 * a.CinT * 4, bias = (int64_t)256 * p->Cout * 2 * 4;
    return v + dm + slabs + bias;
}

int ssdhip_conv2d_wgrad_winograd4(const ssdhip_wgrad_params* p, void* stream) {
    SSDHIP_REQUIRE(p && p->x && p->dy && p->dw_oihw && p->ws, "wgrad_winograd4: null pointer");
    SSDHIP_REQUIRE(p->Cin % 4 == 0 && p->dy_row % 4 == 0 && p->dy_row >= p->Cout, "wgrad_winograd4: Cin and dy_row must be multiples of 4");
    SSDHIP_REQUIRE(p->dy_batch_stride == (int64_t)p->Ho * p->Wo * p->dy_row, "wgrad_winograd4: dy must be dense (N,H,W,dy_row)");
    SSDHIP_REQUIRE(p->compute == 0, "wgrad_winograd4: fp32 only");
    int T, Tp;
    SSDHIP_REQUIRE(wino4_geometry(p, &T, &Tp) == SSDHIP_OK, "wgrad_winograd4: 3x3 / stride 1 / pad 1 layers only");
    hipStream_t st = (hipStream_t)stream;
    // V may be handed in (the forward's F(4x4) input transform of the same activation, planes padded to 128 rows); the workspace keeps its
    // V region either way
    float* V = p->wino_v ? const_cast<float*>(p->wino_v) : p->ws;
    float* dM = p->ws + (size_t)36 * Tp * p->Cin;
    float* slabs = dM + (size_t)36 * Tp * p->dy_row;
    WgradArgs a;
    plan_winograd(p, a, T, Tp, V, dM, slabs, 36);
    if (!p->wino_v) {
        int rc = ssdhip_winograd4_input(p->x, p->in_scale, p->in_shift, V, p->N, p->Hi, p->Wi, p->Cin, Tp, stream);
        if (rc != SSDHIP_OK) return rc;
    }
    const int TH = (p->Hi + 3) / 4, TW = (p->Wi + 3) / 4, C4 = p->dy_row / 4;
    hipLaunchKernelGGL(winograd4_outgrad_kernel, dim3(cdiv((int64_t)T * C4, 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(p->dy),
                       reinterpret_cast<float4*>(dM), p->Hi, p->Wi, C4, TH, TW, T, Tp);
    SSDHIP_CHECK_LAUNCH();
    hipLaunchKernelGGL((wgrad_kernel<false, true>), dim3(a.CB * a.IB * a.taps * a.splitK), dim3(256), 0, st, a);
    SSDHIP_CHECK_LAUNCH();
    int splits_left = a.splitK;
    if (a.splitK > 1) {
        const int64_t slab4 = (int64_t)a.CoutT * 36 * a.CinT / 4;
        int sb = (int)((slab4 + 255) / 256);
        if (sb > 4096) sb = 4096;
        hipLaunchKernelGGL(slab_sum_kernel, dim3(sb), dim3(256), 0, st, reinterpret_cast<float4*>(slabs), slab4, a.splitK);
        SSDHIP_CHECK_LAUNCH();
        splits_left = 1;
    }
    int64_t total = (int64_t)p->Cout * ((p->Cin + 1) / 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(winograd4_wgrad_output_kernel, dim3(blocks), dim3(256), 0, st, slabs, p->dw_oihw, p->Cout, p->Cin, a.CoutT, a.CinT,
                       splits_left);
    SSDHIP_CHECK_LAUNCH();
    if (p->dbias) {
        float* partial = slabs + (size_t)a.splitK * a.CoutT * 36 * a.CinT;
        const int M = p->N * p->Ho * p->Wo;
        int MB = cdiv(M, 64);
        if (MB > 256) MB = 256;
        const int rows_per_block = cdiv(M, MB);
        MB = cdiv(M, rows_per_block);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(p->Cout, 64), MB), dim3(256), 0, st, p->dy, (long long)p->dy_batch_stride,
                           p->Ho * p->Wo, (int64_t)M, p->Cout, p->dy_row, rows_per_block, partial);
        SSDHIP_CHECK_LAUNCH();
        hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(p->Cout, 64)), dim3(256), 0, st, partial, MB, p->Cout, p->dbias);
        SSDHIP_CHECK_LAUNCH();
    }
    return SSDHIP_OK;
}

}  // extern "C"
