// NHWC implicit-GEMM convolution on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32), gfx950.
// Replaces nn.Conv2d forward (model/vgg16.py:33, model/ssd.py:61,70-77) and, in gather mode 1, its data gradient.
//
//   GEMM view   M = N*Ho*Wo output pixels, N = Cout, K = R*S*Cin (k = tap*Cin + ci, ci contiguous in NHWC)
//   block tile  BM x BN x 32, 256 threads = 4 waves (2 x 2), each wave (BM/2) x (BN/2) = MT x NT MFMA tiles of 32x32
//   staging     global -> registers (16 B per lane, whole 128-B channel rows) -> [fused BN scale/shift + ReLU of the
//               producer layer, zero padding after it] -> LDS, rows padded to 36 floats so every ds_read_b128 of a
//               16-lane group hits 16 distinct 4-bank slots (conflict-free); two LDS buffers, one barrier per K-step,
//               next tile's global loads issued before the MFMAs of the current one
//   MFMA feed   a lane reads 4 consecutive k of its row with one ds_read_b128 and issues 4 MFMAs from it: MFMA j of
//               chunk c sums k = 8c+j (lanes 0-31) and 8c+4+j (lanes 32-63).  A and B use the same k permutation, and
//               the sum over k is order-free, so no transposes are needed.  LDS read traffic: 16 B per lane per 4 MFMAs.
//   epilogue    + bias, NHWC store (each store instruction writes two full 128-B lines), optional per-(m-block,
//               channel) partial sum / sum of squares for train-mode BatchNorm (deterministic: no atomics)
//   tile order  XCD-aware: each XCD's L2 sees a contiguous run of m-tiles with all their n-tiles
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_ROW = 36;  // floats per staged row (32 + 4 pad)

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* in_scale;
    const float* in_shift;
    float* y;
    float* stats;
    long long y_batch_stride;
    int M, Hi, Wi, Cin, cpt, Ho, Wo, HoWo, Cout, taps, stride, lgStride, pad, KP, KT, MB, NB, accumulate;
};

template <int KS>
__device__ __forceinline__ void tap_rs(int tap, int& r, int& s) {
    if (KS == 1) { r = 0; s = 0; }
    else { r = (tap * 11) >> 5; s = tap - r * 3; }   // tap/3 for tap < 16
}

template <int BM, int BN, int KS, int MODE, bool XFORM, bool SMALLC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_igemm_kernel(ConvArgs a) {
    constexpr int MT = BM / 64, NT = BN / 64;     // 32x32 MFMA tiles per wave
    constexpr int AL = BM / 32, BL = BN / 32;     // float4 loads per thread per stage
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_ROW];
    float* As = lds;
    float* Bs = lds + 2 * BM * LDS_ROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective tile order (blocks b, b+8, ... share an XCD)
    int t;
    {
        const int T = a.MB * a.NB, bid = blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3, q = T >> 3, r = T & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int mt_idx = t / a.NB, nt_idx = t - mt_idx * a.NB;
    const int m0 = mt_idx * BM, n0 = nt_idx * BN;

    // ---- per-thread gather state: AL rows of the A tile, fixed 16-B column slot --------------------------------
    const int lrow = tid >> 3, col4 = tid & 7;
    int a_base[4], a_h[4], a_w[4];
    a_base[2] = a_base[3] = 0; a_h[2] = a_h[3] = a_w[2] = a_w[3] = -100000;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        int m = m0 + lrow + 32 * i;
        if (m < a.M) {
            int n = m / a.HoWo;
            int rem = m - n * a.HoWo;
            int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            a_base[i] = n * a.Hi * a.Wi;
            if (MODE == 0) { a_h[i] = ho * a.stride - a.pad; a_w[i] = wo * a.stride - a.pad; }
            else { a_h[i] = ho + a.pad; a_w[i] = wo + a.pad; }
        } else {
            a_base[i] = 0; a_h[i] = -100000; a_w[i] = -100000;
        }
    }
    const float* wrow[4];
    wrow[2] = wrow[3] = a.w;
#pragma unroll
    for (int i = 0; i < BL; ++i) wrow[i] = a.w + (size_t)(n0 + lrow + 32 * i) * a.KP + col4 * 4;

    // Staging registers are NAMED scalars driven by macros (no arrays, no lambdas, no conditional stages): hipcc kept
    // array/lambda forms of this code in scratch memory.  Loads are unconditional (out-of-image taps read the tensor base
    // and are zeroed at store time; the last iteration re-loads the final stage) so there are no branches and no early
    // s_waitcnt: the global loads of tile kt+1 stay in flight under the MFMAs of tile kt.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, sc, sh;
    ra2 = ra3 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned okmask;
    // k = tap * Cin + c.  Cin % 32 == 0: a stage is 32 channels of one tap (tap uniform over the block);
    // SMALLC (Cin == 4, the padded RGB image): a stage is 8 taps x 4 channels, one tap per 16-B slot.
#define CONV_LOAD_A(I)                                                                                                \
    {                                                                                                                 \
        int hi, wi;                                                                                                   \
        bool ok = tap < a.taps;                                                                                       \
        if (MODE == 0) { hi = a_h[I] + r; wi = a_w[I] + s; }                                                          \
        else {                                                                                                        \
            int th = a_h[I] - r, tw = a_w[I] - s;                                                                     \
            ok = ok && ((th | tw) & (a.stride - 1)) == 0 && th >= 0 && tw >= 0;                                       \
            hi = th >> a.lgStride; wi = tw >> a.lgStride;                                                             \
        }                                                                                                             \
        ok = ok && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;                                    \
        const int pix = ok ? (a_base[I] + hi * a.Wi + wi) : 0;                                                        \
        ra##I = *reinterpret_cast<const float4*>(a.x + (size_t)pix * a.Cin + c);                                      \
        okmask |= (ok ? 1u : 0u) << I;                                                                                \
    }
#define CONV_LOAD_STAGE(KT_)                                                                                          \
    {                                                                                                                 \
        const int kt_ = (KT_);                                                                                        \
        int tap, c;                                                                                                   \
        if (SMALLC) { tap = kt_ * 8 + col4; c = 0; }                                                                  \
        else { tap = kt_ / a.cpt; c = (kt_ - tap * a.cpt) * BK + col4 * 4; }                                          \
        int r, s;                                                                                                     \
        tap_rs<KS>(tap, r, s);                                                                                        \
        if (XFORM) {                                                                                                  \
            sc = *reinterpret_cast<const float4*>(a.in_scale + c);                                                    \
            sh = *reinterpret_cast<const float4*>(a.in_shift + c);                                                    \
        }                                                                                                             \
        okmask = 0;                                                                                                   \
        rb0 = *reinterpret_cast<const float4*>(wrow[0] + kt_ * BK);                                                   \
        rb1 = *reinterpret_cast<const float4*>(wrow[1] + kt_ * BK);                                                   \
        if (BL > 2) {                                                                                                 \
            rb2 = *reinterpret_cast<const float4*>(wrow[BL > 2 ? 2 : 0] + kt_ * BK);                                  \
            rb3 = *reinterpret_cast<const float4*>(wrow[BL > 2 ? 3 : 0] + kt_ * BK);                                  \
        }                                                                                                             \
        CONV_LOAD_A(0) CONV_LOAD_A(1)                                                                                 \
        if (AL > 2) { CONV_LOAD_A(2) CONV_LOAD_A(3) }                                                                 \
    }
#define CONV_STORE_A(I)                                                                                               \
    {                                                                                                                 \
        float4 v = ra##I;                                                                                             \
        if (XFORM) {                                                                                                  \
            v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);                         \
            v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);                         \
        }                                                                                                             \
        if (!((okmask >> I) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f); /* zero padding AFTER the transform */        \
        *reinterpret_cast<float4*>(Ab_ + 32 * I * LDS_ROW) = v;                                                       \
    }
#define CONV_STORE_STAGE(BUF_)                                                                                        \
    {                                                                                                                 \
        float* Ab_ = As + (BUF_) * BM * LDS_ROW + lrow * LDS_ROW + col4 * 4;                                          \
        float* Bb_ = Bs + (BUF_) * BN * LDS_ROW + lrow * LDS_ROW + col4 * 4;                                          \
        *reinterpret_cast<float4*>(Bb_) = rb0;                                                                        \
        *reinterpret_cast<float4*>(Bb_ + 32 * LDS_ROW) = rb1;                                                         \
        if (BL > 2) {                                                                                                 \
            *reinterpret_cast<float4*>(Bb_ + 64 * LDS_ROW) = rb2;                                                     \
            *reinterpret_cast<float4*>(Bb_ + 96 * LDS_ROW) = rb3;                                                     \
        }                                                                                                             \
        CONV_STORE_A(0) CONV_STORE_A(1)                                                                               \
        if (AL > 2) { CONV_STORE_A(2) CONV_STORE_A(3) }                                                               \
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    CONV_LOAD_STAGE(0)
    CONV_STORE_STAGE(0)
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 4;
    const int last = a.KT - 1;
    for (int kt = 0; kt < a.KT; ++kt) {
        const int buf = kt & 1;
        CONV_LOAD_STAGE(kt < last ? kt + 1 : last)
        __builtin_amdgcn_sched_barrier(0);   // keep every global load of tile kt+1 ahead of the MFMA block (hipcc sinks them)
        const float* Ab = As + buf * BM * LDS_ROW + (wm * (BM / 2) + frow) * LDS_ROW + fk;
        const float* Bb = Bs + buf * BN * LDS_ROW + (wn * (BN / 2) + frow) * LDS_ROW + fk;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            float4 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_ROW + kc * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_ROW + kc * 8);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);   // ... and the wait + LDS write of that tile behind it
        CONV_STORE_STAGE(buf ^ 1)      // on the last iteration this fills a buffer nobody reads
        __syncthreads();
    }
#undef CONV_LOAD_STAGE
#undef CONV_STORE_STAGE
#undef CONV_LOAD_A
#undef CONV_STORE_A

    // ---- epilogue ------------------------------------------------------------------------------------------------
    // C/D map of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const int ccol = lane & 31, rquad = 4 * (lane >> 5);
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = n0 + wn * (BN / 2) + j * 32 + ccol;
        const bool cok = co < a.Cout;
        const float bv = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + rquad;
                if (m < a.M && cok) {
                    float v = acc[i][j][e] + bv;
                    const int n = m / a.HoWo;
                    const int rem = m - n * a.HoWo;
                    float* dst = a.y + (size_t)n * a.y_batch_stride + (size_t)rem * a.Cout + co;
                    if (a.accumulate) v += *dst;
                    *dst = v;
                    t1 += v;
                    t2 += v * v;
                }
            }
        }
        s1[j] = t1; s2[j] = t2;
    }
    if (a.stats != nullptr) {
        __syncthreads();                       // all waves are done with the staging buffers
        float* red = lds;                      // [wm][BN][2]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
            float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            if (lane < 32) {
                const int cl = wn * (BN / 2) + j * 32 + lane;
                red[(wm * BN + cl) * 2] = t1;
                red[(wm * BN + cl) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int co = n0 + tid;
            if (co < a.Cout) {
                float2 o;
                o.x = red[tid * 2] + red[(BN + tid) * 2];
                o.y = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
                *reinterpret_cast<float2*>(a.stats + ((size_t)mt_idx * a.Cout + co) * 2) = o;
            }
        }
    }
}

// OIHW -> packed rows.  mode 0: row = co, inner = ci;  mode 1: row = ci, inner = co.  k = tap * inner_pad + inner.
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int R, int S,
                                   int rows_pad, int inner_pad, int KP, int mode) {
    const int64_t total = (int64_t)rows_pad * KP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / KP), k = (int)(i - (int64_t)row * KP);
        const int tap = k / inner_pad, inner = k - tap * inner_pad;
        float v = 0.f;
        if (tap < R * S) {
            const int r = tap / S, s = tap - r * S;
            const int co = mode == 0 ? row : inner, ci = mode == 0 ? inner : row;
            if (co < Cout && ci < Cin) v = w[(((size_t)co * Cin + ci) * R + r) * S + s];
        }
        packed[i] = v;
    }
}

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

int choose_tile(const ssdhip_conv_params* p, int& BM, int& BN) {
    if (p->tile) { BM = p->tile / 1000; BN = p->tile % 1000; }
    else {
        const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
        BN = (p->Cout <= 64 || cdiv(p->Cout, 64) * 64 < cdiv(p->Cout, 128) * 128) ? 64 : 128;
        BM = 128;
        // small grids: halve the m-tile so the chip (256 CUs x 2 resident blocks) sees enough tiles per round
        if ((int64_t)cdiv(M, 128) * cdiv(p->Cout, BN) < 1024) BM = 64;
    }
    if (!((BM == 128 || BM == 64) && (BN == 128 || BN == 64))) return SSDHIP_EINVAL;
    return SSDHIP_OK;
}

template <int BM, int BN, int KS>
void launch_conv(const ConvArgs& a, int mode, bool xform, hipStream_t st) {
    dim3 grid(a.MB * a.NB), block(256);
    if (a.Cin == 4) {
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, KS, 0, false, true>), grid, block, 0, st, a);
    } else if (mode == 0) {
        if (xform) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, KS, 0, true, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, KS, 0, false, false>), grid, block, 0, st, a);
    } else {
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, KS, 1, false, false>), grid, block, 0, st, a);
    }
}

}  // namespace

extern "C" {

int ssdhip_conv2d_block_m(const ssdhip_conv_params* p) {
    int BM, BN;
    if (!p || choose_tile(p, BM, BN) != SSDHIP_OK) return SSDHIP_EINVAL;
    return BM;
}

int ssdhip_conv2d_tile(const ssdhip_conv_params* p) {
    int BM, BN;
    if (!p || choose_tile(p, BM, BN) != SSDHIP_OK) return SSDHIP_EINVAL;
    return BM * 1000 + BN;
}

int ssdhip_conv2d_nhwc(const ssdhip_conv_params* p, void* stream) {
    SSDHIP_REQUIRE(p && p->x && p->w && p->y, "conv2d: null pointer");
    SSDHIP_REQUIRE(p->N > 0 && p->Hi > 0 && p->Wi > 0 && p->Ho > 0 && p->Wo > 0 && p->Cout > 0, "conv2d: bad shape");
    SSDHIP_REQUIRE(p->Cin == 4 || p->Cin % 32 == 0, "conv2d: Cin must be 4 (padded RGB) or a multiple of 32 (pad channels)");
    SSDHIP_REQUIRE(!(p->Cin == 4 && (p->mode != 0 || p->in_scale)), "conv2d: the Cin=4 path is forward-only without input transform");
    SSDHIP_REQUIRE(p->R == p->S && (p->R == 1 || p->R == 3), "conv2d: 1x1 or 3x3 kernels");
    SSDHIP_REQUIRE(p->stride == 1 || p->stride == 2, "conv2d: stride 1 or 2");
    SSDHIP_REQUIRE(p->KP % BK == 0 && p->KP >= p->R * p->S * p->Cin, "conv2d: KP must be a multiple of 32 covering R*S*Cin");
    SSDHIP_REQUIRE(p->mode == 0 || p->mode == 1, "conv2d: mode 0 (fwd) or 1 (dgrad)");
    SSDHIP_REQUIRE(!(p->mode == 1 && p->in_scale), "conv2d: no input transform in dgrad mode");
    SSDHIP_REQUIRE((p->in_scale == nullptr) == (p->in_shift == nullptr), "conv2d: in_scale and in_shift go together");
    SSDHIP_REQUIRE((int64_t)p->N * p->Hi * p->Wi * p->Cin < (1ll << 31) && (int64_t)p->N * p->Ho * p->Wo < (1ll << 31),
                   "conv2d: tensor too large for 32-bit pixel indexing");
    if (p->mode == 0) {
        SSDHIP_REQUIRE(p->Ho == (p->Hi + 2 * p->pad - p->R) / p->stride + 1 && p->Wo == (p->Wi + 2 * p->pad - p->S) / p->stride + 1,
                       "conv2d: output size does not match input/stride/pad");
    }
    int BM, BN;
    SSDHIP_REQUIRE(choose_tile(p, BM, BN) == SSDHIP_OK, "conv2d: tile must be one of 128128/128064/64128/64064");
    ConvArgs a;
    a.x = p->x; a.w = p->w; a.bias = p->bias; a.in_scale = p->in_scale; a.in_shift = p->in_shift; a.y = p->y; a.stats = p->stats;
    a.y_batch_stride = p->y_batch_stride;
    a.M = p->N * p->Ho * p->Wo; a.Hi = p->Hi; a.Wi = p->Wi; a.Cin = p->Cin; a.cpt = p->Cin >= BK ? p->Cin / BK : 1;
    a.Ho = p->Ho; a.Wo = p->Wo; a.HoWo = p->Ho * p->Wo; a.Cout = p->Cout; a.taps = p->R * p->S;
    a.stride = p->stride; a.lgStride = ilog2(p->stride); a.pad = p->pad; a.KP = p->KP; a.KT = p->KP / BK;
    a.MB = cdiv(a.M, BM); a.NB = cdiv(p->Cout, BN); a.accumulate = p->accumulate;
    hipStream_t st = (hipStream_t)stream;
    const bool xf = p->in_scale != nullptr;
    const int key = BM * 1000 + BN;
#define SSDHIP_CONV_CASE(bm, bn)                                            \
    case bm * 1000 + bn:                                                    \
        if (p->R == 3) launch_conv<bm, bn, 3>(a, p->mode, xf, st);          \
        else launch_conv<bm, bn, 1>(a, p->mode, xf, st);                    \
        break;
    switch (key) {
        SSDHIP_CONV_CASE(128, 128)
        SSDHIP_CONV_CASE(128, 64)
        SSDHIP_CONV_CASE(64, 128)
        SSDHIP_CONV_CASE(64, 64)
        default: ssdhip_set_error("conv2d: unsupported tile"); return SSDHIP_ENOSUP;
    }
#undef SSDHIP_CONV_CASE
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_pack_conv_weight(const float* w_oihw, float* packed, int Cout, int Cin, int R, int S, int rows_pad, int inner_pad,
                            int KP, int mode, void* stream) {
    SSDHIP_REQUIRE(w_oihw && packed && Cout > 0 && Cin > 0 && R > 0 && S > 0, "pack_conv_weight: bad arguments");
    SSDHIP_REQUIRE(mode == 0 || mode == 1, "pack_conv_weight: mode 0 or 1");
    SSDHIP_REQUIRE(rows_pad >= (mode == 0 ? Cout : Cin) && inner_pad >= (mode == 0 ? Cin : Cout) && KP >= R * S * inner_pad,
                   "pack_conv_weight: padding smaller than the tensor");
    int64_t total = (int64_t)rows_pad * KP;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout, Cin, R, S, rows_pad,
                       inner_pad, KP, mode);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
