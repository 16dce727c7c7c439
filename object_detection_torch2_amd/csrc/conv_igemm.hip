// NHWC implicit-GEMM convolution on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32), gfx950.
// Replaces nn.Conv2d forward (model/vgg16.py:33, model/ssd.py:61,70-77) and, in gather mode 1, its data gradient.
//
//   GEMM view   M = N*Ho*Wo output pixels, N = Cout, K = R*S*Cin walked as (32-channel block, tap, channel): ci contiguous in
//               NHWC, and the taps of one channel block adjacent in time (cache reuse of the input lines)
//   block tile  BM x BN x 32, 256 threads = 4 waves (2 x 2), each wave (BM/2) x (BN/2) = MT x NT MFMA tiles of 32x32
//   staging     global -> registers (16 B per lane, whole 128-B channel rows) -> [fused BN scale/shift + ReLU of the
//               producer layer, zero padding after it] -> LDS rows of 128 B whose 16-B chunks are XOR-swizzled with
//               (row >> 1) & 7, so every ds_read_b128 lane group covers all 16 slots of the 256-B bank row (conflict-free,
//               no padding: 64 KiB for 128x128, 48 KiB for 128x64 -> 3 blocks per CU); two LDS buffers, one barrier per
//               K-step; two-ahead software pipeline: registers hold tile kt+1 while LDS holds tile kt, and the staging
//               instructions (LDS writes of kt+1, address math + global loads of kt+2) are sliced between groups of four
//               MFMAs so they issue in the matrix pipe's shadow
//   MFMA feed   a lane reads 4 consecutive k of its row with one ds_read_b128 and issues 4 MFMAs from it: MFMA j of
//               chunk c sums k = 8c+j (lanes 0-31) and 8c+4+j (lanes 32-63).  A and B use the same k permutation, and
//               the sum over k is order-free, so no transposes are needed.  LDS read traffic: 16 B per lane per 4 MFMAs.
//   epilogue    + bias, NHWC store (each store instruction writes two full 128-B lines), optional per-(m-block,
//               channel) partial sum / sum of squares for train-mode BatchNorm (deterministic: no atomics)
//   tile order  XCD-aware: each XCD's L2 sees a contiguous run of m-tiles with all their n-tiles
//   BF16 = true (BASELINE configs[3], opt-in; never the headline): the same gather / staging on the fp32 tensors, but the staged
//               values are rounded to bf16 on the way into LDS (v_cvt_pk_bf16_f32; rows of 64 B, 16-B chunks XOR-swizzled with
//               (row >> 2) & 3) and the products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulators (one ds_read_b128 = the
//               8 k of one MFMA operand); epilogue, BN statistics and every tensor in HBM stay fp32
#include "common.h"
#include <stdlib.h>

// gemm_planes.hip: LDS-DMA staged plane GEMM (Winograd domain)
long long ssdhip_gemm_planes_queue_ws_bytes(void);
int ssdhip_gemm_planes_queue(const float* v, const float* u, float* y, int M, int K, int Cout, int rows_per_plane, long long u_plane_stride,
                             void* ws, int max_cus, hipStream_t st);
int ssdhip_gemm_planes_dma(const float* v, const float* u, float* y, int M, int K, int Cout, int rows_per_plane, long long u_plane_stride,
                           int BM, int BN, int tpb, hipStream_t st, int abl);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {      // round-to-nearest-even pair, lo in bits 0-15
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

constexpr int BK = 32;
constexpr int LDS_ROW = 32;  // floats per staged row: 128 B, no padding; 16-B chunks XOR-swizzled with (row >> 1) & 7

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* in_scale;
    const float* in_shift;
    float* y;
    float* stats;
    long long y_batch_stride;
    int M, Hi, Wi, Cin, cpt, Ho, Wo, HoWo, Cout, taps, stride, lgStride, pad, KP, KT, MB, NB, accumulate, dense;
    // iteration space (Ho, Wo) maps to output pixel (oh0 + ostep*h, ow0 + ostep*w) of an OWf-wide image; taps walked are
    // r = r0 + rstep*ri (ri < nr), s = s0 + rstep*si (si < ns).  Forward and stride-1 dgrad: identity / all taps.  Stride-2
    // dgrad runs once per output-parity class with only the taps whose source pixel exists (2.25 of 9 on average).
    int OWf, oh0, ow0, ostep, r0, s0, rstep, nr, ns;
    // batched GEMM use (Winograd domain): block rows m0.. belong to plane m0 / rows_per_img, whose weights start w_img_stride
    // floats further (rows_per_img is a multiple of the m-tile so no tile straddles two planes); 0 = one weight matrix
    long long w_img_stride;
    int rows_per_img;
    // persistent plane-GEMM blocks: each block walks tpb consecutive m-tiles of one n-tile as ONE software pipeline (the loads
    // of the next tile's first K-steps are in flight during the current tile's last MFMAs and its stores); 1 = one tile per block
    int tpb;
    // split-K for launches whose tile grid cannot fill the chip (the 10x10 .. 1x1 maps of the tail: a few blocks walking 36-144
    // K-steps each, one exposed load latency per step): block z = blockIdx.x / tiles walks K-steps [z * kt_per, (z + 1) * kt_per) and
    // stores its raw accumulators to slab z; conv_splitk_reduce_kernel sums the slabs in fixed order and does the epilogue
    // (bias, accumulate, strided store, BatchNorm partial sums).  ksplit == 1: off.
    int ksplit, kt_per;
    float* slab;
};

template <int KS>
__device__ __forceinline__ void tap_rs(int tap, int& r, int& s) {
    if (KS == 1) { r = 0; s = 0; }
    else { r = (tap * 11) >> 5; s = tap - r * 3; }   // tap/3 for tap < 16
}

// ABL: timing-only ablations (wrong results): 1 = no global loads in the K loop, 2 = no LDS writes in the K loop
template <int BM, int BN>
struct ConvWaves { static constexpr int value = (BM + BN == 256) ? 2 : (BM + BN == 192) ? 3 : 4; };   // blocks per CU the LDS footprint admits

template <int BM, int BN, int KS, int MODE, bool XFORM, bool SMALLC, int ABL = 0, bool BF16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, ConvWaves<BM, BN>::value))) void conv_igemm_kernel(ConvArgs a) {
    constexpr int WROWS = BM / 2, WCOLS = BN / 2;     // per-wave output tile (wave grid 2 x 2)
    constexpr int MT = WROWS / 32, NT = WCOLS / 32;   // 32x32 MFMA tiles per wave
    constexpr int AL = BM / 32, BL = BN / 32;         // float4 loads per thread per stage (2 or 4)
    constexpr int GROUPS = 4 * MT * NT;               // groups of MFMAs per K-step (4 each; MT*NT each in element-major order)
    constexpr int SLICES = 17;                        // staging slices spread over the groups (see CONV_SLICE)
    constexpr int LROW = BF16 ? LDS_ROW / 2 : LDS_ROW;   // floats of LDS per staged row (bf16 rows are 64 B)
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LROW];
    float* As = lds;
    float* Bs = lds + 2 * BM * LROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective tile order (blocks b, b+8, ... share an XCD)
    int t;
    {
        const int T = ((a.MB + a.tpb - 1) / a.tpb) * a.NB, bid = a.ksplit > 1 ? (int)(blockIdx.x % (unsigned)T) : (int)blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3, q = T >> 3, r = T & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int mchunk = t / a.NB, nt_idx = t - mchunk * a.NB;
    const int mt_idx = mchunk * a.tpb;                         // first m-tile of this block
    const int ntl = min(a.tpb, a.MB - mt_idx);                 // m-tiles this block walks
    const int m0 = mt_idx * BM, n0 = nt_idx * BN;
    const int split = a.ksplit > 1 ? (int)(blockIdx.x / (unsigned)(a.MB * a.NB)) : 0;
    const int kt_lo = split * a.kt_per, kt_hi = a.ksplit > 1 ? min(a.KT, kt_lo + a.kt_per) : a.KT;      // this block's K-steps

    // ---- per-thread gather state: AL rows of the A tile, fixed 16-B column slot --------------------------------
    // 32-bit BYTE offsets against uniform base pointers (global_load saddr + voffset form, no 64-bit VALU math)
    const int lrow = tid >> 3, col4 = tid & 7;
    const char* xb = reinterpret_cast<const char*>(a.x);
    const char* wb = reinterpret_cast<const char*>(a.w);
    if (a.rows_per_img > 0) wb += (size_t)(m0 / a.rows_per_img) * (size_t)a.w_img_stride * 4;
    int ld_tile = 0;                       // loader's m-tile within the block (persistent plane-GEMM mode)
    bool ld_wrap = false;
    int a_h[4], a_w[4], a_pix[4];          // MODE 0: a_pix = pixel index of tap (0,0); MODE 1: image base pixel
    unsigned w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + lrow + 32 * i;
        if (i < AL && m < a.M) {
            int n = m / a.HoWo;
            int rem = m - n * a.HoWo;
            int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            if (MODE == 0) {
                a_h[i] = ho * a.stride - a.pad; a_w[i] = wo * a.stride - a.pad;
                a_pix[i] = (n * a.Hi + a_h[i]) * a.Wi + a_w[i];
            } else {
                a_h[i] = a.oh0 + a.ostep * ho + a.pad; a_w[i] = a.ow0 + a.ostep * wo + a.pad;
                a_pix[i] = n * a.Hi * a.Wi;
            }
        } else {
            a_h[i] = -100000; a_w[i] = -100000; a_pix[i] = 0;
        }
        w_off[i] = (unsigned)(((n0 + lrow + 32 * (i < BL ? i : 0)) * a.KP + col4 * 4) * 4);
    }

    // ---- staging state -------------------------------------------------------------------------------------------
    // Registers are NAMED scalars driven by macros (no arrays / lambdas / conditional stages: hipcc kept those forms in
    // scratch).  Loads are unconditional (out-of-image taps read the tensor base and are zeroed at LDS-store time; past
    // the last K-step the final stage is re-loaded), so there are no branches and no early s_waitcnt.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, sc, sh;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = sc = sh = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned okmask = 0;
    // loader position: stage ld_kt = (channel block ld_cc, tap ld_tap), k = (cc * taps + tap) * 32 + c % 32: channel block
    // OUTER, tap INNER, so the 9 taps re-read each 128-B input line in 9 consecutive K-steps (L1/L2 hits; with the tap outer
    // the re-reads were a whole K loop apart and the per-XCD L2 thrashed: 3.3x the algorithmic bytes at the fabric counters).
    // SMALLC (Cin == 4, padded RGB): a stage is 8 taps x 4 channels, k = tap * 4 + c.
    int ld_kt = kt_lo, ld_cc = SMALLC ? 0 : kt_lo / (a.nr * a.ns), ld_tap = SMALLC ? 0 : kt_lo - ld_cc * (a.nr * a.ns);
    int tap, r, s, c;
    unsigned tapoff, woff_k;
#define CONV_PREP()                                                                                                   \
    {                                                                                                                 \
        if (ld_wrap) {             /* the loader enters the block's next m-tile (rows + BM, maybe the next weight plane) */ \
            ld_wrap = false; ++ld_tile;                                                                               \
            a_pix[0] += BM; a_pix[1] += BM; a_pix[2] += BM; a_pix[3] += BM;                                           \
            wb = reinterpret_cast<const char*>(a.w) + (size_t)((m0 + ld_tile * BM) / a.rows_per_img) * (size_t)a.w_img_stride * 4; \
        }                                                                                                             \
        if (SMALLC) { tap = ld_kt * 8 + col4; c = 0; tap_rs<KS>(tap, r, s); woff_k = (unsigned)(ld_kt * BK * 4); }     \
        else {                                                                                                        \
            const int ri = a.ns == 3 ? (ld_tap * 11) >> 5 : a.ns == 2 ? ld_tap >> 1 : ld_tap;                         \
            r = a.r0 + a.rstep * ri; s = a.s0 + a.rstep * (ld_tap - ri * a.ns);                                       \
            tap = r * KS + s; c = ld_cc * BK + col4 * 4;                                                              \
            woff_k = (unsigned)((ld_cc * a.taps + tap) * BK * 4);                                                     \
        }                                                                                                             \
        if (MODE == 0) tapoff = (unsigned)(((r * a.Wi + s) * a.Cin + c) * 4);                                         \
        if (XFORM) {                                                                                                  \
            sc = *reinterpret_cast<const float4*>(a.in_scale + c);                                                    \
            sh = *reinterpret_cast<const float4*>(a.in_shift + c);                                                    \
        }                                                                                                             \
        okmask = 0;                                                                                                   \
        if (ld_kt < kt_hi - 1) {   /* channel block outer, tap inner: the 9 taps of one 128-B line are consecutive */ \
            ++ld_kt;                                                                                                  \
            if (++ld_tap == a.nr * a.ns) { ld_tap = 0; ++ld_cc; }                                                     \
        } else if (ld_tile + 1 < ntl) { ld_kt = 0; ld_tap = 0; ld_cc = 0; ld_wrap = true; }                           \
    }
#define CONV_LOAD_A(I)                                                                                                \
    if (AL > I) {                                                                                                     \
        bool ok = tap < a.taps;                                                                                       \
        unsigned off;                                                                                                 \
        if (MODE == 0) {                                                                                              \
            ok = ok && (unsigned)(a_h[I] + r) < (unsigned)a.Hi && (unsigned)(a_w[I] + s) < (unsigned)a.Wi;            \
            off = (unsigned)(a_pix[I] * a.Cin * 4) + tapoff;                                                          \
        } else {                                                                                                      \
            const int th = a_h[I] - r, tw = a_w[I] - s;                                                               \
            const int hi = th >> a.lgStride, wi = tw >> a.lgStride;                                                   \
            ok = ok && ((th | tw) & (a.stride - 1)) == 0 && th >= 0 && tw >= 0 && hi < a.Hi && wi < a.Wi;             \
            off = (unsigned)(((a_pix[I] + hi * a.Wi + wi) * a.Cin + c) * 4);                                          \
        }                                                                                                             \
        if (ABL != 1 || !in_loop) ra##I = *reinterpret_cast<const float4*>(xb + (ok ? off : 0u));                     \
        else ra##I.x += __uint_as_float(off & 1u);                                                                    \
        okmask |= (ok ? 1u : 0u) << I;                                                                                \
    }
#define CONV_LOAD_B(I) if (BL > I) { if (ABL != 1 || !in_loop) rb##I = *reinterpret_cast<const float4*>(wb + (w_off[I] + woff_k)); else rb##I.x += __uint_as_float(woff_k & 1u); }
#define CONV_STORE_A(I)                                                                                               \
    if (AL > I) {                                                                                                     \
        float4 v = ra##I;                                                                                             \
        if (XFORM) {                                                                                                  \
            v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);                         \
            v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);                         \
        }                                                                                                             \
        if (!((okmask >> I) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f); /* zero padding AFTER the transform */        \
        if (BF16) *reinterpret_cast<uint2*>(Ast_ + 32 * I * LROW) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)); \
        else if (ABL != 2 || !in_loop) *reinterpret_cast<float4*>(Ast_ + 32 * I * LROW) = v;                          \
        else asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));                                              \
    }
#define CONV_STORE_B(I) if (BL > I) { if (BF16) *reinterpret_cast<uint2*>(Bst_ + 32 * I * LROW) = make_uint2(pack_bf16(rb##I.x, rb##I.y), pack_bf16(rb##I.z, rb##I.w)); else if (ABL != 2 || !in_loop) *reinterpret_cast<float4*>(Bst_ + 32 * I * LROW) = rb##I; else asm volatile("" :: "v"(rb##I.x), "v"(rb##I.y), "v"(rb##I.z), "v"(rb##I.w)); }
    // slice N of the staging work of one K-step: 0-7 write tile kt+1 (registers) to LDS, 8 advances the loader,
    // 9-16 issue the global loads of tile kt+2 into the same registers.  N is a constant after unrolling.
#define CONV_SLICE(N)                                                                                                 \
    switch (N) {                                                                                                      \
        case 0: CONV_STORE_B(0) break;  case 1: CONV_STORE_B(1) break;                                                \
        case 2: CONV_STORE_B(2) break;  case 3: CONV_STORE_B(3) break;                                                \
        case 4: CONV_STORE_A(0) break;  case 5: CONV_STORE_A(1) break;                                                \
        case 6: CONV_STORE_A(2) break;  case 7: CONV_STORE_A(3) break;                                                \
        case 8: CONV_PREP() break;                                                                                    \
        case 9: CONV_LOAD_B(0) break;   case 10: CONV_LOAD_B(1) break;                                                \
        case 11: CONV_LOAD_B(2) break;  case 12: CONV_LOAD_B(3) break;                                                \
        case 13: CONV_LOAD_A(0) break;  case 14: CONV_LOAD_A(1) break;                                                \
        case 15: CONV_LOAD_A(2) break;  case 16: CONV_LOAD_A(3) break;                                                \
        default: break;                                                                                               \
    }
#define CONV_LOAD_ALL() { CONV_PREP() CONV_LOAD_B(0) CONV_LOAD_B(1) CONV_LOAD_B(2) CONV_LOAD_B(3) CONV_LOAD_A(0) CONV_LOAD_A(1) CONV_LOAD_A(2) CONV_LOAD_A(3) }
#define CONV_STORE_ALL() { CONV_STORE_B(0) CONV_STORE_B(1) CONV_STORE_B(2) CONV_STORE_B(3) CONV_STORE_A(0) CONV_STORE_A(1) CONV_STORE_A(2) CONV_STORE_A(3) }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- software pipeline, two tiles ahead ------------------------------------------------------------------------
    // LDS holds tile kt, registers hold tile kt+1 (loaded one K-step ago).  K-step kt: 16*MT*NT MFMAs on tile kt in groups
    // of four; after each group one staging slice (registers -> LDS for tile kt+1, then the loads of tile kt+2) issues in
    // the matrix pipe's shadow.  sched_barrier(0) pins the group/slice order; one barrier per K-step.
    // LDS image: row r, logical 16-B chunk q lives at chunk q ^ ((r >> 1) & 7): rows r and r+32 share the swizzle, and any
    // 16 rows of a ds_read_b128 lane group cover all 16 slots of the 256-B bank row (conflict-free without padding)
    // bf16 image: row r, logical 16-B chunk q (8 k) at chunk q ^ ((r >> 2) & 3); a lane's 4 k are one 8-B half of a chunk
    const int st_chunk = BF16 ? ((col4 >> 1) ^ ((lrow >> 2) & 3)) * 4 + (col4 & 1) * 2 : (col4 ^ ((lrow >> 1) & 7)) * 4;
    float* Ast_ = As + lrow * LROW + st_chunk;
    float* Bst_ = Bs + lrow * LROW + st_chunk;
    bool in_loop = false;
    CONV_LOAD_ALL()          // tile 0
    CONV_STORE_ALL()
    CONV_LOAD_ALL()          // tile 1 (or tile 0 again when KT == 1)
    __syncthreads();

    in_loop = true;
    const int frow = lane & 31, fh = lane >> 5, fx = (frow >> 1) & 7;
    int fko[4];                      // float offset of k-chunk kc for this lane: logical chunk 2*kc + fh, swizzled
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) fko[kc] = ((2 * kc + fh) ^ fx) * 4;
    const float* Afr = As + (wm * WROWS + frow) * LROW;
    const float* Bfr = Bs + (wn * WCOLS + frow) * LROW;
    const int fko8[2] = {((0 + fh) ^ ((frow >> 2) & 3)) * 4, ((2 + fh) ^ ((frow >> 2) & 3)) * 4};   // bf16: chunk 2 m + fh of MFMA m
    const int total_steps = ntl * (kt_hi - kt_lo);
    int kt_in = 0, m_cur = m0;             // K-step inside the current m-tile, first row of that tile
    for (int kt = 0; kt < total_steps; ++kt) {
        const int buf = kt & 1;
        Ast_ = As + (buf ^ 1) * BM * LROW + lrow * LROW + st_chunk;
        Bst_ = Bs + (buf ^ 1) * BN * LROW + lrow * LROW + st_chunk;
        const float* Ab = Afr + buf * BM * LROW;
        const float* Bb = Bfr + buf * BN * LROW;
        if constexpr (BF16) {
            // 2 MFMAs of K = 16 per output tile and K-step; the staging slices go behind each half's MT*NT MFMAs
            bf16x8 ha[2][MT], hb[2][NT];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int i = 0; i < MT; ++i) ha[m][i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * LROW + fko8[m]);
#pragma unroll
                for (int j = 0; j < NT; ++j) hb[m][j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * LROW + fko8[m]);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[m][i], hb[m][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int sl = 0; sl < SLICES; ++sl)
                    if (sl >= m * 9 && sl < (m + 1) * 9) { CONV_SLICE(sl) }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        float4 fa[2][MT], fb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LROW + fko[0]);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LROW + fko[0]);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int cur = kc & 1;
            if (kc < 3) {      // fragments of the next k-chunk, in flight under this chunk's MFMAs
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[cur ^ 1][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LROW + fko[kc < 3 ? kc + 1 : 0]);
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[cur ^ 1][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LROW + fko[kc < 3 ? kc + 1 : 0]);
            }
            // element-major order: consecutive MFMAs hit different accumulators, so a staging instruction slipped between
            // them never sits on a dependent-accumulator chain (64-cycle latency == issue interval leaves no slack)
#pragma unroll
            for (int el = 0; el < 4; ++el) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float av = el == 0 ? fa[cur][i].x : el == 1 ? fa[cur][i].y : el == 2 ? fa[cur][i].z : fa[cur][i].w;
                        const float bv = el == 0 ? fb[cur][j].x : el == 1 ? fb[cur][j].y : el == 2 ? fb[cur][j].z : fb[cur][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                if (MT * NT == 4 || (el & (4 / (MT * NT) - 1)) == 4 / (MT * NT) - 1) {     // one group = 4 MFMAs
                    const int g = MT * NT == 4 ? kc * 4 + el : MT * NT == 2 ? kc * 2 + (el >> 1) : kc;
#pragma unroll
                    for (int sl = 0; sl < SLICES; ++sl)
                        if (sl >= g * SLICES / GROUPS && sl < (g + 1) * SLICES / GROUPS) { CONV_SLICE(sl) }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        }
        if (a.tpb > 1) {           // persistent plane-GEMM: a finished m-tile is stored from inside the pipeline (dense rows)
            if (++kt_in == a.KT) {
                kt_in = 0;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int co = n0 + wn * WCOLS + j * 32 + (lane & 31);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int m = m_cur + wm * WROWS + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                            if (co < a.Cout) a.y[(size_t)m * a.Cout + co] = acc[i][j][e];
                            acc[i][j][e] = 0.f;
                        }
                }
                m_cur += BM;
            }
        }
        __syncthreads();
    }
    if (a.tpb > 1) return;
#undef CONV_PREP
#undef CONV_LOAD_A
#undef CONV_LOAD_B
#undef CONV_STORE_A
#undef CONV_STORE_B
#undef CONV_SLICE
#undef CONV_LOAD_ALL
#undef CONV_STORE_ALL

    // ---- epilogue ------------------------------------------------------------------------------------------------
    // C/D map of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const int ccol = lane & 31, rquad = 4 * (lane >> 5);
    if (a.ksplit > 1) {            // raw partial sums -> slab[split][m][co]; the reduce kernel does the rest
        float* slab = a.slab + (size_t)split * a.M * a.Cout;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = n0 + wn * WCOLS + j * 32 + ccol;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * WROWS + i * 32 + (e & 3) + 8 * (e >> 2) + rquad;
                    if (m < a.M && co < a.Cout) slab[(size_t)m * a.Cout + co] = acc[i][j][e];
                }
        }
        return;
    }
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = n0 + wn * WCOLS + j * 32 + ccol;
        const bool cok = co < a.Cout;
        const float bv = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WROWS + i * 32 + (e & 3) + 8 * (e >> 2) + rquad;
                if (m < a.M && cok) {
                    float v = acc[i][j][e] + bv;
                    float* dst;
                    if (a.dense) dst = a.y + (size_t)m * a.Cout + co;      // plain NHWC: no per-element division
                    else {
                        const int n = m / a.HoWo;
                        const int rem = m - n * a.HoWo;
                        const int hh = rem / a.Wo, ww = rem - hh * a.Wo;
                        dst = a.y + (size_t)n * a.y_batch_stride + (size_t)((a.oh0 + a.ostep * hh) * a.OWf + a.ow0 + a.ostep * ww) * a.Cout + co;
                    }
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
        float* red = lds;                      // [2][BN][2]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
            float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            if (lane < 32) {
                const int cl = wn * WCOLS + j * 32 + lane;
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

// Second pass of a split-K launch: y = sum over slabs (fixed order) + bias with the epilogue's addressing (dense / strided /
// accumulate) and its BatchNorm partial sums per BM-row block.  grid (MB, ceil(Cout / 64)), 1,024 threads = 64 channels x 16 row
// lanes: a thread owns BM / 16 rows and has all their slab values in flight at once (the pass is a latency chain otherwise).
__global__ __launch_bounds__(1024) void conv_splitk_reduce_kernel(ConvArgs a, int BM) {
    __shared__ float red[16][64][2];
    const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int co = blockIdx.y * 64 + col;
    const bool cok = co < a.Cout;
    const int mb = blockIdx.x, m_lo = mb * BM, m_hi = min(a.M, m_lo + BM);
    const float bv = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
    const size_t slab_stride = (size_t)a.M * a.Cout;
    float t1 = 0.f, t2 = 0.f;
    if (cok) {
        for (int m = m_lo + rl; m < m_hi; m += 16) {
            const float* sp = a.slab + (size_t)m * a.Cout + co;
            float part[8];
#pragma unroll
            for (int z = 0; z < 8; ++z) part[z] = z < a.ksplit ? sp[z * slab_stride] : 0.f;
            float v = part[0];
#pragma unroll
            for (int z = 1; z < 8; ++z) v += part[z];          // fixed order; the unused slots add +0
            v += bv;
            float* dst;
            if (a.dense) dst = a.y + (size_t)m * a.Cout + co;
            else {
                const int n = m / a.HoWo;
                const int rem = m - n * a.HoWo;
                const int hh = rem / a.Wo, ww = rem - hh * a.Wo;
                dst = a.y + (size_t)n * a.y_batch_stride + (size_t)((a.oh0 + a.ostep * hh) * a.OWf + a.ow0 + a.ostep * ww) * a.Cout + co;
            }
            if (a.accumulate) v += *dst;
            *dst = v;
            t1 += v;
            t2 += v * v;
        }
    }
    if (a.stats != nullptr) {
        red[rl][col][0] = t1; red[rl][col][1] = t2;
        __syncthreads();
        if (rl == 0 && cok) {
            float2 o = make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 16; ++k) { o.x += red[k][col][0]; o.y += red[k][col][1]; }
            *reinterpret_cast<float2*>(a.stats + ((size_t)mb * a.Cout + co) * 2) = o;
        }
    }
}

// launches whose tile grid is this small are latency chains of K-steps: split K over blocks (plan_split)
static int plan_split(int MB, int NB, int KT, int64_t M, int Cout, int64_t ws_bytes, int& kt_per) {
    kt_per = KT;
    const int blocks = MB * NB;
    const int cus = ssdhip_cu_count();
    if (blocks > cus * 3 / 4 || KT < 8) return 1;
    int ks = cus * 3 / 2 / blocks;    // aim at ~1.5 blocks per CU
    if (ks > 8) ks = 8;
    if (ks > KT / 2) ks = KT / 2;
    while (ks > 1 && (int64_t)ks * M * Cout * 4 > ws_bytes) --ks;
    if (ks < 2) return 1;
    kt_per = (KT + ks - 1) / ks;
    return (KT + kt_per - 1) / kt_per;
}

// OIHW -> packed rows.  mode 0: row = co, inner = ci;  mode 1: row = ci, inner = co.  k order as the loader walks it.
__device__ __forceinline__ void pack_weight_body(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int R, int S,
                                                 int rows_pad, int inner_pad, int KP, int mode) {
    const int64_t total = (int64_t)rows_pad * KP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / KP), k = (int)(i - (int64_t)row * KP);
        int tap, inner;
        if (inner_pad < 32) { tap = k / inner_pad; inner = k - tap * inner_pad; }          // padded RGB: k = tap * 4 + c
        else {                                                                              // k = (cblk * taps + tap) * 32 + c % 32
            const int taps = R * S, cb = k / (taps * 32), rem = k - cb * taps * 32;
            tap = rem >> 5; inner = cb * 32 + (rem & 31);
            if (inner >= inner_pad) tap = taps;                                             // K padding beyond the last block
        }
        float v = 0.f;
        if (tap < R * S) {
            const int r = tap / S, s = tap - r * S;
            const int co = mode == 0 ? row : inner, ci = mode == 0 ? inner : row;
            if (co < Cout && ci < Cin) v = w[(((size_t)co * Cin + ci) * R + r) * S + s];
        }
        packed[i] = v;
    }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int R, int S,
                                   int rows_pad, int inner_pad, int KP, int mode) {
    pack_weight_body(w, packed, Cout, Cin, R, S, rows_pad, inner_pad, KP, mode);
}

// every layout of one optimizer step in ONE launch: blockIdx.y = job (the trainable tail repacks ~26 weight tensors per step, 5 us each)
__global__ void pack_weights_batched_kernel(const ssdhip_pack_job* __restrict__ jobs) {
    const ssdhip_pack_job j = jobs[blockIdx.y];
    pack_weight_body(j.w, j.packed, j.Cout, j.Cin, j.R, j.S, j.rows_pad, j.inner_pad, j.KP, j.mode);
}

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// tiles-per-block rule of the one-slot-per-block plane GEMM (see ssdhip_conv2d_nhwc)
static int plane_tpb(int64_t tiles, int Cin) {
    int tpb = (int)(tiles / ((int64_t)ssdhip_cu_count() * 6));
    if (tpb < 1 || tpb == 2) tpb = 1;
    if (tpb > 8) tpb = 8;
    if (Cin >= 256 && tpb > 3) tpb = 3;
    return tpb;
}

// Work-queue kernel or one slot per block?  In tile-times on 2 x CUs resident slots: blocks of tpb tiles run in whole rounds (3,600 one-tile
// blocks on 512 slots: 8; 6,552 tiles in three-tile blocks: 15), the queue pays the exact share (7.03; 12.8) plus ~5 % per item and a
// start-up term.  Measured (scripts/plane_queue_probe.py): conv_3_2 557 -> 481 us, conv_4_1 318 -> 296, conv_4_2 508 -> 497; where the rounds
// already fit (conv_6_1: 3.94 of 4; det_4_3) the slot form stays 7 % ahead.
static bool plane_queue_wins(int64_t MB, int NB, int Cin) {
    const int64_t tiles = MB * NB, slots = 2 * (int64_t)ssdhip_cu_count();
    if (tiles < 4 * slots) return false;        // short launches: the queue's start-up and split-piece hand-off cost more than a partial round (1,188 tiles: 107 vs 92 us)
    const int tpb = plane_tpb(tiles, Cin);
    const int64_t blocks = ((MB + tpb - 1) / tpb) * NB;
    const double slot_cost = (double)((blocks + slots - 1) / slots) * tpb;
    const double queue_cost = (double)tiles / (double)slots * 1.05 + 0.25;
    return queue_cost < 0.98 * slot_cost;
}

int choose_tile(const ssdhip_conv_params* p, int& BM, int& BN) {
    if (p->tile % 1000000) { BM = (p->tile / 1000) % 1000; BN = p->tile % 1000; }
    else {
        const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
        BN = (p->Cout <= 64 || cdiv(p->Cout, 64) * 64 < cdiv(p->Cout, 128) * 128) ? 64 : 128;
        BM = 128;
        // small grids: smaller tiles so the chip (256 CUs x >= 2 resident blocks) sees enough tiles per round
        // (thresholds measured with scripts/conv_bench.py)
        if ((int64_t)cdiv(M, 128) * cdiv(p->Cout, BN) < 1024) BM = 64;
        if (BM == 64 && BN == 128 && (int64_t)cdiv(M, 64) * cdiv(p->Cout, 128) < 512) BN = 64;
    }
    if (!((BM == 128 || BM == 64) && (BN == 128 || BN == 64))) return SSDHIP_EINVAL;
    return SSDHIP_OK;
}

template <int BM, int BN, int KS>
void launch_conv(const ConvArgs& a, int mode, bool xform, bool bf16, hipStream_t st) {
    dim3 grid(cdiv(a.MB, a.tpb) * a.NB * a.ksplit), block(256);
    if (bf16) {          // never the padded-RGB layer, never a plane GEMM (checked by the caller)
        if (mode == 0) {
            if (xform) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, KS, 0, true, false, 0, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, KS, 0, false, false, 0, true>), grid, block, 0, st, a);
        } else {
            hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, KS, 1, false, false, 0, true>), grid, block, 0, st, a);
        }
    } else if (a.Cin == 4) {
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
    SSDHIP_REQUIRE(elems_below(1ll << 30, p->N, p->Hi, p->Wi, p->Cin) && elems_below(1ll << 31, p->N, p->Ho, p->Wo),
                   "conv2d: input tensor must stay below 4 GiB (32-bit byte offsets)");
    if (p->mode == 0) {
        SSDHIP_REQUIRE(p->Ho == (p->Hi + 2 * p->pad - p->R) / p->stride + 1 && p->Wo == (p->Wi + 2 * p->pad - p->S) / p->stride + 1,
                       "conv2d: output size does not match input/stride/pad");
    }
#ifndef SSDHIP_PROBES
    // tile codes >= 1,000,000 select A/B and timing-only ablation kernels (some return wrong results by design): they exist only in
    // libssdhip_probes.so (make probes; used by scripts/*_probe.py, scripts/*_bench.py), never in the product library
    SSDHIP_REQUIRE(p->tile >= 0 && p->tile < 1000000, "conv2d: tile must be 0 (auto) or one of 128128/128064/64128/64064");
#endif
    SSDHIP_REQUIRE(p->compute == 0 || p->compute == 1 || p->compute == 2, "conv2d: compute 0 (fp32 MFMA), 1 (bf16 MFMA, fp32 accumulate) or 2 (split-bf16 plane GEMM)");
    const bool bf16 = p->compute == 1;
    SSDHIP_REQUIRE(!bf16 || (p->Cin != 4 && p->rows_per_img == 0), "conv2d: bf16 compute is for the plain conv layers");
    SSDHIP_REQUIRE(p->compute != 2 || (p->rows_per_img > 0 && p->tile == 128128 && p->KP == p->Cin && p->Cin % 32 == 0 && !p->bias && !p->stats && !p->accumulate && !p->in_scale),
                   "conv2d: compute 2 (fp32 operands as three bf16 terms, six products) is for 128 x 128 plane GEMMs");
    int BM, BN;
    SSDHIP_REQUIRE(choose_tile(p, BM, BN) == SSDHIP_OK, "conv2d: tile must be one of 128128/128064/64128/64064");
    SSDHIP_REQUIRE(p->rows_per_img == 0 || (p->rows_per_img % BM == 0 && p->mode == 0 && p->w_img_stride > 0),
                   "conv2d: rows_per_img must be a multiple of the m-tile (batched GEMM use)");
    ConvArgs a;
    a.x = p->x; a.w = p->w; a.bias = p->bias; a.in_scale = p->in_scale; a.in_shift = p->in_shift; a.y = p->y; a.stats = p->stats;
    a.y_batch_stride = p->y_batch_stride;
    a.M = p->N * p->Ho * p->Wo; a.Hi = p->Hi; a.Wi = p->Wi; a.Cin = p->Cin; a.cpt = p->Cin >= BK ? p->Cin / BK : 1;
    a.Ho = p->Ho; a.Wo = p->Wo; a.HoWo = p->Ho * p->Wo; a.Cout = p->Cout; a.taps = p->R * p->S;
    a.stride = p->stride; a.lgStride = ilog2(p->stride); a.pad = p->pad; a.KP = p->KP; a.KT = p->KP / BK;
    a.MB = cdiv(a.M, BM); a.NB = cdiv(p->Cout, BN); a.accumulate = p->accumulate;
    a.dense = p->y_batch_stride == (int64_t)p->Ho * p->Wo * p->Cout;
    a.w_img_stride = p->w_img_stride; a.rows_per_img = p->rows_per_img;
    a.tpb = 1;
    a.ksplit = 1; a.kt_per = a.KT; a.slab = p->ws;
    if (p->rows_per_img > 0 && !p->bias && !p->stats && !p->accumulate && p->R == 1 && !p->in_scale && a.M % BM == 0 &&
        p->y_batch_stride == (int64_t)p->Ho * p->Wo * p->Cout) {
        // short-K plane GEMMs: amortise pipeline fill/drain over several m-tiles, but keep >= ~6 blocks per CU.  Two tiles per block
        // measured worse than one AND than three on every shape (3,600-tile GEMMs of the 36-plane F(4x4) layers: 537 vs 505 / 507 us on
        // conv_4_2, 287 vs 277 / 276 on conv_4_1; scripts/w4_bench.py with tile codes 40..49 of the probe build), so 2 becomes 1
        // (measured on conv_4_1 / conv_4_2 (K = 256 / 512, 3,600 tiles) and conv_3_2 (K = 256, 6,552 tiles) of SSD300 at batch 32 only;
        // other shapes inherit the rule unmeasured)
        const int64_t tiles = (int64_t)a.MB * a.NB;
        a.tpb = plane_tpb(tiles, p->Cin);                // K >= 256 (conv_3_2 / conv_3_3 F(4x4) planes, 6,552 tiles): 3 tiles 532-537 us, 4 tiles 544-553 (three runs)
        // plain row-major planes: stage with the LDS-DMA engine (gemm_planes.hip) unless tile code 9xxxxxx asks for the
        // register-staged kernel (A/B comparison in scripts/wino_bench.py)
        int probe = 0;
#ifdef SSDHIP_PROBES
        const int code = p->tile / 1000000;        // 6/7: timing-only ablations; 8: the 4-wave block (A/B); 4x: tpb override (scripts/plane_gemm_probe.py)
        if (code >= 40 && code < 50) a.tpb = code - 40 + 1;
        probe = code == 7 ? 1 : code == 6 ? 2 : code == 8 ? 4 : code == 3 ? 3 : code == 9 ? 9 : 0;
#endif
        // 128 x 128 tiles at K >= 256 with a caller workspace: the persistent work-queue kernel (no tile quantisation, one pipeline fill per
        // launch; gemm_planes.hip).  Without a workspace (and for every other tile): the one-launch-slot-per-block kernel below.
        if (p->compute == 2) {
            static const bool nw8 = getenv("SSDHIP_SPLIT_NW") && atoi(getenv("SSDHIP_SPLIT_NW")) == 8;      // A/B of the two wave layouts (probe scripts)
            int rc = ssdhip_gemm_planes_dma(p->x, p->w, p->y, a.M, p->Cin, p->Cout, p->rows_per_img, p->w_img_stride, BM, BN, a.tpb, (hipStream_t)stream, nw8 ? 101 : 100);
            if (rc != SSDHIP_OK) { ssdhip_set_error("conv2d: split-bf16 plane GEMM needs 128 x 128 tiles"); return rc; }
            SSDHIP_CHECK_LAUNCH();
            return SSDHIP_OK;
        }
        if (probe == 0 && p->tile == 128128 && BM == 128 && BN == 128 && p->KP == p->Cin && p->Cin % BK == 0 && p->Cin >= 256 && p->ws &&
            p->ws_bytes >= ssdhip_gemm_planes_queue_ws_bytes() && plane_queue_wins(a.MB, a.NB, p->Cin)) {
            int rc = ssdhip_gemm_planes_queue(p->x, p->w, p->y, a.M, p->Cin, p->Cout, p->rows_per_img, p->w_img_stride, p->ws, p->max_cus, (hipStream_t)stream);
            if (rc != SSDHIP_OK) { ssdhip_set_error("conv2d: plane GEMM work-queue kernel not available on this device"); return rc; }
            SSDHIP_CHECK_LAUNCH();
            return SSDHIP_OK;
        }
        if (probe != 9 && !(BM == 64 && BN == 64) && p->KP == p->Cin && p->Cin % BK == 0 && (p->Cin >= 256 || (p->Cin == 128 && BM == 128 && BN == 128))) {   // other K = 128 tiles: register staging measured faster
            int rc = ssdhip_gemm_planes_dma(p->x, p->w, p->y, a.M, p->Cin, p->Cout, p->rows_per_img, p->w_img_stride, BM, BN, a.tpb,
                                            (hipStream_t)stream, probe);
            if (rc != SSDHIP_OK) { ssdhip_set_error("conv2d: plane GEMM tile not compiled"); return rc; }
            SSDHIP_CHECK_LAUNCH();
            return SSDHIP_OK;
        }
    }
    a.OWf = p->Wo; a.oh0 = 0; a.ow0 = 0; a.ostep = 1; a.r0 = 0; a.s0 = 0; a.rstep = 1; a.nr = p->R; a.ns = p->S;
    hipStream_t st = (hipStream_t)stream;
    const bool xf = p->in_scale != nullptr;
#ifdef SSDHIP_PROBES
    if (p->tile >= 1000000 && p->tile / 1000000 != 9) {   // timing-only ablations of the 128x128 3x3 kernel (scripts/conv_bench.py)
        SSDHIP_REQUIRE(BM == 128 && BN == 128 && p->R == 3 && p->mode == 0 && p->Cin != 4 && xf, "conv2d: ablation tile");
        dim3 grid(a.MB * a.NB), block(256);
        if (p->tile / 1000000 == 1) hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 3, 0, true, false, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 3, 0, true, false, 2>), grid, block, 0, st, a);
        SSDHIP_CHECK_LAUNCH();
        return SSDHIP_OK;
    }
#endif
    const int key = BM * 1000 + BN;
#define SSDHIP_CONV_CASE(bm, bn)                                            \
    case bm * 1000 + bn:                                                    \
        if (p->R == 3) launch_conv<bm, bn, 3>(a, p->mode, xf, bf16, st);    \
        else launch_conv<bm, bn, 1>(a, p->mode, xf, bf16, st);              \
        break;
    // stride-2 data gradient: one launch per output-parity class, each with only the taps that land on a dy pixel
    // (tiny layers stay on one masked launch: four launches cost more than the wasted MFMAs; measured at M = 3200)
    const int classes = (p->mode == 1 && p->stride == 2 && p->R == 3 && (int64_t)p->N * p->Ho * p->Wo >= 8192) ? 4 : 1;
    for (int cls = 0; cls < classes; ++cls) {
        if (classes == 4) {
            const int ph = cls >> 1, pw = cls & 1;                       // parity of (ho + pad), (wo + pad)
            const int h0 = (ph + p->pad) & 1, w0 = (pw + p->pad) & 1;     // first output row / column of the class
            const int Hc = (p->Ho - h0 + 1) / 2, Wc = (p->Wo - w0 + 1) / 2;
            if (Hc <= 0 || Wc <= 0) continue;
            a.Ho = Hc; a.Wo = Wc; a.HoWo = Hc * Wc; a.M = p->N * Hc * Wc;
            a.oh0 = h0; a.ow0 = w0; a.ostep = 2; a.OWf = p->Wo; a.dense = 0;
            a.rstep = 2; a.r0 = ph; a.nr = ph == 0 ? 2 : 1; a.s0 = pw; a.ns = pw == 0 ? 2 : 1;
            a.KT = a.cpt * a.nr * a.ns;
            a.MB = cdiv(a.M, BM);
        }
        a.ksplit = (p->ws && p->Cin != 4 && p->rows_per_img == 0 && a.tpb == 1) ? plan_split(a.MB, a.NB, a.KT, a.M, a.Cout, p->ws_bytes, a.kt_per) : 1;
        switch (key) {
            SSDHIP_CONV_CASE(128, 128)
            SSDHIP_CONV_CASE(128, 64)
            SSDHIP_CONV_CASE(64, 128)
            SSDHIP_CONV_CASE(64, 64)
            default: ssdhip_set_error("conv2d: unsupported tile"); return SSDHIP_ENOSUP;
        }
        SSDHIP_CHECK_LAUNCH();
        if (a.ksplit > 1) {
            hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(a.MB, cdiv(a.Cout, 64)), dim3(1024), 0, st, a, BM);
            SSDHIP_CHECK_LAUNCH();
        }
    }
#undef SSDHIP_CONV_CASE
    return SSDHIP_OK;
}

int64_t ssdhip_conv2d_ws_bytes(const ssdhip_conv_params* p) {
    // bytes of split-K workspace this launch can use (0: the launch does not split).  Sized for the largest split plan_split picks.
    int BM, BN;
    if (!p || choose_tile(p, BM, BN) != SSDHIP_OK || p->Cin == 4 || p->KP <= 0) return 0;
    if (p->rows_per_img != 0) {
        // plane GEMMs (Winograd domain): 128 x 128 tiles at K >= 256 run on the work-queue kernel when given this workspace (its first 4 KiB
        // zero before the first launch; the kernel leaves them zero)
        const bool q = p->tile == 128128 && p->KP == p->Cin && p->Cin % BK == 0 && p->Cin >= 256 && p->R == 1 && !p->bias && !p->stats && !p->accumulate &&
                       !p->in_scale && ((int64_t)p->N * p->Ho * p->Wo) % 128 == 0 && p->y_batch_stride == (int64_t)p->Ho * p->Wo * p->Cout;
        return q && plane_queue_wins(((int64_t)p->N * p->Ho * p->Wo) / 128, cdiv(p->Cout, 128), p->Cin) ? (int64_t)ssdhip_gemm_planes_queue_ws_bytes() : 0;
    }
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    if (p->mode == 1 && p->stride == 2 && p->R == 3 && M >= 8192) return 0;       // parity-class launches: large grids
    int kt_per;
    const int ks = plan_split(cdiv(M, BM), cdiv(p->Cout, BN), p->KP / BK, M, p->Cout, (int64_t)1 << 40, kt_per);
    return ks > 1 ? (int64_t)ks * M * p->Cout * 4 : 0;
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

int ssdhip_pack_conv_weights(const ssdhip_pack_job* jobs_device, int n_jobs, int64_t max_elements, void* stream) {
    SSDHIP_REQUIRE(jobs_device && n_jobs > 0 && n_jobs <= 65535 && max_elements > 0, "pack_conv_weights: bad arguments");
    int blocks = (int)((max_elements + 255) / 256);
    if (blocks > 512) blocks = 512;                      // per job; grid-stride inside
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(blocks, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
