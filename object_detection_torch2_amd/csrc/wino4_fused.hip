// Two-pass Winograd F(4x4, 3x3) convolution for 3x3 / stride 1 / pad 1 layers whose maps are too large for the three-pass form of
// winograd4.hip (reference op: nn.Conv2d forward, model/vgg16.py:33).  OPT-IN (engine: SSD_EARLY_F4=1); the default route of the 300 /
// 150 / 75 px layers stays the one-kernel F(2x2) form of wino_fused.hip.
//
// Why two passes.  F(4x4) needs 2.25 MFMA flops per output and channel pair where F(2x2) needs 4, but the three-pass form writes and
// re-reads the Winograd domain on BOTH sides of the GEMM (V and M: 2.25x the activation bytes each way), which on the 150 px maps is
// more HBM time than the MFMAs it saves (every backbone layer on it: 10.9 ms per step against 9.6, profiles/r04_i_*).  A one-kernel
// F(4x4) would put the input transform's VALU work beside the fp32 MFMAs, where it costs matrix cycles one for one (profiles/r04_h_*).
// This form keeps the input transform as its own HBM-bound pass and folds the OUTPUT side into the GEMM:
//
//   pass A  wino4f_input_kernel   V'[tile block][K-step][xi][32 tiles][4 c] = (B^T d B)[xi]      (winograd4_input_kernel's arithmetic,
//                                 producer's BatchNorm.ReLU on load, zero padding after it; K-step-blocked layout: the image a block of
//                                 pass B needs for one K-step is 18 KiB of contiguous memory)
//   pass B  wino4f_gemm_kernel    one block = 32 tiles x 64 output channels x ALL 36 planes in accumulators (v_mfma_f32_16x16x4_f32 keeps
//                                 the output channel on the lane and the tile on the register index, so A^T m A is register-local: M never
//                                 exists in memory); epilogue: output transform, bias, NHWC store (or the 2x2 pooling windows' maxima /
//                                 minima, as wino_fused.hip stores them), BatchNorm partial sums
//
// Pass B's shape follows from the accumulators: 36 planes x 4 registers per 16 x 16 MFMA tile = 144 per wave, two waves per SIMD.  8 waves
// as 2 (tiles) x 4 (channels).  Operands per K-step of 4 input channels: A image 36 x 32 x 16 B = 18 KiB (streams from HBM, ring of 3
// stages, requested two steps ahead: a step is only ~1.1 us of MFMA time), B image 36 x 64 x 16 B = 36 KiB (pre-packed weights, L2,
// double-buffered), both by LDS-DMA in 1-KiB pieces; fragments by ds_read_b32, requested three planes ahead of their MFMA.
// Measured on the inner loop (scripts/probes/f4_fused_gemm_probe.hip, profiles/r04_j_*): 0.78-0.85 of the fp32 MFMA peak without the
// operand stream, 0.55-0.60 with it -- the item the accumulators allow needs three times the plane GEMM's operand bytes per flop.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int F4_TILES = 32, F4_CH = 64, F4_BK = 4, F4_PL = 36, F4_NA = 3, F4_DEPTH = 3;
constexpr int F4_A_BYTES = F4_PL * F4_TILES * F4_BK * 4;      // 18,432
constexpr int F4_B_BYTES = F4_PL * F4_CH * F4_BK * 4;         // 36,864
constexpr int F4_LDS = F4_NA * F4_A_BYTES + 2 * F4_B_BYTES;   // 129,024
constexpr int F4_A_PSTRIDE = F4_TILES * F4_BK * 4, F4_B_PSTRIDE = F4_CH * F4_BK * 4;

constexpr float kC97_36 = 97.0f / 36.0f, kC4_9 = 4.0f / 9.0f, kC9_4 = 2.25f, kC3_2 = 1.5f, kC2_3 = 2.0f / 3.0f;
constexpr float kC27_8 = 3.375f, kC8_27 = 8.0f / 27.0f;

// v = B^T d for one 6-vector of channel pairs (in place); same operations in the same order as winograd4.hip
__device__ __forceinline__ void f4_bt6(float2& d0, float2& d1, float2& d2, float2& d3, float2& d4, float2& d5) {
#define F4_BT6(c)                                                                                               \
    {                                                                                                           \
        const float v0 = fmaf(-kC97_36, d2.c, d0.c + d4.c);                                                     \
        const float e1 = fmaf(-kC4_9, d2.c, d4.c), o1 = fmaf(-kC2_3, d1.c, kC3_2 * d3.c);                       \
        const float e2 = fmaf(-kC9_4, d2.c, d4.c), o2 = fmaf(-kC3_2, d1.c, kC2_3 * d3.c);                       \
        const float v5 = fmaf(-kC97_36, d3.c, d1.c + d5.c);                                                     \
        d0.c = v0; d1.c = e1 + o1; d2.c = e1 - o1; d3.c = e2 + o2; d4.c = e2 - o2; d5.c = v5;                   \
    }
    F4_BT6(x) F4_BT6(y)
#undef F4_BT6
}
// y = A^T m for one 6-vector
__device__ __forceinline__ void f4_at6(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1, float& y2, float& y3) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y0 = (m0 + s12) + s34;
    y1 = fmaf(kC3_2, d12, kC2_3 * d34);
    y2 = fmaf(kC9_4, s12, kC4_9 * s34);
    y3 = fmaf(kC27_8, d12, kC8_27 * d34) + m5;
}

// ---- pass A: block = 32 consecutive tiles x 32 channels = 8 waves of 8 tiles x 16 channels (lane = (tile, channel pair)): a load
// instruction reads 64 contiguous bytes of 8 pixels (the other half of each 128-B line goes to the neighbouring wave of the same block),
// a store instruction writes four whole 128-B lines (8 tiles x 16 B of each of the wave's four K-steps).  (First version: lane = one of 32
// tiles, 32 lines per load instruction: 3.2-3.7 TB/s.)
template <bool XFORM>
__global__ __launch_bounds__(512) void wino4f_input_kernel(const float2* __restrict__ x, const float2* __restrict__ scale,
                                                           const float2* __restrict__ shift, float2* __restrict__ Vb, int H, int W, int CV,
                                                           int TH, int TW, int T, int KT, int GB) {
    int bid;
    {   // an XCD walks a contiguous eighth of the blocks: neighbouring tiles (overlapping patches) and the GB channel groups of a tile block meet in one L2
        const int nb = gridDim.x, xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int tb = bid / GB, gb = bid - tb * GB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tl = 8 * (wave & 3) + (lane >> 3), cvl = lane & 7, half = cvl & 1;
    const int g = gb * 8 + 4 * (wave >> 2) + (cvl >> 1), cv = 2 * g + half;
    const int t = tb * F4_TILES + tl;
    const bool tv = t < T;
    const int tc = tv ? t : T - 1;
    const int tw = tc % TW, q = tc / TW, th = q % TH, n = q / TH;
    float2 sc = make_float2(1.f, 1.f), sh = make_float2(0.f, 0.f);
    if (XFORM) { sc = scale[cv]; sh = shift[cv]; }
    float2 d[6][6];
    const float2* xn = x + (size_t)n * H * W * CV + cv;
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
        const bool hin = tv && (unsigned)(4 * th - 1 + i) < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const bool in = hin && (unsigned)(4 * tw - 1 + j) < (unsigned)W;
            float2 v = d[i][j];
            if (XFORM) v = make_float2(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f));
            d[i][j] = in ? v : make_float2(0.f, 0.f);
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) f4_bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    // float2 index of (tile block, K-step g, plane 0, tile tl, channel pair half)
    float2* o = Vb + ((size_t)(tb * KT + g) * F4_PL * (F4_TILES * F4_BK) + tl * F4_BK + half * 2) / 2;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f4_bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
        for (int i = 0; i < 6; ++i) o[(size_t)(6 * i + j) * (F4_TILES * F4_BK / 2)] = d[i][j];
    }
}

// U (36, rows_pad, Cin) of winograd4_weight_kernel -> Upk[channel tile nb][K-step g][xi][64 channels][4 c]: the B image of (nb, g) as it lies in LDS
__global__ void wino4f_pack_kernel(const float* __restrict__ U, float* __restrict__ upk, int Cout, int Cin, int rows_pad) {
    const int64_t total = (int64_t)F4_PL * Cout * Cin;
    const int KT = Cin / F4_BK;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i & 3), ch = (int)((i >> 2) & 63);
        int64_t rest = i >> 8;
        const int xi = (int)(rest % F4_PL); rest /= F4_PL;
        const int g = (int)(rest % KT), nb = (int)(rest / KT);
        upk[i] = U[((size_t)xi * rows_pad + nb * F4_CH + ch) * Cin + g * F4_BK + k];
    }
}

struct F4Args {
    const char* vb;       // pass A's output
    const char* upk;
    const float* bias;
    float* z;
    float* zmax;
    float* zmin;
    float* stats;         // (tile blocks, Cout, 2) or null
    int H, W, Cout, KT, NB, TB, TH, TW, T;
};

// ---- pass B.  One item (tile block, channel tile) per block; the NB channel tiles of a tile block get neighbouring block ids inside one
// XCD's contiguous eighth of the items, so its A images cross the fabric once.  (A persistent form -- one block per CU walking its items as
// one K-step stream, operand requests running ahead across item boundaries -- measured SLOWER on every layer, conv_2_1 0.51 ms against 0.41:
// its blocks stay in lockstep, so all 256 CUs store their 128-KiB output tiles in the same few microseconds and then all compute; blocks
// that start and end on their own spread the stores.)
template <bool POOL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino4f_gemm_kernel(F4Args a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    int bid;
    {
        const int nbk = gridDim.x, xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q = nbk >> 3, r = nbk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int tb = bid / a.NB, nb = bid - tb * a.NB;
    const int wm = wave >> 2, wn = wave & 3, li = lane & 15, kq = lane >> 4;
    const int KT = a.KT;
    const unsigned a_rd = lds0 + (unsigned)(((16 * wm + li) * F4_BK + kq) * 4);
    const unsigned b_rd = lds0 + (unsigned)(F4_NA * F4_A_BYTES) + (unsigned)(((16 * wn + li) * F4_BK + kq) * 4);
    const char* ablk = a.vb + (size_t)tb * KT * F4_A_BYTES;
    const char* bblk = a.upk + (size_t)nb * KT * F4_B_BYTES;
    constexpr int APIECES = F4_A_BYTES / 1024, BPIECES = F4_B_BYTES / 1024;       // 18, 36
    auto issue_a = [&](int step, int slot) {       // A image of K-step `step` (past the end: the last one again, never consumed) -> ring slot
        const char* as = ablk + (size_t)(step < KT ? step : KT - 1) * F4_A_BYTES;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int piece = wave + 8 * k;
            if (piece < APIECES) __builtin_amdgcn_global_load_lds((gptr_t)(as + piece * 1024 + lane * 16), (lptr_t)(lds + slot * F4_A_BYTES + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_b = [&](int step) {
        const char* bs = bblk + (size_t)(step < KT ? step : KT - 1) * F4_B_BYTES;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int piece = wave + 8 * k;
            if (piece < BPIECES) __builtin_amdgcn_global_load_lds((gptr_t)(bs + piece * 1024 + lane * 16), (lptr_t)(lds + F4_NA * F4_A_BYTES + (step & 1) * F4_B_BYTES + piece * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[F4_PL];
#pragma unroll
    for (int p = 0; p < F4_PL; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    issue_b(0);
    issue_a(0, 0);
    issue_a(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int slot = 0;                 // ring slot of K-step g
#pragma clang loop unroll(disable)
    for (int g = 0; g < KT; ++g) {
        const unsigned ab = a_rd + (unsigned)(slot * F4_A_BYTES), bb = b_rd + (unsigned)((g & 1) * F4_B_BYTES);
        constexpr int D = F4_DEPTH, R = D + 1;
        float fa[R], fb[R];
#define F4_REQ(P)                                                                                                     \
    do {                                                                                                               \
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[(P) % R]) : "v"(ab), "n"((P) * F4_A_PSTRIDE));            \
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[(P) % R]) : "v"(bb), "n"((P) * F4_B_PSTRIDE));            \
    } while (0)
#pragma unroll
        for (int p = 0; p < D; ++p) F4_REQ(p);
        // the ring slot of step g - 1 and the B buffer of step g - 1 are free: their readers passed the last barrier
        issue_b(g + 1);
        issue_a(g + 2, slot == 0 ? F4_NA - 1 : slot - 1);
#pragma unroll
        for (int p = 0; p < F4_PL; ++p) {
            const int r = p % R;
            if (p + D < F4_PL) F4_REQ(p + D);
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[r]), "+v"(fb[r]) : "n"((p + D < F4_PL ? D : F4_PL - 1 - p) * 2));
            __builtin_amdgcn_sched_barrier(0);
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r], fb[r], acc[p], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef F4_REQ
        // B of step g + 1 and A of step g + 1 (requested one step ago) must have landed; this step's A request -- the newest: 3 pieces on
        // waves 0 and 1, 2 on the others -- may stay in flight
        if (wave < 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot == F4_NA - 1 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the requests past the end: nothing may land in LDS behind the epilogue's back
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: register e of lane (j = lane & 15, r4 = lane >> 4) = tile 16 wm + 4 r4 + e, output channel 16 wn + j
    const int co = nb * F4_CH + 16 * wn + li;
    const float bv = a.bias ? a.bias[co] : 0.f;
    const int H = a.H, W = a.W, Cout = a.Cout;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int t = tb * F4_TILES + 16 * wm + 4 * kq + e;
        float r[4][6];                 // A^T m, column by column (winograd4_output_kernel's order)
#pragma unroll
        for (int j = 0; j < 6; ++j)
            f4_at6(acc[j][e], acc[6 + j][e], acc[12 + j][e], acc[18 + j][e], acc[24 + j][e], acc[30 + j][e], r[0][j], r[1][j], r[2][j], r[3][j]);
        if (t >= a.T) continue;
        const int tw = t % a.TW, q = t / a.TW, th = q % a.TH, n = q / a.TH;
        float o[4][4];
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            f4_at6(r[aa][0], r[aa][1], r[aa][2], r[aa][3], r[aa][4], r[aa][5], o[aa][0], o[aa][1], o[aa][2], o[aa][3]);
#pragma unroll
            for (int b = 0; b < 4; ++b) o[aa][b] += bv;
        }
        if (POOL) {
            // even H, W: a tile holds up to 2 x 2 whole pooling windows; store their maxima / minima at the pooled pixel (see wino_fused.hip)
            const int PW = W >> 1;
#pragma unroll
            for (int wa = 0; wa < 2; ++wa)
#pragma unroll
                for (int wb = 0; wb < 2; ++wb) {
                    if (4 * th + 2 * wa < H && 4 * tw + 2 * wb < W) {
                        const float v0 = o[2 * wa][2 * wb], v1 = o[2 * wa][2 * wb + 1], v2 = o[2 * wa + 1][2 * wb], v3 = o[2 * wa + 1][2 * wb + 1];
                        const size_t pix = ((size_t)n * (H >> 1) + 2 * th + wa) * PW + 2 * tw + wb;
                        a.zmax[pix * Cout + co] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                        a.zmin[pix * Cout + co] = fminf(fminf(v0, v1), fminf(v2, v3));
                        s1 += (v0 + v1) + (v2 + v3);
                        s2 = fmaf(v0, v0, s2); s2 = fmaf(v1, v1, s2); s2 = fmaf(v2, v2, s2); s2 = fmaf(v3, v3, s2);
                    }
                }
        } else {
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const int h = 4 * th + aa;
                if (h < H) {
                    float* dst = a.z + ((size_t)(n * H + h) * W + 4 * tw) * Cout + co;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (4 * tw + b < W) {
                            const float v = o[aa][b];
                            dst[(size_t)b * Cout] = v;
                            s1 += v; s2 = fmaf(v, v, s2);
                        }
                }
            }
        }
    }
    if (a.stats != nullptr) {
        // lanes j, j + 16, j + 32, j + 48 hold the same channel (4 tile groups); then the two tile halves (wm) through LDS, fixed order
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        float* red = reinterpret_cast<float*>(lds);
        if (lane < 16) { red[(wm * F4_CH + 16 * wn + lane) * 2] = s1; red[(wm * F4_CH + 16 * wn + lane) * 2 + 1] = s2; }
        // explicit wait + barrier: __syncthreads() would also wait for vmcnt(0), i.e. for this block's output stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid < F4_CH) {
            float* o = a.stats + ((size_t)tb * Cout + nb * F4_CH + tid) * 2;
            o[0] = red[tid * 2] + red[(F4_CH + tid) * 2];
            o[1] = red[tid * 2 + 1] + red[(F4_CH + tid) * 2 + 1];
        }
    }
}

}  // namespace

extern "C" {

int ssdhip_wino4f_tile_blocks(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return SSDHIP_EINVAL;
    const int64_t T = (int64_t)N * ((H + 3) / 4) * ((W + 3) / 4);
    if (T >= (1ll << 31) - F4_TILES) return SSDHIP_EINVAL;
    return (int)((T + F4_TILES - 1) / F4_TILES);
}

int64_t ssdhip_wino4f_v_floats(int N, int H, int W, int Cin) {
    const int tb = ssdhip_wino4f_tile_blocks(N, H, W);
    if (tb < 0 || Cin <= 0 || Cin % 32) return SSDHIP_EINVAL;
    return (int64_t)tb * F4_TILES * F4_PL * Cin;
}

int ssdhip_wino4f_pack(const float* U, float* upk, int Cout, int Cin, int rows_pad, void* stream) {
    SSDHIP_REQUIRE(U && upk && Cout > 0 && Cin > 0 && Cout % F4_CH == 0 && Cin % 32 == 0 && rows_pad >= Cout,
                   "wino4f_pack: Cout must be a multiple of 64, Cin of 32, rows_pad >= Cout");
    const int64_t total = (int64_t)F4_PL * Cout * Cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wino4f_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, upk, Cout, Cin, rows_pad);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_wino4f_input(const float* x, const float* in_scale, const float* in_shift, float* vb, int N, int H, int W, int C, void* stream) {
    SSDHIP_REQUIRE(x && vb && N > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0, "wino4f_input: C must be a multiple of 32");
    SSDHIP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "wino4f_input: in_scale and in_shift go together");
    SSDHIP_REQUIRE(elems_below(1ll << 31, N, H, W), "wino4f_input: too many pixels");
    const int TH = (H + 3) / 4, TW = (W + 3) / 4, T = N * TH * TW;
    const int TB = ssdhip_wino4f_tile_blocks(N, H, W), GB = C / 32;
    SSDHIP_REQUIRE(TB > 0 && elems_below(1ll << 31, TB, GB), "wino4f_input: bad shape");
    const dim3 grid((unsigned)(TB * GB)), block(512);
    if (in_scale)
        hipLaunchKernelGGL(wino4f_input_kernel<true>, grid, block, 0, (hipStream_t)stream, reinterpret_cast<const float2*>(x),
                           reinterpret_cast<const float2*>(in_scale), reinterpret_cast<const float2*>(in_shift), reinterpret_cast<float2*>(vb), H, W,
                           C / 2, TH, TW, T, C / F4_BK, GB);
    else
        hipLaunchKernelGGL(wino4f_input_kernel<false>, grid, block, 0, (hipStream_t)stream, reinterpret_cast<const float2*>(x), nullptr, nullptr,
                           reinterpret_cast<float2*>(vb), H, W, C / 2, TH, TW, T, C / F4_BK, GB);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

static int wino4f_gemm_launch(const float* vb, const float* upk, const float* bias, float* z, float* zmax, float* zmin, float* stats, int N,
                              int H, int W, int Cin, int Cout, void* stream) {
    SSDHIP_REQUIRE(vb && upk && (z || (zmax && zmin)) && N > 0 && H > 0 && W > 0, "wino4f_gemm: bad arguments");
    SSDHIP_REQUIRE(Cin >= 32 && Cin % 32 == 0 && Cout > 0 && Cout % F4_CH == 0, "wino4f_gemm: Cin must be a multiple of 32 and Cout of 64");
    SSDHIP_REQUIRE(elems_below(1ll << 31, N, H, W), "wino4f_gemm: too many pixels");
    F4Args a;
    a.vb = reinterpret_cast<const char*>(vb); a.upk = reinterpret_cast<const char*>(upk); a.bias = bias;
    a.z = z; a.zmax = zmax; a.zmin = zmin; a.stats = stats;
    a.H = H; a.W = W; a.Cout = Cout; a.KT = Cin / F4_BK; a.NB = Cout / F4_CH;
    a.TH = (H + 3) / 4; a.TW = (W + 3) / 4; a.T = N * a.TH * a.TW;
    a.TB = ssdhip_wino4f_tile_blocks(N, H, W);
    SSDHIP_REQUIRE(a.TB > 0 && a.KT >= 2 && elems_below(1ll << 31, a.TB, a.NB), "wino4f_gemm: bad shape");
    static DeviceOnce attr;
    if (const int dev = attr.pending(); dev != DeviceOnce::kDone) {
        const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4f_gemm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, F4_LDS) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(wino4f_gemm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, F4_LDS) == hipSuccess;
        if (!ok) {
            ssdhip_set_error("wino4f_gemm: cannot enable 126 KiB dynamic LDS");
            return SSDHIP_ELAUNCH;
        }
        attr.mark(dev);
    }
    const dim3 grid((unsigned)(a.TB * a.NB)), block(512);
    if (zmax) hipLaunchKernelGGL(wino4f_gemm_kernel<true>, grid, block, F4_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(wino4f_gemm_kernel<false>, grid, block, F4_LDS, (hipStream_t)stream, a);
    SSDHIP_CHECK_LAUNCH();
    return a.TB;       // rows of the BatchNorm partial table
}

int ssdhip_wino4f_gemm(const float* vb, const float* upk, const float* bias, float* z, float* stats, int N, int H, int W, int Cin, int Cout,
                       void* stream) {
    SSDHIP_REQUIRE(z, "wino4f_gemm: bad arguments");
    return wino4f_gemm_launch(vb, upk, bias, z, nullptr, nullptr, stats, N, H, W, Cin, Cout, stream);
}

int ssdhip_wino4f_gemm_pool(const float* vb, const float* upk, const float* bias, float* zmax, float* zmin, float* stats, int N, int H, int W,
                            int Cin, int Cout, void* stream) {
    SSDHIP_REQUIRE(zmax && zmin && H % 2 == 0 && W % 2 == 0, "wino4f_gemm_pool: even H and W (a tile must hold whole pooling windows)");
    return wino4f_gemm_launch(vb, upk, bias, nullptr, zmax, zmin, stats, N, H, W, Cin, Cout, stream);
}

}  // extern "C"
