// 3x3 / stride 1 / pad 1 convolution with bf16 operands in HBM and fp32 accumulation (v_mfma_f32_32x32x16_bf16), gfx950: the
// frozen VGG16-bn backbone of BASELINE configs[3] / [4] (model/vgg16.py:33) when SSD.set_compute_dtype('bf16') is on.
// Never the fp32 headline path.
//
// What bounds a bf16 convolution on this chip is not the matrix pipe (2.5 PFLOP/s) but the path into LDS: a GEMM-shaped kernel
// that re-fetches its A tile for each of the 9 taps needs > 60 B/clk/CU from L2.  So the input is staged as an image PATCH:
//
//   input   activated (BatchNorm + ReLU already applied) bf16, NHWC with a one-pixel zero halo: (N, H + 2, W + 2, C), written
//           by the producer's apply pass; the halo makes zero padding a property of the tensor -- the kernel never tests a
//           coordinate on the way in
//   item    512 output positions x BN = 128 (or 64) output channels, all of K = 9 C; 512 threads = 8 waves (2 per SIMD: while
//           one wave issues its DMAs or waits at the barrier the other one feeds the matrix pipe), wave = 128 x 64 (or 64 x 64)
//   patch   the pixels under the item's positions and their 3x3 neighbourhoods, 32 channels at a time, by LDS-DMA
//           (buffer_load ... lds, 16 B per lane): each input byte enters the CU once per item and serves all 9 taps.
//           Two geometries, one addressing formula (LDS pixel = mt * mstride + r * pitch + s + lane):
//             tiles  16 rows x 32 columns of one image (maps wider than 93 pixels): patch 18 x 34, pitch 34
//             flat   512 consecutive positions of the haloed tensor taken as one long row (smaller maps): patch 512 + 2 (W + 2) + 2
//                    pixels, pitch W + 2; the halo columns are computed and dropped (2 / (W + 2) of the work)
//   weights pre-packed per (channel tile, 32-channel block, tap) as the LDS image itself (conv_bf16_pack_kernel): 8 KiB per
//           step by LDS-DMA, three buffers so that a step's image is requested two steps ahead
//   LDS     pixel / output-channel rows of 64 B (32 channels); the four 16-B chunks of row p sit at chunk ^ ((p >> 2) & 3): every
//           ds_read_b128 of 32 consecutive rows is conflict-free whatever the first row is (the taps shift it by r * pitch + s)
//   step    one tap of one channel block: per wave 2 x (MT + NT) ds_read_b128 and 2 x MT x NT MFMAs; one barrier per step; the
//           DMAs are counted (s_waitcnt vmcnt(N)): a patch piece has two steps to land, a weight image two steps
//   output  z = conv + bias as fp32 or bf16 (N, H, W, Cout), plus per-(item, channel) sum / sum of squares of the fp32 values
//           for train-mode BatchNorm (deterministic: no atomics)
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int CB_BM = 512;                       // output positions per item
constexpr int CB_PPW = 6;                        // patch DMA pieces (1 KiB = 16 pixels x 64 B) per wave and stage
constexpr int CB_STAGE = 8 * CB_PPW * 1024;      // 48 KiB: one patch stage
constexpr int CB_MAXPIX = CB_STAGE / 64;         // 768 pixels
constexpr int CB_BBUF = 8192;                    // one weight image (128 rows x 64 B); 64-channel tiles use the lower half
constexpr int CB_A_OFF = 0;
constexpr int CB_B_OFF = 2 * CB_STAGE;
constexpr int CB_RED_OFF = CB_B_OFF + 3 * CB_BBUF;
constexpr int CB_LDS = CB_RED_OFF + 8 * 128 * 2 * 4;      // 131,072 B
constexpr int CB_TILE_H = 16, CB_TILE_W = 32, CB_TILE_PITCH = CB_TILE_W + 2;

struct CbArgs {
    const void* a;
    const void* wpk;
    const float* bias;
    void* z;
    float* stats;
    int N, H, W, C, Cout, Hp, Wp;
    int mode;               // 0 = tiles of 16 x 32 pixels, 1 = flat
    int TX, TYX;            // tiles: per tile row, per image
    int MB, NB, KCB;
    int pitch, mstride, npix;
    unsigned a_bytes, w_bytes;
    int skew;
};

// OIHW fp32 (Cout, Cin, 3, 3) -> bf16 [channel tile nb][channel block cb][tap][row = output channel][physical chunk][8]
__global__ void conv_bf16_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin, int BN) {
    const int64_t total = (int64_t)Cout * Cin * 9;
    const int KCB = Cin / 32;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        const int pc = (int)((idx >> 3) & 3);
        int64_t r = idx >> 5;
        const int row = (int)(r % BN); r /= BN;
        const int tap = (int)(r % 9); r /= 9;
        const int cb = (int)(r % KCB);
        const int nb = (int)(r / KCB);
        const int lc = pc ^ ((row >> 2) & 3);
        const int co = nb * BN + row, ci = cb * 32 + lc * 8 + e;
        out[idx] = (__bf16)w[((int64_t)co * Cin + ci) * 9 + tap];
    }
}


// BatchNorm apply + ReLU (+ MaxPool2d(2, 2, pad)) of a pre-BN map z (fp32 or bf16) -> the activated bf16 tensor with a zero halo
// that conv_bf16_kernel reads (model/vgg16.py:28-35).  One thread = one output pixel x 8 channels (one 16-B store).  The value
// is relu(fma(z, scale, shift)) rounded to nearest even: exactly what the implicit-GEMM bf16 template forms when it stages z.
struct F8 { float v[8]; };
template <bool ZBF16>
__device__ __forceinline__ F8 load8(const void* z, int64_t idx8) {
    F8 r;
    if (ZBF16) {
        const bf16x8 t = reinterpret_cast<const bf16x8*>(z)[idx8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r.v[k] = (float)t[k];
    } else {
        const float4 lo = reinterpret_cast<const float4*>(z)[2 * idx8], hi = reinterpret_cast<const float4*>(z)[2 * idx8 + 1];
        r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w; r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    }
    return r;
}
template <bool ZBF16, bool POOL>
__global__ __launch_bounds__(256) void bn_relu_to_bf16_kernel(const void* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
                                                              __bf16* __restrict__ out, int H, int W, int C8, int Ho, int Wo, int pad, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    int64_t pix = i / C8;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int64_t n = pix / Ho;
    const F8 sc = load8<false>(scale, c8), sh = load8<false>(shift, c8);
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = 0.f;          // ReLU output >= 0 and every window holds a real pixel: 0 is the identity of the max
#pragma unroll
    for (int dy = 0; dy < (POOL ? 2 : 1); ++dy) {
        const int h = POOL ? ho * 2 - pad + dy : ho;
        if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < (POOL ? 2 : 1); ++dx) {
            const int w = POOL ? wo * 2 - pad + dx : wo;
            if ((unsigned)w >= (unsigned)W) continue;
            const F8 v = load8<ZBF16>(z, ((n * H + h) * W + w) * C8 + c8);
#pragma unroll
            for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], fmaxf(fmaf(v.v[k], sc.v[k], sh.v[k]), 0.f));
        }
    }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (__bf16)m[k];
    reinterpret_cast<bf16x8*>(out)[((n * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * C8 + c8] = o;
}

template <int MT, int NT, int WN, bool ZBF16>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_bf16_kernel(CbArgs a) {
    constexpr int WM = 8 / WN;
    constexpr int BN = NT * 32 * WN;
    constexpr int IMG = BN * 64;
    static_assert(WM * MT * 32 == CB_BM, "8 waves cover the 512 positions of an item");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;

    // work items (tile block mb, channel tile nb): a block keeps nb (the host makes the grid a multiple of NB) and walks
    // mb0, mb0 + mbstep, ... below mb_end
    const int G = gridDim.x;
    int nb, mb0, mbstep = G / a.NB, mb_end = a.MB;
    if (G % (8 * a.NB) == 0) {
        // an XCD walks a contiguous eighth of the tile blocks, its groups side by side: neighbouring patches (shared halo rows) and
        // the channel tiles of one patch meet in one L2 (wino_fused.hip: halved that kernel's fabric traffic)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, L = (a.MB + 7) >> 3;
        nb = slot % a.NB;
        mbstep = (G >> 3) / a.NB;
        mb0 = xcd * L + slot / a.NB;
        mb_end = min(a.MB, (xcd + 1) * L);
    } else {
        nb = blockIdx.x % a.NB;
        mb0 = blockIdx.x / a.NB;
    }
    if (mb0 >= mb_end) return;
    const int cnt = (mb_end - mb0 + mbstep - 1) / mbstep;
    for (int k = (int)(blockIdx.x >> 3) * a.skew; k > 0; --k) __builtin_amdgcn_s_sleep(16);
    const int n0 = nb * BN;
    const int steps = a.KCB * 9;

    // ---- DMA sources.  Patch: lane -> (LDS pixel p, physical chunk) of each of this wave's 6 pieces; the source offset relative
    // to the item's origin does not depend on the item, so the item and the channel block travel in the scalar offset.
    unsigned dof[CB_PPW];
#pragma unroll
    for (int k = 0; k < CB_PPW; ++k) {
        const int q = (wave * CB_PPW + k) * 64 + lane, p = q >> 2;
        const int lc = (q & 3) ^ ((p >> 2) & 3);
        const int rel = a.mode ? p : (p / CB_TILE_PITCH) * a.Wp + p % CB_TILE_PITCH;
        dof[k] = p < a.npix ? (unsigned)rel * (unsigned)(a.C * 2) + (unsigned)(lc * 16) : 0u;     // slots behind the patch are never read: any valid source
    }
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.a), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wpk), 0, a.w_bytes, 0x00020000);
    auto dma_a = [&](int k, int stage, unsigned soff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (lptr_t)(lds + CB_A_OFF + stage * CB_STAGE + (wave * CB_PPW + k) * 1024), 16, dof[k], soff, 0, 0);
    };
    // weight image of step sidx (of this block's channel tile) -> buffer buf: one piece per wave (64-channel tiles: the pieces of
    // waves 4-7 re-read the image's first KiB into the unused half of the buffer)
    const unsigned b_vo = wave * 1024 < IMG ? (unsigned)(wave * 1024 + lane * 16) : (unsigned)(lane * 16);
    auto dma_b = [&](int buf, int sidx) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lptr_t)(lds + CB_B_OFF + buf * CB_BBUF + wave * 1024), 16, b_vo, (nb * steps + sidx) * IMG, 0, 0);
    };
    auto origin = [&](int mb) -> unsigned {       // byte offset of the item's patch origin in the haloed input
        if (a.mode) return (unsigned)(CB_BM * mb) * (unsigned)(a.C * 2);
        const int n = mb / a.TYX, t = mb - n * a.TYX, ty = t / a.TX, tx = t - ty * a.TX;
        return (unsigned)((n * a.Hp + ty * CB_TILE_H) * a.Wp + tx * CB_TILE_W) * (unsigned)(a.C * 2);
    };

    // ---- operand fragments.  All LDS reads of the loop are asm: the compiler would drain vmcnt in front of any LDS read it can
    // see while a DMA is in flight.
    const int pw = wm * MT * a.mstride + li;                                      // LDS pixel of (mt = 0, tap (0, 0))
    unsigned blane[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
        blane[kc] = lds0 + (unsigned)(CB_B_OFF + (wn * NT * 32 + li) * 64 + ((((2 * kc + h) ^ (li >> 2)) & 3) << 4));
    bf16x8 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    auto issue_reads = [&](bf16x8(&fa)[MT], bf16x8(&fb)[NT], unsigned abuf, int bbuf, int toff, int kc) {
        int pwt = pw;
        asm volatile("" : "+v"(pwt));          // keep the 72 (tap, tile, k-chunk) addresses out of registers: recompute them per step
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pix = pwt + mt * a.mstride + toff;
            const unsigned ad = abuf + (unsigned)(pix * 64) + (unsigned)((((2 * kc + h) ^ (pix >> 2)) & 3) << 4);
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[mt]) : "v"(ad));
        }
        const unsigned bd = blane[kc] + (unsigned)(bbuf * CB_BBUF);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nt]) : "v"(bd), "n"(nt * 2048));
    };
    f32x16 acc[MT][NT];
    auto mfmas = [&](bf16x8(&fa)[MT], bf16x8(&fb)[NT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
    };
#define CB_TIE(fa, fb)                                                       \
    do {                                                                     \
        _Pragma("unroll") for (int q_ = 0; q_ < MT; ++q_) asm volatile("" : "+v"(fa[q_])); \
        _Pragma("unroll") for (int q_ = 0; q_ < NT; ++q_) asm volatile("" : "+v"(fb[q_])); \
    } while (0)

    // ---- prologue: patch of (item 0, channel block 0) in stage 0, weight images of steps 0 and 1, fragments of step 0
    unsigned org_cur = origin(mb0), org_next = origin(cnt > 1 ? mb0 + mbstep : mb0);
#pragma unroll
    for (int k = 0; k < CB_PPW; ++k) dma_a(k, 0, org_cur);
    dma_b(0, 0);
    dma_b(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_reads(fa0, fb0, lds0 + CB_A_OFF, 0, 0, 0);

    const int co_lane = n0 + wn * NT * 32 + li;
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = a.bias ? a.bias[co_lane + nt * 32] : 0.f;
    float* red = reinterpret_cast<float*>(lds + CB_RED_OFF);
    auto flush_stats = [&](int mb) {           // after a barrier behind the epilogue that filled red
        if (tid < BN) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
            float* o = a.stats + ((size_t)mb * a.Cout + n0 + tid) * 2;
            o[0] = t1; o[1] = t2;
        }
    };

    bool drain = false;
    int cbg = 0;                                // stages consumed so far: parity = stage of the current channel block
#pragma clang loop unroll(disable)
    for (int j = 0; j < cnt; ++j) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
#pragma clang loop unroll(disable)
        for (int cb = 0; cb < a.KCB; ++cb, ++cbg) {
            const int st = cbg & 1;
            const unsigned acur = lds0 + (unsigned)(CB_A_OFF + st * CB_STAGE), anxt = lds0 + (unsigned)(CB_A_OFF + (st ^ 1) * CB_STAGE);
            const unsigned a_next = cb + 1 < a.KCB ? org_cur + (unsigned)((cb + 1) * 64) : org_next;     // next stage: next block or next item
            const int sbase = cb * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // step = tap t of channel block cb.  It requests the weight image of step + 2 and one patch piece of the next stage,
                // computes with the fragments of (t, k-chunk 0) read during the previous step, reads (t, 1) and (t + 1, 0).
                int s2 = sbase + t + 2;
                if (s2 >= steps) s2 -= steps;
                dma_b((t + 2) % 3, s2);
                if (t < CB_PPW) dma_a(t, st ^ 1, a_next);
                issue_reads(fa1, fb1, acur, t % 3, (t / 3) * a.pitch + t % 3, 1);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT));
                CB_TIE(fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                // the weight image of step + 1 was requested one step ago, in front of that step's patch piece: everything older
                // than (that piece, this step's two requests) has landed
                if (t == 0 && drain) {           // first step behind an epilogue: its stores are still in the counter
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    drain = false;
                } else {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(((t >= 1 && t <= CB_PPW) ? 1 : 0) + 1 + (t < CB_PPW ? 1 : 0)) : "memory");
                }
                CB_TIE(fa1, fb1);
                __builtin_amdgcn_s_barrier();
                if (t == 0 && cb == 0 && j > 0 && a.stats != nullptr) flush_stats(mb0 + (j - 1) * mbstep);
                issue_reads(fa0, fb0, t < 8 ? acur : anxt, (t + 1) % 3, t < 8 ? ((t + 1) / 3) * a.pitch + (t + 1) % 3 : 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue of item j: bias, store, BatchNorm partial sums
        const int mb = mb0 + j * mbstep;
        int n_img = 0, ty = 0, tx = 0;
        if (!a.mode) { n_img = mb / a.TYX; const int t = mb - n_img * a.TYX; ty = t / a.TX; tx = t - ty * a.TX; }
        float s1[NT], s2[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int mrow = wm * MT + mt;
            int fn = 0, fy = 0, fx = 0;
            if (a.mode) {                      // haloed coordinates of the first of this tile's 32 consecutive positions
                const int j0 = a.Wp + 1 + CB_BM * mb + 32 * mrow, img = a.Hp * a.Wp;
                fn = j0 / img;
                const int rem = j0 - fn * img;
                fy = rem / a.Wp;
                fx = rem - fy * a.Wp;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ie = (e & 3) + 8 * (e >> 2) + 4 * h;
                bool ok;
                int pix;
                if (a.mode) {
                    int xh = fx + ie, yh = fy, nn = fn;
                    if (xh >= a.Wp) { xh -= a.Wp; ++yh; }
                    if (xh >= a.Wp) { xh -= a.Wp; ++yh; }
                    if (yh >= a.Hp) { yh -= a.Hp; ++nn; }
                    ok = nn < a.N && yh >= 1 && yh <= a.H && xh >= 1 && xh <= a.W;
                    pix = (nn * a.H + yh - 1) * a.W + xh - 1;
                } else {
                    const int y = ty * CB_TILE_H + mrow, x = tx * CB_TILE_W + ie;
                    ok = y < a.H && x < a.W;
                    pix = (n_img * a.H + y) * a.W + x;
                }
                if (ok) {          // (pairing neighbouring channels into 4-B bf16 stores by a lane exchange was measured: slower, 0.133 -> 0.141 ms on conv_3_1)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float v = acc[mt][nt][e] + bv[nt];
                        const size_t o = (size_t)pix * a.Cout + co_lane + nt * 32;
                        if (ZBF16) reinterpret_cast<__bf16*>(a.z)[o] = (__bf16)v;
                        else reinterpret_cast<float*>(a.z)[o] = v;
                        s1[nt] += v; s2[nt] += v * v;
                    }
                }
            }
        }
        if (a.stats != nullptr) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float t1 = s1[nt] + __shfl_xor(s1[nt], 32, 64), t2 = s2[nt] + __shfl_xor(s2[nt], 32, 64);
                if (lane < 32) { red[(wm * BN + wn * NT * 32 + nt * 32 + lane) * 2] = t1; red[(wm * BN + wn * NT * 32 + nt * 32 + lane) * 2 + 1] = t2; }
            }
        }
        drain = true;            // the stores share the DMA counter: the next step waits for vmcnt(0) instead of its counted value
        org_cur = org_next;
        org_next = origin(j + 2 < cnt ? mb0 + (j + 2) * mbstep : mb0 + (cnt - 1) * mbstep);
    }
    CB_TIE(fa0, fb0);                              // the fragments read ahead for a step that does not exist
    if (a.stats != nullptr) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        flush_stats(mb0 + (cnt - 1) * mbstep);
    }
#undef CB_TIE
}

}  // namespace

extern "C" {

static int cb_bn(int Cout) { return Cout % 128 == 0 ? 128 : 64; }
static bool cb_flat(int W) { return CB_BM + 2 * (W + 2) + 2 <= CB_MAXPIX; }

int64_t ssdhip_conv_bf16_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64 || Cin % 32) return SSDHIP_EINVAL;
    return (int64_t)Cout * Cin * 9 * 2;
}

int ssdhip_conv_bf16_pack(const float* w_oihw, void* wpk, int Cout, int Cin, void* stream) {
    SSDHIP_REQUIRE(w_oihw && wpk && Cout > 0 && Cin > 0 && Cout % 64 == 0 && Cin % 32 == 0, "conv_bf16_pack: Cout must be a multiple of 64 and Cin of 32");
    const int64_t total = (int64_t)Cout * Cin * 9;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(conv_bf16_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, reinterpret_cast<__bf16*>(wpk), Cout, Cin, cb_bn(Cout));
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

// bytes of the haloed activation tensor (N, H + 2, W + 2, C) bf16 plus the zero slack the last items' patches reach into
int64_t ssdhip_conv_bf16_act_bytes(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return SSDHIP_EINVAL;
    const int64_t px = (int64_t)N * (H + 2) * (W + 2) + 18ll * (W + 2) + 1024;
    return px * C * 2;
}

int ssdhip_conv_bf16_stat_rows(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return SSDHIP_EINVAL;
    int64_t mb;
    if (cb_flat(W)) mb = ((int64_t)N * (H + 2) * (W + 2) - 2 * (W + 2) - 2 + CB_BM - 1) / CB_BM;
    else mb = (int64_t)N * ((H + CB_TILE_H - 1) / CB_TILE_H) * ((W + CB_TILE_W - 1) / CB_TILE_W);
    if (mb <= 0 || mb >= (1ll << 24)) return SSDHIP_EINVAL;
    return (int)mb;
}

int ssdhip_conv_bf16(const void* a, const void* wpk, const float* bias, void* z, int z_bf16, float* stats, int N, int H, int W, int Cin, int Cout,
                     int max_blocks, void* stream) {
    SSDHIP_REQUIRE(a && wpk && z && N > 0 && H > 0 && W > 0 && max_blocks >= 0, "conv_bf16: bad arguments");
    SSDHIP_REQUIRE(Cin >= 32 && Cin % 32 == 0 && Cout > 0 && Cout % 64 == 0, "conv_bf16: Cin must be a multiple of 32 and Cout of 64");
    const int64_t abytes = ssdhip_conv_bf16_act_bytes(N, H, W, Cin);
    SSDHIP_REQUIRE(abytes > 0 && abytes < (1ll << 32) && elems_below(1ll << 31, N, H, W, Cout), "conv_bf16: the haloed input must stay below 4 GiB (32-bit byte offsets) and the output below 2^31 elements");
    SSDHIP_REQUIRE(W >= 14, "conv_bf16: map too narrow (W >= 14)");
    static DeviceOnce attr;
    if (const int dev = attr.pending(); dev != DeviceOnce::kDone) {
        bool ok = true;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_kernel<4, 2, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CB_LDS) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_kernel<4, 2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CB_LDS) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_kernel<2, 2, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CB_LDS) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_kernel<2, 2, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CB_LDS) == hipSuccess;
        if (!ok) { ssdhip_set_error("conv_bf16: cannot enable 128 KiB dynamic LDS"); return SSDHIP_ELAUNCH; }
        attr.mark(dev);
    }
    CbArgs g;
    g.a = a; g.wpk = wpk; g.bias = bias; g.z = z; g.stats = stats;
    g.N = N; g.H = H; g.W = W; g.C = Cin; g.Cout = Cout; g.Hp = H + 2; g.Wp = W + 2;
    g.mode = cb_flat(W) ? 1 : 0;
    g.TX = (W + CB_TILE_W - 1) / CB_TILE_W;
    g.TYX = g.TX * ((H + CB_TILE_H - 1) / CB_TILE_H);
    g.MB = ssdhip_conv_bf16_stat_rows(N, H, W);
    SSDHIP_REQUIRE(g.MB > 0, "conv_bf16: bad shape");
    const int BN = cb_bn(Cout);
    g.NB = Cout / BN; g.KCB = Cin / 32;
    g.pitch = g.mode ? g.Wp : CB_TILE_PITCH;
    g.mstride = g.mode ? 32 : CB_TILE_PITCH;
    g.npix = g.mode ? CB_BM + 2 * g.Wp + 2 : (CB_TILE_H + 2) * CB_TILE_PITCH;
    g.a_bytes = (unsigned)abytes;
    g.w_bytes = (unsigned)((int64_t)Cout * Cin * 18);
    g.skew = 0;
#ifdef SSDHIP_PROBES
    if (const char* e = getenv("SSDHIP_CB_SKEW")) g.skew = atoi(e);
#endif
    SSDHIP_REQUIRE(g.NB <= 256, "conv_bf16: too many channel tiles");
    const int64_t items = (int64_t)g.MB * g.NB;
    const int cus = ssdhip_cu_count();              // one persistent block per CU (128 KiB of LDS each)
    const int cap = max_blocks > 0 && max_blocks < cus ? (max_blocks < g.NB ? g.NB : max_blocks) : cus;
    const int G = items < cap ? (int)items : cap / g.NB * g.NB;
    dim3 grid((unsigned)G), block(512);
    hipStream_t st = (hipStream_t)stream;
    if (BN == 128) {
        if (z_bf16) hipLaunchKernelGGL((conv_bf16_kernel<4, 2, 2, true>), grid, block, CB_LDS, st, g);
        else hipLaunchKernelGGL((conv_bf16_kernel<4, 2, 2, false>), grid, block, CB_LDS, st, g);
    } else {
        if (z_bf16) hipLaunchKernelGGL((conv_bf16_kernel<2, 2, 1, true>), grid, block, CB_LDS, st, g);
        else hipLaunchKernelGGL((conv_bf16_kernel<2, 2, 1, false>), grid, block, CB_LDS, st, g);
    }
    SSDHIP_CHECK_LAUNCH();
    return g.MB;
}

int ssdhip_bn_relu_to_bf16(const void* z, int z_bf16, const float* scale, const float* shift, void* out, int N, int H, int W, int C, int pool,
                           int pool_pad, void* stream) {
    SSDHIP_REQUIRE(z && scale && shift && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "bn_relu_to_bf16: bad arguments (C % 8 == 0)");
    SSDHIP_REQUIRE(pool_pad == 0 || pool_pad == 1, "bn_relu_to_bf16: pool padding 0 or 1");
    hipStream_t st = (hipStream_t)stream;
    const int C8 = C / 8;
    const int Ho = pool ? (H + 2 * pool_pad - 2) / 2 + 1 : H, Wo = pool ? (W + 2 * pool_pad - 2) / 2 + 1 : W;
    const int64_t total = (int64_t)N * Ho * Wo * C8;
    const dim3 grid(cdiv(total, 256)), block(256);
    __bf16* o = reinterpret_cast<__bf16*>(out);
    if (pool) {
        if (z_bf16) hipLaunchKernelGGL((bn_relu_to_bf16_kernel<true, true>), grid, block, 0, st, z, scale, shift, o, H, W, C8, Ho, Wo, pool_pad, total);
        else hipLaunchKernelGGL((bn_relu_to_bf16_kernel<false, true>), grid, block, 0, st, z, scale, shift, o, H, W, C8, Ho, Wo, pool_pad, total);
    } else {
        if (z_bf16) hipLaunchKernelGGL((bn_relu_to_bf16_kernel<true, false>), grid, block, 0, st, z, scale, shift, o, H, W, C8, Ho, Wo, 0, total);
        else hipLaunchKernelGGL((bn_relu_to_bf16_kernel<false, false>), grid, block, 0, st, z, scale, shift, o, H, W, C8, Ho, Wo, 0, total);
    }
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
