// Fused Winograd F(2x2, 3x3) convolution for the 3x3 / stride 1 / pad 1 layers of the VGG16-bn backbone (model/vgg16.py:33) on
// maps large enough to fill the chip: input transform, the 16 Winograd-domain products and the output transform in ONE kernel.
// Nothing of the Winograd domain touches HBM (the three-pass form in winograd.hip + gemm_planes.hip writes and re-reads
// V = 4x the input bytes and M = 4x the output bytes per layer).
//
// Why it fits gfx950: v_mfma_f32_32x32x2_f32 keeps the output channel on the lane and the 32 tile rows in the 16 accumulator
// registers, the same for every plane xi.  A wave that owns 32 tiles x 32 output channels for ALL 16 planes therefore holds, per
// lane and per register index, the 16 values m[xi] of one (tile, channel): the output transform A^T m A is register-local
// arithmetic, no shuffle, no LDS.  16 planes x 16 registers = 256 accumulators = the AGPR half of the 512-register file at one
// wave per SIMD.
//
// What bounds it (measured, scripts/wf_bench.py + scripts/wf_pmc.sh): the fp32 MFMA runs at the fp32 VECTOR rate and shares
// the SIMD's fp32 lanes with the VALU, so the input transform is not free beside it -- every VALU instruction of the loop costs
// the matrix pipe 2-3 cycles, and the clock drops from 2.40 to 2.14 GHz when the VALU works beside the MFMAs.  Hence the
// transform is computed ONCE per block (each thread owns half a patch x 4 channels of one tile: 160 VALU instructions per
// thread and K-step) and shared through LDS, rather than recomputed by every wave that needs it (256: 20 % slower).
//
//   block   256 threads = 4 waves as 2 (tiles) x 2 (channels): 64 consecutive tiles x 64 output channels x 16 planes
//   K loop  8 input channels per step; per wave and step 16 planes x 4 MFMAs (4,096 MFMA cycles)
//   x       the pixels under the block's 64 tiles (per tile-row segment a strip of 4 rows x (2 len + 2) pixels), the step's 8
//           channels, by LDS-DMA into a raw tile (buffer_load ... lds; out-of-image pixels read 0 through the bounds check): every
//           input byte enters the CU once per block.  (A first version fetched each patch position with per-thread 16-B loads,
//           six requests per 32-B piece: 0.48 of the 1.34 ms of conv_4_2 went into waiting on them.)
//   A       thread (tile = lane, upper / lower half of the 4x4 patch, channel quad) reads its 12 patch positions from the raw
//           tile, applies the producer's BatchNorm scale/shift + ReLU as fma + v_med3 (upper bound +inf inside the image, 0
//           outside: zero padding AFTER the activation), runs the row and column passes of B^T d B in registers and stores its 8
//           planes of the next step's A image; a mid-step barrier frees the (single) raw tile for the next DMA
//   B       weights pre-packed per (channel tile, K-step) as the LDS image itself (wino_fused_pack_kernel), so a step's 32 KiB
//           arrive by LDS-DMA (1 KiB contiguous per wave-instruction)
//   LDS     2 x (A 32 KiB + B 32 KiB) + raw tile 20 KiB; operand rows of 32 B (8 k), the two 16-B halves swapped where
//           (row >> 3) & 1: conflict-free ds_read_b128 for the MFMA feed (lane reads 4 consecutive k and issues 4 MFMAs)
//   epilogue A^T m A + bias from the accumulators, NHWC store (128-B lines per half-wave), BatchNorm partial sums per block
//   order   block id -> (channel tile = id % NB, tile block = id / NB): with the round-robin XCD placement an XCD keeps seeing
//           the same one or two weight slices (2 MiB at K = 512) in its L2 while the tile blocks stream (speed only)
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int WF_TILES = 64, WF_COUT = 64, WF_BK = 8;
constexpr int WF_PLANE = 64 * WF_BK * 4;              // bytes of one plane of an operand image: 64 rows x 32 B
constexpr int WF_OPER = 16 * WF_PLANE;                // 32 KiB: one operand image of one K-step
constexpr int WF_SLOTS = 640;                         // pixel slots of the raw input tile (512 + 8 per tile-row segment, <= 16 segments)
constexpr int WF_QUAD = WF_SLOTS * 16;                // bytes of one channel quad of the raw tile
constexpr int WF_MAXC = 1024;
constexpr int WF_A_OFF = 0;                           // 2 A images
constexpr int WF_B_OFF = 2 * WF_OPER;                 // 2 B images
constexpr int WF_RAW_OFF = 4 * WF_OPER;               // raw tile: the step's 2 channel quads of every slot
constexpr int WF_SS_OFF = WF_RAW_OFF + 2 * WF_QUAD;   // scale[Cin], shift[Cin]
constexpr int WF_TAB_OFF = WF_SS_OFF + 2 * WF_MAXC * 4;   // per-tile output table: 64 x int2
constexpr int WF_RED_OFF = WF_TAB_OFF + 64 * 8;       // stats exchange: 2 (wm) x 64 (channel) x 2 floats
constexpr int WF_LDS = WF_RED_OFF + 2 * 64 * 2 * 4;   // 161,280 B of the CU's 163,840

struct WfArgs {
    const float* x;
    const float* upk;
    const float* bias;
    const float* scale;
    const float* shift;
    float* z;
    float* stats;
    int H, W, C, Cout, TH, TW, T, KT, NB, RT;      // RT = N * TH tile rows
    unsigned x_bytes;
    unsigned magic_tw, magic_th;                  // ceil(2^32 / d): n / d == umulhi(n, magic) for n * d < 2^32
};

__device__ __forceinline__ int fdiv(int n, unsigned magic) { return (int)__umulhi((unsigned)n, magic); }

// U[xi] = (G g G^T)[xi], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; same operation order as winograd_weight_kernel
__device__ __forceinline__ float wino_u(const float* __restrict__ g9, int xi) {
    const int r = xi >> 2, s = xi & 3;
    float t[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g0 = g9[c], g1 = g9[3 + c], g2 = g9[6 + c];
        t[c] = r == 0 ? g0 : r == 1 ? 0.5f * (g0 + g1 + g2) : r == 2 ? 0.5f * (g0 - g1 + g2) : g2;
    }
    return s == 0 ? t[0] : s == 1 ? 0.5f * (t[0] + t[1] + t[2]) : s == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2];
}

// OIHW (Cout, Cin, 3, 3) -> [channel tile nb][K-step g][xi][row][physical 16-B chunk][4]: the B image of (nb, g) as it lies in LDS
__global__ void wino_fused_pack_kernel(const float* __restrict__ w, float* __restrict__ upk, int Cout, int Cin) {
    const int64_t total = (int64_t)16 * Cout * Cin;
    const int KT = Cin / WF_BK;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3), chunk = (int)((i >> 2) & 1), row = (int)((i >> 3) & 63), xi = (int)((i >> 9) & 15);
        const int64_t rest = i >> 13;
        const int g = (int)(rest % KT), nb = (int)(rest / KT);
        const int co = nb * 64 + row, c = WF_BK * g + 4 * (chunk ^ ((row >> 3) & 1)) + e;
        const float u = wino_u(w + ((size_t)co * Cin + c) * 9, xi);
        upk[i] = xi >= 12 ? -u : u;          // the kernel's A image holds -V in planes 12..15 (see rows_of)
    }
}

// Geometry of a block's 64 consecutive tiles t0 .. t0 + 63 (row-major over (n, th, tw)): segment 0 is the rest of tile row R0
// from column tw0 (len0 tiles), segments 1.. are whole tile rows, the last one possibly cut short.  Segment k owns the pixel strip
// rows 2 th - 1 .. 2 th + 2, columns 2 tws - 1 .. 2 (tws + len) of its image, 4 x (2 len + 2) slots starting at seg_off(k).
struct WfGeo {
    int R0, tw0, len0, off1, rowslots;
    __device__ __forceinline__ int seg_len(int k, int TW) const { return k == 0 ? len0 : min(TW, WF_TILES - len0 - (k - 1) * TW); }
    __device__ __forceinline__ int seg_off(int k) const { return k == 0 ? 0 : off1 + (k - 1) * rowslots; }
};

// ABL (probe builds only, wrong results by design): bit 0 = no raw-tile DMA in the K loop, bit 1 = no weight DMA in the K loop,
// bit 2 = no activation / input transform in the K loop
template <bool XFORM, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_fused_kernel(WfArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nb = blockIdx.x % a.NB, mb = blockIdx.x / a.NB;
    const int t0 = mb * WF_TILES, n0 = nb * WF_COUT;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const int TW = a.TW, TH = a.TH;

    WfGeo geo;
    geo.R0 = t0 / TW;                       // uniform: scalar division once per block
    geo.tw0 = t0 - geo.R0 * TW;
    geo.len0 = min(WF_TILES, TW - geo.tw0);
    geo.off1 = 4 * (2 * geo.len0 + 2);
    geo.rowslots = 4 * (2 * TW + 2);
    const int nseg = geo.len0 == WF_TILES ? 1 : 1 + (WF_TILES - geo.len0 + TW - 1) / TW;
    const int total_slots = geo.seg_off(nseg - 1) + 4 * (2 * geo.seg_len(nseg - 1, TW) + 2);

    // ---- byte offset in x of every pixel slot (0x80000000 = outside the image / unused: the DMA then writes 0), via LDS:
    // thread t decodes slots t, t + 256, t + 512; afterwards each lane keeps the slots of its 5 DMA pieces
    {
        unsigned* tab = reinterpret_cast<unsigned*>(lds + WF_A_OFF);          // the A images are idle until the prologue fills one
        for (int s = tid; s < WF_SLOTS; s += 256) {
            unsigned o = 0x80000000u;
            if (s < total_slots) {
                const int k = s < geo.off1 ? 0 : 1 + (s - geo.off1) / geo.rowslots;
                const int rem = s - geo.seg_off(k);
                const int width = 2 * geo.seg_len(k, TW) + 2;
                const int r = rem / width, col = rem - r * width;
                const int R = geo.R0 + k;
                const int n = fdiv(R, a.magic_th), th = R - n * TH;
                const int h = 2 * th - 1 + r, w = 2 * (k == 0 ? geo.tw0 : 0) - 1 + col;
                if (R < a.RT && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
                    o = (unsigned)(((n * a.H + h) * a.W + w) * a.C * 4);
            }
            tab[s] = o;
        }
        // BatchNorm scale / shift of the producer, once per block
        if (XFORM) {
            float* ss = reinterpret_cast<float*>(lds + WF_SS_OFF);
            for (int c = tid; c < a.C; c += 256) { ss[c] = a.scale[c]; ss[WF_MAXC + c] = a.shift[c]; }
        }
        // output table: float offset of pixel (n, 2 th, 2 tw) of each tile and what exists of its 2x2 outputs
        if (tid < 64) {
            const int i = tid;
            int2 ent = make_int2(0, 0);
            if (t0 + i < a.T) {
                int k = 0, c = i;
                if (i >= geo.len0) { const int ii = i - geo.len0; const int q = fdiv(ii, a.magic_tw); k = 1 + q; c = ii - q * TW; }
                const int R = geo.R0 + k, tw = (k == 0 ? geo.tw0 : 0) + c;
                const int n = fdiv(R, a.magic_th), th = R - n * TH;
                ent.x = ((n * a.H + 2 * th) * a.W + 2 * tw) * a.Cout;
                ent.y = 1 | (2 * th + 1 < a.H ? 2 : 0) | (2 * tw + 1 < a.W ? 4 : 0);
            }
            *reinterpret_cast<int2*>(lds + WF_TAB_OFF + i * 8) = ent;
        }
    }
    __syncthreads();
    // raw-tile DMA: 2 quads x 10 pieces of 64 slots; wave w moves pieces 5 w .. 5 w + 4 (waves 0, 1: quad 0; waves 2, 3: quad 1)
    unsigned dof[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) dof[k] = reinterpret_cast<const unsigned*>(lds + WF_A_OFF)[64 * ((5 * wave + k) % 10) + lane];
    __syncthreads();

    // ---- staging role of this thread: tile = lane, hh = upper (patch rows 0-2 -> planes 0-7) / lower (rows 3-1 -> planes 8-15)
    // half of the patch, q = which 4 of the step's 8 channels.  hh and q are wave-uniform.
    const int hh = wave & 1, q = wave >> 1;
    unsigned pb[3];
    float hi[3][4];
    {
        const int i = lane;
        int k = 0, c = i;
        if (i >= geo.len0) { const int ii = i - geo.len0; const int qq = fdiv(ii, a.magic_tw); k = 1 + qq; c = ii - qq * TW; }
        const int width = 2 * geo.seg_len(k, TW) + 2;
        const int R = geo.R0 + k, tw = (k == 0 ? geo.tw0 : 0) + c;
        const int n = fdiv(R, a.magic_th), th = R - n * TH;
        const bool tv = t0 + i < a.T;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int pr = hh ? 3 - r : r;                       // upper half: patch rows 0, 1, 2; lower half: rows 3, 2, 1
            pb[r] = lds0 + (unsigned)(WF_RAW_OFF + q * WF_QUAD + (tv ? (geo.seg_off(k) + pr * width + 2 * c) * 16 : 0));
            const int h = 2 * th - 1 + pr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = 2 * tw - 1 + j;
                hi[r][j] = (tv && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) ? __builtin_inff() : 0.0f;
            }
        }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    auto dma_raw = [&](int g) {        // the 8 channels of step g of every pixel slot -> raw tile (this wave's 5 pieces)
        const int quad = wave >> 1;
        char* d = lds + WF_RAW_OFF + quad * WF_QUAD + ((5 * wave) % 10) * 1024;
        const int soff = g * 32 + quad * 16;
#pragma unroll
        for (int k = 0; k < 5; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr_t)(d + k * 1024), 16, dof[k], soff, 0, 0);
    };
    const char* bsrc = reinterpret_cast<const char*>(a.upk) + (size_t)nb * a.KT * WF_OPER + wave * (WF_OPER / 4) + lane * 16;
    auto dma_b = [&](int g) {          // this wave's quarter (8 x 1 KiB) of step g's weight image -> B image g & 1
        const char* s = bsrc + (size_t)g * WF_OPER;
        char* d = lds + WF_B_OFF + (g & 1) * WF_OPER + wave * (WF_OPER / 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_amdgcn_global_load_lds((gptr_t)(s + k * 1024), (lptr_t)(d + k * 1024), 16, 0, 0);
    };

    // ---- patch registers of the NEXT step and the transform.  All LDS reads of the loop are asm: the compiler would put
    // s_waitcnt vmcnt(0) in front of any LDS read it can see while a DMA is in flight.
    f32x4 P[3][4], sc, sh, rx[4], ry[4];
    sc = (f32x4){1.f, 1.f, 1.f, 1.f}; sh = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned ss_rd = lds0 + (unsigned)(WF_SS_OFF + q * 16);
    auto read_patch = [&](int g) {     // 12 (+2) reads
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(P[r][j]) : "v"(pb[r]), "n"(j * 16));
        if (XFORM) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(sc) : "v"(ss_rd + (unsigned)(g * 32)));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sh) : "v"(ss_rd + (unsigned)(g * 32)), "n"(WF_MAXC * 4));
        }
    };
#define WF_TIE_P()                                                                                                                   \
    asm volatile("" : "+v"(P[0][0]), "+v"(P[0][1]), "+v"(P[0][2]), "+v"(P[0][3]), "+v"(P[1][0]), "+v"(P[1][1]), "+v"(P[1][2]),        \
                      "+v"(P[1][3]), "+v"(P[2][0]), "+v"(P[2][1]), "+v"(P[2][2]), "+v"(P[2][3]), "+v"(sc), "+v"(sh))
    auto act = [&](f32x4 f, float h) -> f32x4 {
        if (XFORM) {
            f.x = __builtin_amdgcn_fmed3f(__builtin_fmaf(f.x, sc.x, sh.x), 0.f, h);
            f.y = __builtin_amdgcn_fmed3f(__builtin_fmaf(f.y, sc.y, sh.y), 0.f, h);
            f.z = __builtin_amdgcn_fmed3f(__builtin_fmaf(f.z, sc.z, sh.z), 0.f, h);
            f.w = __builtin_amdgcn_fmed3f(__builtin_fmaf(f.w, sc.w, sh.w), 0.f, h);
        }
        return f;
    };
    // Upper half (hh = 0) holds patch rows (d0, d1, d2): B^T d row 0 = d0 - d2, row 1 = d1 + d2.  The lower half loads its rows in
    // the order (d3, d2, d1), so the same two expressions give d3 - d1 = -(row 3) and, with s = -1, d2 - d1 = row 2: no branch, no
    // select.  Planes 12..15 therefore hold -V; the packed weights of those planes are negated too, so every product is unchanged.
    const float ysign = hh ? -1.f : 1.f;
    const unsigned x_wr = lds0 + (unsigned)(WF_A_OFF + (hh ? 12 : 0) * WF_PLANE + lane * 32 + ((q ^ ((lane >> 3) & 1)) * 16));
    const unsigned y_wr = lds0 + (unsigned)(WF_A_OFF + (hh ? 8 : 4) * WF_PLANE + lane * 32 + ((q ^ ((lane >> 3) & 1)) * 16));
    auto rows_of = [&](int j) {          // column j of the patch: activation, then the two B^T d rows this thread owns
        const f32x4 l0 = act(P[0][j], hi[0][j]), l1 = act(P[1][j], hi[1][j]), l2 = act(P[2][j], hi[2][j]);
        rx[j] = l0 - l2;
        ry[j].x = __builtin_fmaf(ysign, l2.x, l1.x); ry[j].y = __builtin_fmaf(ysign, l2.y, l1.y);
        ry[j].z = __builtin_fmaf(ysign, l2.z, l1.z); ry[j].w = __builtin_fmaf(ysign, l2.w, l1.w);
    };
    auto planes_of = [&](int i, unsigned abase) {     // row i (0: rx, 1: ry) -> its 4 planes of the A image at LDS offset abase
        const f32x4* r = i == 0 ? rx : ry;
        const unsigned dst = (i == 0 ? x_wr : y_wr) + abase;
        const f32x4 o0 = r[0] - r[2], o1 = r[1] + r[2], o2 = r[2] - r[1], o3 = r[1] - r[3];
        asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(o0) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(dst), "v"(o1), "n"(WF_PLANE) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(dst), "v"(o2), "n"(2 * WF_PLANE) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(dst), "v"(o3), "n"(3 * WF_PLANE) : "memory");
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    // ---- prologue: A and B images of step 0 in buffers 0, raw tile of step 1 in flight
    const int KT = a.KT;
    dma_raw(0);
    dma_b(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_patch(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WF_TIE_P();
    __builtin_amdgcn_s_barrier();                       // every thread has its patch of step 0: the raw tile may be overwritten
    dma_raw(KT > 1 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) rows_of(j);
    planes_of(0, 0);
    planes_of(1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const unsigned sw = (unsigned)(((lane >> 5) ^ ((lane >> 3) & 1)) * 16);
    const unsigned a_rd = lds0 + (unsigned)(WF_A_OFF + (32 * wm + (lane & 31)) * 32) + sw;
    const unsigned b_rd = lds0 + (unsigned)(WF_B_OFF + (32 * wn + (lane & 31)) * 32) + sw;

    for (int g = 0; g < KT; ++g) {
        const unsigned cur = (unsigned)(g & 1) * WF_OPER, nxt = cur ^ WF_OPER;
        const unsigned ab = a_rd + cur, bb = b_rd + cur;
        const int g1 = g + 1 < KT ? g + 1 : g, g2 = g + 2 < KT ? g + 2 : KT - 1;      // past the end: valid steps again (never consumed)
        f32x4 fa[2], fb[2];
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0]) : "v"(ab));
        asm volatile("ds_read_b128 %0, %1" : "=v"(fb[0]) : "v"(bb));
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int c = p & 1;
            if (p < 15) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[c ^ 1]) : "v"(ab), "n"((p + 1) * WF_PLANE));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[c ^ 1]) : "v"(bb), "n"((p + 1) * WF_PLANE));
            }
            // Side work in small pieces behind the MFMAs of the planes.  The raw tile holds step g + 1 when the step begins: read it
            // (p = 0, behind the DMA of the weight image of step g + 1), barrier once every thread has its patch (p = 1), refill it for step
            // g + 2 by DMA; transform (p = 2..5), A image of step g + 1 (p = 6, 7).
            if (p == 0) {
                if (!(ABL & 2)) dma_b(g1);          // first thing after the barrier: a whole step of cover for the weight image
                read_patch(g1);
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[c]), "+v"(fb[c]) : "n"(XFORM ? 15 : 14));      // 4-bit field: 15 still retires this slot's fragments (issued first)
            } else if (p == 1) {
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[c]), "+v"(fb[c]));                    // ... also retires the patch reads
                WF_TIE_P();
                __builtin_amdgcn_s_barrier();
                if (!(ABL & 1)) dma_raw(g2);
            } else if (p < 15) {
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[c]), "+v"(fb[c]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c]), "+v"(fb[c]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][e], fb[c][e], acc[p], 0, 0, 0);
            if (p >= 2 && p < 6 && !(ABL & 4)) rows_of(p - 2);
            if (p == 6 && !(ABL & 4)) planes_of(0, nxt);
            if (p == 7 && !(ABL & 4)) planes_of(1, nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // raw tile of step g + 2 and weight image of step g + 1 have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // ... and this thread's A-image stores
        __builtin_amdgcn_s_barrier();
    }
#undef WF_TIE_P

    // ---- epilogue: output transform from the accumulators, bias, NHWC store, BatchNorm partial sums
    const int co = n0 + 32 * wn + (lane & 31);
    const float bv = a.bias ? a.bias[co] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    const int WC = a.W * a.Cout;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = 32 * wm + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int2 ent = *reinterpret_cast<const int2*>(lds + WF_TAB_OFF + i * 8);
        float r[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[0][j] = (acc[j][e] + acc[4 + j][e]) + acc[8 + j][e];
            r[1][j] = (acc[4 + j][e] - acc[8 + j][e]) - acc[12 + j][e];
        }
        float* zp = a.z + ent.x + co;
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) {
            const float y0 = ((r[aa][0] + r[aa][1]) + r[aa][2]) + bv;
            const float y1 = ((r[aa][1] - r[aa][2]) - r[aa][3]) + bv;
            const bool rowok = (ent.y & 1) && (aa == 0 || (ent.y & 2));
            if (rowok) {
                zp[aa * WC] = y0;
                s1 += y0; s2 += y0 * y0;
                if (ent.y & 4) {
                    zp[aa * WC + a.Cout] = y1;
                    s1 += y1; s2 += y1 * y1;
                }
            }
        }
    }
    if (a.stats != nullptr) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        float* red = reinterpret_cast<float*>(lds + WF_RED_OFF);
        if (lane < 32) { red[(wm * 64 + 32 * wn + lane) * 2] = s1; red[(wm * 64 + 32 * wn + lane) * 2 + 1] = s2; }
        __syncthreads();
        if (tid < 64) {
            float* o = a.stats + ((size_t)mb * a.Cout + n0 + tid) * 2;
            o[0] = red[tid * 2] + red[(64 + tid) * 2];
            o[1] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
        }
    }
}

unsigned wf_magic(int d) { return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

}  // namespace

extern "C" {

int64_t ssdhip_wino_fused_weight_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cout % WF_COUT || Cin % 16) return SSDHIP_EINVAL;
    return (int64_t)16 * Cout * Cin;
}

int ssdhip_wino_fused_pack(const float* w_oihw, float* upk, int Cout, int Cin, void* stream) {
    SSDHIP_REQUIRE(w_oihw && upk && Cout > 0 && Cin > 0 && Cout % WF_COUT == 0 && Cin % 16 == 0,
                   "wino_fused_pack: Cout must be a multiple of 64 and Cin of 16");
    const int64_t total = (int64_t)16 * Cout * Cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wino_fused_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, upk, Cout, Cin);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_wino_fused_stat_rows(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return SSDHIP_EINVAL;
    const int64_t T = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2);
    if (T >= (1ll << 31) - 64) return SSDHIP_EINVAL;
    return (int)((T + WF_TILES - 1) / WF_TILES);
}

int ssdhip_wino_fused_conv(const float* x, const float* upk, const float* bias, const float* in_scale, const float* in_shift, float* z,
                           float* stats, int N, int H, int W, int Cin, int Cout, void* stream) {
    SSDHIP_REQUIRE(x && upk && z && N > 0 && H > 0 && W > 0, "wino_fused_conv: bad arguments");
    SSDHIP_REQUIRE(Cin > 0 && Cin % 16 == 0 && Cin <= WF_MAXC && Cout > 0 && Cout % WF_COUT == 0,
                   "wino_fused_conv: Cin must be a multiple of 16 (<= 1024) and Cout of 64");
    SSDHIP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "wino_fused_conv: in_scale and in_shift go together");
    SSDHIP_REQUIRE((int64_t)N * H * W * Cin * 4 < (1ll << 31) && (int64_t)N * H * W * Cout < (1ll << 31),
                   "wino_fused_conv: tensors must stay below 2 GiB (32-bit offsets)");
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    // 64 consecutive tiles touch at most 16 tile rows (raw tile: 512 + 8 slots per segment <= 640) and the fast divisions hold
    SSDHIP_REQUIRE(TW >= 5 && (int64_t)N * TH < 65536, "wino_fused_conv: map too narrow (W >= 9) or too many tile rows");
    static DeviceOnce attr;
    if (const int dev = attr.pending(); dev != DeviceOnce::kDone) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS) != hipSuccess) {
            ssdhip_set_error("wino_fused_conv: cannot enable 158 KiB dynamic LDS");
            return SSDHIP_ELAUNCH;
        }
        attr.mark(dev);
    }
    int abl = 0;
#ifdef SSDHIP_PROBES
    if (const char* e = getenv("SSDHIP_WF_ABL")) abl = atoi(e);      // probe library only (scripts/wf_bench.py)
#endif
    WfArgs a;
    a.x = x; a.upk = upk; a.bias = bias; a.scale = in_scale; a.shift = in_shift; a.z = z; a.stats = stats;
    a.H = H; a.W = W; a.C = Cin; a.Cout = Cout; a.TH = TH; a.TW = TW;
    a.T = N * TH * TW; a.KT = Cin / WF_BK; a.NB = Cout / WF_COUT; a.RT = N * TH;
    a.x_bytes = (unsigned)((int64_t)N * H * W * Cin * 4);
    a.magic_tw = wf_magic(TW); a.magic_th = wf_magic(TH);
    const int MB = (a.T + WF_TILES - 1) / WF_TILES;
    dim3 grid((unsigned)((int64_t)MB * a.NB)), block(256);
    hipStream_t st = (hipStream_t)stream;
#ifdef SSDHIP_PROBES
#define WF_ABL_CASE(k) case k: { static DeviceOnce once; if (const int d = once.pending(); d != DeviceOnce::kDone) { \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<true, k>), hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS); once.mark(d); } \
        hipLaunchKernelGGL((wino_fused_kernel<true, k>), grid, block, WF_LDS, st, a); SSDHIP_CHECK_LAUNCH(); return MB; }
    if (abl && in_scale) switch (abl) { WF_ABL_CASE(1) WF_ABL_CASE(2) WF_ABL_CASE(3) WF_ABL_CASE(4) WF_ABL_CASE(7) default: break; }
#undef WF_ABL_CASE
#endif
    (void)abl;
    if (in_scale) hipLaunchKernelGGL(wino_fused_kernel<true>, grid, block, WF_LDS, st, a);
    else hipLaunchKernelGGL(wino_fused_kernel<false>, grid, block, WF_LDS, st, a);
    SSDHIP_CHECK_LAUNCH();
    return MB;      // rows of the BatchNorm partial table
}

}  // extern "C"
