// Plane GEMM for the Winograd domain: M[p][t][co] = sum_c V[p][t][c] * U[p][co][c], p = 0..15, fp32 on v_mfma_f32_32x32x2_f32.
// Same tile shape, LDS image (128-B rows, 16-B chunks XOR-swizzled with (row >> 1) & 7) and MFMA feed as conv_igemm.hip, but
// both operands are plain row-major planes (no gather, no padding, no fused transform), so the staging is done by the LDS-DMA
// engine instead of registers:
//
//   global_load_lds (16 B per lane)   a wave-instruction lands 8 rows x 128 B straight in LDS (destination = wave-uniform base
//                                     + lane * 16 B); the swizzle is applied on the SOURCE side: the lane that owns physical
//                                     chunk q of row r fetches logical chunk q ^ ((r >> 1) & 7).  No VGPR staging, no ds_write,
//                                     no VALU: per K-step a wave issues (BM + BN) / 32 DMA instructions and 16*MT*NT MFMAs.
//   one-ahead double buffer           K-step g: DMA of step g+1 into the other buffer (its readers passed the previous barrier),
//                                     MFMAs on buffer g & 1, then s_waitcnt vmcnt(0) + s_barrier: a full K-step of latency cover
//                                     (fragment ds_reads are asm with explicit lgkmcnt waits, see the loop)
//   measured (scripts/plane_gemm_probe.py, conv_4_2 planes): 122-126 TF; without the DMA stream 133-137 TF, without the
//   epilogue stores 130 TF; dropping the vmcnt wait or the barrier changes nothing (not latency-bound); grid quantisation
//   (5,824 tiles over 512 resident slots) costs 5 %
//   persistent blocks                 a block walks `tpb` consecutive m-tiles of one n-tile as one pipeline (K = Cin is short:
//                                     4..32 K-steps), storing each finished tile from inside the loop
//   tile order                        XCD-aware, n-tiles of an m-chunk adjacent (V rows are read once per n-tile from L2)
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int BK = 32, LROW = 32;

struct PlaneArgs {
    const float* v;      // (planes, rows_per_plane, K)
    const float* u;      // (planes, u_rows, K)
    float* y;            // (planes * rows_per_plane, Cout)
    long long u_plane_stride;
    int M, K, KT, Cout, rows_per_plane, MB, NB, tpb;
    long long u_term_stride;      // split-bf16 form with pre-split weights: elements between the three bf16 term images of U
};

template <int BM, int BN>
struct PlaneWaves { static constexpr int value = (BM + BN == 256) ? 2 : 3; };

// NW = 4: 2 x 2 waves of (BM/2) x (BN/2); NW = 8: 2 x 4 waves of (BM/2) x (BN/4) on the same block tile (twice the waves per SIMD for the
// same LDS footprint: more latency cover for the barrier and the DMA stream, 1.25x the LDS fragment reads)
template <int BM, int BN, int ABL = 0, int NW = 4, int KTF = 0>      // KTF: K-steps per tile when known at compile time (K = 128: 4), else 0
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, PlaneWaves<BM, BN>::value * NW / 4))) void gemm_planes_dma_kernel(PlaneArgs a) {
    constexpr int WNC = NW / 2;                                    // waves along n
    constexpr int RPP = 8 * NW;                                    // operand rows one DMA pass of the block covers
    constexpr int WROWS = BM / 2, WCOLS = BN / WNC, MT = WROWS / 32, NT = WCOLS / 32, AL = BM / RPP, BL = BN / RPP;
    __shared__ __attribute__((aligned(1024))) float lds[2 * (BM + BN) * LROW];
    float* As = lds;
    float* Bs = lds + 2 * BM * LROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNC, wn = wave % WNC;
    const int KT = KTF ? KTF : a.KT;

    int t;
    {
        const int T = ((a.MB + a.tpb - 1) / a.tpb) * a.NB, bid = blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3, q = T >> 3, r = T & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int mchunk = t / a.NB, nt_idx = t - mchunk * a.NB;
    const int mt0 = mchunk * a.tpb, ntl = min(a.tpb, a.MB - mt0);
    const int m0 = mt0 * BM, n0 = nt_idx * BN;

    // DMA source offsets (bytes): lane (lrow, chunk) of pass i fetches logical chunk `chunk ^ swz` of row 32 i + lrow
    const int lrow = tid >> 3, pchunk = tid & 7;
    const unsigned src_chunk = (unsigned)((pchunk ^ ((lrow >> 1) & 7)) * 16);
    const char* vb = reinterpret_cast<const char*>(a.v);
    const unsigned rowK = (unsigned)a.K * 4u;                       // bytes per operand row
    unsigned a_off = (unsigned)(m0 + lrow) * rowK + src_chunk;      // + 32 i rows, + K-step * 128
    const unsigned b_off = (unsigned)(n0 + lrow) * rowK + src_chunk;
    // wave-uniform LDS destinations: wave w of pass i owns rows 32 i + 8 w .. + 7 (1 KiB)
    float* a_dst = As + (8 * wave) * LROW;
    float* b_dst = Bs + (8 * wave) * LROW;

    int ld_kt = 0, ld_tile = 0;            // loader position (K-step inside the tile, tile inside the block)
    const char* ub = reinterpret_cast<const char*>(a.u) + (size_t)(m0 / a.rows_per_plane) * (size_t)a.u_plane_stride * 4;
    auto issue = [&](int buf) {            // DMA of the loader's current K-step into LDS buffer `buf`, then advance the loader
        const unsigned kb = (unsigned)ld_kt * (BK * 4);
#pragma unroll
        for (int i = 0; i < AL; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + (a_off + (unsigned)(RPP * i) * rowK + kb)),
                                             (lptr_t)(a_dst + (buf * BM + RPP * i) * LROW), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BL; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(ub + (b_off + (unsigned)(RPP * i) * rowK + kb)),
                                             (lptr_t)(b_dst + (buf * BN + RPP * i) * LROW), 16, 0, 0);
        if (++ld_kt == KT) {
            if (ld_tile + 1 < ntl) {
                ld_kt = 0; ++ld_tile;
                a_off += (unsigned)BM * rowK;
                ub = reinterpret_cast<const char*>(a.u) + (size_t)((m0 + ld_tile * BM) / a.rows_per_plane) * (size_t)a.u_plane_stride * 4;
            } else {
                ld_kt = KT - 1;           // past the end: keep re-fetching the last step (never consumed)
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    issue(0);
    __syncthreads();                       // emits s_waitcnt vmcnt(0): step 0 has landed for every wave

    // Fragment reads are written as asm: the compiler cannot tell an LDS-DMA write into the OTHER buffer from the ds_reads of
    // this one and would put s_waitcnt vmcnt(0) in front of the first ds_read after the DMA issue (one quarter of a K-step of
    // latency cover instead of a whole one).  Waits are explicit; the "+v" operands tie each wait to the registers it covers.
    const int frow = lane & 31, fh = lane >> 5, fx = (frow >> 1) & 7;
    unsigned fko[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) fko[kc] = (unsigned)(((2 * kc + fh) ^ fx) * 16);
    const unsigned a_lds = (unsigned)(uintptr_t)(lptr_t)(As + (wm * WROWS + frow) * LROW);
    const unsigned b_lds = (unsigned)(uintptr_t)(lptr_t)(Bs + (wn * WCOLS + frow) * LROW);
    const int total = ntl * KT;
    int kt_in = 0, m_cur = m0;
    for (int g = 0; g < total; ++g) {
        const int buf = g & 1;
        const unsigned ab = a_lds + (unsigned)(buf * BM * LROW * 4), bb = b_lds + (unsigned)(buf * BN * LROW * 4);
        f32x4 fa[2][MT], fb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[0][i]) : "v"(ab + fko[0]), "n"(i * 32 * LROW * 4));
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[0][j]) : "v"(bb + fko[0]), "n"(j * 32 * LROW * 4));
        if (ABL != 2) issue(buf ^ 1);                    // next K-step streams into the other buffer while this one is consumed
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int cur = kc & 1;
            if (kc < 3) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[cur ^ 1][i]) : "v"(ab + fko[kc < 3 ? kc + 1 : 0]), "n"(i * 32 * LROW * 4));
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[cur ^ 1][j]) : "v"(bb + fko[kc < 3 ? kc + 1 : 0]), "n"(j * 32 * LROW * 4));
            }
            // the fragments of `cur` were issued one kc earlier: everything but the (MT + NT) reads just issued has landed
            if constexpr (MT == 2 && NT == 2) {
                if (kc < 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[cur][0]), "+v"(fa[cur][1]), "+v"(fb[cur][0]), "+v"(fb[cur][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cur][0]), "+v"(fa[cur][1]), "+v"(fb[cur][0]), "+v"(fb[cur][1]));
            } else if constexpr (MT == 2) {
                if (kc < 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[cur][0]), "+v"(fa[cur][1]), "+v"(fb[cur][0]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cur][0]), "+v"(fa[cur][1]), "+v"(fb[cur][0]));
            } else if constexpr (NT == 2) {
                if (kc < 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[cur][0]), "+v"(fb[cur][0]), "+v"(fb[cur][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cur][0]), "+v"(fb[cur][0]), "+v"(fb[cur][1]));
            } else {
                if (kc < 3) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[cur][0]), "+v"(fb[cur][0]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cur][0]), "+v"(fb[cur][0]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int el = 0; el < 4; ++el)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][el], fb[cur][j][el], acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++kt_in == KT) {             // tile finished: dense rows of M, then restart the accumulators
            kt_in = 0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = n0 + wn * WCOLS + j * 32 + (lane & 31);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = m_cur + wm * WROWS + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        if (co < a.Cout && (ABL != 1 || acc[i][j][e] == 123.456f)) a.y[(size_t)m * a.Cout + co] = acc[i][j][e];
                        acc[i][j][e] = 0.f;
                    }
            }
            m_cur += BM;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next K-step has landed ...
        __builtin_amdgcn_s_barrier();                      // ... and so has everyone else's; this buffer is free again
    }
}


// ---- Split-bf16 form (round 4; OPT-IN, compute = 2; never the default: the headline keeps exact fp32 products).  Same planes, same fp32
// operands in HBM and LDS, same tile walk as gemm_planes_dma_kernel<128, 128, 0, 8>; only the inner product changes: every fp32 operand
// value is split in registers into three bf16 terms (a = a1 + a2 + a3 exactly: 8 + 8 + 8 significant bits) and the six cross products
// a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 -- everything down to 2^-24 of the product -- run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
// 12 MFMAs of 32 cycles per 16 k and wave where the fp32 form issues 16 of 64.  Measured against fp64 (scripts/split_bf16_probe.py,
// profiles/r04_d_*): 3x SMALLER error than the fp32 MFMA (the bf16 instruction sums 16 products before it rounds into the accumulator).
typedef __bf16 pl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pl_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pl_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pl_pack(float lo, float hi) {          // round-to-nearest-even pair, lo in bits 0-15
    const pl_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pl_bf16x2));
}
// two fp32 values -> the packed pairs of their three bf16 terms
__device__ __forceinline__ void pl_split(float x, float y, unsigned& t1, unsigned& t2, unsigned& t3) {
    t1 = pl_pack(x, y);
    const float rx = x - __uint_as_float(t1 << 16), ry = y - __uint_as_float(t1 & 0xffff0000u);
    t2 = pl_pack(rx, ry);
    t3 = pl_pack(rx - __uint_as_float(t2 << 16), ry - __uint_as_float(t2 & 0xffff0000u));
}
// eight consecutive k of one row -> three MFMA operands
__device__ __forceinline__ void pl_split8(const f32x4& lo, const f32x4& hi, pl_bf16x8 (&t)[3]) {
    unsigned w[3][4];
    pl_split(lo.x, lo.y, w[0][0], w[1][0], w[2][0]);
    pl_split(lo.z, lo.w, w[0][1], w[1][1], w[2][1]);
    pl_split(hi.x, hi.y, w[0][2], w[1][2], w[2][2]);
    pl_split(hi.z, hi.w, w[0][3], w[1][3], w[2][3]);
    pl_u32x4 u[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) u[i] = (pl_u32x4){w[i][0], w[i][1], w[i][2], w[i][3]};
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = __builtin_bit_cast(pl_bf16x8, u[i]);
}

// NW = 8: 2 x 4 waves of 64 x 32 (two blocks per CU); NW = 4: 2 x 2 waves of 64 x 64 -- a value of A is then split by two waves instead of four
// and a value of B serves two m-tiles AND two n-tiles: 32 splits per 24 MFMAs instead of 24 per 12 (the form is bound by the split's VALU work)
// BPRE: the weights arrive PRE-SPLIT (ssdhip_split_bf16x3 of the fp32 planes U: three bf16 images, a.u_term_stride elements apart) -- they
// change once per optimizer step (frozen layers: never), so their split does not belong in the K loop.  LDS then holds, per stage, the fp32
// A tile (16 KiB) and three 64-B-row bf16 B tiles (24 KiB; 16-B chunks XOR-swizzled with (row >> 2) & 3: conflict-free ds_read_b128 for
// the MFMA feed): 80 KiB per block, two blocks per CU.  Only A is split in registers: 88 VALU operations per 24 MFMAs instead of 176.
template <int NW, bool BPRE = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, NW == 8 ? 4 : 2))) void gemm_planes_split_kernel(PlaneArgs a) {
    constexpr int BM = 128, BN = 128, WNC = NW / 2, RPP = 8 * NW, WROWS = 64, WCOLS = BN / WNC, MT = 2, NT = WCOLS / 32, AL = BM / RPP, BL = BN / RPP;
    constexpr int BSTAGE = BPRE ? 3 * BN * 16 : BN * LROW;          // floats of LDS per stage of B: three bf16 tiles of 64-B rows, or one fp32 tile
    __shared__ __attribute__((aligned(1024))) float lds[2 * BM * LROW + 2 * BSTAGE];
    float* As = lds;
    float* Bs = lds + 2 * BM * LROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNC, wn = wave % WNC;
    const int KT = a.KT;
    int t;
    {
        const int T = ((a.MB + a.tpb - 1) / a.tpb) * a.NB, bid = blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3, q = T >> 3, r = T & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int mchunk = t / a.NB, nt_idx = t - mchunk * a.NB;
    const int mt0 = mchunk * a.tpb, ntl = min(a.tpb, a.MB - mt0);
    const int m0 = mt0 * BM, n0 = nt_idx * BN;
    const int lrow = tid >> 3, pchunk = tid & 7;
    const unsigned src_chunk = (unsigned)((pchunk ^ ((lrow >> 1) & 7)) * 16);
    const char* vb = reinterpret_cast<const char*>(a.v);
    const unsigned rowK = (unsigned)a.K * 4u;
    unsigned a_off = (unsigned)(m0 + lrow) * rowK + src_chunk;
    const unsigned b_off = (unsigned)(n0 + lrow) * rowK + src_chunk;
    float* a_dst = As + (8 * wave) * LROW;
    float* b_dst = Bs + (8 * wave) * LROW;
    int ld_kt = 0, ld_tile = 0;
    constexpr int ESZ = BPRE ? 2 : 4;                               // bytes per weight element
    const char* ub = reinterpret_cast<const char*>(a.u) + (size_t)(m0 / a.rows_per_plane) * (size_t)a.u_plane_stride * ESZ;
    // pre-split B: a wave-instruction lands 16 rows x 64 B; lane (row = lane >> 2, physical chunk = lane & 3) fetches the logical chunk
    // (lane & 3) ^ ((row >> 2) & 3) of its row; wave w of pass i owns rows 16 (NW i + w) .. + 15
    const int brow = lane >> 2;
    const unsigned bsrc = (unsigned)(((lane & 3) ^ ((brow >> 2) & 3)) * 16);
    const unsigned rowKb = (unsigned)a.K * 2u;
    auto issue = [&](int buf) {
        const unsigned kb = (unsigned)ld_kt * (BK * 4);
#pragma unroll
        for (int i = 0; i < AL; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + (a_off + (unsigned)(RPP * i) * rowK + kb)), (lptr_t)(a_dst + (buf * BM + RPP * i) * LROW), 16, 0, 0);
        if constexpr (BPRE) {
            const unsigned kb2 = (unsigned)ld_kt * (BK * 2);
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                for (int i = 0; i < BN / (16 * NW); ++i) {
                    const int r0 = 16 * (NW * i + wave);
                    __builtin_amdgcn_global_load_lds((gptr_t)(ub + (size_t)t3 * (size_t)a.u_term_stride * 2 + ((unsigned)(n0 + r0 + brow) * rowKb + kb2 + bsrc)),
                                                     (lptr_t)(Bs + buf * BSTAGE + (t3 * BN + r0) * 16), 16, 0, 0);
                }
        } else {
#pragma unroll
            for (int i = 0; i < BL; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(ub + (b_off + (unsigned)(RPP * i) * rowK + kb)), (lptr_t)(b_dst + (buf * BN + RPP * i) * LROW), 16, 0, 0);
        }
        if (++ld_kt == KT) {
            if (ld_tile + 1 < ntl) {
                ld_kt = 0; ++ld_tile;
                a_off += (unsigned)BM * rowK;
                ub = reinterpret_cast<const char*>(a.u) + (size_t)((m0 + ld_tile * BM) / a.rows_per_plane) * (size_t)a.u_plane_stride * ESZ;
            } else {
                ld_kt = KT - 1;
            }
        }
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    issue(0);
    __syncthreads();
    // bf16 MFMA operand: lane (row = lane & 31, kh = lane >> 5) supplies k = 16 m + 8 kh .. + 7 of its row = the logical 16-B chunks
    // 4 m + 2 kh and + 1 of the 128-B LDS row; physical chunk = logical ^ ((row >> 1) & 7), so the pair stays inside one aligned 32 B
    const int frow = lane & 31, kh = lane >> 5, fx = (frow >> 1) & 7;
    unsigned fko[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { fko[m][0] = (unsigned)(((4 * m + 2 * kh) ^ fx) * 16); fko[m][1] = (unsigned)(((4 * m + 2 * kh + 1) ^ fx) * 16); }
    const unsigned a_lds = (unsigned)(uintptr_t)(lptr_t)(As + (wm * WROWS + frow) * LROW);
    const unsigned b_lds = BPRE ? (unsigned)(uintptr_t)(lptr_t)(Bs + (wn * WCOLS + frow) * 16) : (unsigned)(uintptr_t)(lptr_t)(Bs + (wn * WCOLS + frow) * LROW);
    // pre-split B fragment: lane (row, kh) reads the 16 B = k 16 m + 8 kh .. + 7 of term t3: logical chunk 2 m + kh of its 64-B row
    unsigned fkb[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) fkb[m] = (unsigned)(((2 * m + kh) ^ ((frow >> 2) & 3)) * 16);
    const int total = ntl * KT;
    int kt_in = 0, m_cur = m0;
    for (int g = 0; g < total; ++g) {
        const int buf = g & 1;
        const unsigned ab = a_lds + (unsigned)(buf * BM * LROW * 4), bb = b_lds + (unsigned)(buf * BSTAGE * 4);
        f32x4 fa[2][MT][2], fb[2][NT][BPRE ? 3 : 2];          // [k16 half][m- / n-tile][low / high 16 B of fp32, or the three bf16 terms]
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[m][i][h]) : "v"(ab + fko[m][h]), "n"(i * 32 * LROW * 4));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (BPRE) {
#pragma unroll
                    for (int t3 = 0; t3 < 3; ++t3)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[m][j][t3]) : "v"(bb + fkb[m]), "n"((t3 * BN + j * 32) * 64));
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[m][j][h]) : "v"(bb + fko[m][h]), "n"(j * 32 * LROW * 4));
                }
            }
        }
        issue(buf ^ 1);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            // the reads of half m were issued before those of half m + 1: (MT + NT) * 2 newer ones may stay outstanding for m = 0
            if constexpr (BPRE) {
                static_assert(!BPRE || NT == 2, "pre-split B is wired for the 2 x 2 wave layout");
                // per half m: MT * 2 + NT * 3 = 10 reads; those of half 1 may stay outstanding while half 0 is consumed
                if (m == 0) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][0][2]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]), "+v"(fb[0][1][2]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][0][2]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1]), "+v"(fb[1][1][2]));
            } else if constexpr (NT == 1) {
                if (m == 0) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]));
            } else {
                if (m == 0) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][NT - 1][0]), "+v"(fb[0][NT - 1][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][NT - 1][0]), "+v"(fb[1][NT - 1][1]));
            }
            pl_bf16x8 ta[MT][3], tb[NT][3];
#pragma unroll
            for (int i = 0; i < MT; ++i) pl_split8(fa[m][i][0], fa[m][i][1], ta[i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (BPRE) {
#pragma unroll
                    for (int t3 = 0; t3 < 3; ++t3) tb[j][t3] = __builtin_bit_cast(pl_bf16x8, fb[m][j][t3]);
                } else {
                    pl_split8(fb[m][j][0], fb[m][j][1], tb[j]);
                }
            }
            // small terms first: (a3 b1) (a2 b2) (a1 b3) (a2 b1) (a1 b2) (a1 b1)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[i][2], tb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[i][1], tb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[i][0], tb[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[i][1], tb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[i][0], tb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[i][0], tb[j][0], acc[i][j], 0, 0, 0);
                }
        }
        if (++kt_in == KT) {
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// fp32 array -> its three bf16 term images (dst: 3 x n bf16, term-major): x = t1 + t2 + t3 exactly
__global__ void split_bf16x3_kernel(const float2* __restrict__ src, unsigned* __restrict__ dst, long long n2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = src[i];
        unsigned t1, t2, t3;
        pl_split(v.x, v.y, t1, t2, t3);
        dst[i] = t1; dst[n2 + i] = t2; dst[2 * n2 + i] = t3;
    }
}

// ---- Work-queue form of the 128 x 128 / 8-wave kernel (round 4): the dominant launches of the step (conv_3_2 .. conv_4_3 planes: 3,600 and
// 6,552 tiles of 8 or 16 K-steps) lose 10-15 % to tile quantisation in the form above -- 3,600 one-tile blocks over 512 resident slots
// are 7.03 rounds paid as 8, 6,552 tiles in three-tile blocks are 12.8 tile-times paid as 15 (scripts/plane_k_probe.py: the asymptote
// of the K loop is 131 TF, a 16-step tile reaches 121-125) -- and re-pay launch, first-DMA latency and store drain per block.  Here:
//   grid      2 blocks per CU, persistent; block -> XCD = blockIdx & 7 (the hardware's round-robin placement)
//   queues    one atomic work queue per XCD over that XCD's contiguous eighth of the tiles (n-tiles of an m-chunk adjacent, as above:
//             the blocks of an XCD work on neighbouring tiles at any time, so a V tile is fetched once into their L2); dynamic, so
//             a CU that shares its time with another stream's kernels simply takes fewer tiles
//   pipeline  a block knows its next item before it starts the current one (items are grabbed two ahead), so the first K-step of the
//             next tile streams in under the last K-step of the current one: one pipeline fill per launch instead of one per block
//   last round  when the tiles of an XCD leave R < S / 2 tiles for the last round of its S blocks, those R tiles are queued as
//             R * SP pieces of KT / SP K-steps; a piece stores its 128 x 128 partial sums to the workspace, and the block that
//             arrives LAST at the tile's counter adds the SP partials in slice order (fixed order: bit-reproducible whatever the
//             arrival order) and writes M.  The last round then costs KT / SP K-steps + one 64-KiB-per-slice reduction.
// Control words (first 4 KiB of the workspace, zero before the first launch, left zero by every launch): queue heads [0..7], exited
// blocks [8], piece counters [64 + 64 xcd + tile].
constexpr int PQ_CTL_BYTES = 4096;      // control words
struct PlaneQArgs {
    const float* v;
    const float* u;
    float* y;
    int* ctl;
    float* part;
    long long u_plane_stride;
    int M, K, KT, Cout, rows_per_plane, MB, NB, S;
};

// Queue grab of the work-queue kernel as a raw instruction WITHOUT a wait: the compiler puts s_waitcnt vmcnt(0) behind a
// memory operation whose result it sees (at the join of `if (thread 0)` that is every K-step, right behind the DMA issue: no latency cover
// left, half the speed).  The K loop ends every step with an explicit s_waitcnt vmcnt(0); results are only used behind one.
__device__ __forceinline__ int pq_grab_nowait(int* q) {
    int r, one = 1;
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(r) : "v"(q), "v"(one) : "memory");
    return r;
}
// Sum of the SP slice partials of one 128 x 128 tile in slice order (bit-reproducible whatever the arrival order), written to M.  Kept out
// of line: it runs once per remainder tile, and inlined it costs the K loop 30 registers.
__device__ __noinline__ void plane_piece_reduce(const float* part, int SP, float* y, int m0, int n0, int Cout, int tid) {
    // (thread 0 ran the agent acquire -- this CU's L1 is invalidated -- and every wave passed a barrier behind its vmcnt(0))
    const float4* src = reinterpret_cast<const float4*>(part);
    for (int k = 0; k < (128 * 128 / 4) / 512; ++k) {
        const int e4 = tid + 512 * k, row = e4 >> 5, c4 = e4 & 31;
        float4 s = src[e4];
        for (int sl = 1; sl < SP; ++sl) {
            const float4 t = src[(size_t)sl * (128 * 128 / 4) + e4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int co = n0 + 4 * c4;
        float* o = y + ((size_t)(m0 + row) * Cout + co);
        if (co + 3 < Cout) *reinterpret_cast<float4*>(o) = s;
        else { if (co < Cout) o[0] = s.x; if (co + 1 < Cout) o[1] = s.y; if (co + 2 < Cout) o[2] = s.z; }
    }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_planes_queue_kernel(PlaneQArgs a) {
    constexpr int BM = 128, BN = 128, WNC = 4, RPP = 64, WROWS = 64, WCOLS = 32, MT = 2, AL = 2, BL = 2;
    __shared__ __attribute__((aligned(1024))) float lds[2 * (BM + BN) * LROW];
    float* As = lds;
    float* Bs = lds + 2 * BM * LROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNC, wn = wave % WNC;
    const int KT = a.KT, S = a.S;
    const int xcd = blockIdx.x & 7;
    // this XCD's tiles [base, base + Tx); F full rounds of S tiles, then R tiles as R * SP pieces
    const int T = a.MB * a.NB, tq = T >> 3, tr = T & 7;
    const int base = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, Tx = tq + (xcd < tr ? 1 : 0);
    const int F = Tx / S, R = Tx - F * S;
    int SP = 1;
    if (R > 0 && 2 * R <= S) { SP = S / R; if (SP > KT) SP = KT; if (SP > 16) SP = 16; }
    const int n_full = SP > 1 ? F * S : Tx, n_items = n_full + (SP > 1 ? R * SP : 0);      // SP == 1: the last round's tiles are ordinary items
    int* qhead = a.ctl + xcd;
    // Block-wide broadcasts of thread 0's queue grabs go through two words of LDS behind a workgroup barrier.  (A mailbox in global memory,
    // read by every wave for itself, was tried to keep the block at exactly 64 KiB of LDS: one wave reading a stale word while another
    // read the fresh one sends the two into different items -- mismatched barrier counts, a hung kernel.  64 KiB + 16 B still fits twice
    // into the CU's 160 KiB.)
    __shared__ int pub[4];                 // [0]: next item id; [2]: "this block reduces the tile"

    // an item is wave-uniform: every field is forced into scalar registers (the integer divisions below run on the vector unit)
    struct Item { int m0, n0, k0, k1, piece, pt, plane; };
    auto decode = [&](int id) {
        Item it;
        int tile = base + id, k0 = 0, k1 = KT, piece = -1, pt = 0;
        if (id >= n_full) {
            const int j = id - n_full, sl = j % SP;
            pt = j / SP;
            tile = base + n_full + pt;
            k0 = sl * KT / SP; k1 = (sl + 1) * KT / SP;
            piece = xcd * S + j;
        }
        const int mchunk = tile / a.NB;
        it.m0 = __builtin_amdgcn_readfirstlane(mchunk * BM);
        it.n0 = __builtin_amdgcn_readfirstlane((tile - mchunk * a.NB) * BN);
        it.k0 = __builtin_amdgcn_readfirstlane(k0); it.k1 = __builtin_amdgcn_readfirstlane(k1);
        it.piece = __builtin_amdgcn_readfirstlane(piece); it.pt = __builtin_amdgcn_readfirstlane(pt);
        it.plane = __builtin_amdgcn_readfirstlane(mchunk * BM / a.rows_per_plane);
        return it;
    };

    // the first item of a block is its slot index (no round trip before the first DMA); the queue hands out ids from S on
    int cur_id = (int)(blockIdx.x >> 3);

    const int lrow = tid >> 3, pchunk = tid & 7;
    const unsigned src_chunk = (unsigned)((pchunk ^ ((lrow >> 1) & 7)) * 16);
    const char* vb = reinterpret_cast<const char*>(a.v);
    const unsigned rowK = (unsigned)a.K * 4u;
    float* a_dst = As + (8 * wave) * LROW;
    float* b_dst = Bs + (8 * wave) * LROW;
    auto issue = [&](const Item& it, int kt, int buf) {
        const unsigned kb = (unsigned)kt * (BK * 4);
        const unsigned a_off = (unsigned)(it.m0 + lrow) * rowK + src_chunk + kb, b_off = (unsigned)(it.n0 + lrow) * rowK + src_chunk + kb;
        const char* ub = reinterpret_cast<const char*>(a.u) + (size_t)it.plane * (size_t)a.u_plane_stride * 4;
#pragma unroll
        for (int i = 0; i < AL; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + (a_off + (unsigned)(RPP * i) * rowK)), (lptr_t)(a_dst + (buf * BM + RPP * i) * LROW), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BL; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(ub + (b_off + (unsigned)(RPP * i) * rowK)), (lptr_t)(b_dst + (buf * BN + RPP * i) * LROW), 16, 0, 0);
    };

    const int frow = lane & 31, fh = lane >> 5, fx = (frow >> 1) & 7;
    unsigned fko[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) fko[kc] = (unsigned)(((2 * kc + fh) ^ fx) * 16);
    const unsigned a_lds = (unsigned)(uintptr_t)(lptr_t)(As + (wm * WROWS + frow) * LROW);
    const unsigned b_lds = (unsigned)(uintptr_t)(lptr_t)(Bs + (wn * WCOLS + frow) * LROW);

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    int buf = 0;
#ifdef PQ_TRACE      // timing experiments only: per block (start, then per item: id, end time) in the partial-sum area behind the slices
    long long* trace = reinterpret_cast<long long*>(a.part + (size_t)8 * S * BM * BN) + (size_t)blockIdx.x * 64;
    int tn = 0;
    if (tid == 0) { trace[0] = wall_clock64(); trace[1] = cur_id; tn = 2; }
#endif
    if ((unsigned)cur_id < (unsigned)n_items) {
        Item cur = decode(cur_id);
        issue(cur, cur.k0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                   // the first K-step has landed for every wave
        while (true) {
            // The id of the NEXT item (one ahead: the blocks of an XCD then work on 64 consecutive tiles at any time, like consecutively
            // dispatched blocks would -- grabbing two ahead spread them over twice the span and cost conv_6_1 its L2 hits): atomic grab at
            // the item's first K-step (thread 0, no wait), mailbox store behind that step's vmcnt(0), acknowledged by the next step's;
            // every thread loads it at the top of the last-but-one K-step and decodes it behind that step's wait, in time for the last
            // K-step to stream the next tile's first operands in.  Items of fewer than 4 K-steps (split pieces) fetch it at their end.
            const bool fast = cur.k1 - cur.k0 >= 4;
            int nxt_id = n_items, nn = 0;
            bool known = false, prefetched = false;
            Item nx = cur;
            for (int kt = cur.k0; kt < cur.k1; ++kt) {
                const unsigned ab = a_lds + (unsigned)(buf * BM * LROW * 4), bb = b_lds + (unsigned)(buf * BN * LROW * 4);
                f32x4 fa[2][MT], fb[2];
#pragma unroll
                for (int i = 0; i < MT; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[0][i]) : "v"(ab + fko[0]), "n"(i * 32 * LROW * 4));
                asm volatile("ds_read_b128 %0, %1" : "=v"(fb[0]) : "v"(bb + fko[0]));
                if (kt + 1 < cur.k1) issue(cur, kt + 1, buf ^ 1);
                else if (known && (unsigned)nxt_id < (unsigned)n_items) { issue(nx, nx.k0, buf ^ 1); prefetched = true; }
                const bool grab = kt == cur.k0 && tid == 0;
                const bool fetch = fast && kt + 2 == cur.k1;
                int grabbed = 0;
                if (grab) grabbed = pq_grab_nowait(qhead);      // (not to be touched before the s_waitcnt at the end of the step)
                if (fetch) nn = pub[0];                          // published before this step's opening barrier (the previous step's closing one)
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const int c = kc & 1;
                    if (kc < 3) {
#pragma unroll
                        for (int i = 0; i < MT; ++i)
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[c ^ 1][i]) : "v"(ab + fko[kc < 3 ? kc + 1 : 0]), "n"(i * 32 * LROW * 4));
                        asm volatile("ds_read_b128 %0, %1" : "=v"(fb[c ^ 1]) : "v"(bb + fko[kc < 3 ? kc + 1 : 0]));
                        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[c][0]), "+v"(fa[c][1]), "+v"(fb[c]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c][0]), "+v"(fa[c][1]), "+v"(fb[c]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int el = 0; el < 4; ++el)
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][el], fb[c][el], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kt + 1 == cur.k1) {            // item finished: M rows (whole tile) or the partial sums of this K slice
                    const int col = wn * WCOLS + (lane & 31), co = cur.n0 + col;
                    const int ld = cur.piece < 0 ? a.Cout : BN;
                    const bool ok = cur.piece >= 0 || co < a.Cout;
                    // (the asm keeps the 32 row addresses from being hoisted out of the K loop, where they cost 64 registers and spill)
                    int row0 = wm * WROWS + 4 * (lane >> 5);
                    asm volatile("" : "+v"(row0));
                    float* dst = (cur.piece < 0 ? a.y + ((size_t)cur.m0 * a.Cout + co) : a.part + ((size_t)cur.piece * (BM * BN) + col)) + (size_t)row0 * ld;
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            if (ok) dst[(size_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * ld] = acc[i][e];
                            acc[i][e] = 0.f;
                        }
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(grabbed), "+v"(nn) : : "memory");
                if (grab) pub[0] = S + grabbed;
                if (fetch) {
                    nxt_id = __builtin_amdgcn_readfirstlane(nn);
                    known = true;
                    if ((unsigned)nxt_id < (unsigned)n_items) nx = decode(nxt_id);
                }
                __builtin_amdgcn_s_barrier();
                buf ^= 1;
            }
            if (cur.piece >= 0) {
                // Hand-off of the slice's partial sums to whichever block reduces the tile -- possibly on another XCD, whose L2 does not see
                // this XCD's dirty lines (the block -> XCD mapping above is a locality guess, not a contract).  The form MI355X_MICROARCH.md
                // lists as valid for plain stores: every storing wave has waited vmcnt(0) (end of the K-step) -> workgroup barrier -> ONE
                // lane: agent release (L2 write-back) -> vmcnt(0) -> agent-scope atomic add on the tile's counter; the block whose add
                // returns SP - 1 came last: agent acquire -> vmcnt(0) -> barrier -> plain loads in slice order.
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    int* cnt = a.ctl + 64 + 64 * xcd + cur.pt;
                    const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = old == SP - 1;
                    if (last) {
                        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left zero for the next launch
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    pub[2] = last;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (pub[2]) plane_piece_reduce(a.part + (size_t)(xcd * S + cur.pt * SP) * (BM * BN), SP, a.y, cur.m0, cur.n0, a.Cout, tid);
                __syncthreads();                                 // pub[2] may be rewritten by the next piece
            }
            if (!known) {                                        // short item: its grab is only now guaranteed to have reached the mailbox
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                nxt_id = __builtin_amdgcn_readfirstlane(pub[0]);
                __syncthreads();                                 // everyone has read it before thread 0 publishes the next grab
                if (nxt_id < n_items) nx = decode(nxt_id);
            }
#ifdef PQ_TRACE
            if (tid == 0 && tn < 62) { trace[tn++] = wall_clock64(); trace[tn++] = nxt_id; }
#endif
            if (nxt_id >= n_items) break;
            cur = nx;
            if (!prefetched) {
                issue(cur, cur.k0, buf);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    }
    // the last block out of the whole grid leaves the queues at zero for the next launch (every block has stopped grabbing by then)
    __syncthreads();
#ifdef PQ_TRACE
    if (tid == 0) { trace[tn] = wall_clock64(); trace[63] = tn; }
#endif
    if (tid == 0) {
        const int done = __hip_atomic_fetch_add(a.ctl + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            for (int i = 0; i < 9; ++i) __hip_atomic_store(a.ctl + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

// called from ssdhip_conv2d_nhwc (conv_igemm.hip) for plane GEMMs
#ifdef SSDHIP_PROBES
// libssdhip_probes.so only (scripts/plane_split_probe.py): the split-bf16 plane GEMM with PRE-SPLIT weights.  u3 = the three bf16 term images of the
// weight planes (ssdhip_probe_split_bf16x3), u_term_stride elements apart; M = all rows of V (planes x rows_per_plane, a multiple of 128).
extern "C" int ssdhip_probe_plane_gemm_bpre(const float* v, const void* u3, float* y, int M, int K, int Cout, int rows_per_plane,
                                            long long u_plane_stride, long long u_term_stride, int tpb, void* stream) {
    SSDHIP_REQUIRE(v && u3 && y && M > 0 && M % 128 == 0 && rows_per_plane % 128 == 0 && K >= 32 && K % 32 == 0 && Cout > 0 && tpb >= 1 &&
                       (long long)M * K * 4 < (1ll << 31) && (long long)cdiv(Cout, 128) * 128 * K * 2 < (1ll << 31),
                   "probe_plane_gemm_bpre: 128-row tiles, K a multiple of 32, 32-bit byte offsets inside V and inside one term image of a plane");
    PlaneArgs a;
    a.v = v; a.u = reinterpret_cast<const float*>(u3); a.y = y; a.u_plane_stride = u_plane_stride; a.u_term_stride = u_term_stride;
    a.M = M; a.K = K; a.KT = K / BK; a.Cout = Cout; a.rows_per_plane = rows_per_plane;
    a.MB = M / 128; a.NB = cdiv(Cout, 128); a.tpb = tpb;
    hipLaunchKernelGGL((gemm_planes_split_kernel<4, true>), dim3(cdiv(a.MB, a.tpb) * a.NB), dim3(256), 0, (hipStream_t)stream, a);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

extern "C" int ssdhip_probe_split_bf16x3(const float* src, void* dst3, long long n, void* stream) {
    SSDHIP_REQUIRE(src && dst3 && n > 0 && n % 2 == 0, "split_bf16x3: null pointer or odd element count");
    long long blocks = (n / 2 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(src),
                       reinterpret_cast<unsigned*>(dst3), n / 2);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}
#endif

// Workspace of the work-queue form: control words + one 128 x 128 partial tile per block slot.
long long ssdhip_gemm_planes_queue_ws_bytes(void) {
    const int S = ssdhip_cu_count() * 2 / 8;
    return PQ_CTL_BYTES + (long long)8 * S * 128 * 128 * 4 + 8 * S * 64 * 8;      // (+ the trace area of PQ_TRACE builds)
}

// 128 x 128 tiles, K >= 256: the persistent work-queue kernel.  ws: ssdhip_gemm_planes_queue_ws_bytes() bytes whose first 4 KiB were zero
// before the first launch on it (launches on one workspace must be stream-ordered).
int ssdhip_gemm_planes_queue(const float* v, const float* u, float* y, int M, int K, int Cout, int rows_per_plane, long long u_plane_stride,
                             void* ws, int max_cus, hipStream_t st) {
    PlaneQArgs a;
    a.v = v; a.u = u; a.y = y; a.u_plane_stride = u_plane_stride;
    a.ctl = reinterpret_cast<int*>(ws);
    a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + PQ_CTL_BYTES);
    a.M = M; a.K = K; a.KT = K / BK; a.Cout = Cout; a.rows_per_plane = rows_per_plane;
    a.MB = M / 128; a.NB = cdiv(Cout, 128);
    a.S = ssdhip_cu_count() * 2 / 8;
    // max_cus (ssdhip_conv_params) = CUs the persistent grid fills, two blocks each: a full grid holds every register of every CU for the
    // whole launch, so kernels of other streams only get in between launches.  Measured beside the trainable tail (engine: SSD_PQ_CUS): 208 of
    // 256 CUs 9.53 vs 9.57 ms per step over three alternating runs on one box (216: 9.62, 224: 9.53-9.59, 192: 9.57).  The workspace is sized
    // for the full grid either way.
    if (max_cus >= 8 && max_cus * 2 / 8 < a.S) a.S = max_cus * 2 / 8;
    if (a.S < 1 || a.S > 64) return SSDHIP_ENOSUP;      // piece counters: 64 per XCD
    hipLaunchKernelGGL(gemm_planes_queue_kernel, dim3(8 * a.S), dim3(512), 0, st, a);
    return SSDHIP_OK;
}

int ssdhip_gemm_planes_dma(const float* v, const float* u, float* y, int M, int K, int Cout, int rows_per_plane, long long u_plane_stride,
                           int BM, int BN, int tpb, hipStream_t st, int abl) {
    PlaneArgs a;
    a.v = v; a.u = u; a.y = y; a.u_plane_stride = u_plane_stride;
    a.M = M; a.K = K; a.KT = K / BK; a.Cout = Cout; a.rows_per_plane = rows_per_plane;
    a.MB = M / BM; a.NB = cdiv(Cout, BN); a.tpb = tpb;
    dim3 grid(cdiv(a.MB, a.tpb) * a.NB), block(256);
    if (abl == 100) {          // compute = 2: the split-bf16 form (opt-in; 128 x 128 tiles only)
        if (BM != 128 || BN != 128) return SSDHIP_ENOSUP;
        hipLaunchKernelGGL(gemm_planes_split_kernel<4>, grid, dim3(256), 0, st, a);
        return SSDHIP_OK;
    }
    if (abl == 101) {          // A/B: the 8-wave form of the same (scripts/plane_split_probe.py with SPLIT_NW=8)
        if (BM != 128 || BN != 128) return SSDHIP_ENOSUP;
        hipLaunchKernelGGL(gemm_planes_split_kernel<8>, grid, dim3(512), 0, st, a);
        return SSDHIP_OK;
    }
#ifdef SSDHIP_PROBES      // libssdhip_probes.so only: A/B variants and timing-only ablations (abl 1 / 2 return wrong results by design)
    if (abl == 4) { hipLaunchKernelGGL((gemm_planes_dma_kernel<128, 128, 0, 4>), grid, block, 0, st, a); return SSDHIP_OK; }      // A/B: the 4-wave block
    if (abl == 1) { hipLaunchKernelGGL((gemm_planes_dma_kernel<128, 128, 1>), grid, block, 0, st, a); return SSDHIP_OK; }
    if (abl == 2) { hipLaunchKernelGGL((gemm_planes_dma_kernel<128, 128, 2>), grid, block, 0, st, a); return SSDHIP_OK; }
    if (BM == 64 && BN == 128 && abl == 3) { hipLaunchKernelGGL((gemm_planes_dma_kernel<64, 128>), grid, block, 0, st, a); return SSDHIP_OK; }     // A/B: 4 waves
#else
    if (abl != 0) return SSDHIP_ENOSUP;
#endif
    if (BM == 128 && BN == 128 && K == 128) hipLaunchKernelGGL((gemm_planes_dma_kernel<128, 128, 0, 8, 4>), grid, dim3(512), 0, st, a);   // short K: 4 steps, known at compile time
    else if (BM == 128 && BN == 128) hipLaunchKernelGGL((gemm_planes_dma_kernel<128, 128, 0, 8>), grid, dim3(512), 0, st, a);   // 8 waves: +3 % (scripts/plane_gemm_probe.py)
    else if (BM == 128 && BN == 64) hipLaunchKernelGGL((gemm_planes_dma_kernel<128, 64>), grid, block, 0, st, a);
    else if (BM == 64 && BN == 128) hipLaunchKernelGGL((gemm_planes_dma_kernel<64, 128, 0, 8>), grid, dim3(512), 0, st, a);       // 2 x 4 waves of 32 x 32
    else return SSDHIP_ENOSUP;
    return SSDHIP_OK;
}
