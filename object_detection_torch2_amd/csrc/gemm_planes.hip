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

}  // namespace

// called from ssdhip_conv2d_nhwc (conv_igemm.hip) for plane GEMMs
int ssdhip_gemm_planes_dma(const float* v, const float* u, float* y, int M, int K, int Cout, int rows_per_plane, long long u_plane_stride,
                           int BM, int BN, int tpb, hipStream_t st, int abl) {
    PlaneArgs a;
    a.v = v; a.u = u; a.y = y; a.u_plane_stride = u_plane_stride;
    a.M = M; a.K = K; a.KT = K / BK; a.Cout = Cout; a.rows_per_plane = rows_per_plane;
    a.MB = M / BM; a.NB = cdiv(Cout, BN); a.tpb = tpb;
    dim3 grid(cdiv(a.MB, a.tpb) * a.NB), block(256);
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
