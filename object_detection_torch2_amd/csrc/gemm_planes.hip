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
//   persistent blocks                 a block walks `tpb` consecutive m-tiles of one n-tile as one pipeline (K = Cin is short:
//                                     4..32 K-steps), storing each finished tile from inside the loop
//   tile order                        XCD-aware, n-tiles of an m-chunk adjacent (V rows are read once per n-tile from L2)
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
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

template <int BM, int BN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, PlaneWaves<BM, BN>::value))) void gemm_planes_dma_kernel(PlaneArgs a) {
    constexpr int WROWS = BM / 2, WCOLS = BN / 2, MT = WROWS / 32, NT = WCOLS / 32, AL = BM / 32, BL = BN / 32;
    __shared__ __attribute__((aligned(1024))) float lds[2 * (BM + BN) * LROW];
    float* As = lds;
    float* Bs = lds + 2 * BM * LROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

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
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + (a_off + (unsigned)(32 * i) * rowK + kb)),
                                             (lptr_t)(a_dst + (buf * BM + 32 * i) * LROW), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BL; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(ub + (b_off + (unsigned)(32 * i) * rowK + kb)),
                                             (lptr_t)(b_dst + (buf * BN + 32 * i) * LROW), 16, 0, 0);
        if (++ld_kt == a.KT) {
            if (ld_tile + 1 < ntl) {
                ld_kt = 0; ++ld_tile;
                a_off += (unsigned)BM * rowK;
                ub = reinterpret_cast<const char*>(a.u) + (size_t)((m0 + ld_tile * BM) / a.rows_per_plane) * (size_t)a.u_plane_stride * 4;
            } else {
                ld_kt = a.KT - 1;           // past the end: keep re-fetching the last step (never consumed)
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

    const int frow = lane & 31, fh = lane >> 5, fx = (frow >> 1) & 7;
    int fko[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) fko[kc] = ((2 * kc + fh) ^ fx) * 4;
    const float* Afr = As + (wm * WROWS + frow) * LROW;
    const float* Bfr = Bs + (wn * WCOLS + frow) * LROW;
    const int total = ntl * a.KT;
    int kt_in = 0, m_cur = m0;
    for (int g = 0; g < total; ++g) {
        const int buf = g & 1;
        const float* Ab = Afr + buf * BM * LROW;
        const float* Bb = Bfr + buf * BN * LROW;
        float4 fa[2][MT], fb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LROW + fko[0]);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LROW + fko[0]);
        issue(buf ^ 1);                    // next K-step streams into the other buffer while this one is consumed
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int cur = kc & 1;
            if (kc < 3) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[cur ^ 1][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LROW + fko[kc < 3 ? kc + 1 : 0]);
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[cur ^ 1][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LROW + fko[kc < 3 ? kc + 1 : 0]);
            }
#pragma unroll
            for (int el = 0; el < 4; ++el)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float av = el == 0 ? fa[cur][i].x : el == 1 ? fa[cur][i].y : el == 2 ? fa[cur][i].z : fa[cur][i].w;
                        const float bv = el == 0 ? fb[cur][j].x : el == 1 ? fb[cur][j].y : el == 2 ? fb[cur][j].z : fb[cur][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
        if (++kt_in == a.KT) {             // tile finished: dense rows of M, then restart the accumulators
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
        __syncthreads();                   // vmcnt(0) + barrier: the other buffer is complete and this one is free
    }
}

}  // namespace

// called from ssdhip_conv2d_nhwc (conv_igemm.hip) for plane GEMMs
int ssdhip_gemm_planes_dma(const float* v, const float* u, float* y, int M, int K, int Cout, int rows_per_plane, long long u_plane_stride,
                           int BM, int BN, int tpb, hipStream_t st) {
    PlaneArgs a;
    a.v = v; a.u = u; a.y = y; a.u_plane_stride = u_plane_stride;
    a.M = M; a.K = K; a.KT = K / BK; a.Cout = Cout; a.rows_per_plane = rows_per_plane;
    a.MB = M / BM; a.NB = cdiv(Cout, BN); a.tpb = tpb;
    dim3 grid(cdiv(a.MB, a.tpb) * a.NB), block(256);
    if (BM == 128 && BN == 128) hipLaunchKernelGGL((gemm_planes_dma_kernel<128, 128>), grid, block, 0, st, a);
    else if (BM == 128 && BN == 64) hipLaunchKernelGGL((gemm_planes_dma_kernel<128, 64>), grid, block, 0, st, a);
    else if (BM == 64 && BN == 128) hipLaunchKernelGGL((gemm_planes_dma_kernel<64, 128>), grid, block, 0, st, a);
    else return SSDHIP_ENOSUP;
    return SSDHIP_OK;
}
