// Probe (timing only, no product code): what would the GEMM + output-transform half of a "two-pass" Winograd F(4x4) convolution reach on
// the 150 / 75 px layers (conv_2_1, conv_2_2, conv_3_1) that run the one-kernel F(2x2) form today?  Pass A would be the existing input
// transform writing V in a K-step-blocked layout; pass B -- this probe's inner loop -- keeps ALL 36 planes of a 32-tile x 64-channel item in
// accumulators (v_mfma_f32_16x16x4_f32: output channel on the lane, tile on the register index, so A^T m A stays register-local) and never
// writes M.  The accumulators (36 planes x 4 registers per 16 x 16 tile) dictate the wave tile, and the wave tile dictates the LDS feed:
//   S  wave = 16 tiles x 16 channels x 36 planes (2 x 4 waves):           one ds_read_b32 of A and one of B per MFMA
//   P  wave = 16 tiles x 32 channels x 18 planes (2 plane halves x 2 x 2): one ds_read_b32 of A and one ds_read_b64 of B per two MFMAs
//      (the real kernel would exchange half an output transform between the plane halves through LDS at the end of an item)
// Operand images per K-step (BK = 4: A = 36 planes x 32 tiles x 16 B = 18 KiB, B = 36 planes x 64 channels x 16 B = 36 KiB; B double-buffered,
// A in a ring of NA = 2 .. 4 stages: 108 .. 144 KiB of LDS, one block of 8 waves per CU) arrive by LDS-DMA from contiguous, pre-packed global
// images; the A images stream from HBM (one region per block and K-step), the B images are shared by all blocks (L2-resident weights).
// Measured (profiles/r04_j_*): the inner loop alone 0.78-0.85 of the fp32 MFMA peak, with the operand stream 0.55-0.60 -- the item the
// accumulators allow (32 x 64) needs 54 KiB of operands per 2,304 MFMA cycles and CU, three times the plane GEMM's bytes per flop.
//   hipcc --offload-arch=gfx950 -O3 f4_fused_gemm_probe.hip -o f4_fused_gemm_probe && ./f4_fused_gemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#include <type_traits>
#include <utility>
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

constexpr int PL = 36, TILES = 32, CH = 64, BK = 4;
constexpr int A_BYTES = PL * TILES * BK * 4;      // 18,432
constexpr int B_BYTES = PL * CH * BK * 4;         // 36,864
constexpr int STAGE = A_BYTES + B_BYTES;          // 55,296
constexpr int PIECES = STAGE / 1024;              // 54 wave-instructions of 1 KiB per K-step

// MODE 0 = S, 1 = P;  ABL bit 0: no DMA in the loop, bit 1: no epilogue stores
template <int MODE, int ABL, int DEPTH, int NA = 2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const char* __restrict__ asrc, const char* __restrict__ bsrc,
                                                                                         float* __restrict__ out, int KT, int items) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int PW = MODE == 0 ? 36 : 18, NT = MODE == 0 ? 1 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    int ph, wm, wn;
    if (MODE == 0) { ph = 0; wm = wave >> 2; wn = wave & 3; }
    else { ph = wave >> 2; wm = (wave >> 1) & 1; wn = wave & 1; }
    const int li = lane & 15, kq = lane >> 4;
    // fragment addresses inside a stage
    // LDS: NA stages of the A image (streamed from HBM, requested NA - 1 K-steps ahead), then 2 stages of the B image (weights from L2, one ahead)
    const unsigned a_rd = lds0 + (unsigned)(((ph * PW * TILES + 16 * wm + li) * BK + kq) * 4);
    const unsigned b_rd = lds0 + (unsigned)(NA * A_BYTES) +
                          (MODE == 0 ? (unsigned)(((16 * wn + li) * BK + kq) * 4) : (unsigned)(((ph * PW * 2 + wn) * 16 * BK * 2 + (li * BK + kq) * 2) * 4));
    constexpr int A_PSTRIDE = TILES * BK * 4, B_PSTRIDE = CH * BK * 4;

    const char* ablk = asrc + (size_t)blockIdx.x * (size_t)items * KT * A_BYTES;
    const int total = items * KT;
    constexpr int APIECES = A_BYTES / 1024, BPIECES = B_BYTES / 1024;       // 18, 36
    auto issue_a = [&](int step) {            // A image of K-step `step` -> ring slot step % NA
        const int st = step < total ? step : total - 1;
        const char* as = ablk + (size_t)st * A_BYTES;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int piece = wave + 8 * k;
            if (piece < APIECES) __builtin_amdgcn_global_load_lds((gptr_t)(as + piece * 1024 + lane * 16), (lptr_t)(lds + (step % NA) * A_BYTES + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_b = [&](int step) {
        const char* bs = bsrc + (size_t)(step % KT) * B_BYTES;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int piece = wave + 8 * k;
            if (piece < BPIECES) __builtin_amdgcn_global_load_lds((gptr_t)(bs + piece * 1024 + lane * 16), (lptr_t)(lds + NA * A_BYTES + (step & 1) * B_BYTES + piece * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[PW][NT];
    issue_b(0);
    for (int k = 0; k < NA - 1; ++k) issue_a(k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int g = 0;
    for (int it = 0; it < items; ++it) {
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt, ++g) {
            const int buf = g & 1;
            const unsigned ab = a_rd + (unsigned)((g % NA) * A_BYTES), bb = b_rd + (unsigned)(buf * B_BYTES);
            // fragments are requested D planes ahead of their MFMAs (ring of D + 1 registers)
            constexpr int D = DEPTH, R = D + 1;
            float fa[R], fs[R];
            f32x2 fd[R];
#define F4_REQ(P)                                                                                                                     \
    do {                                                                                                                               \
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[(P) % R]) : "v"(ab), "n"((P) * A_PSTRIDE));                                \
        if constexpr (MODE == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fs[(P) % R]) : "v"(bb), "n"((P) * B_PSTRIDE));      \
        else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fd[(P) % R]) : "v"(bb), "n"((P) * B_PSTRIDE));                          \
    } while (0)
#pragma unroll
            for (int p = 0; p < D; ++p) F4_REQ(p);
            if (!(ABL & 1)) { issue_b(g + 1); issue_a(g + NA - 1); }      // the slot of step g - 1 is free: its readers passed the last barrier
#pragma unroll
            for (int p = 0; p < PW; ++p) {
                const int r = p % R;
                if (p + D < PW) F4_REQ(p + D);
                // reads issued after this plane's pair: 2 per newer plane
                if constexpr (MODE == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[r]), "+v"(fs[r]) : "n"((p + D < PW ? D : PW - 1 - p) * 2));
                else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[r]), "+v"(fd[r]) : "n"((p + D < PW ? D : PW - 1 - p) * 2));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == 0) {
                    acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r], fs[r], acc[p][0], 0, 0, 0);
                } else {
                    acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r], fd[r].x, acc[p][0], 0, 0, 0);
                    acc[p][NT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r], fd[r].y, acc[p][NT - 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef F4_REQ
            // B of step g + 1 and A of step g + 1 (requested NA - 2 steps ago) must have landed; this step's A request (the newest: 3 pieces on
            // waves 0 and 1, 2 on the others) may stay in flight when NA > 2
            if (NA == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (wave < 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // stand-in epilogue: 16 output pixels per tile and channel group from sums of planes (the real A^T m A is ~10 operations per output),
        // NHWC-like stores: lane = channel, 64-B runs per 16 lanes
        {
            float* o = out + ((size_t)(blockIdx.x * items + it) * TILES * 16) * CH;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int px = 0; px < 16; ++px) {
                    const int tile = 16 * wm + 4 * kq + e;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        float v = acc[px % PW][j][e] + acc[(px + 7) % PW][j][e] + acc[(px + 13) % PW][j][e] + acc[(px + 17) % PW][j][e];
                        if (MODE == 1 && ph == 1) v = -v;
                        if ((MODE == 0 || ph == (px >> 3)) && (!(ABL & 2) || v == 123.456f)) o[((size_t)tile * 16 + px) * CH + (MODE == 0 ? 16 * wn : 32 * wn + 16 * j) + li] = v;
                    }
                }
        }
    }
}

__global__ void fill(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int)(h & 0xffff) * (1.0f / 32768.0f) - 1.0f;
    }
}

template <int MODE, int ABL, int DEPTH, int NA = 2>
static void run(const char* name, const char* a, const char* b, float* out, int KT, int items, int grid) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, ABL, DEPTH, NA>), hipFuncAttributeMaxDynamicSharedMemorySize, NA * A_BYTES + 2 * B_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE, ABL, DEPTH, NA>), dim3(grid), dim3(512), NA * A_BYTES + 2 * B_BYTES, 0, a, b, out, KT, items);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<MODE, ABL, DEPTH, NA>), dim3(grid), dim3(512), NA * A_BYTES + 2 * B_BYTES, 0, a, b, out, KT, items);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = 2.0 * grid * (double)items * KT * PL * TILES * CH * BK;
    printf("%-44s K = %3d, %d items per block: %7.3f ms  %6.1f TF executed (%.3f of 157.3)\n", name, KT * BK, items, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = cus > 0 ? cus : 256;
    for (int KT : {16, 32}) {                 // K = 64 (conv_2_1), 128 (conv_2_2, conv_3_1)
        const int items = 256 / KT;           // ~ the same K-steps per block either way
        const size_t abytes = (size_t)grid * items * KT * A_BYTES, bbytes = (size_t)KT * B_BYTES, obytes = (size_t)grid * items * TILES * 16 * CH * 4;
        char *a, *b; float* out;
        hipMalloc(&a, abytes); hipMalloc(&b, bbytes); hipMalloc(&out, obytes);
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (float*)a, abytes / 4); hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, (float*)b, bbytes / 4);
        run<0, 0, 1>("S: 16x16 waves, 36 planes, 1 plane ahead", a, b, out, KT, items, grid);
        run<0, 0, 2>("S, 2 planes ahead", a, b, out, KT, items, grid);
        run<0, 0, 3>("S, 3 planes ahead", a, b, out, KT, items, grid);
        run<0, 0, 2, 3>("S, 2 ahead, A image requested 2 K-steps ahead", a, b, out, KT, items, grid);
        run<0, 0, 2, 4>("S, 2 ahead, A image requested 3 K-steps ahead", a, b, out, KT, items, grid);
        run<0, 0, 3, 4>("S, 3 ahead, A image requested 3 K-steps ahead", a, b, out, KT, items, grid);
        run<0, 1, 2>("S, 2 ahead, no DMA in the loop", a, b, out, KT, items, grid);
        run<0, 3, 2>("S, 2 ahead, no DMA, no epilogue stores", a, b, out, KT, items, grid);
        run<1, 0, 1>("P: 16x32 waves, 18 planes per half, 1 ahead", a, b, out, KT, items, grid);
        run<1, 0, 2>("P, 2 planes ahead", a, b, out, KT, items, grid);
        run<1, 0, 2, 4>("P, 2 ahead, A image requested 3 K-steps ahead", a, b, out, KT, items, grid);
        run<1, 1, 2>("P, 2 ahead, no DMA in the loop", a, b, out, KT, items, grid);
        run<1, 3, 2>("P, 2 ahead, no DMA, no epilogue stores", a, b, out, KT, items, grid);
        hipFree(a); hipFree(b); hipFree(out);
    }
    return 0;
}
