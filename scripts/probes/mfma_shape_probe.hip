// Probe: sustained fp32 MFMA rate of v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 at the same per-wave output tile
// (64x64), operands re-read from LDS each k-chunk like the conv kernel, random data, every CU busy, 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * 32; i += 256) lds[i] = src[(blockIdx.x * 8192 + i) & 0xFFFFF];
    __syncthreads();
    const float* A = lds + (wave >> 1) * 64 * 32;
    const float* B = lds + 128 * 32 + (wave & 1) * 64 * 32;
    float out = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int row = lane & 31, h = lane >> 5;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                float4 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const float4*>(A + (i * 32 + row) * 32 + (((2 * kc + h) ^ ((row >> 1) & 7)) * 4));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const float4*>(B + (j * 32 + row) * 32 + (((2 * kc + h) ^ ((row >> 1) & 7)) * 4));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) out += acc[i][j][e];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        const int row = lane & 15, q = lane >> 4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {      // 16 k per chunk: lane (row, q) reads k = 16kc + 4q .. +3
                float4 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const float4*>(A + (i * 16 + row) * 32 + (((4 * kc + q) ^ (((i * 16 + row) >> 1) & 7)) * 4));
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const float4*>(B + (j * 16 + row) * 32 + (((4 * kc + q) ^ (((j * 16 + row) >> 1) & 7)) * 4));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) out += acc[i][j][e];
    }
    dst[blockIdx.x * 256 + tid] = out;
}

int main() {
    const int blocks = 512, iters = 4000;
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *src, *dst;
    hipMalloc(&src, h.size() * 4); hipMalloc(&dst, blocks * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = (double)blocks * 4 * iters * 2.0 * 64 * 64 * 32;   // per wave per iter: 64x64x32 MACs
    for (int round = 0; round < 4; ++round) {
        for (int shape : {32, 16}) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(blocks), dim3(256), 0, 0, src, dst, iters);
            else hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, src, dst, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("round %d shape %2d: %.3f ms  %.1f TF/s\n", round, shape, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
