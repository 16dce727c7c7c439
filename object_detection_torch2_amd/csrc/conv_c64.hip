// 3x3 / stride 1 / pad 1 convolution with Cin = 64 on big maps (VGG16 conv_1_2: 64 -> 64 at 300x300, conv_2_1: 64 -> 128 at
// 150x150; model/vgg16.py:25,33).  K = 576 is only 18 K-steps of the general implicit-GEMM kernel, whose LDS staging of the A
// tile then costs as much as the MFMAs it feeds (103-108 TF there).  Here nothing is staged for A:
//   * a wave owns 32 consecutive output pixels x 64 output channels per iteration and walks the image persistently;
//   * A operand of v_mfma_f32_32x32x2_f32 straight from global memory: lane (pixel = lane & 31, half = lane >> 5) loads the 32
//     input channels [32 half, 32 half + 32) of one tap as 8 float4 (one full 128-B line), applies the producer's BN.ReLU in
//     registers, zeroes out-of-image taps, and MFMA s of that tap takes channel s of them -- lanes 0-31 feed k = (tap, s),
//     lanes 32-63 k = (tap, 32 + s); the next tap's loads are in flight under this tap's 64 MFMAs (register double buffer);
//   * B operand from LDS: the 64 x 576 weights of the block's output-channel half stay resident (144 KiB, one block of 16 waves
//     per CU), laid out [tap][j][s/4][lane][4] so a lane reads the 4 floats of 4 consecutive MFMAs with one conflict-free
//     ds_read_b128;
//   * no barriers after the weight load; epilogue as conv_rgb.hip (bias, NHWC store, per-wave BatchNorm partial sums).
#include "common.h"

namespace {

__host__ __device__ inline int64_t tiles_of(int N, int H, int W) { return ((int64_t)N * H * W + 31) / 32; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int C64_THREADS = 512, C64_WAVES = C64_THREADS / 64;     // 8 waves = 2 per SIMD; 16 would cap the kernel at 128 registers (it spills)
constexpr int W_IMAGE = 9 * 2 * 8 * 64 * 4;          // floats of one 64-channel weight image (147,456 B)

// OIHW (Cout, 64, 3, 3) -> [half][tap][j][s4][lane][e]: co = 64 half + 32 j + (lane & 31), ci = 32 (lane >> 5) + 4 s4 + e
__global__ void pack_c64_kernel(const float* __restrict__ w, float* __restrict__ img, int Cout) {
    const int halves = Cout / 64;
    const int64_t total = (int64_t)halves * W_IMAGE;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int r = (int)(i % W_IMAGE);
        const int half = (int)(i / W_IMAGE);
        const int e = r & 3; r >>= 2;
        const int lane = r & 63; r >>= 6;
        const int s4 = r & 7; r >>= 3;
        const int j = r & 1; r >>= 1;
        const int tap = r;
        const int co = 64 * half + 32 * j + (lane & 31), ci = 32 * (lane >> 5) + 4 * s4 + e;
        img[i] = w[((size_t)co * 64 + ci) * 9 + tap];
    }
}

template <bool XFORM>
__global__ __launch_bounds__(C64_THREADS) void conv_c64_kernel(const float* __restrict__ x, const float* __restrict__ wimg,
                                                               const float* __restrict__ bias, const float* __restrict__ in_scale,
                                                               const float* __restrict__ in_shift, float* __restrict__ z,
                                                               float* __restrict__ stats, int N, int H, int W, int Cout, int groups) {
    extern __shared__ __attribute__((aligned(16))) float wl[];     // W_IMAGE floats + 128 (scale, shift)
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, prow = lane & 31;
    const int halves = Cout / 64;
    const int chalf = blockIdx.x % halves, group = blockIdx.x / halves;       // output-channel half, pixel group of this block
    {
        const float4* src = reinterpret_cast<const float4*>(wimg + (size_t)chalf * W_IMAGE);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = tid; i < W_IMAGE / 4; i += C64_THREADS) dst[i] = src[i];
        if (XFORM && tid < 64) { wl[W_IMAGE + tid] = in_scale[tid]; wl[W_IMAGE + 64 + tid] = in_shift[tid]; }
    }
    __syncthreads();
    const int64_t M = (int64_t)N * H * W;
    const int wave = group * C64_WAVES + (tid >> 6);
    // a block owns a contiguous range of tiles and its 8 waves walk it side by side: the rows above / below a tile are re-read
    // ~W/32 tiles later by the same CU, i.e. from its own L1 / its XCD's L2 (striding tiles over all waves of the grid made every
    // XCD fetch its own copy of the three input rows: 2.4x the algorithmic HBM bytes)
    const int64_t per_group = (tiles_of(N, H, W) + groups - 1) / groups;
    const int64_t t_lo = (int64_t)group * per_group, t_hi = min(t_lo + per_group, tiles_of(N, H, W));
    const int nwaves = C64_WAVES;
    const int64_t first = t_lo + (tid >> 6);
    const float4* bl = reinterpret_cast<const float4*>(wl) + lane;             // + ((tap * 2 + j) * 8 + s4) * 64
    const float4* scl = reinterpret_cast<const float4*>(wl + W_IMAGE + 32 * half);
    const float4* shl = reinterpret_cast<const float4*>(wl + W_IMAGE + 64 + 32 * half);
    float bv[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[j] = bias ? bias[64 * chalf + 32 * j + prow] : 0.f;

    // per-lane coordinates of a tile's pixel; tap loads of tile t+1 start under the last tap of tile t
    float4 a[2][8];
    bool ok[2];
    int n = 0, y = 0, xx = 0;
    bool pv = false;
    const float4* xrow = reinterpret_cast<const float4*>(x);
    auto locate = [&](int64_t tile) {
        const int64_t p = tile * 32 + prow;
        pv = p < M;
        const int64_t pc = pv ? p : M - 1;
        n = (int)(pc / ((int64_t)H * W));
        const int rem = (int)(pc - (int64_t)n * H * W);
        y = rem / W; xx = rem - y * W;
        xrow = reinterpret_cast<const float4*>(x) + (((size_t)n * H + y) * W + xx) * 16 + 8 * half;      // tap (1,1)
    };
    auto load_tap = [&](int tap, int buf) {
        const int r = tap / 3, s = tap - 3 * r;
        const int yy = y + r - 1, xs = xx + s - 1;
        ok[buf] = pv && (unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W;
        const float4* src = ok[buf] ? xrow + ((r - 1) * W + (s - 1)) * 16 : reinterpret_cast<const float4*>(x);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[buf][q] = src[q];
    };
    if (first < t_hi) { locate(first); load_tap(0, 0); }
    for (int64_t tile = first; tile < t_hi; tile += nwaves) {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        float4 b0 = bl[0], b1 = bl[8 * 64];                         // B fragments one group ahead of their MFMAs
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = (tap & 1);
            if (tap < 8) load_tap(tap + 1, cur ^ 1);
            else if (tile + nwaves < t_hi) { locate(tile + nwaves); load_tap(0, cur ^ 1); }   // 9 taps: tap 0 of the next tile lands in buffer 1 ...
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                float4 v = a[cur][s4];
                if (XFORM) {
                    const float4 sc = scl[s4], sh = shl[s4];
                    v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                    v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
                }
                if (!ok[cur]) v = make_float4(0.f, 0.f, 0.f, 0.f);          // zero padding AFTER the transform
                const float4 c0 = b0, c1 = b1;
                const int nxt = tap * 8 + s4 + 1;                            // next group (tap', s4'), wraps to the next tile's first
                const int nt = nxt >= 72 ? 0 : nxt >> 3, ns = nxt & 7;
                b0 = bl[((nt * 2 + 0) * 8 + ns) * 64]; b1 = bl[((nt * 2 + 1) * 8 + ns) * 64];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, c0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, c1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, c0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, c1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, c0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, c1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, c0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, c1.w, acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);     // keep the B reads of later groups where they are (the scheduler hoists them all: spills)
            }
        }
        // ... so an odd tile count per wave would leave it in the wrong buffer: tap 8 used buffer 0, the prefetch went to buffer 1; swap
#pragma unroll
        for (int q = 0; q < 8; ++q) a[0][q] = a[1][q];
        ok[0] = ok[1];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = 64 * chalf + 32 * j + prow;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m < M) {
                    const float v = acc[j][e] + bv[j];
                    z[m * Cout + co] = v;
                    s1[j] += v; s2[j] += v * v;
                }
            }
        }
    }
    if (stats != nullptr) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64), t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            if (half == 0) *reinterpret_cast<float2*>(stats + ((size_t)wave * Cout + 64 * chalf + 32 * j + prow) * 2) = make_float2(t1, t2);
        }
    }
}

int c64_groups(int Cout) { return 256 / (Cout / 64); }      // one resident block per CU: 256 blocks in all

}  // namespace

extern "C" {

int64_t ssdhip_conv_c64_weight_floats(int Cout) { return Cout > 0 && Cout % 64 == 0 ? (int64_t)(Cout / 64) * W_IMAGE : SSDHIP_EINVAL; }

int ssdhip_conv_c64_pack(const float* w_oihw, float* wimg, int Cout, void* stream) {
    SSDHIP_REQUIRE(w_oihw && wimg && Cout > 0 && Cout % 64 == 0, "conv_c64_pack: Cout must be a multiple of 64");
    hipLaunchKernelGGL(pack_c64_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w_oihw, wimg, Cout);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_conv_c64_stat_rows(int Cout) {
    if (Cout <= 0 || Cout % 64 || Cout > 256) return SSDHIP_EINVAL;
    return c64_groups(Cout) * C64_WAVES;
}

int ssdhip_conv_c64(const float* x, const float* wimg, const float* bias, const float* in_scale, const float* in_shift, float* z,
                    float* stats, int N, int H, int W, int Cout, void* stream) {
    SSDHIP_REQUIRE(x && wimg && z && N > 0 && H > 0 && W > 0, "conv_c64: bad arguments");
    SSDHIP_REQUIRE(Cout == 64 || Cout == 128 || Cout == 256, "conv_c64: Cout must be 64, 128 or 256");
    SSDHIP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv_c64: in_scale and in_shift go together");
    SSDHIP_REQUIRE(elems_below(1ll << 31, N, H, W, 64), "conv_c64: tensor too large");
    static DeviceOnce attr;
    const size_t lds = (size_t)(W_IMAGE + 128) * sizeof(float);
    if (const int dev = attr.pending(); dev != DeviceOnce::kDone) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_c64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_c64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            ssdhip_set_error("conv_c64: cannot enable 145 KiB dynamic LDS");
            return SSDHIP_ELAUNCH;
        }
        attr.mark(dev);
    }
    const int groups = c64_groups(Cout);
    dim3 grid(groups * (Cout / 64)), block(C64_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (in_scale) hipLaunchKernelGGL(conv_c64_kernel<true>, grid, block, lds, st, x, wimg, bias, in_scale, in_shift, z, stats, N, H, W, Cout, groups);
    else hipLaunchKernelGGL(conv_c64_kernel<false>, grid, block, lds, st, x, wimg, bias, in_scale, in_shift, z, stats, N, H, W, Cout, groups);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
