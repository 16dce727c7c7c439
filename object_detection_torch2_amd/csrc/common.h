// Shared helpers for libssdhip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ssdhip.h"

#define WAVE 64

extern "C" void ssdhip_set_error(const char* msg);

#define SSDHIP_REQUIRE(cond, msg)                 \
    do {                                          \
        if (!(cond)) {                            \
            ssdhip_set_error(msg);                \
            return SSDHIP_EINVAL;                 \
        }                                         \
    } while (0)

#define SSDHIP_CHECK_LAUNCH()                                  \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) {                               \
            ssdhip_set_error(hipGetErrorString(e__));          \
            return SSDHIP_ELAUNCH;                             \
        }                                                      \
    } while (0)

// "Done once per device" flag for per-device driver state such as hipFuncSetAttribute (a kernel has one function object per
// device, so a process-wide flag would leave every device but the first without its attribute).
struct DeviceOnce {
    static constexpr int kDone = -1, kUnknown = 64;
    unsigned long long mask = 0;          // bit d: done on device ordinal d (benign race between threads: the action is idempotent)
    int pending() const {                 // kDone, or the ordinal to hand to mark() once the action succeeded
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return kUnknown;
        return ((__atomic_load_n(&mask, __ATOMIC_RELAXED) >> dev) & 1ull) ? kDone : dev;
    }
    void mark(int dev) { if (dev >= 0 && dev < 64) __atomic_fetch_or(&mask, 1ull << dev, __ATOMIC_RELAXED); }
};

// Compute units of the current device (cached per device ordinal): the persistent grids and the split-K targets are derived from
// it, so a partition with fewer CUs (CPX / DPX modes, other gfx9 parts) does not queue surplus persistent blocks behind full-length
// ones.  256 on an MI355X in SPX mode; 256 if the query fails.
extern "C" int ssdhip_cu_count(void);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// a * b * c * d * e < limit for non-negative factors, evaluated without overflowing int64 (the size checks of the launchers take extents
// straight from the caller: four 31-bit extents multiply to 2^124).  A negative factor fails the check.  Found by `make asan` (UBSan).
static inline bool elems_below(int64_t limit, int64_t a, int64_t b = 1, int64_t c = 1, int64_t d = 1, int64_t e = 1) {
    const int64_t f[5] = {a, b, c, d, e};
    int64_t v = 1;
    for (int i = 0; i < 5; ++i) {
        if (f[i] < 0) return false;
        if (f[i] == 0) return limit > 0;
        if (v > (limit - 1) / f[i]) return false;
        v *= f[i];
    }
    return v < limit;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
