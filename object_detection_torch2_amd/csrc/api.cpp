// Library-level entry points of libssdhip.so (no reference equivalent).
#include <string.h>
#include <hip/hip_runtime.h>
#include "../../include/ssdhip.h"

static thread_local char g_err[256] = "";

extern "C" {

void ssdhip_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

const char* ssdhip_last_error(void) { return g_err; }

int ssdhip_version(void) { return 101; }

int ssdhip_cu_count(void) {
    static int cached[64] = {0};          // per device ordinal; benign race: every thread computes the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
    return n;
}

}  // extern "C"
