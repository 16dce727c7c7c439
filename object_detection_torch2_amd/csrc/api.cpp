// Library-level entry points of libssdhip.so (no reference equivalent).
#include <string.h>
#include "../../include/ssdhip.h"

static thread_local char g_err[256] = "";

extern "C" {

void ssdhip_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

const char* ssdhip_last_error(void) { return g_err; }

int ssdhip_version(void) { return 100; }

}  // extern "C"
