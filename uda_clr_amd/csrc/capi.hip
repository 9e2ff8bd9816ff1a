// Error reporting and version for libuda_clr_hip.so.
#include "common.h"

static thread_local char g_err[512] = "";

int uda_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

extern "C" const char* uda_last_error(void) { return g_err; }
extern "C" int uda_version(void) { return 1; }
