// Probes and error plumbing of the C ABI (include/gan_amd.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void gan_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int gan_amd_version(void) { return 25; }
extern "C" const char* gan_amd_arch(void) { return "gfx950"; }
extern "C" const char* gan_amd_last_error(void) { return g_err; }
