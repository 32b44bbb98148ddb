// Error plumbing of the C ABI (thread-local message; nothing throws across the boundary).
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

namespace {
thread_local char g_err[512] = "";
}

int cb_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

int cb_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cb_fail("%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

extern "C" const char* cb_last_error(void) { return g_err; }
extern "C" int cb_version(void) { return 7; }
