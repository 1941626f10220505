// C-ABI plumbing shared by every entry point of libgaot_hip.so: version + thread-local error text.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/gaot_hip.h"

namespace gaot {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gaot

extern "C" int gaot_abi_version(void) { return 10; }
extern "C" const char* gaot_last_error(void) { return gaot::g_err; }
