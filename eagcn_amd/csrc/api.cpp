// Library-level pieces of the C ABI: version, error string.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/eagcn_hip.h"

namespace eagcn {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace eagcn

extern "C" int eagcn_abi_version(void) { return 1; }
extern "C" const char* eagcn_last_error(void) { return eagcn::g_err; }
