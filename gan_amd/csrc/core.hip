// Error reporting and version of libaig_hip.so.
#include "common.h"

namespace aig {

static thread_local char g_err[512] = "";

char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace aig

extern "C" int aig_abi_version(void) { return 1; }

extern "C" const char* aig_last_error(void) { return aig::err_buf(); }
