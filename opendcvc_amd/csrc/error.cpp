// error.cpp - ABI version and the per-thread error message of the C ABI (include/dcvc_amd.h).  Host-only C++: no HIP
// call, so it is also part of the sanitizer build of the host coder (make asan).
#include <cstdarg>
#include <cstdio>

#include "common.hpp"

namespace dcvc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace dcvc

extern "C" {

int dcvc_abi_version(void) { return DCVC_ABI_VERSION; }

const char* dcvc_last_error(void) { return dcvc::g_err; }

}  // extern "C"
