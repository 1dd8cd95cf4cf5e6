// Error plumbing and library-level queries of libs2s_mi355x.
#include "common.h"

#include <stdarg.h>

namespace s2s {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace s2s

extern "C" const char* s2s_last_error(void) { return s2s::g_last_error.c_str(); }
extern "C" int32_t s2s_abi_version(void) { return S2S_ABI_VERSION; }
extern "C" int32_t s2s_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
