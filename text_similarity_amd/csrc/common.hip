// Error reporting and version for libtsim.so.
#include <stdarg.h>

#include "common.h"

namespace tsim {
static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
}  // namespace tsim

extern "C" int tsim_version(void) { return 102; }
extern "C" const char *tsim_last_error(void) { return tsim::g_last_error.c_str(); }
