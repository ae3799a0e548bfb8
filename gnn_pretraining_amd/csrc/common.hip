// libgnnmp: version + error plumbing.
#include <stdarg.h>

#include "gnnmp_internal.h"

namespace gmp {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
static thread_local int g_lane_mode = LANE_ALL;
static thread_local hipStream_t g_lane_stream = nullptr;
int& lane_mode() { return g_lane_mode; }
hipStream_t& lane_stream() { return g_lane_stream; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace gmp

extern "C" int gmp_version(void) { return 100; }
extern "C" const char* gmp_last_error_string(void) { return gmp::err_buf(); }
