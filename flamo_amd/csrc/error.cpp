// Error plumbing of the C ABI: thread-local last-error string, no exceptions across the boundary.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace fl {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// default: what tools/dbg/policy_sweep.py measured best on the replayed config-2 step; FLAMO_STREAM_POLICY overrides
static unsigned g_stream_policy = [] {
    const char* e = getenv("FLAMO_STREAM_POLICY");
    return e ? (unsigned)strtoul(e, nullptr, 0) : 0u;
}();
static thread_local int g_stream_site = 0;
unsigned stream_policy() { return g_stream_policy; }
int stream_site() { return g_stream_site; }

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return FL_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return FL_ERR_HIP;
}
}  // namespace fl

extern "C" {
int fl_version(void) { return FL_ABI_VERSION; }
const char* fl_last_error(void) { return fl::g_err; }
int fl_set_stream_policy(unsigned mask, int site) {
    if (mask != 0xFFFFFFFFu) fl::g_stream_policy = mask;
    if (site >= 0) fl::g_stream_site = site ? 1 : 0;
    return (int)fl::g_stream_policy;
}
}
