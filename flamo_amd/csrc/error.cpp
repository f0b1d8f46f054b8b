// Error plumbing of the C ABI: thread-local last-error string, no exceptions across the boundary.
#include <stdarg.h>
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

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return FL_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return FL_ERR_HIP;
}
}  // namespace fl

extern "C" {
int fl_version(void) { return FL_ABI_VERSION; }
const char* fl_last_error(void) { return fl::g_err; }
}
