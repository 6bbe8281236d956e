// Error reporting and version of the C-ABI (include/iwvi_hip.h).
#include "iwvi_common.h"
#include <cstdarg>
#include <cstdio>

namespace iwvi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return IWVI_ERR_LAUNCH;
    }
    return IWVI_OK;
}

}  // namespace iwvi

extern "C" int iwvi_version(void) { return IWVI_ABI_VERSION; }
extern "C" const char* iwvi_last_error(void) { return iwvi::g_err; }
