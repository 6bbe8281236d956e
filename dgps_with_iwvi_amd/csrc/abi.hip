// Error reporting and version of the C-ABI (include/iwvi_hip.h).
#include "iwvi_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

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

// ---- development route switches: set through iwvi_debug_set_option only (the library does not read the environment) -------------
static const char* const g_opt_names[] = {
    "IWVI_BW_FUSED", "IWVI_CHAIN_EXIT", "IWVI_FW_SLOW_TAIL", "IWVI_FW_MAX_NS", "IWVI_NATGRAD_UNFUSED", "IWVI_NG_ONE_WG", "IWVI_NG_STOP", "IWVI_DEBUG_STOP", "IWVI_PRE_STAMP_P",
    "IWVI_FW_NO_LEAN", "IWVI_PRE_SB_INLINE", "IWVI_BW_P5_F32"};
constexpr int N_OPTS = (int)(sizeof(g_opt_names) / sizeof(g_opt_names[0]));
static int g_opt_values[N_OPTS] = {0};
static int opt_index(const char* name) {
    for (int i = 0; i < N_OPTS; ++i) if (!strcmp(name, g_opt_names[i])) return i;
    return -1;
}
int dbg_opt(const char* name) { const int i = opt_index(name); return i < 0 ? 0 : g_opt_values[i]; }

}  // namespace iwvi

extern "C" int iwvi_debug_set_option(const char* name, int value) {
    const int i = name ? iwvi::opt_index(name) : -1;
    if (i < 0) { iwvi::set_error("iwvi_debug_set_option: unknown option %s", name ? name : "(null)"); return IWVI_ERR_ARG; }
    iwvi::g_opt_values[i] = value;
    return IWVI_OK;
}
extern "C" int iwvi_version(void) { return IWVI_ABI_VERSION; }
extern "C" const char* iwvi_last_error(void) { return iwvi::g_err; }
