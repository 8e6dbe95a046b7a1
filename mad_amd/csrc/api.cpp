// Error reporting and version of libmad_hip.so.
#include "common.h"

#include <string.h>

namespace mad {

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
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

}  // namespace mad

extern "C" int mad_version(void) { return 1; }
extern "C" const char* mad_last_error(void) { return mad::g_err; }
