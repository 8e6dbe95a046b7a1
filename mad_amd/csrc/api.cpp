// Error reporting, version and build identity of libmad_hip.so.
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

#ifndef MAD_BUILD_ID
#error "MAD_BUILD_ID is set by the Makefile (hash of the native sources, mad_amd/buildid.py)"
#endif

extern "C" int mad_version(void) { return 1; }
extern "C" const char* mad_build_id(void) { return MAD_BUILD_ID; }
extern "C" const char* mad_last_error(void) { return mad::g_err; }
