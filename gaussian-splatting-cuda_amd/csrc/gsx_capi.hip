// gsx_capi.hip — error plumbing shared by the C-ABI entry points of libgsx.so (include/gsx.h).
// The reference reports errors as c10::Error exceptions and never checks launches;
// here every entry point returns a gsx_status and leaves a thread-local message.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gsx.h"

namespace gsx {

static thread_local char g_err[512] = "";

void set_error(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return GSX_OK;
    snprintf(g_err, sizeof(g_err), "%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return GSX_ERR_LAUNCH_FAILED;
}

}  // namespace gsx

extern "C" const char* gsx_last_error(void) { return gsx::g_err; }
extern "C" int gsx_abi_version(void) { return GSX_ABI_VERSION; }
