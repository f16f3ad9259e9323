// gsx_capi.hip — error plumbing shared by the C-ABI entry points of libgsx.so (include/gsx.h).
// The reference reports errors as c10::Error exceptions and never checks launches;
// here every entry point returns a gsx_status and leaves a thread-local message.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
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

// Test / A-B switches (GSX_RASTER_PATH, GSX_BWD, GSX_BIN_NB, ... — DESIGN.md §1).  A host that embeds libgsx.so inherits NONE of them unless
// it opts in: they are looked at only when GSX_TEST_SWITCHES=1 is in the environment (read once per process; tests/conftest.py and
// the tools set it).  Returns the variable's value or nullptr.
const char* test_switch(const char* name) {
    static const bool enabled = [] { const char* e = getenv("GSX_TEST_SWITCHES"); return e != nullptr && strcmp(e, "1") == 0; }();
    return enabled ? getenv(name) : nullptr;
}

}  // namespace gsx

extern "C" const char* gsx_last_error(void) { return gsx::g_err; }
extern "C" int gsx_abi_version(void) { return GSX_ABI_VERSION; }
extern "C" const char* gsx_test_switch(const char* name) { return gsx::test_switch(name); }   // (for the C++ shim: one gate for every switch)
