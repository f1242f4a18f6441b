// Error plumbing and small host-only entry points of libmde_hip.so.
#include <stdarg.h>

#include "mde_common.h"

static thread_local char g_err[512] = "";

void mde_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mde_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return MDE_OK;
    mde_set_error("%s: %s", what, hipGetErrorString(e));
    return MDE_EHIP;
}

extern "C" const char* mde_last_error(void) { return g_err; }

extern "C" int mde_abi_version(void) { return 5; }

extern "C" int mde_device_cu_count(int* out) {
    MDE_REQUIRE(out, "mde_device_cu_count: null argument");
    int dev = 0;
    int rc = mde_check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    hipDeviceProp_t p;
    rc = mde_check_hip(hipGetDeviceProperties(&p, dev), "hipGetDeviceProperties");
    if (rc) return rc;
    *out = p.multiProcessorCount;
    return MDE_OK;
}
