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

extern "C" int mde_abi_version(void) { return 12; }
extern "C" int mde_act_dtype(void) { return MDE_ACT_DTYPE_CODE; }       // 0: bf16 storage (libmde_hip.so), 1: fp16 (libmde_hip_f16.so)

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

// ------------------------------------------------------------------ deterministic mode
MdeDet g_mde_det = {0, nullptr, nullptr, 0};

extern "C" size_t mde_det_scratch_bytes(int64_t n) { return (size_t)n * 16; }

extern "C" int mde_set_deterministic(int on, float* gbase, void* scratch, int64_t n) {
    MDE_REQUIRE(!on || (gbase && scratch && n > 0 && ((uintptr_t)scratch % 16) == 0), "mde_set_deterministic: bad argument");
    g_mde_det.on = on != 0;
    g_mde_det.gbase = on ? gbase : nullptr;
    g_mde_det.scratch = on ? reinterpret_cast<long long*>(scratch) : nullptr;
    g_mde_det.n = on ? n : 0;
    return MDE_OK;
}

extern "C" int mde_deterministic(void) { return g_mde_det.on; }

namespace {
__global__ __launch_bounds__(256) void det_flush_k(float* g, long long* scratch, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        long long* p = scratch + 2 * i;
        if (p[0] | p[1]) {
            g[i] += (float)mde_det_value(p);
            p[0] = 0;
            p[1] = 0;
        }
    }
}
}  // namespace

extern "C" int mde_det_flush(void* stream) {
    if (!g_mde_det.on) return MDE_OK;
    const int64_t n = g_mde_det.n;
    det_flush_k<<<(int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256), 256, 0, (hipStream_t)stream>>>(g_mde_det.gbase, g_mde_det.scratch, n);
    MDE_LAUNCH_CHECK("det_flush_k");
    return MDE_OK;
}
