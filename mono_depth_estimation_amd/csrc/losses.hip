// Per-pixel depth losses and metrics as wavefront-reduced streaming kernels (gfx950).
//   SILog           reference criteria.py:724-732
//   depth metrics   reference metrics.py:58-109 (absrel, 'rmse' (sic), delta1-3, log10)
// fp32 in, per-thread fp32 partials over a short strided run, wave/workgroup reduction in
// double, one fp64 atomic per workgroup and quantity.  HBM-bound: 2 x 4 B read per pixel.
#include "mde_common.h"

namespace {

constexpr int NT = 256;

struct SilogWs {
    double s1, s2, cnt;   // sum d, sum d^2, number of valid pixels
    float loss, pad;
};

template <int K>
__device__ __forceinline__ void block_atomic_add(double (&v)[K], double* dst) {
    __shared__ double sh[K][NT / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double r = mde_wave_sum_d(v[k]);
        if (lane == 0) sh[k][w] = r;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double r = 0.0;
        for (int i = 0; i < NT / 64; ++i) r += sh[threadIdx.x][i];
        atomicAdd(dst + threadIdx.x, r);
    }
}

__global__ __launch_bounds__(NT) void silog_reduce_k(const float* __restrict__ est, const float* __restrict__ gt,
                                                     int64_t n, SilogWs* ws) {
    float s1 = 0.f, s2 = 0.f, c = 0.f;
    double acc[3] = {0.0, 0.0, 0.0};
    int run = 0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float g = gt[i];
        if (g > 1e-2f) {
            const float d = logf(est[i]) - logf(g);
            s1 += d;
            s2 += d * d;
            c += 1.f;
        }
        if (++run == 64) {   // spill the fp32 partials before they lose bits
            acc[0] += s1; acc[1] += s2; acc[2] += c;
            s1 = s2 = c = 0.f;
            run = 0;
        }
    }
    acc[0] += s1; acc[1] += s2; acc[2] += c;
    block_atomic_add<3>(acc, &ws->s1);
}

__global__ void silog_finalize_k(SilogWs* ws, float lambda, float* loss) {
    const double m1 = ws->s1 / ws->cnt, m2 = ws->s2 / ws->cnt;
    const float l = (float)(10.0 * sqrt(m2 - (double)lambda * m1 * m1));
    ws->loss = l;
    *loss = l;
}

// d loss / d est_i = (100 / (n * loss)) * (d_i - lambda * mean d) / est_i   for valid pixels
__global__ __launch_bounds__(NT) void silog_bwd_k(const float* __restrict__ est, const float* __restrict__ gt, int64_t n,
                                                  float lambda, const SilogWs* __restrict__ ws,
                                                  const float* __restrict__ gscale, float* __restrict__ grad) {
    const float cnt = (float)ws->cnt;
    const float mean_d = (float)(ws->s1 / ws->cnt);
    const float k = (gscale ? *gscale : 1.f) * 100.f / (cnt * ws->loss);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float g = gt[i], e = est[i];
        float r = 0.f;
        if (g > 1e-2f) r = k * ((logf(e) - logf(g)) - lambda * mean_d) / e;
        grad[i] = r;
    }
}

struct MetricWs { double s[7]; };  // absrel, rmse(sic), d1, d2, d3, log10, count

__global__ __launch_bounds__(NT) void metrics_reduce_k(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       int64_t n, MetricWs* ws) {
    float a[7] = {0, 0, 0, 0, 0, 0, 0};
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    int run = 0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float t = tgt[i];
        if (t > 0.f) {
            const float p = fmaxf(pred[i], 1e-7f);
            const float diff = p - t;
            const float ratio = fmaxf(p / t, t / p);
            a[0] += fabsf(diff) / t;
            a[1] += sqrtf(diff * diff / t);
            a[2] += ratio < 1.25f ? 1.f : 0.f;
            a[3] += ratio < 1.25f * 1.25f ? 1.f : 0.f;
            a[4] += ratio < 1.25f * 1.25f * 1.25f ? 1.f : 0.f;
            a[5] += fabsf(log10f(p) - log10f(t));
            a[6] += 1.f;
        }
        if (++run == 64) {
#pragma unroll
            for (int k = 0; k < 7; ++k) { acc[k] += a[k]; a[k] = 0.f; }
            run = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] += a[k];
    block_atomic_add<7>(acc, ws->s);
}

__global__ void metrics_finalize_k(const MetricWs* ws, float* out) {
    const int k = threadIdx.x;
    if (k < 6) out[k] = (float)(ws->s[k] / ws->s[6]);
}

int grid_for(int64_t n) {
    int64_t nb = (n + NT - 1) / NT;
    return (int)(nb > 256 * 8 ? 256 * 8 : (nb < 1 ? 1 : nb));
}

}  // namespace

extern "C" size_t mde_silog_ws_bytes(void) { return sizeof(SilogWs); }

extern "C" int mde_silog_fwd(const float* est, const float* gt, int64_t n, float variance_focus, void* ws,
                             float* loss, void* stream) {
    MDE_REQUIRE(est && gt && ws && loss && n > 0, "mde_silog_fwd: bad argument");
    MDE_REQUIRE(((uintptr_t)ws % 8) == 0, "mde_silog_fwd: ws must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = mde_check_hip(hipMemsetAsync(ws, 0, sizeof(SilogWs), st), "hipMemsetAsync(silog ws)")) return rc;
    silog_reduce_k<<<grid_for(n), NT, 0, st>>>(est, gt, n, (SilogWs*)ws);
    MDE_LAUNCH_CHECK("silog_reduce_k");
    silog_finalize_k<<<1, 1, 0, st>>>((SilogWs*)ws, variance_focus, loss);
    MDE_LAUNCH_CHECK("silog_finalize_k");
    return MDE_OK;
}

extern "C" int mde_silog_bwd(const float* est, const float* gt, int64_t n, float variance_focus, const void* ws,
                             const float* gscale, float* grad, void* stream) {
    MDE_REQUIRE(est && gt && ws && grad && n > 0, "mde_silog_bwd: bad argument");
    silog_bwd_k<<<grid_for(n), NT, 0, (hipStream_t)stream>>>(est, gt, n, variance_focus, (const SilogWs*)ws, gscale, grad);
    MDE_LAUNCH_CHECK("silog_bwd_k");
    return MDE_OK;
}

extern "C" size_t mde_metrics_ws_bytes(void) { return sizeof(MetricWs); }

extern "C" int mde_depth_metrics(const float* pred, const float* target, int64_t n, void* ws, float* out, void* stream) {
    MDE_REQUIRE(pred && target && ws && out && n > 0, "mde_depth_metrics: bad argument");
    MDE_REQUIRE(((uintptr_t)ws % 8) == 0, "mde_depth_metrics: ws must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = mde_check_hip(hipMemsetAsync(ws, 0, sizeof(MetricWs), st), "hipMemsetAsync(metrics ws)")) return rc;
    metrics_reduce_k<<<grid_for(n), NT, 0, st>>>(pred, target, n, (MetricWs*)ws);
    MDE_LAUNCH_CHECK("metrics_reduce_k");
    metrics_finalize_k<<<1, 64, 0, st>>>((const MetricWs*)ws, out);
    MDE_LAUNCH_CHECK("metrics_finalize_k");
    return MDE_OK;
}
