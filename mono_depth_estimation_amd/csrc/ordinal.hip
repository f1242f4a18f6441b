// Kernels of the DORN path (reference network/Dorn.py, criteria.py:734-787): the ordinal regression head and its loss,
// nn.Dropout2d's per-(image, channel) scale, and the full-image encoder's padded average pool.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/mde_hip.h"
#include "mde_common.h"

namespace {

constexpr int NT = 256;
constexpr int OT_PIX = 64;

int grid_flat(int64_t total) {
    const int64_t g = (total + NT - 1) / NT;
    return (int)(g < 1 ? 1 : (g > 262144 ? 262144 : g));
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {   // result valid in thread 0
    const double r = mde_wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = r;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < NT / 64; ++i) t += sh[i];
    __syncthreads();
    return t;
}

// ------------------------------------------------------------------ Dropout2d scale (Dorn.py:59,107,109)
// out[n][p][c] = x[n][p][c] * m[n][c] (+ out): m holds 0 or 1 / (1 - p) per (image, channel); the same kernel routes the gradient.
__global__ __launch_bounds__(NT) void chan_scale_k(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ m,
                                                   bf16_t* __restrict__ out, int ldo, int64_t HW, int C, int64_t total, int acc) {
    const int cpr = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int col = (int)(i % cpr);
        const int64_t row = i / cpr;
        const int64_t n = row / HW;
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + row * ldx + col * 8);
        const float* mm = m + n * C + col * 8;
        bf16x8_t o;
        if (acc) o = *reinterpret_cast<const bf16x8_t*>(out + row * ldo + col * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)v[e] * mm[e] + (acc ? (float)o[e] : 0.f));
        *reinterpret_cast<bf16x8_t*>(out + row * ldo + col * 8) = o;
    }
}

// ------------------------------------------------------------------ padded average pool, NCHW-flattened output (Dorn.py:58,72-74)
// out[n][c * OH * OW + oy * OW + ox] = m[n][c] / k^2 * sum over the k x k window at (oy * s - p, ox * s - p) clipped to the map
// (count_include_pad: the divisor is always k^2).  The output order is the reference's `x.view(-1, C * h * w)` of an NCHW
// tensor, so nn.Linear's weight [out][C * h * w] contracts it as stored.
__global__ __launch_bounds__(NT) void avgpool_flat_fwd_k(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ m,
                                                         bf16_t* __restrict__ out, int N, int H, int W, int C, int OH, int OW, int k,
                                                         int s, int p) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * OH * OW * cpr;
    const float inv = 1.f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int col = (int)(i % cpr);
        int64_t q = i / cpr;
        const int ox = (int)(q % OW); q /= OW;
        const int oy = (int)(q % OH);
        const int n = (int)(q / OH);
        const int y0 = max(oy * s - p, 0), y1 = min(oy * s - p + k, H);
        const int x0 = max(ox * s - p, 0), x1 = min(ox * s - p + k, W);
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int y = y0; y < y1; ++y)
            for (int xx = x0; xx < x1; ++xx) {
                const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + (((int64_t)n * H + y) * W + xx) * ldx + col * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
            }
        const int64_t plane = (int64_t)OH * OW;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = col * 8 + e;
            out[((int64_t)n * C + c) * plane + oy * OW + ox] = (bf16_t)(a[e] * inv * (m ? m[(int64_t)n * C + c] : 1.f));
        }
    }
}

// dx[n][y][x][c] (+)= m[n][c] / k^2 * sum of dout over the windows that contain (y, x)   (gather form, no atomics)
__global__ __launch_bounds__(NT) void avgpool_flat_bwd_k(const bf16_t* __restrict__ dout, const float* __restrict__ m,
                                                         bf16_t* __restrict__ dx, int lddx, int N, int H, int W, int C, int OH,
                                                         int OW, int k, int s, int p, int acc) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * H * W * cpr;
    const float inv = 1.f / (float)(k * k);
    const int64_t plane = (int64_t)OH * OW;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int col = (int)(i % cpr);
        int64_t q = i / cpr;
        const int xx = (int)(q % W); q /= W;
        const int y = (int)(q % H);
        const int n = (int)(q / H);
        // windows oy with oy * s - p <= y <= oy * s - p + k - 1
        const int ay = y + p - k + 1, ax = xx + p - k + 1;
        const int oy0 = ay > 0 ? (ay + s - 1) / s : 0, ox0 = ax > 0 ? (ax + s - 1) / s : 0;
        const int oy1 = min((y + p) / s, OH - 1), ox1 = min((xx + p) / s, OW - 1);
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int oy = oy0; oy <= oy1; ++oy)
            for (int ox = ox0; ox <= ox1; ++ox)
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += (float)dout[((int64_t)n * C + col * 8 + e) * plane + oy * OW + ox];
        bf16_t* d = dx + (((int64_t)n * H + y) * W + xx) * lddx + col * 8;
        bf16x8_t o;
        if (acc) o = *reinterpret_cast<const bf16x8_t*>(d);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = (bf16_t)(a[e] * inv * (m ? m[(int64_t)n * C + col * 8 + e] : 1.f) + (acc ? (float)o[e] : 0.f));
        *reinterpret_cast<bf16x8_t*>(d) = o;
    }
}

// ------------------------------------------------------------------ ordinal regression head (Dorn.py:288-318)
// x: bf16 [N * HW][ldx], channel 2k = A_k, 2k + 1 = B_k.  prob[n][k][p] = softmax over the pair (clamp(A), clamp(B)) at index 1
// = 1 / (1 + exp(clamp(A) - clamp(B))), clamp to [1e-8, 1e4]; label[n][p] = #{k : prob > 0.5} (int64, as torch.sum of a bool
// tensor).  Pixel rows are read coalesced, transposed through LDS, planes written with 256-byte wave stores.
__device__ __forceinline__ float ord_clamp(float v) { return fminf(fmaxf(v, 1e-8f), 1e4f); }
__device__ __forceinline__ bool ord_inside(float v) { return v >= 1e-8f && v <= 1e4f; }

__global__ __launch_bounds__(NT) void ordinal_fwd_k(const bf16_t* __restrict__ x, int ldx, float* __restrict__ prob,
                                                    long long* __restrict__ label, int N, int64_t HW, int K) {
    extern __shared__ float sm[];                 // [2K][65] tile + [4][64] counts
    float* tile = sm;
    int* cnt = reinterpret_cast<int*>(sm + (size_t)2 * K * 65);
    const int C = 2 * K, cpr = (C + 7) >> 3;
    const int64_t tiles_per_img = (HW + OT_PIX - 1) / OT_PIX;
    const int64_t ntiles = (int64_t)N * tiles_per_img;
    const int p = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t n = t / tiles_per_img;
        const int64_t hw0 = (t - n * tiles_per_img) * OT_PIX;
        const int npix = (int)min((int64_t)OT_PIX, HW - hw0);
        for (int i = threadIdx.x; i < OT_PIX * cpr; i += NT) {
            const int pr = i / cpr, c8 = i - pr * cpr;
            bf16x8_t v;
            if (pr < npix) v = *reinterpret_cast<const bf16x8_t*>(x + (n * HW + hw0 + pr) * ldx + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c8 * 8 + e;
                if (c < C) tile[c * 65 + pr] = pr < npix ? (float)v[e] : 0.f;
            }
        }
        __syncthreads();
        int mine = 0;
        if (p < npix) {
            float* po = prob + n * K * HW + hw0 + p;
            for (int k = q; k < K; k += 4) {
                const float a = ord_clamp(tile[(2 * k) * 65 + p]), b = ord_clamp(tile[(2 * k + 1) * 65 + p]);
                const float mx = fmaxf(a, b);                       // softmax as ATen computes it: exp(v - max) / sum
                const float ea = expf(a - mx), eb = expf(b - mx);
                const float pr1 = eb / (ea + eb);
                po[(int64_t)k * HW] = pr1;
                mine += pr1 > 0.5f;
            }
        }
        cnt[q * 64 + p] = mine;
        __syncthreads();
        if (q == 0 && p < npix) label[n * HW + hw0 + p] = (long long)(cnt[p] + cnt[64 + p] + cnt[128 + p] + cnt[192 + p]);
        __syncthreads();
    }
}

// dA_k = -g, dB_k = +g with g = dprob * prob * (1 - prob), each only where its logit lies inside the clamp range
__global__ __launch_bounds__(NT) void ordinal_bwd_k(const float* __restrict__ dprob, const bf16_t* __restrict__ x, int ldx,
                                                    bf16_t* __restrict__ dx, int lddx, int N, int64_t HW, int K) {
    extern __shared__ float sm[];                 // [2K][65]
    float* tile = sm;
    const int C = 2 * K, cpr = (C + 7) >> 3, cpo = lddx >> 3;
    const int64_t tiles_per_img = (HW + OT_PIX - 1) / OT_PIX;
    const int64_t ntiles = (int64_t)N * tiles_per_img;
    const int p = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t n = t / tiles_per_img;
        const int64_t hw0 = (t - n * tiles_per_img) * OT_PIX;
        const int npix = (int)min((int64_t)OT_PIX, HW - hw0);
        for (int i = threadIdx.x; i < OT_PIX * cpr; i += NT) {
            const int pr = i / cpr, c8 = i - pr * cpr;
            bf16x8_t v;
            if (pr < npix) v = *reinterpret_cast<const bf16x8_t*>(x + (n * HW + hw0 + pr) * ldx + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c8 * 8 + e;
                if (c < C) tile[c * 65 + pr] = pr < npix ? (float)v[e] : 0.f;
            }
        }
        __syncthreads();
        if (p < npix) {
            const float* gp = dprob + n * K * HW + hw0 + p;
            for (int k = q; k < K; k += 4) {
                const float ar = tile[(2 * k) * 65 + p], br = tile[(2 * k + 1) * 65 + p];
                const float a = ord_clamp(ar), b = ord_clamp(br);
                const float mx = fmaxf(a, b);
                const float ea = expf(a - mx), eb = expf(b - mx);
                const float pr1 = eb / (ea + eb);
                const float g = gp[(int64_t)k * HW] * pr1 * (1.f - pr1);
                tile[(2 * k) * 65 + p] = ord_inside(ar) ? -g : 0.f;
                tile[(2 * k + 1) * 65 + p] = ord_inside(br) ? g : 0.f;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < OT_PIX * cpo; i += NT) {
            const int pr = i / cpo, c8 = i - pr * cpo;
            if (pr >= npix) continue;
            bf16x8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c8 * 8 + e;
                o[e] = (bf16_t)(c < C ? tile[c * 65 + pr] : 0.f);
            }
            *reinterpret_cast<bf16x8_t*>(dx + (n * HW + hw0 + pr) * lddx + c8 * 8) = o;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ ordLoss (criteria.py:734-787)
// loss = -( sum_{k <= t} log clamp(P_k, 1e-8, 1e8) + sum_{k > t} log clamp(1 - P_k, 1e-8, 1e8) ) / (N * HW), t = target[n][p]
// as a FLOAT (modules/dorn.py:102-107 hands over the un-truncated SID label; the reference compares the integer plane index
// with it after type promotion; a NaN target takes neither branch, -inf takes the second for every k).
struct OrdHead { double sum; };

__global__ void ord_loss_init_k(OrdHead* h) { h->sum = 0.0; }

__global__ __launch_bounds__(NT) void ord_loss_fwd_k(const float* __restrict__ prob, const float* __restrict__ target, int K,
                                                     int64_t HW, int64_t total, OrdHead* h) {
    __shared__ double sh[NT / 64];
    double part = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW, p = i - n * HW;
        const float t = target[i];
        const float* pp = prob + n * K * HW + p;
        float s = 0.f;
        for (int k = 0; k < K; ++k) {
            const float v = pp[(int64_t)k * HW];
            if ((float)k <= t) s += logf(fminf(fmaxf(v, 1e-8f), 1e8f));
            else if ((float)k > t) s += logf(fminf(fmaxf(1.f - v, 1e-8f), 1e8f));
        }
        part += (double)s;
    }
    const double ps = block_sum_d(part, sh);
    if (threadIdx.x == 0 && ps != 0.0) atomicAdd(&h->sum, ps);
}

__global__ void ord_loss_finalize_k(const OrdHead* h, double count, float* loss) { *loss = (float)(-h->sum / count); }

// d loss / d P_k = -gscale / (N HW) * ( [k <= t] / P  |  -[k > t] / (1 - P) ), zero where the clamp is active
__global__ __launch_bounds__(NT) void ord_loss_bwd_k(const float* __restrict__ prob, const float* __restrict__ target, int K,
                                                     int64_t HW, int64_t total, const float* __restrict__ gscale, float* __restrict__ grad) {
    const float gs = -gscale[0] / (float)total;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW, p = i - n * HW;
        const float t = target[i];
        const float* pp = prob + n * K * HW + p;
        float* gp = grad + n * K * HW + p;
        for (int k = 0; k < K; ++k) {
            const float v = pp[(int64_t)k * HW];
            float g = 0.f;
            if ((float)k <= t) {
                if (v >= 1e-8f && v <= 1e8f) g = gs / v;
            } else if ((float)k > t) {
                const float w = 1.f - v;
                if (w >= 1e-8f && w <= 1e8f) g = -gs / w;
            }
            gp[(int64_t)k * HW] = g;
        }
    }
}

}  // namespace

#define ORD_ALIGNED(p) (((uintptr_t)(p) % 16) == 0)

extern "C" int mde_chan_scale(const void* x, int ldx, const float* m, void* out, int ldo, int N, int64_t HW, int C, int accumulate,
                              void* stream) {
    MDE_REQUIRE(x && m && out && N > 0 && HW > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && ldx >= C && ldo >= C &&
                    ORD_ALIGNED(x) && ORD_ALIGNED(out),
                "mde_chan_scale: bad argument (C=%d, ldx=%d, ldo=%d: multiples of 8, 16-byte aligned bases)", C, ldx, ldo);
    const int64_t total = (int64_t)N * HW * (C / 8);
    chan_scale_k<<<grid_flat(total), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, m, (bf16_t*)out, ldo, HW, C, total, accumulate);
    MDE_LAUNCH_CHECK("chan_scale_k");
    return MDE_OK;
}

static int avgpool_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

extern "C" int mde_avgpool_flat_fwd(const void* x, int ldx, const float* m, void* out, int N, int H, int W, int C, int k, int s, int p,
                                    void* stream) {
    MDE_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldx >= C && ORD_ALIGNED(x),
                "mde_avgpool_flat_fwd: bad argument (C=%d, ldx=%d)", C, ldx);
    MDE_REQUIRE(k > 0 && s > 0 && p >= 0 && 2 * p <= k && H + 2 * p >= k && W + 2 * p >= k, "mde_avgpool_flat_fwd: k=%d s=%d p=%d on %dx%d", k, s, p, H, W);
    const int OH = avgpool_out(H, k, s, p), OW = avgpool_out(W, k, s, p);
    avgpool_flat_fwd_k<<<grid_flat((int64_t)N * OH * OW * (C / 8)), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, m, (bf16_t*)out, N, H, W,
                                                                                                   C, OH, OW, k, s, p);
    MDE_LAUNCH_CHECK("avgpool_flat_fwd_k");
    return MDE_OK;
}

extern "C" int mde_avgpool_flat_bwd(const void* dout, const float* m, void* dx, int lddx, int N, int H, int W, int C, int k, int s, int p,
                                    int accumulate, void* stream) {
    MDE_REQUIRE(dout && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && lddx % 8 == 0 && lddx >= C && ORD_ALIGNED(dx),
                "mde_avgpool_flat_bwd: bad argument (C=%d, lddx=%d)", C, lddx);
    MDE_REQUIRE(k > 0 && s > 0 && p >= 0 && 2 * p <= k && H + 2 * p >= k && W + 2 * p >= k, "mde_avgpool_flat_bwd: k=%d s=%d p=%d on %dx%d", k, s, p, H, W);
    const int OH = avgpool_out(H, k, s, p), OW = avgpool_out(W, k, s, p);
    avgpool_flat_bwd_k<<<grid_flat((int64_t)N * H * W * (C / 8)), NT, 0, (hipStream_t)stream>>>((const bf16_t*)dout, m, (bf16_t*)dx, lddx, N, H, W, C,
                                                                                                 OH, OW, k, s, p, accumulate);
    MDE_LAUNCH_CHECK("avgpool_flat_bwd_k");
    return MDE_OK;
}

static constexpr int ORD_MAX_K = 128;

extern "C" int mde_ordinal_fwd(const void* x, int ldx, float* prob, int64_t* label, int N, int64_t HW, int K, void* stream) {
    MDE_REQUIRE(x && prob && label && N > 0 && HW > 0 && K > 0 && K <= ORD_MAX_K && ldx % 8 == 0 && ldx >= (2 * K + 7) / 8 * 8 && ORD_ALIGNED(x),
                "mde_ordinal_fwd: bad argument (K=%d <= %d, ldx=%d >= 2K rounded up to 8)", K, ORD_MAX_K, ldx);
    const size_t smem = ((size_t)2 * K * 65 + 4 * 64) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        int rc = mde_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&ordinal_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (2 * ORD_MAX_K * 65 + 256) * 4),
                               "hipFuncSetAttribute(ordinal_fwd_k)");
        if (rc) return rc;
        attr = true;
    }
    const int64_t ntiles = (int64_t)N * ((HW + OT_PIX - 1) / OT_PIX);
    ordinal_fwd_k<<<(int)(ntiles < 4096 ? ntiles : 4096), NT, smem, (hipStream_t)stream>>>((const bf16_t*)x, ldx, prob, (long long*)label, N, HW, K);
    MDE_LAUNCH_CHECK("ordinal_fwd_k");
    return MDE_OK;
}

extern "C" int mde_ordinal_bwd(const float* dprob, const void* x, int ldx, void* dx, int lddx, int N, int64_t HW, int K, void* stream) {
    MDE_REQUIRE(dprob && x && dx && N > 0 && HW > 0 && K > 0 && K <= ORD_MAX_K && ldx % 8 == 0 && lddx % 8 == 0 && ldx >= (2 * K + 7) / 8 * 8 &&
                    lddx >= (2 * K + 7) / 8 * 8 && ORD_ALIGNED(x) && ORD_ALIGNED(dx),
                "mde_ordinal_bwd: bad argument (K=%d <= %d, ldx=%d, lddx=%d)", K, ORD_MAX_K, ldx, lddx);
    const size_t smem = ((size_t)2 * K * 65) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        int rc = mde_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&ordinal_bwd_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (2 * ORD_MAX_K * 65) * 4),
                               "hipFuncSetAttribute(ordinal_bwd_k)");
        if (rc) return rc;
        attr = true;
    }
    const int64_t ntiles = (int64_t)N * ((HW + OT_PIX - 1) / OT_PIX);
    ordinal_bwd_k<<<(int)(ntiles < 4096 ? ntiles : 4096), NT, smem, (hipStream_t)stream>>>(dprob, (const bf16_t*)x, ldx, (bf16_t*)dx, lddx, N, HW, K);
    MDE_LAUNCH_CHECK("ordinal_bwd_k");
    return MDE_OK;
}

extern "C" size_t mde_ord_loss_ws_bytes(void) { return sizeof(OrdHead); }

extern "C" int mde_ord_loss_fwd(const float* prob, const float* target, int N, int K, int64_t HW, void* ws, float* loss, void* stream) {
    MDE_REQUIRE(prob && target && ws && loss && N > 0 && K > 0 && HW > 0, "mde_ord_loss_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    OrdHead* h = (OrdHead*)ws;
    const int64_t total = (int64_t)N * HW;
    ord_loss_init_k<<<1, 1, 0, st>>>(h);
    MDE_LAUNCH_CHECK("ord_loss_init_k");
    int grid = grid_flat(total);
    if (grid > 2048) grid = 2048;
    ord_loss_fwd_k<<<grid, NT, 0, st>>>(prob, target, K, HW, total, h);
    MDE_LAUNCH_CHECK("ord_loss_fwd_k");
    ord_loss_finalize_k<<<1, 1, 0, st>>>(h, (double)total, loss);
    MDE_LAUNCH_CHECK("ord_loss_finalize_k");
    return MDE_OK;
}

extern "C" int mde_ord_loss_bwd(const float* prob, const float* target, int N, int K, int64_t HW, const float* gscale, float* grad,
                                void* stream) {
    MDE_REQUIRE(prob && target && gscale && grad && N > 0 && K > 0 && HW > 0, "mde_ord_loss_bwd: bad argument");
    const int64_t total = (int64_t)N * HW;
    ord_loss_bwd_k<<<grid_flat(total), NT, 0, (hipStream_t)stream>>>(prob, target, K, HW, total, gscale, grad);
    MDE_LAUNCH_CHECK("ord_loss_bwd_k");
    return MDE_OK;
}

// =================================================================================== MyNet pieces (network/MyNet.py)
namespace {

// Weighter's tail (MyNet.py:96-119): a [N][HW][lda] bf16 with C channels -> flatten(start_dim=2) -> nn.Linear(HW, 1) over the
// pixel axis -> sum over the C channels -> sigmoid:  scale[n] = sigmoid( sum_p w[p] * (sum_c a[n][p][c]) + C * b ).
// pre[n] accumulates the double sum (zeroed by the caller's init launch).
__global__ void zero_f32_k(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

__global__ __launch_bounds__(NT) void weighted_pool_fwd_k(const bf16_t* __restrict__ a, int lda, const float* __restrict__ w, int64_t HW, int C,
                                                          float* __restrict__ pre) {
    __shared__ double sh[NT / 64];
    const int n = blockIdx.y, cpr = C >> 3;
    const int64_t total = HW * cpr;
    double part = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t p = i / cpr;
        const int col = (int)(i - p * cpr);
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(a + ((int64_t)n * HW + p) * lda + col * 8);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[e];
        part += (double)(s * w[p]);
    }
    const double t = block_sum_d(part, sh);
    if (threadIdx.x == 0) atomicAdd(pre + n, (float)t);
}

__global__ void weighted_pool_finish_k(const float* __restrict__ pre, const float* __restrict__ b, int C, int N, float* __restrict__ scale) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) scale[n] = 1.f / (1.f + expf(-(pre[n] + (float)C * b[0])));
}

// dpre[n] = dscale[n] * s (1 - s);  da[n][p][c] (+)= dpre[n] * w[p];  dw[p] += sum_n dpre[n] * sum_c a[n][p][c];  db += C * sum_n dpre[n]
__global__ __launch_bounds__(NT) void weighted_pool_bwd_k(const float* __restrict__ dscale, const float* __restrict__ scale,
                                                          const bf16_t* __restrict__ a, int lda, const float* __restrict__ w,
                                                          bf16_t* __restrict__ da, int ldda, int acc, float* __restrict__ dw,
                                                          float* __restrict__ db, int N, int64_t HW, int C, MdeDetDev det) {
    const int cpr = C >> 3;
    for (int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x; p < HW; p += (int64_t)gridDim.x * NT) {
        const float wp = w[p];
        float gw = 0.f;
        for (int n = 0; n < N; ++n) {
            const float s = scale[n], dp = dscale[n] * s * (1.f - s);
            float rs = 0.f;
            for (int col = 0; col < cpr; ++col) {
                const int64_t off = ((int64_t)n * HW + p);
                const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(a + off * lda + col * 8);
                bf16x8_t o;
                if (acc) o = *reinterpret_cast<const bf16x8_t*>(da + off * ldda + col * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    rs += (float)v[e];
                    o[e] = (bf16_t)(dp * wp + (acc ? (float)o[e] : 0.f));
                }
                *reinterpret_cast<bf16x8_t*>(da + off * ldda + col * 8) = o;
            }
            gw += dp * rs;
        }
        mde_grad_add(dw + p, gw, det);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float g = 0.f;
        for (int n = 0; n < N; ++n) g += dscale[n] * scale[n] * (1.f - scale[n]);
        mde_grad_add(db, (float)C * g, det);
    }
}

// depth = factor * (m0 * s0[n] + m1 * s1[n] + m2 * s2[n])  (MyNet.py:152-155: / 3.0, * 10.0), fp32 maps [N][HW]
__global__ __launch_bounds__(NT) void combine3_fwd_k(const float* __restrict__ m0, const float* __restrict__ m1, const float* __restrict__ m2,
                                                     const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
                                                     float factor, int64_t HW, int64_t total, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW;
        out[i] = factor * (m0[i] * s0[n] + m1[i] * s1[n] + m2[i] * s2[n]);
    }
}

// dm_k += factor * dout * s_k[n];  ds_k[n] += factor * sum_p dout * m_k   (ds zeroed by the caller's init launch)
__global__ __launch_bounds__(NT) void combine3_bwd_k(const float* __restrict__ dout, const float* __restrict__ m0, const float* __restrict__ m1,
                                                     const float* __restrict__ m2, const float* __restrict__ s0, const float* __restrict__ s1,
                                                     const float* __restrict__ s2, float factor, int64_t HW, float* __restrict__ dm0,
                                                     float* __restrict__ dm1, float* __restrict__ dm2, float* __restrict__ ds) {
    __shared__ double sh[NT / 64];
    const int n = blockIdx.y;
    const float a0 = factor * s0[n], a1 = factor * s1[n], a2 = factor * s2[n];
    double p0 = 0.0, p1 = 0.0, p2 = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x; p < HW; p += (int64_t)gridDim.x * NT) {
        const int64_t i = (int64_t)n * HW + p;
        const float g = dout[i];
        p0 += (double)(g * m0[i]);
        p1 += (double)(g * m1[i]);
        p2 += (double)(g * m2[i]);
        dm0[i] += g * a0;
        dm1[i] += g * a1;
        dm2[i] += g * a2;
    }
    const double t0 = block_sum_d(p0, sh), t1 = block_sum_d(p1, sh), t2 = block_sum_d(p2, sh);
    if (threadIdx.x == 0) {
        atomicAdd(ds + n, factor * (float)t0);
        atomicAdd(ds + gridDim.y + n, factor * (float)t1);
        atomicAdd(ds + 2 * gridDim.y + n, factor * (float)t2);
    }
}

}  // namespace

extern "C" int mde_weighted_pool_fwd(const void* a, int lda, const float* w, const float* b, float* pre, float* scale, int N, int64_t HW,
                                     int C, void* stream) {
    MDE_REQUIRE(a && w && b && pre && scale && N > 0 && HW > 0 && C > 0 && C % 8 == 0 && lda % 8 == 0 && lda >= C && ORD_ALIGNED(a),
                "mde_weighted_pool_fwd: bad argument (C=%d, lda=%d)", C, lda);
    hipStream_t st = (hipStream_t)stream;
    zero_f32_k<<<mde_cdiv(N, 64), 64, 0, st>>>(pre, N);
    MDE_LAUNCH_CHECK("zero_f32_k");
    int gx = grid_flat(HW * (C / 8));
    if (gx > 64) gx = 64;
    weighted_pool_fwd_k<<<dim3(gx, N), NT, 0, st>>>((const bf16_t*)a, lda, w, HW, C, pre);
    MDE_LAUNCH_CHECK("weighted_pool_fwd_k");
    weighted_pool_finish_k<<<mde_cdiv(N, 64), 64, 0, st>>>(pre, b, C, N, scale);
    MDE_LAUNCH_CHECK("weighted_pool_finish_k");
    return MDE_OK;
}

extern "C" int mde_weighted_pool_bwd(const float* dscale, const float* scale, const void* a, int lda, const float* w, void* da, int ldda,
                                     int accumulate, float* dw, float* db, int N, int64_t HW, int C, void* stream) {
    MDE_REQUIRE(dscale && scale && a && w && da && dw && db && N > 0 && HW > 0 && C > 0 && C % 8 == 0 && lda % 8 == 0 && ldda % 8 == 0 &&
                    lda >= C && ldda >= C && ORD_ALIGNED(a) && ORD_ALIGNED(da),
                "mde_weighted_pool_bwd: bad argument (C=%d, lda=%d, ldda=%d)", C, lda, ldda);
    MDE_DET_REQUIRE("mde_weighted_pool_bwd", dw, HW);
    MDE_DET_REQUIRE("mde_weighted_pool_bwd", db, (int64_t)1);
    weighted_pool_bwd_k<<<grid_flat(HW), NT, 0, (hipStream_t)stream>>>(dscale, scale, (const bf16_t*)a, lda, w, (bf16_t*)da, ldda, accumulate, dw, db,
                                                                      N, HW, C, mde_det_dev());
    MDE_LAUNCH_CHECK("weighted_pool_bwd_k");
    return MDE_OK;
}

extern "C" int mde_combine3_fwd(const float* m0, const float* m1, const float* m2, const float* s0, const float* s1, const float* s2,
                                float factor, int N, int64_t HW, float* out, void* stream) {
    MDE_REQUIRE(m0 && m1 && m2 && s0 && s1 && s2 && out && N > 0 && HW > 0, "mde_combine3_fwd: bad argument");
    const int64_t total = (int64_t)N * HW;
    combine3_fwd_k<<<grid_flat(total), NT, 0, (hipStream_t)stream>>>(m0, m1, m2, s0, s1, s2, factor, HW, total, out);
    MDE_LAUNCH_CHECK("combine3_fwd_k");
    return MDE_OK;
}

extern "C" int mde_combine3_bwd(const float* dout, const float* m0, const float* m1, const float* m2, const float* s0, const float* s1,
                                const float* s2, float factor, int N, int64_t HW, float* dm0, float* dm1, float* dm2, float* ds, void* stream) {
    MDE_REQUIRE(dout && m0 && m1 && m2 && s0 && s1 && s2 && dm0 && dm1 && dm2 && ds && N > 0 && HW > 0, "mde_combine3_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    zero_f32_k<<<mde_cdiv(3 * N, 64), 64, 0, st>>>(ds, 3 * N);
    MDE_LAUNCH_CHECK("zero_f32_k");
    int gx = grid_flat(HW);
    if (gx > 128) gx = 128;
    combine3_bwd_k<<<dim3(gx, N), NT, 0, st>>>(dout, m0, m1, m2, s0, s1, s2, factor, HW, dm0, dm1, dm2, ds);
    MDE_LAUNCH_CHECK("combine3_bwd_k");
    return MDE_OK;
}
