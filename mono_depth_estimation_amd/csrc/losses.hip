// Per-pixel depth losses and metrics as wavefront-reduced streaming kernels (gfx950).
//   SILog           reference criteria.py:724-732
//   depth metrics   reference metrics.py:58-109 (absrel, 'rmse' (sic), delta1-3, log10)
//   MaskedL1Loss / MaskedMSELoss / berHuLoss   reference criteria.py:67-90,113-133
//   MaskedDepthLoss (Eigen)                    reference criteria.py:17-64
//   compute_scale_and_shift, mse / l1 / trimmed data terms, GradientLoss, MidasLoss   reference criteria.py:154-332
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

struct MetricWs { double s[11]; };  // absrel, rmse(sic), d1, d2, d3, log10, mae, mse, msle, sqrel, count

__global__ __launch_bounds__(NT) void metrics_reduce_k(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       int64_t n, MetricWs* ws) {
    constexpr int K = 11;
    float a[K];
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { a[k] = 0.f; acc[k] = 0.0; }
    int run = 0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float t = tgt[i];
        if (t > 0.f) {
            const float p = fmaxf(pred[i], 1e-7f);
            const float diff = p - t;
            const float ratio = fmaxf(p / t, t / p);
            const float dl = log1pf(p) - log1pf(t);
            a[0] += fabsf(diff) / t;
            a[1] += sqrtf(diff * diff / t);
            a[2] += ratio < 1.25f ? 1.f : 0.f;
            a[3] += ratio < 1.25f * 1.25f ? 1.f : 0.f;
            a[4] += ratio < 1.25f * 1.25f * 1.25f ? 1.f : 0.f;
            a[5] += fabsf(log10f(p) - log10f(t));
            a[6] += fabsf(diff);              // torchmetrics mean_absolute_error (metrics.py:118)
            a[7] += diff * diff;              // mean_squared_error (:121)
            a[8] += dl * dl;                  // mean_squared_log_error (:120): (log1p p - log1p t)^2
            a[9] += diff * diff / t;          // RelativeSquareError (:100-103)
            a[10] += 1.f;
        }
        if (++run == 64) {
#pragma unroll
            for (int k = 0; k < K; ++k) { acc[k] += a[k]; a[k] = 0.f; }
            run = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] += a[k];
    block_atomic_add<K>(acc, ws->s);
}

__global__ void metrics_finalize_k(const MetricWs* ws, float* out) {
    const int k = threadIdx.x;
    if (k < 10) out[k] = (float)(ws->s[k] / ws->s[10]);
}

// ------------------------------------------------------------------ masked pointwise losses (criteria.py:67-133)
// kind 0: mean |t-p| over t > 0; kind 1: mean (t-p)^2; kind 2 (reverse Huber as the reference writes it):
// c = 0.2 * max(p - t) over ALL pixels, loss = mean(cat(|d|, |d|[|d| > c]^2)) over valid pixels.
struct MaskedWs {
    double s1, s2, n1, n2;   // sum of the first-order terms / of the squared berHu terms, their counts
    int cmax_key;            // max(p - t) as an order-preserving integer key
    float c, loss, inv;      // berHu threshold; result; 1 / (n1 + n2)
};

__device__ __forceinline__ int float_key(float f) {       // monotone float -> int map for atomicMax
    const int i = __builtin_bit_cast(int, f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float key_float(int k) { return __builtin_bit_cast(float, k >= 0 ? k : k ^ 0x7FFFFFFF); }

__global__ void masked_init_k(MaskedWs* ws) {
    ws->s1 = ws->s2 = ws->n1 = ws->n2 = 0.0;
    ws->cmax_key = float_key(-__builtin_inff());
    ws->c = ws->loss = ws->inv = 0.f;
}

__global__ __launch_bounds__(NT) void masked_max_k(const float* __restrict__ pred, const float* __restrict__ tgt, int64_t n,
                                                   MaskedWs* ws) {
    __shared__ float sh[NT / 64];
    float m = -__builtin_inff();
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) m = fmaxf(m, pred[i] - tgt[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < NT / 64; ++i) m = fmaxf(m, sh[i]);
        atomicMax(&ws->cmax_key, float_key(m));
    }
}

__global__ void masked_threshold_k(MaskedWs* ws) { ws->c = 0.2f * key_float(ws->cmax_key); }

template <int KIND>
__global__ __launch_bounds__(NT) void masked_reduce_k(const float* __restrict__ pred, const float* __restrict__ tgt, int64_t n,
                                                      MaskedWs* ws) {
    const float c = KIND == 2 ? ws->c : 0.f;
    float a[4] = {0, 0, 0, 0};
    double acc[4] = {0, 0, 0, 0};
    int run = 0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float t = tgt[i];
        if (t > 0.f) {
            const float d = fabsf(t - pred[i]);
            a[0] += KIND == 1 ? d * d : d;
            a[2] += 1.f;
            if (KIND == 2 && d > c) {
                a[1] += d * d;
                a[3] += 1.f;
            }
        }
        if (++run == 64) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[k] += a[k]; a[k] = 0.f; }
            run = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += a[k];
    block_atomic_add<4>(acc, &ws->s1);
}

__global__ void masked_finalize_k(MaskedWs* ws, float* loss) {
    const double cnt = ws->n1 + ws->n2;
    const float l = (float)((ws->s1 + ws->s2) / cnt);       // empty mask: 0/0 = NaN, as the reference's mean of nothing
    ws->loss = l;
    ws->inv = (float)(1.0 / cnt);
    *loss = l;
}

template <int KIND>
__global__ __launch_bounds__(NT) void masked_bwd_k(const float* __restrict__ pred, const float* __restrict__ tgt, int64_t n,
                                                   const MaskedWs* __restrict__ ws, const float* __restrict__ gscale,
                                                   float* __restrict__ grad) {
    const float k = (gscale ? *gscale : 1.f) * ws->inv, c = ws->c;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float t = tgt[i];
        float r = 0.f;
        if (t > 0.f) {
            const float e = t - pred[i], d = fabsf(e);
            const float sg = e > 0.f ? -1.f : (e < 0.f ? 1.f : 0.f);     // d|e|/dp (0 at e = 0, as autograd)
            if (KIND == 0) r = k * sg;
            if (KIND == 1) r = -2.f * k * e;
            if (KIND == 2) r = k * sg * (1.f + (d > c ? 2.f * d : 0.f));
        }
        grad[i] = r;
    }
}

// ------------------------------------------------------------------ MaskedDepthLoss (criteria.py:17-64)
// ws: double g[4] = {sum mi*gi^2, sum mi, sum mj*gj^2, sum mj}; float out[4] = {1/D, 1/Mi, 1/Mj, loss};
//     double img[N][3] = {n_b, sum d, sum d^2}
struct DepthWsHead { double g[4]; float inv_d, inv_mi, inv_mj, loss; };

__global__ __launch_bounds__(NT) void mdepth_reduce_k(const float* __restrict__ pred, const float* __restrict__ tgt, int H, int W,
                                                      int blocks_per_img, DepthWsHead* head, double* img) {
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    const int64_t hw = (int64_t)H * W;
    const float* p = pred + b * hw;
    const float* t = tgt + b * hw;
    float a[7] = {0, 0, 0, 0, 0, 0, 0};
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    int run = 0;
    for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < hw; i += (int64_t)blocks_per_img * NT) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const float ti = t[i], pi = p[i];
        const bool m = ti > 0.f;
        if (m) {
            const float d = pi - ti;
            a[0] += 1.f;
            a[1] += d;
            a[2] += d * d;
        }
        if (y + 1 < H) {
            const float t2 = t[i + W];
            if (m && t2 > 0.f) {
                const float gi = (p[i + W] - pi) - (t2 - ti);
                a[3] += gi * gi;
                a[4] += 1.f;
            }
        }
        if (x + 1 < W) {
            const float t2 = t[i + 1];
            if (m && t2 > 0.f) {
                const float gj = (p[i + 1] - pi) - (t2 - ti);
                a[5] += gj * gj;
                a[6] += 1.f;
            }
        }
        if (++run == 64) {
#pragma unroll
            for (int k = 0; k < 7; ++k) { acc[k] += a[k]; a[k] = 0.f; }
            run = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] += a[k];
    double lo[3] = {acc[0], acc[1], acc[2]}, hi[4] = {acc[3], acc[4], acc[5], acc[6]};
    block_atomic_add<3>(lo, img + 3 * b);
    __syncthreads();
    block_atomic_add<4>(hi, head->g);
}

__global__ void mdepth_finalize_k(DepthWsHead* head, const double* img, int N, float* loss) {
    double num = 0.0, half = 0.0, den = 0.0;
    for (int b = 0; b < N; ++b) {
        const double n = img[3 * b], s1 = img[3 * b + 1], s2 = img[3 * b + 2];
        num += n * s2;
        half += s1 * s1;
        den += n * n;
    }
    const double l = (num - 0.5 * half) / den + head->g[0] / head->g[1] + head->g[2] / head->g[3];
    head->inv_d = (float)(1.0 / den);
    head->inv_mi = (float)(1.0 / head->g[1]);
    head->inv_mj = (float)(1.0 / head->g[3]);
    head->loss = (float)l;
    *loss = (float)l;
}

__global__ __launch_bounds__(NT) void mdepth_bwd_k(const float* __restrict__ pred, const float* __restrict__ tgt, int N, int H,
                                                   int W, const DepthWsHead* __restrict__ head, const double* __restrict__ img,
                                                   const float* __restrict__ gscale, float* __restrict__ grad) {
    const float gs = gscale ? *gscale : 1.f;
    const float kd = gs * head->inv_d, ki = 2.f * gs * head->inv_mi, kj = 2.f * gs * head->inv_mj;
    const int64_t hw = (int64_t)H * W, n = hw * N;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int b = (int)(i / hw);
        const int64_t r = i - b * hw;
        const int y = (int)(r / W), x = (int)(r - (int64_t)y * W);
        const float ti = tgt[i], pi = pred[i];
        const bool m = ti > 0.f;
        float g = 0.f;
        if (m) {
            g = kd * (2.f * (float)img[3 * b] * (pi - ti) - (float)img[3 * b + 1]);
            // vertical pairs (y-1, y) and (y, y+1); horizontal pairs (x-1, x) and (x, x+1)
            if (y > 0 && tgt[i - W] > 0.f) g += ki * ((pi - pred[i - W]) - (ti - tgt[i - W]));
            if (y + 1 < H && tgt[i + W] > 0.f) g -= ki * ((pred[i + W] - pi) - (tgt[i + W] - ti));
            if (x > 0 && tgt[i - 1] > 0.f) g += kj * ((pi - pred[i - 1]) - (ti - tgt[i - 1]));
            if (x + 1 < W && tgt[i + 1] > 0.f) g -= kj * ((pred[i + 1] - pi) - (tgt[i + 1] - ti));
        }
        grad[i] = g;
    }
}

// ------------------------------------------------------------------ MiDaS losses (criteria.py:154-332)
// q = scale_b * pred + shift_b (per-image least squares, or q = pred), mask = target > 0
//   data term   sum_b sum m r^2 / (2 sum_b M_b)   or   sum m |r| / (2 sum_b M_b)   (r = q - t; the reference's
//               "trimmed" MAE never trims: criteria.py:214-216 slices the (values, indices) tuple)
//   gradient    sum over scales k = 1,2,4,..: on the [::k, ::k] sub-grid, d = m*(q - t),
//               sum |d(x+k) - d(x)| m m' + sum |d(y+k) - d(y)| m m', reduced batch-based (/ sum_b M_{b,k})
//               or image-based (mean_b of / M_{b,k})
constexpr int MSC = 4;              // scales supported (reference default 4)
struct MidasImg {
    double ls[5];                   // a00, a01, a11, b0, b1
    double data, M;                 // data-term sum, valid pixels
    double gsum[MSC], gM[MSC];      // gradient sums and valid sub-grid pixels per scale
    double sg, sgp;                 // sum g, sum g*pred   (g = dL/dq), for the chain through scale / shift
    float scale, shift, det_ok, pad;
    float w[MSC];                   // backward weight of scale k for this image (alpha / divisor)
};
struct MidasHead { float loss, inv_data, pad0, pad1; };

__global__ __launch_bounds__(NT) void midas_ls_k(const float* __restrict__ pred, const float* __restrict__ tgt, int64_t hw,
                                                 int blocks_per_img, MidasImg* img) {
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    const float* p = pred + b * hw;
    const float* t = tgt + b * hw;
    float a[5] = {0, 0, 0, 0, 0};
    double acc[5] = {0, 0, 0, 0, 0};
    int run = 0;
    for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < hw; i += (int64_t)blocks_per_img * NT) {
        const float ti = t[i];
        if (ti > 0.f) {
            const float pi = p[i];
            a[0] += pi * pi; a[1] += pi; a[2] += 1.f; a[3] += pi * ti; a[4] += ti;
        }
        if (++run == 64) {
#pragma unroll
            for (int k = 0; k < 5; ++k) { acc[k] += a[k]; a[k] = 0.f; }
            run = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] += a[k];
    block_atomic_add<5>(acc, img[b].ls);
}

__global__ void midas_fit_k(MidasImg* img, int N, int ssi) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= N) return;
    MidasImg& im = img[b];
    float s = 1.f, h = 0.f, ok = 0.f;
    if (ssi) {
        // the reference solves in fp32 on fp32 sums; the sums here are fp64, the solve follows its formula
        const double a00 = im.ls[0], a01 = im.ls[1], a11 = im.ls[2], b0 = im.ls[3], b1 = im.ls[4];
        const double det = a00 * a11 - a01 * a01;
        s = 0.f;
        if (det != 0.0) {
            s = (float)((a11 * b0 - a01 * b1) / det);
            h = (float)((-a01 * b0 + a00 * b1) / det);
            ok = 1.f;
        }
    }
    im.scale = s;
    im.shift = h;
    im.det_ok = ok;
}

// residual-like quantity d = m * (q - t) at pixel j of image row-major (needs scale/shift); mk: mask source (> 0)
__device__ __forceinline__ float midas_d(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ mk,
                                         int64_t j, float s, float h) {
    return mk[j] > 0.f ? (s * p[j] + h - t[j]) : 0.f;
}

template <int L1>
__global__ __launch_bounds__(NT) void midas_terms_k(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                    const float* __restrict__ msk, int H, int W, int scales, int blocks_per_img,
                                                    MidasImg* img) {
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    const int64_t hw = (int64_t)H * W;
    const float* p = pred + b * hw;
    const float* t = tgt + b * hw;
    const float* mk = msk + b * hw;
    const float s = img[b].scale, h = img[b].shift;
    float a[2 + 2 * MSC];
    double acc[2 + 2 * MSC];
#pragma unroll
    for (int k = 0; k < 2 + 2 * MSC; ++k) { a[k] = 0.f; acc[k] = 0.0; }
    int run = 0;
    for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < hw; i += (int64_t)blocks_per_img * NT) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const bool m = mk[i] > 0.f;
        const float d = midas_d(p, t, mk, i, s, h);
        if (m) {
            a[0] += L1 ? fabsf(d) : d * d;
            a[1] += 1.f;
        }
#pragma unroll
        for (int sc = 0; sc < MSC; ++sc) {
            const int k = 1 << sc;
            if (sc >= scales || (y & (k - 1)) || (x & (k - 1))) continue;
            if (m) a[2 + MSC + sc] += 1.f;
            if (x + k < W && m && mk[i + k] > 0.f) a[2 + sc] += fabsf(midas_d(p, t, mk, i + k, s, h) - d);
            if (y + k < H && m && mk[i + (int64_t)k * W] > 0.f) a[2 + sc] += fabsf(midas_d(p, t, mk, i + (int64_t)k * W, s, h) - d);
        }
        if (++run == 32) {
#pragma unroll
            for (int k = 0; k < 2 + 2 * MSC; ++k) { acc[k] += a[k]; a[k] = 0.f; }
            run = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 2 + 2 * MSC; ++k) acc[k] += a[k];
    double lo[2] = {acc[0], acc[1]}, g[MSC], gm[MSC];
#pragma unroll
    for (int k = 0; k < MSC; ++k) { g[k] = acc[2 + k]; gm[k] = acc[2 + MSC + k]; }
    block_atomic_add<2>(lo, &img[b].data);
    __syncthreads();
    block_atomic_add<MSC>(g, img[b].gsum);
    __syncthreads();
    block_atomic_add<MSC>(gm, img[b].gM);
}

__global__ void midas_loss_k(MidasHead* head, MidasImg* img, int N, float data_weight, float alpha, int scales, int batch_based,
                             float* loss) {
    double data = 0.0, M = 0.0;
    for (int b = 0; b < N; ++b) { data += img[b].data; M += img[b].M; }
    double total = M != 0.0 ? (double)data_weight * data / (2.0 * M) : 0.0;     // reduction_batch_based on (., 2M)
    head->inv_data = M != 0.0 ? (float)((double)data_weight / (2.0 * M)) : 0.f;
    for (int sc = 0; sc < scales; ++sc) {
        if (batch_based) {
            double g = 0.0, gm = 0.0;
            for (int b = 0; b < N; ++b) { g += img[b].gsum[sc]; gm += img[b].gM[sc]; }
            if (alpha > 0.f && gm != 0.0) total += (double)alpha * g / gm;
            for (int b = 0; b < N; ++b) img[b].w[sc] = (alpha > 0.f && gm != 0.0) ? (float)(alpha / gm) : 0.f;
        } else {
            double acc = 0.0;
            for (int b = 0; b < N; ++b) {
                const double gm = img[b].gM[sc];
                acc += gm != 0.0 ? img[b].gsum[sc] / gm : img[b].gsum[sc];
                img[b].w[sc] = alpha > 0.f ? (float)(alpha / (gm != 0.0 ? gm : 1.0) / N) : 0.f;
            }
            if (alpha > 0.f) total += (double)alpha * acc / N;
        }
    }
    head->loss = (float)total;
    *loss = (float)total;
}

__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// g = dL/dq at pixel i of image b
template <int L1>
__device__ __forceinline__ float midas_g(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ mk,
                                         int64_t i, int y, int x, int H, int W, int scales, float s, float h, float inv_data,
                                         const float* __restrict__ w) {
    if (!(mk[i] > 0.f)) return 0.f;
    const float d = s * p[i] + h - t[i];
    float g = L1 ? sgnf(d) * inv_data : 2.f * d * inv_data;
#pragma unroll
    for (int sc = 0; sc < MSC; ++sc) {
        const int k = 1 << sc;
        if (sc >= scales || (y & (k - 1)) || (x & (k - 1))) continue;
        float acc = 0.f;
        if (x + k < W && mk[i + k] > 0.f) acc -= sgnf(midas_d(p, t, mk, i + k, s, h) - d);
        if (x - k >= 0 && mk[i - k] > 0.f) acc += sgnf(d - midas_d(p, t, mk, i - k, s, h));
        if (y + k < H && mk[i + (int64_t)k * W] > 0.f) acc -= sgnf(midas_d(p, t, mk, i + (int64_t)k * W, s, h) - d);
        if (y - k >= 0 && mk[i - (int64_t)k * W] > 0.f) acc += sgnf(d - midas_d(p, t, mk, i - (int64_t)k * W, s, h));
        g += w[sc] * acc;
    }
    return g;
}

template <int L1>
__global__ __launch_bounds__(NT) void midas_gsum_k(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                   const float* __restrict__ msk, int H, int W, int scales, int blocks_per_img,
                                                   const MidasHead* __restrict__ head, MidasImg* img) {
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    const int64_t hw = (int64_t)H * W;
    const float* p = pred + b * hw;
    const float* t = tgt + b * hw;
    const float* mk = msk + b * hw;
    const float s = img[b].scale, h = img[b].shift, inv_data = head->inv_data;
    float wsc[MSC];
#pragma unroll
    for (int k = 0; k < MSC; ++k) wsc[k] = img[b].w[k];
    float a0 = 0.f, a1 = 0.f;
    double acc[2] = {0.0, 0.0};
    int run = 0;
    for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < hw; i += (int64_t)blocks_per_img * NT) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const float g = midas_g<L1>(p, t, mk, i, y, x, H, W, scales, s, h, inv_data, wsc);
        a0 += g;
        a1 += g * p[i];
        if (++run == 64) { acc[0] += a0; acc[1] += a1; a0 = a1 = 0.f; run = 0; }
    }
    acc[0] += a0; acc[1] += a1;
    block_atomic_add<2>(acc, &img[b].sg);
}

// dL/dpred_i = s*g_i + (ds/dp_i) * sum_j g_j p_j + (dh/dp_i) * sum_j g_j   (scale and shift are functions of pred)
template <int L1>
__global__ __launch_bounds__(NT) void midas_bwd_k(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                  const float* __restrict__ msk, int N, int H, int W, int scales, int ssi,
                                                  const MidasHead* __restrict__ head,
                                                  const MidasImg* __restrict__ img, const float* __restrict__ gscale,
                                                  float* __restrict__ grad) {
    const float gs = gscale ? *gscale : 1.f, inv_data = head->inv_data;
    const int64_t hw = (int64_t)H * W, n = hw * N;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int b = (int)(i / hw);
        const int64_t r = i - b * hw;
        const int y = (int)(r / W), x = (int)(r - (int64_t)y * W);
        const MidasImg& im = img[b];
        const float* p = pred + b * hw;
        const float* t = tgt + b * hw;
        const float* mk = msk + b * hw;
        float out = im.scale * midas_g<L1>(p, t, mk, r, y, x, H, W, scales, im.scale, im.shift, inv_data, im.w);
        if (ssi && im.det_ok != 0.f && mk[r] > 0.f) {
            const double a00 = im.ls[0], a01 = im.ls[1], a11 = im.ls[2], b0 = im.ls[3], b1 = im.ls[4];
            const double det = a00 * a11 - a01 * a01;
            const double pi = p[r], ti = t[r];
            const double ddet = 2.0 * pi * a11 - 2.0 * a01;
            const double ds = (a11 * ti - b1 - (double)im.scale * ddet) / det;
            const double dh = (-b0 - a01 * ti + 2.0 * pi * b1 - (double)im.shift * ddet) / det;
            out += (float)(ds * im.sgp + dh * im.sg);
        }
        grad[i] = gs * out;
    }
}

// ------------------------------------------------------------------ robust normalisation (criteria.py:135-152)
// per image: med = torch.median(mask * x) (lower median over ALL H*W values, masked-out pixels count as 0),
// s = clamp(sum m |x - med| / sum m, 1e-6) (med = 0, s = 1 for an image without valid pixels), x' = (x - med) / s.
struct RobustImg {
    unsigned int hist[256];
    unsigned int prefix, rank, pad0, pad1;
    double cnt, sq, sgn, eq;         // sum m, sum m|x-med|, sum m*sign(x-med), #{valid pixels with x == med}
    double sg, sgp;                  // backward: sum g', sum g' * x'
    float med, s, clamped, valid;
};

__device__ __forceinline__ unsigned int order_key(float v) {
    const unsigned int u = __builtin_bit_cast(unsigned int, v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float order_val(unsigned int k) {
    return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__global__ void robust_init_k(RobustImg* img, int N, unsigned int rank) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < N) { img[b].prefix = 0u; img[b].rank = rank; }
}

// histogram of byte `pass` (0 = most significant) of the keys whose higher bytes equal the prefix found so far
__global__ __launch_bounds__(NT) void robust_hist_k(const float* __restrict__ x, const float* __restrict__ msk, int64_t hw,
                                                    int blocks_per_img, int pass, RobustImg* img) {
    __shared__ unsigned int sh[256];
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    sh[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned int prefix = img[b].prefix;
    const int shift = 24 - 8 * pass;
    const float* xi = x + b * hw;
    const float* mk = msk + b * hw;
    for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < hw; i += (int64_t)blocks_per_img * NT) {
        const unsigned int k = order_key(mk[i] > 0.f ? xi[i] : 0.f);
        if (pass == 0 || (k >> (shift + 8)) == prefix) atomicAdd(&sh[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&img[b].hist[threadIdx.x], sh[threadIdx.x]);
}

__global__ void robust_select_k(RobustImg* img, int N, int pass) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= N) return;
    RobustImg& im = img[b];
    unsigned int r = im.rank, bin = 0;
    for (; bin < 255u; ++bin) {
        if (r < im.hist[bin]) break;
        r -= im.hist[bin];
    }
    im.rank = r;
    im.prefix = (im.prefix << 8) | bin;
    for (int i = 0; i < 256; ++i) im.hist[i] = 0u;
    if (pass == 3) im.med = order_val(im.prefix);
}

__global__ __launch_bounds__(NT) void robust_stats_k(const float* __restrict__ x, const float* __restrict__ msk, int64_t hw,
                                                     int blocks_per_img, RobustImg* img) {
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    const float med = img[b].med;
    const float* xi = x + b * hw;
    const float* mk = msk + b * hw;
    float a[4] = {0, 0, 0, 0};
    double acc[4] = {0, 0, 0, 0};
    int run = 0;
    for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < hw; i += (int64_t)blocks_per_img * NT) {
        if (mk[i] > 0.f) {
            const float d = xi[i] - med;
            a[0] += 1.f; a[1] += fabsf(d); a[2] += sgnf(d); a[3] += d == 0.f ? 1.f : 0.f;
        }
        if (++run == 64) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[k] += a[k]; a[k] = 0.f; }
            run = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += a[k];
    block_atomic_add<4>(acc, &img[b].cnt);
}

__global__ void robust_scale_k(RobustImg* img, int N) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= N) return;
    RobustImg& im = img[b];
    if (im.cnt > 0.0) {
        const float raw = (float)(im.sq / im.cnt);
        im.s = raw < 1e-6f ? 1e-6f : raw;
        im.clamped = raw < 1e-6f ? 1.f : 0.f;
        im.valid = 1.f;
    } else {
        im.med = 0.f;
        im.s = 1.f;
        im.clamped = 1.f;
        im.valid = 0.f;
    }
}

__global__ __launch_bounds__(NT) void robust_apply_k(const float* __restrict__ x, int64_t hw, int64_t n, const RobustImg* __restrict__ img,
                                                     float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const RobustImg& im = img[i / hw];
        out[i] = (x[i] - im.med) / im.s;
    }
}

__global__ __launch_bounds__(NT) void robust_bwd_sums_k(const float* __restrict__ g, const float* __restrict__ xn, int64_t hw,
                                                        int blocks_per_img, RobustImg* img) {
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    const float* gi = g + b * hw;
    const float* xi = xn + b * hw;
    float a0 = 0.f, a1 = 0.f;
    double acc[2] = {0.0, 0.0};
    int run = 0;
    for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < hw; i += (int64_t)blocks_per_img * NT) {
        a0 += gi[i];
        a1 += gi[i] * xi[i];
        if (++run == 64) { acc[0] += a0; acc[1] += a1; a0 = a1 = 0.f; run = 0; }
    }
    acc[0] += a0; acc[1] += a1;
    block_atomic_add<2>(acc, &img[b].sg);
}

// x' = (x - med)/s:  dL/dx_i = g'_i/s - (sum g')/s * dmed/dx_i - (sum g' x')/s * ds/dx_i
//   dmed/dx_i = [valid_i and x_i == med] / #such ;  ds/dx_i = (m_i sign(x_i - med) - (sum m sign) dmed/dx_i)/cnt  (0 if clamped)
__global__ __launch_bounds__(NT) void robust_bwd_k(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ msk,
                                                   int64_t hw, int64_t n, const RobustImg* __restrict__ img,
                                                   const float* __restrict__ gscale, float* __restrict__ grad) {
    const float gs = gscale ? *gscale : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const RobustImg& im = img[i / hw];
        const float inv_s = 1.f / im.s;
        float out = g[i] * inv_s;
        if (im.valid != 0.f && msk[i] > 0.f) {
            const float d = x[i] - im.med;
            const float dmed = (d == 0.f && im.eq > 0.0) ? (float)(1.0 / im.eq) : 0.f;
            out -= (float)im.sg * inv_s * dmed;
            if (im.clamped == 0.f) {
                const float ds = (sgnf(d) - (float)im.sgn * dmed) / (float)im.cnt;
                out -= (float)im.sgp * inv_s * ds;
            }
        }
        grad[i] = gs * out;
    }
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

extern "C" size_t mde_masked_loss_ws_bytes(void) { return sizeof(MaskedWs); }

extern "C" int mde_masked_loss_fwd(int kind, const float* pred, const float* target, int64_t n, void* ws, float* loss,
                                   void* stream) {
    MDE_REQUIRE(pred && target && ws && loss && n > 0, "mde_masked_loss_fwd: bad argument");
    MDE_REQUIRE(kind >= 0 && kind <= 2, "mde_masked_loss_fwd: kind=%d (0 L1, 1 MSE, 2 berHu)", kind);
    MDE_REQUIRE(((uintptr_t)ws % 8) == 0, "mde_masked_loss_fwd: ws must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    MaskedWs* w = (MaskedWs*)ws;
    masked_init_k<<<1, 1, 0, st>>>(w);
    if (kind == 2) {
        masked_max_k<<<grid_for(n), NT, 0, st>>>(pred, target, n, w);
        masked_threshold_k<<<1, 1, 0, st>>>(w);
        masked_reduce_k<2><<<grid_for(n), NT, 0, st>>>(pred, target, n, w);
    } else if (kind == 1) {
        masked_reduce_k<1><<<grid_for(n), NT, 0, st>>>(pred, target, n, w);
    } else {
        masked_reduce_k<0><<<grid_for(n), NT, 0, st>>>(pred, target, n, w);
    }
    MDE_LAUNCH_CHECK("masked_reduce_k");
    masked_finalize_k<<<1, 1, 0, st>>>(w, loss);
    MDE_LAUNCH_CHECK("masked_finalize_k");
    return MDE_OK;
}

extern "C" int mde_masked_loss_bwd(int kind, const float* pred, const float* target, int64_t n, const void* ws,
                                   const float* gscale, float* grad, void* stream) {
    MDE_REQUIRE(pred && target && ws && grad && n > 0, "mde_masked_loss_bwd: bad argument");
    MDE_REQUIRE(kind >= 0 && kind <= 2, "mde_masked_loss_bwd: kind=%d (0 L1, 1 MSE, 2 berHu)", kind);
    hipStream_t st = (hipStream_t)stream;
    const MaskedWs* w = (const MaskedWs*)ws;
    if (kind == 2)
        masked_bwd_k<2><<<grid_for(n), NT, 0, st>>>(pred, target, n, w, gscale, grad);
    else if (kind == 1)
        masked_bwd_k<1><<<grid_for(n), NT, 0, st>>>(pred, target, n, w, gscale, grad);
    else
        masked_bwd_k<0><<<grid_for(n), NT, 0, st>>>(pred, target, n, w, gscale, grad);
    MDE_LAUNCH_CHECK("masked_bwd_k");
    return MDE_OK;
}

extern "C" size_t mde_masked_depth_ws_bytes(int N) { return sizeof(DepthWsHead) + (size_t)(N > 0 ? N : 0) * 3 * sizeof(double); }

extern "C" int mde_masked_depth_fwd(const float* pred, const float* target, int N, int H, int W, void* ws, float* loss,
                                    void* stream) {
    MDE_REQUIRE(pred && target && ws && loss && N > 0 && H > 0 && W > 0, "mde_masked_depth_fwd: bad argument");
    MDE_REQUIRE(((uintptr_t)ws % 8) == 0, "mde_masked_depth_fwd: ws must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = mde_check_hip(hipMemsetAsync(ws, 0, mde_masked_depth_ws_bytes(N), st), "hipMemsetAsync(masked depth ws)")) return rc;
    DepthWsHead* head = (DepthWsHead*)ws;
    double* img = (double*)(head + 1);
    const int64_t hw = (int64_t)H * W;
    int bpi = (int)((hw + NT * 16 - 1) / (NT * 16));
    const int cap = (2048 + N - 1) / N;
    bpi = bpi < 1 ? 1 : (bpi > cap ? cap : bpi);
    mdepth_reduce_k<<<N * bpi, NT, 0, st>>>(pred, target, H, W, bpi, head, img);
    MDE_LAUNCH_CHECK("mdepth_reduce_k");
    mdepth_finalize_k<<<1, 1, 0, st>>>(head, img, N, loss);
    MDE_LAUNCH_CHECK("mdepth_finalize_k");
    return MDE_OK;
}

extern "C" int mde_masked_depth_bwd(const float* pred, const float* target, int N, int H, int W, const void* ws,
                                    const float* gscale, float* grad, void* stream) {
    MDE_REQUIRE(pred && target && ws && grad && N > 0 && H > 0 && W > 0, "mde_masked_depth_bwd: bad argument");
    const DepthWsHead* head = (const DepthWsHead*)ws;
    const double* img = (const double*)(head + 1);
    mdepth_bwd_k<<<grid_for((int64_t)N * H * W), NT, 0, (hipStream_t)stream>>>(pred, target, N, H, W, head, img, gscale, grad);
    MDE_LAUNCH_CHECK("mdepth_bwd_k");
    return MDE_OK;
}

extern "C" size_t mde_midas_ws_bytes(int N) { return sizeof(MidasHead) + (size_t)(N > 0 ? N : 0) * sizeof(MidasImg); }

namespace {
int midas_bpi(int N, int64_t hw) {
    int bpi = (int)((hw + NT * 16 - 1) / (NT * 16));
    const int cap = (2048 + N - 1) / N;
    return bpi < 1 ? 1 : (bpi > cap ? cap : bpi);
}
int midas_check(const char* who, const void* pred, const void* target, int N, int H, int W, int data_kind, int scales, const void* ws) {
    MDE_REQUIRE(pred && target && ws && N > 0 && H > 0 && W > 0, "%s: bad argument", who);
    MDE_REQUIRE(data_kind == 0 || data_kind == 1, "%s: data_kind=%d (0 mse, 1 l1 / trimmed)", who, data_kind);
    MDE_REQUIRE(scales >= 0 && scales <= MSC, "%s: scales=%d (0..%d)", who, scales, MSC);
    MDE_REQUIRE(((uintptr_t)ws % 8) == 0, "%s: ws must be 8-byte aligned", who);
    return MDE_OK;
}
}  // namespace

namespace {
int midas_fwd_impl(const float* pred, const float* target, const float* msk, int N, int H, int W, int ssi, int data_kind,
                   float data_weight, float alpha, int scales, int batch_based, void* ws, float* loss, hipStream_t st) {
    if (int rc = mde_check_hip(hipMemsetAsync(ws, 0, mde_midas_ws_bytes(N), st), "hipMemsetAsync(midas ws)")) return rc;
    MidasHead* head = (MidasHead*)ws;
    MidasImg* img = (MidasImg*)(head + 1);
    const int64_t hw = (int64_t)H * W;
    const int bpi = midas_bpi(N, hw);
    if (ssi) {
        midas_ls_k<<<N * bpi, NT, 0, st>>>(pred, target, hw, bpi, img);
        MDE_LAUNCH_CHECK("midas_ls_k");
    }
    midas_fit_k<<<(N + 63) / 64, 64, 0, st>>>(img, N, ssi);
    if (data_kind)
        midas_terms_k<1><<<N * bpi, NT, 0, st>>>(pred, target, msk, H, W, scales, bpi, img);
    else
        midas_terms_k<0><<<N * bpi, NT, 0, st>>>(pred, target, msk, H, W, scales, bpi, img);
    MDE_LAUNCH_CHECK("midas_terms_k");
    midas_loss_k<<<1, 1, 0, st>>>(head, img, N, data_weight, alpha, scales, batch_based, loss);
    MDE_LAUNCH_CHECK("midas_loss_k");
    return MDE_OK;
}

int midas_bwd_impl(const float* pred, const float* target, const float* msk, int N, int H, int W, int ssi, int data_kind,
                   int scales, void* ws, const float* gscale, float* grad, hipStream_t st) {
    MidasHead* head = (MidasHead*)ws;
    MidasImg* img = (MidasImg*)(head + 1);
    const int64_t hw = (int64_t)H * W;
    const int bpi = midas_bpi(N, hw);
    if (ssi) {
        if (data_kind)
            midas_gsum_k<1><<<N * bpi, NT, 0, st>>>(pred, target, msk, H, W, scales, bpi, head, img);
        else
            midas_gsum_k<0><<<N * bpi, NT, 0, st>>>(pred, target, msk, H, W, scales, bpi, head, img);
        MDE_LAUNCH_CHECK("midas_gsum_k");
    }
    if (data_kind)
        midas_bwd_k<1><<<grid_for(hw * N), NT, 0, st>>>(pred, target, msk, N, H, W, scales, ssi, head, img, gscale, grad);
    else
        midas_bwd_k<0><<<grid_for(hw * N), NT, 0, st>>>(pred, target, msk, N, H, W, scales, ssi, head, img, gscale, grad);
    MDE_LAUNCH_CHECK("midas_bwd_k");
    return MDE_OK;
}

// robust normalisation of x (mask source msk) into out; fills img
int robust_normalize(const float* x, const float* msk, int N, int H, int W, RobustImg* img, float* out, hipStream_t st) {
    const int64_t hw = (int64_t)H * W;
    const int bpi = midas_bpi(N, hw);
    robust_init_k<<<(N + 63) / 64, 64, 0, st>>>(img, N, (unsigned int)((hw - 1) / 2));
    for (int pass = 0; pass < 4; ++pass) {
        robust_hist_k<<<N * bpi, NT, 0, st>>>(x, msk, hw, bpi, pass, img);
        MDE_LAUNCH_CHECK("robust_hist_k");
        robust_select_k<<<(N + 63) / 64, 64, 0, st>>>(img, N, pass);
    }
    robust_stats_k<<<N * bpi, NT, 0, st>>>(x, msk, hw, bpi, img);
    MDE_LAUNCH_CHECK("robust_stats_k");
    robust_scale_k<<<(N + 63) / 64, 64, 0, st>>>(img, N);
    robust_apply_k<<<grid_for(hw * N), NT, 0, st>>>(x, hw, hw * N, img, out);
    MDE_LAUNCH_CHECK("robust_apply_k");
    return MDE_OK;
}
}  // namespace

extern "C" int mde_midas_fwd(const float* pred, const float* target, int N, int H, int W, int ssi, int data_kind,
                             float data_weight, float alpha, int scales, int batch_based, void* ws, float* loss, void* stream) {
    if (int rc = midas_check("mde_midas_fwd", pred, target, N, H, W, data_kind, scales, ws)) return rc;
    MDE_REQUIRE(loss, "mde_midas_fwd: null loss");
    return midas_fwd_impl(pred, target, target, N, H, W, ssi, data_kind, data_weight, alpha, scales, batch_based, ws, loss,
                          (hipStream_t)stream);
}

extern "C" int mde_midas_bwd(const float* pred, const float* target, int N, int H, int W, int ssi, int data_kind, int scales,
                             void* ws, const float* gscale, float* grad, void* stream) {
    if (int rc = midas_check("mde_midas_bwd", pred, target, N, H, W, data_kind, scales, ws)) return rc;
    MDE_REQUIRE(grad, "mde_midas_bwd: null grad");
    return midas_bwd_impl(pred, target, target, N, H, W, ssi, data_kind, scales, ws, gscale, grad, (hipStream_t)stream);
}

// TrimmedProcrustesLoss (criteria.py:335-363): both maps robustly normalised, then the (never trimming) trimmed MAE
// + alpha * multi-scale gradient loss, every mask = target > 0 of the ORIGINAL target.
// ws layout: [midas ws][RobustImg pred x N][RobustImg target x N]
extern "C" size_t mde_procrustes_ws_bytes(int N) {
    return ((mde_midas_ws_bytes(N) + 15) / 16) * 16 + 2 * (size_t)(N > 0 ? N : 0) * sizeof(RobustImg);
}

extern "C" int mde_procrustes_fwd(const float* pred, const float* target, int N, int H, int W, float alpha, int scales,
                                  int batch_based, void* ws, float* pred_n, float* target_n, float* loss, void* stream) {
    if (int rc = midas_check("mde_procrustes_fwd", pred, target, N, H, W, 1, scales, ws)) return rc;
    MDE_REQUIRE(pred_n && target_n && loss, "mde_procrustes_fwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    RobustImg* rp = (RobustImg*)((char*)ws + ((mde_midas_ws_bytes(N) + 15) / 16) * 16);
    RobustImg* rt = rp + N;
    if (int rc = mde_check_hip(hipMemsetAsync(rp, 0, 2 * (size_t)N * sizeof(RobustImg), st), "hipMemsetAsync(procrustes ws)")) return rc;
    if (int rc = robust_normalize(pred, target, N, H, W, rp, pred_n, st)) return rc;
    if (int rc = robust_normalize(target, target, N, H, W, rt, target_n, st)) return rc;
    return midas_fwd_impl(pred_n, target_n, target, N, H, W, 0, 1, 1.f, alpha, scales, batch_based, ws, loss, st);
}

// grad = d loss / d pred; gtmp: scratch N*H*W floats (d loss / d pred_n)
extern "C" int mde_procrustes_bwd(const float* pred, const float* target, int N, int H, int W, int scales, void* ws,
                                  const float* pred_n, const float* target_n, const float* gscale, float* gtmp, float* grad,
                                  void* stream) {
    if (int rc = midas_check("mde_procrustes_bwd", pred, target, N, H, W, 1, scales, ws)) return rc;
    MDE_REQUIRE(pred_n && target_n && gtmp && grad, "mde_procrustes_bwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    RobustImg* rp = (RobustImg*)((char*)ws + ((mde_midas_ws_bytes(N) + 15) / 16) * 16);
    if (int rc = midas_bwd_impl(pred_n, target_n, target, N, H, W, 0, 1, scales, ws, nullptr, gtmp, st)) return rc;
    const int64_t hw = (int64_t)H * W;
    const int bpi = midas_bpi(N, hw);
    robust_bwd_sums_k<<<N * bpi, NT, 0, st>>>(gtmp, pred_n, hw, bpi, rp);
    MDE_LAUNCH_CHECK("robust_bwd_sums_k");
    robust_bwd_k<<<grid_for(hw * N), NT, 0, st>>>(gtmp, pred, target, hw, hw * N, rp, gscale, grad);
    MDE_LAUNCH_CHECK("robust_bwd_k");
    return MDE_OK;
}

extern "C" int mde_scale_and_shift(const float* pred, const float* target, int N, int H, int W, void* ws, float* scale,
                                   float* shift, void* stream) {
    MDE_REQUIRE(pred && target && ws && scale && shift && N > 0 && H > 0 && W > 0, "mde_scale_and_shift: bad argument");
    MDE_REQUIRE(((uintptr_t)ws % 8) == 0, "mde_scale_and_shift: ws must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = mde_check_hip(hipMemsetAsync(ws, 0, mde_midas_ws_bytes(N), st), "hipMemsetAsync(midas ws)")) return rc;
    MidasImg* img = (MidasImg*)((MidasHead*)ws + 1);
    const int64_t hw = (int64_t)H * W;
    const int bpi = midas_bpi(N, hw);
    midas_ls_k<<<N * bpi, NT, 0, st>>>(pred, target, hw, bpi, img);
    MDE_LAUNCH_CHECK("midas_ls_k");
    midas_fit_k<<<(N + 63) / 64, 64, 0, st>>>(img, N, 1);
    if (int rc = mde_check_hip(hipMemcpy2DAsync(scale, sizeof(float), &img[0].scale, sizeof(MidasImg), sizeof(float), N,
                                                hipMemcpyDeviceToDevice, st), "hipMemcpy2DAsync(scale)")) return rc;
    return mde_check_hip(hipMemcpy2DAsync(shift, sizeof(float), &img[0].shift, sizeof(MidasImg), sizeof(float), N,
                                          hipMemcpyDeviceToDevice, st), "hipMemcpy2DAsync(shift)");
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
