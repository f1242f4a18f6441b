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

// residual-like quantity d = m * (q - t) at pixel j of image row-major (needs scale/shift)
__device__ __forceinline__ float midas_d(const float* __restrict__ p, const float* __restrict__ t, int64_t j, float s, float h) {
    const float tj = t[j];
    return tj > 0.f ? (s * p[j] + h - tj) : 0.f;
}

template <int L1>
__global__ __launch_bounds__(NT) void midas_terms_k(const float* __restrict__ pred, const float* __restrict__ tgt, int H, int W,
                                                    int scales, int blocks_per_img, MidasImg* img) {
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    const int64_t hw = (int64_t)H * W;
    const float* p = pred + b * hw;
    const float* t = tgt + b * hw;
    const float s = img[b].scale, h = img[b].shift;
    float a[2 + 2 * MSC];
    double acc[2 + 2 * MSC];
#pragma unroll
    for (int k = 0; k < 2 + 2 * MSC; ++k) { a[k] = 0.f; acc[k] = 0.0; }
    int run = 0;
    for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < hw; i += (int64_t)blocks_per_img * NT) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const bool m = t[i] > 0.f;
        const float d = midas_d(p, t, i, s, h);
        if (m) {
            a[0] += L1 ? fabsf(d) : d * d;
            a[1] += 1.f;
        }
#pragma unroll
        for (int sc = 0; sc < MSC; ++sc) {
            const int k = 1 << sc;
            if (sc >= scales || (y & (k - 1)) || (x & (k - 1))) continue;
            if (m) a[2 + MSC + sc] += 1.f;
            if (x + k < W && m && t[i + k] > 0.f) a[2 + sc] += fabsf(midas_d(p, t, i + k, s, h) - d);
            if (y + k < H && m && t[i + (int64_t)k * W] > 0.f) a[2 + sc] += fabsf(midas_d(p, t, i + (int64_t)k * W, s, h) - d);
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
__device__ __forceinline__ float midas_g(const float* __restrict__ p, const float* __restrict__ t, int64_t i, int y, int x, int H, int W,
                                         int scales, float s, float h, float inv_data, const float* __restrict__ w) {
    if (!(t[i] > 0.f)) return 0.f;
    const float d = s * p[i] + h - t[i];
    float g = L1 ? sgnf(d) * inv_data : 2.f * d * inv_data;
#pragma unroll
    for (int sc = 0; sc < MSC; ++sc) {
        const int k = 1 << sc;
        if (sc >= scales || (y & (k - 1)) || (x & (k - 1))) continue;
        float acc = 0.f;
        if (x + k < W && t[i + k] > 0.f) acc -= sgnf(midas_d(p, t, i + k, s, h) - d);
        if (x - k >= 0 && t[i - k] > 0.f) acc += sgnf(d - midas_d(p, t, i - k, s, h));
        if (y + k < H && t[i + (int64_t)k * W] > 0.f) acc -= sgnf(midas_d(p, t, i + (int64_t)k * W, s, h) - d);
        if (y - k >= 0 && t[i - (int64_t)k * W] > 0.f) acc += sgnf(d - midas_d(p, t, i - (int64_t)k * W, s, h));
        g += w[sc] * acc;
    }
    return g;
}

template <int L1>
__global__ __launch_bounds__(NT) void midas_gsum_k(const float* __restrict__ pred, const float* __restrict__ tgt, int H, int W,
                                                   int scales, int blocks_per_img, const MidasHead* __restrict__ head, MidasImg* img) {
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    const int64_t hw = (int64_t)H * W;
    const float* p = pred + b * hw;
    const float* t = tgt + b * hw;
    const float s = img[b].scale, h = img[b].shift, inv_data = head->inv_data;
    float wsc[MSC];
#pragma unroll
    for (int k = 0; k < MSC; ++k) wsc[k] = img[b].w[k];
    float a0 = 0.f, a1 = 0.f;
    double acc[2] = {0.0, 0.0};
    int run = 0;
    for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < hw; i += (int64_t)blocks_per_img * NT) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const float g = midas_g<L1>(p, t, i, y, x, H, W, scales, s, h, inv_data, wsc);
        a0 += g;
        a1 += g * p[i];
        if (++run == 64) { acc[0] += a0; acc[1] += a1; a0 = a1 = 0.f; run = 0; }
    }
    acc[0] += a0; acc[1] += a1;
    block_atomic_add<2>(acc, &img[b].sg);
}

// dL/dpred_i = s*g_i + (ds/dp_i) * sum_j g_j p_j + (dh/dp_i) * sum_j g_j   (scale and shift are functions of pred)
template <int L1>
__global__ __launch_bounds__(NT) void midas_bwd_k(const float* __restrict__ pred, const float* __restrict__ tgt, int N, int H, int W,
                                                  int scales, int ssi, const MidasHead* __restrict__ head,
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
        float out = im.scale * midas_g<L1>(p, t, r, y, x, H, W, scales, im.scale, im.shift, inv_data, im.w);
        if (ssi && im.det_ok != 0.f && t[r] > 0.f) {
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

extern "C" int mde_midas_fwd(const float* pred, const float* target, int N, int H, int W, int ssi, int data_kind,
                             float data_weight, float alpha, int scales, int batch_based, void* ws, float* loss, void* stream) {
    if (int rc = midas_check("mde_midas_fwd", pred, target, N, H, W, data_kind, scales, ws)) return rc;
    MDE_REQUIRE(loss, "mde_midas_fwd: null loss");
    hipStream_t st = (hipStream_t)stream;
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
        midas_terms_k<1><<<N * bpi, NT, 0, st>>>(pred, target, H, W, scales, bpi, img);
    else
        midas_terms_k<0><<<N * bpi, NT, 0, st>>>(pred, target, H, W, scales, bpi, img);
    MDE_LAUNCH_CHECK("midas_terms_k");
    midas_loss_k<<<1, 1, 0, st>>>(head, img, N, data_weight, alpha, scales, batch_based, loss);
    MDE_LAUNCH_CHECK("midas_loss_k");
    return MDE_OK;
}

extern "C" int mde_midas_bwd(const float* pred, const float* target, int N, int H, int W, int ssi, int data_kind, int scales,
                             void* ws, const float* gscale, float* grad, void* stream) {
    if (int rc = midas_check("mde_midas_bwd", pred, target, N, H, W, data_kind, scales, ws)) return rc;
    MDE_REQUIRE(grad, "mde_midas_bwd: null grad");
    hipStream_t st = (hipStream_t)stream;
    MidasHead* head = (MidasHead*)ws;
    MidasImg* img = (MidasImg*)(head + 1);
    const int64_t hw = (int64_t)H * W;
    const int bpi = midas_bpi(N, hw);
    if (ssi) {
        if (data_kind)
            midas_gsum_k<1><<<N * bpi, NT, 0, st>>>(pred, target, H, W, scales, bpi, head, img);
        else
            midas_gsum_k<0><<<N * bpi, NT, 0, st>>>(pred, target, H, W, scales, bpi, head, img);
        MDE_LAUNCH_CHECK("midas_gsum_k");
    }
    if (data_kind)
        midas_bwd_k<1><<<grid_for(hw * N), NT, 0, st>>>(pred, target, N, H, W, scales, ssi, head, img, gscale, grad);
    else
        midas_bwd_k<0><<<grid_for(hw * N), NT, 0, st>>>(pred, target, N, H, W, scales, ssi, head, img, gscale, grad);
    MDE_LAUNCH_CHECK("midas_bwd_k");
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
