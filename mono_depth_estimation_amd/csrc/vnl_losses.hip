// The VNL configuration's criteria (gfx950): weighted cross-entropy over depth bins, the virtual-normal loss on
// host-drawn point triples, and the bins <-> depth mapping that sits between the network and the two.
//   WCEL_Loss      reference criteria.py:839-863
//   VNL_Loss       reference criteria.py:866-1045
//   bins_to_depth / depth_to_bins   reference modules/vnl.py:202-230
// All fp32.  WCEL and the bin mapping stream [N][C][HW] logits with one thread per pixel (lanes walk adjacent
// pixels, so each channel step is one coalesced 256 B row per wave); VNL is a gather + a few dozen flops per
// triple, a radix select for the "drop the lowest quarter" rule and an atomic scatter for the gradient.
#include "mde_common.h"

namespace {

constexpr int NT = 256;

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

// =================================================================================== WCEL
struct WcelHead {
    double sum, valid;   // sum over pixels of sum_c w[bin][c] * logp[c]; number of gt > 0
    float loss, pad;
};
// ws = WcelHead | rowsum[C]

__global__ void wcel_init_k(WcelHead* h, const float* __restrict__ weight, int C, float* __restrict__ rowsum) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { h->sum = h->valid = 0.0; h->loss = 0.f; }
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < C) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += weight[(int64_t)r * C + c];
        rowsum[r] = s;
    }
}

__global__ __launch_bounds__(NT) void wcel_fwd_k(const float* __restrict__ logit, const int* __restrict__ bins,
                                                 const float* __restrict__ gt, const float* __restrict__ weight,
                                                 const float* __restrict__ rowsum, int C, int64_t HW, int64_t total,
                                                 float* __restrict__ lse, WcelHead* h) {
    __shared__ double sh[NT / 64];
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    double part = 0.0, val = 0.0;
    if (i < total) {
        const int64_t n = i / HW, p = i - n * HW;
        const float* x = logit + n * C * HW + p;
        float m = -__builtin_inff();
        for (int c = 0; c < C; ++c) m = fmaxf(m, x[(int64_t)c * HW]);
        const int b = bins[i];
        const bool inside = (unsigned)b < (unsigned)C;
        const float* wr = weight + (int64_t)(inside ? b : 0) * C;
        float s = 0.f, a = 0.f;
        for (int c = 0; c < C; ++c) {          // second walk: the workgroup's 256 x C slab is L2-resident
            const float v = x[(int64_t)c * HW];
            s += expf(v - m);
            a += wr[c] * (v - m);
        }
        const float l = logf(s);
        lse[i] = m + l;
        if (inside) part = (double)(a - l * rowsum[b]);
        val = gt[i] > 0.f ? 1.0 : 0.0;
    }
    const double ps = block_sum_d(part, sh);
    const double vs = block_sum_d(val, sh);
    if (threadIdx.x == 0) {
        if (ps != 0.0) atomicAdd(&h->sum, ps);
        if (vs != 0.0) atomicAdd(&h->valid, vs);
    }
}

__global__ void wcel_finalize_k(WcelHead* h, float* loss) {
    const float l = (float)(-h->sum / h->valid);
    h->loss = l;
    *loss = l;
}

// d loss / d x[c] = -(w[bin][c] - softmax[c] * rowsum[bin]) / valid
__global__ __launch_bounds__(NT) void wcel_bwd_k(const float* __restrict__ logit, const int* __restrict__ bins,
                                                 const float* __restrict__ weight, const float* __restrict__ rowsum,
                                                 int C, int64_t HW, int64_t total, const float* __restrict__ lse,
                                                 const WcelHead* __restrict__ h, const float* __restrict__ gscale,
                                                 float* __restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const int64_t n = i / HW, p = i - n * HW;
    const float* x = logit + n * C * HW + p;
    float* g = grad + n * C * HW + p;
    const int b = bins[i];
    if ((unsigned)b >= (unsigned)C) {
        for (int c = 0; c < C; ++c) g[(int64_t)c * HW] = 0.f;
        return;
    }
    const float k = (gscale ? *gscale : 1.f) / (float)h->valid;
    const float* wr = weight + (int64_t)b * C;
    const float rs = rowsum[b], l = lse[i];
    for (int c = 0; c < C; ++c) g[(int64_t)c * HW] = -k * (wr[c] - expf(x[(int64_t)c * HW] - l) * rs);
}

// ---- the production WCEL path: weight table in LDS (C <= 192: 150 x 150 x 4 B = 90 KB), persistent 1024-thread
// workgroups, V pixels per thread as one b32/b128 access per channel, the channel walk in chunks of U so that U
// loads are in flight per lane, ONE walk over the logits (running max / rescaled sum per chunk).  The per-lane
// weight row lookup is an LDS read instead of a 64-address global gather.
constexpr int WNT = 1024;
constexpr int WCEL_LDS_MAX_C = 192;

template <int V> struct PixVec;
template <> struct PixVec<1> { typedef float T; };
template <> struct PixVec<4> { typedef f32x4_t T; };
template <int V> __device__ __forceinline__ float pv_get(const typename PixVec<V>::T& v, int i);
template <> __device__ __forceinline__ float pv_get<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float pv_get<4>(const f32x4_t& v, int i) { return v[i]; }
template <int V> __device__ __forceinline__ void pv_set(typename PixVec<V>::T& v, int i, float x);
template <> __device__ __forceinline__ void pv_set<1>(float& v, int, float x) { v = x; }
template <> __device__ __forceinline__ void pv_set<4>(f32x4_t& v, int i, float x) { v[i] = x; }

__device__ __forceinline__ double wblock_sum_d(double v, double* sh) {   // WNT threads; result in thread 0
    const double r = mde_wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = r;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < WNT / 64; ++i) t += sh[i];
    __syncthreads();
    return t;
}

template <int V, int U>
__global__ __launch_bounds__(WNT) void wcel_fwd_lds_k(const float* __restrict__ logit, const int* __restrict__ bins,
                                                      const float* __restrict__ gt, const float* __restrict__ weight,
                                                      const float* __restrict__ rowsum, int C, int64_t HW, int64_t total,
                                                      float* __restrict__ lse, WcelHead* h) {
    typedef typename PixVec<V>::T vec_t;
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [C][C]
    __shared__ double sh[WNT / 64];
    for (int i = threadIdx.x; i < C * C; i += WNT) wl[i] = weight[i];
    __syncthreads();
    double part = 0.0, val = 0.0;
    for (int64_t base = ((int64_t)blockIdx.x * WNT + threadIdx.x) * V; base < total; base += (int64_t)gridDim.x * WNT * V) {
        const int64_t n = base / HW, p = base - n * HW;          // HW % V == 0: the V pixels share an image
        const float* x = logit + n * C * HW + p;
        int b[V];
        const float* wr[V];
        float m[V], s[V], a[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            b[j] = bins[base + j];
            wr[j] = wl + ((unsigned)b[j] < (unsigned)C ? b[j] : 0) * C;
            m[j] = -__builtin_inff();
            s[j] = a[j] = 0.f;
        }
        int c = 0;
        for (; c + U <= C; c += U) {
            vec_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *(const vec_t*)(x + (int64_t)(c + u) * HW);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float cm = pv_get<V>(v[0], j);
#pragma unroll
                for (int u = 1; u < U; ++u) cm = fmaxf(cm, pv_get<V>(v[u], j));
                const float mn = fmaxf(m[j], cm);
                float acc = s[j] * __expf(m[j] - mn);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float xv = pv_get<V>(v[u], j);
                    acc += __expf(xv - mn);
                    a[j] += wr[j][c + u] * xv;
                }
                s[j] = acc;
                m[j] = mn;
            }
        }
        for (; c < C; ++c) {
            const vec_t v = *(const vec_t*)(x + (int64_t)c * HW);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float xv = pv_get<V>(v, j);
                const float mn = fmaxf(m[j], xv);
                s[j] = s[j] * __expf(m[j] - mn) + __expf(xv - mn);
                m[j] = mn;
                a[j] += wr[j][c] * xv;
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float l = m[j] + logf(s[j]);
            lse[base + j] = l;
            if ((unsigned)b[j] < (unsigned)C) part += (double)(a[j] - l * rowsum[b[j]]);
            val += gt[base + j] > 0.f ? 1.0 : 0.0;
        }
    }
    const double ps = wblock_sum_d(part, sh);
    const double vs = wblock_sum_d(val, sh);
    if (threadIdx.x == 0) {
        if (ps != 0.0) atomicAdd(&h->sum, ps);
        if (vs != 0.0) atomicAdd(&h->valid, vs);
    }
}

template <int V, int U>
__global__ __launch_bounds__(WNT) void wcel_bwd_lds_k(const float* __restrict__ logit, const int* __restrict__ bins,
                                                      const float* __restrict__ weight, const float* __restrict__ rowsum,
                                                      int C, int64_t HW, int64_t total, const float* __restrict__ lse,
                                                      const WcelHead* __restrict__ h, const float* __restrict__ gscale,
                                                      float* __restrict__ grad) {
    typedef typename PixVec<V>::T vec_t;
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [C][C] then one all-zero row
    for (int i = threadIdx.x; i < C * C + C; i += WNT) wl[i] = i < C * C ? weight[i] : 0.f;
    __syncthreads();
    const float k = -(gscale ? *gscale : 1.f) / (float)h->valid;
    for (int64_t base = ((int64_t)blockIdx.x * WNT + threadIdx.x) * V; base < total; base += (int64_t)gridDim.x * WNT * V) {
        const int64_t n = base / HW, p = base - n * HW;
        const float* x = logit + n * C * HW + p;
        float* g = grad + n * C * HW + p;
        const float* wr[V];
        float l[V], rs[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int b = bins[base + j];
            const bool inside = (unsigned)b < (unsigned)C;
            wr[j] = wl + (inside ? b : C) * C;                  // an outside label reads the zero row ...
            rs[j] = inside ? rowsum[b] : 0.f;                   // ... and has no softmax term: gradient 0
            l[j] = lse[base + j];
        }
        int c = 0;
        for (; c + U <= C; c += U) {
            vec_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *(const vec_t*)(x + (int64_t)(c + u) * HW);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                vec_t o;
#pragma unroll
                for (int j = 0; j < V; ++j) pv_set<V>(o, j, k * (wr[j][c + u] - __expf(pv_get<V>(v[u], j) - l[j]) * rs[j]));
                *(vec_t*)(g + (int64_t)(c + u) * HW) = o;
            }
        }
        for (; c < C; ++c) {
            const vec_t v = *(const vec_t*)(x + (int64_t)c * HW);
            vec_t o;
#pragma unroll
            for (int j = 0; j < V; ++j) pv_set<V>(o, j, k * (wr[j][c] - __expf(pv_get<V>(v, j) - l[j]) * rs[j]));
            *(vec_t*)(g + (int64_t)c * HW) = o;
        }
    }
}

template <typename K>
int wcel_lds_attr(K kernel, size_t smem, const char* what) {
    return mde_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)smem), what);
}

int wcel_grid(int64_t total, int V) {
    int cus = 256;
    mde_device_cu_count(&cus);
    const int64_t chunks = (total / V + WNT - 1) / WNT;
    return (int)(chunks < cus ? (chunks < 1 ? 1 : chunks) : cus);
}

// =================================================================================== bin mapping
template <int V, int U>
__global__ __launch_bounds__(NT) void bins_to_depth_fwd_k(const float* __restrict__ prob, const float* __restrict__ border,
                                                          int C, int64_t HW, int64_t total, float* __restrict__ depth) {
    typedef typename PixVec<V>::T vec_t;
    const int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * V;
    if (i >= total) return;
    const int64_t n = i / HW, p = i - n * HW;
    const float* x = prob + n * C * HW + p;
    float s[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = 0.f;
    int c = 0;
    for (; c + U <= C; c += U) {
        vec_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *(const vec_t*)(x + (int64_t)(c + u) * HW);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < V; ++j) s[j] += pv_get<V>(v[u], j) * border[c + u];
    }
    for (; c < C; ++c) {
        const vec_t v = *(const vec_t*)(x + (int64_t)c * HW);
#pragma unroll
        for (int j = 0; j < V; ++j) s[j] += pv_get<V>(v, j) * border[c];
    }
    vec_t o;
#pragma unroll
    for (int j = 0; j < V; ++j) pv_set<V>(o, j, exp10f(s[j]));
    *(vec_t*)(depth + i) = o;
}

// d depth / d prob[c] = depth * ln 10 * border[c]
__global__ __launch_bounds__(NT) void bins_to_depth_bwd_k(const float* __restrict__ depth, const float* __restrict__ gdepth,
                                                          const float* __restrict__ border, int C, int64_t HW,
                                                          int64_t total, float* __restrict__ gprob) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const int64_t n = i / HW, p = i - n * HW;
    float* g = gprob + n * C * HW + p;
    const float k = gdepth[i] * depth[i] * 2.302585092994046f;
    for (int c = 0; c < C; ++c) g[(int64_t)c * HW] = k * border[c];
}

// modules/vnl.py:202-217: also rewrites depth in place (clamped; invalid = -1).
__global__ __launch_bounds__(NT) void depth_to_bins_k(float* __restrict__ depth, int64_t n, float dmin, float dmax,
                                                      float dmin_log, float interval, int C, int* __restrict__ bins) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    float d = depth[i];
    const bool invalid = d < 0.f;
    if (d < dmin) d = dmin;
    if (d > dmax) d = dmax;
    int b = (int)((log10f(d) - dmin_log) / interval);   // .to(torch.int) truncates toward zero
    if (invalid) b = C + 1;
    if (b == C) b = C - 1;
    bins[i] = b;
    depth[i] = invalid ? -1.f : d;
}

// =================================================================================== VNL
struct VnlHead {
    unsigned int hist[256];
    unsigned int prefix, rank, M, k;   // radix-select state; number of kept-by-filter triples; number dropped
    unsigned int cnt_less, cnt_eq, pad0, pad1;
    double sum_gt;                     // sum of the values above the threshold
    float t, loss, wkeep, weq;         // threshold (smallest kept value); result; 1/(M-k); share kept of the ties at t
};
// ws = VnlHead | val[B*n]   (val < 0: triple rejected by the filter)

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct VnlGeo {
    int H, W, n;
    float u0, v0, fx, fy;
};

// criteria.py:905-910: x = (u - u0) * |d| / fx, y likewise, z = d
__device__ __forceinline__ V3 back_project(float d, int p, const VnlGeo& g, float& u, float& v) {
    const int py = p / g.W, px = p - py * g.W;
    u = (float)px - g.u0;
    v = (float)py - g.v0;
    const float a = fabsf(d);
    return {u * a / g.fx, v * a / g.fy, d};
}

// criteria.py:955-988 with the thresholds select_points_groups passes (:996-1000)
__device__ __forceinline__ bool vnl_keep(const V3 (&g)[3]) {
    constexpr float DC = 0.867f, DD = 0.005f, DZ = 0.0001f;
    const bool pad = g[0].z > DZ && g[1].z > DZ && g[2].z > DZ;
    const V3 D[3] = {g[1] - g[0], g[2] - g[0], g[2] - g[1]};
    float nr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) nr[i] = sqrtf(dot3(D[i], D[i]));
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float e = dot3(D[i], D[j]) / (nr[i] * nr[j] + 1e-8f);
            cnt += (e > DC || e < -DC) ? 1 : 0;
        }
    const bool nx = fabsf(D[0].x) < DD || fabsf(D[1].x) < DD || fabsf(D[2].x) < DD;
    const bool ny = fabsf(D[0].y) < DD || fabsf(D[1].y) < DD || fabsf(D[2].y) < DD;
    const bool nz = fabsf(D[0].z) < DD || fabsf(D[1].z) < DD || fabsf(D[2].z) < DD;
    return pad && !((nx && ny && nz) || cnt > 3);
}

// criteria.py:1004: a predicted point j with z == 0 overwrites COORDINATE j of all three points with 1e-4 (the
// reference indexes [B, n, xyz] with a [B, n, point] mask); fixed[c] marks the coordinates that lost their gradient.
__device__ __forceinline__ void vnl_zero_fix(V3 (&q)[3], bool (&fixed)[3]) {
    fixed[0] = q[0].z == 0.f;
    fixed[1] = q[1].z == 0.f;
    fixed[2] = q[2].z == 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (fixed[0]) q[j].x = 0.0001f;
        if (fixed[1]) q[j].y = 0.0001f;
        if (fixed[2]) q[j].z = 0.0001f;
    }
}

// criteria.py:1025-1038: unit normal with the 0.01 patch for a zero norm
__device__ __forceinline__ V3 unit_normal(const V3 (&q)[3], V3& nrm, float& len, float& den) {
    nrm = cross3(q[1] - q[0], q[2] - q[0]);
    len = sqrtf(dot3(nrm, nrm));
    den = len + (len == 0.f ? 0.01f : 0.f);
    return {nrm.x / den, nrm.y / den, nrm.z / den};
}

__device__ __forceinline__ bool vnl_load(const float* __restrict__ dep, const int* __restrict__ p123, int i,
                                         const VnlGeo& g, int (&p)[3], float (&d)[3], V3 (&pt)[3], float (&u)[3], float (&v)[3]) {
    const int hw = g.H * g.W;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        p[j] = p123[j * g.n + i];
        ok = ok && (unsigned)p[j] < (unsigned)hw;
    }
    if (!ok) return false;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        d[j] = dep[p[j]];
        pt[j] = back_project(d[j], p[j], g, u[j], v[j]);
    }
    return true;
}

__global__ void vnl_init_k(VnlHead* h) {
    h->hist[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        h->prefix = h->rank = h->M = h->k = h->cnt_less = h->cnt_eq = 0u;
        h->sum_gt = 0.0;
        h->t = -0.5f;
        h->loss = h->wkeep = h->weq = 0.f;
    }
}

__global__ __launch_bounds__(NT) void vnl_groups_k(const float* __restrict__ gt, const float* __restrict__ pred,
                                                   const int* __restrict__ p123, VnlGeo g, int64_t total,
                                                   float* __restrict__ val, VnlHead* h) {
    __shared__ double sh[NT / 64];
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    double kept = 0.0;
    if (idx < total) {
        const int b = (int)(idx / g.n), i = (int)(idx - (int64_t)b * g.n);
        const int64_t hw = (int64_t)g.H * g.W;
        int p[3];
        float d[3], u[3], v[3];
        V3 pg[3], pq[3];
        float out = -1.f;
        if (vnl_load(gt + b * hw, p123, i, g, p, d, pg, u, v) && vnl_keep(pg)) {
            vnl_load(pred + b * hw, p123, i, g, p, d, pq, u, v);
            bool fixed[3];
            vnl_zero_fix(pq, fixed);
            V3 n0, n1;
            float l0, l1, d0, d1;
            const V3 a = unit_normal(pg, n0, l0, d0), c = unit_normal(pq, n1, l1, d1);
            out = fabsf(a.x - c.x) + fabsf(a.y - c.y) + fabsf(a.z - c.z);
            if (!(out >= 0.f)) out = __builtin_inff();   // NaN sorts last, like torch.sort
            kept = 1.0;
        }
        val[idx] = out;
    }
    const double m = block_sum_d(kept, sh);
    if (threadIdx.x == 0 && m != 0.0) atomicAdd(&h->M, (unsigned int)m);
}

// k = int(M * 0.25) smallest values are dropped (criteria.py:1041-1043)
__global__ void vnl_rank_k(VnlHead* h, int select) {
    const unsigned int k = select ? (unsigned int)((double)h->M * 0.25) : 0u;
    h->k = k;
    h->rank = k;
}

__global__ __launch_bounds__(NT) void vnl_hist_k(const float* __restrict__ val, int64_t total, int pass, VnlHead* h) {
    __shared__ unsigned int sh[256];
    sh[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned int prefix = h->prefix;
    const int shift = 24 - 8 * pass;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const float v = val[i];
        if (v >= 0.f) {   // non-negative floats order like their bit patterns
            const unsigned int key = __builtin_bit_cast(unsigned int, v);
            if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&sh[(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&h->hist[threadIdx.x], sh[threadIdx.x]);
}

__global__ void vnl_select_k(VnlHead* h, int pass) {
    unsigned int r = h->rank, bin = 0;
    for (; bin < 255u; ++bin) {
        if (r < h->hist[bin]) break;
        r -= h->hist[bin];
    }
    h->rank = r;
    h->prefix = (h->prefix << 8) | bin;
    for (int i = 0; i < 256; ++i) h->hist[i] = 0u;
    if (pass == 3) h->t = __builtin_bit_cast(float, h->prefix);
}

__global__ __launch_bounds__(NT) void vnl_sum_k(const float* __restrict__ val, int64_t total, VnlHead* h) {
    __shared__ double sh[NT / 64];
    const float t = h->t;
    double s = 0.0, less = 0.0, eq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const float v = val[i];
        if (v >= 0.f) {
            if (v > t) s += (double)v;
            else if (v == t) eq += 1.0;
            else less += 1.0;
        }
    }
    s = block_sum_d(s, sh);
    less = block_sum_d(less, sh);
    eq = block_sum_d(eq, sh);
    if (threadIdx.x == 0) {
        if (s != 0.0) atomicAdd(&h->sum_gt, s);
        if (less != 0.0) atomicAdd(&h->cnt_less, (unsigned int)less);
        if (eq != 0.0) atomicAdd(&h->cnt_eq, (unsigned int)eq);
    }
}

__global__ void vnl_finalize_k(VnlHead* h, float* loss) {
    const double keep = (double)h->M - (double)h->k;               // 0 kept triples -> mean of nothing = NaN
    const double kept_eq = (double)h->cnt_eq - ((double)h->k - (double)h->cnt_less);
    double total = h->sum_gt;
    if (h->cnt_eq) total += kept_eq * (double)h->t;
    const float l = (float)(total / keep);
    h->loss = l;
    h->wkeep = (float)(1.0 / keep);
    h->weq = h->cnt_eq ? (float)(kept_eq / (double)h->cnt_eq) : 0.f;
    *loss = l;
}

// Gradient of one kept triple, scattered onto its three pixels.  Ties at the threshold share the weight of the
// tied slots that survive the cut (torch.sort would pick some of them; the value of the loss is the same).
__global__ __launch_bounds__(NT) void vnl_bwd_k(const float* __restrict__ gt, const float* __restrict__ pred,
                                                const int* __restrict__ p123, VnlGeo g, int64_t total,
                                                const float* __restrict__ val, const VnlHead* __restrict__ h,
                                                const float* __restrict__ gscale, float* __restrict__ grad) {
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (idx >= total) return;
    const float vv = val[idx], t = h->t;
    if (!(vv >= t) || vv < 0.f) return;
    const float w = (gscale ? *gscale : 1.f) * h->wkeep * (vv == t ? h->weq : 1.f);
    if (w == 0.f) return;
    const int b = (int)(idx / g.n), i = (int)(idx - (int64_t)b * g.n);
    const int64_t hw = (int64_t)g.H * g.W;
    int p[3];
    float d[3], u[3], v[3], dg[3];
    V3 pg[3], pq[3];
    vnl_load(gt + b * hw, p123, i, g, p, dg, pg, u, v);
    vnl_load(pred + b * hw, p123, i, g, p, d, pq, u, v);
    bool fixed[3];
    vnl_zero_fix(pq, fixed);
    V3 n0, n1;
    float l0, l1, d0, d1;
    const V3 a = unit_normal(pg, n0, l0, d0), c = unit_normal(pq, n1, l1, d1);
    // d loss / d c = sign(c - a) (0 at 0, like torch.abs)
    const V3 s = {w * (float)((c.x > a.x) - (c.x < a.x)), w * (float)((c.y > a.y) - (c.y < a.y)),
                  w * (float)((c.z > a.z) - (c.z < a.z))};
    // c = n / den, den = |n| (+0.01 when |n| == 0, where torch's norm has a zero subgradient)
    V3 gn = {s.x / d1, s.y / d1, s.z / d1};
    if (l1 > 0.f) {
        const float k = dot3(s, n1) / (d1 * d1) / l1;
        gn = {gn.x - k * n1.x, gn.y - k * n1.y, gn.z - k * n1.z};
    }
    // n = e1 x e2, e1 = q1 - q0, e2 = q2 - q0
    const V3 e1 = pq[1] - pq[0], e2 = pq[2] - pq[0];
    const V3 g1 = cross3(e2, gn), g2 = cross3(gn, e1);
    V3 gq[3] = {{-(g1.x + g2.x), -(g1.y + g2.y), -(g1.z + g2.z)}, g1, g2};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (fixed[0]) gq[j].x = 0.f;
        if (fixed[1]) gq[j].y = 0.f;
        if (fixed[2]) gq[j].z = 0.f;
        const float sg = (float)((d[j] > 0.f) - (d[j] < 0.f));
        const float gd = gq[j].x * u[j] * sg / g.fx + gq[j].y * v[j] * sg / g.fy + gq[j].z;
        if (gd != 0.f) atomicAdd(grad + b * hw + p[j], gd);
    }
}

int vnl_check(const char* fn, const void* gt, const void* pred, const void* p123, int B, int H, int W, int n, float fx,
              float fy, const void* ws) {
    MDE_REQUIRE(gt && pred && p123 && ws, "%s: null pointer", fn);
    MDE_REQUIRE(B > 0 && H > 0 && W > 0 && n > 0, "%s: bad shape B=%d H=%d W=%d n=%d", fn, B, H, W, n);
    MDE_REQUIRE((int64_t)H * W < (1ll << 31) && (int64_t)B * n < (1ll << 31), "%s: image or sample count too large", fn);
    MDE_REQUIRE(fx != 0.f && fy != 0.f, "%s: zero focal length", fn);
    return MDE_OK;
}

inline int grid_for(int64_t n, int cap = 4096) {
    const int64_t b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// =================================================================================== criterion-fused softmax head (VNL)
// VNL's head hands the caller fp32 NCHW logits and softmax (network/VNL.py:325-327; 2 x 2.95 GB at 16 x 150 x 480 x 640), and the
// reference's criterion -- ModelLoss(bins_to_depth(softmax), logits, bins, gt), modules/vnl.py:255 -- walks them again forwards
// and backwards.  When the criterion is handed tensors that come straight from this library's head it takes a private route
// instead: these kernels read the head's INPUT (16-bit NHWC, the 3 x 3 prediction conv's output, 1.5 GB) + its bias and form
// logits and softmax on the fly; the backward kernel writes d(input) directly.  A WAVE walks pixels, its lanes own the channels
// (c = lane, lane + 64, lane + 128: C <= 192), so a pixel's row is one coalesced read and a weight-table row one more.
constexpr int HF_NT = 256;

// A THREAD owns a pixel and walks its row in 16-byte chunks (8 channels; the row stride is a multiple of 8): 19 loads of 16 bytes
// for 150 channels.  (First version: a wave per pixel, lanes over channels, 2-byte loads and three wave reductions per pixel --
// 1.5 / 2.4 / 3.2 ms for the three kernels at 16 x 480 x 640, SLOWER than the five passes they replace: ten times the load
// instructions and a chain of LDS-crossbar shuffles per pixel.)
__device__ __forceinline__ void hf_chunk(const bf16_t* __restrict__ row, int j, int C, const float* __restrict__ bias, float (&z)[8]) {
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(row + 8 * j);
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = 8 * j + e < C ? (float)v[e] + bias[8 * j + e] : -__builtin_inff();
}

// depth = 10 ** sum_c softmax_c * border_c (modules/vnl.py:219-230 on the head's softmax), log10(depth) and the log-sum-exp:
// ONE walk with a running maximum (the sums are rescaled when it moves)
__global__ __launch_bounds__(HF_NT) void head_depth_fwd_k(const bf16_t* __restrict__ x, int ld, const float* __restrict__ bias,
                                                          const float* __restrict__ border, int64_t P, int C, float* __restrict__ depth,
                                                          float* __restrict__ l10, float* __restrict__ lse) {
    const int nch = (C + 7) >> 3;
    for (int64_t p = (int64_t)blockIdx.x * HF_NT + threadIdx.x; p < P; p += (int64_t)gridDim.x * HF_NT) {
        const bf16_t* row = x + p * ld;
        float m = -__builtin_inff(), s = 0.f, t = 0.f;
        for (int j = 0; j < nch; ++j) {
            float z[8];
            hf_chunk(row, j, C, bias, z);
            float cm = z[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) cm = fmaxf(cm, z[e]);
            if (cm > m) {
                const float sc = expf(m - cm);          // (exp(-inf) = 0 on the first chunk)
                s *= sc;
                t *= sc;
                m = cm;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ee = expf(z[e] - m);        // (0 for the slots past C)
                s += ee;
                t += 8 * j + e < C ? ee * border[8 * j + e] : 0.f;
            }
        }
        const float l = t / s;
        l10[p] = l;
        depth[p] = exp10f(l);
        lse[p] = m + logf(s);
    }
}

// WCEL (criteria.py:839-863) from the head's input: sum_c w[bin][c] * log softmax_c = sum_c w[bin][c] z_c - lse * rowsum[bin];
// the weight table waits in LDS (C x C floats: 90 KB at 150 bins) as in wcel_fwd_lds_k -- a lane's row lookup is an LDS read
__global__ __launch_bounds__(WNT) void head_wcel_fwd_k(const bf16_t* __restrict__ x, int ld, const float* __restrict__ bias,
                                                       const int* __restrict__ bins, const float* __restrict__ gt,
                                                       const float* __restrict__ weight, const float* __restrict__ rowsum,
                                                       const float* __restrict__ lse, int64_t P, int C, WcelHead* h) {
    extern __shared__ float s_w[];
    __shared__ double s_red[WNT / 64];
    for (int i = threadIdx.x; i < C * C; i += WNT) s_w[i] = weight[i];
    __syncthreads();
    const int nch = (C + 7) >> 3;
    double part = 0.0, val = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * WNT + threadIdx.x; p < P; p += (int64_t)gridDim.x * WNT) {
        const int b = bins[p];
        val += gt[p] > 0.f ? 1.0 : 0.0;
        if ((unsigned)b >= (unsigned)C) continue;
        const bf16_t* row = x + p * ld;
        const float* wr = s_w + b * C;
        float a = 0.f;
        for (int j = 0; j < nch; ++j) {
            float z[8];
            hf_chunk(row, j, C, bias, z);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (8 * j + e < C) a += wr[8 * j + e] * z[e];
        }
        part += (double)(a - lse[p] * rowsum[b]);
    }
    const double ps = wblock_sum_d(part, s_red);
    const double vs = wblock_sum_d(val, s_red);
    if (threadIdx.x == 0) {
        if (ps != 0.0) atomicAdd(&h->sum, ps);
        if (vs != 0.0) atomicAdd(&h->valid, vs);
    }
}

// d(total) / d(head input): the WCEL term -k (w[bin][c] - softmax_c rowsum[bin]) with k = gscale / valid, plus the depth's,
// softmax_c * gd * (border_c - log10 depth) with gd = gdepth * depth * ln 10 (bins_to_depth backward through the softmax:
// sum_k softmax_k border_k IS log10 depth, so no reduction is left): one walk, one 16-byte store per chunk.  (The bias gradient
// is the column sums of dx: the caller takes them with mde_bn_stats.)
template <bool TABLE>
__global__ __launch_bounds__(WNT) void head_fused_bwd_k(const bf16_t* __restrict__ x, int ld, const float* __restrict__ bias,
                                                        const int* __restrict__ bins, const float* __restrict__ weight,
                                                        const float* __restrict__ rowsum, const WcelHead* __restrict__ h,
                                                        const float* __restrict__ gscale, const float* __restrict__ lse,
                                                        const float* __restrict__ depth, const float* __restrict__ l10,
                                                        const float* __restrict__ gdepth, const float* __restrict__ border, int64_t P,
                                                        int C, bf16_t* __restrict__ dx, int lddx) {
    extern __shared__ float s_w[];
    if (TABLE) {
        for (int i = threadIdx.x; i < C * C; i += WNT) s_w[i] = weight[i];
        __syncthreads();
    }
    const int nch = (C + 7) >> 3, nst = lddx >> 3;
    const float k = TABLE ? (gscale ? *gscale : 1.f) / (float)h->valid : 0.f;
    for (int64_t p = (int64_t)blockIdx.x * WNT + threadIdx.x; p < P; p += (int64_t)gridDim.x * WNT) {
        const int bn = TABLE ? bins[p] : -1;
        const bool inside = (unsigned)bn < (unsigned)C;
        const float l = lse[p], rs = inside ? rowsum[bn] : 0.f;
        const float gd = gdepth ? gdepth[p] * depth[p] * 2.302585092994046f : 0.f, ld10 = gdepth ? l10[p] : 0.f;
        const float* wr = s_w + (inside ? bn : 0) * C;
        const bf16_t* row = x + p * ld;
        bf16_t* drow = dx + p * lddx;
        for (int j = 0; j < nst; ++j) {
            bf16x8_t o;
            if (j < nch) {
                float z[8];
                hf_chunk(row, j, C, bias, z);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = 8 * j + e;
                    float dz = 0.f;
                    if (c < C) {
                        const float pc = expf(z[e] - l);
                        dz = pc * gd * (border[c] - ld10);
                        if (TABLE && inside) dz -= k * (wr[c] - pc * rs);
                    }
                    o[e] = (bf16_t)dz;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)0.f;
            }
            *reinterpret_cast<bf16x8_t*>(drow + 8 * j) = o;
        }
    }
}

}  // namespace


extern "C" size_t mde_wcel_ws_bytes(int C) { return sizeof(WcelHead) + (size_t)(C > 0 ? C : 0) * sizeof(float); }

extern "C" int mde_wcel_fwd(const float* logit, const int32_t* bins, const float* gt, const float* weight, int N, int C,
                            int64_t HW, void* ws, float* lse, float* loss, void* stream) {
    MDE_REQUIRE(logit && bins && gt && weight && ws && lse && loss, "mde_wcel_fwd: null pointer");
    MDE_REQUIRE(N > 0 && C > 0 && HW > 0 && C <= 4096, "mde_wcel_fwd: bad shape N=%d C=%d HW=%lld", N, C, (long long)HW);
    hipStream_t st = (hipStream_t)stream;
    WcelHead* h = (WcelHead*)ws;
    float* rowsum = (float*)(h + 1);
    const int64_t total = (int64_t)N * HW;
    wcel_init_k<<<mde_cdiv(C, 64), 64, 0, st>>>(h, weight, C, rowsum);
    MDE_LAUNCH_CHECK("wcel_init_k");
    if (C <= WCEL_LDS_MAX_C) {
        const size_t smem = (size_t)C * C * sizeof(float);
        if (HW % 4 == 0 && ((uintptr_t)logit & 15) == 0) {
            if (int rc = wcel_lds_attr(&wcel_fwd_lds_k<4, 4>, smem, "hipFuncSetAttribute(wcel_fwd_lds_k)")) return rc;
            wcel_fwd_lds_k<4, 4><<<wcel_grid(total, 4), WNT, smem, st>>>(logit, bins, gt, weight, rowsum, C, HW, total, lse, h);
        } else {
            if (int rc = wcel_lds_attr(&wcel_fwd_lds_k<1, 8>, smem, "hipFuncSetAttribute(wcel_fwd_lds_k)")) return rc;
            wcel_fwd_lds_k<1, 8><<<wcel_grid(total, 1), WNT, smem, st>>>(logit, bins, gt, weight, rowsum, C, HW, total, lse, h);
        }
        MDE_LAUNCH_CHECK("wcel_fwd_lds_k");
    } else {
        wcel_fwd_k<<<mde_cdiv(total, NT), NT, 0, st>>>(logit, bins, gt, weight, rowsum, C, HW, total, lse, h);
        MDE_LAUNCH_CHECK("wcel_fwd_k");
    }
    wcel_finalize_k<<<1, 1, 0, st>>>(h, loss);
    MDE_LAUNCH_CHECK("wcel_finalize_k");
    return MDE_OK;
}

extern "C" int mde_wcel_bwd(const float* logit, const int32_t* bins, const float* weight, int N, int C, int64_t HW,
                            const void* ws, const float* lse, const float* gscale, float* grad, void* stream) {
    MDE_REQUIRE(logit && bins && weight && ws && lse && grad, "mde_wcel_bwd: null pointer");
    MDE_REQUIRE(N > 0 && C > 0 && HW > 0 && C <= 4096, "mde_wcel_bwd: bad shape N=%d C=%d HW=%lld", N, C, (long long)HW);
    const WcelHead* h = (const WcelHead*)ws;
    const int64_t total = (int64_t)N * HW;
    hipStream_t st = (hipStream_t)stream;
    const float* rowsum = (const float*)(h + 1);
    if (C <= WCEL_LDS_MAX_C) {
        const size_t smem = ((size_t)C * C + C) * sizeof(float);
        if (HW % 4 == 0 && (((uintptr_t)logit | (uintptr_t)grad) & 15) == 0) {
            if (int rc = wcel_lds_attr(&wcel_bwd_lds_k<4, 4>, smem, "hipFuncSetAttribute(wcel_bwd_lds_k)")) return rc;
            wcel_bwd_lds_k<4, 4><<<wcel_grid(total, 4), WNT, smem, st>>>(logit, bins, weight, rowsum, C, HW, total, lse, h,
                                                                         gscale, grad);
        } else {
            if (int rc = wcel_lds_attr(&wcel_bwd_lds_k<1, 8>, smem, "hipFuncSetAttribute(wcel_bwd_lds_k)")) return rc;
            wcel_bwd_lds_k<1, 8><<<wcel_grid(total, 1), WNT, smem, st>>>(logit, bins, weight, rowsum, C, HW, total, lse, h,
                                                                         gscale, grad);
        }
        MDE_LAUNCH_CHECK("wcel_bwd_lds_k");
    } else {
        wcel_bwd_k<<<mde_cdiv(total, NT), NT, 0, st>>>(logit, bins, weight, rowsum, C, HW, total, lse, h, gscale, grad);
        MDE_LAUNCH_CHECK("wcel_bwd_k");
    }
    return MDE_OK;
}

extern "C" int mde_bins_to_depth_fwd(const float* prob, const float* border, int N, int C, int64_t HW, float* depth,
                                     void* stream) {
    MDE_REQUIRE(prob && border && depth, "mde_bins_to_depth_fwd: null pointer");
    MDE_REQUIRE(N > 0 && C > 0 && HW > 0, "mde_bins_to_depth_fwd: bad shape N=%d C=%d HW=%lld", N, C, (long long)HW);
    const int64_t total = (int64_t)N * HW;
    if (HW % 4 == 0 && (((uintptr_t)prob | (uintptr_t)depth) & 15) == 0)
        bins_to_depth_fwd_k<4, 4><<<mde_cdiv(total / 4, NT), NT, 0, (hipStream_t)stream>>>(prob, border, C, HW, total, depth);
    else
        bins_to_depth_fwd_k<1, 8><<<mde_cdiv(total, NT), NT, 0, (hipStream_t)stream>>>(prob, border, C, HW, total, depth);
    MDE_LAUNCH_CHECK("bins_to_depth_fwd_k");
    return MDE_OK;
}

extern "C" int mde_bins_to_depth_bwd(const float* depth, const float* gdepth, const float* border, int N, int C,
                                     int64_t HW, float* gprob, void* stream) {
    MDE_REQUIRE(depth && gdepth && border && gprob, "mde_bins_to_depth_bwd: null pointer");
    MDE_REQUIRE(N > 0 && C > 0 && HW > 0, "mde_bins_to_depth_bwd: bad shape N=%d C=%d HW=%lld", N, C, (long long)HW);
    const int64_t total = (int64_t)N * HW;
    bins_to_depth_bwd_k<<<mde_cdiv(total, NT), NT, 0, (hipStream_t)stream>>>(depth, gdepth, border, C, HW, total, gprob);
    MDE_LAUNCH_CHECK("bins_to_depth_bwd_k");
    return MDE_OK;
}

extern "C" int mde_depth_to_bins(float* depth, int64_t n, float depth_min, float depth_max, float depth_min_log,
                                 float interval, int C, int32_t* bins, void* stream) {
    MDE_REQUIRE(depth && bins, "mde_depth_to_bins: null pointer");
    MDE_REQUIRE(n > 0 && C > 0 && interval > 0.f && depth_min > 0.f && depth_max >= depth_min,
                "mde_depth_to_bins: bad arguments n=%lld C=%d", (long long)n, C);
    depth_to_bins_k<<<mde_cdiv(n, NT), NT, 0, (hipStream_t)stream>>>(depth, n, depth_min, depth_max, depth_min_log, interval,
                                                                     C, bins);
    MDE_LAUNCH_CHECK("depth_to_bins_k");
    return MDE_OK;
}

extern "C" size_t mde_vnl_ws_bytes(int B, int n) {
    return sizeof(VnlHead) + (size_t)(B > 0 ? B : 0) * (size_t)(n > 0 ? n : 0) * sizeof(float);
}

extern "C" int mde_vnl_fwd(const float* gt, const float* pred, const int32_t* p123, int B, int H, int W, int n, float fx,
                           float fy, int select, void* ws, float* loss, void* stream) {
    if (int rc = vnl_check("mde_vnl_fwd", gt, pred, p123, B, H, W, n, fx, fy, ws)) return rc;
    MDE_REQUIRE(loss, "mde_vnl_fwd: null loss");
    hipStream_t st = (hipStream_t)stream;
    VnlHead* h = (VnlHead*)ws;
    float* val = (float*)(h + 1);
    const VnlGeo g = {H, W, n, (float)(W / 2), (float)(H / 2), fx, fy};
    const int64_t total = (int64_t)B * n;
    vnl_init_k<<<1, 256, 0, st>>>(h);
    vnl_groups_k<<<mde_cdiv(total, NT), NT, 0, st>>>(gt, pred, p123, g, total, val, h);
    MDE_LAUNCH_CHECK("vnl_groups_k");
    vnl_rank_k<<<1, 1, 0, st>>>(h, select);
    if (select) {
        for (int pass = 0; pass < 4; ++pass) {
            vnl_hist_k<<<grid_for(total, 512), NT, 0, st>>>(val, total, pass, h);
            vnl_select_k<<<1, 1, 0, st>>>(h, pass);
        }
        MDE_LAUNCH_CHECK("vnl_select_k");
    }
    vnl_sum_k<<<grid_for(total, 512), NT, 0, st>>>(val, total, h);
    vnl_finalize_k<<<1, 1, 0, st>>>(h, loss);
    MDE_LAUNCH_CHECK("vnl_finalize_k");
    return MDE_OK;
}

extern "C" int mde_vnl_bwd(const float* gt, const float* pred, const int32_t* p123, int B, int H, int W, int n, float fx,
                           float fy, const void* ws, const float* gscale, float* grad, void* stream) {
    if (int rc = vnl_check("mde_vnl_bwd", gt, pred, p123, B, H, W, n, fx, fy, ws)) return rc;
    MDE_REQUIRE(grad, "mde_vnl_bwd: null grad");
    hipStream_t st = (hipStream_t)stream;
    const VnlHead* h = (const VnlHead*)ws;
    const VnlGeo g = {H, W, n, (float)(W / 2), (float)(H / 2), fx, fy};
    const int64_t total = (int64_t)B * n;
    if (int rc = mde_check_hip(hipMemsetAsync(grad, 0, (size_t)B * H * W * sizeof(float), st), "mde_vnl_bwd: memset")) return rc;
    vnl_bwd_k<<<mde_cdiv(total, NT), NT, 0, st>>>(gt, pred, p123, g, total, (const float*)(h + 1), h, gscale, grad);
    MDE_LAUNCH_CHECK("vnl_bwd_k");
    return MDE_OK;
}

// ---- criterion-fused softmax head (see the kernels above)
static int hf_grid(int64_t P, int nt, int per_cu) {
    int cus = 256;
    mde_device_cu_count(&cus);
    const int64_t want = (P + nt - 1) / nt, cap = (int64_t)cus * per_cu;
    return (int)(want < cap ? (want < 1 ? 1 : want) : cap);
}
#define HF_CHECK_X(who) \
    MDE_REQUIRE(x && bias && P > 0 && C > 0 && C <= WCEL_LDS_MAX_C && ldx >= C && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, \
                who ": bad argument (C=%d, ldx=%d: rows of 16-byte chunks, at most %d channels)", C, ldx, WCEL_LDS_MAX_C)

extern "C" int mde_vnl_head_depth_fwd(const void* x, int ldx, const float* bias, const float* border, int64_t P, int C, float* depth,
                                      float* log10_depth, float* lse, void* stream) {
    HF_CHECK_X("mde_vnl_head_depth_fwd");
    MDE_REQUIRE(border && depth && log10_depth && lse, "mde_vnl_head_depth_fwd: null pointer");
    head_depth_fwd_k<<<hf_grid(P, HF_NT, 16), HF_NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, bias, border, P, C, depth, log10_depth, lse);
    MDE_LAUNCH_CHECK("head_depth_fwd_k");
    return MDE_OK;
}

extern "C" int mde_vnl_head_wcel_fwd(const void* x, int ldx, const float* bias, const int32_t* bins, const float* gt, const float* weight,
                                     const float* lse, int64_t P, int C, void* ws, float* loss, void* stream) {
    HF_CHECK_X("mde_vnl_head_wcel_fwd");
    MDE_REQUIRE(bins && gt && weight && lse && ws && loss, "mde_vnl_head_wcel_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    WcelHead* h = (WcelHead*)ws;
    float* rowsum = reinterpret_cast<float*>(h + 1);
    wcel_init_k<<<mde_cdiv(C, 64), 64, 0, st>>>(h, weight, C, rowsum);
    const size_t smem = (size_t)C * C * sizeof(float);
    if (int rc = wcel_lds_attr(&head_wcel_fwd_k, smem, "hipFuncSetAttribute(head_wcel_fwd_k)")) return rc;
    head_wcel_fwd_k<<<hf_grid(P, WNT, 1), WNT, smem, st>>>((const bf16_t*)x, ldx, bias, bins, gt, weight, rowsum, lse, P, C, h);
    wcel_finalize_k<<<1, 1, 0, st>>>(h, loss);
    MDE_LAUNCH_CHECK("head_wcel_fwd_k");
    return MDE_OK;
}

extern "C" int mde_vnl_head_bwd(const void* x, int ldx, const float* bias, const int32_t* bins, const float* weight, const void* ws,
                                const float* gscale, const float* lse, const float* depth, const float* log10_depth, const float* gdepth,
                                const float* border, int64_t P, int C, void* dx, int lddx, void* stream) {
    HF_CHECK_X("mde_vnl_head_bwd");
    MDE_REQUIRE(lse && dx && lddx >= C && lddx % 8 == 0 && ((uintptr_t)dx % 16) == 0, "mde_vnl_head_bwd: dx rows are 16-byte chunks (lddx=%d)", lddx);
    MDE_REQUIRE((bins == nullptr) == (weight == nullptr) && (!bins || ws) && (!gdepth || (depth && log10_depth && border)) && (bins || gdepth),
                "mde_vnl_head_bwd: the WCEL term needs bins + weight + ws, the depth term gdepth + depth + log10_depth + border");
    const WcelHead* h = (const WcelHead*)ws;
    const float* rowsum = h ? reinterpret_cast<const float*>(h + 1) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (bins) {
        const size_t smem = (size_t)C * C * sizeof(float);
        if (int rc = wcel_lds_attr(&head_fused_bwd_k<true>, smem, "hipFuncSetAttribute(head_fused_bwd_k)")) return rc;
        head_fused_bwd_k<true><<<hf_grid(P, WNT, 1), WNT, smem, st>>>((const bf16_t*)x, ldx, bias, bins, weight, rowsum, h, gscale, lse, depth,
                                                                       log10_depth, gdepth, border, P, C, (bf16_t*)dx, lddx);
    } else {
        head_fused_bwd_k<false><<<hf_grid(P, WNT, 2), WNT, 0, st>>>((const bf16_t*)x, ldx, bias, nullptr, nullptr, nullptr, nullptr, nullptr, lse,
                                                                     depth, log10_depth, gdepth, border, P, C, (bf16_t*)dx, lddx);
    }
    MDE_LAUNCH_CHECK("head_fused_bwd_k");
    return MDE_OK;
}
