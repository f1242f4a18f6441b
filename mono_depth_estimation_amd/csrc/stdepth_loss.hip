// The stdepth composite criterion (gfx950): reference modules/base_module.py:124-208 (`setup_criterion` -> `_loss`)
// with stdepth_utils.py's depth_sort (:4-17), composite_layers (:19-42) and separable-Gaussian DSSIM (:63-140).
// pred / targ [N][C][H][W] fp32 (C = 10: front RGBA, back RGBA, 2 depths; C = 20: 3 sortable RGBA layers + back +
// 4 depths), rgba [N][4][H][W].  One pass over the pixels accumulates every masked sum the selected terms need
// (fp64 per workgroup, one atomic per quantity), a finalize kernel forms the terms and the backward coefficients,
// one pass writes the gradient.  The SSIM terms run as LDS-tiled separable 11-tap filters: the forward keeps three
// per-pixel derivative maps, the backward filters those (the Gaussian is symmetric) and adds into the gradient.
#include "mde_common.h"

namespace {

constexpr int NT = 256;
constexpr int NSUM = 13;
constexpr int MAXC = 20;

enum : unsigned {
    T_SILMA = 1u, T_SILMS = 2u, T_MSE = 4u, T_MAE = 8u, T_ALLSSIM = 16u, T_COLORSSIM = 32u, T_COMPOSITE = 64u,
    T_COMPOSITE_SSIM = 128u, T_FBDIV = 256u
};

struct StHead {
    // 0 cnt1 | 1,2 colour |d|, d^2 | 3,4 all-channel |d|, d^2 | 5 cntD | 6,7 depth |d|, d^2 | 8,9,10 silog n, sum, sum^2
    // | 11 composite | 12 fb
    double s[NSUM];
    double ssim_ch[MAXC];     // sum over mask1 of the DSSIM map, per pred channel
    double ssim_comp[4];      // the same for pred_full vs rgba
    float out[12];            // total, depth_silog, color_mae, color_mse, all_mse, all_mae, all_ssim, front_ssim,
                              // back_ssim, composite_mse, composite_ssim, fb_divergence
    float k_cmae, k_cmse, k_amae, k_amse, k_dmae, k_dmse, k_sil, sil_mean, k_comp, k_fb;
    float k_ssim_all, k_ssim_color, k_ssim_comp, pad;
};

struct StCfg {
    int N, C, H, W, d0, d1;   // depth channels [d0, d1)
    unsigned terms;
    float lambda, depth_w, comp_w, fbdiv_w, ssim_w;
};

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

__device__ __forceinline__ float nan_to_num(float v) {   // torch.nan_to_num defaults
    if (v != v) return 0.f;
    if (v == __builtin_inff()) return 3.4028234663852886e38f;
    if (v == -__builtin_inff()) return -3.4028234663852886e38f;
    return v;
}
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// composite_layers over L <= 4 RGBA layers given front to back; un[] = the unclamped result
__device__ __forceinline__ void composite(const float (*ly)[4], int L, float (&un)[4]) {
    float r = ly[0][0], g = ly[0][1], b = ly[0][2], a = ly[0][3];
    for (int i = 1; i < L; ++i) {
        const float k = (1.0f - a) * ly[i][3];
        r = r + k * ly[i][0];
        g = g + k * ly[i][1];
        b = b + k * ly[i][2];
        a = a + k;
    }
    un[0] = r; un[1] = g; un[2] = b; un[3] = a;
}

// the layers of one pixel (its CC channel values in v[]) in compositing order: single layer = front, back;
// multi = 3 layers stably sorted by their depth channel 16 + i, then back
template <int CC, bool SINGLE>
__device__ __forceinline__ int load_layers(const float (&v)[CC], float (*ly)[4]) {
    if constexpr (SINGLE) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) ly[i][c] = v[4 * i + c];
        return 2;
    } else {
        float d[3] = {v[16], v[17], v[18]};
        float l[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) l[i][c] = v[4 * i + c];
        // bubble network with strict compares, rows swapped together with their keys: equal keys keep their order
#define ST_SWAP(i, j)                                                                         \
    if (d[j] < d[i]) {                                                                        \
        const float t = d[i]; d[i] = d[j]; d[j] = t;                                          \
        for (int c = 0; c < 4; ++c) { const float u = l[i][c]; l[i][c] = l[j][c]; l[j][c] = u; } \
    }
        ST_SWAP(0, 1) ST_SWAP(1, 2) ST_SWAP(0, 1)
#undef ST_SWAP
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) ly[i][c] = l[i][c];
#pragma unroll
        for (int c = 0; c < 4; ++c) ly[3][c] = v[12 + c];
        return 4;
    }
}

__global__ void st_init_k(StHead* h) {
    const int t = threadIdx.x;
    if (t < NSUM) h->s[t] = 0.0;
    if (t < MAXC) h->ssim_ch[t] = 0.0;
    if (t < 4) h->ssim_comp[t] = 0.0;
    if (t < 12) h->out[t] = 0.f;
}

// CC: channels of pred / targ; SINGLE: the reference's `single_layer` (depth channels 8:10 and two compositing layers,
// whatever CC is — the laina module's default is out_channels = 20 with single_layer = True), else 16:20 and 3 + 1 layers
template <int CC, bool SINGLE>
__global__ __launch_bounds__(NT) void st_reduce_k(const float* __restrict__ pred, const float* __restrict__ targ,
                                                  const float* __restrict__ rgba, StCfg g, float* __restrict__ pred_full,
                                                  StHead* h) {
    __shared__ double sh[NT / 64];
    const int64_t HW = (int64_t)g.H * g.W, total = HW * g.N;
    double acc[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW, p = i - n * HW;
        const float* pp = pred + n * g.C * HW + p;
        const float* tp = targ + n * g.C * HW + p;
        const float* rp = rgba + n * 4 * HW + p;
        const bool m1 = rp[3 * HW] > 0.f;
        float a[NSUM];
#pragma unroll
        for (int k = 0; k < NSUM; ++k) a[k] = 0.f;
        // every channel of the pixel is loaded up front (CC is a template constant: the loops unroll and the
        // 2*CC + 4 loads are in flight together instead of one dependent trip per channel)
        float pv[CC], tv[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) { pv[c] = pp[(int64_t)c * HW]; tv[c] = tp[(int64_t)c * HW]; }
        if (m1) {
            a[0] = 1.f;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const float d = pv[c] - tv[c];
                if (c < 8) { a[1] += fabsf(d); a[2] += d * d; }
                a[3] += fabsf(d);
                a[4] += d * d;
            }
            if (g.terms & T_FBDIV) {
                float A[3], B[3], A2[3], B2[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { A[c] = pv[c]; B[c] = tv[4 + c]; A2[c] = pv[4 + c]; B2[c] = tv[c]; }
                const float m_1 = sqrtf(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]) * sqrtf(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]) + 1e-3f;
                const float m_2 = sqrtf(A2[0] * A2[0] + A2[1] * A2[1] + A2[2] * A2[2]) * sqrtf(B2[0] * B2[0] + B2[1] * B2[1] + B2[2] * B2[2]) + 1e-3f;
                a[12] = (A[0] * B[0] / m_1 + A[1] * B[1] / m_1 + A[2] * B[2] / m_1) +
                        (A2[0] * B2[0] / m_2 + A2[1] * B2[1] / m_2 + A2[2] * B2[2] / m_2);
            }
        }
#pragma unroll
        for (int c = (SINGLE ? 8 : 16); c < (SINGLE ? 10 : 20); ++c) {   // the depth mask is its own (targ > 0), independent of alpha
            const float t = tv[c];
            if (t > 0.f) {
                const float q = pv[c], d = q - t;
                a[5] += 1.f; a[6] += fabsf(d); a[7] += d * d;
                if (t > 1e-2f) {                         // silog's own validity test inside the masked vector
                    const float dl = logf(q) - logf(t);
                    a[8] += 1.f; a[9] += dl; a[10] += dl * dl;
                }
            }
        }
        if (pred_full || (g.terms & (T_COMPOSITE | T_COMPOSITE_SSIM))) {
            float ly[4][4], un[4];
            const int L = load_layers<CC, SINGLE>(pv, ly);
            composite(ly, L, un);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float pf = clamp01(un[c]);
                if (pred_full) pred_full[n * 4 * HW + (int64_t)c * HW + p] = pf;
                if (m1 && (g.terms & T_COMPOSITE)) {
                    const float e = pf - rp[(int64_t)c * HW];
                    a[11] += nan_to_num(g.comp_w * (e * e));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NSUM; ++k) acc[k] += (double)a[k];
    }
    for (int k = 0; k < NSUM; ++k) {
        const double r = block_sum_d(acc[k], sh);
        if (threadIdx.x == 0 && r != 0.0) atomicAdd(&h->s[k], r);
    }
}

__global__ void st_finalize_k(StHead* h, StCfg g, float* out) {
    const double* s = h->s;
    const double cnt1 = s[0], cntD = s[5];
    float o[12];
    for (int i = 0; i < 12; ++i) o[i] = 0.f;
    // silog over the masked depth vector, nan_to_num'ed twice in the reference (wrapper + call site)
    const double m1 = s[9] / s[8], m2 = s[10] / s[8];
    const float sil = (float)(10.0 * sqrt(m2 - (double)g.lambda * m1 * m1));
    const bool sil_ok = sil == sil && fabsf(sil) != __builtin_inff();
    if (g.terms & (T_SILMA | T_SILMS)) o[1] = g.depth_w * nan_to_num(sil);
    if (g.terms & T_SILMA) o[2] = (float)(s[1] / (8.0 * cnt1));
    if (g.terms & T_SILMS) o[3] = (float)(s[2] / (8.0 * cnt1));
    if (g.terms & T_MSE) o[4] = (float)(s[4] / (g.C * cnt1)) + g.depth_w * (float)(s[7] / cntD);
    if (g.terms & T_MAE) o[5] = (float)(s[3] / (g.C * cnt1)) + g.depth_w * (float)(s[6] / cntD);
    double all = 0.0, front = 0.0, back = 0.0, comp = 0.0;
    for (int c = 0; c < g.C; ++c) all += h->ssim_ch[c];
    for (int c = 0; c < 4; ++c) { front += h->ssim_ch[c]; back += h->ssim_ch[4 + c]; comp += h->ssim_comp[c]; }
    if (g.terms & T_ALLSSIM) o[6] = g.ssim_w * (float)(all / (g.C * cnt1));
    if (g.terms & T_COLORSSIM) { o[7] = g.ssim_w * (float)(front / (4.0 * cnt1)); o[8] = g.ssim_w * (float)(back / (4.0 * cnt1)); }
    if (g.terms & T_COMPOSITE) o[9] = (float)(s[11] / (4.0 * cnt1));
    if (g.terms & T_COMPOSITE_SSIM) o[10] = g.ssim_w * g.comp_w * (float)(comp / (4.0 * cnt1));
    if (g.terms & T_FBDIV) o[11] = g.fbdiv_w * (float)(s[12] / cnt1);
    float total = 0.f;
    for (int i = 1; i < 12; ++i) total += o[i];
    o[0] = total;
    for (int i = 0; i < 12; ++i) { h->out[i] = o[i]; out[i] = o[i]; }
    h->k_cmae = (g.terms & T_SILMA) ? (float)(1.0 / (8.0 * cnt1)) : 0.f;
    h->k_cmse = (g.terms & T_SILMS) ? (float)(2.0 / (8.0 * cnt1)) : 0.f;
    h->k_amae = (g.terms & T_MAE) ? (float)(1.0 / (g.C * cnt1)) : 0.f;
    h->k_amse = (g.terms & T_MSE) ? (float)(2.0 / (g.C * cnt1)) : 0.f;
    h->k_dmae = (g.terms & T_MAE) ? (float)(g.depth_w / cntD) : 0.f;
    h->k_dmse = (g.terms & T_MSE) ? (float)(2.0 * g.depth_w / cntD) : 0.f;
    h->k_sil = ((g.terms & (T_SILMA | T_SILMS)) && sil_ok) ? (float)(g.depth_w * 100.0 / (s[8] * (double)sil)) : 0.f;
    h->sil_mean = (float)m1;
    h->k_comp = (g.terms & T_COMPOSITE) ? (float)(g.comp_w * 2.0 / (4.0 * cnt1)) : 0.f;
    h->k_fb = (g.terms & T_FBDIV) ? (float)(g.fbdiv_w / cnt1) : 0.f;
    h->k_ssim_all = (g.terms & T_ALLSSIM) ? (float)(g.ssim_w / (g.C * cnt1)) : 0.f;
    h->k_ssim_color = (g.terms & T_COLORSSIM) ? (float)(g.ssim_w / (4.0 * cnt1)) : 0.f;
    h->k_ssim_comp = (g.terms & T_COMPOSITE_SSIM) ? (float)(g.ssim_w * g.comp_w / (4.0 * cnt1)) : 0.f;
}

// gfull: d loss / d pred_full from the composite SSIM term (already scaled), or null
template <int CC, bool SINGLE>
__global__ __launch_bounds__(NT) void st_bwd_k(const float* __restrict__ pred, const float* __restrict__ targ,
                                               const float* __restrict__ rgba, StCfg g, const StHead* __restrict__ h,
                                               const float* __restrict__ gscale, const float* __restrict__ gfull,
                                               float* __restrict__ grad) {
    const int64_t HW = (int64_t)g.H * g.W, total = HW * g.N;
    const float gs = gscale ? *gscale : 1.f;
    const float k_cmae = gs * h->k_cmae, k_cmse = gs * h->k_cmse, k_amae = gs * h->k_amae, k_amse = gs * h->k_amse;
    const float k_dmae = gs * h->k_dmae, k_dmse = gs * h->k_dmse, k_sil = gs * h->k_sil, k_comp = gs * h->k_comp;
    const float k_fb = gs * h->k_fb, sil_mean = h->sil_mean;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW, p = i - n * HW;
        const float* pp = pred + n * g.C * HW + p;
        const float* tp = targ + n * g.C * HW + p;
        const float* rp = rgba + n * 4 * HW + p;
        float* gp = grad + n * g.C * HW + p;
        const bool m1 = rp[3 * HW] > 0.f;
        float gr[CC], pv[CC], tv[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) { gr[c] = 0.f; pv[c] = pp[(int64_t)c * HW]; tv[c] = tp[(int64_t)c * HW]; }
        if (m1) {
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const float d = pv[c] - tv[c];
                float v = k_amae * sgn(d) + k_amse * d;
                if (c < 8) v += k_cmae * sgn(d) + k_cmse * d;
                gr[c] = v;
            }
            if (k_fb != 0.f) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {          // (pred[:, :3], targ[:, 4:7]) and (pred[:, 4:7], targ[:, :3])
                    float A[3], B[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { A[c] = pv[4 * s + c]; B[c] = tv[4 * (1 - s) + c]; }
                    const float nA = sqrtf(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]);
                    const float nB = sqrtf(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]);
                    const float mag = nA * nB + 1e-3f;
                    const float dotAB = A[0] * B[0] + A[1] * B[1] + A[2] * B[2];
                    const float q = nA > 0.f ? dotAB / (mag * mag) * nB / nA : 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) gr[4 * s + c] += k_fb * (B[c] / mag - q * A[c]);
                }
            }
        }
#pragma unroll
        for (int c = (SINGLE ? 8 : 16); c < (SINGLE ? 10 : 20); ++c) {
            const float t = tv[c];
            if (t > 0.f) {
                const float q = pv[c], d = q - t;
                float v = k_dmae * sgn(d) + k_dmse * d;
                if (t > 1e-2f) v += k_sil * ((logf(q) - logf(t)) - g.lambda * sil_mean) / q;
                gr[c] += v;
            }
        }
        if (SINGLE && (k_comp != 0.f || gfull)) {
            float ly[4][4], un[4], e[4];
            load_layers<CC, SINGLE>(pv, ly);
            composite(ly, 2, un);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = 0.f;
                if (m1 && k_comp != 0.f) {
                    const float df = clamp01(un[c]) - rp[(int64_t)c * HW];
                    const float sq = g.comp_w * (df * df);
                    if (sq == sq && fabsf(sq) != __builtin_inff()) v = k_comp * df;
                }
                if (gfull) v += gfull[n * 4 * HW + (int64_t)c * HW + p];
                e[c] = (un[c] >= 0.f && un[c] <= 1.f) ? v : 0.f;      // torch.clamp passes the gradient on [min, max]
            }
            const float a0 = ly[0][3], a1 = ly[1][3], k1 = (1.0f - a0) * a1;
            float ga0 = e[3] * (1.0f - a1), ga1 = e[3] * (1.0f - a0);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gr[c] += e[c];
                gr[4 + c] += e[c] * k1;
                ga0 -= e[c] * a1 * ly[1][c];
                ga1 += e[c] * (1.0f - a0) * ly[1][c];
            }
            gr[3] += ga0;
            gr[7] += ga1;
        }
#pragma unroll
        for (int c = 0; c < CC; ++c) gp[(int64_t)c * HW] = gr[c];
    }
}

// ------------------------------------------------------------------------------------------ SSIM
// DSSIM map of clamp(P, 0, 1) vs clamp(T, 0, 1) per plane, 11-tap Gaussian (sigma 1.5), zero padding, summed over
// the pixels with alpha > 0 into sums[plane % Cs].  Tile = 16 x 64 outputs of one plane, halo 5.
constexpr int TH = 16, TW = 64, HALO = 5, SH = TH + 2 * HALO, SW = TW + 2 * HALO, SWP = SW + 1;

struct Gauss { float w[11]; };

__device__ __forceinline__ void ssim_load_tile(const float* __restrict__ src, int H, int W, int y0, int x0, float (*dst)[SWP],
                                               bool clamp) {
    for (int i = threadIdx.x; i < SH * SW; i += NT) {
        const int r = i / SW, c = i - r * SW;
        const int y = y0 + r - HALO, x = x0 + c - HALO;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            v = src[(int64_t)y * W + x];
            if (clamp) v = clamp01(v);
        }
        dst[r][c] = v;
    }
}

__global__ __launch_bounds__(NT) void ssim_fwd_k(const float* __restrict__ P, const float* __restrict__ T,
                                                 const float* __restrict__ rgba, int Cs, int H, int W, int tiles_x,
                                                 Gauss gw, double* __restrict__ sums, float* __restrict__ abc,
                                                 int64_t abc_stride) {
    __shared__ float sp[SH][SWP], st[SH][SWP];
    __shared__ float hz[5][SH][TW];
    __shared__ double sh[NT / 64];
    const int c = blockIdx.y, n = blockIdx.z;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int64_t HW = (int64_t)H * W, plane = ((int64_t)n * Cs + c) * HW;
    ssim_load_tile(P + plane, H, W, y0, x0, sp, true);
    ssim_load_tile(T + plane, H, W, y0, x0, st, true);
    __syncthreads();
    for (int i = threadIdx.x; i < SH * TW; i += NT) {
        const int r = i / TW, x = i - r * TW;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float p = sp[r][x + k], t = st[r][x + k], w = gw.w[k];
            a0 += w * p; a1 += w * t; a2 += w * (p * p); a3 += w * (t * t); a4 += w * (p * t);
        }
        hz[0][r][x] = a0; hz[1][r][x] = a1; hz[2][r][x] = a2; hz[3][r][x] = a3; hz[4][r][x] = a4;
    }
    __syncthreads();
    const int x = threadIdx.x & 63, rb = (threadIdx.x >> 6) * 4;
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = rb + j, y = y0 + r, xx = x0 + x;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = gw.w[k];
            mu1 += w * hz[0][r + k][x]; mu2 += w * hz[1][r + k][x]; e11 += w * hz[2][r + k][x];
            e22 += w * hz[3][r + k][x]; e12 += w * hz[4][r + k][x];
        }
        if (y < H && xx < W) {
            constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
            const float A = 2.f * s12 + C2, B = s1 + s2 + C2;
            const float csr = A / B, cs = fmaxf(csr, 0.f);
            const float ln = 2.f * mu1 * mu2 + C1, ld = mu1 * mu1 + mu2 * mu2 + C1, l = ln / ld;
            const bool m = rgba[((int64_t)n * 4 + 3) * HW + (int64_t)y * W + xx] > 0.f;
            if (m) part += (double)(1.0f - l * cs);
            if (abc) {
                float da = 0.f, db = 0.f, dc = 0.f;       // d DSSIM / d (mu1, E[p^2], E[pt]) at this output
                if (m && csr > 0.f) {
                    const float dl = (2.f * mu2 * ld - ln * 2.f * mu1) / (ld * ld);
                    const float dcs_mu = -2.f * mu2 / B + 2.f * mu1 * A / (B * B);
                    da = -(dl * cs + l * dcs_mu);
                    db = l * A / (B * B);
                    dc = -l * 2.f / B;
                }
                const int64_t o = plane + (int64_t)y * W + xx;
                abc[o] = da; abc[abc_stride + o] = db; abc[2 * abc_stride + o] = dc;
            }
        }
    }
    const double r = block_sum_d(part, sh);
    if (threadIdx.x == 0 && r != 0.0) atomicAdd(&sums[c], r);
}

// grad[plane][x] (+)= k(channel) * [0 <= P(x) <= 1] * (G*a + 2 p G*b + t G*c)(x)
// scale per channel: kc[0] for every channel + kc[1] for channels < 8 (colour SSIM), times gscale
__global__ __launch_bounds__(NT) void ssim_bwd_k(const float* __restrict__ P, const float* __restrict__ T,
                                                 const float* __restrict__ abc, int64_t abc_stride, int Cs, int H, int W,
                                                 int tiles_x, Gauss gw, const float* __restrict__ k_all,
                                                 const float* __restrict__ k_lo8, const float* __restrict__ gscale,
                                                 int accumulate, float* __restrict__ grad) {
    __shared__ float sa[3][SH][SWP];
    __shared__ float hz[3][SH][TW];
    const int c = blockIdx.y, n = blockIdx.z;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int64_t HW = (int64_t)H * W, plane = ((int64_t)n * Cs + c) * HW;
#pragma unroll
    for (int q = 0; q < 3; ++q) ssim_load_tile(abc + q * abc_stride + plane, H, W, y0, x0, sa[q], false);
    __syncthreads();
    for (int i = threadIdx.x; i < SH * TW; i += NT) {
        const int r = i / TW, x = i - r * TW;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = gw.w[k];
            a0 += w * sa[0][r][x + k]; a1 += w * sa[1][r][x + k]; a2 += w * sa[2][r][x + k];
        }
        hz[0][r][x] = a0; hz[1][r][x] = a1; hz[2][r][x] = a2;
    }
    __syncthreads();
    float kc = (k_all ? *k_all : 0.f) + ((k_lo8 && c < 8) ? *k_lo8 : 0.f);
    kc *= gscale ? *gscale : 1.f;
    const int x = threadIdx.x & 63, rb = (threadIdx.x >> 6) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = rb + j, y = y0 + r, xx = x0 + x;
        float fa = 0.f, fb = 0.f, fc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = gw.w[k];
            fa += w * hz[0][r + k][x]; fb += w * hz[1][r + k][x]; fc += w * hz[2][r + k][x];
        }
        if (y < H && xx < W) {
            const int64_t o = plane + (int64_t)y * W + xx;
            const float p = P[o], t = clamp01(T[o]);
            float v = 0.f;
            if (p >= 0.f && p <= 1.f) v = kc * (fa + 2.f * p * fb + t * fc);
            grad[o] = accumulate ? grad[o] + v : v;
        }
    }
}

Gauss make_gauss() {
    Gauss g;
    float s = 0.f;
    for (int i = 0; i < 11; ++i) {
        const float c = (float)(i - 5);
        g.w[i] = expf(-(c * c) / (2.f * 1.5f * 1.5f));
        s += g.w[i];
    }
    for (int i = 0; i < 11; ++i) g.w[i] /= s;
    return g;
}

unsigned needs_ssim_pred(unsigned t) { return t & (T_ALLSSIM | T_COLORSSIM); }

int st_check(const char* fn, const void* pred, const void* targ, const void* rgba, int N, int C, int H, int W, int single,
             unsigned terms, const void* ws, const void* scratch) {
    MDE_REQUIRE(pred && targ && rgba && ws, "%s: null pointer", fn);
    MDE_REQUIRE(N > 0 && H > 0 && W > 0 && (C == 10 || C == 20), "%s: bad shape N=%d C=%d H=%d W=%d (C is 10 or 20)", fn, N, C, H, W);
    MDE_REQUIRE(single || C == 20, "%s: the multi-layer layout needs C = 20", fn);
    MDE_REQUIRE(terms != 0 && terms < 512u, "%s: bad term mask %u", fn, terms);
    MDE_REQUIRE(!(terms & (T_COMPOSITE | T_COMPOSITE_SSIM)) || single,
                "%s: the composite terms are only well-formed single-layer, as in the reference", fn);
    MDE_REQUIRE(!(terms & T_COMPOSITE_SSIM) || (terms & T_COMPOSITE), "%s: composite_ssim without composite", fn);
    MDE_REQUIRE(!(terms & (T_ALLSSIM | T_COLORSSIM | T_COMPOSITE_SSIM)) || scratch, "%s: the SSIM terms need scratch", fn);
    return MDE_OK;
}

inline int grid_for(int64_t n) {
    const int64_t b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// ------------------------------------------------------------------------------------------ the 'ssim' metric
// metrics.py:123 maps 'ssim' to torchmetrics.functional.structural_similarity_index_measure (torchmetrics 0.7.3, absent from
// the image; restated from its published definition, functional/image/ssim.py): 11 x 11 Gaussian window (sigma 1.5),
// data_range = max(max p - min p, max t - min t), c1 = (0.01 R)^2, c2 = (0.03 R)^2, the map of
//   (2 mu_p mu_t + c1)(2 s_pt + c2) / ((mu_p^2 + mu_t^2 + c1)(s_p + s_t + c2))
// computed on the reflect-padded planes and then CROPPED by the 5-pixel pad, so what is averaged is exactly the pixels
// whose window lies inside the image (the padding never reaches them): mean over [5, H-5) x [5, W-5) of every plane.
// The reference hands it clamp_min(pred, 1e-7) and the unmasked target (metrics.py:58-63).
struct SsimMetricWs {
    unsigned lo_p, hi_p, lo_t, hi_t;   // order-preserving integer images of the extrema (float_key)
    unsigned pad0, pad1;
    double sum;
};
__device__ __forceinline__ unsigned float_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
__global__ __launch_bounds__(NT) void ssim_metric_range_k(const float* __restrict__ P, const float* __restrict__ T, int64_t n,
                                                          SsimMetricWs* __restrict__ ws) {
    float lp = __builtin_inff(), hp = -__builtin_inff(), lt = lp, ht = hp;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float p = fmaxf(P[i], 1e-7f), t = T[i];
        lp = fminf(lp, p); hp = fmaxf(hp, p); lt = fminf(lt, t); ht = fmaxf(ht, t);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lp = fminf(lp, __shfl_xor(lp, o, 64)); hp = fmaxf(hp, __shfl_xor(hp, o, 64));
        lt = fminf(lt, __shfl_xor(lt, o, 64)); ht = fmaxf(ht, __shfl_xor(ht, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&ws->lo_p, float_key(lp)); atomicMax(&ws->hi_p, float_key(hp));
        atomicMin(&ws->lo_t, float_key(lt)); atomicMax(&ws->hi_t, float_key(ht));
    }
}
__global__ __launch_bounds__(NT) void ssim_metric_k(const float* __restrict__ P, const float* __restrict__ T, int H, int W, int tiles_x,
                                                    Gauss gw, SsimMetricWs* __restrict__ ws) {
    __shared__ float sp[SH][SWP], st[SH][SWP];
    __shared__ float hz[5][SH][TW];
    __shared__ double sh[NT / 64];
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int64_t plane = (int64_t)blockIdx.y * H * W;
    const float R = fmaxf(key_float(ws->hi_p) - key_float(ws->lo_p), key_float(ws->hi_t) - key_float(ws->lo_t));
    const float C1 = (0.01f * R) * (0.01f * R), C2 = (0.03f * R) * (0.03f * R);
    for (int i = threadIdx.x; i < SH * SW; i += NT) {
        const int r = i / SW, c = i - r * SW;
        const int y = y0 + r - HALO, x = x0 + c - HALO;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;     // (windows that leave the image are never averaged)
        sp[r][c] = in ? fmaxf(P[plane + (int64_t)y * W + x], 1e-7f) : 0.f;
        st[r][c] = in ? T[plane + (int64_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SH * TW; i += NT) {
        const int r = i / TW, x = i - r * TW;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float p = sp[r][x + k], t = st[r][x + k], w = gw.w[k];
            a0 += w * p; a1 += w * t; a2 += w * (p * p); a3 += w * (t * t); a4 += w * (p * t);
        }
        hz[0][r][x] = a0; hz[1][r][x] = a1; hz[2][r][x] = a2; hz[3][r][x] = a3; hz[4][r][x] = a4;
    }
    __syncthreads();
    const int x = threadIdx.x & 63, rb = (threadIdx.x >> 6) * 4;
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = rb + j, y = y0 + r, xx = x0 + x;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = gw.w[k];
            mu1 += w * hz[0][r + k][x]; mu2 += w * hz[1][r + k][x]; e11 += w * hz[2][r + k][x];
            e22 += w * hz[3][r + k][x]; e12 += w * hz[4][r + k][x];
        }
        if (y >= HALO && y < H - HALO && xx >= HALO && xx < W - HALO) {
            const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
            part += (double)(((2.f * mu1 * mu2 + C1) * (2.f * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2)));
        }
    }
    const double r = block_sum_d(part, sh);
    if (threadIdx.x == 0) atomicAdd(&ws->sum, r);
}
__global__ void ssim_metric_init_k(SsimMetricWs* ws) {
    ws->lo_p = ws->lo_t = 0xFFFFFFFFu;
    ws->hi_p = ws->hi_t = 0u;
    ws->sum = 0.0;
}
__global__ void ssim_metric_finish_k(const SsimMetricWs* ws, double count, float* out) { out[0] = (float)(ws->sum / count); }
}  // namespace

extern "C" size_t mde_ssim_metric_ws_bytes(void) { return sizeof(SsimMetricWs); }

extern "C" int mde_ssim_metric(const float* pred, const float* target, int planes, int H, int W, void* ws, float* out, void* stream) {
    MDE_REQUIRE(pred && target && ws && out && planes > 0, "mde_ssim_metric: bad argument");
    MDE_REQUIRE(H > 2 * HALO && W > 2 * HALO, "mde_ssim_metric: the 11 x 11 window needs maps larger than 10 x 10 (got %d x %d)", H, W);
    MDE_REQUIRE(((uintptr_t)ws % 8) == 0, "mde_ssim_metric: ws must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SsimMetricWs* w = reinterpret_cast<SsimMetricWs*>(ws);
    const int64_t n = (int64_t)planes * H * W;
    ssim_metric_init_k<<<1, 1, 0, st>>>(w);
    MDE_LAUNCH_CHECK("ssim_metric_init_k");
    const int64_t nb = (n + NT * 8 - 1) / (NT * 8);
    ssim_metric_range_k<<<(int)(nb > 2048 ? 2048 : nb), NT, 0, st>>>(pred, target, n, w);
    MDE_LAUNCH_CHECK("ssim_metric_range_k");
    const int tiles_x = (W + TW - 1) / TW, tiles = tiles_x * ((H + TH - 1) / TH);
    ssim_metric_k<<<dim3(tiles, planes), NT, 0, st>>>(pred, target, H, W, tiles_x, make_gauss(), w);
    MDE_LAUNCH_CHECK("ssim_metric_k");
    ssim_metric_finish_k<<<1, 1, 0, st>>>(w, (double)planes * (H - 2 * HALO) * (W - 2 * HALO), out);
    MDE_LAUNCH_CHECK("ssim_metric_finish_k");
    return MDE_OK;
}

extern "C" size_t mde_stdepth_ws_bytes(void) { return sizeof(StHead); }

// floats of caller scratch: [pred_full 4][g_full 4][abc of pred 3C][abc of pred_full 12] planes of N*H*W, as needed
extern "C" size_t mde_stdepth_scratch_elems(int N, int C, int H, int W, unsigned terms) {
    size_t planes = 0;
    if (terms & T_COMPOSITE_SSIM) planes += 4 + 4 + 12;
    if (needs_ssim_pred(terms)) planes += 3 * (size_t)C;
    return planes * (size_t)N * H * W;
}

extern "C" int mde_stdepth_fwd(const float* pred, const float* targ, const float* rgba, int N, int C, int H, int W,
                               int single_layer, unsigned terms, float variance_focus, float depth_w, float comp_w, float fbdiv_w,
                               float ssim_w, void* ws, float* scratch, float* pred_full, float* out, void* stream) {
    if (int rc = st_check("mde_stdepth_fwd", pred, targ, rgba, N, C, H, W, single_layer, terms, ws, scratch)) return rc;
    MDE_REQUIRE(out, "mde_stdepth_fwd: null out");
    hipStream_t st = (hipStream_t)stream;
    StHead* h = (StHead*)ws;
    const StCfg g = {N, C, H, W, single_layer ? 8 : 16, single_layer ? 10 : 20, terms, variance_focus, depth_w, comp_w, fbdiv_w, ssim_w};
    const int64_t plane = (int64_t)N * H * W;
    float* full = pred_full;
    float* abc_comp = nullptr;
    float* abc_pred = scratch;
    if (terms & T_COMPOSITE_SSIM) {
        if (!full) full = scratch;
        abc_comp = scratch + 8 * plane;
        abc_pred = scratch + 20 * plane;
    }
    st_init_k<<<1, 64, 0, st>>>(h);
    if (C == 10) st_reduce_k<10, true><<<grid_for(plane), NT, 0, st>>>(pred, targ, rgba, g, full, h);
    else if (single_layer) st_reduce_k<20, true><<<grid_for(plane), NT, 0, st>>>(pred, targ, rgba, g, full, h);
    else st_reduce_k<20, false><<<grid_for(plane), NT, 0, st>>>(pred, targ, rgba, g, full, h);
    MDE_LAUNCH_CHECK("st_reduce_k");
    const int tiles_x = mde_cdiv(W, TW), tiles = tiles_x * mde_cdiv(H, TH);
    const Gauss gw = make_gauss();
    if (needs_ssim_pred(terms)) {
        ssim_fwd_k<<<dim3(tiles, C, N), NT, 0, st>>>(pred, targ, rgba, C, H, W, tiles_x, gw, h->ssim_ch, abc_pred, plane * C);
        MDE_LAUNCH_CHECK("ssim_fwd_k");
    }
    if (terms & T_COMPOSITE_SSIM) {
        ssim_fwd_k<<<dim3(tiles, 4, N), NT, 0, st>>>(full, rgba, rgba, 4, H, W, tiles_x, gw, h->ssim_comp, abc_comp, plane * 4);
        MDE_LAUNCH_CHECK("ssim_fwd_k(composite)");
    }
    st_finalize_k<<<1, 1, 0, st>>>(h, g, out);
    MDE_LAUNCH_CHECK("st_finalize_k");
    return MDE_OK;
}

extern "C" int mde_stdepth_bwd(const float* pred, const float* targ, const float* rgba, int N, int C, int H, int W,
                               int single_layer, unsigned terms, float variance_focus, float depth_w, float comp_w, float fbdiv_w,
                               float ssim_w, const void* ws, float* scratch, const float* pred_full, const float* gscale,
                               float* grad, void* stream) {
    if (int rc = st_check("mde_stdepth_bwd", pred, targ, rgba, N, C, H, W, single_layer, terms, ws, scratch)) return rc;
    MDE_REQUIRE(grad, "mde_stdepth_bwd: null grad");
    hipStream_t st = (hipStream_t)stream;
    const StHead* h = (const StHead*)ws;
    const StCfg g = {N, C, H, W, single_layer ? 8 : 16, single_layer ? 10 : 20, terms, variance_focus, depth_w, comp_w, fbdiv_w, ssim_w};
    const int64_t plane = (int64_t)N * H * W;
    const int tiles_x = mde_cdiv(W, TW), tiles = tiles_x * mde_cdiv(H, TH);
    const Gauss gw = make_gauss();
    const float* gfull = nullptr;
    float* abc_pred = scratch;
    if (terms & T_COMPOSITE_SSIM) {
        const float* full = pred_full ? pred_full : scratch;
        float* gf = scratch + 4 * plane;
        ssim_bwd_k<<<dim3(tiles, 4, N), NT, 0, st>>>(full, rgba, scratch + 8 * plane, plane * 4, 4, H, W, tiles_x, gw,
                                                     &h->k_ssim_comp, nullptr, gscale, 0, gf);
        MDE_LAUNCH_CHECK("ssim_bwd_k(composite)");
        gfull = gf;
        abc_pred = scratch + 20 * plane;
    }
    if (C == 10) st_bwd_k<10, true><<<grid_for(plane), NT, 0, st>>>(pred, targ, rgba, g, h, gscale, gfull, grad);
    else if (single_layer) st_bwd_k<20, true><<<grid_for(plane), NT, 0, st>>>(pred, targ, rgba, g, h, gscale, gfull, grad);
    else st_bwd_k<20, false><<<grid_for(plane), NT, 0, st>>>(pred, targ, rgba, g, h, gscale, gfull, grad);
    MDE_LAUNCH_CHECK("st_bwd_k");
    if (needs_ssim_pred(terms)) {
        ssim_bwd_k<<<dim3(tiles, C, N), NT, 0, st>>>(pred, targ, abc_pred, plane * C, C, H, W, tiles_x, gw, &h->k_ssim_all,
                                                     &h->k_ssim_color, gscale, 1, grad);
        MDE_LAUNCH_CHECK("ssim_bwd_k");
    }
    return MDE_OK;
}
