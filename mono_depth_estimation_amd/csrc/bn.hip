// BatchNorm2d (train + eval) forward/backward for NHWC bf16 activations on gfx950.
// Replaces nn.BatchNorm2d (+ the ReLU / residual add that follows it) everywhere on the
// FCRN path (reference network/FCRN.py:181-188,335,354 and the torchvision Bottlenecks).
//
// All kernels are HBM-bound streaming passes: 16-byte (8 x bf16) accesses per lane, every
// thread owns one fixed 8-channel column so per-channel parameters sit in registers and
// per-channel sums need no cross-lane traffic until the end of the workgroup.
// Statistics: fp32 per-thread partials over <= ~64 rows, combined with LDS atomics per
// workgroup, then fp32 global atomics into one of MDE_STAT_SLOTS rows; the single-wave
// finalize kernels sum the slots in double and re-zero them.
#include "mde_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXC = 2048;
// Rows are walked back to front: the producers before a BN pass (conv epilogues, the previous BN pass) write front
// to back, and the layer-1-sized tensors (315 MB) exceed the 256 MB Infinity Cache, so the END of the tensor is what
// is still cached when the consumer starts; the next consumer in turn finds this pass's last-written rows at the
// front.  Measured in-network: 32.26 -> 32.16 ms/step.
#ifndef MDE_BN_UNROLL
#define MDE_BN_UNROLL 4          // rows in flight per thread in the apply passes (read + write streams: +3 % forward, +7 % backward)
#endif
#ifndef MDE_BN_SNAKE
#define MDE_BN_SNAKE 1
#endif
#ifndef MDE_BN_RED_ROWS
#define MDE_BN_RED_ROWS 32
#endif

__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
    const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    bf16x8_t t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8_t*>(p) = t;
}
__device__ __forceinline__ void ldf8(const float* p, float (&v)[8]) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}

// Workgroup-level combine of per-thread column sums into part[slot][2][C].
// det: the per-thread sums are combined in a FIXED order (row lane 0, 1, ...) and leave the workgroup as integer atomics
// (mde_common.h: deterministic mode); otherwise LDS float atomics + global float atomics.
__device__ __forceinline__ void flush_sums(const float (&s1)[8], const float (&s2)[8], int col, int C,
                                           float* part, float* sh, int det) {
    if (det) {
        const int cpr = C >> 3, rpb = NT / cpr;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sh[threadIdx.x * 16 + e] = s1[e];
            sh[threadIdx.x * 16 + 8 + e] = s2[e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += NT) {
            const int which = i / C, c = i - which * C;
            float t = 0.f;
            for (int q = 0; q < rpb; ++q) t += sh[(q * cpr + (c >> 3)) * 16 + which * 8 + (c & 7)];
            mde_stat_add(part, C, blockIdx.x, which, c, t, 1);
        }
        return;
    }
    for (int i = threadIdx.x; i < 2 * C; i += NT) sh[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        atomicAdd(&sh[col * 8 + e], s1[e]);
        atomicAdd(&sh[C + col * 8 + e], s2[e]);
    }
    __syncthreads();
    float* dst = part + (size_t)(blockIdx.x % MDE_STAT_SLOTS) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += NT) atomicAdd(dst + i, sh[i]);
}

__global__ __launch_bounds__(NT) void bn_stats_k(const bf16_t* __restrict__ x, int64_t M, int C, int ld,
                                                 float* part, int rows_per_blk, int det) {
    __shared__ float sh[2 * MAXC];
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r1 = min(M, r0 + rows_per_blk);
    for (int64_t r_ = rl < rpb ? r0 + rl : r1; r_ < r1; r_ += rpb) {      // (threads past the last whole row group idle)
        const int64_t r = MDE_BN_SNAKE ? M - 1 - r_ : r_;
        float v[8];
        ld8(x + r * ld + col * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
    }
    flush_sums(s1, s2, col, C, part, sh, det);
}

__global__ void bn_finalize_k(float* part, int64_t M, int C, const float* gamma, const float* beta,
                              float* rmean, float* rvar, float momentum, float eps, float* scale,
                              float* shift, float* smean, float* srstd, int det) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double s1 = mde_stat_take(part, C, 0, c, det), s2 = mde_stat_take(part, C, 1, c, det);
    const double mean = s1 / (double)M;
    double var = s2 / (double)M - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    smean[c] = (float)mean;
    srstd[c] = rstd;
    if (rmean) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// Batch moments alone (mean, biased variance) out of a partial-sum buffer, which is zeroed; and the scale / shift of ONE
// BatchNorm from given moments.  DenseNet (Bts.py:283-292 densenet161) normalises the same concatenated channels again in
// every later layer of a block, each with its own gamma / beta / running statistics but the SAME batch moments: they
// are computed once per 48-channel group, when it is produced.
__global__ void bn_moments_k(float* part, int64_t M, int C, float* mean_out, float* var_out, int det) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double s1 = mde_stat_take(part, C, 0, c, det), s2 = mde_stat_take(part, C, 1, c, det);
    const double mean = s1 / (double)M;
    const double var = s2 / (double)M - mean * mean;
    mean_out[c] = (float)mean;
    var_out[c] = (float)(var > 0.0 ? var : 0.0);
}
__global__ void bn_finalize_moments_k(const float* mean_in, const float* var_in, int64_t M, int C, const float* gamma, const float* beta,
                                      float* rmean, float* rvar, float momentum, float eps, float* scale, float* shift,
                                      float* smean, float* srstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = mean_in[c], var = var_in[c];
    const float rstd = (float)(1.0 / sqrt((double)var + (double)eps));
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    smean[c] = mean;
    srstd[c] = rstd;
    if (rmean) {
        const float unb = M > 1 ? (float)((double)var * (double)M / (double)(M - 1)) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
}

__global__ void bn_eval_k(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                          float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

// RES: 0 none, 1 plain residual add, 2 residual with its own scale/shift (second BN site)
template <int RES>
__global__ __launch_bounds__(NT) void bn_apply_k(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                 const float* __restrict__ shift, const bf16_t* __restrict__ r, int ldr,
                                                 const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                 bf16_t* __restrict__ out, int ldo, uint8_t* __restrict__ bits, int64_t M,
                                                 int C, int relu) {
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float sc[8], sh[8], rsc[8], rsh[8];
    ldf8(scale + col * 8, sc);
    ldf8(shift + col * 8, sh);
    if (RES == 2) {
        ldf8(rscale + col * 8, rsc);
        ldf8(rshift + col * 8, rsh);
    }
#pragma unroll MDE_BN_UNROLL
    for (int64_t row_ = rl < rpb ? (int64_t)blockIdx.x * rpb + rl : M; row_ < M; row_ += (int64_t)gridDim.x * rpb) {
        const int64_t row = MDE_BN_SNAKE ? M - 1 - row_ : row_;
        float v[8], q[8];
        ld8(x + row * ldx + col * 8, v);
        if (RES) ld8(r + row * ldr + col * 8, q);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // relu == 2: x is the PRE-activation of an ELU (the eval form of conv -> ELU -> BatchNorm, Bts.py:216-217): ELU here,
            // in fp32 -- a stored ELU output piles up on -1 (every value in (-1, -1 + 2^-10) rounds to -1), a bias per element
            // that the BatchNorm's 1 / sigma then scales
            const float xe = (relu == 2 && v[e] < 0.f) ? expm1f(v[e]) : v[e];
            float y = xe * sc[e] + sh[e];
            if (RES == 1) y += q[e];
            if (RES == 2) y += q[e] * rsc[e] + rsh[e];
            v[e] = relu == 1 ? fmaxf(y, 0.f) : y;
        }
        st8(out + row * ldo + col * 8, v);
        if (bits) {
            uint32_t m = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) m |= (uint32_t)((float)(bf16_t)v[e] > 0.f) << e;   // of the STORED bf16 value
            bits[row * cpr + col] = (uint8_t)m;
        }
    }
}

// MASK: 0 no ReLU, 1 mask from `out` > 0, 2 mask recomputed from x*msc+msh > 0, 3 packed mask bits
template <int MASK>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_k(const bf16_t* __restrict__ dout, int ldd,
                                                      const bf16_t* __restrict__ out, int ldo,
                                                      const bf16_t* __restrict__ x, int ldx,
                                                      const float* __restrict__ smean, const float* __restrict__ srstd,
                                                      const float* __restrict__ msc, const float* __restrict__ msh,
                                                      const uint8_t* __restrict__ bits, int64_t M, int C, float* part,
                                                      int rows_per_blk, int det) {
    __shared__ float sh[2 * MAXC];
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float mu[8], rs[8], ms[8], mh[8];
    ldf8(smean + col * 8, mu);
    ldf8(srstd + col * 8, rs);
    if (MASK == 2) {
        ldf8(msc + col * 8, ms);
        ldf8(msh + col * 8, mh);
    }
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r1 = min(M, r0 + rows_per_blk);
    // (no unrolling here: four rows in flight measured 8-17 % SLOWER for this read-only pass)
    for (int64_t r_ = rl < rpb ? r0 + rl : r1; r_ < r1; r_ += rpb) {      // (threads past the last whole row group idle)
        const int64_t r = MDE_BN_SNAKE ? M - 1 - r_ : r_;
        float g[8], v[8], o[8];
        ld8(dout + r * ldd + col * 8, g);
        ld8(x + r * ldx + col * 8, v);
        if (MASK == 1) ld8(out + r * ldo + col * 8, o);
        uint32_t mb = 0;
        if (MASK == 3) mb = bits[r * cpr + col];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (MASK == 2) o[e] = v[e] * ms[e] + mh[e];
            if (MASK == 3) o[e] = (float)((mb >> e) & 1u);
            const float ge = (MASK && !(o[e] > 0.f)) ? 0.f : g[e];
            s1[e] += ge;
            s2[e] += ge * ((v[e] - mu[e]) * rs[e]);
        }
    }
    flush_sums(s1, s2, col, C, part, sh, det);
}

__global__ void bn_bwd_finalize_k(float* part, int64_t M, int C, const float* gamma, const float* srstd,
                                  float* dgamma, float* dbeta, float* coef, int det) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double s1 = mde_stat_take(part, C, 0, c, det), s2 = mde_stat_take(part, C, 1, c, det);
    if (dgamma) dgamma[c] += (float)s2;
    if (dbeta) dbeta[c] += (float)s1;
    coef[c] = gamma[c] * srstd[c];
    coef[C + c] = (float)(s1 / (double)M);
    coef[2 * C + c] = (float)(s2 / (double)M);
}

template <int MASK>
__global__ __launch_bounds__(NT) void bn_bwd_apply_k(const bf16_t* __restrict__ dout, int ldd,
                                                     const bf16_t* __restrict__ out, int ldo,
                                                     const bf16_t* __restrict__ x, int ldx,
                                                     const float* __restrict__ smean, const float* __restrict__ srstd,
                                                     const float* __restrict__ msc, const float* __restrict__ msh,
                                                     const uint8_t* __restrict__ bits, const float* __restrict__ coef,
                                                     int64_t M, int C, bf16_t* __restrict__ dx, int ldxo, int accumulate, bf16_t* __restrict__ dres,
                                                     int ldres) {
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float mu[8], rs[8], c0[8], c1[8], c2[8], ms[8], mh[8];
    if (MASK == 2) {
        ldf8(msc + col * 8, ms);
        ldf8(msh + col * 8, mh);
    }
    ldf8(smean + col * 8, mu);
    ldf8(srstd + col * 8, rs);
    ldf8(coef + col * 8, c0);
    ldf8(coef + C + col * 8, c1);
    ldf8(coef + 2 * C + col * 8, c2);
#pragma unroll MDE_BN_UNROLL
    for (int64_t row_ = rl < rpb ? (int64_t)blockIdx.x * rpb + rl : M; row_ < M; row_ += (int64_t)gridDim.x * rpb) {
        const int64_t row = MDE_BN_SNAKE ? M - 1 - row_ : row_;
        float g[8], v[8], o[8], d[8];
        ld8(dout + row * ldd + col * 8, g);
        ld8(x + row * ldx + col * 8, v);
        if (MASK == 1) ld8(out + row * ldo + col * 8, o);
        if (accumulate) ld8(dx + row * ldxo + col * 8, d);
        uint32_t mb = 0;
        if (MASK == 3) mb = bits[row * cpr + col];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (MASK == 2) o[e] = v[e] * ms[e] + mh[e];
            if (MASK == 3) o[e] = (float)((mb >> e) & 1u);
            if (MASK && !(o[e] > 0.f)) g[e] = 0.f;
            const float xh = (v[e] - mu[e]) * rs[e];
            const float r = c0[e] * (g[e] - c1[e] - xh * c2[e]);
            d[e] = accumulate ? d[e] + r : r;
        }
        st8(dx + row * ldxo + col * 8, d);
        if (dres) st8(dres + row * ldres + col * 8, g);
    }
}

// ---- residual join of two BN outputs: one masked gradient, two sites (see mde_bn_bwd_reduce2)
__global__ __launch_bounds__(NT) void bn_bwd_reduce2_k(const bf16_t* __restrict__ dout, int ldd,
                                                       const bf16_t* __restrict__ xa, int ldxa,
                                                       const bf16_t* __restrict__ xb, int ldxb,
                                                       const float* __restrict__ mean_a, const float* __restrict__ rstd_a,
                                                       const float* __restrict__ mean_b, const float* __restrict__ rstd_b,
                                                       const uint8_t* __restrict__ bits, int64_t M, int C, float* part_a,
                                                       float* part_b, int rows_per_blk, int det) {
    __shared__ float sh[3 * MAXC];
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float mua[8], rsa[8], mub[8], rsb[8];
    ldf8(mean_a + col * 8, mua);
    ldf8(rstd_a + col * 8, rsa);
    ldf8(mean_b + col * 8, mub);
    ldf8(rstd_b + col * 8, rsb);
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r1 = min(M, r0 + rows_per_blk);
    for (int64_t r_ = rl < rpb ? r0 + rl : r1; r_ < r1; r_ += rpb) {      // (threads past the last whole row group idle)
        const int64_t r = MDE_BN_SNAKE ? M - 1 - r_ : r_;
        float g[8], va[8], vb[8];
        ld8(dout + r * ldd + col * 8, g);
        ld8(xa + r * ldxa + col * 8, va);
        ld8(xb + r * ldxb + col * 8, vb);
        const uint32_t mb = bits ? bits[r * cpr + col] : 0xFFu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ge = ((mb >> e) & 1u) ? g[e] : 0.f;
            s1[e] += ge;
            sa[e] += ge * ((va[e] - mua[e]) * rsa[e]);
            sb[e] += ge * ((vb[e] - mub[e]) * rsb[e]);
        }
    }
    if (det) {                                   // fixed-order combine, integer atomics (see flush_sums)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sh[threadIdx.x * 24 + e] = s1[e];
            sh[threadIdx.x * 24 + 8 + e] = sa[e];
            sh[threadIdx.x * 24 + 16 + e] = sb[e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * C; i += NT) {
            const int which = i / C, c = i - which * C;
            float t = 0.f;
            for (int q = 0; q < rpb; ++q) t += sh[(q * cpr + (c >> 3)) * 24 + which * 8 + (c & 7)];
            if (which == 0) {
                mde_stat_add(part_a, C, blockIdx.x, 0, c, t, 1);
                mde_stat_add(part_b, C, blockIdx.x, 0, c, t, 1);
            } else {
                mde_stat_add(which == 1 ? part_a : part_b, C, blockIdx.x, 1, c, t, 1);
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < 3 * C; i += NT) sh[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        atomicAdd(&sh[col * 8 + e], s1[e]);
        atomicAdd(&sh[C + col * 8 + e], sa[e]);
        atomicAdd(&sh[2 * C + col * 8 + e], sb[e]);
    }
    __syncthreads();
    float* da = part_a + (size_t)(blockIdx.x % MDE_STAT_SLOTS) * 2 * C;
    float* db = part_b + (size_t)(blockIdx.x % MDE_STAT_SLOTS) * 2 * C;
    for (int i = threadIdx.x; i < C; i += NT) {
        atomicAdd(da + i, sh[i]);
        atomicAdd(db + i, sh[i]);
        atomicAdd(da + C + i, sh[C + i]);
        atomicAdd(db + C + i, sh[2 * C + i]);
    }
}

__global__ __launch_bounds__(NT) void bn_bwd_apply2_k(const bf16_t* __restrict__ dout, int ldd,
                                                      const bf16_t* __restrict__ xa, int ldxa,
                                                      const bf16_t* __restrict__ xb, int ldxb,
                                                      const float* __restrict__ mean_a, const float* __restrict__ rstd_a,
                                                      const float* __restrict__ mean_b, const float* __restrict__ rstd_b,
                                                      const uint8_t* __restrict__ bits, const float* __restrict__ coef_a,
                                                      const float* __restrict__ coef_b, int64_t M, int C,
                                                      bf16_t* __restrict__ dxa, int ldda, bf16_t* __restrict__ dxb, int lddb) {
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float mua[8], rsa[8], a0[8], a1[8], a2[8], mub[8], rsb[8], b0[8], b1[8], b2[8];
    ldf8(mean_a + col * 8, mua);
    ldf8(rstd_a + col * 8, rsa);
    ldf8(coef_a + col * 8, a0);
    ldf8(coef_a + C + col * 8, a1);
    ldf8(coef_a + 2 * C + col * 8, a2);
    ldf8(mean_b + col * 8, mub);
    ldf8(rstd_b + col * 8, rsb);
    ldf8(coef_b + col * 8, b0);
    ldf8(coef_b + C + col * 8, b1);
    ldf8(coef_b + 2 * C + col * 8, b2);
#pragma unroll MDE_BN_UNROLL
    for (int64_t row_ = rl < rpb ? (int64_t)blockIdx.x * rpb + rl : M; row_ < M; row_ += (int64_t)gridDim.x * rpb) {
        const int64_t row = MDE_BN_SNAKE ? M - 1 - row_ : row_;
        float g[8], va[8], vb[8], da[8], db[8];
        ld8(dout + row * ldd + col * 8, g);
        ld8(xa + row * ldxa + col * 8, va);
        ld8(xb + row * ldxb + col * 8, vb);
        const uint32_t mb = bits ? bits[row * cpr + col] : 0xFFu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ge = ((mb >> e) & 1u) ? g[e] : 0.f;
            da[e] = a0[e] * (ge - a1[e] - ((va[e] - mua[e]) * rsa[e]) * a2[e]);
            db[e] = b0[e] * (ge - b1[e] - ((vb[e] - mub[e]) * rsb[e]) * b2[e]);
        }
        st8(dxa + row * ldda + col * 8, da);
        st8(dxb + row * lddb + col * 8, db);
    }
}

// ------------------------------------------------------------------ finalize work inside the streaming launches
// The one-workgroup finalize kernels (scale / shift from the statistics; the backward coefficients from the backward sums) sat
// between every pair of dependent passes: 128 launches of ~5 us per FCRN step, ~540 per BTS step, each with the whole chip idle
// around it.  In these forms every workgroup of the streaming pass derives the constants of ITS 8-channel columns from the
// partial sums itself (32 slots x 2 rows from L2), workgroup 0 also writes what later passes need (scale / shift / saved mean /
// 1/std, running statistics; dgamma / dbeta) and zeroes the partial-sum buffer of the OTHER direction -- nobody reads that one
// while this launch runs (the forward sums are dead once backward has started and the other way round), so no workgroup has
// to wait for another.  The sums this launch reads are left as they are: the other direction's launch zeroes them.
__device__ __forceinline__ void fin_sum8(const float* part, int ld, int which, int col, double (&s)[8]) {
    // eight slots in flight at a time: all 32 at once would cost the whole kernel 128 registers more than its streaming loop
    // needs (256 VGPRs = two waves per SIMD: the loop then ran at half the bandwidth)
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.0;
    static_assert(MDE_STAT_SLOTS % 8 == 0, "slots are read in chunks of eight");
#pragma unroll 1
    for (int k0 = 0; k0 < MDE_STAT_SLOTS; k0 += 8) {
        f32x4_t a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float* p = part + ((size_t)(k0 + k) * 2 + which) * ld + col * 8;
            a[k] = *reinterpret_cast<const f32x4_t*>(p);
            b[k] = *reinterpret_cast<const f32x4_t*>(p + 4);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += (double)a[k][e]; s[4 + e] += (double)b[k][e]; }
    }
}

__device__ __forceinline__ void fin_zero(float* z, int64_t n) {
    if (z && blockIdx.x == 0)
        for (int64_t i = threadIdx.x; i < n; i += NT) z[i] = 0.f;
}

// forward: scale / shift of the thread's 8 channels (bn_finalize_k's arithmetic; or bn_finalize_moments_k's from given moments)
__device__ __forceinline__ void fin_fwd(const mde_bn_fin& f, int col, bool writer, float (&sc)[8], float (&sh)[8]) {
    float mean[8], var[8];
    if (f.part) {
        double s1[8], s2[8];
        fin_sum8(f.part, f.part_ld, 0, col, s1);
        fin_sum8(f.part, f.part_ld, 1, col, s2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double m = s1[e] / (double)f.count;
            double v = s2[e] / (double)f.count - m * m;
            mean[e] = (float)m;
            var[e] = (float)(v > 0.0 ? v : 0.0);
            const float rstd = (float)(1.0 / sqrt((v > 0.0 ? v : 0.0) + (double)f.eps));
            sc[e] = f.gamma[col * 8 + e] * rstd;
            sh[e] = f.beta[col * 8 + e] - (float)m * sc[e];
            if (writer) {
                f.srstd[col * 8 + e] = rstd;
                if (f.rmean) {
                    const double unb = f.count > 1 ? (v > 0.0 ? v : 0.0) * (double)f.count / (double)(f.count - 1) : (v > 0.0 ? v : 0.0);
                    f.rmean[col * 8 + e] = (1.f - f.momentum) * f.rmean[col * 8 + e] + f.momentum * (float)m;
                    f.rvar[col * 8 + e] = (1.f - f.momentum) * f.rvar[col * 8 + e] + f.momentum * (float)unb;
                }
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mean[e] = f.mean_in[col * 8 + e];
            var[e] = f.var_in[col * 8 + e];
            const float rstd = (float)(1.0 / sqrt((double)var[e] + (double)f.eps));
            sc[e] = f.gamma[col * 8 + e] * rstd;
            sh[e] = f.beta[col * 8 + e] - mean[e] * sc[e];
            if (writer) {
                f.srstd[col * 8 + e] = rstd;
                if (f.rmean) {
                    const float unb = f.count > 1 ? (float)((double)var[e] * (double)f.count / (double)(f.count - 1)) : var[e];
                    f.rmean[col * 8 + e] = (1.f - f.momentum) * f.rmean[col * 8 + e] + f.momentum * mean[e];
                    f.rvar[col * 8 + e] = (1.f - f.momentum) * f.rvar[col * 8 + e] + f.momentum * unb;
                }
            }
        }
    }
    if (writer) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f.scale[col * 8 + e] = sc[e];
            f.shift[col * 8 + e] = sh[e];
            f.smean[col * 8 + e] = mean[e];
        }
    }
}

template <int RES>
__global__ __launch_bounds__(NT) void bn_apply_fin_k(const bf16_t* __restrict__ x, int ldx, const mde_bn_fin f,
                                                     const bf16_t* __restrict__ r, int ldr, const mde_bn_fin fr,
                                                     bf16_t* __restrict__ out, int ldo, uint8_t* __restrict__ bits, int64_t M,
                                                     int C, int relu) {
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    extern __shared__ float s_fin[];          // [4][C]: the constants, derived ONCE per workgroup by its first row of threads
    float sc[8], sh[8], rsc[8], rsh[8];
    if (rl == 0) {
        fin_fwd(f, col, blockIdx.x == 0, sc, sh);
        if (RES == 2) fin_fwd(fr, col, blockIdx.x == 0, rsc, rsh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s_fin[col * 8 + e] = sc[e];
            s_fin[C + col * 8 + e] = sh[e];
            if (RES == 2) {
                s_fin[2 * C + col * 8 + e] = rsc[e];
                s_fin[3 * C + col * 8 + e] = rsh[e];
            }
        }
    }
    fin_zero(f.zero, f.zero_n);
    if (RES == 2) fin_zero(fr.zero, fr.zero_n);
    __syncthreads();
    if (rl != 0 && rl < rpb) {
        ldf8(s_fin + col * 8, sc);
        ldf8(s_fin + C + col * 8, sh);
        if (RES == 2) {
            ldf8(s_fin + 2 * C + col * 8, rsc);
            ldf8(s_fin + 3 * C + col * 8, rsh);
        }
    }
#pragma unroll MDE_BN_UNROLL
    for (int64_t row_ = rl < rpb ? (int64_t)blockIdx.x * rpb + rl : M; row_ < M; row_ += (int64_t)gridDim.x * rpb) {
        const int64_t row = MDE_BN_SNAKE ? M - 1 - row_ : row_;
        float v[8], q[8];
        ld8(x + row * ldx + col * 8, v);
        if (RES) ld8(r + row * ldr + col * 8, q);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = v[e] * sc[e] + sh[e];
            if (RES == 1) y += q[e];
            if (RES == 2) y += q[e] * rsc[e] + rsh[e];
            v[e] = relu ? fmaxf(y, 0.f) : y;
        }
        st8(out + row * ldo + col * 8, v);
        if (bits) {
            uint32_t m = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) m |= (uint32_t)((float)(bf16_t)v[e] > 0.f) << e;
            bits[row * cpr + col] = (uint8_t)m;
        }
    }
}

// backward: the three coefficients of the thread's 8 channels (bn_bwd_finalize_k's arithmetic)
__device__ __forceinline__ void fin_bwd(const mde_bn_bfin& f, int col, bool writer, float (&c0)[8], float (&c1)[8], float (&c2)[8]) {
    double s1[8], s2[8];
    fin_sum8(f.part, f.part_ld, 0, col, s1);
    fin_sum8(f.part, f.part_ld, 1, col, s2);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        c0[e] = f.gamma[col * 8 + e] * f.srstd[col * 8 + e];
        c1[e] = (float)(s1[e] / (double)f.count);
        c2[e] = (float)(s2[e] / (double)f.count);
        if (writer) {
            if (f.dgamma) f.dgamma[col * 8 + e] += (float)s2[e];
            if (f.dbeta) f.dbeta[col * 8 + e] += (float)s1[e];
        }
    }
}

template <int MASK>
__global__ __launch_bounds__(NT) void bn_bwd_apply_fin_k(const bf16_t* __restrict__ dout, int ldd,
                                                         const bf16_t* __restrict__ out, int ldo,
                                                         const bf16_t* __restrict__ x, int ldx,
                                                         const float* __restrict__ smean, const float* __restrict__ srstd,
                                                         const float* __restrict__ msc, const float* __restrict__ msh,
                                                         const uint8_t* __restrict__ bits, const mde_bn_bfin f,
                                                         int64_t M, int C, bf16_t* __restrict__ dx, int ldxo, int accumulate, bf16_t* __restrict__ dres,
                                                         int ldres) {
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float mu[8], rs[8], c0[8], c1[8], c2[8], ms[8], mh[8];
    extern __shared__ float s_fin[];          // [3][C]
    if (rl < rpb) {
        if (MASK == 2) {
            ldf8(msc + col * 8, ms);
            ldf8(msh + col * 8, mh);
        }
        ldf8(smean + col * 8, mu);
        ldf8(srstd + col * 8, rs);
    }
    if (rl == 0) {
        fin_bwd(f, col, blockIdx.x == 0, c0, c1, c2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s_fin[col * 8 + e] = c0[e];
            s_fin[C + col * 8 + e] = c1[e];
            s_fin[2 * C + col * 8 + e] = c2[e];
        }
    }
    fin_zero(f.zero, f.zero_n);
    __syncthreads();
    if (rl != 0 && rl < rpb) {
        ldf8(s_fin + col * 8, c0);
        ldf8(s_fin + C + col * 8, c1);
        ldf8(s_fin + 2 * C + col * 8, c2);
    }
#pragma unroll MDE_BN_UNROLL
    for (int64_t row_ = rl < rpb ? (int64_t)blockIdx.x * rpb + rl : M; row_ < M; row_ += (int64_t)gridDim.x * rpb) {
        const int64_t row = MDE_BN_SNAKE ? M - 1 - row_ : row_;
        float g[8], v[8], o[8], d[8];
        ld8(dout + row * ldd + col * 8, g);
        ld8(x + row * ldx + col * 8, v);
        if (MASK == 1) ld8(out + row * ldo + col * 8, o);
        if (accumulate) ld8(dx + row * ldxo + col * 8, d);
        uint32_t mb = 0;
        if (MASK == 3) mb = bits[row * cpr + col];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (MASK == 2) o[e] = v[e] * ms[e] + mh[e];
            if (MASK == 3) o[e] = (float)((mb >> e) & 1u);
            if (MASK && !(o[e] > 0.f)) g[e] = 0.f;
            const float xh = (v[e] - mu[e]) * rs[e];
            const float r = c0[e] * (g[e] - c1[e] - xh * c2[e]);
            d[e] = accumulate ? d[e] + r : r;
        }
        st8(dx + row * ldxo + col * 8, d);
        if (dres) st8(dres + row * ldres + col * 8, g);
    }
}

__global__ __launch_bounds__(NT) void bn_bwd_apply2_fin_k(const bf16_t* __restrict__ dout, int ldd,
                                                          const bf16_t* __restrict__ xa, int ldxa,
                                                          const bf16_t* __restrict__ xb, int ldxb,
                                                          const float* __restrict__ mean_a, const float* __restrict__ rstd_a,
                                                          const float* __restrict__ mean_b, const float* __restrict__ rstd_b,
                                                          const uint8_t* __restrict__ bits, const mde_bn_bfin fa, const mde_bn_bfin fb,
                                                          int64_t M, int C, bf16_t* __restrict__ dxa, int ldda, bf16_t* __restrict__ dxb, int lddb) {
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float mua[8], rsa[8], a0[8], a1[8], a2[8], mub[8], rsb[8], b0[8], b1[8], b2[8];
    extern __shared__ float s_fin[];          // [6][C]
    if (rl < rpb) {
        ldf8(mean_a + col * 8, mua);
        ldf8(rstd_a + col * 8, rsa);
        ldf8(mean_b + col * 8, mub);
        ldf8(rstd_b + col * 8, rsb);
    }
    if (rl == 0) {
        fin_bwd(fa, col, blockIdx.x == 0, a0, a1, a2);
        fin_bwd(fb, col, blockIdx.x == 0, b0, b1, b2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s_fin[col * 8 + e] = a0[e];
            s_fin[C + col * 8 + e] = a1[e];
            s_fin[2 * C + col * 8 + e] = a2[e];
            s_fin[3 * C + col * 8 + e] = b0[e];
            s_fin[4 * C + col * 8 + e] = b1[e];
            s_fin[5 * C + col * 8 + e] = b2[e];
        }
    }
    fin_zero(fa.zero, fa.zero_n);
    fin_zero(fb.zero, fb.zero_n);
    __syncthreads();
    if (rl != 0 && rl < rpb) {
        ldf8(s_fin + col * 8, a0);
        ldf8(s_fin + C + col * 8, a1);
        ldf8(s_fin + 2 * C + col * 8, a2);
        ldf8(s_fin + 3 * C + col * 8, b0);
        ldf8(s_fin + 4 * C + col * 8, b1);
        ldf8(s_fin + 5 * C + col * 8, b2);
    }
#pragma unroll MDE_BN_UNROLL
    for (int64_t row_ = rl < rpb ? (int64_t)blockIdx.x * rpb + rl : M; row_ < M; row_ += (int64_t)gridDim.x * rpb) {
        const int64_t row = MDE_BN_SNAKE ? M - 1 - row_ : row_;
        float g[8], va[8], vb[8], da[8], db[8];
        ld8(dout + row * ldd + col * 8, g);
        ld8(xa + row * ldxa + col * 8, va);
        ld8(xb + row * ldxb + col * 8, vb);
        const uint32_t mb = bits ? bits[row * cpr + col] : 0xFFu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ge = ((mb >> e) & 1u) ? g[e] : 0.f;
            da[e] = a0[e] * (ge - a1[e] - ((va[e] - mua[e]) * rsa[e]) * a2[e]);
            db[e] = b0[e] * (ge - b1[e] - ((vb[e] - mub[e]) * rsb[e]) * b2[e]);
        }
        st8(dxa + row * ldda + col * 8, da);
        st8(dxb + row * lddb + col * 8, db);
    }
}

int check_site(const char* who, int64_t M, int C) {
    MDE_REQUIRE(M > 0 && C > 0, "%s: non-positive size", who);
    // (C / 8 need not divide the 256 threads: the threads past the last whole row group idle -- DenseNet's 48-channel growth)
    MDE_REQUIRE(C % 8 == 0 && C <= MAXC, "%s: C=%d unsupported (need C %% 8 == 0, C <= %d)", who, C, MAXC);
    return MDE_OK;
}
bool al16(const void* p, int ld) { return ((uintptr_t)p % 16) == 0 && ld % 8 == 0; }

void reduce_geometry(int64_t M, int C, int* nblk, int* rows_per_blk) {
    // at least MDE_BN_RED_ROWS row groups per workgroup (amortises the LDS + global-atomic flush), at most 2048
    // workgroups.  In-network sweep (ms/step): 4 -> 33.26, 8 -> 32.95, 16 -> 32.75, 32 -> 32.74, 64 -> 33.15.
    const int rpb = NT / (C / 8);
    int64_t nb = (M + (int64_t)rpb * MDE_BN_RED_ROWS - 1) / ((int64_t)rpb * MDE_BN_RED_ROWS);
    nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
    int64_t rows = (M + nb - 1) / nb;
    rows = (rows + rpb - 1) / rpb * rpb;
    *nblk = (int)((M + rows - 1) / rows);
    *rows_per_blk = (int)rows;
}
int stream_grid(int64_t M, int C) {
    const int rpb = NT / (C / 8);
    int64_t nb = (M + rpb - 1) / rpb;
    return (int)(nb > 256 * 8 ? 256 * 8 : nb);
}

}  // namespace

extern "C" int mde_stat_slots(void) { return MDE_STAT_SLOTS; }

extern "C" int mde_bn_stats(const void* x, int64_t M, int C, int ld, float* part, void* stream) {
    MDE_REQUIRE(x && part, "mde_bn_stats: null argument");
    if (int rc = check_site("mde_bn_stats", M, C)) return rc;
    MDE_REQUIRE(al16(x, ld), "mde_bn_stats: x must be 16-byte aligned with ld %% 8 == 0");
    int nblk, rows;
    reduce_geometry(M, C, &nblk, &rows);
    bn_stats_k<<<nblk, NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, M, C, ld, part, rows, g_mde_det.on);
    MDE_LAUNCH_CHECK("bn_stats_k");
    return MDE_OK;
}

extern "C" int mde_bn_finalize(float* part, int64_t M, int C, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* save_mean, float* save_rstd, void* stream) {
    MDE_REQUIRE(part && gamma && beta && scale && shift && save_mean && save_rstd, "mde_bn_finalize: null argument");
    MDE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "mde_bn_finalize: running stats must come in pairs");
    MDE_REQUIRE(M > 0 && C > 0, "mde_bn_finalize: non-positive size");
    bn_finalize_k<<<mde_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(part, M, C, gamma, beta, running_mean, running_var,
                                                                 momentum, eps, scale, shift, save_mean, save_rstd, g_mde_det.on);
    MDE_LAUNCH_CHECK("bn_finalize_k");
    return MDE_OK;
}

extern "C" int mde_bn_moments(float* part, int64_t M, int C, float* mean, float* var, void* stream) {
    MDE_REQUIRE(part && mean && var && M > 0 && C > 0, "mde_bn_moments: bad argument");
    bn_moments_k<<<mde_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(part, M, C, mean, var, g_mde_det.on);
    MDE_LAUNCH_CHECK("bn_moments_k");
    return MDE_OK;
}

extern "C" int mde_bn_finalize_moments(const float* mean, const float* var, int64_t M, int C, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps, float* scale,
                                       float* shift, float* save_mean, float* save_rstd, void* stream) {
    MDE_REQUIRE(mean && var && gamma && beta && scale && shift && save_mean && save_rstd, "mde_bn_finalize_moments: null argument");
    MDE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "mde_bn_finalize_moments: running stats must come in pairs");
    MDE_REQUIRE(M > 0 && C > 0, "mde_bn_finalize_moments: non-positive size");
    bn_finalize_moments_k<<<mde_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(mean, var, M, C, gamma, beta, running_mean, running_var,
                                                                         momentum, eps, scale, shift, save_mean, save_rstd);
    MDE_LAUNCH_CHECK("bn_finalize_moments_k");
    return MDE_OK;
}

extern "C" int mde_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                                       const float* running_var, float eps, int C, float* scale, float* shift,
                                       void* stream) {
    MDE_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0,
                "mde_bn_eval_scale_shift: bad argument");
    bn_eval_k<<<mde_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var, eps, C, scale, shift);
    MDE_LAUNCH_CHECK("bn_eval_k");
    return MDE_OK;
}

extern "C" int mde_bn_apply(const void* x, int ldx, const float* scale, const float* shift, const void* r,
                            int ldr, const float* rscale, const float* rshift, void* out, int ldo,
                            uint8_t* relu_bits, int64_t M, int C, int relu, void* stream) {
    MDE_REQUIRE(x && scale && shift && out, "mde_bn_apply: null argument");
    if (int rc = check_site("mde_bn_apply", M, C)) return rc;
    MDE_REQUIRE(al16(x, ldx) && al16(out, ldo) && (!r || al16(r, ldr)), "mde_bn_apply: tensors must be 16-byte aligned, ld %% 8 == 0");
    MDE_REQUIRE((rscale == nullptr) == (rshift == nullptr) && (!rscale || r), "mde_bn_apply: rscale/rshift need r and each other");
    MDE_REQUIRE(relu >= 0 && relu <= 2 && !(relu == 2 && relu_bits), "mde_bn_apply: relu=%d (0 none, 1 ReLU behind, 2 ELU in front; no mask bits with 2)", relu);
    const int grid = stream_grid(M, C);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t *xp = (const bf16_t*)x, *rp = (const bf16_t*)r;
    if (!r)
        bn_apply_k<0><<<grid, NT, 0, st>>>(xp, ldx, scale, shift, nullptr, 0, nullptr, nullptr, (bf16_t*)out, ldo, relu_bits, M, C, relu);
    else if (!rscale)
        bn_apply_k<1><<<grid, NT, 0, st>>>(xp, ldx, scale, shift, rp, ldr, nullptr, nullptr, (bf16_t*)out, ldo, relu_bits, M, C, relu);
    else
        bn_apply_k<2><<<grid, NT, 0, st>>>(xp, ldx, scale, shift, rp, ldr, rscale, rshift, (bf16_t*)out, ldo, relu_bits, M, C, relu);
    MDE_LAUNCH_CHECK("bn_apply_k");
    return MDE_OK;
}

extern "C" int mde_bn_bwd_reduce(const void* dout, int ldd, const void* out, int ldo, const void* x, int ldx,
                                 const float* save_mean, const float* save_rstd, const float* mask_scale,
                                 const float* mask_shift, const uint8_t* relu_bits, int64_t M, int C, int relu,
                                 float* part, void* stream) {
    const bool recompute = relu && mask_scale && mask_shift;
    const bool packed = relu && !recompute && relu_bits;
    MDE_REQUIRE(dout && x && save_mean && save_rstd && part && (!relu || recompute || packed || out), "mde_bn_bwd_reduce: null argument");
    MDE_REQUIRE((mask_scale == nullptr) == (mask_shift == nullptr), "mde_bn_bwd_reduce: mask_scale/mask_shift come in pairs");
    if (int rc = check_site("mde_bn_bwd_reduce", M, C)) return rc;
    MDE_REQUIRE(al16(dout, ldd) && al16(x, ldx) && (!relu || recompute || packed || al16(out, ldo)), "mde_bn_bwd_reduce: alignment");
    int nblk, rows;
    reduce_geometry(M, C, &nblk, &rows);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t *d = (const bf16_t*)dout, *o = (const bf16_t*)out, *xp = (const bf16_t*)x;
    if (!relu)
        bn_bwd_reduce_k<0><<<nblk, NT, 0, st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, nullptr, nullptr, nullptr, M, C, part, rows, g_mde_det.on);
    else if (recompute)
        bn_bwd_reduce_k<2><<<nblk, NT, 0, st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, mask_scale, mask_shift, nullptr, M, C, part, rows, g_mde_det.on);
    else if (packed)
        bn_bwd_reduce_k<3><<<nblk, NT, 0, st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, nullptr, nullptr, relu_bits, M, C, part, rows, g_mde_det.on);
    else
        bn_bwd_reduce_k<1><<<nblk, NT, 0, st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, nullptr, nullptr, nullptr, M, C, part, rows, g_mde_det.on);
    MDE_LAUNCH_CHECK("bn_bwd_reduce_k");
    return MDE_OK;
}

extern "C" int mde_bn_bwd_reduce2(const void* dout, int ldd, const void* xa, int ldxa, const void* xb, int ldxb,
                                  const float* save_mean_a, const float* save_rstd_a, const float* save_mean_b,
                                  const float* save_rstd_b, const uint8_t* relu_bits, int64_t M, int C, float* part_a,
                                  float* part_b, void* stream) {
    MDE_REQUIRE(dout && xa && xb && save_mean_a && save_rstd_a && save_mean_b && save_rstd_b && part_a && part_b,
                "mde_bn_bwd_reduce2: null argument");
    if (int rc = check_site("mde_bn_bwd_reduce2", M, C)) return rc;
    MDE_REQUIRE(al16(dout, ldd) && al16(xa, ldxa) && al16(xb, ldxb), "mde_bn_bwd_reduce2: alignment");
    int nblk, rows;
    reduce_geometry(M, C, &nblk, &rows);
    bn_bwd_reduce2_k<<<nblk, NT, 0, (hipStream_t)stream>>>((const bf16_t*)dout, ldd, (const bf16_t*)xa, ldxa, (const bf16_t*)xb,
                                                           ldxb, save_mean_a, save_rstd_a, save_mean_b, save_rstd_b, relu_bits,
                                                           M, C, part_a, part_b, rows, g_mde_det.on);
    MDE_LAUNCH_CHECK("bn_bwd_reduce2_k");
    return MDE_OK;
}

extern "C" int mde_bn_bwd_apply2(const void* dout, int ldd, const void* xa, int ldxa, const void* xb, int ldxb,
                                 const float* save_mean_a, const float* save_rstd_a, const float* save_mean_b,
                                 const float* save_rstd_b, const uint8_t* relu_bits, const float* coef_a,
                                 const float* coef_b, int64_t M, int C, void* dxa, int ldda, void* dxb, int lddb,
                                 void* stream) {
    MDE_REQUIRE(dout && xa && xb && save_mean_a && save_rstd_a && save_mean_b && save_rstd_b && coef_a && coef_b && dxa && dxb,
                "mde_bn_bwd_apply2: null argument");
    if (int rc = check_site("mde_bn_bwd_apply2", M, C)) return rc;
    MDE_REQUIRE(al16(dout, ldd) && al16(xa, ldxa) && al16(xb, ldxb) && al16(dxa, ldda) && al16(dxb, lddb),
                "mde_bn_bwd_apply2: alignment");
    bn_bwd_apply2_k<<<stream_grid(M, C), NT, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dout, ldd, (const bf16_t*)xa, ldxa, (const bf16_t*)xb, ldxb, save_mean_a, save_rstd_a, save_mean_b,
        save_rstd_b, relu_bits, coef_a, coef_b, M, C, (bf16_t*)dxa, ldda, (bf16_t*)dxb, lddb);
    MDE_LAUNCH_CHECK("bn_bwd_apply2_k");
    return MDE_OK;
}

extern "C" int mde_bn_bwd_finalize(float* part, int64_t M, int C, const float* gamma, const float* save_rstd,
                                   float* dgamma, float* dbeta, float* coef, void* stream) {
    MDE_REQUIRE(part && gamma && save_rstd && coef && M > 0 && C > 0, "mde_bn_bwd_finalize: bad argument");
    bn_bwd_finalize_k<<<mde_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(part, M, C, gamma, save_rstd, dgamma, dbeta, coef, g_mde_det.on);
    MDE_LAUNCH_CHECK("bn_bwd_finalize_k");
    return MDE_OK;
}

extern "C" int mde_bn_bwd_apply(const void* dout, int ldd, const void* out, int ldo, const void* x, int ldx,
                                const float* save_mean, const float* save_rstd, const float* mask_scale,
                                const float* mask_shift, const uint8_t* relu_bits, const float* coef, int64_t M,
                                int C, int relu, void* dx, int ldxo, int accumulate_dx, void* dres, int ldres,
                                void* stream) {
    const bool recompute = relu && mask_scale && mask_shift;
    const bool packed = relu && !recompute && relu_bits;
    MDE_REQUIRE(dout && x && save_mean && save_rstd && coef && dx && (!relu || recompute || packed || out), "mde_bn_bwd_apply: null argument");
    MDE_REQUIRE((mask_scale == nullptr) == (mask_shift == nullptr), "mde_bn_bwd_apply: mask_scale/mask_shift come in pairs");
    if (int rc = check_site("mde_bn_bwd_apply", M, C)) return rc;
    MDE_REQUIRE(al16(dout, ldd) && al16(x, ldx) && al16(dx, ldxo) && (!relu || recompute || packed || al16(out, ldo)) &&
                    (!dres || al16(dres, ldres)), "mde_bn_bwd_apply: alignment");
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid(M, C);
    const bf16_t *d = (const bf16_t*)dout, *o = (const bf16_t*)out, *xp = (const bf16_t*)x;
    if (!relu)
        bn_bwd_apply_k<0><<<grid, NT, 0, st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, nullptr, nullptr, nullptr, coef, M, C,
                                               (bf16_t*)dx, ldxo, accumulate_dx, (bf16_t*)dres, ldres);
    else if (recompute)
        bn_bwd_apply_k<2><<<grid, NT, 0, st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, mask_scale, mask_shift, nullptr, coef,
                                               M, C, (bf16_t*)dx, ldxo, accumulate_dx, (bf16_t*)dres, ldres);
    else if (packed)
        bn_bwd_apply_k<3><<<grid, NT, 0, st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, nullptr, nullptr, relu_bits, coef, M,
                                               C, (bf16_t*)dx, ldxo, accumulate_dx, (bf16_t*)dres, ldres);
    else
        bn_bwd_apply_k<1><<<grid, NT, 0, st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, nullptr, nullptr, nullptr, coef, M, C,
                                               (bf16_t*)dx, ldxo, accumulate_dx, (bf16_t*)dres, ldres);
    MDE_LAUNCH_CHECK("bn_bwd_apply_k");
    return MDE_OK;
}

static int check_fin(const char* who, const mde_bn_fin* f, int C) {
    MDE_REQUIRE(f && f->gamma && f->beta && f->scale && f->shift && f->smean && f->srstd && f->count > 0, "%s: incomplete mde_bn_fin", who);
    MDE_REQUIRE((f->part != nullptr) != (f->mean_in != nullptr && f->var_in != nullptr), "%s: partial sums OR given moments", who);
    MDE_REQUIRE(!f->part || (f->part_ld >= C && f->part_ld % 4 == 0 && ((uintptr_t)f->part % 16) == 0), "%s: partial sums: ld >= C, 16-byte aligned", who);
    MDE_REQUIRE((f->rmean == nullptr) == (f->rvar == nullptr) && (f->zero == nullptr || f->zero_n > 0), "%s: running statistics come in pairs", who);
    MDE_REQUIRE(!g_mde_det.on, "%s: the fused finalize reads float partial sums; deterministic mode keeps integer ones (use mde_bn_finalize)", who);
    return MDE_OK;
}
static int check_bfin(const char* who, const mde_bn_bfin* f, int C) {
    MDE_REQUIRE(f && f->part && f->gamma && f->srstd && f->count > 0 && f->part_ld >= C && f->part_ld % 4 == 0 && ((uintptr_t)f->part % 16) == 0,
                "%s: incomplete mde_bn_bfin", who);
    MDE_REQUIRE(!g_mde_det.on, "%s: the fused finalize reads float partial sums; deterministic mode keeps integer ones (use mde_bn_bwd_finalize)", who);
    return MDE_OK;
}

extern "C" int mde_bn_apply_fin(const void* x, int ldx, const mde_bn_fin* fin, const void* r, int ldr, const mde_bn_fin* fin_r, void* out, int ldo,
                                uint8_t* relu_bits, int64_t M, int C, int relu, void* stream) {
    MDE_REQUIRE(x && out, "mde_bn_apply_fin: null argument");
    if (int rc = check_site("mde_bn_apply_fin", M, C)) return rc;
    if (int rc = check_fin("mde_bn_apply_fin", fin, C)) return rc;
    if (fin_r) {
        if (int rc = check_fin("mde_bn_apply_fin (residual site)", fin_r, C)) return rc;
    }
    MDE_REQUIRE(al16(x, ldx) && al16(out, ldo) && (!r || al16(r, ldr)) && (!fin_r || r), "mde_bn_apply_fin: tensors must be 16-byte aligned, ld %% 8 == 0");
    const int grid = stream_grid(M, C);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t *xp = (const bf16_t*)x, *rp = (const bf16_t*)r;
    const mde_bn_fin none = {};
    if (!r)
        bn_apply_fin_k<0><<<grid, NT, 4 * C * sizeof(float), st>>>(xp, ldx, *fin, nullptr, 0, none, (bf16_t*)out, ldo, relu_bits, M, C, relu);
    else if (!fin_r)
        bn_apply_fin_k<1><<<grid, NT, 4 * C * sizeof(float), st>>>(xp, ldx, *fin, rp, ldr, none, (bf16_t*)out, ldo, relu_bits, M, C, relu);
    else
        bn_apply_fin_k<2><<<grid, NT, 4 * C * sizeof(float), st>>>(xp, ldx, *fin, rp, ldr, *fin_r, (bf16_t*)out, ldo, relu_bits, M, C, relu);
    MDE_LAUNCH_CHECK("bn_apply_fin_k");
    return MDE_OK;
}

extern "C" int mde_bn_bwd_apply_fin(const void* dout, int ldd, const void* out, int ldo, const void* x, int ldx, const float* save_mean,
                                    const float* save_rstd, const float* mask_scale, const float* mask_shift, const uint8_t* relu_bits,
                                    const mde_bn_bfin* fin, int64_t M, int C, int relu, void* dx, int ldxo, int accumulate_dx, void* dres,
                                    int ldres, void* stream) {
    const bool recompute = relu && mask_scale && mask_shift;
    const bool packed = relu && !recompute && relu_bits;
    MDE_REQUIRE(dout && x && save_mean && save_rstd && dx && (!relu || recompute || packed || out), "mde_bn_bwd_apply_fin: null argument");
    MDE_REQUIRE((mask_scale == nullptr) == (mask_shift == nullptr), "mde_bn_bwd_apply_fin: mask_scale/mask_shift come in pairs");
    if (int rc = check_site("mde_bn_bwd_apply_fin", M, C)) return rc;
    if (int rc = check_bfin("mde_bn_bwd_apply_fin", fin, C)) return rc;
    MDE_REQUIRE(al16(dout, ldd) && al16(x, ldx) && al16(dx, ldxo) && (!relu || recompute || packed || al16(out, ldo)) &&
                    (!dres || al16(dres, ldres)), "mde_bn_bwd_apply_fin: alignment");
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid(M, C);
    const bf16_t *d = (const bf16_t*)dout, *o = (const bf16_t*)out, *xp = (const bf16_t*)x;
    if (!relu)
        bn_bwd_apply_fin_k<0><<<grid, NT, 3 * C * sizeof(float), st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, nullptr, nullptr, nullptr, *fin, M, C,
                                                   (bf16_t*)dx, ldxo, accumulate_dx, (bf16_t*)dres, ldres);
    else if (recompute)
        bn_bwd_apply_fin_k<2><<<grid, NT, 3 * C * sizeof(float), st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, mask_scale, mask_shift, nullptr, *fin,
                                                   M, C, (bf16_t*)dx, ldxo, accumulate_dx, (bf16_t*)dres, ldres);
    else if (packed)
        bn_bwd_apply_fin_k<3><<<grid, NT, 3 * C * sizeof(float), st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, nullptr, nullptr, relu_bits, *fin, M,
                                                   C, (bf16_t*)dx, ldxo, accumulate_dx, (bf16_t*)dres, ldres);
    else
        bn_bwd_apply_fin_k<1><<<grid, NT, 3 * C * sizeof(float), st>>>(d, ldd, o, ldo, xp, ldx, save_mean, save_rstd, nullptr, nullptr, nullptr, *fin, M, C,
                                                   (bf16_t*)dx, ldxo, accumulate_dx, (bf16_t*)dres, ldres);
    MDE_LAUNCH_CHECK("bn_bwd_apply_fin_k");
    return MDE_OK;
}

extern "C" int mde_bn_bwd_apply2_fin(const void* dout, int ldd, const void* xa, int ldxa, const void* xb, int ldxb, const float* save_mean_a,
                                     const float* save_rstd_a, const float* save_mean_b, const float* save_rstd_b, const uint8_t* relu_bits,
                                     const mde_bn_bfin* fin_a, const mde_bn_bfin* fin_b, int64_t M, int C, void* dxa, int ldda, void* dxb,
                                     int lddb, void* stream) {
    MDE_REQUIRE(dout && xa && xb && save_mean_a && save_rstd_a && save_mean_b && save_rstd_b && dxa && dxb, "mde_bn_bwd_apply2_fin: null argument");
    if (int rc = check_site("mde_bn_bwd_apply2_fin", M, C)) return rc;
    if (int rc = check_bfin("mde_bn_bwd_apply2_fin (a)", fin_a, C)) return rc;
    if (int rc = check_bfin("mde_bn_bwd_apply2_fin (b)", fin_b, C)) return rc;
    MDE_REQUIRE(al16(dout, ldd) && al16(xa, ldxa) && al16(xb, ldxb) && al16(dxa, ldda) && al16(dxb, lddb), "mde_bn_bwd_apply2_fin: alignment");
    bn_bwd_apply2_fin_k<<<stream_grid(M, C), NT, 6 * C * sizeof(float), (hipStream_t)stream>>>(
        (const bf16_t*)dout, ldd, (const bf16_t*)xa, ldxa, (const bf16_t*)xb, ldxb, save_mean_a, save_rstd_a, save_mean_b, save_rstd_b,
        relu_bits, *fin_a, *fin_b, M, C, (bf16_t*)dxa, ldda, (bf16_t*)dxb, lddb);
    MDE_LAUNCH_CHECK("bn_bwd_apply2_fin_k");
    return MDE_OK;
}
