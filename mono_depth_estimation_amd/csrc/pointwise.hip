// Pointwise / pooling / resize kernels of the VNL, MiDaS and BTS networks on NHWC bf16 activations (gfx950).
// Everything here is HBM-bound streaming work: 16-byte accesses, a thread owns 8 channels of one pixel, and a
// thread keeps a FIXED 8-channel column wherever a per-channel reduction rides along (bias gradients), so the
// partial sums live in registers until one LDS reduction + one atomic per column per workgroup.
//
// Reference call sites (file:line under /root/reference):
//   bias + activation + residual add   network/MiDaS.py:163-229 (ResidualConvUnit / FeatureFusionBlock, biased convs),
//                                      network/VNL.py:330-350 (FTB_block: `out += residual; relu`), network/Bts.py:69-80 (ELU)
//   global average pool / broadcast    network/VNL.py:207-225 (ASPP image pooling), :353-373 (AFA_block)
//   channel gate                       network/VNL.py:372  (`w * lateral + top`)
//   bilinear resize                    network/VNL.py:308,384,386 (align_corners=True), network/MiDaS.py:132-160,224-227
//   softmax head                       network/VNL.py:314-327 (fcn_topdown_predict: conv bias + nn.Softmax(dim=1))
//   NHWC -> NCHW + sigmoid             network/MiDaS.py:49-57, network/Bts.py:202-203
//   grouped-conv weight packing        network/VNL.py:638 (groups=cardinality), MiDaS' resnext101_32x8d trunk
#include "mde_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float act_fwd(float v, int act) {
    switch (act) {
        case 1: return fmaxf(v, 0.f);
        case 2: return v > 0.f ? v : expm1f(v);
        case 3: return 1.f / (1.f + expf(-v));
        case 4: return fminf(fmaxf(v, 0.f), 6.f);          // nn.ReLU6 (MobileNetV2: VNL.py:402,410)
        default: return v;
    }
}
// derivative of the activation expressed through its OUTPUT y (what the forward pass kept)
__device__ __forceinline__ float act_grad(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? 1.f : 0.f;
        case 2: return y > 0.f ? 1.f : y + 1.f;
        case 3: return y * (1.f - y);
        case 4: return (y > 0.f && y < 6.f) ? 1.f : 0.f;
        default: return 1.f;
    }
}

int grid_rows(int64_t rows, int rows_per_block) {
    int64_t nb = (rows + rows_per_block - 1) / rows_per_block;
    return (int)(nb > 256 * 8 ? 256 * 8 : (nb < 1 ? 1 : nb));
}
int grid_flat(int64_t total) {
    int64_t nb = (total + NT - 1) / NT;
    return (int)(nb > 256 * 16 ? 256 * 16 : (nb < 1 ? 1 : nb));
}

// ------------------------------------------------------------------ out = act(x + bias + r)
__global__ __launch_bounds__(NT) void pw_fwd_k(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ bias,
                                               const bf16_t* __restrict__ r, int ldr, bf16_t* __restrict__ out, int ldo,
                                               int64_t M, int C, int act) {
    const int cpr = C >> 3;
    const int tpr = cpr < NT ? cpr : NT;
    const int rpb = NT / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    if (tr >= rpb) return;
    for (int c8 = tc; c8 < cpr; c8 += tpr) {
        float b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = bias ? bias[c8 * 8 + e] : 0.f;
        for (int64_t row = (int64_t)blockIdx.x * rpb + tr; row < M; row += (int64_t)gridDim.x * rpb) {
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + row * ldx + c8 * 8);
            bf16x8_t rv;
            if (r) rv = *reinterpret_cast<const bf16x8_t*>(r + row * ldr + c8 * 8);
            bf16x8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)act_fwd((float)v[e] + b[e] + (r ? (float)rv[e] : 0.f), act);
            *reinterpret_cast<bf16x8_t*>(out + row * ldo + c8 * 8) = o;
        }
    }
}

// g = dout * act'(out);  dx (+)= g;  dr (+)= g;  dbias += sum_rows g
__global__ __launch_bounds__(NT) void pw_bwd_k(const bf16_t* __restrict__ dout, int ldd, const bf16_t* __restrict__ out, int ldo,
                                               bf16_t* __restrict__ dx, int lddx, int acc_x, bf16_t* __restrict__ dr, int lddr,
                                               int acc_r, float* __restrict__ dbias, float* __restrict__ part, int64_t M, int C, int act,
                                               MdeDetDev det) {
    __shared__ float red[NT * 8];
    const int cpr = C >> 3;
    const int tpr = cpr < NT ? cpr : NT;
    const int rpb = NT / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const bool live = tr < rpb;
    for (int c0 = 0; c0 < cpr; c0 += tpr) {          // (uniform trip count: the reduction below synchronises)
        const int c8 = c0 + tc;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live && c8 < cpr) {
            for (int64_t row = (int64_t)blockIdx.x * rpb + tr; row < M; row += (int64_t)gridDim.x * rpb) {
                const bf16x8_t g0 = *reinterpret_cast<const bf16x8_t*>(dout + row * ldd + c8 * 8);
                float g[8];
                if (act) {
                    const bf16x8_t y = *reinterpret_cast<const bf16x8_t*>(out + row * ldo + c8 * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = (float)g0[e] * act_grad((float)y[e], act);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = (float)g0[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += g[e];
                if (dx) {
                    bf16_t* p = dx + row * lddx + c8 * 8;
                    bf16x8_t o;
                    if (acc_x) {
                        const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)old[e] + g[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)g[e];
                    }
                    *reinterpret_cast<bf16x8_t*>(p) = o;
                }
                if (dr) {
                    bf16_t* p = dr + row * lddr + c8 * 8;
                    bf16x8_t o;
                    if (acc_r) {
                        const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)old[e] + g[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)g[e];
                    }
                    *reinterpret_cast<bf16x8_t*>(p) = o;
                }
            }
        }
        if (dbias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
            __syncthreads();
            if (tr == 0 && c8 < cpr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = 0.f;
                    for (int q = 0; q < rpb; ++q) t += red[(q * tpr + tc) * 8 + e];
                    if (part) mde_stat_add(part, C, blockIdx.x, 0, c8 * 8 + e, t, det.scratch != nullptr);
                    else mde_grad_add(dbias + c8 * 8 + e, t, det);
                }
            }
            __syncthreads();
        }
    }
}

// dbias[c] += the slots of a partial-sum buffer (left zeroed): thousands of workgroups adding into one cache line of
// dbias serialise in L2 (measured: pw_bwd_k 7x slower than pw_fwd_k on MiDaS' 256-channel maps); 32 slots spread them.
__global__ void bias_take_k(float* part, int C, float* dbias, MdeDetDev det) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) mde_grad_add(dbias + c, (float)mde_stat_take(part, C, 0, c, det.scratch != nullptr), det);
}

// ------------------------------------------------------------------ per-image spatial sums / broadcasts
// out[n][c] = scale * sum_p x[n][p][c]          grid (ceil(C/64), N), block = 8 chunk lanes x 32 row lanes
__global__ __launch_bounds__(NT) void spatial_sum_k(const bf16_t* __restrict__ x, int ldx, int64_t HW, int C, float scale,
                                                    bf16_t* __restrict__ out, int ldo) {
    __shared__ float red[32][8][8];
    const int cpr = C >> 3;
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c8 = blockIdx.x * 8 + cl;
    const int64_t n = blockIdx.y;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c8 < cpr) {
        const bf16_t* b = x + n * HW * ldx + c8 * 8;
        for (int64_t p = rl; p < HW; p += 32) {
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(b + p * ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cl][e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x >> 3, e = threadIdx.x & 7;
        if (blockIdx.x * 8 + c < cpr) {
            float t = 0.f;
            for (int q = 0; q < 32; ++q) t += red[q][c][e];
            out[n * ldo + (blockIdx.x * 8 + c) * 8 + e] = (bf16_t)(t * scale);
        }
    }
}

// out[n][p][c] (+)= scale * src[n][c]
__global__ __launch_bounds__(NT) void spatial_bcast_k(const bf16_t* __restrict__ src, int lds, float scale, bf16_t* __restrict__ out,
                                                      int ldo, int N, int64_t HW, int C, int acc) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * HW * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c8 = (int)(i % cpr);
        const int64_t row = i / cpr;
        const int64_t n = row / HW;
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(src + n * lds + c8 * 8);
        bf16_t* p = out + row * ldo + c8 * 8;
        bf16x8_t o;
        if (acc) {
            const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)old[e] + scale * (float)v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(scale * (float)v[e]);
        }
        *reinterpret_cast<bf16x8_t*>(p) = o;
    }
}

// ------------------------------------------------------------------ channel gate: out = w[n][c] * lat + top
__global__ __launch_bounds__(NT) void gate_fwd_k(const bf16_t* __restrict__ w, int ldw, const bf16_t* __restrict__ lat, int ldl,
                                                 const bf16_t* __restrict__ top, int ldt, bf16_t* __restrict__ out, int ldo, int N,
                                                 int64_t HW, int C) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * HW * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c8 = (int)(i % cpr);
        const int64_t row = i / cpr;
        const int64_t n = row / HW;
        const bf16x8_t wv = *reinterpret_cast<const bf16x8_t*>(w + n * ldw + c8 * 8);
        const bf16x8_t lv = *reinterpret_cast<const bf16x8_t*>(lat + row * ldl + c8 * 8);
        const bf16x8_t tv = *reinterpret_cast<const bf16x8_t*>(top + row * ldt + c8 * 8);
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)wv[e] * (float)lv[e] + (float)tv[e]);
        *reinterpret_cast<bf16x8_t*>(out + row * ldo + c8 * 8) = o;
    }
}

// dlat (+)= w * dout; dtop (+)= dout; dw[n][c] = sum_p dout * lat      grid (ceil(C/64), N), as spatial_sum_k
__global__ __launch_bounds__(NT) void gate_bwd_k(const bf16_t* __restrict__ dout, int ldd, const bf16_t* __restrict__ w, int ldw,
                                                 const bf16_t* __restrict__ lat, int ldl, bf16_t* __restrict__ dlat, int lddl,
                                                 int acc_lat, bf16_t* __restrict__ dtop, int lddt, int acc_top,
                                                 bf16_t* __restrict__ dw, int lddw, int64_t HW, int C) {
    __shared__ float red[32][8][8];
    const int cpr = C >> 3;
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c8 = blockIdx.x * 8 + cl;
    const int64_t n = blockIdx.y;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c8 < cpr) {
        const bf16x8_t wv = *reinterpret_cast<const bf16x8_t*>(w + n * ldw + c8 * 8);
        for (int64_t p = rl; p < HW; p += 32) {
            const int64_t row = n * HW + p;
            const bf16x8_t g = *reinterpret_cast<const bf16x8_t*>(dout + row * ldd + c8 * 8);
            const bf16x8_t lv = *reinterpret_cast<const bf16x8_t*>(lat + row * ldl + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)g[e] * (float)lv[e];
            bf16_t* pl = dlat + row * lddl + c8 * 8;
            bf16_t* pt = dtop + row * lddt + c8 * 8;
            bf16x8_t ol, ot;
            if (acc_lat) {
                const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(pl);
#pragma unroll
                for (int e = 0; e < 8; ++e) ol[e] = (bf16_t)((float)old[e] + (float)wv[e] * (float)g[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) ol[e] = (bf16_t)((float)wv[e] * (float)g[e]);
            }
            if (acc_top) {
                const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(pt);
#pragma unroll
                for (int e = 0; e < 8; ++e) ot[e] = (bf16_t)((float)old[e] + (float)g[e]);
            } else {
                ot = g;
            }
            *reinterpret_cast<bf16x8_t*>(pl) = ol;
            *reinterpret_cast<bf16x8_t*>(pt) = ot;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cl][e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x >> 3, e = threadIdx.x & 7;
        if (blockIdx.x * 8 + c < cpr) {
            float t = 0.f;
            for (int q = 0; q < 32; ++q) t += red[q][c][e];
            dw[n * lddw + (blockIdx.x * 8 + c) * 8 + e] = (bf16_t)t;
        }
    }
}

// ------------------------------------------------------------------ bilinear resize, NHWC bf16
// source coordinate exactly as ATen computes it in float:
//   align_corners: src = dst * (in-1)/(out-1);  otherwise: src = max(0, (dst + 0.5) * in/out - 0.5)
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_of(int dst, float scale, int in_size, int align) {
    float src = align ? scale * (float)dst : scale * ((float)dst + 0.5f) - 0.5f;
    if (!align && src < 0.f) src = 0.f;
    Lerp r;
    r.i0 = min((int)src, in_size - 1);
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
    r.l0 = 1.f - r.l1;
    return r;
}

__global__ __launch_bounds__(NT) void resize_fwd_k(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ out, int ldo, int N,
                                                   int H, int W, int C, int OH, int OW, float sh, float sw, int align) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * OH * OW * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c8 = (int)(i % cpr);
        int64_t p = i / cpr;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int64_t n = p / OH;
        const Lerp ly = lerp_of(oh, sh, H, align), lx = lerp_of(ow, sw, W, align);
        const bf16_t* b = x + n * H * W * ldx + c8 * 8;
        const bf16x8_t v00 = *reinterpret_cast<const bf16x8_t*>(b + ((int64_t)ly.i0 * W + lx.i0) * ldx);
        const bf16x8_t v01 = *reinterpret_cast<const bf16x8_t*>(b + ((int64_t)ly.i0 * W + lx.i1) * ldx);
        const bf16x8_t v10 = *reinterpret_cast<const bf16x8_t*>(b + ((int64_t)ly.i1 * W + lx.i0) * ldx);
        const bf16x8_t v11 = *reinterpret_cast<const bf16x8_t*>(b + ((int64_t)ly.i1 * W + lx.i1) * ldx);
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = (bf16_t)(ly.l0 * (lx.l0 * (float)v00[e] + lx.l1 * (float)v01[e]) + ly.l1 * (lx.l0 * (float)v10[e] + lx.l1 * (float)v11[e]));
        *reinterpret_cast<bf16x8_t*>(out + ((n * OH + oh) * (int64_t)OW + ow) * ldo + c8 * 8) = o;
    }
}

// gather form of the transposed interpolation (deterministic): a source pixel collects every destination pixel whose
// stencil touches it; the candidate window comes from the inverse scale, widened, and each candidate is tested exactly
__global__ __launch_bounds__(NT) void resize_bwd_k(const bf16_t* __restrict__ dout, int ldd, bf16_t* __restrict__ dx, int lddx, int N,
                                                   int H, int W, int C, int OH, int OW, float sh, float sw, float ish, float isw,
                                                   int align, int acc) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * H * W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c8 = (int)(i % cpr);
        int64_t p = i / cpr;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int64_t n = p / H;
        const int oh_lo = max(0, (int)floorf((float)(h - 1) * ish) - 2), oh_hi = min(OH - 1, (int)ceilf((float)(h + 1) * ish) + 2);
        const int ow_lo = max(0, (int)floorf((float)(w - 1) * isw) - 2), ow_hi = min(OW - 1, (int)ceilf((float)(w + 1) * isw) + 2);
        const bf16_t* go = dout + n * OH * OW * ldd + c8 * 8;
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const Lerp ly = lerp_of(oh, sh, H, align);
            float wy = 0.f;
            if (ly.i0 == h) wy += ly.l0;
            if (ly.i1 == h) wy += ly.l1;
            if (wy == 0.f) continue;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const Lerp lx = lerp_of(ow, sw, W, align);
                float wx = 0.f;
                if (lx.i0 == w) wx += lx.l0;
                if (lx.i1 == w) wx += lx.l1;
                if (wx == 0.f) continue;
                const bf16x8_t g = *reinterpret_cast<const bf16x8_t*>(go + ((int64_t)oh * OW + ow) * ldd);
                const float ww = wy * wx;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += ww * (float)g[e];
            }
        }
        bf16_t* po = dx + ((n * H + h) * (int64_t)W + w) * lddx + c8 * 8;
        bf16x8_t o;
        if (acc) {
            const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(po);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)old[e] + a[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)a[e];
        }
        *reinterpret_cast<bf16x8_t*>(po) = o;
    }
}

// ------------------------------------------------------------------ nearest x2 / 2x2 average pool (BTS: Bts.py:69-80, DenseNet transitions)
// up: out[n][2y+a][2x+b][c] = x[n][y][x][c]; its gradient: dx (+)= sum of the 4
__global__ __launch_bounds__(NT) void nearest2_fwd_k(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ out, int ldo, int N,
                                                     int H, int W, int C) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * H * W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c8 = (int)(i % cpr);
        int64_t p = i / cpr;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int64_t n = p / H;
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + ((n * H + h) * (int64_t)W + w) * ldx + c8 * 8);
        bf16_t* o = out + ((n * 2 * H + 2 * h) * (int64_t)(2 * W) + 2 * w) * ldo + c8 * 8;
        *reinterpret_cast<bf16x8_t*>(o) = v;
        *reinterpret_cast<bf16x8_t*>(o + ldo) = v;
        *reinterpret_cast<bf16x8_t*>(o + (int64_t)2 * W * ldo) = v;
        *reinterpret_cast<bf16x8_t*>(o + (int64_t)2 * W * ldo + ldo) = v;
    }
}
// dst[n][y][x][c] (+)= scale * sum_{a,b} src[n][2y+a][2x+b][c]    (nearest-x2 backward: scale 1; avg-pool forward: 0.25)
__global__ __launch_bounds__(NT) void sum2x2_k(const bf16_t* __restrict__ src, int lds, bf16_t* __restrict__ dst, int ldd, int N, int H,
                                               int W, int C, float scale, int acc) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * H * W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c8 = (int)(i % cpr);
        int64_t p = i / cpr;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int64_t n = p / H;
        const bf16_t* s = src + ((n * 2 * H + 2 * h) * (int64_t)(2 * W) + 2 * w) * lds + c8 * 8;
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(s), b = *reinterpret_cast<const bf16x8_t*>(s + lds);
        const bf16x8_t c = *reinterpret_cast<const bf16x8_t*>(s + (int64_t)2 * W * lds);
        const bf16x8_t d = *reinterpret_cast<const bf16x8_t*>(s + (int64_t)2 * W * lds + lds);
        bf16_t* po = dst + ((n * H + h) * (int64_t)W + w) * ldd + c8 * 8;
        bf16x8_t o;
        if (acc) {
            const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(po);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)old[e] + scale * ((float)a[e] + (float)b[e] + (float)c[e] + (float)d[e]));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(scale * ((float)a[e] + (float)b[e] + (float)c[e] + (float)d[e]));
        }
        *reinterpret_cast<bf16x8_t*>(po) = o;
    }
}
// out[n][2y+a][2x+b][c] (+)= scale * x[n][y][x][c]     (avg-pool backward: scale 0.25)
__global__ __launch_bounds__(NT) void spread2x2_k(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ out, int ldo, int N, int H,
                                                  int W, int C, float scale, int acc) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * H * W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c8 = (int)(i % cpr);
        int64_t p = i / cpr;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int64_t n = p / H;
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + ((n * H + h) * (int64_t)W + w) * ldx + c8 * 8);
        bf16_t* o = out + ((n * 2 * H + 2 * h) * (int64_t)(2 * W) + 2 * w) * ldo + c8 * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16_t* po = o + (int64_t)(q >> 1) * 2 * W * ldo + (q & 1) * ldo;
            bf16x8_t r;
            if (acc) {
                const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(po);
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = (bf16_t)((float)old[e] + scale * (float)v[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = (bf16_t)(scale * (float)v[e]);
            }
            *reinterpret_cast<bf16x8_t*>(po) = r;
        }
    }
}

// ------------------------------------------------------------------ softmax head (VNL.py:314-327)
// x: bf16 [N*HW][ldx] conv output (no bias yet);  logit = x + bias, prob = softmax_c(logit), both fp32 NCHW.
// One 64-pixel tile per iteration: the rows are read coalesced (full pixel rows), transposed through LDS, and each
// channel plane is written with 256-byte wave stores.  Wave q handles channels c = q mod 4.
constexpr int SM_PIX = 64;
__global__ __launch_bounds__(NT) void softmax_head_fwd_k(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ bias,
                                                         float* __restrict__ logit, float* __restrict__ prob, int N, int64_t HW, int C) {
    extern __shared__ float sm[];                 // [C][65] tile + [4][64] reductions
    float* tile = sm;
    float* red = sm + (size_t)C * 65;
    const int cpr = (C + 7) >> 3;
    const int64_t tiles_per_img = (HW + SM_PIX - 1) / SM_PIX;
    const int64_t ntiles = (int64_t)N * tiles_per_img;
    const int p = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t n = t / tiles_per_img;
        const int64_t hw0 = (t - n * tiles_per_img) * SM_PIX;
        const int npix = (int)min((int64_t)SM_PIX, HW - hw0);
        for (int i = threadIdx.x; i < SM_PIX * cpr; i += NT) {
            const int pr = i / cpr, c8 = i - pr * cpr;
            bf16x8_t v;
            if (pr < npix) v = *reinterpret_cast<const bf16x8_t*>(x + (n * HW + hw0 + pr) * ldx + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c8 * 8 + e;
                if (c < C) tile[c * 65 + pr] = pr < npix ? (float)v[e] + (bias ? bias[c] : 0.f) : 0.f;
            }
        }
        __syncthreads();
        float mx = -INFINITY;
        for (int c = q; c < C; c += 4) mx = fmaxf(mx, tile[c * 65 + p]);
        red[q * 64 + p] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[p], red[64 + p]), fmaxf(red[128 + p], red[192 + p]));
        __syncthreads();
        float s = 0.f;
        for (int c = q; c < C; c += 4) s += expf(tile[c * 65 + p] - mx);
        red[q * 64 + p] = s;
        __syncthreads();
        const float inv = 1.f / (red[p] + red[64 + p] + red[128 + p] + red[192 + p]);
        if (p < npix) {
            float* lo = logit + n * C * HW + hw0 + p;
            float* po = prob + n * C * HW + hw0 + p;
            for (int c = q; c < C; c += 4) {
                const float v = tile[c * 65 + p];
                lo[(int64_t)c * HW] = v;
                po[(int64_t)c * HW] = expf(v - mx) * inv;
            }
        }
        __syncthreads();
    }
}

// dx[p][c] = dlogit[c][p] + prob[c][p] * (dprob[c][p] - sum_c' dprob[c'][p] * prob[c'][p]);  dbias[c] += sum_p dx[p][c]
// A thread (pixel p, wave slot q) owns channels q, q + 4, ...: their prob / dprob values are loaded ONCE into registers by an
// unrolled loop (KMAX loads in flight per operand; the first version walked them in a run-time loop, one dependent load pair
// at a time, and was latency-bound at 0.7 TB/s), used for the dot product and again for the gradient.  The bias gradient is
// summed from the finished LDS tile by one thread per channel (one register) and added once per workgroup at the end.
constexpr int SM_MAXC = 64;
template <int KMAX>
__global__ __launch_bounds__(NT) void softmax_head_bwd_k(const float* __restrict__ dlogit, const float* __restrict__ dprob,
                                                         const float* __restrict__ prob, bf16_t* __restrict__ dx, int lddx,
                                                         float* __restrict__ dbias, int N, int64_t HW, int C, MdeDetDev det) {
    extern __shared__ float sm[];
    float* tile = sm;
    float* red = sm + (size_t)C * 65;
    const int cpr = (C + 7) >> 3;
    const int64_t tiles_per_img = (HW + SM_PIX - 1) / SM_PIX;
    const int64_t ntiles = (int64_t)N * tiles_per_img;
    const int p = threadIdx.x & 63, q = threadIdx.x >> 6;
    float bacc = 0.f;                              // bias gradient of channel threadIdx.x
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t n = t / tiles_per_img;
        const int64_t hw0 = (t - n * tiles_per_img) * SM_PIX;
        const int npix = (int)min((int64_t)SM_PIX, HW - hw0);
        const bool in = p < npix;
        const int64_t base = n * C * HW + hw0 + p;
        float P[KMAX], D[KMAX];
        float dot = 0.f;
        if (dprob) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int c = q + 4 * k;
                const bool ok = in && c < C;
                P[k] = ok ? prob[base + (int64_t)c * HW] : 0.f;
                D[k] = ok ? dprob[base + (int64_t)c * HW] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k) dot += P[k] * D[k];
        }
        red[q * 64 + p] = dot;
        __syncthreads();
        dot = red[p] + red[64 + p] + red[128 + p] + red[192 + p];
        float L[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = q + 4 * k;
            L[k] = (dlogit && in && c < C) ? dlogit[base + (int64_t)c * HW] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = q + 4 * k;
            if (c < C) tile[c * 65 + p] = L[k] + (dprob ? P[k] * (D[k] - dot) : 0.f);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < SM_PIX * cpr; i += NT) {
            const int pr = i / cpr, c8 = i - pr * cpr;
            if (pr < npix) {
                bf16x8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c8 * 8 + e;
                    o[e] = (bf16_t)(c < C ? tile[c * 65 + pr] : 0.f);
                }
                *reinterpret_cast<bf16x8_t*>(dx + (n * HW + hw0 + pr) * lddx + c8 * 8) = o;
            }
        }
        if (dbias && (int)threadIdx.x < C) {       // (pixels beyond npix hold zeros: written as 0 above for !in)
            const float* row = tile + threadIdx.x * 65;
            float sacc = 0.f;
#pragma unroll 8
            for (int j = 0; j < SM_PIX; ++j) sacc += row[j];
            bacc += sacc;
        }
        __syncthreads();
    }
    if (dbias && (int)threadIdx.x < C) mde_grad_add(dbias + threadIdx.x, bacc, det);
}

// y = scale act(p) on an fp32 map, and its backward through the kept output: the activation behind a one-channel head conv that
// already produced fp32 (mde_head_conv_fwd), e.g. BTS' get_depth + Sigmoid x max_depth (Bts.py:168,262).
__global__ __launch_bounds__(NT) void map_act_fwd_k(const float* __restrict__ p, float* __restrict__ y, int64_t n4, int act, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        f32x4_t v = reinterpret_cast<const f32x4_t*>(p)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = scale * act_fwd(v[e], act);
        reinterpret_cast<f32x4_t*>(y)[i] = v;
    }
}
__global__ __launch_bounds__(NT) void map_act_bwd_k(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dp,
                                                    int64_t n4, int act, float scale) {
    const float inv = 1.f / scale;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        const f32x4_t g = reinterpret_cast<const f32x4_t*>(dy)[i], o = reinterpret_cast<const f32x4_t*>(y)[i];
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = g[e] * scale * act_grad(o[e] * inv, act);
        reinterpret_cast<f32x4_t*>(dp)[i] = v;
    }
}

// ------------------------------------------------------------------ out[n][c][p] = scale * act(x[n][p][c] + bias[c])   (small C heads)
__global__ __launch_bounds__(NT) void to_nchw_act_fwd_k(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ bias,
                                                        float* __restrict__ out, int N, int64_t HW, int C, int act, float scale) {
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW, p = i - n * HW;
        for (int c0 = 0; c0 < C; c0 += 8) {
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + i * ldx + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c0 + e < C) out[(n * C + c0 + e) * HW + p] = scale * act_fwd((float)v[e] + (bias ? bias[c0 + e] : 0.f), act);
        }
    }
}
// dx[n][p][c] = dout * scale * act'(out / scale);  dbias[c] += sum dx   (C <= 64 for the bias gradient registers)
__global__ __launch_bounds__(NT) void to_nchw_act_bwd_k(const float* __restrict__ dout, const float* __restrict__ out,
                                                        bf16_t* __restrict__ dx, int lddx, float* __restrict__ dbias, int N, int64_t HW,
                                                        int C, int act, float scale, MdeDetDev det) {
    __shared__ float red[NT / 64][64];
    const int64_t total = (int64_t)N * HW;
    float bs[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) bs[k] = 0.f;
    const float inv = 1.f / scale;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW, p = i - n * HW;
#pragma unroll
        for (int c0 = 0; c0 < 64; c0 += 8) {
            if (c0 < C) {
                bf16x8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float g = 0.f;
                    if (c0 + e < C) {
                        const int64_t a = (n * C + c0 + e) * HW + p;
                        g = dout[a] * scale * act_grad(out[a] * inv, act);
                    }
                    bs[c0 + e] += g;
                    o[e] = (bf16_t)g;
                }
                *reinterpret_cast<bf16x8_t*>(dx + i * lddx + c0) = o;
            }
        }
    }
    if (dbias) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (k < C) {
                const float t = mde_wave_sum(bs[k]);
                if (lane == 0) red[wv][k] = t;
            }
        }
        __syncthreads();
        if (threadIdx.x < C) mde_grad_add(dbias + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x], det);
    }
}

// ------------------------------------------------------------------ grouped-conv weight packing
// src fp32 [O][T][G] (G input channels per group, O == I);  fwd[o][t][j] = src[o][t][64b + j - g(o) G] if input channel
// 64b + j (b = o / 64) lies in o's group, else 0;  dgrad[i][t][j] = src[64b + j][t][i - g(i) G] if output channel 64b + j lies
// in i's group, else 0 (the transposed blocks).
__global__ __launch_bounds__(NT) void pack_grouped_k(const float* __restrict__ src, bf16_t* __restrict__ fwd, bf16_t* __restrict__ dgrad,
                                                     int O, int T, int G) {
    const int64_t total = (int64_t)O * T * 64;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int j = (int)(i & 63);
        const int64_t ot = i >> 6;
        const int t = (int)(ot % T);
        const int o = (int)(ot / T);
        const int other = (o & ~63) + j;              // the channel on the other side of the 64x64 block
        const bool same = (other / G) == (o / G);
        const int g0 = (o / G) * G;
        if (fwd) fwd[i] = (bf16_t)(same ? src[((int64_t)o * T + t) * G + (other - g0)] : 0.f);
        if (dgrad) dgrad[i] = (bf16_t)(same ? src[((int64_t)other * T + t) * G + (o - g0)] : 0.f);
    }
}

// The eval-mode two-term shadow of a grouped weight (mde_pack_split_batch's counterpart): fwd2 [O][2T][64], the block-diagonal
// packing of hi = (bf16)w in taps [0, T) and of lo = (bf16)(w - hi) in taps [T, 2T).
__global__ __launch_bounds__(NT) void pack_grouped_split_k(const float* __restrict__ src, bf16_t* __restrict__ fwd2, int O, int T, int G) {
    const int64_t total = (int64_t)O * T * 64;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int j = (int)(i & 63);
        const int64_t ot = i >> 6;
        const int t = (int)(ot % T);
        const int o = (int)(ot / T);
        const int other = (o & ~63) + j;
        const bool same = (other / G) == (o / G);
        const int g0 = (o / G) * G;
        const float v = same ? src[((int64_t)o * T + t) * G + (other - g0)] : 0.f;
        const bf16_t h = (bf16_t)v;
        fwd2[((int64_t)o * 2 * T + t) * 64 + j] = h;
        fwd2[((int64_t)o * 2 * T + T + t) * 64 + j] = (bf16_t)(v - (float)h);
    }
}

}  // namespace

#define PW_ALIGNED(p) (((uintptr_t)(p) % 16) == 0)

extern "C" int mde_pw_fwd(const void* x, int ldx, const float* bias, const void* r, int ldr, void* out, int ldo, int64_t M,
                          int C, int act, void* stream) {
    MDE_REQUIRE(x && out && M > 0 && C > 0 && C % 8 == 0 && act >= 0 && act <= 4, "mde_pw_fwd: bad argument (C=%d, act=%d)", C, act);
    MDE_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && (!r || ldr % 8 == 0) && PW_ALIGNED(x) && PW_ALIGNED(out) && (!r || PW_ALIGNED(r)),
                "mde_pw_fwd: operands must be 16-byte aligned with ld %% 8 == 0");
    const int tpr = C / 8 < NT ? C / 8 : NT;
    pw_fwd_k<<<grid_rows(M, NT / tpr), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, bias, (const bf16_t*)r, ldr, (bf16_t*)out, ldo, M, C, act);
    MDE_LAUNCH_CHECK("pw_fwd_k");
    return MDE_OK;
}

extern "C" int mde_pw_bwd(const void* dout, int ldd, const void* out, int ldo, void* dx, int lddx, int acc_x, void* dr, int lddr,
                          int acc_r, float* dbias, float* bias_part, int64_t M, int C, int act, void* stream) {
    MDE_REQUIRE(dout && (out || act == 0) && (dx || dr || dbias) && M > 0 && C > 0 && C % 8 == 0 && act >= 0 && act <= 4,
                "mde_pw_bwd: bad argument (C=%d, act=%d)", C, act);
    MDE_REQUIRE(ldd % 8 == 0 && (!out || ldo % 8 == 0) && (!dx || lddx % 8 == 0) && (!dr || lddr % 8 == 0) && PW_ALIGNED(dout) &&
                    (!out || PW_ALIGNED(out)) && (!dx || PW_ALIGNED(dx)) && (!dr || PW_ALIGNED(dr)),
                "mde_pw_bwd: operands must be 16-byte aligned with ld %% 8 == 0");
    MDE_DET_REQUIRE("mde_pw_bwd", dbias, (int64_t)C);
    const int tpr = C / 8 < NT ? C / 8 : NT;
    pw_bwd_k<<<grid_rows(M, NT / tpr), NT, 0, (hipStream_t)stream>>>((const bf16_t*)dout, ldd, (const bf16_t*)out, ldo, (bf16_t*)dx, lddx,
                                                                   acc_x, (bf16_t*)dr, lddr, acc_r, dbias, dbias ? bias_part : nullptr, M, C, act,
                                                                   mde_det_dev());
    MDE_LAUNCH_CHECK("pw_bwd_k");
    if (dbias && bias_part) {
        bias_take_k<<<mde_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(bias_part, C, dbias, mde_det_dev());
        MDE_LAUNCH_CHECK("bias_take_k");
    }
    return MDE_OK;
}

extern "C" int mde_spatial_sum(const void* x, int ldx, int N, int64_t HW, int C, float scale, void* out, int ldo, void* stream) {
    MDE_REQUIRE(x && out && N > 0 && HW > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldo >= C && PW_ALIGNED(x),
                "mde_spatial_sum: bad argument (C=%d, ldx=%d, ldo=%d)", C, ldx, ldo);
    spatial_sum_k<<<dim3(mde_cdiv(C / 8, 8), N), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, HW, C, scale, (bf16_t*)out, ldo);
    MDE_LAUNCH_CHECK("spatial_sum_k");
    return MDE_OK;
}

extern "C" int mde_spatial_bcast(const void* src, int lds, float scale, void* out, int ldo, int N, int64_t HW, int C, int accumulate,
                                 void* stream) {
    MDE_REQUIRE(src && out && N > 0 && HW > 0 && C > 0 && C % 8 == 0 && lds % 8 == 0 && ldo % 8 == 0 && PW_ALIGNED(src) && PW_ALIGNED(out),
                "mde_spatial_bcast: bad argument (C=%d, lds=%d, ldo=%d)", C, lds, ldo);
    spatial_bcast_k<<<grid_flat((int64_t)N * HW * (C / 8)), NT, 0, (hipStream_t)stream>>>((const bf16_t*)src, lds, scale, (bf16_t*)out, ldo,
                                                                                        N, HW, C, accumulate);
    MDE_LAUNCH_CHECK("spatial_bcast_k");
    return MDE_OK;
}

extern "C" int mde_gate_fwd(const void* w, int ldw, const void* lat, int ldl, const void* top, int ldt, void* out, int ldo, int N,
                            int64_t HW, int C, void* stream) {
    MDE_REQUIRE(w && lat && top && out && N > 0 && HW > 0 && C > 0 && C % 8 == 0 && ldw % 8 == 0 && ldl % 8 == 0 && ldt % 8 == 0 &&
                    ldo % 8 == 0 && PW_ALIGNED(w) && PW_ALIGNED(lat) && PW_ALIGNED(top) && PW_ALIGNED(out),
                "mde_gate_fwd: bad argument (C=%d)", C);
    gate_fwd_k<<<grid_flat((int64_t)N * HW * (C / 8)), NT, 0, (hipStream_t)stream>>>((const bf16_t*)w, ldw, (const bf16_t*)lat, ldl,
                                                                                   (const bf16_t*)top, ldt, (bf16_t*)out, ldo, N, HW, C);
    MDE_LAUNCH_CHECK("gate_fwd_k");
    return MDE_OK;
}

extern "C" int mde_gate_bwd(const void* dout, int ldd, const void* w, int ldw, const void* lat, int ldl, void* dlat, int lddl,
                            int acc_lat, void* dtop, int lddt, int acc_top, void* dw, int lddw, int N, int64_t HW, int C,
                            void* stream) {
    MDE_REQUIRE(dout && w && lat && dlat && dtop && dw && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "mde_gate_bwd: bad argument (C=%d)", C);
    MDE_REQUIRE(ldd % 8 == 0 && ldw % 8 == 0 && ldl % 8 == 0 && lddl % 8 == 0 && lddt % 8 == 0 && lddw >= C && PW_ALIGNED(dout) &&
                    PW_ALIGNED(w) && PW_ALIGNED(lat) && PW_ALIGNED(dlat) && PW_ALIGNED(dtop),
                "mde_gate_bwd: operands must be 16-byte aligned with ld %% 8 == 0");
    gate_bwd_k<<<dim3(mde_cdiv(C / 8, 8), N), NT, 0, (hipStream_t)stream>>>((const bf16_t*)dout, ldd, (const bf16_t*)w, ldw,
                                                                          (const bf16_t*)lat, ldl, (bf16_t*)dlat, lddl, acc_lat,
                                                                          (bf16_t*)dtop, lddt, acc_top, (bf16_t*)dw, lddw, HW, C);
    MDE_LAUNCH_CHECK("gate_bwd_k");
    return MDE_OK;
}

static inline float rs_scale(int in, int out, int align) {
    if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

extern "C" int mde_resize_bilinear_fwd(const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, int OH, int OW,
                                       int align_corners, void* stream) {
    MDE_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 &&
                    PW_ALIGNED(x) && PW_ALIGNED(out),
                "mde_resize_bilinear_fwd: bad argument (C=%d, ldx=%d, ldo=%d)", C, ldx, ldo);
    resize_fwd_k<<<grid_flat((int64_t)N * OH * OW * (C / 8)), NT, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, ldx, (bf16_t*)out, ldo, N, H, W, C, OH, OW, rs_scale(H, OH, align_corners), rs_scale(W, OW, align_corners),
        align_corners);
    MDE_LAUNCH_CHECK("resize_fwd_k");
    return MDE_OK;
}

extern "C" int mde_resize_bilinear_bwd(const void* dout, int ldd, void* dx, int lddx, int N, int H, int W, int C, int OH, int OW,
                                       int align_corners, int accumulate, void* stream) {
    MDE_REQUIRE(dout && dx && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 8 == 0 && ldd % 8 == 0 && lddx % 8 == 0 &&
                    PW_ALIGNED(dout) && PW_ALIGNED(dx),
                "mde_resize_bilinear_bwd: bad argument (C=%d, ldd=%d, lddx=%d)", C, ldd, lddx);
    const float sh = rs_scale(H, OH, align_corners), sw = rs_scale(W, OW, align_corners);
    const float ish = sh > 0.f ? 1.f / sh : (float)OH, isw = sw > 0.f ? 1.f / sw : (float)OW;
    resize_bwd_k<<<grid_flat((int64_t)N * H * W * (C / 8)), NT, 0, (hipStream_t)stream>>>((const bf16_t*)dout, ldd, (bf16_t*)dx, lddx, N, H, W,
                                                                                        C, OH, OW, sh, sw, ish, isw, align_corners, accumulate);
    MDE_LAUNCH_CHECK("resize_bwd_k");
    return MDE_OK;
}

#define PW_2X2_CHECK(name, a, lda, b, ldb)                                                                                   \
    MDE_REQUIRE(a && b && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && PW_ALIGNED(a) && \
                    PW_ALIGNED(b),                                                                                           \
                name ": bad argument (C=%d)", C)

extern "C" int mde_nearest2_fwd(const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, void* stream) {
    PW_2X2_CHECK("mde_nearest2_fwd", x, ldx, out, ldo);
    nearest2_fwd_k<<<grid_flat((int64_t)N * H * W * (C / 8)), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (bf16_t*)out, ldo, N, H, W, C);
    MDE_LAUNCH_CHECK("nearest2_fwd_k");
    return MDE_OK;
}

extern "C" int mde_sum2x2(const void* src, int lds, void* dst, int ldd, int N, int H, int W, int C, float scale, int accumulate,
                          void* stream) {
    PW_2X2_CHECK("mde_sum2x2", src, lds, dst, ldd);
    sum2x2_k<<<grid_flat((int64_t)N * H * W * (C / 8)), NT, 0, (hipStream_t)stream>>>((const bf16_t*)src, lds, (bf16_t*)dst, ldd, N, H, W, C,
                                                                                    scale, accumulate);
    MDE_LAUNCH_CHECK("sum2x2_k");
    return MDE_OK;
}

extern "C" int mde_spread2x2(const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, float scale, int accumulate,
                             void* stream) {
    PW_2X2_CHECK("mde_spread2x2", x, ldx, out, ldo);
    spread2x2_k<<<grid_flat((int64_t)N * H * W * (C / 8)), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (bf16_t*)out, ldo, N, H, W, C,
                                                                                       scale, accumulate);
    MDE_LAUNCH_CHECK("spread2x2_k");
    return MDE_OK;
}

extern "C" int mde_softmax_head_fwd(const void* x, int ldx, const float* bias, float* logit, float* prob, int N, int64_t HW, int C,
                                    void* stream) {
    MDE_REQUIRE(x && logit && prob && N > 0 && HW > 0 && C > 0 && C <= 4 * SM_MAXC && ldx % 8 == 0 && ldx >= (C + 7) / 8 * 8 && PW_ALIGNED(x),
                "mde_softmax_head_fwd: bad argument (C=%d <= %d, ldx=%d >= C rounded up to 8)", C, 4 * SM_MAXC, ldx);
    const size_t smem = ((size_t)C * 65 + 4 * 64) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        int rc = mde_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&softmax_head_fwd_k),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (4 * SM_MAXC * 65 + 256) * 4),
                               "hipFuncSetAttribute(softmax_head_fwd_k)");
        if (rc) return rc;
        attr = true;
    }
    const int64_t ntiles = (int64_t)N * ((HW + SM_PIX - 1) / SM_PIX);
    softmax_head_fwd_k<<<(int)(ntiles < 2048 ? ntiles : 2048), NT, smem, (hipStream_t)stream>>>((const bf16_t*)x, ldx, bias, logit, prob, N, HW, C);
    MDE_LAUNCH_CHECK("softmax_head_fwd_k");
    return MDE_OK;
}

extern "C" int mde_softmax_head_bwd(const float* dlogit, const float* dprob, const float* prob, void* dx, int lddx, float* dbias,
                                    int N, int64_t HW, int C, void* stream) {
    MDE_REQUIRE((dlogit || dprob) && (prob || !dprob) && dx && N > 0 && HW > 0 && C > 0 && C <= 4 * SM_MAXC && lddx % 8 == 0 &&
                    lddx >= (C + 7) / 8 * 8 && PW_ALIGNED(dx),
                "mde_softmax_head_bwd: bad argument (C=%d <= %d, lddx=%d)", C, 4 * SM_MAXC, lddx);
    MDE_DET_REQUIRE("mde_softmax_head_bwd", dbias, (int64_t)C);
    const size_t smem = ((size_t)C * 65 + 4 * 64) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        for (const void* f : {reinterpret_cast<const void*>(&softmax_head_bwd_k<40>), reinterpret_cast<const void*>(&softmax_head_bwd_k<SM_MAXC>)}) {
            int rc = mde_check_hip(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (4 * SM_MAXC * 65 + 256) * 4),
                                   "hipFuncSetAttribute(softmax_head_bwd_k)");
            if (rc) return rc;
        }
        attr = true;
    }
    const int64_t ntiles = (int64_t)N * ((HW + SM_PIX - 1) / SM_PIX);
    const int grid = (int)(ntiles < 2048 ? ntiles : 2048);
    if (C <= 160)
        softmax_head_bwd_k<40><<<grid, NT, smem, (hipStream_t)stream>>>(dlogit, dprob, prob, (bf16_t*)dx, lddx, dbias, N, HW, C, mde_det_dev());
    else
        softmax_head_bwd_k<SM_MAXC><<<grid, NT, smem, (hipStream_t)stream>>>(dlogit, dprob, prob, (bf16_t*)dx, lddx, dbias, N, HW, C, mde_det_dev());
    MDE_LAUNCH_CHECK("softmax_head_bwd_k");
    return MDE_OK;
}

extern "C" int mde_map_act_fwd(const float* p, float* y, int64_t n, int act, float scale, void* stream) {
    MDE_REQUIRE(p && y && n > 0 && n % 4 == 0 && act >= 0 && act <= 4 && ((uintptr_t)p % 16) == 0 && ((uintptr_t)y % 16) == 0,
                "mde_map_act_fwd: bad argument (n=%lld must be a multiple of 4, 16-byte aligned maps)", (long long)n);
    map_act_fwd_k<<<grid_flat(n / 4), NT, 0, (hipStream_t)stream>>>(p, y, n / 4, act, scale);
    MDE_LAUNCH_CHECK("map_act_fwd_k");
    return MDE_OK;
}

extern "C" int mde_map_act_bwd(const float* dy, const float* y, float* dp, int64_t n, int act, float scale, void* stream) {
    MDE_REQUIRE(dy && y && dp && n > 0 && n % 4 == 0 && act >= 0 && act <= 4 && scale != 0.f && ((uintptr_t)dy % 16) == 0 &&
                    ((uintptr_t)y % 16) == 0 && ((uintptr_t)dp % 16) == 0,
                "mde_map_act_bwd: bad argument (n=%lld must be a multiple of 4, 16-byte aligned maps, scale != 0)", (long long)n);
    map_act_bwd_k<<<grid_flat(n / 4), NT, 0, (hipStream_t)stream>>>(dy, y, dp, n / 4, act, scale);
    MDE_LAUNCH_CHECK("map_act_bwd_k");
    return MDE_OK;
}

extern "C" int mde_to_nchw_act_fwd(const void* x, int ldx, const float* bias, float* out, int N, int64_t HW, int C, int act, float scale,
                                   void* stream) {
    MDE_REQUIRE(x && out && N > 0 && HW > 0 && C > 0 && ldx % 8 == 0 && ldx >= (C + 7) / 8 * 8 && act >= 0 && act <= 3 && PW_ALIGNED(x),
                "mde_to_nchw_act_fwd: bad argument (C=%d, ldx=%d)", C, ldx);
    to_nchw_act_fwd_k<<<grid_flat((int64_t)N * HW), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, bias, out, N, HW, C, act, scale);
    MDE_LAUNCH_CHECK("to_nchw_act_fwd_k");
    return MDE_OK;
}

extern "C" int mde_to_nchw_act_bwd(const float* dout, const float* out, void* dx, int lddx, float* dbias, int N, int64_t HW, int C,
                                   int act, float scale, void* stream) {
    MDE_REQUIRE(dout && out && dx && N > 0 && HW > 0 && C > 0 && C <= 64 && lddx % 8 == 0 && lddx >= (C + 7) / 8 * 8 && act >= 0 &&
                    act <= 3 && scale != 0.f && PW_ALIGNED(dx),
                "mde_to_nchw_act_bwd: bad argument (C=%d <= 64, lddx=%d)", C, lddx);
    MDE_DET_REQUIRE("mde_to_nchw_act_bwd", dbias, (int64_t)C);
    int grid = grid_flat((int64_t)N * HW);
    if (grid > 1024) grid = 1024;
    to_nchw_act_bwd_k<<<grid, NT, 0, (hipStream_t)stream>>>(dout, out, (bf16_t*)dx, lddx, dbias, N, HW, C, act, scale, mde_det_dev());
    MDE_LAUNCH_CHECK("to_nchw_act_bwd_k");
    return MDE_OK;
}

// ------------------------------------------------------------------ BTS' image-residual head (Bts.py:264-271)
// depth = get_depth(iconv1): ten sigmoid channels = two RGBA layers + two depths.  The colour channels are RESIDUALS on the input
// image: front = clamp(2 d[0:3] - 1 + rgb, 0, 1), front alpha = clamp(2 d[3] - 1 + mean(rgb), 0, 1), back likewise from d[4:8];
// d[8:] passes through.  fp32 NCHW in and out.  torch.clamp hands the gradient on where min <= v <= max.
__global__ __launch_bounds__(NT) void image_residual_fwd_k(const float* __restrict__ d, const float* __restrict__ rgb, float* __restrict__ out,
                                                           int N, int64_t HW, int C) {
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW, p = i - n * HW;
        const float r = rgb[(n * 3 + 0) * HW + p], g = rgb[(n * 3 + 1) * HW + p], b = rgb[(n * 3 + 2) * HW + p];
        const float add[4] = {r, g, b, (r + g + b) / 3.0f};
        for (int c = 0; c < C; ++c) {
            const float v = d[(n * C + c) * HW + p];
            out[(n * C + c) * HW + p] = c < 8 ? fminf(fmaxf(v * 2.0f - 1.0f + add[c & 3], 0.0f), 1.0f) : v;
        }
    }
}
__global__ __launch_bounds__(NT) void image_residual_bwd_k(const float* __restrict__ dout, const float* __restrict__ d,
                                                           const float* __restrict__ rgb, float* __restrict__ dd, int N, int64_t HW, int C) {
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW, p = i - n * HW;
        const float r = rgb[(n * 3 + 0) * HW + p], g = rgb[(n * 3 + 1) * HW + p], b = rgb[(n * 3 + 2) * HW + p];
        const float add[4] = {r, g, b, (r + g + b) / 3.0f};
        for (int c = 0; c < C; ++c) {
            const int64_t o = (n * C + c) * HW + p;
            float gsum = dout[o];
            if (c < 8) {
                const float v = d[o] * 2.0f - 1.0f + add[c & 3];
                gsum = (v >= 0.0f && v <= 1.0f) ? 2.0f * gsum : 0.0f;
            }
            dd[o] = gsum;
        }
    }
}

extern "C" int mde_image_residual_fwd(const float* d, const float* rgb, float* out, int N, int64_t HW, int C, void* stream) {
    MDE_REQUIRE(d && rgb && out && N > 0 && HW > 0 && C >= 8, "mde_image_residual_fwd: bad argument (C=%d >= 8: two RGBA layers)", C);
    image_residual_fwd_k<<<grid_flat((int64_t)N * HW), NT, 0, (hipStream_t)stream>>>(d, rgb, out, N, HW, C);
    MDE_LAUNCH_CHECK("image_residual_fwd_k");
    return MDE_OK;
}

extern "C" int mde_image_residual_bwd(const float* dout, const float* d, const float* rgb, float* dd, int N, int64_t HW, int C, void* stream) {
    MDE_REQUIRE(dout && d && rgb && dd && N > 0 && HW > 0 && C >= 8, "mde_image_residual_bwd: bad argument (C=%d >= 8)", C);
    image_residual_bwd_k<<<grid_flat((int64_t)N * HW), NT, 0, (hipStream_t)stream>>>(dout, d, rgb, dd, N, HW, C);
    MDE_LAUNCH_CHECK("image_residual_bwd_k");
    return MDE_OK;
}

extern "C" int mde_pack_grouped(const float* src, void* fwd, void* dgrad, int O, int T, int G, void* stream) {
    MDE_REQUIRE(src && (fwd || dgrad) && O > 0 && T > 0 && G > 0 && O % 64 == 0 && 64 % G == 0,
                "mde_pack_grouped: O=%d must be a multiple of 64 and the group size %d must divide 64", O, G);
    pack_grouped_k<<<grid_flat((int64_t)O * T * 64), NT, 0, (hipStream_t)stream>>>(src, (bf16_t*)fwd, (bf16_t*)dgrad, O, T, G);
    MDE_LAUNCH_CHECK("pack_grouped_k");
    return MDE_OK;
}

extern "C" int mde_pack_grouped_split(const float* src, void* fwd2, int O, int T, int G, void* stream) {
    MDE_REQUIRE(src && fwd2 && O > 0 && O % 64 == 0 && T > 0 && G > 0 && 64 % G == 0,
                "mde_pack_grouped_split: O=%d must be a multiple of 64 and the group size %d must divide 64", O, G);
    pack_grouped_split_k<<<grid_flat((int64_t)O * T * 64), NT, 0, (hipStream_t)stream>>>(src, (bf16_t*)fwd2, O, T, G);
    MDE_LAUNCH_CHECK("pack_grouped_split_k");
    return MDE_OK;
}

// ------------------------------------------------------------------ BTS: plane coefficients -> local planar guidance depth
// (Bts.py:105-122 reduction_1x1's tail, :228-231 F.normalize, :124-146 local_planar_guidance, /max_depth)
namespace {

struct Plane { float s0, s1, s2, st, ct, sp, cp, n[3], r, dist; };
__device__ __forceinline__ Plane plane_of(const bf16_t* x, float max_depth) {
    Plane p;
    p.s0 = 1.f / (1.f + expf(-(float)x[0]));
    p.s1 = 1.f / (1.f + expf(-(float)x[1]));
    p.s2 = 1.f / (1.f + expf(-(float)x[2]));
    const float theta = p.s0 * 1.0471975511965976f, phi = p.s1 * 6.283185307179586f;
    sincosf(theta, &p.st, &p.ct);
    sincosf(phi, &p.sp, &p.cp);
    const float n0 = p.st * p.cp, n1 = p.st * p.sp, n2 = p.ct;
    p.r = fmaxf(sqrtf(n0 * n0 + n1 * n1 + n2 * n2), 1e-12f);
    p.n[0] = n0 / p.r; p.n[1] = n1 / p.r; p.n[2] = n2 / p.r;
    p.dist = p.s2 * max_depth;
    return p;
}

// x: bf16 [N][h][w][ldx] (channels 0..2);  out: fp32 [N][h*up][w*up] = dist / (n1 u + n2 v + n3) / max_depth
__global__ __launch_bounds__(NT) void plane_depth_fwd_k(const bf16_t* __restrict__ x, int ldx, float* __restrict__ out, int N, int h, int w,
                                                        int up, float max_depth) {
    const int64_t total = (int64_t)N * h * w * up;          // one thread per (plane pixel, sub-row): `up` consecutive outputs
    const int W = w * up, H = h * up;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int a = (int)(i % up);
        int64_t p = i / up;
        const int j = (int)(p % w); p /= w;
        const int ii = (int)(p % h);
        const int64_t n = p / h;
        const Plane pl = plane_of(x + ((n * h + ii) * (int64_t)w + j) * ldx, max_depth);
        const float v = ((float)a - (float)(up - 1) * 0.5f) / (float)up;
        float* o = out + (n * H + (int64_t)ii * up + a) * W + (int64_t)j * up;
        for (int b = 0; b < up; ++b) {
            const float u = ((float)b - (float)(up - 1) * 0.5f) / (float)up;
            o[b] = pl.dist / (pl.n[0] * u + pl.n[1] * v + pl.n[2]) / max_depth;
        }
    }
}

// dx[n][i][j][0..2] from dout fp32 [N][H][W]; channels 3..7 of the 16-byte chunk are written as zero
__global__ __launch_bounds__(NT) void plane_depth_bwd_k(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ dout,
                                                        bf16_t* __restrict__ dx, int lddx, int N, int h, int w, int up, float max_depth) {
    const int64_t total = (int64_t)N * h * w;
    const int W = w * up, H = h * up;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int64_t p = i;
        const int j = (int)(p % w); p /= w;
        const int ii = (int)(p % h);
        const int64_t n = p / h;
        const Plane pl = plane_of(x + i * ldx, max_depth);
        float dn[3] = {0.f, 0.f, 0.f}, ddist = 0.f;
        for (int a = 0; a < up; ++a) {
            const float v = ((float)a - (float)(up - 1) * 0.5f) / (float)up;
            const float* g = dout + (n * H + (int64_t)ii * up + a) * W + (int64_t)j * up;
            for (int b = 0; b < up; ++b) {
                const float u = ((float)b - (float)(up - 1) * 0.5f) / (float)up;
                const float den = pl.n[0] * u + pl.n[1] * v + pl.n[2];
                const float go = g[b] / max_depth;
                ddist += go / den;
                const float dden = -go * pl.dist / (den * den);
                dn[0] += dden * u;
                dn[1] += dden * v;
                dn[2] += dden;
            }
        }
        // F.normalize: d raw = (d unit - unit (unit . d unit)) / r
        const float dot = pl.n[0] * dn[0] + pl.n[1] * dn[1] + pl.n[2] * dn[2];
        const float r0 = (dn[0] - pl.n[0] * dot) / pl.r, r1 = (dn[1] - pl.n[1] * dot) / pl.r, r2 = (dn[2] - pl.n[2] * dot) / pl.r;
        const float dtheta = r0 * pl.ct * pl.cp + r1 * pl.ct * pl.sp - r2 * pl.st;
        const float dphi = -r0 * pl.st * pl.sp + r1 * pl.st * pl.cp;
        bf16x8_t o;
        o[0] = (bf16_t)(dtheta * 1.0471975511965976f * pl.s0 * (1.f - pl.s0));
        o[1] = (bf16_t)(dphi * 6.283185307179586f * pl.s1 * (1.f - pl.s1));
        o[2] = (bf16_t)(ddist * max_depth * pl.s2 * (1.f - pl.s2));
#pragma unroll
        for (int e = 3; e < 8; ++e) o[e] = (bf16_t)0.f;
        *reinterpret_cast<bf16x8_t*>(dx + i * lddx) = o;
    }
}

// One-channel fp32 map [N][H][W] <-> one bf16 channel of an NHWC tensor [N][H/step][W/step][ld] (F.interpolate(nearest,
// scale 1/step) picks source pixel (step*y, step*x): Bts.py:234,247; torch.cat into a wider tensor: :238,251,263)
__global__ __launch_bounds__(NT) void map_to_slot_k(const float* __restrict__ src, bf16_t* __restrict__ dst, int ld, int N, int H, int W, int step) {
    const int h = H / step, w = W / step;
    const int64_t total = (int64_t)N * h * w;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int64_t p = i;
        const int x = (int)(p % w); p /= w;
        const int y = (int)(p % h);
        const int64_t n = p / h;
        dst[i * ld] = (bf16_t)src[(n * H + (int64_t)y * step) * W + (int64_t)x * step];
    }
}
__global__ __launch_bounds__(NT) void slot_to_map_add_k(const bf16_t* __restrict__ dslot, int ld, float* __restrict__ dsrc, int N, int H, int W,
                                                        int step) {
    const int h = H / step, w = W / step;
    const int64_t total = (int64_t)N * h * w;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int64_t p = i;
        const int x = (int)(p % w); p /= w;
        const int y = (int)(p % h);
        const int64_t n = p / h;
        dsrc[(n * H + (int64_t)y * step) * W + (int64_t)x * step] += (float)dslot[i * ld];
    }
}

}  // namespace

extern "C" int mde_plane_depth_fwd(const void* x, int ldx, float* out, int N, int h, int w, int up, float max_depth, void* stream) {
    MDE_REQUIRE(x && out && N > 0 && h > 0 && w > 0 && up >= 1 && up <= 16 && ldx >= 8 && ldx % 8 == 0 && max_depth > 0.f,
                "mde_plane_depth_fwd: bad argument (up=%d, ldx=%d)", up, ldx);
    plane_depth_fwd_k<<<grid_flat((int64_t)N * h * w * up), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, out, N, h, w, up, max_depth);
    MDE_LAUNCH_CHECK("plane_depth_fwd_k");
    return MDE_OK;
}

extern "C" int mde_plane_depth_bwd(const void* x, int ldx, const float* dout, void* dx, int lddx, int N, int h, int w, int up, float max_depth,
                                   void* stream) {
    MDE_REQUIRE(x && dout && dx && N > 0 && h > 0 && w > 0 && up >= 1 && up <= 16 && ldx >= 8 && ldx % 8 == 0 && lddx >= 8 &&
                    lddx % 8 == 0 && PW_ALIGNED(dx) && max_depth > 0.f,
                "mde_plane_depth_bwd: bad argument (up=%d, ldx=%d, lddx=%d)", up, ldx, lddx);
    plane_depth_bwd_k<<<grid_flat((int64_t)N * h * w), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, dout, (bf16_t*)dx, lddx, N, h, w, up,
                                                                                   max_depth);
    MDE_LAUNCH_CHECK("plane_depth_bwd_k");
    return MDE_OK;
}

extern "C" int mde_map_to_slot(const float* src, void* dst, int ld, int N, int H, int W, int step, void* stream) {
    MDE_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && step >= 1 && H % step == 0 && W % step == 0 && ld > 0,
                "mde_map_to_slot: bad argument (H=%d, W=%d, step=%d)", H, W, step);
    map_to_slot_k<<<grid_flat((int64_t)N * (H / step) * (W / step)), NT, 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, ld, N, H, W, step);
    MDE_LAUNCH_CHECK("map_to_slot_k");
    return MDE_OK;
}

extern "C" int mde_slot_to_map_add(const void* dslot, int ld, float* dsrc, int N, int H, int W, int step, void* stream) {
    MDE_REQUIRE(dslot && dsrc && N > 0 && H > 0 && W > 0 && step >= 1 && H % step == 0 && W % step == 0 && ld > 0,
                "mde_slot_to_map_add: bad argument (H=%d, W=%d, step=%d)", H, W, step);
    slot_to_map_add_k<<<grid_flat((int64_t)N * (H / step) * (W / step)), NT, 0, (hipStream_t)stream>>>((const bf16_t*)dslot, ld, dsrc, N, H, W, step);
    MDE_LAUNCH_CHECK("slot_to_map_add_k");
    return MDE_OK;
}
