// Optimiser step and parameter plumbing on flat fp32 ranges (gfx950).
//   Adam as torch.optim.Adam computes it (reference modules/laina.py:51-57); the same pass
//   refreshes the bf16 shadow copy the conv kernels read, so no separate cast pass per step.
//   mde_pack_wt builds the transposed ("dgrad") bf16 weight packing [I][T][O] from the
//   fp32 master [O][T][I] through a 32x32 LDS tile (coalesced on both sides).
#include "mde_common.h"

namespace {

constexpr int NT = 256;

// DEC = false: torch.optim.Adam (L2: the decay joins the gradient); DEC = true: torch.optim.AdamW (the parameter is
// first scaled by 1 - lr * wd, the moments never see the decay)
template <bool DEC>
__global__ __launch_bounds__(NT) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, bf16_t* __restrict__ pb, int64_t n, float lr,
                                             float b1, float b2, float eps, float wd, float gs, float bc1, float bc2s) {
    // bc1 = 1 - b1^t ; bc2s = sqrt(1 - b2^t)
    const float step = lr / bc1;
    const float keep = DEC ? 1.f - lr * wd : 1.f, l2 = DEC ? 0.f : wd;
    for (int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * NT * 4) {
        if (i + 4 <= n) {
            f32x4_t pp = *reinterpret_cast<f32x4_t*>(p + i);
            const f32x4_t gg = *reinterpret_cast<const f32x4_t*>(g + i);
            f32x4_t mm = *reinterpret_cast<f32x4_t*>(m + i), vv = *reinterpret_cast<f32x4_t*>(v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (DEC) pp[e] *= keep;
                const float ge = gg[e] * gs + l2 * pp[e];
                mm[e] = b1 * mm[e] + (1.f - b1) * ge;
                vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
                pp[e] -= step * mm[e] / (sqrtf(vv[e]) / bc2s + eps);
            }
            *reinterpret_cast<f32x4_t*>(p + i) = pp;
            *reinterpret_cast<f32x4_t*>(m + i) = mm;
            *reinterpret_cast<f32x4_t*>(v + i) = vv;
            if (pb) {
                bf16x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)pp[e];
                *reinterpret_cast<bf16x4_t*>(pb + i) = o;
            }
        } else {
            for (int64_t j = i; j < n; ++j) {
                if (DEC) p[j] *= keep;
                const float ge = g[j] * gs + l2 * p[j];
                m[j] = b1 * m[j] + (1.f - b1) * ge;
                v[j] = b2 * v[j] + (1.f - b2) * ge * ge;
                p[j] -= step * m[j] / (sqrtf(v[j]) / bc2s + eps);
                if (pb) pb[j] = (bf16_t)p[j];
            }
        }
    }
}

// torch.optim.SGD with momentum (dampening 0, no Nesterov): g += wd * p; buf = mu * buf + g; p -= lr * buf.
// A zero-initialised buf makes the first step buf = g, as torch does.
__global__ __launch_bounds__(NT) void sgd_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                            bf16_t* __restrict__ pb, int64_t n, float lr, float mu, float wd, float gs) {
    for (int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * NT * 4) {
        if (i + 4 <= n) {
            f32x4_t pp = *reinterpret_cast<f32x4_t*>(p + i), bb = *reinterpret_cast<f32x4_t*>(buf + i);
            const f32x4_t gg = *reinterpret_cast<const f32x4_t*>(g + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bb[e] = mu * bb[e] + (gg[e] * gs + wd * pp[e]);
                pp[e] -= lr * bb[e];
            }
            *reinterpret_cast<f32x4_t*>(p + i) = pp;
            *reinterpret_cast<f32x4_t*>(buf + i) = bb;
            if (pb) {
                bf16x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)pp[e];
                *reinterpret_cast<bf16x4_t*>(pb + i) = o;
            }
        } else {
            for (int64_t j = i; j < n; ++j) {
                buf[j] = mu * buf[j] + (g[j] * gs + wd * p[j]);
                p[j] -= lr * buf[j];
                if (pb) pb[j] = (bf16_t)p[j];
            }
        }
    }
}

__global__ __launch_bounds__(NT) void cast_k(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n,
                                             const uint32_t* __restrict__ gate) {
    if (gate && *gate == 0u) return;                 // gated form: the fingerprint found the masters unchanged
    for (int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * NT * 4) {
        if (i + 4 <= n) {
            const f32x4_t s = *reinterpret_cast<const f32x4_t*>(src + i);
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)s[e];
            *reinterpret_cast<bf16x4_t*>(dst + i) = o;
        } else {
            for (int64_t j = i; j < n; ++j) dst[j] = (bf16_t)src[j];
        }
    }
}

// dst[i][t][o] = bf16(src[o][t][i]); grid (ceil(I/32), ceil(O/32), T), block (32, 8)
__global__ void pack_wt_k(const float* __restrict__ src, bf16_t* __restrict__ dst, int O, int T, int I) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int i0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int o = o0 + r, i = i0 + threadIdx.x;
        tile[r][threadIdx.x] = (o < O && i < I) ? src[((size_t)o * T + t) * I + i] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = i0 + r, o = o0 + threadIdx.x;
        if (i < I && o < O) dst[((size_t)i * T + t) * O + o] = (bf16_t)tile[threadIdx.x][r];
    }
}

// one launch for many weights: block b belongs to the last job whose first_block <= b
__global__ void pack_wt_batch_k(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                const mde_pack_job* __restrict__ jobs, int njobs, const uint32_t* __restrict__ gate) {
    __shared__ float tile[32][33];
    if (gate && *gate == 0u) return;
    int lo = 0, hi = njobs - 1;                      // uniform binary search (the table is L2-resident)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const mde_pack_job j = jobs[lo];
    const int O = (int)j.O, T = (int)j.T, I = (int)j.I;
    const int nib = (I + 31) / 32, nob = (O + 31) / 32;
    const int local = (int)((int64_t)blockIdx.x - j.first_block);
    const int ib = local % nib, ob = (local / nib) % nob, t = local / (nib * nob);
    if (t >= T) return;
    const float* s = src + j.off;
    bf16_t* d = dst + j.off;
    const int i0 = ib * 32, o0 = ob * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int o = o0 + r, i = i0 + threadIdx.x;
        tile[r][threadIdx.x] = (o < O && i < I) ? s[((size_t)o * T + t) * I + i] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = i0 + r, o = o0 + threadIdx.x;
        if (i < I && o < O) d[((size_t)i * T + t) * O + o] = (bf16_t)tile[threadIdx.x][r];
    }
}

// The eval-mode two-term weight shadow of many weights in one launch (blocks as pack_wt_batch_k): hi = (bf16)w,
// lo = (bf16)(w - hi), written as the tap-doubled GEMM operand of job j at dst + 2 * off:
//   TR = false: dst[o][h * T + t][i]   (forward operand [O][2T][I]);   TR = true: dst[i][h * T + t][o]   (transposed, [I][2T][O])
template <bool TR>
__global__ void pack_split_batch_k(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                   const mde_pack_job* __restrict__ jobs, int njobs) {
    __shared__ float tile[32][33];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const mde_pack_job j = jobs[lo];
    const int O = (int)j.O, T = (int)j.T, I = (int)j.I;
    const int nib = (I + 31) / 32, nob = (O + 31) / 32;
    const int local = (int)((int64_t)blockIdx.x - j.first_block);
    const int ib = local % nib, ob = (local / nib) % nob, t = local / (nib * nob);
    if (t >= T) return;
    const float* s = src + j.off;
    bf16_t* d = dst + 2 * j.off;
    const int i0 = ib * 32, o0 = ob * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int o = o0 + r, i = i0 + threadIdx.x;
        const float v = (o < O && i < I) ? s[((size_t)o * T + t) * I + i] : 0.f;
        if (TR) {
            tile[r][threadIdx.x] = v;
        } else if (o < O && i < I) {
            const bf16_t h = (bf16_t)v;
            d[((size_t)o * 2 * T + t) * I + i] = h;
            d[((size_t)o * 2 * T + T + t) * I + i] = (bf16_t)(v - (float)h);
        }
    }
    if (!TR) return;
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = i0 + r, o = o0 + threadIdx.x;
        if (i < I && o < O) {
            const float v = tile[threadIdx.x][r];
            const bf16_t h = (bf16_t)v;
            d[((size_t)i * 2 * T + t) * O + o] = h;
            d[((size_t)i * 2 * T + T + t) * O + o] = (bf16_t)(v - (float)h);
        }
    }
}

// Position-weighted 64-bit sum of the raw words of a flat fp32 range: any in-place change of a parameter that torch's
// version counters do not see (writes through `.data`, collectives into detached views) changes it.
struct FpState { unsigned long long acc, last; uint32_t changed, pad; };

__global__ __launch_bounds__(NT) void fingerprint_k(const uint32_t* __restrict__ w, int64_t n, FpState* st) {
    __shared__ unsigned long long sh[NT / 64];
    unsigned long long a = 0ull;
    for (int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * NT * 4) {
        if (i + 4 <= n) {
            const i32x4_t v = *reinterpret_cast<const i32x4_t*>(w + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) a += (unsigned long long)(uint32_t)v[e] * (unsigned long long)(2 * (i + e) + 1);
        } else {
            for (int64_t j = i; j < n; ++j) a += (unsigned long long)w[j] * (unsigned long long)(2 * j + 1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0ull;
        for (int k = 0; k < NT / 64; ++k) t += sh[k];
        atomicAdd(&st->acc, t);
    }
}

__global__ void fingerprint_finish_k(FpState* st) {
    st->changed = st->acc != st->last ? 1u : 0u;
    st->last = st->acc;
    st->acc = 0ull;
}

int grid_for4(int64_t n) {
    int64_t nb = (n / 4 + NT - 1) / NT + 1;
    return (int)(nb > 256 * 8 ? 256 * 8 : nb);
}

}  // namespace

extern "C" int mde_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, float grad_scale, int step,
                             void* stream) {
    MDE_REQUIRE(p && g && m && v && n > 0 && step >= 1, "mde_adam_step: bad argument");
    MDE_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                    ((uintptr_t)v % 16) == 0 && (!p_bf16 || ((uintptr_t)p_bf16 % 8) == 0),
                "mde_adam_step: ranges must be 16-byte aligned");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    adam_k<false><<<grid_for4(n), NT, 0, (hipStream_t)stream>>>(p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps,
                                                               weight_decay, grad_scale, bc1, bc2s);
    MDE_LAUNCH_CHECK("adam_k");
    return MDE_OK;
}

extern "C" int mde_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float grad_scale, int step, void* stream) {
    MDE_REQUIRE(p && g && m && v && n > 0 && step >= 1, "mde_adamw_step: bad argument");
    MDE_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                    ((uintptr_t)p_bf16 % 8) == 0,
                "mde_adamw_step: ranges must be 16-byte aligned");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    adam_k<true><<<grid_for4(n), NT, 0, (hipStream_t)stream>>>(p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps,
                                                              weight_decay, grad_scale, bc1, bc2s);
    MDE_LAUNCH_CHECK("adam_k<AdamW>");
    return MDE_OK;
}

extern "C" int mde_sgd_step(float* p, const float* g, float* buf, void* p_bf16, int64_t n, float lr, float momentum,
                            float weight_decay, float grad_scale, void* stream) {
    MDE_REQUIRE(p && g && buf && n > 0, "mde_sgd_step: bad argument");
    MDE_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)buf % 16) == 0 && ((uintptr_t)p_bf16 % 8) == 0,
                "mde_sgd_step: ranges must be 16-byte aligned");
    sgd_k<<<grid_for4(n), NT, 0, (hipStream_t)stream>>>(p, g, buf, (bf16_t*)p_bf16, n, lr, momentum, weight_decay, grad_scale);
    MDE_LAUNCH_CHECK("sgd_k");
    return MDE_OK;
}

extern "C" int mde_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
    MDE_REQUIRE(src && dst && n > 0, "mde_cast_bf16: bad argument");
    MDE_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "mde_cast_bf16: alignment");
    cast_k<<<grid_for4(n), NT, 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, n, nullptr);
    MDE_LAUNCH_CHECK("cast_k");
    return MDE_OK;
}

extern "C" size_t mde_param_fingerprint_state_bytes(void) { return sizeof(FpState); }

extern "C" int mde_param_fingerprint(const float* p, int64_t n, void* state, void* stream) {
    MDE_REQUIRE(p && state && n > 0 && ((uintptr_t)p % 16) == 0, "mde_param_fingerprint: bad argument");
    fingerprint_k<<<grid_for4(n), NT, 0, (hipStream_t)stream>>>(reinterpret_cast<const uint32_t*>(p), n, (FpState*)state);
    fingerprint_finish_k<<<1, 1, 0, (hipStream_t)stream>>>((FpState*)state);
    MDE_LAUNCH_CHECK("fingerprint_k");
    return MDE_OK;
}

extern "C" int mde_refresh_if_changed(const float* src, void* shadow, void* packed, const mde_pack_job* jobs, int njobs,
                                      int64_t nblocks, int64_t n, const void* state, void* stream) {
    MDE_REQUIRE(src && shadow && state && n > 0, "mde_refresh_if_changed: bad argument");
    MDE_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)shadow % 8) == 0, "mde_refresh_if_changed: alignment");
    const uint32_t* gate = &((const FpState*)state)->changed;
    cast_k<<<grid_for4(n), NT, 0, (hipStream_t)stream>>>(src, (bf16_t*)shadow, n, gate);
    if (packed && jobs && njobs > 0 && nblocks > 0) {
        MDE_REQUIRE(nblocks < (1ll << 31), "mde_refresh_if_changed: grid too large");
        pack_wt_batch_k<<<dim3((unsigned)nblocks), dim3(32, 8), 0, (hipStream_t)stream>>>(src, (bf16_t*)packed, jobs, njobs, gate);
    }
    MDE_LAUNCH_CHECK("mde_refresh_if_changed");
    return MDE_OK;
}

extern "C" int mde_pack_wt_batch(const float* src, void* dst, const mde_pack_job* jobs, int njobs, int64_t nblocks,
                                 void* stream) {
    MDE_REQUIRE(src && dst && jobs && njobs > 0 && nblocks > 0 && nblocks < (1ll << 31), "mde_pack_wt_batch: bad argument");
    pack_wt_batch_k<<<dim3((unsigned)nblocks), dim3(32, 8), 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, jobs, njobs, nullptr);
    MDE_LAUNCH_CHECK("pack_wt_batch_k");
    return MDE_OK;
}

extern "C" int mde_pack_wt(const float* src, void* dst, int O, int T, int I, void* stream) {
    MDE_REQUIRE(src && dst && O > 0 && T > 0 && I > 0 && T <= 65535, "mde_pack_wt: bad argument");
    pack_wt_k<<<dim3(mde_cdiv(I, 32), mde_cdiv(O, 32), T), dim3(32, 8), 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, O, T, I);
    MDE_LAUNCH_CHECK("pack_wt_k");
    return MDE_OK;
}

extern "C" int mde_pack_split_batch(const float* src, void* dst, const mde_pack_job* jobs, int njobs, int64_t nblocks,
                                    int transposed, void* stream) {
    MDE_REQUIRE(src && dst && jobs && njobs > 0 && nblocks > 0 && nblocks < (1ll << 31), "mde_pack_split_batch: bad argument");
    if (transposed)
        pack_split_batch_k<true><<<dim3((unsigned)nblocks), dim3(32, 8), 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, jobs, njobs);
    else
        pack_split_batch_k<false><<<dim3((unsigned)nblocks), dim3(32, 8), 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, jobs, njobs);
    MDE_LAUNCH_CHECK("pack_split_batch_k");
    return MDE_OK;
}
