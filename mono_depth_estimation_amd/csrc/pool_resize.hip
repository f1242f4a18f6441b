// MaxPool 3x3/2 (torchvision stem, used at reference network/FCRN.py:319,356), the
// bilinear(align_corners=True)+sigmoid head (FCRN.py:341,369-371) and the NCHW<->NHWC
// boundary layout changes.  HBM-bound streaming kernels, 16-byte accesses where the layout
// allows.
#include "mde_common.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------ maxpool 3x3 s2 p1
// one thread = one output pixel x 8 channels.  idx = kh*3+kw of the FIRST maximum in scan
// order (ATen: `val > max || isnan(val)`), kept for the backward routing of ties (post-ReLU
// inputs tie at 0 all the time).
__global__ __launch_bounds__(NT) void maxpool_fwd_k(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                    uint8_t* __restrict__ idx, int N, int H, int W, int C,
                                                    int OH, int OW) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * OH * OW * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int col = (int)(i % cpr);
        int64_t p = i / cpr;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int n = (int)(p / OH);
        float best[8];
        int bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = -1; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(x + (((int64_t)n * H + ih) * W + iw) * C + col * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (float)t[e];
                    if (bi[e] < 0 || v > best[e] || v != v) { best[e] = v; bi[e] = kh * 3 + kw; }
                }
            }
        }
        bf16x8_t o;
        uint64_t packed = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = (bf16_t)best[e];
            packed |= (uint64_t)(uint8_t)bi[e] << (8 * e);
        }
        const int64_t ob = (((int64_t)n * OH + oh) * OW + ow) * C + col * 8;
        *reinterpret_cast<bf16x8_t*>(out + ob) = o;
        *reinterpret_cast<uint64_t*>(idx + ob) = packed;
    }
}

// gather form (no atomics): an input pixel sums dout of the <= 4 windows whose argmax is it.
__global__ __launch_bounds__(NT) void maxpool_bwd_k(const bf16_t* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                    bf16_t* __restrict__ dx, int N, int H, int W, int C, int OH, int OW) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * H * W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int col = (int)(i % cpr);
        int64_t p = i / cpr;
        const int iw = (int)(p % W); p /= W;
        const int ih = (int)(p % H);
        const int n = (int)(p / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // windows oh with oh*2-1 <= ih <= oh*2+1
        const int oh0 = ih >> 1, oh1 = (ih + 1) >> 1;
        const int ow0 = iw >> 1, ow1 = (iw + 1) >> 1;
        for (int oh = oh0; oh <= oh1; ++oh) {
            if (oh >= OH) continue;
            const int kh = ih - (oh * 2 - 1);
            for (int ow = ow0; ow <= ow1; ++ow) {
                if (ow >= OW) continue;
                const int k = kh * 3 + (iw - (ow * 2 - 1));
                const int64_t ob = (((int64_t)n * OH + oh) * OW + ow) * C + col * 8;
                const uint64_t packed = *reinterpret_cast<const uint64_t*>(idx + ob);
                const bf16x8_t g = *reinterpret_cast<const bf16x8_t*>(dout + ob);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if ((int)((packed >> (8 * e)) & 0xFF) == k) acc[e] += (float)g[e];
            }
        }
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[e];
        *reinterpret_cast<bf16x8_t*>(dx + (((int64_t)n * H + ih) * W + iw) * C + col * 8) = o;
    }
}

// ------------------------------------------------------------------ maxpool k x k, stride s, no padding, over a spatial VIEW
// nn.MaxPool2d(2, 2) of VGG-19-BN and the cropped pools of Eigen's scale-2 / scale-3 stacks (Eigen.py:23,41 `pool(x)[:, :,
// 1:-1, 1:-1]`; :52,65 `conv(img)[:, :, 2:-3, 2:-3]` then MaxPool2d(3, 1)): the crop is folded into the view -- x points at
// the view's first pixel inside the producing tensor, whose own row / image pitches (in pixels) are passed in.
struct PoolView { int ld, wpitch; int64_t ipitch; int H, W; };     // channel stride, pixels per row, pixels per image, view size

__global__ __launch_bounds__(NT) void maxpool_g_fwd_k(const bf16_t* __restrict__ x, PoolView v, bf16_t* __restrict__ out, int ldo,
                                                      uint8_t* __restrict__ idx, int N, int C, int OH, int OW, int k, int s) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * OH * OW * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int col = (int)(i % cpr);
        int64_t p = i / cpr;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int n = (int)(p / OH);
        float best[8];
        int bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = -1; }
        for (int kh = 0; kh < k; ++kh)
            for (int kw = 0; kw < k; ++kw) {
                const int ih = oh * s + kh, iw = ow * s + kw;              // (inside the view by the definition of OH / OW)
                const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(x + ((int64_t)n * v.ipitch + (int64_t)ih * v.wpitch + iw) * v.ld + col * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float val = (float)t[e];
                    if (bi[e] < 0 || val > best[e] || val != val) { best[e] = val; bi[e] = kh * k + kw; }     // ATen's tie / NaN rule
                }
            }
        bf16x8_t o;
        uint64_t packed = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = (bf16_t)best[e];
            packed |= (uint64_t)(uint8_t)bi[e] << (8 * e);
        }
        const int64_t opix = ((int64_t)n * OH + oh) * OW + ow;
        *reinterpret_cast<bf16x8_t*>(out + opix * ldo + col * 8) = o;
        *reinterpret_cast<uint64_t*>(idx + opix * C + col * 8) = packed;
    }
}

// gather form (no atomics): a view pixel sums dout of the windows whose argmax it is; dx (+)= that
__global__ __launch_bounds__(NT) void maxpool_g_bwd_k(const bf16_t* __restrict__ dout, int ldd, const uint8_t* __restrict__ idx,
                                                      bf16_t* __restrict__ dx, PoolView v, int N, int C, int OH, int OW, int k, int s,
                                                      int accumulate) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)N * v.H * v.W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int col = (int)(i % cpr);
        int64_t p = i / cpr;
        const int iw = (int)(p % v.W); p /= v.W;
        const int ih = (int)(p % v.H);
        const int n = (int)(p / v.H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // windows oh with oh * s <= ih <= oh * s + k - 1
        const int oh1 = min(ih / s, OH - 1), ow1 = min(iw / s, OW - 1);
        const int oh0 = max(0, (ih - k + s) / s), ow0 = max(0, (iw - k + s) / s);       // ceil((ih - k + 1) / s) for ih - k + 1 > -s
        for (int oh = oh0; oh <= oh1; ++oh) {
            const int kh = ih - oh * s;
            if (kh < 0 || kh >= k) continue;
            for (int ow = ow0; ow <= ow1; ++ow) {
                const int kw = iw - ow * s;
                if (kw < 0 || kw >= k) continue;
                const int64_t opix = ((int64_t)n * OH + oh) * OW + ow;
                const uint64_t packed = *reinterpret_cast<const uint64_t*>(idx + opix * C + col * 8);
                const bf16x8_t g = *reinterpret_cast<const bf16x8_t*>(dout + opix * ldd + col * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if ((int)((packed >> (8 * e)) & 0xFF) == kh * k + kw) acc[e] += (float)g[e];
            }
        }
        bf16_t* dst = dx + ((int64_t)n * v.ipitch + (int64_t)ih * v.wpitch + iw) * v.ld + col * 8;
        bf16x8_t o;
        if (accumulate) {
            const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(dst);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(acc[e] + (float)old[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[e];
        }
        *reinterpret_cast<bf16x8_t*>(dst) = o;
    }
}

// ------------------------------------------------------------------ bilinear (align_corners) + sigmoid
// source coordinate exactly as ATen computes it in float: src = dst * ((in-1)/(out-1)).
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_of(int dst, float scale, int in_size) {
    const float src = scale * (float)dst;
    Lerp r;
    r.i0 = min((int)src, in_size - 1);
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
    r.l0 = 1.f - r.l1;
    return r;
}

__global__ __launch_bounds__(NT) void upsig_fwd_k(const float* __restrict__ x, float* __restrict__ out, int N, int H,
                                                  int W, int C, int OH, int OW, float sh, float sw) {
    const int64_t total = (int64_t)N * C * OH * OW;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int ow = (int)(i % OW);
        int64_t p = i / OW;
        const int oh = (int)(p % OH); p /= OH;
        const int c = (int)(p % C);
        const int n = (int)(p / C);
        const Lerp ly = lerp_of(oh, sh, H), lx = lerp_of(ow, sw, W);
        const float* b = x + (int64_t)n * H * W * C + c;
        const float v00 = b[((int64_t)ly.i0 * W + lx.i0) * C], v01 = b[((int64_t)ly.i0 * W + lx.i1) * C];
        const float v10 = b[((int64_t)ly.i1 * W + lx.i0) * C], v11 = b[((int64_t)ly.i1 * W + lx.i1) * C];
        const float v = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
        out[i] = 1.f / (1.f + expf(-v));
    }
}

// gather form of the transposed interpolation: source pixel (h,w) collects every destination
// pixel whose stencil touches it (deterministic, no atomics); sigmoid' = out*(1-out).
__global__ __launch_bounds__(NT) void upsig_bwd_k(const float* __restrict__ dout, const float* __restrict__ out,
                                                  float* __restrict__ dx, int N, int H, int W, int C, int OH, int OW,
                                                  float sh, float sw, float ish, float isw) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        // destination rows whose i0 is h-1 or h lie in [(h-1)/s, (h+1)/s]; widen by one and test exactly
        const int oh_lo = max(0, (int)floorf((float)(h - 1) * ish) - 1), oh_hi = min(OH - 1, (int)ceilf((float)(h + 1) * ish) + 1);
        const int ow_lo = max(0, (int)floorf((float)(w - 1) * isw) - 1), ow_hi = min(OW - 1, (int)ceilf((float)(w + 1) * isw) + 1);
        const float* go = dout + ((int64_t)n * C + c) * OH * OW;
        const float* oo = out + ((int64_t)n * C + c) * OH * OW;
        float acc = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const Lerp ly = lerp_of(oh, sh, H);
            float wy = 0.f;
            if (ly.i0 == h) wy += ly.l0;
            if (ly.i1 == h) wy += ly.l1;
            if (wy == 0.f) continue;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const Lerp lx = lerp_of(ow, sw, W);
                float wx = 0.f;
                if (lx.i0 == w) wx += lx.l0;
                if (lx.i1 == w) wx += lx.l1;
                if (wx == 0.f) continue;
                const float o = oo[(int64_t)oh * OW + ow];
                acc += wy * wx * (go[(int64_t)oh * OW + ow] * o * (1.f - o));
            }
        }
        dx[i] = acc;
    }
}

// ------------------------------------------------------------------ layout changes
__global__ __launch_bounds__(NT) void nchw2nhwc_k(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int C,
                                                  int H, int W) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int64_t hw = p % ((int64_t)H * W);
        const int64_t n = p / ((int64_t)H * W);
        dst[i] = (bf16_t)src[(n * C + c) * (int64_t)H * W + hw];
    }
}
// the same into a wider pixel stride: channels [C, Cpad) are written as zeros (generic-stem input, in_channels != 3)
__global__ __launch_bounds__(NT) void nchw2nhwc_pad_k(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int C,
                                                      int H, int W, int Cpad) {
    const int64_t total = (int64_t)N * H * W * Cpad;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % Cpad);
        const int64_t p = i / Cpad;
        const int64_t hw = p % ((int64_t)H * W);
        const int64_t n = p / ((int64_t)H * W);
        dst[i] = c < C ? (bf16_t)src[(n * C + c) * (int64_t)H * W + hw] : (bf16_t)0.f;
    }
}
__global__ __launch_bounds__(NT) void nhwc2nchw_k(const bf16_t* __restrict__ src, float* __restrict__ dst, int N, int C,
                                                  int H, int W) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t hw = i % ((int64_t)H * W);
        const int64_t q = i / ((int64_t)H * W);
        const int c = (int)(q % C);
        const int64_t n = q / C;
        dst[i] = (float)src[(n * (int64_t)H * W + hw) * C + c];
    }
}

int grid_for(int64_t total) {
    int64_t nb = (total + NT - 1) / NT;
    return (int)(nb > 256 * 16 ? 256 * 16 : (nb < 1 ? 1 : nb));
}

// The eval-mode image / stem-weight operands of an image conv on the GEMM kernel (graph.ImageStem): 16 channel slots
//   image  [xh | xl | xh | 0 ...]   xh = (bf16)x, xl = (bf16)(x - xh)          (C <= 5 image channels, 3 C <= 16)
//   weight [wh | wh | wl | 0 ...]   wh = (bf16)w, wl = (bf16)(w - wh)
// so one contraction over the 16 slots is xh wh + xl wh + xh wl = x w up to 2^-16 relative: the fp32 image against the
// two-term weight shadow, in the launches the training step makes (a tap list doubled instead would split a 7x7 into four
// accumulating launches whose partial sums each pass through a 16-bit rounding).
__global__ __launch_bounds__(NT) void nchw2nhwc_split16_k(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int C, int H, int W) {
    const int64_t HW = (int64_t)H * W, total = (int64_t)N * HW;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t n = i / HW, px = i - n * HW;
        bf16_t o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = (bf16_t)0.f;
        for (int c = 0; c < C; ++c) {
            const float v = src[(n * C + c) * HW + px];
            const bf16_t h = (bf16_t)v;
            o[c] = h;
            o[C + c] = (bf16_t)(v - (float)h);
            o[2 * C + c] = h;
        }
        bf16x8_t a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = o[e]; b[e] = o[8 + e]; }
        *reinterpret_cast<bf16x8_t*>(dst + i * 16) = a;
        *reinterpret_cast<bf16x8_t*>(dst + i * 16 + 8) = b;
    }
}

// src fp32 [rows][Cp] (rows = O * T of a stem weight stored with its C image channels padded to Cp) -> dst bf16 [rows][16]
__global__ __launch_bounds__(NT) void stem_weight_split16_k(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows, int Cp, int C) {
    for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * NT) {
        bf16_t o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = (bf16_t)0.f;
        for (int c = 0; c < C; ++c) {
            const float v = src[r * Cp + c];
            const bf16_t h = (bf16_t)v;
            o[c] = h;
            o[C + c] = h;
            o[2 * C + c] = (bf16_t)(v - (float)h);
        }
        bf16x8_t a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = o[e]; b[e] = o[8 + e]; }
        *reinterpret_cast<bf16x8_t*>(dst + r * 16) = a;
        *reinterpret_cast<bf16x8_t*>(dst + r * 16 + 8) = b;
    }
}

}  // namespace

// nn.PixelShuffle(2) on NHWC bf16 (reference FCRN.py:236,245 — the FasterUpProj decoder):
//   dst[n][2y+a][2x+b][c] = src[n][y][x][4c + 2a + b],  src has 4C channels (pixel stride ld_src), dst C (ld_dst).
// A thread moves 32 source channels (64 B) = 8 destination channels at each of the 4 sub-pixels (16 B each).
// INV: the same permutation read the other way (gradient of the forward: src is written from dst).
template <bool INV>
__global__ __launch_bounds__(256) void pixel_shuffle_k(bf16_t* __restrict__ src, int ld_src, bf16_t* __restrict__ dst,
                                                       int ld_dst, int64_t pixels, int h, int w, int C) {
    const int groups = C / 8;
    const int64_t total = pixels * groups;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / groups;
        const int gq = (int)(i - pix * groups);
        const int x = (int)(pix % w);
        const int64_t ny = pix / w;                 // n * h + y
        const int y = (int)(ny % h);
        const int64_t n = ny / h;
        bf16_t* sp = src + pix * ld_src + gq * 32;
        bf16_t* dp = dst + ((n * 2 * h + 2 * y) * (int64_t)(2 * w) + 2 * x) * ld_dst + gq * 8;
        bf16x8_t v[4], o[4];
        if (!INV) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const bf16x8_t*>(sp + 8 * q);
#pragma unroll
            for (int ab = 0; ab < 4; ++ab)
#pragma unroll
                for (int c = 0; c < 8; ++c) o[ab][c] = v[(4 * c + ab) >> 3][(4 * c + ab) & 7];
#pragma unroll
            for (int ab = 0; ab < 4; ++ab)
                *reinterpret_cast<bf16x8_t*>(dp + ((int64_t)(ab >> 1) * 2 * w + (ab & 1)) * ld_dst) = o[ab];
        } else {
#pragma unroll
            for (int ab = 0; ab < 4; ++ab)
                o[ab] = *reinterpret_cast<const bf16x8_t*>(dp + ((int64_t)(ab >> 1) * 2 * w + (ab & 1)) * ld_dst);
#pragma unroll
            for (int ab = 0; ab < 4; ++ab)
#pragma unroll
                for (int c = 0; c < 8; ++c) v[(4 * c + ab) >> 3][(4 * c + ab) & 7] = o[ab][c];
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<bf16x8_t*>(sp + 8 * q) = v[q];
        }
    }
}

extern "C" int mde_pixel_shuffle2(void* src, int ld_src, void* dst, int ld_dst, int N, int h, int w, int C, int inverse,
                                  void* stream) {
    MDE_REQUIRE(src && dst && N > 0 && h > 0 && w > 0 && C > 0, "mde_pixel_shuffle2: bad argument");
    MDE_REQUIRE(C % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && ld_src >= 4 * C && ld_dst >= C &&
                    ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                "mde_pixel_shuffle2: C=%d, ld_src=%d, ld_dst=%d must be multiples of 8 with 16-byte aligned bases", C, ld_src, ld_dst);
    const int64_t pixels = (int64_t)N * h * w, total = pixels * (C / 8);
    const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    if (inverse) pixel_shuffle_k<true><<<grid, 256, 0, (hipStream_t)stream>>>((bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, pixels, h, w, C);
    else pixel_shuffle_k<false><<<grid, 256, 0, (hipStream_t)stream>>>((bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, pixels, h, w, C);
    MDE_LAUNCH_CHECK("pixel_shuffle_k");
    return MDE_OK;
}

static inline int maxpool_out(int in, int ceil_mode) {
    if (!ceil_mode) return (in + 2 - 3) / 2 + 1;
    int o = (in + 2 - 3 + 1) / 2 + 1;             // ceil((in + 2p - k) / s) + 1
    if ((o - 1) * 2 >= in + 1) --o;               // ATen: the last window must start inside the input or its left padding
    return o;
}

extern "C" int mde_maxpool_fwd2(const void* x, void* out, uint8_t* idx, int N, int H, int W, int C, int ceil_mode, void* stream) {
    MDE_REQUIRE(x && out && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "mde_maxpool_fwd: bad argument (C %% 8 == 0)");
    MDE_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)idx % 8) == 0, "mde_maxpool_fwd: alignment");
    const int OH = maxpool_out(H, ceil_mode), OW = maxpool_out(W, ceil_mode);
    maxpool_fwd_k<<<grid_for((int64_t)N * OH * OW * (C / 8)), NT, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, (bf16_t*)out, idx, N, H, W, C, OH, OW);
    MDE_LAUNCH_CHECK("maxpool_fwd_k");
    return MDE_OK;
}

extern "C" int mde_maxpool_bwd2(const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C, int ceil_mode, void* stream) {
    MDE_REQUIRE(dout && dx && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "mde_maxpool_bwd: bad argument (C %% 8 == 0)");
    MDE_REQUIRE(((uintptr_t)dout % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)idx % 8) == 0, "mde_maxpool_bwd: alignment");
    const int OH = maxpool_out(H, ceil_mode), OW = maxpool_out(W, ceil_mode);
    maxpool_bwd_k<<<grid_for((int64_t)N * H * W * (C / 8)), NT, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dout, idx, (bf16_t*)dx, N, H, W, C, OH, OW);
    MDE_LAUNCH_CHECK("maxpool_bwd_k");
    return MDE_OK;
}

static int pool_view_check(const char* who, const void* x, int ldx, int wpitch, int64_t ipitch, int Hv, int Wv, int N, int C, int k, int s) {
    MDE_REQUIRE(x && N > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldx >= C && ((uintptr_t)x % 16) == 0,
                "%s: channels (C=%d, ld=%d) must be multiples of 8 on a 16-byte aligned base", who, C, ldx);
    MDE_REQUIRE(k >= 1 && k <= 15 && s >= 1 && Hv >= k && Wv >= k && wpitch >= Wv && ipitch >= (int64_t)(Hv - 1) * wpitch + Wv,
                "%s: window %d / stride %d over a %d x %d view (row pitch %d)", who, k, s, Hv, Wv, wpitch);
    return MDE_OK;
}

extern "C" int mde_maxpool_view_fwd(const void* x, int ldx, int wpitch, int64_t ipitch, int Hv, int Wv, void* out, int ldo, uint8_t* idx,
                                    int N, int C, int k, int s, void* stream) {
    if (int rc = pool_view_check("mde_maxpool_view_fwd", x, ldx, wpitch, ipitch, Hv, Wv, N, C, k, s)) return rc;
    MDE_REQUIRE(out && idx && ldo % 8 == 0 && ldo >= C && ((uintptr_t)out % 16) == 0 && ((uintptr_t)idx % 8) == 0,
                "mde_maxpool_view_fwd: output alignment");
    const int OH = (Hv - k) / s + 1, OW = (Wv - k) / s + 1;
    maxpool_g_fwd_k<<<grid_for((int64_t)N * OH * OW * (C / 8)), NT, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, PoolView{ldx, wpitch, ipitch, Hv, Wv}, (bf16_t*)out, ldo, idx, N, C, OH, OW, k, s);
    MDE_LAUNCH_CHECK("maxpool_g_fwd_k");
    return MDE_OK;
}

extern "C" int mde_maxpool_view_bwd(const void* dout, int ldd, const uint8_t* idx, void* dx, int lddx, int wpitch, int64_t ipitch, int Hv, int Wv,
                                    int N, int C, int k, int s, int accumulate, void* stream) {
    if (int rc = pool_view_check("mde_maxpool_view_bwd", dx, lddx, wpitch, ipitch, Hv, Wv, N, C, k, s)) return rc;
    MDE_REQUIRE(dout && idx && ldd % 8 == 0 && ldd >= C && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)idx % 8) == 0,
                "mde_maxpool_view_bwd: gradient alignment");
    const int OH = (Hv - k) / s + 1, OW = (Wv - k) / s + 1;
    maxpool_g_bwd_k<<<grid_for((int64_t)N * Hv * Wv * (C / 8)), NT, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dout, ldd, idx, (bf16_t*)dx, PoolView{lddx, wpitch, ipitch, Hv, Wv}, N, C, OH, OW, k, s, accumulate);
    MDE_LAUNCH_CHECK("maxpool_g_bwd_k");
    return MDE_OK;
}

extern "C" int mde_maxpool_fwd(const void* x, void* out, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    return mde_maxpool_fwd2(x, out, idx, N, H, W, C, 0, stream);
}

extern "C" int mde_maxpool_bwd(const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C, void* stream) {
    return mde_maxpool_bwd2(dout, idx, dx, N, H, W, C, 0, stream);
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

extern "C" int mde_upsample_sigmoid_fwd(const float* x, float* out, int N, int H, int W, int C, int OH, int OW,
                                        void* stream) {
    MDE_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "mde_upsample_sigmoid_fwd: bad argument");
    upsig_fwd_k<<<grid_for((int64_t)N * C * OH * OW), NT, 0, (hipStream_t)stream>>>(x, out, N, H, W, C, OH, OW,
                                                                                  ac_scale(H, OH), ac_scale(W, OW));
    MDE_LAUNCH_CHECK("upsig_fwd_k");
    return MDE_OK;
}

extern "C" int mde_upsample_sigmoid_bwd(const float* dout, const float* out, float* dx, int N, int H, int W, int C,
                                        int OH, int OW, void* stream) {
    MDE_REQUIRE(dout && out && dx && N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "mde_upsample_sigmoid_bwd: bad argument");
    const float sh = ac_scale(H, OH), sw = ac_scale(W, OW);
    // inverse scales only bound the candidate window; a degenerate scale (out size 1) scans everything
    const float ish = sh > 0.f ? 1.f / sh : (float)OH, isw = sw > 0.f ? 1.f / sw : (float)OW;
    upsig_bwd_k<<<grid_for((int64_t)N * H * W * C), NT, 0, (hipStream_t)stream>>>(dout, out, dx, N, H, W, C, OH, OW, sh,
                                                                                sw, ish, isw);
    MDE_LAUNCH_CHECK("upsig_bwd_k");
    return MDE_OK;
}

extern "C" int mde_nchw_to_nhwc_bf16(const float* src, void* dst, int N, int C, int H, int W, void* stream) {
    MDE_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "mde_nchw_to_nhwc_bf16: bad argument");
    nchw2nhwc_k<<<grid_for((int64_t)N * C * H * W), NT, 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, N, C, H, W);
    MDE_LAUNCH_CHECK("nchw2nhwc_k");
    return MDE_OK;
}

extern "C" int mde_nchw_to_nhwc_bf16_pad(const float* src, void* dst, int N, int C, int H, int W, int Cpad, void* stream) {
    MDE_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "mde_nchw_to_nhwc_bf16_pad: bad argument");
    nchw2nhwc_pad_k<<<grid_for((int64_t)N * Cpad * H * W), NT, 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, N, C, H, W, Cpad);
    MDE_LAUNCH_CHECK("nchw2nhwc_pad_k");
    return MDE_OK;
}

extern "C" int mde_nchw_to_nhwc_split16(const float* src, void* dst, int N, int C, int H, int W, void* stream) {
    MDE_REQUIRE(src && dst && N > 0 && C > 0 && 3 * C <= 16 && H > 0 && W > 0 && ((uintptr_t)dst % 16) == 0, "mde_nchw_to_nhwc_split16: bad argument (C=%d)", C);
    nchw2nhwc_split16_k<<<grid_for((int64_t)N * H * W), NT, 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, N, C, H, W);
    MDE_LAUNCH_CHECK("nchw2nhwc_split16_k");
    return MDE_OK;
}

extern "C" int mde_stem_weight_split16(const float* src, void* dst, int64_t rows, int Cp, int C, void* stream) {
    MDE_REQUIRE(src && dst && rows > 0 && C > 0 && 3 * C <= 16 && Cp >= C && ((uintptr_t)dst % 16) == 0, "mde_stem_weight_split16: bad argument (C=%d)", C);
    stem_weight_split16_k<<<grid_for(rows), NT, 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, rows, Cp, C);
    MDE_LAUNCH_CHECK("stem_weight_split16_k");
    return MDE_OK;
}

extern "C" int mde_nhwc_bf16_to_nchw(const void* src, float* dst, int N, int C, int H, int W, void* stream) {
    MDE_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "mde_nhwc_bf16_to_nchw: bad argument");
    nhwc2nchw_k<<<grid_for((int64_t)N * C * H * W), NT, 0, (hipStream_t)stream>>>((const bf16_t*)src, dst, N, C, H, W);
    MDE_LAUNCH_CHECK("nhwc2nchw_k");
    return MDE_OK;
}
