// Depthwise 3x3 convolution (groups == channels) for NHWC 16-bit activations on gfx950: forward, input gradient, weight gradient.
// Replaces nn.Conv2d(hidden, hidden, 3, stride, groups=hidden, padding=dilation, dilation=dilation, bias=False) of the reference's
// MobileNetV2 encoder option (network/VNL.py:427-444 InvertedResidual, :389-397 the stride-8 body with dilations 2 / 4).
//
// Nine multiply-adds per output element: an HBM-bound streaming pass, nothing for the matrix cores (a block-diagonal GEMM tile
// at group size 1 would waste 63 of 64 of the MFMA).  As in bn.hip a thread owns a fixed 8-channel column (16-byte accesses)
// and keeps that column's 9 x 8 fp32 weights in registers; the nine taps of neighbouring pixels are cache hits.  Weights and their
// gradient are fp32 [C][9] -- the flat master / gradient slices themselves, no 16-bit shadow.
#include "mde_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ void ld8f(const bf16_t* p, float (&v)[8]) {
    const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}

// DGRAD = false: out[n, oy, ox, c] = sum_ij x[n, oy s + (i - 1) d, ox s + (j - 1) d, c] w[c][3 i + j]         (grid: OH x OW)
// DGRAD = true : dx[n, y, x, c]    = sum_ij dy[n, (y - (i - 1) d) / s, (x - (j - 1) d) / s, c] w[c][3 i + j]  where divisible, in range
//                (grid: the INPUT's H x W; src = dy of size SH x SW)
template <bool DGRAD>
__global__ __launch_bounds__(NT) void dw3x3_k(const bf16_t* __restrict__ src, int lds, const float* __restrict__ w, bf16_t* __restrict__ dst,
                                              int ldd, int N, int SH, int SW, int DH, int DW, int C, int s, int d, int accumulate) {
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    if (rl >= rpb) return;
    float wt[9][8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t][e] = w[(col * 8 + e) * 9 + t];
    const int64_t M = (int64_t)N * DH * DW;
    for (int64_t row = (int64_t)blockIdx.x * rpb + rl; row < M; row += (int64_t)gridDim.x * rpb) {
        const int x0 = (int)(row % DW);
        const int64_t q = row / DW;
        const int y0 = (int)(q % DH), n = (int)(q / DH);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int sy;
            if (!DGRAD) {
                sy = y0 * s + (i - 1) * d;
            } else {
                const int t = y0 - (i - 1) * d;
                if (t < 0 || t % s) continue;
                sy = t / s;
            }
            if (sy < 0 || sy >= SH) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int sx;
                if (!DGRAD) {
                    sx = x0 * s + (j - 1) * d;
                } else {
                    const int t = x0 - (j - 1) * d;
                    if (t < 0 || t % s) continue;
                    sx = t / s;
                }
                if (sx < 0 || sx >= SW) continue;
                float v[8];
                ld8f(src + (((int64_t)n * SH + sy) * SW + sx) * lds + col * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e] * wt[i * 3 + j][e];
            }
        }
        bf16_t* o = dst + row * ldd + col * 8;
        if (accumulate) {
            float old[8];
            ld8f(o, old);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += old[e];
        }
        bf16x8_t t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (bf16_t)acc[e];
        *reinterpret_cast<bf16x8_t*>(o) = t;
    }
}

// dw[c][3 i + j] += sum over output pixels of x[n, oy s + (i - 1) d, ox s + (j - 1) d, c] dy[n, oy, ox, c]: per-thread sums over the
// rows of its column (72 registers), combined over the workgroup's row lanes through LDS in a fixed order, one add per entry
__global__ __launch_bounds__(NT) void dw3x3_wgrad_k(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ dy, int ldy,
                                                    float* __restrict__ dw, int N, int H, int W, int OH, int OW, int C, int s, int d,
                                                    int rows_per_blk, MdeDetDev det) {
    extern __shared__ float sh[];                   // [NT][9] per channel element, processed one e at a time
    const int cpr = C >> 3, rpb = NT / cpr, col = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    const int64_t M = (int64_t)N * OH * OW;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk, r1 = min(M, r0 + rows_per_blk);
    for (int64_t row = rl < rpb ? r0 + rl : r1; row < r1; row += rpb) {
        const int ox = (int)(row % OW);
        const int64_t q = row / OW;
        const int oy = (int)(q % OH), n = (int)(q / OH);
        float g[8];
        ld8f(dy + row * ldy + col * 8, g);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int sy = oy * s + (i - 1) * d;
            if (sy < 0 || sy >= H) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int sx = ox * s + (j - 1) * d;
                if (sx < 0 || sx >= W) continue;
                float v[8];
                ld8f(x + (((int64_t)n * H + sy) * W + sx) * ldx + col * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i * 3 + j][e] += v[e] * g[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) sh[threadIdx.x * 9 + t] = acc[t][e];
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float sum = 0.f;
                for (int k = 0; k < rpb; ++k) sum += sh[(k * cpr + col) * 9 + t];
                if (sum != 0.f) mde_grad_add(dw + (col * 8 + e) * 9 + t, sum, det);
            }
        }
    }
}

int dw_check(const char* who, const void* a, int lda, const void* b, int ldb, int N, int H, int W, int C, int s, int d) {
    MDE_REQUIRE(a && b && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 2048 && s >= 1 && s <= 2 && d >= 1,
                "%s: bad argument (C=%d: a multiple of 8 up to 2048; stride %d: 1 or 2; dilation %d)", who, C, s, d);
    MDE_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= C && ldb >= C && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0,
                "%s: tensors are walked in 16-byte chunks (ld %% 8 == 0, 16-byte aligned)", who);
    return MDE_OK;
}
int dw_grid(int64_t M, int C) {
    const int rpb = NT / (C / 8);
    const int64_t nb = (M + rpb - 1) / rpb;
    return (int)(nb > 256 * 8 ? 256 * 8 : (nb < 1 ? 1 : nb));
}

}  // namespace

extern "C" int mde_dwconv3x3_fwd(const void* x, int ldx, const float* w, void* out, int ldo, int N, int H, int W, int C, int stride,
                                 int dilation, void* stream) {
    if (int rc = dw_check("mde_dwconv3x3_fwd", x, ldx, out, ldo, N, H, W, C, stride, dilation)) return rc;
    MDE_REQUIRE(w, "mde_dwconv3x3_fwd: null weights");
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;            // padding == dilation: (n + 2 d - 2 d - 1) / s + 1
    dw3x3_k<false><<<dw_grid((int64_t)N * OH * OW, C), NT, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, w, (bf16_t*)out, ldo, N, H, W, OH, OW,
                                                                                     C, stride, dilation, 0);
    MDE_LAUNCH_CHECK("dw3x3_k<fwd>");
    return MDE_OK;
}

extern "C" int mde_dwconv3x3_dgrad(const void* dy, int ldy, const float* w, void* dx, int lddx, int N, int H, int W, int C, int stride,
                                   int dilation, int accumulate, void* stream) {
    if (int rc = dw_check("mde_dwconv3x3_dgrad", dy, ldy, dx, lddx, N, H, W, C, stride, dilation)) return rc;
    MDE_REQUIRE(w, "mde_dwconv3x3_dgrad: null weights");
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    dw3x3_k<true><<<dw_grid((int64_t)N * H * W, C), NT, 0, (hipStream_t)stream>>>((const bf16_t*)dy, ldy, w, (bf16_t*)dx, lddx, N, OH, OW, H, W, C,
                                                                                  stride, dilation, accumulate);
    MDE_LAUNCH_CHECK("dw3x3_k<dgrad>");
    return MDE_OK;
}

extern "C" int mde_dwconv3x3_wgrad(const void* x, int ldx, const void* dy, int ldy, float* dw, int N, int H, int W, int C, int stride,
                                   int dilation, void* stream) {
    if (int rc = dw_check("mde_dwconv3x3_wgrad", x, ldx, dy, ldy, N, H, W, C, stride, dilation)) return rc;
    MDE_REQUIRE(dw, "mde_dwconv3x3_wgrad: null gradient");
    MDE_DET_REQUIRE("mde_dwconv3x3_wgrad", dw, (int64_t)C * 9);
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const int64_t M = (int64_t)N * OH * OW;
    const int rpb = NT / (C / 8);
    int64_t nb = (M + (int64_t)rpb * 32 - 1) / ((int64_t)rpb * 32);
    nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    int64_t rows = (M + nb - 1) / nb;
    rows = (rows + rpb - 1) / rpb * rpb;
    const int grid = (int)((M + rows - 1) / rows);
    dw3x3_wgrad_k<<<grid, NT, NT * 9 * sizeof(float), (hipStream_t)stream>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, dw, N, H, W, OH, OW, C,
                                                                             stride, dilation, (int)rows, mde_det_dev());
    MDE_LAUNCH_CHECK("dw3x3_wgrad_k");
    return MDE_OK;
}
