// The reference's input pipeline on the device (modules/base_module.py:234-284): Pillow's 8-bit arithmetic, bit for bit.
// Images are uint8 H x W x C (PIL's layout; the D depth layers of a sample travel as ONE D-channel image: every operation
// here is per channel).  Integer work, HBM-bound on images of a few hundred KB: one thread per output element.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mde_hip.h"
#include "mde_common.h"

namespace {

constexpr int NT = 256;
constexpr int PRECISION_BITS = 32 - 8 - 2;      // Pillow src/libImaging/Resample.c

int grid_for(int64_t total) {
    const int64_t g = (total + NT - 1) / NT;
    return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

// functional.to_pil_image of a float tensor: pic.mul(255).byte(), C x H x W -> H x W x C; `divisor`: the depth / s of
// base_module.py:236 in front of it (a float32 division, correctly rounded as torch's)
__global__ __launch_bounds__(NT) void aug_to_u8_k(const float* __restrict__ src, int C, int64_t HW, float divisor, uint8_t* __restrict__ dst) {
    const int64_t total = HW * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t p = i / C;
        const int c = (int)(i - p * C);
        float v = src[(int64_t)c * HW + p];
        if (divisor != 1.0f) v = __fdiv_rn(v, divisor);
        dst[i] = (uint8_t)(int)__fmul_rn(v, 255.0f);
    }
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// ImagingResampleHorizontal_8bpc over source rows [y0, y0 + rows): out[y][xx][c] = clip8(2^21 + sum_x in[y0 + y][xmin + x][c] * k[xx][x])
__global__ __launch_bounds__(NT) void aug_resample_h_k(const uint8_t* __restrict__ src, int W, int C, int y0, int rows,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int OW,
                                                       uint8_t* __restrict__ dst) {
    const int64_t total = (int64_t)rows * OW * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        int64_t q = i / C;
        const int xx = (int)(q % OW);
        const int y = (int)(q / OW);
        const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
        const uint8_t* row = src + ((int64_t)(y0 + y) * W + xmin) * C + c;
        const int* k = kk + (int64_t)xx * ksize;
        int ss = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < n; ++x) ss += (int)row[(int64_t)x * C] * k[x];
        dst[i] = clip8(ss);
    }
}

// ImagingResampleVertical_8bpc: src rows are offset by y0 (the horizontal pass kept only the rows this one reads)
__global__ __launch_bounds__(NT) void aug_resample_v_k(const uint8_t* __restrict__ src, int W, int C, int y0,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int OH,
                                                       uint8_t* __restrict__ dst) {
    const int64_t rowlen = (int64_t)W * C, total = (int64_t)OH * rowlen;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int yy = (int)(i / rowlen);
        const int64_t r = i - (int64_t)yy * rowlen;
        const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
        const uint8_t* col = src + (int64_t)(ymin - y0) * rowlen + r;
        const int* k = kk + (int64_t)yy * ksize;
        int ss = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < n; ++y) ss += (int)col[(int64_t)y * rowlen] * k[y];
        dst[i] = clip8(ss);
    }
}

// Geometry.c affine_fixed (nearest neighbour, 16.16 fixed point, fill 0): xin = (a2 + a1 * y + a0 * x) >> 16
__global__ __launch_bounds__(NT) void aug_affine_nearest_k(const uint8_t* __restrict__ src, int H, int W, int C, int a0, int a1, int a2,
                                                           int a3, int a4, int a5, uint8_t* __restrict__ dst) {
    const int64_t total = (int64_t)H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        int64_t q = i / C;
        const int x = (int)(q % W);
        const int y = (int)(q / W);
        const long long xin = ((long long)a2 + (long long)a1 * y + (long long)a0 * x) >> 16;
        const long long yin = ((long long)a5 + (long long)a4 * y + (long long)a3 * x) >> 16;
        uint8_t v = 0;
        if (xin >= 0 && xin < W && yin >= 0 && yin < H) v = src[((int64_t)yin * W + xin) * C + c];
        dst[i] = v;
    }
}

// CenterCrop -> hflip -> np.array(img, float32) / 255.0 -> to_tensor: H x W x C uint8 -> C x oh x ow float32 through a
// 256-entry table of the quotients (computed on the host by numpy itself)
__global__ __launch_bounds__(NT) void aug_crop_flip_k(const uint8_t* __restrict__ src, int W, int C, int top, int left, int oh, int ow,
                                                      int flip, const float* __restrict__ lut, float* __restrict__ dst, int lut_stride = 0) {
    const int64_t total = (int64_t)C * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % ow);
        int64_t q = i / ow;
        const int y = (int)(q % oh);
        const int c = (int)(q / oh);
        const int sx = left + (flip ? ow - 1 - x : x);
        dst[i] = lut[c * lut_stride + src[((int64_t)(top + y) * W + sx) * C + c]];
    }
}

// flip (columns) -> constant pad -> crop of an H x W x C image of ELEM-byte elements, as numpy does it (modules/vnl.py:59-78:
// np.flip(img, axis=1), np.pad(..., 'constant'), the slice): pure data movement.  Output pixel (y, x) is padded-image pixel
// (cy + y, cx + x); the padded image has `pt` rows / `pl` columns of `fill` in front of the (flipped) image.
template <typename T>
__global__ __launch_bounds__(NT) void aug_flip_pad_crop_k(const T* __restrict__ src, int H, int W, int C, int flip, int pt, int pl, int cy, int cx,
                                                          int oh, int ow, T fill, T* __restrict__ dst) {
    const int64_t total = (int64_t)oh * ow * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        const int64_t q = i / C;
        const int x = (int)(q % ow), y = (int)(q / ow);
        const int sy = cy + y - pt;
        int sx = cx + x - pl;
        const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
        if (flip) sx = W - 1 - sx;
        dst[i] = in ? src[((int64_t)sy * W + sx) * C + c] : fill;
    }
}

}  // namespace

extern "C" int mde_aug_to_u8(const float* src, int C, int H, int W, float divisor, uint8_t* dst, void* stream) {
    MDE_REQUIRE(src && dst && C > 0 && H > 0 && W > 0 && divisor > 0.f, "mde_aug_to_u8: bad argument");
    aug_to_u8_k<<<grid_for((int64_t)H * W * C), NT, 0, (hipStream_t)stream>>>(src, C, (int64_t)H * W, divisor, dst);
    MDE_LAUNCH_CHECK("aug_to_u8_k");
    return MDE_OK;
}

extern "C" int mde_aug_resample_u8(const uint8_t* src, int H, int W, int C, const int32_t* hbounds, const int32_t* hk, int hksize, int OW,
                                   const int32_t* vbounds, const int32_t* vk, int vksize, int OH, int y0, int rows, uint8_t* tmp,
                                   uint8_t* dst, void* stream) {
    MDE_REQUIRE(src && dst && H > 0 && W > 0 && C > 0 && OW > 0 && OH > 0, "mde_aug_resample_u8: bad argument");
    MDE_REQUIRE((hbounds != nullptr) == (hk != nullptr) && (vbounds != nullptr) == (vk != nullptr) && (hbounds || vbounds),
                "mde_aug_resample_u8: coefficient tables come in pairs, at least one axis");
    MDE_REQUIRE((hbounds || OW == W) && (vbounds || OH == H), "mde_aug_resample_u8: an axis without coefficients keeps its size");
    MDE_REQUIRE(y0 >= 0 && rows > 0 && y0 + rows <= H && (!hbounds || !vbounds || tmp), "mde_aug_resample_u8: row window / scratch");
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* vin = src;
    int vy0 = 0;
    if (hbounds) {
        uint8_t* hout = vbounds ? tmp : dst;
        MDE_REQUIRE(vbounds || (y0 == 0 && rows == H), "mde_aug_resample_u8: a horizontal-only pass covers every row");
        aug_resample_h_k<<<grid_for((int64_t)rows * OW * C), NT, 0, st>>>(src, W, C, y0, rows, hbounds, hk, hksize, OW, hout);
        MDE_LAUNCH_CHECK("aug_resample_h_k");
        vin = hout;
        vy0 = y0;
    }
    if (vbounds) {
        aug_resample_v_k<<<grid_for((int64_t)OH * OW * C), NT, 0, st>>>(vin, OW, C, vy0, vbounds, vk, vksize, OH, dst);
        MDE_LAUNCH_CHECK("aug_resample_v_k");
    }
    return MDE_OK;
}

extern "C" int mde_aug_affine_nearest_u8(const uint8_t* src, int H, int W, int C, const int32_t* coef, uint8_t* dst, void* stream) {
    MDE_REQUIRE(src && dst && coef && H > 0 && W > 0 && C > 0, "mde_aug_affine_nearest_u8: bad argument");
    aug_affine_nearest_k<<<grid_for((int64_t)H * W * C), NT, 0, (hipStream_t)stream>>>(src, H, W, C, coef[0], coef[1], coef[2], coef[3], coef[4],
                                                                                      coef[5], dst);
    MDE_LAUNCH_CHECK("aug_affine_nearest_k");
    return MDE_OK;
}

extern "C" int mde_aug_crop_flip_to_float_c(const uint8_t* src, int H, int W, int C, int top, int left, int oh, int ow, int flip,
                                            const float* lut, int lut_stride, float* dst, void* stream) {
    MDE_REQUIRE(src && dst && lut && H > 0 && W > 0 && C > 0 && oh > 0 && ow > 0 && top >= 0 && left >= 0 && top + oh <= H && left + ow <= W &&
                    lut_stride >= 0,
                "mde_aug_crop_flip_to_float_c: crop %dx%d at (%d, %d) of a %dx%d image", oh, ow, top, left, H, W);
    aug_crop_flip_k<<<grid_for((int64_t)C * oh * ow), NT, 0, (hipStream_t)stream>>>(src, W, C, top, left, oh, ow, flip, lut, dst, lut_stride);
    MDE_LAUNCH_CHECK("aug_crop_flip_k");
    return MDE_OK;
}

extern "C" int mde_aug_flip_pad_crop(const void* src, int elem_bytes, int H, int W, int C, int flip, int pad_top, int pad_left, int crop_y,
                                     int crop_x, int oh, int ow, const void* fill, void* dst, void* stream) {
    MDE_REQUIRE(src && dst && fill && (elem_bytes == 1 || elem_bytes == 4) && H > 0 && W > 0 && C > 0 && oh > 0 && ow > 0 && pad_top >= 0 &&
                    pad_left >= 0 && crop_y >= 0 && crop_x >= 0 && crop_y + oh <= H + pad_top && crop_x + ow <= W + pad_left,
                "mde_aug_flip_pad_crop: bad argument (the crop must lie inside the padded image)");
    const int g = grid_for((int64_t)oh * ow * C);
    if (elem_bytes == 1)
        aug_flip_pad_crop_k<uint8_t><<<g, NT, 0, (hipStream_t)stream>>>((const uint8_t*)src, H, W, C, flip, pad_top, pad_left, crop_y, crop_x, oh, ow,
                                                                         *(const uint8_t*)fill, (uint8_t*)dst);
    else
        aug_flip_pad_crop_k<uint32_t><<<g, NT, 0, (hipStream_t)stream>>>((const uint32_t*)src, H, W, C, flip, pad_top, pad_left, crop_y, crop_x, oh, ow,
                                                                          *(const uint32_t*)fill, (uint32_t*)dst);
    MDE_LAUNCH_CHECK("aug_flip_pad_crop_k");
    return MDE_OK;
}

extern "C" int mde_aug_crop_flip_to_float(const uint8_t* src, int H, int W, int C, int top, int left, int oh, int ow, int flip,
                                          const float* lut, float* dst, void* stream) {
    MDE_REQUIRE(src && dst && lut && H > 0 && W > 0 && C > 0 && oh > 0 && ow > 0 && top >= 0 && left >= 0 && top + oh <= H && left + ow <= W,
                "mde_aug_crop_flip_to_float: crop %dx%d at (%d, %d) of a %dx%d image", oh, ow, top, left, H, W);
    aug_crop_flip_k<<<grid_for((int64_t)C * oh * ow), NT, 0, (hipStream_t)stream>>>(src, W, C, top, left, oh, ow, flip, lut, dst);
    MDE_LAUNCH_CHECK("aug_crop_flip_k");
    return MDE_OK;
}
