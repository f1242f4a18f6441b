// Shared device/host helpers for libmde_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mde_hip.h"

// The 16-bit storage type of activations, their gradients and the GEMM weight shadows.  Default build (libmde_hip.so): bf16 --
// hence the names.  -DMDE_ACT_F16 (libmde_hip_f16.so, build.sh): IEEE half, the storage type BASELINE configuration 5 names
// (reference train.py:139-140, precision=16): 11 significand bits instead of 8, range 6e-5 ... 65504 -- the caller scales the
// loss as the reference's GradScaler does.  Same kernels: every conversion goes through (float) / (bf16_t), the MFMA and dot
// instructions of the two types have the same shapes and rates (MDE_MFMA_16x16x32, MDE_FDOT2), the LDS-DMA and transposed LDS
// reads move 16-bit words of either.  fp32 everywhere a sum crosses pixels (BatchNorm statistics, weight gradients, losses).
#ifdef MDE_ACT_F16
typedef _Float16 bf16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2_t;
#define MDE_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define MDE_FDOT2(a, b, c) __builtin_amdgcn_fdot2(a, b, c, false)
#define MDE_ACT_DTYPE_CODE 1
#else
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#define MDE_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MDE_FDOT2(a, b, c) __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false)
#define MDE_ACT_DTYPE_CODE 0
#endif
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(2))) int i32x2_t;

#define MDE_WAVE 64
#define MDE_STAT_SLOTS 32       // rows of a BatchNorm partial-sum buffer (mde_stat_slots())
#define MDE_OOB_OFFSET 0x80000000u  // any voffset >= num_records reads as zero through a raw buffer

// ---------------------------------------------------------------- error plumbing (host)
void mde_set_error(const char* fmt, ...);
int mde_check_hip(hipError_t e, const char* what);

#define MDE_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            mde_set_error(__VA_ARGS__);   \
            return MDE_EINVAL;            \
        }                                 \
    } while (0)

#define MDE_LAUNCH_CHECK(what)                                     \
    do {                                                           \
        int _rc = mde_check_hip(hipGetLastError(), what);          \
        if (_rc) return _rc;                                       \
    } while (0)

static inline int mde_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- deterministic mode (mde_set_deterministic)
// Every cross-workgroup floating-point sum (BatchNorm partial sums, split-K weight gradients, bias gradients) is an
// atomic add whose order changes from run to run; a last-bit difference flips a bf16 rounding and a deep net amplifies it.
// In deterministic mode each addend is split EXACTLY into two integers (v * 2^20 rounded, and the remainder * 2^60) that
// are added with 64-bit integer atomics: integer addition is associative, so the sum does not depend on the order.
struct MdeDet {
    int on;
    float* gbase;          // the flat fp32 gradient buffer weight / bias gradients are accumulated into ...
    long long* scratch;    // ... and its integer shadow: 2 x int64 per element, zero between backward passes
    long long n;           // elements of gbase
};
extern MdeDet g_mde_det;
struct MdeDetDev {         // kernel argument: scratch == nullptr -> plain float atomics
    float* gbase;
    long long* scratch;
};
static inline MdeDetDev mde_det_dev() { return g_mde_det.on ? MdeDetDev{g_mde_det.gbase, g_mde_det.scratch} : MdeDetDev{nullptr, nullptr}; }
// In deterministic mode a gradient destination is addressed as scratch + 2 * (dst - gbase): every entry point that hands
// mde_det_dev() to a kernel first checks that [dst, dst + n) lies inside the registered buffer (a destination outside it
// would turn into out-of-bounds 64-bit atomics).  dst == nullptr (gradient not requested) passes.
static inline bool mde_det_in_range(const float* dst, int64_t n) {
    return !g_mde_det.on || !dst || (dst >= g_mde_det.gbase && dst + n <= g_mde_det.gbase + g_mde_det.n);
}
#define MDE_DET_REQUIRE(who, dst, n)                    \
    MDE_REQUIRE(mde_det_in_range((dst), (n)),           \
                "%s: deterministic mode is on and " #dst " lies outside the registered gradient buffer", who)
#define MDE_DET_SLOTS 8    // a partial-sum buffer [MDE_STAT_SLOTS][2][C] floats holds [MDE_DET_SLOTS][2][C][2] int64 in this mode

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mde_rsrc(const void* p, uint32_t bytes) {
    // raw buffer, stride 0: loads whose byte offset is >= bytes return 0
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// Sum over the 16 lanes of a DPP row (lanes sharing lane>>4), result in every lane of the row.
// row_ror rotations are plain VALU DPP modifiers; __shfl_xor would go through the LDS crossbar
// (ds_bpermute_b32), which cost ~6 us per 256x256 tile in the conv epilogue.
__device__ __forceinline__ float mde_row16_sum(float v) {
    // (the empty asm keeps each add scalar: paired up by the SLP vectorizer they become
    //  v_mov + 2 x v_mov_b32_dpp + v_pk_add_f32 instead of two fused v_add_f32_dpp)
#define MDE_DPP_ADD(ctrl)                                                                                           \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false)); \
    asm("" : "+v"(v));
    MDE_DPP_ADD(0x128)  // row_ror:8
    MDE_DPP_ADD(0x124)  // row_ror:4
    MDE_DPP_ADD(0x122)  // row_ror:2
    MDE_DPP_ADD(0x121)  // row_ror:1
#undef MDE_DPP_ADD
    return v;
}

// A non-finite addend has no integer image (the conversion would give 0 or INT64_MIN and the sum would come out finite):
// it POISONS the slot instead -- the low word is overwritten with INT64_MAX, which no sum of remainders can bring back
// under 2^61 (each remainder is < 2^39) -- and mde_det_value reads a poisoned slot as NaN, so divergence still shows.
// (The split is exact for normal floats; for float denormals r * 2^60 is rounded: still order-independent.)
#define MDE_DET_POISON 0x7FFFFFFFFFFFFFFFll
__device__ __forceinline__ void mde_det_add2(long long* p, float v) {
    if (!(fabsf(v) <= 3.4028234663852886e38f)) {
        atomicExch(reinterpret_cast<unsigned long long*>(p + 1), (unsigned long long)MDE_DET_POISON);
        return;
    }
    const double vd = (double)v;
    const long long a = __double2ll_rn(vd * 1048576.0);
    const double r = vd - (double)a * (1.0 / 1048576.0);
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)a);
    atomicAdd(reinterpret_cast<unsigned long long*>(p + 1), (unsigned long long)__double2ll_rn(r * 1152921504606846976.0));
}
__device__ __forceinline__ double mde_det_value(const long long* p) {
    if (p[1] >= (1ll << 61) || p[1] <= -(1ll << 61)) return __builtin_nan("");
    return (double)p[0] * (1.0 / 1048576.0) + (double)p[1] * (1.0 / 1152921504606846976.0);
}
// one addend of a BatchNorm partial-sum buffer `part` (logical [slot][2][C]); wg picks the slot
__device__ __forceinline__ void mde_stat_add(float* part, int C, uint32_t wg, int which, int c, float v, int det) {
    if (!det) atomicAdd(part + ((size_t)(wg % MDE_STAT_SLOTS) * 2 + which) * C + c, v);
    else mde_det_add2(reinterpret_cast<long long*>(part) + (((size_t)(wg % MDE_DET_SLOTS) * 2 + which) * C + c) * 2, v);
}
// sum over the slots of entry (which, c), leaving it zeroed.  All loads are issued before the first store: with the zeroing
// store inside the load loop the compiler must keep load k+1 behind store k (same array), and the 32 L2 round trips ran back
// to back — 8 us for a finalize kernel that does nothing else, 128 of them per step.
__device__ __forceinline__ double mde_stat_take(float* part, int C, int which, int c, int det) {
    double s = 0.0;
    if (!det) {
        float v[MDE_STAT_SLOTS];
#pragma unroll
        for (int k = 0; k < MDE_STAT_SLOTS; ++k) v[k] = part[((size_t)k * 2 + which) * C + c];
#pragma unroll
        for (int k = 0; k < MDE_STAT_SLOTS; ++k) s += (double)v[k];
#pragma unroll
        for (int k = 0; k < MDE_STAT_SLOTS; ++k) part[((size_t)k * 2 + which) * C + c] = 0.f;
    } else {
        long long v[MDE_DET_SLOTS][2];
#pragma unroll
        for (int k = 0; k < MDE_DET_SLOTS; ++k) {
            const long long* p = reinterpret_cast<const long long*>(part) + (((size_t)k * 2 + which) * C + c) * 2;
            v[k][0] = p[0];
            v[k][1] = p[1];
        }
#pragma unroll
        for (int k = 0; k < MDE_DET_SLOTS; ++k) s += mde_det_value(v[k]);
#pragma unroll
        for (int k = 0; k < MDE_DET_SLOTS; ++k) {
            long long* p = reinterpret_cast<long long*>(part) + (((size_t)k * 2 + which) * C + c) * 2;
            p[0] = 0;
            p[1] = 0;
        }
    }
    return s;
}
// one addend of a weight / bias gradient element
__device__ __forceinline__ void mde_grad_add(float* dst, float v, const MdeDetDev& dd) {
    if (!dd.scratch) atomicAdd(dst, v);
    else mde_det_add2(dd.scratch + 2 * (dst - dd.gbase), v);
}

__device__ __forceinline__ float mde_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double mde_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum of `v`; result valid in thread 0.  `scratch` >= blockDim/64 floats of LDS.
__device__ __forceinline__ float mde_block_sum(float v, float* scratch) {
    v = mde_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += scratch[i];
    __syncthreads();
    return r;
}

// floor(m / d) for m < 2^32 with inv = floor(2^32 / d) (one correction step; d >= 2).
// For d == 1 pass inv = 0xFFFFFFFF.
__device__ __forceinline__ uint32_t mde_fastdiv(uint32_t m, uint32_t d, uint32_t inv) {
    uint32_t q = __umulhi(m, inv);
    uint32_t r = m - q * d;
    return q + (r >= d ? 1u : 0u);
}

// XCD-aware block remap (bijective for any grid size): blocks b, b+8, ... share an XCD under
// round-robin dispatch, so give each of the 8 groups a contiguous run of the tile order.
// Speed only, never correctness (guide T1).
__device__ __forceinline__ uint32_t mde_xcd_remap(uint32_t bid, uint32_t nblk) {
    const uint32_t q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
    const uint32_t start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return start + i;
}
#endif
