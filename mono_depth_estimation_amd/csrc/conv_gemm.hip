// Implicit-GEMM convolution on MFMA for gfx950 — the "NT" form used by conv forward,
// dgrad (stride-1 directly, strided as output phases) and the phase-decomposed FCRN
// up-projection (reference network/FCRN.py:31-44,170-198).
//
//   out[pix][col] = sum_t sum_c in[src(pix,t)][c] * w[col][wtap[t]][c]
//
// Data layout: activations NHWC bf16, weights [col][tap][c] bf16 (K contiguous for both
// operands, so each MFMA fragment is one 16-byte load per lane).  Accumulation fp32.
//
// Tiling (one workgroup = 256 threads = 4 waves, one 64x64 output sub-tile per wave):
//   BP pixels x BC columns per workgroup, BK = 64 contracted channels per K-step (always
//   inside one tap because C % 64 == 0).  (BP,BC) = (128,128): 2x2 waves of 64 px x 64 ch;
//   (128,64) for <=64-column layers: 4x1 waves of 32 px x 64 ch.
//   v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand (rows = output channels)
//   and the gathered pixels as B (cols = pixels): a lane then owns 4 consecutive channels
//   of one pixel, which makes the epilogue's LDS re-tiling a ds_write_b64.
//
// Global -> LDS: register staged (raw buffer loads; out-of-image taps and tile tails read
// as zero through the buffer bounds check, so padding costs no branches), written to a
// fragment-linear XOR-swizzled image: fragment (16 rows x 32 k) = 1 KiB, lane l reads
// 16 B at l*16 ^ swizzle -> conflict-free ds_read_b128 and conflict-free ds_write_b128.
// Two LDS buffers, one barrier per K-step, next step's loads in flight under the MFMAs.
//
// Epilogue: fp32 accumulators -> bf16 -> LDS [pixel][channel] -> full-line 16-byte stores
// (optionally read-modify-write for dgrad accumulation), plus optional BatchNorm partial
// sums (sum, sum of squares per channel, reduced per workgroup then added with fp32 atomics
// to one of MDE_STAT_SLOTS rows) so the BN statistics pass over the conv output is not needed.
#include "mde_common.h"

namespace {

constexpr int BK = 64;
constexpr int NT = 256;

struct KArgs {
    mde_conv_desc d;
    const void* in;
    const void* w;
    void* out;
    float* stats;
    uint32_t w_bytes;
    uint32_t out_bytes;
    int32_t M;        // N*GH*GW
    int32_t nP, nC;   // tiles along pixels / columns
    int32_t vec_ok;   // 16-byte stores allowed
};

// byte offset of the 16-byte chunk (row, kslot8) inside a swizzled tile
__device__ __forceinline__ int chunk_off(int row, int kslot8) {
    return ((((row >> 4) * 2 + (kslot8 >> 2)) * 64) + (kslot8 & 3) * 16 + ((row & 15) ^ kslot8)) * 16;
}

template <int BP, int BC>
__global__ __launch_bounds__(NT, 2) void conv_gemm_nt(const KArgs a) {
    constexpr int XP = BP / 32;          // X chunks per thread per K-step
    constexpr int WP = BC / 32;          // W chunks per thread per K-step
    constexpr int WAVES_C = BC / 64;     // waves along columns
    constexpr int WAVES_P = 4 / WAVES_C; // waves along pixels
    constexpr int PF = BP / WAVES_P / 16;// 16-pixel fragments per wave
    constexpr int XT_BYTES = BP * BK * 2;
    constexpr int WT_BYTES = BC * BK * 2;
    constexpr int BUF_BYTES = XT_BYTES + WT_BYTES;
    constexpr int ROWB = BC * 2 + 16;    // epilogue tile row pitch (bytes)
    static_assert(WAVES_P * WAVES_C == 4 && PF * 16 * WAVES_P == BP && (PF == 2 || PF == 4), "wave tiling");
    static_assert(BP * ROWB <= 2 * BUF_BYTES, "epilogue tile fits in the staging buffers");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_inbase = reinterpret_cast<int*>(smem + 2 * BUF_BYTES);
    int* s_yx = s_inbase + BP;
    int* s_out = s_yx + BP;
    float* s_stat = reinterpret_cast<float*>(s_out + BP);  // [WAVES_P][2][BC]

    const mde_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave % WAVES_C;       // wave position along columns
    const int wp = wave / WAVES_C;       // wave position along pixels

    const uint32_t bid = mde_xcd_remap(blockIdx.x, (uint32_t)(a.nP * a.nC));
    const int pi = bid % a.nP, ci = bid / a.nP;
    const int m0 = pi * BP, n0 = ci * BC;

    // ---- per-row decode, once per workgroup
    for (int r = tid; r < BP; r += NT) {
        const int m = m0 + r;
        int inbase = 0, yx = 0x7FFF7FFF, oo = -1;
        if (m < a.M) {
            const int gw = d.GW, ghw = d.GH * d.GW;
            const int n = m / ghw, rem = m - n * ghw;
            const int gy = rem / gw, gx = rem - gy * gw;
            const int iy0 = gy * d.sy, ix0 = gx * d.sx;
            inbase = ((n * d.H + iy0) * d.W + ix0) * d.ld_in;
            yx = (iy0 << 16) | ix0;
            oo = ((n * d.OH + gy * d.osy + d.ooy) * d.OW + gx * d.osx + d.oox) * d.ld_out;
        }
        s_inbase[r] = inbase;
        s_yx[r] = yx;
        s_out[r] = oo;
    }
    __syncthreads();

    const int kslot8 = tid & 7;
    const int lrow = tid >> 3;           // 0..31
    int x_base[XP], x_yx[XP];
#pragma unroll
    for (int p = 0; p < XP; ++p) {
        x_base[p] = s_inbase[p * 32 + lrow] + kslot8 * 8;
        x_yx[p] = s_yx[p * 32 + lrow];
    }
    const int wrow_len = d.wtaps_total * d.C;
    int w_base[WP];
#pragma unroll
    for (int p = 0; p < WP; ++p) w_base[p] = (n0 + p * 32 + lrow) * wrow_len + kslot8 * 8;

    const __amdgpu_buffer_rsrc_t rs_in = mde_rsrc(a.in, d.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = mde_rsrc(a.w, a.w_bytes);

    int st_off[XP > WP ? XP : WP];       // LDS chunk offsets of this thread's rows (same for X and W)
#pragma unroll
    for (int p = 0; p < (XP > WP ? XP : WP); ++p) st_off[p] = chunk_off(p * 32 + lrow, kslot8);

    // fragment read offsets: lane l reads row (l&15), k-slot (l>>4) of fragment (rb, kb)
    int rd_off[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int ks = lane >> 4, r = lane & 15;
        rd_off[kb] = (kb * 64 + ks * 16 + (r ^ (kb * 4 + ks))) * 16;
    }

    f32x4_t acc[4][PF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < PF; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int csteps = d.C / BK;
    const int nsteps = d.ntaps * csteps;
    i32x4_t xr[XP], wr[WP];

    auto issue_loads = [&](int tap, int cs) {
        const int tdy = d.dy[tap], tdx = d.dx[tap];
        const int c0 = cs * BK;
        const int tapoff = (tdy * d.W + tdx) * d.ld_in + c0;
        const int woff = d.wtap[tap] * d.C + c0;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int iy = (int)((uint32_t)x_yx[p] >> 16) + tdy;
            const int ix = (x_yx[p] & 0xFFFF) + tdx;
            const bool ok = ((uint32_t)iy < (uint32_t)d.H) & ((uint32_t)ix < (uint32_t)d.W);
            const uint32_t off = ok ? (uint32_t)(x_base[p] + tapoff) * 2u : MDE_OOB_OFFSET;
            xr[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < WP; ++p)
            wr[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (uint32_t)(w_base[p] + woff) * 2u, 0, 0);
    };
    auto stage_write = [&](int buf) {
        char* xb = smem + buf * BUF_BYTES;
        char* wb = xb + XT_BYTES;
#pragma unroll
        for (int p = 0; p < XP; ++p) *reinterpret_cast<i32x4_t*>(xb + st_off[p]) = xr[p];
#pragma unroll
        for (int p = 0; p < WP; ++p) *reinterpret_cast<i32x4_t*>(wb + st_off[p]) = wr[p];
    };

    issue_loads(0, 0);
    stage_write(0);
    __syncthreads();

    int tap = 0, cs = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (++cs == csteps) { cs = 0; ++tap; }
        const bool more = s + 1 < nsteps;
        if (more) issue_loads(tap, cs);

        const char* xb = smem + buf * BUF_BYTES + wp * (PF * 2048);
        const char* wb = smem + buf * BUF_BYTES + XT_BYTES + wc * (4 * 2048);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8_t fa[4], fb[PF];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[i] = *reinterpret_cast<const bf16x8_t*>(wb + i * 2048 + rd_off[kb]);
#pragma unroll
            for (int j = 0; j < PF; ++j)
                fb[j] = *reinterpret_cast<const bf16x8_t*>(xb + j * 2048 + rd_off[kb]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) stage_write(buf ^ 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    // (the loop's last barrier guarantees every wave is done reading the staging tiles)
    {
        const int prow = wp * (PF * 16) + (lane & 15);
        const int chb = wc * 64 + (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                bf16x4_t v;
                v[0] = (bf16_t)acc[i][j][0];
                v[1] = (bf16_t)acc[i][j][1];
                v[2] = (bf16_t)acc[i][j][2];
                v[3] = (bf16_t)acc[i][j][3];
                *reinterpret_cast<bf16x4_t*>(smem + (prow + j * 16) * ROWB + (chb + i * 16) * 2) = v;
            }
    }
    if (a.stats) {
        // per-wave channel sums over its pixels (rows beyond M are exact zeros)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const float v = acc[i][j][r];
                    s1 += v;
                    s2 += v * v;
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if ((lane & 15) == 0) {
                    const int ch = wc * 64 + i * 16 + (lane >> 4) * 4 + r;
                    s_stat[(wp * 2 + 0) * BC + ch] = s1;
                    s_stat[(wp * 2 + 1) * BC + ch] = s2;
                }
            }
    }
    __syncthreads();
    if (a.stats) {
        for (int e = tid; e < 2 * BC; e += NT) {
            const int which = e / BC, ch = e - which * BC;
            if (n0 + ch < d.ncols) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < WAVES_P; ++q) s += s_stat[(q * 2 + which) * BC + ch];
                atomicAdd(a.stats + ((size_t)(pi % MDE_STAT_SLOTS) * 2 + which) * d.ncols + n0 + ch, s);
            }
        }
    }
    {
        constexpr int CPR = BC / 8;          // 16-byte chunks per pixel row
        constexpr int RPP = NT / CPR;        // rows per pass
        const int chunk = tid % CPR, r0 = tid / CPR;
        const int col = n0 + chunk * 8;
        bf16_t* outp = reinterpret_cast<bf16_t*>(a.out);
        if (col < d.ncols) {
            const bool full = a.vec_ok && (col + 8 <= d.ncols);
#pragma unroll 4
            for (int r = r0; r < BP; r += RPP) {
                const int oo = s_out[r];
                if (oo < 0) continue;
                bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(smem + r * ROWB + chunk * 16);
                bf16_t* dst = outp + (size_t)oo + col;
                if (full) {
                    if (d.accumulate) {
                        const bf16x8_t old = *reinterpret_cast<const bf16x8_t*>(dst);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)old[e]);
                    }
                    *reinterpret_cast<bf16x8_t*>(dst) = v;
                } else {
                    const int nv = min(8, d.ncols - col);
                    for (int e = 0; e < nv; ++e) {
                        float x = (float)v[e];
                        if (d.accumulate) x += (float)dst[e];
                        dst[e] = (bf16_t)x;
                    }
                }
            }
        }
    }
}

template <int BP, int BC>
constexpr size_t smem_bytes() {
    return 2 * (size_t)(BP + BC) * BK * 2 + 3 * BP * sizeof(int) + (4 / (BC / 64)) * 2 * BC * sizeof(float);
}

template <int BP, int BC>
int launch(const KArgs& ka, hipStream_t st) {
    static bool attr_done = false;
    constexpr size_t smem = smem_bytes<BP, BC>();
    if (!attr_done) {
        int rc = mde_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_nt<BP, BC>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),
                               "hipFuncSetAttribute(conv_gemm_nt)");
        if (rc) return rc;
        attr_done = true;
    }
    conv_gemm_nt<BP, BC><<<dim3(ka.nP * ka.nC), dim3(NT), smem, st>>>(ka);
    MDE_LAUNCH_CHECK("conv_gemm_nt");
    return MDE_OK;
}

}  // namespace

extern "C" int mde_conv_gemm(const mde_conv_desc* d, const void* in, const void* w, void* out,
                             float* stats, void* stream) {
    MDE_REQUIRE(d && in && w && out, "mde_conv_gemm: null argument");
    MDE_REQUIRE(d->C > 0 && d->C % BK == 0, "mde_conv_gemm: C=%d must be a positive multiple of %d", d->C, BK);
    MDE_REQUIRE(d->ntaps >= 1 && d->ntaps <= MDE_MAX_TAPS, "mde_conv_gemm: ntaps=%d out of range", d->ntaps);
    MDE_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->GH > 0 && d->GW > 0 && d->ncols > 0,
                "mde_conv_gemm: non-positive dimension");
    MDE_REQUIRE(d->H < 32000 && d->W < 32000, "mde_conv_gemm: H/W too large for packed coordinates");
    MDE_REQUIRE(d->ld_in % 8 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)w % 16) == 0,
                "mde_conv_gemm: input/weight must be 16-byte aligned with ld_in %% 8 == 0");
    MDE_REQUIRE(d->in_bytes > 0 && d->in_bytes < MDE_OOB_OFFSET, "mde_conv_gemm: in_bytes must be < 2 GiB");
    const int64_t M = (int64_t)d->N * d->GH * d->GW;
    const int64_t out_elems = (int64_t)d->N * d->OH * d->OW * d->ld_out;
    MDE_REQUIRE(M < (1ll << 31) && out_elems < (1ll << 31), "mde_conv_gemm: tensor too large for 32-bit indexing");
    MDE_REQUIRE((d->GH - 1) * d->osy + d->ooy < d->OH && (d->GW - 1) * d->osx + d->oox < d->OW,
                "mde_conv_gemm: output grid exceeds the output tensor");
    for (int t = 0; t < d->ntaps; ++t)
        MDE_REQUIRE(d->wtap[t] >= 0 && d->wtap[t] < d->wtaps_total, "mde_conv_gemm: wtap[%d] out of range", t);
    MDE_REQUIRE(!(stats && d->accumulate), "mde_conv_gemm: stats with accumulate is not defined");
    const int64_t wbytes = (int64_t)d->ncols * d->wtaps_total * d->C * 2;
    MDE_REQUIRE(wbytes < MDE_OOB_OFFSET, "mde_conv_gemm: weight tensor must be < 2 GiB");

    KArgs ka;
    ka.d = *d;
    ka.in = in;
    ka.w = w;
    ka.out = out;
    ka.stats = stats;
    ka.w_bytes = (uint32_t)wbytes;
    ka.out_bytes = (uint32_t)(out_elems * 2);
    ka.M = (int32_t)M;
    ka.vec_ok = (d->ld_out % 8 == 0) && (((uintptr_t)out % 16) == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ka.nP = mde_cdiv(M, 128);
    if (d->ncols <= 64) {
        ka.nC = 1;
        return launch<128, 64>(ka, st);
    }
    ka.nC = mde_cdiv(d->ncols, 128);
    return launch<128, 128>(ka, st);
}
