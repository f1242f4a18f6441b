// The two convolutions that do not fit the implicit-GEMM kernel's "C % 64 == 0" contract:
//   stem  7x7/2, 3 -> 64   on the raw fp32 NCHW image (torchvision conv1, used at reference
//         network/FCRN.py:308,353)  — forward + weight gradient (no input gradient needed)
//   head  3x3,  Cin -> Cout<=32  with fp32 output (conv3, FCRN.py:340,368) — fwd, dgrad, wgrad
// The stem runs on MFMA with its im2col fragments built on the fly from an LDS image patch; the
// head runs on the vector ALUs in fp32.  Both use LDS-staged operands and coalesced global traffic.
#include "mde_common.h"

namespace {

constexpr int NT = 256;

// =================================================================== stem 7x7 / stride 2 / pad 3
constexpr int SK = 147;            // 7*7*3, weight layout [64][7][7][3] -> k = (kh*7+kw)*3+c
constexpr int SPW = 136;           // patch row pitch (133 used)
constexpr int STILE = 64;          // output pixels per tile (one output-row segment)

__device__ __forceinline__ void stem_load_patch(float* patch, const float* __restrict__ x, int n, int oy, int ox0,
                                                int H, int W) {
    for (int i = threadIdx.x; i < 3 * 7 * 133; i += NT) {
        const int j = i % 133, r = i / 133, kh = r % 7, c = r / 7;
        const int iy = 2 * oy - 3 + kh, ix = 2 * ox0 - 3 + j;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(((int64_t)n * 3 + c) * H + iy) * W + ix];
        patch[(c * 7 + kh) * SPW + j] = v;
    }
}

// MFMA formulation.  K = 147 (k = (kh*7+kw)*3+c, the OHWI weight order) padded to 160 = 5 steps of
// v_mfma_f32_16x16x32_bf16.  A operand = weights (rows = 64 output channels, register resident for
// the whole persistent workgroup), B operand = pixels: its fragments are built on the fly from the
// fp32 image patch in LDS (lane (px = l&15, g = l>>4) needs patch[c][kh][2*px + kw] for the 8 k's
// of its group; the 40 LDS offsets are per-lane constants).  The image is split into bf16 hi + lo
// parts (two MFMAs) so the stem keeps ~fp32 accuracy on the raw input; weights are bf16.
constexpr int SKP = 160;
constexpr int SKS = SKP / 32;      // 5 MFMA k-steps

__device__ __forceinline__ int stem_patch_off(int k) {   // k -> offset of tap (kh,kw), channel c in the patch
    const int c = k % 3, kw = (k / 3) % 7, kh = k / 21;
    return (c * 7 + kh) * SPW + kw;
}

__device__ __forceinline__ void split_bf16(float v, bf16_t& hi, bf16_t& lo) {
    hi = (bf16_t)v;
    lo = (bf16_t)(v - (float)hi);
}

__global__ __launch_bounds__(NT, 2) void stem_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                    bf16_t* __restrict__ out, int N, int H, int W, int OH, int OW) {
    __shared__ float patch[3 * 7 * SPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    // A fragments: weights of channels cb*16 + lr, k = ks*32 + 8*lg + j
    bf16x8_t wa[4][SKS];
    int poff[SKS][8];
#pragma unroll
    for (int ks = 0; ks < SKS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 32 + 8 * lg + j;
            poff[ks][j] = k < SK ? stem_patch_off(k) : -1;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) wa[cb][ks][j] = (bf16_t)(k < SK ? w[(cb * 16 + lr) * SK + k] : 0.f);
        }
    const int tiles_x = (OW + STILE - 1) / STILE;
    const int64_t ntiles = (int64_t)N * OH * tiles_x;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = (int)(t % tiles_x);
        const int oy = (int)((t / tiles_x) % OH);
        const int n = (int)(t / ((int64_t)tiles_x * OH));
        const int ox0 = tx * STILE;
        __syncthreads();
        stem_load_patch(patch, x, n, oy, ox0, H, W);
        __syncthreads();
        const int px = wave * 16 + lr;            // this lane's pixel inside the 64-pixel tile
        f32x4_t acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < SKS; ++ks) {
            bf16x8_t bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = poff[ks][j] >= 0 ? patch[poff[ks][j] + 2 * px] : 0.f;
                bf16_t hi, lo;
                split_bf16(v, hi, lo);
                bh[j] = hi;
                bl[j] = lo;
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb][ks], bh, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb][ks], bl, acc[cb], 0, 0, 0);
            }
        }
        // D: col = pixel (lane&15), rows = channels cb*16 + lg*4 + r
        if (ox0 + px < OW) {
            bf16_t* o = out + ((((int64_t)n * OH + oy) * OW) + ox0 + px) * 64 + lg * 4;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                bf16x4_t v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16_t)acc[cb][r];
                *reinterpret_cast<bf16x4_t*>(o + cb * 16) = v;
            }
        }
    }
}

// dw[ch][k] += sum_px dY[px][ch] * patch[px][k]:  A[ch][px] comes from the [pixel][channel] dY tile
// by transposed LDS reads (as in conv_wgrad.hip), B[px][k] from the patch (hi + lo parts).
// Wave w owns the k-column fragments {w, w+4, w+8}; all 4 channel fragments.
__device__ __forceinline__ int stem_dy_off(int row, int ch) {   // [64 px][64 ch] bf16, 128-B rows, XOR swizzle
    return row * 128 + 16 * (ch ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1));
}

__global__ __launch_bounds__(NT, 2) void stem_wgrad_k(const float* __restrict__ x, const bf16_t* __restrict__ dout,
                                                      float* __restrict__ dw, int N, int H, int W, int OH, int OW) {
    __shared__ float patch[3 * 7 * SPW];
    __shared__ __attribute__((aligned(16))) char dyt[STILE * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    constexpr int NF = 3;                    // k-column fragments per wave (10 in total: wave, wave+4, wave+8)
    int poff[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int k = (wave + 4 * f) * 16 + lr;
        poff[f] = (wave + 4 * f < SKP / 16 && k < SK) ? stem_patch_off(k) : -1;
    }
    f32x4_t acc[4][NF];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[cb][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int tiles_x = (OW + STILE - 1) / STILE;
    const int64_t ntiles = (int64_t)N * OH * tiles_x;
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = (int)(t % tiles_x);
        const int oy = (int)((t / tiles_x) % OH);
        const int n = (int)(t / ((int64_t)tiles_x * OH));
        const int ox0 = tx * STILE;
        __syncthreads();
        stem_load_patch(patch, x, n, oy, ox0, H, W);
        for (int i = threadIdx.x; i < STILE * 8; i += NT) {   // dY tile: 64 px x 8 chunks of 8 channels
            const int p = i >> 3, ch8 = i & 7;
            i32x4_t g = {0, 0, 0, 0};
            if (ox0 + p < OW)
                g = *reinterpret_cast<const i32x4_t*>(dout + ((((int64_t)n * OH + oy) * OW) + ox0 + p) * 64 + ch8 * 8);
            *reinterpret_cast<i32x4_t*>(dyt + stem_dy_off(p, ch8)) = g;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < STILE; kk += 32) {
            // A fragments (channels cb*16 + lr as rows, pixels kk + 8*lg + j as k): two transposed reads each
            bf16x8_t fa[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int q = lr >> 2, pp = lr & 3;
                const int row = kk + 8 * lg + q, ch = cb * 2 + (pp >> 1), sub = (pp & 1) * 8;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)(dyt + stem_dy_off(row, ch) + sub));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)(dyt + stem_dy_off(row + 4, ch) + sub));
                union { struct { s16x4_t a, b; } s; bf16x8_t v; } u;
                u.s.a = lo;
                u.s.b = hi;
                fa[cb] = u.v;
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (wave + 4 * f >= SKP / 16) continue;      // wave-uniform
                bf16x8_t bh, bl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int px = kk + 8 * lg + j;
                    const float v = poff[f] >= 0 ? patch[poff[f] + 2 * px] : 0.f;
                    bf16_t h, l;
                    split_bf16(v, h, l);
                    bh[j] = h;
                    bl[j] = l;
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    acc[cb][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cb], bh, acc[cb][f], 0, 0, 0);
                    acc[cb][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cb], bl, acc[cb][f], 0, 0, 0);
                }
            }
        }
    }
    // D: col = k index (lane&15), rows = channels cb*16 + lg*4 + r
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int k = (wave + 4 * f) * 16 + lr;
        if (wave + 4 * f >= SKP / 16 || k >= SK) continue;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(dw + (cb * 16 + lg * 4 + r) * SK + k, acc[cb][f][r]);
    }
}

// =================================================================== head 3x3 / pad 1, fp32 out
// 8 lanes per pixel, lane c8 owns channels [c8*8, c8*8+8) of every 64-channel group.
template <int CO>
__global__ __launch_bounds__(NT) void head_fwd_k(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                 float* __restrict__ out, int N, int H, int W, int Cin, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float ws[];   // [9][CO][Cin]
    for (int i = threadIdx.x; i < 9 * CO * Cin; i += NT) {
        const int c = i % Cin, co = (i / Cin) % CO, tap = i / (Cin * CO);
        ws[i] = co < Cout ? w[((int64_t)co * 9 + tap) * Cin + c] : 0.f;
    }
    __syncthreads();
    const int lpp = Cin >> 3;                       // lanes per pixel (8 for Cin = 64)
    const int c8 = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppb = NT / lpp;
    const int64_t npx = (int64_t)N * H * W;
    for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < npx; p += (int64_t)gridDim.x * ppb) {
        const int ix = (int)(p % W), iy = (int)((p / W) % H);
        const int64_t nb = p - (int64_t)iy * W - ix;   // n*H*W
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int y = iy + tap / 3 - 1, xx = ix + tap % 3 - 1;
            if ((unsigned)y >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
            const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(x + (nb + (int64_t)y * W + xx) * Cin + c8 * 8);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                const float* wr = ws + (tap * CO + co) * Cin + c8 * 8;
                const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wr), w1 = *reinterpret_cast<const f32x4_t*>(wr + 4);
                acc[co] += v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3] + v[4] * w1[0] + v[5] * w1[1] +
                           v[6] * w1[2] + v[7] * w1[3];
            }
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float s = acc[co];
            for (int o = 1; o < lpp; o <<= 1) s += __shfl_xor(s, o, 64);
            if (c8 == 0 && co < Cout) out[p * Cout + co] = s;
        }
    }
}

template <int CO>
__global__ __launch_bounds__(NT) void head_dgrad_k(const float* __restrict__ w, const float* __restrict__ dout,
                                                   bf16_t* __restrict__ dx, int N, int H, int W, int Cin, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float ws[];   // [9][CO][Cin]
    for (int i = threadIdx.x; i < 9 * CO * Cin; i += NT) {
        const int c = i % Cin, co = (i / Cin) % CO, tap = i / (Cin * CO);
        ws[i] = co < Cout ? w[((int64_t)co * 9 + tap) * Cin + c] : 0.f;
    }
    __syncthreads();
    const int lpp = Cin >> 3;
    const int c8 = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppb = NT / lpp;
    const int64_t npx = (int64_t)N * H * W;
    for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < npx; p += (int64_t)gridDim.x * ppb) {
        const int ix = (int)(p % W), iy = (int)((p / W) % H);
        const int64_t nb = p - (int64_t)iy * W - ix;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // output pixel q whose tap `tap` reads this input pixel: q = p - (tap offset)
            const int y = iy - (tap / 3 - 1), xx = ix - (tap % 3 - 1);
            if ((unsigned)y >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
            const float* g = dout + (nb + (int64_t)y * W + xx) * Cout;
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                if (co >= Cout) break;
                const float d = g[co];
                const float* wr = ws + (tap * CO + co) * Cin + c8 * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += d * wr[e];
            }
        }
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[e];
        *reinterpret_cast<bf16x8_t*>(dx + p * Cin + c8 * 8) = o;
    }
}

// dw[co][tap][c] += sum_px dout[px][co] * x[px + off(tap)][c]; one launch handles output
// channels [co0, co0 + COG).
template <int COG>
__global__ __launch_bounds__(NT) void head_wgrad_k(const bf16_t* __restrict__ x, const float* __restrict__ dout,
                                                   float* __restrict__ dw, int N, int H, int W, int Cin, int Cout, int co0) {
    __shared__ float red[NT / 64][COG * 9 * 64];   // Cin <= 64 per pass (lpp <= 8)
    const int lpp = Cin >> 3;
    const int c8 = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppb = NT / lpp;
    const int64_t npx = (int64_t)N * H * W;
    float acc[COG][9][8];
#pragma unroll
    for (int g = 0; g < COG; ++g)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[g][t][e] = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < npx; p += (int64_t)gridDim.x * ppb) {
        const int ix = (int)(p % W), iy = (int)((p / W) % H);
        const int64_t nb = p - (int64_t)iy * W - ix;
        float d[COG];
#pragma unroll
        for (int g = 0; g < COG; ++g) d[g] = co0 + g < Cout ? dout[p * Cout + co0 + g] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int y = iy + tap / 3 - 1, xx = ix + tap % 3 - 1;
            if ((unsigned)y >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
            const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(x + (nb + (int64_t)y * W + xx) * Cin + c8 * 8);
#pragma unroll
            for (int g = 0; g < COG; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][tap][e] += d[g] * (float)t[e];
        }
    }
    // lanes with equal c8 inside a wave, then the 4 waves through LDS, then one atomic per value
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < COG; ++g)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s = acc[g][t][e];
                for (int o = lpp; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
                if (lane < lpp) red[wv][(g * 9 + t) * 64 + lane * 8 + e] = s;
            }
    __syncthreads();
    for (int i = threadIdx.x; i < COG * 9 * Cin; i += NT) {
        const int c = i % Cin, t = (i / Cin) % 9, g = i / (Cin * 9);
        if (co0 + g < Cout) {
            float s = 0.f;
            for (int q = 0; q < NT / 64; ++q) s += red[q][(g * 9 + t) * 64 + c];
            atomicAdd(dw + ((int64_t)(co0 + g) * 9 + t) * Cin + c, s);
        }
    }
}

int px_grid(int64_t npx, int ppb) {
    int64_t nb = (npx + ppb - 1) / ppb;
    return (int)(nb > 256 * 8 ? 256 * 8 : (nb < 1 ? 1 : nb));
}

template <int CO>
int head_fwd_launch(const void* x, const float* w, float* out, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
    const size_t smem = (size_t)9 * CO * Cin * sizeof(float);
    if (smem > 48 * 1024) {
        int rc = mde_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&head_fwd_k<CO>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), "hipFuncSetAttribute(head_fwd_k)");
        if (rc) return rc;
    }
    head_fwd_k<CO><<<px_grid((int64_t)N * H * W, NT / (Cin / 8)), NT, smem, st>>>((const bf16_t*)x, w, out, N, H, W, Cin, Cout);
    MDE_LAUNCH_CHECK("head_fwd_k");
    return MDE_OK;
}
template <int CO>
int head_dgrad_launch(const float* w, const float* dout, void* dx, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
    const size_t smem = (size_t)9 * CO * Cin * sizeof(float);
    if (smem > 48 * 1024) {
        int rc = mde_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&head_dgrad_k<CO>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), "hipFuncSetAttribute(head_dgrad_k)");
        if (rc) return rc;
    }
    head_dgrad_k<CO><<<px_grid((int64_t)N * H * W, NT / (Cin / 8)), NT, smem, st>>>(w, dout, (bf16_t*)dx, N, H, W, Cin, Cout);
    MDE_LAUNCH_CHECK("head_dgrad_k");
    return MDE_OK;
}

#define HEAD_DISPATCH(fn, ...)                          \
    (Cout <= 1 ? fn<1>(__VA_ARGS__) : Cout <= 2 ? fn<2>(__VA_ARGS__) : Cout <= 4 ? fn<4>(__VA_ARGS__) \
     : Cout <= 8 ? fn<8>(__VA_ARGS__) : Cout <= 16 ? fn<16>(__VA_ARGS__) : fn<32>(__VA_ARGS__))

int head_check(const char* who, int N, int H, int W, int Cin, int Cout) {
    MDE_REQUIRE(N > 0 && H > 0 && W > 0, "%s: non-positive size", who);
    MDE_REQUIRE(Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64, "%s: Cin=%d unsupported (8, 16, 32 or 64)", who, Cin);
    MDE_REQUIRE(Cout >= 1 && Cout <= 32, "%s: Cout=%d unsupported (1..32)", who, Cout);
    return MDE_OK;
}

}  // namespace

extern "C" int mde_stem_conv_fwd(const float* x, const float* w, void* out, int N, int H, int W, void* stream) {
    MDE_REQUIRE(x && w && out && N > 0 && H > 0 && W > 0, "mde_stem_conv_fwd: bad argument");
    MDE_REQUIRE(((uintptr_t)out % 16) == 0, "mde_stem_conv_fwd: out must be 16-byte aligned");
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    const int64_t ntiles = (int64_t)N * OH * ((OW + STILE - 1) / STILE);
    const int grid = (int)(ntiles > 2048 ? 2048 : ntiles);
    stem_fwd_k<<<grid, NT, 0, (hipStream_t)stream>>>(x, w, (bf16_t*)out, N, H, W, OH, OW);
    MDE_LAUNCH_CHECK("stem_fwd_k");
    return MDE_OK;
}

extern "C" int mde_stem_conv_wgrad(const float* x, const void* dout, float* dw, int N, int H, int W, void* stream) {
    MDE_REQUIRE(x && dout && dw && N > 0 && H > 0 && W > 0, "mde_stem_conv_wgrad: bad argument");
    MDE_REQUIRE(((uintptr_t)dout % 16) == 0, "mde_stem_conv_wgrad: dout must be 16-byte aligned");
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    const int64_t ntiles = (int64_t)N * OH * ((OW + STILE - 1) / STILE);
    const int grid = (int)(ntiles > 1024 ? 1024 : ntiles);
    stem_wgrad_k<<<grid, NT, 0, (hipStream_t)stream>>>(x, (const bf16_t*)dout, dw, N, H, W, OH, OW);
    MDE_LAUNCH_CHECK("stem_wgrad_k");
    return MDE_OK;
}

extern "C" int mde_head_conv_fwd(const void* x, const float* w, float* out, int N, int H, int W, int Cin, int Cout,
                                 void* stream) {
    MDE_REQUIRE(x && w && out, "mde_head_conv_fwd: null argument");
    if (int rc = head_check("mde_head_conv_fwd", N, H, W, Cin, Cout)) return rc;
    MDE_REQUIRE(((uintptr_t)x % 16) == 0, "mde_head_conv_fwd: x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    return HEAD_DISPATCH(head_fwd_launch, x, w, out, N, H, W, Cin, Cout, st);
}

extern "C" int mde_head_conv_bwd(const void* x, const float* w, const float* dout, void* dx, float* dw, int N, int H,
                                 int W, int Cin, int Cout, void* stream) {
    MDE_REQUIRE(x && w && dout && (dx || dw), "mde_head_conv_bwd: null argument");
    if (int rc = head_check("mde_head_conv_bwd", N, H, W, Cin, Cout)) return rc;
    MDE_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dx % 16) == 0, "mde_head_conv_bwd: x/dx must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        int rc = HEAD_DISPATCH(head_dgrad_launch, w, dout, dx, N, H, W, Cin, Cout, st);
        if (rc) return rc;
    }
    if (dw) {
        const int grid = px_grid((int64_t)N * H * W, NT / (Cin / 8));
        const int g = grid > 1024 ? 1024 : grid;
        if (Cout == 1) {
            head_wgrad_k<1><<<g, NT, 0, st>>>((const bf16_t*)x, dout, dw, N, H, W, Cin, Cout, 0);
            MDE_LAUNCH_CHECK("head_wgrad_k");
        } else {
            for (int co0 = 0; co0 < Cout; co0 += 2) {
                head_wgrad_k<2><<<g, NT, 0, st>>>((const bf16_t*)x, dout, dw, N, H, W, Cin, Cout, co0);
                MDE_LAUNCH_CHECK("head_wgrad_k");
            }
        }
    }
    return MDE_OK;
}
