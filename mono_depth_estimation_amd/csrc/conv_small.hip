// The two convolutions that do not fit the implicit-GEMM kernel's "C % 64 == 0" contract:
//   stem  7x7/2, 3 -> 64   on the raw fp32 NCHW image (torchvision conv1, used at reference
//         network/FCRN.py:308,353)  — forward + weight gradient (no input gradient needed)
//   head  3x3,  Cin -> Cout<=32  with fp32 output (conv3, FCRN.py:340,368) — fwd, dgrad, wgrad
// The stem runs on MFMA with its im2col fragments built on the fly from an LDS image patch; the
// head runs on the vector ALUs in fp32.  Both use LDS-staged operands and coalesced global traffic.
#include "mde_common.h"
#include <type_traits>

namespace {

constexpr int NT = 256;

// =================================================================== stem 7x7 / stride 2 / pad 3
// MFMA formulation (v_mfma_f32_16x16x32_bf16).  A persistent workgroup walks tiles of TY x 64
// output pixels; the fp32 NCHW image patch of a tile (3 x (2*TY+5) x 133) is staged in LDS ONCE,
// already split into bf16 hi + lo parts packed in one 32-bit word (two MFMAs per product keep
// ~fp32 accuracy on the raw image; weights are bf16 as everywhere).  Even and odd image columns
// live in separate planes of a patch row, so the 16 pixels of a fragment (image columns
// 2*px + kw) read consecutive words.  The next tile's patch is prefetched into registers while
// the current one is computed.  All global loads are unconditional buffer loads (an out-of-image
// element gets an out-of-range offset and reads as 0): a load under a per-lane condition
// compiles to its own block with a full vmcnt(0) wait.
constexpr int SK = 147;            // 7*7*3, weight layout [64][7][7][3] -> k = (kh*7+kw)*3+c
constexpr int STX = 64;            // output pixels per tile row
constexpr int SPC = 2 * STX + 5;   // 133 input columns
static_assert(SPC == 128 + 5, "the patch loader covers columns 0..127 plus a 5-column tail");
constexpr int SPH = 68;            // words per column plane (even columns, then odd columns)
constexpr int SPW = 2 * SPH;       // patch row pitch (words)

// Patch geometry for TY output rows per tile
template <int TY>
struct StemGeo {
    static constexpr int PR = 2 * TY + 5;          // input rows per channel
    static constexpr int ROWS = 3 * PR;            // patch rows (channel-major)
    static constexpr int WORDS = ROWS * SPW;
    // zero words behind the patch (target of padded k): last tile row, odd plane, last pixel, 4 words
    static constexpr int ZPAD = 2 * (TY - 1) * SPW + SPH + STX + 8;
};

// Tile walk of a persistent workgroup: (n, ty, tx) advanced by the grid size with scalar carries
// (a 64-bit t % tiles_x per tile costs ~130 instructions of software division, three times over).
struct StemTile { int n, oy0, ox0; };
template <int TY>
struct StemWalk {
    int tx, ty, n, dx, dy, dn, tiles_x, tiles_y;
    __device__ __forceinline__ void init(int first, int step, int tiles_x_, int tiles_y_) {
        tiles_x = tiles_x_;
        tiles_y = tiles_y_;
        tx = first % tiles_x;
        ty = (first / tiles_x) % tiles_y;
        n = first / (tiles_x * tiles_y);
        dx = step % tiles_x;
        dy = (step / tiles_x) % tiles_y;
        dn = step / (tiles_x * tiles_y);
    }
    __device__ __forceinline__ StemTile tile() const { return StemTile{n, ty * TY, tx * STX}; }
    __device__ __forceinline__ void next() {
        tx += dx;
        int c = tx >= tiles_x;
        tx -= c ? tiles_x : 0;
        ty += dy + c;
        c = ty >= tiles_y;
        ty -= c ? tiles_y : 0;
        n += dn + c;
    }
};

__device__ __forceinline__ int stem_patch_word(int row, int col) {   // word of image column `col` in patch row `row`
    return row * SPW + (col & 1) * SPH + (col >> 1);
}

__device__ __forceinline__ uint32_t pack_hilo(float v) {  // bf16(v) | bf16(v - bf16(v)) << 16
    const bf16_t hi = (bf16_t)v;
    const bf16_t lo = (bf16_t)(v - (float)hi);
    return (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
}

// Patch loader for NTT threads.  Thread -> (rp = tid >> 7, column j = tid & 127) covers columns
// 0..127 of rows RP*q + rp; the 5 remaining columns of every row are one more load for the first
// ROWS*5 threads.  The image-relative word offset and the row-within-channel of every load are
// per-thread constants built once; per tile a load costs an add, a row-range compare and a select.
template <int NTT, int TY>
struct StemLoader {
    using G = StemGeo<TY>;
    static constexpr int RP = NTT / 128;                       // patch rows per pass
    static constexpr int Q = (G::ROWS + RP - 1) / RP + 1;      // loads per thread (last: tail columns)
    static_assert(G::ROWS * 5 <= NTT && G::PR < 31, "tail columns fit one pass");
    int relpr[Q];      // ((c*H + pr)*W + column) | pr << 27; pr = 31: no such row
    int col, tcol;

    __device__ __forceinline__ void init(int H, int W) {
        const int rp = threadIdx.x >> 7;
        col = threadIdx.x & 127;
#pragma unroll
        for (int q = 0; q < Q - 1; ++q) {
            const int r = RP * q + rp, c = r / G::PR, pr = r - c * G::PR;
            relpr[q] = r < G::ROWS ? (((c * H + pr) * W + col) | (pr << 27)) : (31 << 27);
        }
        const int r = threadIdx.x / 5, c = r / G::PR, pr = r - c * G::PR;
        tcol = 128 + threadIdx.x % 5;
        relpr[Q - 1] = r < G::ROWS ? (((c * H + pr) * W + tcol) | (pr << 27)) : (31 << 27);
    }
    __device__ __forceinline__ void issue(float (&pv)[Q], const __amdgpu_buffer_rsrc_t rs, const StemTile tl, int H, int W) const {
        const int iy0 = 2 * tl.oy0 - 3, ix0 = 2 * tl.ox0 - 3;
        const int base = (tl.n * 3 * H + iy0) * W + ix0;
        const int lo = -iy0, hi = min(H - iy0, G::PR);         // image rows: pr in [lo, hi); pr = 31 (no row) fails
        const bool okx = (unsigned)(ix0 + col) < (unsigned)W, okt = (unsigned)(ix0 + tcol) < (unsigned)W;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int pr = (int)((uint32_t)relpr[q] >> 27), rel = relpr[q] & 0x7FFFFFF;
            const bool ok = (q < Q - 1 ? okx : okt) & (pr >= lo) & (pr < hi);
            const uint32_t off = ok ? (uint32_t)(base + rel) * 4u : MDE_OOB_OFFSET;
            pv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
        }
    }
    __device__ __forceinline__ void write(uint32_t* patch, const float (&pv)[Q]) const {
        const int rp = threadIdx.x >> 7;
#pragma unroll
        for (int q = 0; q < Q - 1; ++q) {
            const int r = RP * q + rp;
            if (r < G::ROWS) patch[stem_patch_word(r, col)] = pack_hilo(pv[q]);
        }
        if (threadIdx.x < G::ROWS * 5) patch[stem_patch_word(threadIdx.x / 5, tcol)] = pack_hilo(pv[Q - 1]);
    }
    // words the loader never writes (plane tails) must hold finite values: they meet zero weights
    __device__ __forceinline__ static void clear_tails(uint32_t* patch) {
        for (int r = threadIdx.x; r < G::ROWS; r += NTT) {
            patch[r * SPW + SPH - 1] = 0u;                     // even plane: columns 0..132 fill indices 0..66
            patch[r * SPW + 2 * SPH - 2] = patch[r * SPW + 2 * SPH - 1] = 0u;   // odd plane: indices 0..65
        }
    }
};

// eight packed words (hi | lo << 16) -> the hi and lo bf16x8 fragments
__device__ __forceinline__ void stem_unpack8(const uint32_t (&wv)[8], bf16x8_t& bh, bf16x8_t& bl) {
    i32x4_t h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (int)__builtin_amdgcn_perm(wv[2 * i + 1], wv[2 * i], 0x05040100u);
        l[i] = (int)__builtin_amdgcn_perm(wv[2 * i + 1], wv[2 * i], 0x07060302u);
    }
    bh = __builtin_bit_cast(bf16x8_t, h);
    bl = __builtin_bit_cast(bf16x8_t, l);
}

// Forward.  K is re-ordered for the gather: k' = g*8 + e with g = c*7 + kh (21 groups, padded to
// 24 = 6 MFMA k-steps) and e = kw (e = 7: zero weight), so the 8 k's of a lane group are the image
// columns 2*px .. 2*px+7 of ONE patch row: four consecutive words of the even plane and four of
// the odd plane, read with immediates from one per-k-step base (the plain (kh,kw,c) order needs
// 40 per-lane offsets and an address add per word: measured issue-bound, MFMA pipe 22 % busy;
// profiles/).  8 waves, two per SIMD; wave w owns output row w of the 8 x 64 tile and all 64
// channels (A operand = weights, register resident for the whole persistent workgroup).
// stats (optional): BatchNorm partial sums of the fp32 results, kept in registers across tiles.
// CB: channel blocks of 16 this launch computes (4: torchvision's 64-channel conv1 in one launch; 3: one half of
// densenet161's 96-channel conv0, Bts.py:289 -- the weights of a launch stay in registers, 6 blocks would not fit);
// `w` and `out` point at the launch's first channel, ldo is the whole tensor's channel count, c0 the first channel.
constexpr int NTF = 512;
constexpr int FTY = 8;             // output rows per forward tile
constexpr int FKS = 6;             // k-steps of 32 (24 groups of 8)
#ifndef MDE_STEM_ABLATE
#define MDE_STEM_ABLATE 0   // diagnostics: 1 no output stores, 2 no MFMAs, 4 no patch loads (bit mask)
#endif
template <int CB>
__global__ __launch_bounds__(NTF, 1) void stem_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                     bf16_t* __restrict__ out, float* stats, int N, int H, int W, int OH,
                                                     int OW, int ldo, int c0, int det) {
    using G = StemGeo<FTY>;
    __shared__ uint32_t patch[2][G::WORDS + G::ZPAD];
    __shared__ float s_red[NTF / 64][2][CB * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    bf16x8_t wa[CB][FKS];
    int poffb[FKS];                          // byte offset of the lane's group row (even plane, its pixel) in a patch buffer
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) {
        const int g = ks * 4 + lg, c = g / 7, kh = g - c * 7;
        poffb[ks] = ((g < 21 ? (c * G::PR + kh) * SPW : G::WORDS) + 2 * wave * SPW + lr) * 4;   // padded groups: the zero words
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                wa[cb][ks][e] = (bf16_t)((g < 21 && e < 7) ? w[(cb * 16 + lr) * SK + (kh * 7 + e) * 3 + c] : 0.f);
    }
    for (int i = threadIdx.x; i < G::ZPAD; i += NTF) patch[0][G::WORDS + i] = patch[1][G::WORDS + i] = 0u;
    StemLoader<NTF, FTY>::clear_tails(patch[0]);
    StemLoader<NTF, FTY>::clear_tails(patch[1]);
    f32x4_t s1[CB], s2[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) s1[cb] = s2[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int tiles_x = (OW + STX - 1) / STX, tiles_y = (OH + FTY - 1) / FTY;
    const int ntiles = N * tiles_y * tiles_x;
    const __amdgpu_buffer_rsrc_t rs_x = mde_rsrc(x, (uint32_t)((int64_t)N * 3 * H * W * 4));
    StemLoader<NTF, FTY> ld;
    ld.init(H, W);
    StemWalk<FTY> walk;
    walk.init(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    float pv[StemLoader<NTF, FTY>::Q];
    if ((int)blockIdx.x < ntiles) {
        ld.issue(pv, rs_x, walk.tile(), H, W);
        ld.write(patch[0], pv);
    }
    __syncthreads();
    StemTile tl;
    auto compute = [&](auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value;
        const char* pb = reinterpret_cast<const char*>(&patch[CUR][0]);
        const int oy = tl.oy0 + wave;
        if (oy >= OH) return;
#pragma unroll
        for (int f = 0; f < STX / 16; ++f) {
            if (tl.ox0 + f * 16 >= OW) break;
            f32x4_t acc[CB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < FKS; ++ks) {
                const uint32_t* pe = reinterpret_cast<const uint32_t*>(pb + poffb[ks] + f * 64);
                uint32_t wv[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wv[2 * i] = pe[i];                // image column 2*px + 2i
                    wv[2 * i + 1] = pe[SPH + i];      // image column 2*px + 2i + 1
                }
                bf16x8_t bh, bl;
                stem_unpack8(wv, bh, bl);
#if MDE_STEM_ABLATE & 2
                acc[0][0] += (float)bh[0] + (float)bl[1];
                acc[1][1] += (float)bh[2] + (float)bl[3] + (float)bh[4] + (float)bl[5] + (float)bh[6] + (float)bl[7];
#else
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[cb] = MDE_MFMA_16x16x32(wa[cb][ks], bh, acc[cb]);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[cb] = MDE_MFMA_16x16x32(wa[cb][ks], bl, acc[cb]);
#endif
            }
            // D: col = pixel (lane&15), rows = channels cb*16 + lg*4 + r
            const int ox = tl.ox0 + f * 16 + lr;
            if (ox < OW) {
                bf16_t* o = out + ((((int64_t)tl.n * OH + oy) * OW) + ox) * ldo + lg * 4;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    bf16x4_t v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (bf16_t)acc[cb][r];
                    if (!(MDE_STEM_ABLATE & 1) || v[0] == (bf16_t)123.f) *reinterpret_cast<bf16x4_t*>(o + cb * 16) = v;
                    s1[cb] += acc[cb];
                    s2[cb] += acc[cb] * acc[cb];
                }
            }
        }
    };
    int cur = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        tl = walk.tile();
        if (t + (int)gridDim.x < ntiles) walk.next();                     // last round: reload the same tile (unused)
        if (!(MDE_STEM_ABLATE & 4)) ld.issue(pv, rs_x, walk.tile(), H, W);
        if (cur)
            compute(std::integral_constant<int, 1>{});
        else
            compute(std::integral_constant<int, 0>{});
        ld.write(patch[cur ^ 1], pv);
        __syncthreads();
        cur ^= 1;
    }
    if (stats) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a1 = mde_row16_sum(s1[cb][r]), a2 = mde_row16_sum(s2[cb][r]);
                if (lr == 0) {
                    s_red[wave][0][cb * 16 + lg * 4 + r] = a1;
                    s_red[wave][1][cb * 16 + lg * 4 + r] = a2;
                }
            }
        __syncthreads();
        if (threadIdx.x < 2 * CB * 16) {
            const int which = threadIdx.x / (CB * 16), ch = threadIdx.x % (CB * 16);
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < NTF / 64; ++q) v += s_red[q][which][ch];
            mde_stat_add(stats, ldo, blockIdx.x, which, c0 + ch, v, det);
        }
    }
}

// ---- weight gradient (NT threads, tiles of WTY x 64 pixels, the plain k = (kh*7+kw)*3+c column order)
constexpr int SKP = 160;           // 147 output columns padded to 10 fragments of 16
constexpr int WTY = 4;
__device__ __forceinline__ int stem_patch_off(int k) {   // k -> word offset of tap (kh,kw), channel c for pixel (0,0)
    const int c = k % 3, kw = (k / 3) % 7, kh = k / 21;
    return stem_patch_word(c * StemGeo<WTY>::PR + kh, kw);
}

// dw[ch][k] += sum_px dY[px][ch] * patch[px][k]:  A[ch][px] comes from the [pixel][channel] dY tile
// by transposed LDS reads (as in conv_wgrad.hip), B[px][k] from the packed patch (hi + lo parts).
// Wave w owns the k-column fragments {w, w+4, w+8}; all 4 channel fragments.  Patch and dY tile of
// the next tile are prefetched into registers during the MFMAs.
__device__ __forceinline__ int stem_dy_off(int row, int ch) {   // [px][64 ch] bf16, 128-B rows, XOR swizzle
    return row * 128 + 16 * (ch ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1));
}
// (CB channel blocks of 16 per launch = 2 CB chunks of 16 bytes per pixel; the LDS rows keep their 128-byte pitch)
template <int CB>
__device__ __forceinline__ void stem_dy_issue(i32x4_t (&dv)[WTY * STX * 2 * CB / NT], const __amdgpu_buffer_rsrc_t rs, const StemTile tl,
                                              int OH, int OW, int ldo) {
#pragma unroll
    for (int q = 0; q < WTY * STX * 2 * CB / NT; ++q) {
        const int i = threadIdx.x + q * NT;
        const int p = i / (2 * CB), ch8 = i % (2 * CB);
        const int oy = tl.oy0 + p / STX, ox = tl.ox0 + p % STX;       // p < WTY * STX
        const bool ok = (oy < OH) & (ox < OW);                       // pixels outside the image read as zeros
        const uint32_t off = ok ? (uint32_t)(((tl.n * OH + oy) * OW + ox) * ldo + ch8 * 8) * 2u : MDE_OOB_OFFSET;
        dv[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    }
}

template <int CB>
__global__ __launch_bounds__(NT, 2) void stem_wgrad_k(const float* __restrict__ x, const bf16_t* __restrict__ dout,
                                                      float* __restrict__ dw, int N, int H, int W, int OH, int OW, int ldo,
                                                      MdeDetDev det) {
    constexpr int SDQ = WTY * STX * 2 * CB / NT;   // 16-byte dY chunks per thread per tile (8 for 64 channels)
    static_assert(WTY * STX * 2 * CB % NT == 0, "whole chunks per thread");
    using G = StemGeo<WTY>;
    __shared__ uint32_t patch[G::WORDS + G::ZPAD];
    __shared__ __attribute__((aligned(16))) char dyt[WTY * STX * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    constexpr int NF = 3;                    // k-column fragments per wave (10 in total: wave, wave+4, wave+8)
    int poff[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int k = (wave + 4 * f) * 16 + lr;
        poff[f] = (wave + 4 * f < SKP / 16 && k < SK) ? stem_patch_off(k) : G::WORDS;
    }
    for (int i = threadIdx.x; i < G::ZPAD; i += NT) patch[G::WORDS + i] = 0u;
    StemLoader<NT, WTY>::clear_tails(patch);
    f32x4_t acc[CB][NF];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[cb][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int tiles_x = (OW + STX - 1) / STX, tiles_y = (OH + WTY - 1) / WTY;
    const int ntiles = N * tiles_y * tiles_x;
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const __amdgpu_buffer_rsrc_t rs_x = mde_rsrc(x, (uint32_t)((int64_t)N * 3 * H * W * 4));
    const __amdgpu_buffer_rsrc_t rs_d = mde_rsrc(dout, (uint32_t)((((int64_t)N * OH * OW - 1) * ldo + CB * 16) * 2));
    StemLoader<NT, WTY> ld;
    ld.init(H, W);
    StemWalk<WTY> walk;
    walk.init(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    float pv[StemLoader<NT, WTY>::Q];
    i32x4_t dv[SDQ];
    if ((int)blockIdx.x < ntiles) {
        ld.issue(pv, rs_x, walk.tile(), H, W);
        stem_dy_issue<CB>(dv, rs_d, walk.tile(), OH, OW, ldo);
    }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                               // everyone is done with the previous tile's LDS
        ld.write(patch, pv);
#pragma unroll
        for (int q = 0; q < SDQ; ++q) {
            const int i = threadIdx.x + q * NT;
            *reinterpret_cast<i32x4_t*>(dyt + stem_dy_off(i / (2 * CB), i % (2 * CB))) = dv[q];
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) walk.next();
        ld.issue(pv, rs_x, walk.tile(), H, W);
        stem_dy_issue<CB>(dv, rs_d, walk.tile(), OH, OW, ldo);
#pragma unroll 2
        for (int kk = 0; kk < WTY * STX; kk += 32) {
            // A fragments (channels cb*16 + lr as rows, pixels kk + 8*lg + j as k): two transposed reads each
            bf16x8_t fa[CB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int q = lr >> 2, pp = lr & 3;
                const int row = kk + 8 * lg + q, ch = cb * 2 + (pp >> 1), sub = (pp & 1) * 8;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)(dyt + stem_dy_off(row, ch) + sub));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)(dyt + stem_dy_off(row + 4, ch) + sub));
                union { struct { s16x4_t a, b; } s; bf16x8_t v; } u;
                u.s.a = lo;
                u.s.b = hi;
                fa[cb] = u.v;
            }
            // pixels kk + 8*lg + j: tile row kk/64, columns (kk%64) + 8*lg + j
            const uint32_t* P = patch + 2 * (kk / STX) * SPW + ((kk % STX) + 8 * lg);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (wave + 4 * f >= SKP / 16) continue;      // wave-uniform
                uint32_t wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = P[poff[f] + j];
                bf16x8_t bh, bl;
                stem_unpack8(wv, bh, bl);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    acc[cb][f] = MDE_MFMA_16x16x32(fa[cb], bh, acc[cb][f]);
                    acc[cb][f] = MDE_MFMA_16x16x32(fa[cb], bl, acc[cb][f]);
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
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mde_grad_add(dw + (cb * 16 + lg * 4 + r) * SK + k, acc[cb][f][r], det);
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
                                                   float* __restrict__ dw, int N, int H, int W, int Cin, int Cout, int co0,
                                                   MdeDetDev det) {
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
            mde_grad_add(dw + ((int64_t)(co0 + g) * 9 + t) * Cin + c, s, det);
        }
    }
}

// ------------------------------------------------------------------- head, Cin = 64 or 32 -> Cout = 1 (FCRN conv3; BTS get_depth)
// LPP = Cin / 8 lanes per pixel: eight or four (lane c8 owns channels [c8*8, c8*8+8)); a lane group walks runs of HR
// consecutive pixels of one row, so the 3x3 windows of neighbouring outputs share their loads
// (18 instead of 36 per run).  All weights sit in registers; out-of-image taps are buffer loads
// with an out-of-range offset (read as 0).
constexpr int HR = 4;
typedef bf16x2_t bf16pair_t;

template <int LPP>
__device__ __forceinline__ float head_sum(float v) {     // sum over the LPP (8 or 4) lanes of a pixel, result in all of them
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    asm("" : "+v"(v));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    if constexpr (LPP == 8) {
        asm("" : "+v"(v));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    }
    return v;
}

// forward: weights as bf16 hi + lo pairs, v_dot2c_f32_bf16 against the bf16 activations (no conversions)
template <int LPP>
__global__ __launch_bounds__(NT) void head1_fwd_k(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                  float* __restrict__ out, int N, int H, int W) {
    constexpr int CIN = 8 * LPP;
    static_assert((LPP == 8 || LPP == 4) && LPP >= HR, "a pixel owns 8 or 4 lanes, one per output of a run");
    const int c8 = threadIdx.x & (LPP - 1), grp = threadIdx.x / LPP;
    bf16pair_t wh[9][4], wl[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float v = w[t * CIN + c8 * 8 + 2 * i + h];
                const bf16_t hi = (bf16_t)v;
                wh[t][i][h] = hi;
                wl[t][i][h] = (bf16_t)(v - (float)hi);
            }
    const __amdgpu_buffer_rsrc_t rs = mde_rsrc(x, (uint32_t)((int64_t)N * H * W * CIN * 2));
    const int runs_x = (W + HR - 1) / HR;
    const int nruns = N * H * runs_x;
    for (int it = blockIdx.x * (NT / LPP) + grp; it < nruns; it += gridDim.x * (NT / LPP)) {
        const int xb = it % runs_x, ny = it / runs_x, y = ny % H;
        const int x0 = xb * HR;
        float acc[HR];
#pragma unroll
        for (int p = 0; p < HR; ++p) acc[p] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            const bool rok = (unsigned)yy < (unsigned)H;
            const int rowbase = ((ny - y + yy) * W) * CIN + c8 * 8;      // (n*H + yy) * W pixels
            i32x4_t v[HR + 2];
#pragma unroll
            for (int j = 0; j < HR + 2; ++j) {
                const int col = x0 - 1 + j;
                const bool ok = rok & ((unsigned)col < (unsigned)W);
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (uint32_t)(rowbase + col * CIN) * 2u : MDE_OOB_OFFSET, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < HR; ++p)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int xi = v[p + kx][i];      // (scalar copy on purpose: __builtin_bit_cast applied to a
                        const bf16pair_t xv = __builtin_bit_cast(bf16pair_t, xi);   //  vector-element expression reads element 0)
                        acc[p] = MDE_FDOT2(xv, wh[ky * 3 + kx][i], acc[p]);
                        acc[p] = MDE_FDOT2(xv, wl[ky * 3 + kx][i], acc[p]);
                    }
        }
        float mine = 0.f;
#pragma unroll
        for (int p = 0; p < HR; ++p) {
            const float sp = head_sum<LPP>(acc[p]);
            mine = c8 == p ? sp : mine;
        }
        if (c8 < HR && x0 + c8 < W) out[(int64_t)ny * W + x0 + c8] = mine;
    }
}

// input gradient: dx[p][c] = sum_tap dout[p - off(tap)] * w[tap][c]   (fp32 weights in registers)
template <int LPP>
__global__ __launch_bounds__(NT) void head1_dgrad_k(const float* __restrict__ w, const float* __restrict__ dout,
                                                    bf16_t* __restrict__ dx, int N, int H, int W) {
    constexpr int CIN = 8 * LPP;
    static_assert((LPP == 8 || LPP == 4) && LPP >= HR, "a pixel owns 8 or 4 lanes, one per output of a run");
    const int c8 = threadIdx.x & (LPP - 1), grp = threadIdx.x / LPP;
    f32x2_t wr[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) wr[t][i] = f32x2_t{w[t * CIN + c8 * 8 + 2 * i], w[t * CIN + c8 * 8 + 2 * i + 1]};
    const __amdgpu_buffer_rsrc_t rs = mde_rsrc(dout, (uint32_t)((int64_t)N * H * W * 4));
    const int runs_x = (W + HR - 1) / HR;
    const int nruns = N * H * runs_x;
    for (int it = blockIdx.x * (NT / LPP) + grp; it < nruns; it += gridDim.x * (NT / LPP)) {
        const int xb = it % runs_x, ny = it / runs_x, y = ny % H;
        const int x0 = xb * HR;
        f32x2_t acc[HR][4];
#pragma unroll
        for (int p = 0; p < HR; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[p][i] = f32x2_t{0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            // tap (ky, kx) of output pixel q reads input pixel q + (ky-1, kx-1): q = p - (ky-1, kx-1)
            const int yy = y - (ky - 1);
            const bool rok = (unsigned)yy < (unsigned)H;
            float d[HR + 2];
#pragma unroll
            for (int j = 0; j < HR + 2; ++j) {
                const int col = x0 - 1 + j;
                const bool ok = rok & ((unsigned)col < (unsigned)W);
                d[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs, ok ? (uint32_t)((ny - y + yy) * W + col) * 4u : MDE_OOB_OFFSET, 0, 0));
            }
#pragma unroll
            for (int p = 0; p < HR; ++p)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float dq = d[p + 2 - kx];                 // column x0 + p - (kx - 1)
                    const f32x2_t dd = {dq, dq};
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[p][i] = __builtin_elementwise_fma(dd, wr[ky * 3 + kx][i], acc[p][i]);
                }
        }
#pragma unroll
        for (int p = 0; p < HR; ++p) {
            if (x0 + p >= W) break;
            bf16x8_t o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[2 * i] = (bf16_t)acc[p][i][0];
                o[2 * i + 1] = (bf16_t)acc[p][i][1];
            }
            *reinterpret_cast<bf16x8_t*>(dx + ((int64_t)ny * W + x0 + p) * CIN + c8 * 8) = o;
        }
    }
}

// weight gradient, input-stationary: every activation is converted once and meets the nine dout
// values around it:  dw[(ky,kx)][c] += dout[p - (ky-1, kx-1)] * x[p][c]
template <int LPP>
__global__ __launch_bounds__(NT) void head1_wgrad_k(const bf16_t* __restrict__ x, const float* __restrict__ dout,
                                                    float* __restrict__ dw, int N, int H, int W, MdeDetDev det) {
    __shared__ float red[NT / 64][9 * 8 * LPP];
    constexpr int CIN = 8 * LPP;
    static_assert((LPP == 8 || LPP == 4) && LPP >= HR, "a pixel owns 8 or 4 lanes, one per output of a run");
    const int c8 = threadIdx.x & (LPP - 1), grp = threadIdx.x / LPP;
    f32x2_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x2_t{0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rs_d = mde_rsrc(dout, (uint32_t)((int64_t)N * H * W * 4));
    const __amdgpu_buffer_rsrc_t rs_x = mde_rsrc(x, (uint32_t)((int64_t)N * H * W * CIN * 2));
    const int runs_x = (W + HR - 1) / HR;
    const int nruns = N * H * runs_x;
    for (int it = blockIdx.x * (NT / LPP) + grp; it < nruns; it += gridDim.x * (NT / LPP)) {
        const int xb = it % runs_x, ny = it / runs_x, y = ny % H;
        const int x0 = xb * HR;
        i32x4_t xv[HR];
#pragma unroll
        for (int p = 0; p < HR; ++p)
            xv[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, x0 + p < W ? (uint32_t)((ny * W + x0 + p) * CIN + c8 * 8) * 2u : MDE_OOB_OFFSET, 0, 0);
        float d[3][HR + 2];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y - (ky - 1);
            const bool rok = (unsigned)yy < (unsigned)H;
#pragma unroll
            for (int j = 0; j < HR + 2; ++j) {
                const int col = x0 - 1 + j;
                const bool ok = rok & ((unsigned)col < (unsigned)W);
                d[ky][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs_d, ok ? (uint32_t)((ny - y + yy) * W + col) * 4u : MDE_OOB_OFFSET, 0, 0));
            }
        }
#pragma unroll
        for (int p = 0; p < HR; ++p) {
            f32x2_t xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t u = (uint32_t)xv[p][i];
#ifdef MDE_ACT_F16
                const bf16x2_t h2 = __builtin_bit_cast(bf16x2_t, u);
                xf[i] = f32x2_t{(float)h2[0], (float)h2[1]};
#else
                xf[i] = f32x2_t{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xFFFF0000u)};   // bf16 -> fp32 is a shift
#endif
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float dq = d[ky][p + 2 - kx];
                    const f32x2_t dd = {dq, dq};
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[ky * 3 + kx][i] = __builtin_elementwise_fma(dd, xf[i], acc[ky * 3 + kx][i]);
                }
        }
    }
    // lanes with equal c8 inside the wave (pixel groups), then the waves through LDS, then one atomic per value
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float s = acc[t][i][h];
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x128, 0xF, 0xF, false));   // row_ror:8
                if constexpr (LPP == 4) {
                    asm("" : "+v"(s));
                    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x124, 0xF, 0xF, false));   // row_ror:4
                }
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (lane < LPP) red[wv][t * CIN + lane * 8 + 2 * i + h] = s;
            }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * CIN; i += NT) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NT / 64; ++q) s += red[q][i];
        mde_grad_add(dw + i, s, det);
    }
}

int cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return cus;
}

int head1_grid(int N, int H, int W, int lpp = 8) {          // persistent-ish: up to 8 workgroups per CU over the pixel runs
    const int64_t nruns = (int64_t)N * H * ((W + HR - 1) / HR);
    const int64_t nb = (nruns + NT / lpp - 1) / (NT / lpp), cap = 8 * (int64_t)cu_count();
    return (int)(nb > cap ? cap : (nb < 1 ? 1 : nb));
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

// Cout: 64 (one launch) or 96 (two launches of 48 channels: the weights of a launch are register resident)
extern "C" int mde_stem_conv_fwd_c(const float* x, const float* w, void* out, float* stats, int N, int H, int W, int Cout,
                                   void* stream) {
    MDE_REQUIRE(x && w && out && N > 0 && H > 0 && W > 0, "mde_stem_conv_fwd: bad argument");
    MDE_REQUIRE(Cout == 64 || Cout == 96, "mde_stem_conv_fwd: Cout=%d unsupported (64 or 96)", Cout);
    MDE_REQUIRE(((uintptr_t)out % 16) == 0, "mde_stem_conv_fwd: out must be 16-byte aligned");
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    MDE_REQUIRE((int64_t)N * 3 * H * W * 4 < MDE_OOB_OFFSET && (int64_t)N * OH * OW < (1 << 30),
                "mde_stem_conv_fwd: image tensor must be < 2 GiB");
    const int64_t ntiles = (int64_t)N * ((OH + FTY - 1) / FTY) * ((OW + STX - 1) / STX);
    MDE_REQUIRE((int64_t)3 * H * W < (1 << 27), "mde_stem_conv_fwd: image plane too large");
    const int64_t cap = cu_count();                         // persistent: one 8-wave workgroup per CU
    const int grid = (int)(ntiles > cap ? cap : ntiles);
    bf16_t* o = (bf16_t*)out;
    if (Cout == 64) {
        stem_fwd_k<4><<<grid, NTF, 0, (hipStream_t)stream>>>(x, w, o, stats, N, H, W, OH, OW, 64, 0, g_mde_det.on);
    } else {
        stem_fwd_k<3><<<grid, NTF, 0, (hipStream_t)stream>>>(x, w, o, stats, N, H, W, OH, OW, 96, 0, g_mde_det.on);
        stem_fwd_k<3><<<grid, NTF, 0, (hipStream_t)stream>>>(x, w + 48 * SK, o + 48, stats, N, H, W, OH, OW, 96, 48, g_mde_det.on);
    }
    MDE_LAUNCH_CHECK("stem_fwd_k");
    return MDE_OK;
}

extern "C" int mde_stem_conv_fwd(const float* x, const float* w, void* out, float* stats, int N, int H, int W,
                                 void* stream) {
    return mde_stem_conv_fwd_c(x, w, out, stats, N, H, W, 64, stream);
}

extern "C" int mde_stem_conv_wgrad_c(const float* x, const void* dout, float* dw, int N, int H, int W, int Cout, void* stream) {
    MDE_REQUIRE(x && dout && dw && N > 0 && H > 0 && W > 0, "mde_stem_conv_wgrad: bad argument");
    MDE_REQUIRE(Cout == 64 || Cout == 96, "mde_stem_conv_wgrad: Cout=%d unsupported (64 or 96)", Cout);
    MDE_REQUIRE(((uintptr_t)dout % 16) == 0, "mde_stem_conv_wgrad: dout must be 16-byte aligned");
    MDE_REQUIRE((int64_t)3 * H * W < (1 << 27), "mde_stem_conv_wgrad: image plane too large");
    MDE_REQUIRE((int64_t)N * 3 * H * W * 4 < MDE_OOB_OFFSET && (int64_t)N * (H / 2 + 1) * (W / 2 + 1) * 2 * Cout < MDE_OOB_OFFSET,
                "mde_stem_conv_wgrad: tensors must be < 2 GiB");
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    const int64_t ntiles = (int64_t)N * ((OH + WTY - 1) / WTY) * ((OW + STX - 1) / STX);
    const int64_t cap = 2 * (int64_t)cu_count();
    const int grid = (int)(ntiles > cap ? cap : ntiles);
    MDE_DET_REQUIRE("mde_stem_conv_wgrad", dw, (int64_t)Cout * 7 * 7 * 3);
    const bf16_t* d = (const bf16_t*)dout;
    if (Cout == 64) {
        stem_wgrad_k<4><<<grid, NT, 0, (hipStream_t)stream>>>(x, d, dw, N, H, W, OH, OW, 64, mde_det_dev());
    } else {
        stem_wgrad_k<3><<<grid, NT, 0, (hipStream_t)stream>>>(x, d, dw, N, H, W, OH, OW, 96, mde_det_dev());
        stem_wgrad_k<3><<<grid, NT, 0, (hipStream_t)stream>>>(x, d + 48, dw + 48 * SK, N, H, W, OH, OW, 96, mde_det_dev());
    }
    MDE_LAUNCH_CHECK("stem_wgrad_k");
    return MDE_OK;
}

extern "C" int mde_stem_conv_wgrad(const float* x, const void* dout, float* dw, int N, int H, int W, void* stream) {
    return mde_stem_conv_wgrad_c(x, dout, dw, N, H, W, 64, stream);
}

extern "C" int mde_head_conv_fwd(const void* x, const float* w, float* out, int N, int H, int W, int Cin, int Cout,
                                 void* stream) {
    MDE_REQUIRE(x && w && out, "mde_head_conv_fwd: null argument");
    if (int rc = head_check("mde_head_conv_fwd", N, H, W, Cin, Cout)) return rc;
    MDE_REQUIRE(((uintptr_t)x % 16) == 0, "mde_head_conv_fwd: x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (Cout == 1 && (Cin == 64 || Cin == 32) && (int64_t)N * H * W * Cin * 2 < MDE_OOB_OFFSET) {
        if (Cin == 64) head1_fwd_k<8><<<head1_grid(N, H, W), NT, 0, st>>>((const bf16_t*)x, w, out, N, H, W);
        else head1_fwd_k<4><<<head1_grid(N, H, W, 4), NT, 0, st>>>((const bf16_t*)x, w, out, N, H, W);
        MDE_LAUNCH_CHECK("head1_fwd_k");
        return MDE_OK;
    }
    return HEAD_DISPATCH(head_fwd_launch, x, w, out, N, H, W, Cin, Cout, st);
}

extern "C" int mde_head_conv_bwd(const void* x, const float* w, const float* dout, void* dx, float* dw, int N, int H,
                                 int W, int Cin, int Cout, void* stream) {
    MDE_REQUIRE(x && w && dout && (dx || dw), "mde_head_conv_bwd: null argument");
    if (int rc = head_check("mde_head_conv_bwd", N, H, W, Cin, Cout)) return rc;
    MDE_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dx % 16) == 0, "mde_head_conv_bwd: x/dx must be 16-byte aligned");
    MDE_DET_REQUIRE("mde_head_conv_bwd", dw, (int64_t)Cout * 9 * Cin);
    hipStream_t st = (hipStream_t)stream;
    if (Cout == 1 && (Cin == 64 || Cin == 32) && (int64_t)N * H * W * Cin * 2 < MDE_OOB_OFFSET) {
        const int lpp = Cin / 8;
        if (dx) {
            if (lpp == 8) head1_dgrad_k<8><<<head1_grid(N, H, W), NT, 0, st>>>(w, dout, (bf16_t*)dx, N, H, W);
            else head1_dgrad_k<4><<<head1_grid(N, H, W, 4), NT, 0, st>>>(w, dout, (bf16_t*)dx, N, H, W);
            MDE_LAUNCH_CHECK("head1_dgrad_k");
        }
        if (dw) {
            const int g = head1_grid(N, H, W, lpp), gc = g > 1024 ? 1024 : g;
            if (lpp == 8) head1_wgrad_k<8><<<gc, NT, 0, st>>>((const bf16_t*)x, dout, dw, N, H, W, mde_det_dev());
            else head1_wgrad_k<4><<<gc, NT, 0, st>>>((const bf16_t*)x, dout, dw, N, H, W, mde_det_dev());
            MDE_LAUNCH_CHECK("head1_wgrad_k");
        }
        return MDE_OK;
    }
    if (dx) {
        int rc = HEAD_DISPATCH(head_dgrad_launch, w, dout, dx, N, H, W, Cin, Cout, st);
        if (rc) return rc;
    }
    if (dw) {
        const int grid = px_grid((int64_t)N * H * W, NT / (Cin / 8));
        const int g = grid > 1024 ? 1024 : grid;
        if (Cout == 1) {
            head_wgrad_k<1><<<g, NT, 0, st>>>((const bf16_t*)x, dout, dw, N, H, W, Cin, Cout, 0, mde_det_dev());
            MDE_LAUNCH_CHECK("head_wgrad_k");
        } else {
            for (int co0 = 0; co0 < Cout; co0 += 2) {
                head_wgrad_k<2><<<g, NT, 0, st>>>((const bf16_t*)x, dout, dw, N, H, W, Cin, Cout, co0, mde_det_dev());
                MDE_LAUNCH_CHECK("head_wgrad_k");
            }
        }
    }
    return MDE_OK;
}
