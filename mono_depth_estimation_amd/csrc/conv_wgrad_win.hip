// Weight-gradient GEMM, windowed form, for multi-tap convolutions on gfx950.
//
//   dw[r][otap[t]][c] += sum_pix direct[pix][.] x gathered[src(pix, t)][.]            (mde_conv_wgrad's contract)
//
// conv_wgrad.hip gives every tap its own workgroups: a K-step (64 pixels) of a 128 x 128 tile moves 32 KB through the
// LDS-DMA path for 2.1 MFLOP, and that path (one 1-KiB instruction per ~20 cycles per CU) is what bounds it, as it bounded
// the plain forward tiles (conv_gemm.hip, "halo-tiled form").  Here ONE workgroup owns the taps of a kernel ROW (same dy,
// up to three dx): the 64 pixels of a K-step are a 4 x 16 block of one image, the direct operand's tile is fetched once
// and used by all the row's taps, and the gathered operand is a WINDOW of 4 x (16 + dx span) pixels that the taps read at
// shifted rows.  A 3x3 moves 25 KB per 3.1 MFLOP (128 direct x 64 gathered channels x 3 taps): half the bytes per flop.
// 8 waves, two workgroups per CU, K-step s+1 fetched under the MFMAs of step s (two staging buffers).
//
// MEASURED (round 3, in-network, FCRN-50 step, `bench.py --per-shape` with MDE_WGRAD_WIN=0 against =1; gpurun_out/r3_ww*.txt):
// this kernel is SLOWER than the per-tap one on every multi-tap shape of the step -- +22 % (614 400 px, 128 x 128 ch, 9 taps:
// 261 -> 318 us) to +110 % (9 600 px, 1024 x 1024 ch, 25 taps: 622 -> 1 309 us); weight-gradient time 8.00 -> 11.59 ms per
// step, no shape faster.  The byte argument above holds and did not decide it: a wave here issues 8 T MFMAs per 16
// transposed-read pairs where the per-tap kernel's 128 x 128 tile issues 32 (the DMA bytes were traded for LDS reads), the
// stride-2 windows of the up-projection layers take 2-way bank conflicts on every read, and a 5-tap row runs as 3 + 2 with
// the third slot of the short group computed and thrown away.  It is correct (tests/test_conv_wgrad_gpu.py runs every case
// through both kernels) and is kept as a diagnostic: mde_conv_wgrad calls it only under MDE_WGRAD_WIN=1.
//
// Tiles are row-major [pixel][channel] and consumed with ds_read_b64_tr_b16 exactly as in conv_wgrad.hip (same swizzles);
// the window's rows are addressed through per-lane offsets computed once per workgroup (they do not depend on the K-step).
#include <stdlib.h>

#include "mde_common.h"

namespace {

constexpr int NT = 512;
constexpr int BD = 128, BG = 64;           // direct / gathered channels per tile
constexpr int BH = 4, BWX = 16;            // pixel block of a K-step
constexpr int DT_BYTES = 64 * BD * 2;      // direct tile
constexpr int MAX_WROWS = 144;             // window rows (pixels) a buffer holds
constexpr int MAX_GROUPS = 12;

struct WArgs {
    mde_wgrad_desc d;
    const void* direct;
    const void* gathered;
    float* dw;
    int32_t nD, nG;               // tiles along the direct / gathered channels
    int32_t Crows, Ccols;
    int32_t ngroups;              // tap groups (taps of one kernel row, at most three)
    int32_t g_dy[MAX_GROUPS], g_dx0[MAX_GROUPS], g_cnt[MAX_GROUPS];
    int32_t g_dx[MAX_GROUPS][3], g_otap[MAX_GROUPS][3];
    int32_t ww, wrows;            // window width (pixels), rows in all = BH * ww
    int32_t nby, nbx, nblocks;    // 4 x 16 pixel blocks per image column / row, in all
    int32_t bchunk;               // blocks per split-K slice
    MdeDetDev det;
};

template <int CH>
__device__ __forceinline__ int tile_off(int row, int ch) {      // as conv_wgrad.hip
    if constexpr (CH == 128) return row * 256 + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    else return row * 128 + 16 * (ch ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1));
}

__device__ __forceinline__ bf16x8_t read_tr_pair(const char* lo, const char* hi) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)lo);
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)hi);
    union { struct { s16x4_t a, b; } s; bf16x8_t v; } u;
    u.s.a = a;
    u.s.b = b;
    return u.v;
}

// T = taps per group (2 or 3).  GA: the gathered tensor's channels are the rows of dw.
template <int T, bool GA>
__global__ __launch_bounds__(NT, 4) void conv_wgrad_win(const WArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const mde_wgrad_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wv >> 1, wg = wv & 1;                       // wave position along direct / gathered channels
    const int wbytes = ((a.wrows + 7) >> 3) * 1024;
    const int buf_bytes = DT_BYTES + wbytes;

    // block -> (group, gathered tile, direct tile, split-K slice): groups fastest, neighbours share operand tiles in L2
    uint32_t b = mde_xcd_remap(blockIdx.x, gridDim.x);
    const int grp = b % a.ngroups; b /= a.ngroups;
    const int tg = b % a.nG; b /= a.nG;
    const int td = b % a.nD; b /= a.nD;
    const int ks = b;
    const int d0 = td * BD, g0 = tg * BG;
    const int blk0 = ks * a.bchunk, blk1 = min(a.nblocks, blk0 + a.bchunk);
    if (blk0 >= blk1) return;
    const int nsteps = blk1 - blk0;
    const int dyg = a.g_dy[grp], dx0 = a.g_dx0[grp];

    const __amdgpu_buffer_rsrc_t rs_d = mde_rsrc(a.direct, d.d_bytes);
    const __amdgpu_buffer_rsrc_t rs_g = mde_rsrc(a.gathered, d.g_bytes);
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // ---- DMA lanes.  Direct tile: 16 pieces of 4 rows x 256 B, two per wave; piece p = wv + 8 q holds rows 4 p .. 4 p + 3.
    const int a_lr = lane >> 4;
    const int a_cs = d0 + ((lane & 15) ^ ((a_lr << 2) | (wv & 3))) * 8;               // source channel of this lane's chunk
    const int dr = wv * 4 + a_lr;                                                      // tile row of piece q = 0 (q = 1: + 32)
    const int d_ry = dr >> 4, d_rx = dr & 15;
    const uint32_t d_delta = (uint32_t)((d_ry * d.GW + d_rx) * d.ld_d + a_cs) * 2u;
    const bool d_cok = a_cs < d.Cd;
    // Window: pieces of 8 rows x 128 B, up to three per wave; piece p = wv + 8 q holds window rows 8 p .. 8 p + 7.
    const int b_lr = lane >> 3;
    const int b_cs = g0 + ((lane & 7) ^ ((((b_lr >> 1) & 1) | ((wv & 1) << 1)) << 1)) * 8;
    const bool g_cok = b_cs < d.Cg;
    const int npieces = (a.wrows + 7) >> 3;
    int w_pos[3];                                                                      // window row | window column << 3 (one register)
    uint32_t w_delta[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int h = (wv + 8 * q) * 8 + b_lr;
        const int wy = h / a.ww, wx = h - wy * a.ww;
        w_delta[q] = (uint32_t)((wy * d.sy * d.W + wx) * d.ld_g + b_cs) * 2u;
        w_pos[q] = h < a.wrows ? (wy | (wx << 3)) : (1 << 27);                         // past the window: never inside the tensor
    }

    // ---- fragment read offsets (bytes inside a buffer), fixed for the whole workgroup
    const int fi = lane & 15, fg = lane >> 4, fq = fi >> 2, fp = fi & 3;
    const int sub = (fp & 1) * 8;
    int doff[2][2];                           // direct tile: [k half][lo / hi], fragment i = 0 (i = 1: ^ 32)
    int woff[T][2][2];                        // window: [tap][k half][lo / hi], fragment j = 0 (j = 1: ^ 32)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int k = kk * 32 + 8 * fg + fq + 4 * hi;                              // pixel of the block, raster order
            doff[kk][hi] = tile_off<128>(k, wd * 4 + (fp >> 1)) + sub;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int h = (k >> 4) * a.ww + (k & 15) * d.sx + (a.g_dx[grp][t] - dx0);
                woff[t][kk][hi] = DT_BYTES + tile_off<64>(h, wg * 4 + (fp >> 1)) + sub;
            }
        }

    f32x4_t acc[T][2][2];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- block cursor of the next K-step to fetch
    const int bpi = a.nby * a.nbx;
    int ln = blk0 / bpi, lby = (blk0 - ln * bpi) / a.nbx, lbx = blk0 - ln * bpi - lby * a.nbx;
    auto issue = [&](int buf) {
        char* dst = smem + buf * buf_bytes;
        const int gy0 = lby * BH, gx0 = lbx * BWX;
        const uint32_t dbase = (uint32_t)(((ln * d.GH + gy0) * d.GW + gx0) * d.ld_d) * 2u;
        const bool xok = d_cok & (gx0 + d_rx < d.GW);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool ok = xok & (gy0 + d_ry + 2 * q < d.GH);
            const uint32_t off = ok ? dbase + d_delta + (uint32_t)(2 * q * d.GW * d.ld_d) * 2u : MDE_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lds_ptr)(dst + (wv + 8 * q) * 1024), 16, off, 0, 0, 0);
        }
        const int iy0 = gy0 * d.sy + dyg, ix0 = gx0 * d.sx + dx0;
        // (the base may lie before the tensor: computed in 64 bits, used only where the pixel itself is inside)
        const int64_t gbase = ((int64_t)(ln * d.H + iy0) * d.W + ix0) * d.ld_g * 2;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (wv + 8 * q < npieces) {                       // wave-uniform
                const int iy = iy0 + (w_pos[q] & 7) * d.sy, ix = ix0 + (w_pos[q] >> 3);
                const bool ok = g_cok & ((uint32_t)iy < (uint32_t)d.H) & ((uint32_t)ix < (uint32_t)d.W);
                const uint32_t off = ok ? (uint32_t)(gbase + (int64_t)w_delta[q]) : MDE_OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr)(dst + DT_BYTES + (wv + 8 * q) * 1024), 16, off, 0, 0, 0);
            }
        if (++lbx == a.nbx) { lbx = 0; if (++lby == a.nby) { lby = 0; ++ln; } }
    };

    issue(0);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // step s is in; everyone left step s-1
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nsteps) issue((s + 1) & 1);
        const char* base = smem + (s & 1) * buf_bytes;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fd[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fd[i] = read_tr_pair(base + (doff[kk][0] ^ (i << 5)), base + (doff[kk][1] ^ (i << 5)));
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bf16x8_t fw[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) fw[j] = read_tr_pair(base + (woff[t][kk][0] ^ (j << 5)), base + (woff[t][kk][1] ^ (j << 5)));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = GA ? MDE_MFMA_16x16x32(fw[j], fd[i], acc[t][i][j])
                                          : MDE_MFMA_16x16x32(fd[i], fw[j], acc[t][i][j]);
            }
            if constexpr (T > 2) __builtin_amdgcn_sched_barrier(0);   // (keeps the second half's fragments out of the first's registers: no spills at 128)
        }
    }

    // ---- epilogue: fp32 atomic accumulation into dw[row][otap][col]
    const size_t rstride = (size_t)d.otaps_total * a.Ccols;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (t >= a.g_cnt[grp]) break;
        const int otap = a.g_otap[grp][t];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int dch = d0 + wd * 32 + i * 16, gch = g0 + wg * 32 + j * 16;
                const int col = (GA ? dch : gch) + (lane & 15);
                const int rbase = (GA ? gch : dch) + (lane >> 4) * 4;
                if (col < a.Ccols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rbase + r < a.Crows)
                            mde_grad_add(a.dw + (size_t)(rbase + r) * rstride + (size_t)otap * a.Ccols + col, acc[t][i][j][r], a.det);
                }
            }
    }
}

}  // namespace

// Called by mde_conv_wgrad (conv_wgrad.hip) after its argument checks.  Returns 1 when the windowed form took the launch,
// 0 when the geometry is not eligible (the caller then runs the per-tap form), < 0 on error.
int mde_conv_wgrad_windowed(const mde_wgrad_desc* d, const void* direct, const void* gathered, float* dw, hipStream_t st, int force) {
    if (d->group_size || d->ntaps < 2) return 0;
    WArgs wa;
    // group the taps by dy, at most three per group (a 5-tap row becomes 3 + 2), each group's dx sorted ascending
    int order[MDE_MAX_TAPS];
    for (int t = 0; t < d->ntaps; ++t) order[t] = t;
    for (int i = 1; i < d->ntaps; ++i)
        for (int j = i; j > 0; --j) {
            const int x = order[j], y = order[j - 1];
            if (d->dy[x] < d->dy[y] || (d->dy[x] == d->dy[y] && d->dx[x] < d->dx[y])) { order[j] = y; order[j - 1] = x; } else break;
        }
    int ng = 0, maxcnt = 0, maxspan = 0;
    for (int i = 0; i < d->ntaps;) {
        int cnt = 1;
        while (i + cnt < d->ntaps && cnt < 3 && d->dy[order[i + cnt]] == d->dy[order[i]]) ++cnt;
        if (ng == MAX_GROUPS) return 0;
        wa.g_dy[ng] = d->dy[order[i]];
        wa.g_dx0[ng] = d->dx[order[i]];
        wa.g_cnt[ng] = cnt;
        for (int t = 0; t < 3; ++t) {
            const int src = order[i + (t < cnt ? t : cnt - 1)];          // (unused slots repeat the last tap; never stored)
            wa.g_dx[ng][t] = d->dx[src];
            wa.g_otap[ng][t] = d->otap[src];
        }
        const int span = d->dx[order[i + cnt - 1]] - d->dx[order[i]];
        maxspan = span > maxspan ? span : maxspan;
        maxcnt = cnt > maxcnt ? cnt : maxcnt;
        i += cnt;
        ++ng;
    }
    if (maxcnt < 2) return 0;
    const int ww = (BWX - 1) * d->sx + maxspan + 1, wrows = BH * ww;
    if (wrows > MAX_WROWS || d->sx < 1 || d->sy < 1) return 0;
    const bool ga = d->rows_from_gathered != 0;
    wa.d = *d;
    wa.direct = direct;
    wa.gathered = gathered;
    wa.dw = dw;
    wa.Crows = ga ? d->Cg : d->Cd;
    wa.Ccols = ga ? d->Cd : d->Cg;
    wa.nD = mde_cdiv(d->Cd, BD);
    wa.nG = mde_cdiv(d->Cg, BG);
    wa.ngroups = ng;
    wa.ww = ww;
    wa.wrows = wrows;
    wa.nby = mde_cdiv(d->GH, BH);
    wa.nbx = mde_cdiv(d->GW, BWX);
    wa.nblocks = d->N * wa.nby * wa.nbx;
    wa.det = mde_det_dev();
    // Tile efficiency: blocks hang over the grid's edge (their pixels read as zero) and channel tiles over the channel counts
    const double fill = ((double)d->N * d->GH * d->GW / ((double)wa.nblocks * 64)) * ((double)d->Cd / (wa.nD * BD)) * ((double)d->Cg / (wa.nG * BG));
    if (!force && (fill < 0.7 || wa.nblocks < 64)) return 0;
    // split-K: about two rounds of the chip's 512 resident workgroups, at least 8 K-steps per workgroup
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int64_t base = (int64_t)ng * wa.nD * wa.nG;
    int64_t ks = (4 * cus + base - 1) / base;
    if (const char* e = getenv("MDE_WGRAD_WIN_KS")) ks = atoi(e);
    const int64_t cap = wa.nblocks / 8 > 1 ? wa.nblocks / 8 : 1;
    ks = ks < 1 ? 1 : ks > cap ? cap : ks;
    wa.bchunk = (int32_t)((wa.nblocks + ks - 1) / ks);
    const int kslices = mde_cdiv(wa.nblocks, wa.bchunk);
    const int64_t nblk = base * kslices;
    MDE_REQUIRE(nblk < (1ll << 31), "mde_conv_wgrad: grid too large");
    const size_t smem = 2 * ((size_t)DT_BYTES + (size_t)((wrows + 7) >> 3) * 1024);
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[4] = {reinterpret_cast<const void*>(&conv_wgrad_win<2, false>), reinterpret_cast<const void*>(&conv_wgrad_win<2, true>),
                              reinterpret_cast<const void*>(&conv_wgrad_win<3, false>), reinterpret_cast<const void*>(&conv_wgrad_win<3, true>)};
        for (const void* f : fns) {
            int rc = mde_check_hip(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (DT_BYTES + MAX_WROWS / 8 * 1024)),
                                   "hipFuncSetAttribute(conv_wgrad_win)");
            if (rc) return rc;
        }
        attr_done = true;
    }
    if (maxcnt == 2) {
        if (ga) conv_wgrad_win<2, true><<<dim3((unsigned)nblk), dim3(NT), smem, st>>>(wa);
        else conv_wgrad_win<2, false><<<dim3((unsigned)nblk), dim3(NT), smem, st>>>(wa);
    } else {
        if (ga) conv_wgrad_win<3, true><<<dim3((unsigned)nblk), dim3(NT), smem, st>>>(wa);
        else conv_wgrad_win<3, false><<<dim3((unsigned)nblk), dim3(NT), smem, st>>>(wa);
    }
    MDE_LAUNCH_CHECK("conv_wgrad_win");
    return 1;
}
