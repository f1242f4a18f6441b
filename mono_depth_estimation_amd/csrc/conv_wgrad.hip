// Weight-gradient GEMM on MFMA for gfx950 — the "TN" form: both operands are pixel-major
// (NHWC bf16) and the contraction runs over PIXELS, so each MFMA fragment needs 8
// consecutive pixels of one channel: a transposed read.  The tiles are staged row-major
// ([pixel][channel], full 128/256-byte lines from HBM) and consumed with
// ds_read_b64_tr_b16 (hardware transpose), XOR-swizzled so both the 16-byte staging writes
// and the transposed reads are bank-conflict-free.
//
//   dw[r][otap[t]][c] += sum_{pix} A[pix or src(pix,t)][r] * B[src(pix,t) or pix][c]
//
// One workgroup (256 threads, 2x2 waves) owns a BA x BB tile of one tap and one slice of
// the pixel range (split-K); partial tiles are added with fp32 global atomics.
// Replaces the autograd weight-gradient of every nn.Conv2d on the FCRN path (reference
// network/FCRN.py:180-188,334 and the torchvision Bottleneck convs) including the
// up-projection 5x5, whose zero-stuffed input (FCRN.py:31-44) becomes a stride-2 gather on
// the dY side.
#include <stdlib.h>

#include "mde_common.h"

int mde_conv_wgrad_windowed(const mde_wgrad_desc* d, const void* direct, const void* gathered, float* dw, hipStream_t st, int force);

namespace {

constexpr int BKP = 64;   // pixels per K-step
constexpr int NT = 256;

struct KArgs {
    mde_wgrad_desc d;
    const void* direct;
    const void* gathered;
    float* dw;
    int32_t M;            // N*GH*GW
    int32_t kchunk;       // pixels per split-K slice (multiple of 64)
    int32_t nA, nB;       // tiles along rows / cols
    int32_t Crows, Ccols;
    int32_t gsize;        // grouped convolution: channels per group (0 = dense)
    MdeDetDev det;        // deterministic mode: partial tiles are added as integers into the gradient's int64 shadow
    uint32_t inv_gw, inv_ghw;
    int32_t row_off;      // first row of this launch's tile grid (a row range may be split over two launches with different tile heights)
    int32_t skip_store;   // diagnostics (MDE_WGRAD_NOSTORE=1): the epilogue's atomics are skipped (timing only: results are wrong)
    // two-stage reduction (mde_conv_wgrad_ws): every workgroup STORES its partial tile into slice `ks` of the workspace,
    // fp32 [kslices][Crows][ntaps][Ccols] (tap index of this launch, not the output slot), and wgrad_reduce_k adds the slices
    // into dw.  nullptr: fp32 atomics straight into dw.
    float* ws;
};

// byte offset of 16-byte chunk `ch` of row `row` in a [64][CH] bf16 tile, CH = 128, 64 or 32 (32: row tiles only)
template <int CH>
__device__ __forceinline__ int tile_off(int row, int ch) {
    if constexpr (CH == 128) {
        return row * 256 + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    } else if constexpr (CH == 32) {
        // 64-byte rows: four of them span the banks once, so the four pixel groups of a transposed read (rows 8 apart) would
        // meet on the same banks; bit 3 of the row swaps the two 32-byte halves
        return row * 64 + 16 * (ch ^ (((row >> 3) & 1) << 1));
    } else {
        return row * 128 + 16 * (ch ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1));
    }
}

// One 16(channel) x 32(pixel) MFMA operand fragment out of a row-major [pixel][channel] tile:
// two transposed reads of 4 pixel-rows x 16 channels each (guide T10).
template <int CH>
__device__ __forceinline__ bf16x8_t read_frag_tr(const char* tile, int k0, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int q = i >> 2, p = i & 3;
    const int row = k0 + 8 * g + q;
    const int ch = (c0 >> 3) + (p >> 1);
    const int sub = (p & 1) * 8;
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)(tile + tile_off<CH>(row, ch) + sub));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)(tile + tile_off<CH>(row + 4, ch) + sub));
    union { struct { s16x4_t a, b; } s; bf16x8_t v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

// GA: the gathered tensor supplies the A operand (rows of dw); otherwise the direct one does.
// DMA: tiles are filled by LDS-DMA (buffer_load ... lds; one wave-instruction = 1 KiB = 4 rows x 256 B
// or 8 rows x 128 B) instead of VGPR staging + ds_write; the destination is lane-linear, so the XOR
// chunk swizzle of tile_off<> is applied to the SOURCE chunk each lane fetches.
// NBUF = 1: ONE staging buffer (fetch -> wait -> MFMA per K-step), so twice as many workgroups fit a CU and overlap each
// other's fetches and MFMAs (conv_gemm.hip's single-buffer tiles: occupancy beat the ring there on nearly every shape).
// BA2 = 32 / 64: the tile carries 32 / 64 MORE rows behind its 128 (a second operand tile next to the first: one / two more
// fragments per wave against the same column fragments) -- a row count of 128 + a remainder <= 64 (VNL's 152-row prediction
// conv; DenseNet's 192-channel bottlenecks) in ONE launch instead of a 128-row launch and a second launch that stages the whole
// column operand again for a quarter / a half of the MFMAs.
template <int BA, int BB, bool GA, bool DMA, int NBUF = 2, int BA2 = 0>
__global__ __launch_bounds__(NT, NBUF == 1 ? 4 : 2) void conv_wgrad_tn(const KArgs a) {
    static_assert(NBUF == 2 || (NBUF == 1 && DMA), "single-buffer form: LDS-DMA loop only");
    static_assert(BA2 == 0 || ((BA2 == 32 || BA2 == 64) && BA == 128 && DMA && NBUF == 2), "the extra rows ride on the 128-row ring tile");
    constexpr int FA2 = BA2 / 32;                  // extra 16-row fragments per wave
    constexpr int FA = BA / 32, FB = BB / 32;      // 16-wide fragments per wave along rows / cols
    constexpr int AT_BYTES = BKP * BA * 2, BT_BYTES = BKP * BB * 2;
    constexpr int AT2_BYTES = BKP * BA2 * 2;       // the extra rows' tile, behind the first operand tile
    constexpr int B_OFF = AT_BYTES + AT2_BYTES;
    constexpr int BUF_BYTES = B_OFF + BT_BYTES;
    constexpr int CPR_A = BA / 8, CPR_B = BB / 8;  // 16-byte chunks per tile row
    constexpr int RPP_A = NT / CPR_A, RPP_B = NT / CPR_B;
    constexpr int PA = BKP / RPP_A, PB = BKP / RPP_B;  // loads per thread per step

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const mde_wgrad_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;

    // block -> (tap, tileB, tileA, kslice), tap fastest so neighbours share operand tiles; the XCD remap
    // keeps those neighbours on ONE XCD (one L2) instead of dealing them round-robin over the eight
    uint32_t b = mde_xcd_remap(blockIdx.x, gridDim.x);
    const int tap = b % d.ntaps; b /= d.ntaps;
    const int tb = b % a.nB; b /= a.nB;
    const int ta = b % a.nA; b /= a.nA;
    const int ks = b;
    // grouped (BA == BB == 64, nB == 1): the tile on the diagonal, rows and columns are the same 64-channel window
    const int row0 = a.row_off + ta * BA, col0 = a.gsize ? row0 : tb * BB;
    const int kbeg = ks * a.kchunk;
    const int kend = min(a.M, kbeg + a.kchunk);
    if (kbeg >= kend) return;
    const int nsteps = (kend - kbeg + BKP - 1) / BKP;

    const int tdy = d.dy[tap], tdx = d.dx[tap];
    const __amdgpu_buffer_rsrc_t rs_d = mde_rsrc(a.direct, d.d_bytes);
    const __amdgpu_buffer_rsrc_t rs_g = mde_rsrc(a.gathered, d.g_bytes);

    const int a_chunk = tid % CPR_A, a_row = tid / CPR_A;
    const int b_chunk = tid % CPR_B, b_row = tid / CPR_B;
    // channel offsets inside the source tensors
    const int a_c = row0 + a_chunk * 8, b_c = col0 + b_chunk * 8;

    int a_st[PA], b_st[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p) a_st[p] = tile_off<BA>(a_row + p * RPP_A, a_chunk);
#pragma unroll
    for (int p = 0; p < PB; ++p) b_st[p] = tile_off<BB>(b_row + p * RPP_B, b_chunk);

    // (channel chunks past the tensor's channel count -- tiles of a count that is not a multiple of 64 -- read as zero)
    auto direct_off = [&](int m, int c) -> uint32_t {
        return ((m < kend) & (c < d.Cd)) ? (uint32_t)(m * d.ld_d + c) * 2u : MDE_OOB_OFFSET;
    };
    auto gathered_off = [&](int m, int c) -> uint32_t {
        const uint32_t n = mde_fastdiv((uint32_t)m, (uint32_t)(d.GH * d.GW), a.inv_ghw);
        const uint32_t rem = (uint32_t)m - n * (uint32_t)(d.GH * d.GW);
        const uint32_t gy = mde_fastdiv(rem, (uint32_t)d.GW, a.inv_gw);
        const uint32_t gx = rem - gy * (uint32_t)d.GW;
        const int iy = (int)gy * d.sy + tdy, ix = (int)gx * d.sx + tdx;
        const bool ok = (m < kend) & ((uint32_t)iy < (uint32_t)d.H) & ((uint32_t)ix < (uint32_t)d.W) & (c < d.Cg);
        return ok ? (uint32_t)((((int)n * d.H + iy) * d.W + ix) * d.ld_g + c) * 2u : MDE_OOB_OFFSET;
    };

    i32x4_t ar[PA], br[PB];
    auto issue_loads = [&](int s) {
        const int mb = kbeg + s * BKP;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int m = mb + a_row + p * RPP_A;
            ar[p] = GA ? __builtin_amdgcn_raw_buffer_load_b128(rs_g, gathered_off(m, a_c), 0, 0)
                       : __builtin_amdgcn_raw_buffer_load_b128(rs_d, direct_off(m, a_c), 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int m = mb + b_row + p * RPP_B;
            br[p] = GA ? __builtin_amdgcn_raw_buffer_load_b128(rs_d, direct_off(m, b_c), 0, 0)
                       : __builtin_amdgcn_raw_buffer_load_b128(rs_g, gathered_off(m, b_c), 0, 0);
        }
    };
    auto stage_write = [&](int buf) {
        char* at = smem + buf * BUF_BYTES;
        char* bt = at + B_OFF;
#pragma unroll
        for (int p = 0; p < PA; ++p) *reinterpret_cast<i32x4_t*>(at + a_st[p]) = ar[p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *reinterpret_cast<i32x4_t*>(bt + b_st[p]) = br[p];
    };

    f32x4_t acc[FA][FB];
    f32x4_t acc2[BA2 ? FA2 : 1][BA2 ? FB : 1];     // rows row0 + BA + wa * (BA2 / 2) + i * 16 .. + 16 (BA2)
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (BA2 ? FA2 : 1); ++i)
#pragma unroll
        for (int j = 0; j < (BA2 ? FB : 1); ++j) acc2[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char* at = smem + buf * BUF_BYTES;
        const char* bt = at + B_OFF;
#pragma unroll
        for (int kk = 0; kk < BKP; kk += 32) {
            bf16x8_t fa[FA], fb[FB];
#pragma unroll
            for (int i = 0; i < FA; ++i) fa[i] = read_frag_tr<BA>(at, kk, wa * (BA / 2) + i * 16, lane);
#pragma unroll
            for (int j = 0; j < FB; ++j) fb[j] = read_frag_tr<BB>(bt, kk, wb * (BB / 2) + j * 16, lane);
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < FB; ++j)
                    acc[i][j] = MDE_MFMA_16x16x32(fa[i], fb[j], acc[i][j]);
            if constexpr (BA2 != 0) {
#pragma unroll
                for (int i = 0; i < FA2; ++i) {
                    const bf16x8_t f2 = read_frag_tr<BA2 ? BA2 : 32>(at + AT_BYTES, kk, wa * (BA2 / 2) + i * 16, lane);
#pragma unroll
                    for (int j = 0; j < FB; ++j) acc2[i][j] = MDE_MFMA_16x16x32(f2, fb[j], acc2[i][j]);
                }
            }
        }
    };

    if constexpr (DMA) {
        constexpr int RPI_A = 1024 / (BA * 2), RPI_B = 1024 / (BB * 2);   // tile rows per DMA piece (4 or 8)
        constexpr int QA = BKP / RPI_A / 4, QB = BKP / RPI_B / 4;         // pieces per wave per K-step
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        // this lane's tile-row inside a piece and the source chunk that lands at its lane-linear slot
        const int a_lr = lane / CPR_A, b_lr = lane / CPR_B;
        int a_sw, b_sw;
        if constexpr (BA == 128) a_sw = (a_lr << 2) | (wv & 3);
        else if constexpr (BA == 32) a_sw = ((a_lr >> 3) & 1) << 1;            // (16 rows per piece: row = 16 wv + a_lr)
        else a_sw = (((a_lr >> 1) & 1) | ((wv & 1) << 1)) << 1;
        if constexpr (BB == 128) b_sw = (b_lr << 2) | (wv & 3); else b_sw = (((b_lr >> 1) & 1) | ((wv & 1) << 1)) << 1;
        const int a_cs = row0 + ((lane % CPR_A) ^ a_sw) * 8, b_cs = col0 + ((lane % CPR_B) ^ b_sw) * 8;
        // (BA2 = 32: 16 rows of 64 bytes per piece, one piece per wave; 64: 8 rows of 128 bytes, two pieces; the swizzle of
        //  tile_off<BA2> on the source chunk, as for the first operand tile)
        constexpr int CPR_A2 = BA2 ? BA2 / 8 : 4, RPI_A2 = 1024 / (CPR_A2 * 16), QA2 = BKP / RPI_A2 / 4;
        const int a2_lr = lane / CPR_A2;
        const int a2_sw = BA2 == 64 ? (((a2_lr >> 1) & 1) | ((wv & 1) << 1)) << 1 : ((a2_lr >> 3) & 1) << 1;
        const int a2_cs = row0 + BA + ((lane % CPR_A2) ^ a2_sw) * 8;
        typedef __attribute__((address_space(3))) void* lds_ptr;
        // Gathered-operand addressing: the 64 pixels of a K-step are decoded ONCE per workgroup (one
        // wave, one pixel per lane, two steps ahead, waves taking turns) into s_goff[step&1][64] =
        // byte offset of the pixel's channel 0, or an out-of-range offset.  Decoding per lane and per
        // DMA made this loop VALU-bound (~170 VALU per wave per K-step against 32 MFMAs).
        uint32_t* s_goff = reinterpret_cast<uint32_t*>(smem + NBUF * BUF_BYTES);
        auto decode_step = [&](int s) {
            const uint32_t o = gathered_off(kbeg + s * BKP + lane, 0);
            s_goff[(s & 1) * BKP + lane] = o;
        };
        auto issue_dma = [&](int s, int buf) {
            const int mb = kbeg + s * BKP;
            char* at = smem + buf * BUF_BYTES + wv * 1024;
            const uint32_t* go = s_goff + (s & 1) * BKP;
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                const int r = (wv + 4 * q) * RPI_A + a_lr;
                // (an out-of-range pixel stays out of range after adding the small channel offset)
                const uint32_t off = GA ? (a_cs < d.Cg ? go[r] + (uint32_t)a_cs * 2u : MDE_OOB_OFFSET) : direct_off(mb + r, a_cs);
                if (GA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr)(at + q * 4096), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lds_ptr)(at + q * 4096), 16, off, 0, 0, 0);
            }
            if constexpr (BA2 != 0) {
#pragma unroll
                for (int q = 0; q < QA2; ++q) {
                    const int r = (wv + 4 * q) * RPI_A2 + a2_lr;
                    const uint32_t off = GA ? (a2_cs < d.Cg ? go[r] + (uint32_t)a2_cs * 2u : MDE_OOB_OFFSET) : direct_off(mb + r, a2_cs);
                    if (GA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr)(at + AT_BYTES + q * 4096), 16, off, 0, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lds_ptr)(at + AT_BYTES + q * 4096), 16, off, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int r = (wv + 4 * q) * RPI_B + b_lr;
                const uint32_t off = GA ? direct_off(mb + r, b_cs) : (b_cs < d.Cg ? go[r] + (uint32_t)b_cs * 2u : MDE_OOB_OFFSET);
                if (GA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lds_ptr)(at + B_OFF + q * 4096), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr)(at + B_OFF + q * 4096), 16, off, 0, 0, 0);
            }
        };
        if (wv == 0) decode_step(0);
        if (wv == 1 && nsteps > 1) decode_step(1);
        __syncthreads();
        if constexpr (NBUF == 1) {
            for (int s = 0; s < nsteps; ++s) {
                if (s) __builtin_amdgcn_s_barrier();        // everyone left step s-1: the buffer and offset slot (s+1)&1 are free
                issue_dma(s, 0);
                if (s >= 1 && s + 1 < nsteps && wv == (s & 3)) decode_step(s + 1);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                compute(0);
            }
        } else {
        issue_dma(0, 0);
        for (int s = 0; s < nsteps; ++s) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();     // pieces of step s landed, offsets of step s+1 visible, everyone left step s-1
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < nsteps) issue_dma(s + 1, (s + 1) & 1);
            if (s + 2 < nsteps && wv == (s & 3)) decode_step(s + 2);   // slot (s&1) was last read before this barrier
            compute(s & 1);
        }
        }
    } else {
        issue_loads(0);
        stage_write(0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const bool more = s + 1 < nsteps;
            if (more) issue_loads(s + 1);
            compute(s & 1);
            if (more) stage_write((s & 1) ^ 1);
            __syncthreads();
        }
    }

    // epilogue: fp32 atomic accumulation into dw[row][otap][col]
    if (a.skip_store) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if constexpr (BA2 != 0) keep += acc2[0][0][0];
        if (keep == 123456.789f) a.dw[0] = keep;      // (keeps the accumulators alive)
        return;
    }
    if (a.ws) {
        // plain stores: the L2 atomic units add one dword per clock and channel (~1.1 TB/s chip-wide), stores
        // and the summing pass run at the streaming rate
        float* wsl = a.ws + ((size_t)ks * a.Crows * d.ntaps + tap) * a.Ccols;
        const size_t wstride = (size_t)d.ntaps * a.Ccols;
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const int col = col0 + wb * (BB / 2) + j * 16 + (lane & 15);
                const int rbase = row0 + wa * (BA / 2) + i * 16 + (lane >> 4) * 4;
                if (col < a.Ccols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rbase + r < a.Crows) wsl[(size_t)(rbase + r) * wstride + col] = acc[i][j][r];
                }
            }
        if constexpr (BA2 != 0) {
#pragma unroll
            for (int i = 0; i < FA2; ++i)
#pragma unroll
                for (int j = 0; j < FB; ++j) {
                    const int col = col0 + wb * (BB / 2) + j * 16 + (lane & 15);
                    const int rbase = row0 + BA + wa * (BA2 / 2) + i * 16 + (lane >> 4) * 4;
                    if (col < a.Ccols) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (rbase + r < a.Crows) wsl[(size_t)(rbase + r) * wstride + col] = acc2[i][j][r];
                    }
                }
        }
        return;
    }
    const int otap = d.otap[tap];
    const size_t rstride = (size_t)d.otaps_total * a.Ccols;
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            const int col = col0 + wb * (BB / 2) + j * 16 + (lane & 15);
            const int rbase = row0 + wa * (BA / 2) + i * 16 + (lane >> 4) * 4;
            if (a.gsize) {
                // block-diagonal: dw is [rows][otaps_total][gsize]; keep the columns of the row's own group
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rbase + r;
                    const int gc = col - (row / a.gsize) * a.gsize;
                    if (row < a.Crows && gc >= 0 && gc < a.gsize)
                        mde_grad_add(a.dw + ((size_t)row * d.otaps_total + otap) * a.gsize + gc, acc[i][j][r], a.det);
                }
            } else if (col < a.Ccols) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rbase + r;
                    if (row < a.Crows)
                        mde_grad_add(a.dw + (size_t)row * rstride + (size_t)otap * a.Ccols + col, acc[i][j][r], a.det);
                }
            }
        }
    if constexpr (BA2 != 0) {
#pragma unroll
        for (int i = 0; i < FA2; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const int col = col0 + wb * (BB / 2) + j * 16 + (lane & 15);
                const int rbase = row0 + BA + wa * (BA2 / 2) + i * 16 + (lane >> 4) * 4;
                if (col < a.Ccols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rbase + r;
                        if (row < a.Crows)
                            mde_grad_add(a.dw + (size_t)row * rstride + (size_t)otap * a.Ccols + col, acc2[i][j][r], a.det);
                    }
                }
            }
    }
}

template <int BA, int BB, bool GA>
int launch(const KArgs& ka, int nblk, hipStream_t st) {
    constexpr size_t smem = 2 * (size_t)BKP * (BA + BB) * 2 + 2 * BKP * sizeof(uint32_t);
    constexpr size_t smem1 = (size_t)BKP * (BA + BB) * 2 + 2 * BKP * sizeof(uint32_t);
    // MDE_WGRAD_PATH=reg: register-staged main loop (diagnostics); MDE_WGRAD_NBUF=1|2 forces the single-buffer / ring form.
    // Default: the ring, except for grids of at least four workgroups per CU (the large-channel 9- and 25-tap layers whose
    // fitted split-K is small): there the single-buffer form's doubled occupancy wins (in-network: 512x512 channels, 25 taps,
    // 38 400 pixels 693 -> 544 us; 1024x1024, 25 taps, 9 600 pixels 745 -> 598 us), everywhere else it loses 10-20 %.
    static int reg = -1, nbuf = 0, cus = 256;
    if (reg < 0) {
        const char* e = getenv("MDE_WGRAD_PATH");
        reg = e && !strcmp(e, "reg");
        const char* nb = getenv("MDE_WGRAD_NBUF");
        nbuf = !nb ? 0 : !strcmp(nb, "1") ? 1 : 2;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    }
    const bool single = nbuf == 1 || (nbuf == 0 && nblk >= 4 * cus && BA * BB >= 128 * 128);
    if (reg) conv_wgrad_tn<BA, BB, GA, false><<<dim3(nblk), dim3(NT), smem, st>>>(ka);
    else if (single) conv_wgrad_tn<BA, BB, GA, true, 1><<<dim3(nblk), dim3(NT), smem1, st>>>(ka);
    else conv_wgrad_tn<BA, BB, GA, true><<<dim3(nblk), dim3(NT), smem, st>>>(ka);
    MDE_LAUNCH_CHECK("conv_wgrad_tn");
    return MDE_OK;
}

inline bool reg_path() {
    const char* e = getenv("MDE_WGRAD_PATH");
    return e && !strcmp(e, "reg");
}

template <bool GA>
int dispatch(const KArgs& ka, int ba, int bb, int nblk, hipStream_t st) {
    if (ba == 32) return bb == 128 ? launch<32, 128, GA>(ka, nblk, st) : launch<32, 64, GA>(ka, nblk, st);
    if (ba == 128 && bb == 128) return launch<128, 128, GA>(ka, nblk, st);
    if (ba == 128 && bb == 64) return launch<128, 64, GA>(ka, nblk, st);
    if (ba == 64 && bb == 128) return launch<64, 128, GA>(ka, nblk, st);
    return launch<64, 64, GA>(ka, nblk, st);
}

// the 128 + 32 / 64-row tiles (conv_wgrad_tn<128, BB, GA, true, 2, 32 | 64>): one row tile, the ring form.  The 64-row extension
// runs with 64-column tiles only (128 + 64 rows against 128 columns would be 80 KB of staging: one workgroup per CU)
template <bool GA, int BB, int BA2>
int launch_ext1(const KArgs& ka, int nblk, hipStream_t st) {
    constexpr size_t smem = 2 * (size_t)BKP * (128 + BA2 + BB) * 2 + 2 * BKP * sizeof(uint32_t);
    static bool done = false;
    if (!done && smem > 64 * 1024) {
        int rc = mde_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tn<128, BB, GA, true, 2, BA2>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), "hipFuncSetAttribute(conv_wgrad_tn 128 + extra rows)");
        if (rc) return rc;
        done = true;
    }
    conv_wgrad_tn<128, BB, GA, true, 2, BA2><<<dim3(nblk), dim3(NT), smem, st>>>(ka);
    MDE_LAUNCH_CHECK("conv_wgrad_tn (128 + extra rows)");
    return MDE_OK;
}
template <bool GA>
int launch_ext(const KArgs& ka, int bb, int extra, int nblk, hipStream_t st) {
    if (extra == 64) return launch_ext1<GA, 64, 64>(ka, nblk, st);
    return bb == 128 ? launch_ext1<GA, 128, 32>(ka, nblk, st) : launch_ext1<GA, 64, 32>(ka, nblk, st);
}

inline uint32_t inv32(uint32_t dv) { return dv <= 1 ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / dv); }

// Second stage of the two-stage reduction: dw[row][otap[t]][col] += sum_k ws[k][row][t][col].  One thread per four columns;
// the k loop keeps four slices in flight.  Plain read-modify-write of dw: the caller's launches into one dw are ordered
// on one stream (mde_conv_wgrad_ws).
struct RedArgs {
    const float* ws;
    float* dw;
    int32_t ks, Crows, ntaps, Ccols, otaps_total;
    int16_t otap[MDE_MAX_TAPS];
};
// KG: slices are dealt to KG thread groups of a workgroup (32 x KG float4 columns x groups per 256 threads) and combined through
// LDS -- a small dw split many ways (64 x 256 weights, 384 slices) would otherwise be a few thousand threads walking hundreds of
// slices one after the other.
template <int KG>
__global__ __launch_bounds__(256) void wgrad_reduce_k(const RedArgs a) {
    constexpr int EPB = 256 / KG;                                  // float4 elements per workgroup pass
    __shared__ f32x4_t sh[KG > 1 ? 256 : 1];
    const int c4 = a.Ccols >> 2;
    const int64_t per = (int64_t)a.Crows * a.ntaps * c4;           // float4 elements per slice
    const int64_t slice = per * 4;
    const int el = threadIdx.x % EPB, kg = threadIdx.x / EPB;
    for (int64_t i0 = (int64_t)blockIdx.x * EPB; i0 < per; i0 += (int64_t)gridDim.x * EPB) {
        const int64_t i = i0 + el;
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        if (i < per) {
            const float* src = a.ws + i * 4;
            int k = kg;
            for (; k + 3 * KG < a.ks; k += 4 * KG) {
                s0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(k + 0 * KG) * slice);
                s1 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(k + 1 * KG) * slice);
                s2 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(k + 2 * KG) * slice);
                s3 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(k + 3 * KG) * slice);
            }
            for (; k < a.ks; k += KG) s0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)k * slice);
        }
        f32x4_t sum = (s0 + s1) + (s2 + s3);
        if constexpr (KG > 1) {
            __syncthreads();                                       // (the previous pass has been read out)
            sh[threadIdx.x] = sum;
            __syncthreads();
            if (kg == 0) {
#pragma unroll
                for (int g = 1; g < KG; ++g) sum += sh[g * EPB + el];
            }
        }
        if (kg == 0 && i < per) {
            const int c = (int)(i % c4);
            const int64_t rt = i / c4;
            const int t = (int)(rt % a.ntaps);
            const int64_t row = rt / a.ntaps;
            f32x4_t* dst = reinterpret_cast<f32x4_t*>(a.dw + ((size_t)row * a.otaps_total + a.otap[t]) * a.Ccols + c * 4);
            *dst = *dst + sum;
        }
    }
}

}  // namespace

extern "C" int mde_conv_wgrad(const mde_wgrad_desc* d, const void* direct, const void* gathered,
                              float* dw, void* stream) {
    return mde_conv_wgrad_ws(d, direct, gathered, dw, nullptr, 0, stream);
}

// The two-stage split-K reduction of a launch: bytes of partial tiles it needs, and whether the launch would take that form
// given a workspace (MDE_WGRAD_TWOSTAGE: 0 never, 2 wherever possible, otherwise the measured rule).  ONE place for
// mde_conv_wgrad_ws (which decides) and mde_conv_wgrad_ws_bytes (which tells the caller what to allocate): a caller that sizes
// its workspace from the latter no longer reserves memory for launches that keep the atomic path.
static bool ws_choice(const mde_wgrad_desc* d, int64_t* need_out) {
    const int64_t M = (int64_t)d->N * d->GH * d->GW;
    const bool ga = d->rows_from_gathered != 0;
    const int crows = ga ? d->Cg : d->Cd, ccols = ga ? d->Cd : d->Cg;
    int64_t chunk = (M + d->ksplit - 1) / d->ksplit;
    chunk = (chunk + BKP - 1) / BKP * BKP;
    const int64_t kslices = (M + chunk - 1) / chunk;
    const int64_t need = kslices * crows * ccols * d->ntaps * 4;
    *need_out = need;
    if (d->group_size) return false;
    const char* e = getenv("MDE_WGRAD_TWOSTAGE");           // (read per call: the tests switch it between launches)
    const int two = !e ? 1 : atoi(e);
    if (!two) return false;
    if (two == 2) return true;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    static int rows32 = -1;
    if (rows32 < 0) {
        const char* r = getenv("MDE_WGRAD_ROWS32");
        rows32 = !(r && !strcmp(r, "0"));
    }
    const int ba = crows % 128 == 0 ? 128 : (rows32 && crows <= 32) ? 32 : 64;
    const int bb = ccols % 128 == 0 ? 128 : 64;
    // Where it pays (in-network, tools/per_shape_diff.py over `bench.py --per-shape` with MDE_WGRAD_TWOSTAGE=0 / 2): what the
    // atomics cost is the burst at the END of a launch whose workgroups all finish together -- one round of short workgroups
    // (512 x 128 weights split 128 ways: 92 -> 67 us; 1024 x 256 split 32: 63 -> 50; 256 x 256 x 9 split 14: 87 -> 77).  A grid
    // of several rounds, or workgroups of hundreds of K-steps, hides its atomics under the other workgroups' MFMAs and
    // only pays for the extra pass (1024 x 1024 x 25 split 2: +3 %; 128 x 128 x 9 over 614 400 pixels: +5 %); below ~20 MB
    // of partial tiles the second launch costs more than the burst (64 x 64 x 9 split 113: +6 %).
    const int64_t blocks = (int64_t)d->ntaps * mde_cdiv(crows, ba) * mde_cdiv(ccols, bb) * kslices;
    return need >= (20ll << 20) && blocks <= 5 * (int64_t)ncu && chunk / BKP <= 96;
}

extern "C" int64_t mde_conv_wgrad_ws_bytes(const mde_wgrad_desc* d) {
    if (!d || d->group_size || d->Cd <= 0 || d->Cg <= 0 || d->ntaps < 1 || d->ksplit < 1) return 0;
    int64_t need = 0;
    return ws_choice(d, &need) ? need : 0;
}

extern "C" int mde_conv_wgrad_ws(const mde_wgrad_desc* d, const void* direct, const void* gathered,
                                 float* dw, float* ws, int64_t ws_bytes, void* stream) {
    MDE_REQUIRE(d && direct && gathered && dw, "mde_conv_wgrad: null argument");
    MDE_REQUIRE(d->Cd > 0 && d->Cg > 0 && d->Cd % 8 == 0 && d->Cg % 8 == 0,
                "mde_conv_wgrad: channel counts (%d, %d) must be positive multiples of 8", d->Cd, d->Cg);
    MDE_REQUIRE(d->group_size == 0 || (d->Cd == d->Cg && d->Cd % 64 == 0 && 64 % d->group_size == 0),
                "mde_conv_wgrad: grouped needs equal channel counts, a multiple of 64, and a group size dividing 64 (%d, %d, %d)",
                d->Cd, d->Cg, d->group_size);
    MDE_REQUIRE(d->ntaps >= 1 && d->ntaps <= MDE_MAX_TAPS, "mde_conv_wgrad: ntaps=%d out of range", d->ntaps);
    MDE_REQUIRE(d->N > 0 && d->GH > 0 && d->GW > 0 && d->H > 0 && d->W > 0, "mde_conv_wgrad: non-positive dimension");
    MDE_REQUIRE(d->ld_d % 8 == 0 && d->ld_g % 8 == 0 && ((uintptr_t)direct % 16) == 0 && ((uintptr_t)gathered % 16) == 0,
                "mde_conv_wgrad: operands must be 16-byte aligned with ld %% 8 == 0");
    MDE_REQUIRE(d->d_bytes > 0 && d->d_bytes < MDE_OOB_OFFSET && d->g_bytes > 0 && d->g_bytes < MDE_OOB_OFFSET,
                "mde_conv_wgrad: operand sizes must be < 2 GiB");
    MDE_REQUIRE(d->ksplit >= 1, "mde_conv_wgrad: ksplit must be >= 1");
    for (int t = 0; t < d->ntaps; ++t)
        MDE_REQUIRE(d->otap[t] >= 0 && d->otap[t] < d->otaps_total, "mde_conv_wgrad: otap[%d] out of range", t);
    const int64_t M = (int64_t)d->N * d->GH * d->GW;
    MDE_REQUIRE(M < (1ll << 31) && M * d->ld_d < (1ll << 30) && (int64_t)d->N * d->H * d->W * d->ld_g < (1ll << 30),
                "mde_conv_wgrad: tensor too large for 32-bit indexing");

    {
        // the windowed form (conv_wgrad_win.hip): one workgroup per kernel ROW of taps.  It measured slower than the per-tap
        // kernel below on every shape (that file's header has the numbers), so it is never chosen: MDE_WGRAD_WIN=1 runs it
        // wherever the geometry is eligible (diagnostics, tests); unset or 0 = never.  Read per call.
        const char* we = getenv("MDE_WGRAD_WIN");
        const int win = !we ? 0 : atoi(we);
        if (win) {
            const int rc = mde_conv_wgrad_windowed(d, direct, gathered, dw, reinterpret_cast<hipStream_t>(stream), win == 1);
            if (rc != 0) return rc < 0 ? rc : MDE_OK;
        }
    }
    KArgs ka;
    ka.d = *d;
    ka.direct = direct;
    ka.gathered = gathered;
    ka.dw = dw;
    ka.M = (int32_t)M;
    const bool ga = d->rows_from_gathered != 0;
    ka.Crows = ga ? d->Cg : d->Cd;
    ka.Ccols = ga ? d->Cd : d->Cg;
    ka.gsize = d->group_size;
    ka.det = mde_det_dev();
    {
        static int ns = -1;
        if (ns < 0) {
            const char* e = getenv("MDE_WGRAD_NOSTORE");
            ns = e && !strcmp(e, "1");
        }
        ka.skip_store = ns;
    }
    if (ka.det.scratch) {
        const int64_t nw = (int64_t)ka.Crows * d->otaps_total * (d->group_size ? d->group_size : ka.Ccols);
        MDE_REQUIRE(dw >= g_mde_det.gbase && dw + nw <= g_mde_det.gbase + g_mde_det.n,
                    "mde_conv_wgrad: deterministic mode is on and dw lies outside the registered gradient buffer");
    }
    // (<= 32 rows -- the full-resolution decoder layers of BTS and MiDaS, 32 output channels over millions of pixels -- take the
    //  32-row tile: half the padding of the 64-row one.  MDE_WGRAD_ROWS32=0: off)
    static int rows32 = -1;
    if (rows32 < 0) {
        const char* e = getenv("MDE_WGRAD_ROWS32");
        rows32 = !(e && !strcmp(e, "0"));
    }
    const int ba = (ka.Crows % 128 == 0 && !ka.gsize) ? 128 : (rows32 && !ka.gsize && ka.Crows <= 32) ? 32 : 64;
    const int bb = (ka.Ccols % 128 == 0 && !ka.gsize) ? 128 : 64;
    ka.nB = ka.gsize ? 1 : mde_cdiv(ka.Ccols, bb);
    int64_t chunk = (M + d->ksplit - 1) / d->ksplit;
    chunk = (chunk + BKP - 1) / BKP * BKP;
    ka.kchunk = (int32_t)chunk;
    const int kslices = mde_cdiv(M, chunk);
    ka.inv_gw = inv32((uint32_t)d->GW);
    ka.inv_ghw = inv32((uint32_t)(d->GH * d->GW));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // Two-stage reduction where a workspace is given and holds every partial tile (MDE_WGRAD_TWOSTAGE=0: never).  Not in
    // deterministic mode (its integer atomics are order-free already), not for grouped weights (block-diagonal dw).
    ka.ws = nullptr;
    {
        int64_t need = 0;
        const bool pays = ws_choice(d, &need);          // (MDE_WGRAD_TWOSTAGE and the measured rule: above)
        if (pays && ws && !ka.gsize && !ka.det.scratch && !ka.skip_store && need <= ws_bytes && ka.Ccols % 4 == 0 &&
            ((uintptr_t)ws % 16) == 0 && ((uintptr_t)dw % 16) == 0)
            ka.ws = ws;
    }
    auto reduce = [&]() -> int {
        if (!ka.ws) return MDE_OK;
        RedArgs ra;
        ra.ws = ka.ws;
        ra.dw = dw;
        ra.ks = kslices;
        ra.Crows = ka.Crows;
        ra.ntaps = d->ntaps;
        ra.Ccols = ka.Ccols;
        ra.otaps_total = d->otaps_total;
        for (int t = 0; t < MDE_MAX_TAPS; ++t) ra.otap[t] = t < d->ntaps ? d->otap[t] : 0;
        const int64_t per = (int64_t)ka.Crows * d->ntaps * (ka.Ccols / 4);
        if (kslices >= 32 && per < (1 << 20)) {
            const int64_t nb = (per + 31) / 32;
            wgrad_reduce_k<8><<<dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, st>>>(ra);
        } else {
            const int64_t nb = (per + 255) / 256;
            wgrad_reduce_k<1><<<dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, st>>>(ra);
        }
        MDE_LAUNCH_CHECK("wgrad_reduce_k");
        return MDE_OK;
    };
    auto go = [&](int tile_a, int row_off, int rows) -> int {
        ka.row_off = row_off;
        ka.nA = mde_cdiv(rows, tile_a);
        const int64_t nblk = (int64_t)d->ntaps * ka.nA * ka.nB * kslices;
        MDE_REQUIRE(nblk < (1ll << 31), "mde_conv_wgrad: grid too large");
        return ga ? dispatch<true>(ka, tile_a, bb, (int)nblk, st) : dispatch<false>(ka, tile_a, bb, (int)nblk, st);
    };
    // A row count that is not a multiple of 128 (DenseNet's 192-channel bottlenecks): the 128-row tile, whose operand re-use is
    // twice the 64-row tile's, takes the multiple of 128 in front and the 64-row tile only the remainder — if the remainder
    // fills more than half of that tile (VNL's 152 rows = 128 + 24 measured 1 % slower split than as three 64-row tiles).
    // BTS 56.0 -> 55.3 ms per step, MyNet 37.3 -> 36.8.  (MDE_WGRAD_MIXED=0: 64-row tiles throughout, as before.)
    static int mixed = -1;
    if (mixed < 0) {
        const char* e = getenv("MDE_WGRAD_MIXED");
        mixed = !(e && !strcmp(e, "0"));
    }
    static int mixed_min = -1;
    if (mixed_min < 0) {
        const char* e = getenv("MDE_WGRAD_MIXED_MIN");          // (diagnostics: the smallest remainder that still takes a 64-row tile of its own)
        mixed_min = e ? atoi(e) : 16;
    }
    if (mixed && ba == 64 && !ka.gsize && ka.Crows > 128 && ka.Crows % 128 > mixed_min) {
        const int head = ka.Crows / 128 * 128;
        // The split factor was chosen for a grid of 64-row tiles; each of the two launches here has a fraction of those
        // workgroups (152 rows: 1 of 3 row tiles each) and would leave CUs empty: with atomics (no shared workspace layout)
        // both take twice the slices.  VNL's 256 -> 152 prediction conv, 2 x 2 457 600 pixels: step 68.9 -> 67.4 ms.
        static int mks = -1;
        if (mks < 0) {
            const char* e = getenv("MDE_WGRAD_MIXED_KS");
            mks = e ? atoi(e) : 2;
        }
        int ks2 = kslices;
        if (!ka.ws && mks > 1 && head == 128) {
            int64_t c2 = (M + (int64_t)d->ksplit * mks - 1) / ((int64_t)d->ksplit * mks);
            c2 = (c2 + BKP - 1) / BKP * BKP;
            ka.kchunk = (int32_t)c2;
            ks2 = mde_cdiv(M, c2);
        }
        auto go2 = [&](int tile_a, int row_off, int rows) -> int {
            ka.row_off = row_off;
            ka.nA = mde_cdiv(rows, tile_a);
            const int64_t nblk = (int64_t)d->ntaps * ka.nA * ka.nB * ks2;
            MDE_REQUIRE(nblk < (1ll << 31), "mde_conv_wgrad: grid too large");
            return ga ? dispatch<true>(ka, tile_a, bb, (int)nblk, st) : dispatch<false>(ka, tile_a, bb, (int)nblk, st);
        };
        // 128 + a remainder <= 64: ONE launch of the 128 + 32 / 64-row tile (MDE_WGRAD_ROWS160=0: two launches, as before; =32: only
        // the 32-row extension)
        static int rows160 = -1;
        if (rows160 < 0) {
            const char* e = getenv("MDE_WGRAD_ROWS160");
            rows160 = !e ? 64 : atoi(e);
        }
        const int rem = ka.Crows - head;
        if (rows160 && rows32 && head == 128 && rem <= rows160 && !reg_path()) {
            const int extra = rem <= 32 ? 32 : 64;
            const int bbx = extra == 64 ? 64 : bb;                 // (see launch_ext1)
            ka.row_off = 0;
            ka.nA = 1;
            ka.nB = mde_cdiv(ka.Ccols, bbx);
            const int64_t nblk = (int64_t)d->ntaps * ka.nB * ks2;
            MDE_REQUIRE(nblk < (1ll << 31), "mde_conv_wgrad: grid too large");
            if (int rc = ga ? launch_ext<true>(ka, bbx, extra, (int)nblk, st) : launch_ext<false>(ka, bbx, extra, (int)nblk, st)) return rc;
            return reduce();
        }
        if (int rc = go2(128, 0, head)) return rc;
        if (int rc = go2((rows32 && ka.Crows - head <= 32) ? 32 : 64, head, ka.Crows - head)) return rc;
        return reduce();
    }
    if (int rc = go(ba, 0, ka.Crows)) return rc;
    return reduce();
}
