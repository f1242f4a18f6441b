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
// Two LDS buffers and two register staging sets, one barrier per K-step: every global load is
// issued two K-steps before it is needed (the single-stage version was latency-bound: 41 % of
// wave cycles parked in s_waitcnt, MFMA busy 28 %; profiles/r01_b_pmc_conv.txt).
//
// Epilogue: fp32 accumulators -> bf16 -> LDS [pixel][channel] -> full-line 16-byte stores
// (optionally read-modify-write for dgrad accumulation), plus optional BatchNorm partial
// sums (sum, sum of squares per channel, reduced per workgroup then added with fp32 atomics
// to one of MDE_STAT_SLOTS rows) so the BN statistics pass over the conv output is not needed.
#include <stdlib.h>

#include <stdio.h>

#include "mde_common.h"
#include <type_traits>

// MDE_ABLATE (timing-only diagnostic builds, results are wrong): 1 = no global loads in the
// K-loop, 2 = also no LDS staging writes, 3 = no MFMAs, 4 = no barriers in the K-loop (register
// path), 5 = no LDS-DMA in the K-loop (DMA path: LDS reads + MFMA + barrier only).
#ifndef MDE_ABLATE
#define MDE_ABLATE 0
#endif
// Schedule knobs of the DMA loop, A/B-tested with tools/conv_microbench.py (M=153600 N=256 K=2304) and bench.py:
//   8-wave tiles: s_setprio(1) around each MFMA cluster (+4 %).  Issuing the next tile's DMAs BETWEEN the two MFMA
//   halves of a K-step is +7 % on the microbenchmark (operands warm in L2/MALL: 941 vs 878 TFLOP/s) but LOSES in the
//   network, where operands come from HBM: issuing them at the top of the step, a full step ahead, is 0.8 ms/step
//   faster (991 -> 1015 images/s).  Latency cover beats issue placement once the data is cold -> default off.
//   4-wave 128x128 (two workgroups per CU): mid-step issue -6 %, setprio neutral -> both off.
//   A four-stage ring of 32-channel stages (same LDS, DMA 1.5 steps ahead) was built and measured: correct, but
//   twice the barriers and LDS-latency exposures per MFMA: -15 % microbench, -3 % in-network (32.75 vs 31.9 ms/step).
#ifndef MDE_DMA_MID
#define MDE_DMA_MID 0
#endif
#ifndef MDE_RATE_256
#define MDE_RATE_256 1.15   // fitted per-flop rates of the 8-wave tiles relative to 128x128 (pick_and_launch)
#define MDE_RATE_192 1.10
#endif
#ifndef MDE_SETPRIO
#define MDE_SETPRIO (NT == 512)
#endif

namespace {

constexpr int BK = 64;
struct KArgs {
    mde_conv_desc d;
    const void* in;
    const void* w;
    void* out;
    float* stats;
    uint32_t w_bytes;
    uint32_t out_bytes;
    int32_t M;        // N*GH*GW
    int32_t m_begin;  // first pixel row of this launch (a layer may be split into two launches by pixel range)
    int32_t m_end;    // one past the last pixel row of this launch
    int32_t nP, nC;   // tiles along pixels / columns
    int32_t vec_ok;   // 16-byte stores allowed
    int32_t col_fastest;  // tile order: 1 = all column tiles of a pixel tile are neighbours (activation tile reused from L2)
    int32_t det;          // deterministic mode: BatchNorm partial sums leave the workgroup as integer atomics (mde_common.h)
    // fused epilogue (mde_conv_gemm_act): the pass a biased / activated conv would otherwise make over its own output
    // (pointwise.hip pw_fwd_k).  The bias joins the fp32 accumulator BEFORE the one rounding to the storage type; without a
    // residual the activation does too: out = bf16(act(result + bias)).  With one: out = bf16(act(bf16(result + bias) + residual)).
    // (Round 3 rounded the bare result first, as the separate pass sees it.  bf16(result) + bias, rounded again, is a
    // DOUBLE rounding across a per-channel CONSTANT: where |result + bias| is a binade above |result| the second grid is
    // coarser and the error of the pair depends on (bias mod ulp) only -- the same for every pixel of the channel, up to half
    // an ulp, and it does not average out of a mean over pixels: MiDaS' biased decoder convs shifted the eval output's mean by
    // 8e-5 relative on the CPU oracle's emulation of that order.)
    const float* bias;    // [ncols] or nullptr
    const void* resid;    // bf16, addressed exactly like `out` (same offsets), or nullptr
    int32_t act;          // 0 none, 1 ReLU, 2 ELU, 3 sigmoid
    // fused BatchNorm-backward sums (mde_conv_gemm_bnred): this launch writes the gradient g of a BatchNorm (+ ReLU) site's
    // OUTPUT; its epilogue also adds that site's sum(g') and sum(g' * xhat) (g' = g under the ReLU mask, xhat the site's
    // normalised input) into the site's partial-sum buffer -- the pass bn.hip's bn_bwd_reduce_k would make over g and x
    const void* red_x;        // the site's input (bf16, addressed exactly like `out`), nullptr = off
    const float* red_mu;      // saved batch mean / 1 / std of the site [ncols]
    const float* red_rs;
    const float* red_msc;     // ReLU mask recomputed as x * msc + msh > 0 (the site's scale / shift), or
    const float* red_msh;
    const uint8_t* red_bits;  // packed mask bits, one byte per 8 channels of a dense row (residual joins), or neither: no ReLU
    float* red_part;          // [MDE_STAT_SLOTS][2][ncols]
    int32_t red_ldmul;        // the site's input has row stride red_ldmul * ld_out, or (0) any stride red_ldx: the row index is
    int32_t red_ldx;          // then recovered from the output offset by a division (DenseNet: a prefix of a wider concatenation)
    // a residual join out = relu(bn_a(x) + bn_b(x2)): one masked gradient (bits), the sums of both sites (bn.hip bn_bwd_reduce2_k)
    const void* red_x2;       // nullptr = one site
    const float* red_mu2;
    const float* red_rs2;
    float* red_part2;
    int32_t red_ldmul2, red_ldx2;
    // with the sums (RED instances only): out = result + (add under its own mask bits) -- an identity shortcut's gradient taken
    // straight from the block OUTPUT's gradient, which the BatchNorm-backward pass then need not copy into `out` first
    const void* add_src;      // bf16, addressed exactly like `out`; nullptr = off
    const uint8_t* add_bits;  // one byte per 8 channels of a dense row, or nullptr: unmasked
    // halo-tiled form (HALO kernels): the workgroup's 128 pixels are a th x tw block of ONE image (tw = 1 << h_tws), whose
    // input window (th + dy span) x (tw + dx span) is staged ONCE per 64-channel chunk and read by every tap
    int32_t h_tws, h_th;          // log2(tile width), tile height
    int32_t h_nty, h_ntx;         // tiles per image
    int32_t h_hw, h_rows;         // halo width (pixels), halo rows (pixels) in all
    int32_t h_dy0, h_dx0;         // smallest tap offsets: halo pixel (0, 0) is input (gy0 + dy0, gx0 + dx0)
    int32_t h_dil;                // dilation D: the tile is a block of ONE parity class (rows % D, columns % D) of the output grid,
                                  // in whose own coordinates the taps are D times closer (a dilated 3x3 has a (th + 2) x (tw + 2) window)
    int32_t h_step_y, h_step_x;   // a wave's next window instruction is (waves x 8) rows on: that many / hw lines down, % hw pixels right (+ one carry)
};

__device__ __forceinline__ float epi_act(float v, int act) {      // == pointwise.hip act_fwd
    switch (act) {
        case 1: return fmaxf(v, 0.f);
        case 2: return v > 0.f ? v : expm1f(v);
        case 3: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// byte offset of the 16-byte chunk (row, kslot8) inside a swizzled tile
__device__ __forceinline__ int chunk_off(int row, int kslot8) {
    return ((((row >> 4) * 2 + (kslot8 >> 2)) * 64) + (kslot8 & 3) * 16 + ((row & 15) ^ kslot8)) * 16;
}

// NT threads (4 or 8 waves).  Waves are laid out WAVES_P (pixels) x WAVES_C (columns); a wave
// owns PF x CF fragments of 16 pixels x 16 columns.
//
// DMA = true: the tiles are filled by LDS-DMA (buffer_load ... lds, 1 KiB = 8 rows x 128 B per
// wave-instruction, no VGPR staging, no ds_write) into a ring of NBUF LDS buffers, issued
// NBUF-1 K-steps ahead; ordering is a counted s_waitcnt vmcnt(N) + one raw s_barrier per step.
// The LDS destination of a DMA is lane-linear, so the bank swizzle lives in the SOURCE k-slot
// each lane fetches and in the fragment read address (guide rule 21).  Out-of-image taps use
// an out-of-range buffer offset: the DMA then writes zeros (tools/probes/lds_dma_probe.hip).
// PP = true ("ping-pong", 8 waves, 2-buffer DMA ring): the two waves of every SIMD run the SAME K-loop one barrier apart,
// so one of them is always in its MFMA cluster while the other issues its LDS reads and DMA pieces (guide: the 8-phase
// template's stagger; MI355X_MICROARCH "Two waves per SIMD" item 9).  In the lockstep loop both waves read together (LDS
// saturated, matrix pipe idle) and then compute together (pipe shared): MFMA busy measured 41 %.
//
// HALO = true (single staging buffer, stride-1 gathers, >= 2 taps): the pixel operand of a K-step is not gathered per tap.
// The 128 pixels of a tile form a th x tw block of one image; the block's input window is staged once per 64-channel chunk
// ("LDS-staged im2col": (th + 2) x (tw + 2) rows of 128 B for a 3x3) and the taps of that chunk read it at shifted
// rows.  Why: the 128x128 tile moves 32 KB through the L2 -> LDS DMA path per K-step for 2 MFLOP (64 FLOP/B); that path
// delivers ~70 GB/s per CU (MI355X_MICROARCH.md, "Indexed rows: gather into LDS"), i.e. it caps the tile near 1.1 PFLOP/s
// -- where the best 9- and 25-tap layers sit.  With the window shared by the taps a 3x3 K-step moves 18.6 KB instead.
#ifndef MDE_RING3_MAX_TILES_PER_CU
#define MDE_RING3_MAX_TILES_PER_CU 1
#endif
#ifndef MDE_DEEP_RING
#define MDE_DEEP_RING 4
#endif
#ifndef MDE_PAD_LDS
#define MDE_PAD_LDS 0
#endif
// waves along the columns of a tile: two for 128+ columns, and for the 8-wave form of the 128 x 64 tile (each wave needs two
// 16-pixel fragments at least)
template <int BP, int BC, int NT>
constexpr int waves_c() { return (BC >= 128 || NT / 64 > BP / 32) ? 2 : 1; }
constexpr int HALO_MAX_Q = 8;    // halo DMA instructions per wave and chunk (4 waves x 8 x 8 rows = 256 halo rows)
// RED = 1 / 2: the instances behind mde_conv_gemm_bnred (the epilogue also reduces the backward sums of a BatchNorm site / of the
// two sites of a residual join); kept apart so that the registers that epilogue needs do not cost the other launches their occupancy
template <int BP, int BC, int NT, bool DMA, int NBUF, bool PP = false, bool HALO = false, int RED = 0>
__global__ __launch_bounds__(NT, HALO ? (NT == 512 ? 4 : BC <= 64 ? 5 : 4)
                                     : (NT == 256 && BP * BC >= 256 * 256) ? 1 : (NBUF == 1 ? 4 : 2)) void conv_gemm_nt(const KArgs a) {
    constexpr int NW = NT / 64;
    constexpr int RPL = NT / 8;          // tile rows covered by one load pass (8 chunks per row)
    constexpr int XP = BP / RPL;         // X chunks per thread per K-step
    constexpr int WP = BC / RPL;         // W chunks per thread per K-step
    constexpr int WAVES_C = waves_c<BP, BC, NT>();   // waves along columns
    constexpr int WAVES_P = NW / WAVES_C;        // waves along pixels
    constexpr int PF = BP / WAVES_P / 16;        // 16-pixel fragments per wave
    constexpr int CF = BC / WAVES_C / 16;        // 16-column fragments per wave
    constexpr int XT_BYTES = BP * BK * 2;
    constexpr int WT_BYTES = BC * BK * 2;
    constexpr int BUF_BYTES = XT_BYTES + WT_BYTES;
    constexpr int ROWB = BC * 2 + 16;    // epilogue tile row pitch (bytes)
    // the epilogue re-tiles through LDS; when the whole tile does not fit, in EPASS pixel slabs
    // LDS the epilogue may stage through (HALO: the smallest window, one row per tile pixel, + the weight buffers)
    constexpr int STG_BYTES = HALO ? XT_BYTES + NBUF * WT_BYTES : (NBUF < 2 ? NBUF : 2) * BUF_BYTES;
    constexpr int EPASS = (BP * ROWB <= STG_BYTES) ? 1 : (BP * ROWB <= 2 * STG_BYTES) ? 2 : 4;
    constexpr int EROWS = BP / EPASS;    // pixel rows per epilogue pass
    static_assert(WAVES_P * WAVES_C == NW && PF * 16 * WAVES_P == BP && CF * 16 * WAVES_C == BC && PF >= 2 && PF <= 8, "wave tiling");
    static_assert(XP >= 1 && WP >= 1 && XP * RPL == BP && WP * RPL == BC, "load tiling");
    static_assert(EROWS * ROWB <= STG_BYTES && (WAVES_P % EPASS) == 0, "epilogue slab fits in the staging buffers");
    static_assert(NBUF == 2 || (DMA && (NBUF == 1 || (NBUF >= 3 && NBUF <= 6 && !PP && !HALO))), "register staging uses two LDS buffers");
    static_assert(NBUF != 1 || !PP, "the single-buffer loop has no ping-pong form");
    static_assert(!HALO || (DMA && !PP && ((NBUF == 1 && BP == 128 && NT == 256) || (NBUF == 2 && BP == 256 && NT == 512))),
                  "the halo form: 128-pixel single-buffer tiles (4 waves) or 256-pixel tiles with a two-deep weight ring (8 waves)");
    constexpr int XR = BP / 8 / NW, WR = BC / 8 / NW;   // DMA regions (8 rows) per wave per K-step
    static_assert(!DMA || (XR >= 1 && WR >= 1 && XR * NW * 8 == BP && WR * NW * 8 == BC && NW % 2 == 0), "DMA tiling");

#ifdef MDE_SB_STAMP
    const uint64_t sb_k0 = __builtin_amdgcn_s_memtime(), sb_r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t sb_loop0 = 0, sb_loop1 = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // HALO: [halo window: ceil(rows / 8) KiB][weight tile][s_out][s_tap]; the BatchNorm partial sums (epilogue only) live
    // in the staging area behind the first epilogue slab, so that four (128 columns) / five (64) workgroups fit a CU
    const int halo_bytes = HALO ? ((a.h_rows + 7) >> 3) * 1024 : 0;
    static_assert(!HALO || EROWS * ROWB + WAVES_P * 2 * BC * 4 <= STG_BYTES, "statistics fit behind the epilogue slab");
    int* s_inbase = reinterpret_cast<int*>(smem + (HALO ? halo_bytes + NBUF * WT_BYTES : NBUF * BUF_BYTES));
    int* s_yx = HALO ? s_inbase : s_inbase + BP;                      // (HALO: no per-row gather tables)
    int* s_out = HALO ? s_inbase : s_yx + BP;
    float* s_stat = HALO ? reinterpret_cast<float*>(smem + EROWS * ROWB) : reinterpret_cast<float*>(s_out + BP);  // [WAVES_P][2][BC]
    int* s_tap = HALO ? s_out + BP : reinterpret_cast<int*>(s_stat + WAVES_P * 2 * BC);   // [3][MDE_MAX_TAPS]: x offset, w offset, dy|dx

    const mde_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave % WAVES_C;       // wave position along columns
    const int wp = wave / WAVES_C;       // wave position along pixels

    const uint32_t bid = mde_xcd_remap(blockIdx.x, (uint32_t)(a.nP * a.nC));
    const int pi = a.col_fastest ? bid / a.nC : bid % a.nP;
    const int ci = a.col_fastest ? bid % a.nC : bid / a.nP;
    const int m0 = a.m_begin + pi * BP, n0 = ci * BC;

    // HALO: tile pi = (image, tile row, tile column); pixel r of the tile = (r >> tws, r & (tw - 1)) of the block
    int h_n = 0, h_gy0 = 0, h_gx0 = 0, h_ry = 0, h_rx = 0;
    if constexpr (HALO) {
        // tile pi = (image, parity class, tile row, tile column); h_gy0 / h_gx0 are in the class's own coordinates: class pixel
        // (y, x) is grid pixel (y * D + ry, x * D + rx)
        const int tpc = a.h_nty * a.h_ntx, D = a.h_dil, tpi = tpc * D * D;
        h_n = pi / tpi;
        const int irem = pi - h_n * tpi, cls = irem / tpc, trem = irem - cls * tpc, ty = trem / a.h_ntx;
        h_ry = cls / D;
        h_rx = cls - h_ry * D;
        h_gy0 = ty * a.h_th;
        h_gx0 = (trem - ty * a.h_ntx) << a.h_tws;
        for (int r = tid; r < BP; r += NT) {
            const int gy = (h_gy0 + (r >> a.h_tws)) * D + h_ry, gx = (h_gx0 + (r & ((1 << a.h_tws) - 1))) * D + h_rx;
            s_out[r] = (gy < d.GH && gx < d.GW) ? ((h_n * d.OH + gy * d.osy + d.ooy) * d.OW + gx * d.osx + d.oox) * d.ld_out : -1;
        }
        if (tid < d.ntaps) {
            s_tap[tid] = (d.dy[tid] / D - a.h_dy0) * a.h_hw + (d.dx[tid] / D - a.h_dx0);     // halo-row offset of the tap (class coordinates)
            s_tap[MDE_MAX_TAPS + tid] = d.wtap[tid] * d.C;
        }
    } else {
    // ---- per-row decode, once per workgroup
    for (int r = tid; r < BP; r += NT) {
        const int m = m0 + r;
        int inbase = 0, yx = 0x7FFF7FFF, oo = -1;
        if (m < a.m_end) {
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
    // per-tap scalars into LDS: indexing the kernarg tables dynamically inside the K-loop makes
    // hipcc fetch them with VECTOR global loads and wait vmcnt(0) for them, which drains every
    // prefetched tile each step (seen in the .s; MFMA busy 28 % -> see profiles/).
    if (tid < d.ntaps) {
        const int tdy = d.dy[tid], tdx = d.dx[tid];
        s_tap[tid] = (tdy * d.W + tdx) * d.ld_in;
        s_tap[MDE_MAX_TAPS + tid] = d.wtap[tid] * d.C;
        s_tap[2 * MDE_MAX_TAPS + tid] = (tdy << 16) | (tdx & 0xFFFF);
    }
    }   // !HALO
    __syncthreads();

    const int kslot8 = tid & 7;
    const int lrow = tid >> 3;           // 0..RPL-1
    // grouped convolution (block-diagonal weights, 64-channel blocks): the column tile [n0, n0+64) contracts
    // with the input channels [n0, n0+64) only -- the gathered operand's channel window moves with the tile
    const int goff = d.grouped ? n0 : 0;
    int x_base[XP], x_yx[XP];
#pragma unroll
    for (int p = 0; p < XP; ++p) {
        x_base[p] = HALO ? 0 : s_inbase[p * RPL + lrow] + goff + kslot8 * 8;
        x_yx[p] = HALO ? 0 : s_yx[p * RPL + lrow];
    }
    const int wrow_len = d.wtaps_total * d.C;
    int w_base[WP];
#pragma unroll
    for (int p = 0; p < WP; ++p) w_base[p] = (n0 + p * RPL + lrow) * wrow_len + kslot8 * 8;

    const __amdgpu_buffer_rsrc_t rs_in = mde_rsrc(a.in, d.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = mde_rsrc(a.w, a.w_bytes);

    int st_off[XP > WP ? XP : WP];       // LDS chunk offsets of this thread's rows (same for X and W)
#pragma unroll
    for (int p = 0; p < (XP > WP ? XP : WP); ++p) st_off[p] = chunk_off(p * RPL + lrow, kslot8);

    // fragment read offsets: lane l reads row (l&15), k-slot (l>>4) of fragment (rb, kb)
    int rd_off[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int ks = lane >> 4, r = lane & 15;
        if constexpr (DMA)   // region (r>>3) of 8 rows x 8 k-slots, k-slot XORed with (row>>1)&7
            rd_off[kb] = (r >> 3) * 1024 + ((r & 7) * 8 + ((kb * 4 + ks) ^ ((r >> 1) & 7))) * 16;
        else
            rd_off[kb] = (kb * 64 + ks * 16 + (r ^ (kb * 4 + ks))) * 16;
    }

    f32x4_t acc[CF][PF];
#pragma unroll
    for (int i = 0; i < CF; ++i)
#pragma unroll
        for (int j = 0; j < PF; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // C need not be a multiple of BK: the last K-step of a tap is zero-filled past C (both operands) by sending
    // the 16-byte chunks beyond it to an out-of-range buffer offset (C % 8 == 0)
    // K-steps run chunk-major in EVERY form of the loop: (64-channel chunk, tap), taps fastest.  The halo-tiled form needs that
    // order (one window per chunk); the plain forms follow it so that a layer's result does not depend on which form its grid
    // size selects: same operands into the same sequence of MFMAs = bit-identical output (eval outputs stay batch-invariant).
    const int csteps = (d.C + BK - 1) / BK;
    const int nsteps = d.ntaps * csteps;
    // two register staging sets: loads are issued TWO K-steps ahead of their use, so each has
    // two full compute phases to land (the loop was global-latency-bound with one).
    constexpr bool DEEP = (CF * PF <= 16);   // second staging set only when the accumulators leave room
    i32x4_t xa[XP], wa[WP], xb[DEEP ? XP : 1], wb_[DEEP ? WP : 1];
    int ltap = 0, lcs = 0;                // cursor of the next K-step to load

    auto issue_loads = [&](i32x4_t (&xr)[XP], i32x4_t (&wr)[WP]) {
        const int c0 = lcs * BK;
        const int tapoff = s_tap[ltap] + c0;
        const int woff = s_tap[MDE_MAX_TAPS + ltap] + c0;
        const int dydx = s_tap[2 * MDE_MAX_TAPS + ltap];
        const int tdy = dydx >> 16, tdx = (int)(short)(dydx & 0xFFFF);
        const bool kin = c0 + kslot8 * 8 < d.C;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int iy = (int)((uint32_t)x_yx[p] >> 16) + tdy;
            const int ix = (x_yx[p] & 0xFFFF) + tdx;
            const bool ok = ((uint32_t)iy < (uint32_t)d.H) & ((uint32_t)ix < (uint32_t)d.W) & kin;
            const uint32_t off = ok ? (uint32_t)(x_base[p] + tapoff) * 2u : MDE_OOB_OFFSET;
            xr[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < WP; ++p)
            wr[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, kin ? (uint32_t)(w_base[p] + woff) * 2u : MDE_OOB_OFFSET, 0, 0);
        if (++ltap == d.ntaps) { ltap = 0; ++lcs; }
    };
    auto stage_write = [&](int buf, const i32x4_t (&xr)[XP], const i32x4_t (&wr)[WP]) {
        char* xbuf = smem + buf * BUF_BYTES;
        char* wbuf = xbuf + XT_BYTES;
#pragma unroll
        for (int p = 0; p < XP; ++p) *reinterpret_cast<i32x4_t*>(xbuf + st_off[p]) = xr[p];
#pragma unroll
        for (int p = 0; p < WP; ++p) *reinterpret_cast<i32x4_t*>(wbuf + st_off[p]) = wr[p];
    };
    auto compute_half = [&](int buf, int kb) {
        const char* xbuf = smem + buf * BUF_BYTES + wp * (PF * 2048);
        const char* wbuf = smem + buf * BUF_BYTES + XT_BYTES + wc * (CF * 2048);
        bf16x8_t fa[CF], fb[PF];
#pragma unroll
        for (int i = 0; i < CF; ++i)
            fa[i] = *reinterpret_cast<const bf16x8_t*>(wbuf + i * 2048 + rd_off[kb]);
#pragma unroll
        for (int j = 0; j < PF; ++j)
            fb[j] = *reinterpret_cast<const bf16x8_t*>(xbuf + j * 2048 + rd_off[kb]);
#if MDE_ABLATE == 3
#pragma unroll
        for (int i = 0; i < CF; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
        for (int j = 0; j < PF; ++j) asm volatile("" ::"v"(fb[j]));
#else
        if constexpr (MDE_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < CF; ++i)
#pragma unroll
            for (int j = 0; j < PF; ++j)
                acc[i][j] = MDE_MFMA_16x16x32(fa[i], fb[j], acc[i][j]);
        if constexpr (MDE_SETPRIO) __builtin_amdgcn_s_setprio(0);
#endif
    };
    auto compute = [&](int buf) {
        compute_half(buf, 0);
        compute_half(buf, 1);
    };
#if MDE_ABLATE == 1 || MDE_ABLATE == 2
#define LOOP_LOADS(...)
#else
#define LOOP_LOADS(...) issue_loads(__VA_ARGS__)
#endif
#if MDE_ABLATE == 2
#define LOOP_WRITE(...)
#else
#define LOOP_WRITE(...) stage_write(__VA_ARGS__)
#endif
#if MDE_ABLATE == 4
#define LOOP_SYNC()
#else
#define LOOP_SYNC() __syncthreads()
#endif
    if constexpr (DMA) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int lr = lane >> 3;
        const int k8 = (lane & 7) ^ ((((wv & 1) << 2) + (lr >> 1)) & 7);   // source k-slot of this lane
        // per row: byte offset of the tap-origin pixel and a bitmask of the taps that stay inside
        // the image, computed once per tile so the K-loop spends ~3 VALU per DMA, not ~9
        uint32_t dx_base[XR], dx_ok[XR], dw_base[WR];
        // HALO: byte offset of this lane's chunk of halo row h = (q * NW + wv) * 8 + lr (MDE_OOB_OFFSET: outside the image
        // or past the window -> the DMA writes zeros = the convolution's padding); the window's fragment-read bases and
        // the validity of this lane's pixels (a tile may hang over the image's edge)
        // (the offsets are re-derived for every chunk by walking the window 32 rows at a time: eight of them held in registers
        //  across the K-loop pushed the 128-column kernel over its 128-VGPR budget at four workgroups per CU)
        int hb[PF];
        uint32_t pvalid = 0;
        int h_iy0 = 0, h_ix0 = 0;                   // input pixel of this lane's first halo row (q = 0)
        if constexpr (HALO) {
            const int h = wv * 8 + lr;
            const int hy = h / a.h_hw;
            h_iy0 = h_gy0 + a.h_dy0 + hy;
            h_ix0 = h_gx0 + a.h_dx0 + (h - hy * a.h_hw);
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int p = wp * (PF * 16) + j * 16 + (lane & 15);
                const int py = p >> a.h_tws, px = p & ((1 << a.h_tws) - 1);
                hb[j] = py * a.h_hw + px;
                pvalid |= (uint32_t)(((h_gy0 + py) * a.h_dil + h_ry < d.GH) & ((h_gx0 + px) * a.h_dil + h_rx < d.GW)) << j;
            }
        }
#pragma unroll
        for (int q = 0; q < XR; ++q) {
            if constexpr (HALO) { dx_base[q] = 0; dx_ok[q] = 0; continue; }
            const int row = (wv + q * NW) * 8 + lr;
            dx_base[q] = (uint32_t)(s_inbase[row] + goff + k8 * 8) * 2u;
            const int yx = s_yx[row];
            const int iy0 = (int)((uint32_t)yx >> 16), ix0 = yx & 0xFFFF;
            uint32_t m = 0;
            for (int t = 0; t < d.ntaps; ++t) {
                const int dydx = s_tap[2 * MDE_MAX_TAPS + t];
                const int iy = iy0 + (dydx >> 16), ix = ix0 + (int)(short)(dydx & 0xFFFF);
                m |= (uint32_t)(((uint32_t)iy < (uint32_t)d.H) & ((uint32_t)ix < (uint32_t)d.W)) << t;
            }
            dx_ok[q] = m;
        }
#pragma unroll
        for (int q = 0; q < WR; ++q) dw_base[q] = (uint32_t)((n0 + (wv + q * NW) * 8 + lr) * wrow_len + k8 * 8) * 2u;
        typedef __attribute__((address_space(3))) void* lds_ptr;
        // K tail: this lane's chunk of a K-step lies past C once c0b >= klimb.  Its tap / weight offsets are then
        // replaced by MDE_OOB_OFFSET: base + 2 GiB is out of range of either buffer (< 2 GiB each), the DMA writes zeros.
        const uint32_t klimb = (uint32_t)max(d.C - k8 * 8, 0) * 2u;
        auto issue_dma = [&](int buf) {
            // per-tap offsets in bytes (LDS broadcast reads)
            const uint32_t c0b = (uint32_t)(lcs * BK) * 2u;
            const bool kin = c0b < klimb;
            const uint32_t tapoff = kin ? (uint32_t)s_tap[ltap] * 2u + c0b : MDE_OOB_OFFSET;
            const uint32_t woff = kin ? (uint32_t)s_tap[MDE_MAX_TAPS + ltap] * 2u + c0b : MDE_OOB_OFFSET;
            const uint32_t bit = 1u << ltap;
            char* xbuf = smem + buf * BUF_BYTES + wv * 1024;
#pragma unroll
            for (int q = 0; q < XR; ++q) {
                const uint32_t off = (dx_ok[q] & bit) ? dx_base[q] + tapoff : MDE_OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(xbuf + q * NW * 1024), 16, off, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < WR; ++q) {
                // (named temporary on purpose: with the bare sum as the builtin's argument hipcc 7.2's
                //  host pass drops this kernel's stub without a diagnostic; build.sh checks for that)
                const uint32_t offw = dw_base[q] + woff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(xbuf + XT_BYTES + q * NW * 1024), 16, offw, 0, 0, 0);
            }
            if (++ltap == d.ntaps) { ltap = 0; ++lcs; }
        };
        constexpr int IPS = XR + WR;          // DMA instructions per wave per K-step
        constexpr int DIST = NBUF - 1;        // prefetch distance in K-steps
        int lbuf = 0, cbuf = 0;
        if constexpr (PP) {
            static_assert(!PP || (NW == 8 && NBUF == 2 && CF == 8 && WR == 4), "ping-pong: 8 waves, two LDS buffers, 256 columns");
            // Group A = waves 0-3, group B = waves 4-7 (one of each per SIMD) run the SAME program, B one barrier interval
            // behind A (one extra barrier up front).  A K-step is four intervals
            //   R0: LDS reads of the pixel fragments (both k-halves, kept in registers) and of the first four column
            //       fragments | M0: their 32 MFMAs | R1: reads of the other four column fragments | M1: 32 MFMAs
            // so one group's M intervals are the other's R intervals: the matrix pipe always has a cluster to run, and
            // the LDS / DMA traffic of one group hides under the other's MFMAs.
            // DMA: after R0(s) of both groups the pixel tile and the first-half weight rows of buffer s&1 are dead (the
            // pixel fragments live in registers), after R1(s) the rest.  So step s+2's pieces go out in two batches:
            //   "early" (half the pixel pieces + weight rows of the first column half) in R1(s),
            //   "late"  (the other pixel pieces + the second-half weight rows) in R0(s+1),
            // 4 + 4 pieces per wave: every piece has >= 3 intervals (a late one) or 5 (an early one) to land, against one
            // K-step = 4 intervals in the lockstep loop, with the same two buffers.  A wave waits (counted vmcnt: the
            // newest early batch stays in flight) at the end of M1(s) (group A) / R1(s) (group B): both in front of the
            // barrier after which step s+1 is first read.  No DMA is issued inside an M interval (measured +110 cycles
            // per piece there).
            constexpr int XE = (XR + 1) / 2;                    // pixel pieces of the early batch
            constexpr int NEARLY = XE + 2;                      // pieces per early batch
            const bool grpB = wv >= NW / 2;
            int etap = 0, ecs = 0, ltap2 = 0, lcs2 = 0;        // load cursors of the early / late batch streams
            auto advance = [&](int& tap, int& cs) { if (++tap == d.ntaps) { tap = 0; ++cs; } };
            auto issue_part = [&](int buf, int tap, int cs, int x0, int x1, int wpar) {
                const uint32_t c0b = (uint32_t)(cs * BK) * 2u;
                const bool kin = c0b < klimb;
                const uint32_t tapoff = kin ? (uint32_t)s_tap[tap] * 2u + c0b : MDE_OOB_OFFSET;
                const uint32_t woff = kin ? (uint32_t)s_tap[MDE_MAX_TAPS + tap] * 2u + c0b : MDE_OOB_OFFSET;
                const uint32_t bit = 1u << tap;
                char* xbuf = smem + buf * BUF_BYTES + wv * 1024;
#pragma unroll
                for (int q = 0; q < XR; ++q)
                    if (q >= x0 && q < x1) {
                        const uint32_t off = (dx_ok[q] & bit) ? dx_base[q] + tapoff : MDE_OOB_OFFSET;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(xbuf + q * NW * 1024), 16, off, 0, 0, 0);
                    }
#pragma unroll
                for (int q = 0; q < WR; ++q)
                    if ((q & 1) == wpar) {                      // pieces 0, 2: weight rows of the waves' first column halves
                        const uint32_t offw = dw_base[q] + woff;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(xbuf + XT_BYTES + q * NW * 1024), 16, offw, 0, 0, 0);
                    }
            };
            auto issue_early = [&](int buf) { issue_part(buf, etap, ecs, 0, XE, 0); advance(etap, ecs); };
            auto issue_late = [&](int buf) { issue_part(buf, ltap2, lcs2, XE, XR, 1); advance(ltap2, lcs2); };
            bf16x8_t fa[4][2], fb[PF][2];
#define PP_BARRIER()                         \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0)
#ifdef MDE_PP_STAMP
            // diagnostic build: cycles per phase of waves 0 and 4 of one workgroup into the `stats` buffer (as uint64):
            // [0] R body, [1] barrier after R, [2] M body, [3] barrier after M, [4] vmcnt waits
            uint64_t tacc[5] = {0, 0, 0, 0, 0};
#define PP_T() __builtin_amdgcn_s_memtime()
#define PP_ADD(i, t0) tacc[i] += PP_T() - (t0)
#else
#define PP_T() 0
#define PP_ADD(i, t0)
#endif
            issue_early(0);
            issue_late(0);                                      // step 0, both batches
            if (nsteps > 1) {
                issue_early(1);                                 // step 1's early batch ("R1(-1)")
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEARLY) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            PP_BARRIER();
            if (grpB) { PP_BARRIER(); }                         // the stagger: pairs with the end of A's R0(0)
            for (int s = 0; s < nsteps; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                const char* xb = smem + cur * BUF_BYTES + wp * (PF * 2048);
                const char* wb = smem + cur * BUF_BYTES + XT_BYTES + wc * (CF * 2048);
                // ---- R0
                [[maybe_unused]] uint64_t t0 = PP_T();
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int j = 0; j < PF; ++j) fb[j][kb] = *reinterpret_cast<const bf16x8_t*>(xb + j * 2048 + rd_off[kb]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[i][kb] = *reinterpret_cast<const bf16x8_t*>(wb + i * 2048 + rd_off[kb]);
                }
                if (s + 1 < nsteps) issue_late(nxt);            // step s+1: second pixel half + second-half weight rows
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_ADD(0, t0); t0 = PP_T();
                PP_BARRIER();
                PP_ADD(1, t0); t0 = PP_T();
                // ---- M0
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < PF; ++j)
                            acc[i][j] = MDE_MFMA_16x16x32(fa[i][kb], fb[j][kb], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
                PP_ADD(2, t0); t0 = PP_T();
                PP_BARRIER();
                PP_ADD(3, t0); t0 = PP_T();
                // ---- R1
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[i][kb] = *reinterpret_cast<const bf16x8_t*>(wb + (4 + i) * 2048 + rd_off[kb]);
                const bool ahead = s + 2 < nsteps;
                if (ahead) issue_early(cur);                    // step s+2 into the parts of this buffer nobody reads any more
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_ADD(0, t0); t0 = PP_T();
                if (grpB) {
                    if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEARLY) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                PP_ADD(4, t0); t0 = PP_T();
                PP_BARRIER();
                PP_ADD(1, t0); t0 = PP_T();
                // ---- M1
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < PF; ++j)
                            acc[4 + i][j] = MDE_MFMA_16x16x32(fa[i][kb], fb[j][kb], acc[4 + i][j]);
                __builtin_amdgcn_s_setprio(0);
                PP_ADD(2, t0); t0 = PP_T();
                if (!grpB) {
                    if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEARLY) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                PP_ADD(4, t0); t0 = PP_T();
                PP_BARRIER();
                PP_ADD(3, t0);
            }
#ifdef MDE_PP_STAMP
            if (a.stats && blockIdx.x == 7 && lane == 0 && (wv == 0 || wv == 4)) {
                uint64_t* o = reinterpret_cast<uint64_t*>(a.stats) + (wv ? 8 : 0);
                for (int i = 0; i < 5; ++i) o[i] = tacc[i];
                o[5] = (uint64_t)nsteps;
            }
#endif
#undef PP_T
#undef PP_ADD
            if (!grpB) { PP_BARRIER(); }                        // A's matching barrier for B's stagger
#undef PP_BARRIER
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else if constexpr (HALO) {
            // K-steps run chunk-major: (64-channel chunk cc, tap t); the window of chunk cc is fetched once, ahead of its
            // first tap.  The window's rows are read at ANY row offset (tile row x tap), so its bank swizzle is
            // k-slot ^ (h & 7): every run of 16 consecutive rows then covers the 16 16-byte bank groups once whatever its
            // start (with the weight tile's (row >> 1) & 7 only runs starting at a multiple of 4 do).
            const int ninstr = (a.h_rows + 7) >> 3;
            const int ks4 = lane >> 4;
            const int k8h = (lane & 7) ^ lr;                   // h & 7 == lr: an instruction starts at a multiple of 8 rows
            const uint32_t klimbh = (uint32_t)max(d.C - k8h * 8, 0) * 2u;
#ifdef MDE_SB_STAMP
#define SB_T() __builtin_amdgcn_s_memtime()
#else
#define SB_T() 0
#endif
            auto issue_window = [&](int cc) {
                const uint32_t c0b = (uint32_t)(cc * BK) * 2u;
                const bool kinh = c0b < klimbh;
                int iy = h_iy0, ix = h_ix0, hrow = wv * 8 + lr;
                const int xlim = h_gx0 + a.h_dx0 + a.h_hw;              // one past the window's last input column
#pragma unroll
                for (int q = 0; q < HALO_MAX_Q; ++q) {
                    if (q * NW + wv < ninstr) {               // wave-uniform
                        const int py = iy * a.h_dil + h_ry, px = ix * a.h_dil + h_rx;       // class coordinates -> input pixel
                        const bool ok = kinh & (hrow < a.h_rows) & ((uint32_t)py < (uint32_t)d.H) & ((uint32_t)px < (uint32_t)d.W);
                        const uint32_t off = ok ? (uint32_t)(((h_n * d.H + py) * d.W + px) * d.ld_in + k8h * 8) * 2u + c0b : MDE_OOB_OFFSET;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(smem + (q * NW + wv) * 1024), 16, off, 0, 0, 0);
                    }
                    hrow += NW * 8;                           // the next instruction of this wave: NW x 8 window rows on
                    ix += a.h_step_x;
                    iy += a.h_step_y;
                    if (ix >= xlim) { ix -= a.h_hw; ++iy; }
                }
            };
            auto issue_weights = [&](int buf, int cc, int t) {
                const uint32_t c0b = (uint32_t)(cc * BK) * 2u;
                const uint32_t woff = c0b < klimb ? (uint32_t)s_tap[MDE_MAX_TAPS + t] * 2u + c0b : MDE_OOB_OFFSET;
                char* wdst = smem + halo_bytes + buf * WT_BYTES + wv * 1024;
#pragma unroll
                for (int q = 0; q < WR; ++q) {
                    const uint32_t offw = dw_base[q] + woff;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(wdst + q * NW * 1024), 16, offw, 0, 0, 0);
                }
            };
            auto window_addr = [&](int t, int (&ha)[PF]) {    // row h of the window, k-slot XORed with h & 7 (the DMA's swizzle)
                const int toff = s_tap[t];
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const int h = hb[j] + toff;
                    ha[j] = h * 128 + ((ks4 ^ (h & 7)) << 4);
                }
            };
            // half: the chunk has at most 32 channels (a 32-channel layer; the tail of 152 = 64 + 64 + 24): the upper half of the
            // K-step is zero fill on both operands, its MFMAs and fragment reads are skipped (wave-uniform)
            auto compute_step = [&](int buf, const int (&ha)[PF], bool half) {
                const char* wbuf = smem + halo_bytes + buf * WT_BYTES + wc * (CF * 2048);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (kb == 1 && half) break;
                    bf16x8_t fa[CF], fb[PF];
#pragma unroll
                    for (int i = 0; i < CF; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(wbuf + i * 2048 + rd_off[kb]);
#pragma unroll
                    for (int j = 0; j < PF; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(smem + (ha[j] ^ (kb << 6)));
#pragma unroll
                    for (int i = 0; i < CF; ++i)
#pragma unroll
                        for (int j = 0; j < PF; ++j)
                            acc[i][j] = MDE_MFMA_16x16x32(fa[i], fb[j], acc[i][j]);
                }
            };
            int cc = 0, t = 0;
#ifdef MDE_SB_STAMP
            sb_loop0 = SB_T();
#endif
            if constexpr (NBUF == 1) {
                for (int s = 0; s < nsteps; ++s) {
                    if (s) __builtin_amdgcn_s_barrier();      // every wave is done reading step s-1 (weights; window if t == 0)
                    if (t == 0) issue_window(cc);
                    issue_weights(0, cc, t);
                    int ha[PF];
                    window_addr(t, ha);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    compute_step(0, ha, d.C - cc * BK <= 32);
                    if (++t == d.ntaps) { t = 0; ++cc; }
                }
            } else {
                // 8 waves, 256 pixels, a two-deep ring for the weight tiles: the weights of step s+1 are fetched under the
                // MFMAs of step s (a wave issues 2 weight pieces per 32 MFMAs here, against 8 pieces in the plain 128x128
                // tile: the wave's issue slot, ~180 cycles per LDS-DMA instruction, was what the plain loop spent half its
                // K-step in).  The window has ONE buffer: at a chunk boundary every wave must have finished the old chunk's
                // last tap before the new window is fetched, so that fetch is exposed once per chunk (ntaps steps); the
                // other workgroup of the CU runs meanwhile.
                issue_window(0);
                issue_weights(0, 0, 0);
                for (int s = 0; s < nsteps; ++s) {
                    const int cur = s & 1;
                    int tn = t + 1, cn = cc;
                    if (tn == d.ntaps) { tn = 0; ++cn; }
                    int ha[PF];
                    window_addr(t, ha);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of step s (and of a new window)
                    __builtin_amdgcn_s_barrier();                         // everyone's are in; everyone left step s-1
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < nsteps) issue_weights(cur ^ 1, cn, tn);
                    compute_step(cur, ha, d.C - cc * BK <= 32);
                    if (tn == 0 && s + 1 < nsteps) {
                        __builtin_amdgcn_s_barrier();                     // the old window is dead
                        issue_window(cn);
                    }
                    t = tn;
                    cc = cn;
                }
            }
#ifdef MDE_SB_STAMP
            sb_loop1 = SB_T();
#endif
            __syncthreads();
            // a tile may hang over the image's edge: those pixels' results are never stored and must not reach the
            // BatchNorm sums (their window rows are real pixels of the image)
            if (a.stats) {
#pragma unroll
                for (int j = 0; j < PF; ++j)
                    if (!((pvalid >> j) & 1u))
#pragma unroll
                        for (int i = 0; i < CF; ++i) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        } else if constexpr (NBUF == 1) {
            // Short-K shapes (1x1 convolutions over 64 ... 256 channels are one to four K-steps: all prologue and
            // epilogue, HBM-bound): ONE staging buffer, so that twice as many workgroups fit a CU and one workgroup's
            // operand fetch overlaps its neighbours' epilogue stores — the overlap a ring inside one workgroup cannot give
            // a loop this short.
#ifdef MDE_SB_STAMP
            // diagnostic build: cycles of wave 0 of one mid-grid workgroup per loop phase, into the `stats` buffer (uint64):
            // [0] top barrier, [1] DMA issue, [3] vmcnt wait, [4] barrier after the wait, [5] LDS reads + MFMA issue, [6] K-steps
            uint64_t tacc[7] = {0, 0, 0, 0, 0, 0, 0};
#define SB_ADD(i, tt) tacc[i] += SB_T() - (tt)
            sb_loop0 = SB_T();
#else
#define SB_ADD(i, tt)
#endif
            for (int s = 0; s < nsteps; ++s) {
                [[maybe_unused]] uint64_t t0 = SB_T();
                if (s) __builtin_amdgcn_s_barrier();          // every wave is done reading step s-1
                SB_ADD(0, t0); t0 = SB_T();
                issue_dma(0);
                SB_ADD(1, t0); t0 = SB_T();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SB_ADD(3, t0); t0 = SB_T();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                SB_ADD(4, t0); t0 = SB_T();
                compute_half(0, 0);
                if (d.C - (s / d.ntaps) * BK > 32) compute_half(0, 1);      // (a chunk of <= 32 channels: the upper half is zero fill)
                SB_ADD(5, t0);
            }
#ifdef MDE_SB_STAMP
            sb_loop1 = SB_T();
            if (a.stats && blockIdx.x == gridDim.x / 2 && tid == 0) {
                uint64_t* o = reinterpret_cast<uint64_t*>(a.stats + MDE_STAT_SLOTS * 2 * d.ncols);   // behind the partial sums
                for (int i = 0; i < 6; ++i) o[i] = tacc[i];
                o[6] = (uint64_t)nsteps;
            }
#endif
            __syncthreads();
        } else {
#pragma unroll
        for (int q = 0; q < DIST; ++q)
            if (q < nsteps) { issue_dma(lbuf); lbuf = lbuf + 1 == NBUF ? 0 : lbuf + 1; }
        // One wave per SIMD (4 waves x 128x128 wave tiles): nothing else hides LDS latency, so the
        // fragment reads are software-pipelined one fragment ahead of the MFMAs that use them (the
        // compiler's own order was read -> wait -> 8 MFMAs per pixel fragment: pipe idle on every
        // wait) and the next K-step's DMA instructions are spread over the second half's MFMAs.
        constexpr bool W4 = (NW == 4 && PF == 8 && CF == 8);
        if constexpr (W4) {
            static_assert(!W4 || (XR == PF && WR == PF && DIST == 1), "one x and one w DMA per pixel fragment");
            // The K-step barrier sits in the LAST pixel-fragment iteration of a step, after every LDS read of
            // the current buffer has completed (lgkmcnt(0)): the first fragments of the next step are then
            // fetched from the other buffer under that iteration's MFMAs, so a step starts with its operands
            // in registers instead of behind vmcnt(0) + barrier + nine reads.
            bf16x8_t fa0[CF], fa1[CF], fb[2];
            {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const char* xb = smem + wp * (PF * 2048);
                const char* wb = smem + XT_BYTES + wc * (CF * 2048);
#pragma unroll
                for (int i = 0; i < CF; ++i) fa0[i] = *reinterpret_cast<const bf16x8_t*>(wb + i * 2048 + rd_off[0]);
                fb[0] = *reinterpret_cast<const bf16x8_t*>(xb + rd_off[0]);
            }
            for (int s = 0; s < nsteps; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                const char* xb = smem + cbuf * BUF_BYTES + wp * (PF * 2048);
                const char* wb = smem + cbuf * BUF_BYTES + XT_BYTES + wc * (CF * 2048);
                auto RA = [&](int kb, int i) { return *reinterpret_cast<const bf16x8_t*>(wb + i * 2048 + rd_off[kb]); };
                auto RB = [&](int kb, int j) { return *reinterpret_cast<const bf16x8_t*>(xb + j * 2048 + rd_off[kb]); };
                // the other buffer was released by the previous barrier: the next K-step's DMA goes out during
                // k-half 0, one x and one w instruction per pixel fragment
#if MDE_ABLATE == 5
                const bool more = false;               // diagnostics: DMA instructions issue, nothing is fetched
#else
                const bool more = s + DIST < nsteps;
#endif
                const uint32_t c0b = (uint32_t)(lcs * BK) * 2u;
                const bool kin = c0b < klimb;
                const uint32_t tapoff = kin ? (uint32_t)s_tap[ltap] * 2u + c0b : MDE_OOB_OFFSET;
                const uint32_t woff = kin ? (uint32_t)s_tap[MDE_MAX_TAPS + ltap] * 2u + c0b : MDE_OOB_OFFSET;
                const uint32_t bitm = more ? 1u << ltap : 0u;
                char* dbuf = smem + lbuf * BUF_BYTES + wv * 1024;
                __builtin_amdgcn_sched_barrier(0);
                // sched_group_barrier masks: 0x008 MFMA, 0x100 LDS read, 0x020 VMEM read, 0x002 VALU.  Every
                // non-MFMA instruction of an iteration is placed in the shadow of an MFMA (16 pipe cycles,
                // 4 issue cycles): issued in a clump between MFMA groups they left the pipe idle.
#pragma unroll
                for (int j = 0; j < PF; ++j) {         // k-half 0; fetch B[j+1] and the j-th A fragment of half 1
                    fb[(j + 1) & 1] = j + 1 < PF ? RB(0, j + 1) : RB(1, 0);
                    fa1[j] = RA(1, j);
                    {   // unconditional (a branch would split the scheduling region): after the last K-step the
                        // offsets are out of range and the DMA writes zeros into the buffer nobody reads again
                        const uint32_t off = (dx_ok[j] & bitm) ? dx_base[j] + tapoff : MDE_OOB_OFFSET;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(dbuf + j * NW * 1024), 16, off, 0, 0, 0);
                        const uint32_t offw = more ? dw_base[j] + woff : MDE_OOB_OFFSET;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(dbuf + XT_BYTES + j * NW * 1024), 16, offw, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < CF; ++i)
                        acc[i][j] = MDE_MFMA_16x16x32(fa0[i], fb[j & 1], acc[i][j]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < PF - 1; ++j) {     // k-half 1
                    fb[(j + 1) & 1] = RB(1, j + 1);
#pragma unroll
                    for (int i = 0; i < CF; ++i)
                        acc[i][j] = MDE_MFMA_16x16x32(fa1[i], fb[(PF + j) & 1], acc[i][j]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // last iteration: every read of this buffer is in (lgkmcnt(0)), this wave's DMA of the next
                // step landed (vmcnt(0)); after the barrier the other buffer is complete and this one is free
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#if MDE_ABLATE != 4
                __builtin_amdgcn_s_barrier();
#endif
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int nb = cbuf + 1 == NBUF ? 0 : cbuf + 1;
                    const char* nxb = smem + nb * BUF_BYTES + wp * (PF * 2048);
                    const char* nwb = smem + nb * BUF_BYTES + XT_BYTES + wc * (CF * 2048);
#pragma unroll
                    for (int i = 0; i < CF; ++i) fa0[i] = *reinterpret_cast<const bf16x8_t*>(nwb + i * 2048 + rd_off[0]);
                    fb[0] = *reinterpret_cast<const bf16x8_t*>(nxb + rd_off[0]);
#pragma unroll
                    for (int i = 0; i < CF; ++i)
                        acc[i][PF - 1] = MDE_MFMA_16x16x32(fa1[i], fb[(PF + PF - 1) & 1], acc[i][PF - 1]);
#pragma unroll
                    for (int i = 0; i < CF; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (more) {
                    if (++ltap == d.ntaps) { ltap = 0; ++lcs; }
                    lbuf = lbuf + 1 == NBUF ? 0 : lbuf + 1;
                }
                cbuf = cbuf + 1 == NBUF ? 0 : cbuf + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (zero) DMA must land before the epilogue reuses LDS
        } else
        for (int s = 0; s < nsteps; ++s) {
            // step s landed once only the younger groups remain outstanding: min(DIST - 1, steps still to come) of them
            {
                const int rem = nsteps - 1 - s;
                static_assert(IPS * (DIST > 1 ? DIST - 1 : 1) <= 63, "vmcnt is a 6-bit count");
                if (DIST >= 2 && rem >= DIST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPS * (DIST > 1 ? DIST - 1 : 0)) : "memory");
                else if (DIST >= 5 && rem == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPS * 3 <= 63 ? IPS * 3 : 0) : "memory");
                else if (DIST >= 4 && rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPS * 2 <= 63 ? IPS * 2 : 0) : "memory");
                else if (DIST >= 3 && rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();     // every wave's DMA of step s is in; everyone left step s-1
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MDE_DMA_MID) {
                compute_half(cbuf, 0);            // reads -> MFMAs start right behind the barrier ...
                __builtin_amdgcn_sched_barrier(0);
#if MDE_ABLATE != 5
                if (s + DIST < nsteps) { issue_dma(lbuf); lbuf = lbuf + 1 == NBUF ? 0 : lbuf + 1; }   // ... DMA issue under them
#endif
                __builtin_amdgcn_sched_barrier(0);
                compute_half(cbuf, 1);
            } else {
#if MDE_ABLATE != 5
                if (s + DIST < nsteps) { issue_dma(lbuf); lbuf = lbuf + 1 == NBUF ? 0 : lbuf + 1; }
#endif
                compute(cbuf);
            }
            cbuf = cbuf + 1 == NBUF ? 0 : cbuf + 1;
        }
        __syncthreads();                      // staging ring is free for the epilogue
        }   // !PP
    } else
    if constexpr (DEEP) {
        issue_loads(xa, wa);                              // step 0
        if (nsteps > 1) issue_loads(xb, wb_);             // step 1
        stage_write(0, xa, wa);
        __syncthreads();
#if MDE_ABLATE == 1 || MDE_ABLATE == 2
        stage_write(1, xb, wb_);
        __syncthreads();
#endif
        for (int s = 0; s < nsteps; s += 2) {
            // even step s: LDS buffer 0; set B holds step s+1 (in flight); set A is free
            if (s + 2 < nsteps) LOOP_LOADS(xa, wa);
            compute(0);
            if (s + 1 < nsteps) LOOP_WRITE(1, xb, wb_);
            LOOP_SYNC();
            if (s + 1 >= nsteps) break;
            // odd step s+1: LDS buffer 1; set A holds step s+2 (in flight); set B is free
            if (s + 3 < nsteps) LOOP_LOADS(xb, wb_);
            compute(1);
            if (s + 2 < nsteps) LOOP_WRITE(0, xa, wa);
            LOOP_SYNC();
        }
    } else {
        issue_loads(xa, wa);
        stage_write(0, xa, wa);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const bool more = s + 1 < nsteps;
            if (more) LOOP_LOADS(xa, wa);
            compute(s & 1);
            if (more) LOOP_WRITE((s & 1) ^ 1, xa, wa);
            LOOP_SYNC();
        }
    }
#undef LOOP_LOADS
#undef LOOP_WRITE
#undef LOOP_SYNC

    // ---------------------------------------------------------------- epilogue
    // (the loop's last barrier guarantees every wave is done reading the staging tiles)
    if (a.stats) {
        // per-wave channel sums over its pixels (rows beyond M are exact zeros): packed fp32 over the
        // wave's pixel fragments, then a DPP reduction over the 16 lanes that hold different pixels
        constexpr int SG = CF < 4 ? CF : 4;      // column fragments reduced per group (bounds the live registers)
#pragma unroll
        for (int g = 0; g < CF; g += SG) {
            f32x4_t t1[SG], t2[SG];
#pragma unroll
            for (int i = 0; i < SG; ++i) {
                f32x2_t s1a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2a = {0.f, 0.f}, s2b = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const f32x2_t va = {acc[g + i][j][0], acc[g + i][j][1]}, vb = {acc[g + i][j][2], acc[g + i][j][3]};
                    s1a += va;
                    s1b += vb;
                    s2a = __builtin_elementwise_fma(va, va, s2a);
                    s2b = __builtin_elementwise_fma(vb, vb, s2b);
                }
                t1[i] = f32x4_t{mde_row16_sum(s1a[0]), mde_row16_sum(s1a[1]), mde_row16_sum(s1b[0]), mde_row16_sum(s1b[1])};
                t2[i] = f32x4_t{mde_row16_sum(s2a[0]), mde_row16_sum(s2a[1]), mde_row16_sum(s2b[0]), mde_row16_sum(s2b[1])};
            }
            if ((lane & 15) == 0) {
                const int ch = wc * (CF * 16) + (lane >> 4) * 4 + g * 16;
#pragma unroll
                for (int i = 0; i < SG; ++i) {
                    *reinterpret_cast<f32x4_t*>(s_stat + (wp * 2 + 0) * BC + ch + i * 16) = t1[i];
                    *reinterpret_cast<f32x4_t*>(s_stat + (wp * 2 + 1) * BC + ch + i * 16) = t2[i];
                }
            }
        }
    }
    if constexpr (RED == 0) {
        if (a.bias || (a.act && !a.resid)) {
            // fused epilogue, on the fp32 accumulators: + bias (the lane's 4 channels per column fragment), and the activation
            // where nothing else joins before it
            const int ch0 = n0 + wc * (CF * 16) + (lane >> 4) * 4;
            const int actn = a.resid ? 0 : a.act;
#pragma unroll
            for (int i = 0; i < CF; ++i) {
                float b4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = ch0 + i * 16 + e;
                    b4[e] = (a.bias && c < d.ncols) ? a.bias[c] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < PF; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] = epi_act(acc[i][j][e] + b4[e], actn);
            }
        }
    }
    constexpr int CPR = BC / 8;          // 16-byte chunks per pixel row
    constexpr int RPP = NT / CPR;        // rows per store pass
    static_assert(RPP >= 1 && RPP * CPR == NT, "store tiling");
    static_assert(RPP * 2 * BC * 4 <= STG_BYTES, "the fused BatchNorm-backward sums are combined in the staging area");
    bf16_t* outp = reinterpret_cast<bf16_t*>(a.out);
    // fused BatchNorm-backward sums: every thread stores (and sums) the same 8 channels of every row it handles
    float red1[RED ? 8 : 1] = {}, red2[RED ? 8 : 1] = {}, red3[RED == 2 ? 8 : 1] = {};
    if constexpr (RED) {
        // the site's per-channel constants [mean | 1/std | mask scale | mask shift][BC] (a join: [mean | 1/std] of both sites) wait
        // in the statistics area (no forward statistics in such a launch) and are read per row: held in registers they cost the
        // 64-column tiles a workgroup per CU
        static_assert(WAVES_P >= 2, "the statistics area holds four vectors");
        for (int e = tid; e < 4 * BC; e += NT) {
            const int which = e / BC, c = n0 + (e - which * BC);
            const float* src = which == 0 ? a.red_mu : which == 1 ? a.red_rs : which == 2 ? (a.red_x2 ? a.red_mu2 : a.red_msc)
                                                                                            : (a.red_x2 ? a.red_rs2 : a.red_msh);
            s_stat[e] = (src && c < d.ncols) ? src[c] : 0.f;
        }
    }
#pragma unroll
    for (int ep = 0; ep < EPASS; ++ep) {
        if (ep) __syncthreads();         // previous slab fully stored before it is overwritten
        if (wp / (WAVES_P / EPASS) == ep) {
            const int prow = (wp % (WAVES_P / EPASS)) * (PF * 16) + (lane & 15);
            const int chb = wc * (CF * 16) + (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < CF; ++i)
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
        __syncthreads();
        if (ep == 0 && a.stats) {
            for (int e = tid; e < 2 * BC; e += NT) {
                const int which = e / BC, ch = e - which * BC;
                if (n0 + ch < d.ncols) {
                    float sum = 0.f;
#pragma unroll
                    for (int q = 0; q < WAVES_P; ++q) sum += s_stat[(q * 2 + which) * BC + ch];
                    mde_stat_add(a.stats, d.ncols, (uint32_t)pi, which, n0 + ch, sum, a.det);
                }
            }
        }
        const int chunk = tid % CPR, r0 = tid / CPR;
        const int col = n0 + chunk * 8;
        if (col < d.ncols) {
            const bool full = a.vec_ok && (col + 8 <= d.ncols);
            if (full) {
                // Batches of RB rows.  For accumulate, every LOAD of a batch is issued before the batch's
                // first STORE: loads and stores retire through one in-order counter (vmcnt), so a load
                // issued behind a store waits for it.  The loads are unconditional (rows past the end
                // read row 0 and are not stored): a load under a per-lane condition becomes its own
                // block with a full vmcnt(0) wait, which serialised the old read-modify-write loop.
                constexpr int ROWS_PT = EROWS / RPP;                 // rows per thread per pass
                constexpr int RB = ROWS_PT % 4 == 0 ? 4 : ROWS_PT % 3 == 0 ? 3 : ROWS_PT % 2 == 0 ? 2 : 1;   // bounded by registers
                static_assert(ROWS_PT * RPP == EROWS && ROWS_PT % RB == 0, "store batches");
                // ACC 0: plain store; 1: out += result (read-modify-write); 2: out = act(result + bias + residual), the
                // residual read like ACC 1 reads the old output (same offsets, loads ahead of the stores)
                // RED 1 / 2 / 3: also the BatchNorm-backward sums of the site this gradient belongs to (mask recomputed from the
                // site's input / packed mask bits / no ReLU); the site's input is read like ACC 1 reads the old output
                auto red_off = [&](int oo_, int ldmul, int ldx) -> size_t {      // offset of the row in the site's input
                    const uint32_t o = (uint32_t)max(oo_, 0);
                    return ldmul ? (size_t)o * ldmul : (size_t)(o / (uint32_t)d.ld_out) * ldx;
                };
                auto store_rows = [&](auto acc_tag, auto red_tag) {
                    constexpr int ACC = decltype(acc_tag)::value;
                    constexpr int RM = decltype(red_tag)::value;        // mask mode of the fused sums, 0 = off
                    constexpr int RBL = (RM != 0 && RB > 2) ? 2 : RB;   // (the sums' constants take the registers of two rows in flight)
                    const bf16_t* resp = reinterpret_cast<const bf16_t*>(a.resid);
                    const bf16_t* redx = reinterpret_cast<const bf16_t*>(a.red_x);
                    const bf16_t* redx2 = reinterpret_cast<const bf16_t*>(a.red_x2);
#pragma unroll 1
                    for (int rb = 0; rb < ROWS_PT; rb += RBL) {
                        int oo[RBL];
                        i32x4_t oldv[ACC ? RBL : 1];
                        uint32_t ab[ACC == 3 ? RBL : 1];
                        i32x4_t xin[RM ? RBL : 1];
                        i32x4_t xin2[RM == 4 ? RBL : 1];
                        uint32_t mb[(RM == 2 || RM == 4) ? RBL : 1];
#pragma unroll
                        for (int q = 0; q < RBL; ++q) {
                            oo[q] = s_out[ep * EROWS + r0 + (rb + q) * RPP];
                            if constexpr (ACC == 1) oldv[q] = *reinterpret_cast<const i32x4_t*>(outp + (size_t)max(oo[q], 0) + col);
                            if constexpr (ACC == 3) {      // (ACC 3: the addend comes from another tensor, under that tensor's mask bits)
                                oldv[q] = *reinterpret_cast<const i32x4_t*>(reinterpret_cast<const bf16_t*>(a.add_src) + (size_t)max(oo[q], 0) + col);
                                ab[q] = a.add_bits ? a.add_bits[((size_t)max(oo[q], 0) + col) >> 3] : 0xFFu;
                            }
                            if constexpr (ACC == 2) oldv[q] = resp ? *reinterpret_cast<const i32x4_t*>(resp + (size_t)max(oo[q], 0) + col) : i32x4_t{0, 0, 0, 0};
                            if constexpr (RM != 0) xin[q] = *reinterpret_cast<const i32x4_t*>(redx + red_off(oo[q], a.red_ldmul, a.red_ldx) + col);
                            if constexpr (RM == 2 || RM == 4) mb[q] = a.red_bits[((size_t)max(oo[q], 0) + col) >> 3];
                            if constexpr (RM == 4) xin2[q] = *reinterpret_cast<const i32x4_t*>(redx2 + red_off(oo[q], a.red_ldmul2, a.red_ldx2) + col);
                        }
#pragma unroll
                        for (int q = 0; q < RBL; ++q) {
                            const int r = r0 + (rb + q) * RPP;
                            bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(smem + r * ROWB + chunk * 16);
                            if constexpr (ACC == 1) {
                                const bf16x8_t old = __builtin_bit_cast(bf16x8_t, oldv[q]);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)old[e]);
                            }
                            if constexpr (ACC == 3) {
                                const bf16x8_t old = __builtin_bit_cast(bf16x8_t, oldv[q]);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (((ab[q] >> e) & 1u) ? (float)old[e] : 0.f));
                            }
                            if constexpr (ACC == 2) {
                                const bf16x8_t rv = __builtin_bit_cast(bf16x8_t, oldv[q]);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (bf16_t)epi_act((float)v[e] + (float)rv[e], a.act);     // (the bias is in already)
                            }
                            if (oo[q] >= 0) *reinterpret_cast<bf16x8_t*>(outp + (size_t)oo[q] + col) = v;
                            if constexpr (RM != 0) {
                                // (bn.hip bn_bwd_reduce_k's arithmetic on the value just stored)
                                const bf16x8_t xv = __builtin_bit_cast(bf16x8_t, xin[q]);
                                const bf16x8_t xv2 = __builtin_bit_cast(bf16x8_t, xin2[RM == 4 ? q : 0]);
                                if (oo[q] >= 0) {
#pragma unroll
                                    for (int h = 0; h < 8; h += 4) {
                                        int cofs = chunk * 8 + h;
                                        asm volatile("" : "+v"(cofs));           // (keeps the constants' loads inside the row loop)
                                        const f32x4_t mu = *reinterpret_cast<const f32x4_t*>(s_stat + cofs);
                                        const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(s_stat + BC + cofs);
                                        f32x4_t ms = {0.f, 0.f, 0.f, 0.f}, mh = {0.f, 0.f, 0.f, 0.f};
                                        if constexpr (RM == 1 || RM == 4) {      // (a join: the second site's mean and 1 / std)
                                            ms = *reinterpret_cast<const f32x4_t*>(s_stat + 2 * BC + cofs);
                                            mh = *reinterpret_cast<const f32x4_t*>(s_stat + 3 * BC + cofs);
                                        }
#pragma unroll
                                        for (int e = 0; e < 4; ++e) {
                                            const float xe = (float)xv[h + e];
                                            bool on = true;
                                            if constexpr (RM == 1) on = xe * ms[e] + mh[e] > 0.f;
                                            if constexpr (RM == 2 || RM == 4) on = (mb[q] >> (h + e)) & 1u;
                                            const float ge = on ? (float)v[h + e] : 0.f;
                                            red1[h + e] += ge;
                                            red2[h + e] += ge * ((xe - mu[e]) * rs[e]);
                                            if constexpr (RM == 4) red3[h + e] += ge * (((float)xv2[h + e] - ms[e]) * mh[e]);
                                        }
                                    }
                                }
                            }
                        }
                    }
                };
                using T0 = std::integral_constant<int, 0>;
                using T1 = std::integral_constant<int, 1>;
                using T2 = std::integral_constant<int, 2>;
                using T3 = std::integral_constant<int, 3>;
                using T4 = std::integral_constant<int, 4>;
                if constexpr (RED == 2) {
                    if (a.add_src) store_rows(T3{}, T4{}); else if (d.accumulate) store_rows(T1{}, T4{}); else store_rows(T0{}, T4{});
                } else if constexpr (RED == 1) {
                    if (a.red_bits) {
                        if (a.add_src) store_rows(T3{}, T2{}); else if (d.accumulate) store_rows(T1{}, T2{}); else store_rows(T0{}, T2{});
                    } else if (a.red_msc) {
                        if (d.accumulate) store_rows(T1{}, T1{}); else store_rows(T0{}, T1{});
                    } else {
                        if (d.accumulate) store_rows(T1{}, T3{}); else store_rows(T0{}, T3{});
                    }
                } else {
                    if (a.resid)
                        store_rows(T2{}, T0{});                              // (without a residual the epilogue ran on the accumulators)
                    else if (d.accumulate)
                        store_rows(T1{}, T0{});
                    else
                        store_rows(T0{}, T0{});
                }
            } else {
                for (int r = r0; r < EROWS; r += RPP) {
                    const int oo = s_out[ep * EROWS + r];
                    if (oo < 0) continue;
                    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(smem + r * ROWB + chunk * 16);
                    bf16_t* dst = outp + (size_t)oo + col;
                    const int nv = min(8, d.ncols - col);
                    const bf16_t* resp = reinterpret_cast<const bf16_t*>(a.resid);
                    for (int e = 0; e < nv; ++e) {
                        float x = (float)v[e];
                        if (resp) {
                            x = epi_act(x + (float)resp[(size_t)oo + col + e], a.act);
                        } else if (d.accumulate) {
                            x += (float)dst[e];
                        }
                        dst[e] = (bf16_t)x;
                    }
                }
            }
        }
    }
    if constexpr (RED) {
        // row lanes combined in a fixed order (as bn.hip's deterministic flush does), one addend per column and workgroup
        __syncthreads();                 // the last slab has been read out of the staging area
        float* red = reinterpret_cast<float*>(smem);     // [RPP][2][BC]
        const int chunk = tid % CPR, rl = tid / CPR;
        // (as 16-byte stores: eight scalar stores 32 bytes apart put 16 lanes on 4 banks -- 31 % of this form's LDS cycles
        //  were bank conflicts, profiles/r03_pmc_gemm_in_network.txt)
        *reinterpret_cast<f32x4_t*>(red + (rl * 2 + 0) * BC + chunk * 8) = f32x4_t{red1[0], red1[1], red1[2], red1[3]};
        *reinterpret_cast<f32x4_t*>(red + (rl * 2 + 0) * BC + chunk * 8 + 4) = f32x4_t{red1[4], red1[5], red1[6], red1[7]};
        *reinterpret_cast<f32x4_t*>(red + (rl * 2 + 1) * BC + chunk * 8) = f32x4_t{red2[0], red2[1], red2[2], red2[3]};
        *reinterpret_cast<f32x4_t*>(red + (rl * 2 + 1) * BC + chunk * 8 + 4) = f32x4_t{red2[4], red2[5], red2[6], red2[7]};
        __syncthreads();
        for (int e = tid; e < 2 * BC; e += NT) {
            const int which = e / BC, ch = e - which * BC;
            if (n0 + ch < d.ncols) {
                float sum = 0.f;
                for (int q = 0; q < RPP; ++q) sum += red[(q * 2 + which) * BC + ch];
                mde_stat_add(a.red_part, d.ncols, (uint32_t)pi, which, n0 + ch, sum, a.det);
                if (which == 0 && RED == 2) mde_stat_add(a.red_part2, d.ncols, (uint32_t)pi, 0, n0 + ch, sum, a.det);
            }
        }
        if constexpr (RED == 2) {        // the second site's sum(g' * xhat), through the same area
            __syncthreads();
            *reinterpret_cast<f32x4_t*>(red + rl * BC + chunk * 8) = f32x4_t{red3[0], red3[1], red3[2], red3[3]};
            *reinterpret_cast<f32x4_t*>(red + rl * BC + chunk * 8 + 4) = f32x4_t{red3[4], red3[5], red3[6], red3[7]};
            __syncthreads();
            for (int ch = tid; ch < BC; ch += NT) {
                if (n0 + ch < d.ncols) {
                    float sum = 0.f;
                    for (int q = 0; q < RPP; ++q) sum += red[q * BC + ch];
                    mde_stat_add(a.red_part2, d.ncols, (uint32_t)pi, 1, n0 + ch, sum, a.det);
                }
            }
        }
    }
#ifdef MDE_SB_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.stats && blockIdx.x == gridDim.x / 2 && tid == 0) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.stats + MDE_STAT_SLOTS * 2 * d.ncols);
        const uint64_t k1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        o[7] = sb_loop0 - sb_k0;      // prologue
        o[8] = sb_loop1 - sb_loop0;   // K-loop
        o[9] = k1 - sb_loop1;         // epilogue (stores drained)
        o[10] = r1 - sb_r0;           // the same span in 100 MHz ticks
    }
#endif
}

// waves of the deep-ring tiles (grids of at most one tile per CU): a K-step there costs what its DMA instructions take to
// ISSUE (~190 cycles each, item 27 of DESIGN section 3) -- eight waves issue a step's pieces in half the time of four.
// bc: 128 = the 4-deep 128-column tile, 64 = the 4-deep 64-column tile, 1 = the 3-deep 64-column tile of grids between one and
// two rounds.  All three run with eight waves.  (The 64-column forms were held back for most of round 4: their fused
// BatchNorm-backward sums came out wrong whenever a weight-gradient workgroup of the other stream shared the CU.  Cause, found
// with tools/probes/conv_concurrency3.py: hipcc's SLP vectoriser had paired the sums' accumulations crosswise in exactly these
// two instances -- v_pk_add_f32 / v_pk_mul_f32 with op_sel:[0,1] -- and on gfx950 the low lane of such an instruction is
// unreliable while a wave of another kernel shares the SIMD.  build.sh now compiles this file without the SLP pass and
// check_isa.py rejects a library that holds such an instruction: DESIGN section 3, item 44.)
// MDE_CONV_DEEP_WAVES: 4 = four waves everywhere, 64 / 128 / d64 = eight for that one form only (diagnostics).
inline int deep_waves(int bc = 0) {
    const char* e = getenv("MDE_CONV_DEEP_WAVES");             // (read per call: the tests switch it between launches)
    if (e && !strcmp(e, "4")) return 4;
    if (e && !strcmp(e, "64")) return bc == 64 ? 8 : 4;
    if (e && !strcmp(e, "128")) return bc == 128 ? 8 : 4;
    if (e && !strcmp(e, "d64")) return bc == 1 ? 8 : 4;
    return 8;
}

template <int BP, int BC, int NT, int NBUF>
constexpr size_t smem_bytes() {
    return NBUF * (size_t)(BP + BC) * BK * 2 + 3 * BP * sizeof(int) +
           ((NT / 64) / waves_c<BP, BC, NT>()) * 2 * BC * sizeof(float) + 3 * MDE_MAX_TAPS * sizeof(int);
}

template <int BP, int BC, int NT, bool DMA, int NBUF, bool PP = false>
int launch(KArgs& ka, int64_t M, hipStream_t st) {
    static bool attr_done = false;
    // (diagnostics, MDE_CONV_PAD_LDS=1: the 8-wave 64-column deep tiles ask for 132 KB so that no other kernel's workgroup
    //  fits a CU beside them)
    constexpr size_t smem0 = smem_bytes<BP, BC, NT, NBUF>();
    constexpr size_t smem = (NT == 512 && BC == 64 && NBUF == 4 && smem0 < 132 * 1024) ? (MDE_PAD_LDS ? 132 * 1024 : smem0) : smem0;
    static_assert(smem <= 160 * 1024, "LDS budget");
    if (!attr_done) {
        int rc = mde_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_nt<BP, BC, NT, DMA, NBUF, PP>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),
                               "hipFuncSetAttribute(conv_gemm_nt)");
        if (rc) return rc;
        attr_done = true;
    }
    (void)M;
    ka.nP = mde_cdiv(ka.m_end - ka.m_begin, BP);
    ka.nC = mde_cdiv(ka.d.ncols, BC);
    if (ka.red_x) {
        static bool red_attr_done[2] = {false, false};
        const int join = ka.red_x2 != nullptr;
        const void* fn = join ? reinterpret_cast<const void*>(&conv_gemm_nt<BP, BC, NT, DMA, NBUF, PP, false, 2>)
                              : reinterpret_cast<const void*>(&conv_gemm_nt<BP, BC, NT, DMA, NBUF, PP, false, 1>);
        if (!red_attr_done[join]) {
            int rc = mde_check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),
                                   "hipFuncSetAttribute(conv_gemm_nt, fused BatchNorm-backward sums)");
            if (rc) return rc;
            red_attr_done[join] = true;
        }
        if (join)
            conv_gemm_nt<BP, BC, NT, DMA, NBUF, PP, false, 2><<<dim3(ka.nP * ka.nC), dim3(NT), smem, st>>>(ka);
        else
            conv_gemm_nt<BP, BC, NT, DMA, NBUF, PP, false, 1><<<dim3(ka.nP * ka.nC), dim3(NT), smem, st>>>(ka);
    } else {
        conv_gemm_nt<BP, BC, NT, DMA, NBUF, PP><<<dim3(ka.nP * ka.nC), dim3(NT), smem, st>>>(ka);
    }
    MDE_LAUNCH_CHECK("conv_gemm_nt");
    return MDE_OK;
}

// ---- the halo-tiled form: eligibility, tile shape, launch
// Eligible: stride-1 gathers (forward 3x3 / dilated 3x3, stride-1 input gradients, the up-projection's and the strided
// input gradient's output phases), >= 2 taps, not grouped, and a window of at most 256 pixels for some 128-pixel block shape.
struct HaloPlan {
    int ok;
    int tws, th, nty, ntx, hw, rows, dy0, dx0, dil;
    double dma_rows;      // 128-byte rows through the DMA path per 64-channel chunk, whole launch (what the form is chosen by)
    int64_t tiles;
};
HaloPlan halo_plan(const mde_conv_desc& d, int bp, int bc) {
    HaloPlan best{};
    if (d.sy != 1 || d.sx != 1 || d.grouped || d.ntaps < 2) return best;
    int dy0 = d.dy[0], dy1 = d.dy[0], dx0 = d.dx[0], dx1 = d.dx[0];
    for (int t = 1; t < d.ntaps; ++t) {
        dy0 = d.dy[t] < dy0 ? d.dy[t] : dy0;
        dy1 = d.dy[t] > dy1 ? d.dy[t] : dy1;
        dx0 = d.dx[t] < dx0 ? d.dx[t] : dx0;
        dx1 = d.dx[t] > dx1 ? d.dx[t] : dx1;
    }
    // Dilation: when every tap offset is a multiple of D (an atrous 3x3: D = its dilation), the output grid splits into D x D
    // parity classes that never share an input pixel; a tile is a block of ONE class, in whose coordinates the taps are D
    // times closer -- the window of a dilated 3x3 is (th + 2) x (tw + 2) pixels like a plain one's, not (th + 2 D) x (tw + 2 D).
    auto gcd = [](int a, int b) { a = a < 0 ? -a : a; b = b < 0 ? -b : b; while (b) { const int t = a % b; a = b; b = t; } return a; };
    int D = 0;
    for (int t = 0; t < d.ntaps; ++t) D = gcd(gcd(D, d.dy[t]), d.dx[t]);
    if (D < 1) D = 1;
    if (D > 8) return best;
    dy0 /= D; dy1 /= D; dx0 /= D; dx1 /= D;
    const int nw = bp / 32;                                   // waves: 4 for the 128-pixel tile, 8 for the 256-pixel one
    const int gh = mde_cdiv(d.GH, D), gw = mde_cdiv(d.GW, D); // a class's grid
    // LDS: the window must leave room for the resident workgroups the form is built for (two 8-wave / four 4-wave per CU)
    const int fixed = (bp == 128 ? 1 : 2) * bc * BK * 2 + bp * 4 + 3 * MDE_MAX_TAPS * 4;
    const int max_rows = ((bp == 128 ? (bc <= 64 ? 32768 : 40960) : 81920) - fixed) / 128;
    for (int tws = 1; (bp >> tws) >= 2; ++tws) {
        const int tw = 1 << tws, th = bp >> tws;
        const int hw = tw + (dx1 - dx0), rows = (th + (dy1 - dy0)) * hw;
        if (rows > HALO_MAX_Q * nw * 8 || ((rows + 7) & ~7) > max_rows) continue;
        const int nty = mde_cdiv(gh, th), ntx = mde_cdiv(gw, tw);
        const int64_t tiles = (int64_t)d.N * D * D * nty * ntx * mde_cdiv(d.ncols, bc);
        const double cost = (double)tiles * (((rows + 7) & ~7) + d.ntaps * bc);
        if (!best.ok || cost < best.dma_rows) best = HaloPlan{1, tws, th, nty, ntx, hw, rows, dy0, dx0, D, cost, tiles};
    }
    return best;
}

template <int BP, int BC>
int launch_halo(KArgs& ka, const HaloPlan& hp, hipStream_t st) {
    constexpr int NT = BP * 2, NBUF = BP == 128 ? 1 : 2, NW = NT / 64;
    static bool attr_done[3] = {false, false, false};
    constexpr size_t fixed = (size_t)NBUF * BC * BK * 2 + BP * sizeof(int) + 3 * MDE_MAX_TAPS * sizeof(int);
    const int red = !ka.red_x ? 0 : ka.red_x2 ? 2 : 1;
    const void* fn = red == 2 ? reinterpret_cast<const void*>(&conv_gemm_nt<BP, BC, NT, true, NBUF, false, true, 2>)
                     : red  ? reinterpret_cast<const void*>(&conv_gemm_nt<BP, BC, NT, true, NBUF, false, true, 1>)
                            : reinterpret_cast<const void*>(&conv_gemm_nt<BP, BC, NT, true, NBUF, false, true>);
    if (!attr_done[red]) {
        int rc = mde_check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)HALO_MAX_Q * NW * 1024 + fixed)),
                               "hipFuncSetAttribute(conv_gemm_nt halo)");
        if (rc) return rc;
        attr_done[red] = true;
    }
    ka.h_tws = hp.tws; ka.h_th = hp.th; ka.h_nty = hp.nty; ka.h_ntx = hp.ntx;
    ka.h_hw = hp.hw; ka.h_rows = hp.rows; ka.h_dy0 = hp.dy0; ka.h_dx0 = hp.dx0; ka.h_dil = hp.dil;
    ka.h_step_y = (NW * 8) / hp.hw; ka.h_step_x = (NW * 8) % hp.hw;
    ka.nP = ka.d.N * hp.dil * hp.dil * hp.nty * hp.ntx;
    ka.nC = mde_cdiv(ka.d.ncols, BC);
    const size_t smem = (size_t)((hp.rows + 7) >> 3) * 1024 + fixed;
    if (getenv("MDE_CONV_OCC")) {       // diagnostics: resident workgroups per CU the runtime computes for this launch
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, NT, smem);
        fprintf(stderr, "conv_gemm_nt<%d,%d,halo>: %zu B LDS, %d workgroups per CU, window %d rows (tile %dx%d), grid %d\n", BP, BC, smem, nb,
                hp.rows, hp.th, 1 << hp.tws, ka.nP * ka.nC);
        if (hp.dil > 1) fprintf(stderr, "   dilation %d: %d parity classes\n", hp.dil, hp.dil * hp.dil);
    }
    if (red == 2)
        conv_gemm_nt<BP, BC, NT, true, NBUF, false, true, 2><<<dim3(ka.nP * ka.nC), dim3(NT), smem, st>>>(ka);
    else if (red)
        conv_gemm_nt<BP, BC, NT, true, NBUF, false, true, 1><<<dim3(ka.nP * ka.nC), dim3(NT), smem, st>>>(ka);
    else
        conv_gemm_nt<BP, BC, NT, true, NBUF, false, true><<<dim3(ka.nP * ka.nC), dim3(NT), smem, st>>>(ka);
    MDE_LAUNCH_CHECK("conv_gemm_nt(halo)");
    return MDE_OK;
}

int cus_() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return cus;
}

// Tile choice: the biggest workgroup tile whose grid still fills the chip about twice over
// (one 8-wave workgroup per CU for the 256-pixel tiles).  Diagnostics: MDE_CONV_TILE=<BP>x<BC>
// forces a tile, MDE_CONV_PATH=reg selects the register-staged main loop instead of LDS-DMA.
int pick_and_launch(KArgs& ka, int64_t M, hipStream_t st) {
    static int forced = -1, reg = 0, gpp = 0;
    if (forced < 0) {
        // 8-wave tiles: MDE_CONV_PP=1 always the ping-pong loop, 0 never; default: where a tile has >= 18 K-steps (measured
        // in-network: +2..4 % on the 9- and 25-tap layers, -3..6 % on the 1x1 layers whose 1-8 K-steps do not amortise
        // its longer prologue)
        const char* ppe = getenv("MDE_CONV_PP");
        gpp = !ppe ? 2 : (strcmp(ppe, "0") != 0);
        const char* e = getenv("MDE_CONV_TILE");
        forced = !e ? 0 : !strcmp(e, "256x256") ? 1 : !strcmp(e, "128x128") ? 3 : !strcmp(e, "192x256") ? 5 : !strcmp(e, "128x64") ? 6 : !strcmp(e, "256x256w4") ? 7 : !strcmp(e, "128x128s") ? 8 : !strcmp(e, "128x256s") ? 9 : !strcmp(e, "128x64s") ? 10 : 0;
        const char* q = getenv("MDE_CONV_PATH");
        reg = q && !strcmp(q, "reg");
    }
    const int n = ka.d.ncols;
    {
        // MDE_CONV_HALO: 1 = the halo-tiled form wherever it is eligible (diagnostics / tests), 0 = never
        // MDE_CONV_HALO (read per call: the tests switch it between launches): 0 = never the halo-tiled form; 1 / 2 = the
        // 128-pixel single-buffer / the 256-pixel two-deep-weight-ring form wherever eligible (diagnostics, tests); unset =
        // the 256-pixel form where it measured faster in-network (tools/per_shape_diff.py over `bench.py --per-shape` runs
        // with the form forced): >= 4 taps (2-tap phases lost 7-24 %), a grid of at least two workgroups per CU for each of the
        // two resident slots (150 x 2 tiles lost 3-12 %, 600 tiles won 4-9 %), and tiles that mostly lie inside the image (a
        // 15 x 20 map covers 59 % of its two 16 x 16 blocks: -10 %).  The 128-pixel form lost to the plain tiles on most
        // shapes (its K-step still waits out a full DMA latency) and is never chosen.
        const char* he = getenv("MDE_CONV_HALO");
        const int halo = !he ? -1 : atoi(he);
        if (halo != 0 && forced == 0 && !reg) {
            // (diagnostics: 3 / 4 = the 128- / 256-pixel form with 64-column tiles whatever the layer's width)
            // 64-column tiles for <= 64 columns, and where 128-column tiles would be mostly padding (VNL's 150-bin
            // prediction conv: 152 columns = 3 x 64 at 79 % against 2 x 128 at 59 %)
            const bool narrow = n <= 64 || mde_cdiv(n, 64) * 64 * 10 <= mde_cdiv(n, 128) * 128 * 8;
            const int bc = (narrow || halo == 3 || halo == 4) ? 64 : 128, bp = (halo == 1 || halo == 3) ? 128 : 256;
            const HaloPlan hp = halo_plan(ka.d, bp, bc);
            bool take = hp.ok;
            if (take && halo < 0) {
                const double fill = (double)M / ((double)ka.d.N * hp.dil * hp.dil * hp.nty * hp.ntx * bp);
                take = ka.d.ntaps >= 4 && hp.tiles >= 2 * cus_() && fill >= 0.75;
            }
            // <= 32 columns on a map of at least a million pixels (BTS' and MiDaS' full-resolution decoder layers): the 128-pixel
            // form with a 32-column tile -- half the MFMAs and weight reads of the 64-column tile, which is mostly padding there
            {
                const char* ne = getenv("MDE_CONV_NARROW");        // (read per call; "0": off, "2": wherever eligible -- tests)
                const int nar = !ne ? 1 : atoi(ne);
                if (n <= 32 && nar && ((halo < 0 && ka.d.ntaps >= 4 && M >= (1 << 20)) || nar == 2)) {
                    const HaloPlan h32 = halo_plan(ka.d, 128, 32);
                    if (h32.ok) return launch_halo<128, 32>(ka, h32, st);
                }
            }
            if (take) {
                if (bp == 128) return bc == 64 ? launch_halo<128, 64>(ka, hp, st) : launch_halo<128, 128>(ka, hp, st);
                return bc == 64 ? launch_halo<256, 64>(ka, hp, st) : launch_halo<256, 128>(ka, hp, st);
            }
        }
    }
    const bool pp = gpp == 1 || (gpp == 2 && ka.d.ntaps * ((ka.d.C + BK - 1) / BK) >= 18);
    if (forced == 10 && !ka.d.grouped) return launch<128, 64, 256, true, 1>(ka, M, st);      // diagnostics: 64-column single-buffer tile everywhere
    {
        // <= 32 columns, one tap, a map of a million pixels or more (BTS' reduc1x1 chain at full resolution): the single-buffer
        // tile with 32 columns (20 KB of LDS: more workgroups per CU, half the padding)
        const char* ne = getenv("MDE_CONV_NARROW");
        const int nar = !ne ? 1 : atoi(ne);
        if (n <= 32 && !ka.d.grouped && !reg && forced == 0 && nar && (M >= (1 << 20) || nar == 2) && ka.d.ntaps * ((ka.d.C + BK - 1) / BK) <= 2)
            return launch<128, 32, 256, true, 1>(ka, M, st);
    }
    if (n <= 64 || ka.d.grouped) {
        // 2-deep ring = 48 KB LDS = three workgroups per CU.  Measured alternatives, all slower on M = 2 457 600 / 614 400,
        // 64->64 3x3: 256x64 with 8 waves (423 / 423 TFLOP/s), 256x64 with 4 waves (365 / 343), 3-deep ring at two
        // workgroups per CU (443 / 456) against 487 / 576: occupancy beats prefetch depth and bigger tiles here.
        // and a single 24 KB buffer at FIVE workgroups per CU beats the 2-deep ring at three on every 64-column shape of the
        // step (in-network: 9 taps at 2 457 600 pixels 348 -> 301 us, at 614 400 90 -> 77; 1x1 256 -> 64: 81 -> 75)
        static int ring64 = -1;
        if (ring64 < 0) {
            const char* e = getenv("MDE_CONV_RING64");
            ring64 = !e ? -2 : atoi(e);          // diagnostics: 0 never a ring, 1 the 2-deep ring everywhere, 3 the 3-deep ring everywhere
        }
        if (reg) return launch<128, 64, 256, false, 2>(ka, M, st);
        // A grid that leaves CUs with one workgroup or none (DenseNet's 3x3 192 -> 48 on 19 200 or 4 800 pixels: 150 or 38
        // tiles of 27 K-steps) is a chain of DMA latencies: no neighbour overlaps them, so the tile prefetches two K-steps
        // ahead itself (3-deep ring, 72 KB).
        const int64_t tiles64 = (int64_t)mde_cdiv(M, 128) * mde_cdiv(n, 64);
        if (ring64 == 3) return launch<128, 64, 256, true, 3>(ka, M, st);
        const char* de = getenv("MDE_CONV_DEEP");
        const int deep64 = !(de && !strcmp(de, "0"));
        if (ring64 == 4 || (ring64 == -2 && deep64 && forced == 0 && tiles64 <= MDE_RING3_MAX_TILES_PER_CU * cus_() && ka.d.ntaps * ((ka.d.C + BK - 1) / BK) >= 4))
            return deep_waves(64) == 8 ? launch<128, 64, 512, true, MDE_DEEP_RING>(ka, M, st) : launch<128, 64, 256, true, MDE_DEEP_RING>(ka, M, st);
        return (ring64 == 1 || forced == 6) ? launch<128, 64, 256, true, 2>(ka, M, st) : launch<128, 64, 256, true, 1>(ka, M, st);
    }
    const int cus = cus_();
    // Cost model fitted to in-network timings: kernel time = rounds x work per resident
    // slot per round / per-flop rate; a CU hosts one 8-wave workgroup (256x256: rate 1.15, 192x256: 1.10)
    // or two 4-wave 128x128 workgroups (rate 1.0).  What decides between them is the tail round.
    static double r256 = 0.0, r192 = 0.0;                       // per-flop rates relative to the 128x128 tile
    if (r256 == 0.0) {
        r256 = MDE_RATE_256;
        r192 = MDE_RATE_192;
        if (const char* e = getenv("MDE_CONV_RATES")) sscanf(e, "%lf,%lf", &r256, &r192);   // diagnostics: refit
    }
    const int nc256 = mde_cdiv(n, 256), nc128 = mde_cdiv(n, 128);
    // Short K (1x1 convolutions over <= 512 channels: at most 8 K-steps, HBM-bound, mostly prologue and epilogue): the
    // single-buffer 128x128 tile at four workgroups per CU, whose fetches overlap the neighbours' epilogue stores.  Measured
    // in-network (tools/short_k_sweep.sh): -12 % summed over the 1x1 shapes of the step that qualify, e.g. 64 -> 256 channels
    // at 614 400 pixels 116 -> 97 us (accumulating: 158 -> 123); shapes with 16+ K-steps or too few tiles are not faster.
    {
        static int sk = -1;
        if (sk < 0) {
            const char* e = getenv("MDE_CONV_SHORTK");
            sk = !(e && !strcmp(e, "0"));
        }
        const int nst = ka.d.ntaps * ((ka.d.C + BK - 1) / BK);
        if (sk && forced == 0 && !reg && nst <= 8 && (int64_t)mde_cdiv(M, 128) * nc128 >= 2 * cus)
            return launch<128, 128, 256, true, 1>(ka, M, st);
    }
    {
        // a grid of at most one 128x128 tile per CU (DenseNet's 3x3 input gradients, 48 -> 192 channels on 19 200 or 4 800 pixels):
        // a chain of DMA latencies, see the 64-column rule above -- the deep ring (MDE_CONV_DEEP=0: off)
        const char* de = getenv("MDE_CONV_DEEP");              // (read per call: the tests switch it between launches)
        const int deep = !(de && !strcmp(de, "0"));
        const int nst = ka.d.ntaps * ((ka.d.C + BK - 1) / BK);
        if (deep && forced == 0 && !reg && nst >= 4 && (int64_t)mde_cdiv(M, 128) * nc128 <= MDE_RING3_MAX_TILES_PER_CU * cus)
            return deep_waves(128) == 8 ? launch<128, 128, 512, true, MDE_DEEP_RING>(ka, M, st) : launch<128, 128, 256, true, MDE_DEEP_RING>(ka, M, st);
        // ... and a grid that is a little more than one 128-column tile per CU but at most two 64-column tiles (DenseNet's 3x3
        // input gradients on 19 200 pixels, 48 -> 192 channels: 300 / 450 tiles): the 8-wave 64-column tile with a 3-deep ring
        // (72 KB: two per CU), everything resident in one round.  MDE_CONV_DEEP64=0: off.
        const char* d6 = getenv("MDE_CONV_DEEP64");
        if (deep && !(d6 && !strcmp(d6, "0")) && forced == 0 && !reg && nst >= 4 && nst <= 32 && (int64_t)mde_cdiv(M, 128) * nc128 <= 2 * cus &&
            (int64_t)mde_cdiv(M, 128) * mde_cdiv(n, 64) <= 2 * cus && deep_waves(1) == 8)
            return launch<128, 64, 512, true, 3>(ka, M, st);
    }
    const int64_t p256 = mde_cdiv(M, 256), t256 = p256 * nc256;
    auto rounds = [](int64_t tiles, int64_t slots) { return (tiles + slots - 1) / slots; };
    const int64_t t128 = (int64_t)mde_cdiv(M, 128) * nc128;
    const double c128 = (double)rounds(t128, 2 * cus) * 32768.0;
    double best = c128;
    int pick = 0;                                               // 0: 128x128, 1: 256x256, 2: 192x256, 3: 256x256 + 128x128 tail, 4: 128x128 single buffer
    int32_t split = 0;
    if (n >= 256 && !reg) {
        const double c256 = (double)rounds(t256, cus) * 65536.0 / r256;
        const double c192 = (double)rounds((int64_t)mde_cdiv(M, 192) * nc256, cus) * 49152.0 / r192;
        if (c256 < best) { best = c256; pick = 1; }
        if (c192 < best) { best = c192; pick = 2; }
        if (t256 > cus) {                                       // full 256x256 rounds, remaining pixel rows on 128x128 tiles
            const int64_t p1 = ((t256 / cus) * cus) / nc256;
            if (p1 > 0 && p1 < p256) {
                const int64_t rem_px = M - p1 * 256;
                const double cs = (double)rounds(p1 * nc256, cus) * 65536.0 / r256 +
                                  (double)rounds((int64_t)mde_cdiv(rem_px, 128) * nc128, 2 * cus) * 32768.0;
                if (cs < best) { best = cs; pick = 3; split = (int32_t)(p1 * 256); }
            }
        }
    } else if (n >= 256) {
        if ((double)rounds(t256, cus) * 65536.0 / r256 < best) pick = 1;
    }
    if (!reg) {
        // the single-buffer 128x128 tile, four workgroups per CU: a partial last round costs it little (the workgroups left
        // share the CU among fewer), so its rounds count fractionally, with a floor of one (fitted in-network).
        // Against a 256-column candidate it must win by a margin: where the two models tie, the 256x256 split launch measured
        // 7 % faster (256 -> 256 channels, 25 taps, 153 600 pixels: 432 vs 462 us).
        static double rs = 0.0;
        if (rs == 0.0) {
            rs = 1.05;
            if (const char* e = getenv("MDE_CONV_RATE_S")) rs = atof(e);
        }
        const double fr = (double)t128 / (double)(4 * cus);
        const double cs = (fr > 1.0 ? fr : 1.0) * 65536.0 / rs;
        if (rs > 0.0 && cs < (pick != 0 ? 0.9 : 1.0) * best) { best = cs; pick = 4; }
    }
    if (forced == 0 && pick == 3) {
        KArgs k1 = ka, k2 = ka;
        k1.m_end = split;
        k2.m_begin = split;
        int rc = pp ? launch<256, 256, 512, true, 2, true>(k1, M, st) : launch<256, 256, 512, true, 2>(k1, M, st);
        if (rc) return rc;
        return launch<128, 128, 256, true, 2>(k2, M, st);
    }
    if (forced == 0 && pick == 4) return launch<128, 128, 256, true, 1>(ka, M, st);
    if (forced == 0 && pick == 2) return pp ? launch<192, 256, 512, true, 2, true>(ka, M, st) : launch<192, 256, 512, true, 2>(ka, M, st);
    if (forced == 1 || (forced == 0 && pick == 1))
        return reg ? launch<256, 256, 512, false, 2>(ka, M, st)
                   : (pp ? launch<256, 256, 512, true, 2, true>(ka, M, st) : launch<256, 256, 512, true, 2>(ka, M, st));
    if (forced == 5) return pp ? launch<192, 256, 512, true, 2, true>(ka, M, st) : launch<192, 256, 512, true, 2>(ka, M, st);
    if (forced == 6) return launch<128, 64, 256, true, 2>(ka, M, st);
    if (forced == 7) return launch<256, 256, 256, true, 2>(ka, M, st);   // 4 waves x (128 px x 128 ch), one workgroup per CU
    if (forced == 8) return launch<128, 128, 256, true, 1>(ka, M, st);   // single staging buffer: four workgroups per CU
    if (forced == 9) return launch<128, 256, 512, true, 1>(ka, M, st);   // single staging buffer, 256 columns: two per CU
    return reg ? launch<128, 128, 256, false, 2>(ka, M, st) : launch<128, 128, 256, true, 2>(ka, M, st);
}

}  // namespace

static int conv_gemm_impl(const mde_conv_desc* d, const void* in, const void* w, void* out, float* stats, const float* bias,
                          const void* resid, int act, void* stream, const mde_bn_red* red = nullptr);

extern "C" int mde_conv_gemm(const mde_conv_desc* d, const void* in, const void* w, void* out,
                             float* stats, void* stream) {
    return conv_gemm_impl(d, in, w, out, stats, nullptr, nullptr, 0, stream);
}

extern "C" int mde_conv_gemm_act(const mde_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                                 const void* residual, int act, void* stream) {
    MDE_REQUIRE(d && !d->accumulate, "mde_conv_gemm_act: an accumulating launch has no fused epilogue");
    MDE_REQUIRE(act >= 0 && act <= 3, "mde_conv_gemm_act: act=%d (0 none, 1 ReLU, 2 ELU, 3 sigmoid)", act);
    MDE_REQUIRE(!residual || ((uintptr_t)residual % 16) == 0, "mde_conv_gemm_act: residual must be 16-byte aligned");
    return conv_gemm_impl(d, in, w, out, nullptr, bias, residual, act, stream);
}

extern "C" int mde_conv_gemm_bnred(const mde_conv_desc* d, const void* in, const void* w, void* out, const mde_bn_red* r, void* stream) {
    MDE_REQUIRE(d && r && r->x && r->save_mean && r->save_rstd && r->part, "mde_conv_gemm_bnred: null argument");
    MDE_REQUIRE((r->mask_scale == nullptr) == (r->mask_shift == nullptr), "mde_conv_gemm_bnred: mask_scale / mask_shift come in pairs");
    MDE_REQUIRE(!(r->mask_scale && r->relu_bits), "mde_conv_gemm_bnred: the ReLU mask is either recomputed or read from bits");
    MDE_REQUIRE(d->ncols % 8 == 0 && d->ld_out % 8 == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)r->x % 16) == 0,
                "mde_conv_gemm_bnred: ncols=%d and ld_out=%d must be multiples of 8, out and x 16-byte aligned", d->ncols, d->ld_out);
    MDE_REQUIRE(r->x_ld >= 0 && r->x_ld % 8 == 0 && (r->x_ld == 0 || r->x_ld >= d->ncols),
                "mde_conv_gemm_bnred: x_ld=%d must be 0 or a multiple of 8 that holds the %d columns", r->x_ld, d->ncols);
    MDE_REQUIRE(!r->x2 || (r->save_mean2 && r->save_rstd2 && r->part2 && !r->mask_scale && ((uintptr_t)r->x2 % 16) == 0 &&
                           r->x2_ld >= 0 && r->x2_ld % 8 == 0),
                "mde_conv_gemm_bnred: a join needs the second site's mean, 1 / std and partial sums, takes its mask from bits, and "
                "x2 16-byte aligned with x2_ld a multiple of 8");
    MDE_REQUIRE(!r->add || (!d->accumulate && (r->relu_bits || r->x2) && d->ld_out == d->ncols && ((uintptr_t)r->add % 16) == 0),
                "mde_conv_gemm_bnred: `add` goes with a launch that does not accumulate, takes its sums' mask from relu_bits (or is a "
                "join), and addresses dense rows");
    MDE_REQUIRE(!r->relu_bits || d->ld_out == d->ncols, "mde_conv_gemm_bnred: packed mask bits address dense rows (ld_out=%d, ncols=%d)",
                d->ld_out, d->ncols);
    return conv_gemm_impl(d, in, w, out, nullptr, nullptr, nullptr, 0, stream, r);
}

static int conv_gemm_impl(const mde_conv_desc* d, const void* in, const void* w, void* out, float* stats, const float* bias,
                          const void* resid, int act, void* stream, const mde_bn_red* red) {
    MDE_REQUIRE(d && in && w && out, "mde_conv_gemm: null argument");
    MDE_REQUIRE(d->C > 0 && d->C % 8 == 0, "mde_conv_gemm: C=%d must be a positive multiple of 8", d->C);
    MDE_REQUIRE(!d->grouped || (d->C == BK && d->ncols % BK == 0),
                "mde_conv_gemm: a grouped launch contracts %d channels per column tile and needs ncols %% %d == 0 (C=%d, ncols=%d)",
                BK, BK, d->C, d->ncols);
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
    ka.m_begin = 0;
    ka.m_end = (int32_t)M;
    {
        static int order = -1;             // MDE_CONV_ORDER=pix restores pixel-fastest tile order (A/B)
        if (order < 0) {
            const char* e = getenv("MDE_CONV_ORDER");
            order = !(e && !strcmp(e, "pix"));
        }
        ka.col_fastest = order;
    }
    ka.det = g_mde_det.on;
    ka.bias = bias;
    ka.resid = resid;
    ka.act = act;
    ka.red_x = red ? red->x : nullptr;
    ka.red_mu = red ? red->save_mean : nullptr;
    ka.red_rs = red ? red->save_rstd : nullptr;
    ka.red_msc = red ? red->mask_scale : nullptr;
    ka.red_msh = red ? red->mask_shift : nullptr;
    ka.red_bits = red ? red->relu_bits : nullptr;
    ka.red_part = red ? red->part : nullptr;
    ka.red_ldmul = !(red && red->x_ld) ? 1 : (red->x_ld % d->ld_out == 0) ? red->x_ld / d->ld_out : 0;
    ka.red_ldx = red ? red->x_ld : 0;
    ka.red_x2 = red ? red->x2 : nullptr;
    ka.red_mu2 = red ? red->save_mean2 : nullptr;
    ka.red_rs2 = red ? red->save_rstd2 : nullptr;
    ka.red_part2 = red ? red->part2 : nullptr;
    ka.add_src = red ? red->add : nullptr;
    ka.add_bits = red ? red->add_bits : nullptr;
    ka.red_ldmul2 = !(red && red->x2_ld) ? 1 : (red->x2_ld % d->ld_out == 0) ? red->x2_ld / d->ld_out : 0;
    ka.red_ldx2 = red ? red->x2_ld : 0;
    ka.h_tws = ka.h_th = ka.h_nty = ka.h_ntx = ka.h_hw = ka.h_rows = ka.h_dy0 = ka.h_dx0 = ka.h_step_y = ka.h_step_x = 0;
    ka.h_dil = 1;
    ka.vec_ok = (d->ld_out % 8 == 0) && (((uintptr_t)out % 16) == 0);
    return pick_and_launch(ka, M, reinterpret_cast<hipStream_t>(stream));
}
