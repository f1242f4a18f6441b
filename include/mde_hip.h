/* libmde_hip.so — C ABI of the MI355X (gfx950) monocular-depth training hot path.
 *
 * Drop-in boundary for xeTaiz/mono-depth-estimation's FCRN path.  The
 * reference has no native layer: every op below replaces an ATen/cuDNN call that the
 * reference reaches through torch.nn (file:line of the call site is cited per entry).
 *
 * Conventions
 *   - every function returns 0 on success or a negative MDE_E* code; the message is in
 *     mde_last_error() (thread-local).  No C++ exception crosses this boundary.
 *   - the CALLER owns every buffer (activations, workspaces, outputs).  Nothing here
 *     allocates or frees device memory, and nothing synchronises: kernels are enqueued
 *     on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *   - activations are NHWC ("pixel-major") bf16 unless stated; `ld` arguments are the
 *     element distance between consecutive pixels so channel slices of a wider tensor can
 *     be passed without copies.  Master weights / statistics / gradients are fp32.
 *   - thread-safe: no global mutable state besides the thread-local error string.
 */
#ifndef MDE_HIP_H
#define MDE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDE_OK 0
#define MDE_EINVAL (-1)    /* bad argument (shape, alignment, null pointer)           */
#define MDE_EHIP (-2)      /* a HIP runtime call / kernel launch failed               */
#define MDE_ENOTSUP (-3)   /* valid request this build has no kernel for              */

#define MDE_MAX_TAPS 32

const char* mde_last_error(void);
/* ABI version of this header (checked by the Python loader). */
int mde_abi_version(void);
/* The 16-bit storage type this build of the library uses for activations, their gradients and the GEMM weight shadows
 * (every `void*` tensor argument documented as bf16 below): 0 = bf16 (libmde_hip.so), 1 = IEEE fp16 (libmde_hip_f16.so, the same
 * sources compiled with -DMDE_ACT_F16: BASELINE configuration 5 / reference train.py:139-140 precision=16). */
int mde_act_dtype(void);
/* Number of CUs of the current device (used by hosts to size split-K). */
int mde_device_cu_count(int* out);

/* ------------------------------------------------------------------------------------
 * Deterministic mode (opt-in).  Every cross-workgroup floating-point sum of a training step — BatchNorm partial sums,
 * split-K weight gradients, bias gradients — is an atomic add whose order changes from run to run; a last-bit
 * difference flips a bf16 rounding somewhere and a deep net amplifies it (two identical train steps at 32 x 480 x 640
 * differ by ~0.26 relative L2 in the trunk gradients).  With the mode on, every such addend is split exactly into two
 * integers and added with 64-bit integer atomics, which are order-independent: the step becomes bit-reproducible.
 * gbase: the flat fp32 gradient buffer all weight / bias gradients of the step land in (n elements); scratch: caller-owned,
 * mde_det_scratch_bytes(n) bytes, ZEROED by the caller once.  mde_det_flush adds the integer sums into gbase (and
 * re-zeroes them): call it after the last backward kernel, before anything reads the gradients.  While the mode is on,
 * BatchNorm partial-sum buffers are interpreted as integers too (same size; they must be zero when the mode is switched).
 * Process-wide state (one process per GPU); set it from the thread that launches the kernels.
 * ---------------------------------------------------------------------------------- */
size_t mde_det_scratch_bytes(int64_t n);
int mde_set_deterministic(int on, float* gbase, void* scratch, int64_t n);
int mde_deterministic(void);
int mde_det_flush(void* stream);

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA (bf16 in, fp32 accumulate).
 * Replaces every nn.Conv2d on the FCRN path: torchvision Bottleneck convs (called at
 * reference network/FCRN.py:305,318-323), conv2 (FCRN.py:334), the UpProj 5x5/3x3 convs
 * (FCRN.py:180-188) incl. the zero-insertion Unpool (FCRN.py:31-44) folded in as four
 * output phases, and their autograd backward (dgrad / wgrad).
 *
 * One descriptor covers forward, stride-1 dgrad, strided dgrad (as output phases) and the
 * phase-decomposed up-projection, because all are "for every pixel of an output grid,
 * sum over taps t and channels c of   in[(gy*sy+dy[t], gx*sx+dx[t]), c] * w[col][wtap[t]][c]".
 * ---------------------------------------------------------------------------------- */
typedef struct mde_conv_desc {
    /* gathered operand: bf16 [N][H][W][ld_in], C contracted channels per tap (C % 8 == 0; a K-step that runs past
     * C is zero-filled by the kernel, so DenseNet's 48-channel growth or a 32-channel head need no padded storage) */
    int32_t N, H, W, ld_in, C;
    uint32_t in_bytes;          /* bytes addressable from `in` (bounds for zero-fill)       */
    /* output pixel grid of this launch: M = N*GH*GW rows */
    int32_t GH, GW;
    int32_t sy, sx;             /* input step per grid step                                */
    int32_t ntaps;              /* 1..MDE_MAX_TAPS                                         */
    int16_t dy[MDE_MAX_TAPS];   /* input row  = gy*sy + dy[t]  (out of range -> zero)      */
    int16_t dx[MDE_MAX_TAPS];   /* input col  = gx*sx + dx[t]                              */
    int16_t wtap[MDE_MAX_TAPS]; /* tap slot of t inside the packed weight row              */
    int32_t wtaps_total;        /* weight row = [wtaps_total][C] bf16                       */
    /* output tensor: bf16 [N][OH][OW][ld_out]; grid pixel -> (gy*osy+ooy, gx*osx+oox)     */
    int32_t OH, OW, ld_out;
    int32_t osy, osx, ooy, oox;
    int32_t ncols;              /* output channels (GEMM columns); weight rows             */
    int32_t accumulate;         /* !=0: out += result (bf16 read-modify-write)             */
    /* !=0: grouped convolution in block-diagonal form (ResNeXt: VNL.py:638, MiDaS' resnext101_32x8d).  C must be 64
     * and ncols a multiple of 64: output columns [64b, 64b+64) contract with input channels [64b, 64b+64) of `in`
     * only; w is [ncols][wtaps_total][64] with zeros outside each column's own group (mde_pack_grouped writes it). */
    int32_t grouped;
} mde_conv_desc;

/* out[pix][col] (+)= sum_t sum_c in[src(pix,t)][c] * w[col][wtap[t]][c]
 * w: bf16 [ncols][wtaps_total][C].  stats (optional, may be NULL): a BatchNorm partial-sum
 * buffer fp32 [mde_stat_slots()][2][ncols] (see the BatchNorm section) that receives the
 * per-column sum and sum of squares of the fp32 results — fuses the BN batch-statistics pass
 * into the conv epilogue; feed it to mde_bn_finalize as `part`. */
int mde_conv_gemm(const mde_conv_desc* d, const void* in, const void* w, void* out,
                  float* stats, void* stream);
/* The same launch with a fused epilogue -- a conv bias (Conv2d(bias=True): MiDaS.py:163-229, VNL.py:331-350, Dorn.py:58-80), an
 * activation (act: 0 none, 1 ReLU, 2 ELU (Bts.py:69-80), 3 sigmoid) and a residual sum (ResidualConvUnit, FTB_block) without
 * the extra pass over the conv's output.  The bias joins the fp32 accumulator, so the sum is rounded to bf16 ONCE:
 * out = bf16(act(result + bias)) without a residual, out = bf16(act(bf16(result + bias) + residual)) with one.  (A rounding
 * of the bare result before a per-channel constant is added and the sum rounded again has an error that depends on the
 * constant only, the same for every pixel of the channel: a mean over pixels keeps it.)  bias: fp32 [ncols] or NULL;
 * residual: bf16, laid out and addressed exactly like `out` (same N, OH, OW, ld_out), or NULL; d->accumulate must be 0. */
int mde_conv_gemm_act(const mde_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                      const void* residual, int act, void* stream);
/* An input-gradient launch whose output is the gradient g of a BatchNorm (+ ReLU) site's OUTPUT (torchvision Bottleneck:
 * conv3's input gradient is d(relu(bn2(.))), FCRN.py:320-323): the epilogue also adds that site's backward sums
 * sum(g') and sum(g' * xhat) -- g' = g under the ReLU mask, xhat = (x - save_mean) * save_rstd -- into `part`, so that
 * mde_bn_bwd_reduce's pass over g and x is not made; continue with mde_bn_bwd_finalize / mde_bn_bwd_apply as after
 * mde_bn_bwd_reduce.  With d->accumulate the sums are of the accumulated gradient: the launch must be its last writer.
 * x: the site's input, bf16, the same pixels as `out`; x_ld its row stride in elements (0: ld_out; the upper half of the
 * up-projection's two-branch tensor, FCRN.py:181-188; a prefix of a DenseNet block's concatenation, Bts.py:283-292).  ReLU mask: mask_scale / mask_shift (recomputed as
 * x * scale + shift > 0), or relu_bits (one byte per 8 channels, dense rows: ld_out == ncols), or neither (no ReLU).
 * Several launches that together cover the gradient (the output phases of a strided conv) each add their part. */
typedef struct mde_bn_red {
    const void* x;
    const float* save_mean;
    const float* save_rstd;
    const float* mask_scale;
    const float* mask_shift;
    const uint8_t* relu_bits;
    float* part;               /* fp32 [mde_stat_slots()][2][ncols] */
    int32_t x_ld;
    /* a residual join out = relu(bn(x) + bn2(x2)) (Bottleneck with a projection shortcut; the up-projection's two branches,
     * FCRN.py:190-197): the gradient reaches both sites under the same mask (relu_bits, or none); the launch adds the sums of
     * both, as mde_bn_bwd_reduce2 does.  x2 == NULL: one site. */
    const void* x2;
    const float* save_mean2;
    const float* save_rstd2;
    float* part2;
    int32_t x2_ld;
    /* out = result + add under add_bits: the launch also brings in the gradient that reaches `out` through an identity shortcut
     * (torchvision Bottleneck: out += identity) -- `add` is the gradient of the block's OUTPUT, bf16, addressed exactly like
     * `out`, add_bits that output's ReLU mask (one byte per 8 channels; NULL: unmasked).  The BatchNorm-backward pass of the
     * block then need not write that masked copy into `out` first (mde_bn_bwd_apply's dres).  d->accumulate must be 0; only
     * with relu_bits or x2 (dense rows).  NULL: off. */
    const void* add;
    const uint8_t* add_bits;
} mde_bn_red;
int mde_conv_gemm_bnred(const mde_conv_desc* d, const void* in, const void* w, void* out, const mde_bn_red* r, void* stream);

/* Weight gradient ("TN" GEMM over pixels), fp32 output accumulated with atomics:
 *   dw[r][otap[t]][c] += sum_{pix in grid} direct[pix][.] x gathered[src(pix,t)][.]
 * rows index the channels of `rows_from` (0: direct tensor, 1: gathered tensor), columns the
 * other one.  The caller zeroes dw.  ksplit >= 1 splits the pixel range over workgroups. */
typedef struct mde_wgrad_desc {
    int32_t N, GH, GW;          /* pixel grid (of the direct tensor)                        */
    int32_t ld_d, Cd;           /* direct tensor bf16 [N][GH][GW][ld_d], Cd channels used (% 8 == 0) */
    int32_t H, W, ld_g, Cg;     /* gathered tensor bf16 [N][H][W][ld_g], Cg channels used   */
    uint32_t d_bytes, g_bytes;
    int32_t sy, sx, ntaps;
    int16_t dy[MDE_MAX_TAPS], dx[MDE_MAX_TAPS];
    int16_t otap[MDE_MAX_TAPS]; /* tap slot in the output                                   */
    int32_t otaps_total;        /* dw is fp32 [rows][otaps_total][cols]                      */
    int32_t rows_from_gathered; /* 0: rows = direct channels, cols = gathered channels       */
    int32_t ksplit;
    /* > 0: grouped convolution with this many channels per group (divides 64; Cd == Cg, multiples of 64): only the
     * block-diagonal part is computed and dw is fp32 [rows][otaps_total][group_size].  0: dense. */
    int32_t group_size;
} mde_wgrad_desc;

int mde_conv_wgrad(const mde_wgrad_desc* d, const void* direct, const void* gathered,
                   float* dw, void* stream);
/* The same with a workspace for a two-stage split-K reduction: every workgroup stores its partial tile (fp32
 * [slices][rows][ntaps][cols]) and a second kernel adds the slices into dw -- plain stores and one streaming pass instead of
 * slices x |dw| fp32 atomics (the L2 atomic units add one dword per clock and channel).  ws: fp32, 16-byte aligned, ws_bytes
 * >= mde_conv_wgrad_ws_bytes(d), otherwise (or ws == NULL, a grouped weight, deterministic mode) the atomic path runs;
 * mde_conv_wgrad_ws_bytes returns 0 for a launch that would keep the atomic path anyway (the measured rule in
 * conv_wgrad.hip: ws_choice), so a caller sizing its workspace from it reserves nothing for those.  The
 * second kernel updates dw with plain read-modify-writes: launches into the same dw, and launches that share a workspace,
 * must be ordered on one stream. */
int64_t mde_conv_wgrad_ws_bytes(const mde_wgrad_desc* d);
int mde_conv_wgrad_ws(const mde_wgrad_desc* d, const void* direct, const void* gathered, float* dw, float* ws, int64_t ws_bytes,
                      void* stream);

/* Stem 7x7/2 convolution on the raw image (torchvision conv1, used at FCRN.py:308,353).
 * x: fp32 NCHW [N][3][H][W] (the tensor the LightningModule hands to model(x), laina.py:18)
 * w: fp32 [64][7][7][3] (OHWI).  out: bf16 NHWC [N][H/2][W/2][64].  H, W even.
 * stats (optional, may be NULL): BatchNorm partial-sum buffer fp32 [mde_stat_slots()][2][64]
 * that receives the per-channel sum / sum of squares of the fp32 results (as mde_conv_gemm). */
int mde_stem_conv_fwd(const float* x, const float* w, void* out, float* stats, int N, int H, int W,
                      void* stream);
/* dw (fp32 [64][7][7][3], caller-zeroed) += wgrad from dout bf16 [N][H/2][W/2][64]. */
int mde_stem_conv_wgrad(const float* x, const void* dout, float* dw, int N, int H, int W, void* stream);
/* The same kernels for a stem of Cout = 64 or 96 output channels (densenet161's conv0, reference network/Bts.py:289 through
 * torchvision's densenet: nn.Conv2d(3, 96, 7, 2, 3)): w fp32 [Cout][7][7][3], out / dout bf16 [N][H/2][W/2][Cout], stats
 * [mde_stat_slots()][2][Cout], dw fp32 [Cout][7][7][3].  96 channels run as two launches of 48 (the weights of a launch
 * are register resident); the image is read once per launch. */
int mde_stem_conv_fwd_c(const float* x, const float* w, void* out, float* stats, int N, int H, int W, int Cout, void* stream);
int mde_stem_conv_wgrad_c(const float* x, const void* dout, float* dw, int N, int H, int W, int Cout, void* stream);

/* Head conv3 3x3, Cin -> Cout (Cout <= 32), fp32 output (FCRN.py:340,368).
 * x: bf16 [N][H][W][Cin]; w: fp32 [Cout][3][3][Cin]; out: fp32 [N][H][W][Cout]. */
int mde_head_conv_fwd(const void* x, const float* w, float* out, int N, int H, int W, int Cin,
                      int Cout, void* stream);
/* dx: bf16 [N][H][W][Cin];  dw: fp32 [Cout][3][3][Cin] caller-zeroed, atomically added. */
int mde_head_conv_bwd(const void* x, const float* w, const float* dout, void* dx, float* dw,
                      int N, int H, int W, int Cin, int Cout, void* stream);

/* ------------------------------------------------------------------------------------
 * BatchNorm2d, training and eval mode (nn.BatchNorm2d everywhere in FCRN.py / torchvision).
 * Statistics are fp32; a "BN site" normalises C channels (C % 8 == 0, C <= 2048; wider ones are split by the caller) of a
 * [M = N*H*W][ld] bf16 tensor.
 *
 * Partial-sum buffers ("part"): fp32 [mde_stat_slots()][2][C].  Producers (mde_bn_stats,
 * mde_conv_gemm's epilogue, mde_bn_bwd_reduce) ADD into slot (workgroup % slots) with fp32
 * atomics; the finalize kernels consume the buffer and leave it ZEROED, so a buffer that
 * starts zeroed (caller's job, once) can be reused every step without a memset.
 * ---------------------------------------------------------------------------------- */
int mde_stat_slots(void);
int mde_bn_stats(const void* x, int64_t M, int C, int ld, float* part, void* stream);
/* part -> mean, biased var -> scale = gamma*rstd, shift = beta - mean*scale;
 *   save_mean/save_rstd kept for backward; running stats updated with `momentum`
 *   (unbiased variance), exactly as nn.BatchNorm2d in train mode.  part is zeroed. */
int mde_bn_finalize(float* part, int64_t M, int C, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* scale,
                    float* shift, float* save_mean, float* save_rstd, void* stream);
/* The two halves of mde_bn_finalize, for BatchNorms that share batch moments (DenseNet: every later layer of a block
 * normalises the same concatenated channels with its own gamma / beta / running statistics; torchvision densenet161 as
 * Bts.py:283-292 uses it): mde_bn_moments turns a partial-sum buffer into mean and BIASED variance (fp32 [C]) and zeroes
 * it; mde_bn_finalize_moments derives one BatchNorm's scale / shift / saved statistics from such moments and updates its
 * running statistics (unbiased variance, `momentum`) exactly as mde_bn_finalize does. */
int mde_bn_moments(float* part, int64_t M, int C, float* mean, float* var, void* stream);
int mde_bn_finalize_moments(const float* mean, const float* var, int64_t M, int C, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                            float* save_mean, float* save_rstd, void* stream);
/* Eval mode: scale/shift from running statistics. */
int mde_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, int C, float* scale, float* shift,
                            void* stream);
/* out = act( x*scale + shift  [+ r  |  + r*rscale + rshift] );  relu == 1 applies ReLU.
 * relu == 2: out = ELU(x)*scale + shift [+ ...] -- x is the pre-activation of an ELU in front of the BatchNorm (the eval form
 * of BTS' upconv -> ELU -> bn, reference network/Bts.py:76-80,216-229): the ELU value is never stored in 16 bits.
 * r may be NULL; rscale/rshift may be NULL (plain residual add).  All bf16, own ld each.
 * relu_bits (optional, uint8 [M][C/8]): bit e of byte (row, c/8) = (out[row][c/8*8+e] > 0), the
 * packed ReLU mask the backward passes can read instead of `out` (16x fewer bytes). */
int mde_bn_apply(const void* x, int ldx, const float* scale, const float* shift, const void* r,
                 int ldr, const float* rscale, const float* rshift, void* out, int ldo,
                 uint8_t* relu_bits, int64_t M, int C, int relu, void* stream);
/* Backward of  out = act(bn(x) [+ ...]):  g = dout * (relu ? out > 0 : 1).
 * If mask_scale/mask_shift are given (sites WITHOUT a residual), the ReLU mask is recomputed
 * as (x*mask_scale + mask_shift > 0) from the tensor already being read and `out` is not
 * touched (may be NULL): one full-tensor read less per pass.  If relu_bits (from mde_bn_apply) is
 * given, the mask is read from it and `out` is likewise not touched.
 * pass 1: part += (sum g, sum g*xhat) per channel, xhat = (x - save_mean)*save_rstd. */
int mde_bn_bwd_reduce(const void* dout, int ldd, const void* out, int ldo, const void* x, int ldx,
                      const float* save_mean, const float* save_rstd, const float* mask_scale,
                      const float* mask_shift, const uint8_t* relu_bits, int64_t M, int C, int relu,
                      float* part, void* stream);
/* pass 2 (tiny): dgamma += sum g*xhat, dbeta += sum g; coef[3][C] = (gamma*rstd, mean g,
 * mean g*xhat); part is zeroed. */
int mde_bn_bwd_finalize(float* part, int64_t M, int C, const float* gamma, const float* save_rstd,
                        float* dgamma, float* dbeta, float* coef, void* stream);
/* pass 3: dx = coef0*(g - coef1 - xhat*coef2) (bf16, ld ldxo; accumulate_dx != 0 adds into dx);
 *   dres (optional) receives g, the masked upstream gradient, for the residual branch. */
int mde_bn_bwd_apply(const void* dout, int ldd, const void* out, int ldo, const void* x, int ldx,
                     const float* save_mean, const float* save_rstd, const float* mask_scale,
                     const float* mask_shift, const uint8_t* relu_bits, const float* coef, int64_t M,
                     int C, int relu, void* dx, int ldxo, int accumulate_dx, void* dres, int ldres,
                     void* stream);

/* ---- finalize work inside the streaming launches.  The one-workgroup finalize kernels above (mde_bn_finalize,
 * mde_bn_finalize_moments, mde_bn_bwd_finalize) sit between every pair of dependent BatchNorm passes: ~5 us with the chip idle
 * around each, 128 per FCRN step, ~540 per DenseNet-161 step.  In the *_fin forms every workgroup of the streaming pass derives
 * the constants of its own channels from the partial sums, workgroup 0 also writes what later passes read (scale / shift /
 * save_mean / save_rstd and the running statistics; dgamma / dbeta) and zeroes `zero[0 .. zero_n)` -- the partial-sum buffer of
 * the OTHER direction, which nothing reads while this launch runs.  The sums the launch reads are left as they are: a site
 * therefore keeps two buffers, forward statistics (zeroed by its backward launch) and backward sums (zeroed by its forward
 * launch); the caller zeroes them itself where a pass is repeated without its counterpart.  Not in deterministic mode
 * (integer partial sums): MDE_EINVAL there. */
typedef struct mde_bn_fin {
    const float* part;          /* forward sums [mde_stat_slots()][2][part_ld], already offset to the site's first channel, or NULL with */
    int32_t part_ld;
    const float* mean_in;       /* ... given batch moments [C] (mde_bn_moments: DenseNet's shared moments) */
    const float* var_in;
    int64_t count;              /* elements per channel (M) */
    const float* gamma;
    const float* beta;
    float* rmean;               /* running statistics, updated with `momentum` (both NULL: not tracked) */
    float* rvar;
    float momentum, eps;
    float* scale;               /* written by workgroup 0: gamma / std, beta - mean * scale, batch mean, 1 / std */
    float* shift;
    float* smean;
    float* srstd;
    float* zero;                /* NULL or a buffer workgroup 0 zeroes (the site's backward sums) */
    int64_t zero_n;
} mde_bn_fin;
typedef struct mde_bn_bfin {
    const float* part;          /* backward sums [mde_stat_slots()][2][part_ld]: sum(g'), sum(g' xhat) */
    int32_t part_ld;
    int64_t count;
    const float* gamma;
    const float* srstd;
    float* dgamma;              /* += by workgroup 0 (NULL: not wanted) */
    float* dbeta;
    float* zero;                /* NULL or a buffer workgroup 0 zeroes (the site's forward sums) */
    int64_t zero_n;
} mde_bn_bfin;
/* mde_bn_apply with the statistics' finalize inside: fin_r != NULL = a second BatchNorm on the residual (r must be given). */
int mde_bn_apply_fin(const void* x, int ldx, const mde_bn_fin* fin, const void* r, int ldr, const mde_bn_fin* fin_r, void* out, int ldo,
                     uint8_t* relu_bits, int64_t M, int C, int relu, void* stream);
/* mde_bn_bwd_apply / mde_bn_bwd_apply2 with mde_bn_bwd_finalize inside. */
int mde_bn_bwd_apply_fin(const void* dout, int ldd, const void* out, int ldo, const void* x, int ldx, const float* save_mean,
                         const float* save_rstd, const float* mask_scale, const float* mask_shift, const uint8_t* relu_bits,
                         const mde_bn_bfin* fin, int64_t M, int C, int relu, void* dx, int ldxo, int accumulate_dx, void* dres,
                         int ldres, void* stream);
int mde_bn_bwd_apply2_fin(const void* dout, int ldd, const void* xa, int ldxa, const void* xb, int ldxb, const float* save_mean_a,
                          const float* save_rstd_a, const float* save_mean_b, const float* save_rstd_b, const uint8_t* relu_bits,
                          const mde_bn_bfin* fin_a, const mde_bn_bfin* fin_b, int64_t M, int C, void* dxa, int ldda, void* dxb,
                          int lddb, void* stream);
/* The same two passes for a residual JOIN whose two summands are both BatchNorm outputs
 * (out = relu(bn_a(xa) + bn_b(xb)): the downsample bottlenecks and every up-projection, FCRN.py:170-198):
 * both sites see the same masked gradient g = dout * mask, so dout and the mask are read once per pass
 * instead of once per site.  relu_bits may be NULL (no ReLU).  part_a/part_b, coef_a/coef_b as above. */
int mde_bn_bwd_reduce2(const void* dout, int ldd, const void* xa, int ldxa, const void* xb, int ldxb,
                       const float* save_mean_a, const float* save_rstd_a, const float* save_mean_b,
                       const float* save_rstd_b, const uint8_t* relu_bits, int64_t M, int C,
                       float* part_a, float* part_b, void* stream);
int mde_bn_bwd_apply2(const void* dout, int ldd, const void* xa, int ldxa, const void* xb, int ldxb,
                      const float* save_mean_a, const float* save_rstd_a, const float* save_mean_b,
                      const float* save_rstd_b, const uint8_t* relu_bits, const float* coef_a,
                      const float* coef_b, int64_t M, int C, void* dxa, int ldda, void* dxb, int lddb,
                      void* stream);

/* ------------------------------------------------------------------------------------
 * Pooling / resize / pointwise.
 * ---------------------------------------------------------------------------------- */
/* MaxPool 3x3 stride 2 pad 1 (torchvision maxpool, FCRN.py:319,356). idx: uint8 argmax
 * (first maximum in window scan order, as ATen) kept for backward. */
int mde_maxpool_fwd(const void* x, void* out, uint8_t* idx, int N, int H, int W, int C, void* stream);
int mde_maxpool_bwd(const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C,
                    void* stream);
/* nn.PixelShuffle(2) on NHWC bf16 (FCRN.py:236,245): dst[n][2y+a][2x+b][c] = src[n][y][x][4c+2a+b]; src has 4C
 * channels at pixel stride ld_src, dst C channels at ld_dst (multiples of 8).  inverse != 0 writes src from dst
 * (the gradient of the forward permutation). */
int mde_pixel_shuffle2(void* src, int ld_src, void* dst, int ld_dst, int N, int h, int w, int C, int inverse,
                       void* stream);
/* Bilinear resize align_corners=True followed by sigmoid (FCRN.py:341,369-371).
 * x: fp32 [N][H][W][C] -> out fp32 NCHW [N][C][OH][OW] (the module's return tensor). */
int mde_upsample_sigmoid_fwd(const float* x, float* out, int N, int H, int W, int C, int OH, int OW,
                             void* stream);
/* dx fp32 [N][H][W][C] = d(loss)/dx given dout = d(loss)/d(out) and out (both NCHW fp32). */
int mde_upsample_sigmoid_bwd(const float* dout, const float* out, float* dx, int N, int H, int W,
                             int C, int OH, int OW, void* stream);

/* ---- Pointwise / pooling / resize ops of the VNL, MiDaS and BTS networks (NHWC bf16, own `ld` per tensor, C % 8 == 0) ----
 * act codes: 0 none, 1 ReLU, 2 ELU (alpha 1), 3 sigmoid.
 * out = act(x + bias[c] + r): conv bias + activation + residual add in one pass (MiDaS.py:163-229 ResidualConvUnit /
 * FeatureFusionBlock, VNL.py:348-349 FTB_block's `out += residual; relu`, Bts.py:69-80 conv + ELU).  bias (fp32 [C]) and r
 * may be NULL. */
int mde_pw_fwd(const void* x, int ldx, const float* bias, const void* r, int ldr, void* out, int ldo, int64_t M, int C,
               int act, void* stream);
/* g = dout * act'(out) (the derivative is taken from the forward OUTPUT; out may be NULL for act 0);
 * dx = g or dx += g (acc_x), likewise dr (the residual's gradient; may be NULL); dbias[c] += sum over rows of g (fp32;
 * may be NULL).  bias_part (optional): a zeroed BatchNorm-style partial-sum buffer [mde_stat_slots()][2][C] the workgroups
 * spread their bias-gradient atomics over (returned zeroed; one extra tiny launch folds it into dbias) — without it every
 * workgroup adds into dbias directly, which serialises on wide maps.  dx may be NULL when only dbias / dr are wanted. */
int mde_pw_bwd(const void* dout, int ldd, const void* out, int ldo, void* dx, int lddx, int acc_x, void* dr, int lddr,
               int acc_r, float* dbias, float* bias_part, int64_t M, int C, int act, void* stream);
/* out[n][c] = scale * sum_p x[n][p][c]: nn.AdaptiveAvgPool2d(1) with scale = 1/HW (VNL.py:207,221,359,367), and the
 * gradient of a spatial broadcast with scale = 1.  x: [N][HW][ldx]; out: bf16 [N][ldo]. */
int mde_spatial_sum(const void* x, int ldx, int N, int64_t HW, int C, float scale, void* out, int ldo, void* stream);
/* out[n][p][c] = scale * src[n][c] (accumulate != 0: +=): bilinear upsampling of a 1x1 map (VNL.py:225) and the gradient
 * of the average pool (scale = 1/HW). */
int mde_spatial_bcast(const void* src, int lds, float scale, void* out, int ldo, int N, int64_t HW, int C, int accumulate,
                      void* stream);
/* AFA_block's gate (VNL.py:372): out = w[n][c] * lat + top, w: bf16 [N][ldw].  Backward: dlat (+)= w * dout,
 * dtop (+)= dout, dw[n][c] = sum_p dout * lat (bf16 [N][lddw], overwritten). */
int mde_gate_fwd(const void* w, int ldw, const void* lat, int ldl, const void* top, int ldt, void* out, int ldo, int N,
                 int64_t HW, int C, void* stream);
int mde_gate_bwd(const void* dout, int ldd, const void* w, int ldw, const void* lat, int ldl, void* dlat, int lddl,
                 int acc_lat, void* dtop, int lddt, int acc_top, void* dw, int lddw, int N, int64_t HW, int C, void* stream);
/* F.interpolate(mode='bilinear') on NHWC bf16, ATen's source-index arithmetic for both align_corners settings
 * (VNL.py:308,384,386 True; MiDaS.py:155-157 False, :224-227 True).  bwd is the exact transpose in gather form
 * (deterministic); accumulate != 0 adds into dx. */
int mde_resize_bilinear_fwd(const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, int OH, int OW,
                            int align_corners, void* stream);
int mde_resize_bilinear_bwd(const void* dout, int ldd, void* dx, int lddx, int N, int H, int W, int C, int OH, int OW,
                            int align_corners, int accumulate, void* stream);
/* Nearest x2 upsampling (Bts.py:77): out[n][2y+a][2x+b] = x[n][y][x], x is [N][H][W].  mde_sum2x2 is its gradient
 * (scale 1) and nn.AvgPool2d(2, 2) forward (scale 0.25): dst[n][y][x] (+)= scale * sum of the 2x2 block of src
 * [N][2H][2W]; mde_spread2x2 is the average pool's gradient: out[n][2y+a][2x+b] (+)= scale * x[n][y][x]. */
int mde_nearest2_fwd(const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, void* stream);
int mde_sum2x2(const void* src, int lds, void* dst, int ldd, int N, int H, int W, int C, float scale, int accumulate,
               void* stream);
int mde_spread2x2(const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, float scale, int accumulate,
                  void* stream);
/* fcn_topdown_predict (VNL.py:314-327): x = the 3x3 conv's output bf16 [N*HW][ldx] (bias not yet added);
 * logit = x + bias, prob = softmax over channels; both fp32 NCHW [N][C][HW] (the module's two return tensors).  C <= 256.
 * bwd: dx = dlogit + prob * (dprob - sum_c dprob * prob) as bf16 [N*HW][lddx] (channels [C, C rounded up to 8) written
 * as zero); dbias[c] += sum_p dx (fp32, may be NULL).  dlogit or dprob may be NULL (treated as zero). */
int mde_softmax_head_fwd(const void* x, int ldx, const float* bias, float* logit, float* prob, int N, int64_t HW, int C,
                         void* stream);
int mde_softmax_head_bwd(const float* dlogit, const float* dprob, const float* prob, void* dx, int lddx, float* dbias, int N,
                         int64_t HW, int C, void* stream);
/* y = scale * act(p) on an fp32 map of n elements (n % 4 == 0), and its backward through the kept output:
 * dp = dy * scale * act'(.) with act' expressed through y / scale.  The activation behind a one-channel head convolution that
 * already produced fp32 (mde_head_conv_fwd) -- BTS' get_depth + nn.Sigmoid x max_depth (reference network/Bts.py:168,262).
 * act: 0 none, 1 ReLU, 2 ELU, 3 sigmoid, 4 ReLU6. */
int mde_map_act_fwd(const float* p, float* y, int64_t n, int act, float scale, void* stream);
int mde_map_act_bwd(const float* dy, const float* y, float* dp, int64_t n, int act, float scale, void* stream);

/* out[n][c][p] = scale * act(x[n][p][c] + bias[c]), fp32 NCHW: the small-C output heads (MiDaS.py:54-56: 7 channels +
 * sigmoid; Bts.py:202-203,273: sigmoid * max_depth).  bwd: dx bf16 [N*HW][lddx] (pad channels zero), dbias += (C <= 64). */
int mde_to_nchw_act_fwd(const void* x, int ldx, const float* bias, float* out, int N, int64_t HW, int C, int act, float scale,
                        void* stream);
int mde_to_nchw_act_bwd(const float* dout, const float* out, void* dx, int lddx, float* dbias, int N, int64_t HW, int C,
                        int act, float scale, void* stream);
/* BTS' image-residual head (Bts.py:264-271; out_channels == 10 with image_residuals): d = the ten sigmoid channels of get_depth
 * (two RGBA layers + two depths), fp32 [N][C][HW]; rgb = the input image fp32 [N][3][HW].  out[:, 0:3] = clamp(2 d - 1 + rgb, 0, 1),
 * out[:, 3] = clamp(2 d - 1 + mean(rgb), 0, 1), [4:8] likewise, [8:] = d.  Backward: dd = d(loss)/d(d) (no gradient to the image). */
int mde_image_residual_fwd(const float* d, const float* rgb, float* out, int N, int64_t HW, int C, void* stream);
int mde_image_residual_bwd(const float* dout, const float* d, const float* rgb, float* dd, int N, int64_t HW, int C, void* stream);
/* BTS plane heads (Bts.py:105-122 reduction_1x1's tail, :228-231 F.normalize, :124-146 local_planar_guidance): x bf16
 * [N][h][w][ldx] holds the three plane parameters in channels 0..2; theta = sigmoid(x0) pi/3, phi = sigmoid(x1) 2 pi,
 * dist = sigmoid(x2) max_depth, n = normalize(sin theta cos phi, sin theta sin phi, cos theta);
 * out[n][i*up + a][j*up + b] = dist / (n1 u_b + n2 v_a + n3) / max_depth with u, v = (k - (up-1)/2) / up; out is fp32
 * [N][h*up][w*up].  bwd writes dx bf16 [N][h][w][lddx] (channels 3..7 zero) from dout fp32 [N][h*up][w*up]. */
int mde_plane_depth_fwd(const void* x, int ldx, float* out, int N, int h, int w, int up, float max_depth, void* stream);
int mde_plane_depth_bwd(const void* x, int ldx, const float* dout, void* dx, int lddx, int N, int h, int w, int up, float max_depth,
                        void* stream);
/* A one-channel fp32 map [N][H][W] as one bf16 channel of an NHWC tensor at 1/step resolution (dst points at the channel,
 * ld = its pixel stride): torch.cat([...], dim=1) of a depth map, after F.interpolate(scale_factor=1/step, mode='nearest')
 * when step > 1 (Bts.py:234,238,247,251,263).  mde_slot_to_map_add is the gradient: dsrc[picked pixels] += dslot. */
int mde_map_to_slot(const float* src, void* dst, int ld, int N, int H, int W, int step, void* stream);
int mde_slot_to_map_add(const void* dslot, int ld, float* dsrc, int N, int H, int W, int step, void* stream);
/* Block-diagonal packings of a grouped conv weight (VNL.py:638 `groups=cardinality`): src fp32 [O][T][G] (G = channels per
 * group, O == I, O % 64 == 0, G divides 64) -> fwd bf16 [O][T][64] for mde_conv_gemm(grouped) and dgrad bf16 [O][T][64]
 * (the transposed blocks, for the input gradient); either may be NULL. */
int mde_pack_grouped(const float* src, void* fwd, void* dgrad, int O, int T, int G, void* stream);
/* The two-term eval shadow (mde_pack_split_batch) of a grouped weight: fwd2 bf16 [O][2T][64]. */
int mde_pack_grouped_split(const float* src, void* fwd2, int O, int T, int G, void* stream);

/* Depthwise 3x3 convolution, groups == channels (the reference's MobileNetV2 encoder option: VNL.py:427-444, nn.Conv2d(h, h, 3, stride,
 * groups=h, padding=dilation, dilation=dilation, bias=False)): NHWC 16-bit activations, fp32 weights w [C][9] (the flat master
 * slice itself) and fp32 weight gradient dw [C][9] (+=).  C % 8 == 0, stride 1 or 2, padding == dilation, so the output is
 * ((H - 1) / stride + 1) x ((W - 1) / stride + 1).  HBM-bound streaming passes (nine multiply-adds per element): nothing for MFMA.
 *   fwd  : out[n, oy, ox, c] = sum_ij x[n, oy s + (i - 1) d, ox s + (j - 1) d, c] w[c][3 i + j]
 *   dgrad: dx over the INPUT's H x W from dy over the output's grid; accumulate != 0: dx += .
 *   wgrad: dw[c][3 i + j] += sum over output pixels of x[...] dy[...]. */
int mde_dwconv3x3_fwd(const void* x, int ldx, const float* w, void* out, int ldo, int N, int H, int W, int C, int stride, int dilation,
                      void* stream);
int mde_dwconv3x3_dgrad(const void* dy, int ldy, const float* w, void* dx, int lddx, int N, int H, int W, int C, int stride, int dilation,
                        int accumulate, void* stream);
int mde_dwconv3x3_wgrad(const void* x, int ldx, const void* dy, int ldy, float* dw, int N, int H, int W, int C, int stride, int dilation,
                        void* stream);

/* ---- DORN pieces (network/Dorn.py) ---- */
/* nn.MaxPool2d(3, 2, 1, ceil_mode=True) (Dorn.py:235): as mde_maxpool_fwd / _bwd with the output size of ATen's ceil rule
 * (OH = ceil((H - 1) / 2) + 1, minus one if the last window would start beyond the padded input); out / idx are
 * [N][OH][OW][C].  ceil_mode == 0 is exactly mde_maxpool_fwd / _bwd. */
/* nn.MaxPool2d(k, s) without padding over a spatial VIEW of an NHWC tensor: VGG-19-BN's MaxPool2d(2, 2) (Eigen.py:74) and the
 * cropped pools of Eigen's scale 2 / 3 (`pool(x)[:, :, 1:-1, 1:-1]`, Eigen.py:23,41; `conv(img)[:, :, 2:-3, 2:-3]` then
 * MaxPool2d(3, 1), Eigen.py:52,65-67) -- the crop is the view.  x / dx point at the view's first pixel inside the producing
 * tensor; ld = its channel stride, wpitch / ipitch = its pixels per row / per image, Hv x Wv = the view.  out: bf16
 * [N][OH][OW][ldo >= C] with OH = (Hv - k) / s + 1; idx: uint8 [N][OH][OW][C] (argmax inside the window, ATen's tie rule).
 * The backward pass writes (accumulate != 0: adds onto) the VIEW of dx only: the caller zeroes what a crop leaves out. */
int mde_maxpool_view_fwd(const void* x, int ldx, int wpitch, int64_t ipitch, int Hv, int Wv, void* out, int ldo, uint8_t* idx,
                         int N, int C, int k, int s, void* stream);
int mde_maxpool_view_bwd(const void* dout, int ldd, const uint8_t* idx, void* dx, int lddx, int wpitch, int64_t ipitch, int Hv, int Wv,
                         int N, int C, int k, int s, int accumulate, void* stream);
int mde_maxpool_fwd2(const void* x, void* out, uint8_t* idx, int N, int H, int W, int C, int ceil_mode, void* stream);
int mde_maxpool_bwd2(const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C, int ceil_mode, void* stream);
/* nn.Dropout2d as data (Dorn.py:59,107,109): out[n][p][c] = x[n][p][c] * m[n][c] (+ out when accumulate), m fp32 [N][C] holding
 * 0 or 1 / (1 - p); bf16 [N * HW][ld] tensors, C % 8 == 0.  The gradient is the same call on the output gradient. */
int mde_chan_scale(const void* x, int ldx, const float* m, void* out, int ldo, int N, int64_t HW, int C, int accumulate, void* stream);
/* FullImageEncoder's pooling (Dorn.py:58,72-74): nn.AvgPool2d(k, stride=s, padding=p) (count_include_pad, floor mode) of a
 * bf16 NHWC map, times m[n][c] (the Dropout2d scale, may be NULL), written in the order of `x.view(-1, C * OH * OW)` of the
 * NCHW result: out bf16 [N][C * OH * OW], OH = (H + 2p - k) / s + 1 — what nn.Linear's weight contracts as stored.
 * bwd: dx[n][y][x][c] (+)= m[n][c] / k^2 * sum of dout over the windows containing (y, x). */
int mde_avgpool_flat_fwd(const void* x, int ldx, const float* m, void* out, int N, int H, int W, int C, int k, int s, int p, void* stream);
int mde_avgpool_flat_bwd(const void* dout, const float* m, void* dx, int lddx, int N, int H, int W, int C, int k, int s, int p,
                         int accumulate, void* stream);
/* OrdinalRegressionLayer (Dorn.py:288-318): x bf16 [N * HW][ldx] with channel 2k = A_k, 2k + 1 = B_k (K <= 128);
 * prob fp32 [N][K][HW] = softmax over the pair (clamp(A_k, 1e-8, 1e4), clamp(B_k, 1e-8, 1e4)) at index 1;
 * label int64 [N][HW] = #{k : prob > 0.5}.  bwd: dx bf16 [N * HW][lddx] (every channel written, padding zero):
 * dB_k = dprob * prob * (1 - prob) where B_k lies inside the clamp range, dA_k = -(the same) where A_k does. */
int mde_ordinal_fwd(const void* x, int ldx, float* prob, int64_t* label, int N, int64_t HW, int K, void* stream);
int mde_ordinal_bwd(const float* dprob, const void* x, int ldx, void* dx, int lddx, int N, int64_t HW, int K, void* stream);

/* ---- MyNet pieces (network/MyNet.py) ---- */
/* Weighter's tail (MyNet.py:96-119): flatten(start_dim=2) -> nn.Linear(HW, 1) over the pixel axis -> sum over channels ->
 * sigmoid, of a bf16 [N][HW][lda] map with C channels: scale[n] = sigmoid(sum_p w[p] * sum_c a[n][p][c] + C * b[0]).
 * pre: N floats of scratch (the pre-activation).  bwd: da (+)= dpre[n] * w[p] with dpre = dscale * s * (1 - s),
 * dw[p] += sum_n dpre[n] * sum_c a, db[0] += C * sum_n dpre[n]. */
int mde_weighted_pool_fwd(const void* a, int lda, const float* w, const float* b, float* pre, float* scale, int N, int64_t HW, int C,
                          void* stream);
int mde_weighted_pool_bwd(const float* dscale, const float* scale, const void* a, int lda, const float* w, void* da, int ldda,
                          int accumulate, float* dw, float* db, int N, int64_t HW, int C, void* stream);
/* my_decoder's output (MyNet.py:152-155): out[n][p] = factor * (m0 * s0[n] + m1 * s1[n] + m2 * s2[n]) over fp32 maps [N][HW]
 * and per-image scales [N].  bwd: dm_k += factor * dout * s_k[n] (accumulating), ds [3][N] = factor * sum_p dout * m_k. */
int mde_combine3_fwd(const float* m0, const float* m1, const float* m2, const float* s0, const float* s1, const float* s2, float factor,
                     int N, int64_t HW, float* out, void* stream);
int mde_combine3_bwd(const float* dout, const float* m0, const float* m1, const float* m2, const float* s0, const float* s1,
                     const float* s2, float factor, int N, int64_t HW, float* dm0, float* dm1, float* dm2, float* ds, void* stream);

/* ---- Input pipeline (modules/base_module.py:234-284), Pillow's 8-bit arithmetic bit for bit; images are uint8 H x W x C ---- */
/* functional.to_pil_image of a float tensor C x H x W: (v / divisor) * 255 truncated to uint8 (divisor: the `depth / s` in front,
 * 1 for the image). */
int mde_aug_to_u8(const float* src, int C, int H, int W, float divisor, uint8_t* dst, void* stream);
/* PIL Image.resize(BILINEAR) of an 8-bit image (Resample.c): the horizontal pass over source rows [y0, y0 + rows) into tmp
 * (only the rows the vertical pass reads), then the vertical pass; each rounds to uint8.  bounds: int32 [out][2] = (first
 * input index, count), k: int32 [out][ksize] 22-bit fixed-point coefficients (the caller computes them in double as Pillow's
 * precompute_coeffs does); an axis whose size does not change passes NULL tables. */
int mde_aug_resample_u8(const uint8_t* src, int H, int W, int C, const int32_t* hbounds, const int32_t* hk, int hksize, int OW,
                        const int32_t* vbounds, const int32_t* vk, int vksize, int OH, int y0, int rows, uint8_t* tmp, uint8_t* dst,
                        void* stream);
/* PIL Image.transform(AFFINE, NEAREST) as Image.rotate uses it (Geometry.c affine_fixed): coef = the six 16.16 fixed-point
 * coefficients (a0, a1, a2 + half-pixel terms, a3, a4, a5 + ...); pixels that map outside the image are 0. */
int mde_aug_affine_nearest_u8(const uint8_t* src, int H, int W, int C, const int32_t* coef, uint8_t* dst, void* stream);
/* CenterCrop -> hflip -> np.array(img, float32) / 255 -> to_tensor: uint8 H x W x C -> float32 C x oh x ow; lut: the 256
 * quotients (computed by the host exactly as numpy does). */
int mde_aug_crop_flip_to_float(const uint8_t* src, int H, int W, int C, int top, int left, int oh, int ow, int flip, const float* lut,
                               float* dst, void* stream);
/* The same with one 256-entry table PER CHANNEL (lut[c * lut_stride + v]; lut_stride 0 = one shared table): the last step of
 * modules/midas.py:107-150 is the hub's `default_transform`, which on an image that already has its 384 x 384 size is
 * float32((v / 255.0 - mean[c]) / std[c]) evaluated in double -- 3 x 256 values the host computes as numpy does. */
int mde_aug_crop_flip_to_float_c(const uint8_t* src, int H, int W, int C, int top, int left, int oh, int ow, int flip, const float* lut,
                                 int lut_stride, float* dst, void* stream);
/* modules/vnl.py:59-78 (flip_pad_reshape_crop up to its cv2.resize): np.flip(img, axis=1) -> np.pad(img, ((pad_top, 0),
 * (pad_left, 0)), 'constant', fill) -> img[crop_y : crop_y + oh, crop_x : crop_x + ow] for an H x W x C image of 1-byte
 * (uint8 RGB) or 4-byte (float32 depth) elements; `fill`: one HOST element.  The crop must lie inside the padded image. */
int mde_aug_flip_pad_crop(const void* src, int elem_bytes, int H, int W, int C, int flip, int pad_top, int pad_left, int crop_y, int crop_x,
                          int oh, int ow, const void* fill, void* dst, void* stream);

/* ------------------------------------------------------------------------------------
 * Losses and metrics (criteria.py / metrics.py), fp32 in, fp32/fp64 accumulation.
 * ---------------------------------------------------------------------------------- */
/* ordLoss (criteria.py:734-787): prob fp32 [N][K][HW], target fp32 [N][HW] (the SID label, NOT truncated: the reference
 * compares the plane index k with it as floats; NaN takes neither branch):
 * loss = -(sum_{k <= t} log clamp(P_k, 1e-8, 1e8) + sum_{k > t} log clamp(1 - P_k, 1e-8, 1e8)) / (N * HW).
 * ws >= mde_ord_loss_ws_bytes().  bwd: grad fp32 [N][K][HW] = *gscale * d loss / d prob (zero where a clamp is active). */
size_t mde_ord_loss_ws_bytes(void);
int mde_ord_loss_fwd(const float* prob, const float* target, int N, int K, int64_t HW, void* ws, float* loss, void* stream);
int mde_ord_loss_bwd(const float* prob, const float* target, int N, int K, int64_t HW, const float* gscale, float* grad, void* stream);
/* SILog (criteria.py:724-732).  ws: >= mde_silog_ws_bytes() bytes, zeroed by the call.
 * loss: 1 float.  grad (optional): d loss / d est, same shape as est, scaled by *gscale
 * (device pointer to 1 float = upstream gradient; NULL -> 1). */
size_t mde_silog_ws_bytes(void);
int mde_silog_fwd(const float* est, const float* gt, int64_t n, float variance_focus, void* ws,
                  float* loss, void* stream);
int mde_silog_bwd(const float* est, const float* gt, int64_t n, float variance_focus, const void* ws,
                  const float* gscale, float* grad, void* stream);
/* Masked pointwise losses (criteria.py:67-90 MaskedMSELoss / MaskedL1Loss, :113-133 berHuLoss), mask = target > 0.
 * kind 0: mean |t-p|; 1: mean (t-p)^2; 2: reverse Huber exactly as the reference computes it (threshold
 * c = 0.2*max(pred-target) over ALL pixels, loss = mean(cat(|d|, |d|[|d|>c]^2))).  An empty mask gives NaN, as there.
 * ws >= mde_masked_loss_ws_bytes(), written by fwd and read by bwd; grad = d loss / d pred * (*gscale or 1). */
size_t mde_masked_loss_ws_bytes(void);
int mde_masked_loss_fwd(int kind, const float* pred, const float* target, int64_t n, void* ws, float* loss,
                        void* stream);
int mde_masked_loss_bwd(int kind, const float* pred, const float* target, int64_t n, const void* ws,
                        const float* gscale, float* grad, void* stream);
/* MaskedDepthLoss (criteria.py:17-64, the loss modules/eigen.py pairs with Eigen): linear-space scale-invariant
 * term over the per-image masked residuals + masked forward-difference gradient MSE in y and x.
 * pred/target: fp32 [N][H][W] (or [N][1][H][W]).  ws >= mde_masked_depth_ws_bytes(N). */
size_t mde_masked_depth_ws_bytes(int N);
int mde_masked_depth_fwd(const float* pred, const float* target, int N, int H, int W, void* ws, float* loss,
                         void* stream);
int mde_masked_depth_bwd(const float* pred, const float* target, int N, int H, int W, const void* ws,
                         const float* gscale, float* grad, void* stream);
/* MiDaS losses (criteria.py:154-332), pred/target fp32 [N][H][W], mask = target > 0:
 *   q = scale_b*pred + shift_b when ssi (compute_scale_and_shift, per-image closed-form least squares; det == 0 -> 0, 0)
 *   loss = data_weight * data(q) + alpha * sum_{k < scales} gradient_loss(q[::2^k, ::2^k])
 *          (MidasLoss.forward: data_weight 1; GradientLoss alone: data_weight 0, alpha 1)
 *   data_kind 0: mse_loss, 1: l1_loss == trimmed_mae_loss (the reference's trimming is a no-op), both
 *   reduction_batch_based on 2*M; the gradient term is batch-based or image-based (batch_based = 0).
 * mde_midas_bwd: grad = d loss / d pred * (*gscale or 1), including the dependence of scale and shift on pred.
 * ws >= mde_midas_ws_bytes(N): zeroed and written by fwd, read (and completed) by bwd. */
size_t mde_midas_ws_bytes(int N);
int mde_midas_fwd(const float* pred, const float* target, int N, int H, int W, int ssi, int data_kind,
                  float data_weight, float alpha, int scales, int batch_based, void* ws, float* loss,
                  void* stream);
int mde_midas_bwd(const float* pred, const float* target, int N, int H, int W, int ssi, int data_kind,
                  int scales, void* ws, const float* gscale, float* grad, void* stream);
/* TrimmedProcrustesLoss (criteria.py:335-363): prediction and target are robustly normalised per image
 * (normalize_prediction_robust, criteria.py:135-152: subtract the lower median of mask*x over all H*W values, divide
 * by the mean absolute deviation over valid pixels clamped at 1e-6), then the reference's "trimmed" MAE (it never
 * trims) + alpha * multi-scale gradient loss, all masks = target > 0 of the original target.
 * pred_n / target_n: caller-owned fp32 [N][H][W] scratch that fwd fills (the normalised maps) and bwd reads;
 * gtmp: another such scratch for bwd.  ws >= mde_procrustes_ws_bytes(N). */
size_t mde_procrustes_ws_bytes(int N);
int mde_procrustes_fwd(const float* pred, const float* target, int N, int H, int W, float alpha, int scales,
                       int batch_based, void* ws, float* pred_n, float* target_n, float* loss, void* stream);
int mde_procrustes_bwd(const float* pred, const float* target, int N, int H, int W, int scales, void* ws,
                       const float* pred_n, const float* target_n, const float* gscale, float* gtmp,
                       float* grad, void* stream);
/* compute_scale_and_shift alone (criteria.py:154-176): scale[N], shift[N]. */
int mde_scale_and_shift(const float* pred, const float* target, int N, int H, int W, void* ws, float* scale,
                        float* shift, void* stream);
/* ---- The VNL configuration's criteria (reference criteria.py:839-1062, modules/vnl.py:202-230) ----
 * WCEL_Loss (criteria.py:839-863): logit [N][C][HW] fp32 (NCHW), bins [N][HW] int32 labels (a label outside
 * [0, C) contributes nothing, like the reference's one-hot comparison), gt [N][HW] depth (only gt > 0 is
 * counted, for the divisor), weight [C][C] = the matrix AFTER the constructor's row normalisation.
 * loss = -sum_px sum_c weight[bin][c] * log_softmax(logit)[c] / #{gt > 0}.  lse: caller-owned [N][HW] scratch
 * fwd fills (log-sum-exp per pixel) and bwd reads.  ws >= mde_wcel_ws_bytes(C). */
size_t mde_wcel_ws_bytes(int C);
int mde_wcel_fwd(const float* logit, const int32_t* bins, const float* gt, const float* weight, int N, int C,
                 int64_t HW, void* ws, float* lse, float* loss, void* stream);
int mde_wcel_bwd(const float* logit, const int32_t* bins, const float* weight, int N, int C, int64_t HW,
                 const void* ws, const float* lse, const float* gscale, float* grad, void* stream);
/* bins_to_depth (modules/vnl.py:219-230): depth[n][p] = 10 ** sum_c prob[n][c][p] * border[c]; bwd fills
 * gprob [N][C][HW] from gdepth [N][HW] and the depth fwd produced. */
int mde_bins_to_depth_fwd(const float* prob, const float* border, int N, int C, int64_t HW, float* depth,
                          void* stream);
int mde_bins_to_depth_bwd(const float* depth, const float* gdepth, const float* border, int N, int C,
                          int64_t HW, float* gprob, void* stream);
/* depth_to_bins (modules/vnl.py:202-217): bins = trunc((log10(clamp(depth)) - depth_min_log) / interval),
 * C + 1 where depth < 0, C -> C - 1; depth is rewritten IN PLACE (clamped; -1 where it was negative), as the
 * reference does. */
int mde_depth_to_bins(float* depth, int64_t n, float depth_min, float depth_max, float depth_min_log,
                      float interval, int C, int32_t* bins, void* stream);
/* VNL_Loss (criteria.py:866-1045).  gt, pred: [B][H][W] fp32 depth.  p123: DEVICE int32 [3][n] linear pixel
 * indices (y * W + x) of the three points of each of the n triples, drawn by the host exactly as
 * criteria.py:912-932 does and shared by every image of the batch; a triple with an index outside [0, H*W)
 * is rejected.  Back-projection with principal point (W/2, H/2) (integer division), the reference's
 * collinear / near / invalid-depth filter (thresholds 0.867, 0.005, 1e-4), |n_gt - n_pred|_1 of the unit
 * normals; select != 0 drops the int(0.25 * M) smallest of the M surviving values before the mean
 * (M == 0 -> NaN, like the reference).  bwd zero-fills grad [B][H][W] and scatters with fp32 atomics.
 * ws >= mde_vnl_ws_bytes(B, n), carried from fwd to bwd. */
size_t mde_vnl_ws_bytes(int B, int n);
int mde_vnl_fwd(const float* gt, const float* pred, const int32_t* p123, int B, int H, int W, int n, float fx,
                float fy, int select, void* ws, float* loss, void* stream);
int mde_vnl_bwd(const float* gt, const float* pred, const int32_t* p123, int B, int H, int W, int n, float fx,
                float fy, const void* ws, const float* gscale, float* grad, void* stream);

/* ---- criterion-fused softmax head (a PRIVATE route between VNL's head and its criterion; the public tensors stay what they are).
 * network/VNL.py:325-327 returns fp32 NCHW logits and softmax (2 x 2.95 GB at 16 x 150 x 480 x 640) and the reference's criterion
 * -- ModelLoss(bins_to_depth(softmax), logits, depth_to_bins(gt), gt), modules/vnl.py:255 -- walks them again forwards and
 * backwards.  When that criterion is handed tensors which come straight from this library's head, it reads the head's INPUT
 * instead (x: 16-bit [P][ldx], P = N*H*W pixels, the prediction conv's output; logits = x + bias, C <= 192 channels) and the
 * backward writes d(x) directly: 20 GB less HBM traffic per configuration-5 step.
 *   mde_vnl_head_depth_fwd: depth[p] = 10 ** sum_c softmax_c border_c (modules/vnl.py:219-230), log10_depth[p], lse[p] = logsumexp.
 *   mde_vnl_head_wcel_fwd : criteria.WCEL_Loss (criteria.py:839-863) from x and the lse of the call above; ws >= mde_wcel_ws_bytes(C).
 *   mde_vnl_head_bwd      : dx[p][c] = gscale/valid * -(w[bin][c] - softmax_c rowsum[bin])            (bins / weight / ws given)
 *                                    + softmax_c * gdepth[p] depth[p] ln10 * (border_c - log10_depth[p])   (gdepth given)
 *                           as 16-bit [P][lddx] (channels [C, lddx) zero).  (The bias gradient is the column sums of dx: mde_bn_stats.)
 * x and dx rows are walked in 16-byte chunks: ldx, lddx multiples of 8, 16-byte aligned. */
int mde_vnl_head_depth_fwd(const void* x, int ldx, const float* bias, const float* border, int64_t P, int C, float* depth,
                           float* log10_depth, float* lse, void* stream);
int mde_vnl_head_wcel_fwd(const void* x, int ldx, const float* bias, const int32_t* bins, const float* gt, const float* weight,
                          const float* lse, int64_t P, int C, void* ws, float* loss, void* stream);
int mde_vnl_head_bwd(const void* x, int ldx, const float* bias, const int32_t* bins, const float* weight, const void* ws,
                     const float* gscale, const float* lse, const float* depth, const float* log10_depth, const float* gdepth,
                     const float* border, int64_t P, int C, void* dx, int lddx, void* stream);
/* ---- The stdepth composite criterion (reference modules/base_module.py:124-208 `_loss`, stdepth_utils.py) ----
 * pred, targ: [N][C][H][W] fp32, C = 10 or 20; single_layer != 0 is the reference's default layout (front RGBA, back
 * RGBA, depths in channels 8:10, whatever C is: laina's default is 20 output channels with single_layer), 0 the
 * multi-layer one (C = 20: 3 depth-sorted RGBA layers, back RGBA, depths 16:20); rgba [N][4][H][W].  terms = OR of MDE_ST_* (the reference selects them by substrings
 * of method.loss; COMPOSITE_SSIM = 'composite' and 'ssim' both present).  The composite terms need single_layer, where
 * the reference's own indexing is well-formed.  out[12] = total, depth_silog, color_mae, color_mse, all_mse, all_mae,
 * all_ssim, front_ssim, back_ssim, composite_mse, composite_ssim, fb_divergence (unselected = 0).
 * pred_full: optional [N][4][H][W] output, the clamped composite (any C); required when COMPOSITE_SSIM is set.
 * scratch: caller-owned fp32, >= mde_stdepth_scratch_elems(...) elements, carried from fwd to bwd with ws
 * (>= mde_stdepth_ws_bytes()); may be NULL when no SSIM term is selected. */
#define MDE_ST_SILMA 1u
#define MDE_ST_SILMS 2u
#define MDE_ST_MSE 4u
#define MDE_ST_MAE 8u
#define MDE_ST_ALLSSIM 16u
#define MDE_ST_COLORSSIM 32u
#define MDE_ST_COMPOSITE 64u
#define MDE_ST_COMPOSITE_SSIM 128u
#define MDE_ST_FBDIV 256u
size_t mde_stdepth_ws_bytes(void);
size_t mde_stdepth_scratch_elems(int N, int C, int H, int W, unsigned terms);
int mde_stdepth_fwd(const float* pred, const float* targ, const float* rgba, int N, int C, int H, int W,
                    int single_layer, unsigned terms, float variance_focus, float depth_w, float comp_w, float fbdiv_w, float ssim_w,
                    void* ws, float* scratch, float* pred_full, float* out, void* stream);
int mde_stdepth_bwd(const float* pred, const float* targ, const float* rgba, int N, int C, int H, int W,
                    int single_layer, unsigned terms, float variance_focus, float depth_w, float comp_w, float fbdiv_w, float ssim_w,
                    const void* ws, float* scratch, const float* pred_full, const float* gscale, float* grad,
                    void* stream);
/* Depth metrics (metrics.py:58-123) over target > 0 with pred clamped at 1e-7: out[10] = absrel, 'rmse' (= mean
 * sqrt((p-t)^2/t), sic), delta1, delta2, delta3, log10, mae, mse, msle (= mean (log1p p - log1p t)^2, the
 * torchmetrics definitions the reference maps those names to), sqrel.  ws >= mde_metrics_ws_bytes(). */
size_t mde_metrics_ws_bytes(void);
int mde_depth_metrics(const float* pred, const float* target, int64_t n, void* ws, float* out,
                      void* stream);
/* The 'ssim' entry of the reference's metric list (metrics.py:63,123: torchmetrics 0.7.3
 * structural_similarity_index_measure with its defaults, on clamp_min(pred, 1e-7) and the unmasked target): 11 x 11
 * Gaussian window (sigma 1.5), data_range from the two tensors' extrema, mean over the pixels whose window lies inside
 * the image.  pred / target: fp32 [planes][H][W] (planes = N * C), H, W > 10; ws >= mde_ssim_metric_ws_bytes(), 8-byte
 * aligned; out: one float. */
size_t mde_ssim_metric_ws_bytes(void);
int mde_ssim_metric(const float* pred, const float* target, int planes, int H, int W, void* ws, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Optimiser / parameter plumbing (torch.optim.Adam as configured at modules/laina.py:51-57).
 * ---------------------------------------------------------------------------------- */
/* One Adam step over a flat fp32 range; also refreshes the bf16 shadow copy used by the convs.
 * step >= 1.  weight_decay is L2 (added to grad) like torch.optim.Adam.  grad_scale multiplies
 * the gradient first (1/world_size after a sum all-reduce). */
int mde_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, float grad_scale, int step,
                  void* stream);
/* torch.optim.AdamW (modules/bts.py:139-152: eps 1e-3, weight_decay 1e-2 / 0): same contract as mde_adam_step, the
 * decay is decoupled (p *= 1 - lr * weight_decay before the Adam update; the moments never see it). */
int mde_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float grad_scale, int step,
                   void* stream);
/* torch.optim.SGD with momentum (modules/vnl.py:289-326: momentum 0.9, weight_decay 5e-4; dampening 0, no Nesterov):
 * g' = grad_scale * g + weight_decay * p; buf = momentum * buf + g'; p -= lr * buf.  buf starts at zero. */
int mde_sgd_step(float* p, const float* g, float* buf, void* p_bf16, int64_t n, float lr, float momentum,
                 float weight_decay, float grad_scale, void* stream);
/* bf16 cast of a flat fp32 range (weight shadow refresh after load_state_dict). */
int mde_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst[i][t][o] = bf16(src[o][t][i])  — the dgrad ("transposed") weight packing. */
int mde_pack_wt(const float* src, void* dst, int O, int T, int I, void* stream);
/* The same for many weights of one flat parameter buffer in ONE launch: job j transposes
 * src[off .. off + O*T*I) into dst[off ..) (same element offset in a flat bf16 buffer).
 * jobs: DEVICE array sorted by first_block; first_block = sum over earlier jobs of
 * ceil(I/32)*ceil(O/32)*T; nblocks = that sum over all jobs. */
typedef struct mde_pack_job {
    int64_t off;
    int64_t O, T, I;
    int64_t first_block;
} mde_pack_job;
int mde_pack_wt_batch(const float* src, void* dst, const mde_pack_job* jobs, int njobs, int64_t nblocks,
                      void* stream);
/* The eval-mode TWO-TERM weight shadow (north_star: "AbsRel within 1e-4 of CPU reference on identical weights"; the
 * reference's eval forward, e.g. network/FCRN.py:351-371, multiplies with fp32 weights): w = hi + lo + O(2^-17 |w|) with
 * hi = (bf16)w and lo = (bf16)(w - hi).  Job j (as above) writes the TAP-DOUBLED GEMM operand of its weight at dst + 2*off:
 * transposed == 0: [O][2T][I], hi in taps [0, T), lo in taps [T, 2T); transposed != 0: [I][2T][O] likewise.  A convolution
 * whose descriptor lists every tap twice -- (dy, dx, wtap) and (dy, dx, wtap + T), wtaps_total = 2T -- then contracts the
 * activation with both terms in ONE fp32 accumulation: the weight-rounding error of an eval-mode output drops from 2^-9 to
 * 2^-17 relative at twice the MFMA work.  Training keeps the one-term shadow (mde_cast_bf16 / the optimiser steps). */
int mde_pack_split_batch(const float* src, void* dst, const mde_pack_job* jobs, int njobs, int64_t nblocks,
                         int transposed, void* stream);
/* Detecting parameter writes torch's version counters do not see (`.data` writes, collectives into detached views):
 * mde_param_fingerprint folds the raw words of the flat fp32 range into a 64-bit position-weighted sum and records in
 * `state` (DEVICE, >= mde_param_fingerprint_state_bytes(), zero-initialised once) whether it differs from the previous
 * call's; mde_refresh_if_changed then re-derives the bf16 shadow and the transposed packings ON DEVICE only if it did
 * (no host synchronisation: the kernels read the flag and return).  Cost when nothing changed: one read of the range. */
size_t mde_param_fingerprint_state_bytes(void);
int mde_param_fingerprint(const float* p, int64_t n, void* state, void* stream);
int mde_refresh_if_changed(const float* src, void* shadow, void* packed, const mde_pack_job* jobs, int njobs,
                           int64_t nblocks, int64_t n, const void* state, void* stream);

/* fp32 NCHW -> bf16 NHWC (and back) layout changes at the module boundary. */
int mde_nchw_to_nhwc_bf16(const float* src, void* dst, int N, int C, int H, int W, void* stream);
/* The same into [N][H][W][Cpad], channels [C, Cpad) zero: the input of the generic stem (FCRN.py:307-313,
 * in_channels != 3), whose 7x7/2 conv runs on the GEMM kernel with the input channels padded to 64. */
int mde_nchw_to_nhwc_bf16_pad(const float* src, void* dst, int N, int C, int H, int W, int Cpad, void* stream);
int mde_nhwc_bf16_to_nchw(const void* src, float* dst, int N, int C, int H, int W, void* stream);
/* Eval-mode operands of an image convolution on the GEMM kernel (densenet161's conv0, Bts.py:289; DORN's, Eigen's image convs):
 * 16 channel slots per pixel / per (output channel, tap) -- image [xh | xl | xh | 0 ...], weight [wh | wh | wl | 0 ...] with
 * h = (bf16)v, l = (bf16)(v - h) -- so ONE contraction over the slots is x w to 2^-16 relative: the fp32 image (as the
 * reference's forward sees it) against the two-term weight shadow (mde_pack_split_batch), in as many launches as the training
 * step's.  src image fp32 NCHW with C <= 5 channels -> dst bf16 [N][H][W][16]; src weight fp32 [rows = O * T][Cp] (C real
 * channels) -> dst bf16 [rows][16]. */
int mde_nchw_to_nhwc_split16(const float* src, void* dst, int N, int C, int H, int W, void* stream);
int mde_stem_weight_split16(const float* src, void* dst, int64_t rows, int Cp, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDE_HIP_H */
