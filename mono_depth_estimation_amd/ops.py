"""Host-side wrappers over the C ABI (include/mde_hip.h): descriptor builders and launchers.

Tensors are torch CUDA tensors used purely as device-memory handles (data_ptr + current
stream); all arithmetic happens in libmde_hip.so.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvDesc, WgradDesc, check


# the torch dtype of the library's 16-bit storage type: every activation, activation gradient and GEMM weight shadow
ACT_DTYPE = torch.float16 if _lib.ACT_NAME == "fp16" else torch.bfloat16


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _nbytes_from(t):
    """Bytes addressable from t.data_ptr() inside its storage."""
    return t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()


# ------------------------------------------------------------------------------ conv descriptors
def _fill_taps(d, taps, field3):
    assert 1 <= len(taps) <= _lib.MAX_TAPS, len(taps)
    d.ntaps = len(taps)
    third = getattr(d, field3)
    for i, (dy, dx, wt) in enumerate(taps):
        d.dy[i], d.dx[i], third[i] = dy, dx, wt


def conv_desc(N, H, W, ld_in, Cc, in_bytes, GH, GW, sy, sx, taps, wtaps_total, OH, OW, ld_out,
              osy=1, osx=1, ooy=0, oox=0, ncols=0, accumulate=False):
    d = ConvDesc()
    d.N, d.H, d.W, d.ld_in, d.C, d.in_bytes = N, H, W, ld_in, Cc, in_bytes
    d.GH, d.GW, d.sy, d.sx = GH, GW, sy, sx
    _fill_taps(d, taps, "wtap")
    d.wtaps_total = wtaps_total
    d.OH, d.OW, d.ld_out = OH, OW, ld_out
    d.osy, d.osx, d.ooy, d.oox = osy, osx, ooy, oox
    d.ncols, d.accumulate = ncols, int(accumulate)
    return d


def out_size(n, k, s, p, dil=1):
    return (n + 2 * p - dil * (k - 1) - 1) // s + 1


def fwd_desc(N, H, W, ld_in, Cin, in_bytes, k, stride, pad, Cout, ld_out, dil=1):
    """Plain Conv2d forward, square kernel k, weights [Cout][k*k][Cin]."""
    OH, OW = out_size(H, k, stride, pad, dil), out_size(W, k, stride, pad, dil)
    taps = [(i * dil - pad, j * dil - pad, i * k + j) for i in range(k) for j in range(k)]
    return conv_desc(N, H, W, ld_in, Cin, in_bytes, OH, OW, stride, stride, taps, k * k, OH, OW, ld_out,
                     ncols=Cout)


def taps_fwd_desc(N, H, W, ld_in, Cin, in_bytes, taps, ntaps_total, Cout, ld_out):
    """Stride-1 same-size conv given as an explicit tap list [(dy, dx, weight tap)] (asymmetric padding, non-square
    kernels: FCRN.py:236-239).  Weights [Cout][ntaps_total][Cin]."""
    return conv_desc(N, H, W, ld_in, Cin, in_bytes, H, W, 1, 1, taps, ntaps_total, H, W, ld_out, ncols=Cout)


def taps_dgrad_desc(N, H, W, ld_dx, Cin, ld_dy, Cout, dy_bytes, taps, ntaps_total, accumulate=False):
    """Input gradient of taps_fwd_desc's conv: the mirrored taps over dY with the transposed weights."""
    return conv_desc(N, H, W, ld_dy, Cout, dy_bytes, H, W, 1, 1, [(-dy, -dx, t) for dy, dx, t in taps], ntaps_total, H, W,
                     ld_dx, ncols=Cin, accumulate=accumulate)


def taps_wgrad_desc(N, H, W, ld_x, Cin, x_bytes, ld_dy, Cout, dy_bytes, taps, ntaps_total, ksplit):
    """dW[Cout][ntaps_total][Cin] of taps_fwd_desc's conv: direct = dY, gathered = x."""
    return wgrad_desc(N, H, W, ld_dy, Cout, dy_bytes, H, W, ld_x, Cin, x_bytes, 1, 1, taps, ntaps_total, False, ksplit)


def dgrad_descs(N, H, W, ld_dx, Cin, OH, OW, ld_dy, Cout, dy_bytes, k, stride, pad, dil=1, accumulate=False):
    """Conv2d input gradient as 1 (stride 1) or stride^2 output-phase launches over
    dY [N][OH][OW][ld_dy] with the transposed weights [Cin][k*k][Cout].
    Returns (descs, needs_zero_fill): phases without any tap are not launched."""
    s = stride
    descs, covered = [], 0
    for a in range(s):
        for b in range(s):
            ti = [(i, (a + pad - i * dil) // s) for i in range(k) if (a + pad - i * dil) % s == 0]
            tj = [(j, (b + pad - j * dil) // s) for j in range(k) if (b + pad - j * dil) % s == 0]
            gh, gw = (H - a + s - 1) // s, (W - b + s - 1) // s
            if not ti or not tj or gh <= 0 or gw <= 0:
                continue
            taps = [(dy, dx, i * k + j) for (i, dy) in ti for (j, dx) in tj]
            descs.append(conv_desc(N, OH, OW, ld_dy, Cout, dy_bytes, gh, gw, 1, 1, taps, k * k, H, W, ld_dx,
                                   osy=s, osx=s, ooy=a, oox=b, ncols=Cin, accumulate=accumulate))
            covered += 1
    return descs, covered != s * s


def upproj_fwd_descs(N, h, w, ld_in, Cin, in_bytes, ncols, ld_out):
    """5x5/pad-2 conv over the zero-stuffed 2x upsampling of x [N][h][w][Cin], as four
    output phases over x itself (FCRN.py:31-44 + 180,187).  Weights [ncols][25][Cin]."""
    descs = []
    for a in range(2):
        for b in range(2):
            taps = [((a + i - 2) // 2, (b + j - 2) // 2, i * 5 + j)
                    for i in range(5) if (a + i) % 2 == 0 for j in range(5) if (b + j) % 2 == 0]
            descs.append(conv_desc(N, h, w, ld_in, Cin, in_bytes, h, w, 1, 1, taps, 25, 2 * h, 2 * w, ld_out,
                                   osy=2, osx=2, ooy=a, oox=b, ncols=ncols))
    return descs


def upproj_dgrad_desc(N, h, w, ld_dx, Cin, ld_dy, Cdy, dy_bytes, accumulate=False):
    """dx[gy,gx] = sum_{i,j} dY[2gy+2-i, 2gx+2-j] * W[.,i,j,.]: a stride-2 5x5 gather over
    dY [N][2h][2w][ld_dy] with transposed weights [Cin][25][Cdy]."""
    taps = [(2 - i, 2 - j, i * 5 + j) for i in range(5) for j in range(5)]
    return conv_desc(N, 2 * h, 2 * w, ld_dy, Cdy, dy_bytes, h, w, 2, 2, taps, 25, h, w, ld_dx, ncols=Cin,
                     accumulate=accumulate)


class LaunchTimer:
    """Optional per-launch HIP-event timing of the GEMM kernels (bench.py's roofline leg).
    Events are recorded on the stream the kernels run on (torch's current stream)."""

    def __init__(self):
        self.records = []          # (kind, flops, start_event, end_event, shape tag)
        self.bytes = []            # algorithmic HBM bytes of the same launches (operands and results once each), same order

    def binding(self, peak_flops, peak_bytes):
        """kind -> (seconds at whichever roofline binds each launch -- max(flops / peak_flops, bytes / peak_bytes), summed --,
        launches bound by HBM, measured seconds); call after a device synchronize."""
        out = {}
        for (kind, flops, e0, e1, _), nbytes in zip(self.records, self.bytes):
            tf, tb = flops / peak_flops, nbytes / peak_bytes
            lo, nh, t = out.get(kind, (0.0, 0, 0.0))
            out[kind] = (lo + max(tf, tb), nh + int(tb > tf), t + e0.elapsed_time(e1) * 1e-3)
        return out

    def summary(self):
        """kind -> (launches, total flops, total seconds); call after a device synchronize."""
        out = {}
        for kind, flops, e0, e1, _ in self.records:
            n, f, t = out.get(kind, (0, 0.0, 0.0))
            out[kind] = (n + 1, f + flops, t + e0.elapsed_time(e1) * 1e-3)
        return out


TIMER = None   # set to a LaunchTimer to time every conv_gemm / conv_wgrad launch


def _timed(kind, flops, fn, tag="", nbytes=0.0):
    if TIMER is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    TIMER.records.append((kind, flops, e0, e1, tag))
    TIMER.bytes.append(nbytes)


def set_deterministic(on, gbase=None, scratch=None):
    n = gbase.numel() if gbase is not None else 0
    check(_lib.load().mde_set_deterministic(int(on), _p(gbase), _p(scratch), n), "mde_set_deterministic")


def det_scratch(gbase):
    """Zeroed integer shadow of a flat gradient buffer (2 x int64 per element)."""
    return torch.zeros(2 * gbase.numel(), dtype=torch.int64, device=gbase.device)


def det_flush():
    check(_lib.load().mde_det_flush(_stream()), "mde_det_flush")


ACT_CODE = {None: 0, "none": 0, "relu": 1, "elu": 2, "sigmoid": 3}


def bn_red(x, save_mean, save_rstd, part, mask_scale=None, mask_shift=None, relu_bits=None, x_ld=0, second=None):
    """mde_bn_red: the BatchNorm site whose backward sums an input-gradient launch adds from its epilogue (conv_gemm(red=...)).
    second = (x2, save_mean2, save_rstd2, part2, x2_ld): the other site of a residual join.  The returned object keeps the
    tensors alive."""
    dp = lambda t: t.data_ptr() if t is not None else None
    r = _lib.BnRed(*[dp(t) for t in (x, save_mean, save_rstd, mask_scale, mask_shift, relu_bits, part)], x_ld)
    r._keep = (x, save_mean, save_rstd, part, mask_scale, mask_shift, relu_bits, second)
    if second is not None:
        r.x2, r.save_mean2, r.save_rstd2, r.part2 = (dp(t) for t in second[:4])
        r.x2_ld = second[4]
    return r


def bn_red_with_add(red, add, add_bits):
    """A copy of `red` whose launch also adds `add` (bf16, laid out like the launch's output) under `add_bits` to its result:
    the gradient an identity shortcut hands to the block's input (mde_bn_red.add)."""
    r = _lib.BnRed.from_buffer_copy(red)
    r.add = add.data_ptr()
    r.add_bits = add_bits.data_ptr() if add_bits is not None else None
    r._keep = (red, add, add_bits)
    return r


def conv_gemm(desc, x, w, out, stats=None, bias=None, res=None, act=None, red=None):
    """bias / res / act: the fused epilogue out = act(conv + bias + res) (mde_conv_gemm_act; no statistics with it).
    red (ops.bn_red): the launch writes the gradient of a BatchNorm site's output and adds that site's backward sums
    (mde_conv_gemm_bnred)."""
    lib = _lib.load()
    if red is not None:
        assert stats is None and bias is None and res is None and not ACT_CODE[act]
        call = lambda: check(lib.mde_conv_gemm_bnred(C.byref(desc), _p(x), _p(w), _p(out), C.byref(red), _stream()), "mde_conv_gemm_bnred")
    elif bias is not None or res is not None or ACT_CODE[act]:
        assert stats is None, "BatchNorm statistics and a fused activation epilogue do not combine"
        call = lambda: check(lib.mde_conv_gemm_act(C.byref(desc), _p(x), _p(w), _p(out), _p(bias), _p(res), ACT_CODE[act], _stream()),
                             "mde_conv_gemm_act")
    else:
        call = lambda: check(lib.mde_conv_gemm(C.byref(desc), _p(x), _p(w), _p(out), _p(stats), _stream()), "mde_conv_gemm")
    if TIMER is None:
        call()
        return
    M = desc.N * desc.GH * desc.GW
    # algorithmic FLOP: a grouped launch contracts 64 channels per column tile of which only the group's own are the
    # convolution's (graph.Conv sets `_useful` = group size / 64 on its descriptors); the rest are zeros of the block-diagonal packing
    flops = 2.0 * M * desc.ncols * desc.ntaps * desc.C * getattr(desc, "_useful", 1.0)
    # algorithmic bytes: the input tensor, the weights and the output once each (bf16), + what the epilogue is asked to read:
    # the old output (accumulate), a residual, the BatchNorm site input(s) of a launch that carries backward sums
    extra = int(bool(desc.accumulate)) + int(res is not None) + (0 if red is None else (2 if red.x2 else 1) + int(bool(red.add)))
    nbytes = 2.0 * (desc.N * desc.H * desc.W * desc.C + desc.ncols * desc.ntaps * desc.C + M * desc.ncols * (1 + extra))
    _timed("conv_gemm_nt", flops, call,
           "M=%d N=%d taps=%d C=%d s=%d%s" % (M, desc.ncols, desc.ntaps, desc.C, desc.sy, " acc" if desc.accumulate else ""), nbytes)


def split_descs(d):
    """The eval-mode form of a conv descriptor over the TWO-TERM weight operand (mde_pack_split_batch: [rows][2T][C], hi = bf16(w)
    in taps [0, T), lo = bf16(w - hi) in [T, 2T)): every tap listed twice, (dy, dx, wtap) and (dy, dx, wtap + T), so ONE fp32
    accumulation contracts the activation with both terms.  More than MDE_MAX_TAPS taps run as several launches, the later ones
    accumulating.  Cached on the descriptor (a forward descriptor's fields are fixed at plan time)."""
    sp = getattr(d, "_split", None)
    if sp is None:
        T = d.wtaps_total
        taps = [(d.dy[i], d.dx[i], d.wtap[i]) for i in range(d.ntaps)]
        both = [(dy, dx, wt + h * T) for (dy, dx, wt) in taps for h in (0, 1)]
        sp = []
        for t0 in range(0, len(both), _lib.MAX_TAPS):
            e = ConvDesc.from_buffer_copy(d)
            _fill_taps(e, both[t0:t0 + _lib.MAX_TAPS], "wtap")
            e.wtaps_total = 2 * T
            if t0:
                e.accumulate = 1
            if hasattr(d, "_useful"):
                e._useful = d._useful
            sp.append(e)
        d._split = sp
    return sp


def conv_gemm_eval(desc, x, w2, out, bias=None, res=None, act=None):
    """An eval-mode forward conv with the two-term weight operand `w2` (split_descs).  A fused epilogue needs the whole
    contraction in one launch (an accumulating launch has none): the caller checks split_fits first."""
    ds = split_descs(desc)
    fused = bias is not None or res is not None or ACT_CODE[act]
    assert len(ds) == 1 or not fused
    for d in ds:
        conv_gemm(d, x, w2, out, bias=bias, res=res, act=act)


def split_fits(desc):
    """True if the doubled tap list of split_descs(desc) is ONE launch (what a fused epilogue needs)."""
    return 2 * desc.ntaps <= _lib.MAX_TAPS


def stat_slots():
    return _lib.load().mde_stat_slots()


def new_stat_buffer(C_, device="cuda"):
    """A zeroed BatchNorm partial-sum buffer [slots][2][C] (finalize kernels re-zero it)."""
    return torch.zeros(stat_slots(), 2, C_, dtype=torch.float32, device=device)


# ------------------------------------------------------------------------------ wgrad descriptors
def wgrad_desc(N, GH, GW, ld_d, Cd, d_bytes, H, W, ld_g, Cg, g_bytes, sy, sx, taps, otaps_total,
               rows_from_gathered, ksplit):
    d = WgradDesc()
    d.N, d.GH, d.GW, d.ld_d, d.Cd = N, GH, GW, ld_d, Cd
    d.H, d.W, d.ld_g, d.Cg = H, W, ld_g, Cg
    d.d_bytes, d.g_bytes = d_bytes, g_bytes
    d.sy, d.sx = sy, sx
    _fill_taps(d, taps, "otap")
    d.otaps_total, d.rows_from_gathered, d.ksplit = otaps_total, int(rows_from_gathered), ksplit
    return d


def conv_wgrad_desc(N, H, W, ld_x, Cin, x_bytes, OH, OW, ld_dy, Cout, dy_bytes, k, stride, pad, ksplit, dil=1):
    """dW[Cout][k*k][Cin] of a plain Conv2d: direct = dY, gathered = x."""
    taps = [(i * dil - pad, j * dil - pad, i * k + j) for i in range(k) for j in range(k)]
    return wgrad_desc(N, OH, OW, ld_dy, Cout, dy_bytes, H, W, ld_x, Cin, x_bytes, stride, stride, taps, k * k,
                      False, ksplit)


def upproj_wgrad_desc(N, h, w, ld_x, Cin, x_bytes, ld_dy, Cdy, dy_bytes, ksplit):
    """dW[Cdy][25][Cin] of the up-projection 5x5: direct = x (grid h x w), gathered = dY."""
    taps = [(2 - i, 2 - j, i * 5 + j) for i in range(5) for j in range(5)]
    return wgrad_desc(N, h, w, ld_x, Cin, x_bytes, 2 * h, 2 * w, ld_dy, Cdy, dy_bytes, 2, 2, taps, 25, True, ksplit)


def wgrad_ws_bytes(desc):
    """Bytes of workspace the two-stage split-K reduction of this launch needs (0: no such form)."""
    return int(_lib.load().mde_conv_wgrad_ws_bytes(C.byref(desc)))


def conv_wgrad(desc, direct, gathered, dw, ws=None):
    """ws: a workspace (any dtype, 16-byte aligned) for the two-stage split-K reduction (mde_conv_wgrad_ws); launches that
    share it, or add into the same dw, must be ordered on one stream.  None: fp32 atomics into dw."""
    lib = _lib.load()
    nb = ws.numel() * ws.element_size() if ws is not None else 0
    call = lambda: check(lib.mde_conv_wgrad_ws(C.byref(desc), _p(direct), _p(gathered), _p(dw), _p(ws), nb, _stream()), "mde_conv_wgrad_ws")
    if TIMER is None:
        call()
        return
    flops = 2.0 * desc.N * desc.GH * desc.GW * desc.Cd * (desc.group_size if desc.group_size else desc.Cg) * desc.ntaps   # (grouped: the block diagonal)
    nbytes = 2.0 * (desc.N * desc.GH * desc.GW * desc.Cd + desc.N * desc.H * desc.W * desc.Cg) + 4.0 * desc.Cd * desc.Cg * desc.ntaps
    _timed("conv_wgrad_tn", flops, call,
           "K=%d Cd=%d Cg=%d taps=%d ks=%d" % (desc.N * desc.GH * desc.GW, desc.Cd, desc.Cg, desc.ntaps, desc.ksplit), nbytes)


def wgrad_time_model(pixels, row_tiles, col_tiles, ntaps, ks, cus=256, wg_per_cu=2, tile_elems=128 * 128):
    """Estimated conv_wgrad_tn time (s) at split-K factor ks, fitted to MI355X sweeps (tools/conv_microbench.py
    wgrad with MB_KS=...):  rounds of resident workgroups, a last round that is shorter
    when it leaves CUs with fewer co-resident workgroups (a lone 4-wave workgroup reaches ~62 % of a CU), plus
    the split-K reduction traffic: every workgroup adds its fp32 tile with atomics (~500 G elements/s chip-wide)."""
    blocks = max(1, row_tiles * col_tiles * ntaps) * ks
    slots = cus * wg_per_cu
    rounds = -(-blocks // slots)
    need = 2 if tile_elems >= 16384 else 3 if tile_elems >= 8192 else 4      # co-resident workgroups for full rate
    f = lambda k: min(1.0, 0.62 + 0.38 * (k - 1) / (need - 1))               # CU throughput with k workgroups
    full = wg_per_cu / f(wg_per_cu)
    last = blocks - (rounds - 1) * slots
    k = min(wg_per_cu, -(-last // cus))
    h = (k / f(k)) / full                                                     # last round relative to a full one
    rate = 750e12 if tile_elems >= 16384 else 600e12 if tile_elems >= 8192 else 560e12
    t_round = 2.0 * (pixels / ks) * tile_elems * wg_per_cu / (rate / cus)
    return t_round * ((rounds - 1) + h) + blocks * tile_elems / 500e9


def choose_ksplit(pixels, row_tiles, col_tiles, ntaps, cus=256, min_steps=8, wg_per_cu=2, tile_elems=128 * 128):
    """Split-K factor for the weight-gradient GEMM (grid = ntaps*row_tiles*col_tiles*ksplit workgroups):
    the factor with the smallest modelled time (wgrad_time_model) among those that keep >= min_steps
    64-pixel K-steps per workgroup.  Filling the last round matters, and so does the reduction: each extra
    split adds one fp32 atomic per output element (the old fill-only rule chose 227 splits where 56 is 40 % faster)."""
    base = max(1, row_tiles * col_tiles * ntaps)
    slots = cus * wg_per_cu
    cap = max(1, pixels // (64 * min_steps))
    hi = min(cap, max(1, (4 * slots + base - 1) // base))
    best, best_t = 1, None
    for ks in range(1, hi + 1):
        t = wgrad_time_model(pixels, row_tiles, col_tiles, ntaps, ks, cus, wg_per_cu, tile_elems)
        if best_t is None or t < best_t * (1.0 - 1e-9):
            best, best_t = ks, t
    scale = float(os.environ.get("MDE_WGRAD_KS_SCALE", "1"))       # (diagnostics: the model is fitted to launches running alone)
    return best if scale == 1.0 else max(1, min(cap, int(round(best * scale))))


# ------------------------------------------------------------------------------ small convs
def stem_conv_fwd(x, w, out, stats=None, cout=64):
    N, _, H, W = x.shape
    check(_lib.load().mde_stem_conv_fwd_c(_p(x), _p(w), _p(out), _p(stats), N, H, W, cout, _stream()), "mde_stem_conv_fwd_c")


def stem_conv_wgrad(x, dout, dw, cout=64):
    N, _, H, W = x.shape
    check(_lib.load().mde_stem_conv_wgrad_c(_p(x), _p(dout), _p(dw), N, H, W, cout, _stream()), "mde_stem_conv_wgrad_c")


def head_conv_fwd(x, w, out, N, H, W, Cin, Cout):
    check(_lib.load().mde_head_conv_fwd(_p(x), _p(w), _p(out), N, H, W, Cin, Cout, _stream()), "mde_head_conv_fwd")


def head_conv_bwd(x, w, dout, dx, dw, N, H, W, Cin, Cout):
    check(_lib.load().mde_head_conv_bwd(_p(x), _p(w), _p(dout), _p(dx), _p(dw), N, H, W, Cin, Cout, _stream()),
          "mde_head_conv_bwd")


# ------------------------------------------------------------------------------ batch norm
def bn_stats(x, M, C_, ld, part):
    check(_lib.load().mde_bn_stats(_p(x), M, C_, ld, _p(part), _stream()), "mde_bn_stats")


def bn_finalize(part, M, C_, gamma, beta, rmean, rvar, momentum, eps, scale, shift, smean, srstd):
    check(_lib.load().mde_bn_finalize(_p(part), M, C_, _p(gamma), _p(beta), _p(rmean), _p(rvar), momentum, eps,
                                      _p(scale), _p(shift), _p(smean), _p(srstd), _stream()), "mde_bn_finalize")


def bn_moments(part, M, C_, mean, var):
    check(_lib.load().mde_bn_moments(_p(part), M, C_, _p(mean), _p(var), _stream()), "mde_bn_moments")


def bn_finalize_moments(mean, var, M, C_, gamma, beta, rmean, rvar, momentum, eps, scale, shift, smean, srstd):
    check(_lib.load().mde_bn_finalize_moments(_p(mean), _p(var), M, C_, _p(gamma), _p(beta), _p(rmean), _p(rvar), momentum, eps,
                                              _p(scale), _p(shift), _p(smean), _p(srstd), _stream()), "mde_bn_finalize_moments")


def bn_eval_scale_shift(gamma, beta, rmean, rvar, eps, C_, scale, shift):
    check(_lib.load().mde_bn_eval_scale_shift(_p(gamma), _p(beta), _p(rmean), _p(rvar), eps, C_, _p(scale), _p(shift),
                                              _stream()), "mde_bn_eval_scale_shift")


def bn_apply(x, ldx, scale, shift, out, ldo, M, C_, relu, r=None, ldr=0, rscale=None, rshift=None, relu_bits=None):
    check(_lib.load().mde_bn_apply(_p(x), ldx, _p(scale), _p(shift), _p(r), ldr, _p(rscale), _p(rshift), _p(out), ldo,
                                   _p(relu_bits), M, C_, int(relu), _stream()), "mde_bn_apply")


def bn_bwd_reduce(dout, ldd, out, ldo, x, ldx, smean, srstd, M, C_, relu, part, mask_scale=None, mask_shift=None,
                  relu_bits=None):
    check(_lib.load().mde_bn_bwd_reduce(_p(dout), ldd, _p(out), ldo, _p(x), ldx, _p(smean), _p(srstd), _p(mask_scale),
                                        _p(mask_shift), _p(relu_bits), M, C_, int(relu), _p(part), _stream()),
          "mde_bn_bwd_reduce")


def bn_bwd_reduce2(dout, ldd, xa, ldxa, xb, ldxb, mean_a, rstd_a, mean_b, rstd_b, relu_bits, M, C_, part_a, part_b):
    check(_lib.load().mde_bn_bwd_reduce2(_p(dout), ldd, _p(xa), ldxa, _p(xb), ldxb, _p(mean_a), _p(rstd_a), _p(mean_b),
                                         _p(rstd_b), _p(relu_bits), M, C_, _p(part_a), _p(part_b), _stream()),
          "mde_bn_bwd_reduce2")


def bn_bwd_apply2(dout, ldd, xa, ldxa, xb, ldxb, mean_a, rstd_a, mean_b, rstd_b, relu_bits, coef_a, coef_b, M, C_, dxa, ldda,
                  dxb, lddb):
    check(_lib.load().mde_bn_bwd_apply2(_p(dout), ldd, _p(xa), ldxa, _p(xb), ldxb, _p(mean_a), _p(rstd_a), _p(mean_b),
                                        _p(rstd_b), _p(relu_bits), _p(coef_a), _p(coef_b), M, C_, _p(dxa), ldda, _p(dxb), lddb,
                                        _stream()), "mde_bn_bwd_apply2")


def bn_bwd_finalize(part, M, C_, gamma, srstd, dgamma, dbeta, coef):
    check(_lib.load().mde_bn_bwd_finalize(_p(part), M, C_, _p(gamma), _p(srstd), _p(dgamma), _p(dbeta), _p(coef),
                                          _stream()), "mde_bn_bwd_finalize")


def bn_bwd_apply(dout, ldd, out, ldo, x, ldx, smean, srstd, coef, M, C_, relu, dx, ldxo, accumulate=False, dres=None,
                 ldres=0, mask_scale=None, mask_shift=None, relu_bits=None):
    check(_lib.load().mde_bn_bwd_apply(_p(dout), ldd, _p(out), ldo, _p(x), ldx, _p(smean), _p(srstd), _p(mask_scale),
                                       _p(mask_shift), _p(relu_bits), _p(coef), M, C_, int(relu), _p(dx), ldxo, int(accumulate),
                                       _p(dres), ldres, _stream()), "mde_bn_bwd_apply")


def bn_apply_fin(x, ldx, fin, out, ldo, M, C_, relu, r=None, ldr=0, fin_r=None, relu_bits=None):
    """mde_bn_apply_fin: the statistics' finalize inside the apply launch (fin / fin_r: _lib.BnFin from BNSite.fin())."""
    check(_lib.load().mde_bn_apply_fin(_p(x), ldx, C.byref(fin), _p(r), ldr, C.byref(fin_r) if fin_r is not None else None, _p(out), ldo,
                                       _p(relu_bits), M, C_, int(relu), _stream()), "mde_bn_apply_fin")


def bn_bwd_apply_fin(dout, ldd, out, ldo, x, ldx, smean, srstd, bfin, M, C_, relu, dx, ldxo, accumulate=False, dres=None, ldres=0,
                     mask_scale=None, mask_shift=None, relu_bits=None):
    check(_lib.load().mde_bn_bwd_apply_fin(_p(dout), ldd, _p(out), ldo, _p(x), ldx, _p(smean), _p(srstd), _p(mask_scale), _p(mask_shift),
                                           _p(relu_bits), C.byref(bfin), M, C_, int(relu), _p(dx), ldxo, int(accumulate), _p(dres), ldres,
                                           _stream()), "mde_bn_bwd_apply_fin")


def bn_bwd_apply2_fin(dout, ldd, xa, ldxa, xb, ldxb, mean_a, rstd_a, mean_b, rstd_b, relu_bits, bfin_a, bfin_b, M, C_, dxa, ldda, dxb, lddb):
    check(_lib.load().mde_bn_bwd_apply2_fin(_p(dout), ldd, _p(xa), ldxa, _p(xb), ldxb, _p(mean_a), _p(rstd_a), _p(mean_b), _p(rstd_b),
                                            _p(relu_bits), C.byref(bfin_a), C.byref(bfin_b), M, C_, _p(dxa), ldda, _p(dxb), lddb, _stream()),
          "mde_bn_bwd_apply2_fin")


# ------------------------------------------------------------------------------ pool / resize
def pixel_shuffle2(src, ld_src, dst, ld_dst, N, h, w, C_, inverse=False):
    check(_lib.load().mde_pixel_shuffle2(_p(src), ld_src, _p(dst), ld_dst, N, h, w, C_, int(inverse), _stream()),
          "mde_pixel_shuffle2")


def maxpool_view_fwd(x, ldx, wpitch, ipitch, Hv, Wv, out, ldo, idx, N, C_, k, s):
    check(_lib.load().mde_maxpool_view_fwd(_p(x), ldx, wpitch, ipitch, Hv, Wv, _p(out), ldo, _p(idx), N, C_, k, s, _stream()),
          "mde_maxpool_view_fwd")


def maxpool_view_bwd(dout, ldd, idx, dx, lddx, wpitch, ipitch, Hv, Wv, N, C_, k, s, accumulate=False):
    check(_lib.load().mde_maxpool_view_bwd(_p(dout), ldd, _p(idx), _p(dx), lddx, wpitch, ipitch, Hv, Wv, N, C_, k, s, int(accumulate),
                                           _stream()), "mde_maxpool_view_bwd")


def maxpool_fwd(x, out, idx, N, H, W, C_, ceil_mode=False):
    check(_lib.load().mde_maxpool_fwd2(_p(x), _p(out), _p(idx), N, H, W, C_, int(ceil_mode), _stream()), "mde_maxpool_fwd2")


def maxpool_bwd(dout, idx, dx, N, H, W, C_, ceil_mode=False):
    check(_lib.load().mde_maxpool_bwd2(_p(dout), _p(idx), _p(dx), N, H, W, C_, int(ceil_mode), _stream()), "mde_maxpool_bwd2")


def maxpool_out_size(n, ceil_mode=False):
    """Output size of nn.MaxPool2d(3, 2, 1[, ceil_mode=True]) (ATen's rule; Dorn.py:235)."""
    if not ceil_mode:
        return (n + 2 - 3) // 2 + 1
    o = -(-(n + 2 - 3) // 2) + 1
    return o - 1 if (o - 1) * 2 >= n + 1 else o


# ---- DORN pieces (csrc/ordinal.hip)
def chan_scale(x, ldx, m, out, ldo, N, HW, C_, accumulate=False):
    check(_lib.load().mde_chan_scale(_p(x), ldx, _p(m), _p(out), ldo, N, HW, C_, int(accumulate), _stream()), "mde_chan_scale")


def avgpool_flat_fwd(x, ldx, m, out, N, H, W, C_, k, s, p):
    check(_lib.load().mde_avgpool_flat_fwd(_p(x), ldx, _p(m) if m is not None else None, _p(out), N, H, W, C_, k, s, p, _stream()),
          "mde_avgpool_flat_fwd")


def avgpool_flat_bwd(dout, m, dx, lddx, N, H, W, C_, k, s, p, accumulate=False):
    check(_lib.load().mde_avgpool_flat_bwd(_p(dout), _p(m) if m is not None else None, _p(dx), lddx, N, H, W, C_, k, s, p, int(accumulate),
                                           _stream()), "mde_avgpool_flat_bwd")


def ordinal_fwd(x, ldx, prob, label, N, HW, K):
    check(_lib.load().mde_ordinal_fwd(_p(x), ldx, _p(prob), _p(label), N, HW, K, _stream()), "mde_ordinal_fwd")


def ordinal_bwd(dprob, x, ldx, dx, lddx, N, HW, K):
    check(_lib.load().mde_ordinal_bwd(_p(dprob), _p(x), ldx, _p(dx), lddx, N, HW, K, _stream()), "mde_ordinal_bwd")


def weighted_pool_fwd(a, lda, w, b, pre, scale, N, HW, C_):
    check(_lib.load().mde_weighted_pool_fwd(_p(a), lda, _p(w), _p(b), _p(pre), _p(scale), N, HW, C_, _stream()), "mde_weighted_pool_fwd")


def weighted_pool_bwd(dscale, scale, a, lda, w, da, ldda, accumulate, dw, db, N, HW, C_):
    check(_lib.load().mde_weighted_pool_bwd(_p(dscale), _p(scale), _p(a), lda, _p(w), _p(da), ldda, int(accumulate), _p(dw), _p(db), N, HW, C_,
                                            _stream()), "mde_weighted_pool_bwd")


def combine3_fwd(maps, scales, factor, N, HW, out):
    check(_lib.load().mde_combine3_fwd(_p(maps[0]), _p(maps[1]), _p(maps[2]), _p(scales[0]), _p(scales[1]), _p(scales[2]), factor, N, HW,
                                       _p(out), _stream()), "mde_combine3_fwd")


def combine3_bwd(dout, maps, scales, factor, N, HW, dmaps, ds):
    check(_lib.load().mde_combine3_bwd(_p(dout), _p(maps[0]), _p(maps[1]), _p(maps[2]), _p(scales[0]), _p(scales[1]), _p(scales[2]), factor,
                                       N, HW, _p(dmaps[0]), _p(dmaps[1]), _p(dmaps[2]), _p(ds), _stream()), "mde_combine3_bwd")


def ord_loss_ws(device="cuda"):
    return torch.zeros((_lib.load().mde_ord_loss_ws_bytes() + 7) // 8, dtype=torch.float64, device=device)


def ord_loss_fwd(prob, target, N, K, HW, ws, loss):
    check(_lib.load().mde_ord_loss_fwd(_p(prob), _p(target), N, K, HW, _p(ws), _p(loss), _stream()), "mde_ord_loss_fwd")


def ord_loss_bwd(prob, target, N, K, HW, gscale, grad):
    check(_lib.load().mde_ord_loss_bwd(_p(prob), _p(target), N, K, HW, _p(gscale), _p(grad), _stream()), "mde_ord_loss_bwd")


def upsample_sigmoid_fwd(x, out, N, H, W, C_, OH, OW):
    check(_lib.load().mde_upsample_sigmoid_fwd(_p(x), _p(out), N, H, W, C_, OH, OW, _stream()), "mde_upsample_sigmoid_fwd")


def upsample_sigmoid_bwd(dout, out, dx, N, H, W, C_, OH, OW):
    check(_lib.load().mde_upsample_sigmoid_bwd(_p(dout), _p(out), _p(dx), N, H, W, C_, OH, OW, _stream()),
          "mde_upsample_sigmoid_bwd")


def nchw_to_nhwc_bf16(src, dst):
    N, C_, H, W = src.shape
    check(_lib.load().mde_nchw_to_nhwc_bf16(_p(src), _p(dst), N, C_, H, W, _stream()), "mde_nchw_to_nhwc_bf16")


def nchw_to_nhwc_bf16_pad(src, dst, Cpad):
    N, C_, H, W = src.shape
    check(_lib.load().mde_nchw_to_nhwc_bf16_pad(_p(src), _p(dst), N, C_, H, W, Cpad, _stream()), "mde_nchw_to_nhwc_bf16_pad")


def nchw_to_nhwc_split16(src, dst):
    N, C_, H, W = src.shape
    check(_lib.load().mde_nchw_to_nhwc_split16(_p(src), _p(dst), N, C_, H, W, _stream()), "mde_nchw_to_nhwc_split16")


def stem_weight_split16(src, dst, rows, Cp, C_):
    check(_lib.load().mde_stem_weight_split16(_p(src), _p(dst), rows, Cp, C_, _stream()), "mde_stem_weight_split16")


def nhwc_bf16_to_nchw(src, dst):
    N, C_, H, W = dst.shape
    check(_lib.load().mde_nhwc_bf16_to_nchw(_p(src), _p(dst), N, C_, H, W, _stream()), "mde_nhwc_bf16_to_nchw")


# ------------------------------------------------------------------------------ pointwise / pooling / resize (VNL, MiDaS, BTS)
ACT = {None: 0, "none": 0, "relu": 1, "elu": 2, "sigmoid": 3, "relu6": 4}


def dwconv3x3_fwd(x, ldx, w, out, ldo, N, H, W, C_, stride, dil):
    check(_lib.load().mde_dwconv3x3_fwd(_p(x), ldx, _p(w), _p(out), ldo, N, H, W, C_, stride, dil, _stream()), "mde_dwconv3x3_fwd")


def dwconv3x3_dgrad(dy, ldy, w, dx, lddx, N, H, W, C_, stride, dil, accumulate=False):
    check(_lib.load().mde_dwconv3x3_dgrad(_p(dy), ldy, _p(w), _p(dx), lddx, N, H, W, C_, stride, dil, int(accumulate), _stream()), "mde_dwconv3x3_dgrad")


def dwconv3x3_wgrad(x, ldx, dy, ldy, dw, N, H, W, C_, stride, dil):
    check(_lib.load().mde_dwconv3x3_wgrad(_p(x), ldx, _p(dy), ldy, _p(dw), N, H, W, C_, stride, dil, _stream()), "mde_dwconv3x3_wgrad")


def pw_fwd(x, ldx, bias, r, ldr, out, ldo, M, C_, act):
    check(_lib.load().mde_pw_fwd(_p(x), ldx, _p(bias), _p(r), ldr, _p(out), ldo, M, C_, ACT[act], _stream()), "mde_pw_fwd")


def pw_bwd(dout, ldd, out, ldo, dx, lddx, acc_x, dr, lddr, acc_r, dbias, M, C_, act, bias_part=None):
    check(_lib.load().mde_pw_bwd(_p(dout), ldd, _p(out), ldo, _p(dx), lddx, int(acc_x), _p(dr), lddr, int(acc_r), _p(dbias),
                                 _p(bias_part), M, C_, ACT[act], _stream()), "mde_pw_bwd")


def spatial_sum(x, ldx, N, HW, C_, scale, out, ldo):
    check(_lib.load().mde_spatial_sum(_p(x), ldx, N, HW, C_, scale, _p(out), ldo, _stream()), "mde_spatial_sum")


def spatial_bcast(src, lds, scale, out, ldo, N, HW, C_, accumulate=False):
    check(_lib.load().mde_spatial_bcast(_p(src), lds, scale, _p(out), ldo, N, HW, C_, int(accumulate), _stream()),
          "mde_spatial_bcast")


def gate_fwd(w, ldw, lat, ldl, top, ldt, out, ldo, N, HW, C_):
    check(_lib.load().mde_gate_fwd(_p(w), ldw, _p(lat), ldl, _p(top), ldt, _p(out), ldo, N, HW, C_, _stream()), "mde_gate_fwd")


def gate_bwd(dout, ldd, w, ldw, lat, ldl, dlat, lddl, acc_lat, dtop, lddt, acc_top, dw, lddw, N, HW, C_):
    check(_lib.load().mde_gate_bwd(_p(dout), ldd, _p(w), ldw, _p(lat), ldl, _p(dlat), lddl, int(acc_lat), _p(dtop), lddt,
                                   int(acc_top), _p(dw), lddw, N, HW, C_, _stream()), "mde_gate_bwd")


def resize_bilinear_fwd(x, ldx, out, ldo, N, H, W, C_, OH, OW, align_corners):
    check(_lib.load().mde_resize_bilinear_fwd(_p(x), ldx, _p(out), ldo, N, H, W, C_, OH, OW, int(align_corners), _stream()),
          "mde_resize_bilinear_fwd")


def resize_bilinear_bwd(dout, ldd, dx, lddx, N, H, W, C_, OH, OW, align_corners, accumulate=False):
    check(_lib.load().mde_resize_bilinear_bwd(_p(dout), ldd, _p(dx), lddx, N, H, W, C_, OH, OW, int(align_corners),
                                              int(accumulate), _stream()), "mde_resize_bilinear_bwd")


def nearest2_fwd(x, ldx, out, ldo, N, H, W, C_):
    check(_lib.load().mde_nearest2_fwd(_p(x), ldx, _p(out), ldo, N, H, W, C_, _stream()), "mde_nearest2_fwd")


def sum2x2(src, lds, dst, ldd, N, H, W, C_, scale, accumulate=False):
    check(_lib.load().mde_sum2x2(_p(src), lds, _p(dst), ldd, N, H, W, C_, scale, int(accumulate), _stream()), "mde_sum2x2")


def spread2x2(x, ldx, out, ldo, N, H, W, C_, scale, accumulate=False):
    check(_lib.load().mde_spread2x2(_p(x), ldx, _p(out), ldo, N, H, W, C_, scale, int(accumulate), _stream()), "mde_spread2x2")


def softmax_head_fwd(x, ldx, bias, logit, prob, N, HW, C_):
    check(_lib.load().mde_softmax_head_fwd(_p(x), ldx, _p(bias), _p(logit), _p(prob), N, HW, C_, _stream()), "mde_softmax_head_fwd")


def softmax_head_bwd(dlogit, dprob, prob, dx, lddx, dbias, N, HW, C_):
    check(_lib.load().mde_softmax_head_bwd(_p(dlogit), _p(dprob), _p(prob), _p(dx), lddx, _p(dbias), N, HW, C_, _stream()),
          "mde_softmax_head_bwd")


def map_act_fwd(p, y, act, scale=1.0):
    check(_lib.load().mde_map_act_fwd(_p(p), _p(y), p.numel(), ACT[act], scale, _stream()), "mde_map_act_fwd")


def map_act_bwd(dy, y, dp, act, scale=1.0):
    check(_lib.load().mde_map_act_bwd(_p(dy), _p(y), _p(dp), y.numel(), ACT[act], scale, _stream()), "mde_map_act_bwd")


def to_nchw_act_fwd(x, ldx, bias, out, N, HW, C_, act, scale=1.0):
    check(_lib.load().mde_to_nchw_act_fwd(_p(x), ldx, _p(bias), _p(out), N, HW, C_, ACT[act], scale, _stream()), "mde_to_nchw_act_fwd")


def to_nchw_act_bwd(dout, out, dx, lddx, dbias, N, HW, C_, act, scale=1.0):
    check(_lib.load().mde_to_nchw_act_bwd(_p(dout), _p(out), _p(dx), lddx, _p(dbias), N, HW, C_, ACT[act], scale, _stream()),
          "mde_to_nchw_act_bwd")


def image_residual_fwd(d, rgb, out):
    N, C_, H, W = d.shape
    check(_lib.load().mde_image_residual_fwd(_p(d), _p(rgb), _p(out), N, H * W, C_, _stream()), "mde_image_residual_fwd")


def image_residual_bwd(dout, d, rgb, dd):
    N, C_, H, W = d.shape
    check(_lib.load().mde_image_residual_bwd(_p(dout), _p(d), _p(rgb), _p(dd), N, H * W, C_, _stream()), "mde_image_residual_bwd")


def plane_depth_fwd(x, ldx, out, N, h, w, up, max_depth):
    check(_lib.load().mde_plane_depth_fwd(_p(x), ldx, _p(out), N, h, w, up, max_depth, _stream()), "mde_plane_depth_fwd")


def plane_depth_bwd(x, ldx, dout, dx, lddx, N, h, w, up, max_depth):
    check(_lib.load().mde_plane_depth_bwd(_p(x), ldx, _p(dout), _p(dx), lddx, N, h, w, up, max_depth, _stream()), "mde_plane_depth_bwd")


def map_to_slot(src, dst, ld, N, H, W, step=1):
    check(_lib.load().mde_map_to_slot(_p(src), _p(dst), ld, N, H, W, step, _stream()), "mde_map_to_slot")


def slot_to_map_add(dslot, ld, dsrc, N, H, W, step=1):
    check(_lib.load().mde_slot_to_map_add(_p(dslot), ld, _p(dsrc), N, H, W, step, _stream()), "mde_slot_to_map_add")


def pack_grouped(src, fwd, dgrad, O, T, G):
    check(_lib.load().mde_pack_grouped(_p(src), _p(fwd), _p(dgrad), O, T, G, _stream()), "mde_pack_grouped")


# ------------------------------------------------------------------------------ losses / metrics
def silog_ws(device="cuda"):
    return torch.zeros((_lib.load().mde_silog_ws_bytes() + 7) // 8, dtype=torch.float64, device=device)


def silog_fwd(est, gt, variance_focus, ws, loss):
    check(_lib.load().mde_silog_fwd(_p(est), _p(gt), est.numel(), variance_focus, _p(ws), _p(loss), _stream()),
          "mde_silog_fwd")


def silog_bwd(est, gt, variance_focus, ws, gscale, grad):
    check(_lib.load().mde_silog_bwd(_p(est), _p(gt), est.numel(), variance_focus, _p(ws), _p(gscale), _p(grad),
                                    _stream()), "mde_silog_bwd")


MASKED_KINDS = {"l1": 0, "mse": 1, "berhu": 2}


def masked_loss_ws(device="cuda"):
    return torch.zeros((_lib.load().mde_masked_loss_ws_bytes() + 7) // 8, dtype=torch.float64, device=device)


def masked_loss_fwd(kind, pred, target, ws, loss):
    check(_lib.load().mde_masked_loss_fwd(MASKED_KINDS[kind], _p(pred), _p(target), pred.numel(), _p(ws), _p(loss), _stream()),
          "mde_masked_loss_fwd")


def masked_loss_bwd(kind, pred, target, ws, gscale, grad):
    check(_lib.load().mde_masked_loss_bwd(MASKED_KINDS[kind], _p(pred), _p(target), pred.numel(), _p(ws), _p(gscale), _p(grad),
                                          _stream()), "mde_masked_loss_bwd")


def masked_depth_ws(N, device="cuda"):
    return torch.zeros((_lib.load().mde_masked_depth_ws_bytes(N) + 7) // 8, dtype=torch.float64, device=device)


def masked_depth_fwd(pred, target, N, H, W, ws, loss):
    check(_lib.load().mde_masked_depth_fwd(_p(pred), _p(target), N, H, W, _p(ws), _p(loss), _stream()), "mde_masked_depth_fwd")


def masked_depth_bwd(pred, target, N, H, W, ws, gscale, grad):
    check(_lib.load().mde_masked_depth_bwd(_p(pred), _p(target), N, H, W, _p(ws), _p(gscale), _p(grad), _stream()),
          "mde_masked_depth_bwd")


def midas_ws(N, device="cuda"):
    return torch.zeros((_lib.load().mde_midas_ws_bytes(N) + 7) // 8, dtype=torch.float64, device=device)


def midas_fwd(pred, target, N, H, W, ssi, data_kind, data_weight, alpha, scales, batch_based, ws, loss):
    check(_lib.load().mde_midas_fwd(_p(pred), _p(target), N, H, W, int(ssi), data_kind, data_weight, alpha, scales,
                                    int(batch_based), _p(ws), _p(loss), _stream()), "mde_midas_fwd")


def midas_bwd(pred, target, N, H, W, ssi, data_kind, scales, ws, gscale, grad):
    check(_lib.load().mde_midas_bwd(_p(pred), _p(target), N, H, W, int(ssi), data_kind, scales, _p(ws), _p(gscale), _p(grad),
                                    _stream()), "mde_midas_bwd")


def procrustes_ws(N, device="cuda"):
    return torch.zeros((_lib.load().mde_procrustes_ws_bytes(N) + 7) // 8, dtype=torch.float64, device=device)


def procrustes_fwd(pred, target, N, H, W, alpha, scales, batch_based, ws, pred_n, target_n, loss):
    check(_lib.load().mde_procrustes_fwd(_p(pred), _p(target), N, H, W, alpha, scales, int(batch_based), _p(ws), _p(pred_n),
                                         _p(target_n), _p(loss), _stream()), "mde_procrustes_fwd")


def procrustes_bwd(pred, target, N, H, W, scales, ws, pred_n, target_n, gscale, gtmp, grad):
    check(_lib.load().mde_procrustes_bwd(_p(pred), _p(target), N, H, W, scales, _p(ws), _p(pred_n), _p(target_n), _p(gscale),
                                         _p(gtmp), _p(grad), _stream()), "mde_procrustes_bwd")


def scale_and_shift(pred, target, N, H, W, ws, scale, shift):
    check(_lib.load().mde_scale_and_shift(_p(pred), _p(target), N, H, W, _p(ws), _p(scale), _p(shift), _stream()),
          "mde_scale_and_shift")


# ------------------------------------------------------------------------------ VNL configuration criteria
def wcel_ws(C, device="cuda"):
    return torch.zeros((_lib.load().mde_wcel_ws_bytes(C) + 7) // 8, dtype=torch.float64, device=device)


def wcel_fwd(logit, bins, gt, weight, N, C, HW, ws, lse, loss):
    check(_lib.load().mde_wcel_fwd(_p(logit), _p(bins), _p(gt), _p(weight), N, C, HW, _p(ws), _p(lse), _p(loss), _stream()),
          "mde_wcel_fwd")


def wcel_bwd(logit, bins, weight, N, C, HW, ws, lse, gscale, grad):
    check(_lib.load().mde_wcel_bwd(_p(logit), _p(bins), _p(weight), N, C, HW, _p(ws), _p(lse), _p(gscale), _p(grad),
                                   _stream()), "mde_wcel_bwd")


def bins_to_depth_fwd(prob, border, N, C, HW, depth):
    check(_lib.load().mde_bins_to_depth_fwd(_p(prob), _p(border), N, C, HW, _p(depth), _stream()), "mde_bins_to_depth_fwd")


def bins_to_depth_bwd(depth, gdepth, border, N, C, HW, gprob):
    check(_lib.load().mde_bins_to_depth_bwd(_p(depth), _p(gdepth), _p(border), N, C, HW, _p(gprob), _stream()),
          "mde_bins_to_depth_bwd")


def depth_to_bins(depth, depth_min, depth_max, depth_min_log, interval, C, bins):
    check(_lib.load().mde_depth_to_bins(_p(depth), depth.numel(), depth_min, depth_max, depth_min_log, interval, C, _p(bins),
                                        _stream()), "mde_depth_to_bins")


def vnl_ws(B, n, device="cuda"):
    return torch.zeros((_lib.load().mde_vnl_ws_bytes(B, n) + 7) // 8, dtype=torch.float64, device=device)


def vnl_fwd(gt, pred, p123, B, H, W, n, fx, fy, select, ws, loss):
    check(_lib.load().mde_vnl_fwd(_p(gt), _p(pred), _p(p123), B, H, W, n, fx, fy, int(select), _p(ws), _p(loss), _stream()),
          "mde_vnl_fwd")


def vnl_bwd(gt, pred, p123, B, H, W, n, fx, fy, ws, gscale, grad):
    check(_lib.load().mde_vnl_bwd(_p(gt), _p(pred), _p(p123), B, H, W, n, fx, fy, _p(ws), _p(gscale), _p(grad), _stream()),
          "mde_vnl_bwd")


def vnl_head_depth_fwd(x, ldx, bias, border, P, C_, depth, l10, lse):
    check(_lib.load().mde_vnl_head_depth_fwd(_p(x), ldx, _p(bias), _p(border), P, C_, _p(depth), _p(l10), _p(lse), _stream()), "mde_vnl_head_depth_fwd")


def vnl_head_wcel_fwd(x, ldx, bias, bins, gt, weight, lse, P, C_, ws, loss):
    check(_lib.load().mde_vnl_head_wcel_fwd(_p(x), ldx, _p(bias), _p(bins), _p(gt), _p(weight), _p(lse), P, C_, _p(ws), _p(loss), _stream()),
          "mde_vnl_head_wcel_fwd")


def vnl_head_bwd(x, ldx, bias, bins, weight, ws, gscale, lse, depth, l10, gdepth, border, P, C_, dx, lddx):
    check(_lib.load().mde_vnl_head_bwd(_p(x), ldx, _p(bias), _p(bins), _p(weight), _p(ws), _p(gscale), _p(lse), _p(depth), _p(l10), _p(gdepth),
                                       _p(border), P, C_, _p(dx), lddx, _stream()), "mde_vnl_head_bwd")


# ------------------------------------------------------------------------------ stdepth composite criterion
def stdepth_ws(device="cuda"):
    return torch.zeros((_lib.load().mde_stdepth_ws_bytes() + 7) // 8, dtype=torch.float64, device=device)


def stdepth_scratch(N, C, H, W, terms, device="cuda"):
    n = _lib.load().mde_stdepth_scratch_elems(N, C, H, W, terms)
    return torch.empty(n, dtype=torch.float32, device=device) if n else None


def stdepth_fwd(pred, targ, rgba, N, C, H, W, single, terms, weights, ws, scratch, pred_full, out):
    vf, dw, cw, fw, sw = weights
    check(_lib.load().mde_stdepth_fwd(_p(pred), _p(targ), _p(rgba), N, C, H, W, int(single), terms, vf, dw, cw, fw, sw, _p(ws),
                                      _p(scratch), _p(pred_full), _p(out), _stream()), "mde_stdepth_fwd")


def stdepth_bwd(pred, targ, rgba, N, C, H, W, single, terms, weights, ws, scratch, pred_full, gscale, grad):
    vf, dw, cw, fw, sw = weights
    check(_lib.load().mde_stdepth_bwd(_p(pred), _p(targ), _p(rgba), N, C, H, W, int(single), terms, vf, dw, cw, fw, sw, _p(ws),
                                      _p(scratch), _p(pred_full), _p(gscale), _p(grad), _stream()), "mde_stdepth_bwd")


def metrics_ws(device="cuda"):
    return torch.zeros((_lib.load().mde_metrics_ws_bytes() + 7) // 8, dtype=torch.float64, device=device)


def depth_metrics(pred, target, ws, out):
    check(_lib.load().mde_depth_metrics(_p(pred), _p(target), pred.numel(), _p(ws), _p(out), _stream()),
          "mde_depth_metrics")


def ssim_metric(pred, target, out):
    """torchmetrics-0.7.3-style SSIM of clamp_min(pred, 1e-7) against target (fp32 [N][C][H][W]) into out[0]."""
    lib = _lib.load()
    ws = torch.zeros((lib.mde_ssim_metric_ws_bytes() + 7) // 8, dtype=torch.float64, device=pred.device)
    check(lib.mde_ssim_metric(_p(pred), _p(target), pred.numel() // (pred.shape[-1] * pred.shape[-2]), pred.shape[-2], pred.shape[-1],
                              _p(ws), _p(out), _stream()), "mde_ssim_metric")


# ------------------------------------------------------------------------------ optimiser plumbing
def fingerprint_state(device="cuda"):
    return torch.zeros((_lib.load().mde_param_fingerprint_state_bytes() + 7) // 8, dtype=torch.int64, device=device)


def param_fingerprint(p, state):
    check(_lib.load().mde_param_fingerprint(_p(p), p.numel(), _p(state), _stream()), "mde_param_fingerprint")


def refresh_if_changed(src, shadow, packed, jobs, njobs, nblocks, state):
    check(_lib.load().mde_refresh_if_changed(_p(src), _p(shadow), _p(packed), _p(jobs), njobs, nblocks, src.numel(), _p(state),
                                             _stream()), "mde_refresh_if_changed")


def adamw_step(p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    check(_lib.load().mde_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, lr, beta1, beta2, eps, weight_decay,
                                     grad_scale, step, _stream()), "mde_adamw_step")


def sgd_step(p, g, buf, p_bf16, n, lr, momentum, weight_decay, grad_scale):
    check(_lib.load().mde_sgd_step(_p(p), _p(g), _p(buf), _p(p_bf16), n, lr, momentum, weight_decay, grad_scale, _stream()),
          "mde_sgd_step")


def adam_step(p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    check(_lib.load().mde_adam_step(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, lr, beta1, beta2, eps, weight_decay,
                                    grad_scale, step, _stream()), "mde_adam_step")


def cast_bf16(src, dst, n=None):
    check(_lib.load().mde_cast_bf16(_p(src), _p(dst), src.numel() if n is None else n, _stream()), "mde_cast_bf16")


def pack_jobs(convs, device):
    """Device job table (mde_pack_job rows) for mde_pack_wt_batch: convs = [(off, O, T, I)]."""
    rows, first = [], 0
    for off, O, T, I in convs:
        rows.append([off, O, T, I, first])
        first += ((I + 31) // 32) * ((O + 31) // 32) * T
    return torch.tensor(rows, dtype=torch.int64, device=device), first


def pack_wt_batch(src, dst, jobs, nblocks):
    check(_lib.load().mde_pack_wt_batch(_p(src), _p(dst), _p(jobs), jobs.shape[0], nblocks, _stream()), "mde_pack_wt_batch")


def pack_split_batch(src, dst, jobs, nblocks, transposed=False):
    check(_lib.load().mde_pack_split_batch(_p(src), _p(dst), _p(jobs), jobs.shape[0], nblocks, int(transposed), _stream()),
          "mde_pack_split_batch")


def pack_grouped_split(src, fwd2, O, T, G):
    check(_lib.load().mde_pack_grouped_split(_p(src), _p(fwd2), O, T, G, _stream()), "mde_pack_grouped_split")


def pack_wt(src, dst, O, T, I):
    check(_lib.load().mde_pack_wt(_p(src), _p(dst), O, T, I, _stream()), "mde_pack_wt")
