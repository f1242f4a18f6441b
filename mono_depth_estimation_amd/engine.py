"""Execution engine of the HIP FCRN path: flat parameter storage + a static plan of kernel
launches for forward and backward of reference network/FCRN.py:351-371 (ResNet-50/101/152
trunk, conv2/bn2, UpProj decoder, conv3, bilinear, sigmoid).

Design:
  * parameters, gradients, Adam moments and BN running statistics live in a few FLAT fp32
    buffers; the nn.Module's Parameters are views into them (conv weights stored OHWI, i.e.
    torch channels_last, so the bf16 shadow copy *is* the packed GEMM weight and the wgrad
    kernel's output *is* the .grad tensor).  One Adam launch and one all-reduce cover
    everything; encoder and decoder are two contiguous ranges (1x / 10x learning rates,
    reference modules/laina.py:51-57).
  * activations are NHWC bf16, allocated once per input shape; the plan is a list of
    pre-built descriptors so a step is a fixed sequence of C-ABI calls on the current stream
    (capturable in a HIP graph).
  * BatchNorm batch statistics come out of the conv epilogue; ReLU and residual adds are
    fused into the BN apply; the zero-insertion Unpool never materialises (four output
    phases of the 5x5 convs); both 5x5 branches of an UpProj module run as one GEMM.
"""
import os

import torch

from . import _lib, ops

_ALIGN = 64
# BatchNorm-backward sums from the epilogue of the input-gradient launch that writes the site's output gradient
# (mde_conv_gemm_bnred) instead of mde_bn_bwd_reduce's pass over gradient and input; MDE_FUSE_BN_RED=0: the separate pass (A/B)
FUSE_BN_RED = os.environ.get("MDE_FUSE_BN_RED", "1") != "0"
FUSE_DRES = FUSE_BN_RED and os.environ.get("MDE_FUSE_DRES", "1") != "0"      # identity shortcuts: see Bottleneck.bwd
_CHECK_FUSED_SUMS = os.environ.get("MDE_FUSE_BN_RED_CHECK", "0") == "1"    # (tests switch it on in-process)
# The one-workgroup BatchNorm finalize kernels folded into the streaming launches that follow them (mde_bn_apply_fin,
# mde_bn_bwd_apply_fin / _apply2_fin).  MEASURED SLOWER and therefore OFF unless MDE_FUSE_BN_FIN=1 (FCRN
# 28.3 -> 30.1 ms per step, BTS 51.0 -> 56.1): what the 128 (FCRN) / ~540 (BTS) five-microsecond launches cost is less than what
# every workgroup of every streaming pass pays for deriving its constants from the 32 partial-sum slots itself -- four dependent
# L2 round trips in front of its first row, and 50-60 more registers for the whole kernel (114 against 62: half the waves per
# SIMD for a pass that lives on memory-level parallelism).  Never in deterministic mode.
FUSE_BN_FIN = os.environ.get("MDE_FUSE_BN_FIN", "0") == "1"
_JOIN_RANGE = tuple(int(v) for v in os.environ["MDE_WGRAD_JOIN"].split(":")) if os.environ.get("MDE_WGRAD_JOIN") else None


def _round_up(n, a=_ALIGN):
    return (n + a - 1) // a * a


class Act:
    """NHWC bf16 activation (or a channel slice of one) with a lazily allocated gradient."""

    def __init__(self, dev, N, H, W, C, parent=None, c0=0):
        self.N, self.H, self.W, self.C = N, H, W, C
        self.parent, self.c0 = parent, c0
        if parent is None:
            self.ld = C
            self.t = torch.empty(N, H, W, C, dtype=ops.ACT_DTYPE, device=dev)
        else:                                             # (c0 is absolute: the channel offset inside the ROOT tensor)
            self.ld = parent.ld
            self.t = parent.t[..., c0 - parent.c0:c0 - parent.c0 + C]
        self._g = None
        self._gw = False         # gradient already written during the current backward (slices share their root's flag)
        self.producer = None     # the ConvBN whose BatchNorm (+ residual, ReLU) writes this activation
        self.reduced = False     # this backward: the producer site's sums have been added by the launch that completed .g
        self.concat_root = False  # TapeEngine.buf(): gradient zeroed and flagged written before every backward

    @property
    def gw(self):
        return self._gw if self.parent is None else self.parent.gw

    @gw.setter
    def gw(self, v):
        if self.parent is None:
            self._gw = v
        else:
            self.parent.gw = v

    @property
    def M(self):
        return self.N * self.H * self.W

    @property
    def nbytes(self):            # bytes addressable from the first element
        root = self
        while root.parent is not None:
            root = root.parent
        return (root.t.numel() - self.c0) * 2

    def root(self):
        a = self
        while a.parent is not None:
            a = a.parent
        return a

    def slice(self, c0, C):
        return Act(None, self.N, self.H, self.W, C, parent=self, c0=self.c0 + c0)

    @property
    def g(self):
        if self._g is None:
            if self.parent is None:
                self._g = torch.empty_like(self.t)
            else:
                self._g = self.parent.g[..., self.c0 - self.parent.c0:self.c0 - self.parent.c0 + self.C]
        return self._g


class BNSite:
    """Per-BatchNorm device state: fp32 views into the flat stores + saved statistics."""

    def __init__(self, eng, C, gamma, beta, g_off, b_off, rmean, rvar, bn, eps):
        dev = eng.dev
        self.C, self.bn, self.eps = C, bn, eps        # momentum is read from the module at run time
        self.store, self.g_off, self.b_off = eng.store, g_off, b_off     # gradient slices: offsets into store.Gcur
        self.gamma, self.beta, self.rmean, self.rvar = gamma, beta, rmean, rvar
        self.scale, self.shift, self.smean, self.srstd = (torch.empty(C, device=dev) for _ in range(4))
        # two partial-sum buffers [slots][2][C]: the forward statistics (conv epilogues / mde_bn_stats add into it) and the backward
        # sums (mde_bn_bwd_reduce / the input-gradient launches' epilogues).  Separate finalize kernels zero the one they read; in
        # the fused forms (FUSE_BN_FIN) each direction's launch zeroes the OTHER direction's buffer (EngineCore.begin_* keep watch)
        self.part = ops.new_stat_buffer(C, dev)
        self.part_b = ops.new_stat_buffer(C, dev)
        self.fpart, self.fpart_ld, self.fzero = self.part, C, self.part     # where the forward sums of THIS site's channels start
        self.coef = torch.empty(3, C, device=dev)
        self._fin, self._bfin = {}, {}
        eng.sites.append(self)

    def half(self, eng, i):
        """View of channels [i*C/2, (i+1)*C/2) of a fused site (own partial buffer for backward)."""
        h = self.C // 2
        s = BNSite.__new__(BNSite)
        s.C, s.bn, s.eps = h, self.bn, self.eps
        s.store, s.g_off, s.b_off = self.store, self.g_off + i * h, self.b_off + i * h
        for k in ("gamma", "beta", "rmean", "rvar", "scale", "shift", "smean", "srstd"):
            setattr(s, k, getattr(self, k)[i * h:(i + 1) * h])
        s.part = None                                   # (the forward statistics are the parent's: the fused conv writes 2C columns)
        s.part_b = ops.new_stat_buffer(h, eng.dev)
        s.fpart, s.fpart_ld, s.fzero = self.part.view(-1)[i * h:], self.C, self.part
        s.coef = torch.empty(3, h, device=eng.dev)
        s._fin, s._bfin = {}, {}
        eng.sites.append(s)
        return s

    @property
    def dgamma(self):
        return self.store.Gcur[self.g_off:self.g_off + self.C]

    @property
    def dbeta(self):
        return self.store.Gcur[self.b_off:self.b_off + self.C]

    def fin(self, M, mean=None, var=None):
        """mde_bn_fin of this site's training-mode forward: its scale / shift come out of the apply launch itself (from the forward
        sums, or from given batch moments: DenseNet), which also zeroes the site's backward sums."""
        key = (M, None if mean is None else mean.data_ptr())
        f = self._fin.get(key)
        mom = self.bn.momentum if self.bn.momentum is not None else 0.1
        if f is None:
            dp = lambda t: t.data_ptr() if t is not None else None
            f = _lib.BnFin(dp(self.fpart) if mean is None else None, self.fpart_ld if mean is None else 0, dp(mean), dp(var), M,
                           dp(self.gamma), dp(self.beta), dp(self.rmean), dp(self.rvar), mom, self.eps, dp(self.scale), dp(self.shift),
                           dp(self.smean), dp(self.srstd), dp(self.part_b), self.part_b.numel())
            self._fin[key] = f
        f.momentum = mom                                   # (read from the module at run time, as finalize does)
        return f

    def bfin(self, M):
        """mde_bn_bfin of this site's backward: coefficients, dgamma / dbeta and the zeroing of the forward sums inside the
        backward apply launch.  (dgamma / dbeta point into the buffer the current backward accumulates into.)"""
        G = self.store.Gcur
        key = (M, G.data_ptr())
        f = self._bfin.get(key)
        if f is None:
            dp = lambda t: t.data_ptr() if t is not None else None
            f = _lib.BnBfin(dp(self.part_b), self.C, M, dp(self.gamma), dp(self.srstd), dp(self.dgamma), dp(self.dbeta), dp(self.fzero),
                            self.fzero.numel())
            f._keep = G
            self._bfin = {key: f}
        return f

    def apply(self, eng, x, ldx, out, ldo, M, relu, train, r=None, ldr=0, res_site=None, relu_bits=None, finalized=False,
              mean=None, var=None):
        """out = [relu](bn(x) [+ r | + bn_res(r)]) with this site's statistics finalized on the way: inside the launch
        (mde_bn_apply_fin) in a training-mode pass, by the separate kernels otherwise (eval; deterministic mode; MDE_FUSE_BN_FIN=0;
        `finalized`: the caller has run them already).  mean / var: batch moments already reduced (graph.PrefixBN)."""
        if train and eng.fin_fused and not finalized:
            ops.bn_apply_fin(x, ldx, self.fin(M, mean, var), out, ldo, M, self.C, relu, r=r, ldr=ldr,
                             fin_r=res_site.fin(M) if res_site is not None else None, relu_bits=relu_bits)
            return
        if not finalized:
            if mean is not None:
                self.finalize_moments(mean, var, M, train)
            else:
                self.finalize(M, train)
            if res_site is not None:
                res_site.finalize(M, train)
        ops.bn_apply(x, ldx, self.scale, self.shift, out, ldo, M, self.C, relu, r=r, ldr=ldr,
                     rscale=res_site.scale if res_site is not None else None, rshift=res_site.shift if res_site is not None else None,
                     relu_bits=relu_bits)

    def finalize(self, M, train):
        if train:
            mom = self.bn.momentum if self.bn.momentum is not None else 0.1
            assert self.part is not None, "a half site's forward statistics are finalized through its parent (or inside its apply launch)"
            ops.bn_finalize(self.part, M, self.C, self.gamma, self.beta, self.rmean, self.rvar, mom, self.eps,
                            self.scale, self.shift, self.smean, self.srstd)
        else:
            ops.bn_eval_scale_shift(self.gamma, self.beta, self.rmean, self.rvar, self.eps, self.C, self.scale, self.shift)

    def finalize_moments(self, mean, var, M, train):
        """As finalize, from batch moments another site already reduced (DenseNet: graph.PrefixBN)."""
        if train:
            mom = self.bn.momentum if self.bn.momentum is not None else 0.1
            ops.bn_finalize_moments(mean, var, M, self.C, self.gamma, self.beta, self.rmean, self.rvar, mom, self.eps,
                                    self.scale, self.shift, self.smean, self.srstd)
        else:
            ops.bn_eval_scale_shift(self.gamma, self.beta, self.rmean, self.rvar, self.eps, self.C, self.scale, self.shift)

    def backward(self, dout, out, x, relu, dx, accumulate=False, dres=None, mask_from_x=False, relu_bits=None, reduced=False):
        """g = dout*(out>0 if relu); writes dgamma/dbeta (+=), dx (bf16) and optionally dres = g.
        mask_from_x: out == relu(x*scale+shift) exactly (no residual), so the mask is recomputed
        from x with this site's scale/shift and `out` is never read.
        reduced: the sums are in `part` already (red_spec: added by the launch that wrote dout)."""
        M, C = x.M, self.C
        ms, mh = (self.scale, self.shift) if (relu and mask_from_x) else (None, None)
        if not reduced:
            ops.bn_bwd_reduce(dout, _ld(dout, out), out.t if out is not None else None, out.ld if out is not None else 0,
                              x.t, x.ld, self.smean, self.srstd, M, C, relu, self.part_b, ms, mh, relu_bits)
        elif _CHECK_FUSED_SUMS:
            # diagnostics (MDE_FUSE_BN_RED_CHECK=1): the sums that came with the conv launch against the reduction pass
            tmp = torch.zeros_like(self.part_b)
            ops.bn_bwd_reduce(dout, _ld(dout, out), out.t if out is not None else None, out.ld if out is not None else 0,
                              x.t, x.ld, self.smean, self.srstd, M, C, relu, tmp, ms, mh, relu_bits)
            _check_fused_sums(self.part_b, tmp, "M=%d C=%d x.ld=%d relu=%s mask_from_x=%s bits=%s" % (M, C, x.ld, relu, mask_from_x,
                                                                                                   relu_bits is not None))
        if self.store.fin_fused:
            ops.bn_bwd_apply_fin(dout, _ld(dout, out), out.t if out is not None else None, out.ld if out is not None else 0,
                                 x.t, x.ld, self.smean, self.srstd, self.bfin(M), M, C, relu, dx, _ld(dx, x), accumulate,
                                 dres, _ld(dres, x) if dres is not None else 0, ms, mh, relu_bits)
            return
        ops.bn_bwd_finalize(self.part_b, M, C, self.gamma, self.srstd, self.dgamma, self.dbeta, self.coef)
        ops.bn_bwd_apply(dout, _ld(dout, out), out.t if out is not None else None, out.ld if out is not None else 0,
                         x.t, x.ld, self.smean, self.srstd, self.coef, M, C, relu, dx, _ld(dx, x), accumulate,
                         dres, _ld(dres, x) if dres is not None else 0, ms, mh, relu_bits)


    def red_spec(self, x, relu, mask_from_x=False, relu_bits=None):
        """What an input-gradient launch needs to add this site's backward sums from its epilogue (ops.conv_gemm(red=)):
        the arguments of backward() that decide the mask.  None where that launch cannot (a mask read from `out`)."""
        if not FUSE_BN_RED or self.C % 8 or (relu and not mask_from_x and relu_bits is None):
            return None
        ms, mh = (self.scale, self.shift) if (relu and mask_from_x) else (None, None)
        return ops.bn_red(x.t, self.smean, self.srstd, self.part_b, ms, mh, relu_bits if (relu and not mask_from_x) else None, x_ld=x.ld)


FUSED_SUM_CHECKS = []      # diagnostics: (site description, largest relative difference) per checked site and backward


def _check_fused_sums(part, ref_part, what):
    """|fused - reduction pass| per (sum, channel) relative to the reduction pass's value + 1e-3 of its largest."""
    a, b = part.double().sum(0), ref_part.double().sum(0)
    scale = b.abs() + 1e-3 * b.abs().amax(dim=1, keepdim=True) + 1e-30
    FUSED_SUM_CHECKS.append((what, float(((a - b).abs() / scale).max())))


def bn_join_backward(sa, sb, dout, out, xa, xb, dxa, dxb, relu_bits):
    """Backward of out = relu(bn_a(xa) + bn_b(xb)) for both sites at once: dout and the mask are read once
    per pass (ops.bn_bwd_reduce2 / bn_bwd_apply2) instead of once per site.  `out.reduced`: the sums of both sites came with
    the launch that completed dout (ConvBN.red_spec)."""
    M, C = xa.M, sa.C
    ldd = _ld(dout, out)
    reduced, out.reduced = out.reduced, False
    if not reduced:
        ops.bn_bwd_reduce2(dout, ldd, xa.t, xa.ld, xb.t, xb.ld, sa.smean, sa.srstd, sb.smean, sb.srstd, relu_bits, M, C,
                           sa.part_b, sb.part_b)
    elif _CHECK_FUSED_SUMS:
        ta, tb = torch.zeros_like(sa.part_b), torch.zeros_like(sb.part_b)
        ops.bn_bwd_reduce2(dout, ldd, xa.t, xa.ld, xb.t, xb.ld, sa.smean, sa.srstd, sb.smean, sb.srstd, relu_bits, M, C, ta, tb)
        _check_fused_sums(sa.part_b, ta, "join a: M=%d C=%d" % (M, C))
        _check_fused_sums(sb.part_b, tb, "join b: M=%d C=%d xb.ld=%d" % (M, C, xb.ld))
    if sa.store.fin_fused:
        ops.bn_bwd_apply2_fin(dout, ldd, xa.t, xa.ld, xb.t, xb.ld, sa.smean, sa.srstd, sb.smean, sb.srstd, relu_bits, sa.bfin(M), sb.bfin(M),
                              M, C, dxa, _ld(dxa, xa), dxb, _ld(dxb, xb))
        return
    ops.bn_bwd_finalize(sa.part_b, M, C, sa.gamma, sa.srstd, sa.dgamma, sa.dbeta, sa.coef)
    ops.bn_bwd_finalize(sb.part_b, M, C, sb.gamma, sb.srstd, sb.dgamma, sb.dbeta, sb.coef)
    ops.bn_bwd_apply2(dout, ldd, xa.t, xa.ld, xb.t, xb.ld, sa.smean, sa.srstd, sb.smean, sb.srstd, relu_bits, sa.coef,
                      sb.coef, M, C, dxa, _ld(dxa, xa), dxb, _ld(dxb, xb))


def _ld(t, like):
    """Pixel stride (elements) of a gradient/activation tensor view."""
    return t.stride(2) if t is not None and t.dim() == 4 else like.ld


class Conv:
    """A conv parameter: fp32 master [O][T][I] slice of P, bf16 shadow slice, grad slice, and
    (when its input gradient is needed) the transposed bf16 packing [I][T][O]."""

    def __init__(self, store, off, O, T, I, need_dgrad=True):
        n = O * T * I
        self.O, self.T, self.I = O, T, I
        self.store, self.off, self.n = store, off, n
        self.w32 = store.P[off:off + n]
        self.wf = store.Pb[off:off + n]
        self.wd = store.WD[off:off + n] if need_dgrad else None    # slice of the flat transposed-weight buffer
        self.fwd_transposed = False

    @property
    def dw(self):                         # gradient slice of the buffer the current backward accumulates into
        return self.store.Gcur[self.off:self.off + self.n]

    # the eval-mode two-term operands (FlatStore.ensure_split): [O][2T][I] and, for the layers whose FORWARD contracts with the
    # transposed packing (ConvTranspose2d: DeConvLayer, graph.ConvT), [I][2T][O]
    @property
    def w2(self):
        return self.store.W2[2 * self.off:2 * (self.off + self.n)]

    @property
    def w2d(self):
        return self.store.W2D[2 * self.off:2 * (self.off + self.n)]

    def want_w2d(self):
        if not self.fwd_transposed:
            self.fwd_transposed = True
            self.store._split_jobs = None


class GroupedConv:
    """A grouped conv parameter (reference VNL.py:638, MiDaS' resnext101_32x8d): fp32 master [O][T][G] slice of P (G input
    channels per group) and its two block-diagonal bf16 packings [O][T][64] — forward and transposed (input gradient) —
    in buffers of their own, rewritten by mde_pack_grouped whenever the masters change."""

    def __init__(self, store, off, O, T, G):
        self.O, self.T, self.I, self.G = O, T, 64, G
        self.store, self.off, self.n = store, off, O * T * G
        self.w32 = store.P[off:off + self.n]
        self.wf = torch.zeros(O * T * 64, dtype=ops.ACT_DTYPE, device=store.dev)
        self.wd = torch.zeros(O * T * 64, dtype=ops.ACT_DTYPE, device=store.dev)
        self.w2 = None                  # eval-mode two-term operand [O][2T][64], built on first use (pack_split)

    @property
    def dw(self):
        return self.store.Gcur[self.off:self.off + self.n]

    def pack(self):
        ops.pack_grouped(self.w32, self.wf, self.wd, self.O, self.T, self.G)

    def pack_split(self):
        if self.w2 is None:
            self.w2 = torch.zeros(2 * self.O * self.T * 64, dtype=ops.ACT_DTYPE, device=self.store.dev)
        ops.pack_grouped_split(self.w32, self.w2, self.O, self.T, self.G)


class FlatStore:
    """Flat fp32 parameter / gradient / BN-buffer storage of one network module (created once; every per-shape plan
    shares it).  Re-points the module's Parameters and buffers at views.  Subclasses say how the module's tensors
    are laid out (`_layout`): the order (encoder range first: the two learning-rate groups every reference module
    configures), which tensors are stored back to back as one fused GEMM operand, and the channel padding."""

    pad_to = 8                    # stored channel counts are rounded up to this (the GEMM kernels take C % 8 == 0)

    def __init__(self, module, device):
        self.m, self.dev = module, device
        self.convs = {}           # id(first weight) -> Conv (shared packings)
        self.grouped = []         # GroupedConv objects (their block-diagonal packings follow every weight update)
        self.packed_version = -1
        self.adam_state = None
        self.sgd_state = None
        self._fp_state, self._fp_valid = None, False     # device fingerprint of P (refresh_weights(check_data=True))
        self.step_count = 0
        # eval-mode forward passes contract with a TWO-TERM weight shadow, bf16(w) + bf16(w - bf16(w)) (ensure_split; north_star:
        # "AbsRel within 1e-4 of CPU reference on identical weights" for weights that are not on the 16-bit grid); MDE_EVAL_SPLIT=0:
        # the one-term shadow the training step uses (A/B, tests)
        self.split_eval = os.environ.get("MDE_EVAL_SPLIT", "1") != "0"
        self.W2 = self.W2D = None
        self._split_jobs, self._split_epoch, self.shadow_epoch = None, -1, 0
        self.grad_reducer = None          # module path: a dp.FlatGradReducer fed by backward (TapeModule.set_grad_reducer)
        self._flatten_parameters()
        self.deterministic, self._det_scratch = False, None
        if os.environ.get("MDE_DETERMINISTIC", "0") == "1":
            self.set_deterministic(True)

    # ---- opt-in deterministic mode (include/mde_hip.h: mde_set_deterministic): bit-reproducible training steps
    def set_deterministic(self, on=True):
        """Every cross-workgroup sum of a step (BatchNorm statistics, split-K weight gradients, bias gradients) becomes an
        order-independent integer accumulation: two runs from the same state give bit-identical gradients.  Costs one
        extra pass over the gradient buffer per step and 16 bytes of scratch per parameter; the gradient exchange of
        dp.FlatGradReducer then starts after backward instead of overlapping it.  Process-wide kernel state: switch it
        between steps, and use it on one module at a time."""
        self.deterministic = bool(on)
        if not on:
            ops.set_deterministic(False)

    @property
    def fin_fused(self):
        """The BatchNorm finalize work runs inside the streaming launches (FUSE_BN_FIN; never in deterministic mode, whose partial
        sums are integers)."""
        return FUSE_BN_FIN and not self.deterministic

    def det_begin(self):
        """Program the library's (process-wide) accumulation mode for THIS store's forward / backward: the integer gradient
        shadow pointed at the buffer this backward accumulates into, or the mode switched OFF -- a store that is not
        deterministic must not run under a mode (and a gradient base) another store left behind."""
        if self.deterministic:
            if self._det_scratch is None:
                self._det_scratch = ops.det_scratch(self.G)
            ops.set_deterministic(True, self.Gcur, self._det_scratch)
        else:
            ops.set_deterministic(False)

    def det_end(self):
        if self.deterministic:
            ops.det_flush()

    def _layout(self):
        """-> (plist, blist, n_encoder_entries, no_pad).  plist: [(kind, [tensors])] in flat order, a fused entry lists
        several tensors back to back; blist: [[buffers]]; no_pad: {id(weight): (pad O?, pad I?)} exceptions."""
        raise NotImplementedError

    def _flatten_parameters(self):
        m, dev = self.m, self.dev
        plist, blist, self.n_encoder_entries, no_pad = self._layout()
        padc = lambda c: (c + self.pad_to - 1) // self.pad_to * self.pad_to
        pad64 = padc

        # Storage shape of every parameter.  The GEMM kernels take channel counts that are multiples of 64; the
        # decoders of the ResNet-18/34 variants end in 32- and 16-channel layers (FCRN.py:329-349 with
        # num_channels = 512), so such tensors are stored zero-padded to 64 channels ([Op][kh][kw][Ip], BN vectors
        # [Cp]) and the Parameter is the strided view of the real entries.  Padded entries stay exactly zero: their
        # activations, gradients and Adam moments are all zero.  Nothing is padded for the 50/101/152 networks.
        self.sdims = {}
        for _, ts in plist:
            for t in ts:
                if t.dim() == 4:
                    po, pi = no_pad.get(id(t), (True, True))
                    O, I, kh, kw = t.shape
                    self.sdims[id(t)] = (pad64(O) if po else O, kh, kw, pad64(I) if pi else I)
                else:
                    self.sdims[id(t)] = (pad64(t.numel()),)
        snumel = lambda t: int(torch.Size(self.sdims[id(t)]).numel())
        offs, size = [], 0
        for i, (_, ts) in enumerate(plist):
            if i == self.n_encoder_entries:
                self.encoder_numel = size
            offs.append(size)
            size = _round_up(size + sum(snumel(t) for t in ts))
        if self.n_encoder_entries >= len(plist):
            self.encoder_numel = size
        self.P = torch.zeros(size, dtype=torch.float32, device=dev)
        self.G = torch.zeros(size, dtype=torch.float32, device=dev)
        self.G2 = None                    # second gradient buffer, only for the autograd path (see begin_autograd_backward)
        self.Gcur = self.G                # where the engine's backward accumulates parameter gradients
        self.Pb = torch.zeros(size, dtype=ops.ACT_DTYPE, device=dev)
        self.WD = torch.zeros(size, dtype=ops.ACT_DTYPE, device=dev)     # transposed conv weights, same offsets as P
        self._pack_jobs = None
        self.p_off = {}
        for (kind, ts), off in zip(plist, offs):
            o = off
            for t in ts:
                self.p_off[id(t)] = o
                view, gview = self.view_of(self.P, t), self.view_of(self.G, t)       # OIHW view of OHWI storage
                with torch.no_grad():
                    view.copy_(t.detach().to(dev))
                t.data = view
                t._mde_grad = gview
                o += snumel(t)
        bsize, boffs = 0, []
        for ts in blist:
            boffs.append(bsize)
            bsize = _round_up(bsize + sum(pad64(t.numel()) for t in ts))
        self.B = torch.zeros(bsize, dtype=torch.float32, device=dev)
        self.b_off = {}
        for ts, off in zip(blist, boffs):
            o = off
            for t in ts:
                n = t.numel()
                view = self.B[o:o + n]
                view.copy_(t.detach().to(dev))
                t.data = view
                self.b_off[id(t)] = o
                o += pad64(n)
        # one shared int64 counter vector for every BN's num_batches_tracked
        bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
        self.nbt = torch.zeros(len(bns), dtype=torch.int64, device=dev)
        for i, bn in enumerate(bns):
            self.nbt[i] = int(bn.num_batches_tracked)
            bn.num_batches_tracked.data = self.nbt[i]
        self.params = [p for p in m.parameters()]
        self._g_base, self._g2_base = self._uses(self.G), None      # storage use counts with no outside views

    def attach_grads(self):
        """Make the .grad of every Parameter that requires grad the matching view of the flat gradient buffer
        (frozen parameters keep .grad = None, as under torch autograd; their slice of the buffer is scratch).
        Returns True if the buffer had to be (re)attached and zeroed."""
        live = [p for p in self.params if p.requires_grad]
        fresh = any(p.grad is None or p.grad.data_ptr() != p._mde_grad.data_ptr() for p in live)
        if fresh:
            self.G.zero_()
            for p in live:
                p.grad = p._mde_grad
        return fresh

    # ---- gradients handed to torch autograd (module path).  The engine's direct path (bench.py, eng.backward)
    # accumulates into self.G and that is all.  Through autograd the Function must RETURN gradient tensors, otherwise
    # nothing flows through AccumulateGrad and wrappers that hook it (DistributedDataParallel) never see gradients.
    def _in(self, t, buf):
        return buf is not None and buf.data_ptr() <= t.data_ptr() < buf.data_ptr() + buf.numel() * 4

    def grad_buffer(self):
        """The flat buffer the Parameters' .grad tensors live in (G unless autograd adopted views of G2)."""
        for p in self.params:
            if p.requires_grad and p.grad is not None:
                return self.G2 if self._in(p.grad, self.G2) else self.G
        return self.G

    @staticmethod
    def _uses(buf):
        """Number of live tensors sharing buf's storage (every view counts), or None if torch does not tell."""
        try:
            return torch._C._storage_Use_Count(buf.untyped_storage()._cdata)
        except Exception:
            return None

    def begin_autograd_backward(self):
        """Pick and zero the buffer this backward writes: one that holds neither the live .grad tensors (so that
        autograd can add the returned views onto them — accumulation — or adopt them without a copy when .grad is
        None) nor views somebody else still owns: the gradients torch.autograd.grad() returned from an earlier
        backward are views of a flat buffer too, and zeroing it under them would silently change those tensors.
        The storage's use count says whether such views exist; a buffer still referenced is left to its owners."""
        held = None
        for p in self.params:
            if p.requires_grad and p.grad is not None:
                held = p.grad
                break

        def free(buf, base):
            if buf is None or (held is not None and self._in(held, buf)):
                return False
            u = self._uses(buf)
            return u is not None and u <= base
        if self._g_base is None:                      # no use counts: the old rule (and backward returns copies)
            target = self.G
            if held is not None and self._in(held, self.G):
                if self.G2 is None:
                    self.G2 = torch.zeros_like(self.G)
                target = self.G2
        elif free(self.G, self._g_base):
            target = self.G
        else:
            if not free(self.G2, self._g2_base):
                self.G2 = torch.zeros_like(self.G)  # the previous G2 (if any) lives on with whoever references it
                self._g2_base = self._uses(self.G2)
            target = self.G2
        target.zero_()
        self.Gcur = target
        return target

    def view_of(self, flat, p):
        """p's entries inside a flat buffer laid out like P (its storage may be zero-padded to 64 channels)."""
        off, sd = self.p_off[id(p)], self.sdims[id(p)]
        n = int(torch.Size(sd).numel())
        if p.dim() == 4:
            O, I, kh, kw = p.shape
            return flat[off:off + n].view(sd)[:O, :, :, :I].permute(0, 3, 1, 2)
        return flat[off:off + p.numel()].view(p.shape)

    def grad_view(self, p, buf):
        """A FRESH tensor (nobody else references it: autograd may adopt it as .grad) viewing p's slice of buf."""
        return self.view_of(buf, p)

    def storage_is_current(self):
        p = self.params[0]
        return p.data_ptr() == self.view_of(self.P, p).data_ptr()

    def conv(self, weights, need_dgrad=True):
        t0 = weights[0]
        c = self.convs.get(id(t0))
        if c is None:
            O = sum(self.sdims[id(t)][0] for t in weights)            # storage (possibly padded) channel counts
            _, kh, kw, I = self.sdims[id(t0)]
            c = Conv(self, self.p_off[id(t0)], O, kh * kw, I, need_dgrad)
            self.convs[id(t0)] = c
            self._pack_jobs = self._split_jobs = None
        return c

    def linear(self, w, need_dgrad=True):
        """An nn.Linear weight [O][I] as the 1x1 convolution it is: its natural storage is already the GEMM operand
        [O][1][I] (Dorn.py:64 global_fc; O and I multiples of 8, so nothing is padded)."""
        c = self.convs.get(id(w))
        if c is None:
            O, I = w.shape
            assert w.dim() == 2 and O % 8 == 0 and I % 8 == 0 and self.sdims[id(w)] == (O * I,), (tuple(w.shape), self.sdims[id(w)])
            c = Conv(self, self.p_off[id(w)], O, 1, I, need_dgrad)
            self.convs[id(w)] = c
            self._pack_jobs = self._split_jobs = None
        return c

    def layer_boundaries(self):
        """Flat offsets at which conv weights start (forward order): where dp.FlatGradReducer may cut its buckets."""
        return sorted({0} | {self.p_off[id(p)] for p in self.params if p.dim() == 4})

    def conv_grouped(self, w, G):
        c = self.convs.get(id(w))
        if c is None:
            O, kh, kw, I = self.sdims[id(w)]
            assert I == G and O % 64 == 0 and 64 % G == 0, (O, I, G)
            c = GroupedConv(self, self.p_off[id(w)], O, kh * kw, G)
            self.convs[id(w)] = c
            self.grouped.append(c)
            self.packed_version = -1           # its packings do not exist yet
            self._split_jobs = None
        return c

    def params_version(self):
        """Changes whenever torch modified a Parameter in place (optimizer.step, load_state_dict, p.data.copy_ ...).
        The Parameters are views of the flat buffer P, but an in-place op on such a view bumps the PARAMETER's
        version counter, not P's (``p.data = view`` keeps the parameter's own counter): P._version never moves."""
        return sum(p._version for p in self.params) + self.P._version

    def refresh_weights(self, force=False, check_data=False):
        """bf16 shadow + transposed packings follow the fp32 masters (after any in-place update).
        check_data (the nn.Module path): also catch writes torch's version counters do not record — `p.data.mul_()`,
        the reference's own `weights_init` (`m.weight.data.normal_`), a collective into `p.data`: a device-side
        fingerprint of the flat masters gates the same two kernels, without a host synchronisation (one extra read
        of the 254 MB range per forward, ~0.07 ms)."""
        v = self.params_version()
        if force or v != self.packed_version:
            ops.cast_bf16(self.P, self.Pb)
            self.repack()
            self.packed_version = v
            self.shadow_epoch += 1
            self._fp_valid = False
            if check_data:
                self._record_fingerprint()     # the shadows are fresh NOW: a .data write before the next forward must show
        elif check_data:
            if self._fp_state is None:
                self._fp_state = ops.fingerprint_state(self.dev)
            ops.param_fingerprint(self.P, self._fp_state)
            if self._fp_valid:
                jobs, nblocks = self._jobs()
                ops.refresh_if_changed(self.P, self.Pb, self.WD if jobs is not None else None, jobs,
                                       jobs.shape[0] if jobs is not None else 0, nblocks, self._fp_state)
                for g in self.grouped:         # (not gated by the device flag: a few small launches per forward)
                    g.pack()
            self._fp_valid = True          # (a first call only records the fingerprint: the shadows are known fresh)

    def _record_fingerprint(self):
        """Fingerprint of the masters the shadows were just derived from (module path only: one 254 MB read)."""
        if self._fp_state is None:
            self._fp_state = ops.fingerprint_state(self.dev)
        ops.param_fingerprint(self.P, self._fp_state)
        self._fp_valid = True

    def _jobs(self):
        if self._pack_jobs is None:
            todo = sorted((c.off, c.O, c.T, c.I) for c in self.convs.values() if isinstance(c, Conv) and c.wd is not None)
            self._pack_jobs = ops.pack_jobs(todo, self.dev) if todo else (None, 0)
        return self._pack_jobs

    def repack(self):
        """Transposed ("dgrad") packings of every conv weight that needs one, in a single launch."""
        jobs, nblocks = self._jobs()
        if jobs is not None:
            ops.pack_wt_batch(self.P, self.WD, jobs, nblocks)
        for g in self.grouped:
            g.pack()

    def _step_ranges(self):
        """Contiguous flat ranges the fused optimiser steps may touch: [(begin, end, is_decoder)].  Parameters with
        requires_grad=False are left out (torch.optim skips them: the reference's get_*_lr_params filter on
        requires_grad, laina.py:51-57) — the engine's backward writes a gradient for every parameter, frozen or
        not, and a fused step over the whole range used to move frozen weights.  All trainable: the two ranges."""
        key = tuple(p.requires_grad for p in self.params)
        if getattr(self, "_ranges_key", None) != key:
            e, n = self.encoder_numel, self.P.numel()
            if all(key):
                self._ranges = [(0, e, False), (e, n, True)]
            else:
                spans = sorted((self.p_off[id(p)], self.p_off[id(p)] + int(torch.Size(self.sdims[id(p)]).numel()),
                                p.requires_grad) for p in self.params)
                runs = []
                for b, en, live in spans:
                    dec = b >= e
                    if live and runs and runs[-1][3] and runs[-1][2] == dec:
                        runs[-1][1] = en                   # extend the current run over the alignment gap
                    elif live:
                        runs.append([b, en, dec, True])
                    elif runs:
                        runs[-1][3] = False                # a frozen tensor ends the run
                self._ranges = [(b, en, dec) for b, en, dec, _ in runs]
            self._ranges_key = key
        return self._ranges

    def _sync_external_grads(self, G):
        """Gradients torch holds OUTSIDE the flat buffer G must be copied in before a fused step reads G:
        AccumulateGrad clones a returned gradient view when its strides do not match the parameter's (the
        zero-padded ResNet-18/34 decoder weights), and gradient accumulation / zero_grad(set_to_none=False) / DDP
        then act on the clone, not on G."""
        with torch.no_grad():
            for p in self.params:
                if p.requires_grad and p.grad is not None and not self._in(p.grad, G):
                    self.view_of(G, p).copy_(p.grad)

    # fused Adam over the two flat ranges (encoder 1x LR, decoder 10x LR: modules/laina.py:51-57)
    def adam_step(self, lr_encoder, lr_decoder, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0,
                  decoupled=False):
        """decoupled=True: torch.optim.AdamW (modules/bts.py:139-152), weight_decay a scalar or an (encoder, decoder)
        pair as BTS configures it (1e-2 / 0)."""
        if self.adam_state is None:
            self.adam_state = (torch.zeros_like(self.P), torch.zeros_like(self.P))
        mom, var = self.adam_state
        self.step_count += 1
        G = self.grad_buffer()
        self._sync_external_grads(G)
        wd = weight_decay if isinstance(weight_decay, (tuple, list)) else (weight_decay, weight_decay)
        step = ops.adamw_step if decoupled else ops.adam_step
        for b, en, dec in self._step_ranges():
            step(self.P[b:en], G[b:en], mom[b:en], var[b:en], self.Pb[b:en], en - b, lr_decoder if dec else lr_encoder,
                 betas[0], betas[1], eps, wd[1] if dec else wd[0], grad_scale, self.step_count)
        self._after_fused_step()

    def ensure_split(self, always=False):
        """The eval-mode two-term operands of every conv weight follow the fp32 masters (mde_pack_split_batch: one launch per
        layout over a device job table; grouped weights one small launch each).  Re-derived when the one-term shadows were
        (`shadow_epoch`), or on every call (`always`: the nn.Module path, where a `.data` write is only known to the device)."""
        if not self.split_eval:
            return False
        if self._split_jobs is None:
            fw = sorted((c.off, c.O, c.T, c.I) for c in self.convs.values() if isinstance(c, Conv))
            tr = sorted((c.off, c.O, c.T, c.I) for c in self.convs.values() if isinstance(c, Conv) and c.fwd_transposed)
            self._split_jobs = (ops.pack_jobs(fw, self.dev) if fw else (None, 0), ops.pack_jobs(tr, self.dev) if tr else (None, 0))
            self._split_epoch = -1
        if always or self._split_epoch != self.shadow_epoch:
            (jf, nf), (jt, nt) = self._split_jobs
            if jf is not None:
                if self.W2 is None:
                    self.W2 = torch.zeros(2 * self.P.numel(), dtype=ops.ACT_DTYPE, device=self.dev)
                ops.pack_split_batch(self.P, self.W2, jf, nf)
            if jt is not None:
                if self.W2D is None:
                    self.W2D = torch.zeros(2 * self.P.numel(), dtype=ops.ACT_DTYPE, device=self.dev)
                ops.pack_split_batch(self.P, self.W2D, jt, nt, transposed=True)
            for g in self.grouped:
                g.pack_split()
            self._split_epoch = self.shadow_epoch
        return True

    def _after_fused_step(self):
        self.shadow_epoch += 1
        self.repack()
        self.packed_version = self.params_version()   # shadow + packings are current (the kernels bump no version counter)
        self._fp_valid = False
        if self._fp_state is not None:                 # module path in use: keep the .data-write detector armed
            self._record_fingerprint()

    # fused SGD with momentum over the same two ranges (modules/vnl.py:289-326: momentum 0.9, weight_decay 5e-4)
    def sgd_step(self, lr_encoder, lr_decoder, momentum=0.9, weight_decay=0.0, grad_scale=1.0):
        if self.sgd_state is None:
            self.sgd_state = torch.zeros_like(self.P)
        G, buf = self.grad_buffer(), self.sgd_state
        self._sync_external_grads(G)
        for b, en, dec in self._step_ranges():
            ops.sgd_step(self.P[b:en], G[b:en], buf[b:en], self.Pb[b:en], en - b, lr_decoder if dec else lr_encoder, momentum,
                         weight_decay, grad_scale)
        self._after_fused_step()


class ParamStore(FlatStore):
    """FlatStore of the FCRN module (network/FCRN.py): trunk, then conv2 / bn2 / decoder / conv3; each UpProj module's two
    5x5 convs (and their BN vectors) adjacent, so they run as ONE GEMM / one BN site.  Channel counts are stored padded to
    64 (only the ResNet-18/34 decoders have smaller ones), as the first round's kernels required."""

    pad_to = 64

    def _blocks(self):
        for li in (1, 2, 3, 4):
            for blk in getattr(self.m, "layer%d" % li):
                yield li, blk

    def _layout(self):
        m = self.m
        plist, blist = [], []     # (key, [tensors])  -- a fused entry lists several tensors back to back

        def conv_bn(conv, bn):
            plist.append(("w", [conv.weight]))
            plist.append(("g", [bn.weight]))
            plist.append(("b", [bn.bias]))
            blist.append([bn.running_mean])
            blist.append([bn.running_var])

        conv_bn(m.conv1, m.bn1)
        for _, blk in self._blocks():
            conv_bn(blk.conv1, blk.bn1)
            conv_bn(blk.conv2, blk.bn2)
            if hasattr(blk, "conv3"):
                conv_bn(blk.conv3, blk.bn3)
            if blk.downsample is not None:
                conv_bn(blk.downsample[0], blk.downsample[1])
        n_encoder_entries = len(plist)
        conv_bn(m.conv2, m.bn2)
        kind = getattr(m.upSample, "kind", "upproj")
        for name in ("layer1", "layer2", "layer3", "layer4"):
            up = getattr(m.upSample, name)
            if kind == "upconv":
                conv_bn(up.conv, up.batchnorm)
                continue
            if kind == "deconv":
                conv_bn(getattr(up, "deconv%d" % m.upSample.kernel_size), up.batchnorm)
                continue
            if kind == "fasterupconv":
                for cn in ("conv1_", "conv2_", "conv3_", "conv4_"):
                    seq = getattr(up, cn)
                    conv_bn(seq.conv1, seq.bn1)
                    plist.append(("b", [seq.conv1.bias]))
                continue
            if kind == "fasterupproj":
                for fu in (up.upper_branch.faster_upconv, up.bottom_branch):
                    for cn in ("conv1_", "conv2_", "conv3_", "conv4_"):
                        seq = getattr(fu, cn)
                        conv_bn(seq.conv1, seq.bn1)
                        plist.append(("b", [seq.conv1.bias]))
                conv_bn(up.upper_branch.conv, up.upper_branch.batchnorm)
                continue
            ub, bb = up.upper_branch, up.bottom_branch
            plist.append(("w", [ub.conv1.weight, bb.conv.weight]))          # fused [2C][25][Cin]
            plist.append(("g", [ub.batchnorm1.weight, bb.batchnorm.weight]))
            plist.append(("b", [ub.batchnorm1.bias, bb.batchnorm.bias]))
            blist.append([ub.batchnorm1.running_mean, bb.batchnorm.running_mean])
            blist.append([ub.batchnorm1.running_var, bb.batchnorm.running_var])
            conv_bn(ub.conv2, ub.batchnorm2)
        plist.append(("w", [m.conv3.weight]))
        # (pad O, pad I): the 3-channel stem and the head have their own kernels; a stem for in_channels != 3 runs
        # on the GEMM kernel with its input channels padded
        no_pad = {id(m.conv1.weight): (False, m.conv1.in_channels != 3), id(m.conv3.weight): (False, True)}
        return plist, blist, n_encoder_entries, no_pad


class EngineCore:
    """What every per-shape launch plan shares: the store's flat buffers, conv / BN-site lookup, the split-K choice and the
    (optional) weight-gradient side stream."""

    def __init__(self, module, store, N, H, W):
        self.m, self.store, self.N, self.H, self.W, self.dev = module, store, N, H, W, store.dev
        self.P, self.G, self.B, self.Pb = store.P, store.G, store.B, store.Pb
        self.p_off, self.b_off, self.params = store.p_off, store.b_off, store.params
        # (a plan can be BUILT over a CPU store -- the multi-process CPU tests walk real tapes with stub kernels -- but not run:
        #  every launch needs the GPU)
        on_gpu = torch.device(self.dev).type == "cuda"
        self.cus = torch.cuda.get_device_properties(self.dev).multi_processor_count if on_gpu else 256
        # The weight-gradient GEMMs run on a second stream beside the input-gradient / BatchNorm chain (MDE_WGRAD_STREAM=0: one
        # stream).  With the round-1 kernels (64-128 KB of LDS per workgroup: nothing else fits beside them on a CU) this bought
        # 1 %; with the single-buffer tiles (36 KB) the two streams' workgroups share the CUs and the HBM-bound BatchNorm passes
        # overlap the MFMA-bound weight gradients: 999 -> 1 048 images/s.  Per-kernel durations then include the overlap, so
        # while ops.TIMER is recording (bench.py's roofline leg: one instrumented step) everything runs on ONE stream.
        self.side = torch.cuda.Stream(self.dev) if (on_gpu and os.environ.get("MDE_WGRAD_STREAM", "1") == "1") else None
        self.side_busy = False
        self.split = False            # this forward runs over the two-term eval operands (begin_forward)
        self.sites = []               # every BNSite of the plan (BNSite.__init__ / half register themselves)
        # partial sums a pass left behind: the fused-finalize launches read the sums without zeroing them, the OTHER direction's
        # launch does (BNSite.fin / bfin); a pass that is repeated without its counterpart zeroes them here first
        self._fwd_sums, self._bwd_sums = False, False

    def attach_grads(self):
        return self.store.attach_grads()

    @property
    def fin_fused(self):
        return self.store.fin_fused

    def begin_forward(self, train, check_data):
        """Weight shadows current for this forward: the one-term shadow always, the two-term eval operands in eval mode."""
        self.store.det_begin()
        self.store.refresh_weights(check_data=check_data)
        self.split = (not train) and self.store.ensure_split(always=check_data)
        if train:
            if self._fwd_sums:                 # a training-mode forward without a backward since (or an interrupted pass)
                for s in self.sites:
                    if s.part is not None:
                        s.part.zero_()
            # fused: this pass leaves its sums for the backward launches to zero, and zeroes the backward sums itself;
            # separate finalize kernels zero what they read
            self._fwd_sums = True

    def reset_sums(self):
        """Zero every BatchNorm partial-sum buffer of the plan.  forward() / backward() keep them consistent by themselves; a driver
        that runs single layers of the plan (the teacher-forced layer tests) calls this before each training-mode layer pass."""
        for s in self.sites:
            if s.part is not None:
                s.part.zero_()
            s.part_b.zero_()
        self._fwd_sums = self._bwd_sums = False

    def end_forward(self, train):
        if train:
            if self.fin_fused:
                self._bwd_sums = False
            else:
                self._fwd_sums = False

    def begin_backward(self):
        self._wcount = 0
        if self._bwd_sums:                     # a backward pass repeated, or interrupted, without a training-mode forward in between
            for s in self.sites:
                s.part_b.zero_()
            for op in getattr(self, "tape", ()):       # ... whose launches may have left "sums already added" marks behind
                for a in op.acts():
                    a.reduced = False
        self._bwd_sums = True

    def end_backward(self):
        if self.fin_fused:
            self._fwd_sums = False
        else:
            self._bwd_sums = False

    def fwd_conv(self, desc, x, conv, out, stats=None, transposed=False, bias=None, res=None, act=None):
        """A FORWARD convolution launch.  Training (and MDE_EVAL_SPLIT=0): the one-term bf16 shadow (`conv.wf`, or `conv.wd`
        where the forward contracts with the transposed packing), BatchNorm statistics from the epilogue.  Eval mode: the
        tap-doubled launch over the two-term operand (ops.conv_gemm_eval) -- except a fused-epilogue conv whose doubled tap
        list does not fit one launch (a dense 5x5 with bias / activation: Eigen's stacks), which keeps the one-term shadow."""
        fused = bias is not None or res is not None or bool(ops.ACT_CODE[act])
        if self.split and (not fused or ops.split_fits(desc)):
            w2 = conv.w2d if transposed else conv.w2
            ops.conv_gemm_eval(desc, x, w2, out, bias=bias, res=res, act=act)
            return
        ops.conv_gemm(desc, x, conv.wd if transposed else conv.wf, out, stats, bias=bias, res=res, act=act)

    # ------------------------------------------------------------------ plan helpers
    def _conv(self, weights, need_dgrad=True):
        return self.store.conv(weights, need_dgrad)

    def _site(self, bns):
        g0, b0, rm0, rv0 = bns[0].weight, bns[0].bias, bns[0].running_mean, bns[0].running_var
        C = sum(self.store.sdims[id(b.weight)][0] for b in bns)     # storage (possibly padded) channel count
        po, pb = self.p_off[id(g0)], self.p_off[id(b0)]
        bo, bv = self.b_off[id(rm0)], self.b_off[id(rv0)]
        return BNSite(self, C, self.P[po:po + C], self.P[pb:pb + C], po, pb,
                      self.B[bo:bo + C], self.B[bv:bv + C], bns[0], bns[0].eps)

    def _site_chunk(self, bn, c0, cn):
        """A BN site over channels [c0, c0 + cn) of one BatchNorm (sites wider than the kernels' 2048 channels are split)."""
        po, pb = self.p_off[id(bn.weight)] + c0, self.p_off[id(bn.bias)] + c0
        bo, bv = self.b_off[id(bn.running_mean)] + c0, self.b_off[id(bn.running_var)] + c0
        return BNSite(self, cn, self.P[po:po + cn], self.P[pb:pb + cn], po, pb, self.B[bo:bo + cn], self.B[bv:bv + cn], bn, bn.eps)

    def _ksplit(self, pixels, rows, cols, ntaps):
        ba, bb = (128 if rows % 128 == 0 else 64), (128 if cols % 128 == 0 else 64)
        lds = 2 * 64 * (ba + bb) * 2 + 512           # conv_wgrad_tn's staging ring + offset table
        wgpc = int(os.environ.get("MDE_WGRAD_WGPC", "0")) or min(8, (160 * 1024) // lds)      # (diagnostics: occupancy the split-K model assumes)
        return ops.choose_ksplit(pixels, -(-rows // ba), -(-cols // bb), ntaps, self.cus, wg_per_cu=wgpc, tile_elems=ba * bb)


    def wgrad(self, desc, a, b, dw):
        """Weight-gradient GEMM of one conv.  It only reads dY and the activation, so it can run beside the
        input-gradient GEMM of the same layer: on a second stream its workgroups fill the CUs that the other
        kernel leaves idle and its MFMAs overlap the HBM-bound BatchNorm passes (MDE_WGRAD_STREAM=0 turns it off; a step that
        is being timed per launch — ops.TIMER — stays on one stream so that each duration describes one kernel).  (Round 4: the
        launches dealt over TWO side streams measured slower -- FCRN 28.0 -> 28.4 ms, BTS 44.8 -> 45.4: the weight-gradient
        kernels then compete with each other, not only with the main stream.)"""
        ws = self._wgrad_ws(desc)
        if self.side is None or ops.TIMER is not None:
            ops.conv_wgrad(desc, a, b, dw, ws)
            return
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)                  # dY (and everything before it) is ready
        with torch.cuda.stream(self.side):
            ops.conv_wgrad(desc, a, b, dw, ws)
        self.side_busy = True
        if _JOIN_RANGE is not None:                 # diagnostics: MDE_WGRAD_JOIN=lo:hi joins the side stream right behind launches lo..hi-1 of a pass
            self._wcount = getattr(self, "_wcount", 0) + 1
            if _JOIN_RANGE[0] <= self._wcount - 1 < _JOIN_RANGE[1]:
                cur.wait_stream(self.side)

    _WS_MIN, _WS_MAX = 256 << 20, 2 << 30

    def _wgrad_ws(self, desc, slot="_ws"):
        """The workspace of the two-stage split-K reduction (ops.conv_wgrad): ONE buffer per engine, shared by all of its
        weight-gradient launches -- they are ordered on one stream (the side stream, or the main one without it).  Grown on
        demand after a device synchronisation; launches that would need more than 2 GiB keep the atomic path."""
        need = getattr(desc, "_ws_need", None)
        if need is None:
            need = desc._ws_need = ops.wgrad_ws_bytes(desc)
        if need == 0 or need > self._WS_MAX:
            return None
        ws = getattr(self, slot, None)
        if ws is None or ws.numel() < need:
            torch.cuda.synchronize(self.dev)
            ws = torch.empty(max(need, self._WS_MIN), dtype=torch.uint8, device=self.dev)
            setattr(self, slot, ws)
        return ws

    def join_side(self):
        if self.side is not None and self.side_busy:
            torch.cuda.current_stream().wait_stream(self.side)
            self.side_busy = False


class FCRNEngine(EngineCore):
    """Static launch plan for one (batch, height, width) input shape over a ParamStore."""

    def __init__(self, module, store, N, H, W):
        super().__init__(module, store, N, H, W)
        self.out_channels = module.conv3.out_channels
        self.OH, self.OW = module.output_size
        self._plan()

    def _blocks(self):
        return self.store._blocks()

    # ------------------------------------------------------------------ the plan
    def _plan(self):
        m, dev, N, H, W = self.m, self.dev, self.N, self.H, self.W
        self.layers = []
        # stem: conv1 (fp32 NCHW in) -> bn1 + relu -> maxpool
        H2, W2 = ops.out_size(H, 7, 2, 3), ops.out_size(W, 7, 2, 3)
        self.stem_w = self._conv([m.conv1.weight], need_dgrad=False)
        self.stem_c = Act(dev, N, H2, W2, 64)
        self.cin = m.conv1.in_channels
        if self.cin != 3:
            # reference FCRN.py:307-313: a fresh 7x7/2 conv for in_channels != 3.  No special kernel: the image goes
            # to NHWC bf16 with its channels zero-padded to the stored weight's 64, the 49 taps run as two launches
            # of the GEMM kernel (32 + 17, the second accumulating), BN statistics by the stand-alone reduction.
            Cp = self.stem_w.I
            self.xin = Act(dev, N, H, W, Cp)
            taps = [(i - 3, j - 3, i * 7 + j) for i in range(7) for j in range(7)]
            self.stem_fd, self.stem_wd = [], []
            ks = self._ksplit(N * H2 * W2, 64, Cp, 32)
            for part, acc in ((taps[:32], False), (taps[32:], True)):
                self.stem_fd.append(ops.conv_desc(N, H, W, Cp, Cp, self.xin.nbytes, H2, W2, 2, 2, part, 49, H2, W2, 64,
                                                  ncols=64, accumulate=acc))
                self.stem_wd.append(ops.wgrad_desc(N, H2, W2, 64, 64, self.stem_c.nbytes, H, W, Cp, Cp, self.xin.nbytes,
                                                   2, 2, part, 49, False, ks))
        self.stem_a = Act(dev, N, H2, W2, 64)
        self.stem_site = self._site([m.bn1])
        H4, W4 = ops.out_size(H2, 3, 2, 1), ops.out_size(W2, 3, 2, 1)
        self.pool = Act(dev, N, H4, W4, 64)
        self.pool_idx = torch.empty(N, H4, W4, 64, dtype=torch.uint8, device=dev)
        x = self.pool
        for _, blk in self._blocks():
            L = Bottleneck(self, x, blk) if hasattr(blk, "conv3") else BasicBlock(self, x, blk)
            self.layers.append(L)
            x = L.out
        L = ConvBN(self, x, self._conv([m.conv2.weight]), self._site([m.bn2]), 1, 1, 0, relu=False)
        L.conv_weight0 = m.conv2.weight
        L.last_writer = True                       # the trunk's output feeds conv2 only
        self.layers.append(L)
        x = L.out
        for name in ("layer1", "layer2", "layer3", "layer4"):
            mod = getattr(m.upSample, name)
            kind = getattr(m.upSample, "kind", "upproj")
            if kind == "upproj":
                L = UpProjLayer(self, x, mod)
            elif kind == "fasterupproj":
                L = FasterUpProjLayer(self, x, mod)
            elif kind == "fasterupconv":
                L = FasterUpConvLayer(self, x, mod)
            elif kind == "upconv":
                L = UpConvLayer(self, x, mod)
            else:
                L = DeConvLayer(self, x, mod, m.upSample.kernel_size)
            self.layers.append(L)
            x = L.out
        self.feat = x                                   # [N][H/2][W/2][64] for the standard net
        self.head_w = self._conv([m.conv3.weight], need_dgrad=False)
        Co = self.out_channels
        self.logits = torch.empty(N, x.H, x.W, Co, device=dev)
        self.dlogits = torch.empty(N, x.H, x.W, Co, device=dev)
        self.y = torch.empty(N, Co, self.OH, self.OW, device=dev)

    # ------------------------------------------------------------------ execution
    def forward(self, x, train, check_data=False):
        assert x.shape == (self.N, self.cin, self.H, self.W) and x.dtype == torch.float32 and x.is_contiguous()
        self.begin_forward(train, check_data)
        self.x = x
        s = self.stem_site
        if self.cin == 3:
            ops.stem_conv_fwd(x, self.stem_w.w32, self.stem_c.t, s.part if train else None)
        else:
            ops.nchw_to_nhwc_bf16_pad(x, self.xin.t, self.xin.C)
            for d in self.stem_fd:
                self.fwd_conv(d, self.xin.t, self.stem_w, self.stem_c.t)
            if train:
                ops.bn_stats(self.stem_c.t, self.stem_c.M, 64, 64, s.part)
        s.apply(self, self.stem_c.t, 64, self.stem_a.t, 64, self.stem_c.M, True, train)
        ops.maxpool_fwd(self.stem_a.t, self.pool.t, self.pool_idx, self.N, self.stem_a.H, self.stem_a.W, 64)
        for L in self.layers:
            L.fwd(train)
        f = self.feat
        ops.head_conv_fwd(f.t, self.head_w.w32, self.logits, f.N, f.H, f.W, f.C, self.out_channels)
        ops.upsample_sigmoid_fwd(self.logits, self.y, f.N, f.H, f.W, self.out_channels, self.OH, self.OW)
        if train:
            self.store.nbt += 1
        self.end_forward(train)
        return self.y

    def grad_boundaries(self):
        """Flat-gradient offsets at which each plan layer's parameters start (forward order)."""
        offs = [self.store.p_off[id(self.m.conv1.weight)]]
        for L in self.layers:
            offs.append(L.first_param_offset())
        offs.append(self.store.p_off[id(self.m.conv3.weight)])
        return sorted(offs)

    def backward(self, dy, on_progress=None, consumer_waits_side=False):
        """dy: fp32 NCHW gradient w.r.t. the output.  Adds parameter gradients into self.G.
        on_progress(offset), if given, is called whenever every gradient element >= offset of
        the flat buffer is final (backward walks the forward-ordered buffer from its tail) once the
        work queued so far on the current stream AND on self.side (the weight-gradient stream) has run.
        consumer_waits_side: the callback orders itself behind self.side (dp.FlatGradReducer(extra_streams=
        [eng.side])); otherwise the side stream is joined into the current one before every call."""
        assert dy.shape == self.y.shape and dy.dtype == torch.float32 and dy.is_contiguous()
        det = self.store.deterministic
        self.store.det_begin()
        if det:
            progress, on_progress = on_progress, None     # the gradients reach G only with the final flush
        f = self.feat
        for L in self.layers:
            L.reset_grad_flags()
        self.pool.gw = self.stem_a.gw = False
        self.begin_backward()
        ops.upsample_sigmoid_bwd(dy, self.y, self.dlogits, f.N, f.H, f.W, self.out_channels, self.OH, self.OW)
        ops.head_conv_bwd(f.t, self.head_w.w32, self.dlogits, f.g, self.head_w.dw, f.N, f.H, f.W, f.C, self.out_channels)
        f.gw = True
        for L in reversed(self.layers):
            L.bwd()
            if on_progress is not None:
                if not consumer_waits_side:
                    self.join_side()                   # the layer's weight gradients must be final too
                on_progress(L.first_param_offset())
        ops.maxpool_bwd(self.pool.g, self.pool_idx, self.stem_a.g, self.N, self.stem_a.H, self.stem_a.W, 64)
        s = self.stem_site
        s.backward(self.stem_a.g, self.stem_a, self.stem_c, True, self.stem_c.g, mask_from_x=True)
        if self.cin == 3:
            ops.stem_conv_wgrad(self.x, self.stem_c.g, self.stem_w.dw)
        else:
            for d in self.stem_wd:
                self.wgrad(d, self.stem_c.g, self.xin.t, self.stem_w.dw)
        self.join_side()
        self.store.det_end()
        self.end_backward()
        if det:
            on_progress = progress
        if on_progress is not None:
            on_progress(0)


class ConvBN:
    """conv (implicit GEMM, BN statistics from its epilogue) -> BN -> [+ residual] -> [ReLU]."""

    def __init__(self, eng, x, conv, site, k, stride, pad, relu, res=None, res_site=None, has_out=True):
        self.eng, self.x, self.conv, self.site = eng, x, conv, site
        self.conv_weight0 = None      # set by the owner (first Parameter of this unit)
        self.k, self.stride, self.pad, self.relu, self.res, self.res_site = k, stride, pad, relu, res, res_site
        Cout, dev = conv.O, eng.dev
        OH, OW = ops.out_size(x.H, k, stride, pad), ops.out_size(x.W, k, stride, pad)
        self.c = Act(dev, x.N, OH, OW, Cout)      # pre-BN
        self.out = Act(dev, x.N, OH, OW, Cout) if has_out else None   # post BN / residual / ReLU
        # residual joins keep a bit-packed ReLU mask (1 byte per 8 channels) for the backward passes
        self.bits = (torch.empty(x.N * OH * OW * (Cout // 8), dtype=torch.uint8, device=dev)
                     if (relu and res is not None and has_out) else None)
        if has_out:
            self.out.producer = self
        self._red = False
        self._red_add = None
        self.last_writer = False      # set by the owner where x feeds this unit only: see conv_bwd
        self.fdesc = ops.fwd_desc(x.N, x.H, x.W, x.ld, x.C, x.nbytes, k, stride, pad, Cout, Cout)
        self.ddescs, self.dzero = ops.dgrad_descs(x.N, x.H, x.W, x.ld, x.C, OH, OW, Cout, Cout, self.c.nbytes, k, stride, pad)
        ks = eng._ksplit(self.c.M, Cout, x.C, k * k)
        self.wdesc = ops.conv_wgrad_desc(x.N, x.H, x.W, x.ld, x.C, x.nbytes, OH, OW, Cout, Cout, self.c.nbytes, k,
                                         stride, pad, ks)

    def first_param_offset(self):
        return self.eng.store.p_off[id(self.conv_weight0)]

    def reset_grad_flags(self):
        self.c.gw = False
        if self.out is not None:
            self.out.gw = False

    def conv_fwd(self, train):
        self.eng.fwd_conv(self.fdesc, self.x.t, self.conv, self.c.t, self.site.part if train else None)
        if not (train and self.eng.fin_fused):
            self.site.finalize(self.c.M, train)          # (fused: inside the apply launch that reads this site -- fwd below, or the join's)

    def fwd(self, train):
        self.conv_fwd(train)
        s, c, o = self.site, self.c, self.out
        fused = train and self.eng.fin_fused
        if self.res is None:
            s.apply(self.eng, c.t, c.ld, o.t, o.ld, c.M, self.relu, train, finalized=not fused)
        elif self.res_site is None:
            s.apply(self.eng, c.t, c.ld, o.t, o.ld, c.M, self.relu, train, r=self.res.t, ldr=self.res.ld,
                    relu_bits=self.bits if train else None, finalized=not fused)
        else:
            # (the second site was finalized by its own unit's conv_fwd -- a shortcut conv, the up-projection's 5x5 -- unless fused)
            s.apply(self.eng, c.t, c.ld, o.t, o.ld, c.M, self.relu, train, r=self.res.t, ldr=self.res.ld, res_site=self.res_site,
                    relu_bits=self.bits if train else None, finalized=not fused)

    def red_spec(self):
        """For the launch that completes d(out): this unit's BatchNorm-backward sums (a join: both sites') from that launch's
        epilogue; None where it cannot."""
        if self._red is False:
            self._red = None
            s, rs = self.site, self.res_site
            if self.out is not None and self.out.ld == self.c.ld and rs is None:
                self._red = s.red_spec(self.c, self.relu, mask_from_x=self.res is None, relu_bits=self.bits)
            elif self.out is not None and self.out.ld == self.c.ld and FUSE_BN_RED and self.bits is not None and s.C % 8 == 0:
                # a join: the sums of both sites under the one mask (what bn_join_backward's first pass computes)
                self._red = ops.bn_red(self.c.t, s.smean, s.srstd, s.part_b, relu_bits=self.bits, x_ld=self.c.ld,
                                       second=(self.res.t, rs.smean, rs.srstd, rs.part_b, self.res.ld))
        return self._red

    def bn_bwd(self, dres_to=None):
        """d(out) -> d(c) (in self.c.g); optionally routes the masked gradient to an identity residual."""
        dres = None
        if dres_to is not None:
            assert not dres_to.gw, "identity-residual gradient must be the first writer"
            dres = dres_to.g
            dres_to.gw = True
        reduced, self.out.reduced = self.out.reduced, False
        self.site.backward(self.out.g, self.out, self.c, self.relu, self.c.g, dres=dres, mask_from_x=self.res is None,
                           relu_bits=self.bits, reduced=reduced)
        self.c.gw = True

    def conv_bwd(self, last_writer=False, red=None, add=None):
        """last_writer: no other gradient reaches x after this one, so the unit that produced x gets its BatchNorm-backward sums
        from these launches' epilogues (red: the same for a site that is not a ConvBN's, given by the caller).
        add = (gradient, mask bits): these launches are the FIRST writer of d(x) too and bring that gradient in themselves (an
        identity shortcut's: Bottleneck.bwd)."""
        x, eng = self.x, self.eng
        eng.wgrad(self.wdesc, self.c.g, x.t, self.conv.dw)
        acc = x.gw
        if self.dzero and not acc:
            x.g.zero_()
        if red is None and last_writer and x.producer is not None and x.parent is None:
            red = x.producer.red_spec()
        if add is not None:
            assert red is not None and not acc and not self.dzero and len(self.ddescs) == 1
            if self._red_add is None:
                self._red_add = ops.bn_red_with_add(red, add[0], add[1])
            red = self._red_add
        for d in self.ddescs:
            d.accumulate = int(acc)
            ops.conv_gemm(d, self.c.g, self.conv.wd, x.g, red=red)
        x.gw = True
        x.reduced = red is not None

    def bwd(self):
        self.bn_bwd()
        self.conv_bwd(last_writer=self.last_writer)


class Bottleneck:
    """torchvision Bottleneck v1.5 (called from reference network/FCRN.py:320-323)."""

    def __init__(self, eng, x, blk):
        self.x, self.eng, self.blk = x, eng, blk
        s = blk.conv2.stride[0]
        self.a = ConvBN(eng, x, eng._conv([blk.conv1.weight]), eng._site([blk.bn1]), 1, 1, 0, True)
        self.b = ConvBN(eng, self.a.out, eng._conv([blk.conv2.weight]), eng._site([blk.bn2]), 3, s, 1, True)
        self.ds = None
        if blk.downsample is not None:
            self.ds = ConvBN(eng, x, eng._conv([blk.downsample[0].weight]), eng._site([blk.downsample[1]]), 1,
                             blk.downsample[0].stride[0], 0, False, has_out=False)
            self.c = ConvBN(eng, self.b.out, eng._conv([blk.conv3.weight]), eng._site([blk.bn3]), 1, 1, 0, True,
                            res=self.ds.c, res_site=self.ds.site)
        else:
            self.c = ConvBN(eng, self.b.out, eng._conv([blk.conv3.weight]), eng._site([blk.bn3]), 1, 1, 0, True, res=x)
        self.out = self.c.out

    def first_param_offset(self):
        return self.eng.store.p_off[id(self.blk.conv1.weight)]

    def reset_grad_flags(self):
        for u in (self.a, self.b, self.c, self.ds):
            if u is not None:
                u.reset_grad_flags()

    def fwd(self, train):
        self.a.fwd(train)
        self.b.fwd(train)
        if self.ds is not None:
            self.ds.conv_fwd(train)          # its BN apply is folded into c's join
        self.c.fwd(train)

    def bwd(self):
        c, ds, x = self.c, self.ds, self.x
        add = None
        if ds is None:
            # identity shortcut: d(x) = masked d(out) + conv1's input gradient.  Where conv1's launch can bring the masked d(out)
            # in itself (it carries the producer's BatchNorm-backward sums anyway: mde_bn_red.add), bn3's pass does not write it
            spec = x.producer.red_spec() if (FUSE_DRES and x.producer is not None and x.parent is None) else None
            if (spec is not None and (spec.relu_bits or spec.x2) and not x.gw and c.bits is not None and c.out.ld == x.ld == x.C
                    and len(self.a.ddescs) == 1 and not self.a.dzero):
                add = (c.out.g, c.bits)
            c.bn_bwd(dres_to=None if add is not None else x)
        else:
            # both BN sites of the join (conv3 and the shortcut conv) see the same masked gradient
            bn_join_backward(c.site, ds.site, c.out.g, c.out, c.c, ds.c, c.c.g, ds.c.g, c.bits)
            c.c.gw = ds.c.gw = True
            ds.conv_bwd()
        c.conv_bwd(last_writer=True)         # (b.out and a.out feed one conv each; x: the shortcut's gradient is in already)
        self.b.bn_bwd()
        self.b.conv_bwd(last_writer=True)
        self.a.bn_bwd()
        self.a.conv_bwd(last_writer=True, add=add)


class BasicBlock:
    """torchvision BasicBlock (resnet18 / 34, reference network/FCRN.py:305 with layers <= 34): 3x3(stride) -> BN -> ReLU
    -> 3x3 -> BN -> (+ identity or 1x1/stride shortcut-BN) -> ReLU."""

    def __init__(self, eng, x, blk):
        self.x, self.eng, self.blk = x, eng, blk
        s = blk.conv1.stride[0]
        self.a = ConvBN(eng, x, eng._conv([blk.conv1.weight]), eng._site([blk.bn1]), 3, s, 1, True)
        self.ds = None
        if blk.downsample is not None:
            self.ds = ConvBN(eng, x, eng._conv([blk.downsample[0].weight]), eng._site([blk.downsample[1]]), 1,
                             blk.downsample[0].stride[0], 0, False, has_out=False)
            self.c = ConvBN(eng, self.a.out, eng._conv([blk.conv2.weight]), eng._site([blk.bn2]), 3, 1, 1, True,
                            res=self.ds.c, res_site=self.ds.site)
        else:
            self.c = ConvBN(eng, self.a.out, eng._conv([blk.conv2.weight]), eng._site([blk.bn2]), 3, 1, 1, True, res=x)
        self.out = self.c.out

    def first_param_offset(self):
        return self.eng.store.p_off[id(self.blk.conv1.weight)]

    def reset_grad_flags(self):
        for u in (self.a, self.c, self.ds):
            if u is not None:
                u.reset_grad_flags()

    def fwd(self, train):
        self.a.fwd(train)
        if self.ds is not None:
            self.ds.conv_fwd(train)
        self.c.fwd(train)

    def bwd(self):
        c, ds = self.c, self.ds
        if ds is None:
            c.bn_bwd(dres_to=self.x)
        else:
            bn_join_backward(c.site, ds.site, c.out.g, c.out, c.c, ds.c, c.c.g, ds.c.g, c.bits)
            c.c.gw = ds.c.gw = True
            ds.conv_bwd()
        c.conv_bwd()
        self.a.bwd()


class UpProjLayer:
    """reference network/FCRN.py:170-198 without the zero-stuffed tensor: both 5x5 branches as
    one 4-phase GEMM into y55 [N][2h][2w][2C]; upper half -> BN -> ReLU -> 3x3 -> BN, joined
    with BN(lower half) and ReLU."""

    def __init__(self, eng, x, mod):
        self.eng, self.x, self.mod = eng, x, mod
        dev, N, h, w, Cin = eng.dev, x.N, x.H, x.W, x.C
        ub, bb = mod.upper_branch, mod.bottom_branch
        self.w55 = eng._conv([ub.conv1.weight, bb.conv.weight])
        C = self.w55.O // 2                               # (storage channels: Cin // 2, or 64 where that is smaller)
        self.site55 = eng._site([ub.batchnorm1, bb.batchnorm])
        self.site_u, self.site_b = self.site55.half(eng, 0), self.site55.half(eng, 1)
        self.y55 = Act(dev, N, 2 * h, 2 * w, 2 * C)
        self.y_u, self.y_b = self.y55.slice(0, C), self.y55.slice(C, C)
        self.a1 = Act(dev, N, 2 * h, 2 * w, C)
        self.fdescs = ops.upproj_fwd_descs(N, h, w, x.ld, Cin, x.nbytes, 2 * C, 2 * C)
        self.ddesc = ops.upproj_dgrad_desc(N, h, w, x.ld, Cin, 2 * C, 2 * C, self.y55.nbytes)
        ks = eng._ksplit(x.M, 2 * C, Cin, 25)
        self.wdesc = ops.upproj_wgrad_desc(N, h, w, x.ld, Cin, x.nbytes, 2 * C, 2 * C, self.y55.nbytes, ks)
        self.c2 = ConvBN(eng, self.a1, eng._conv([ub.conv2.weight]), eng._site([ub.batchnorm2]), 3, 1, 1, True,
                         res=self.y_b, res_site=self.site_b)
        self.out = self.c2.out
        self._red_u = False

    def first_param_offset(self):
        return self.eng.store.p_off[id(self.mod.upper_branch.conv1.weight)]

    def reset_grad_flags(self):
        self.y55.gw = self.a1.gw = False
        self.c2.reset_grad_flags()

    def fwd(self, train):
        x, y = self.x, self.y55
        for d in self.fdescs:
            self.eng.fwd_conv(d, x.t, self.w55, y.t, self.site55.part if train else None)
        su = self.site_u
        if train and self.eng.fin_fused:
            # the upper half's constants come out of its own apply launch; the lower half's out of the join's (c2.fwd: res_site)
            su.apply(self.eng, self.y_u.t, y.ld, self.a1.t, self.a1.ld, y.M, True, train)
        else:
            self.site55.finalize(y.M, train)
            su.apply(self.eng, self.y_u.t, y.ld, self.a1.t, self.a1.ld, y.M, True, train, finalized=True)
        self.c2.fwd(train)

    def bwd(self):
        x, y, c2 = self.x, self.y55, self.c2
        yg = y.g
        C = self.site_u.C
        # join of bn2(conv3x3) and bn(bottom 5x5): d(out) -> d(c2.c) and d(y55 lower half) in one pair of passes
        bn_join_backward(c2.site, self.site_b, c2.out.g, c2.out, c2.c, self.y_b, c2.c.g, yg[..., C:], c2.bits)
        c2.c.gw = True
        if self._red_u is False:
            self._red_u = self.site_u.red_spec(self.y_u, True, mask_from_x=True)
        c2.conv_bwd(red=self._red_u)                                        # -> d(a1) (+ site_u's sums), dW(conv2)
        self.site_u.backward(self.a1.g, self.a1, self.y_u, True, yg[..., :C], mask_from_x=True, reduced=self._red_u is not None)
        self.a1.reduced = False
        self.eng.wgrad(self.wdesc, x.t, yg, self.w55.dw)
        self.ddesc.accumulate = int(x.gw)
        red = x.producer.red_spec() if (x.producer is not None and x.parent is None) else None   # (x feeds this layer only)
        ops.conv_gemm(self.ddesc, yg, self.w55.wd, x.g, red=red)
        x.gw = True
        x.reduced = red is not None


class UpConvLayer:
    """reference network/FCRN.py:91-110 (`UpConv.upconv_module`): unpool -> 5x5 conv -> BN -> ReLU, the zero-stuffed
    tensor never materialised: the 5x5 runs as four output phases over x (a quarter of the dense taps)."""

    def __init__(self, eng, x, mod):
        self.eng, self.x, self.mod = eng, x, mod
        dev, N, h, w, Cin = eng.dev, x.N, x.H, x.W, x.C
        C = Cin // 2
        self.w = eng._conv([mod.conv.weight])
        self.site = eng._site([mod.batchnorm])
        self.y = Act(dev, N, 2 * h, 2 * w, C)          # pre-BN
        self.out = Act(dev, N, 2 * h, 2 * w, C)
        self.fdescs = ops.upproj_fwd_descs(N, h, w, x.ld, Cin, x.nbytes, C, C)
        self.ddesc = ops.upproj_dgrad_desc(N, h, w, x.ld, Cin, C, C, self.y.nbytes)
        self.wdesc = ops.upproj_wgrad_desc(N, h, w, x.ld, Cin, x.nbytes, C, C, self.y.nbytes, eng._ksplit(x.M, C, Cin, 25))

    def first_param_offset(self):
        return self.eng.store.p_off[id(self.mod.conv.weight)]

    def reset_grad_flags(self):
        self.y.gw = self.out.gw = False

    def fwd(self, train):
        x, y, s = self.x, self.y, self.site
        for d in self.fdescs:
            self.eng.fwd_conv(d, x.t, self.w, y.t, s.part if train else None)
        s.apply(self.eng, y.t, y.ld, self.out.t, self.out.ld, y.M, True, train)

    def bwd(self):
        x, y = self.x, self.y
        self.site.backward(self.out.g, self.out, y, True, y.g, mask_from_x=True)
        self.eng.wgrad(self.wdesc, x.t, y.g, self.w.dw)
        self.ddesc.accumulate = int(x.gw)
        ops.conv_gemm(self.ddesc, y.g, self.w.wd, x.g)
        x.gw = True


class DeConvLayer:
    """reference network/FCRN.py:68-88 (`DeConv.convt`): ConvTranspose2d(k, stride 2, pad (k-1)//2, output_padding
    k%2, no bias) -> BN -> ReLU.  A transposed convolution IS the input gradient of the strided convolution with the
    same weight tensor [Cin][Cout][k][k] read as that convolution's [O][I][kh][kw]: the forward pass runs the
    conv kernel's stride-2 output phases with the transposed packing, the input gradient is the plain strided
    forward convolution of d(out), and the weight gradient is that convolution's weight gradient with the roles of
    activation and output gradient exchanged."""

    def __init__(self, eng, x, mod, k):
        self.eng, self.x, self.mod, self.k = eng, x, mod, k
        dev, N, h, w, Cin = eng.dev, x.N, x.H, x.W, x.C
        C = Cin // 2
        pad = (k - 1) // 2
        assert ops.out_size(2 * h, k, 2, pad) == h and ops.out_size(2 * w, k, 2, pad) == w
        self.convt = getattr(mod, "deconv%d" % k)
        self.w = eng._conv([self.convt.weight])          # O = Cin (of the ConvTranspose), I = C
        self.w.want_w2d()
        self.site = eng._site([mod.batchnorm])
        self.y = Act(dev, N, 2 * h, 2 * w, C)          # pre-BN
        self.out = Act(dev, N, 2 * h, 2 * w, C)
        # forward: "dgrad" of the virtual conv  y-shaped [2h][2w][C] -> x-shaped [h][w][Cin]
        self.fdescs, self.fzero = ops.dgrad_descs(N, 2 * h, 2 * w, self.y.ld, C, h, w, x.ld, Cin, x.nbytes, k, 2, pad)
        # input gradient: the virtual conv's forward over d(y)
        self.ddesc = ops.fwd_desc(N, 2 * h, 2 * w, self.y.ld, C, self.y.nbytes, k, 2, pad, Cin, x.ld)
        self.wdesc = ops.conv_wgrad_desc(N, 2 * h, 2 * w, self.y.ld, C, self.y.nbytes, h, w, x.ld, Cin, x.nbytes, k, 2, pad,
                                         eng._ksplit(x.M, Cin, C, k * k))

    def first_param_offset(self):
        return self.eng.store.p_off[id(self.convt.weight)]

    def reset_grad_flags(self):
        self.y.gw = self.out.gw = False

    def fwd(self, train):
        x, y, s = self.x, self.y, self.site
        if self.fzero:
            y.t.zero_()
        for d in self.fdescs:
            d.accumulate = 0
            self.eng.fwd_conv(d, x.t, self.w, y.t, s.part if train else None, transposed=True)
        s.apply(self.eng, y.t, y.ld, self.out.t, self.out.ld, y.M, True, train)

    def bwd(self):
        x, y = self.x, self.y
        self.site.backward(self.out.g, self.out, y, True, y.g, mask_from_x=True)
        self.eng.wgrad(self.wdesc, x.t, y.g, self.w.dw)
        self.ddesc.accumulate = int(x.gw)
        ops.conv_gemm(self.ddesc, y.g, self.w.wf, x.g)
        x.gw = True


# (pad_top, pad_left, kh, kw) of faster_upconv's conv1_ .. conv4_ (FCRN.py:214-239: F.pad(x, (l, r, t, b)) + Conv2d)
_FASTER_GEOMETRY = ((1, 1, 3, 3), (0, 1, 2, 3), (1, 0, 3, 2), (0, 0, 2, 2))


class _BiasedConvBN:
    """One of faster_upconv's four {Conv2d WITH bias -> BatchNorm2d} units writing a channel slice.  In training
    mode the bias cancels in the normalisation (its gradient is exactly zero and stays zero in the flat buffer); it
    only moves the running mean, and in eval mode it joins the shift."""

    def __init__(self, eng, x, seq, geo, y, yb):
        pt, pl, kh, kw = geo
        self.eng, self.x, self.y, self.yb = eng, x, y, yb            # y: pre-BN slice, yb: post-BN slice
        self.conv = eng._conv([seq.conv1.weight])
        self.site = eng._site([seq.bn1])
        self.bias = eng.P[eng.p_off[id(seq.conv1.bias)]:eng.p_off[id(seq.conv1.bias)] + y.C]
        taps = [(i - pt, j - pl, i * kw + j) for i in range(kh) for j in range(kw)]
        N, H, W, C = x.N, x.H, x.W, y.C
        self.fdesc = ops.taps_fwd_desc(N, H, W, x.ld, x.C, x.nbytes, taps, kh * kw, C, y.ld)
        self.ddesc = ops.taps_dgrad_desc(N, H, W, x.ld, x.C, y.ld, C, y.nbytes, taps, kh * kw)
        self.wdesc = ops.taps_wgrad_desc(N, H, W, x.ld, x.C, x.nbytes, y.ld, C, y.nbytes, taps, kh * kw,
                                         eng._ksplit(x.M, C, x.C, kh * kw))

    def fwd(self, train, relu):
        s, y = self.site, self.y
        self.eng.fwd_conv(self.fdesc, self.x.t, self.conv, y.t, s.part if train else None)
        if train:
            s.apply(self.eng, y.t, y.ld, self.yb.t, self.yb.ld, y.M, relu, True)
            with torch.no_grad():
                mom = s.bn.momentum if s.bn.momentum is not None else 0.1
                s.rmean.add_(self.bias, alpha=mom)           # running mean of (conv + bias), after the statistics' own update
        else:
            s.finalize(y.M, False)
            with torch.no_grad():
                s.shift.addcmul_(s.scale, self.bias)
            s.apply(self.eng, y.t, y.ld, self.yb.t, self.yb.ld, y.M, relu, False, finalized=True)

    def bwd(self, relu):
        x, y, yb = self.x, self.y, self.yb
        self.site.backward(yb.g, yb, y, relu, y.g, mask_from_x=relu)
        self.eng.wgrad(self.wdesc, y.g, x.t, self.conv.dw)
        self.ddesc.accumulate = int(x.gw)
        ops.conv_gemm(self.ddesc, y.g, self.conv.wd, x.g)
        x.gw = True


class FasterUpProjLayer:
    """reference network/FCRN.py:206-281 (`FasterUpProj.FasterUpProjModule`): per branch four convs (3x3, 2x3, 3x2,
    2x2 with asymmetric zero padding, biased) + BN each, concatenated and pixel-shuffled to 2x resolution; upper
    branch -> ReLU -> 3x3 conv -> BN, joined with the bottom branch and ReLU.  The eight convs write channel slices
    of ONE [N][h][w][8C] tensor (the `cat` never exists separately), BN + ReLU run before the shuffle (a pure
    permutation commutes with them), and the shuffle is one 64-byte-per-thread permutation kernel."""

    def __init__(self, eng, x, mod):
        self.eng, self.x, self.mod = eng, x, mod
        dev, N, h, w, Cin = eng.dev, x.N, x.H, x.W, x.C
        C = Cin // 2
        self.C = C
        self.Y = Act(dev, N, h, w, 8 * C)               # pre-BN outputs of the 8 convs: upper 4, bottom 4
        self.T = Act(dev, N, h, w, 8 * C)               # post-BN (upper: + ReLU), still un-shuffled
        self.units = []
        for bi, fu in enumerate((mod.upper_branch.faster_upconv, mod.bottom_branch)):
            for ci, cn in enumerate(("conv1_", "conv2_", "conv3_", "conv4_")):
                c0 = (4 * bi + ci) * C
                self.units.append(_BiasedConvBN(eng, x, getattr(fu, cn), _FASTER_GEOMETRY[ci], self.Y.slice(c0, C),
                                                self.T.slice(c0, C)))
        self.a1 = Act(dev, N, 2 * h, 2 * w, C)          # relu(shuffle(upper))
        self.r = Act(dev, N, 2 * h, 2 * w, C)           # shuffle(bottom)
        ub = mod.upper_branch
        self.c2 = ConvBN(eng, self.a1, eng._conv([ub.conv.weight]), eng._site([ub.batchnorm]), 3, 1, 1, True, res=self.r)
        self.out = self.c2.out

    def first_param_offset(self):
        return self.eng.store.p_off[id(self.mod.upper_branch.faster_upconv.conv1_.conv1.weight)]

    def reset_grad_flags(self):
        self.Y.gw = self.T.gw = self.a1.gw = self.r.gw = False
        self.c2.reset_grad_flags()

    def _shuffle(self, inverse):
        T, C, x = self.T, self.C, self.x
        for half, dst in ((0, self.a1), (1, self.r)):
            src = (T.g if inverse else T.t)[..., half * 4 * C:(half + 1) * 4 * C]
            ops.pixel_shuffle2(src, T.ld, dst.g if inverse else dst.t, dst.ld, x.N, x.H, x.W, C, inverse)

    def fwd(self, train):
        for i, u in enumerate(self.units):
            u.fwd(train, relu=i < 4)
        self._shuffle(False)
        self.c2.fwd(train)

    def bwd(self):
        c2 = self.c2
        c2.bn_bwd(dres_to=self.r)                       # d(out) -> d(c2.c), and the masked gradient to the bottom branch
        c2.conv_bwd()                                   # -> d(a1), dW(conv)
        self._shuffle(True)                             # d(a1), d(r) -> d(T)
        for i, u in enumerate(self.units):
            u.bwd(relu=i < 4)


class FasterUpConvLayer:
    """reference network/FCRN.py:113-164 (`FasterUpConv.faster_upconv_module`): the four biased convs + BN, the
    pixel shuffle, ReLU — one branch of FasterUpProjLayer without the 3x3 and the join."""

    def __init__(self, eng, x, mod):
        self.eng, self.x, self.mod = eng, x, mod
        dev, N, h, w, Cin = eng.dev, x.N, x.H, x.W, x.C
        C = Cin // 2
        self.C = C
        self.Y = Act(dev, N, h, w, 4 * C)
        self.T = Act(dev, N, h, w, 4 * C)
        self.units = [_BiasedConvBN(eng, x, getattr(mod, cn), _FASTER_GEOMETRY[ci], self.Y.slice(ci * C, C), self.T.slice(ci * C, C))
                      for ci, cn in enumerate(("conv1_", "conv2_", "conv3_", "conv4_"))]
        self.out = Act(dev, N, 2 * h, 2 * w, C)

    def first_param_offset(self):
        return self.eng.store.p_off[id(self.mod.conv1_.conv1.weight)]

    def reset_grad_flags(self):
        self.Y.gw = self.T.gw = self.out.gw = False

    def fwd(self, train):
        for u in self.units:
            u.fwd(train, relu=True)
        ops.pixel_shuffle2(self.T.t, self.T.ld, self.out.t, self.out.ld, self.x.N, self.x.H, self.x.W, self.C)

    def bwd(self):
        ops.pixel_shuffle2(self.T.g, self.T.ld, self.out.g, self.out.ld, self.x.N, self.x.H, self.x.W, self.C, inverse=True)
        for u in self.units:
            u.bwd(relu=True)
