"""A small tape of kernel launches for the networks beside FCRN (reference network/VNL.py, MiDaS.py, Bts.py).

engine.py's FCRN plan is hand-ordered; these networks are wider graphs (lateral connections, concatenations, gates),
so their plans are TAPES: an ordered list of ops over NHWC bf16 activations (`engine.Act`: tensor, channel-slice view,
lazily allocated gradient).  `forward` runs the tape, `backward` runs it in reverse.  Gradient convention: every
activation has one gradient buffer; the first op that writes it in a backward pass overwrites, later ones accumulate
(`Act.gw`), so fan-out needs no extra add kernels.  A concatenation is never executed: producers write channel slices
of one wider tensor (pointer + pixel stride), consumers read it whole.

Every op is a thin wrapper over C-ABI entry points (include/mde_hip.h); there is no torch arithmetic here.
"""
import os

import torch

from . import _lib, ops
from .engine import FUSE_BN_RED, FUSE_DRES, Act, BNSite, EngineCore, FlatStore, bn_join_backward


# ---------------------------------------------------------------------------------------------- parameter storage
class NetStore(FlatStore):
    """Generic flat storage: every parameter its own entry, the module's ENCODER parameters first (`is_encoder(name)`;
    the reference's two learning-rate groups: vnl.py:289-326 'res' in key, midas.py:94-105 `pretrained`, bts.py:139-152
    `encoder`), channel counts padded to 8.  `raw` names weights that keep their exact shape (grouped convs [O][G][k][k],
    the 3-channel stem)."""

    def __init__(self, module, device, is_encoder, raw=()):
        self._is_encoder, self._raw = is_encoder, set(raw)
        super().__init__(module, device)

    def _layout(self):
        named = list(self.m.named_parameters())
        enc = [(n, p) for n, p in named if self._is_encoder(n)]
        dec = [(n, p) for n, p in named if not self._is_encoder(n)]
        plist = [("w" if p.dim() == 4 else "v", [p]) for _, p in enc + dec]
        blist = []
        for mod in self.m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                blist.append([mod.running_mean])
                blist.append([mod.running_var])
        no_pad = {id(p): (False, False) for n, p in named if n in self._raw}
        return plist, blist, len(enc), no_pad

    def vec(self, p):
        """fp32 view of a 1-D parameter's storage (padded length) inside P, and its offset."""
        off, n = self.p_off[id(p)], self.sdims[id(p)][0]
        return self.P[off:off + n], off


# ---------------------------------------------------------------------------------------------- ops
class Op:
    def fwd(self, train):
        raise NotImplementedError

    def bwd(self):
        raise NotImplementedError

    def acts(self):
        """Activations this op produces (their gradient flags are reset before a backward pass)."""
        return ()

    def grad_ranges(self):
        """[(begin, end)] element ranges of the flat gradient buffer this op's backward writes (TapeEngine.backward's
        on_progress: a range is final once the last op that lists it has run)."""
        return ()


def _site_ranges(site):
    return [(site.g_off, site.g_off + site.C), (site.b_off, site.b_off + site.C)]


# The first backward of a plan records, per activation, the op that took its gradient LAST (every accumulating writer asks
# _take for its flag; the few writers that do not are asserted first writers).  Where that op is a Conv reading the output of a
# BatchNorm op, the conv's input-gradient launches add that BatchNorm's backward sums from their epilogue from then on
# (TapeEngine._plan_fused_sums, mde_conv_gemm_bnred) and the BatchNorm op skips its reduction pass.
_TRACE = None
_CUR_OP = None


def _take(x):
    """-> accumulate flag for a write into x.g, marking it written."""
    if _TRACE is not None:
        _TRACE.setdefault(id(x.root()), []).append(_CUR_OP)
    if x.parent is not None and not x.root().concat_root:
        # the written-flag is shared by a root and all its slices: a first writer that covers only part of the channels
        # would make the writers of the other channels accumulate onto uninitialised memory, unless the root is a
        # concatenation target (TapeEngine.buf), whose gradient is zeroed and flagged before every backward
        raise AssertionError("gradient write into a channel slice whose root is not a TapeEngine.buf() concatenation target")
    acc = x.gw
    x.gw = True
    return acc


class Stem(Op):
    """7x7/2 conv on the fp32 NCHW image -> BN -> ReLU -> 3x3/2 max-pool (torchvision stem; VNL.py:593-599 basic_bn_stem,
    MiDaS.py:93-96).  The image is the graph's input: no input gradient."""

    def __init__(self, eng, conv, bn, N, H, W):
        dev = eng.dev
        self.eng = eng
        H2, W2 = ops.out_size(H, 7, 2, 3), ops.out_size(W, 7, 2, 3)
        assert H % 2 == 0 and W % 2 == 0, "stem kernel: even image sizes"
        self.w = eng._conv([conv.weight], need_dgrad=False)
        self.site = eng._site([bn])
        self.c, self.a = Act(dev, N, H2, W2, 64), Act(dev, N, H2, W2, 64)
        H4, W4 = ops.out_size(H2, 3, 2, 1), ops.out_size(W2, 3, 2, 1)
        self.out = Act(dev, N, H4, W4, 64)
        self.idx = torch.empty(N, H4, W4, 64, dtype=torch.uint8, device=dev)
        self.x = None

    def acts(self):
        return (self.c, self.a, self.out)

    def grad_ranges(self):
        return [(self.w.off, self.w.off + self.w.n)] + _site_ranges(self.site)

    def fwd(self, train):
        s, c, a = self.site, self.c, self.a
        ops.stem_conv_fwd(self.x, self.w.w32, c.t, s.part if train else None)
        s.apply(self.eng, c.t, 64, a.t, 64, c.M, True, train)
        ops.maxpool_fwd(a.t, self.out.t, self.idx, a.N, a.H, a.W, 64)

    def bwd(self):
        a, c = self.a, self.c
        ops.maxpool_bwd(self.out.g, self.idx, a.g, a.N, a.H, a.W, 64)
        self.site.backward(a.g, a, c, True, c.g, mask_from_x=True)
        ops.stem_conv_wgrad(self.x, c.g, self.w.dw)


class Conv(Op):
    """A convolution (no bias, no activation): dense or grouped, any stride / dilation, writing `out` (its own tensor or a
    channel slice of a wider one).  `site`: the BatchNorm that follows, whose batch statistics come out of this
    launch's epilogue in training mode."""

    def __init__(self, eng, x, w, k, stride=1, pad=0, dil=1, groups=1, out=None, site=None, need_dgrad=True):
        self.eng, self.x, self.site, self.need_dgrad = eng, x, site, need_dgrad
        store = eng.store
        O = store.sdims[id(w)][0] if w.dim() == 4 else w.shape[0]
        OH, OW = ops.out_size(x.H, k, stride, pad, dil), ops.out_size(x.W, k, stride, pad, dil)
        self.out = out if out is not None else Act(eng.dev, x.N, OH, OW, O)
        o = self.out
        assert (o.N, o.H, o.W, o.C) == (x.N, OH, OW, O), ((o.N, o.H, o.W, o.C), (x.N, OH, OW, O))
        self.own_out = out is None
        if groups > 1:
            G = w.shape[1]
            assert x.C == O == G * groups, (x.C, O, G, groups)
            self.conv = store.conv_grouped(w, G)
            Cin = 64
        else:
            self.conv = store.conv([w], need_dgrad) if w.dim() == 4 else store.linear(w, need_dgrad)   # nn.Linear: a 1x1 conv
            Cin = self.conv.I
            assert x.C == Cin, "conv input has %d channels, the stored weight %d" % (x.C, Cin)
            G = 0
        # The kernels address an operand through a 32-bit buffer offset (< 2 GiB): a batch whose tensors are larger (VNL's
        # 16 x 480 x 640 x 256 input of the prediction conv is 2.5 GB) runs as equal sub-batches over pointer-offset views;
        # BatchNorm statistics and weight gradients accumulate across the launches anyway (atomics).
        lim = (1 << 31) - (1 << 20)
        per_img = max(x.H * x.W * x.ld, OH * OW * o.ld) * 2
        self.chunk = x.N
        while self.chunk * per_img > lim:
            self.chunk = max(d for d in range(1, self.chunk) if x.N % d == 0)
        n = self.chunk
        xb, ob = (n * x.H * x.W * x.ld - x.c0) * 2, (n * OH * OW * o.ld - o.c0) * 2       # bytes addressable from the first element
        self.fdesc = ops.fwd_desc(n, x.H, x.W, x.ld, Cin, xb, k, stride, pad, O, o.ld, dil=dil)
        self.fdesc.grouped = int(groups > 1)
        if groups > 1:
            self.fdesc._useful = G / 64.0                # (read by ops.conv_gemm's launch timer: algorithmic FLOP)
        if need_dgrad:
            # (grouped: every 64-column tile of dx contracts the matching 64-channel window of dY)
            self.ddescs, self.dzero = ops.dgrad_descs(n, x.H, x.W, x.ld, x.C, OH, OW, o.ld, O if groups == 1 else 64, ob,
                                                      k, stride, pad, dil=dil)
            for d in self.ddescs:
                d.grouped = int(groups > 1)
                if groups > 1:
                    d._useful = G / 64.0
        M = n * OH * OW
        if groups > 1:
            ks = ops.choose_ksplit(M, O // 64, 1, k * k, eng.cus, wg_per_cu=4, tile_elems=64 * 64)
        else:
            ks = eng._ksplit(M, O, Cin, k * k)
        self.wdesc = ops.conv_wgrad_desc(n, x.H, x.W, x.ld, x.C, xb, OH, OW, o.ld, O, ob, k, stride, pad, ks, dil=dil)
        self.wdesc.group_size = G
        self.groups = groups
        # fused epilogue (TapeEngine.pw): out = act(conv + bias + res) written by the conv launch itself
        self.f_bias = self.f_boff = self.f_res = self.f_act = self.f_part = self.pre_g = None
        self.fused = False
        self.eval_raw = False
        self.red = None        # ops.bn_red of the BatchNorm op that wrote x, when this conv completes d(x) (see _TRACE)
        self.first_writer = False   # ... and is its only writer: the identity shortcut's share comes in through red.add

    def fuse(self, bias, b_off, res, act, out):
        """Take over the pointwise pass that would follow this conv: the launch writes act(conv + bias + res) into `out`
        (mde_conv_gemm_act; bit-identical to conv + mde_pw_fwd) and the pre-activation tensor is never materialised.  Its
        GRADIENT still is (pre_g): the weight- and input-gradient GEMMs contract with d(pre-activation)."""
        o = self.out
        assert (out.N, out.H, out.W, out.C) == (o.N, o.H, o.W, o.C) and self.site is None and self.groups == 1
        old_ld = o.ld
        o.t = None                                   # (anything that still reads the pre-activation fails loudly)
        self.out, self.own_out = out, out.parent is None and out is not o
        self.f_bias, self.f_boff, self.f_res, self.f_act, self.fused = bias, b_off, res, act, True
        self.f_part = ops.new_stat_buffer(out.C, self.eng.dev) if bias is not None else None
        assert old_ld == o.C, "only a conv that owns a dense output is fused (its gradient descriptors address pre_g, stride C)"
        self.fdesc.ld_out = out.ld                   # the forward descriptor carries the NEW output's pixel stride
        self.pre_g = torch.empty(out.N, out.H, out.W, out.C, dtype=ops.ACT_DTYPE, device=self.eng.dev)
        return self

    def acts(self):
        return (self.out,) if self.own_out else ()

    def grad_ranges(self):
        r = [(self.conv.off, self.conv.off + self.conv.n)]
        if self.f_bias is not None:
            r.append((self.f_boff, self.f_boff + self.out.C))
        return r

    def _chunks(self):
        return range(0, self.x.N, self.chunk)

    def fwd(self, train):
        stats = self.site.part if (train and self.site is not None) else None
        n = self.chunk
        if self.fused:
            r = self.f_res
            # (eval_raw: an ELU whose only reader is a BatchNorm -- in eval mode that BatchNorm's pass applies it, BN.pre_elu)
            act = None if (self.eval_raw and not train) else self.f_act
            for i in self._chunks():
                self.eng.fwd_conv(self.fdesc, self.x.t[i:i + n], self.conv, self.out.t[i:i + n], bias=self.f_bias,
                                  res=r.t[i:i + n] if r is not None else None, act=act)
            return
        for i in self._chunks():
            self.eng.fwd_conv(self.fdesc, self.x.t[i:i + n], self.conv, self.out.t[i:i + n], stats)

    def bwd(self):
        x, o, eng, n = self.x, self.out, self.eng, self.chunk
        og = o.g
        if self.fused:
            # g = dout * act'(out): the conv's own output gradient; the bias gradient and the residual's ride along
            r = self.f_res
            acc_r = _take(r) if r is not None else False
            dbias = eng.store.Gcur[self.f_boff:self.f_boff + o.C] if self.f_bias is not None else None
            ops.pw_bwd(og, _ldg(o), o.t, o.ld, self.pre_g, o.C, False, r.g if r is not None else None, _ldg(r) if r is not None else 0,
                       acc_r, dbias, o.M, o.C, self.f_act, bias_part=self.f_part)
            og = self.pre_g
        for i in self._chunks():
            eng.wgrad(self.wdesc, og[i:i + n], x.t[i:i + n], self.conv.dw)
        if not self.need_dgrad:
            return
        acc = _take(x)
        assert not (self.first_writer and acc), "a launch that brings the shortcut's gradient in itself is the first writer"
        xg = x.g
        if self.dzero and not acc:
            xg.zero_()
        for d in self.ddescs:
            d.accumulate = int(acc)
            for i in self._chunks():
                ops.conv_gemm(d, og[i:i + n], self.conv.wd, xg[i:i + n], red=self.red)


class DwConv(Op):
    """A depthwise 3x3 convolution (groups == channels, padding == dilation, no bias): MobileNetV2's `dw` layers (VNL.py:427-444).
    Its weight keeps its exact [C][1][3][3] shape in the flat store (NetStore `raw`): the fp32 master IS the kernels' [C][9]
    operand and its gradient slice their output; the BatchNorm behind it takes its statistics from a reduction pass."""

    def __init__(self, eng, x, w, stride=1, dil=1):
        assert tuple(w.shape) == (x.C, 1, 3, 3) and eng.store.sdims[id(w)] == (x.C, 3, 3, 1), (tuple(w.shape), x.C)
        self.eng, self.x, self.stride, self.dil = eng, x, stride, dil
        self.off, self.n = eng.store.p_off[id(w)], x.C * 9
        self.w32 = eng.store.P[self.off:self.off + self.n]
        self.out = Act(eng.dev, x.N, (x.H - 1) // stride + 1, (x.W - 1) // stride + 1, x.C)

    def acts(self):
        return (self.out,)

    def grad_ranges(self):
        return [(self.off, self.off + self.n)]

    def fwd(self, train):
        x, o = self.x, self.out
        ops.dwconv3x3_fwd(x.t, x.ld, self.w32, o.t, o.ld, x.N, x.H, x.W, x.C, self.stride, self.dil)

    def bwd(self):
        x, o = self.x, self.out
        ops.dwconv3x3_wgrad(x.t, x.ld, o.g, _ldg(o), self.eng.store.Gcur[self.off:self.off + self.n], x.N, x.H, x.W, x.C, self.stride, self.dil)
        ops.dwconv3x3_dgrad(o.g, _ldg(o), self.w32, x.g, _ldg(x), x.N, x.H, x.W, x.C, self.stride, self.dil, accumulate=_take(x))


class BN(Op):
    """out = act(bn(c) [+ res | + bn_r(res)]) with the batch statistics the producing Conv accumulated (training) or the
    running ones (eval).  `bias`: the conv in front had a bias (VNL.py:336 FTB_block.conv2): under batch statistics it
    cancels (zero gradient, only the running mean sees it); in eval mode it joins the shift."""

    def __init__(self, eng, c, site, relu, out=None, res=None, res_site=None, bias=None):
        self.eng, self.c, self.site, self.relu, self.res, self.res_site = eng, c, site, relu, res, res_site
        self.out = out if out is not None else Act(eng.dev, c.N, c.H, c.W, c.C)
        self.own_out = out is None
        self.bias = eng.store.vec(bias)[0][:site.C] if bias is not None else None
        self.bits = (torch.empty(c.M * (c.C // 8), dtype=torch.uint8, device=eng.dev) if (relu and res is not None) else None)
        if res is not None and res_site is None and not relu:
            raise NotImplementedError("BN + identity residual without ReLU")
        self.reduced = False   # the backward sums come with the launch that completes d(out) (see _TRACE)
        self.skip_dres = False
        self.pre_elu = False   # eval mode: c holds the pre-activation of an ELU (Conv.eval_raw); this pass applies it in fp32

    def red_spec(self):
        s, c, rs = self.site, self.c, self.res_site
        if not self.own_out or self.out.parent is not None or s.C % 8:
            return None
        if self.res is None:
            return s.red_spec(c, self.relu, mask_from_x=self.relu)
        if rs is None:
            return s.red_spec(c, self.relu, relu_bits=self.bits)
        if self.bits is None:
            return None
        return ops.bn_red(c.t, s.smean, s.srstd, s.part_b, relu_bits=self.bits, x_ld=c.ld,
                          second=(self.res.t, rs.smean, rs.srstd, rs.part_b, self.res.ld))

    def acts(self):
        return (self.out,) if self.own_out else ()

    def grad_ranges(self):
        return _site_ranges(self.site) + (_site_ranges(self.res_site) if self.res_site is not None else [])

    def fwd(self, train):
        s, c, o = self.site, self.c, self.out
        rs, r = self.res_site, self.res
        bits = self.bits if (train and r is not None) else None
        if train:
            # (the statistics' finalize runs inside the apply launch unless the store says otherwise: BNSite.apply)
            s.apply(self.eng, c.t, c.ld, o.t, o.ld, c.M, self.relu, True, r=r.t if r is not None else None, ldr=r.ld if r is not None else 0,
                    res_site=rs, relu_bits=bits)
            if self.bias is not None:
                with torch.no_grad():          # running mean of (conv + bias), after the statistics' own update of it
                    s.rmean.add_(self.bias, alpha=s.bn.momentum if s.bn.momentum is not None else 0.1)
            return
        s.finalize(c.M, False)
        if rs is not None:
            rs.finalize(r.M, False)
        if self.bias is not None:
            with torch.no_grad():
                s.shift.addcmul_(s.scale, self.bias)
        s.apply(self.eng, c.t, c.ld, o.t, o.ld, c.M, 2 if self.pre_elu else self.relu, False, r=r.t if r is not None else None,
                ldr=r.ld if r is not None else 0, res_site=rs, relu_bits=bits, finalized=True)

    def bwd(self):
        c, o, res = self.c, self.out, self.res
        acc = _take(c)
        if res is None:
            self.site.backward(o.g, o, c, self.relu, c.g, accumulate=acc, mask_from_x=self.relu, reduced=self.reduced)
            return
        assert not acc, "a pre-BN tensor joined with a residual has one consumer"
        if self.res_site is None:
            assert not res.gw, "identity-residual gradient must be the first writer"
            if self.skip_dres:           # the convolution that completes d(res) adds the masked d(out) itself (Conv.first_writer)
                self.site.backward(o.g, o, c, self.relu, c.g, relu_bits=self.bits, reduced=self.reduced)
                return
            res.gw = True
            self.site.backward(o.g, o, c, self.relu, c.g, dres=res.g, relu_bits=self.bits, reduced=self.reduced)
        else:
            assert not res.gw
            res.gw = True
            o.reduced = self.reduced
            bn_join_backward(self.site, self.res_site, o.g, o, c, res, c.g, res.g, self.bits)


class Pw(Op):
    """out = act(x + bias + r): conv bias, activation and residual add in one pass (MiDaS.py:163-229, VNL.py:348-349,
    Bts.py:69-80).  bias: a 1-D Parameter (or None); r: another activation (or None)."""

    def __init__(self, eng, x, bias=None, r=None, act=None, out=None):
        self.eng, self.x, self.r, self.act = eng, x, r, act
        self.out = out if out is not None else Act(eng.dev, x.N, x.H, x.W, x.C)
        self.own_out = out is None
        self.bias, self.b_off = (None, None)
        if bias is not None:
            v, off = eng.store.vec(bias)
            assert v.numel() == x.C, (v.numel(), x.C)
            self.bias, self.b_off = v, off
        self.part = ops.new_stat_buffer(x.C, eng.dev) if bias is not None else None

    def acts(self):
        return (self.out,) if self.own_out else ()

    def grad_ranges(self):
        return [(self.b_off, self.b_off + self.x.C)] if self.bias is not None else []

    def fwd(self, train):
        x, r, o = self.x, self.r, self.out
        ops.pw_fwd(x.t, x.ld, self.bias, r.t if r is not None else None, r.ld if r is not None else 0, o.t, o.ld, x.M, x.C, self.act)

    def bwd(self):
        x, r, o = self.x, self.r, self.out
        acc_x = _take(x)
        acc_r = _take(r) if r is not None else False
        dbias = self.eng.store.Gcur[self.b_off:self.b_off + x.C] if self.bias is not None else None
        ops.pw_bwd(o.g, _ldg(o), o.t, o.ld, x.g, _ldg(x), acc_x, r.g if r is not None else None, _ldg(r) if r is not None else 0,
                   acc_r, dbias, x.M, x.C, self.act, bias_part=self.part)


def _ldg(a):
    """Pixel stride of an activation's gradient tensor (slices share their parent's)."""
    return a.g.stride(2) if a is not None else 0


class Resize(Op):
    """F.interpolate(mode='bilinear', align_corners=...) to (OH, OW)."""

    def __init__(self, eng, x, OH, OW, align_corners, out=None):
        self.x, self.align = x, align_corners
        self.out = out if out is not None else Act(eng.dev, x.N, OH, OW, x.C)
        self.own_out = out is None

    def acts(self):
        return (self.out,) if self.own_out else ()

    def fwd(self, train):
        x, o = self.x, self.out
        ops.resize_bilinear_fwd(x.t, x.ld, o.t, o.ld, x.N, x.H, x.W, x.C, o.H, o.W, self.align)

    def bwd(self):
        x, o = self.x, self.out
        acc = _take(x)
        ops.resize_bilinear_bwd(o.g, _ldg(o), x.g, _ldg(x), x.N, x.H, x.W, x.C, o.H, o.W, self.align, acc)


class GlobalAvgPool(Op):
    """nn.AdaptiveAvgPool2d(1): [N][H][W][C] -> [N][1][1][C] (written into `out`, possibly a channel slice)."""

    def __init__(self, eng, x, out=None):
        self.x = x
        self.out = out if out is not None else Act(eng.dev, x.N, 1, 1, x.C)
        self.own_out = out is None

    def acts(self):
        return (self.out,) if self.own_out else ()

    def fwd(self, train):
        x, o = self.x, self.out
        ops.spatial_sum(x.t, x.ld, x.N, x.H * x.W, x.C, 1.0 / (x.H * x.W), o.t, o.ld)

    def bwd(self):
        x, o = self.x, self.out
        acc = _take(x)
        ops.spatial_bcast(o.g, _ldg(o), 1.0 / (x.H * x.W), x.g, _ldg(x), x.N, x.H * x.W, x.C, acc)


class Broadcast(Op):
    """[N][1][1][C] -> [N][H][W][C]: F.interpolate of a 1x1 map (VNL.py:225)."""

    def __init__(self, eng, v, out):
        self.v, self.out = v, out

    def fwd(self, train):
        v, o = self.v, self.out
        ops.spatial_bcast(v.t, v.ld, 1.0, o.t, o.ld, o.N, o.H * o.W, o.C)

    def bwd(self):
        v, o = self.v, self.out
        assert not v.gw
        v.gw = True
        ops.spatial_sum(o.g, _ldg(o), o.N, o.H * o.W, o.C, 1.0, v.g, _ldg(v))


class Gate(Op):
    """AFA_block's output (VNL.py:372): out = w * lateral + top, w: [N][1][1][C]."""

    def __init__(self, eng, w, lat, top):
        self.w, self.lat, self.top = w, lat, top
        self.out = Act(eng.dev, lat.N, lat.H, lat.W, lat.C)

    def acts(self):
        return (self.out,)

    def fwd(self, train):
        w, l, t, o = self.w, self.lat, self.top, self.out
        ops.gate_fwd(w.t, w.ld, l.t, l.ld, t.t, t.ld, o.t, o.ld, l.N, l.H * l.W, l.C)

    def bwd(self):
        w, l, t, o = self.w, self.lat, self.top, self.out
        assert not w.gw
        w.gw = True
        acc_l, acc_t = _take(l), _take(t)
        ops.gate_bwd(o.g, _ldg(o), w.t, w.ld, l.t, l.ld, l.g, _ldg(l), acc_l, t.g, _ldg(t), acc_t, w.g, _ldg(w), l.N, l.H * l.W, l.C)


class SoftmaxHead(Op):
    """fcn_topdown_predict's tail (VNL.py:325-327): logits = x + bias, softmax over channels; both fp32 NCHW outputs."""

    def __init__(self, eng, x, bias, C):
        self.eng, self.x, self.C = eng, x, C
        self.bias, self.b_off = eng.store.vec(bias)
        self.logit = torch.empty(x.N, C, x.H, x.W, device=eng.dev)
        self.prob = torch.empty(x.N, C, x.H, x.W, device=eng.dev)
        self.outputs = (self.logit, self.prob)
        self.douts = [None, None]
        self.fresh = False
        self.serial = 0          # which forward the current outputs belong to (criteria's private route checks it)
        self.stash = None        # what the criterion-fused route leaves for bwd (criteria._FusedDepthFunction / _FusedWcelFunction)
        self._tmp = self._part = None

    def tag(self, logit, prob):
        """Mark the tensors the module hands out as THIS forward's head outputs: criteria.bins_to_depth / WCEL_Loss then read the
        head's 16-bit input instead of walking 2 x 2.95 GB of fp32 planes, and their backward leaves its pieces in `stash` for
        bwd to turn into d(input) in one launch (include/mde_hip.h: mde_vnl_head_*).  A tensor derived from these (a clone, a
        slice, another dtype) carries no mark and takes the general route."""
        import weakref
        ref = (weakref.ref(self), self.serial)
        logit._mde_head, prob._mde_head = ref, ref

    def new_outputs(self):
        """The module path hands its outputs to the caller, who may keep them across steps: these two are 2.95 GB each at
        configuration 5's size, so instead of cloning what a plan-owned buffer holds (2 x 5.9 GB of copy traffic per step, 2.4 ms)
        every forward of the module path writes into tensors of its own (the caching allocator recycles them) and hands THOSE out."""
        x = self.x
        self.logit = torch.empty(x.N, self.C, x.H, x.W, device=self.eng.dev)
        self.prob = torch.empty(x.N, self.C, x.H, x.W, device=self.eng.dev)
        self.outputs = (self.logit, self.prob)
        self.fresh = True
        self.serial += 1
        self.stash = None

    def grad_ranges(self):
        return [(self.b_off, self.b_off + self.C)]

    def fwd(self, train):
        x = self.x
        ops.softmax_head_fwd(x.t, x.ld, self.bias, self.logit, self.prob, x.N, x.H * x.W, self.C)

    def bwd(self):
        x = self.x
        assert not x.gw
        x.gw = True
        dl, dp = self.douts
        dbias = self.eng.store.Gcur[self.b_off:self.b_off + self.C]
        st, self.stash = self.stash, None
        if st is not None and st.get("serial") == self.serial and (st.get("gdepth") is not None or st.get("gscale") is not None):
            # the criterion took the private route: its backward left (gdepth, depth, log10 depth, border) and / or (bins, weight,
            # ws, gscale) here and returned zero-stride placeholders for d(logit) / d(softmax)
            w = st.get("wcel") if st.get("gscale") is not None else None
            gd = st.get("gdepth")
            ops.vnl_head_bwd(x.t, x.ld, self.bias, w[0] if w else None, w[1] if w else None, w[2] if w else None, st.get("gscale"),
                             st["lse"], st.get("depth") if gd is not None else None, st.get("l10") if gd is not None else None, gd,
                             st.get("border") if gd is not None else None, x.M, self.C, x.g, _ldg(x))
            # the bias gradient = the column sums of d(input): one reduction pass over the 16-bit tensor just written
            if self._part is None:
                self._part = ops.new_stat_buffer(x.g.shape[-1], self.eng.dev)
            ops.bn_stats(x.g, x.M, x.g.shape[-1], _ldg(x), self._part)
            with torch.no_grad():
                dbias.add_(self._part[:, 0, :self.C].sum(0))
                self._part.zero_()
            if dl is not None or dp is not None:           # gradients from OTHER consumers of the public tensors: the general pass, added
                if self._tmp is None:
                    self._tmp = torch.empty_like(x.g)
                ops.softmax_head_bwd(dl, dp, self.prob, self._tmp, _ldg(x), dbias, x.N, x.H * x.W, self.C)
                ops.pw_fwd(self._tmp, _ldg(x), None, x.g, _ldg(x), x.g, _ldg(x), x.M, x.g.shape[-1], None)
            return
        if dl is None and dp is None:
            x.g.zero_()
            return
        ops.softmax_head_bwd(dl, dp, self.prob, x.g, _ldg(x), dbias, x.N, x.H * x.W, self.C)


class ToNCHW(Op):
    """out = scale * act(x + bias) as fp32 NCHW: the small-channel output heads (MiDaS.py:54-56, Bts.py:202-203)."""

    def __init__(self, eng, x, bias, C, act, scale=1.0):
        self.eng, self.x, self.C, self.act, self.scale = eng, x, C, act, scale
        self.bias, self.b_off = eng.store.vec(bias) if bias is not None else (None, None)
        self.y = torch.empty(x.N, C, x.H, x.W, device=eng.dev)
        self.outputs = (self.y,)
        self.douts = [None]

    def grad_ranges(self):
        return [(self.b_off, self.b_off + self.C)] if self.bias is not None else []

    def fwd(self, train):
        x = self.x
        ops.to_nchw_act_fwd(x.t, x.ld, self.bias, self.y, x.N, x.H * x.W, self.C, self.act, self.scale)

    def bwd(self):
        x = self.x
        assert not x.gw
        x.gw = True
        dbias = self.eng.store.Gcur[self.b_off:self.b_off + self.C] if self.bias is not None else None
        ops.to_nchw_act_bwd(self.douts[0], self.y, x.g, _ldg(x), dbias, x.N, x.H * x.W, self.C, self.act, self.scale)


class HeadConvMap(Op):
    """A 3x3 / pad 1 convolution to ONE channel with an activation and a scale, as an fp32 map (NCHW == NHWC for one channel):
    BTS' get_depth -> Sigmoid -> x max_depth (Bts.py:168,262) on the head kernels of csrc/conv_small.hip -- fp32 accumulation
    straight to the fp32 result -- instead of a 64-column GEMM tile for 1 useful column (16 x 480 x 640 pixels, 32 channels:
    390 + 389 + 731 us forward / input gradient / weight gradient on the GEMM kernels)."""

    def __init__(self, eng, x, weight, act, scale=1.0):
        assert weight.shape[0] == 1 and tuple(weight.shape[2:]) == (3, 3) and x.C in (8, 16, 32, 64) and x.ld == x.C and weight.shape[1] == x.C
        self.eng, self.x, self.act, self.scale = eng, x, act, float(scale)
        self.off, self.n = eng.store.p_off[id(weight)], 9 * x.C           # (fp32 master [O padded][3][3][C]: row 0 is the filter)
        self.w32 = eng.store.P[self.off:self.off + self.n]
        self.pre = torch.empty(x.N, 1, x.H, x.W, device=eng.dev)
        self.y = torch.empty(x.N, 1, x.H, x.W, device=eng.dev)
        self.outputs, self.douts = (self.y,), [None]

    def grad_ranges(self):
        return [(self.off, self.off + self.n)]

    def fwd(self, train):
        x = self.x
        ops.head_conv_fwd(x.t, self.w32, self.pre, x.N, x.H, x.W, x.C, 1)
        ops.map_act_fwd(self.pre, self.y, self.act, self.scale)

    def bwd(self):
        x = self.x
        assert not x.gw
        x.gw = True
        if self.douts[0] is None:              # this output took no part in the loss: no gradient flows through the head
            x.g.zero_()
            return
        ops.map_act_bwd(self.douts[0], self.y, self.pre, self.act, self.scale)           # (pre: its own gradient from here on)
        ops.head_conv_bwd(x.t, self.w32, self.pre, x.g, self.eng.store.Gcur[self.off:self.off + self.n], x.N, x.H, x.W, x.C, 1)


class ImageResidualHead(ToNCHW):
    """BTS' final_depth with image_residuals (Bts.py:264-271): the sigmoid head's ten channels, the colour ones as residuals on
    the input image.  The image is the plan's input (`eng.stem.x`), read as it is (fp32 NCHW)."""

    def __init__(self, eng, x, C):
        super().__init__(eng, x, None, C, "sigmoid", 1.0)
        self.d = self.y                                   # the sigmoid channels
        self.y = torch.empty_like(self.d)
        self.outputs = (self.y,)
        self.dd = torch.empty_like(self.d)

    def fwd(self, train):
        x = self.x
        ops.to_nchw_act_fwd(x.t, x.ld, None, self.d, x.N, x.H * x.W, self.C, self.act, self.scale)
        ops.image_residual_fwd(self.d, self.eng.stem.x, self.y)

    def bwd(self):
        x = self.x
        assert not x.gw
        x.gw = True
        ops.image_residual_bwd(self.douts[0], self.d, self.eng.stem.x, self.dd)
        ops.to_nchw_act_bwd(self.dd, self.d, x.g, _ldg(x), None, x.N, x.H * x.W, self.C, self.act, self.scale)


# ---------------------------------------------------------------------------------------------- ops of the BTS / DenseNet plans
class ImageStem(Op):
    """A k x k stem conv on the image with any number of output channels (densenet161's 7x7/2 conv0: 96 channels, Bts.py:289;
    DORN's 3x3/2 conv1, Dorn.py:224): the fp32 NCHW image goes to NHWC bf16 with its 3 channels zero-padded to 8, the taps
    run as launches of the GEMM kernel of at most 32 taps each (7x7: 32 + 17, the second accumulating), the BatchNorm
    statistics come from a stand-alone reduction.  No input gradient."""

    def __init__(self, eng, conv, site, N, H, W, xin=None):
        """xin: the converted image of another ImageStem of the same plan (Eigen reads the image with three convs): shared,
        and converted once by the stem that owns it."""
        self.eng, self.site = eng, site
        self.w = eng._conv([conv.weight], need_dgrad=False)
        O, Cp = self.w.O, self.w.I
        k, st, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        assert Cp == 8 and conv.kernel_size == (k, k) and conv.stride == (st, st) and conv.padding == (pd, pd) and conv.dilation == (1, 1)
        H2, W2 = ops.out_size(H, k, st, pd), ops.out_size(W, k, st, pd)
        self.owns_xin = xin is None
        self.xin = Act(eng.dev, N, H, W, Cp) if xin is None else xin
        assert (self.xin.N, self.xin.H, self.xin.W, self.xin.C) == (N, H, W, Cp)
        self.out = Act(eng.dev, N, H2, W2, O)
        taps = [(i - pd, j - pd, i * k + j) for i in range(k) for j in range(k)]
        ks = eng._ksplit(N * H2 * W2, O, Cp, min(32, k * k))
        self.fd, self.wd = [], []
        self.fd16 = []          # eval mode: the same launches over the 16-slot hi / lo operands (mde_nchw_to_nhwc_split16)
        for t0 in range(0, k * k, 32):
            part = taps[t0:t0 + 32]
            self.fd.append(ops.conv_desc(N, H, W, Cp, Cp, self.xin.nbytes, H2, W2, st, st, part, k * k, H2, W2, O, ncols=O, accumulate=t0 > 0))
            self.fd16.append(ops.conv_desc(N, H, W, 16, 16, N * H * W * 16 * 2, H2, W2, st, st, part, k * k, H2, W2, O, ncols=O, accumulate=t0 > 0))
            self.wd.append(ops.wgrad_desc(N, H2, W2, O, O, self.out.nbytes, H, W, Cp, Cp, self.xin.nbytes, st, st, part, k * k, False, ks))
        self.cin, self.ntaps = conv.in_channels, k * k
        self.w16 = None
        self.x = None

    def acts(self):
        return (self.out,)

    def grad_ranges(self):
        return [(self.w.off, self.w.off + self.w.n)]

    def fwd(self, train):
        if self.eng.split and 3 * self.cin <= 16:
            # eval: the fp32 image (hi + lo) against the two-term weight shadow, one contraction over 16 channel slots
            xin = self.xin
            if getattr(xin, "split16", None) is None:
                xin.split16 = torch.empty(xin.N, xin.H, xin.W, 16, dtype=ops.ACT_DTYPE, device=self.eng.dev)
            if self.owns_xin:
                ops.nchw_to_nhwc_split16(self.x, xin.split16)
            if self.w16 is None:
                self.w16 = torch.empty(self.w.O * self.ntaps * 16, dtype=ops.ACT_DTYPE, device=self.eng.dev)
            ops.stem_weight_split16(self.w.w32, self.w16, self.w.O * self.ntaps, self.w.I, self.cin)
            for d in self.fd16:
                ops.conv_gemm(d, xin.split16, self.w16, self.out.t)
            return
        if self.owns_xin:
            ops.nchw_to_nhwc_bf16_pad(self.x, self.xin.t, self.xin.C)
        for d in self.fd:
            self.eng.fwd_conv(d, self.xin.t, self.w, self.out.t)
        if train and self.site is not None:               # (Eigen's 9 x 9 / 2 image convs have no BatchNorm: Eigen.py:22,51)
            ops.bn_stats(self.out.t, self.out.M, self.out.C, self.out.ld, self.site.part)

    def bwd(self):
        for d in self.wd:
            self.eng.wgrad(d, self.out.g, self.xin.t, self.w.dw)


class Stem7(Op):
    """A 7x7 / 2 / pad 3 convolution of the fp32 NCHW image to 64 or 96 channels on the stem kernels (csrc/conv_small.hip:
    the image patch staged once per tile as bf16 hi + lo pairs, BatchNorm sums from the epilogue): densenet161's conv0
    (Bts.py:289, 96 channels), where the general ImageStem -- 49 taps over 8 zero-padded channel slots through the GEMM
    kernel, 64-column tiles -- spent 0.8 ms forward and 2.3 ms on the weight gradient of a 16-image step against 0.15 / 0.2 ms
    here.  The weight keeps its exact [O][7][7][3] shape in the flat store (NetStore `raw`).  Eval with the two-term weight
    shadow goes through ImageStem's 16-slot hi / lo operands (GEMM kernel).  No input gradient."""

    def __init__(self, eng, conv, site, N, H, W):
        O = conv.out_channels
        assert conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.in_channels == 3 and O in (64, 96)
        assert H % 2 == 0 and W % 2 == 0, "stem kernel: even image sizes"
        assert eng.store.sdims[id(conv.weight)] == (O, 7, 7, 3), "Stem7: the weight must be stored raw"
        self.eng, self.site, self.O = eng, site, O
        self.w = eng._conv([conv.weight], need_dgrad=False)
        H2, W2 = H // 2, W // 2
        self.out = Act(eng.dev, N, H2, W2, O)
        taps = [(i - 3, j - 3, i * 7 + j) for i in range(7) for j in range(7)]
        self.fd16 = [ops.conv_desc(N, H, W, 16, 16, N * H * W * 16 * 2, H2, W2, 2, 2, taps[t0:t0 + 32], 49, H2, W2, O, ncols=O, accumulate=t0 > 0)
                     for t0 in (0, 32)]
        self.split16 = self.w16 = self.x = None

    def acts(self):
        return (self.out,)

    def grad_ranges(self):
        return [(self.w.off, self.w.off + self.w.n)]

    def fwd(self, train):
        o = self.out
        if self.eng.split:
            if self.split16 is None:
                self.split16 = torch.empty(o.N, 2 * o.H, 2 * o.W, 16, dtype=ops.ACT_DTYPE, device=self.eng.dev)
                self.w16 = torch.empty(self.O * 49 * 16, dtype=ops.ACT_DTYPE, device=self.eng.dev)
            ops.nchw_to_nhwc_split16(self.x, self.split16)
            ops.stem_weight_split16(self.w.w32, self.w16, self.O * 49, 3, 3)
            for d in self.fd16:
                ops.conv_gemm(d, self.split16, self.w16, o.t)
            return
        ops.stem_conv_fwd(self.x, self.w.w32, o.t, self.site.part if (train and self.site is not None) else None, self.O)

    def bwd(self):
        ops.stem_conv_wgrad(self.x, self.out.g, self.w.dw, self.O)


def stem7_weights(module):
    """Names of the 7x7 / 2 image convs Stem7 can run (3 -> 64 or 96 channels): for a NetStore's `raw` list."""
    return [n + ".weight" for n, m in module.named_modules()
            if isinstance(m, torch.nn.Conv2d) and m.in_channels == 3 and m.out_channels in (64, 96) and m.kernel_size == (7, 7)
            and m.stride == (2, 2) and m.padding == (3, 3) and m.dilation == (1, 1) and m.groups == 1 and m.bias is None]


def image_stem(eng, conv, site, N, H, W):
    """Stem7 where the conv's weight is stored raw (see stem7_weights) and the image size is even, ImageStem otherwise."""
    if eng.store.sdims[id(conv.weight)] == (conv.out_channels, 7, 7, 3):
        return Stem7(eng, conv, site, N, H, W)
    return ImageStem(eng, conv, site, N, H, W)


class MaxPool(Op):
    """nn.MaxPool2d(3, 2, 1[, ceil_mode=True]) on a contiguous NHWC tensor."""

    def __init__(self, eng, x, ceil_mode=False):
        assert x.parent is None
        self.x, self.ceil = x, ceil_mode
        H2, W2 = ops.maxpool_out_size(x.H, ceil_mode), ops.maxpool_out_size(x.W, ceil_mode)
        self.out = Act(eng.dev, x.N, H2, W2, x.C)
        self.idx = torch.empty(x.N, H2, W2, x.C, dtype=torch.uint8, device=eng.dev)
        self.tmp = None

    def acts(self):
        return (self.out,)

    def fwd(self, train):
        x = self.x
        ops.maxpool_fwd(x.t, self.out.t, self.idx, x.N, x.H, x.W, x.C, self.ceil)

    def bwd(self):
        x = self.x
        if not _take(x):
            ops.maxpool_bwd(self.out.g, self.idx, x.g, x.N, x.H, x.W, x.C, self.ceil)
            return
        # the input also feeds a later consumer (densenet's relu0 is a decoder skip, Bts.py:206,250): route into a scratch
        # tensor, then one in-place add pass
        if self.tmp is None:
            self.tmp = torch.empty_like(x.t)
        ops.maxpool_bwd(self.out.g, self.idx, self.tmp, x.N, x.H, x.W, x.C, self.ceil)
        ops.pw_fwd(self.tmp, x.C, None, x.g, _ldg(x), x.g, _ldg(x), x.M, x.C, None)


class MaxPoolView(Op):
    """nn.MaxPool2d(k, s) (no padding) over the spatial view x[:, y0:y0 + Hv, x0:x0 + Wv] of an NHWC tensor, the crops of
    Eigen's stacks folded into the view: VGG-19-BN's MaxPool2d(2, 2) (whole tensor), scale 2's `pool(x)[:, :, 1:-1, 1:-1]`
    (Eigen.py:41: MaxPool2d(3, 2), output rows 1 .. OH - 2 = windows starting at 2, 4, ...: view origin (2, 2)) and scale 3's
    `conv(img)[:, :, 2:-3, 2:-3]` -> ReLU -> MaxPool2d(3, 1) (Eigen.py:65-67: the ReLU commutes with the crop).  `out` may be
    a channel slice of a concatenation.  Pixels the view leaves out receive a zero gradient."""

    def __init__(self, eng, x, k, s, y0=0, x0=0, Hv=None, Wv=None, out=None):
        assert x.parent is None, "the view is taken on a tensor of its own"
        self.x, self.k, self.s, self.y0, self.x0 = x, k, s, y0, x0
        self.Hv, self.Wv = (x.H - y0 if Hv is None else Hv), (x.W - x0 if Wv is None else Wv)
        assert 0 <= y0 and 0 <= x0 and y0 + self.Hv <= x.H and x0 + self.Wv <= x.W and self.Hv >= k and self.Wv >= k
        OH, OW = (self.Hv - k) // s + 1, (self.Wv - k) // s + 1
        self.out = out if out is not None else Act(eng.dev, x.N, OH, OW, x.C)
        assert (self.out.N, self.out.H, self.out.W, self.out.C) == (x.N, OH, OW, x.C)
        self.own_out = out is None
        self.idx = torch.empty(x.N, OH, OW, x.C, dtype=torch.uint8, device=eng.dev)
        self.partial = (y0, x0, self.Hv, self.Wv) != (0, 0, x.H, x.W) or (OH - 1) * s + k < self.Hv or (OW - 1) * s + k < self.Wv

    def acts(self):
        return (self.out,) if self.own_out else ()

    def _view(self, t):
        return t[:, self.y0:, self.x0:]

    def fwd(self, train):
        x, o = self.x, self.out
        ops.maxpool_view_fwd(self._view(x.t), x.ld, x.W, x.H * x.W, self.Hv, self.Wv, o.t, o.ld, self.idx, x.N, x.C, self.k, self.s)

    def bwd(self):
        x, o = self.x, self.out
        acc = _take(x)
        if self.partial and not acc:
            x.g.zero_()                                  # what the view (or a stride that does not reach the edge) leaves out
            acc = True
        ops.maxpool_view_bwd(o.g, _ldg(o), self.idx, self._view(x.g), _ldg(x), x.W, x.H * x.W, self.Hv, self.Wv, x.N, x.C, self.k, self.s,
                             accumulate=acc)


class Unflatten(Op):
    """`x.reshape(-1, C, h, w)` of a Linear layer's output row (Eigen.py:87): the [N][1][1][C h w] row is in the NCHW order of
    the reference's reshape, the convolutions want [N][h][w][C].  The inverse of PooledFlat's layout change (pool 1 x 1, no
    dropout), so it is that op's two kernels with forward and backward exchanged."""

    def __init__(self, eng, x, C, h, w):
        assert (x.H, x.W) == (1, 1) and x.C == C * h * w and x.parent is None
        self.x = x
        self.out = Act(eng.dev, x.N, h, w, C)
        self.ones = torch.ones(x.N, C, device=eng.dev)

    def acts(self):
        return (self.out,)

    def fwd(self, train):
        o = self.out
        ops.avgpool_flat_bwd(self.x.t, self.ones, o.t, o.ld, o.N, o.H, o.W, o.C, 1, 1, 0)

    def bwd(self):
        o, x = self.out, self.x
        assert not _take(x), "the Linear layer's output row has one consumer"
        ops.avgpool_flat_fwd(o.g, _ldg(o), self.ones, x.g, o.N, o.H, o.W, o.C, 1, 1, 0)


class Nearest2(Op):
    """F.interpolate(scale_factor=2, mode='nearest') (Bts.py:77) / nn.Upsample(scale_factor=2) (MyNet.py:20,44)."""

    def __init__(self, eng, x, out=None):
        self.x = x
        self.out = out if out is not None else Act(eng.dev, x.N, 2 * x.H, 2 * x.W, x.C)
        self.own_out = out is None

    def acts(self):
        return (self.out,) if self.own_out else ()

    def fwd(self, train):
        x, o = self.x, self.out
        ops.nearest2_fwd(x.t, x.ld, o.t, o.ld, x.N, x.H, x.W, x.C)

    def bwd(self):
        x, o = self.x, self.out
        ops.sum2x2(o.g, _ldg(o), x.g, _ldg(x), x.N, x.H, x.W, x.C, 1.0, _take(x))


class AvgPool2(Op):
    """nn.AvgPool2d(2, 2) (densenet transitions), writing `out` (a channel slice of the next block's tensor)."""

    def __init__(self, eng, x, out=None):
        self.x = x
        self.out = out if out is not None else Act(eng.dev, x.N, x.H // 2, x.W // 2, x.C)
        self.own_out = out is None
        assert x.H % 2 == 0 and x.W % 2 == 0

    def acts(self):
        return (self.out,) if self.own_out else ()

    def fwd(self, train):
        x, o = self.x, self.out
        ops.sum2x2(x.t, x.ld, o.t, o.ld, o.N, o.H, o.W, o.C, 0.25)

    def bwd(self):
        x, o = self.x, self.out
        ops.spread2x2(o.g, _ldg(o), x.g, _ldg(x), o.N, o.H, o.W, o.C, 0.25, _take(x))


class StatsPass(Op):
    """Batch statistics of a tensor no conv epilogue produced (a BatchNorm after an ELU: Bts.py:209,215,219): one reduction
    pass into the site's partial sums, training mode only."""

    def __init__(self, eng, x, site):
        self.x, self.site = x, site

    def fwd(self, train):
        if train:
            x = self.x
            ops.bn_stats(x.t, x.M, x.C, x.ld, self.site.part)

    def bwd(self):
        pass


class Moments(Op):
    """Batch mean / biased variance of `x` (channels [c0, c0 + x.C) of a DenseNet block) into the block's moment vectors:
    from `part` if a conv epilogue already accumulated the sums, else by a reduction pass.  Training mode only."""

    def __init__(self, eng, x, mean, var, c0, part=None):
        self.x, self.mean, self.var = x, mean[c0:c0 + x.C], var[c0:c0 + x.C]
        self.part, self.own = (part, False) if part is not None else (ops.new_stat_buffer(x.C, eng.dev), True)

    def fwd(self, train):
        if train:
            x = self.x
            if self.own:
                ops.bn_stats(x.t, x.M, x.C, x.ld, self.part)
            ops.bn_moments(self.part, x.M, x.C, self.mean, self.var)

    def bwd(self):
        pass


class PrefixBN(Op):
    """BatchNorm (+ ReLU) over the first `x.C` channels of a concatenation whose batch moments are already known (DenseNet's
    norm1 / transition norm / norm5: torchvision densenet161 as Bts.py:283-292 uses it).  Output is its own tensor; the input
    gradient ACCUMULATES into the concatenation's gradient (every layer of the block adds its share).  Sites wider than the
    kernels' 2048 channels run as equal channel chunks."""

    def __init__(self, eng, x, bn, mean, var, relu=True, out=None):
        self.eng, self.x, self.relu = eng, x, relu
        self.out = out if out is not None else Act(eng.dev, x.N, x.H, x.W, x.C)
        self.own_out = out is None
        C = x.C
        nch = -(-C // 2048)
        step = -(-(C // 8) // nch) * 8
        self.chunks = []
        for c0 in range(0, C, step):
            cn = min(step, C - c0)
            self.chunks.append((eng._site_chunk(bn, c0, cn), x.slice(c0, cn), self.out.slice(c0, cn), mean[c0:c0 + cn], var[c0:c0 + cn]))
        self.reduced = False

    def red_spec(self):
        if len(self.chunks) != 1 or not self.own_out or self.out.parent is not None or self.x.C % 8:
            return None
        s, xs = self.chunks[0][:2]
        return s.red_spec(xs, self.relu, mask_from_x=self.relu)

    def acts(self):
        return (self.out,) if self.own_out else ()

    def grad_ranges(self):
        return [r for c in self.chunks for r in _site_ranges(c[0])]

    def fwd(self, train):
        for s, xs, os_, mean, var in self.chunks:
            s.apply(self.eng, xs.t, xs.ld, os_.t, os_.ld, xs.M, self.relu, train, mean=mean, var=var)

    def bwd(self):
        acc = _take(self.x)
        for s, xs, os_, _, _ in self.chunks:
            s.backward(os_.g, os_, xs, self.relu, xs.g, accumulate=acc, mask_from_x=self.relu, reduced=self.reduced)


class F32Map:
    """A one-channel fp32 map [N][1][H][W] with a gradient that every consumer ADDS into (zeroed, or set to the caller's
    output gradient, at the start of a backward pass)."""

    def __init__(self, dev, N, H, W):
        self.N, self.H, self.W = N, H, W
        self.t = torch.empty(N, 1, H, W, device=dev)
        self.g = torch.zeros(N, 1, H, W, device=dev)


class PlaneDepth(Op):
    """reduction_1x1's plane parameters -> local planar guidance depth / max_depth (Bts.py:105-146,228-232), an output head."""

    def __init__(self, eng, x, up, max_depth):
        self.x, self.up, self.max_depth = x, up, max_depth
        self.map = F32Map(eng.dev, x.N, x.H * up, x.W * up)
        self.outputs, self.douts = (self.map.t,), [None]

    def fwd(self, train):
        x = self.x
        ops.plane_depth_fwd(x.t, x.ld, self.map.t, x.N, x.H, x.W, self.up, self.max_depth)

    def bwd(self):
        x = self.x
        assert not x.gw
        x.gw = True
        if self.douts[0] is not None:
            self.map.g.add_(self.douts[0])
        ops.plane_depth_bwd(x.t, x.ld, self.map.g, x.g, _ldg(x), x.N, x.H, x.W, self.up, self.max_depth)
        self.map.g.zero_()


class MapSlot(Op):
    """torch.cat of a one-channel fp32 map into channel `ch` of a wider NHWC tensor, after F.interpolate(nearest,
    1/step) when step > 1 (Bts.py:234,238,247,251,263)."""

    def __init__(self, eng, m, cat, ch, step=1):
        self.m, self.cat, self.ch, self.step = m, cat, ch, step
        assert (cat.H * step, cat.W * step) == (m.H, m.W)

    def fwd(self, train):
        m, cat = self.m, self.cat
        ops.map_to_slot(m.t, cat.t[..., self.ch:], cat.ld, m.N, m.H, m.W, self.step)

    def bwd(self):
        m, cat = self.m, self.cat
        ops.slot_to_map_add(cat.g[..., self.ch:], _ldg(cat), m.g, m.N, m.H, m.W, self.step)


class SigmoidMap(Op):
    """reduction_1x1(is_final=True)'s tail (Bts.py:95-97,262): sigmoid of a one-channel conv output as an fp32 map that is
    both an output head and a concatenated feature."""

    def __init__(self, eng, x):
        self.x = x
        self.map = F32Map(eng.dev, x.N, x.H, x.W)
        self.outputs, self.douts = (self.map.t,), [None]

    def fwd(self, train):
        x = self.x
        ops.to_nchw_act_fwd(x.t, x.ld, None, self.map.t, x.N, x.H * x.W, 1, "sigmoid", 1.0)

    def bwd(self):
        x = self.x
        assert not x.gw
        x.gw = True
        if self.douts[0] is not None:
            self.map.g.add_(self.douts[0])
        ops.to_nchw_act_bwd(self.map.g, self.map.t, x.g, _ldg(x), None, x.N, x.H * x.W, 1, "sigmoid", 1.0)
        self.map.g.zero_()


# ---------------------------------------------------------------------------------------------- ops of the DORN plan
class ChannelDropout(Op):
    """nn.Dropout2d(p) (Dorn.py:59,107,109): whole channels of an image are zeroed with probability p, the rest scaled by
    1 / (1 - p); identity in eval mode.  The mask [N][C] is drawn with torch's device generator (torch.manual_seed governs
    it) unless `fixed` holds one (parity tests hand the same mask to the oracle); the kernels take it as data."""

    def __init__(self, eng, x, p, out=None):
        self.x, self.p = x, float(p)
        self.out = out if out is not None else Act(eng.dev, x.N, x.H, x.W, x.C)
        self.own_out = out is None
        self.mask = torch.ones(x.N, x.C, device=eng.dev)
        self.fixed = None

    def acts(self):
        return (self.out,) if self.own_out else ()

    def draw(self, train):
        if not train or self.p == 0.0:
            self.mask.fill_(1.0)
        elif self.fixed is not None:
            self.mask.copy_(self.fixed)
        elif self.p >= 1.0:
            self.mask.zero_()
        else:
            self.mask.bernoulli_(1.0 - self.p).div_(1.0 - self.p)

    def fwd(self, train):
        x, o = self.x, self.out
        self.draw(train)
        ops.chan_scale(x.t, x.ld, self.mask, o.t, o.ld, x.N, x.H * x.W, x.C)

    def bwd(self):
        x, o = self.x, self.out
        ops.chan_scale(o.g, _ldg(o), self.mask, x.g, _ldg(x), x.N, x.H * x.W, x.C, accumulate=_take(x))


class PooledFlat(ChannelDropout):
    """FullImageEncoder's front (Dorn.py:58-62,72-74): AvgPool2d(k, k, k // 2) -> Dropout2d -> `view(-1, C * h * w)`; the
    output is the [N][1][1][C * h * w] row nn.Linear contracts, in the NCHW order of the reference's flatten."""

    def __init__(self, eng, x, k, p):
        self.x, self.p, self.k, self.pad = x, float(p), k, k // 2
        self.oh, self.ow = (x.H + 2 * self.pad - k) // k + 1, (x.W + 2 * self.pad - k) // k + 1
        self.out = Act(eng.dev, x.N, 1, 1, x.C * self.oh * self.ow)
        self.own_out = True
        self.mask = torch.ones(x.N, x.C, device=eng.dev)
        self.fixed = None

    def fwd(self, train):
        x = self.x
        self.draw(train)
        ops.avgpool_flat_fwd(x.t, x.ld, self.mask, self.out.t, x.N, x.H, x.W, x.C, self.k, self.k, self.pad)

    def bwd(self):
        x = self.x
        ops.avgpool_flat_bwd(self.out.g, self.mask, x.g, _ldg(x), x.N, x.H, x.W, x.C, self.k, self.k, self.pad, accumulate=_take(x))


class OrdinalHead(Op):
    """OrdinalRegressionLayer (Dorn.py:288-318): outputs (decode_c int64 [N][1][H][W], ord_c1 fp32 [N][K][H][W])."""

    def __init__(self, eng, x, K):
        self.x, self.K = x, K
        self.label = torch.empty(x.N, 1, x.H, x.W, dtype=torch.int64, device=eng.dev)
        self.prob = torch.empty(x.N, K, x.H, x.W, device=eng.dev)
        self.outputs = (self.label, self.prob)
        self.douts = [None, None]

    def fwd(self, train):
        x = self.x
        ops.ordinal_fwd(x.t, x.ld, self.prob, self.label, x.N, x.H * x.W, self.K)

    def bwd(self):
        x = self.x
        assert not x.gw
        x.gw = True
        if self.douts[1] is None:
            x.g.zero_()
            return
        ops.ordinal_bwd(self.douts[1], x.t, x.ld, x.g, _ldg(x), x.N, x.H * x.W, self.K)


# ---------------------------------------------------------------------------------------------- ops of the MyNet plan
class ConvT(Op):
    """nn.ConvTranspose2d(Cin, Cout, k, stride=2, padding=p) with (2h + 2p - k) / 2 + 1 == h, no bias here (MyNet.py:61-63: k 4,
    p 1).  A transposed convolution IS the input gradient of the strided convolution with the same weight tensor
    [Cin][Cout][k][k] read as that convolution's [O][I][kh][kw] (engine.DeConvLayer): forward = the stride-2 output phases
    over the transposed packing, input gradient = that convolution's forward over d(out), weight gradient = its weight
    gradient with activation and output gradient exchanged."""

    def __init__(self, eng, x, w, k, pad, stride=2):
        """stride 2 with k = 2 pad + 2 doubles the size (MyNet); Eigen.py:79 upsamples by stride 4 with k 3 (14 x 19 -> 55 x 75:
        sixteen output phases, seven of them without a tap: bias only) and Eigen.py:34 ends scale 2 in k 5 / stride 2 /
        pad 2 (h -> 2 h - 1).  Output size (h - 1) stride - 2 pad + k, as nn.ConvTranspose2d without output_padding."""
        self.eng, self.x = eng, x
        self.w = eng.store.conv([w])                       # O = Cin, I = Cout (storage, possibly padded to 8)
        self.w.want_w2d()
        Cin, C = self.w.O, self.w.I
        N, h, wd, st = x.N, x.H, x.W, stride
        H2, W2 = (h - 1) * st - 2 * pad + k, (wd - 1) * st - 2 * pad + k
        assert x.C == Cin and ops.out_size(H2, k, st, pad) == h and ops.out_size(W2, k, st, pad) == wd, (x.C, Cin, k, pad, st)
        self.out = Act(eng.dev, N, H2, W2, C)
        o = self.out
        self.fdescs, self.fzero = ops.dgrad_descs(N, H2, W2, o.ld, C, h, wd, x.ld, Cin, x.nbytes, k, st, pad)
        self.ddesc = ops.fwd_desc(N, H2, W2, o.ld, C, o.nbytes, k, st, pad, Cin, x.ld)
        self.wdesc = ops.conv_wgrad_desc(N, H2, W2, o.ld, C, o.nbytes, h, wd, x.ld, Cin, x.nbytes, k, st, pad,
                                         eng._ksplit(x.M, Cin, C, k * k))

    def acts(self):
        return (self.out,)

    def grad_ranges(self):
        return [(self.w.off, self.w.off + self.w.n)]

    def fwd(self, train):
        if self.fzero:
            self.out.t.zero_()
        for d in self.fdescs:
            self.eng.fwd_conv(d, self.x.t, self.w, self.out.t, transposed=True)

    def bwd(self):
        x, o = self.x, self.out
        self.eng.wgrad(self.wdesc, x.t, o.g, self.w.dw)
        self.ddesc.accumulate = int(_take(x))
        ops.conv_gemm(self.ddesc, o.g, self.w.wf, x.g)


class PixelShuffle2(Op):
    """nn.PixelShuffle(2) (MyNet.py:37,46,48): [N][h][w][4C] -> [N][2h][2w][C], a permutation (and its inverse for the gradient)."""

    def __init__(self, eng, x, out=None):
        assert x.C % 32 == 0, "PixelShuffle(2): 4 x (a multiple of 8) channels"
        self.x = x
        self.out = out if out is not None else Act(eng.dev, x.N, 2 * x.H, 2 * x.W, x.C // 4)
        self.own_out = out is None
        self.tmp = None

    def acts(self):
        return (self.out,) if self.own_out else ()

    def fwd(self, train):
        x, o = self.x, self.out
        ops.pixel_shuffle2(x.t, x.ld, o.t, o.ld, x.N, x.H, x.W, o.C)

    def bwd(self):
        x, o = self.x, self.out
        if not _take(x):
            ops.pixel_shuffle2(x.g, _ldg(x), o.g, _ldg(o), x.N, x.H, x.W, o.C, inverse=True)
            return
        if self.tmp is None:                               # the input has another consumer: permute into scratch, then add
            self.tmp = torch.empty(x.N, x.H, x.W, x.C, dtype=ops.ACT_DTYPE, device=x.t.device)
        ops.pixel_shuffle2(self.tmp, x.C, o.g, _ldg(o), x.N, x.H, x.W, o.C, inverse=True)
        ops.pw_fwd(self.tmp, x.C, None, x.g, _ldg(x), x.g, _ldg(x), x.M, x.C, None)


class WeightedPool(Op):
    """Weighter's tail (MyNet.py:96-119): flatten -> nn.Linear(HW, 1) -> sum over channels -> sigmoid: one fp32 scale per image
    (`scale`, with its gradient `dscale` written by the consumer)."""

    def __init__(self, eng, x, linear, dscale):
        self.eng, self.x = eng, x
        self.w, self.w_off = eng.store.vec(linear.weight)
        self.b, self.b_off = eng.store.vec(linear.bias)
        assert linear.weight.shape == (1, x.H * x.W), (tuple(linear.weight.shape), x.H, x.W)
        self.pre = torch.empty(x.N, device=eng.dev)
        self.scale = torch.empty(x.N, device=eng.dev)
        self.dscale = dscale

    def grad_ranges(self):
        return [(self.w_off, self.w_off + self.x.H * self.x.W), (self.b_off, self.b_off + 1)]

    def fwd(self, train):
        x = self.x
        ops.weighted_pool_fwd(x.t, x.ld, self.w, self.b, self.pre, self.scale, x.N, x.H * x.W, x.C)

    def bwd(self):
        x, G_ = self.x, self.eng.store.Gcur
        HW = x.H * x.W
        ops.weighted_pool_bwd(self.dscale, self.scale, x.t, x.ld, self.w, x.g, _ldg(x), _take(x), G_[self.w_off:self.w_off + HW],
                              G_[self.b_off:self.b_off + 1], x.N, HW, x.C)


class Combine3(Op):
    """my_decoder's output (MyNet.py:152-155): factor * sum_k map_k * scale_k[n], the module's fp32 N x 1 x H x W result."""

    def __init__(self, eng, maps, factor):
        m = maps[0]
        self.maps, self.factor = maps, float(factor)
        self.ds = torch.zeros(3, m.N, device=eng.dev)              # the three WeightedPool ops read their row of it
        self.scales = None                                           # set once the WeightedPool ops exist
        self.y = torch.empty(m.N, 1, m.H, m.W, device=eng.dev)
        self.outputs, self.douts = (self.y,), [None]

    def fwd(self, train):
        m = self.maps[0]
        ops.combine3_fwd([a.t for a in self.maps], self.scales, self.factor, m.N, m.H * m.W, self.y)

    def bwd(self):
        m = self.maps[0]
        if self.douts[0] is None:
            self.ds.zero_()
            return
        ops.combine3_bwd(self.douts[0], [a.t for a in self.maps], self.scales, self.factor, m.N, m.H * m.W, [a.g for a in self.maps], self.ds)


# ---------------------------------------------------------------------------------------------- the tape
class TapeEngine(EngineCore):
    """Launch plan of one input shape: subclasses fill `self.tape` in `_plan` and name the image op (`self.stem`) and the
    output ops (`self.heads`: ops with `.outputs` / `.douts`)."""

    def __init__(self, module, store, N, H, W):
        super().__init__(module, store, N, H, W)
        self.tape, self.heads, self.stem, self._bufs = [], [], None, []
        self._sums_planned = False
        self._thr = None
        self._plan()
        self._acts = [a for op in self.tape for a in op.acts()] + self._bufs

    _fuse_pw = os.environ.get("MDE_FUSE_PW", "1") != "0"

    def add(self, op):
        self.tape.append(op)
        return op

    def buf(self, N, H, W, C):
        """A zero-initialised activation the plan owns (concatenation targets: producers write channel slices of it)."""
        a = Act(self.dev, N, H, W, C)
        a.t.zero_()
        a.concat_root = True
        self._bufs.append(a)
        return a

    # ---- building blocks shared by the plans
    def pw(self, x, bias=None, r=None, act=None, out=None):
        """out = act(x + bias + r) (Pw).  When x is the dense output of the convolution just added to the tape and nothing else
        has read it, the pass is folded into that convolution's epilogue instead (Conv.fuse): one launch and one tensor
        less per biased / activated conv (MiDaS' ResidualConvUnits and output head, BTS' conv + ELU chains, VNL's FTB
        blocks, DORN's and MyNet's biased convs).  MDE_FUSE_PW=0 keeps the separate pass (A/B, tests)."""
        last = self.tape[-1] if self.tape else None
        ok = (self._fuse_pw and isinstance(last, Conv) and last.out is x and last.own_out and last.site is None and last.groups == 1
              and not last.fused and x.parent is None and (bias is not None or r is not None or act is not None))
        if ok and r is not None:
            o_ld = out.ld if out is not None else x.C
            ok = (r.N, r.H, r.W, r.C) == (x.N, x.H, x.W, x.C) and r.ld == o_ld and r is not x
        if ok and out is not None:
            ok = (out.N, out.H, out.W, out.C) == (x.N, x.H, x.W, x.C)
        if not ok:
            return self.add(Pw(self, x, bias=bias, r=r, act=act, out=out)).out
        b, b_off = (None, None)
        if bias is not None:
            b, b_off = self.store.vec(bias)
            assert b.numel() == x.C, (b.numel(), x.C)
        dst = out if out is not None else Act(self.dev, x.N, x.H, x.W, x.C)
        return last.fuse(b, b_off, r, act, dst).out

    def conv_bn(self, x, conv, bn, relu, out=None):
        """nn.Conv2d (its own stride / padding / dilation / groups) -> nn.BatchNorm2d -> [ReLU]."""
        k, s, p, d, g = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0], conv.groups
        site = self._site([bn])
        c = self.add(Conv(self, x, conv.weight, k, s, p, d, g, site=site))
        return self.add(BN(self, c.out, site, relu, out=out, bias=conv.bias)).out

    def bottleneck(self, x, conv1, bn1, conv2, bn2, conv3, bn3, ds_conv=None, ds_bn=None):
        """1x1 -> BN -> ReLU -> 3x3 (stride / dilation / groups) -> BN -> ReLU -> 1x1 -> BN -> (+ identity | + BN(1x1 shortcut))
        -> ReLU: torchvision's Bottleneck v1.5 and VNL.py:618-669 alike.  The join is one pass over both BN sites."""
        a = self.conv_bn(x, conv1, bn1, True)
        b = self.conv_bn(a, conv2, bn2, True)
        s3 = self._site([bn3])
        c3 = self.add(Conv(self, b, conv3.weight, 1, site=s3)).out
        if ds_conv is not None:
            sd = self._site([ds_bn])
            ds = self.add(Conv(self, x, ds_conv.weight, 1, ds_conv.stride[0], site=sd)).out
            return self.add(BN(self, c3, s3, True, res=ds, res_site=sd)).out
        return self.add(BN(self, c3, s3, True, res=x)).out

    def forward(self, x, train, check_data=False):
        assert x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (self.N, 3, self.H, self.W), tuple(x.shape)
        self.begin_forward(train, check_data)
        self.stem.x = x
        for op in self.tape:
            op.fwd(train)
        if train:
            self.store.nbt += 1
        self.end_forward(train)
        return tuple(t for h in self.heads for t in h.outputs)

    def progress_thresholds(self):
        """thr[i]: once the i-th op of the REVERSED tape has run, every element >= thr[i] of the flat gradient buffer is
        final -- no op that has not run yet writes at or above it (Op.grad_ranges; parameters no op writes, unused or frozen
        ones, are final from the start).  The flat order is the module's parameter order, encoder first, so a backward pass
        completes the buffer from its tail and the thresholds fall towards 0."""
        if self._thr is None:
            rev = list(reversed(self.tape))
            last = {}
            for i, op in enumerate(rev):
                for r in op.grad_ranges():
                    last[r] = i                                    # the last position that writes r
            ends_at = {}
            for (b, e), i in last.items():
                ends_at[i] = max(ends_at.get(i, 0), e)
            thr, pending = [0] * len(rev), 0
            for i in range(len(rev) - 1, -1, -1):                    # pending(i) = ranges whose last writer comes after i
                thr[i] = pending
                pending = max(pending, ends_at.get(i, 0))
            self._thr = thr
        return self._thr

    def backward(self, douts, on_progress=None, marks=None, consumer_waits_side=False):
        """douts: one fp32 gradient (or None) per output tensor, in the order forward returned them.  Adds the parameter
        gradients into store.Gcur.
        on_progress(offset), if given, is called as backward walks the tape whenever every gradient element >= offset of the
        flat buffer has become final (progress_thresholds) -- dp.FlatGradReducer.ready: the bucket's all-reduce goes out on its
        own stream while the rest of backward runs, as engine.FCRNEngine.backward does.  marks: descending offsets the consumer
        cares about (its bucket starts): the callback then fires only when a mark is passed.  consumer_waits_side: the
        consumer orders itself behind the weight-gradient stream (FlatGradReducer(extra_streams=[eng.side])); otherwise that
        stream is joined into the current one before every call."""
        self.store.det_begin()
        self.begin_backward()
        if self.store.deterministic and on_progress is not None:
            final_cb, on_progress = on_progress, None            # the gradients reach the buffer only with the final flush
        else:
            final_cb = on_progress
        thr = self.progress_thresholds() if on_progress is not None else None
        marks = sorted(set(marks), reverse=True) if marks is not None else None
        mi, reported = 0, None
        for a in self._acts:
            a.gw = False
        for a in self._bufs:           # concatenation targets: consumers may cover only part of the channels, so every
            a.g.zero_()                # writer accumulates onto a zeroed gradient
            a.gw = True
        i = 0
        for h in self.heads:
            for k in range(len(h.outputs)):
                d = douts[i]
                if d is not None and d.dim() > 0 and d.numel() > 1 and all(st_ == 0 for st_ in d.stride()):
                    d = None               # a zero-stride placeholder of the criterion-fused route (SoftmaxHead.stash has the gradient)
                h.douts[k] = d.contiguous() if d is not None else None
                i += 1
        global _TRACE, _CUR_OP
        tracing = FUSE_BN_RED and not self._sums_planned
        if tracing:
            _TRACE = {}
        try:
            for i, op in enumerate(reversed(self.tape)):
                _CUR_OP = op
                op.bwd()
                if thr is not None and thr[i] != reported and thr[i] > 0:
                    if marks is not None:
                        if mi >= len(marks) or thr[i] > marks[mi]:
                            continue
                        while mi < len(marks) and marks[mi] >= thr[i]:
                            mi += 1
                    if not consumer_waits_side:
                        self.join_side()
                    on_progress(thr[i])
                    reported = thr[i]
        finally:
            trace, _TRACE, _CUR_OP = _TRACE, None, None
        if tracing:
            self._plan_fused_sums(trace)
        self.join_side()
        self.store.det_end()
        self.end_backward()
        if final_cb is not None:
            final_cb(0)

    def _plan_fused_sums(self, last_taker):
        """After the first backward: every BatchNorm op whose output gradient is completed by a convolution's input-gradient
        launch hands that launch its backward sums (Conv.red) and drops its own reduction pass."""
        self._sums_planned = True
        self.fused_sums = 0
        self._fused_pairs = []
        for op in self.tape:
            if not isinstance(op, (BN, PrefixBN)):
                continue
            conv = (last_taker.get(id(op.out)) or [None])[-1]
            if (isinstance(conv, Conv) and conv.x is op.out and conv.need_dgrad and conv.red is None and len(conv._chunks()) == 1
                    and op.out.parent is None):
                spec = op.red_spec()
                if spec is not None:
                    conv.red, op.reduced = spec, True
                    self._fused_pairs.append((conv, op, spec))
                    self.fused_sums += 1

        # identity shortcuts (out = relu(bn(c) + x)): where x's gradient has exactly two writers -- this BatchNorm's backward pass
        # (the masked d(out), first) and ONE convolution that already carries x's producer's sums -- that convolution's launch
        # takes the masked d(out) from out.g itself (mde_bn_red.add) and the pass stops writing the copy
        self._dres_pairs = []
        if FUSE_DRES:
            by_conv = {id(conv): i for i, (conv, _, _) in enumerate(self._fused_pairs)}
            for op in self.tape:
                if not (isinstance(op, BN) and op.res is not None and op.res_site is None and op.bits is not None):
                    continue
                x = op.res
                takers = last_taker.get(id(x.root()), [])
                if x.parent is not None or len(takers) != 1 or id(takers[0]) not in by_conv:
                    continue
                conv, prod, spec = self._fused_pairs[by_conv[id(takers[0])]]
                if (conv.x is x and len(conv.ddescs) == 1 and not conv.dzero and (spec.relu_bits or spec.x2) and op.out.parent is None
                        and op.out.ld == x.ld == x.C and (op.out.N, op.out.H, op.out.W, op.out.C) == (x.N, x.H, x.W, x.C)):
                    spec2 = ops.bn_red_with_add(spec, op.out.g, op.bits)
                    self._fused_pairs[by_conv[id(conv)]] = (conv, prod, spec2)
                    self._dres_pairs.append((op, conv, spec, spec2))
                    conv.red, conv.first_writer, op.skip_dres = spec2, True, True

    def set_fused_sums(self, on):
        """Diagnostics / tests: switch the planned fusions off (every BatchNorm runs its own reduction pass again) and back on."""
        for conv, op, spec in self._fused_pairs:
            conv.red, op.reduced = (spec, True) if on else (None, False)
        for op, conv, _, _ in self._dres_pairs:
            conv.first_writer = op.skip_dres = on

    def grad_boundaries(self):
        """Flat-gradient offsets at which conv weights start: where dp.FlatGradReducer may cut its buckets."""
        return self.store.layer_boundaries()


class _TapeFunction(torch.autograd.Function):
    """One autograd node for a whole network (see network/FCRN.py:_FCRNFunction for why backward RETURNS gradient views)."""

    @staticmethod
    def forward(ctx, x, engine, train, *params):
        ctx.engine, ctx.train = engine, train
        ctx.set_materialize_grads(False)          # an unused output's gradient stays None (no zero tensors of GBs)
        for h in engine.heads:               # heads with large outputs write into fresh tensors instead of being cloned
            if hasattr(h, "new_outputs"):
                h.new_outputs()
        engine.forward(x, train, check_data=True)
        engine.forward_serial = ctx.serial = getattr(engine, "forward_serial", 0) + 1
        return tuple(y if getattr(h, "fresh", False) else y.clone() for h in engine.heads for y in h.outputs)

    @staticmethod
    def backward(ctx, *douts):
        _lib.note_fp16_backward()
        eng = ctx.engine
        if not ctx.train:
            raise RuntimeError("mono_depth_estimation_amd: backward() through a forward pass run in eval() mode is not supported "
                               "(no BatchNorm-backward kernel for running statistics); call .train() first or use torch.no_grad().")
        if ctx.needs_input_grad[0]:
            raise RuntimeError("mono_depth_estimation_amd: the gradient with respect to the input image is not computed; detach it.")
        if ctx.serial != eng.forward_serial:
            raise RuntimeError("mono_depth_estimation_amd: backward() of a forward pass whose activations were overwritten by a later "
                               "forward of the same input shape (forward #%d, latest #%d)." % (ctx.serial, eng.forward_serial))
        st = eng.store
        buf = st.begin_autograd_backward()
        red = st.grad_reducer
        try:
            if red is not None:
                # the gradient exchange overlapped with this backward (TapeModule.set_grad_reducer): buckets go out on the
                # reducer's stream as the tape passes their first element; the caller joins with reducer.finish()
                red.begin(buf)
                eng.backward(douts, on_progress=red.ready, marks=[b for b, _ in red.buckets], consumer_waits_side=eng.side in red.extra_streams)
            else:
                eng.backward(douts)
        finally:
            st.Gcur = st.G
        grads = tuple(st.grad_view(p, buf) if need else None for p, need in zip(eng.params, ctx.needs_input_grad[3:]))
        if st._g_base is None:
            grads = tuple(g.clone() if g is not None else None for g in grads)
        return (None, None, None) + grads


class TapeModule(torch.nn.Module):
    """nn.Module surface of a tape-run network: parameters live in the reference's submodule tree (same state_dict keys),
    `forward` runs the HIP plan.  Subclasses set `_engine_cls` and implement `_make_store(device)`."""

    _engine_cls = None

    def _init_runtime(self):
        self._engines, self._store = {}, None

    def _make_store(self, device):
        raise NotImplementedError

    def _first_param(self):
        return next(self.parameters())

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, "_engines", None) is not None:
            dev = self._first_param().device
            self._engines, self._store = {}, None
            if dev.type == "cuda":
                self._store = self._make_store(dev)
        return out

    def __deepcopy__(self, memo):
        import copy
        with torch.no_grad():
            for p in self.parameters():
                memo[id(p)] = torch.nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad)
            for b in self.buffers():
                memo[id(b)] = b.detach().clone()
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in ("_engines", "_store"):
                new.__dict__[k] = copy.deepcopy(v, memo)
        new._engines, new._store = {}, None
        return new

    def __getstate__(self):
        import copy
        if self._store is None:
            state = dict(self.__dict__)
            state["_engines"] = {}
            return state
        return copy.deepcopy(self).__dict__

    def set_grad_reducer(self, reducer):
        """Overlap the data-parallel gradient exchange with backward on the nn.Module path (what Lightning's DDP cannot do
        for a network that is ONE autograd node: its bucket hooks all fire when the node returns).  `reducer`: a
        dp.FlatGradReducer over this module's flat gradient buffer (`module._store.G`, boundaries `_store.layer_boundaries()`);
        `loss.backward()` then issues each bucket's all-reduce as soon as the tape has passed the bucket's first element, and
        the caller joins with `reducer.finish()` before the optimiser step.  None switches it off."""
        if self._store is None:
            raise RuntimeError("set_grad_reducer: move the module to its GPU first (the flat gradient buffer lives there)")
        self._store.grad_reducer = reducer

    def _engine(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected an N x 3 x H x W image batch, got %s" % (tuple(x.shape),))
        if not x.is_cuda:
            raise RuntimeError("mono_depth_estimation_amd networks run on MI355X only (input is on %s); there is no CPU fallback" % x.device)
        st = self._store
        if st is None or st.dev != x.device or not st.storage_is_current():
            self._store = st = self._make_store(x.device)
            self._engines.clear()
        key = tuple(x.shape)
        eng = self._engines.pop(key, None)
        if eng is None:
            cap = max(1, int(os.environ.get("MDE_MAX_PLANS", "4")))
            while len(self._engines) >= cap:
                self._engines.pop(next(iter(self._engines)))
            eng = self._engine_cls(self, st, x.shape[0], x.shape[2], x.shape[3])
        self._engines[key] = eng
        return eng

    def _run(self, x):
        eng = self._engine(x)
        outs = _TapeFunction.apply(x.contiguous().float(), eng, self.training, *eng.params)
        i = 0
        for h in eng.heads:
            n = len(h.outputs)
            if hasattr(h, "tag") and getattr(h, "fresh", False):
                h.tag(*outs[i:i + n])
            i += n
        return outs
