"""Drop-in for the reference's criteria.py depth losses (same class names, constructors and call
signatures; 0-dim result with grad; the reductions and their gradients are wavefront-reduced HIP
kernels in csrc/losses.hip; no CPU fallback):

  silog_loss(variance_focus)(depth_est, depth_gt)      criteria.py:724-732   (the FCRN bench loss)
  MaskedL1Loss()(pred, target), MaskedMSELoss(), berHuLoss()   criteria.py:67-90,113-133
  MaskedDepthLoss()(pred, target)                      criteria.py:17-64    (Eigen's loss)
  compute_scale_and_shift(prediction, target, mask=None)       criteria.py:154-176
  GradientLoss(scales, reduction)(prediction, target, mask)    criteria.py:227-244,283-303
  MidasLoss(alpha, scales, loss, reduction)(prediction, target)   criteria.py:306-332
      loss in {'mse','l1','trim','ssimse','ssil1','ssitrim'}; as in the reference, 'trim' computes the plain
      L1 term (its sort-and-slice trims nothing) and only the batch-based reduction is well-formed for the
      data terms; every mask is target > 0.
  TrimmedProcrustesLoss(alpha, scales, reduction)(prediction, target)   criteria.py:335-363 (+ :135-152)
  WCEL_Loss(args)(pred_logit, gt_bins, gt)             criteria.py:839-863
  VNL_Loss(focal_x, focal_y, input_size, ...)(gt_depth, pred_depth, select=True)   criteria.py:866-1045
      the triples are drawn on the host from the GLOBAL numpy RNG with the reference's exact sequence of
      calls (select_index), so a seeded run samples the same pixels as the reference.
  ModelLoss(args)(pred_depth, pred_logit, depth_bins, depth_gt)   criteria.py:1047-1062
  ordLoss()(ord_labels, target)                        criteria.py:734-787   (DORN's loss; csrc/ordinal.hip)
      ord_labels: the ordinal probabilities N x K x H x W; target: the SID label map N x 1 x H x W as modules/dorn.py:102-107
      computes it (a float tensor, not truncated: plane k counts as "<= target" by a float comparison, as in the reference).
  bins_to_depth(depth_bin, depth_bin_border), depth_to_bins(depth, depth_min, depth_max, dec_out_c)
      modules/vnl.py:202-230 (methods of the reference's VNLModule; free functions here, kernels in
      csrc/vnl_losses.hip)
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops


class _SilogFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, gt, variance_focus):
        ctx.in_dtype = est.dtype
        est = est.contiguous().float()
        gt = gt.contiguous().float()
        ws = ops.silog_ws(est.device)
        loss = torch.empty(1, device=est.device)
        ops.silog_fwd(est, gt, variance_focus, ws, loss)
        ctx.save_for_backward(est, gt, ws)
        ctx.variance_focus = variance_focus
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        est, gt, ws = ctx.saved_tensors
        grad = torch.empty_like(est)
        ops.silog_bwd(est, gt, ctx.variance_focus, ws, gout.contiguous().float().reshape(1), grad)
        return grad.to(ctx.in_dtype), None, None


class silog_loss(nn.Module):
    def __init__(self, variance_focus):
        super(silog_loss, self).__init__()
        self.variance_focus = variance_focus

    def forward(self, depth_est, depth_gt):
        if not depth_est.is_cuda:
            raise RuntimeError("mono_depth_estimation_amd.criteria.silog_loss runs on MI355X only; no CPU fallback")
        return _SilogFunction.apply(depth_est, depth_gt, float(self.variance_focus))


def _need_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError("mono_depth_estimation_amd.criteria.%s runs on MI355X only; no CPU fallback" % name)


class _MaskedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind):
        ctx.in_dtype = pred.dtype
        pred, target = pred.contiguous().float(), target.contiguous().float()
        ws = ops.masked_loss_ws(pred.device)
        loss = torch.empty(1, device=pred.device)
        ops.masked_loss_fwd(kind, pred, target, ws, loss)
        ctx.save_for_backward(pred, target, ws)
        ctx.kind = kind
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        pred, target, ws = ctx.saved_tensors
        grad = torch.empty_like(pred)
        ops.masked_loss_bwd(ctx.kind, pred, target, ws, gout.contiguous().float().reshape(1), grad)
        return grad.to(ctx.in_dtype), None, None


class _MaskedLoss(nn.Module):
    kind = None

    def forward(self, pred, target):
        assert pred.dim() == target.dim(), "inconsistent dimensions"
        _need_gpu(pred, type(self).__name__)
        self.loss = _MaskedFunction.apply(pred, target, self.kind)
        return self.loss


class MaskedMSELoss(_MaskedLoss):
    """criteria.py:67-77: mean (target - pred)^2 over target > 0."""
    kind = "mse"


class MaskedL1Loss(_MaskedLoss):
    """criteria.py:80-90: mean |target - pred| over target > 0."""
    kind = "l1"


class berHuLoss(_MaskedLoss):
    """criteria.py:113-133: reverse Huber with c = 0.2 * max(pred - target) (over all pixels, as written there)."""
    kind = "berhu"


class _MaskedDepthFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        ctx.in_dtype = pred.dtype
        pred, target = pred.contiguous().float(), target.contiguous().float()
        N, H, W = pred.shape[0], pred.shape[-2], pred.shape[-1]
        ws = ops.masked_depth_ws(N, pred.device)
        loss = torch.empty(1, device=pred.device)
        ops.masked_depth_fwd(pred, target, N, H, W, ws, loss)
        ctx.save_for_backward(pred, target, ws)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        pred, target, ws = ctx.saved_tensors
        grad = torch.empty_like(pred)
        ops.masked_depth_bwd(pred, target, pred.shape[0], pred.shape[-2], pred.shape[-1], ws,
                             gout.contiguous().float().reshape(1), grad)
        return grad.to(ctx.in_dtype), None


class MaskedDepthLoss(nn.Module):
    """criteria.py:17-64 (used by modules/eigen.py:8-9): linear-space scale-invariant term + masked
    forward-difference gradient cost.  pred/target: [N][H][W] or [N][1][H][W]."""

    def forward(self, pred, target):
        assert pred.dim() == target.dim(), "inconsistent dimensions"
        _need_gpu(pred, "MaskedDepthLoss")
        if pred.dim() == 4 and pred.shape[1] != 1:
            raise NotImplementedError("MaskedDepthLoss: single-channel depth maps only (as the reference's slicing assumes)")
        self.loss = _MaskedDepthFunction.apply(pred, target)
        return self.loss


def _squeeze_pair(prediction, target):
    if prediction.dim() == 4:
        prediction = prediction.squeeze(1)
    if target.dim() == 4:
        target = target.squeeze(1)
    if prediction.dim() != 3 or prediction.shape != target.shape:
        raise ValueError("expected [N][H][W] (or [N][1][H][W]) depth maps of equal shape, got %s and %s"
                         % (tuple(prediction.shape), tuple(target.shape)))
    return prediction.contiguous().float(), target.contiguous().float()


def _check_mask(mask, target, who):
    """The HIP kernels derive the mask as target > 0 (what MidasLoss does); an explicit mask must be that one."""
    if mask is not None and not torch.equal(mask.reshape(target.shape) > 0, target > 0):
        raise NotImplementedError("%s: only the mask (target > 0) is supported on the HIP path" % who)


def compute_scale_and_shift(prediction, target, mask=None):
    """criteria.py:154-176: per-image closed-form least squares (scale, shift); zeros where det == 0."""
    _need_gpu(prediction, "compute_scale_and_shift")
    p, t = _squeeze_pair(prediction, target)
    _check_mask(mask, t, "compute_scale_and_shift")
    N, H, W = p.shape
    scale, shift = torch.empty(N, device=p.device), torch.empty(N, device=p.device)
    ops.scale_and_shift(p, t, N, H, W, ops.midas_ws(N, p.device), scale, shift)
    return scale, shift


class _MidasFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, target, ssi, data_kind, data_weight, alpha, scales, batch_based):
        ctx.in_dtype = prediction.dtype
        p, t = _squeeze_pair(prediction, target)
        N, H, W = p.shape
        ws = ops.midas_ws(N, p.device)
        loss = torch.empty(1, device=p.device)
        ops.midas_fwd(p, t, N, H, W, ssi, data_kind, data_weight, alpha, scales, batch_based, ws, loss)
        ctx.save_for_backward(p, t, ws)
        ctx.cfg = (ssi, data_kind, scales, prediction.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        p, t, ws = ctx.saved_tensors
        ssi, data_kind, scales, shape = ctx.cfg
        grad = torch.empty_like(p)
        ops.midas_bwd(p, t, p.shape[0], p.shape[1], p.shape[2], ssi, data_kind, scales, ws,
                      gout.contiguous().float().reshape(1), grad)
        return grad.reshape(shape).to(ctx.in_dtype), None, None, None, None, None, None, None


class GradientLoss(nn.Module):
    """criteria.py:283-303: sum over `scales` sub-samplings [::2^k] of the masked absolute forward differences
    of mask*(prediction - target), reduced batch-based or image-based."""

    def __init__(self, scales=4, reduction='batch-based'):
        super().__init__()
        if not 0 <= scales <= 4:
            raise NotImplementedError("GradientLoss: up to 4 scales on the HIP path")
        self.scales, self.batch_based = scales, reduction == 'batch-based'

    def forward(self, prediction, target, mask):
        _need_gpu(prediction, "GradientLoss")
        _check_mask(mask, target.squeeze(1) if target.dim() == 4 else target, "GradientLoss")
        return _MidasFunction.apply(prediction, target, False, 0, 0.0, 1.0, self.scales, self.batch_based)


class MidasLoss(nn.Module):
    """criteria.py:306-332 (modules/midas.py:29-37 uses it batch-based)."""

    def __init__(self, alpha=0.5, scales=4, loss='ssimse', reduction='batch-based'):
        super().__init__()
        self.loss = loss
        if 'trim' in loss or 'l1' in loss:
            self.data_kind = 1            # TrimmedMAELoss never trims (criteria.py:214-216): identical to L1Loss
        elif 'mse' in loss:
            self.data_kind = 0
        else:
            raise ValueError()
        if reduction != 'batch-based':
            raise NotImplementedError("MidasLoss: the reference's data terms are only well-formed with the batch-based "
                                      "reduction (image-based indexes a per-pixel map with per-image counts)")
        if not 0 <= scales <= 4:
            raise NotImplementedError("MidasLoss: up to 4 scales on the HIP path")
        self.alpha, self.scales = float(alpha), scales

    def forward(self, prediction, target):
        _need_gpu(prediction, "MidasLoss")
        return _MidasFunction.apply(prediction, target, "ssi" in self.loss, self.data_kind, 1.0,
                                    self.alpha if self.alpha > 0 else 0.0, self.scales, True)


class _ProcrustesFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, target, alpha, scales, batch_based):
        ctx.in_dtype = prediction.dtype
        p, t = _squeeze_pair(prediction, target)
        N, H, W = p.shape
        ws = ops.procrustes_ws(N, p.device)
        pn, tn, loss = torch.empty_like(p), torch.empty_like(p), torch.empty(1, device=p.device)
        ops.procrustes_fwd(p, t, N, H, W, alpha, scales, batch_based, ws, pn, tn, loss)
        ctx.save_for_backward(p, t, ws, pn, tn)
        ctx.cfg = (scales, prediction.shape)
        return loss.reshape(()), pn

    @staticmethod
    def backward(ctx, gout, _gpn):
        p, t, ws, pn, tn = ctx.saved_tensors
        scales, shape = ctx.cfg
        gtmp, grad = torch.empty_like(p), torch.empty_like(p)
        ops.procrustes_bwd(p, t, p.shape[0], p.shape[1], p.shape[2], scales, ws, pn, tn,
                           gout.contiguous().float().reshape(1), gtmp, grad)
        return grad.reshape(shape).to(ctx.in_dtype), None, None, None, None


class TrimmedProcrustesLoss(nn.Module):
    """criteria.py:335-363.  `prediction_ssi` holds the robustly normalised prediction of the last call
    (detached), as the reference's property does."""

    def __init__(self, alpha=0.5, scales=4, reduction="batch-based"):
        super().__init__()
        if reduction != "batch-based":
            raise NotImplementedError("TrimmedProcrustesLoss: the reference's data term is only well-formed batch-based")
        if not 0 <= scales <= 4:
            raise NotImplementedError("TrimmedProcrustesLoss: up to 4 scales on the HIP path")
        self.alpha, self.scales = float(alpha), scales
        self.prediction_ssi = None

    def forward(self, prediction, target):
        _need_gpu(prediction, "TrimmedProcrustesLoss")
        loss, pn = _ProcrustesFunction.apply(prediction, target, self.alpha if self.alpha > 0 else 0.0, self.scales, True)
        self.prediction_ssi = pn.detach()
        return loss


# ------------------------------------------------------------------------------ DORN (csrc/ordinal.hip)
class _OrdLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, target):
        ctx.in_dtype = prob.dtype
        x = prob.contiguous().float()
        N, K = x.shape[0], x.shape[1]
        HW = x.numel() // (N * K)
        if target.numel() != N * HW:
            raise ValueError("ordLoss: ordinal probabilities %s need %d targets, got %s" % (tuple(prob.shape), N * HW, tuple(target.shape)))
        t = target.to(device=x.device, dtype=torch.float32).contiguous()
        ws = ops.ord_loss_ws(x.device)
        loss = torch.empty(1, device=x.device)
        ops.ord_loss_fwd(x, t, N, K, HW, ws, loss)
        ctx.save_for_backward(x, t)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        x, t = ctx.saved_tensors
        N, K = x.shape[0], x.shape[1]
        grad = torch.empty_like(x)
        ops.ord_loss_bwd(x, t, N, K, x.numel() // (N * K), gout.contiguous().float().reshape(1), grad)
        return grad.to(ctx.in_dtype), None


class ordLoss(nn.Module):
    """criteria.py:734-787: the average over pixels of the ordinal log-likelihood."""

    def __init__(self):
        super(ordLoss, self).__init__()
        self.loss = 0.0

    def forward(self, ord_labels, target):
        _need_gpu(ord_labels, "ordLoss")
        if ord_labels.dim() != 4:
            raise ValueError("ordLoss: ord_labels must be N x K x H x W, got %s" % (tuple(ord_labels.shape),))
        self.loss = _OrdLossFunction.apply(ord_labels, target)
        return self.loss


# ------------------------------------------------------------------------------ VNL configuration (csrc/vnl_losses.hip)
class _WcelFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, bins, gt, weight):
        ctx.in_dtype = logit.dtype
        x = logit.contiguous().float()
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        if bins.numel() != N * HW or gt.numel() != N * HW:
            raise ValueError("WCEL_Loss: logits %s need %d labels and depths, got %d and %d"
                             % (tuple(logit.shape), N * HW, bins.numel(), gt.numel()))
        if tuple(weight.shape) != (C, C):
            raise ValueError("WCEL_Loss: weight %s for %d bins" % (tuple(weight.shape), C))
        b = bins.to(device=x.device, dtype=torch.int32).contiguous()
        g = gt.to(device=x.device, dtype=torch.float32).contiguous()
        ws = ops.wcel_ws(C, x.device)
        lse = torch.empty(N * HW, device=x.device)
        loss = torch.empty(1, device=x.device)
        ops.wcel_fwd(x, b, g, weight, N, C, HW, ws, lse, loss)
        ctx.save_for_backward(x, b, weight, ws, lse)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        x, b, weight, ws, lse = ctx.saved_tensors
        N, C = x.shape[0], x.shape[1]
        grad = torch.empty_like(x)
        ops.wcel_bwd(x, b, weight, N, C, x.numel() // (N * C), ws, lse, gout.contiguous().float().reshape(1), grad)
        return grad.to(ctx.in_dtype), None, None, None


class WCEL_Loss(nn.Module):
    """criteria.py:839-863.  `args` carries dec_out_c and wce_loss_weight (a [C][C] list or array, row-normalised
    here exactly as the reference's constructor does, in float64)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        w = np.asarray(self.args.wce_loss_weight, dtype=np.float64)
        w = w / np.sum(w, 1, keepdims=True)
        self.weight = torch.from_numpy(w)

    def forward(self, pred_logit, gt_bins, gt):
        _need_gpu(pred_logit, "WCEL_Loss")
        if pred_logit.shape[1] != self.args.dec_out_c:
            raise ValueError("WCEL_Loss: %d logit channels, dec_out_c = %d" % (pred_logit.shape[1], self.args.dec_out_c))
        self.weight = self.weight.to(device=pred_logit.device, dtype=torch.float).contiguous()
        head = _head_of(pred_logit, pred_logit.shape[1])
        if head is not None:                  # this forward's logits, straight from the HIP head: the private route
            return _FusedWcelFunction.apply(pred_logit, gt_bins, gt, self.weight, head)
        return _WcelFunction.apply(pred_logit, gt_bins, gt, self.weight)


class _VnlFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, pred, p123, fx, fy, select):
        ctx.in_dtype = pred.dtype
        B, _, H, W = pred.shape
        p = pred.contiguous().float()
        g = gt.to(device=p.device, dtype=torch.float32).contiguous()
        n = p123.shape[1]
        ws = ops.vnl_ws(B, n, p.device)
        loss = torch.empty(1, device=p.device)
        ops.vnl_fwd(g, p, p123, B, H, W, n, fx, fy, select, ws, loss)
        ctx.save_for_backward(g, p, p123, ws)
        ctx.cfg = (fx, fy)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        g, p, p123, ws = ctx.saved_tensors
        B, _, H, W = p.shape
        grad = torch.empty_like(p)
        ops.vnl_bwd(g, p, p123, B, H, W, p123.shape[1], ctx.cfg[0], ctx.cfg[1], ws,
                    gout.contiguous().float().reshape(1), grad)
        return None, grad.to(ctx.in_dtype), None, None, None, None


class VNL_Loss(nn.Module):
    """criteria.py:866-1045.  As in the reference, the filter thresholds that actually apply are the ones
    select_points_groups hard-codes (0.867 / 0.005; the delta_cos / delta_diff_* constructor arguments are stored
    and never used) plus delta_z; unlike the reference, tensors follow the prediction's device instead of being
    pinned to cuda:0, and the ground truth receives no gradient."""

    def __init__(self, focal_x, focal_y, input_size, delta_cos=0.867, delta_diff_x=0.01, delta_diff_y=0.01,
                 delta_diff_z=0.01, delta_z=0.0001, sample_ratio=0.15, device_sampling=False, generator=None):
        """device_sampling (not in the reference; off by default): draw the point triples ON THE DEVICE — three i.i.d.
        uniform index vectors, the distribution of the reference's `choice` + `shuffle` — from `generator` (a
        torch.Generator on the prediction's device, or the device's default stream).  It removes the per-step host draw
        (3 x np.random.choice + shuffle over H*W: ~2.5 ms at 480 x 640) and its host-to-device copy from the training
        loop, at the price of a different random stream than the reference's numpy one."""
        super().__init__()
        self.device_sampling, self.generator = bool(device_sampling), generator
        if delta_z != 0.0001:
            raise NotImplementedError("VNL_Loss: the HIP path fixes delta_z = 1e-4 (the reference's only call site)")
        self.fx, self.fy = float(focal_x), float(focal_y)
        self.input_size = (int(input_size[0]), int(input_size[1]))
        self.delta_cos, self.delta_z, self.sample_ratio = delta_cos, delta_z, sample_ratio
        self.delta_diff_x, self.delta_diff_y, self.delta_diff_z = delta_diff_x, delta_diff_y, delta_diff_z

    def select_index(self):
        """The reference's draw, call for call (criteria.py:912-932): three `choice(num, int(num * ratio))` with
        replacement, each followed by a `shuffle`, all on the global numpy RNG."""
        H, W = self.input_size
        num = W * H
        out = {}
        for i in (1, 2, 3):
            p = np.random.choice(num, int(num * self.sample_ratio), replace=True)
            np.random.shuffle(p)
            out["p%d_x" % i] = p % W
            out["p%d_y" % i] = (p / W).astype(int)
        return out

    def forward(self, gt_depth, pred_depth, select=True):
        _need_gpu(pred_depth, "VNL_Loss")
        if pred_depth.ndim != 4 or pred_depth.shape[1] != 1 or tuple(pred_depth.shape[2:]) != self.input_size:
            raise ValueError("VNL_Loss: prediction %s, expected [B, 1, %d, %d]" % ((tuple(pred_depth.shape),) + self.input_size))
        if gt_depth.shape != pred_depth.shape:
            raise ValueError("VNL_Loss: ground truth %s vs prediction %s" % (tuple(gt_depth.shape), tuple(pred_depth.shape)))
        H, W = self.input_size
        n = int(H * W * self.sample_ratio)
        if n == 0:
            raise ValueError("VNL_Loss: input_size %s samples no triples" % (self.input_size,))
        if self.device_sampling:
            p123 = torch.randint(0, H * W, (3, n), dtype=torch.int32, device=pred_depth.device, generator=self.generator)
        else:
            s = self.select_index()
            lin = np.stack([s["p%d_y" % i] * W + s["p%d_x" % i] for i in (1, 2, 3)]).astype(np.int32)
            p123 = torch.from_numpy(lin).to(pred_depth.device, non_blocking=True)
        return _VnlFunction.apply(gt_depth, pred_depth, p123, self.fx, self.fy, bool(select))


class ModelLoss(nn.Module):
    """criteria.py:1047-1062: WCEL + diff_loss_weight * VNL."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.weight_cross_entropy_loss = WCEL_Loss(args)
        # (args.vnl_device_sampling: opt-in, not a reference field — see VNL_Loss)
        self.virtual_normal_loss = VNL_Loss(focal_x=args.focal_x, focal_y=args.focal_y, input_size=args.crop_size,
                                            device_sampling=bool(getattr(args, "vnl_device_sampling", False)))

    def forward(self, pred_depth, pred_logit, depth_bins, depth_gt):
        loss_metric = self.weight_cross_entropy_loss(pred_logit, depth_bins, depth_gt)
        loss_normal = self.virtual_normal_loss(depth_gt, pred_depth)
        return loss_metric + self.args.diff_loss_weight * loss_normal


# ---------------------------------------------------------------------------------------------- the private route to VNL's head
_FUSE_HEAD = os.environ.get("MDE_FUSE_VNL_HEAD", "1") != "0"


def _head_of(t, C):
    """The graph.SoftmaxHead whose CURRENT forward produced `t` (graph.SoftmaxHead.tag), or None: the tensor is not one the
    module handed out, belongs to an earlier forward, or the route does not apply (deterministic mode, > 192 channels)."""
    ref = getattr(t, "_mde_head", None) if _FUSE_HEAD else None
    if ref is None:
        return None
    h = ref[0]()
    if h is None or ref[1] != h.serial or h.C != C or C > 192 or h.eng.store.deterministic or not h.fresh:
        return None
    return h


def _placeholder(t):
    """A zero gradient of t's shape that owns no memory (stride 0): what the fused route returns for d(logit) / d(softmax); the
    real gradient travels in SoftmaxHead.stash.  (Should autograd add it to another consumer's gradient, the sum is that
    consumer's gradient alone -- and the head adds the general pass for it.)"""
    return torch.zeros((), dtype=t.dtype, device=t.device).expand(t.shape)


class _FusedDepthFunction(torch.autograd.Function):
    """bins_to_depth on the head's own softmax, read from the head's 16-bit input (mde_vnl_head_depth_fwd)."""

    @staticmethod
    def forward(ctx, prob, border, head):
        x = head.x
        P = x.M
        depth = torch.empty(x.N, 1, x.H, x.W, device=prob.device)
        l10, lse = torch.empty(P, device=prob.device), torch.empty(P, device=prob.device)
        ops.vnl_head_depth_fwd(x.t, x.ld, head.bias, border, P, head.C, depth, l10, lse)
        head.stash = dict(head.stash or {}, serial=head.serial, depth=depth, l10=l10, lse=lse, border=border)
        ctx.head, ctx.serial, ctx.like = head, head.serial, prob
        return depth

    @staticmethod
    def backward(ctx, gdepth):
        h = ctx.head
        if h.stash is None or h.stash.get("serial") != ctx.serial:
            raise RuntimeError("mono_depth_estimation_amd: bins_to_depth's backward after the network ran another forward pass")
        h.stash["gdepth"] = gdepth.contiguous().float()
        return _placeholder(ctx.like), None, None


class _FusedWcelFunction(torch.autograd.Function):
    """WCEL_Loss on the head's own logits, read from the head's 16-bit input (mde_vnl_head_wcel_fwd)."""

    @staticmethod
    def forward(ctx, logit, bins, gt, weight, head):
        x = head.x
        P, C = x.M, head.C
        if bins.numel() != P or gt.numel() != P:
            raise ValueError("WCEL_Loss: logits %s need %d labels and depths, got %d and %d" % (tuple(logit.shape), P, bins.numel(), gt.numel()))
        if tuple(weight.shape) != (C, C):
            raise ValueError("WCEL_Loss: weight %s for %d bins" % (tuple(weight.shape), C))
        b = bins.to(device=logit.device, dtype=torch.int32).contiguous()
        g = gt.to(device=logit.device, dtype=torch.float32).contiguous()
        st = head.stash if (head.stash is not None and head.stash.get("serial") == head.serial) else None
        if st is None or "lse" not in st:                     # (bins_to_depth was not called on this forward's softmax: the lse alone)
            lse, scratch = torch.empty(P, device=logit.device), torch.empty(2, P, device=logit.device)
            ops.vnl_head_depth_fwd(x.t, x.ld, head.bias, torch.zeros(C, device=logit.device), P, C, scratch[0], scratch[1], lse)
            st = dict(serial=head.serial, lse=lse)
        ws = ops.wcel_ws(C, logit.device)
        loss = torch.empty(1, device=logit.device)
        ops.vnl_head_wcel_fwd(x.t, x.ld, head.bias, b, g, weight, st["lse"], P, C, ws, loss)
        st["wcel"] = (b, weight, ws)
        head.stash = st
        ctx.head, ctx.serial, ctx.like = head, head.serial, logit
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        h = ctx.head
        if h.stash is None or h.stash.get("serial") != ctx.serial:
            raise RuntimeError("mono_depth_estimation_amd: WCEL_Loss's backward after the network ran another forward pass")
        h.stash["gscale"] = gout.contiguous().float().reshape(1)
        return _placeholder(ctx.like), None, None, None, None


class _BinsToDepthFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, border):
        ctx.in_dtype = prob.dtype
        x = prob.contiguous().float()
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        depth = torch.empty((N, 1) + tuple(x.shape[2:]), device=x.device)
        ops.bins_to_depth_fwd(x, border, N, C, HW, depth)
        ctx.save_for_backward(depth, border)
        ctx.shape = tuple(x.shape)
        return depth

    @staticmethod
    def backward(ctx, gdepth):
        depth, border = ctx.saved_tensors
        N, C = ctx.shape[0], ctx.shape[1]
        gprob = torch.empty(ctx.shape, device=depth.device)
        ops.bins_to_depth_bwd(depth, gdepth.contiguous().float(), border, N, C, depth.numel() // N, gprob)
        return gprob.to(ctx.in_dtype), None


def bins_to_depth(depth_bin, depth_bin_border):
    """modules/vnl.py:219-230: [b, c, h, w] bin probabilities -> [b, 1, h, w] depth = 10 ** sum_c p_c * border_c
    (fp32 result).  depth_bin_border: the C log10 bin centres (array or tensor)."""
    _need_gpu(depth_bin, "bins_to_depth")
    border = torch.as_tensor(np.asarray(depth_bin_border.cpu() if torch.is_tensor(depth_bin_border) else depth_bin_border),
                             dtype=torch.float32).to(depth_bin.device).contiguous()
    if depth_bin.ndim != 4 or border.numel() != depth_bin.shape[1]:
        raise ValueError("bins_to_depth: %s probabilities, %d borders" % (tuple(depth_bin.shape), border.numel()))
    head = _head_of(depth_bin, depth_bin.shape[1])
    if head is not None:                      # this forward's softmax, straight from the HIP head: the private route
        return _FusedDepthFunction.apply(depth_bin, border, head)
    return _BinsToDepthFunction.apply(depth_bin, border)


def depth_to_bins(depth, depth_min, depth_max, dec_out_c):
    """modules/vnl.py:202-217.  Returns int32 bins shaped like `depth`; like the reference it also rewrites `depth`
    IN PLACE (clamped to [depth_min, depth_max]; -1 where it was negative = invalid padding, label dec_out_c + 1)."""
    _need_gpu(depth, "depth_to_bins")
    if depth.dtype != torch.float32 or not depth.is_contiguous():
        raise ValueError("depth_to_bins: a contiguous fp32 depth map is rewritten in place")
    dmin_log = np.log10(depth_min)
    interval = (np.log10(depth_max) - dmin_log) / dec_out_c
    bins = torch.empty(depth.shape, dtype=torch.int32, device=depth.device)
    ops.depth_to_bins(depth, float(depth_min), float(depth_max), float(dmin_log), float(interval), int(dec_out_c), bins)
    return bins
