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
"""
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
