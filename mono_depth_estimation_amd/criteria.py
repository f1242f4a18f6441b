"""Drop-in for the reference's criteria.py depth losses (same class names, constructors and call
signatures; 0-dim result with grad; the reductions and their gradients are wavefront-reduced HIP
kernels in csrc/losses.hip; no CPU fallback):

  silog_loss(variance_focus)(depth_est, depth_gt)      criteria.py:724-732   (the FCRN bench loss)
  MaskedL1Loss()(pred, target), MaskedMSELoss(), berHuLoss()   criteria.py:67-90,113-133
  MaskedDepthLoss()(pred, target)                      criteria.py:17-64    (Eigen's loss)
"""
import torch
import torch.nn as nn

from . import ops


class _SilogFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, gt, variance_focus):
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
        return grad, None, None


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
        return grad, None, None


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
        return grad, None


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
