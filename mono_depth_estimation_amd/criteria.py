"""Drop-in for the reference's criteria.py losses that sit on the FCRN hot path.

``silog_loss(variance_focus)(depth_est, depth_gt)`` keeps the reference's constructor and
call signature (criteria.py:724-732) and returns a 0-dim tensor with grad; the reduction and
its gradient are wavefront-reduced HIP kernels (csrc/losses.hip).
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
