"""Drop-in for the reference's composite "stdepth" criterion — the closure `BaseModule.setup_criterion` builds
(modules/base_module.py:124-208) and every stdepth-trained module calls as `self.criterion(y_hat, y, rgba)`
(laina default 'mae+composite', bts default 'silma').  Same call signature and return tuple; all arithmetic in
csrc/stdepth_loss.hip (masked sums, alpha compositing, depth sort, separable-Gaussian DSSIM, fwd + bwd); no CPU
fallback.

    criterion = setup_criterion(method, single_layer=True)
    loss, = criterion(pred, targ, rgba)
    loss, pred_full, terms = criterion(pred, targ, rgba, return_composited=True, return_loss_dict=True)

`method` carries loss, variance_focus, depth_loss_weight, comp_loss_weight, fbdiv_loss_weight, ssim_loss_weight
(base_module.py:125-131,331-334).  Terms are selected by SUBSTRING of method.loss, as the reference does."""
import torch

from . import ops

SILMA, SILMS, MSE, MAE, ALLSSIM, COLORSSIM, COMPOSITE, COMPOSITE_SSIM, FBDIV = (1 << i for i in range(9))

# (bit, name, slot in the kernel's out[12]) in the order the reference fills its loss_dict
_ORDER = [(SILMA, "depth_silog", 1), (SILMA, "color_mae", 2), (SILMS, "depth_silog", 1), (SILMS, "color_mse", 3),
          (MSE, "all_mse", 4), (MAE, "all_mae", 5), (ALLSSIM, "all_ssim", 6), (COLORSSIM, "front_ssim", 7),
          (COLORSSIM, "back_ssim", 8), (COMPOSITE, "composite_mse", 9), (COMPOSITE_SSIM, "composite_ssim", 10),
          (FBDIV, "fb_divergence", 11)]


def term_mask(loss):
    """base_module.py:156-196: which terms a loss string switches on."""
    t = 0
    for key, bit in (("silma", SILMA), ("silms", SILMS), ("mse", MSE), ("mae", MAE), ("allssim", ALLSSIM),
                     ("colorssim", COLORSSIM), ("composite", COMPOSITE), ("fbdivergence", FBDIV)):
        if key in loss:
            t |= bit
    if (t & COMPOSITE) and "ssim" in loss:
        t |= COMPOSITE_SSIM
    return t


class _StdepthFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, targ, rgba, terms, weights, want_full, single):
        ctx.in_dtype = pred.dtype
        p = pred.contiguous().float()
        t = targ.to(device=p.device, dtype=torch.float32).contiguous()
        r = rgba.to(device=p.device, dtype=torch.float32).contiguous()
        N, C, H, W = p.shape
        ws = ops.stdepth_ws(p.device)
        scratch = ops.stdepth_scratch(N, C, H, W, terms, p.device)
        need_full = want_full or bool(terms & COMPOSITE_SSIM)
        full = torch.empty((N, 4, H, W), device=p.device) if need_full else None
        out = torch.empty(12, device=p.device)
        ops.stdepth_fwd(p, t, r, N, C, H, W, single, terms, weights, ws, scratch, full, out)
        ctx.save_for_backward(p, t, r, ws)
        ctx.extra = (terms, weights, scratch, full, single)
        if full is None:
            full = torch.empty(0, device=p.device)
        ctx.mark_non_differentiable(out, full)
        return out[0].clone(), out, full

    @staticmethod
    def backward(ctx, gout, _gterms, _gfull):
        p, t, r, ws = ctx.saved_tensors
        terms, weights, scratch, full, single = ctx.extra
        N, C, H, W = p.shape
        grad = torch.empty_like(p)
        ops.stdepth_bwd(p, t, r, N, C, H, W, single, terms, weights, ws, scratch, full, gout.contiguous().float().reshape(1), grad)
        return grad.to(ctx.in_dtype), None, None, None, None, None, None


def setup_criterion(method, single_layer=True):
    """Returns `_loss(pred, targ, rgba, return_composited=False, return_loss_dict=False)` -> tuple, as
    BaseModule.setup_criterion does.  pred / targ: [N, C, H, W] with C = 10 or 20.  single_layer (the reference's
    default, base_module.py:57) reads front / back RGBA from channels 0:8 and depths from 8:10 whatever C is — the laina
    module's default is out_channels = 20 WITH single_layer; single_layer=False is the 20-channel multi-layer layout."""
    terms = term_mask(method.loss)
    weights = (float(method.variance_focus), float(method.depth_loss_weight), float(method.comp_loss_weight),
               float(method.fbdiv_loss_weight), float(method.ssim_loss_weight))
    channels = (10, 20) if single_layer else (20,)

    def _loss(pred, targ, rgba, return_composited=False, return_loss_dict=False):
        if not pred.is_cuda:
            raise RuntimeError("mono_depth_estimation_amd.stdepth runs on MI355X only; no CPU fallback")
        if terms == 0:
            raise ValueError("stdepth criterion: loss string %r selects no term" % (method.loss,))
        if pred.ndim != 4 or pred.shape[1] not in channels or targ.shape != pred.shape:
            raise ValueError("stdepth criterion: pred %s / targ %s, expected [N, %s, H, W] (single_layer=%s)"
                             % (tuple(pred.shape), tuple(targ.shape), " or ".join(map(str, channels)), single_layer))
        if tuple(rgba.shape) != (pred.shape[0], 4) + tuple(pred.shape[2:]):
            raise ValueError("stdepth criterion: rgba %s for pred %s" % (tuple(rgba.shape), tuple(pred.shape)))
        if (terms & COMPOSITE) and not single_layer:
            # the reference indexes a 4-channel composite and a 5-channel target with one 4-channel mask here
            # (base_module.py:149,180) and raises; so do we
            raise ValueError("stdepth criterion: 'composite' is only well-formed with single_layer=True")
        loss, out, full = _StdepthFunction.apply(pred, targ, rgba, terms, weights, bool(return_composited), bool(single_layer))
        ret = [loss]
        if return_composited:
            ret.append(full)
        if return_loss_dict:
            d = {}
            for bit, name, slot in _ORDER:
                if terms & bit:
                    d[name] = out[slot].detach()
            ret.append(d)
        return tuple(ret)

    return _loss
