"""CPU fp32 oracle of the depth metrics — TEST INFRASTRUCTURE ONLY.

Restates reference metrics.py:58-123.  ``compute`` follows MetricComputation.compute
(:58-67): clamp pred >= 1e-7, keep pixels with target > 0, then the pure functions.
NB the reference's 'rmse' key is RelativeMeanSquareError = mean(sqrt((p-t)^2 / t))
(metrics.py:106-109,122), not a true RMSE; reproduced as is.  'mae' / 'mse' / 'msle' are torchmetrics 0.7.3 functions in the reference
(metrics.py:116-121; the package is absent here): restated from their published definitions
(plain means; msle on log1p) — pinned by definition only, unlike the other seven.
"""
import torch

NAMES = ("absrel", "rmse", "delta1", "delta2", "delta3", "log10", "mae", "mse", "msle", "sqrel")
PINNED = ("absrel", "rmse", "delta1", "delta2", "delta3", "log10", "sqrel")      # present in tests/golden/metrics.npz


def compute(pred, target):
    pred = torch.clamp_min(pred, 1e-7)
    v = target > 0
    p, t = pred[v], target[v]
    ratio = torch.max(p / t, t / p)
    return {
        "absrel": ((p - t).abs() / t).mean(),
        "rmse": torch.sqrt((p - t) ** 2 / t).mean(),
        "delta1": (ratio < 1.25).float().mean(),
        "delta2": (ratio < 1.25 ** 2).float().mean(),
        "delta3": (ratio < 1.25 ** 3).float().mean(),
        "log10": (torch.log10(p) - torch.log10(t)).abs().mean(),
        "mae": (p - t).abs().mean(),
        "mse": ((p - t) ** 2).mean(),
        "msle": ((torch.log1p(p) - torch.log1p(t)) ** 2).mean(),
        "sqrel": ((p - t) ** 2 / t).mean(),
    }


def ssim(pred, target):
    """'ssim' of the reference's metric list (metrics.py:63,123): torchmetrics 0.7.3's
    structural_similarity_index_measure(preds, target) with its defaults, as MetricComputation.compute calls it -- on the
    prediction clamped to >= 1e-7 and the UNMASKED target.  torchmetrics is absent from the image: restated step by step from
    its published functional/image/ssim.py (pinned by definition only): 11 x 11 Gaussian (sigma 1.5) as an outer product of
    the normalised 1-D kernel, data_range = max(preds.max() - preds.min(), target.max() - target.min()), reflect padding
    by 5, one grouped convolution over [p, t, p*p, t*t, p*t], the SSIM map, then the padding's width cropped from every
    side of the map before the mean."""
    import torch.nn.functional as F
    pred = torch.clamp_min(pred, 1e-7).double()
    target = target.double()
    c = pred.shape[1]
    data_range = max(float(pred.max() - pred.min()), float(target.max() - target.min()))
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    dist = torch.arange((1 - 11) / 2, (1 + 11) / 2, 1, dtype=torch.float64)
    g = torch.exp(-(dist / 1.5) ** 2 / 2)
    g = (g / g.sum()).unsqueeze(0)
    kernel = (g.t() @ g).expand(c, 1, 11, 11)
    p, t = F.pad(pred, (5, 5, 5, 5), mode="reflect"), F.pad(target, (5, 5, 5, 5), mode="reflect")
    out = F.conv2d(torch.cat((p, t, p * p, t * t, p * t)), kernel, groups=c)
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(pred.shape[0])
    s_p, s_t, s_pt = e_pp - mu_p ** 2, e_tt - mu_t ** 2, e_pt - mu_p * mu_t
    idx = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_p + s_t + c2))
    return idx[..., 5:-5, 5:-5].mean().float()
