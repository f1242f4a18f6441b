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
