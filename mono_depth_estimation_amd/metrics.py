"""Drop-in for the depth metrics of reference metrics.py:58-123, computed on device in one
pass (no per-step device->host copy as in metrics.py:63).  'rmse' is the reference's
RelativeMeanSquareError = mean(sqrt((p-t)^2/t)) (metrics.py:106-109,122), reproduced as is.
'mae' / 'mse' / 'msle' are what the reference maps those names to (torchmetrics 0.7.3
mean_absolute_error / mean_squared_error / mean_squared_log_error, metrics.py:116-121: plain
means over the masked vectors, msle on log1p).  'ssim' (torchmetrics' SSIM on CPU copies,
metrics.py:63,123) is not provided: torchmetrics is not in this image, so its exact windowing
could not be pinned.
"""
import torch

from . import ops

NAMES = ("absrel", "rmse", "delta1", "delta2", "delta3", "log10", "mae", "mse", "msle", "sqrel")


class MetricComputation(object):
    """Same interface as the reference class: names, compute(pred, target), avg(metric), reset()."""

    def __init__(self, metrics):
        for m in metrics:
            if m not in NAMES:
                raise NotImplementedError("metric '%s' has no HIP kernel (available: %s)" % (m, ", ".join(NAMES)))
        self.names = list(metrics)
        self.reset()

    def reset(self):
        self.count = 0
        self.sum = [0.0 for _ in self.names]

    def compute(self, pred, target):
        if not pred.is_cuda:
            raise RuntimeError("mono_depth_estimation_amd.metrics runs on MI355X only; no CPU fallback")
        pred = pred.detach().contiguous().float()
        target = target.detach().contiguous().float()
        out = torch.empty(len(NAMES), device=pred.device)
        ops.depth_metrics(pred, target, ops.metrics_ws(pred.device), out)
        vals = [out[NAMES.index(n)] for n in self.names]
        self.count += 1
        for i, v in enumerate(vals):
            self.sum[i] = self.sum[i] + v
        return vals

    def avg(self, metric):
        if isinstance(metric, int):
            return self.sum[metric] / self.count
        if isinstance(metric, str):
            return self.sum[self.names.index(metric)] / self.count
        assert False, "metric must be int or str"
