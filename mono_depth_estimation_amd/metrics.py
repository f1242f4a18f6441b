"""Drop-in for the depth metrics of reference metrics.py:58-123, computed on device in one
pass (no per-step device->host copy as in metrics.py:63).  'rmse' is the reference's
RelativeMeanSquareError = mean(sqrt((p-t)^2/t)) (metrics.py:106-109,122), reproduced as is.
'mae' / 'mse' / 'msle' are what the reference maps those names to (torchmetrics 0.7.3
mean_absolute_error / mean_squared_error / mean_squared_log_error, metrics.py:116-121: plain
means over the masked vectors, msle on log1p).  'ssim' (metrics.py:63,123: torchmetrics'
structural_similarity_index_measure on CPU copies of the clamped prediction and the UNMASKED
target) runs on the device too, from torchmetrics 0.7.3's published definition (11 x 11 Gaussian,
sigma 1.5, data_range from the tensors' extrema, the 5-pixel border cropped): torchmetrics is not
in this image, so that entry is pinned by definition only (oracle/metrics.py says the same).
"""
import torch

from . import ops

NAMES = ("absrel", "rmse", "delta1", "delta2", "delta3", "log10", "mae", "mse", "msle", "sqrel")
EXTRA = ("ssim",)                     # not part of the one-pass kernel: a windowed statistic over whole maps


class MetricComputation(object):
    """Same interface as the reference class: names, compute(pred, target), avg(metric), reset()."""

    def __init__(self, metrics):
        for m in metrics:
            if m not in NAMES + EXTRA:
                raise NotImplementedError("metric '%s' has no HIP kernel (available: %s)" % (m, ", ".join(NAMES + EXTRA)))
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
        vals = []
        for n in self.names:
            if n == "ssim":
                if pred.dim() < 3 or pred.shape != target.shape:
                    raise ValueError("'ssim' needs prediction and target maps of one shape [..., H, W]")
                o = torch.empty(1, device=pred.device)
                ops.ssim_metric(pred, target, o)
                vals.append(o[0])
            else:
                vals.append(out[NAMES.index(n)])
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
