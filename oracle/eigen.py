"""CPU fp32 oracle of the Eigen network (reference network/Eigen.py:5-90) — TEST INFRASTRUCTURE ONLY.

BASELINE.json configuration 1 is "Eigen coarse net, CPU PyTorch forward + SILog loss (plumbing, no GPU)": this module is
that plumbing.  `EigenOracle` holds the reference's parameter tree (same state_dict keys and shapes: VGG-19-BN features from
the torchvision stand-in of oracle/trunks.py, the two Linear layers, the transposed convs, Scale2 / Scale3 stacks) and its
forward is the functional restatement nets.eigen_forward; pinned against the reference's own class by
tests/golden/eigen.npz.  The reference's two Linear layers fix the input at 240 x 320 (Eigen.py:77-78): the 64 x 64 of
BASELINE.json's wording is rejected by the reference itself, so the configuration runs at 4 x 3 x 240 x 320.
"""
import torch.nn as nn

from . import nets, trunks


class _VGG(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature_extractor = trunks.VGG19BN().features
        self.flatten = nn.Flatten()
        self.mlp1 = nn.Linear(512 * 10 * 7, 4096)
        self.mlp2 = nn.Linear(4096, 64 * 19 * 14)
        self.upsample = nn.ConvTranspose2d(64, 64, kernel_size=3, stride=4)


class _Scale2(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 96, kernel_size=9, stride=2)
        self.scale2_onestack = nn.Sequential(
            nn.Conv2d(160, 64, 5, padding=2), nn.ReLU(), nn.Conv2d(64, 64, 5, padding=2), nn.ReLU(), nn.Conv2d(64, 64, 5, padding=2), nn.ReLU(),
            nn.ConvTranspose2d(64, 1, kernel_size=5, padding=2, stride=2))


class _Scale3(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 96, kernel_size=9, stride=2)
        self.scale3_onestack = nn.Sequential(
            nn.Conv2d(97, 64, 5, padding=2), nn.ReLU(), nn.Conv2d(64, 64, 5, padding=2), nn.ReLU(), nn.Conv2d(64, 64, 5, padding=2), nn.ReLU(),
            nn.Conv2d(64, 1, 5, padding=2), nn.ReLU())


class EigenOracle(nn.Module):
    def __init__(self):
        super().__init__()
        self.scale1, self.scale2, self.scale3 = _VGG(), _Scale2(), _Scale3()

    def forward(self, img, q=None):
        """q: the storage-rounding hook of nets.eigen_forward (nets.bf16_round / nets.rounding_draw(k)); None = plain fp32."""
        P = dict(self.named_parameters())
        P.update(dict(self.named_buffers()))
        # (every BatchNorm here has the default momentum; weights.calibrate_running_stats sets them all to 1.0 for one pass)
        return nets.eigen_forward(P, img, self.training, momentum=self.scale1.feature_extractor[1].momentum, q=q)
