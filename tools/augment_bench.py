#!/usr/bin/env python3
"""Per-sample time of the input pipeline (base_module.py:234-265 with FCRNModule's sizes: 480 x 640 -> resize 250 -> rotate ->
resize -> crop 240 x 320), three depth layers: the device path against PIL on one host core (what one DataLoader worker does)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_depth_estimation_amd import augment  # noqa: E402
from oracle import augment as OA  # noqa: E402

rng = np.random.RandomState(0)
rgb = torch.from_numpy(rng.rand(3, 480, 640).astype(np.float32))
depth = [torch.from_numpy(rng.rand(1, 480, 640).astype(np.float32)) for _ in range(3)]
rgb_d, depth_d = rgb.cuda(), [d.cuda() for d in depth]
n = 200
np.random.seed(0)
for _ in range(10):
    augment.train_preprocess(rgb_d, depth_d, 250, (240, 320))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    augment.train_preprocess(rgb_d, depth_d, 250, (240, 320))
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / n
torch.set_num_threads(1)
np.random.seed(0)
t0 = time.perf_counter()
for _ in range(40):
    OA.train_preprocess(rgb, depth, 250, (240, 320))
t_cpu = (time.perf_counter() - t0) / 40
print("device: %.3f ms per sample (%.0f samples/s, host-launch bound);  PIL, one core: %.3f ms per sample (%.0f samples/s)" % (
    1e3 * t_dev, 1 / t_dev, 1e3 * t_cpu, 1 / t_cpu))
