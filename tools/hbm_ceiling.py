#!/usr/bin/env python3
"""What streaming kernels reach on this box: torch's device copy (read + write) and a read-only sum, 1 GiB operands."""
import time
import torch
n = 1 << 29
a = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
b = torch.empty_like(a)
def t(f, k=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
dt = t(lambda: b.copy_(a)); print("copy  : %.2f TB/s (read + write)" % (2 * a.numel() * 2 / dt / 1e12))
dt = t(lambda: a.float().sum()) ; print("cast+sum (read 2B, write 4B, read 4B): %.2f TB/s" % (a.numel() * 10 / dt / 1e12))
c = torch.empty(n // 2, dtype=torch.float32, device="cuda").normal_()
dt = t(lambda: c.sum()); print("sum   : %.2f TB/s (read only)" % (c.numel() * 4 / dt / 1e12))
dt = t(lambda: torch.add(a, b, out=b)); print("add   : %.2f TB/s (2 reads + 1 write)" % (3 * a.numel() * 2 / dt / 1e12))
