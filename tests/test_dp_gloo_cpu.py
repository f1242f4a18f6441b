"""The N > 1 path on CPU: two processes, gloo backend, the flat-gradient bucket reducer
(mono_depth_estimation_amd/dp.py) driven tail-first exactly as the engine's backward does."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mono_depth_estimation_amd import dp
    n = 10007
    bounds = list(range(0, n, 1300))
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = dp.FlatGradReducer(g, bounds, target_bytes=8192)
    assert len(red.buckets) > 2
    for off in reversed(bounds):          # backward walks the forward-ordered buffer from its tail
        red.ready(off)
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(g, expect)
    # second step reuses the reducer
    g.fill_(float(rank))
    red.ready(0)
    red.finish()
    ok = ok and torch.equal(g, torch.full((n,), float(sum(range(world)))))
    # bf16 wire format (half the bytes on the links): small integers survive the round trip exactly
    g = torch.arange(n, dtype=torch.float32).remainder(64.0) * (rank + 1)
    red = dp.FlatGradReducer(g, bounds, target_bytes=8192, wire_dtype=torch.bfloat16)
    for off in reversed(bounds):
        red.ready(off)
    red.finish()
    ok = ok and torch.equal(g, torch.arange(n, dtype=torch.float32).remainder(64.0) * sum(r + 1 for r in range(world)))
    # ... and rounds like bf16 does when they do not
    g = torch.full((n,), 1.0 + 2.0 ** -10)
    red = dp.FlatGradReducer(g, bounds, target_bytes=8192, wire_dtype=torch.bfloat16)
    red.ready(0)
    red.finish()
    ok = ok and torch.equal(g, torch.full((n,), float(world)))
    q.put((rank, ok, len(red.buckets)))
    dist.destroy_process_group()


def test_flat_grad_reducer_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
