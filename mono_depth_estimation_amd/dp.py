"""Data-parallel gradient exchange for the flat gradient buffer (one process per GPU).

The reference reaches multi-GPU only through PyTorch-Lightning's DDP (train.py:132-145):
per-rank BatchNorm statistics, per-rank loss, one gradient all-reduce (mean) per step.
Here the whole gradient is one flat fp32 buffer laid out in FORWARD order, so backward
completes it from the tail to the head; it is cut into a few large contiguous buckets that
are all-reduced (RCCL over xGMI; `nccl` backend on ROCm) on a side stream as soon as
backward has passed the bucket's first element, overlapping the exchange with the rest of
backward.  Few, large messages: xGMI is point-to-point, a ring is bound by one link.
The averaging (1/world) is folded into the optimiser's grad_scale, not a separate pass.
"""
import os

import torch
import torch.distributed as dist


def make_buckets(total, boundaries, target_bytes, elem_bytes=4, head_bytes=None):
    """Cut [0, total) into contiguous buckets, walking from the TAIL (backward order), closing a
    bucket at the first layer boundary after it reached target_bytes.  `boundaries`: sorted
    element offsets at which layers start.  Returns [(start, end)] in launch (tail-first) order.
    The LAST bucket (the head of the buffer: the first layers, whose gradients backward finishes last) cannot overlap
    with anything — its exchange is exposed in full — so a small piece of at most head_bytes (default target / 8) is cut
    off its front at a layer boundary: the bulk of it goes out while the first layers are still in backward."""
    cuts = sorted(set(b for b in boundaries if 0 < b < total))
    buckets, end = [], total
    want = max(1, target_bytes // elem_bytes)
    for b in reversed(cuts):
        if end - b >= want:
            buckets.append((b, end))
            end = b
    if end > 0:
        head = max(1, (head_bytes if head_bytes is not None else target_bytes // 8) // elem_bytes)
        inner = [b for b in cuts if b < end and b <= head]
        if inner and end > 2 * head:
            buckets.append((inner[-1], end))
            end = inner[-1]
        buckets.append((0, end))
    return buckets


class _Pending:
    """A wire-format bucket whose collective is in flight: wait() joins it and widens the result back into the gradient."""

    def __init__(self, work, view, buf, n):
        self.work, self.view, self.buf, self.n = work, view, buf, n

    def wait(self):
        self.work.wait()
        self.view.copy_(self.buf[:self.n])


class FlatGradReducer:
    """Sum-all-reduce of a flat gradient tensor in tail-first buckets, overlapped with backward.

    ready(offset): backward promises every gradient element >= offset is final.
    finish(): wait for all buckets (call before the optimiser step)."""

    def __init__(self, flat, boundaries, target_bytes=64 << 20, group=None, extra_streams=(), wire_dtype=None,
                 algo=None):
        """extra_streams: streams besides the current one that also write gradients (the engine's
        weight-gradient stream); a bucket's all-reduce waits for the work queued on them as well, so the
        producer does not have to join them into the main stream at every layer.
        wire_dtype: torch.bfloat16 sends each bucket as bf16 (half the bytes on xGMI: 127 MB instead of 254 MB
        for FCRN-50): the bucket is cast into a wire buffer on the exchange stream, reduced
        there, and cast back into the fp32 gradient; None keeps fp32 on the wire.
        algo: "allreduce" (default) or "rs_ag" ($MDE_DP_ALGO): reduce-scatter + all-gather of the (padded) wire
        buffer, the two halves of a ring all-reduce as separate collectives, each rank reducing one shard."""
        self.flat, self.group = flat, group
        self.wire_dtype = wire_dtype if wire_dtype is not None and wire_dtype != flat.dtype else None
        self.algo = algo or os.environ.get("MDE_DP_ALGO", "allreduce")
        if self.algo not in ("allreduce", "rs_ag"):
            raise ValueError("FlatGradReducer: algo must be 'allreduce' or 'rs_ag', got %r" % (self.algo,))
        backend = dist.get_backend(group) if dist.is_initialized() else None
        if self.algo == "rs_ag" and backend is not None and backend != "nccl":
            raise ValueError("FlatGradReducer: algo='rs_ag' needs reduce_scatter_tensor / all_gather_into_tensor, which the "
                             "%r backend does not implement; use algo='allreduce' there" % (backend,))
        # nccl (RCCL): Work.wait() only orders the CURRENT stream behind the collective, so the widening copy can be queued
        # right away on the exchange stream; other backends (gloo, the CPU rehearsal) block the host in wait(), so there the
        # wait and the copy-back are deferred to finish() and backward keeps running while the bucket is on the wire
        self._defer = backend != "nccl"
        self.extra_streams = [s for s in extra_streams if s is not None]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # MDE_DP_FORCE=1: run the collectives even with one rank (rehearses the RCCL / stream / event
        # path on a single GPU; an all-reduce over one rank is the identity)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("MDE_DP_FORCE") == "1")
        if os.environ.get("MDE_DP_DISABLE") == "1":     # diagnostics: no exchange at all (the multi-rank test's negative control)
            self.active = False
        self.buckets = make_buckets(flat.numel(), boundaries, target_bytes, flat.element_size())
        self.stream = torch.cuda.Stream() if flat.is_cuda else None
        self.early = 0
        self.wire = {}                     # bucket start -> (wire buffer, shard) when the wire format differs / rs_ag
        if self.active and (self.wire_dtype is not None or self.algo == "rs_ag"):
            dt = self.wire_dtype or flat.dtype
            for start, end in self.buckets:
                n = end - start
                npad = -(-n // self.world) * self.world if self.algo == "rs_ag" else n
                buf = torch.zeros(npad, dtype=dt, device=flat.device)
                shard = torch.empty(npad // self.world, dtype=dt, device=flat.device) if self.algo == "rs_ag" else None
                self.wire[start] = (buf, shard)
        self.reset()

    def reset(self):
        self.next, self.works = 0, []
        self._early = 0

    def begin(self, flat):
        """A backward pass is about to accumulate into `flat` (same layout; the autograd path alternates between two flat
        buffers: FlatStore.begin_autograd_backward)."""
        assert flat.numel() == self.flat.numel() and flat.dtype == self.flat.dtype
        self.flat = flat
        self.reset()

    def _exchange(self, start, end):
        """Sum one bucket over the ranks (runs on the exchange stream when there is one)."""
        view = self.flat[start:end]
        if start not in self.wire:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        buf, shard = self.wire[start]
        n = end - start
        buf[:n].copy_(view)                                     # fp32 -> wire dtype (padding stays zero)
        if shard is None:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            work = dist.all_gather_into_tensor(buf, shard, group=self.group, async_op=True)   # same communicator: ordered
        if self._defer:
            self.works.append(_Pending(work, view, buf, n))
        else:
            work.wait()                                         # orders the exchange stream, does not block the host
            view.copy_(buf[:n])

    def _launch(self, start, end):
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                for s in self.extra_streams:
                    self.stream.wait_stream(s)
                self._exchange(start, end)
        else:
            self._exchange(start, end)

    def ready(self, offset):
        if not self.active:
            return
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= offset:
            self._launch(*self.buckets[self.next])
            self.next += 1
            self._early += int(offset > 0)         # issued while backward was still running (diagnostics: `early`)

    def finish(self):
        if self.active:
            self.ready(0)
            self.early = self._early               # buckets of the pass just joined that went out before backward ended
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    for w in self.works:
                        w.wait()
            else:
                for w in self.works:
                    w.wait()
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
        self.reset()
