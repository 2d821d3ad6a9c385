"""Data-parallel wrapper for the engine-backed modules (collectives C1-C3).

The reference wraps the model in ``DistributedDataParallel(find_unused_parameters=True)``
(src/run.py:165-167).  Here the whole backward is one autograd node that writes into a flat
gradient arena laid out in backward-completion order, so DDP reduces to: after each backward
STAGE finishes, all-reduce the matching contiguous slice of the arena on a side stream while the
next stage computes (no gradient copies, no bucket rebuild, the never-used parameters are simply
not in the arena).  One process per GPU; backend "nccl" is RCCL over xGMI on ROCm; "gloo" runs
the same code path on CPU tensors for the tests.
"""
import torch
import torch.distributed as dist
from torch import nn


class _GradSync:
    def __init__(self, buckets, world, group, use_side_stream):
        self.buckets, self.world, self.group = buckets, world, group
        self.stream = torch.cuda.Stream() if use_side_stream else None
        self.works = []

    def bucket_ready(self, i):
        buf = self.buckets[i]
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                buf.div_(self.world)                       # pre-divide: SUM of (g / W) == mean, overflow-safe
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            buf.div_(self.world)
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.works:
            w.wait()
        self.works = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class DistributedDataParallel(nn.Module):
    """``DistributedDataParallel(model)`` for a RealiseModule (or any module exposing
    ``bucket_views()``, ``flat_parameters()``, ``flat_bn_buffers()`` and a ``grad_sync`` slot)."""

    def __init__(self, module, process_group=None, broadcast_buffers=True):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.broadcast_buffers = broadcast_buffers
        flat = module.flat_parameters()
        dist.broadcast(flat, src=0, group=process_group)                       # C3: rank 0's parameters win
        if hasattr(module, "_arenas"):
            for a in (module._arenas[1], module._arenas[2]):
                if a.numel() > 1:
                    dist.broadcast(a, src=0, group=process_group)
            module.mark_parameters_updated()
        module.grad_sync = _GradSync(module.bucket_views(), self.world, process_group, flat.is_cuda)

    def forward(self, *args, **kw):
        if self.broadcast_buffers and self.module.training:
            buf = self.module.flat_bn_buffers()
            if buf.numel() > 1:
                dist.broadcast(buf, src=0, group=self.group)                     # C2: BN running stats, rank 0 -> all
        return self.module(*args, **kw)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)
