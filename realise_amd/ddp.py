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
    """grad_dtype "bf16": the buckets cross the wire as bf16 copies (half the bytes of the 679 MB fp32 exchange, SURVEY.md
    section 5); the fp32 arena receives the reduced values back.  Pre-division by the world size happens in fp32 first."""

    def __init__(self, buckets, world, group, use_side_stream, grad_dtype="fp32", bucket_cap_mb=None, collect_stats=False, algo="allreduce"):
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_dtype must be 'fp32' or 'bf16'")
        if algo not in ("allreduce", "mesh"):
            raise ValueError("algo must be 'allreduce' or 'mesh'")
        # algo "mesh" (SURVEY.md section 5): the exchange written for a fully connected xGMI node instead of left to the library's
        # choice of ring - every rank sends shard j of the bucket straight to rank j (one all-to-all: W - 1 concurrent point-to-point
        # transfers per rank, one per link), adds the W copies of ITS shard in rank order in fp32 (also when the wire is bf16: one
        # rounding per element instead of W - 1), and an all-gather hands the reduced shards round.  2 (W - 1)/W of the bucket crosses
        # each rank's links in total, 1/W of it per link.  Every rank ends with bit-identical gradients by construction (each element
        # is reduced on exactly one rank).
        self.algo = algo
        self._mesh = {}
        self.buckets, self.world, self.group = buckets, world, group
        self.stream = torch.cuda.Stream() if use_side_stream else None
        self.wire = [torch.empty_like(b, dtype=torch.bfloat16) for b in buckets] if grad_dtype == "bf16" else None
        self.avg_op = dist.get_backend(group) == "nccl"
        self.works = []
        # bucket_cap_mb: an engine bucket (one backward stage: 60-230 MB in fp32) goes out as several collectives of at most this
        # many MB of wire bytes each (None: one collective per bucket).  On a point-to-point xGMI mesh a ring all-reduce is
        # per-link bound, so the knob trades launch count against how early the first bytes of a stage move.
        self.bucket_cap_mb = bucket_cap_mb
        self.collect_stats = bool(collect_stats) and use_side_stream
        self._ev = None
        self.last_stats = None

    def _chunks(self, t):
        if not self.bucket_cap_mb:
            return [t]
        n = max(1, int(self.bucket_cap_mb * (1 << 20)) // t.element_size())
        return [t[o:o + n] for o in range(0, t.numel(), n)]

    def wire_bytes(self):
        """bytes every rank hands to the collectives per step, bucket by bucket (communication order is the caller's)"""
        return [(self.wire[i] if self.wire is not None else b).numel() * (2 if self.wire is not None else 4) for i, b in enumerate(self.buckets)]

    def _mesh_bufs(self, key, n, dtype, device):
        b = self._mesh.get(key)
        if b is None:
            w = self.world
            b = (torch.empty(n, dtype=dtype, device=device), torch.empty(n // w, dtype=torch.float32, device=device),
                 torch.empty(n // w, dtype=dtype, device=device) if dtype != torch.float32 else None)
            self._mesh[key] = b
        return b

    def _launch_mesh(self, i):
        """direct reduce-scatter + all-gather of bucket i.  fp32 wire: the raw gradients cross the links and the 1 / W rides on the
        owner's shard sum - a pass over 1 / W of the bucket instead of a pre-divide pass over all of it.
        bf16 wire: the narrowing pass exists anyway and carries the 1 / W (overflow-safe), the shard sum IS the mean."""
        buf, w = self.buckets[i], self.world
        if self.wire is not None:
            torch.mul(buf, 1.0 / w, out=self.wire[i])
            src, post = self.wire[i], None
        else:
            src, post = buf, 1.0 / w
        for k, c in enumerate(self._chunks(src)):
            n = c.numel() - c.numel() % w
            if n:
                recv, acc, narrow = self._mesh_bufs((i, k), n, c.dtype, c.device)
                dist.all_to_all_single(recv, c[:n], group=self.group)              # row r of recv = rank r's copy of my shard
                torch.sum(recv.view(w, n // w), dim=0, dtype=torch.float32, out=acc)
                if post is not None:
                    acc.mul_(post)
                if narrow is not None:
                    narrow.copy_(acc)
                self.works.append((i, dist.all_gather_into_tensor(c[:n], acc if narrow is None else narrow, group=self.group, async_op=True)))
            if n != c.numel():                                                      # fewer than W trailing elements
                if post is not None:
                    c[n:].mul_(post)
                self.works.append((i, dist.all_reduce(c[n:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def _launch(self, i):
        buf = self.buckets[i]
        if self.collect_stats:
            self._mark(i, 0)
        if self.algo == "mesh":
            self._launch_mesh(i)
        elif self.wire is not None:
            # pre-divide and narrow in ONE pass over the bucket: SUM of bf16(g / W) == mean, overflow-safe
            torch.mul(buf, 1.0 / self.world, out=self.wire[i])
            for c in self._chunks(self.wire[i]):
                self.works.append((i, dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        elif self.avg_op:
            # RCCL averages inside the collective (ncclAvg): no pre-divide pass over the 679 MB arena
            for c in self._chunks(buf):
                self.works.append((i, dist.all_reduce(c, op=dist.ReduceOp.AVG, group=self.group, async_op=True)))
        else:
            buf.div_(self.world)                           # gloo has no AVG: pre-divide, then SUM
            for c in self._chunks(buf):
                self.works.append((i, dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    # ---- per-bucket timeline of the last step (collect_stats): when each bucket became ready on the communication stream, when its
    # collectives had completed, and when the backward itself was done - what says whether the exchange hides under the backward
    def _mark(self, i, which):
        if self._ev is None:
            self._ev = {"ready": [torch.cuda.Event(enable_timing=True) for _ in self.buckets],
                        "done": [torch.cuda.Event(enable_timing=True) for _ in self.buckets],
                        "bwd_done": torch.cuda.Event(enable_timing=True), "order": []}
        if which == 0:
            if not self._ev["order"] or len(self._ev["order"]) >= len(self.buckets):
                self._ev["order"] = []
            self._ev["order"].append(i)
            self._ev["ready"][i].record()
        else:
            self._ev["done"][i].record()

    def read_stats(self):
        """milliseconds relative to the first bucket's ready time; synchronises.  None before the first collected step."""
        if not self.collect_stats or self._ev is None or len(self._ev["order"]) != len(self.buckets):
            return None
        torch.cuda.synchronize()
        order = list(self._ev["order"])
        t0 = self._ev["ready"][order[0]]
        ready = [t0.elapsed_time(self._ev["ready"][i]) for i in order]
        done = [t0.elapsed_time(self._ev["done"][i]) for i in order]
        bwd = t0.elapsed_time(self._ev["bwd_done"])
        wb = self.wire_bytes()
        self.last_stats = {"comm_order": order, "bucket_wire_bytes": [wb[i] for i in order], "bucket_ready_ms": [round(x, 3) for x in ready],
                           "bucket_done_ms": [round(x, 3) for x in done], "backward_done_ms": round(bwd, 3),
                           "exposed_tail_ms": round(max(0.0, done[-1] - bwd), 3), "tail_bucket_wire_bytes": wb[order[-1]],
                           "bucket_cap_mb": self.bucket_cap_mb, "algo": self.algo, "collectives_per_step": (2 if self.algo == "mesh" else 1) * sum(len(self._chunks(self.wire[i] if self.wire is not None else self.buckets[i])) for i in order)}
        return self.last_stats

    # ---- event-driven form: the engine runs the whole (branch-overlapped) backward in one call and records events[i] when
    # bucket i is final; the communication stream waits on each event in a FIXED order (identical on every rank, as the
    # collectives require) and all-reduces the bucket while the rest of the backward runs.
    def bucket_events(self):
        if self.stream is None:
            return None
        if getattr(self, "events", None) is None:
            self.events = [torch.cuda.Event() for _ in self.buckets]
            for e in self.events:
                e.record()                                  # forces the handle into existence
        return self.events

    def buckets_signalled(self, order):
        with torch.cuda.stream(self.stream):
            for i in order:
                self.stream.wait_event(self.events[i])
                self._launch(i)

    def bucket_ready(self, i):
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self._launch(i)
        else:
            self._launch(i)

    def _drain(self):
        last = {i: k for k, (i, w) in enumerate(self.works)}
        for k, (i, w) in enumerate(self.works):
            w.wait()
            if last[i] == k:                    # the bucket's last collective
                if self.wire is not None:
                    self.buckets[i].copy_(self.wire[i])
                if self.collect_stats:
                    self._mark(i, 1)
        self.works = []

    def finish(self):
        if self.collect_stats and self._ev is not None:
            self._ev["bwd_done"].record(torch.cuda.current_stream())
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._drain()
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            self._drain()


class DistributedDataParallel(nn.Module):
    """``DistributedDataParallel(model)`` for a RealiseModule (or any module exposing
    ``bucket_views()``, ``flat_parameters()``, ``flat_bn_buffers()`` and a ``grad_sync`` slot)."""

    def __init__(self, module, process_group=None, broadcast_buffers=True, grad_dtype="fp32", bucket_cap_mb=None, collect_stats=False, algo="allreduce"):
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
        module.grad_sync = _GradSync(module.bucket_views(), self.world, process_group, flat.is_cuda, grad_dtype, bucket_cap_mb, collect_stats, algo)

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
