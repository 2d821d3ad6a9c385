"""A bench.py stand-in for the CPU suite (tools/scale_sweep.py --bench): the REAL gradient exchange of realise_amd/ddp.py (_GradSync: bucketed
all-reduce or mesh reduce-scatter + all-gather, fp32 or bf16 wire) over gloo on CPU buckets shaped like the engine's seven - no model, no GPU.
Prints one bench-contract JSON line from rank 0; checks on every rank that the buckets hold the mean afterwards."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realise_amd.ddp import _GradSync  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ddp-algo", default="allreduce")
    ap.add_argument("--grad-dtype", default="fp32")
    ap.add_argument("--backend", default="gloo")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    assert world == a.gpus
    if world > 1:
        dist.init_process_group(backend="gloo")
    sizes = [1001, 4096, 777, 2048, 2048, 513, 3000]
    buckets = [torch.zeros(n) for n in sizes]
    sync = _GradSync(buckets, world, None, False, a.grad_dtype, None, False, a.ddp_algo) if world > 1 else None
    t0 = None
    for step in range(a.warmup + a.steps):
        if step == a.warmup:
            t0 = time.perf_counter()
        for i, b in enumerate(buckets):
            b.copy_(torch.arange(b.numel(), dtype=torch.float32) * 1e-3 + (rank + 1) * (i + 1))
        if sync is not None:
            for i in range(len(buckets)):
                sync.bucket_ready(i)
            sync.finish()
            for i, b in enumerate(buckets):
                want = torch.arange(b.numel(), dtype=torch.float32) * 1e-3 + (world + 1) / 2.0 * (i + 1)
                tol = 1e-5 if a.grad_dtype == "fp32" else 2e-2
                assert (b - want).abs().max().item() <= tol * want.abs().max().item(), (i, a.ddp_algo, a.grad_dtype)
    dt = (time.perf_counter() - t0) / a.steps
    if rank == 0:
        wire = 2 if a.grad_dtype == "bf16" else 4
        print(json.dumps({"metric": "fake exchange steps/sec", "value": world / dt, "unit": "steps/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
                          "ddp": None if sync is None else {"algo": a.ddp_algo, "bucket_wire_bytes": [n * wire for n in sizes], "exposed_tail_ms": 0.0,
                                                            "backward_done_ms": 0.0, "collectives_per_step": (2 if a.ddp_algo == "mesh" else 1) * len(sizes)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
