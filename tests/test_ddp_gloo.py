"""world_size-2 gloo test of the data-parallel wrapper on CPU (the N>1 path of bench.py): parameter
broadcast (C3), BatchNorm-buffer broadcast (C2) and the bucket-by-bucket gradient all-reduce (C1) over the
module's real flat arenas and bucket layout.  The engine itself needs a GPU, so the backward stages are
emulated by writing rank-dependent values into each bucket before signalling it ready."""
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


def _mesh_worker(rank, world, port, grad_dtype, cap_mb, q):
    """the direct reduce-scatter + all-gather exchange against the analytic mean, on buckets whose lengths are NOT multiples of the
    world size and whose contents differ per element and per rank (a shard routed to the wrong rank or row would show)"""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from realise_amd.ddp import _GradSync
        sizes = [1000003, 7, 4099, world - 1 if world > 1 else 1, 65536]
        base = [torch.arange(n, dtype=torch.float32).remainder_(251.0) - 100.0 for n in sizes]
        buckets = [b * float(rank + 1) + float(i) for i, b in enumerate(base)]
        sync = _GradSync(buckets, world, None, False, grad_dtype, cap_mb, False, "mesh")
        ref = _GradSync([b.clone() for b in buckets], world, None, False, grad_dtype, cap_mb, False, "allreduce")
        for s in (sync, ref):
            for i in range(len(sizes)):
                s.bucket_ready(i)
            s.finish()
        mean_scale = sum(r + 1 for r in range(world)) / world
        for i, b in enumerate(base):
            want = b * mean_scale + float(i)
            got = sync.buckets[i]
            if grad_dtype == "fp32":
                assert torch.allclose(got, want, rtol=1e-6, atol=1e-5), (i, (got - want).abs().max())
            else:
                # bf16 wire: inputs and the gathered result are rounded to bf16; the shard sum itself runs in fp32, so the mesh form is
                # at least as close to the exact mean as the bf16 all-reduce
                assert torch.allclose(got, want, rtol=2e-2, atol=2e-2), (i, (got - want).abs().max())
                assert (got - want).abs().max() <= (ref.buckets[i] - want).abs().max() + 1e-6
        # every rank ends with the SAME bits
        flat = torch.cat([b.reshape(-1) for b in sync.buckets])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world,grad_dtype,cap_mb", [(2, "fp32", None), (3, "fp32", 0.5), (2, "bf16", None), (3, "bf16", 1.0)])
def test_mesh_reduce_scatter_all_gather_gloo(world, grad_dtype, cap_mb):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mesh_worker, args=(r, world, port, grad_dtype, cap_mb, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _worker(rank, world, port, model_type, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from realise_amd.config import RealiseConfig
        from realise_amd.ddp import DistributedDataParallel
        from realise_amd.modeling import SpellBert, SpellBertPho2ResArch3
        cfg = RealiseConfig(num_hidden_layers=1, vocab_size=21128)
        cls = SpellBertPho2ResArch3 if model_type == "arch3" else SpellBert
        torch.manual_seed(rank)
        m = cls(cfg, compute_dtype="fp32", seed=100 + rank, init_scheme="perturbed")      # ranks start DIFFERENT
        before = m.flat_parameters().clone()
        ddp = DistributedDataParallel(m)
        # C3: everyone now holds rank 0's parameters
        gathered = [torch.empty_like(before[:4096]) for _ in range(world)]
        dist.all_gather(gathered, m.flat_parameters()[:4096].contiguous())
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        if rank != 0:
            assert not torch.equal(before[:4096], m.flat_parameters()[:4096])
        # C1: emulate the staged backward: fill bucket i, signal it, finish
        buckets = m.bucket_views()
        assert len(buckets) >= 2
        sizes = [b.numel() for b in buckets]
        assert sum(sizes) == m.flat_gradients().numel()
        for i, b in enumerate(buckets):
            b.fill_(float((rank + 1) * (i + 1)))
            m.grad_sync.bucket_ready(i)
        m.grad_sync.finish()
        mean_factor = sum(r + 1 for r in range(world)) / world
        for i, b in enumerate(buckets):
            assert torch.allclose(b, torch.full_like(b, mean_factor * (i + 1))), i
        # C2: BN running statistics follow rank 0 at every training forward
        if model_type == "arch3":
            buf = m.flat_bn_buffers()
            buf.fill_(float(rank + 7))
            m.train()
            try:
                ddp({"src_idx": torch.zeros(1, 4, dtype=torch.long)})
            except Exception:
                pass                                    # the engine refuses CPU tensors; the broadcast ran first
            assert torch.allclose(buf, torch.full_like(buf, 7.0))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("model_type", ["bert", "arch3"])
def test_ddp_wrapper_world2_gloo(model_type):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, model_type, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_scale_sweep_tool_world2_gloo(tmp_path):
    """tools/scale_sweep.py (the table the first 8-GPU run prints: gpus x exchange x wire dtype with exposed_tail_ms) driven end to end on
    CPU: world sizes 1 and 2 over gloo, both exchanges, both wire dtypes, through torch.distributed.run - with tests/fake_bench_ddp.py
    standing in for bench.py (the real _GradSync exchange on engine-shaped buckets, checked against the mean on every rank)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "rows.json"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scale_sweep.py"), "--gpus", "1", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--bench", os.path.join(root, "tests", "fake_bench_ddp.py"), "--master-port", "29741", "--json", str(out), "--timeout", "300"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = json.load(open(out))
    assert [(x["gpus"], x["algo"], x["grad_dtype"]) for x in rows] == [(1, "-", "-"), (2, "allreduce", "fp32"), (2, "allreduce", "bf16"),
                                                                     (2, "mesh", "fp32"), (2, "mesh", "bf16")]
    assert all(x["error"] is None and x["value"] > 0 for x in rows)
    assert rows[3]["collectives"] == 14 and rows[1]["collectives"] == 7 and rows[2]["wire_mb"] < rows[1]["wire_mb"]
    assert "exposed_tail_ms" in r.stdout and "| 2 | mesh | bf16 |" in r.stdout
