#!/usr/bin/env python3
"""First-8-GPU-run helper (readiness: no scaling curve exists in this tree - gpurun boxes have one GPU).

Runs bench.py over  gpus x {allreduce, mesh} x {fp32, bf16}  on ONE node through torch.distributed.run and prints ONE table: whole-job
sentences/s, ms/step, scaling efficiency against the 1-GPU row of the same sweep, and - from the `ddp` object of each line - how much of
the gradient exchange was left exposed behind the backward (`exposed_tail_ms`), so that the first multi-GPU run says by itself which
exchange / wire dtype to make the default (reference: src/run.py:165-167 DistributedDataParallel, train.sh:5).

    python tools/scale_sweep.py                          # gpus 1 2 4 8, both exchanges, both wire dtypes, RCCL
    python tools/scale_sweep.py --gpus 1 2 --backend gloo --bench tests/fake_bench_ddp.py      # what the CPU suite runs (world 2, gloo)

Every run is `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <bench> --gpus N
--steps K --warmup W [--ddp-algo A --grad-dtype D --backend B]`; N = 1 runs the bench directly.  A failed cell is reported, not fatal.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cell(bench, n, algo, dtype, backend, steps, warmup, port, extra, timeout):
    args = [bench, "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup)] + list(extra)
    if n > 1:
        args += ["--ddp-algo", algo, "--grad-dtype", dtype, "--backend", backend]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return None, "timeout after %d s" % timeout
    line = None
    for ln in r.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if r.returncode != 0 or line is None:
        return None, "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:].replace("\n", " | "))
    return json.loads(line), None


def sweep(bench, gpus, algos, dtypes, backend, steps, warmup, port, extra, timeout):
    rows, base = [], None
    for n in gpus:
        cells = [("-", "-")] if n == 1 else [(a, d) for a in algos for d in dtypes]
        for k, (algo, dtype) in enumerate(cells):
            out, err = run_cell(bench, n, algo, dtype, backend, steps, warmup, port + len(rows), extra, timeout)
            row = {"gpus": n, "algo": algo, "grad_dtype": dtype, "error": err}
            if out is not None:
                d = out.get("ddp") or {}
                row.update(value=out["value"], ms_per_step=out["ms_per_step"], unit=out.get("unit", ""),
                           exposed_tail_ms=d.get("exposed_tail_ms"), backward_done_ms=d.get("backward_done_ms"),
                           wire_mb=round(sum(d.get("bucket_wire_bytes", [])) / 1e6, 1) if d else None,
                           collectives=d.get("collectives_per_step"))
                if n == 1 and base is None:
                    base = out["value"]
                row["efficiency"] = round(out["value"] / (base * n), 4) if base else None
            rows.append(row)
    return rows


def table(rows):
    head = ("gpus", "exchange", "wire", "sentences/s", "ms/step", "efficiency", "exposed_tail_ms", "backward_ms", "wire MB", "collectives")
    lines = ["| " + " | ".join(head) + " |", "|" + "---|" * len(head)]
    for r in rows:
        if r["error"]:
            lines.append("| %d | %s | %s | FAILED: %s |" % (r["gpus"], r["algo"], r["grad_dtype"], r["error"]))
            continue
        f = lambda v: "-" if v is None else (("%.3f" % v) if isinstance(v, float) else str(v))
        lines.append("| %d | %s | %s | %.1f | %.3f | %s | %s | %s | %s | %s |" % (
            r["gpus"], r["algo"], r["grad_dtype"], r["value"], r["ms_per_step"], f(r["efficiency"]), f(r["exposed_tail_ms"]),
            f(r["backward_done_ms"]), f(r["wire_mb"]), f(r["collectives"])))
    return "\n".join(lines)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--algos", nargs="+", default=["allreduce", "mesh"], choices=["allreduce", "mesh"])
    ap.add_argument("--grad-dtypes", nargs="+", default=["fp32", "bf16"], choices=["fp32", "bf16"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--master-port", type=int, default=29611)
    ap.add_argument("--timeout", type=int, default=1200)
    ap.add_argument("--bench", default=os.path.join(ROOT, "bench.py"), help="bench script (default: bench.py; the CPU suite passes a gloo stand-in)")
    ap.add_argument("--json", default=None, help="also write the rows as JSON here")
    ap.add_argument("extra", nargs="*", help="passed through to the bench (after --), e.g. -- --no-cpu-baseline --no-fp32-parity")
    a = ap.parse_args(argv)
    extra = a.extra or (["--no-cpu-baseline", "--no-fp32-parity", "--no-glyph256", "--no-forward", "--no-dense-rows-ab"] if a.bench.endswith("bench.py") else [])
    rows = sweep(a.bench, a.gpus, a.algos, a.grad_dtypes, a.backend, a.steps, a.warmup, a.master_port, extra, a.timeout)
    print(table(rows))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)
    return 0 if all(r["error"] is None for r in rows) else 1


if __name__ == "__main__":
    sys.exit(main())
