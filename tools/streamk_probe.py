#!/usr/bin/env python3
"""Stream-K layer GEMM (realise_gemm_nt_streamk, gemm_nt8s.hip) against the 128 x 192 live / dense kernels on the shapes of a training
step: results (same operands, same dropout masks), dead rows untouched, run-to-run bits, and time per launch warm (back to back) and
cold (a 512 MiB fill between launches).  GPU box: python tools/streamk_probe.py [check|time|all]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from realise_amd import _capi  # noqa: E402

lib = _capi.load()
PART_BYTES = 256 * 24 * 512 * 16


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def epilogue(mode, out, N, accumulate=0, out2=None, bias=None, aux=None, drop=0.0, seed=1234):
    ep = _capi.Epilogue()
    ep.mode, ep.accumulate, ep.out, ep.ldo, ep.alpha, ep.drop_scale = mode, accumulate, out.data_ptr(), N, 1.0, 1.0
    if out2 is not None:
        ep.out2 = out2.data_ptr()
    if bias is not None:
        ep.bias = bias.data_ptr()
    if aux is not None:
        ep.aux, ep.ldaux = aux.data_ptr(), N
    if drop > 0.0:
        ep.drop_seed, ep.drop_thresh, ep.drop_scale = seed, int(drop * 4294967296.0), 1.0 / (1.0 - drop)
    return ep


class Ctx:
    def __init__(self):
        self.part = torch.empty(PART_BYTES // 4, dtype=torch.float32, device="cuda")
        self.flags = torch.zeros(256 * 64 + 64, dtype=torch.int32, device="cuda")
        self.timeout = torch.zeros(4, dtype=torch.int32, device="cuda")
        self.tag = 0

    def next_tag(self):
        self.tag += 1
        return self.tag


def live_blocks(kind, nb, rng):
    if kind == "bench":          # sentence lengths as the synthetic batch: 8 blocks per sentence (S = 128), a prefix of each live
        return np.concatenate([np.arange(s * 8, s * 8 + int(rng.integers(3, 9))) for s in range(nb // 8)])
    if kind == "ragged":
        return np.concatenate([np.arange(s * 8, s * 8 + int(rng.integers(1, 9))) for s in range(nb // 8)])
    if kind == "one":
        return np.array([min(37, nb - 1)])
    if kind == "all":
        return np.arange(nb)
    if kind == "odd":
        return np.arange(1, nb, 2)
    raise ValueError(kind)


def make(M, N, K, mode, seed):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(M, K, generator=g) * 0.1).bfloat16().cuda()
    b = (torch.randn(N, K, generator=g) * 0.1).bfloat16().cuda()
    bias = (torch.randn(N, generator=g) * 0.1).float().cuda()
    aux = torch.randn(M, N, generator=g).bfloat16().cuda() if mode in (2, 4) else None
    old = torch.randn(M, N, generator=g).bfloat16().cuda()
    return a, b, bias, aux, old


def launch(ctx, which, a, b, M, N, K, ep, lst, cnt):
    if which == "sk":
        rc = lib.realise_gemm_nt_streamk(stream(), P(a), K, P(b), K, M, N, K, C.byref(ep), P(lst), P(cnt), P(ctx.part), P(ctx.flags), ctx.next_tag(), P(ctx.timeout))
    elif lst is not None:
        rc = lib.realise_gemm_nt_live(stream(), P(a), K, P(b), K, M, N, K, C.byref(ep), P(lst), P(cnt))
    else:
        rc = lib.realise_gemm_nt(stream(), _capi.BF16, P(a), K, P(b), K, M, N, K, C.byref(ep))
    if rc != 0:
        raise RuntimeError("%s launch failed: %d" % (which, rc))


def check(ctx, M, N, K, mode, accumulate, kind, label):
    rng = np.random.default_rng(M + N + K + mode)
    nb = M // 16
    a, b, bias, aux, old = make(M, N, K, mode, M * 7 + N + K + mode)
    drop = 0.1 if mode == 2 else 0.0
    if kind == "dense":
        live, lst, cnt = np.arange(nb), None, None
    else:
        live = live_blocks(kind, nb, rng)
        lst = torch.full((nb + 8,), -7, dtype=torch.int32, device="cuda")
        lst[:len(live)] = torch.from_numpy(live.astype(np.int32)).cuda()
        cnt = torch.tensor([len(live)], dtype=torch.int32, device="cuda")
    rows = torch.from_numpy((live[:, None] * 16 + np.arange(16)[None, :]).reshape(-1)).long().cuda()
    dead = torch.ones(M, dtype=torch.bool, device="cuda")
    dead[rows] = False

    def run(which):
        out = old.clone()
        out2 = old.clone() if mode == 1 else None
        x = a.clone()
        if lst is not None:
            x[dead] = float("nan")
        ep = epilogue(mode, out, N, accumulate, out2, bias if mode != 4 else None, aux, drop)
        launch(ctx, which, x, b, M, N, K, ep, lst, cnt)
        torch.cuda.synchronize()
        return out, out2

    o_sk, o2_sk = run("sk")
    o_sk_b, _ = run("sk")
    o_ref, o2_ref = run("ref")
    ok = True
    msgs = []
    if int(ctx.timeout[0].item()) != 0:
        ok = False; msgs.append("TIMEOUT flag set")
        ctx.timeout.zero_()
    if not torch.isfinite(o_sk.float()).all():
        ok = False; msgs.append("non-finite outputs")
    if not torch.equal(o_sk, o_sk_b):
        ok = False; msgs.append("two launches differ: %d elements" % int((o_sk != o_sk_b).sum().item()))
    if not torch.equal(o_sk[dead], old[dead]):
        ok = False; msgs.append("dead rows written: %d elements" % int((o_sk[dead] != old[dead]).sum().item()))
    d = (o_sk[rows].float() - o_ref[rows].float()).abs()
    scale = o_ref[rows].float().abs().max().item() + 1e-9
    nd = int((o_sk[rows] != o_ref[rows]).sum().item())
    worst = d.max().item() if d.numel() else 0.0
    # a cut tile sums its K range in two or three fp32 chains: the bf16 results may differ by one rounding step
    rel = (d / (o_ref[rows].float().abs() + 1e-3 * scale)).max().item() if d.numel() else 0.0
    if rel > 2.0 ** -6:
        ok = False; msgs.append("results differ beyond a bf16 rounding step")
    if mode == 1:
        d2 = (o2_sk[rows].float() - o2_ref[rows].float()).abs()
        rel2 = (d2 / (o2_ref[rows].float().abs() + 1e-3 * scale)).max().item() if d2.numel() else 0.0
        if rel2 > 2.0 ** -6 or not torch.equal(o2_sk[dead], old[dead]):
            ok = False; msgs.append("pre-activation output differs / dead rows written")
    print("%-4s %-34s M %5d N %5d K %5d mode %d acc %d %-6s live %4d/%4d | differing elements %8d of %9d, worst |d| %.3g (scale %.3g), worst rel %.3g %s"
          % ("ok" if ok else "FAIL", label, M, N, K, mode, accumulate, kind, len(live), nb, nd, rows.numel() * N, worst, scale, rel, "; ".join(msgs)), flush=True)
    return ok


def timeit(ctx, M, N, K, mode, accumulate, kind, label, reps=30):
    rng = np.random.default_rng(5)
    nb = M // 16
    a, b, bias, aux, old = make(M, N, K, mode, 11)
    drop = 0.1 if mode == 2 else 0.0
    if kind == "dense":
        live, lst, cnt = np.arange(nb), None, None
    else:
        live = live_blocks(kind, nb, rng)
        lst = torch.full((nb + 8,), -7, dtype=torch.int32, device="cuda")
        lst[:len(live)] = torch.from_numpy(live.astype(np.int32)).cuda()
        cnt = torch.tensor([len(live)], dtype=torch.int32, device="cuda")
    out = old.clone()
    out2 = old.clone() if mode == 1 else None
    ep = epilogue(mode, out, N, accumulate, out2, bias if mode != 4 else None, aux, drop)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    res = {}
    for which in ("ref", "sk"):
        for cold in (False, True):
            for _ in range(3):
                launch(ctx, which, a, b, M, N, K, ep, lst, cnt)
            torch.cuda.synchronize()
            tot = 0.0
            n = reps if not cold else 10
            if not cold:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    launch(ctx, which, a, b, M, N, K, ep, lst, cnt)
                e1.record(); torch.cuda.synchronize()
                tot = e0.elapsed_time(e1) * 1e3 / n
            else:
                for _ in range(n):
                    flush.fill_(1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch(ctx, which, a, b, M, N, K, ep, lst, cnt)
                    e1.record(); torch.cuda.synchronize()
                    tot += e0.elapsed_time(e1) * 1e3
                tot /= n
            res[(which, cold)] = tot
    gf = 2.0 * len(live) * 16 * N * K / 1e9
    print("%-34s N %5d K %5d mode %d acc %d %-6s live %4d/%4d | 128x192: warm %6.1f us (%4.0f TF) cold %6.1f | stream-K: warm %6.1f us (%4.0f TF) cold %6.1f | warm x%.2f cold x%.2f"
          % (label, N, K, mode, accumulate, kind, len(live), nb, res[("ref", False)], gf / res[("ref", False)] * 1e3, res[("ref", True)],
             res[("sk", False)], gf / res[("sk", False)] * 1e3, res[("sk", True)], res[("ref", False)] / res[("sk", False)], res[("ref", True)] / res[("sk", True)]), flush=True)


LAYER = [
    (2304, 768, 0, 0, "qkv"),
    (3072, 768, 1, 0, "FFN-up + GELU"),
    (768, 3072, 2, 0, "FFN-down + dropout + residual"),
    (768, 768, 2, 0, "attn-out + dropout + residual"),
    (3072, 768, 4, 0, "GELU' data gradient"),
    (768, 3072, 0, 1, "FFN-up data gradient (acc)"),
    (768, 768, 0, 0, "attn-out data gradient"),
    (768, 2304, 0, 1, "qkv data gradient (acc)"),
]

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ctx = Ctx()
    bad = 0
    if what in ("check", "all"):
        for N, K, mode, acc, label in LAYER:
            for kind in ("bench", "dense"):
                bad += 0 if check(ctx, 8192, N, K, mode, acc, kind, label) else 1
        for M, N, K, mode, acc, kind in [(1024, 768, 768, 2, 0, "one"), (2048, 768, 768, 0, 0, "all"), (256, 192, 128, 0, 0, "odd"), (4096, 2304, 768, 0, 0, "ragged"),
                                         (8192, 768, 3072, 2, 0, "ragged"), (512, 200, 256, 0, 1, "dense"), (1008, 776, 128, 0, 0, "dense"), (8192, 3072, 768, 1, 0, "one")]:
            bad += 0 if check(ctx, M, N, K, mode, acc, kind, "edge") else 1
        print("TOTAL failures: %d" % bad, flush=True)
    if what in ("time", "all"):
        for N, K, mode, acc, label in LAYER:
            for kind in ("bench", "dense"):
                timeit(ctx, 8192, N, K, mode, acc, kind, label)
    sys.exit(1 if bad else 0)
