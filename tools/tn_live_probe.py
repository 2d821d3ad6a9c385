"""Times the grouped weight-gradient launch of one transformer layer (four problems, P = 8192 reduction rows, bf16) with and without the
live-block lists, cold operands (every repetition works on another operand set, > 1 GB between two uses of the same bytes):
    none          dense reduction, no list
    64/100, 64/84 whole 64-row tiles: all 128 listed / the 84 % a SIGHAN-shaped batch keeps
    16/100, 16/68 16-row blocks packed four to a tile: all 512 listed / the 68 % such a batch keeps
(16/100 against none = what the per-wave block addressing costs; 16/68 against 64/84 = what the finer granularity buys)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from realise_amd import _capi  # noqa: E402

lib = _capi.load()
P, SETS, REPS = 8192, 6, 24
shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
sets = []
for s in range(SETS):
    probs = (_capi.TnProblem * 4)()
    keep = []
    for k, (I, J) in enumerate(shapes):
        a = (torch.randn(P, I, device="cuda") * 0.1).bfloat16()
        b = (torch.randn(P, J, device="cuda") * 0.1).bfloat16()
        o = torch.zeros(I, J, device="cuda")
        cs = torch.zeros(I, device="cuda")
        keep += [a, b, o, cs]
        probs[k].A, probs[k].lda, probs[k].B, probs[k].ldb = a.data_ptr(), I, b.data_ptr(), J
        probs[k].I, probs[k].J, probs[k].out, probs[k].ldo, probs[k].colsum = I, J, o.data_ptr(), J, cs.data_ptr()
    sets.append((probs, keep))


def lists(rows, frac):
    n = P // rows
    g = torch.Generator().manual_seed(rows)
    if frac >= 1.0:
        idx = torch.arange(n)
    else:            # sentences of 128 rows with a live prefix: the blocks of a padded batch, not a random subset
        per = 128 // rows
        live = []
        target = frac * n
        for sent in range(P // 128):
            k = max(1, min(per, int(round(per * frac + (torch.rand(1, generator=g).item() - 0.5) * per * 0.6))))
            live += [sent * per + j for j in range(k)]
        idx = torch.tensor(live)
    return idx.int().cuda(), torch.tensor([idx.numel()], dtype=torch.int32, device="cuda"), idx.numel() * rows / P


def run(name, rows, frac):
    if rows:
        l, n, f = lists(rows, frac)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(REPS + 4):
        probs, _ = sets[r % SETS]
        e0.record()
        if rows:
            rc = lib.realise_gemm_tn_grouped_live(st, _capi.BF16, 4, probs, P, C.c_void_p(l.data_ptr()), C.c_void_p(n.data_ptr()), rows, 1)
        else:
            rc = lib.realise_gemm_tn_grouped(st, _capi.BF16, 4, probs, P)
        e1.record()
        assert rc == 0
        torch.cuda.synchronize()
        if r >= 4:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    live = f if rows else 1.0
    gf = 2.0 * P * sum(i * j for i, j in shapes) * live * 1e-9
    med = ts[len(ts) // 2]
    print("%-8s live rows %5.1f %%  median %7.1f us  min %7.1f  %6.0f TF executed  (%.2f us per 64-row K-tile)"
          % (name, 100 * live, med, ts[0], gf / med * 1e3, med / (P * live / 64)))


for g8 in (0, 1):
    lib.realise_set_engine(7, g8)
    print("--- %s" % ("8-wave 256x128 tiles, one per CU (gemm_tn8_group)" if g8 else "4-wave 128x128 tiles, two per CU (gemm_tn_group)"))
    run("none", 0, 1.0)
    if not g8:
        run("64/100", 64, 1.0)
        run("64/84", 64, 0.84)
    run("16/100", 16, 1.0)
    run("16/68", 16, 0.68)
    run("none", 0, 1.0)
