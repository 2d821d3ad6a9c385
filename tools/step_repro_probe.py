#!/usr/bin/env python3
"""Training-step reproducibility at full size: N times forward + backward of ONE batch from ONE module state (bf16, B = 64, S = 128, full
model, dropout ON - the masks are a hash of (seed, site, element), identical in every repetition), loss and gradients compared with the
first repetition.  Order-fixed kernels (every layer GEMM, LayerNorm, attention, the grouped weight gradients) must give the same bits;
the tensors behind float atomics (embedding tables, BatchNorm / gate sums, the GRU table) may differ in their last bits and are held to
1e-5 of their scale.  Any other difference is a race or an uninitialised read.  N=... KNOBS as tools/perm_probe.py.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, ".")
from realise_amd import _capi  # noqa: E402
from realise_amd.config import RealiseConfig  # noqa: E402
from realise_amd.data import synthetic_batch  # noqa: E402
from realise_amd.modeling import SpellBertPho2ResArch3  # noqa: E402

lib = _capi.load()
for kv in os.environ.get("KNOBS", "").split():
    name, rest = kv.split(":")
    k, v = rest.split("=")
    getattr(lib, "realise_set_" + name)(int(k), int(v))
N = int(os.environ.get("N", "200"))
cfg = RealiseConfig()
m = SpellBertPho2ResArch3(cfg, compute_dtype="bf16", seed=3).to("cuda").train()
batch = synthetic_batch(64, 128, seed=77)
seed0 = m._step_seed                       # (the module draws one dropout seed per forward: rewound before every repetition)
fixed = lambda n: ".layer." in n and n.endswith("weight") and "LayerNorm" not in n      # noqa: E731
ref, bad_fixed, bad_loose, bad_loss = None, {}, {}, 0
for it in range(N):
    m._step_seed = seed0
    m.zero_grad()
    loss, _ = m(batch)
    loss.backward()
    torch.cuda.synchronize()
    cur = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    cur["__loss__"] = loss.detach().clone().reshape(1)
    if ref is None:
        ref = cur
        continue
    if not torch.equal(cur["__loss__"], ref["__loss__"]):
        bad_loss += 1
    for n in cur:
        if n == "__loss__" or torch.equal(cur[n], ref[n]):
            continue
        d = (cur[n].float() - ref[n].float()).abs().max().item()
        s = ref[n].float().abs().max().item()
        if fixed(n):
            bad_fixed.setdefault(n, []).append((it, d, s))
        elif d > 1e-5 * s + 1e-12:
            bad_loose.setdefault(n, []).append((it, d, s))
try:
    m.check_ids()
    flag = "flags clear"
except Exception as e:      # noqa: BLE001
    flag = "FLAG: %s" % type(e).__name__
nfixed = len([n for n in ref if fixed(n)])
print("KNOBS [%s] %s, %d steps (forward + backward, dropout %s): loss differs %d x; order-fixed gradients (%d tensors) differing: %s; atomics-backed gradients beyond 1e-5: %s"
      % (os.environ.get("KNOBS", ""), flag, N, cfg.hidden_dropout_prob, bad_loss, nfixed,
         "none" if not bad_fixed else "; ".join("%s: %d x (max %.3g of %.3g)" % (k, len(v), max(x[1] for x in v), v[0][2]) for k, v in list(bad_fixed.items())[:8]),
         "none" if not bad_loose else "; ".join("%s: %d x (max %.3g of %.3g)" % (k, len(v), max(x[1] for x in v), v[0][2]) for k, v in list(bad_loose.items())[:8])))
