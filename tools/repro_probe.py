"""forward reproducibility at full size: N training-mode forwards of one batch (no dropout), each compared bit for bit with the first.
Every forward kernel is order-fixed, so ANY difference is a race / uninitialised read.  KNOBS as in tools/perm_probe.py.
GRAD=1: the forwards run with gradients enabled - the form a training step calls, i.e. the live-row path (realise_set_engine(10, .))."""
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
    getattr(lib, "realise_set_" + name)(*([int(k), int(v)] if name != "attn_probe" else [int(v)]))
N = int(os.environ.get("N", "40"))
cfg = RealiseConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
m = SpellBertPho2ResArch3(cfg, compute_dtype="bf16", seed=3).to("cuda").train()
batch = synthetic_batch(64, 128, seed=77)
taps = ["bert.emb", "bert.layer.0.qkv", "bert.layer.0.ctx", "bert.layer.0.attn_out", "bert.layer.0.inter", "bert.layer.0.out", "bert.layer.11.out", "pho_gru", "pho_model.layer.3.out", "res_h", "fused", "output_block.layer.2.out"]
ref, bad = None, {}
with torch.set_grad_enabled(os.environ.get("GRAD", "0") == "1"):
    for it in range(N):
        loss, logits = m(batch)
        torch.cuda.synchronize()
        cur = {"logits": logits.detach().clone()}
        for t in taps:
            cur[t] = m.tap(t).clone()
        if ref is None:
            ref = cur
            continue
        for k in cur:
            if not torch.equal(cur[k], ref[k]):
                bad.setdefault(k, []).append((it, (cur[k].float() - ref[k].float()).abs().max().item()))
                if k in ("bert.layer.0.attn_out", "bert.layer.0.out") and len(bad[k]) <= 2:
                    d = (cur[k].float() - ref[k].float()).reshape(-1, 768) != 0
                    rows = torch.nonzero(d.any(1)).reshape(-1).tolist()
                    cols = torch.nonzero(d.any(0)).reshape(-1).tolist()
                    print("  forward %d, %s: %d differing elements in %d rows %s, %d cols (%s .. %s); per row: %s"
                          % (it, k, int(d.sum()), len(rows), rows[:12], len(cols), cols[:1], cols[-1:], d.sum(1)[rows[:12]].tolist()))
try:
    m.check_ids()
    flag = "flags clear"
except Exception as e:      # noqa: BLE001
    flag = "FLAG: %s" % type(e).__name__
print("KNOBS [%s] GRAD=%s %s, %d forwards: %s" % (os.environ.get("KNOBS", ""), os.environ.get("GRAD", "0"), flag, N,
                                      "all identical" if not bad else "; ".join("%s: %d x (max %.4f)" % (k, len(v), max(x[1] for x in v)) for k, v in bad.items())))
