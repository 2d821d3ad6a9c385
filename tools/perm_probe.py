"""permutation-invariance diagnostic at full size (tests/test_engine_gpu.py::test_full_size_config2_properties_bf16 (2)) under knobs"""
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
cfg = RealiseConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
m = SpellBertPho2ResArch3(cfg, compute_dtype="bf16", seed=3).to("cuda").train()
batch = synthetic_batch(64, 128, seed=77)
with torch.no_grad():
    l1, lg1 = m(batch)
    l1b, lg1b = m(batch)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(5))
    pb = {k: (v[perm] if torch.is_tensor(v) and v.shape[0] == 64 else v) for k, v in batch.items()}
    tok = (perm[:, None] * 128 + torch.arange(128)[None, :]).reshape(-1)
    pb["pho_idx"] = batch["pho_idx"][tok]
    pb["pho_lens"] = [batch["pho_lens"][i] for i in tok.tolist()]
    lp, lgp = m(pb)
torch.cuda.synchronize()
d = (lgp.float() - lg1.float()[perm.cuda()]).abs()
print("KNOBS [%s] run-to-run max %.4f | permuted: loss diff %.2e, logits max diff %.4f, mean %.5f, frac > 0.06: %.2e"
      % (os.environ.get("KNOBS", ""), (lg1.float() - lg1b.float()).abs().max().item(), abs(lp.item() - l1.item()), d.max().item(), d.mean().item(),
         (d > 0.06).float().mean().item()))
