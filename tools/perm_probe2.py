"""the exact sequence of tests/test_engine_gpu.py::test_full_size_config2_properties_bf16 up to the permutation check, under knobs"""
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
res = {}
steps = os.environ.get("SEQ", "1b 0b").split()
for s in steps:
    on, bw = int(s[0]), s.endswith("b")
    lib.realise_set_glyph_dedup(on)
    m.zero_grad()
    loss, logits = m(batch)
    if bw:
        loss.backward()
    torch.cuda.synchronize()
    res[s] = logits.detach().float()
lib.realise_set_glyph_dedup(1)
lg1 = res[steps[0]]
perm = torch.randperm(64, generator=torch.Generator().manual_seed(5))
pb = {k: (v[perm] if torch.is_tensor(v) and v.shape[0] == 64 else v) for k, v in batch.items()}
tok = (perm[:, None] * 128 + torch.arange(128)[None, :]).reshape(-1)
pb["pho_idx"] = batch["pho_idx"][tok]
pb["pho_lens"] = [batch["pho_lens"][i] for i in tok.tolist()]
m.zero_grad()
lp, lgp = m(pb)
m.zero_grad()
l3, lg3 = m(batch)
torch.cuda.synchronize()
d = (lgp.detach().float() - lg1[perm]).abs()
d3 = (lg3.detach().float() - lg1).abs()
print("KNOBS [%s] SEQ [%s]: permuted vs first: max %.4f mean %.5f | same batch again vs first: max %.4f | steps vs first: %s"
      % (os.environ.get("KNOBS", ""), " ".join(steps), d.max().item(), d.mean().item(), d3.max().item(),
         " ".join("%s=%.4f" % (s, (res[s] - lg1).abs().max().item()) for s in steps[1:])))
