"""Round 6 probe: where does a pipelined optimizer sweep differ from the plain one?  realise_set_engine(15, v): 1 = pipelined, 2 = every
reader waits for the whole sweep, 3 = the sliced launches on the caller's stream."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from realise_amd import _capi
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.init import init_state_dict_numpy
from realise_amd.modeling import SpellBertPho2ResArch3
from realise_amd.optim import FusedAdamW

lib = _capi.load()
cfg = RealiseConfig(num_hidden_layers=5, pho_layers=1, out_layers=1)
sd = init_state_dict_numpy(cfg, seed=71)
batches = []
for k in range(3):
    b = synthetic_batch(8, 64, seed=600 + k)
    batches.append({kk: (v.cuda() if torch.is_tensor(v) else v) for kk, v in b.items()})


def run(pipe, knob):
    lib.realise_set_engine(15, knob)
    m = SpellBertPho2ResArch3(cfg, compute_dtype="bf16")
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(x)) for k, x in sd.items()})
    m.to("cuda"); m.train()
    m.trust_fused_optimizer = True
    m.pipeline_optimizer = pipe
    opt = FusedAdamW(m, [{"params": [p for p in m.parameters() if p.requires_grad], "weight_decay": 0.01}], lr=3e-4, eps=1e-8, max_grad_norm=1.0)
    losses, snaps = [], []
    for b in batches:
        m.zero_grad()
        loss = m(b)[0]
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        losses.append(float(loss.item()))
        snaps.append({n: p.detach().clone() for n, p in m.named_parameters()})
    lib.realise_set_engine(15, 1)
    return losses, snaps


base_l, base_s = run(False, 1)
again_l, again_s = run(False, 1)
print("plain twice:", base_l == again_l, sum(not torch.equal(base_s[0][n], again_s[0][n]) for n in base_s[0]), "tensors differ after step 1")
for knob in (1, 2, 3):
    l, s = run(True, knob)
    bad = [n for n in base_s[0] if not torch.equal(base_s[0][n], s[0][n])]
    print("knob", knob, "losses", base_l, l)
    print("   parameters that differ after step 1:", len(bad), [(n, float((base_s[0][n] - s[0][n]).abs().max())) for n in bad[:6]])
