"""Round 6 probe: the wide row-list GEMMs on 256 x 256 one-per-CU tiles (realise_set_nt8p(5, v)) against the shipped 128 x 192 two-per-CU
tiles - a two-layer training step at the bench's batch shape must give the same loss and bit-identical layer gradients (the K order of
an output element does not depend on the tile)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from realise_amd import _capi
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.init import init_state_dict_numpy
from realise_amd.modeling import SpellBertPho2ResArch3

lib = _capi.load()
KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 5      # realise_set_nt8p key under test (5: 256 x 256 row-list tiles, probe build; 7: CU pairing)
VALUES = (1, 2) if KEY == 5 else ((10 + (4 << 4),) if KEY == 8 else (1,))
cfg = RealiseConfig(num_hidden_layers=2, pho_layers=1, out_layers=1)
sd = init_state_dict_numpy(cfg, seed=5)
b = synthetic_batch(64, 128, seed=77)
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}


def run(v):
    lib.realise_set_nt8p(KEY, v)
    m = SpellBertPho2ResArch3(cfg, compute_dtype="bf16")
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(x)) for k, x in sd.items()})
    m.to("cuda"); m.train()
    m.zero_grad()
    loss, _ = m(b)
    loss.backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    lib.realise_set_nt8p(KEY, 0)
    return float(loss.item()), g


l0, g0 = run(0)
for v in VALUES:
    l1, g1 = run(v)
    bad = [n for n in g0 if "encoder.layer" in n and not torch.equal(g0[n], g1[n])]
    print("knob", v, "loss", l0, l1, "layer gradient tensors that differ:", len(bad), bad[:4])
