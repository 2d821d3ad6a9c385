"""GPU diagnostic: which tensors differ between same-seed train steps (atomic-order noise vs real races)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.modeling import SpellBertPho2ResArch3

for pdrop in (0.1, 0.0):
    cfg = RealiseConfig(num_hidden_layers=2, hidden_dropout_prob=pdrop, attention_probs_dropout_prob=pdrop)
    batch = synthetic_batch(4, 32, seed=9)
    m = SpellBertPho2ResArch3(cfg, compute_dtype="fp32", seed=5).to("cuda")
    m.train()
    runs = []
    for r in range(3):
        m.zero_grad()
        m._step_seed = 100
        loss, logits = m(batch)
        taps = {k: m.tap(k).float().clone() for k in ("bert.layer.1.out", "pho_gru", "pho_model.layer.3.out", "res_h", "fused",
                                                      "output_block.layer.2.out", "resnet.block1", "resnet.block5")}
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.item(), logits.float().clone(), m.flat_gradients().clone(), taps))
    print("== dropout", pdrop, "losses", [r[0] for r in runs])
    for a, b in ((0, 1), (1, 2)):
        print(" logits maxdiff run%d-run%d: %.3e" % (a, b, (runs[a][1] - runs[b][1]).abs().max().item()))
        for k in runs[a][3]:
            print("   tap %-28s maxdiff %.3e" % (k, (runs[a][3][k] - runs[b][3][k]).abs().max().item()))
        rows = []
        for name, (arena, off, shape, p) in m._views.items():
            if arena != 0 or p is None:
                continue
            n = p.numel()
            ga, gb = runs[a][2][off:off + n], runs[b][2][off:off + n]
            d = (ga - gb).norm().item()
            rows.append((d / (ga.norm().item() + 1e-30), d, name))
        rows.sort(reverse=True)
        for r in rows[:12]:
            print("   grad rel %.3e abs %.3e %s" % r)
