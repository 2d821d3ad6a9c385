"""Shows that the residual glyph-ResNet gradient mismatches vs the oracle are ReLU-boundary flips: an activation whose
pre-ReLU value is ~1e-7 from zero gets a different mask than the oracle's (or than the previous run's)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "oracle")]
import torch
from realise_amd import _capi
from realise_amd.config import RealiseConfig
from realise_amd.init import init_state_dict_numpy
import test_engine_gpu as TE
import realise_ref as RR
lib = _capi.load()
cfg = RealiseConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
sd_np = init_state_dict_numpy(cfg, "arch3", seed=41, scheme="perturbed")
batch = TE._edge_batch("full_length")
sd = TE.oracle_state_dict(sd_np, requires_grad=False)
taps = {}
with torch.no_grad():
    RR.arch3_forward(sd, cfg, batch, training=True, new_buffers={}, taps=taps)
lib.realise_set_glyph_dedup(0)
m = TE.build("arch3", cfg, sd_np, "fp32", train=True)
prev = {}
for rep in range(4):
    m.zero_grad(); loss, logits = m(batch); loss.backward(); torch.cuda.synchronize()
    for k in range(1, 6):
        o = taps["resnet.block%d" % k]                       # [N, C, h, w]
        N, C, h, w = o.shape
        ours = m.tap("resnet.block%d" % k).float().cpu().reshape(N, h, w, C).permute(0, 3, 1, 2)
        flips = ((ours > 0) != (o > 0)).nonzero()
        msg = "rep %d block%d: max|diff| %.2e, mask flips vs oracle %d" % (rep, k, (ours - o).abs().max().item(), len(flips))
        for f in flips[:4]:
            f = tuple(f.tolist()); msg += " | ch %d ours %.2e oracle %.2e" % (f[1], ours[f].item(), o[f].item())
        if k in prev:
            d = ((ours > 0) != (prev[k] > 0)).sum().item()
            msg += " | flips vs previous rep %d, bitwise equal %s" % (d, bool((ours == prev[k]).all()))
        prev[k] = ours
        print(msg)
