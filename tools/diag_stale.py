import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "oracle")]
import torch
from realise_amd import _capi
from realise_amd.config import RealiseConfig
from realise_amd.init import init_state_dict_numpy
import test_engine_gpu as TE
lib = _capi.load()
cfg = RealiseConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
sd_np = init_state_dict_numpy(cfg, "arch3", seed=41, scheme="perturbed")
batch = TE._edge_batch("full_length")
sd, nb, oloss, ologits = TE._oracle_train("arch3", cfg, sd_np, batch)
lib.realise_set_glyph_dedup(0)
m = TE.build("arch3", cfg, sd_np, "fp32", train=True)
def worst():
    w = (0, "")
    for pname, p in m.named_parameters():
        og = sd[pname].grad
        if og is None or og.abs().max() < 1e-6: continue
        r = ((p.grad.cpu() - og).abs().max() / og.abs().max()).item()
        if r > w[0]: w = (r, pname)
    return w
def step(zero=None, when="before_bwd"):
    m.zero_grad()
    if zero == "ALL": m._ws.zero_()
    loss, logits = m(batch)
    if zero and zero != "ALL": m.tap("scratch." + zero).zero_()
    loss.backward(); torch.cuda.synchronize()
    return worst()
print("rep0", step()); print("rep1", step()); print("zero ALL before fwd", step("ALL")); print("again", step())
for name in ["tn_slab", "ln_slots", "gA", "gB", "gC", "gE", "gD", "gF", "r_dout", "r_dc2", "r_dcs", "r_dh1", "r_dc1", "r_dx", "bn_sums", "bn_slots", "seg_acc", "X1", "X2", "X3"]:
    print("zero %-9s before bwd ->" % name, step(name))
