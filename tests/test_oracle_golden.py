"""Pin the CPU restatement (oracle/realise_ref.py) to outputs of the upstream
reference captured by oracle/make_golden.py.  CPU only."""
import numpy as np
import pytest
import torch

import realise_ref as R
from helpers import check_summary, golden_case_inputs, load_golden, oracle_state_dict

FWD_ATOL = 2e-5     # fp32 CPU vs fp32 CPU; measured 2-4e-6


@pytest.mark.parametrize("name,model_type", [
    ("spellbert_b2s16_eval", "bert"),
    ("arch3_b2s16_eval", "arch3"),
    ("spellbert_b8s64_eval", "bert"),
])
def test_forward_eval_matches_reference(golden_dir, name, model_type):
    g = load_golden(golden_dir, name)
    cfg, sd_np, batch = golden_case_inputs(g, model_type)
    sd = oracle_state_dict(sd_np)
    taps = {}
    fwd = R.arch3_forward if model_type == "arch3" else R.spellbert_forward
    with torch.no_grad():
        loss, logits = fwd(sd, cfg, batch, training=False, taps=taps)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    check_summary(g, "logits", logits, FWD_ATOL)
    assert np.array_equal(logits.argmax(-1).numpy().astype(np.int32), g["argmax"])   # bit-exact ids
    for k in ("bert_h", "pho_gru", "pho_h", "res", "res_h", "out", "bert.emb", "bert.encoder.layer.0.out",
              "resnet.block1", "resnet.block3", "resnet.block5"):
        if "tap/%s/n" % k in g:
            check_summary(g, "tap/" + k, taps[k], FWD_ATOL, what="tap")


@pytest.mark.parametrize("name,model_type", [
    ("spellbert_b2s16_train", "bert"),
    ("arch3_b2s16_train", "arch3"),
    ("spellbert_b8s64_train", "bert"),          # BASELINE configs[0] at its stated size (B=8, S=64, 12 layers)
])
def test_train_step_grads_match_reference(golden_dir, name, model_type):
    g = load_golden(golden_dir, name)
    cfg, sd_np, batch = golden_case_inputs(g, model_type)
    sd = oracle_state_dict(sd_np, requires_grad=True)
    nb = {}
    if model_type == "arch3":
        loss, logits = R.arch3_forward(sd, cfg, batch, training=True, new_buffers=nb)
    else:
        loss, logits = R.spellbert_forward(sd, cfg, batch, training=True)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    check_summary(g, "logits", logits, FWD_ATOL)
    n_checked = 0
    for key in [k[:-2] for k in g if k.startswith("grad/") and k.endswith("/n")]:
        pname = key[len("grad/"):]
        if pname == "classifier.weight":
            continue
        grad = sd[pname].grad
        assert grad is not None, pname
        scale = float(g[key + "/abssum"]) / int(g[key + "/n"])
        check_summary(g, key, grad, atol=2e-6 + 2e-3 * scale, what="grad")
        n_checked += 1
    assert n_checked > 30
    # the 34.2 M never-used parameters get no gradient in the restatement either
    for key in [k for k in g if k.startswith("gradnone/")]:
        assert sd[key[len("gradnone/"):]].grad is None
    # BatchNorm running statistics after one train-mode forward
    for key in [k[:-2] for k in g if k.startswith("buf/") and k.endswith("/n")]:
        check_summary(g, key, nb[key[len("buf/"):]].to(torch.float64), 1e-5, what="buffer")


def test_gru_all_lengths(golden_dir):
    g = load_golden(golden_dir, "gru_lengths")
    from realise_amd.config import RealiseConfig
    from realise_amd.init import tensor_init, tensor_specs
    cfg = RealiseConfig()
    sd = {n: torch.from_numpy(tensor_init(n, s, k, cfg, seed=3, scheme="perturbed"))
          for n, s, k in tensor_specs(cfg, "arch3") if n.startswith("pho_gru") or n.startswith("pho_emb")}
    h = R.pho_gru_last_hidden(sd, torch.from_numpy(g["pho_idx"]), [int(x) for x in g["pho_lens"]])
    assert np.abs(h.numpy() - g["h_last"]).max() < 1e-6


def test_adamw_and_schedule_known_answers(golden_dir):
    g = load_golden(golden_dir, "adamw_steps")
    # linear warm-up table of transformers/tests/optimization_test.py:93-146
    table = [10.0 * R.linear_schedule_with_warmup(s, 2, 10) for s in range(1, 11)]
    assert np.allclose(table, [5.0, 10.0, 8.75, 7.5, 6.25, 5.0, 3.75, 2.5, 1.25, 0.0])
    assert np.allclose(table, g["sched_table"])
    p = torch.from_numpy(g["p0"].copy())
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for i in range(3):
        lr = 1e-2 * R.linear_schedule_with_warmup(i, 2, 10)
        assert abs(lr - g["lrs"][i]) < 1e-12
        p, m, v = R.adamw_step(p, torch.from_numpy(g["grads"][i]), m, v, i + 1, lr, weight_decay=0.01)
        assert np.abs(p.numpy() - g["traj"][i]).max() < 1e-6


def test_adamw_converges_like_reference_test():
    """transformers/tests/optimization_test.py:67-79: lr 2e-1, 100 steps -> [0.4,0.2,-0.5] +-1e-2."""
    w = torch.tensor([0.1, -0.2, -0.1])
    target = torch.tensor([0.4, 0.2, -0.5])
    m = torch.zeros(3)
    v = torch.zeros(3)
    for step in range(1, 101):
        grad = 2.0 * (w - target) / 3.0          # d/dw MSE
        w, m, v = R.adamw_step(w, grad, m, v, step, 2e-1)
    assert torch.allclose(w, target, atol=1e-2)
