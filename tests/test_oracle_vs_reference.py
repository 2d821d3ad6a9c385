"""The CPU restatement against the REFERENCE ITSELF, run live (this container only).

/root/reference does not exist on the GPU box, so everything here skips there; the same comparison is pinned for
travel by the committed vectors (tests/test_oracle_golden.py).  Importing the reference needs the stub modules of
oracle/_ref_import.py.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
from _ref_import import import_reference, reference_available  # noqa: E402
import realise_ref as R  # noqa: E402

from realise_amd.config import RealiseConfig  # noqa: E402
from realise_amd.data import synthetic_batch  # noqa: E402
from realise_amd.init import init_state_dict_numpy  # noqa: E402

pytestmark = pytest.mark.skipif(not reference_available(), reason="the upstream reference tree is not present on this machine")


@pytest.fixture(scope="module")
def ref():
    return import_reference()


def _reference_model(ref, cfg, model_type, sd_np, train):
    models, BertConfig = ref
    bc = BertConfig(vocab_size_or_config_json_file=cfg["vocab_size"])
    for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "hidden_dropout_prob",
              "attention_probs_dropout_prob", "max_position_embeddings", "type_vocab_size", "layer_norm_eps", "initializer_range"):
        setattr(bc, k, cfg[k])
    bc.image_model_type, bc.num_fonts = 0, cfg["num_fonts"]
    m = (models.SpellBertPho2ResArch3 if model_type == "arch3" else models.SpellBert)(bc)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}, strict=True)
    m.tie_cls_weight()
    return m.train(train)


def _oracle_sd(sd_np, grad):
    sd = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.array(v, copy=True))
        if grad and t.dtype == torch.float32 and k not in ("char_images_multifonts", "char_images.weight") and "running_" not in k:
            t.requires_grad_(True)
        sd[k] = t
    sd["classifier.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    return sd


@pytest.mark.parametrize("model_type,B,S", [("bert", 2, 12), ("arch3", 2, 12), ("arch3", 1, 33)])
def test_train_forward_backward_matches_the_reference(ref, model_type, B, S):
    """loss, logits, arg-max ids, every parameter gradient and the updated BatchNorm buffers (train mode, dropout 0)."""
    torch.manual_seed(0)
    cfg = RealiseConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, model_type, seed=100 + S, scheme="perturbed")
    batch = synthetic_batch(B, S, seed=S, with_pho=(model_type == "arch3"))
    m = _reference_model(ref, cfg, model_type, sd_np, train=True)
    loss, logits = m(batch)[:2]
    loss.backward()
    osd, nb = _oracle_sd(sd_np, grad=True), {}
    if model_type == "arch3":
        oloss, ologits = R.arch3_forward(osd, cfg, batch, training=True, new_buffers=nb)
    else:
        oloss, ologits = R.spellbert_forward(osd, cfg, batch, training=True)
    oloss.backward()
    assert abs(loss.item() - oloss.item()) < 1e-5
    assert (logits - ologits).abs().max().item() < 1e-4
    assert torch.equal(logits.argmax(-1), ologits.argmax(-1))
    for k, p in m.named_parameters():
        og = osd[k].grad
        if p.grad is None:                                     # poolers, unused word tables: no gradient on either side
            assert og is None or float(og.abs().max()) == 0.0, k
            continue
        assert og is not None, k
        assert (og - p.grad).abs().max().item() <= 1e-6 + 2e-4 * p.grad.abs().max().item(), k
    msd = m.state_dict()
    for k, v in nb.items():
        assert (msd[k].double() - v.double()).abs().max().item() < 1e-5, k


@pytest.mark.parametrize("num_fonts", [1, 2])
def test_one_and_two_font_models_match_the_reference(ref, num_fonts):
    """models.py:674-679, 831-834: one font keeps the glyph table as an nn.Embedding [V, 1024] reshaped to [n, 1, 32, 32]; the ResNet's
    first convolutions take num_fonts channels.  Pins the oracle the device tests of these configurations compare against."""
    cfg = RealiseConfig(num_hidden_layers=1, num_fonts=num_fonts, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=60 + num_fonts, scheme="perturbed")
    batch = synthetic_batch(3, 24, seed=61)
    m = _reference_model(ref, cfg, "arch3", sd_np, train=True)
    loss, logits = m(batch)[:2]
    loss.backward()
    osd, nb = _oracle_sd(sd_np, grad=True), {}
    oloss, ologits = R.arch3_forward(osd, cfg, batch, training=True, new_buffers=nb)
    oloss.backward()
    assert abs(loss.item() - oloss.item()) < 1e-5 and (logits - ologits).abs().max().item() < 1e-4
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert (osd[k].grad - p.grad).abs().max().item() <= 1e-6 + 2e-4 * p.grad.abs().max().item(), k
    for k, v in nb.items():
        assert (m.state_dict()[k].double() - v.double()).abs().max().item() < 1e-5, k


def test_eval_forward_uses_running_statistics_like_the_reference(ref):
    cfg = RealiseConfig(num_hidden_layers=1)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=5, scheme="perturbed")
    batch = synthetic_batch(2, 10, seed=5)
    m = _reference_model(ref, cfg, "arch3", sd_np, train=False)
    with torch.no_grad():
        loss, logits = m(batch)[:2]
        oloss, ologits = R.arch3_forward(_oracle_sd(sd_np, grad=False), cfg, batch, training=False)
    assert abs(loss.item() - oloss.item()) < 1e-5 and (logits - ologits).abs().max().item() < 1e-4


def test_optimizer_and_schedule_match_the_reference(ref):
    """three AdamW steps + the linear warm-up schedule against the vendored transformers/optimization.py"""
    from transformers import AdamW, get_linear_schedule_with_warmup
    g = torch.Generator().manual_seed(3)
    p0, grads = torch.randn(257, generator=g), [torch.randn(257, generator=g) for _ in range(3)]
    p = torch.nn.Parameter(p0.clone())
    opt = AdamW([p], lr=2e-3, eps=1e-8, weight_decay=0.01)
    sched = get_linear_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=10)
    q, mm, vv = p0.clone(), torch.zeros(257), torch.zeros(257)
    for step, gr in enumerate(grads, start=1):
        lr = 2e-3 * R.linear_schedule_with_warmup(step - 1, 2, 10)
        p.grad = gr.clone()
        opt.step()
        sched.step()
        q, mm, vv = R.adamw_step(q, gr, mm, vv, step, lr, eps=1e-8, weight_decay=0.01)
        assert (p.detach() - q).abs().max().item() < 1e-6
