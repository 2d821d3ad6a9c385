"""Whole-model parity of the HIP engine (through the nn.Module drop-in and the C ABI) against

  (a) the committed golden vectors generated from the upstream reference (tests/golden/*.npz), and
  (b) the CPU oracle (oracle/realise_ref.py) run on the same seeded weights / batch.

Bars (BASELINE.json north_star): fp32 mode - argmax token ids bit-exact, logits within 1e-3;
bf16 mode - logits within the bf16 error band measured for the reference itself under bf16
autocast (2.5e-2 max), argmax equal wherever the reference's own top-1/top-2
margin exceeds twice that band.
"""
import numpy as np
import pytest
import torch

import realise_ref as R
from helpers import check_summary, golden_case_inputs, load_golden, oracle_state_dict
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.modeling import SpellBert, SpellBertPho2ResArch3

pytestmark = pytest.mark.gpu

FP32_LOGIT_TOL = 1e-3
BF16_LOGIT_TOL = 4e-2       # the reference itself under bf16 autocast: 2.5e-2 max


def build(model_type, cfg, sd_np, dtype, train=False):
    cls = SpellBertPho2ResArch3 if model_type == "arch3" else SpellBert
    m = cls(cfg, compute_dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()})
    m.to("cuda")
    m.train(train)
    return m


@pytest.mark.parametrize("name,model_type", [("spellbert_b2s16_eval", "bert"), ("arch3_b2s16_eval", "arch3"),
                                             ("spellbert_b8s64_eval", "bert"), ("arch3_b4s128_eval", "arch3")])
def test_eval_forward_fp32_matches_reference_golden(golden_dir, name, model_type):
    g = load_golden(golden_dir, name)
    cfg, sd_np, batch = golden_case_inputs(g, model_type)
    m = build(model_type, cfg, sd_np, "fp32")
    with torch.no_grad():
        loss, logits = m(batch)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    check_summary(g, "logits", logits.float(), FP32_LOGIT_TOL)
    assert np.array_equal(logits.argmax(-1).cpu().numpy().astype(np.int32), g["argmax"])      # bit-exact ids
    B, S = int(g["meta/B"]), int(g["meta/S"])
    if "tap/bert_h/n" in g:
        last = "bert.layer.%d.out" % (int(g["meta/n_layers"]) - 1)
        check_summary(g, "tap/bert_h", m.tap(last)[:B * S * 768].float(), 1e-4, what="tap")
    if model_type == "arch3":
        check_summary(g, "tap/pho_gru", m.tap("pho_gru").float(), 1e-4, what="tap")
        check_summary(g, "tap/res_h", m.tap("res_h").float(), 2e-4, what="tap")
        check_summary(g, "tap/out", m.tap("output_block.layer.2.out").float(), 2e-4, what="tap")


@pytest.mark.parametrize("name,model_type", [("arch3_b2s16_eval", "arch3"), ("arch3_b4s128_eval", "arch3"),
                                             ("spellbert_b8s64_eval", "bert")])
def test_eval_forward_bf16_within_band(golden_dir, name, model_type):
    g = load_golden(golden_dir, name)
    cfg, sd_np, batch = golden_case_inputs(g, model_type)
    m = build(model_type, cfg, sd_np, "bf16")
    with torch.no_grad():
        loss, logits = m(batch)
    assert logits.dtype == torch.float32          # eval mode: the reference's fp32 logits (logits_dtype='auto')
    check_summary(g, "logits", logits.float(), BF16_LOGIT_TOL)
    assert abs(loss.item() - float(g["loss"])) < 5e-2
    am = logits.float().argmax(-1).cpu().numpy().astype(np.int32)
    decided = g["margin"] > 2 * BF16_LOGIT_TOL
    real = batch["masks"].numpy() == 1
    assert np.array_equal(am[decided & real], g["argmax"][decided & real])
    agree = (am == g["argmax"])[real].mean()
    assert agree > 0.92, agree


def _oracle_train(model_type, cfg, sd_np, batch, taps=None):
    sd = oracle_state_dict(sd_np, requires_grad=True)
    nb = {}
    if model_type == "arch3":
        loss, logits = R.arch3_forward(sd, cfg, batch, training=True, new_buffers=nb, taps=taps)
    else:
        loss, logits = R.spellbert_forward(sd, cfg, batch, training=True)
    loss.backward()
    return sd, nb, loss, logits


def _relu_boundary_flips(m, taps):
    """The glyph ResNet has 10 ReLUs over ~10^7 activations per batch; one whose pre-activation is within rounding
    (~1e-6) of zero can land on the other side than in the oracle, which legitimately selects a different subgradient
    for everything upstream of it (tests/diag_relu_flip.py shows such a case).  Returns the deepest block (1..5) with a
    mask disagreement (0 = masks identical) after checking that every disagreement IS such a boundary case."""
    deepest = 0
    T_ = taps["resnet.block1"].shape[0]
    inv = m.tap("glyph.inv").view(torch.int32)[:T_].cpu().long()          # token -> slot of its distinct glyph
    for k in range(1, 6):
        for site in ("resnet.block%d" % k, "resnet.block%d.h1" % k):
            o = taps[site].detach()
            N, C, h, w = o.shape
            ours = m.tap(site).float().cpu().reshape(N, h, w, C)[inv].permute(0, 3, 1, 2)
            assert (ours - o).abs().max().item() < 5e-3, site          # tiny batches make BatchNorm ill-conditioned
            flip = (ours > 0) != (o > 0)
            if flip.any():
                assert flip.sum().item() <= 4 and max(ours[flip].abs().max().item(), o[flip].abs().max().item()) < 2e-5, site
                deepest = k
    return deepest


def _assert_grads_match_oracle(m, sd, flip_block, rtol, tag=""):
    bad = []
    for pname, p in m.named_parameters():
        og = sd[pname].grad
        if og is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, pname
            continue
        assert p.grad is not None, pname
        mine = p.grad.cpu()
        scale = og.abs().max().item()
        if pname.startswith("resnet.res_block") and int(pname[len("resnet.res_block")]) <= flip_block:
            # upstream of a boundary flip: only a few rows/channels may differ, bounded in the L2 sense
            assert (mine - og).norm().item() <= 0.25 * og.norm().item() + 1e-6, (tag, pname)
            continue
        d = (mine - og).abs().max().item()
        if d > 1e-6 + rtol * scale:
            bad.append((d / (scale + 1e-9), d, scale, pname))
    bad.sort(reverse=True)
    assert not bad, "gradient mismatch vs oracle %s: %s" % (tag, bad[:8])


@pytest.mark.parametrize("name,model_type", [("spellbert_b2s16_train", "bert"), ("arch3_b2s16_train", "arch3"),
                                             ("arch3_b3s40_train", "arch3"), ("spellbert_b8s64_train", "bert")])
def test_train_step_fp32_grads_match_oracle_and_golden(golden_dir, name, model_type):
    g = load_golden(golden_dir, name)
    cfg, sd_np, batch = golden_case_inputs(g, model_type)
    m = build(model_type, cfg, sd_np, "fp32", train=True)
    loss, logits = m(batch)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    check_summary(g, "logits", logits.float(), FP32_LOGIT_TOL)
    taps = {}
    sd, nb, oloss, _ = _oracle_train(model_type, cfg, sd_np, batch, taps)
    flip_block = _relu_boundary_flips(m, taps) if model_type == "arch3" else 0
    _assert_grads_match_oracle(m, sd, flip_block, 2e-3, name)
    for pname, p in m.named_parameters():
        gk = "grad/" + pname
        if gk + "/n" in g and not (pname.startswith("resnet.res_block") and int(pname[len("resnet.res_block")]) <= flip_block):
            check_summary(g, gk, p.grad, atol=2e-6 + 5e-3 * float(g[gk + "/abssum"]) / int(g[gk + "/n"]), what="grad(golden)")
    if model_type == "arch3":
        for k, v in nb.items():
            mine = m.state_dict()[k].cpu().double()
            assert (mine - v.double()).abs().max().item() < 1e-4, k


def test_train_step_bf16_grads_close_to_oracle(golden_dir):
    g = load_golden(golden_dir, "arch3_b2s16_train")
    cfg, sd_np, batch = golden_case_inputs(g, "arch3")
    m = build("arch3", cfg, sd_np, "bf16", train=True)
    loss, logits = m(batch)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 5e-2
    sd, nb, oloss, _ = _oracle_train("arch3", cfg, sd_np, batch)
    cos = []
    for pname, p in m.named_parameters():
        og = sd[pname].grad
        if og is None or og.numel() < 64 or og.abs().max() < 1e-7:
            continue
        a, b = p.grad.cpu().double().reshape(-1), og.double().reshape(-1)
        c = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        cos.append((c, pname))
    cos.sort()
    # direction of every gradient tensor agrees with fp32: 0.99 for the transformer / GRU / gate tensors; the glyph ResNet's
    # BatchNorm at this tiny batch (2 x 16 tokens) amplifies bf16 rounding, measured 0.972 at worst
    worst_other = min([c for c, n in cos if not n.startswith("resnet.")] or [1.0])
    assert worst_other > 0.99, [x for x in cos if not x[1].startswith("resnet.")][:8]
    assert cos[0][0] > 0.96, cos[:8]


def test_gradient_accumulation_and_zero_grad(golden_dir):
    g = load_golden(golden_dir, "spellbert_b2s16_train")
    cfg, sd_np, batch = golden_case_inputs(g, "bert")
    m = build("bert", cfg, sd_np, "fp32", train=True)
    m(batch)[0].backward()
    g1 = m.flat_gradients().clone()
    m(batch)[0].backward()
    assert torch.allclose(m.flat_gradients(), 2 * g1, rtol=1e-4, atol=1e-7)      # engine accumulates like autograd
    m.zero_grad()
    assert float(m.flat_gradients().abs().max()) == 0.0
    (m(batch)[0] * 0.5).backward()
    assert torch.allclose(m.flat_gradients(), 0.5 * g1, rtol=1e-4, atol=1e-7)    # upstream gradient is honoured


def test_dropout_training_mode_is_seeded_and_unbiased():
    cfg = RealiseConfig(num_hidden_layers=2)             # p = 0.1 everywhere, like train.sh
    from realise_amd.data import synthetic_batch
    batch = synthetic_batch(4, 32, seed=9)
    m = SpellBertPho2ResArch3(cfg, compute_dtype="fp32", seed=5).to("cuda")
    m.train()
    m._step_seed = 100
    l1 = m(batch)[0]
    l1.backward()
    g1 = m.flat_gradients().clone()
    assert torch.isfinite(l1) and torch.isfinite(g1).all()
    m.zero_grad()
    m._step_seed = 100
    l2 = m(batch)[0]
    l2.backward()
    assert l1.item() == l2.item() or abs(l1.item() - l2.item()) < 1e-5   # same seed -> same masks (atomics reorder sums)
    # identical masks => identical gradients, up to the summation order of the fp32 atomics left in the backward
    # (LayerNorm / embedding / segment sums); the forward, BatchNorm statistics included, is bitwise reproducible.
    g2 = m.flat_gradients()
    for name, (arena, off, shape, p) in m._views.items():
        if arena != 0 or p is None or "key.bias" in name:
            continue
        n = p.numel()
        a, b = g1[off:off + n], g2[off:off + n]
        rel = ((a - b).norm() / (a.norm() + 1e-12)).item()
        assert rel < (1e-3 if name.startswith("resnet") else 1e-4), (name, rel)
    l3 = m(batch)[0]
    assert l3.item() != l1.item()                                         # next step -> new masks
    m.eval()
    with torch.no_grad():
        le = m(batch)[0]
    assert torch.isfinite(le)            # eval uses BatchNorm running stats (3 momentum-0.1 updates so far): value differs


def test_bert_only_config1_shapes_and_contract():
    """BASELINE config 1: SpellBert, seq_len 64, batch 8 - forward contract of src/models.py:50-73"""
    from realise_amd.data import synthetic_batch
    cfg = RealiseConfig()
    m = SpellBert(cfg, compute_dtype="bf16").to("cuda").eval()
    batch = synthetic_batch(8, 64, seed=1, with_pho=False)
    with torch.no_grad():
        out = m(batch)
        assert len(out) == 2 and out[0].shape == () and out[1].shape == (8, 64, 21128)
        del batch["tgt_idx"]
        out = m(batch)
        assert len(out) == 1 and out[0].shape == (8, 64, 21128)


def test_glyph_dedup_bookkeeping():
    """the ResNet runs once per distinct token id: slots in order of first occurrence, multiplicities, inverse map"""
    from realise_amd.data import synthetic_batch
    cfg = RealiseConfig(num_hidden_layers=1)
    batch = synthetic_batch(8, 64, seed=21)
    m = SpellBertPho2ResArch3(cfg, compute_dtype="bf16", seed=2).to("cuda").eval()
    with torch.no_grad():
        m(batch)
    ids = batch["src_idx"].reshape(-1)
    T_ = ids.numel()
    bounds = m.tap("glyph.bounds").view(torch.int32).cpu()
    inv = m.tap("glyph.inv").view(torch.int32)[:T_].cpu().long()
    counts = m.tap("glyph.counts").view(torch.float32)[:T_].cpu()
    uids = m.tap("glyph.ids").view(torch.int64)[:T_].cpu()
    uniq = torch.unique(ids)
    U = int(bounds[0])
    assert U == uniq.numel() and U < T_
    assert [int(x) for x in bounds[1:6]] == [U * 256, U * 64, U * 16, U * 4, U]
    assert torch.equal(uids[:U][inv], ids)                                   # inverse map reproduces every token id
    assert torch.equal(counts[:U], torch.bincount(inv, minlength=U).float())  # multiplicities
    assert float(counts[:U].sum()) == T_
    first_pos = torch.tensor([int((ids == u).nonzero()[0]) for u in uids[:U]])
    assert torch.all(first_pos[1:] > first_pos[:-1])                          # deterministic: order of first occurrence


def _edge_batch(kind):
    """hand-built ragged / extreme batches (the reference pads every sentence to max_seq_length, run.py:68-101)"""
    from realise_amd.data import synthetic_batch
    if kind == "single_short":          # B=1, one real character
        b = synthetic_batch(1, 8, seed=31)
        b["src_idx"][0] = torch.tensor([101, 21127, 102, 0, 0, 0, 0, 0])
        b["tgt_idx"][0] = torch.tensor([101, 670, 102, 0, 0, 0, 0, 0])
        b["masks"][0] = torch.tensor([1, 1, 1, 0, 0, 0, 0, 0])
        b["loss_masks"][0] = torch.tensor([0, 1, 0, 0, 0, 0, 0, 0])
        b["pho_idx"] = torch.zeros((8, 7), dtype=torch.long)
        b["pho_idx"][:, 0] = 32
        b["pho_idx"][1] = torch.tensor([5, 6, 31, 17, 9, 22, 13])          # longest pinyin (7 letters)
        b["pho_lens"] = [1, 7, 1, 1, 1, 1, 1, 1]
        return b
    if kind == "full_length":           # maximum size: S = 128, no padding at all
        return synthetic_batch(2, 128, seed=32, full_length=True)
    if kind == "ragged":                # very different lengths, S not a multiple of 16
        b = synthetic_batch(5, 37, seed=33)
        return b
    if kind == "all_len1_pinyin":       # every token maps to 'U' (length 1): the GRU runs a single step
        b = synthetic_batch(2, 16, seed=34)
        n = 2 * 16
        b["pho_idx"] = torch.full((n, 1), 32, dtype=torch.long)
        b["pho_lens"] = [1] * n
        return b
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["single_short", "full_length", "ragged", "all_len1_pinyin"])
def test_edge_case_batches_fp32_match_oracle(kind):
    from realise_amd.init import init_state_dict_numpy
    cfg = RealiseConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=41, scheme="perturbed")
    batch = _edge_batch(kind)
    m = build("arch3", cfg, sd_np, "fp32", train=True)
    loss, logits = m(batch)
    loss.backward()
    torch.cuda.synchronize()
    taps = {}
    sd, nb, oloss, ologits = _oracle_train("arch3", cfg, sd_np, batch, taps)
    assert abs(loss.item() - oloss.item()) < 1e-4
    assert (logits.float().cpu() - ologits).abs().max().item() < FP32_LOGIT_TOL
    assert torch.equal(logits.argmax(-1).cpu(), ologits.argmax(-1))
    _assert_grads_match_oracle(m, sd, _relu_boundary_flips(m, taps), 3e-3, kind)


@pytest.mark.parametrize("num_fonts", [1, 2])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_one_and_two_font_models_match_oracle(num_fonts, dtype):
    """num_fonts in {1, 2} on the device (models.py:674-679, 831-834: one font keeps the table as an nn.Embedding [V, 1024], the
    ResNet's first convolutions take num_fonts input channels): forward + backward against the oracle - VERDICT round 3, item 9."""
    from realise_amd.init import init_state_dict_numpy
    cfg = RealiseConfig(num_hidden_layers=1, num_fonts=num_fonts, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=60 + num_fonts, scheme="perturbed")
    assert ("char_images.weight" in sd_np) == (num_fonts == 1)
    batch = synthetic_batch(3, 24, seed=61)
    m = build("arch3", cfg, sd_np, dtype, train=True)
    loss, logits = m(batch)
    loss.backward()
    torch.cuda.synchronize()
    taps = {}
    sd, nb, oloss, ologits = _oracle_train("arch3", cfg, sd_np, batch, taps)
    if dtype == "fp32":
        assert abs(loss.item() - oloss.item()) < 1e-4
        assert (logits.float().cpu() - ologits).abs().max().item() < FP32_LOGIT_TOL
        assert torch.equal(logits.argmax(-1).cpu(), ologits.argmax(-1))
        _assert_grads_match_oracle(m, sd, _relu_boundary_flips(m, taps), 3e-3, "num_fonts=%d" % num_fonts)
        for k, v in nb.items():
            assert (m.state_dict()[k].cpu().double() - v.double()).abs().max().item() < 1e-4, k
    else:
        # (train mode: BatchNorm statistics of a 72-token batch in bf16 - wider than the eval band of the large-batch tests)
        assert abs(loss.item() - oloss.item()) < 5e-2
        assert (logits.float().cpu() - ologits).abs().max().item() < 0.15
        w1 = dict(m.named_parameters())["resnet.res_block1.residual_function.0.weight"]
        assert w1.shape[1] == num_fonts and torch.isfinite(w1.grad).all() and w1.grad.abs().max().item() > 0


def test_glyph_branch_is_bitwise_reproducible():
    """BatchNorm statistics are folded in a fixed order (no float atomics), so two passes over the same batch give
    identical activations and therefore identical ReLU masks (before this, a boundary activation could flip between runs)."""
    from realise_amd.init import init_state_dict_numpy
    cfg = RealiseConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=41, scheme="perturbed")
    batch = _edge_batch("ragged")
    m = build("arch3", cfg, sd_np, "fp32", train=True)
    runs = []
    for _ in range(3):
        m.zero_grad()
        loss, logits = m(batch)
        loss.backward()
        torch.cuda.synchronize()
        runs.append([m.tap("resnet.block%d" % k).clone() for k in range(1, 6)] + [logits.detach().clone()])
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)


def test_rejects_sequences_longer_than_the_position_table():
    """S > 128 runs (round 5: tiled attention kernels; the reference pads to --max_seq_length, run.py:304) up to the position table
    (config.max_position_embeddings = 512, modeling_bert.py:160); beyond it the reference's embedding lookup fails and so does the engine."""
    from realise_amd import _capi
    from realise_amd.data import synthetic_batch
    cfg = RealiseConfig(num_hidden_layers=1)
    m = SpellBert(cfg, compute_dtype="bf16").to("cuda").eval()
    with torch.no_grad():
        loss, logits = m(synthetic_batch(1, 160, with_pho=False))
    assert tuple(logits.shape[:2]) == (1, 160) and torch.isfinite(logits.float()).all() and torch.isfinite(loss)
    with pytest.raises(_capi.RealiseHipError):
        m(synthetic_batch(1, cfg.max_position_embeddings + 16, with_pho=False))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_decode_equals_host_argmax_and_golden_in_fp32(golden_dir, dtype):
    """model.decode == np.argmax(logits.cpu()) (run.py:262-263) with only the ids leaving the device; in fp32 the ids are
    also the reference's own (golden) arg-max ids."""
    g = load_golden(golden_dir, "arch3_b4s128_eval")
    cfg, sd_np, batch = golden_case_inputs(g, "arch3")
    m = build("arch3", cfg, sd_np, dtype)
    with torch.no_grad():
        loss, logits = m(batch)
    ids = m.decode(logits)
    assert ids.dtype == torch.int64 and ids.is_cuda and tuple(ids.shape) == tuple(logits.shape[:2])
    assert np.array_equal(ids.cpu().numpy(), np.argmax(logits.float().cpu().numpy(), axis=-1))
    assert torch.equal(m.decode(batch), ids)
    if dtype == "fp32":
        assert np.array_equal(ids.cpu().numpy().astype(np.int32), g["argmax"])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_device_build_batch_equals_host_batch(dtype):
    """a batch completed on the device from src_idx alone (per-vocabulary pinyin table, device sort, device
    alive counts) gives the same loss, logits and gradients as the reference-style batch with host pho_idx / pho_lens."""
    from realise_amd.init import init_state_dict_numpy
    from realise_amd.pinyin import PinyinTable
    from realise_amd.data import synthetic_batch
    cfg = RealiseConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=43, scheme="perturbed")
    g = np.random.default_rng(5)
    V = cfg.vocab_size
    vlens = g.integers(2, 8, V).astype(np.int32)
    vlens[[0, 101, 102]] = 1
    table = np.zeros((V, 7), np.int64)
    for v in range(V):
        table[v, :vlens[v]] = 32 if vlens[v] == 1 else np.concatenate([g.integers(1, 6, 1), g.integers(6, 32, vlens[v] - 1)])
    tab = PinyinTable(table, vlens)
    batch = synthetic_batch(3, 40, seed=9)
    host = dict(batch)
    host["pho_idx"], host["pho_lens"] = tab.convert(batch["src_idx"].numpy())
    host["pho_idx"] = torch.from_numpy(host["pho_idx"])
    dev = {k: v for k, v in batch.items() if k not in ("pho_idx", "pho_lens")}
    results = []
    for b in (host, dev):
        m = build("arch3", cfg, sd_np, dtype, train=True)
        if b is dev:
            m.set_pinyin_table(tab)
        loss, logits = m(b)
        loss.backward()
        torch.cuda.synchronize()
        results.append((loss.item(), logits.detach().float().cpu(), m.flat_gradients().clone().cpu()))
    (l0, lg0, g0), (l1, lg1, g1) = results
    tol = 1e-5 if dtype == "fp32" else 2e-2
    assert abs(l0 - l1) < tol
    assert (lg0 - lg1).abs().max().item() < (1e-4 if dtype == "fp32" else 6e-2)
    assert ((g0 - g1).norm() / g0.norm()).item() < (1e-5 if dtype == "fp32" else 2e-2)


def test_wgrad_side_stream_overlap_gives_the_same_gradients(golden_dir):
    """realise_set_wgrad_overlap(1): the weight-gradient GEMMs of the BERT layers run on the engine's side stream with
    double-buffered operands; losses and weight gradients must be identical to the in-order schedule."""
    from realise_amd import _capi
    lib = _capi.load()
    g = load_golden(golden_dir, "arch3_b3s40_train")
    cfg, sd_np, batch = golden_case_inputs(g, "arch3")
    out = []
    try:
        for on in (0, 1):
            lib.realise_set_wgrad_overlap(on)
            m = build("arch3", cfg, sd_np, "fp32", train=True)
            for _ in range(2):                                  # two steps: the second re-uses both operand sets
                m.zero_grad()
                loss, logits = m(batch)
                loss.backward()
            torch.cuda.synchronize()
            out.append((loss.item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    finally:
        lib.realise_set_wgrad_overlap(1)
    assert abs(out[0][0] - out[1][0]) < 1e-5                 # the loss sum itself uses fp32 atomics
    for n, g0 in out[0][1].items():
        g1 = out[1][1][n]
        if n.endswith("weight") and g0.dim() == 2 and ".layer." in n and "LayerNorm" not in n:
            assert torch.equal(g0, g1), n                      # slab-reduced GEMM results: bitwise
        else:
            assert (g0 - g1).abs().max().item() <= 1e-5 * (g0.abs().max().item() + 1e-12) + 1e-9, n


def test_full_size_config2_properties_bf16():
    """BASELINE configs[1] at full size (B=64, S=128, 12+4+3 layers, bf16): the oracle would need minutes per pass, so
    the checks are size-independent properties: (1) running the glyph ResNet once per distinct token (dedup) equals the
    dense per-token pass; (2) the loss is permutation-invariant over the sentences of the batch and the logits permute
    with them (BatchNorm statistics and the masked-mean gate see the same multiset); (3) decode == host arg-max;
    (4) every gradient is finite and its global norm is dedup-invariant; (5) a step with lr = 0 leaves the weights and
    the next loss untouched (clip + AdamW plumbing at full arena size)."""
    from realise_amd import _capi
    from realise_amd.data import synthetic_batch
    from realise_amd.optim import FusedAdamW
    lib = _capi.load()
    cfg = RealiseConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = SpellBertPho2ResArch3(cfg, compute_dtype="bf16", seed=3).to("cuda").train()
    batch = synthetic_batch(64, 128, seed=77)
    res = {}
    try:
        for on in (1, 0):
            lib.realise_set_glyph_dedup(on)
            m.zero_grad()
            loss, logits = m(batch)
            loss.backward()
            torch.cuda.synchronize()
            res[on] = (loss.item(), logits.detach().float(), m.flat_gradients().clone())
    finally:
        lib.realise_set_glyph_dedup(1)
    (l1, lg1, g1), (l0, lg0, g0) = res[1], res[0]
    assert np.isfinite(l1) and abs(l1 - l0) < 2e-3 * abs(l0)
    # the logits that mean something: rows before their sentence's last attended / loss position (a bf16 training step does not compute
    # the transformer stacks on the padding rows behind it - realise_set_engine(10, .) - as nothing reads them; they stay finite)
    flagged = (batch["masks"] == 1) | (batch["loss_masks"] == 1)
    last = (flagged * torch.arange(1, 129)[None, :]).max(dim=1).values
    live = (torch.arange(128)[None, :] < last[:, None]).to(lg1.device)
    assert torch.isfinite(lg1).all() and torch.isfinite(lg0).all()
    assert (lg1 - lg0)[live].abs().max().item() < 6e-2
    assert torch.isfinite(g1).all() and abs(g1.norm().item() - g0.norm().item()) < 2e-2 * g0.norm().item()
    rel = ((g1 - g0).norm() / g0.norm()).item()
    assert rel < 5e-2, rel
    # (3)
    assert torch.equal(m.decode(lg1.to(torch.bfloat16)).cpu(), lg1.to(torch.bfloat16).float().cpu().argmax(-1))
    # (2) permute the sentences
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(5))
    pb = {k: (v[perm] if torch.is_tensor(v) and v.shape[0] == 64 else v) for k, v in batch.items()}
    tok = (perm[:, None] * 128 + torch.arange(128)[None, :]).reshape(-1)
    pb["pho_idx"] = batch["pho_idx"][tok]
    pb["pho_lens"] = [batch["pho_lens"][i] for i in tok.tolist()]
    m.zero_grad()
    lp, lgp = m(pb)
    assert abs(lp.item() - l1) < 2e-3 * abs(l1)
    assert (lgp.detach().float() - lg1[perm])[live[perm]].abs().max().item() < 6e-2
    # (5)
    lp.backward()
    before = m.flat_parameters().clone()
    FusedAdamW(m, [{"params": list(m.parameters()), "weight_decay": 0.0}], lr=0.0, max_grad_norm=1.0).step()
    torch.cuda.synchronize()
    assert torch.equal(before, m.flat_parameters())
