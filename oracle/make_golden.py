"""Generate tests/golden/*.npz by running the UPSTREAM REFERENCE in this container.

TEST INFRASTRUCTURE; run only where /root/reference exists:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference (a Python program) cannot travel to the GPU box, so its outputs
on seeded inputs are committed as small fixtures: inputs are regenerated from
seeds by realise_amd.data / realise_amd.init (torch-independent generators), the
fixture stores output *samples and statistics* (strided samples, sums) plus the
argmax ids, never the reference's code.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from realise_amd.config import RealiseConfig          # noqa: E402
from realise_amd.data import synthetic_batch          # noqa: E402
from realise_amd.init import init_state_dict_numpy    # noqa: E402
from _ref_import import import_reference              # noqa: E402
import realise_ref as R                               # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N_SAMPLE = 192


def summarize(t):
    """strided sample + sums of a tensor -> dict of small numpy arrays"""
    a = t.detach().to(torch.float64).reshape(-1)
    n = a.numel()
    stride = max(1, n // N_SAMPLE)
    return {
        "sample": a[::stride][:N_SAMPLE].to(torch.float32).numpy(),
        "sum": np.float64(a.sum().item()),
        "abssum": np.float64(a.abs().sum().item()),
        "n": np.int64(n),
    }


def put(store, key, t):
    for k, v in summarize(t).items():
        store["%s/%s" % (key, k)] = v


def build_reference(models, BertConfig, cfg, model_type, sd_np, train):
    bc = BertConfig(vocab_size_or_config_json_file=cfg["vocab_size"])
    for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
              "hidden_dropout_prob", "attention_probs_dropout_prob", "max_position_embeddings",
              "type_vocab_size", "layer_norm_eps", "initializer_range"):
        setattr(bc, k, cfg[k])
    bc.image_model_type = 0
    bc.num_fonts = cfg["num_fonts"]
    cls = models.SpellBertPho2ResArch3 if model_type == "arch3" else models.SpellBert
    m = cls(bc)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    m.tie_cls_weight()
    m.train(train)
    return m


def oracle_sd(sd_np, requires_grad=False):
    sd = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.array(v, copy=True))
        if requires_grad and t.dtype == torch.float32 and k != "char_images_multifonts" \
                and "running_" not in k:
            t.requires_grad_(True)
        sd[k] = t
    sd["classifier.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    return sd


def case_model(models, BertConfig, name, model_type, B, S, train, seed, n_layers=12, with_grads=False):
    t0 = time.time()
    cfg = RealiseConfig(num_hidden_layers=n_layers, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, model_type, seed=seed, scheme="perturbed")
    batch = synthetic_batch(B, S, seed=seed, with_pho=(model_type == "arch3"))
    m = build_reference(models, BertConfig, cfg, model_type, sd_np, train)
    store = {"meta/B": np.int64(B), "meta/S": np.int64(S), "meta/seed": np.int64(seed),
             "meta/n_layers": np.int64(n_layers), "meta/train": np.int64(train)}
    # hooks for intermediate taps
    taps = {}
    if model_type == "arch3":
        hooks = [
            m.bert.register_forward_hook(lambda mod, i, o: taps.__setitem__("bert_h", o[0])),
            m.pho_gru.register_forward_hook(lambda mod, i, o: taps.__setitem__("pho_gru", o[1].squeeze(0))),
            m.pho_model.register_forward_hook(lambda mod, i, o: taps.__setitem__("pho_h", o[0])),
            m.resnet.register_forward_hook(lambda mod, i, o: taps.__setitem__("res", o)),
            m.resnet_layernorm.register_forward_hook(lambda mod, i, o: taps.__setitem__("res_h", o)),
            m.output_block.register_forward_hook(lambda mod, i, o: taps.__setitem__("out", o[0])),
        ]
        for b in range(1, 6):
            hooks.append(getattr(m.resnet, "res_block%d" % b).register_forward_hook(
                lambda mod, i, o, b=b: taps.__setitem__("resnet.block%d" % b, o)))
    else:
        hooks = [m.bert.register_forward_hook(lambda mod, i, o: taps.__setitem__("bert_h", o[0]))]
    hooks.append(m.bert.embeddings.register_forward_hook(lambda mod, i, o: taps.__setitem__("bert.emb", o)))
    hooks.append(m.bert.encoder.layer[0].register_forward_hook(
        lambda mod, i, o: taps.__setitem__("bert.encoder.layer.0.out", o[0])))
    if with_grads:
        loss, logits = m(batch)[:2]
        loss.backward()
    else:
        with torch.no_grad():
            loss, logits = m(batch)[:2]
    for k, v in taps.items():
        put(store, "tap/" + k, v)
    store["loss"] = np.float64(loss.item())
    put(store, "logits", logits)
    store["argmax"] = logits.argmax(-1).to(torch.int32).numpy()
    top2 = logits.topk(2, dim=-1).values
    store["margin"] = (top2[..., 0] - top2[..., 1]).detach().to(torch.float32).numpy()
    if train and model_type == "arch3":
        for k, v in m.state_dict().items():
            if "running_" in k or "num_batches" in k:
                put(store, "buf/" + k, v.to(torch.float64))
    if with_grads:
        for k, p in m.named_parameters():
            if p.grad is None:
                store["gradnone/" + k] = np.int64(1)
            else:
                put(store, "grad/" + k, p.grad)
    # ---- check the restatement right here ---------------------------------------------
    osd = oracle_sd(sd_np, requires_grad=with_grads)
    otaps, nb = {}, {}
    fwd = R.arch3_forward if model_type == "arch3" else R.spellbert_forward
    kw = dict(training=train, taps=otaps)
    if model_type == "arch3":
        kw["new_buffers"] = nb
    if with_grads:
        oloss, ologits = fwd(osd, cfg, batch, **kw)
        oloss.backward()
    else:
        with torch.no_grad():
            oloss, ologits = fwd(osd, cfg, batch, **kw)
    dl = (ologits - logits).abs().max().item()
    print("[%s] loss ref %.6f oracle %.6f | logits maxdiff %.3e | argmax equal %s | %.1fs"
          % (name, loss.item(), oloss.item(), dl,
             bool((ologits.argmax(-1) == logits.argmax(-1)).all()), time.time() - t0))
    if with_grads:
        worst = 0.0
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            kk = k
            g = osd[kk].grad
            d = (g - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-12)
            worst = max(worst, d)
        print("   worst relative grad diff %.3e" % worst)
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)


def case_gru(models, BertConfig):
    """GRU last hidden state for every pinyin length 1..7 (SURVEY 8c list)."""
    cfg = RealiseConfig()
    sd_np = {}
    from realise_amd.init import tensor_init, tensor_specs
    for name, shape, kind in tensor_specs(cfg, "arch3"):
        if name.startswith("pho_gru") or name.startswith("pho_embeddings"):
            sd_np[name] = tensor_init(name, shape, kind, cfg, seed=3, scheme="perturbed")
    H = cfg["hidden_size"]
    emb = torch.nn.Embedding(33, H, padding_idx=0)
    gru = torch.nn.GRU(H, H, 1, batch_first=True)
    emb.weight.data.copy_(torch.from_numpy(sd_np["pho_embeddings.weight"]))
    for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
        getattr(gru, k).data.copy_(torch.from_numpy(sd_np["pho_gru." + k]))
    g = np.random.Generator(np.random.Philox(key=[7, 7]))
    lens = [1, 2, 3, 4, 5, 6, 7, 3, 1, 7, 2, 5, 4, 6, 1, 1]
    idx = np.zeros((len(lens), 7), np.int64)
    for i, l in enumerate(lens):
        idx[i, :l] = g.integers(1, 33, size=l)
    idx_t = torch.from_numpy(idx)
    with torch.no_grad():
        packed = torch.nn.utils.rnn.pack_padded_sequence(emb(idx_t), lens, batch_first=True, enforce_sorted=False)
        _, hn = gru(packed)                      # src/models.py:818-826
        hn = hn.squeeze(0)
        o = R.pho_gru_last_hidden({k: torch.from_numpy(v) for k, v in sd_np.items()}, idx_t, lens)
    print("[gru] maxdiff oracle-vs-torch.nn.GRU %.3e" % (o - hn).abs().max().item())
    np.savez_compressed(os.path.join(OUT, "gru_lengths.npz"), pho_idx=idx, pho_lens=np.array(lens),
                        h_last=hn.numpy())


def case_optim():
    """3 AdamW steps (transformers/optimization.py:110-169) + the linear
    warm-up table of transformers/tests/optimization_test.py:93-146."""
    sys.path.insert(0, "/root/reference")
    from transformers.optimization import AdamW, get_linear_schedule_with_warmup
    g = np.random.Generator(np.random.Philox(key=[11, 11]))
    p0 = g.standard_normal((5, 7)).astype(np.float32)
    grads = [g.standard_normal((5, 7)).astype(np.float32) for _ in range(3)]
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = AdamW([p], lr=1e-2, weight_decay=0.01, eps=1e-8)
    sched = get_linear_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=10)
    traj, lrs = [], []
    for gr in grads:
        p.grad = torch.from_numpy(gr.copy())
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        traj.append(p.detach().numpy().copy())
    # schedule KAT as in the reference's own test: lr=10, warmup 2, total 10
    q = torch.nn.Parameter(torch.zeros(1))
    o2 = AdamW([q], lr=10.0)
    s2 = get_linear_schedule_with_warmup(o2, num_warmup_steps=2, num_training_steps=10)
    table = []
    for _ in range(10):
        s2.step()
        table.append(o2.param_groups[0]["lr"])
    print("[optim] schedule table", table)
    np.savez_compressed(os.path.join(OUT, "adamw_steps.npz"), p0=p0, grads=np.stack(grads),
                        traj=np.stack(traj), lrs=np.array(lrs, np.float64), sched_table=np.array(table))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    models, BertConfig = import_reference()
    if "--only-spellbert-b8s64-train" in sys.argv:      # round 4: BASELINE configs[0] at its stated size, training step
        case_model(models, BertConfig, "spellbert_b8s64_train", "bert", 8, 64, True, seed=8, n_layers=12, with_grads=True)
        return
    case_gru(models, BertConfig)
    case_optim()
    case_model(models, BertConfig, "spellbert_b2s16_eval", "bert", 2, 16, False, seed=1, n_layers=2)
    case_model(models, BertConfig, "spellbert_b2s16_train", "bert", 2, 16, True, seed=2, n_layers=2, with_grads=True)
    case_model(models, BertConfig, "arch3_b2s16_eval", "arch3", 2, 16, False, seed=3, n_layers=2)
    case_model(models, BertConfig, "arch3_b2s16_train", "arch3", 2, 16, True, seed=4, n_layers=2, with_grads=True)
    case_model(models, BertConfig, "arch3_b3s40_train", "arch3", 3, 40, True, seed=5, n_layers=12, with_grads=True)
    case_model(models, BertConfig, "spellbert_b8s64_eval", "bert", 8, 64, False, seed=6, n_layers=12)
    case_model(models, BertConfig, "arch3_b4s128_eval", "arch3", 4, 128, False, seed=7, n_layers=12)
    case_model(models, BertConfig, "spellbert_b8s64_train", "bert", 8, 64, True, seed=8, n_layers=12, with_grads=True)


if __name__ == "__main__":
    main()
