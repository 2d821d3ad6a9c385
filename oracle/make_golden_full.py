"""Full-size golden vectors (BASELINE configs[1] and configs[3]) from the UPSTREAM REFERENCE run in this container.

TEST INFRASTRUCTURE; run only where /root/reference exists (a few minutes on 8 cores):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_full.py

* ``arch3_b64s128_eval``  - the reference SpellBertPho2ResArch3 (12 + 4 + 3 layers) eval forward on the B=64, S=128
  SIGHAN-shaped synthetic batch of bench.py: arg-max ids of all 8192 tokens, top-1/top-2 margins, loss, 8192 sampled logits
  (one vocabulary slot per token) and strided samples of the intermediate taps.
* ``resnet_b256s128``     - the reference CharResNet (src/char_cnn.py:35-55) alone on the 32768 glyph stacks of a B=256, S=128
  batch (BASELINE configs[3]), in train mode (batch statistics; updated BN buffers stored) and in eval mode: per-block output
  statistics and strided samples of the [32768, 768] result.

* ``arch3_b64s128_train`` - the reference's TRAINING step at the same size (train mode, dropout 0): loss, forward taps, the
  gradient of every gradient-receiving parameter (L2 norm, sum, |sum|, 192 strided samples) after ``loss.backward()``
  (src/run.py:191-200) and the updated BatchNorm buffers.
* ``arch3_b8s256_train`` / ``arch3_b4s512_train`` - the same training step at B=8, S=256 and B=4, S=512 (max_seq_length beyond the
  default 128: src/run.py:304; position table of 512 rows).
* ``resnet_b256s128_bwd`` - the reference CharResNet's backward on the 32768 glyph stacks for a seeded upstream gradient
  (``glyph_upstream_grad``): gradient norm / sums / samples of its 45 parameters.

Inputs are regenerated from seeds (realise_amd.data / realise_amd.init); only outputs are stored.
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
from make_golden import build_reference, put          # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def case_full_eval(models, BertConfig, name="arch3_b64s128_eval", B=64, S=128, seed=8):
    t0 = time.time()
    cfg = RealiseConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=seed, scheme="perturbed")
    batch = synthetic_batch(B, S, seed=seed)
    m = build_reference(models, BertConfig, cfg, "arch3", sd_np, False)
    taps = {}
    hooks = [
        m.bert.register_forward_hook(lambda mod, i, o: taps.__setitem__("bert_h", o[0])),
        m.pho_gru.register_forward_hook(lambda mod, i, o: taps.__setitem__("pho_gru", o[1].squeeze(0))),
        m.pho_model.register_forward_hook(lambda mod, i, o: taps.__setitem__("pho_h", o[0])),
        m.resnet.register_forward_hook(lambda mod, i, o: taps.__setitem__("res", o)),
        m.resnet_layernorm.register_forward_hook(lambda mod, i, o: taps.__setitem__("res_h", o)),
        m.output_block.register_forward_hook(lambda mod, i, o: taps.__setitem__("out", o[0])),
    ]
    with torch.no_grad():
        loss, logits = m(batch)[:2]
    for h in hooks:
        h.remove()
    store = {"meta/B": np.int64(B), "meta/S": np.int64(S), "meta/seed": np.int64(seed), "meta/n_layers": np.int64(12),
             "meta/train": np.int64(0), "loss": np.float64(loss.item())}
    for k, v in taps.items():
        put(store, "tap/" + k, v)
    put(store, "logits", logits)
    store["argmax"] = logits.argmax(-1).to(torch.int32).numpy()
    top2 = logits.topk(2, dim=-1).values
    store["margin"] = (top2[..., 0] - top2[..., 1]).to(torch.float32).numpy()
    # one sampled vocabulary slot per token (deterministic): the arg-max slot of every 4th token, a hashed slot otherwise
    V = logits.shape[-1]
    flat = logits.reshape(-1, V)
    tok = np.arange(flat.shape[0], dtype=np.int64)
    slot = (tok * 2654435761 % V).astype(np.int64)
    am = store["argmax"].reshape(-1).astype(np.int64)
    slot[::4] = am[::4]
    store["sample_slot"] = slot.astype(np.int32)
    store["sample_logit"] = flat[torch.from_numpy(tok), torch.from_numpy(slot)].to(torch.float32).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
    print("[%s] loss %.6f, min margin %.3e, %.1fs" % (name, loss.item(), float(store["margin"][batch["masks"].numpy() == 1].min()),
                                                      time.time() - t0))


def case_resnet(models, BertConfig, name="resnet_b256s128", B=256, S=128, seed=9):
    t0 = time.time()
    cfg = RealiseConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=seed, scheme="perturbed")
    batch = synthetic_batch(B, S, seed=seed, with_pho=False)
    m = build_reference(models, BertConfig, cfg, "arch3", sd_np, True)
    ids = batch["src_idx"].view(-1)
    store = {"meta/B": np.int64(B), "meta/S": np.int64(S), "meta/seed": np.int64(seed)}
    for mode in ("eval", "train"):        # eval first: it must see the INITIAL running statistics, the train pass updates them
        m.resnet.train(mode == "train")
        taps = {}
        hooks = [getattr(m.resnet, "res_block%d" % b).register_forward_hook(
            lambda mod, i, o, b=b: taps.__setitem__("block%d" % b, o)) for b in range(1, 6)]
        with torch.no_grad():
            images = m.char_images_multifonts.index_select(0, ids)        # src/models.py:831-834
            res = m.resnet(images)                                        # [B*S, 768]
        for h in hooks:
            h.remove()
        for k, v in taps.items():
            put(store, "%s/%s" % (mode, k), v)
        put(store, mode + "/res", res)
        rows = np.arange(0, res.shape[0], 61)                             # full 768-wide rows of ~540 tokens
        store[mode + "/rows"] = rows.astype(np.int32)
        store[mode + "/res_rows"] = res[torch.from_numpy(rows)].to(torch.float32).numpy()
        if mode == "train":
            for k, v in m.resnet.state_dict().items():
                if "running_" in k or "num_batches" in k:
                    put(store, "buf/resnet." + k, v.to(torch.float64))
        print("[%s] %s forward done, %.1fs" % (name, mode, time.time() - t0))
        del taps, images, res
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)


N_GRAD_SAMPLE = 1024


def put_grad(store, key, g):
    """gradient summary: the 192-sample summary of every fixture + L2 norm, |max| and 1024 strided samples (the bf16 test
    estimates the direction cosine on them)"""
    put(store, key, g)
    a = g.detach().to(torch.float64).reshape(-1)
    store[key + "/l2"] = np.float64(a.norm().item())
    store[key + "/absmax"] = np.float64(a.abs().max().item())
    stride = max(1, a.numel() // N_GRAD_SAMPLE)
    store[key + "/sample1k"] = a[::stride][:N_GRAD_SAMPLE].to(torch.float32).numpy()


def case_full_train(models, BertConfig, name="arch3_b64s128_train", B=64, S=128, seed=8):
    t0 = time.time()
    cfg = RealiseConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=seed, scheme="perturbed")
    batch = synthetic_batch(B, S, seed=seed)
    m = build_reference(models, BertConfig, cfg, "arch3", sd_np, True)
    taps = {}
    hooks = [
        m.bert.register_forward_hook(lambda mod, i, o: taps.__setitem__("bert_h", o[0])),
        m.pho_gru.register_forward_hook(lambda mod, i, o: taps.__setitem__("pho_gru", o[1].squeeze(0))),
        m.resnet.register_forward_hook(lambda mod, i, o: taps.__setitem__("res", o)),
        m.resnet_layernorm.register_forward_hook(lambda mod, i, o: taps.__setitem__("res_h", o)),
        m.output_block.register_forward_hook(lambda mod, i, o: taps.__setitem__("out", o[0])),
    ]
    loss, logits = m(batch)[:2]
    print("[%s] forward %.1fs" % (name, time.time() - t0))
    loss.backward()                                                        # src/run.py:200
    print("[%s] backward %.1fs" % (name, time.time() - t0))
    for h in hooks:
        h.remove()
    store = {"meta/B": np.int64(B), "meta/S": np.int64(S), "meta/seed": np.int64(seed), "meta/n_layers": np.int64(12),
             "meta/train": np.int64(1), "loss": np.float64(loss.item())}
    for k, v in taps.items():
        put(store, "tap/" + k, v)
    put(store, "logits", logits)
    store["argmax"] = logits.argmax(-1).to(torch.int32).numpy()
    for k, v in m.state_dict().items():
        if "running_" in k or "num_batches" in k:
            put(store, "buf/" + k, v.to(torch.float64))
    n_none = 0
    for k, p in m.named_parameters():
        if p.grad is None:
            store["gradnone/" + k] = np.int64(1)
            n_none += 1
        else:
            put_grad(store, "grad/" + k, p.grad)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
    print("[%s] loss %.6f, %d tensors without gradient, %.1fs" % (name, loss.item(), n_none, time.time() - t0))


def case_resnet_bwd(models, BertConfig, name="resnet_b256s128_bwd", B=256, S=128, seed=9):
    from realise_amd.data import glyph_upstream_grad
    t0 = time.time()
    cfg = RealiseConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=seed, scheme="perturbed")
    batch = synthetic_batch(B, S, seed=seed, with_pho=False)
    m = build_reference(models, BertConfig, cfg, "arch3", sd_np, True)
    ids = batch["src_idx"].view(-1)
    d_res = torch.from_numpy(glyph_upstream_grad(B * S, 768, seed=seed))
    images = m.char_images_multifonts.index_select(0, ids)                # src/models.py:831-834 (frozen table: no gradient)
    res = m.resnet(images)                                                # src/char_cnn.py:35-55, train mode
    print("[%s] forward %.1fs" % (name, time.time() - t0))
    res.backward(d_res)
    print("[%s] backward %.1fs" % (name, time.time() - t0))
    store = {"meta/B": np.int64(B), "meta/S": np.int64(S), "meta/seed": np.int64(seed)}
    put(store, "res", res)
    for k, p in m.resnet.named_parameters():
        put_grad(store, "grad/resnet." + k, p.grad)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
    print("[%s] %d gradient tensors, %.1fs" % (name, len(list(m.resnet.parameters())), time.time() - t0))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    models, BertConfig = import_reference()
    which = sys.argv[1:] or ["eval", "resnet"]
    if "eval" in which:
        case_full_eval(models, BertConfig)
    if "resnet" in which:
        case_resnet(models, BertConfig)
    if "train" in which:
        case_full_train(models, BertConfig)
    if "resnet_bwd" in which:
        case_resnet_bwd(models, BertConfig)
    if "train256" in which:                               # S > 128 (run.py:304 --max_seq_length 256): the tiled attention kernels
        case_full_train(models, BertConfig, name="arch3_b8s256_train", B=8, S=256, seed=12)
    if "train512" in which:
        case_full_train(models, BertConfig, name="arch3_b4s512_train", B=4, S=512, seed=13)


if __name__ == "__main__":
    main()
