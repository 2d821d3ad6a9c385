"""Host-side logic around the path (CPU): feature padding / masks (run.py:68-101), sharding (run.py:130-137),
the synthetic SIGHAN-shaped batch generator, and the deterministic weight generator."""
import pytest
import os
import numpy as np
import torch

from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.init import init_state_dict_numpy, tensor_init
from realise_amd.trainer import data_helper, make_features, shard


def _item(i, n):
    ids = [101] + list(range(700, 700 + n)) + [102]
    return {"id": i, "src": "x" * n, "tgt": "x" * n, "tokens_size": [1] * n, "src_idx": ids, "tgt_idx": ids, "lengths": n}


def test_make_features_masks_match_reference_layout():
    b = make_features([_item(0, 3), _item(1, 5)], 10)
    assert b["src_idx"].shape == (2, 10) and b["src_idx"].dtype == torch.long
    assert b["masks"].tolist() == [[1] * 5 + [0] * 5, [1] * 7 + [0] * 3]
    assert b["loss_masks"].tolist() == [[0, 1, 1, 1] + [0] * 6, [0, 1, 1, 1, 1, 1] + [0] * 4]     # chars only, run.py:86-92
    assert b["lengths"] == [3, 5]


def test_shard_and_data_helper():
    items = [_item(i, 4) for i in range(11)]
    s0, s1 = shard(items, 0, 2), shard(items, 1, 2)
    assert [e["id"] for e in s0] == [0, 2, 4, 6, 8] and [e["id"] for e in s1] == [1, 3, 5, 7, 9]       # tail dropped
    batches = list(data_helper(items, 4, 8, lambda b, t: b, is_eval=True))
    assert [len(b["id"]) for b in batches] == [4, 4, 3]
    a = [b["id"] for b in data_helper(items, 4, 8, lambda b, t: b, seed=3)]
    c = [b["id"] for b in data_helper(items, 4, 8, lambda b, t: b, seed=3)]
    assert a == c and sorted(sum(a, [])) == list(range(11))


def test_synthetic_batch_is_sighan_shaped_and_deterministic():
    b1, b2 = synthetic_batch(4, 32, seed=5), synthetic_batch(4, 32, seed=5)
    for k in ("src_idx", "tgt_idx", "masks", "loss_masks", "pho_idx"):
        assert torch.equal(b1[k], b2[k])
    assert b1["pho_lens"] == b2["pho_lens"] and len(b1["pho_lens"]) == 4 * 32
    src, m, lm = b1["src_idx"], b1["masks"], b1["loss_masks"]
    assert (src[:, 0] == 101).all()
    for r in range(4):
        L = int(m[r].sum()) - 2
        assert src[r, L + 1] == 102 and (src[r, L + 2:] == 0).all()
        assert lm[r].tolist() == [0] + [1] * L + [0] * (32 - 1 - L)
    lens = np.array(b1["pho_lens"]).reshape(4, 32)
    assert ((lens >= 1) & (lens <= 7)).all()
    assert (lens[m.numpy() == 0] == 1).all()                       # PAD -> 'U'
    pho = b1["pho_idx"]
    assert pho.shape[0] == 128 and pho.shape[1] == lens.max()
    assert ((pho > 0).sum(1).numpy() == lens.reshape(-1)).all()    # zero padded past each length


def test_weight_generator_is_reproducible_and_named():
    cfg = RealiseConfig(num_hidden_layers=1)
    a = tensor_init("bert.encoder.layer.0.output.dense.weight", (768, 3072), "normal", cfg, seed=1)
    b = tensor_init("bert.encoder.layer.0.output.dense.weight", (768, 3072), "normal", cfg, seed=1)
    c = tensor_init("bert.encoder.layer.0.intermediate.dense.weight", (768, 3072), "normal", cfg, seed=1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(a.std() - 0.02) < 1e-3                             # initializer_range, modeling_bert.py:496-506
    sd = init_state_dict_numpy(cfg, "bert", seed=0)
    assert sd["classifier.weight"] is sd["bert.embeddings.word_embeddings.weight"]
    assert float(sd["bert.embeddings.LayerNorm.weight"].min()) == 1.0 and float(np.abs(sd["classifier.bias"]).max()) == 0.0


# ---- eval tail: label lines and sentence-level scores against vectors made by the reference ----
def _metric_cases():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "metric_cases.json"), encoding="utf-8") as f:
        return json.load(f)


def test_label_lines_match_reference_vectors(tmp_path):
    from realise_amd.metric import Metric
    cases = _metric_cases()
    (tmp_path / "vocab.txt").write_text("\n".join(cases["vocab"]) + "\n", encoding="utf-8")
    m = Metric(str(tmp_path))
    for c in cases["decode"]:
        batch = {"id": [c["id"]], "src": [c["src"]], "lengths": [c["length"]], "tokens_size": [c["tokens_size"]],
                 "pred_idx": np.array([c["pred_idx"]]), "src_idx": np.zeros((1, len(c["pred_idx"])), np.int64)}
        txt, lbl = m.process_batch_item(batch, 0)
        assert txt == c["txt"] and lbl == c["lbl"]


def test_sentence_scores_match_reference_vectors(tmp_path):
    from realise_amd.metric import metric_file
    for k, c in enumerate(_metric_cases()["score"]):
        p, g = tmp_path / ("p%d" % k), tmp_path / ("g%d" % k)
        p.write_text("\n".join(c["pred"]), encoding="utf-8")
        g.write_text("\n".join(c["gold"]), encoding="utf-8")
        got = metric_file(str(p), str(g))
        assert set(got) == set(c["results"])
        for key, v in c["results"].items():
            assert got[key] == pytest.approx(v, abs=1e-12), key


def test_metric_end_to_end_writes_reference_format(tmp_path):
    from realise_amd.metric import Metric
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "a", "b", "c"]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    batch = {"id": ["x1", "x2"], "src": ["abc", "cb"], "lengths": [3, 2], "tokens_size": [[1, 1, 1], [1, 1]],
             "src_idx": np.zeros((2, 6), np.int64), "pred_idx": np.array([[2, 4, 5, 6, 3, 0], [2, 6, 4, 3, 0, 0]])}
    (tmp_path / "gold.txt").write_text("x1, 0\nx2, 2, a", encoding="utf-8")
    res = Metric(str(tmp_path)).metric([batch], str(tmp_path / "out" / "preds.txt"), str(tmp_path / "out" / "labels.txt"), str(tmp_path / "gold.txt"))
    assert (tmp_path / "out" / "labels.txt").read_text(encoding="utf-8") == "x1, 0\nx2, 2, a"
    assert (tmp_path / "out" / "preds.txt").read_text(encoding="utf-8") == "x1\tabc\nx2\tca"
    assert res["sent-correct-f1"] == 100.0 and res["sent-detect-acc"] == 100.0


# ---- device-side build_batch, host half: the per-vocabulary table against Pinyin2 vectors -----------
def _pinyin_cases():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "pinyin_cases.json"), encoding="utf-8") as f:
        return json.load(f)


def test_pinyin_table_reproduces_reference_convert():
    from realise_amd.pinyin import PHO_INDEX, PinyinTable, token_pinyin
    c = _pinyin_cases()
    assert PHO_INDEX == c["pho_vocab"]
    tone3 = lambda ch: c["tone3"][ch]
    assert [token_pinyin(t, tone3) for t in c["tokens"]] == c["per_token"]
    tab = PinyinTable.build(c["tokens"], tone3)
    assert tab.table.shape == (len(c["tokens"]), 7) and tab.lens.min() >= 1
    for case in c["cases"]:
        pho_idx, lens = tab.convert(case["src_idx"])
        assert lens == case["pho_lens"]
        assert pho_idx.tolist() == case["pho_idx"]            # same width (batch max) and 0-padding as pad_sequence


def test_evaluate_plumbing_writes_label_files_and_scores(tmp_path):
    """run.py:239-280 end to end on the host side: features -> batches -> (stub) model -> decode -> preds.txt / labels.txt ->
    sentence scores.  The stub returns logits whose arg-max is the target except one planted error."""
    from realise_amd.trainer import evaluate
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]"] + list("abcdefgh")
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    tok = {c: i for i, c in enumerate(vocab)}

    def item(sid, src, tgt):
        ids = lambda s: [2] + [tok[c] for c in s] + [3]
        return {"id": sid, "src": src, "tgt": tgt, "tokens_size": [1] * len(src), "lengths": len(src), "src_idx": ids(src), "tgt_idx": ids(tgt)}

    items = [item("s1", "abca", "abcd"), item("s2", "efgh", "efgh"), item("s3", "aabb", "abab")]
    (tmp_path / "gold.txt").write_text("s1, 4, d\ns2, 0\ns3, 2, b, 3, a", encoding="utf-8")

    class Stub(torch.nn.Module):
        @staticmethod
        def build_batch(batch, tokenizer=None):
            return batch

        def forward(self, batch):
            tgt = batch["tgt_idx"].clone()
            if "s3" in batch["id"]:
                tgt[batch["id"].index("s3"), 2] = tok["h"]          # planted wrong correction at position 2 of s3
            logits = torch.nn.functional.one_hot(tgt, len(vocab)).float()
            return torch.tensor(0.5), logits

        def decode(self, logits):
            return logits.argmax(-1)

    loss, preds, res = evaluate(Stub(), items, batch_size=2, max_seq_length=8, vocab_path=str(tmp_path), label_path=str(tmp_path / "gold.txt"),
                                output_dir=str(tmp_path / "out"))
    assert loss == pytest.approx(0.5) and tuple(preds.shape) == (3, 8)
    assert (tmp_path / "out" / "labels.txt").read_text(encoding="utf-8") == "s1, 4, d\ns2, 0\ns3, 2, h, 3, a"
    assert (tmp_path / "out" / "preds.txt").read_text(encoding="utf-8") == "s1\tabcd\ns2\tefgh\ns3\tahab"
    assert res["sent-detect-f1"] == pytest.approx(100.0)             # positions all found
    assert res["sent-correct-p"] == pytest.approx(50.0) and res["sent-correct-r"] == pytest.approx(50.0)
    assert res["sent-correct-acc"] == pytest.approx(100.0 * 2 / 3)


def test_label_reader_keeps_blank_lines_and_trailing_comma_lines(tmp_path):
    """a blank line is an unnamed sentence without edits (it raised IndexError), ``id, `` / ``id,`` mean: no edits"""
    from realise_amd.metric import read_label_file
    f = tmp_path / "labels.txt"
    f.write_text("A1, 3, x, 1, y\n\nA2, 0\nA3, \nA4,\n", encoding="utf-8")
    assert read_label_file(str(f)) == [("A1", [(1, "y"), (3, "x")]), ("", []), ("A2", []), ("A3", []), ("A4", [])]
