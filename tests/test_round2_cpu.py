"""CPU tests added in round 2: the drop-in boundary (build_batch, glyph builders, checkpoint I/O, make_features), the
full-size goldens against the oracle, the SIGHAN13 post-filter, optimizer state round trip and the bf16 gradient buckets."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import realise_ref as R
from helpers import check_summary, golden_case_inputs, load_golden, oracle_state_dict
from realise_amd import glyph, pinyin
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch, synthetic_vocab
from realise_amd.modeling import SpellBert, SpellBertPho2ResArch3, pinyin_table_for
from realise_amd.trainer import make_features

FONT_A = "/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf"
FONT_B = "/usr/share/fonts/truetype/dejavu/DejaVuSerif.ttf"


class StubTokenizer:
    """the two members of BertTokenizer the trainer path touches (tokenization_utils.py:1126-1146)"""

    def __init__(self, vocab):
        self.vocab = {t: i for i, t in enumerate(vocab)}
        self.ids_to_tokens = list(vocab)
        self.vocab_size = len(vocab)

    def convert_ids_to_tokens(self, ids):
        return [self.ids_to_tokens[i] for i in ids]


def fake_tone3(c):
    """stand-in for pypinyin (not installed here): a deterministic TONE3-style syllable for CJK characters, 'U' otherwise"""
    cp = ord(c)
    if not glyph.is_cjk(cp):
        return "U"
    syl = ["a", "zhong", "shuang", "yi", "lve", "er"][cp % 6]
    return syl + str(cp % 5 + 1)


# ------------------------------------------------------------------------------------------------ glyph table builder
needs_fonts = pytest.mark.skipif(not (os.path.exists(FONT_A) and os.path.exists(FONT_B)), reason="DejaVu fonts not installed")


@needs_fonts
def test_glyph_renderer_matches_reference_tables(golden_dir):
    """fixture: the reference's own build_glyce_embed / build_glyce_embed_multifonts / _onefont run on these fonts
    (oracle/make_golden_glyph.py); bit-exact rows"""
    pytest.importorskip("PIL")
    g = dict(np.load(os.path.join(golden_dir, "glyph_render.npz")))
    rows = g["rows"]
    vocab = synthetic_vocab()
    m3 = glyph.render_multifont_table(vocab, 3, True, font_paths=[FONT_A, FONT_B, FONT_A], to_traditional=str.swapcase)
    assert m3.shape == (21128, 3, 32, 32)
    assert np.array_equal(m3[rows], g["multi3_trad1"])
    assert abs(float(m3.astype(np.float64).sum()) - float(g["multi3_trad1/sum"])) < 1e-3
    m2 = glyph.render_multifont_table(vocab, 2, False, font_paths=[FONT_A, FONT_B])
    assert np.array_equal(m2[rows], g["multi2_trad0"])
    s1 = glyph.render_font_table(vocab, FONT_A, cjk_only=True).reshape(len(vocab), -1)
    assert np.array_equal(s1[rows], g["single"])
    # blank rules: multi-character tokens are constant rows in both paths; non-CJK single characters only in the single-font path
    assert all(np.ptp(m3[0, f]) == 0.0 for f in range(3)) and np.ptp(s1[0]) == 0.0               # [PAD]
    a_id = vocab.index("A")
    assert np.ptp(m3[a_id, 0]) > 0.0 and np.ptp(s1[a_id]) == 0.0


@needs_fonts
def test_build_glyce_embed_reference_signatures(tmp_path):
    pytest.importorskip("PIL")
    vocab = synthetic_vocab()
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    cfg = RealiseConfig(num_hidden_layers=1)
    m = SpellBertPho2ResArch3(cfg)
    m.build_glyce_embed_multifonts(str(tmp_path), 3, True, font_paths=[FONT_A, FONT_B, FONT_A], to_traditional=str.swapcase)   # run.py:436-440
    want = glyph.render_multifont_table(vocab, 3, True, font_paths=[FONT_A, FONT_B, FONT_A], to_traditional=str.swapcase)
    assert np.array_equal(m.char_images_multifonts.detach().numpy(), want)
    assert not m.char_images_multifonts.requires_grad
    with pytest.raises(TypeError):
        m.build_glyce_embed_multifonts(want)                  # a table is not a vocab_dir: set_glyph_table is the entry for that
    m.set_glyph_table(want * 0.5)
    assert np.allclose(m.char_images_multifonts.detach().numpy(), want * 0.5)
    # single-font model: nn.Embedding-style key, build_glyce_embed(vocab_dir, font_path) (run.py:433-435)
    m1 = SpellBertPho2ResArch3(RealiseConfig(num_hidden_layers=1, num_fonts=1))
    assert "char_images.weight" in m1.state_dict() and m1.state_dict()["char_images.weight"].shape == (21128, 1024)
    m1.build_glyce_embed(str(tmp_path), FONT_A)
    assert np.array_equal(m1.state_dict()["char_images.weight"].numpy(), glyph.render_font_table(vocab, FONT_A, cjk_only=True).reshape(21128, -1))
    with pytest.raises(RuntimeError):
        m.build_glyce_embed(str(tmp_path), FONT_A)


# ------------------------------------------------------------------------------------------------ build_batch / make_features
def test_build_batch_adds_pinyin_like_the_reference(golden_dir, monkeypatch):
    """models.py:797-804: pho_idx [B*S, max len] long + host list pho_lens (the Pinyin2 encoding itself is pinned to the
    reference's vectors in tests/test_host_logic.py; pypinyin is not installed, so a TONE3 stand-in feeds it)"""
    vocab = synthetic_vocab(400)
    tok = StubTokenizer(vocab)
    monkeypatch.setattr(pinyin, "DEFAULT_TONE3", fake_tone3)
    src = torch.tensor([[101, 300, 301, 150, 102, 0, 0, 0], [101, 399, 102, 0, 0, 0, 0, 0]])
    batch = {"src_idx": src.clone(), "masks": (src != 0).long()}
    out = SpellBertPho2ResArch3.build_batch(batch, tok)
    assert out is batch and out["pho_idx"].dtype == torch.long and out["pho_idx"].shape[0] == src.numel()
    assert isinstance(out["pho_lens"], list) and len(out["pho_lens"]) == src.numel()
    flat = src.reshape(-1).tolist()
    for i, t in enumerate(flat):
        s = pinyin.token_pinyin(vocab[t], fake_tone3)
        assert out["pho_lens"][i] == len(s)
        assert out["pho_idx"][i, :len(s)].tolist() == [pinyin.PHO_INDEX[ch] for ch in s]
        assert out["pho_idx"][i, len(s):].abs().sum().item() == 0
    assert out["pho_idx"].shape[1] == max(out["pho_lens"])           # padded to the batch maximum, as pad_sequence does
    assert pinyin_table_for(tok) is pinyin_table_for(tok)            # built once per tokenizer
    b2 = {"src_idx": src.clone()}
    assert SpellBert.build_batch(b2, tok) is b2 and "pho_idx" not in b2      # SpellBert.build_batch is the identity (models.py:46-48)


def test_make_features_truncates_and_clamps_like_run_py():
    ex = [{"id": 0, "src": "a", "tgt": "a", "tokens_size": [1] * 5, "lengths": 5, "src_idx": [101, 1, 2, 3, 4, 5, 102], "tgt_idx": [101, 1, 2, 3, 9, 5, 102]},
          {"id": 1, "src": "b", "tgt": "b", "tokens_size": [1] * 9, "lengths": 9, "src_idx": [101] + list(range(1, 10)) + [102],
           "tgt_idx": [101] + list(range(1, 10)) + [102]}]
    b = make_features(ex, 8)
    assert b["src_idx"].tolist() == [[101, 1, 2, 3, 4, 5, 102, 0], [101, 1, 2, 3, 4, 5, 6, 7]]          # run.py:79 truncation
    assert b["tgt_idx"].tolist()[0] == [101, 1, 2, 3, 9, 5, 102, 0]
    assert b["masks"].tolist() == [[1, 1, 1, 1, 1, 1, 1, 0], [1] * 8]
    assert b["loss_masks"].tolist() == [[0, 1, 1, 1, 1, 1, 0, 0], [0, 1, 1, 1, 1, 1, 1, 1]]            # min(1 + lengths, max_length)
    assert b["lengths"] == [5, 9] and b["id"] == [0, 1]


def test_remove_de_matches_reference(golden_dir, tmp_path):
    from realise_amd.metric import drop_de_corrections
    with open(os.path.join(golden_dir, "remove_de_cases.json"), encoding="utf-8") as f:
        cases = json.load(f)
    for k, c in enumerate(cases):
        a, b = tmp_path / ("in%d" % k), tmp_path / ("out%d" % k)
        a.write_text(c["input"], encoding="utf-8")
        drop_de_corrections(str(a), str(b))
        assert b.read_text(encoding="utf-8") == c["output"], k


# ------------------------------------------------------------------------------------------------ checkpoint I/O
def test_save_pretrained_from_pretrained_round_trip(tmp_path):
    cfg = RealiseConfig(num_hidden_layers=2)
    m = SpellBertPho2ResArch3(cfg, seed=3, init_scheme="perturbed")
    with torch.no_grad():
        m.state_dict()["resnet.res_block2.residual_function.1.num_batches_tracked"].fill_(17)
    m.save_pretrained(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["config.json", "pytorch_model.bin"]
    m2 = SpellBertPho2ResArch3.from_pretrained(str(tmp_path))
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert list(sd.keys()) == list(sd2.keys())
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k
    assert sd2["classifier.weight"].data_ptr() == sd2["bert.embeddings.word_embeddings.weight"].data_ptr()     # still tied
    assert int(sd2["resnet.res_block2.residual_function.1.num_batches_tracked"]) == 17
    assert m2.config.num_fonts == 3 and m2.config.num_hidden_layers == 2


def test_from_pretrained_loads_a_reference_written_checkpoint(tmp_path):
    """a pytorch_model.bin written by the REFERENCE's save_pretrained (transformers/modeling_utils.py:236-251) loads unchanged"""
    from _ref_import import import_reference, reference_available
    if not reference_available():
        pytest.skip("reference tree not present")
    models, BertConfig = import_reference()
    bc = BertConfig(vocab_size_or_config_json_file=21128)
    bc.num_hidden_layers = 1
    bc.image_model_type = 0
    bc.num_fonts = 3
    torch.manual_seed(5)
    ref = models.SpellBertPho2ResArch3(bc)
    ref.tie_cls_weight()
    ref.save_pretrained(str(tmp_path))
    cfg = RealiseConfig(num_hidden_layers=1)
    ours = SpellBertPho2ResArch3.from_pretrained(str(tmp_path), config=cfg)
    rsd, osd = ref.state_dict(), ours.state_dict()
    assert set(rsd.keys()) == set(osd.keys())
    for k, v in rsd.items():
        assert torch.equal(v, osd[k]), k


# ------------------------------------------------------------------------------------------------ full-size goldens vs oracle
def test_oracle_matches_reference_at_full_config2_size(golden_dir):
    """arch3_b64s128_eval: the reference's eval forward on the B=64, S=128 batch (BASELINE configs[1]); the oracle gives the
    same arg-max ids wherever the reference's top-1/top-2 margin exceeds fp32 noise, and the same sampled logits"""
    g = load_golden(golden_dir, "arch3_b64s128_eval")
    cfg, sd_np, batch = golden_case_inputs(g, "arch3")
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    taps = {}
    with torch.no_grad():
        loss, logits = R.arch3_forward(oracle_state_dict(sd_np), cfg, batch, training=False, taps=taps)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    flat = logits.reshape(-1, logits.shape[-1])
    tok = torch.arange(flat.shape[0])
    mine = flat[tok, torch.from_numpy(g["sample_slot"].astype(np.int64))].numpy()
    assert np.abs(mine - g["sample_logit"]).max() < 5e-5
    am = logits.argmax(-1).numpy().astype(np.int32)
    decided = g["margin"] > 1e-4
    assert np.array_equal(am[decided], g["argmax"][decided])
    assert (~decided).sum() < 16
    for k in ("bert_h", "pho_gru", "pho_h", "res", "res_h", "out"):
        check_summary(g, "tap/" + k, taps[k], 5e-5, what="tap")


def test_oracle_resnet_eval_rows_match_config4_fixture(golden_dir):
    """resnet_b256s128 (BASELINE configs[3]): eval-mode BatchNorm is per sample, so the oracle is checked on the fixture's sampled
    tokens only (the train-mode statistics over all 32768 stacks are checked on the GPU, tests/test_round2_gpu.py)"""
    g = load_golden(golden_dir, "resnet_b256s128")
    B, S, seed = int(g["meta/B"]), int(g["meta/S"]), int(g["meta/seed"])
    cfg = RealiseConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    from realise_amd.init import init_state_dict_numpy
    sd = oracle_state_dict(init_state_dict_numpy(cfg, "arch3", seed=seed, scheme="perturbed"))
    ids = synthetic_batch(B, S, seed=seed, with_pho=False)["src_idx"].view(-1)
    rows = torch.from_numpy(g["eval/rows"].astype(np.int64))
    with torch.no_grad():
        res = R.char_resnet(sd, sd["char_images_multifonts"][ids[rows]], training=False)
    assert (res - torch.from_numpy(g["eval/res_rows"])).abs().max().item() < 2e-5


# ------------------------------------------------------------------------------------------------ optimizer state / DDP
def test_fused_adamw_state_dict_round_trip():
    from realise_amd.optim import FusedAdamW
    cfg = RealiseConfig(num_hidden_layers=1)
    m = SpellBert(cfg, compute_dtype="fp32")
    opt = FusedAdamW(m, lr=1e-3)
    opt._m.normal_()
    opt._v.uniform_()
    opt._step = 41
    sd = opt.state_dict()
    assert "realise_flat" in sd and sd["realise_flat"]["step"] == 41
    opt2 = FusedAdamW(m, lr=5e-4)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2._m, opt._m) and torch.equal(opt2._v, opt._v) and opt2._step == 41
    assert opt2.param_groups[0]["lr"] == 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bf16_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from realise_amd.ddp import DistributedDataParallel
        m = SpellBert(RealiseConfig(num_hidden_layers=1), compute_dtype="fp32", seed=rank)
        DistributedDataParallel(m, grad_dtype="bf16")
        gen = torch.Generator().manual_seed(100 + rank)
        grads = []
        for b in m.bucket_views():
            b.copy_(torch.randn(b.shape, generator=gen) * 1e-2)
            grads.append(b.clone())
        for i in range(len(grads)):
            m.grad_sync.bucket_ready(i)
        m.grad_sync.finish()
        # reference: fp32 mean over the ranks, recomputed from the seeds
        for i, b in enumerate(m.bucket_views()):
            want = torch.zeros_like(b)
            for r in range(world):
                gr = torch.Generator().manual_seed(100 + r)
                for j, bb in enumerate(m.bucket_views()):
                    t = torch.randn(bb.shape, generator=gr) * 1e-2
                    if j == i:
                        want += t / world
            err = (b - want).abs().max().item()
            assert err < 8e-4, (i, err)                  # bf16 keeps 8 mantissa bits: |g| up to ~6e-2 -> ulp 2.4e-4, three roundings
            assert err > 0.0                             # ... and the wire really was bf16
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_ddp_bf16_gradient_buckets_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
