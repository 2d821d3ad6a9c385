"""SIGHAN-shaped synthetic batches.

Produces exactly the dict ``make_features`` + ``build_batch`` hand to
``model(batch)`` in the reference trainer (src/run.py:68-101,
src/models.py:797-804): ``src_idx / tgt_idx / masks / loss_masks`` int64
[B,S], ``pho_idx`` int64 [B*S, Tp], ``pho_lens`` python list (host) of B*S ints,
plus the python-list keys the model ignores.
"""
import numpy as np
import torch

CLS, SEP, PAD = 101, 102, 0
PHO_UNK = 32  # 'U' in Pinyin2's vocab (src/utils.py:61-67)


def synthetic_pinyin_table(vocab_size=21128, seed=0, id_lo=670):
    """A per-vocabulary pinyin table with the statistics of the synthetic batches (tone first, 2..7 symbols per character, 'U' for
    the ids below ``id_lo``: specials, ASCII, word pieces): the stand-in for ``PinyinTable.build(tokenizer vocabulary)`` when neither
    vocab.txt nor pypinyin is at hand.  ``synthetic_batch(..., pinyin_table=t)`` and ``model.set_pinyin_table(t)`` then describe the
    same batch through the host and the device ``build_batch``."""
    from .pinyin import MAX_LEN, PinyinTable
    g = np.random.Generator(np.random.Philox(key=[0x91A71E, seed]))
    lens = g.integers(2, MAX_LEN + 1, size=vocab_size).astype(np.int32)
    table = g.integers(6, 32, size=(vocab_size, MAX_LEN)).astype(np.int64)
    table[:, 0] = g.integers(1, 6, size=vocab_size)
    table[np.arange(MAX_LEN)[None, :] >= lens[:, None]] = 0
    special = np.arange(vocab_size) < id_lo
    table[special] = 0
    table[special, 0] = PHO_UNK
    lens[special] = 1
    return PinyinTable(table, lens)


def synthetic_batch(batch_size, seq_len, vocab_size=21128, seed=0, with_pho=True, full_length=False,
                    id_lo=670, id_hi=7992, pinyin_table=None):
    g = np.random.Generator(np.random.Philox(key=[0x5EA115E, seed]))
    id_hi = min(id_hi, vocab_size)
    id_lo = min(id_lo, max(1, id_hi - 1))
    max_chars = seq_len - 2
    lo = max(1, min(32, max_chars // 4))
    src = np.zeros((batch_size, seq_len), np.int64)
    tgt = np.zeros((batch_size, seq_len), np.int64)
    masks = np.zeros((batch_size, seq_len), np.int64)
    loss_masks = np.zeros((batch_size, seq_len), np.int64)
    lengths = []
    for b in range(batch_size):
        L = max_chars if full_length else int(g.integers(lo, max_chars + 1))
        ids = g.integers(id_lo, id_hi, size=L)
        row = np.concatenate([[CLS], ids, [SEP]])
        src[b, :L + 2] = row
        t = row.copy()
        flip = g.random(L) < 0.02
        t[1:L + 1] = np.where(flip, g.integers(id_lo, id_hi, size=L), ids)
        tgt[b, :L + 2] = t
        masks[b, :L + 2] = 1
        loss_masks[b, 1:L + 1] = 1          # run.py:86-92
        lengths.append(L)
    batch = {
        "src_idx": torch.from_numpy(src), "tgt_idx": torch.from_numpy(tgt),
        "masks": torch.from_numpy(masks), "loss_masks": torch.from_numpy(loss_masks),
        "lengths": lengths, "tokens_size": [[1] * l for l in lengths],
        "id": list(range(batch_size)), "src": [""] * batch_size, "tgt": [""] * batch_size,
    }
    if with_pho and pinyin_table is not None:       # the reference's build_batch: pinyin as a function of the token (models.py:797-804)
        pho, lens = pinyin_table.convert(src)
        batch["pho_idx"] = torch.from_numpy(np.ascontiguousarray(pho))
        batch["pho_lens"] = [int(x) for x in lens]
    elif with_pho:
        flat = src.reshape(-1)
        is_char = (flat != CLS) & (flat != SEP) & (flat != PAD)
        n = flat.shape[0]
        lens = np.where(is_char, g.integers(2, 8, size=n), 1)
        tp = int(lens.max())
        pho = np.zeros((n, tp), np.int64)
        tone = g.integers(1, 6, size=n)
        letters = g.integers(6, 32, size=(n, tp))
        for t in range(tp):
            v = tone if t == 0 else letters[:, t]
            pho[:, t] = np.where(lens > t, v, 0)
        pho[~is_char, 0] = PHO_UNK
        batch["pho_idx"] = torch.from_numpy(pho)
        batch["pho_lens"] = [int(x) for x in lens]
    return batch


def synthetic_vocab(vocab_size=21128):
    """A BERT-style Chinese vocabulary stand-in (``vocab.txt`` of chinese-roberta-wwm-ext is not in the tree): the special
    tokens at their real ids ([PAD] 0, [UNK] 100, [CLS] 101, [SEP] 102, [MASK] 103), ``[unusedN]`` fillers, printable ASCII,
    a few ``##`` word pieces, then consecutive CJK Unified Ideographs from U+4E00.  Deterministic; used by the glyph /
    pinyin / trainer tests and their fixture generators."""
    toks = []
    special = {0: "[PAD]", 100: "[UNK]", 101: "[CLS]", 102: "[SEP]", 103: "[MASK]"}
    for i in range(106):
        toks.append(special.get(i, "[unused%d]" % i))
    toks += [chr(c) for c in range(33, 127)]                   # single printable ASCII characters
    toks += ["##%s" % chr(c) for c in range(97, 123)]          # word pieces: multi-character -> blank glyph, pinyin 'U'
    toks += ["the", "##ing", "2019"]
    cp = 0x4E00
    while len(toks) < vocab_size:
        toks.append(chr(cp))
        cp += 1
    return toks[:vocab_size]


def glyph_upstream_grad(rows, width=768, seed=0, scale=1e-3):
    """A seeded upstream gradient ``d res`` [rows, width] (fp32) for the glyph-ResNet-only backward (BASELINE configs[3]):
    the same array on the reference side (oracle/make_golden_full.py) and on the device side (tests, bench.py)."""
    g = np.random.Generator(np.random.Philox(key=[0x61F9, seed]))
    return (g.standard_normal((rows, width), dtype=np.float32) * np.float32(scale)).astype(np.float32)
