"""Per-vocabulary pinyin table for the device-side ``build_batch``.

The reference converts every character of every batch on the host: ``build_batch`` (src/models.py:797-804) maps ids to
tokens and calls ``Pinyin2.convert`` (src/utils.py:58-99), one pypinyin call per character.  The conversion is a pure
function of the token, so it is done ONCE per vocabulary entry here; a batch is then a device gather
(``realise_build_pho``): ``pho_idx = table[src_idx]``, ``pho_lens = lens[src_idx]``.

Alphabet and encoding as in the reference: ``'P'`` = 0 (padding), tones ``'1'..'5'`` = 1-5, ``'a'..'z'`` = 6-31,
``'U'`` = 32 (no pinyin: multi-character tokens such as ``[CLS]`` / ``[PAD]`` / word pieces, and characters pypinyin
does not know).  A syllable is written tone FIRST (``zhong1`` -> ``1zhong``), at most 7 symbols.
"""
import numpy as np

PHO_VOCAB = ["P"] + [chr(x) for x in range(ord("1"), ord("5") + 1)] + [chr(x) for x in range(ord("a"), ord("z") + 1)] + ["U"]
PHO_INDEX = {c: i for i, c in enumerate(PHO_VOCAB)}
MAX_LEN = 7


def pypinyin_tone3(c):
    """The reference's pypinyin call (utils.py:78-83); needs the ``pypinyin`` package."""
    import pypinyin
    return pypinyin.pinyin(c, style=pypinyin.Style.TONE3, neutral_tone_with_five=True, errors=lambda x: ["U" for _ in x])[0][0]


DEFAULT_TONE3 = pypinyin_tone3      # hosts without pypinyin (and tests) may point this at another str -> TONE3 function


def token_pinyin(token, tone3=None):
    """Pinyin2.get_pinyin (utils.py:74-90): 'U' for multi-character tokens and unknown characters, else tone-first."""
    if len(token) > 1:
        return "U"
    s = (tone3 or DEFAULT_TONE3)(token)
    if s == "U":
        return s
    if not isinstance(s, str) or s[-1] not in "12345":
        raise AssertionError("unexpected pinyin %r for %r" % (s, token))
    return s[-1] + s[:-1]


class PinyinTable:
    """``table`` int64 [V, 7] (0-padded symbol ids) and ``lens`` int32 [V]."""

    def __init__(self, table, lens):
        self.table = np.ascontiguousarray(table, dtype=np.int64)
        self.lens = np.ascontiguousarray(lens, dtype=np.int32)
        if self.table.ndim != 2 or self.table.shape[0] != self.lens.shape[0] or self.lens.min() < 1 or self.lens.max() > self.table.shape[1]:
            raise ValueError("inconsistent pinyin table")

    @classmethod
    def build(cls, tokens, tone3=None):
        """tokens: the vocabulary in id order (``tokenizer.convert_ids_to_tokens(range(V))``)."""
        table = np.zeros((len(tokens), MAX_LEN), np.int64)
        lens = np.zeros(len(tokens), np.int32)
        for i, tok in enumerate(tokens):
            s = token_pinyin(tok, tone3)
            if len(s) > MAX_LEN:
                raise ValueError("pinyin %r of %r is longer than %d symbols" % (s, tok, MAX_LEN))
            table[i, :len(s)] = [PHO_INDEX[ch] for ch in s]
            lens[i] = len(s)
        return cls(table, lens)

    def convert(self, src_idx):
        """Host form of the gather, shaped like Pinyin2.convert's result: (pho_idx [n, max len in batch], list of lens)."""
        ids = np.asarray(src_idx).reshape(-1)
        lens = self.lens[ids]
        return self.table[ids][:, :int(lens.max())], lens.tolist()
