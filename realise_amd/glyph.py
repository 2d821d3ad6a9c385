"""Glyph table builder: vocabulary -> normalised 32x32 font bitmaps.

Behavioural restatement of the reference's ``build_glyce_embed`` / ``build_glyce_embed_multifonts`` /
``build_glyce_embed_onefont`` (src/models.py:703-795), written as one vectorisable function:

* one image per vocabulary entry, in ``vocab.txt`` order; the bitmap of a token is PIL's ``ImageFont.getmask(token)``
  (8-bit coverage, rows x columns = ``mask.size[::-1]``), cropped to the top-left ``size x size`` window and then pasted
  CENTRED into a zero ``size x size`` canvas (the reference's ``image.size != (font_size, font_size)`` compares an int with a
  tuple, so its padding branch always runs - models.py:721-728, 777-784);
* blank (all-zero) rows: tokens longer than one character (multi-font path, models.py:772-774) and additionally every
  non-CJK character on the single-font path (models.py:713-715);
* the whole per-font table is standardised with ITS OWN mean / std: ``(x - mean) / std`` (models.py:731, 787);
* multi-font: fonts ``simhei.ttf``, ``xiaozhuan.ttf``, ``simhei.ttf`` (traditional forms) cut to ``num_fonts``; with
  ``use_traditional_font`` the last slot is replaced by ``simhei.ttf`` rendering the OpenCC ``s2t`` form of every
  single-character token (models.py:737-758, 766-767).

The fonts (``simhei.ttf`` / ``xiaozhuan.ttf``) and OpenCC are not part of the reference tree (.MISSING_LARGE_BLOBS); font
paths and the simplified->traditional converter are therefore parameters with the reference's values as defaults.
"""
import os

import numpy as np

REFERENCE_FONTS = (("simhei.ttf", False), ("xiaozhuan.ttf", False), ("simhei.ttf", True))      # models.py:738-742


def is_cjk(cp):
    """code point inside the CJK blocks BERT's tokenizer treats as Chinese characters (models.py:20-31)"""
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def read_vocab(vocab_dir):
    """``vocab.txt`` of a pretrained directory, one stripped token per line (models.py:705-706, 762-764)."""
    with open(os.path.join(vocab_dir, "vocab.txt"), "r", encoding="utf-8") as f:
        return [line.strip() for line in f]


def glyph_bitmap(font, token, size):
    """centred ``size x size`` float32 coverage bitmap of one token"""
    mask = font.getmask(token)
    w, h = mask.size
    canvas = np.zeros((size, size), np.float32)
    if w == 0 or h == 0:
        return canvas
    img = np.asarray(mask, dtype=np.float32).reshape(h, w)[:size, :size]
    r0, c0 = (size - img.shape[0]) // 2, (size - img.shape[1]) // 2
    canvas[r0:r0 + img.shape[0], c0:c0 + img.shape[1]] = img
    return canvas


def render_font_table(vocab, font_path, size=32, cjk_only=False, to_traditional=None):
    """[len(vocab), size, size] float32, standardised over the whole table.

    ``cjk_only``: single-font rule (blank unless exactly one CJK character); otherwise only multi-character tokens are
    blank.  ``to_traditional``: callable applied to single-character tokens before rendering (OpenCC ``s2t``)."""
    from PIL import ImageFont
    font = ImageFont.truetype(font_path, size=size)
    out = np.zeros((len(vocab), size, size), np.float32)
    for i, tok in enumerate(vocab):
        if to_traditional is not None and len(tok) == 1:
            tok = to_traditional(tok)
        if len(tok) > 1 or (cjk_only and (len(tok) != 1 or not is_cjk(ord(tok)))):
            continue
        if len(tok) == 0:
            continue
        out[i] = glyph_bitmap(font, tok, size)
    std = out.std()
    if std == 0:
        raise ValueError("font %s renders every token of the vocabulary blank" % font_path)
    return ((out - out.mean()) / std).astype(np.float32)


def _opencc_s2t():
    try:
        import opencc
    except ImportError as e:     # pragma: no cover - depends on the host
        raise RuntimeError("use_traditional_font needs OpenCC ('s2t.json'); pass to_traditional= a str -> str callable instead") from e
    return opencc.OpenCC("s2t.json").convert


def render_multifont_table(vocab, num_fonts, use_traditional_font, size=32, font_paths=None, to_traditional=None, font_dir=""):
    """[len(vocab), num_fonts, size, size] float32 as ``build_glyce_embed_multifonts`` stores it (models.py:737-758)."""
    fonts = list(font_paths) if font_paths is not None else list(REFERENCE_FONTS)
    fonts = [(f, False) if isinstance(f, str) else tuple(f) for f in fonts][:num_fonts]
    if use_traditional_font:
        fonts = fonts[:-1] + [(REFERENCE_FONTS[2][0] if font_paths is None else fonts[-1][0], True)]
    if len(fonts) != num_fonts:
        raise ValueError("need %d fonts, have %d" % (num_fonts, len(fonts)))
    conv = None
    if any(trad for _, trad in fonts):
        conv = to_traditional or _opencc_s2t()
    tables = [render_font_table(vocab, os.path.join(font_dir, path), size, cjk_only=False, to_traditional=conv if trad else None)
              for path, trad in fonts]
    return np.ascontiguousarray(np.stack(tables, axis=1))
