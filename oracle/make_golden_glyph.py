"""Glyph-table fixtures from the UPSTREAM REFERENCE's own renderer, run in this container (TEST INFRASTRUCTURE).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_glyph.py

The reference's fonts (simhei.ttf / xiaozhuan.ttf) and OpenCC are not in the tree, so its methods
(src/models.py:703-795) are driven with TTFs that exist in this image (DejaVu) through a scratch directory whose
``simhei.ttf`` / ``xiaozhuan.ttf`` are symlinks to them, and with a stand-in simplified->traditional converter
(``str.swapcase`` - it only has to change which glyph is drawn).  The vocabulary is realise_amd.data.synthetic_vocab.
Stored: sampled rows of the tables the reference's methods produce, per font.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from realise_amd.data import synthetic_vocab          # noqa: E402
from _ref_import import import_reference              # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FONT_A = "/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf"
FONT_B = "/usr/share/fonts/truetype/dejavu/DejaVuSerif.ttf"
ROWS = np.concatenate([np.arange(100, 240, 3), np.arange(240, 21128, 977)]).astype(np.int64)


def main():
    models, _ = import_reference()

    class FakeCC:
        def __init__(self, cfg):
            assert cfg == "s2t.json"

        def convert(self, c):
            return c.swapcase()
    models.opencc.OpenCC = FakeCC
    vocab = synthetic_vocab()
    store = {"rows": ROWS}
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "vocab.txt"), "w", encoding="utf-8") as f:
            f.write("\n".join(vocab) + "\n")
        os.symlink(FONT_A, os.path.join(d, "simhei.ttf"))
        os.symlink(FONT_B, os.path.join(d, "xiaozhuan.ttf"))
        cwd = os.getcwd()
        os.chdir(d)                      # the reference opens the fonts by bare file name (models.py:738-742)
        try:
            cls = models.SpellBertPho2ResArch3
            for nf, trad in ((3, True), (2, False)):
                ns = types.SimpleNamespace()
                ns.char_images_multifonts = torch.nn.Parameter(torch.zeros(21128, nf, 32, 32), requires_grad=False)
                ns.build_glyce_embed_onefont = lambda **kw: cls.build_glyce_embed_onefont(ns, **kw)
                cls.build_glyce_embed_multifonts(ns, d, nf, trad)
                t = ns.char_images_multifonts.data
                store["multi%d_trad%d" % (nf, int(trad))] = t[torch.from_numpy(ROWS)].numpy()
                store["multi%d_trad%d/sum" % (nf, int(trad))] = np.float64(t.double().sum().item())
            ns = types.SimpleNamespace()
            ns.char_images = torch.nn.Embedding(21128, 1024)
            cls.build_glyce_embed(ns, d, os.path.join(d, "simhei.ttf"))
            t = ns.char_images.weight.data
            store["single"] = t[torch.from_numpy(ROWS)].numpy()
            store["single/sum"] = np.float64(t.double().sum().item())
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "glyph_render.npz"), **store)
    print("wrote glyph_render.npz:", {k: getattr(v, "shape", v) for k, v in store.items()})


if __name__ == "__main__":
    main()
