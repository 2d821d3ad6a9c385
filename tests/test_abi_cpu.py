"""CPU checks of the boundary: the C-ABI library loads and exports every symbol the header declares;
the parameter layout reproduces the reference's state_dict; the module shell refuses to run without a GPU."""
import os
import re

import pytest
import torch

from realise_amd import _capi
from realise_amd.config import RealiseConfig
from realise_amd.init import tensor_specs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    header = "".join(open(os.path.join(ROOT, "include", f)).read() for f in sorted(os.listdir(os.path.join(ROOT, "include"))) if f.endswith(".h"))
    main = open(os.path.join(ROOT, "include", "realise_hip.h")).read()
    assert "realise_set_" not in main and "realise_profile_" not in main, "diagnostic knobs belong in realise_hip_debug.h"
    declared = set(re.findall(r"\b(realise_[a-z0-9_]+)\s*\(", header))
    declared -= {"realise_engine"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    assert b"gfx950" in lib.realise_version()


@pytest.mark.parametrize("model_type,count", [("arch3", 427), ("bert", 201)])
def test_layout_matches_reference_state_dict(model_type, count):
    cfg = RealiseConfig()
    c = _capi.make_config(cfg, model_type, _capi.BF16)
    entries, sizes, buckets = _capi.layout(c)
    specs = {n: tuple(s) for n, s, k in tensor_specs(cfg, model_type)}
    assert len(entries) == count and {e[0] for e in entries} == set(specs)
    for name, arena, off, shape in entries:
        assert shape == specs[name], name
        assert off % 64 == 0
    # buckets tile the trainable arena in order
    assert buckets[0][0] == 0 and buckets[-1][1] == sizes[0]
    for (a0, a1), (b0, b1) in zip(buckets, buckets[1:]):
        assert a1 == b0 and a0 < a1
    # q/k/v weights adjacent (fused [3H,H] projection)
    d = {e[0]: e for e in entries}
    q, k, v = (d["bert.encoder.layer.0.attention.self.%s.weight" % n][2] for n in ("query", "key", "value"))
    assert k - q == 768 * 768 and v - k == 768 * 768
    if model_type == "arch3":
        # 34.2 M never-used parameters live in their own arena and get no gradient
        assert sizes[1] >= 2 * 21128 * 768 + 3 * 768 * 768


def test_module_shell_contract_without_gpu():
    from realise_amd.data import synthetic_batch
    from realise_amd.modeling import MODEL_CLASSES, SpellBert
    assert set(MODEL_CLASSES) == {"bert", "bert-pho2-res-arch3"}
    cfg = RealiseConfig(num_hidden_layers=1)
    m = SpellBert(cfg, compute_dtype="fp32")
    sd = m.state_dict()
    assert set(sd) == {n for n, _, _ in tensor_specs(cfg, "bert")}
    assert m.classifier.weight is m.bert.embeddings.word_embeddings.weight          # tie_cls_weight
    m.tie_cls_weight()
    no_decay = ["bias", "LayerNorm.weight"]                                         # run.py:146-151 split works
    assert any(any(nd in n for nd in no_decay) for n, _ in m.named_parameters())
    sd2 = {k: v.clone() + 1.0 if v.is_floating_point() else v.clone() for k, v in sd.items()}
    m.load_state_dict(sd2)
    assert torch.allclose(m.bert.embeddings.LayerNorm.bias, sd2["bert.embeddings.LayerNorm.bias"])
    with pytest.raises(_capi.RealiseHipError):
        m(synthetic_batch(2, 8, with_pho=False))                                     # no CPU fallback, fails loudly
