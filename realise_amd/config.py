"""Model configuration for the ReaLiSe hot path.

Mirrors the fields of the reference's ``BertConfig`` that the path reads
(transformers/configuration_bert.py:83-116) plus the attributes src/run.py
pokes onto it (run.py:418-425).  A plain dict subclass so it serialises to the
same ``config.json`` the reference writes.
"""
import copy
import json
import os


class RealiseConfig(dict):
    DEFAULTS = dict(
        vocab_size=21128,
        hidden_size=768,
        num_hidden_layers=12,
        num_attention_heads=12,
        intermediate_size=3072,
        hidden_act="gelu",
        hidden_dropout_prob=0.1,
        attention_probs_dropout_prob=0.1,
        max_position_embeddings=512,
        type_vocab_size=2,
        initializer_range=0.02,
        layer_norm_eps=1e-12,
        # run.py:421-425
        image_model_type=0,
        num_fonts=3,
        # hard-wired sub-encoder depths (src/models.py:671,692)
        pho_layers=4,
        out_layers=3,
        pho_vocab_size=33,          # src/utils.py:61-67
        glyph_size=32,
    )

    def __init__(self, **kw):
        super().__init__(copy.deepcopy(self.DEFAULTS))
        self.update(kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def validate(self):
        if self["hidden_size"] % self["num_attention_heads"] != 0:
            raise ValueError("hidden size must be a multiple of the head count (modeling_bert.py:199-202)")
        if self["hidden_size"] // self["num_attention_heads"] != 64:
            raise ValueError("the HIP attention kernels are built for head_dim 64")
        if self["hidden_act"] != "gelu":
            raise ValueError("only the erf GELU of the reference path is implemented")
        if self["image_model_type"] != 0:
            raise NotImplementedError("invalid image_model_type %d" % self["image_model_type"])

    # config.json round trip (transformers/configuration_utils.py:204-227)
    def save_pretrained(self, d):
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump(dict(self), f, indent=2, sort_keys=True)

    @classmethod
    def from_pretrained(cls, d, **kw):
        path = os.path.join(d, "config.json") if os.path.isdir(d) else d
        with open(path) as f:
            c = cls(**{k: v for k, v in json.load(f).items()})
        c.update(kw)
        return c
