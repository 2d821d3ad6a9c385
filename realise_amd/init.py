"""Deterministic, torch-independent weight initialisation.

Weights are generated per tensor from a counter-based numpy generator keyed by
the tensor's state_dict name, so that this container (where the reference is
imported to make golden vectors) and the GPU box materialise bit-identical
weights without shipping a 1 GB checkpoint.

``scheme='reference'`` follows the reference's own initialisers
(transformers/modeling_bert.py:496-506 for Linear/Embedding/LayerNorm; PyTorch
defaults for Conv2d / BatchNorm2d / GRU, src/models.py:662-669, char_cnn.py).
``scheme='perturbed'`` additionally randomises biases, LayerNorm/BatchNorm
affine parameters and running statistics so parity tests exercise every term.
"""
import zlib

import numpy as np


def _rng(name, seed):
    key = (zlib.crc32(name.encode()) << 32) | (zlib.adler32(name.encode()) & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=[key & (2**64 - 1), seed & (2**64 - 1)]))


def synth_glyph_table(vocab_size, num_fonts, size=32, seed=0):
    """A stand-in for build_glyce_embed_multifonts (src/models.py:737-795):
    sparse {0,255} stroke bitmaps, constant rows for non-character ids, then
    (x - mean) / std over each font's table as at models.py:793.  The real
    fonts are absent from the tree (.MISSING_LARGE_BLOBS)."""
    out = np.empty((vocab_size, num_fonts, size, size), dtype=np.float32)
    for f in range(num_fonts):
        g = _rng("glyph_font_%d" % f, seed)
        img = np.zeros((vocab_size, size, size), dtype=np.float32)
        # a few horizontal / vertical strokes per glyph
        n_strokes = 6
        r0 = g.integers(2, size - 2, size=(vocab_size, n_strokes))
        c0 = g.integers(2, size - 10, size=(vocab_size, n_strokes))
        ln = g.integers(4, 10, size=(vocab_size, n_strokes))
        vert = g.integers(0, 2, size=(vocab_size, n_strokes))
        for s in range(n_strokes):
            for k in range(10):
                on = (k < ln[:, s])
                rr = np.where(vert[:, s] == 1, np.minimum(c0[:, s] + k, size - 1), r0[:, s])
                cc = np.where(vert[:, s] == 1, r0[:, s], np.minimum(c0[:, s] + k, size - 1))
                idx = np.nonzero(on)[0]
                img[idx, rr[idx], cc[idx]] = 255.0
        img[:670] = 0.0          # [PAD], [unused*], punctuation ... -> blank (len(char) > 1 rows)
        img = (img - img.mean()) / img.std()
        out[:, f] = img
    return out


def tensor_init(name, shape, kind, cfg, seed=0, scheme="reference"):
    """kind in {'normal','zeros','ones','conv','gru','glyph','bn_var','count'}"""
    g = _rng(name, seed)
    pert = scheme == "perturbed"
    std = cfg["initializer_range"]
    if kind == "normal":
        return (g.standard_normal(shape, dtype=np.float32) * std).astype(np.float32)
    if kind == "zeros":
        if pert:
            return (g.standard_normal(shape, dtype=np.float32) * std).astype(np.float32)
        return np.zeros(shape, np.float32)
    if kind == "ones":
        if pert:
            return (1.0 + 0.1 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        return np.ones(shape, np.float32)
    if kind == "bn_var":
        if pert:
            return (0.5 + g.random(shape, dtype=np.float32)).astype(np.float32)
        return np.ones(shape, np.float32)
    if kind == "conv":      # kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        return g.uniform(-b, b, shape).astype(np.float32)
    if kind == "gru":       # nn.GRU.reset_parameters: U(-1/sqrt(hidden), 1/sqrt(hidden))
        b = 1.0 / np.sqrt(cfg["hidden_size"])
        return g.uniform(-b, b, shape).astype(np.float32)
    if kind == "count":
        return np.zeros(shape, np.int64)
    raise ValueError(kind)


def tensor_specs(cfg, model_type="arch3"):
    """Ordered (name, shape, kind) for every state_dict entry of the reference
    model (names/shapes as probed from the reference)."""
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    P, TV = cfg["max_position_embeddings"], cfg["type_vocab_size"]
    specs = []

    def bert(prefix, n_layers):
        e = prefix + "embeddings."
        specs.append((e + "word_embeddings.weight", (V, H), "normal"))
        specs.append((e + "position_embeddings.weight", (P, H), "normal"))
        specs.append((e + "token_type_embeddings.weight", (TV, H), "normal"))
        specs.append((e + "LayerNorm.weight", (H,), "ones"))
        specs.append((e + "LayerNorm.bias", (H,), "zeros"))
        for i in range(n_layers):
            l = "%sencoder.layer.%d." % (prefix, i)
            for nm in ("query", "key", "value"):
                specs.append((l + "attention.self.%s.weight" % nm, (H, H), "normal"))
                specs.append((l + "attention.self.%s.bias" % nm, (H,), "zeros"))
            specs.append((l + "attention.output.dense.weight", (H, H), "normal"))
            specs.append((l + "attention.output.dense.bias", (H,), "zeros"))
            specs.append((l + "attention.output.LayerNorm.weight", (H,), "ones"))
            specs.append((l + "attention.output.LayerNorm.bias", (H,), "zeros"))
            specs.append((l + "intermediate.dense.weight", (I, H), "normal"))
            specs.append((l + "intermediate.dense.bias", (I,), "zeros"))
            specs.append((l + "output.dense.weight", (H, I), "normal"))
            specs.append((l + "output.dense.bias", (H,), "zeros"))
            specs.append((l + "output.LayerNorm.weight", (H,), "ones"))
            specs.append((l + "output.LayerNorm.bias", (H,), "zeros"))
        specs.append((prefix + "pooler.dense.weight", (H, H), "normal"))
        specs.append((prefix + "pooler.dense.bias", (H,), "zeros"))

    if model_type == "arch3":
        F_ = cfg["num_fonts"]
        if F_ == 1:      # models.py:674-676: the single-font model keeps the table as an nn.Embedding [V, 1024]
            specs.append(("char_images.weight", (V, cfg["glyph_size"] * cfg["glyph_size"]), "glyph"))
        else:
            specs.append(("char_images_multifonts", (V, F_, cfg["glyph_size"], cfg["glyph_size"]), "glyph"))
    bert("bert.", cfg["num_hidden_layers"])
    if model_type == "arch3":
        specs.append(("pho_embeddings.weight", (cfg["pho_vocab_size"], H), "normal"))
        specs.append(("pho_gru.weight_ih_l0", (3 * H, H), "gru"))
        specs.append(("pho_gru.weight_hh_l0", (3 * H, H), "gru"))
        specs.append(("pho_gru.bias_ih_l0", (3 * H,), "gru"))
        specs.append(("pho_gru.bias_hh_l0", (3 * H,), "gru"))
        bert("pho_model.", cfg["pho_layers"])
        chans = [cfg["num_fonts"], 64, 128, 256, 512, 768]
        for b in range(1, 6):
            ci, co = chans[b - 1], chans[b]
            p = "resnet.res_block%d." % b

            def bn(q):
                specs.append((q + "weight", (co,), "ones"))
                specs.append((q + "bias", (co,), "zeros"))
                specs.append((q + "running_mean", (co,), "zeros"))
                specs.append((q + "running_var", (co,), "bn_var"))
                specs.append((q + "num_batches_tracked", (), "count"))
            specs.append((p + "residual_function.0.weight", (co, ci, 3, 3), "conv"))
            bn(p + "residual_function.1.")
            specs.append((p + "residual_function.3.weight", (co, co, 3, 3), "conv"))
            bn(p + "residual_function.4.")
            specs.append((p + "shortcut.0.weight", (co, ci, 1, 1), "conv"))
            bn(p + "shortcut.1.")
        specs.append(("resnet_layernorm.weight", (H,), "ones"))
        specs.append(("resnet_layernorm.bias", (H,), "zeros"))
        specs.append(("gate_net.weight", (3, 4 * H), "normal"))
        specs.append(("gate_net.bias", (3,), "zeros"))
        bert("output_block.", cfg["out_layers"])
    specs.append(("classifier.weight", (V, H), "normal"))
    specs.append(("classifier.bias", (V,), "zeros"))
    return specs


def init_state_dict_numpy(cfg, model_type="arch3", seed=0, scheme="reference", tie=True):
    sd = {}
    glyph = None
    for name, shape, kind in tensor_specs(cfg, model_type):
        if kind == "glyph":
            if glyph is None:
                glyph = synth_glyph_table(shape[0], cfg["num_fonts"], cfg["glyph_size"], seed).reshape(shape)
            sd[name] = glyph
        else:
            sd[name] = tensor_init(name, shape, kind, cfg, seed, scheme)
    if tie:
        sd["classifier.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    return sd
