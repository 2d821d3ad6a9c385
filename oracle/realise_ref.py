"""CPU restatement (fp32, plain PyTorch ops) of the ReaLiSe hot path.

TEST INFRASTRUCTURE.  This module is the *checker*: it is imported only by
tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg.  The
product path (realise_amd/) never imports it and fails loudly when the HIP
library is missing.

Every function states the math of one reference call site (file:line relative
to /root/reference) in functional form over a plain ``dict[str, Tensor]`` that
uses the reference's own state_dict key names, so the same weights drive the
reference (in this container, via oracle/_ref_import.py), this restatement, and
the HIP engine.

Parity pin: the reference holds no golden tensors for this path (SURVEY.md
section 4 / 8c); the restatement is pinned against outputs of the reference
itself, generated here by oracle/make_golden.py and committed under
tests/golden/ (see tests/test_oracle_golden.py), and against the reference run
live when /root/reference is present (tests/test_oracle_vs_reference.py).
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12       # transformers/configuration_bert.py:95
BN_EPS = 1e-5        # torch BatchNorm2d default, src/char_cnn.py:17
BN_MOMENTUM = 0.1
MASK_VALUE = -10000.0  # transformers/modeling_bert.py:697


# ----------------------------------------------------------------------------
# BERT pieces (transformers/modeling_bert.py)
# ----------------------------------------------------------------------------
def gelu_erf(x):
    """modeling_bert.py:125-131 - exact erf GELU."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, w, b, eps=LN_EPS):
    """torch.nn.LayerNorm: biased variance over the last dim."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def dropout(x, p, training, gen=None):
    if not training or p == 0.0:
        return x
    keep = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype)
    return x * keep / (1.0 - p)


def bert_embeddings(sd, prefix, input_ids=None, inputs_embeds=None, position_mode="arange",
                    p_drop=0.0, training=False):
    """BertEmbeddings.forward, modeling_bert.py:169-193.

    position_mode 'arange' -> position_ids = arange(S) (default, :177-179);
    'zeros' -> position_ids == 0 everywhere (src/models.py:852-854).
    token_type_ids are always 0 on this path (:180-181).
    """
    if inputs_embeds is None:
        inputs_embeds = sd[prefix + "embeddings.word_embeddings.weight"][input_ids]
    B, S, H = inputs_embeds.shape
    pos = sd[prefix + "embeddings.position_embeddings.weight"]
    if position_mode == "arange":
        pe = pos[:S].unsqueeze(0)
    else:
        pe = pos[0].view(1, 1, H)
    te = sd[prefix + "embeddings.token_type_embeddings.weight"][0].view(1, 1, H)
    e = inputs_embeds + pe + te
    e = layer_norm(e, sd[prefix + "embeddings.LayerNorm.weight"], sd[prefix + "embeddings.LayerNorm.bias"])
    return dropout(e, p_drop, training)


def bert_self_attention(sd, lp, x, ext_mask, n_heads, p_drop=0.0, training=False):
    """BertSelfAttention.forward, modeling_bert.py:220-263."""
    B, S, H = x.shape
    d = H // n_heads

    def split(t):
        return t.view(B, S, n_heads, d).permute(0, 2, 1, 3)

    q = split(F.linear(x, sd[lp + "attention.self.query.weight"], sd[lp + "attention.self.query.bias"]))
    k = split(F.linear(x, sd[lp + "attention.self.key.weight"], sd[lp + "attention.self.key.bias"]))
    v = split(F.linear(x, sd[lp + "attention.self.value.weight"], sd[lp + "attention.self.value.bias"]))
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)   # scale BEFORE the mask (:239-243)
    scores = scores + ext_mask
    probs = torch.softmax(scores, dim=-1)
    probs = dropout(probs, p_drop, training)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous().view(B, S, H)
    return ctx


def bert_layer(sd, lp, x, ext_mask, n_heads, p_drop=0.0, training=False, taps=None):
    """BertLayer.forward, modeling_bert.py:356-369 (a4..a7 of SURVEY 8a)."""
    ctx = bert_self_attention(sd, lp, x, ext_mask, n_heads, p_drop, training)
    a = F.linear(ctx, sd[lp + "attention.output.dense.weight"], sd[lp + "attention.output.dense.bias"])
    a = dropout(a, p_drop, training)
    a = layer_norm(a + x, sd[lp + "attention.output.LayerNorm.weight"], sd[lp + "attention.output.LayerNorm.bias"])
    inter = gelu_erf(F.linear(a, sd[lp + "intermediate.dense.weight"], sd[lp + "intermediate.dense.bias"]))
    o = F.linear(inter, sd[lp + "output.dense.weight"], sd[lp + "output.dense.bias"])
    o = dropout(o, p_drop, training)
    o = layer_norm(o + a, sd[lp + "output.LayerNorm.weight"], sd[lp + "output.LayerNorm.bias"])
    if taps is not None:
        taps[lp + "ctx"] = ctx
        taps[lp + "attn_out"] = a
        taps[lp + "inter"] = inter
        taps[lp + "out"] = o
    return o


def extended_mask(attention_mask):
    """modeling_bert.py:687,696-697: (1 - m) * -10000, broadcast [B,1,1,S]."""
    return (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * MASK_VALUE


def bert_model(sd, prefix, n_layers, n_heads, attention_mask, input_ids=None, inputs_embeds=None,
               position_mode="arange", p_drop=0.0, training=False, taps=None):
    """BertModel.forward, modeling_bert.py:639-745, sequence output only.

    The pooler (:410-416) is computed by the reference and dropped by every
    caller on this path ([0] only), so it is not restated.
    """
    x = bert_embeddings(sd, prefix, input_ids, inputs_embeds, position_mode, p_drop, training)
    if taps is not None:
        taps[prefix + "emb"] = x
    em = extended_mask(attention_mask)
    for i in range(n_layers):
        x = bert_layer(sd, "%sencoder.layer.%d." % (prefix, i), x, em, n_heads, p_drop, training, taps)
    return x


# ----------------------------------------------------------------------------
# Pinyin GRU (src/models.py:661-669, 818-826)
# ----------------------------------------------------------------------------
def pho_gru_last_hidden(sd, pho_idx, pho_lens):
    """Embedding(33,768) -> packed 1-layer unidirectional GRU -> h at each
    sequence's last valid step.  PyTorch gate order r,z,n; h0 = 0.

        r = sig(W_ir x + b_ir + W_hr h + b_hr)
        z = sig(W_iz x + b_iz + W_hz h + b_hz)
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn))
        h' = (1 - z) * n + z * h
    """
    emb = sd["pho_embeddings.weight"][pho_idx]                # [N, Tp, H]
    N, Tp, H = emb.shape
    w_ih, w_hh = sd["pho_gru.weight_ih_l0"], sd["pho_gru.weight_hh_l0"]
    b_ih, b_hh = sd["pho_gru.bias_ih_l0"], sd["pho_gru.bias_hh_l0"]
    lens = torch.as_tensor(pho_lens, dtype=torch.long)
    h = torch.zeros(N, H, dtype=emb.dtype)
    for t in range(Tp):
        gi = F.linear(emb[:, t], w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        i_r, i_z, i_n = gi.chunk(3, 1)
        h_r, h_z, h_n = gh.chunk(3, 1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h_new = (1.0 - z) * n + z * h
        alive = (lens > t).unsqueeze(1)
        h = torch.where(alive, h_new, h)
    return h


# ----------------------------------------------------------------------------
# Glyph ResNet (src/char_cnn.py)
# ----------------------------------------------------------------------------
def batch_norm(sd, p, x, training, new_buffers=None):
    """nn.BatchNorm2d: train -> biased batch var to normalise, running stats
    updated with the UNBIASED var, momentum 0.1; eval -> running stats."""
    w, b = sd[p + "weight"], sd[p + "bias"]
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))
        if new_buffers is not None:
            unbiased = var * (n / max(n - 1, 1))
            new_buffers[p + "running_mean"] = (1 - BN_MOMENTUM) * sd[p + "running_mean"] + BN_MOMENTUM * mean.detach()
            new_buffers[p + "running_var"] = (1 - BN_MOMENTUM) * sd[p + "running_var"] + BN_MOMENTUM * unbiased.detach()
            new_buffers[p + "num_batches_tracked"] = sd[p + "num_batches_tracked"] + 1
    else:
        mean, var = sd[p + "running_mean"], sd[p + "running_var"]
    xh = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS)
    return xh * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def basic_block(sd, p, x, training, new_buffers=None, taps=None, tap_name=None):
    """BasicBlock.forward, char_cnn.py:9-32 (always stride 2 with a 1x1
    shortcut conv in CharResNet)."""
    r = F.conv2d(x, sd[p + "residual_function.0.weight"], None, stride=2, padding=1)
    r = torch.relu(batch_norm(sd, p + "residual_function.1.", r, training, new_buffers))
    if taps is not None:
        taps[tap_name + ".h1"] = r
    r = F.conv2d(r, sd[p + "residual_function.3.weight"], None, stride=1, padding=1)
    r = batch_norm(sd, p + "residual_function.4.", r, training, new_buffers)
    s = F.conv2d(x, sd[p + "shortcut.0.weight"], None, stride=2, padding=0)
    s = batch_norm(sd, p + "shortcut.1.", s, training, new_buffers)
    return torch.relu(r + s)


def char_resnet(sd, images, training, new_buffers=None, taps=None):
    """CharResNet.forward, char_cnn.py:46-55: [N,F,32,32] -> [N,768]."""
    h = images
    for i in range(1, 6):
        h = basic_block(sd, "resnet.res_block%d." % i, h, training, new_buffers, taps, "resnet.block%d" % i)
        if taps is not None:
            taps["resnet.block%d" % i] = h
    return h.squeeze(-1).squeeze(-1)


# ----------------------------------------------------------------------------
# Heads
# ----------------------------------------------------------------------------
def masked_cross_entropy(logits, loss_mask, labels):
    """src/models.py:862-869: mean CE over positions with loss_mask == 1."""
    V = logits.shape[-1]
    active = loss_mask.reshape(-1) == 1
    return F.cross_entropy(logits.reshape(-1, V)[active], labels.reshape(-1)[active])


def gate_fuse(sd, bert_h, pho_h, res_h, attention_mask):
    """src/models.py:840-850: masked mean + 3 independent sigmoid gates."""
    m = attention_mask.to(torch.float32)
    mean = (bert_h * m.unsqueeze(2)).sum(dim=1) / m.sum(dim=1, keepdim=True)
    mean = mean.unsqueeze(1).expand(-1, bert_h.size(1), -1)
    cat = torch.cat((bert_h, pho_h, res_h, mean), dim=-1)
    g = torch.sigmoid(F.linear(cat, sd["gate_net.weight"], sd["gate_net.bias"]))
    return g[..., 0:1] * bert_h + g[..., 1:2] * pho_h + g[..., 2:3] * res_h


def spellbert_forward(sd, cfg, batch, training=False, taps=None):
    """SpellBert.forward, src/models.py:50-73 (BASELINE config 1)."""
    p = cfg["hidden_dropout_prob"] if training else 0.0
    h = bert_model(sd, "bert.", cfg["num_hidden_layers"], cfg["num_attention_heads"], batch["masks"],
                   input_ids=batch["src_idx"], p_drop=p, training=training, taps=taps)
    if taps is not None:
        taps["bert_h"] = h
    h = dropout(h, p, training)
    logits = F.linear(h, sd["classifier.weight"], sd["classifier.bias"])
    if "tgt_idx" in batch:
        return masked_cross_entropy(logits, batch["loss_masks"], batch["tgt_idx"]), logits
    return (logits,)


def arch3_forward(sd, cfg, batch, training=False, new_buffers=None, taps=None):
    """SpellBertPho2ResArch3.forward, src/models.py:806-870.

    ``training`` selects BatchNorm batch statistics AND dropout; parity runs
    use cfg dropout probs of 0 (reference RNG streams are not reproducible).
    """
    p = cfg["hidden_dropout_prob"] if training else 0.0
    ids, mask = batch["src_idx"], batch["masks"]
    B, S = ids.shape
    nh = cfg["num_attention_heads"]
    bert_h = bert_model(sd, "bert.", cfg["num_hidden_layers"], nh, mask, input_ids=ids,
                        p_drop=p, training=training, taps=taps)                          # :816
    pho_h = pho_gru_last_hidden(sd, batch["pho_idx"], batch["pho_lens"]).view(B, S, -1)      # :818-826
    if taps is not None:
        taps["pho_gru"] = pho_h
    pho_h = bert_model(sd, "pho_model.", 4, nh, mask, inputs_embeds=pho_h,
                       p_drop=p, training=training, taps=taps)                           # :827
    if "char_images.weight" in sd:          # num_fonts == 1: an nn.Embedding [V, 1024] viewed as one 32 x 32 font            # :674-676, :831-832
        images = sd["char_images.weight"].index_select(0, ids.reshape(-1)).reshape(-1, 1, 32, 32)
    else:
        images = sd["char_images_multifonts"].index_select(0, ids.reshape(-1))              # :677-679, :833-834
    res = char_resnet(sd, images, training, new_buffers, taps)                              # :836
    res_h = layer_norm(res.view(B, S, -1), sd["resnet_layernorm.weight"], sd["resnet_layernorm.bias"])  # :838
    fused = gate_fuse(sd, bert_h, pho_h, res_h, mask)                                       # :840-850
    out = bert_model(sd, "output_block.", 3, nh, mask, inputs_embeds=fused, position_mode="zeros",
                     p_drop=p, training=training, taps=taps)                             # :852-856
    out_d = dropout(out, p, training)                                                       # :858
    logits = F.linear(out_d, sd["classifier.weight"], sd["classifier.bias"])                # :859
    if taps is not None:
        taps.update(bert_h=bert_h, pho_h=pho_h, res=res, res_h=res_h, fused=fused, out=out)
    if "tgt_idx" in batch:
        return masked_cross_entropy(logits, batch["loss_masks"], batch["tgt_idx"]), logits
    return (logits,)


# ----------------------------------------------------------------------------
# Optimizer step adjacent to the path
# ----------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (src/run.py:207): global L2 norm;
    scale by max_norm / (norm + 1e-6) when that is < 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        grads = [g * coef for g in grads]
    return grads, total


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, correct_bias=True):
    """transformers/optimization.py:110-169 (decoupled weight decay applied
    AFTER the Adam update with the un-corrected lr)."""
    m = m * beta1 + (1.0 - beta1) * g
    v = v * beta2 + (1.0 - beta2) * g * g
    denom = v.sqrt() + eps
    step_size = lr
    if correct_bias:
        step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p = p - step_size * (m / denom)
    if weight_decay > 0.0:
        p = p - lr * weight_decay * p
    return p, m, v


def linear_schedule_with_warmup(step, warmup, total):
    """transformers/optimization.py:45-54 lr multiplier."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))
