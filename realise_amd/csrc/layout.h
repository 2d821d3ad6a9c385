// Parameter layout of the reference's state_dict inside the engine's flat arenas.
// Names and shapes follow src/models.py:652-698, transformers/modeling_bert.py:155-416 and
// src/char_cnn.py:9-55 exactly (427 keys for SpellBertPho2ResArch3).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/realise_hip.h"

namespace rl {

enum Arena { AR_TRAIN = 0, AR_UNUSED = 1, AR_FROZEN = 2, AR_BUF_F32 = 3, AR_BUF_I64 = 4, AR_COUNT = 5 };

struct TensorInfo {
  std::string name;
  int arena;
  int64_t offset;   // elements
  int ndim;
  int64_t shape[4];
  int64_t numel() const { int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; }
};

struct LayerOff {      // offsets into AR_TRAIN
  int64_t qkv_w, qkv_b, ao_w, ao_b, ao_ln_g, ao_ln_b, in_w, in_b, out_w, out_b, out_ln_g, out_ln_b;
};
struct StackOff {
  int64_t word;        // AR_TRAIN offset, or -1 when the table is unused (inputs_embeds stacks)
  int64_t pos, type, ln_g, ln_b;
  std::vector<LayerOff> layers;
};
struct BnOff { int64_t g, b; int64_t rmean, rvar; int64_t nbt; };   // g,b: AR_TRAIN; rmean,rvar: AR_BUF_F32; nbt: AR_BUF_I64
struct BlockOff { int cin, cout; int64_t w1, w2, ws; BnOff bn1, bn2, bns; };

struct Layout {
  std::vector<TensorInfo> tensors;
  int64_t arena_elems[AR_COUNT] = {0, 0, 0, 0, 0};
  std::vector<std::pair<int64_t, int64_t>> buckets;    // [begin, end) in AR_TRAIN, backward completion order
  StackOff bert, pho, outb;
  int64_t cls_w = -1, cls_b = -1;
  int64_t pho_emb = -1, gru_w_ih = -1, gru_w_hh = -1, gru_b_ih = -1, gru_b_hh = -1;
  int64_t res_ln_g = -1, res_ln_b = -1, gate_w = -1, gate_b = -1;
  int64_t glyph = -1;                                  // AR_FROZEN
  BlockOff blocks[5];
  int bert_groups = 0;                                 // number of buckets the bert layers are split in
};

inline int64_t align64(int64_t x) { return (x + 63) & ~(int64_t)63; }

inline Layout build_layout(const realise_config& c) {
  Layout L;
  const int64_t H = c.hidden, I = c.intermediate, V = c.vocab;
  auto add = [&](int arena, const std::string& name, std::initializer_list<int64_t> shape) -> int64_t {
    TensorInfo t;
    t.name = name; t.arena = arena; t.ndim = (int)shape.size();
    int k = 0;
    for (auto s : shape) t.shape[k++] = s;
    for (; k < 4; ++k) t.shape[k] = 1;
    t.offset = L.arena_elems[arena];
    L.arena_elems[arena] = align64(t.offset + (t.ndim == 0 ? 1 : t.numel()));
    L.tensors.push_back(t);
    return t.offset;
  };
  auto alias = [&](const std::string& name, int arena, int64_t offset, std::initializer_list<int64_t> shape) {
    TensorInfo t;
    t.name = name; t.arena = arena; t.ndim = (int)shape.size(); t.offset = offset;
    int k = 0;
    for (auto s : shape) t.shape[k++] = s;
    for (; k < 4; ++k) t.shape[k] = 1;
    L.tensors.push_back(t);
  };
  auto add_layer = [&](const std::string& p) -> LayerOff {
    LayerOff o;
    // q,k,v adjacent so the fused [3H,H] projection (and its gradient) is one contiguous matrix
    o.qkv_w = add(AR_TRAIN, p + "attention.self.query.weight", {H, H});
    add(AR_TRAIN, p + "attention.self.key.weight", {H, H});
    add(AR_TRAIN, p + "attention.self.value.weight", {H, H});
    o.qkv_b = add(AR_TRAIN, p + "attention.self.query.bias", {H});
    add(AR_TRAIN, p + "attention.self.key.bias", {H});
    add(AR_TRAIN, p + "attention.self.value.bias", {H});
    o.ao_w = add(AR_TRAIN, p + "attention.output.dense.weight", {H, H});
    o.ao_b = add(AR_TRAIN, p + "attention.output.dense.bias", {H});
    o.ao_ln_g = add(AR_TRAIN, p + "attention.output.LayerNorm.weight", {H});
    o.ao_ln_b = add(AR_TRAIN, p + "attention.output.LayerNorm.bias", {H});
    o.in_w = add(AR_TRAIN, p + "intermediate.dense.weight", {I, H});
    o.in_b = add(AR_TRAIN, p + "intermediate.dense.bias", {I});
    o.out_w = add(AR_TRAIN, p + "output.dense.weight", {H, I});
    o.out_b = add(AR_TRAIN, p + "output.dense.bias", {H});
    o.out_ln_g = add(AR_TRAIN, p + "output.LayerNorm.weight", {H});
    o.out_ln_b = add(AR_TRAIN, p + "output.LayerNorm.bias", {H});
    return o;
  };
  auto add_layers_desc = [&](StackOff& s, const std::string& prefix, int hi, int lo) {   // layers hi..lo descending
    for (int l = hi; l >= lo; --l) s.layers[l] = add_layer(prefix + "encoder.layer." + std::to_string(l) + ".");
  };
  auto add_emb = [&](StackOff& s, const std::string& prefix, bool word_used) {
    s.pos = add(AR_TRAIN, prefix + "embeddings.position_embeddings.weight", {c.max_pos, H});
    s.type = add(AR_TRAIN, prefix + "embeddings.token_type_embeddings.weight", {c.type_vocab, H});
    s.ln_g = add(AR_TRAIN, prefix + "embeddings.LayerNorm.weight", {H});
    s.ln_b = add(AR_TRAIN, prefix + "embeddings.LayerNorm.bias", {H});
    if (word_used) s.word = add(AR_TRAIN, prefix + "embeddings.word_embeddings.weight", {V, H});
    else { s.word = -1; add(AR_UNUSED, prefix + "embeddings.word_embeddings.weight", {V, H}); }
    add(AR_UNUSED, prefix + "pooler.dense.weight", {H, H});      // BertPooler output is dropped by every
    add(AR_UNUSED, prefix + "pooler.dense.bias", {H});           // caller on this path (modeling_bert.py:410-416)
  };
  auto close_bucket = [&](int64_t& begin) {
    L.buckets.push_back({begin, L.arena_elems[AR_TRAIN]});
    begin = L.arena_elems[AR_TRAIN];
  };
  auto add_bn = [&](const std::string& p, int64_t C) -> BnOff {
    BnOff b;
    b.g = add(AR_TRAIN, p + "weight", {C});
    b.b = add(AR_TRAIN, p + "bias", {C});
    b.rmean = add(AR_BUF_F32, p + "running_mean", {C});
    b.rvar = add(AR_BUF_F32, p + "running_var", {C});
    b.nbt = add(AR_BUF_I64, p + "num_batches_tracked", {});
    return b;
  };

  const bool arch3 = c.model_type == 1;
  int64_t begin = 0;
  L.bert.layers.resize(c.bert_layers);
  // ---- bucket 0: classifier bias (+ untied weight), output_block
  L.cls_b = add(AR_TRAIN, "classifier.bias", {V});
  if (!c.tie_classifier) L.cls_w = add(AR_TRAIN, "classifier.weight", {V, H});
  if (arch3) {
    L.outb.layers.resize(c.out_layers);
    add_layers_desc(L.outb, "output_block.", c.out_layers - 1, 0);
    add_emb(L.outb, "output_block.", false);
    close_bucket(begin);
    // ---- bucket 1: gate, resnet LN, glyph ResNet (blocks 5..1)
    L.gate_w = add(AR_TRAIN, "gate_net.weight", {3, 4 * H});
    L.gate_b = add(AR_TRAIN, "gate_net.bias", {3});
    L.res_ln_g = add(AR_TRAIN, "resnet_layernorm.weight", {H});
    L.res_ln_b = add(AR_TRAIN, "resnet_layernorm.bias", {H});
    const int chans[6] = {c.num_fonts, 64, 128, 256, 512, 768};
    for (int b = 5; b >= 1; --b) {
      BlockOff& k = L.blocks[b - 1];
      k.cin = chans[b - 1]; k.cout = chans[b];
      const std::string p = "resnet.res_block" + std::to_string(b) + ".";
      k.w2 = add(AR_TRAIN, p + "residual_function.3.weight", {k.cout, k.cout, 3, 3});
      k.bn2 = add_bn(p + "residual_function.4.", k.cout);
      k.ws = add(AR_TRAIN, p + "shortcut.0.weight", {k.cout, k.cin, 1, 1});
      k.bns = add_bn(p + "shortcut.1.", k.cout);
      k.w1 = add(AR_TRAIN, p + "residual_function.0.weight", {k.cout, k.cin, 3, 3});
      k.bn1 = add_bn(p + "residual_function.1.", k.cout);
    }
    close_bucket(begin);
    // ---- bucket 2: pho_model, GRU, pinyin embedding
    L.pho.layers.resize(c.pho_layers);
    add_layers_desc(L.pho, "pho_model.", c.pho_layers - 1, 0);
    add_emb(L.pho, "pho_model.", false);
    L.gru_w_hh = add(AR_TRAIN, "pho_gru.weight_hh_l0", {3 * H, H});
    L.gru_b_hh = add(AR_TRAIN, "pho_gru.bias_hh_l0", {3 * H});
    L.gru_w_ih = add(AR_TRAIN, "pho_gru.weight_ih_l0", {3 * H, H});
    L.gru_b_ih = add(AR_TRAIN, "pho_gru.bias_ih_l0", {3 * H});
    L.pho_emb = add(AR_TRAIN, "pho_embeddings.weight", {c.pho_vocab, H});
    close_bucket(begin);
    // models.py:674-679: one font -> nn.Embedding "char_images.weight" [V, 1024]; several -> Parameter [V, F, 32, 32].  Same bytes.
    if (c.num_fonts == 1) L.glyph = add(AR_FROZEN, "char_images.weight", {V, (int64_t)c.glyph_size * c.glyph_size});
    else L.glyph = add(AR_FROZEN, "char_images_multifonts", {V, c.num_fonts, c.glyph_size, c.glyph_size});
  }
  // ---- bert layers in groups of <= 4 (one bucket each)
  {
    int hi = c.bert_layers - 1;
    L.bert_groups = 0;
    while (hi >= 0) {
      const int lo = hi - 3 > 0 ? hi - 3 : 0;
      add_layers_desc(L.bert, "bert.", hi, lo);
      close_bucket(begin);          // SpellBert: the first group's bucket also holds the classifier bias
      ++L.bert_groups;
      hi = lo - 1;
    }
  }
  // ---- last bucket: bert embeddings (word table last: its gradient is complete last)
  add_emb(L.bert, "bert.", true);
  close_bucket(begin);
  if (c.tie_classifier) {
    L.cls_w = L.bert.word;
    alias("classifier.weight", AR_TRAIN, L.bert.word, {V, H});
  }
  return L;
}

}  // namespace rl
