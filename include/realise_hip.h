/*
 * realise_hip.h - C ABI of the MI355X-native ReaLiSe hot path (librealise_hip.so).
 *
 * The reference (DaDaMrX/ReaLiSe) is 100 % Python on PyTorch; there is no FFI layer in it.  The
 * boundary this library sits behind is the nn.Module protocol of `SpellBertPho2ResArch3` /
 * `SpellBert` (src/models.py:652-870, :32-73) as consumed by src/run.py:191,200,258.  Each entry
 * point below names the reference call site(s) it replaces.  All pointers are DEVICE pointers
 * unless stated otherwise; `stream` is a hipStream_t; no torch types cross this boundary.
 * Every function returns 0 on success, non-zero on error (1 = bad argument, 2 = launch failure).
 * dtype: 0 = float32 (exact-fp32 parity mode, v_mfma_f32_16x16x4_f32), 1 = bfloat16 (speed mode,
 * v_mfma_f32_16x16x32_bf16; fp32 accumulation / statistics).
 */
#ifndef REALISE_HIP_H
#define REALISE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REALISE_F32 0
#define REALISE_BF16 1

/* ------------------------------------------------------------------------------------------
 * Fine-grained operators (each one kernel family).  Used by the engine and by the parity tests.
 * ---------------------------------------------------------------------------------------- */

/* Epilogue of realise_gemm_nt / realise_conv_nt.  mode: 0 store(+bias,+accumulate), 1 bias+erf-GELU
 * (out2 = pre-activation), 2 dropout(acc+bias)+aux (residual), 4 acc * gelu'(aux). */
typedef struct {
  int32_t mode;
  int32_t accumulate;
  void* out;
  int64_t ldo;
  void* out2;
  const float* bias;
  const void* aux;
  int64_t ldaux;
  float alpha;
  uint32_t drop_seed;
  uint32_t drop_thresh;   /* p * 2^32 ; 0 = no dropout */
  float drop_scale;       /* 1 / (1 - p) */
} realise_epilogue;

/* Implicit-im2col geometry over an NHWC tensor: rows are the pixels of an Hr x Wr map, K = KH*KW*C.
 * mode 0 = forward gather, 1 = data-gradient gather.  img_index (nullable) maps image n to a row
 * of `src` (glyph lookup char_images_multifonts.index_select, src/models.py:829-834). */
typedef struct {
  const void* src;
  const int64_t* img_index;
  int32_t rows, Hr, Wr, Hs, Ws, C, KH, KW, stride, pad, mode;
} realise_conv_geom;

/* C[M,N] = A[M,K] . B[N,K]^T  - nn.Linear forward (modeling_bert.py:221,231-232,273,326,339;
 * models.py:859), and its data gradient when B is the transposed weight. */
int realise_gemm_nt(void* stream, int dtype, const void* A, int64_t lda, const void* B, int64_t ldb,
                    int M, int N, int K, const realise_epilogue* ep);
/* Same with A gathered on the fly: nn.Conv2d forward / input gradient (src/char_cnn.py:15-28). */
int realise_conv_nt(void* stream, int dtype, const realise_conv_geom* a, const void* B, int64_t ldb,
                    int M, int N, int K, const realise_epilogue* ep);
/* Input gradient of a stride-2 nn.Conv2d (char_cnn.py:16,24 under loss.backward()) by parity classes of the input pixel: pixel
 * (2yy+py, 2xx+px) is reached only by the taps kh = ((py+pad)&1) + 2i, kw = ((px+pad)&1) + 2j, so four GEMMs over a quarter of the
 * pixels each with 1 / 2 / 2 / 4 taps (3x3, pad 1) replace one GEMM over all pixels with 9 taps of which 75 % are stride misses.
 * `a` is the full-map mode-1 geometry (rows = N*Hr*Wr, Hr = Wr a power of two, no img_index); B_classes is the [Cin][slot][Co]
 * weight copy whose tap slots are ordered class by class (class c = 2*py+px; inside a class i-major); the result rows are written
 * to their pixel positions.  With a 1x1 / pad 0 kernel only class 0 has a tap: ep->accumulate must be set. */
int realise_conv_dgrad_s2(void* stream, int dtype, const realise_conv_geom* a, const void* B_classes, int64_t ldb, int Cin,
                          const realise_epilogue* ep);
/* out[I,J] += sum_p A[p,i] * B[p,j] (fp32) - nn.Linear weight gradient.  The reduction over p is split
 * over workgroups; `scratch` (fp32, scratch_elems >= I*J, ideally several times that) receives the partial
 * slabs that a second kernel folds into `out`.  scratch == NULL falls back to fp32 atomics. */
int realise_gemm_tn(void* stream, int dtype, const void* A, int64_t lda, const void* B, int64_t ldb,
                    int P, int I, int J, float* out, int64_t ldo, float* scratch, int64_t scratch_elems,
                    float* colsum_out /* nullable: colsum_out[i] += sum_p A[p,i], the matching bias gradient */);
/* Up to 4 Linear weight gradients that share the token count P in ONE launch (the four Linear layers of a transformer
 * layer: BertSelfAttention q/k/v packed, BertSelfOutput.dense, BertIntermediate.dense, BertOutput.dense,
 * modeling_bert.py:221-232,273,326,339 under loss.backward(), run.py:200): every workgroup owns one 128x128 tile of one
 * problem over the whole reduction, so there is no reduction split, no scratch and no fold pass.
 * out_k[I_k, J_k] += A_k^T B_k, colsum_k[i] += sum_p A_k[p,i] (nullable).  ldo % 4 == 0. */
typedef struct realise_tn_problem {
  const void* A; int64_t lda;      /* dY [P, I] */
  const void* B; int64_t ldb;      /* X  [P, J] */
  int32_t I, J;
  float* out; int64_t ldo;
  float* colsum;
} realise_tn_problem;
int realise_gemm_tn_grouped(void* stream, int dtype, int n, const realise_tn_problem* problems, int P);
/* Conv2d weight gradient into the reference's [Co][Ci][KH][KW] layout. */
int realise_conv_tn(void* stream, int dtype, const void* A, int64_t lda, const realise_conv_geom* b,
                    int P, int Co, int Ci, float* out, float* scratch, int64_t scratch_elems);
/* A/B knobs, probe modes and the per-launch timing hooks live in include/realise_hip_debug.h (diagnostics, not operators). */

/* BertSelfAttention core (modeling_bert.py:239-260): softmax(QK^T/8 + mask_add) -> dropout -> .V
 * q/k/v: [B*S][ldq] token-major, head h at columns 64h..64h+63; ctx [B*S][ldc]; lse [B][nh][S].
 * S <= 128: one workgroup per (batch, head), the whole score tile in registers; S > 128 (max_seq_length 256 / 512, run.py:304): tiles of
 * 128 keys / queries, the forward with a running row maximum and sum (B * nh * S * S < 2^32). */
int realise_attention_fwd(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ldq,
                          const float* mask_add, void* ctx, int64_t ldc, float* lse, int B, int nh, int S,
                          uint32_t drop_seed, uint32_t drop_thresh, float drop_scale);
int realise_attention_bwd(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ldq,
                          const float* mask_add, const void* ctx, const void* dctx, int64_t ldc, const float* lse,
                          float* rowdot, void* dq, void* dk, void* dv, int64_t ldd, int B, int nh, int S,
                          uint32_t drop_seed, uint32_t drop_thresh, float drop_scale);
int realise_mask_to_additive(void* stream, const int64_t* masks, float* out, int n);

/* torch.nn.LayerNorm over the last dim (eps inside the sqrt), y/xhat of `dtype`, rstd fp32. */
int realise_layernorm_fwd(void* stream, int dtype, const void* x, const float* gamma, const float* beta, float eps,
                          void* y, void* xhat, float* rstd, int rows, int H);
int realise_layernorm_bwd(void* stream, int dtype, const void* dy, const void* xhat, const float* rstd,
                          const float* gamma, void* dx, float* dgamma, float* dbeta, int rows, int H);
/* CrossEntropyLoss over rows with loss_mask == 1 (src/models.py:862-869); dlogits nullable. */
int realise_masked_ce(void* stream, int dtype, const void* logits, int64_t ld, const int64_t* labels,
                      const int64_t* loss_mask, int rows, int V, float* loss_out, float* count_scratch, void* dlogits);

/* One time step of the pinyin GRU (nn.GRU over pack_padded_sequence, src/models.py:818-826) on the n_alive longest sequences
 * (tokens sorted by decreasing length: perm[i] = original token of sorted row i, lens[i] its length).  The input projection is a
 * table [33][3H] = W_ih Emb^T + b_ih (fp32); gh = h_prev W_hh^T + b_hh comes from realise_gemm_nt (NULL at t == 0: gh = b_hh,
 * h_prev = 0).  Forward: r, z, n gates -> h_new (sorted rows), rzn saved, out[perm[i]] = h when the sequence ends at this step.
 * Backward (one BPTT step): dh (sorted, in/out: running dL/dh; rows that end here start from dout[perm[i]]) -> dgi / dgh
 * [n_alive][3H] for the table / W_hh gradients, onehot[n_alive][64] of the step's pinyin id. */
typedef struct {
  int32_t n_alive, H, Tp, t;
  const float* table; const int64_t* pho_idx; const int32_t* perm; const int32_t* lens;
  const void* gh; const float* b_hh; const void* h_prev; void* h_new; void* rzn; void* out;
  const void* dout; void* dh; void* dgi; void* dgh; void* onehot;
} realise_gru_step;
int realise_gru_step_fwd(void* stream, int dtype, const realise_gru_step* a);
int realise_gru_step_bwd(void* stream, int dtype, const realise_gru_step* a);
/* Gated fusion (src/models.py:840-850): masked mean of the bert states per sentence, three sigmoid gates from
 * [bert | pho | res | mean] . W[3][4H] + bias, fused = g0 bert + g1 pho + g2 res.  Backward fills dbert / dpho / dres and
 * accumulates dW [3][4H], dbias [3]; mean [B][H], msum [B], g [B*S][4] are saved by the forward, dz [B*S][4] is scratch. */
typedef struct {
  int32_t B, S, H;
  const void* bert; const void* pho; const void* res; const int64_t* masks; const float* W; const float* bias;
  float* mean; float* msum; float* g; void* fused;
  const void* dfused; void* dbert; void* dpho; void* dres; float* dz; float* dW; float* dbias;
} realise_gate;
int realise_gate_fwd(void* stream, int dtype, const realise_gate* a);
int realise_gate_bwd(void* stream, int dtype, const realise_gate* a);
/* nn.BatchNorm2d over an NHWC activation viewed as [P = N*H*W][C] (src/char_cnn.py:17-28), optional fused ReLU.  training:
 * batch statistics (two passes: mean, then centred squares), running statistics updated with the unbiased variance and
 * `momentum`, num_batches_tracked += 1 (nullable), save_mean / save_rstd [C] for the backward.  scratch: 4*C floats.
 * Backward: relu_src = the forward's y when ReLU was fused (NULL otherwise); dgamma / dbeta are ACCUMULATED; scratch 2*C floats. */
int realise_batchnorm_fwd(void* stream, int dtype, const void* x, int P, int C, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, int training, int relu, void* y,
                          float* save_mean, float* save_rstd, float* scratch);
int realise_batchnorm_bwd(void* stream, int dtype, const void* dy, const void* relu_src, const void* x, const float* save_mean, const float* save_rstd,
                          const float* gamma, int P, int C, void* dx, float* dgamma, float* dbeta, float* scratch);
/* Gradient of BertEmbeddings' three table lookups (modeling_bert.py:183-190) from de = d(word + position + type sum) [B*S][H]:
 * word_grad[ids[t]] += de[t] (nullable), pos_grad[s] += sum_b de[b, s] (pos_zero: everything into row 0), type_grad[0] += sum_t de[t]. */
int realise_embedding_bwd(void* stream, int dtype, const void* de, const int64_t* ids, int B, int S, int H, float* word_grad, float* pos_grad,
                          int pos_zero, float* type_grad);
/* Glyph dedup bookkeeping (no reference counterpart: char_images_multifonts.index_select(0, ids) depends on the id only): the distinct
 * ids in order of first occurrence (uniq_ids), their multiplicities (counts), inv[t] = slot of token t, bounds[0] = #distinct,
 * bounds[1 + k] = #distinct * hw[k].  first_scratch: V ints, flag_scratch: T ints.  realise_segment_sum: out[u] = sum over tokens
 * with inv[t] == u of x[t] (acc: T*C floats of scratch; rows >= *nuniq_dev untouched). */
int realise_glyph_unique(void* stream, const int64_t* ids, int T, int V, int32_t* first_scratch, int32_t* flag_scratch, int64_t* uniq_ids, float* counts,
                         int32_t* inv, int32_t* bounds, int nhw, const int32_t* hw);
int realise_segment_sum(void* stream, int dtype, const void* x, const int32_t* inv, int T, int C, float* acc, void* out, const int32_t* nuniq_dev);

/* Eval decode: ids[row] = argmax over the V logits of the row, first maximum wins - replaces
 * `logits.detach().cpu().numpy()` + `np.argmax(preds, axis=-1)` (src/run.py:262-263): only the ids cross PCIe. */
int realise_argmax(void* stream, int dtype, const void* logits, int64_t ld, int rows, int V, int64_t* ids);

/* ------------------------------------------------------------------------------------------
 * Whole-model engine: SpellBert.forward (models.py:50-73) / SpellBertPho2ResArch3.forward
 * (models.py:806-870) and their autograd (`loss.backward()`, run.py:200), one C call each.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t model_type;        /* 0 = SpellBert (BERT only), 1 = SpellBertPho2ResArch3 */
  int32_t dtype;
  int32_t hidden, heads, intermediate, vocab, max_pos, type_vocab;
  int32_t bert_layers, pho_layers, out_layers;
  int32_t num_fonts, glyph_size, pho_vocab;
  float hidden_dropout, attn_dropout, ln_eps;
  int32_t tie_classifier;    /* classifier.weight aliases bert word embeddings (models.py:700-701) */
} realise_config;

/* Parameter layout: the library is the source of truth for where each state_dict tensor of the
 * reference lives inside five flat arenas:
 *   0 trainable fp32 (ordered by backward completion, so DDP buckets are contiguous slices)
 *   1 trainable-but-never-used fp32 (poolers, unused word embeddings: no gradient, as in the reference)
 *   2 frozen fp32 (char_images_multifonts)   3 fp32 buffers (BN running stats)   4 int64 buffers */
int realise_layout_count(const realise_config* cfg);
int realise_layout_entry(const realise_config* cfg, int index, char* name, int name_cap, int32_t* arena,
                         int64_t* offset, int32_t* ndim, int64_t* shape4);
int64_t realise_arena_elems(const realise_config* cfg, int arena);
/* gradient-bucket boundaries (element offsets into arena 0, in backward completion order) */
int realise_bucket_count(const realise_config* cfg);
int realise_bucket_bounds(const realise_config* cfg, int bucket, int64_t* begin, int64_t* end);

typedef struct realise_engine realise_engine;
realise_engine* realise_engine_create(const realise_config* cfg, float* params, float* grads, float* unused_params,
                                      float* frozen, float* buffers_f32, int64_t* buffers_i64);
void realise_engine_destroy(realise_engine* e);
int64_t realise_engine_shadow_bytes(const realise_engine* e);
int64_t realise_engine_workspace_bytes(const realise_engine* e, int B, int S, int Tp);   /* Tp = -1: glyph-only plan */
/* hand the engine its operand-shadow arena and its activation workspace (caller-allocated).  The shadow arena must be
 * ZERO-FILLED by the caller: padded operand rows (the classifier's W^T copy is pitched to 64 columns) rely on it. */
int realise_engine_bind(realise_engine* e, void* shadow, void* workspace, int64_t workspace_bytes);
/* Workspace slots (round 6).  The engine installs one plan per (B, S, Tp) INTO the bound workspace (zero-filled once, then kept
 * self-cleaning) and remembers it per workspace address: a caller that keeps one buffer per batch shape (train batch, eval batch, the
 * short last batch of run.py:104-123's windows, the glyph-only plan) and re-binds the matching one before a forward pays the zero
 * fill once per shape instead of on every switch.  Call this before freeing a buffer the engine has been bound to. */
void realise_engine_forget_workspace(realise_engine* e, void* workspace);
/* re-derive the compute-dtype operand copies from the fp32 masters (after any parameter update) */
int realise_engine_refresh_shadows(realise_engine* e, void* stream);
/* the same, for a caller whose last parameter update was realise_engine_adamw (which wrote the Linear weights' copies itself):
 * linear_current != 0 re-derives only the conv-weight copies (and the glyph table's image when it changed) */
int realise_engine_refresh_shadows_ex(realise_engine* e, void* stream, int linear_current);
/* the frozen glyph table (arena 2) changed: rebuild its NHWC operand image at the next refresh */
void realise_engine_invalidate_frozen(realise_engine* e);
/* zero_grad() without the memset: fresh != 0 tells the engine that the gradient arena holds nothing the caller wants kept - the
 * next backward (or glyph backward) then zero-fills only what it accumulates into (one launch) and STORES the Linear weight
 * gradients of the transformer layers (80 % of the arena) instead of adding to them.  The flag clears itself at that backward;
 * without it gradients are accumulated into the arena as autograd does (run.py:200 under gradient_accumulation_steps). */
void realise_engine_set_grads_fresh(realise_engine* e, int fresh);
/* Token-id range errors.  nn.Embedding raises IndexError for an id outside its table (modeling_bert.py:183-186 word ids,
 * models.py:818 pinyin ids, :831 glyph lookup).  The engine never indexes with a caller's id: every forward first copies src_idx /
 * pho_idx into its workspace with out-of-range ids replaced by 0 and sets *flag = 1 (sticky; device memory or host-mapped pinned
 * memory, nullable) when it replaced one.  The caller reads the flag when it next touches the host (the Python module: at the
 * next forward / backward / decode) and raises; results of a flagged step are meaningless, but no memory outside the tables was read.
 * `flag` points to TWO int32: flag[0] = the id-range flag above; flag[1] = set to 1 if a workgroup of the fused dense + LayerNorm
 * launch gave up waiting for the other column tiles of its rows (never observed; the wait is bounded so that a scheduling surprise
 * cannot hang the device - the step's results are meaningless then). */
void realise_engine_set_id_flag(realise_engine* e, int32_t* flag);
/* The gradient arriving at the loss (`grad_output` of loss.backward(): 1, or 1 / gradient_accumulation_steps, src/run.py:197-200) as
 * a DEVICE fp32 scalar that realise_engine_backward reads when it runs (NULL = 1, the default).  It scales the three gradients that
 * leave the classifier head inside the kernels that store them; the host never reads it and the cross-entropy gradient rows are
 * not touched.  The pointer must stay valid until the backward's kernels have run; it stays installed until changed. */
void realise_engine_set_loss_grad(realise_engine* e, const float* grad_dev);

typedef struct {
  int32_t B, S, Tp;
  int32_t training;           /* dropout + BatchNorm batch statistics + keep activations for backward */
  int32_t want_dlogits;       /* compute d loss / d logits during the loss pass (needed before backward) */
  uint64_t seed;              /* dropout stream for this step */
  const int64_t* src_idx;     /* [B,S] */
  const int64_t* masks;       /* [B,S] */
  const int64_t* loss_masks;  /* [B,S] nullable when tgt_idx is null */
  const int64_t* tgt_idx;     /* [B,S] nullable */
  const int64_t* pho_idx;     /* [B*S,Tp] (arch3) */
  const int32_t* pho_perm;    /* [B*S] tokens sorted by decreasing pinyin length (device) */
  const int32_t* pho_lens_sorted; /* [B*S] (device) */
  const int32_t* n_alive;     /* HOST array [Tp]: #sequences with length > t; NULL when n_alive_dev is given */
  float* loss_out;            /* 1 float (device), nullable when tgt_idx is null */
  void* logits_out;           /* [B*S, vocab] in `dtype` (device).  A bf16 TRAINING call with tgt_idx, loss_masks, masks and want_dlogits set
                               * (B*S % 64 == 0) computes the transformer stacks only on the rows that precede a sentence's last position with
                               * masks == 1 or loss_masks == 1: the loss, those rows' logits and every gradient equal the dense computation bit
                               * for bit; the logits rows behind that position are finite and meaningless (the reference never reads them:
                               * src/run.py:200, :262-270).  Engine knob 10 = 0 (the setter is declared in include/realise_hip_debug.h) computes every row. */
  const int32_t* n_alive_dev; /* DEVICE array [Tp] written by realise_build_pho (used when n_alive == NULL) */
  int32_t eval_live_rows;     /* != 0 on an evaluation call (training == 0, bf16, B*S % 64 == 0): the transformer stacks run over the rows up to every
                               * sentence's last position with masks == 1 (or loss_masks == 1) only, as training steps do - those rows' logits and the
                               * loss are bit-identical to the dense forward's, the logits rows of the padding behind that position are finite and
                               * meaningless (the reference's evaluation cuts its predictions at `lengths`: src/run.py:262-270).  Default 0: dense. */
  float* logits_f32_out;      /* nullable: [B*S, vocab] fp32 (device) - the logits widened to fp32, the dtype the reference returns them in
                               * (src/models.py:859); bf16 engines write them from the classifier kernel's epilogue (the same bf16-rounded
                               * values as logits_out), no pass over logits_out.  Needs logits_out. */
} realise_batch;

/* Device-side `build_batch` (src/models.py:797-804 with the per-character Pinyin2.convert of src/utils.py:58-99 folded into
 * a per-vocabulary table built once on the host): for every token t of src_idx[T]
 *     pho_idx[t][0..Tw) = table[src_idx[t]][0..Tw),   len[t] = vlens[src_idx[t]]
 * then the tokens stably sorted by decreasing length (perm, lens_sorted) and n_alive[k] = #{t : len[t] > k}, all on the
 * device: no per-character host loop and no host list `pho_lens` in the forward contract.  Feed the outputs to
 * realise_batch {pho_idx, pho_perm, pho_lens_sorted, n_alive = NULL, n_alive_dev, Tp = Tw}. */
int realise_build_pho(void* stream, const int64_t* src_idx, int T, const int64_t* table, const int32_t* vlens, int V, int Tw,
                      int64_t* pho_idx, int32_t* perm, int32_t* lens_sorted, int32_t* n_alive_dev);
int realise_engine_forward(realise_engine* e, void* stream, const realise_batch* batch);
/* gradients of the loss computed by the last training forward are ACCUMULATED into the grads arena.
 * Runs buckets [first_bucket, last_bucket] of the backward pass (see realise_bucket_bounds) so a
 * caller can overlap the all-reduce of finished buckets; pass 0, -1 for the whole pass. */
int realise_engine_backward(realise_engine* e, void* stream, int first_bucket, int last_bucket);
/* The whole pass (as first_bucket = 0, last_bucket = -1: the three model branches on three streams, weight gradients deferred
 * to the engine's side stream) that also tells a data-parallel caller when each gradient bucket is final: bucket_events[i]
 * (n_events = realise_bucket_count hipEvent_t handles owned by the caller) is recorded on the stream that finishes bucket i;
 * the caller's communication stream waits on it (hipStreamWaitEvent) and all-reduces the bucket while the rest of the backward
 * runs - what DistributedDataParallel's gradient hooks do for run.py:165-167,200.  On return the caller's stream is ordered after
 * the whole pass. */
int realise_engine_backward_signalled(realise_engine* e, void* stream, void* const* bucket_events, int n_events);
/* Glyph-only entry points (BASELINE configs[3], the glyph-CNN stress run): CharResNet.forward (src/char_cnn.py:46-55) on the
 * B*S glyph stacks char_images_multifonts[src_idx] (src/models.py:829-836), before resnet_layernorm.  res_out / d_res are
 * [B*S, 768] in the engine's compute dtype.  The workspace is sized with realise_engine_workspace_bytes(e, B, S, -1).
 * training != 0: BatchNorm uses batch statistics and updates the running buffers; realise_engine_glyph_backward then
 * ACCUMULATES the conv / BatchNorm parameter gradients for d_res into the grads arena. */
int realise_engine_glyph_forward(realise_engine* e, void* stream, const int64_t* src_idx, int B, int S, int training, void* res_out);
int realise_engine_glyph_backward(realise_engine* e, void* stream, const void* d_res);
/* named view of an internal activation (parity tests): returns 0 and fills ptr/numel if it exists */
int realise_engine_tap(realise_engine* e, const char* name, void** ptr, int64_t* numel);

/* ------------------------------------------------------------------------------------------
 * Optimizer step adjacent to the path (transformers/optimization.py:110-169, run.py:207-211)
 * ---------------------------------------------------------------------------------------- */
int realise_sumsq(void* stream, const float* g, int64_t n, float* out_accum);
int realise_adamw(void* stream, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, int correct_bias,
                  const float* grad_norm_sq, float max_grad_norm);
/* clip_grad_norm_ (run.py:207) for a caller that steps with a stock optimizer: g *= min(1, max_norm / (sqrt(*grad_norm_sq) + 1e-6)),
 * grad_norm_sq from realise_sumsq - no host synchronisation */
int realise_clip_scale(void* stream, float* g, int64_t n, const float* grad_norm_sq, float max_grad_norm);
/* The same sweep with per-group hyper-parameters (the trainer's decay / no-decay groups, run.py:146-151, with a non-zero
 * --weight_decay): group_of_block64[i / 64] is the group (index into `groups`, HOST array of <= 8) of elements [64 b, 64 b + 64)
 * - every tensor of the parameter arena starts on a 64-element boundary - and 255 marks elements no group holds. */
typedef struct { float lr, beta1, beta2, eps, weight_decay; int32_t correct_bias; } realise_adamw_group;
int realise_adamw_grouped(void* stream, float* p, const float* g, float* m, float* v, int64_t n, const uint8_t* group_of_block64,
                          const realise_adamw_group* groups, int n_groups, int64_t step, const float* grad_norm_sq, float max_grad_norm);
/* The grouped sweep over the ENGINE's parameter / gradient arenas (transformers/optimization.py:110-169, run.py:207-211), with the
 * operand copies of the Linear weights written by the pass that updates them: those tensors (90 % of the parameters) are stepped in
 * the 64 x 64 tiles of the operand-copy kernel, which stores the new fp32 value and its compute-dtype W / W^T copies together; the
 * rest by the flat sweep.  m / v: the caller's moment arenas (same layout as the parameter arena).  Follow it with
 * realise_engine_refresh_shadows_ex(e, stream, 1) instead of the full refresh. */
int realise_engine_adamw(realise_engine* e, void* stream, float* m, float* v, const uint8_t* group_of_block64,
                         const realise_adamw_group* groups, int n_groups, int64_t step, const float* grad_norm_sq, float max_grad_norm);
/* realise_engine_adamw, PIPELINED across the step boundary (round 6; run.py:207-211 followed by the next iteration's run.py:191).  The
 * sweep streams 32 bytes per parameter and nothing overlaps it while it sits between the backward and the next forward on `stream`.
 * This form orders the engine's side stream behind `stream` (gradients and grad_norm_sq are final there) and runs the sweep on it in the
 * order the next forward consumes the parameters, an event behind every piece: [everything outside the Linear weights, the tied
 * word table / classifier, BERT layers 0-1] [layers 2-3] [pinyin + output stacks, GRU] [the remaining BERT layers two by two].
 * realise_engine_forward and realise_engine_refresh_shadows_ex make their streams wait for a piece right in front of its first reader
 * and for all of it before they return, so the caller's stream is held for the first piece only (~0.3 of the ~1.1 ms at the full model
 * size).  Same kernels and per-element arithmetic as realise_engine_adamw: the same bits.  CONTRACT: between this call and the next
 * realise_engine_forward / realise_engine_refresh_shadows_ex(e, s, 0) / realise_engine_sync_optimizer nothing else may read or write
 * the parameter, gradient or moment arenas on any stream.  (Engine knob 15 = 0, declared in realise_hip_debug.h, turns it into the plain sweep.) */
int realise_engine_adamw_pipelined(realise_engine* e, void* stream, float* m, float* v, const uint8_t* group_of_block64,
                                   const realise_adamw_group* groups, int n_groups, int64_t step, const float* grad_norm_sq, float max_grad_norm);
/* order `stream` behind every piece of a pending pipelined sweep (no-op without one): before a checkpoint, a state_dict, any access to
 * the arenas from outside the engine */
int realise_engine_sync_optimizer(realise_engine* e, void* stream);
int realise_fill_f32(void* stream, float* p, float value, int64_t n);
/* widen a compute-dtype tensor to fp32 (the reference returns fp32 logits, src/models.py:859); 16-byte aligned pointers */
int realise_cast_to_f32(void* stream, int dtype, const void* src, float* dst, int64_t n);

const char* realise_version(void);

#ifdef __cplusplus
}
#endif
#endif /* REALISE_HIP_H */
