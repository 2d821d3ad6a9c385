// Memory-bound kernels of the path: embeddings, LayerNorm, BatchNorm, gate fusion, GRU gate
// math, masked cross-entropy, dropout, bias / column reductions, weight shadows, optimizer.
#pragma once
#include "common.h"

namespace rl {

struct DropParams { uint32_t seed = 0, thresh = 0; float scale = 1.0f; };

// Device-side row bound + per-image multiplicity for the glyph branch, which runs once per DISTINCT token id of the
// batch: effective rows = min(P, *rows_dev); weight(row) = counts[row / hw] (the number of tokens sharing that glyph).
constexpr int COL_SLOT_FLOATS = 262144;   // >= (row chunks) x 2C for every launch shape of the column reductions
constexpr int COL_SLOT_BYTES = COL_SLOT_FLOATS * 4;
constexpr int LN_SLOT_FLOATS = 1024 * 2 * 1024;   // LayerNorm backward: one [dgamma | dbeta] record per workgroup (<= 1024 workgroups, H <= 1024)
constexpr int LN_SLOT_BYTES = LN_SLOT_FLOATS * 4;
// tap order of the data-gradient weight copy ([ci][slot][co]): n == 0 identity, else slot s holds original tap t[s] (stride-2 convs
// store their taps parity class by parity class, conv_s2_class())
struct TapOrder { int n = 0; int t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; };

struct RowBound {
  const int* rows_dev = nullptr;
  const float* counts = nullptr;
  int hw = 1;
  float* slots = nullptr;          // optional scratch of COL_SLOT_BYTES: per-row-chunk partial sums, folded in a fixed order by a second kernel;
                                   // the outputs are then OVERWRITTEN, not accumulated (no zero-fill needed)
                                   // (bitwise reproducible BatchNorm statistics; without it the reductions use float atomics)
};
__device__ __forceinline__ int rb_rows(const RowBound& b, int P) { return b.rows_dev ? min(P, *b.rows_dev) : P; }
__device__ __forceinline__ float rb_weight(const RowBound& b, int row) { return b.counts ? b.counts[row / b.hw] : 1.0f; }

int mask_to_additive(hipStream_t st, const int64_t* masks, float* out, int n);
// out[i] = ids[i] if 0 <= ids[i] < V else 0, *flag = 1 when any id was out of range (flag nullable)
int sanitize_ids(hipStream_t st, const int64_t* ids, int64_t n, int V, int64_t* out, int* flag);

// ---- LayerNorm (K1, K4 tail, K10) ------------------------------------------------------------
template <typename T> struct LnFwdArgs {
  int rows = 0, H = 0, S = 1;
  int in_mode = 0;                 // 0: x ; 1: word[ids] + pos + type0 ; 2: x + pos + type0
  const T* x = nullptr;
  const int* row_index = nullptr;  // in_mode 0: read row row_index[r] of x instead of row r (glyph dedup gather)
  const int64_t* ids = nullptr;
  const float* word = nullptr;
  const float* pos = nullptr;      // position table [P][H]
  const float* type0 = nullptr;    // token_type row 0
  int pos_zero = 0;                // 1: position id 0 everywhere (models.py:852-854)
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float eps = 1e-12f;
  T* y = nullptr;
  T* xhat = nullptr;               // may alias x
  float* rstd = nullptr;
  DropParams drop;                 // applied to y (embeddings)
  // optional (bf16 fast path, in_mode 0): one byte per row, 0 = padding (engine row_liveness; readable up to the next multiple of 16
  // rows).  A 16-row block without a live row is not visited - y / xhat / rstd keep what they held, as the rows of an unlisted block do in
  // the live-row GEMM that produced x (gemm_nt8_live); blocks with a live row are computed whole.
  const uint8_t* row_live = nullptr;
};
template <typename T> int ln_fwd(hipStream_t st, const LnFwdArgs<T>& a);

template <typename T> struct LnBwdArgs {
  int rows = 0, H = 0;
  const T* dy = nullptr;
  DropParams in_drop;              // mask applied to dy first (embedding dropout sits after the LN)
  const T* xhat = nullptr;
  const float* rstd = nullptr;
  const float* gamma = nullptr;
  T* dx = nullptr;
  T* dx_drop = nullptr;            // optional second output: dx * mask(out_drop) (dense dropout before the LN)
  DropParams out_drop;
  float* dgamma = nullptr;         // accumulated
  float* dbeta = nullptr;
  float* slots = nullptr;          // optional scratch of LN_SLOT_BYTES: per-workgroup partial records, folded in a fixed order (no atomics)
  const uint8_t* row_live = nullptr; // optional: row_live[r] == 0 -> dy[r] is known to be exact zeros (a padding token): the row is not read,
                                   // its outputs are written as zeros (bf16 fast path; the generic kernel reads the zeros)
  int* deferred_records = nullptr; // non-null: do NOT launch the fold; *deferred_records = number of records written to `slots` (0: none, the
                                   // gradients were accumulated directly) - the caller folds several sites in one launch (ln_fold_multi)
};
// Rows of a [B, S] token batch whose gradients can be non-zero: row (b, s) is live iff s < 1 + the last position of sentence b with
// masks == 1 or loss_masks == 1 (everything after it is padding that no query attends to and no loss term reads: every backward
// activation row there is an exact zero).  row_live[B * S] bytes; tiles64 / tiles32 / tiles16 (nullable): ascending indices of the
// 64- / 32- / 16-row blocks of the token rows that hold a live row, n_tiles[0..2] their counts (the live-block lists of the
// weight-gradient reductions).  rows (nullable): the live rows themselves, ascending, n_tiles[3] of them (round 6: the row-granular
// list of the layer GEMMs).
int row_liveness(hipStream_t st, const int64_t* masks, const int64_t* loss_masks, int B, int S, uint8_t* row_live, int* tiles64, int* tiles32,
                 int* tiles16, int* n_tiles, int* rlen, int* rows = nullptr);       // rlen[b] = live rows of sentence b (the attention backward's bound)
template <typename T> int ln_bwd(hipStream_t st, const LnBwdArgs<T>& a);
// dgamma / dbeta += fixed-order sum of the per-workgroup records of up to LN_FOLD_MAX LayerNorm sites, one launch
constexpr int LN_FOLD_MAX = 8;
struct LnFoldSite { const float* recs; int nrec; float* dgamma; float* dbeta; };
struct LnFoldSites { int n = 0; int H = 0; LnFoldSite s[LN_FOLD_MAX]; };
int ln_fold_multi(hipStream_t st, const LnFoldSites& sites);
void set_ln_fast(int on);            // 1 (default): bf16 rows of 256 / 512 / 768 / 1024 columns take the half-wave-per-row kernels (16-byte accesses)
void set_ln_v2(int on);              // 1 (default, round 5): DPP row reductions + ln_bwd16v2_kernel; 0: the round-4 kernels
void set_ln_bwd_blocks(int n);       // workgroups of the fast LayerNorm backward (default 512)

// scatter d(embedding sum) to the word / position / type tables
template <typename T>
int embed_bwd(hipStream_t st, const T* de, const int64_t* ids, int B, int S, int H, float* word_grad, float* pos_grad,
              int pos_zero, float* type_grad);

template <typename T> int bias_grad(hipStream_t st, const T* dy, int64_t ld, int rows, int N, float* out, const int* rows_dev = nullptr);
template <typename T> int dropout_apply(hipStream_t st, const T* x, T* y, int rows, int H, DropParams d);

// ---- masked cross-entropy (K13) ---------------------------------------------------------------
// loss_out[0] = mean over rows with loss_mask==1 of CE(logits[row], labels[row]); dlogits (optional)
// = d loss / d logits.  count_buf: 1 float scratch.  Rows whose label is -100 (CrossEntropyLoss's ignore_index) are skipped in
// the mean; any other label outside [0, V) turns the loss into NaN.  row_loss (optional, `rows` floats): per-row terms folded in
// a fixed order - the loss is then bitwise reproducible; nullptr: one float atomicAdd per row.
// cc.act_idx != nullptr (needs row_loss): COMPACTED gradient rows - dlogits row j belongs to the j-th row that enters the loss
// (cc.act_idx[j] = its token row, cc.inv[row] = j or -1, *cc.n_act = how many; all written here); rows outside the loss are neither
// visited nor written.  The classifier's data / weight gradients then run over *n_act rows instead of all of them.
struct CeCompact { int* act_idx = nullptr; int* inv = nullptr; int* n_act = nullptr; int phase = 0; int logits_compact = 0; };      // (phase / logits_compact: ce_loss)
template <typename T>
int ce_loss(hipStream_t st, const T* logits, int64_t ld, const int64_t* labels, const int64_t* loss_mask, int rows, int V,
            float* loss_out, float* count_buf, T* dlogits, float* row_loss = nullptr, int64_t ld_dl = 0, const CeCompact& cc = CeCompact());
// out[j] = in[idx[j]], j < *n_dev (rows of H elements); out[row] = inv[row] >= 0 ? in[inv[row]] * dropout mask : 0
template <typename T> int gather_rows(hipStream_t st, const T* in, const int* idx, const int* n_dev, int max_rows, int H, T* out);
// scale_dev (nullable): a device scalar multiplied into every stored value - the incoming loss gradient, never read on the host
template <typename T> int scatter_rows_drop(hipStream_t st, const T* in, const int* inv, int rows, int H, T* out, DropParams d, const float* scale_dev = nullptr);
template <typename T> int scale_by_dev(hipStream_t st, T* x, int64_t n, const float* scale_dev);
template <typename T> int scatter_rows_drop_slab(hipStream_t st, const float* slab, int nsplit, int64_t stride, const int* inv, int rows, int H, T* out,
                                                 DropParams d, const float* scale_dev = nullptr);      // rows of a split-K GEMM's fp32 planes, folded in plane order   // ld_dl: row pitch of dlogits (0: ld)

// ---- eval decode (run.py:262-263): ids[row] = argmax_v logits[row][v], first maximum wins (numpy / torch semantics), a NaN
// counts as the maximum.  Only the ids leave the device (32 KB instead of the 692 MB fp32 logits of run.py:262).
template <typename T> int argmax_rows(hipStream_t st, const T* logits, int64_t ld, int rows, int V, int64_t* ids);

// ---- gate fusion (K11) --------------------------------------------------------------------------
template <typename T> struct GateArgs {
  int B = 0, S = 0, H = 0;
  const T* bert = nullptr; const T* pho = nullptr; const T* res = nullptr;
  const int64_t* masks = nullptr;
  const float* W = nullptr;        // [3][4H]
  const float* bias = nullptr;     // [3]
  float* mean = nullptr;           // [B][H]  (saved)
  float* msum = nullptr;           // [B]
  float* g = nullptr;              // [B*S][4] sigmoid gates (saved)
  T* fused = nullptr;
  // backward
  const T* dfused = nullptr;
  T* dbert = nullptr; T* dpho = nullptr; T* dres = nullptr;
  float* dz = nullptr;             // [B*S][4] scratch
  float* dW = nullptr; float* dbias = nullptr;
  // optional byte per token row (row_liveness): 0 = a padding row behind the sentence's last attended / loss position.  Its d fused row is an
  // exact zero, so the backward writes its zeros without reading the row's (possibly stale - live-row steps) forward activations.
  const uint8_t* row_live = nullptr;
};
template <typename T> int gate_fwd(hipStream_t st, const GateArgs<T>& a);
template <typename T> int gate_bwd(hipStream_t st, const GateArgs<T>& a);

// ---- pinyin GRU (K6) ------------------------------------------------------------------------------
template <typename T> struct GruStepArgs {
  int n_alive = 0, H = 0, Tp = 0, t = 0;
  const int* n_alive_dev = nullptr; // optional device-side count: rows >= *n_alive_dev are skipped (n_alive is then the launch bound)
  const float* table = nullptr;     // [V][3H]
  const int64_t* pho_idx = nullptr; // [N][Tp] original order
  const int* perm = nullptr;        // sorted position -> original token
  const int* lens = nullptr;        // sorted lengths
  const T* gh = nullptr;            // [n_alive][3H] = h_prev W_hh^T + b_hh (nullptr at t == 0)
  const float* b_hh = nullptr;
  const T* h_prev = nullptr;        // [N][H] sorted (nullptr at t == 0)
  T* h_new = nullptr;               // [N][H] sorted
  T* rzn = nullptr;                 // [N][3H] saved gates of this step
  T* out = nullptr;                 // [N][H] original order, written when a sequence ends
  // backward
  const T* dout = nullptr;          // [N][H] original order
  T* dh = nullptr;                  // [N][H] sorted, running dL/dh
  T* dgi = nullptr;                 // [n_alive][3H]
  T* dgh = nullptr;                 // [n_alive][3H]
  T* onehot = nullptr;              // [n_alive][64]
};
template <typename T> int gru_step_fwd(hipStream_t st, const GruStepArgs<T>& a);
template <typename T> int gru_step_bwd(hipStream_t st, const GruStepArgs<T>& a);
// Device-side build_batch (models.py:797-804 + utils.py:58-99 as a per-vocabulary table): pho_idx[t] = table[src[t]],
// len[t] = vlens[src[t]]; tokens stably sorted by decreasing length -> perm / lens_sorted; n_alive[k] = #{len > k}.
int pho_prepare(hipStream_t st, const int64_t* src, int T_, const int64_t* table, const int32_t* vlens, int V, int Tw,
                int64_t* pho_idx, int32_t* perm, int32_t* lens_sorted, int32_t* n_alive);
// table[v][j] = b_ih[j] + emb[v, :] . w_ih[j, :]   (fp32; V x 3H from the 33 pinyin symbols)
int gru_table_fwd(hipStream_t st, const float* emb, const float* w_ih, const float* b_ih, int V, int H, float* table);
int gru_table_bwd(hipStream_t st, const float* dtable, int ld_dtable, const float* emb, const float* w_ih, int V, int H,
                  float* d_emb, float* d_w_ih, float* d_b_ih);

// ---- BatchNorm over NHWC activations [P][C] (K9) -------------------------------------------------
// bf16 fast paths of the BatchNorm kernels (16-byte accesses; C, hw powers of two): knobs for A/B runs, and the two-branch backward of
// a BasicBlock's bn2 + shortcut BN (shared dy and ReLU mask read once).  *2 functions return RL_ERR_ARG when the fast path does not apply.
void set_bn_fast(int on);
void set_bn_chunks(int n);
void set_adamw_reg(int on);         // realise_set_ln key 6: AdamW + operand-copy tiles with the in-register transpose (1) or through LDS (0, default)
void set_ce_fast(int on);           // bf16 masked cross-entropy with the logits row held in registers (default 1)
int bn_fast();
// one-pass training statistics of a bf16 [P, C] map + everything bn_finalize_train does (2 launches instead of 5, x read once);
// RL_ERR_ARG when the fast path does not apply
int bn_stats_train16(hipStream_t st, const bf16_t* x, int P, int C, int n_stat, const float* gamma, const float* beta, float eps, float momentum,
                     float* rmean, float* rvar, float* mean, float* rstd, float* scale, float* shift, int64_t* nbt, RowBound rb);
int bn_bwd_reduce2(hipStream_t st, const bf16_t* dy, const bf16_t* relu_src, const bf16_t* xa, const float* mean_a, const float* rstd_a,
                   const bf16_t* xb, const float* mean_b, const float* rstd_b, int P, int C, float* sums4, RowBound rb);
int bn_bwd_apply2(hipStream_t st, const bf16_t* dy, const bf16_t* relu_src, const bf16_t* xa, const float* mean_a, const float* rstd_a,
                  const float* gamma_a, bf16_t* dxa, float* dgamma_a, float* dbeta_a, const bf16_t* xb, const float* mean_b, const float* rstd_b,
                  const float* gamma_b, bf16_t* dxb, float* dgamma_b, float* dbeta_b, const float* sums4, int P, int C, RowBound rb, int n_stat);
template <typename T> int col_sum(hipStream_t st, const T* x, int P, int C, float* out, RowBound rb = RowBound(), float scale = 1.0f);   // scale != 1 needs rb.slots
template <typename T> int col_sumsq_centered(hipStream_t st, const T* x, int P, int C, const float* mean, float* out, RowBound rb = RowBound());
// train-mode finalize: stats -> (mean, rstd, scale, shift), running-stat update (unbiased var, momentum)
int bn_finalize_mean(hipStream_t st, const float* sum, int C, int P, float* mean);
int bn_finalize_train(hipStream_t st, const float* mean, const float* sqsum, int C, int P, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var, float* rstd, float* scale, float* shift,
                      int64_t* num_batches_tracked = nullptr /* += 1 when given */);
int bn_finalize_eval(hipStream_t st, int C, const float* gamma, const float* beta, float eps, const float* running_mean,
                     const float* running_var, float* scale, float* shift);
// y = [relu]( x1*sc1 + sh1 [+ x2*sc2 + sh2] )
template <typename T>
int bn_apply(hipStream_t st, const T* x1, const float* sc1, const float* sh1, const T* x2, const float* sc2, const float* sh2,
             T* y, int P, int C, int relu, RowBound rb = RowBound());
// sums[0..C) += sum g ; sums[C..2C) += sum g*xhat  with g = dy * (relu_src > 0)
template <typename T>
int bn_bwd_reduce(hipStream_t st, const T* dy, const T* relu_src, const T* x, const float* mean, const float* rstd, int P, int C,
                  float* sums, RowBound rb = RowBound());
// dx = gamma*rstd*(g - w*sum_g/N - w*xhat*sum_gx/N); dgamma += sum_gx ; dbeta += sum_g   (w = multiplicity, N = n_stat)
template <typename T>
int bn_bwd_apply(hipStream_t st, const T* dy, const T* relu_src, const T* x, const float* mean, const float* rstd,
                 const float* gamma, const float* sums, int P, int C, T* dx, float* dgamma, float* dbeta,
                 RowBound rb = RowBound(), int n_stat = 0);
// g = dy * (relu_src > 0)
template <typename T> int relu_bwd(hipStream_t st, const T* dy, const T* relu_src, T* g, int64_t n);

// ---- glyph dedup (the ResNet input depends only on the token id) ------------------------------------
// ids[T] -> uniq_ids[<=T] (order of first occurrence), counts[slot], inv[t] = slot of token t, and bounds[0] = U,
// bounds[1+k] = U * hw[k] for k < nhw.  first_scratch: V ints, flag_scratch: T ints.
struct HwList { int n = 0; int v[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
// dedup == 0: identity bookkeeping (every token its own slot) - the dense reference pass, for A/B runs
void set_glyph_dedup(int on);
int glyph_unique(hipStream_t st, const int64_t* ids, int T_, int V, int* first_scratch, int* flag_scratch, int64_t* uniq_ids,
                 float* counts, int* inv, int* bounds, HwList hw);
// out[u][c] = sum over tokens t with inv[t] == u of x[t][c]   (fp32 scratch `acc` of T*C floats, out rows >= U untouched)
template <typename T> int segment_sum(hipStream_t st, const T* x, const int* inv, int T_, int C, float* acc, T* out, const int* nuniq_dev);
// dense[u] = table[ids[u]], u < *nuniq (images of `elems` elements)
template <typename T> int gather_images(hipStream_t st, const T* table, const int64_t* ids, const int* nuniq, int max_images, int64_t elems, T* dense);
// out[t][:] = x[inv[t]][:]
template <typename T> int gather_rows(hipStream_t st, const T* x, const int* inv, int T_, int C, T* out);

// ---- weight shadows (operand copies in the compute dtype) ----------------------------------------
template <typename T> int cast_copy(hipStream_t st, const float* src, T* dst, int64_t n);
template <typename T> int cast_to_f32(hipStream_t st, const T* src, float* dst, int64_t n);     // 16-byte aligned pointers
// src fp32 [R][C] -> dst [R][C] (optional) and dstT [C][R] (optional)
template <typename T> int cast_transpose(hipStream_t st, const float* src, int R, int C, T* dst, T* dstT);
// many matrices in ONE launch (the per-step refresh of every Linear weight's W and W^T operand copies): `descs` is a
// device array of n descriptors sorted by tile_begin, total_tiles 64x64 tiles overall
struct CastDesc { const float* src; void* dst; void* dstT; int R, C; int tile_begin, tiles_c; int ldT; };   // ldT: row pitch of dstT (>= R)
template <typename T> int cast_transpose_multi(hipStream_t st, const CastDesc* descs, int n, int total_tiles);
// conv weight [Co][Ci][KH][KW] fp32 -> fwd [Co][KH*KW][Cpad] and dgrad [Ci_rows][KH*KW][Co] (rows >= Ci zero)
template <typename T>
int conv_weight_shadow(hipStream_t st, const float* w, int Co, int Ci, int KHW, int Cpad, int CiRows, T* fwd, T* dgrad, const TapOrder& order = TapOrder());
// every conv weight of the glyph ResNet in one launch
constexpr int CONV_SHADOW_MAX = 20;
struct ConvShadowDesc { const float* w; void* fwd; void* dgrad; int Co, Ci, KHW, Cpad, CiRows, block_begin; TapOrder order; };
struct ConvShadowDescs { int n = 0; ConvShadowDesc d[CONV_SHADOW_MAX]; };
template <typename T> int conv_weight_shadow_multi(hipStream_t st, ConvShadowDescs& ds);      // fills block_begin
// glyph table [V][F][HW] fp32 -> NHWC [V][HW][Cpad]
template <typename T> int glyph_shadow(hipStream_t st, const float* tbl, int V, int F, int HW, int Cpad, T* out);

// ---- optimizer (K15, K16) -------------------------------------------------------------------------
int sumsq_accum(hipStream_t st, const float* g, int64_t n, float* out);          // out[0] += sum g^2
// AdamW of transformers/optimization.py:110-169 over a flat range; grads are scaled by
// min(1, max_norm / (sqrt(*norm_sq) + 1e-6)) when norm_sq != nullptr (clip_grad_norm_, run.py:207)
int adamw_flat(hipStream_t st, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, float bias_c1, float bias_c2, const float* norm_sq, float max_norm);

constexpr int ADAMW_MAX_GROUPS = 8;
struct AdamwGroup { float lr, beta1, beta2, eps, weight_decay, step_size; };      // step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t) (or lr)
struct AdamwGroups { int n; AdamwGroup g[ADAMW_MAX_GROUPS]; };
// skip_block (nullable): a byte per 64 elements, != 0 = stepped elsewhere (adamw_cast_multi)
int adamw_grouped(hipStream_t st, float* p, const float* g, float* m, float* v, int64_t n, const uint8_t* group_of_block,
                  const AdamwGroups& gs, const float* norm_sq, float max_norm, const uint8_t* skip_block = nullptr);
// AdamW over the Linear weights named by the cast descriptors (tile ids restart at 0 for this launch) + their compute-dtype W / W^T
// copies from the updated values, one pass; P0 / G0 / M0 / V0 = bases of the parameter / gradient / moment arenas
template <typename T>
int adamw_cast_multi(hipStream_t st, const CastDesc* descs, int n, int total_tiles, float* P0, const float* G0, float* M0, float* V0,
                     const uint8_t* group_of_block, const AdamwGroups& gs, const float* norm_sq, float max_norm, int tile_base = 0);
int clip_scale(hipStream_t st, float* g, int64_t n, const float* norm_sq, float max_norm);
int fill_f32(hipStream_t st, float* p, float v, int64_t n);
struct FillChunk { int64_t off; int32_t len, pad; };     // off: 4-element aligned
int zero_chunks(hipStream_t st, float* base, const FillChunk* chunks_dev, int n);
// the grouped AdamW sweep over a chunk list (off, len multiples of 64 elements; one workgroup per chunk)
int adamw_chunks(hipStream_t st, float* p, const float* g, float* m, float* v, const FillChunk* chunks_dev, int n_chunks,
                 const uint8_t* group_of_block, const AdamwGroups& gs, const float* norm_sq, float max_norm);
int add_i64(hipStream_t st, int64_t* p, int64_t v, int n);

}  // namespace rl
