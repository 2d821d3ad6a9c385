// The step engine: SpellBert.forward / SpellBertPho2ResArch3.forward (src/models.py:50-73, 806-870)
// and their backward as ONE host call each.  The engine owns the launch order, the activation
// workspace plan and the compute-dtype operand shadows; it enqueues every kernel on the caller's HIP
// stream and never synchronises or allocates, so a step is two C calls from Python (forward,
// backward) plus the optimizer call, independent of the ~900 kernels in between.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "attention.h"
#include "engine.h"
#include "gemm.h"
#include "layout.h"
#include "ops.h"

namespace rl {

// Weight-gradient overlap (default on): the four wgrad GEMMs of a BERT layer run on an engine-owned side stream while the caller's
// stream continues with the data-gradient chain (two dY buffer sets, by layer parity).  +2.3 % step throughput on top of the branch
// overlap below (bench.py: 23.6 -> 23.1 ms); identical results.  Per-kernel durations become overlap-dependent, so bench.py
// switches both overlaps off on the steps whose launches it brackets with events.
static int g_wgrad_overlap = 1;
void set_wgrad_overlap(int on) { g_wgrad_overlap = on; }
// Grouped weight gradients (default on): the four wgrad GEMMs of a transformer layer go out as ONE launch of 432 tiles with no
// reduction split (gemm_tn_group) once the layer's last dY exists, instead of four launches that each split the reduction 3-4 ways
// and fold the slabs afterwards.
static int g_wgrad_group = 1;
void set_wgrad_group(int on) { g_wgrad_group = on; }
// Stride-2 data gradients of the glyph ResNet by input-pixel parity classes (default on): four class GEMMs over a quarter of the rows
// with 1 / 2 / 2 / 4 taps instead of one GEMM over all rows with 9 taps of which 75 % are stride misses (zeros).
static int g_dgrad_parity = 1;
void set_dgrad_parity(int on) { g_dgrad_parity = on; }
// Branch overlap (default on): the three branches of SpellBertPho2ResArch3 that are independent between the inputs and the gate
// (models.py:816 bert, :818-827 pinyin GRU + pho_model, :829-838 glyph ResNet) - and their backward passes after the gate -
// run on three HIP streams (the caller's + two engine-owned ones), forked and joined with events inside one engine call.  The
// HBM-bound glyph kernels (BatchNorm passes, 64-channel convolutions) then fill the launch gaps / epilogue tails of the
// MFMA-bound BERT GEMMs and co-reside on the CUs next to them (tools/overlap_probe.py: -25 % of the glyph branch's time).
// Results are identical to the serial order (the branches touch disjoint tensors and scratch).
static int g_branch_overlap = 1;
void set_branch_overlap(int on) { g_branch_overlap = on; }
static int g_fwd_order = 0;
void set_fwd_order(int o) { g_fwd_order = o; }
// Priority classes of the engine-owned streams, applied when a stream is created (realise_set_engine keys 1..3, before the first
// forward): 0 = the device default, -1 = the highest priority the device offers, +1 = the lowest.  [0] pinyin branch, [1] glyph
// branch, [2] weight-gradient side stream.
static int g_skip_dead = 1;       // backward skips the rows of padding tokens: LayerNorm backward rows, blocks of the weight-gradient reductions (key 5;
                                  // 1: 16-row blocks in bf16, 2: whole 64-row tiles - the round-3 form, bit-identical to no skipping -, 0: off)
void set_skip_dead(int on) { g_skip_dead = on; }
static int g_cls_compact = 1;     // classifier backward over the rows that enter the loss only (realise_set_engine key 4)
void set_cls_compact(int on) { g_cls_compact = on; }
// Split-K of the classifier's data gradient (realise_set_engine key 6; bf16, compacted rows): K = 21184 vocabulary columns, N = 768,
// ~4.9 k live rows are 156 tiles of 128 x 192 on 512 workgroup slots - three K-ranges fill them (468), the fp32 partial planes are
// folded in plane order by the scatter that follows anyway.  0 = off (one launch over the whole K).
// K4 (realise_set_engine key 8): BertSelfOutput / BertOutput as ONE launch - dense + bias + dropout + residual + LayerNorm (bf16,
// the 128 x 192 two-per-CU kernel: the four column tiles of a row band exchange their LayerNorm partial sums through self-validating
// device-scope slots).  0 (DEFAULT) = the GEMM and the LayerNorm as two launches; 1 = fused; 2 = fused without the hand-off (diagnostics).
// Measured and left OFF (profiles/round4_ab.log, round4_repro_probe.log): the first form (normalise by re-reading the tile from LDS) was
// perf-neutral (17.50 vs 17.55 ms; alone 53 us against 41 + 9) but one launch in ~2000 left that re-read with a wrong element in one
// aligned 16-lane group; the shipped form normalises from registers - 1400 forwards bit-identical - but its 96-byte-per-thread store
// pattern makes the step 0.3 ms SLOWER than the two launches (17.69 vs 17.40 ms).  Passes every parity test either way.
static int g_ln_fuse = 0;
void set_ln_fuse(int on) { g_ln_fuse = on; }
// K6 (realise_set_engine key 9): a GRU time step t > 0 as ONE launch - the recurrent projection with the gate math in its epilogue
// (bf16; gemm_nt8_gru).  0 = GEMM + gru_step_fwd.
static int g_gru_fuse = 1;
void set_gru_fuse(int on) { g_gru_fuse = on; }
static int g_cls_splitk = 3;
void set_cls_splitk(int n) { g_cls_splitk = n < 0 ? 0 : (n > 4 ? 4 : n); }
// Live rows (realise_set_engine key 10; bf16 training steps, needs the 16-row block lists of g_skip_dead = 1): the layer GEMMs of the
// three transformer stacks - forward AND data gradients - run over the live 16-row blocks of the batch only (gemm_nt8_live), the
// attention forward skips the query blocks and key rows beyond a sentence's last live row.  Rows after a sentence's last attended /
// loss position never reach the loss: no query attends to them (additive -10000 mask: their probabilities are exact fp32 zeros), the
// masked mean and the loss skip them, their gradient rows are exact zeros.  The reference computes them anyway; here their
// activations are simply not produced (the workspace rows keep old, finite values - the workspace is zero-filled when a plan is
// installed), the returned training logits of such rows are computed from those stale states and mean nothing, as the reference's
// mean nothing.  The loss, every live row's activations and every gradient are bit-identical to the dense pass (same kernels, same
// accumulation order per row; tests/test_round4_gpu.py).  Evaluation / inference forwards are always dense.
// Round 6, value 2 (DEFAULT): the list the layer GEMMs walk holds the live ROWS, not 16-row blocks (gemm_nt8_live, EpiParams::live_unit
// = 1): a GEMM tile is any 128 live rows.  The blocks that complete a sentence's last 16 rows cost 0.709 - 0.648 = 6 % of the batch's
// rows, and 42 instead of 46 tile rows put the 2304-wide qkv launch into ONE round of the 512 two-per-CU slots.  Every other consumer
// (LayerNorm, attention, the weight-gradient reductions) keeps the 16-row block tables: rows 16-row blocks hold beyond the live
// length are simply not produced by the GEMMs any more, and they are rows every consumer already treats as padding (row_live == 0).
// 1 = the round-4 / 5 form (16-row blocks), 0 = dense.
static int g_live_rows = 2;
void set_live_rows(int on) { g_live_rows = on; }
// realise_set_engine(11, v): 1 = the layer GEMMs of the transformer stacks on the stream-K 256 x 192 kernel (gemm_nt8s.hip) where its
// one-round launch has at least g_streamk_min (realise_set_engine(12, n), default 10) K-tiles per workgroup to share out; 0 (default): the
// 128 x 192 two-per-CU kernels only.  Measured (tools/streamk_probe.py, profiles/round5_streamk_probe.log): correct, reproducible,
// and slower than the two-per-CU kernel on every layer shape - DESIGN.md section 6.6.
// Round 6: the kernel lives in the PROBE build only (librealise_hip_probes.so, python -m realise_amd.build --probes), like every other
// measured-and-rejected variant; in the production library the knob stays 0 and no exchange buffers are planned.
static int g_streamk = 0;
static int g_streamk_min = 10;
void set_streamk(int v) { g_streamk = (RL_PROBES && v > 0) ? 1 : 0; }
void set_streamk_min(int n) { g_streamk_min = n > 0 ? n : 0; }
// K7 (realise_set_engine key 13, default 1; models.py:831-834): the glyph lookup `char_images_multifonts.index_select(0, ids)` is fused
// into the loaders of block 1's two forward convolutions - they gather the 3x3 / 1x1 taps straight from the NHWC glyph table through
// the list of distinct ids (ConvLoader::img_index: one index load per 8-row piece in the tile prologue, nothing in the K loop) - so
// the forward materialises no [U, 32, 32, 8] image batch.  A TRAINING step still needs the gathered images as the B operand of block
// 1's two weight-gradient reductions (an index load inside that kernel's K loop costs a full drain per fetch): they are
// gathered at the head of the backward, off the forward's path.  0 = gather_images in the forward, dense loaders (round 2-5 form).
static int g_glyph_fuse = 1;
void set_glyph_fuse(int on) { g_glyph_fuse = on; }
// K9, evaluation mode (realise_set_engine key 14): BatchNorm on running statistics applied in the convolutions' epilogues (resnet_forward)
static int g_bn_fold = 1;
void set_bn_fold(int on) { g_bn_fold = on; }
// realise_engine_adamw_pipelined runs as such (1, default; 2 / 3: diagnostics - every reader waits for the whole sweep / the sliced
// launches on the caller's stream) or as the plain sweep on the caller's stream (0): realise_set_engine key 15
static int g_opt_pipe = 1;
void set_opt_pipe(int on) { g_opt_pipe = on; }
static int g_stream_pri[3] = {0, 0, 0};
void set_stream_priority(int which, int pri) { if (which >= 0 && which < 3) g_stream_pri[which] = pri < 0 ? -1 : (pri > 0 ? 1 : 0); }
static hipError_t create_stream(hipStream_t* s, int which) {
  if (g_stream_pri[which] == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, g_stream_pri[which] < 0 ? greatest : least);
}

#define RL_TRY(expr) do { const int _rc = (expr); if (_rc != RL_OK) { fprintf(stderr, "[realise_hip] %s failed (%d) at %s:%d\n", #expr, _rc, __FILE__, __LINE__); return _rc; } } while (0)

static constexpr int64_t TN_SLAB_ELEMS = 16LL << 20;   // fp32 partial slabs of the split weight-gradient reductions (64 MiB)
static inline int64_t al256(int64_t x) { return (x + 255) & ~(int64_t)255; }
struct Bump {
  int64_t off = 0;
  int64_t take(int64_t bytes) { const int64_t o = off; off = al256(off + bytes); return o; }
};

static inline uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <typename T> struct Engine : EngineBase {
  realise_config cfg;
  Layout L;
  float *P = nullptr, *G = nullptr, *PU = nullptr, *FZ = nullptr, *BF = nullptr;
  int64_t* BI = nullptr;
  char* sh = nullptr;      // shadow arena
  char* ws = nullptr;      // workspace
  int64_t ws_bytes = 0;
  int H, nh, I, V;
  int Vp;                  // V rounded up to 64: row pitch of dlogits and of the classifier's W^T shadow (128-byte aligned rows)

  // ---------------------------------------------------------------- shadows (offsets in bytes)
  struct LayerSh { int64_t qkv_w, qkv_wT, ao_w, ao_wT, in_w, in_wT, out_w, out_wT; };
  struct BlockSh { int64_t w1f, w1d, w1p, w2f, w2d, wsf, wsd; int cin_pad; };
  std::vector<LayerSh> sh_bert, sh_pho, sh_out;
  BlockSh sh_blk[5];
  int64_t sh_cls_w = 0, sh_cls_wT = 0, sh_gru_hh = 0, sh_gru_hhT = 0, sh_glyph = 0;
  int64_t shadow_total = 0;
  bool glyph_built = false, descs_built = false;
  int64_t sh_descs = 0, sh_fill = 0, sh_skip = 0;
  bool skip_built = false;
  static constexpr int ACHUNK_MAX = 8192, ACHUNK_LEN = 32768;
  std::vector<FillChunk> achunk_host;
  int n_achunks = 0;
  static constexpr int FILL_MAX = 2048;
  int n_descs = 0, desc_tiles = 0;
  std::vector<CastDesc> desc_host;

  // ---------------------------------------------------------------- workspace plan (byte offsets)
  struct LayerAct { int64_t qkv, lse, ctx, s1, rstd1, y1, pre, post, s2, rstd2, y2; };
  struct StackAct { int64_t emb_y, emb_xhat, emb_rstd; std::vector<LayerAct> layers; };
  struct BnAct { int64_t mean, rstd, scale, shift; };
  struct BlockAct { int64_t c1, h1, c2, cs, out; BnAct bn1, bn2, bns; int Hin, Hout, Pout; };
  struct Plan {
    int B = 0, S = 0, Tp = 0;
    int64_t total = 0;
    StackAct bert, pho, outb;
    BlockAct blk[5];
    int64_t mask_add, out_d, dlogits, count, loss_internal;
    int64_t cls_act, cls_inv, cls_nact, cls_xc, cls_gc, cls_slab;
    int64_t sk_part[3] = {0, 0, 0}, sk_flag[3] = {0, 0, 0};       // stream-K GEMMs: exchange buffer + workgroup flags, one set per stack (= stream)
    int64_t ln_part[3] = {0, 0, 0}, ln_flag[3] = {0, 0, 0};       // fused GEMM + LayerNorm: per-row tile partials + arrival counters, one set per stack (= stream)
    int64_t live_rows = 0;                // ascending list of the live rows (row-granular layer GEMMs)
    int64_t row_live, live_t64, live_t32, live_t16, live_n, live_rlen;          // padding rows: exact-zero gradient rows the backward skips (row_liveness)     // classifier backward over the rows that enter the loss only (stage_head)
    int64_t ids_clean = 0, pho_clean = 0;                 // range-checked copies of src_idx / pho_idx (sanitize_ids)
    int64_t gru_table, gru_hs, gru_rzn, gru_gh, gru_out;
    int64_t res_xhat, res_rstd, res_h, gate_mean, gate_msum, gate_g, fused;
    int64_t bn_sums, bn_slots;
    int64_t gu_first, gu_flag, gu_ids, gu_counts, gu_inv, gu_bounds, seg_acc, gu_dense;   // glyph dedup
    // backward scratch
    int64_t gA, gB, gC, gE, gD, gF, rowdot, X1, X2, X3, dz, tn_slab, ln_slots;
    int64_t wC1[2], wC2[2], wD[2], wF[2], tn_slab2;      // per-parity dY copies + second slab: weight gradients on the side stream
    // backward scratch of the three concurrent branches: 0 = bert (the fields above), 1 = pho_model + GRU, 2 = glyph ResNet
    struct Scratch { int64_t gB, gE, rowdot, tn_slab, tn_slab2, ln_slots, wC1[2], wC2[2], wD[2], wF[2]; } sc[3];
    std::vector<std::pair<int64_t, int64_t>> zero_once;      // (offset, bytes): self-cleaning accumulators, zero-filled when the plan is installed
    int64_t gru_dh, gru_dgi, gru_dgh, gru_onehot, gru_dtable;
    int64_t r_dout, r_dc2, r_dcs, r_dh1, r_dc1, r_dx;
  } pl;
  std::map<std::string, std::pair<int64_t, int64_t>> taps;   // name -> (byte offset, numel)

  // last forward
  realise_batch last;
  std::vector<int> last_alive;
  // weight-gradient overlap: the four wgrad GEMMs of a BERT layer run on an engine-owned side stream while the caller's
  // stream continues with the data-gradient chain (see layers_backward)
  hipStream_t side = nullptr;
  hipEvent_t ev_ready[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
  hipEvent_t ev_done[2] = {nullptr, nullptr};
  bool done_pending[2] = {false, false};
  int bw_layers = 0;
  bool side_ok() {
    if (side != nullptr) return true;
    if (create_stream(&side, 2) != hipSuccess) { side = nullptr; return false; }
    for (int p = 0; p < 2; ++p) {
      for (int k = 0; k < 4; ++k) (void)hipEventCreateWithFlags(&ev_ready[p][k], hipEventDisableTiming);
      (void)hipEventCreateWithFlags(&ev_done[p], hipEventDisableTiming);
    }
    return true;
  }
  int join_side(hipStream_t st) {
    for (int p = 0; p < 2; ++p)
      if (done_pending[p]) { if (hipStreamWaitEvent(st, ev_done[p], 0) != hipSuccess) return RL_ERR_LAUNCH; done_pending[p] = false; }
    return RL_OK;
  }
  ~Engine() override {
    if (bst[0] != nullptr) {
      for (int k = 0; k < 2; ++k) { (void)hipStreamSynchronize(bst[k]); (void)hipStreamDestroy(bst[k]); (void)hipEventDestroy(ev_join[k]); }
      (void)hipEventDestroy(ev_fork);
    }
    if (side != nullptr) {
      (void)hipStreamSynchronize(side);
      for (int p = 0; p < 2; ++p) { for (int k = 0; k < 4; ++k) (void)hipEventDestroy(ev_ready[p][k]); (void)hipEventDestroy(ev_done[p]); }
      (void)hipStreamDestroy(side);
    }
    for (hipEvent_t e : ev_sig) if (e != nullptr) (void)hipEventDestroy(e);
    if (ev_grads != nullptr) (void)hipEventDestroy(ev_grads);
    for (int k = 0; k < OPT_EV_MAX; ++k) if (ev_opt[k] != nullptr) (void)hipEventDestroy(ev_opt[k]);
    for (int k = 0; k < 2; ++k) if (ev_shadow[k] != nullptr) (void)hipEventDestroy(ev_shadow[k]);
  }
  // branch overlap: two engine-owned streams next to the caller's, fork / join events
  hipStream_t bst[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  int cs = 0;                              // scratch set the backward helpers currently enqueue with (pl.sc[cs])
  bool branches_ok() {
    if (bst[0] != nullptr) return true;
    for (int k = 0; k < 2; ++k)
      if (create_stream(&bst[k], k) != hipSuccess) { bst[0] = bst[1] = nullptr; return false; }
    (void)hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
    for (int k = 0; k < 2; ++k) (void)hipEventCreateWithFlags(&ev_join[k], hipEventDisableTiming);
    return true;
  }
  int fork(hipStream_t st) {
    if (hipEventRecord(ev_fork, st) != hipSuccess) return RL_ERR_LAUNCH;
    for (int k = 0; k < 2; ++k) if (hipStreamWaitEvent(bst[k], ev_fork, 0) != hipSuccess) return RL_ERR_LAUNCH;
    return RL_OK;
  }
  int join(hipStream_t st) {
    for (int k = 0; k < 2; ++k)
      if (hipEventRecord(ev_join[k], bst[k]) != hipSuccess || hipStreamWaitEvent(st, ev_join[k], 0) != hipSuccess) return RL_ERR_LAUNCH;
    return RL_OK;
  }
  const int32_t* alive_dev = nullptr;      // device-side n_alive[Tp] of the last batch (nullptr: host counts)
  bool have_fwd = false;
  bool dead_ok = false;                    // the last forward produced the row-liveness tables (training batch, B * S % 64 == 0)
  const uint8_t* live_rows() const { return dead_ok ? wp<uint8_t>(pl.row_live) : nullptr; }
  bool live16() const { return sizeof(T) == 2 && g_skip_dead == 1; }
  const int* live_tiles() const { return dead_ok ? wp<int>(sizeof(T) == 2 ? (live16() ? pl.live_t16 : pl.live_t64) : pl.live_t32) : nullptr; }
  const int* live_tile_count() const { return dead_ok ? wp<int>(pl.live_n) + (sizeof(T) == 2 ? (live16() ? 2 : 0) : 1) : nullptr; }
  int live_list_rows() const { return sizeof(T) == 2 ? (live16() ? 16 : 64) : 32; }
  bool rows_live = false;                  // the last forward ran its layer GEMMs over the live 16-row blocks (g_live_rows); the backward follows
  // a layer GEMM over the token rows: the live blocks of a live-row step (no dense fall-back: the dead rows of its operands are stale)
  int nt_sid = -1;                         // stack whose layers are being enqueued (stack_forward / layers_backward): selects the stream-K buffers
  unsigned sk_epoch[3] = {0, 0, 0};
  // whole_blocks: the output is the dY operand of a weight-gradient reduction, which walks whole live 16-row BLOCKS - its rows behind a
  // sentence's last live row inside such a block must be (re)computed (exact zeros: their dY inputs are LayerNorm-backward zeros), not
  // left stale, so that launch keeps the block list (the GELU' data gradient of BertOutput, one of a layer's eight GEMMs).
  int nt_rows(hipStream_t st, const T* A, int64_t lda, const T* B, int64_t ldb, int M, int N, int K, const EpiParams<T>& ep, bool whole_blocks = false) {
    if constexpr (sizeof(T) == 2) {
      EpiParams<T> e2 = ep;
      if (rows_live && g_live_rows == 2 && !whole_blocks) { e2.live_list = wp<int>(pl.live_rows); e2.live_count = wp<int>(pl.live_n) + 3; e2.live_unit = 1; }
      else if (rows_live) { e2.live_list = wp<int>(pl.live_t16); e2.live_count = wp<int>(pl.live_n) + 2; }
      // one round of 256 workgroups over 256 x 192 tiles (stream-K) where there is enough to share out; a stack's launches are
      // serialised by its stream, so one exchange buffer per stack does
#if RL_PROBES
      if (g_streamk && nt_sid >= 0 && nt_sid < 3 && pl.sk_part[nt_sid] != 0 &&
          (int64_t)((M + 255) / 256) * ((N + 191) / 192) * (K / 64) >= (int64_t)g_streamk_min * NT8S_GRID) {
        e2.sk_part = wp<float>(pl.sk_part[nt_sid]); e2.sk_flag = wp<int>(pl.sk_flag[nt_sid]);
        e2.sk_tag = (int)(sk_epoch[nt_sid] % 0x7FFFFFF0u) + 1;
        e2.sk_timeout = id_flag != nullptr ? id_flag + 1 : nullptr;
        const int rc = gemm_nt8s(st, A, lda, B, ldb, M, N, K, e2);
        if (rc == RL_OK) { ++sk_epoch[nt_sid]; return RL_OK; }
        if (rc != RL_ERR_ARG) return rc;
        e2.sk_part = nullptr; e2.sk_flag = nullptr; e2.sk_tag = 0; e2.sk_timeout = nullptr;
      }
#endif
      if (rows_live) return gemm_nt8_live(st, A, lda, B, ldb, M, N, K, e2);
    }
    return gemm_nt<T>(st, A, lda, B, ldb, M, N, K, ep);
  }
  const int* live_rlen() const { return dead_ok ? wp<int>(pl.live_rlen) : nullptr; }
  bool cls_compact = false;                // the last forward wrote compacted classifier-gradient rows (stage_head must match)

  template <typename U> U* wp(int64_t off) const { return (U*)(ws + off); }
  template <typename U> U* sp(int64_t off) const { return (U*)(sh + off); }
  const float* pp(int64_t off) const { return P + off; }
  float* gp(int64_t off) const { return G + off; }

  Engine(const realise_config& c, float* p, float* g, float* pu, float* fz, float* bf, int64_t* bi)
      : cfg(c), L(build_layout(c)), P(p), G(g), PU(pu), FZ(fz), BF(bf), BI(bi) {
    H = c.hidden; nh = c.heads; I = c.intermediate; V = c.vocab; Vp = (V + 63) & ~63;
    plan_shadows();
  }

  static int pad8(int c) { return (c + 7) & ~7; }

  void plan_shadows() {
    Bump b;
    const int64_t e = sizeof(T);
    auto plan_stack = [&](std::vector<LayerSh>& v, int n) {
      v.resize(n);
      for (int l = 0; l < n; ++l) {
        v[l].qkv_w = b.take(3LL * H * H * e); v[l].qkv_wT = b.take(3LL * H * H * e);
        v[l].ao_w = b.take((int64_t)H * H * e); v[l].ao_wT = b.take((int64_t)H * H * e);
        v[l].in_w = b.take((int64_t)I * H * e); v[l].in_wT = b.take((int64_t)I * H * e);
        v[l].out_w = b.take((int64_t)I * H * e); v[l].out_wT = b.take((int64_t)I * H * e);
      }
    };
    plan_stack(sh_bert, cfg.bert_layers);
    sh_cls_w = b.take((int64_t)V * H * e);
    sh_cls_wT = b.take((int64_t)Vp * H * e);      // [H][Vp], columns >= V stay zero (the shadow arena is zero-initialised)
    if (cfg.model_type == 1) {
      plan_stack(sh_pho, cfg.pho_layers);
      plan_stack(sh_out, cfg.out_layers);
      sh_gru_hh = b.take(3LL * H * H * e);
      sh_gru_hhT = b.take(3LL * H * H * e);
      for (int k = 0; k < 5; ++k) {
        const BlockOff& o = L.blocks[k];
        BlockSh& s = sh_blk[k];
        s.cin_pad = pad8(o.cin);
        s.w1f = b.take((int64_t)o.cout * 9 * s.cin_pad * e);
        s.w1d = b.take((int64_t)s.cin_pad * 9 * o.cout * e);
        s.w1p = b.take((int64_t)s.cin_pad * 9 * o.cout * e);
        s.w2f = b.take((int64_t)o.cout * 9 * o.cout * e);
        s.w2d = b.take((int64_t)o.cout * 9 * o.cout * e);
        s.wsf = b.take((int64_t)o.cout * s.cin_pad * e);
        s.wsd = b.take((int64_t)s.cin_pad * o.cout * e);
      }
      const int gs = cfg.glyph_size;
      sh_glyph = b.take((int64_t)V * gs * gs * 8 * e);
    }
    sh_descs = b.take(256 * (int64_t)sizeof(CastDesc));      // device table for the one-launch refresh of the Linear weights
    sh_fill = b.take(FILL_MAX * (int64_t)sizeof(FillChunk));  // chunk table of the fresh-gradient zero fill
    sh_skip = b.take(ACHUNK_MAX * (int64_t)sizeof(FillChunk));   // chunk list of the parameters the tiled AdamW does NOT own (adamw_step)
    shadow_total = b.off;
  }

  int64_t shadow_bytes() const override { return shadow_total; }

  // descriptor table of the Linear weights -> (W, W^T) operand copies: group A = the bert stack (what the forward needs first), group
  // B = classifier, pinyin / output stacks, GRU; tile ids restart per group (one launch each)
  int ensure_descs(hipStream_t st) {
    if (descs_built) return RL_OK;
    std::vector<CastDesc>& d = desc_host;       // member: stays alive while the async upload is in flight
    d.clear();
    int tiles = 0;
    auto add = [&](int64_t src, int R, int C, int64_t dst, int64_t dstT, int ldT = 0) {
      CastDesc x;
      x.src = pp(src); x.dst = sp<T>(dst); x.dstT = sp<T>(dstT); x.R = R; x.C = C; x.ldT = ldT > 0 ? ldT : R;
      x.tile_begin = tiles; x.tiles_c = (C + 63) / 64;
      tiles += x.tiles_c * ((R + 63) / 64);
      d.push_back(x);
    };
    auto add_stack = [&](const StackOff& so, const std::vector<LayerSh>& v) {
      for (size_t l = 0; l < v.size(); ++l) {
        const LayerOff& o = so.layers[l];
        add(o.qkv_w, 3 * H, H, v[l].qkv_w, v[l].qkv_wT);
        add(o.ao_w, H, H, v[l].ao_w, v[l].ao_wT);
        add(o.in_w, I, H, v[l].in_w, v[l].in_wT);
        add(o.out_w, H, I, v[l].out_w, v[l].out_wT);
      }
    };
    add_stack(L.bert, sh_bert);
    n_descs_a = (int)d.size(); desc_tiles_a = tiles;
    tiles = 0;                                   // second group: its own launch, tile ids restart
    add(L.cls_w, V, H, sh_cls_w, sh_cls_wT, Vp);
    if (cfg.model_type == 1) {
      add_stack(L.pho, sh_pho);
      add_stack(L.outb, sh_out);
      add(L.gru_w_hh, 3 * H, H, sh_gru_hh, sh_gru_hhT);
    }
    if (d.size() > 256) return RL_ERR_ARG;
    if (hipMemcpyAsync(sh + sh_descs, d.data(), d.size() * sizeof(CastDesc), hipMemcpyHostToDevice, st) != hipSuccess) return RL_ERR_LAUNCH;
    n_descs = (int)d.size(); desc_tiles = tiles; descs_built = true;
    return RL_OK;
  }
  int refresh_shadows(hipStream_t st) override { return refresh_shadows_ex(st, 0); }
  // skip_linear != 0: the Linear weights' copies are current (adamw_step wrote them from the values it stored); only the conv-weight
  // copies (and the glyph table's image when it changed) are re-derived
  int refresh_shadows_ex(hipStream_t st, int skip_linear) override {
    if (!sh) return RL_ERR_ARG;
    RL_TRY(ensure_descs(st));
    // a pipelined optimizer sweep (adamw_step): the convolution weights this call reads are in its first piece; a full refresh reads everything
    RL_TRY(wait_opt(st, skip_linear ? 0 : -1));
    if (!skip_linear) opt_pending = false;
    // Three independent pieces on three streams (when the branch streams exist): the bert stack's copies on the caller's stream -
    // the forward that follows needs nothing else for its first 12 layers -, classifier + pinyin / output stacks + GRU on the pinyin
    // stream, the conv-weight launch (+ the glyph table when it changed) on the glyph stream, where the glyph branch of the
    // forward queues up behind it by itself.  ev_shadow[k] orders any OTHER consumer (a forward without branch streams, the
    // glyph-only entry points): wait_shadows().
    const bool ovl = cfg.model_type == 1 && g_branch_overlap && branches_ok() && shadow_events_ok();
    hipStream_t s_b = ovl ? bst[0] : st, s_c = ovl ? bst[1] : st;
    if (ovl) RL_TRY(fork(st));
    const CastDesc* dd = (const CastDesc*)(sh + sh_descs);
    if (!skip_linear) {
      RL_TRY(cast_transpose_multi<T>(st, dd, n_descs_a, desc_tiles_a));
      RL_TRY(cast_transpose_multi<T>(s_b, dd + n_descs_a, n_descs - n_descs_a, desc_tiles));
    }
    if (cfg.model_type == 1) {
      ConvShadowDescs cd;
      auto addc = [&](const float* w, int Co, int Ci, int KHW, int Cpad, T* fwd, T* dgrad, const TapOrder& ord = TapOrder()) {
        ConvShadowDesc& x = cd.d[cd.n++];
        x.w = w; x.fwd = fwd; x.dgrad = dgrad; x.Co = Co; x.Ci = Ci; x.KHW = KHW; x.Cpad = Cpad; x.CiRows = Cpad; x.block_begin = 0; x.order = ord;
      };
      for (int k = 0; k < 5; ++k) {
        const BlockOff& o = L.blocks[k];
        const BlockSh& s = sh_blk[k];
        addc(pp(o.w1), o.cout, o.cin, 9, s.cin_pad, sp<T>(s.w1f), sp<T>(s.w1d));
        {   // second data-gradient copy of the stride-2 conv with its taps stored parity class by parity class
          TapOrder ord; ord.n = 9; int first = 0;
          (void)conv_s2_class(3, 3, 1, 0, &first, ord.t);
          addc(pp(o.w1), o.cout, o.cin, 9, s.cin_pad, (T*)nullptr, sp<T>(s.w1p), ord);
        }
        addc(pp(o.w2), o.cout, o.cout, 9, o.cout, sp<T>(s.w2f), sp<T>(s.w2d));
        addc(pp(o.ws), o.cout, o.cin, 1, s.cin_pad, sp<T>(s.wsf), sp<T>(s.wsd));
      }
      RL_TRY(conv_weight_shadow_multi<T>(s_c, cd));
      // the glyph table is frozen (requires_grad=False, models.py:679): its NHWC image is rebuilt only after
      // invalidate_frozen() (load_state_dict / build_glyce_embed*)
      if (!glyph_built) {
        RL_TRY(glyph_shadow<T>(s_c, FZ + L.glyph, V, cfg.num_fonts, cfg.glyph_size * cfg.glyph_size, 8, sp<T>(sh_glyph)));
        glyph_built = true;
      }
    }
    shadows_pending = false;
    if (ovl) {
      for (int k = 0; k < 2; ++k) if (hipEventRecord(ev_shadow[k], bst[k]) != hipSuccess) return RL_ERR_LAUNCH;
      shadows_pending = true;
    }
    return RL_OK;
  }
  // FusedAdamW's step as the engine runs it: the Linear weights (90 % of the parameters) in the tiles of the operand-copy kernel,
  // which stores the new fp32 value AND its compute-dtype W / W^T copies in the same pass; everything else by the flat grouped
  // kernel, told by a byte per 64 parameters what the tiled launches own.  The next forward then calls refresh_shadows_ex(st, 1).
  // ---- pipelined form (round 6, realise_engine_adamw_pipelined).  The sweep is 1.1 ms of pure HBM streaming (32 B per parameter) that
  // nothing could overlap while it sat between the backward and the next forward on the caller's stream.  Here it runs on the
  // engine's side stream in the order the next forward consumes the parameters - everything outside the Linear weights (biases,
  // LayerNorm, embeddings, convolution weights), the tied word table / classifier, BERT layers 0-1 | layers 2-3 | pinyin stack +
  // output stack + GRU | the remaining BERT layers two by two - with an event behind every piece, and the forward (and
  // refresh_shadows_ex) waits for a piece right in front of its first reader: the caller's stream is held for ~0.3 ms instead of the
  // whole sweep, the rest streams under the first layers of the forward.  Same kernels, same arithmetic per element: same bits.
  // Contract: between this call and the next forward / realise_engine_sync_optimizer nothing else may touch parameters, gradients or
  // moments (the module opts in per loop: pipeline_optimizer).
  static constexpr int OPT_EV_MAX = 16;
  hipEvent_t ev_opt[OPT_EV_MAX] = {}; hipEvent_t ev_grads = nullptr;
  int opt_groups = 0;                      // bert layer groups of the pending sweep: events 0 .. opt_groups - 1, event opt_groups = the other stacks
  bool opt_pending = false;
  bool opt_events_ok() {
    if (ev_grads != nullptr) return true;
    if (hipEventCreateWithFlags(&ev_grads, hipEventDisableTiming) != hipSuccess) { ev_grads = nullptr; return false; }
    for (int k = 0; k < OPT_EV_MAX; ++k)
      if (hipEventCreateWithFlags(&ev_opt[k], hipEventDisableTiming) != hipSuccess) return false;
    return true;
  }
  // order stream s behind piece k of the pending sweep (k < 0: behind all of it)
  int wait_opt(hipStream_t s, int k) {
    if (!opt_pending) return RL_OK;
    if (g_opt_pipe >= 2) k = -1;             // (diagnostics: every reader waits for the whole sweep)
    const int lo = k < 0 ? 0 : std::min(k, opt_groups), hi = k < 0 ? opt_groups : lo;
    for (int i = lo; i <= hi; ++i) if (hipStreamWaitEvent(s, ev_opt[i], 0) != hipSuccess) return RL_ERR_LAUNCH;
    return RL_OK;
  }
  int sync_optimizer(hipStream_t st) override {
    RL_TRY(wait_opt(st, -1));
    opt_pending = false;
    return RL_OK;
  }
  int adamw_step(hipStream_t st, float* m, float* v, const uint8_t* group_of_block, const AdamwGroups& gs, const float* norm_sq,
                 float max_norm, int pipelined) override {
    if (!sh || !m || !v || !group_of_block) return RL_ERR_ARG;
    RL_TRY(ensure_descs(st));
    const int64_t n = L.arena_elems[AR_TRAIN];
    if (!skip_built) {
      // everything outside the Linear weights, as chunks of at most ACHUNK_LEN floats (all boundaries are multiples of 64 elements)
      std::vector<std::pair<int64_t, int64_t>> own;
      for (const CastDesc& d : desc_host) {
        const int64_t off = d.src - P, cnt = (int64_t)d.R * d.C;
        if ((off % 64) != 0 || (cnt % 64) != 0 || (d.C % 4) != 0) return RL_ERR_ARG;
        own.push_back({off, off + cnt});
      }
      std::sort(own.begin(), own.end());
      achunk_host.clear();
      int64_t at = 0;
      auto emit = [&](int64_t b, int64_t e) {
        for (int64_t o = b; o < e; o += ACHUNK_LEN) achunk_host.push_back(FillChunk{o, (int32_t)std::min<int64_t>(ACHUNK_LEN, e - o), 0});
      };
      for (auto& r : own) { if (r.first > at) emit(at, r.first); at = std::max(at, r.second); }
      if (at < n) emit(at, n);
      if ((int)achunk_host.size() > ACHUNK_MAX) return RL_ERR_ARG;
      if (!achunk_host.empty() &&
          hipMemcpyAsync(sh + sh_skip, achunk_host.data(), achunk_host.size() * sizeof(FillChunk), hipMemcpyHostToDevice, st) != hipSuccess) return RL_ERR_LAUNCH;
      n_achunks = (int)achunk_host.size();
      skip_built = true;
    }
    RL_TRY(wait_shadows(st));           // (the previous refresh may still be writing the copies on the branch streams)
    const CastDesc* dd = (const CastDesc*)(sh + sh_descs);
    RL_TRY(sync_optimizer(st));         // (a pipelined sweep nobody waited for: two steps without a forward in between)
    const int Lb = (int)sh_bert.size();
    if (pipelined && g_opt_pipe && side_ok() && opt_events_ok() && (Lb + 1) / 2 + 1 <= OPT_EV_MAX && n_descs_a == 4 * Lb) {
      RL_TRY(join_side(st));
      hipStream_t side = g_opt_pipe == 3 ? st : this->side;      // (diagnostics, 3: the sliced launches on the caller's stream)
      if (hipEventRecord(ev_grads, st) != hipSuccess || hipStreamWaitEvent(side, ev_grads, 0) != hipSuccess) return RL_ERR_LAUNCH;
      auto tile_begin = [&](int i, bool group_b) -> int {      // descriptor i of group A (0 .. n_descs_a) / B (0 .. n_descs - n_descs_a); the end = the group's tile count
        if (!group_b) return i < n_descs_a ? desc_host[i].tile_begin : desc_tiles_a;
        return n_descs_a + i < n_descs ? desc_host[n_descs_a + i].tile_begin : desc_tiles;
      };
      auto bert_layers = [&](int l0, int l1) -> int {          // the Linear weights of BERT layers [l0, l1)
        const int a = 4 * l0, b = 4 * std::min(l1, Lb);
        if (b <= a) return RL_OK;
        return adamw_cast_multi<T>(side, dd + a, b - a, tile_begin(b, false) - tile_begin(a, false), P, G, m, v, group_of_block, gs, norm_sq, max_norm, tile_begin(a, false));
      };
      const int nb = n_descs - n_descs_a;                      // group B: [0] = tied word table / classifier, then pinyin / output stacks, GRU
      opt_groups = (Lb + 1) / 2;
      RL_TRY(adamw_chunks(side, P, G, m, v, (const FillChunk*)(sh + sh_skip), n_achunks, group_of_block, gs, norm_sq, max_norm));
      RL_TRY(adamw_cast_multi<T>(side, dd + n_descs_a, 1, tile_begin(1, true), P, G, m, v, group_of_block, gs, norm_sq, max_norm, 0));
      RL_TRY(bert_layers(0, 2));
      if (hipEventRecord(ev_opt[0], side) != hipSuccess) return RL_ERR_LAUNCH;
      for (int g = 1; g < opt_groups; ++g) {
        RL_TRY(bert_layers(2 * g, 2 * g + 2));
        if (hipEventRecord(ev_opt[g], side) != hipSuccess) return RL_ERR_LAUNCH;
        if (g == 1 && nb > 1) {
          RL_TRY(adamw_cast_multi<T>(side, dd + n_descs_a + 1, nb - 1, desc_tiles - tile_begin(1, true), P, G, m, v, group_of_block, gs, norm_sq, max_norm, tile_begin(1, true)));
          if (hipEventRecord(ev_opt[opt_groups], side) != hipSuccess) return RL_ERR_LAUNCH;
        }
      }
      if (opt_groups < 2 || nb <= 1) {                          // (a one- or two-layer stack, or no other stacks: the last event still has to exist)
        if (opt_groups < 2 && nb > 1)
          RL_TRY(adamw_cast_multi<T>(side, dd + n_descs_a + 1, nb - 1, desc_tiles - tile_begin(1, true), P, G, m, v, group_of_block, gs, norm_sq, max_norm, tile_begin(1, true)));
        if (hipEventRecord(ev_opt[opt_groups], side) != hipSuccess) return RL_ERR_LAUNCH;
      }
      opt_pending = true;
      return RL_OK;
    }
    RL_TRY(adamw_cast_multi<T>(st, dd, n_descs_a, desc_tiles_a, P, G, m, v, group_of_block, gs, norm_sq, max_norm));
    RL_TRY(adamw_cast_multi<T>(st, dd + n_descs_a, n_descs - n_descs_a, desc_tiles, P, G, m, v, group_of_block, gs, norm_sq, max_norm));
    return adamw_chunks(st, P, G, m, v, (const FillChunk*)(sh + sh_skip), n_achunks, group_of_block, gs, norm_sq, max_norm);
  }
  // operand copies refreshed on the branch streams: order `s` behind them (no-op when they ran on the caller's stream)
  hipEvent_t ev_shadow[2] = {nullptr, nullptr};
  bool shadows_pending = false;
  int n_descs_a = 0, desc_tiles_a = 0;
  bool shadow_events_ok() {
    for (int k = 0; k < 2; ++k)
      if (ev_shadow[k] == nullptr && hipEventCreateWithFlags(&ev_shadow[k], hipEventDisableTiming) != hipSuccess) return false;
    return true;
  }
  int wait_shadows(hipStream_t s) {
    if (!shadows_pending) return RL_OK;
    for (int k = 0; k < 2; ++k) if (hipStreamWaitEvent(s, ev_shadow[k], 0) != hipSuccess) return RL_ERR_LAUNCH;
    return RL_OK;
  }

  // ---------------------------------------------------------------- workspace planning
  void tap(const std::string& name, int64_t off, int64_t numel) { taps[name] = {off, numel}; }

  // Tp < 0: glyph-only plan (BASELINE configs[3]: the CharResNet alone on B*S glyph stacks) - no BERT stacks, no logits
  Plan make_plan(int B, int S, int Tp) {
    Plan p;
    const bool glyph_only = Tp < 0;
    p.B = B; p.S = S; p.Tp = Tp;
    Bump b;
    const int64_t e = sizeof(T), Tk = (int64_t)B * S;
    taps.clear();
    auto plan_stack = [&](StackAct& a, int n, const std::string& name) {
      a.emb_y = b.take(Tk * H * e); a.emb_xhat = b.take(Tk * H * e); a.emb_rstd = b.take(Tk * 4);
      tap(name + ".emb", a.emb_y, Tk * H);
      a.layers.resize(n);
      for (int l = 0; l < n; ++l) {
        LayerAct& x = a.layers[l];
        x.qkv = b.take(Tk * 3 * H * e); x.lse = b.take((int64_t)B * nh * S * 4); x.ctx = b.take(Tk * H * e);
        x.s1 = b.take(Tk * H * e); x.rstd1 = b.take(Tk * 4); x.y1 = b.take(Tk * H * e);
        x.pre = b.take(Tk * I * e); x.post = b.take(Tk * I * e);
        x.s2 = b.take(Tk * H * e); x.rstd2 = b.take(Tk * 4); x.y2 = b.take(Tk * H * e);
        const std::string ln = name + ".layer." + std::to_string(l);
        tap(ln + ".qkv", x.qkv, Tk * 3 * H); tap(ln + ".ctx", x.ctx, Tk * H); tap(ln + ".attn_out", x.y1, Tk * H);
        tap(ln + ".inter", x.post, Tk * I); tap(ln + ".out", x.y2, Tk * H);
        tap(ln + ".sum1", x.s1, Tk * H); tap(ln + ".sum2", x.s2, Tk * H);      // (pre-LayerNorm sums; after a training forward: their xhat)
      }
    };
    p.mask_add = b.take(Tk * 4);
    p.ids_clean = b.take(Tk * 8);
    p.pho_clean = (!glyph_only && cfg.model_type == 1) ? b.take(Tk * (int64_t)(Tp > 0 ? Tp : 1) * 8) : 0;
    if (!glyph_only) {
      plan_stack(p.bert, cfg.bert_layers, "bert");
      p.out_d = b.take(Tk * H * e);
      p.dlogits = b.take(Tk * Vp * e);
      tap("dlogits", p.dlogits, Tk * Vp);
      p.zero_once.push_back({p.dlogits, Tk * Vp * e});      // compacted gradient rows: the rows beyond the live count hold old (finite) values
    }
    p.count = b.take(256);
    p.loss_internal = b.take(Tk * 4 + 256);       // per-row loss terms (ordered fold: reproducible loss)
    p.cls_act = b.take(Tk * 4); p.cls_inv = b.take(Tk * 4); p.cls_nact = b.take(256);
    p.row_live = b.take(Tk + 64); p.live_t64 = b.take((Tk / 32 + 2) * 4); p.live_t32 = b.take((Tk / 32 + 2) * 4); p.live_t16 = b.take((Tk / 16 + 4) * 4); p.live_n = b.take(256); p.live_rlen = b.take(B * 4 + 64); p.live_rows = b.take((Tk + 128) * 4);
    p.cls_xc = b.take(Tk * H * e); p.cls_gc = b.take(Tk * H * e);
    p.zero_once.push_back({p.cls_xc, Tk * H * e}); p.zero_once.push_back({p.cls_gc, Tk * H * e});
    p.cls_slab = (!glyph_only && sizeof(T) == 2) ? b.take(4 * Tk * H * 4) : 0;      // fp32 planes of the split-K classifier data gradient
    if (!glyph_only && sizeof(T) == 2) {
      for (int k = 0; k < 3; ++k) {
        if (g_streamk) {       // (default off, DESIGN 6.6: 150 MB of exchange buffers are planned only for a plan built with the knob set)
          p.sk_part[k] = b.take(NT8S_PART_BYTES); p.sk_flag[k] = b.take((int64_t)NT8S_GRID * NT8S_FLAG_STRIDE * 4 + 256);      // (flags: zero with the workspace; tags are never 0)
        }
        p.ln_part[k] = b.take(Tk * 8 * 2 * 8);
        p.ln_flag[k] = 0;
        p.zero_once.push_back({p.ln_part[k], Tk * 8 * 2 * 8});
      }
    }
    // shared backward scratch
    p.gA = b.take(Tk * H * e); p.gB = b.take(Tk * H * e); p.gC = b.take(Tk * H * e); p.gE = b.take(Tk * H * e);
    const int64_t Tw = glyph_only ? 1 : Tk;               // the wide BERT scratch is not needed by the glyph-only plan
    p.gD = b.take(Tw * I * e); p.gF = b.take(Tw * 3 * H * e); p.rowdot = b.take((int64_t)B * nh * S * 4);
    // operands of the deferred weight gradients: a layer's four dY matrices live in the buffer set of its parity until the
    // side stream has consumed them (set 0 aliases the classic gC / gD / gF scratch)
    p.wC1[0] = p.gC; p.wD[0] = p.gD; p.wF[0] = p.gF; p.wC2[0] = b.take(Tw * H * e);
    p.wC1[1] = b.take(Tw * H * e); p.wC2[1] = b.take(Tw * H * e); p.wD[1] = b.take(Tw * I * e); p.wF[1] = b.take(Tw * 3 * H * e);
    p.tn_slab2 = b.take(TN_SLAB_ELEMS * 4);
    p.tn_slab = b.take(TN_SLAB_ELEMS * 4);
    p.ln_slots = b.take((int64_t)(LN_FOLD_MAX + 1) * LN_SLOT_BYTES);     // LN_FOLD_MAX deferred sites + one immediate region
    tap("d_x0", p.gB, Tk * H);
    {
      typename Plan::Scratch& s0 = p.sc[0];
      s0.gB = p.gB; s0.gE = p.gE; s0.rowdot = p.rowdot; s0.tn_slab = p.tn_slab; s0.tn_slab2 = p.tn_slab2; s0.ln_slots = p.ln_slots;
      for (int k = 0; k < 2; ++k) { s0.wC1[k] = p.wC1[k]; s0.wC2[k] = p.wC2[k]; s0.wD[k] = p.wD[k]; s0.wF[k] = p.wF[k]; }
      p.sc[1] = s0; p.sc[2] = s0;
      if (cfg.model_type == 1) {
        if (!glyph_only) {          // pho branch: a full private set (its layers run next to the bert layers)
          typename Plan::Scratch& s1 = p.sc[1];
          s1.gB = b.take(Tk * H * e); s1.gE = b.take(Tk * H * e); s1.rowdot = b.take((int64_t)B * nh * S * 4);
          s1.tn_slab = b.take(TN_SLAB_ELEMS * 4); s1.tn_slab2 = s1.tn_slab; s1.ln_slots = b.take((int64_t)(LN_FOLD_MAX + 1) * LN_SLOT_BYTES);
          s1.wC1[0] = b.take(Tk * H * e); s1.wC2[0] = b.take(Tk * H * e); s1.wD[0] = b.take(Tk * I * e); s1.wF[0] = b.take(Tk * 3 * H * e);
          s1.wC1[1] = s1.wC1[0]; s1.wC2[1] = s1.wC2[0]; s1.wD[1] = s1.wD[0]; s1.wF[1] = s1.wF[0];
        }
        typename Plan::Scratch& s2 = p.sc[2];   // glyph branch: LayerNorm-backward output, slabs of the conv weight gradients
        s2.gE = b.take(Tk * H * e); s2.tn_slab = b.take(TN_SLAB_ELEMS * 4); s2.ln_slots = b.take((int64_t)(LN_FOLD_MAX + 1) * LN_SLOT_BYTES);
      }
    }
    if (cfg.model_type == 1) {
      if (!glyph_only) {
        plan_stack(p.pho, cfg.pho_layers, "pho_model");
        plan_stack(p.outb, cfg.out_layers, "output_block");
        p.gru_table = b.take(64LL * 3 * H * 4);
        p.gru_hs = b.take((int64_t)Tp * Tk * H * e);
        p.gru_rzn = b.take((int64_t)Tp * Tk * 3 * H * e);
        p.gru_gh = b.take((int64_t)Tp * Tk * 3 * H * e);
        p.gru_out = b.take(Tk * H * e);
        tap("pho_gru", p.gru_out, Tk * H);
      }
      int hin = cfg.glyph_size;
      for (int k = 0; k < 5; ++k) {
        BlockAct& a = p.blk[k];
        const int C = L.blocks[k].cout;
        a.Hin = hin; a.Hout = hin / 2; a.Pout = (int)(Tk * a.Hout * a.Hout);
        const int64_t n = (int64_t)a.Pout * C * e;
        a.c1 = b.take(n); a.h1 = b.take(n); a.c2 = b.take(n); a.cs = b.take(n); a.out = b.take(n);
        for (BnAct* q : {&a.bn1, &a.bn2, &a.bns}) {
          q->mean = b.take(C * 4); q->rstd = b.take(C * 4); q->scale = b.take(C * 4); q->shift = b.take(C * 4);
        }
        tap("resnet.block" + std::to_string(k + 1), a.out, (int64_t)a.Pout * C);
        tap("resnet.block" + std::to_string(k + 1) + ".h1", a.h1, (int64_t)a.Pout * C);
        hin = a.Hout;
      }
      p.bn_sums = b.take(2 * 1024 * 4);
      p.bn_slots = b.take((int64_t)COL_SLOT_BYTES);
      p.gu_first = b.take((int64_t)V * 4); p.gu_flag = b.take(Tk * 4); p.gu_ids = b.take(Tk * 8); p.gu_counts = b.take(Tk * 4);
      p.gu_inv = b.take(Tk * 4); p.gu_bounds = b.take(64); p.seg_acc = b.take(Tk * H * 4);
      p.gu_dense = b.take(Tk * (int64_t)cfg.glyph_size * cfg.glyph_size * 8 * e);      // the distinct glyph images, contiguous (NHWC, 8 channels)
      tap("glyph.bounds", p.gu_bounds, 64 / (int64_t)sizeof(T)); tap("glyph.inv", p.gu_inv, Tk * 4 / (int64_t)sizeof(T));
      tap("glyph.counts", p.gu_counts, Tk * 4 / (int64_t)sizeof(T)); tap("glyph.ids", p.gu_ids, Tk * 8 / (int64_t)sizeof(T));
      p.res_xhat = b.take(Tk * H * e); p.res_rstd = b.take(Tk * 4); p.res_h = b.take(Tk * H * e);
      p.gate_mean = b.take((int64_t)B * H * 4); p.gate_msum = b.take(B * 4 + 256); p.gate_g = b.take(Tk * 16);
      p.fused = b.take(Tk * H * e);
      tap("res_h", p.res_h, Tk * H); tap("fused", p.fused, Tk * H);
      p.X1 = b.take(Tk * H * e); p.X2 = b.take(Tk * H * e); p.X3 = b.take(Tk * H * e); p.dz = b.take(Tk * 16);
      p.gru_dh = b.take(Tw * H * e); p.gru_dgi = b.take(Tw * 3 * H * e); p.gru_dgh = b.take(Tw * 3 * H * e);
      p.gru_onehot = b.take(Tw * 64 * e); p.gru_dtable = b.take(64LL * 3 * H * 4);
      const int64_t big = (int64_t)p.blk[0].Pout * 64 * e;     // block 1 is the largest activation
      p.r_dout = b.take(big); p.r_dc2 = b.take(big); p.r_dcs = b.take(big); p.r_dh1 = b.take(big); p.r_dc1 = b.take(big);
      p.r_dx = b.take(big);
    }
    {   // named views of the scratch regions (diagnostics: tools/diag_stale.py zeroes them one at a time)
      const int64_t e1 = (int64_t)sizeof(T);
      auto stap = [&](const char* n, int64_t off, int64_t bytes) { tap(std::string("scratch.") + n, off, bytes / e1); };
      stap("tn_slab", p.tn_slab, TN_SLAB_ELEMS * 4); stap("ln_slots", p.ln_slots, LN_SLOT_BYTES);
      stap("gA", p.gA, Tk * H * e); stap("gB", p.gB, Tk * H * e); stap("gC", p.gC, Tk * H * e); stap("gE", p.gE, Tk * H * e);
      stap("gD", p.gD, Tw * I * e); stap("gF", p.gF, Tw * 3 * H * e);
      if (cfg.model_type == 1) {
        const int64_t big = (int64_t)p.blk[0].Pout * 64 * e;
        stap("r_dout", p.r_dout, big); stap("r_dc2", p.r_dc2, big); stap("r_dcs", p.r_dcs, big); stap("r_dh1", p.r_dh1, big);
        stap("r_dc1", p.r_dc1, big); stap("r_dx", p.r_dx, big); stap("bn_sums", p.bn_sums, 2 * 1024 * 4);
        stap("bn_slots", p.bn_slots, (int64_t)COL_SLOT_BYTES); stap("seg_acc", p.seg_acc, Tk * H * 4);
        stap("X1", p.X1, Tk * H * e); stap("X2", p.X2, Tk * H * e); stap("X3", p.X3, Tk * H * e);
      }
    }
    p.total = b.off;
    return p;
  }

  int64_t workspace_bytes(int B, int S, int Tp) override {
    std::map<std::string, std::pair<int64_t, int64_t>> keep = taps;
    const Plan p = make_plan(B, S, Tp != 0 ? Tp : 1);
    taps = keep;
    return p.total;
  }
  // Workspace slots (round 6): a plan lives IN a workspace buffer (self-cleaning accumulators at zero, finite
  // stale rows where a live-row step leaves them), so the engine remembers, per caller buffer it has been bound to, the plan installed
  // there.  A caller that keeps one buffer per (B, S, Tp) key - modeling.py: train batch, eval batch, the short last batch, the
  // glyph-only plan - pays the zero fill of install_plan ONCE per key; re-binding a remembered buffer restores its plan as it was.
  // forget_workspace() before the caller frees a buffer (an allocator may hand the same address out again).
  struct Slot { char* ws; int64_t bytes; Plan pl; std::map<std::string, std::pair<int64_t, int64_t>> taps; int ln_epoch[3]; unsigned sk_epoch[3]; uint64_t stamp; };
  std::vector<Slot> slots;
  uint64_t slot_clock = 0;
  static constexpr size_t MAX_SLOTS = 8;
  void stash_current() {
    if (ws == nullptr) return;
    Slot* s = nullptr;
    for (Slot& x : slots) if (x.ws == ws) s = &x;
    if (s == nullptr) {
      if (slots.size() >= MAX_SLOTS) {       // (callers forget what they free; this bound only guards a caller that never does)
        size_t old = 0;
        for (size_t i = 1; i < slots.size(); ++i) if (slots[i].stamp < slots[old].stamp) old = i;
        slots.erase(slots.begin() + old);
      }
      slots.push_back(Slot());
      s = &slots.back();
    }
    s->ws = ws; s->bytes = ws_bytes; s->pl = pl; s->taps = taps; s->stamp = ++slot_clock;
    for (int k = 0; k < 3; ++k) { s->ln_epoch[k] = ln_epoch[k]; s->sk_epoch[k] = sk_epoch[k]; }
  }
  int bind(void* shadow, void* workspace, int64_t bytes) override {
    // everything the engine keeps inside the caller's shadow buffer (operand copies, the cast descriptor table, the chunk table of the
    // fresh-gradient zero fill) is rebuilt after a re-bind to another buffer
    if (shadow != (void*)sh) { glyph_built = false; descs_built = false; fill_built = false; n_fill = 0; skip_built = false; }
    stash_current();
    sh = (char*)shadow; ws = (char*)workspace; ws_bytes = bytes; pl = Plan(); have_fwd = false; have_glyph_fwd = false;
    ln_epoch[0] = ln_epoch[1] = ln_epoch[2] = 0;
    for (Slot& x : slots)
      if (x.ws == ws && x.bytes == bytes) {
        pl = x.pl; taps = x.taps; x.stamp = ++slot_clock;
        for (int k = 0; k < 3; ++k) { ln_epoch[k] = x.ln_epoch[k]; sk_epoch[k] = x.sk_epoch[k]; }
        break;
      }
    return RL_OK;
  }
  void forget_workspace(void* workspace) override {
    for (size_t i = 0; i < slots.size();) { if (slots[i].ws == (char*)workspace) slots.erase(slots.begin() + i); else ++i; }
    if (ws == (char*)workspace) { ws = nullptr; ws_bytes = 0; pl = Plan(); have_fwd = false; have_glyph_fwd = false; }
  }
  void invalidate_frozen() override { glyph_built = false; }
  int* id_flag = nullptr;
  void set_id_flag(int* flag) override { id_flag = flag; }
  int get_tap(const char* name, void** ptr, int64_t* numel) override {
    auto it = taps.find(name);
    if (it == taps.end() || !ws) return RL_ERR_ARG;
    *ptr = ws + it->second.first; *numel = it->second.second;
    return RL_OK;
  }

  // ---------------------------------------------------------------- dropout sites
  DropParams site(int id, float p) const {
    DropParams d;
    if (!last.training || p <= 0.f) return d;
    d.seed = (uint32_t)splitmix(last.seed * 0x100000001B3ull + (uint64_t)id);
    d.thresh = (uint32_t)((double)p * 4294967296.0);
    d.scale = 1.0f / (1.0f - p);
    return d;
  }
  static void set_drop(EpiParams<T>& ep, const DropParams& d) { ep.drop_seed = d.seed; ep.drop_thresh = d.thresh; ep.drop_scale = d.scale; }

  // ---------------------------------------------------------------- BERT stack
  int stack_forward(hipStream_t st, int sid, const StackOff& so, const std::vector<LayerSh>& shs, StackAct& a,
                    const int64_t* ids, const T* embeds, int pos_zero, const T** out) {
    const int B = pl.B, S = pl.S, Tk = B * S;
    nt_sid = sid;
    {
      LnFwdArgs<T> ln;
      ln.rows = Tk; ln.H = H; ln.S = S;
      ln.in_mode = ids ? 1 : 2; ln.x = embeds; ln.ids = ids;
      ln.word = so.word >= 0 ? pp(so.word) : nullptr;
      ln.pos = pp(so.pos); ln.type0 = pp(so.type); ln.pos_zero = pos_zero;
      ln.gamma = pp(so.ln_g); ln.beta = pp(so.ln_b); ln.eps = cfg.ln_eps;
      ln.y = wp<T>(a.emb_y); ln.xhat = bwd_follows() ? wp<T>(a.emb_xhat) : nullptr; ln.rstd = wp<float>(a.emb_rstd);      // (xhat: the backward's operand)
      ln.drop = site(sid * 1000 + 900, cfg.hidden_dropout);
      RL_TRY(ln_fwd<T>(st, ln));
    }
    const T* x = wp<T>(a.emb_y);
    for (size_t l = 0; l < a.layers.size(); ++l) {
      // a pipelined optimizer sweep (adamw_step) steps the BERT layers two by two: wait in front of a pair's first reader
      if (sid == 0 && l >= 2 && (l % 2) == 0) RL_TRY(wait_opt(st, (int)l / 2));
      const LayerOff& o = so.layers[l];
      const LayerSh& w = shs[l];
      LayerAct& t = a.layers[l];
      {  // fused QKV projection (modeling_bert.py:221,231-232)
        EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = wp<T>(t.qkv); ep.ldo = 3 * H; ep.bias = pp(o.qkv_b);
        RL_TRY(nt_rows(st, x, H, sp<T>(w.qkv_w), H, Tk, 3 * H, H, ep));
      }
      {
        const DropParams d = site(sid * 1000 + (int)l * 10 + 1, cfg.attn_dropout);
        const T* q = wp<T>(t.qkv);
        // INVARIANT: rlen goes to the attention FORWARD on live-row steps only.  With rlen the kernels leave the
        // ctx / lse rows in [rlen[b], S) unwritten (attn_fwd_long: in some waves, zeros from others): only a step whose every consumer
        // skips or masks those rows (rows_live: the layer GEMMs walk the live blocks, the backward masks what it reads) may pass it;
        // a dense forward (evaluation, taps, engine:10=0) always gets nullptr and writes every row.
        RL_TRY(attn_fwd<T>(st, q, q + H, q + 2 * H, 3 * H, wp<float>(pl.mask_add), wp<T>(t.ctx), H, wp<float>(t.lse), B, nh, S,
                           d.seed, d.thresh, d.scale, rows_live ? live_rlen() : nullptr));
      }
      // BertSelfOutput: dense -> dropout -> + input -> LayerNorm (modeling_bert.py:273-277)
      RL_TRY(dense_resid_ln(st, sid, wp<T>(t.ctx), H, sp<T>(w.ao_w), pp(o.ao_b), x, site(sid * 1000 + (int)l * 10 + 2, cfg.hidden_dropout),
                            pp(o.ao_ln_g), pp(o.ao_ln_b), wp<T>(t.s1), wp<float>(t.rstd1), wp<T>(t.y1)));
      {  // BertIntermediate (modeling_bert.py:326-329)
        // (the pre-activation is what the backward's GELU' reads: a forward nothing differentiates - evaluation, no_grad - does not store it)
        EpiParams<T> ep; ep.mode = EPI_GELU; ep.out = wp<T>(t.post); ep.out2 = bwd_follows() ? wp<T>(t.pre) : nullptr; ep.ldo = I; ep.bias = pp(o.in_b);
        RL_TRY(nt_rows(st, wp<T>(t.y1), H, sp<T>(w.in_w), H, Tk, I, H, ep));
      }
      // BertOutput (modeling_bert.py:339-343)
      RL_TRY(dense_resid_ln(st, sid, wp<T>(t.post), I, sp<T>(w.out_w), pp(o.out_b), wp<T>(t.y1), site(sid * 1000 + (int)l * 10 + 3, cfg.hidden_dropout),
                            pp(o.out_ln_g), pp(o.out_ln_b), wp<T>(t.s2), wp<float>(t.rstd2), wp<T>(t.y2)));
      x = wp<T>(t.y2);
    }
    *out = x;
    return RL_OK;
  }

  // LayerNorm gamma / beta gradients: every ln_bwd leaves one record per workgroup; the records of up to LN_FOLD_MAX sites (the two
  // LayerNorms of four transformer layers) are folded by ONE launch at the end of the layer group instead of one fold per site
  // (38 launches of ~7 us on the critical chain per step).  Per scratch set (= per branch stream).
  LnFoldSites pend[3];
  float* ln_region(int k) const { return wp<float>(pl.sc[cs].ln_slots) + (int64_t)k * LN_SLOT_FLOATS; }
  int flush_ln_folds(hipStream_t st) {
    LnFoldSites& f = pend[cs];
    if (f.n == 0) return RL_OK;
    f.H = H;
    const int rc = ln_fold_multi(st, f);
    f.n = 0;
    return rc;
  }
  int ln_bwd_deferred(hipStream_t st, LnBwdArgs<T>& ln) {
    LnFoldSites& f = pend[cs];
    int nrec = 0;
    ln.slots = ln_region(f.n);
    ln.deferred_records = &nrec;
    RL_TRY(ln_bwd<T>(st, ln));
    if (nrec > 0) {
      f.s[f.n++] = LnFoldSite{ln.slots, nrec, ln.dgamma, ln.dbeta};
      if (f.n == LN_FOLD_MAX) RL_TRY(flush_ln_folds(st));
    }
    return RL_OK;
  }

  // backward of layers [hi .. lo] of a stack; gA holds d(output of layer hi) on entry and d(input of layer lo) on exit
  int layers_backward(hipStream_t st, int sid, const StackOff& so, const std::vector<LayerSh>& shs, StackAct& a, int hi, int lo,
                      T* gA) {
    const int B = pl.B, S = pl.S, Tk = B * S;
    nt_sid = sid;
    const typename Plan::Scratch& sc = pl.sc[cs];
    T* gB = wp<T>(sc.gB); T* gE = wp<T>(sc.gE);
    const bool ov = g_wgrad_overlap && (!branch_mode || cs == 0) && side_ok();      // under branch overlap only the bert branch owns two dY sets
    for (int l = hi; l >= lo; --l) {
      const LayerOff& o = so.layers[l];
      const LayerSh& w = shs[l];
      LayerAct& t = a.layers[l];
      const T* x_in = l > 0 ? wp<T>(a.layers[l - 1].y2) : wp<T>(a.emb_y);
      const DropParams d3 = site(sid * 1000 + l * 10 + 3, cfg.hidden_dropout);
      const DropParams d2 = site(sid * 1000 + l * 10 + 2, cfg.hidden_dropout);
      const DropParams d1 = site(sid * 1000 + l * 10 + 1, cfg.attn_dropout);
      // Overlap: this layer's dY matrices go to the buffer set of its parity and stay untouched until the side stream has
      // run the four weight-gradient GEMMs on them; the caller's stream only waits before it re-uses that set (two layers
      // later).  Without overlap everything stays on `st` and set 0 is the plain scratch.
      const int p = ov ? (bw_layers++ & 1) : 0;
      T* gC1 = wp<T>(sc.wC1[p]); T* gC2 = wp<T>(sc.wC2[p]); T* gD = wp<T>(sc.wD[p]); T* gF = wp<T>(sc.wF[p]);
      if (ov && done_pending[p]) {
        if (hipStreamWaitEvent(st, ev_done[p], 0) != hipSuccess) return RL_ERR_LAUNCH;
        done_pending[p] = false;
      }
      const bool grouped = g_wgrad_group != 0;
      TnGroupProblem<T> gp4[4];
      auto wgrad = [&](int k, const T* dy, int64_t ldy, const T* x, int64_t ldx, int Iw, int Jw, float* bias_g, float* w_g) -> int {
        if (grouped) {        // collected; launched once after the attention backward produced the last dY
          gp4[k].A = dy; gp4[k].lda = ldy; gp4[k].B = x; gp4[k].ldb = ldx; gp4[k].I = Iw; gp4[k].J = Jw; gp4[k].out = w_g; gp4[k].ldo = Jw;
          gp4[k].colsum = bias_g;
          return RL_OK;
        }
        hipStream_t ws_ = st;
        TnEpi te; te.slab = wp<float>(sc.tn_slab); te.slab_elems = TN_SLAB_ELEMS; te.colsum = bias_g; te.out = w_g; te.ldo = Jw;
        if (ov) {
          if (hipEventRecord(ev_ready[p][k], st) != hipSuccess || hipStreamWaitEvent(side, ev_ready[p][k], 0) != hipSuccess) return RL_ERR_LAUNCH;
          ws_ = side; te.slab = wp<float>(sc.tn_slab2);
        }
        return gemm_tn<T>(ws_, dy, ldy, x, ldx, Tk, Iw, Jw, te);
      };
      const bool keep_dy = ov || grouped;      // the dense-output gradients must outlive the in-place updates of gA / gB
      {  // output LayerNorm: gA = d y2 -> gB = d s2 (residual part of d y1), gC1 = d(dense out) = d s2 * dropmask
        LnBwdArgs<T> ln; ln.rows = Tk; ln.H = H; ln.dy = gA; ln.xhat = wp<T>(t.s2); ln.rstd = wp<float>(t.rstd2); ln.row_live = live_rows();
        ln.gamma = pp(o.out_ln_g); ln.dx = gB; ln.dx_drop = (d3.thresh || keep_dy) ? gC1 : nullptr; ln.out_drop = d3;
        ln.dgamma = gp(o.out_ln_g); ln.dbeta = gp(o.out_ln_b);
        RL_TRY(ln_bwd_deferred(st, ln));
      }
      const T* dso = (d3.thresh || keep_dy) ? gC1 : gB;
      RL_TRY(wgrad(0, dso, H, wp<T>(t.post), I, H, I, gp(o.out_b), gp(o.out_w)));
      {  // d pre = (d s2' . W_out) * gelu'(pre)
        EpiParams<T> ep; ep.mode = EPI_GELU_BWD; ep.out = gD; ep.ldo = I; ep.aux = wp<T>(t.pre); ep.ldaux = I;
        RL_TRY(nt_rows(st, dso, H, sp<T>(w.out_wT), H, Tk, I, H, ep, true));      // gD feeds wgrad(1): whole live blocks
      }
      RL_TRY(wgrad(1, gD, I, wp<T>(t.y1), H, I, H, gp(o.in_b), gp(o.in_w)));
      {  // d y1 = d s2 + d pre . W_in
        EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = gB; ep.ldo = H; ep.accumulate = 1;
        RL_TRY(nt_rows(st, gD, I, sp<T>(w.in_wT), I, Tk, H, I, ep));
      }
      {  // attention-output LayerNorm: gB = d y1 -> gA = d s1, gC2 = d(dense out)
        LnBwdArgs<T> ln; ln.rows = Tk; ln.H = H; ln.dy = gB; ln.xhat = wp<T>(t.s1); ln.rstd = wp<float>(t.rstd1); ln.row_live = live_rows();
        ln.gamma = pp(o.ao_ln_g); ln.dx = gA; ln.dx_drop = (d2.thresh || keep_dy) ? gC2 : nullptr; ln.out_drop = d2;
        ln.dgamma = gp(o.ao_ln_g); ln.dbeta = gp(o.ao_ln_b);
        RL_TRY(ln_bwd_deferred(st, ln));
      }
      const T* dsa = (d2.thresh || keep_dy) ? gC2 : gA;
      RL_TRY(wgrad(2, dsa, H, wp<T>(t.ctx), H, H, H, gp(o.ao_b), gp(o.ao_w)));
      {  // d ctx = d s1' . W_ao
        EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = gE; ep.ldo = H;
        RL_TRY(nt_rows(st, dsa, H, sp<T>(w.ao_wT), H, Tk, H, H, ep));
      }
      {
        const T* q = wp<T>(t.qkv);
        RL_TRY(attn_bwd<T>(st, q, q + H, q + 2 * H, 3 * H, wp<float>(pl.mask_add), wp<T>(t.ctx), gE, H, wp<float>(t.lse),
                           wp<float>(sc.rowdot), gF, gF + H, gF + 2 * H, 3 * H, B, nh, S, d1.seed, d1.thresh, d1.scale, dead_ok ? wp<int>(pl.live_rlen) : nullptr));      // (backward: padding rows' gradients are exact zeros either way)
      }
      RL_TRY(wgrad(3, gF, 3 * H, x_in, H, 3 * H, H, gp(o.qkv_b), gp(o.qkv_w)));
      if (grouped) {
        hipStream_t ws_ = st;
        if (ov) {
          if (hipEventRecord(ev_ready[p][3], st) != hipSuccess || hipStreamWaitEvent(side, ev_ready[p][3], 0) != hipSuccess) return RL_ERR_LAUNCH;
          ws_ = side;
        }
        RL_TRY(gemm_tn_group<T>(ws_, 4, gp4, Tk, 1.0f, pass_overwrite ? 1 : 0, live_tiles(), live_tile_count(), live_list_rows()));
      }
      {  // d x_in = d s1 + d qkv . W_qkv
        EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = gA; ep.ldo = H; ep.accumulate = 1;
        RL_TRY(nt_rows(st, gF, 3 * H, sp<T>(w.qkv_wT), 3 * H, Tk, H, 3 * H, ep));
      }
      if (ov) {
        if (hipEventRecord(ev_done[p], side) != hipSuccess) return RL_ERR_LAUNCH;
        done_pending[p] = true;
      }
    }
    return flush_ln_folds(st);
  }

  // embeddings backward: gA = d(embedding output) -> dx (T*, d of the pre-LayerNorm sum) ; scatters table grads
  int emb_backward(hipStream_t st, int sid, const StackOff& so, StackAct& a, const int64_t* ids, int pos_zero, const T* gA, T* dx) {
    const int B = pl.B, S = pl.S, Tk = B * S;
    LnBwdArgs<T> ln; ln.slots = ln_region(LN_FOLD_MAX); ln.rows = Tk; ln.H = H; ln.dy = gA; ln.in_drop = site(sid * 1000 + 900, cfg.hidden_dropout);
    ln.xhat = wp<T>(a.emb_xhat); ln.rstd = wp<float>(a.emb_rstd); ln.gamma = pp(so.ln_g); ln.dx = dx;
    ln.dgamma = gp(so.ln_g); ln.dbeta = gp(so.ln_b);
    ln.row_live = live_rows();      // (padding rows: dy is an exact zero - dx written as zeros without reading the row's saved xhat)
    RL_TRY(ln_bwd<T>(st, ln));
    RL_TRY(embed_bwd<T>(st, dx, ids, B, S, H, (ids && so.word >= 0) ? gp(so.word) : nullptr, gp(so.pos), pos_zero, gp(so.type)));
    return RL_OK;
  }

  // ---------------------------------------------------------------- glyph ResNet (char_cnn.py)
  // rows_dev: device-side bound on the row space (glyph dedup: #distinct ids x pixels per image of that map)
  ConvLoader<T> geom(const T* src, const int64_t* index, int rows, int Hr, int Hs, int C, int ksz, int stride, int pad, int mode,
                     const int* rows_dev) const {
    ConvLoader<T> g;
    g.src = src; g.img_index = index; g.rows = rows; g.rows_dev = rows_dev; g.Hr = Hr; g.Wr = Hr; g.Hs = Hs; g.Ws = Hs; g.C = C;
    g.KH = ksz; g.KW = ksz; g.stride = stride; g.pad = pad; g.mode = mode; g.K = ksz * ksz * C;
    if (index != nullptr) g.index_rows = V;          // (gathers from the whole glyph table: the 32-bit offset span check uses the table size)
    return g;
  }
  // glyph dedup bookkeeping of block k's output map: device row bound, multiplicities, pixels per image
  RowBound rbound(int k) const {
    RowBound rb;
    rb.rows_dev = wp<int>(pl.gu_bounds) + 1 + k;
    rb.counts = wp<float>(pl.gu_counts);
    rb.hw = pl.blk[k].Hout * pl.blk[k].Hout;
    rb.slots = wp<float>(pl.bn_slots);
    return rb;
  }
  // Pn = B*S*Hout^2 is the TRUE sample count of the statistics (every token counts, PADs included, as in the
  // reference); the kernels visit only the distinct glyphs and weight them by their multiplicity.
  int bn_forward(hipStream_t st, const T* x, int Pn, int C, const BnOff& o, const BnAct& a, const RowBound& rb) {
    if (last.training) {
      float* sums = wp<float>(pl.bn_sums);
      if constexpr (sizeof(T) == 2) {
        if (bn_stats_train16(st, x, Pn, C, Pn, pp(o.g), pp(o.b), 1e-5f, 0.1f, BF + o.rmean, BF + o.rvar, wp<float>(a.mean), wp<float>(a.rstd),
                             wp<float>(a.scale), wp<float>(a.shift), BI + o.nbt, rb) == RL_OK)
          return RL_OK;
      }
      RL_TRY(col_sum<T>(st, x, Pn, C, wp<float>(a.mean), rb, 1.0f / (float)Pn));          // the fold writes the mean directly
      RL_TRY(col_sumsq_centered<T>(st, x, Pn, C, wp<float>(a.mean), sums + C, rb));
      RL_TRY(bn_finalize_train(st, wp<float>(a.mean), sums + C, C, Pn, pp(o.g), pp(o.b), 1e-5f, 0.1f, BF + o.rmean, BF + o.rvar,
                               wp<float>(a.rstd), wp<float>(a.scale), wp<float>(a.shift), BI + o.nbt));
    } else {
      RL_TRY(bn_finalize_eval(st, C, pp(o.g), pp(o.b), 1e-5f, BF + o.rmean, BF + o.rvar, wp<float>(a.scale), wp<float>(a.shift)));
    }
    return RL_OK;
  }
  bool glyph_gathered = false;             // gu_dense holds the distinct glyph images of the last resnet_forward
  int gather_glyphs(hipStream_t st) {
    return gather_images<T>(st, sp<T>(sh_glyph), wp<int64_t>(pl.gu_ids), wp<int>(pl.gu_bounds), pl.B * pl.S,
                            (int64_t)cfg.glyph_size * cfg.glyph_size * 8, wp<T>(pl.gu_dense));
  }
  int resnet_forward(hipStream_t st, const int64_t* ids, const T** out) {
    // The glyph stack of a token depends only on its id, so the ResNet runs once per DISTINCT id of the batch
    // (order of first occurrence) and BatchNorm weights each glyph by its multiplicity: identical statistics
    // to the reference's dense [B*S, F, 32, 32] pass, a fraction of the work (PAD alone is ~1/3 of the rows).
    {
      HwList hw; hw.n = 5;
      for (int k = 0; k < 5; ++k) hw.v[k] = pl.blk[k].Hout * pl.blk[k].Hout;
      RL_TRY(glyph_unique(st, ids, pl.B * pl.S, V, wp<int>(pl.gu_first), wp<int>(pl.gu_flag), wp<int64_t>(pl.gu_ids),
                          wp<float>(pl.gu_counts), wp<int>(pl.gu_inv), wp<int>(pl.gu_bounds), hw));
    }
    const bool fuse7 = g_glyph_fuse != 0 && (int64_t)V * cfg.glyph_size * cfg.glyph_size * 8 * (int64_t)sizeof(T) < 0xFFFFFE00ll;
    glyph_gathered = !fuse7;
    if (!fuse7) RL_TRY(gather_glyphs(st));
    const T* x = fuse7 ? sp<T>(sh_glyph) : wp<T>(pl.gu_dense);
    const int64_t* index = fuse7 ? wp<int64_t>(pl.gu_ids) : nullptr;
    for (int k = 0; k < 5; ++k) {
      const BlockOff& o = L.blocks[k];
      const BlockSh& s = sh_blk[k];
      BlockAct& a = pl.blk[k];
      const int Co = o.cout, Cin = s.cin_pad, Pn = a.Pout;
      const RowBound rb = rbound(k);
      if (!last.training && g_bn_fold) {
        // K9, evaluation (round 6; char_cnn.py:15-32 with BatchNorm2d on its running statistics = a per-channel affine map): the three
        // convolutions apply their BatchNorm in the epilogue - shortcut first (its normalised output is the second convolution's `aux`),
        // ReLU where the reference has one.  Three launches + three 1-workgroup scale / shift kernels per block, no pass over a raw
        // convolution output (was: 3 convolutions + 3 finalize + 2 apply launches, 5 extra passes over [Pn, Co]).
        RL_TRY(bn_finalize_eval(st, Co, pp(o.bns.g), pp(o.bns.b), 1e-5f, BF + o.bns.rmean, BF + o.bns.rvar, wp<float>(a.bns.scale), wp<float>(a.bns.shift)));
        RL_TRY(bn_finalize_eval(st, Co, pp(o.bn1.g), pp(o.bn1.b), 1e-5f, BF + o.bn1.rmean, BF + o.bn1.rvar, wp<float>(a.bn1.scale), wp<float>(a.bn1.shift)));
        RL_TRY(bn_finalize_eval(st, Co, pp(o.bn2.g), pp(o.bn2.b), 1e-5f, BF + o.bn2.rmean, BF + o.bn2.rvar, wp<float>(a.bn2.scale), wp<float>(a.bn2.shift)));
        EpiParams<T> ea; ea.mode = EPI_AFFINE; ea.ldo = Co;
        ea.out = wp<T>(a.cs); ea.col_scale = wp<float>(a.bns.scale); ea.bias = wp<float>(a.bns.shift); ea.relu = 0;
        RL_TRY(gemm_nt_conv<T>(st, geom(x, index, Pn, a.Hout, a.Hin, Cin, 1, 2, 0, 0, rb.rows_dev), sp<T>(s.wsf), Cin, Pn, Co, Cin, ea));
        ea.out = wp<T>(a.h1); ea.col_scale = wp<float>(a.bn1.scale); ea.bias = wp<float>(a.bn1.shift); ea.relu = 1;
        RL_TRY(gemm_nt_conv<T>(st, geom(x, index, Pn, a.Hout, a.Hin, Cin, 3, 2, 1, 0, rb.rows_dev), sp<T>(s.w1f), 9 * Cin, Pn, Co, 9 * Cin, ea));
        ea.out = wp<T>(a.out); ea.col_scale = wp<float>(a.bn2.scale); ea.bias = wp<float>(a.bn2.shift); ea.relu = 1;
        ea.aux = wp<T>(a.cs); ea.ldaux = Co;
        if (a.Hout == 1) RL_TRY(gemm_nt<T>(st, wp<T>(a.h1), Co, sp<T>(s.w2f) + 4 * Co, 9 * Co, Pn, Co, Co, ea, rb.rows_dev));
        else RL_TRY(gemm_nt_conv<T>(st, geom(wp<T>(a.h1), nullptr, Pn, a.Hout, a.Hout, Co, 3, 1, 1, 0, rb.rows_dev), sp<T>(s.w2f), 9 * Co, Pn, Co, 9 * Co, ea));
        x = wp<T>(a.out);
        index = nullptr;
        continue;
      }
      EpiParams<T> ep; ep.mode = EPI_STORE; ep.ldo = Co;
      // residual_function.0: 3x3 stride 2 pad 1 (char_cnn.py:16)
      ep.out = wp<T>(a.c1);
      RL_TRY(gemm_nt_conv<T>(st, geom(x, index, Pn, a.Hout, a.Hin, Cin, 3, 2, 1, 0, rb.rows_dev), sp<T>(s.w1f), 9 * Cin, Pn, Co, 9 * Cin, ep));
      RL_TRY(bn_forward(st, wp<T>(a.c1), Pn, Co, o.bn1, a.bn1, rb));
      RL_TRY(bn_apply<T>(st, wp<T>(a.c1), wp<float>(a.bn1.scale), wp<float>(a.bn1.shift), nullptr, nullptr, nullptr, wp<T>(a.h1), Pn, Co, 1, rb));
      // residual_function.3: 3x3 stride 1 pad 1 (char_cnn.py:19)
      ep.out = wp<T>(a.c2);
      if (a.Hout == 1) {
        // a 3x3 / pad 1 convolution on a 1x1 map touches only its centre tap (the other eight read padding): a dense GEMM against
        // the centre-tap slice of the [Co][tap][Ci] operand copy, 1/9 of the implicit-GEMM work (block 5: 113 -> ~15 us)
        RL_TRY(gemm_nt<T>(st, wp<T>(a.h1), Co, sp<T>(s.w2f) + 4 * Co, 9 * Co, Pn, Co, Co, ep, rb.rows_dev));
      } else {
        RL_TRY(gemm_nt_conv<T>(st, geom(wp<T>(a.h1), nullptr, Pn, a.Hout, a.Hout, Co, 3, 1, 1, 0, rb.rows_dev), sp<T>(s.w2f), 9 * Co, Pn, Co, 9 * Co, ep));
      }
      RL_TRY(bn_forward(st, wp<T>(a.c2), Pn, Co, o.bn2, a.bn2, rb));
      // shortcut: 1x1 stride 2 (char_cnn.py:26-28)
      ep.out = wp<T>(a.cs);
      RL_TRY(gemm_nt_conv<T>(st, geom(x, index, Pn, a.Hout, a.Hin, Cin, 1, 2, 0, 0, rb.rows_dev), sp<T>(s.wsf), Cin, Pn, Co, Cin, ep));
      RL_TRY(bn_forward(st, wp<T>(a.cs), Pn, Co, o.bns, a.bns, rb));
      RL_TRY(bn_apply<T>(st, wp<T>(a.c2), wp<float>(a.bn2.scale), wp<float>(a.bn2.shift), wp<T>(a.cs), wp<float>(a.bns.scale),
                         wp<float>(a.bns.shift), wp<T>(a.out), Pn, Co, 1, rb));
      x = wp<T>(a.out);
      index = nullptr;
    }
    *out = x;
    return RL_OK;
  }
  // d_top: gradient w.r.t. block-5 output per DISTINCT glyph [U,768] (already summed over the tokens sharing it)
  int resnet_backward(hipStream_t st, const T* d_top) {
    float* sums = wp<float>(pl.bn_sums);
    const T* d_out = d_top;
    if (!glyph_gathered) { RL_TRY(gather_glyphs(st)); glyph_gathered = true; }      // K7: block 1's weight gradients read the gathered images
    for (int k = 4; k >= 0; --k) {
      const BlockOff& o = L.blocks[k];
      const BlockSh& s = sh_blk[k];
      BlockAct& a = pl.blk[k];
      const int Co = o.cout, Cin = s.cin_pad, Pn = a.Pout;
      const RowBound rb = rbound(k);
      const T* x_in = k > 0 ? wp<T>(pl.blk[k - 1].out) : wp<T>(pl.gu_dense);
      const int64_t* index = nullptr;
      T* dc2 = wp<T>(pl.r_dc2); T* dcs = wp<T>(pl.r_dcs); T* dh1 = wp<T>(pl.r_dh1); T* dc1 = wp<T>(pl.r_dc1);
      // out = relu(bn2(c2) + bns(cs))
      // bn2 and the shortcut's BN see the same incoming gradient and ReLU mask: bf16 reads them once for both (two passes over four
      // tensors instead of four passes over three)
      bool paired = false;
      if constexpr (sizeof(T) == 2) {
        if (4 * Co <= 2048 &&
            bn_bwd_reduce2(st, d_out, wp<T>(a.out), wp<T>(a.c2), wp<float>(a.bn2.mean), wp<float>(a.bn2.rstd), wp<T>(a.cs), wp<float>(a.bns.mean),
                           wp<float>(a.bns.rstd), Pn, Co, sums, rb) == RL_OK) {
          RL_TRY(bn_bwd_apply2(st, d_out, wp<T>(a.out), wp<T>(a.c2), wp<float>(a.bn2.mean), wp<float>(a.bn2.rstd), pp(o.bn2.g), dc2, gp(o.bn2.g),
                               gp(o.bn2.b), wp<T>(a.cs), wp<float>(a.bns.mean), wp<float>(a.bns.rstd), pp(o.bns.g), dcs, gp(o.bns.g), gp(o.bns.b),
                               sums, Pn, Co, rb, Pn));
          paired = true;
        }
      }
      if (!paired) {
        RL_TRY(bn_bwd_reduce<T>(st, d_out, wp<T>(a.out), wp<T>(a.c2), wp<float>(a.bn2.mean), wp<float>(a.bn2.rstd), Pn, Co, sums, rb));
        RL_TRY(bn_bwd_apply<T>(st, d_out, wp<T>(a.out), wp<T>(a.c2), wp<float>(a.bn2.mean), wp<float>(a.bn2.rstd), pp(o.bn2.g), sums,
                               Pn, Co, dc2, gp(o.bn2.g), gp(o.bn2.b), rb, Pn));
        RL_TRY(bn_bwd_reduce<T>(st, d_out, wp<T>(a.out), wp<T>(a.cs), wp<float>(a.bns.mean), wp<float>(a.bns.rstd), Pn, Co, sums, rb));
        RL_TRY(bn_bwd_apply<T>(st, d_out, wp<T>(a.out), wp<T>(a.cs), wp<float>(a.bns.mean), wp<float>(a.bns.rstd), pp(o.bns.g), sums,
                               Pn, Co, dcs, gp(o.bns.g), gp(o.bns.b), rb, Pn));
      }
      // conv2 (3x3 s1): weight grad and data grad
      { TnEpi te; te.slab = wp<float>(pl.sc[cs].tn_slab); te.slab_elems = TN_SLAB_ELEMS; te.mode = TN_CONVW; te.out = gp(o.w2); te.Cin = Co; te.Cpad = Co; te.KHW = 9;
        if (a.Hout == 1) { te.tap0 = 4; RL_TRY(gemm_tn<T>(st, dc2, Co, wp<T>(a.h1), Co, Pn, Co, Co, te, rb.rows_dev)); }       // centre tap only
        else RL_TRY(gemm_tn_conv<T>(st, dc2, Co, geom(wp<T>(a.h1), nullptr, Pn, a.Hout, a.Hout, Co, 3, 1, 1, 0, rb.rows_dev), Pn, Co, 9 * Co, te)); }
      { EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = dh1; ep.ldo = Co;
        if (a.Hout == 1) RL_TRY(gemm_nt<T>(st, dc2, Co, sp<T>(s.w2d) + 4 * Co, 9 * Co, Pn, Co, Co, ep, rb.rows_dev));          // centre tap only
        else RL_TRY(gemm_nt_conv<T>(st, geom(dc2, nullptr, Pn, a.Hout, a.Hout, Co, 3, 1, 1, 1, rb.rows_dev), sp<T>(s.w2d), 9 * Co, Pn, Co, 9 * Co, ep)); }
      // h1 = relu(bn1(c1))
      RL_TRY(bn_bwd_reduce<T>(st, dh1, wp<T>(a.h1), wp<T>(a.c1), wp<float>(a.bn1.mean), wp<float>(a.bn1.rstd), Pn, Co, sums, rb));
      RL_TRY(bn_bwd_apply<T>(st, dh1, wp<T>(a.h1), wp<T>(a.c1), wp<float>(a.bn1.mean), wp<float>(a.bn1.rstd), pp(o.bn1.g), sums,
                             Pn, Co, dc1, gp(o.bn1.g), gp(o.bn1.b), rb, Pn));
      // conv1 (3x3 s2) and shortcut (1x1 s2) weight grads
      { TnEpi te; te.slab = wp<float>(pl.sc[cs].tn_slab); te.slab_elems = TN_SLAB_ELEMS; te.mode = TN_CONVW; te.out = gp(o.w1); te.Cin = o.cin; te.Cpad = Cin; te.KHW = 9;
        RL_TRY(gemm_tn_conv<T>(st, dc1, Co, geom(x_in, index, Pn, a.Hout, a.Hin, Cin, 3, 2, 1, 0, rb.rows_dev), Pn, Co, 9 * Cin, te)); }
      { TnEpi te; te.slab = wp<float>(pl.sc[cs].tn_slab); te.slab_elems = TN_SLAB_ELEMS; te.mode = TN_CONVW; te.out = gp(o.ws); te.Cin = o.cin; te.Cpad = Cin; te.KHW = 1;
        RL_TRY(gemm_tn_conv<T>(st, dcs, Co, geom(x_in, index, Pn, a.Hout, a.Hin, Cin, 1, 2, 0, 0, rb.rows_dev), Pn, Co, Cin, te)); }
      if (k > 0) {   // d x_in = dgrad(conv1) + dgrad(shortcut); the glyph table itself is frozen
        const int Pin = pl.blk[k - 1].Pout;
        const int* in_bound = rbound(k - 1).rows_dev;
        T* dx = (d_out == wp<T>(pl.r_dx)) ? wp<T>(pl.r_dout) : wp<T>(pl.r_dx);
        EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = dx; ep.ldo = Cin;
        int hsh = 0; while ((1 << hsh) < a.Hin) ++hsh;
        if (g_dgrad_parity && (1 << hsh) == a.Hin && hsh >= 1) {
          // by parity class of the input pixel: only the taps that reach it, a quarter of the rows per launch; the 1x1 stride-2
          // shortcut reaches the (even, even) pixels only
          ep.rm_hw_shift = 2 * hsh; ep.rm_w_shift = hsh;
          for (int c = 0; c < 4; ++c) {
            int first = 0;
            int nt = conv_s2_class(3, 3, 1, c, &first, nullptr);
            ConvLoader<T> g = geom(dc1, nullptr, Pin / 4, a.Hin, a.Hout, Co, 3, 2, 1, 1, in_bound);
            g.par = c; ep.rm_par = c; ep.accumulate = 0;
            if (a.Hout == 1) {      // 2x2 map under a 1x1 output: pixel (py, px) is reached by tap (py + 1, px + 1) alone
              g.tap_sel = ((c >> 1) + 1) * 3 + (c & 1) + 1;
              first = conv_s2_slot(3, 3, 1, (c >> 1) + 1, (c & 1) + 1); nt = 1;
            }
            RL_TRY(gemm_nt_conv<T>(st, g, sp<T>(s.w1p) + (int64_t)first * Co, 9 * Co, Pin / 4, Cin, nt * Co, ep));
          }
          ConvLoader<T> g = geom(dcs, nullptr, Pin / 4, a.Hin, a.Hout, Co, 1, 2, 0, 1, in_bound);
          g.par = 0; ep.rm_par = 0; ep.accumulate = 1;
          RL_TRY(gemm_nt_conv<T>(st, g, sp<T>(s.wsd), Co, Pin / 4, Cin, Co, ep));
        } else {
          RL_TRY(gemm_nt_conv<T>(st, geom(dc1, nullptr, Pin, a.Hin, a.Hout, Co, 3, 2, 1, 1, in_bound), sp<T>(s.w1d), 9 * Co, Pin, Cin, 9 * Co, ep));
          ep.accumulate = 1;
          RL_TRY(gemm_nt_conv<T>(st, geom(dcs, nullptr, Pin, a.Hin, a.Hout, Co, 1, 2, 0, 1, in_bound), sp<T>(s.wsd), Co, Pin, Cin, Co, ep));
        }
        d_out = dx;
      }
    }
    return RL_OK;
  }

  // ---------------------------------------------------------------- pinyin GRU (models.py:818-826)
  int gru_forward(hipStream_t st) {
    const int N = pl.B * pl.S, Tp = last.Tp;
    // W_ih x + b_ih for the 33 pinyin symbols, from the fp32 masters.  Parity mode: one exact-fp32 MFMA GEMM ([33,768] x [2304,768]^T,
    // a k-ordered fmaf chain); speed mode: the dedicated table kernel (W_ih row in registers, 56 -> ~10 us)
    bool table_done = false;
    if constexpr (sizeof(T) == 2) table_done = gru_table_fwd(st, pp(L.pho_emb), pp(L.gru_w_ih), pp(L.gru_b_ih), cfg.pho_vocab, H, wp<float>(pl.gru_table)) == RL_OK;
    if (!table_done) {
      EpiParams<float> ep; ep.mode = EPI_STORE; ep.out = wp<float>(pl.gru_table); ep.ldo = 3 * H; ep.bias = pp(L.gru_b_ih);
      RL_TRY(gemm_nt<float>(st, pp(L.pho_emb), H, pp(L.gru_w_ih), H, cfg.pho_vocab, 3 * H, H, ep));
    }
    for (int t = 0; t < Tp; ++t) {
      const int n = last_alive[t];
      if (n <= 0) break;
      const int* nd = alive_dev ? (const int*)alive_dev + t : nullptr;
      T* hs_t = wp<T>(pl.gru_hs) + (int64_t)t * N * H;
      T* gh_t = wp<T>(pl.gru_gh) + (int64_t)t * N * 3 * H;
      const T* hs_prev = t > 0 ? wp<T>(pl.gru_hs) + (int64_t)(t - 1) * N * H : nullptr;
      if constexpr (sizeof(T) == 2) {
        if (t > 0 && g_gru_fuse && (H % 64) == 0) {      // recurrent projection + gate math in one launch
          EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = hs_t; ep.ldo = H; ep.bias = pp(L.gru_b_hh); ep.m_dev = nd;
          ep.gru_table = wp<float>(pl.gru_table); ep.gru_pho_idx = last.pho_idx; ep.gru_perm = last.pho_perm; ep.gru_lens = last.pho_lens_sorted;
          ep.gru_hprev = hs_prev; ep.gru_rzn = wp<T>(pl.gru_rzn) + (int64_t)t * N * 3 * H; ep.gru_gh = gh_t; ep.gru_out = wp<T>(pl.gru_out);
          ep.gru_Tp = Tp; ep.gru_t = t;
          const int rc = gemm_nt8_gru(st, hs_prev, H, sp<T>(sh_gru_hh), H, n, 3 * H, H, ep);
          if (rc == RL_OK) continue;
          if (rc != RL_ERR_ARG) return rc;
        }
      }
      if (t > 0) {
        EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = gh_t; ep.ldo = 3 * H; ep.bias = pp(L.gru_b_hh);
        RL_TRY(gemm_nt<T>(st, hs_prev, H, sp<T>(sh_gru_hh), H, n, 3 * H, H, ep, nd));
      }
      GruStepArgs<T> a;
      a.n_alive = n; a.n_alive_dev = nd; a.H = H; a.Tp = Tp; a.t = t; a.table = wp<float>(pl.gru_table); a.pho_idx = last.pho_idx;
      a.perm = last.pho_perm; a.lens = last.pho_lens_sorted; a.gh = t > 0 ? gh_t : nullptr; a.b_hh = pp(L.gru_b_hh);
      a.h_prev = hs_prev; a.h_new = hs_t; a.rzn = wp<T>(pl.gru_rzn) + (int64_t)t * N * 3 * H; a.out = wp<T>(pl.gru_out);
      RL_TRY(gru_step_fwd<T>(st, a));
    }
    return RL_OK;
  }
  int gru_backward(hipStream_t st, const T* dout) {
    const int N = pl.B * pl.S, Tp = last.Tp;
    float* dtable = wp<float>(pl.gru_dtable);
    RL_TRY(fill_f32(st, dtable, 0.f, 64LL * 3 * H));
    for (int t = Tp - 1; t >= 0; --t) {
      const int n = last_alive[t];
      if (n <= 0) continue;
      const int* nd = alive_dev ? (const int*)alive_dev + t : nullptr;
      const T* hs_prev = t > 0 ? wp<T>(pl.gru_hs) + (int64_t)(t - 1) * N * H : nullptr;
      T* gh_t = wp<T>(pl.gru_gh) + (int64_t)t * N * 3 * H;
      GruStepArgs<T> a;
      a.n_alive = n; a.n_alive_dev = nd; a.H = H; a.Tp = Tp; a.t = t; a.pho_idx = last.pho_idx; a.perm = last.pho_perm; a.lens = last.pho_lens_sorted;
      a.gh = t > 0 ? gh_t : nullptr; a.b_hh = pp(L.gru_b_hh); a.h_prev = hs_prev;
      a.rzn = wp<T>(pl.gru_rzn) + (int64_t)t * N * 3 * H; a.dout = dout; a.dh = wp<T>(pl.gru_dh); a.dgi = wp<T>(pl.gru_dgi);
      a.dgh = wp<T>(pl.gru_dgh); a.onehot = wp<T>(pl.gru_onehot);
      RL_TRY(gru_step_bwd<T>(st, a));
      { TnEpi te; te.slab = wp<float>(pl.sc[cs].tn_slab); te.slab_elems = TN_SLAB_ELEMS; te.out = dtable; te.ldo = 3 * H; RL_TRY(gemm_tn<T>(st, a.onehot, 64, a.dgi, 3 * H, n, 64, 3 * H, te, nd)); }
      // d b_hh = column sums of dgh: rides along the W_hh weight-gradient GEMM of the step (fused ones-vector MFMA, TnEpi::colsum) -
      // six column-reduction + fold launch pairs less; step 0 has no recurrent GEMM and keeps the reduction
      if (t == 0) RL_TRY(bias_grad<T>(st, a.dgh, 3 * H, n, 3 * H, gp(L.gru_b_hh), nd));
      if (t > 0) {
        { TnEpi te; te.slab = wp<float>(pl.sc[cs].tn_slab); te.slab_elems = TN_SLAB_ELEMS; te.out = gp(L.gru_w_hh); te.ldo = H; te.colsum = gp(L.gru_b_hh);
          RL_TRY(gemm_tn<T>(st, a.dgh, 3 * H, hs_prev, H, n, 3 * H, H, te, nd)); }
        EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = a.dh; ep.ldo = H; ep.accumulate = 1;
        RL_TRY(gemm_nt<T>(st, a.dgh, 3 * H, sp<T>(sh_gru_hhT), 3 * H, n, H, 3 * H, ep, nd));
      }
    }
    RL_TRY(gru_table_bwd(st, dtable, 3 * H, pp(L.pho_emb), pp(L.gru_w_ih), cfg.pho_vocab, H, gp(L.pho_emb), gp(L.gru_w_ih), gp(L.gru_b_ih)));
    return RL_OK;
  }

  GateArgs<T> gate_args() const {
    GateArgs<T> g;
    g.B = pl.B; g.S = pl.S; g.H = H;
    g.bert = wp<T>(pl.bert.layers.back().y2); g.pho = wp<T>(pl.pho.layers.back().y2); g.res = wp<T>(pl.res_h);
    g.masks = last.masks; g.W = pp(L.gate_w); g.bias = pp(L.gate_b);
    g.mean = wp<float>(pl.gate_mean); g.msum = wp<float>(pl.gate_msum); g.g = wp<float>(pl.gate_g); g.fused = wp<T>(pl.fused);
    return g;
  }

  // ---------------------------------------------------------------- forward
  // a freshly planned workspace: the self-cleaning LayerNorm-backward accumulators start at zero
  int ln_epoch[3] = {0, 0, 0};
  // dense -> dropout -> + residual -> LayerNorm (modeling_bert.py:273-277, 339-343): one launch when the fused form applies
  // the forward being enqueued is one a backward can follow (realise_engine_backward checks the same): it must leave what that pass reads
  bool bwd_follows() const { return last.training && last.tgt_idx != nullptr && last.want_dlogits; }
  int dense_resid_ln(hipStream_t st, int sid, const T* a, int K, const T* w, const float* bias, const T* resid, const DropParams& drop,
                     const float* gamma, const float* beta, T* s_xhat, float* rstd, T* y) {
    const int Tk = pl.B * pl.S;
    EpiParams<T> ep; ep.mode = EPI_DROP_RESID; ep.out = s_xhat; ep.ldo = H; ep.bias = bias; ep.aux = resid; ep.ldaux = H; set_drop(ep, drop);
    if constexpr (sizeof(T) == 2) {
      if (g_ln_fuse && !rows_live && sid >= 0 && sid < 3 && pl.ln_part[sid] != 0 && (Tk % 128) == 0 && (H % 192) == 0) {
        EpiParams<T> e2 = ep;
        e2.ln_gamma = gamma; e2.ln_beta = beta; e2.ln_eps = cfg.ln_eps; e2.ln_y = y; e2.ln_rstd = rstd;
        e2.ln_part = wp<float>(pl.ln_part[sid]);
        if (g_ln_fuse == 2) e2.ln_flag = (int*)e2.ln_part;       // diagnostics: no cross-tile hand-off (see nt8_ln_epilogue)
        e2.ln_target = (ln_epoch[sid] % 0x7FFFFFF0) + 1;          // launch tag: never 0 (the zero-filled buffer), != the previous launch's
        e2.ln_timeout = id_flag != nullptr ? id_flag + 1 : nullptr;
        const int rc = gemm_nt8_ln(st, a, K, w, K, Tk, H, K, e2);
        if (rc == RL_OK) { ++ln_epoch[sid]; return RL_OK; }
        if (rc != RL_ERR_ARG) return rc;
      }
    }
    RL_TRY(nt_rows(st, a, K, w, K, Tk, H, K, ep));
    LnFwdArgs<T> ln; ln.rows = Tk; ln.H = H; ln.x = s_xhat; ln.gamma = gamma; ln.beta = beta;
    ln.eps = cfg.ln_eps; ln.y = y; ln.xhat = bwd_follows() ? s_xhat : nullptr; ln.rstd = rstd;      // (a forward nothing differentiates does not store the normalised rows)
    if (rows_live && (Tk % 16) == 0) ln.row_live = live_rows();      // x holds fresh rows in the listed blocks only: the others are not worth a pass
    return ln_fwd<T>(st, ln);
  }
  int64_t plan_installs = 0;               // workspace zero fills so far (realise_engine_plan_installs: tests / bench count them)
  int64_t plan_install_count() const override { return plan_installs; }
  int install_plan(hipStream_t st, const Plan& p) {
    if (p.total > ws_bytes) { fprintf(stderr, "[realise_hip] workspace too small: need %lld have %lld\n", (long long)p.total, (long long)ws_bytes); return RL_ERR_ARG; }
    pl = p;
    ++plan_installs;
    ln_epoch[0] = ln_epoch[1] = ln_epoch[2] = 0;           // (the arrival counters are zero-filled below)
    // The whole workspace starts at zero (once per plan): the self-cleaning accumulators need it (zero_once), and a live-row step
    // leaves the activation rows of padding tokens as they are - what they hold must be finite wherever a later pass multiplies it by
    // an exact zero (masked mean, gate gradients).
    if (hipMemsetAsync(ws, 0, (size_t)pl.total, st) != hipSuccess) return RL_ERR_LAUNCH;
    return RL_OK;
  }
  int forward(hipStream_t st, const realise_batch& b) override {
    if (!sh || !ws) return RL_ERR_ARG;
    if (b.B < 1 || b.S < 1 || b.S > cfg.max_pos) return RL_ERR_ARG;       // (S > 128: the tiled attention kernels, attention.hip)
    const int Tp = cfg.model_type == 1 ? b.Tp : 1;
    if (cfg.model_type == 1 && (Tp < 1 || !b.pho_idx || !b.pho_perm || !b.pho_lens_sorted || (!b.n_alive && !b.n_alive_dev))) return RL_ERR_ARG;
    if (pl.B != b.B || pl.S != b.S || pl.Tp != Tp) {
      RL_TRY(install_plan(st, make_plan(b.B, b.S, Tp)));
    }
    last = b;
    last.Tp = Tp;
    RL_TRY(wait_opt(st, 0));           // a pipelined optimizer sweep: its first piece holds everything the first kernels read
    have_glyph_fwd = false;            // the activations a glyph_backward would read are about to be overwritten
    dead_ok = false; rows_live = false; cls_compact = false;
    last_alive.assign(Tp, 0);
    // host counts (the reference's contract: pho_lens is a host list) or, after realise_build_pho, device counts: every
    // step is then launched over all B*S rows and bounded on the device
    if (cfg.model_type == 1) for (int t = 0; t < Tp; ++t) last_alive[t] = b.n_alive ? b.n_alive[t] : b.B * b.S;
    alive_dev = (cfg.model_type == 1 && !b.n_alive) ? b.n_alive_dev : nullptr;
    last.n_alive = nullptr;
    have_fwd = false;
    const int Tk = b.B * b.S;
    RL_TRY(mask_to_additive(st, b.masks, wp<float>(pl.mask_add), Tk));
    // everything below (and the backward) reads range-checked copies of the ids: a bad id raises in the module, never faults here
    RL_TRY(sanitize_ids(st, b.src_idx, Tk, V, wp<int64_t>(pl.ids_clean), id_flag));
    last.src_idx = wp<int64_t>(pl.ids_clean);
    if (cfg.model_type == 1) {
      RL_TRY(sanitize_ids(st, b.pho_idx, (int64_t)Tk * Tp, cfg.pho_vocab, wp<int64_t>(pl.pho_clean), id_flag));
      last.pho_idx = wp<int64_t>(pl.pho_clean);
    }
    // Rows after a sentence's last real / loss position are padding no query attends to and no loss term reads: every backward
    // activation row there is an exact zero (the embedding scatter has relied on it since round 1).  The backward skips them
    // (LayerNorm backward rows, blocks of the weight-gradient reductions), and a live-row step (g_live_rows) does not compute
    // their forward activations in the transformer stacks either.
    dead_ok = b.tgt_idx != nullptr && b.loss_masks != nullptr && g_skip_dead && b.want_dlogits && b.masks != nullptr && (Tk % 64) == 0 &&
              Tk / live_list_rows() <= TN_LIST_MAX_ENTRIES;      // (the weight-gradient kernels keep the block list in LDS: beyond 65536 bf16 token rows the step is dense)
    if (dead_ok) RL_TRY(row_liveness(st, b.masks, b.loss_masks, pl.B, pl.S, wp<uint8_t>(pl.row_live), wp<int>(pl.live_t64), wp<int>(pl.live_t32), wp<int>(pl.live_t16), wp<int>(pl.live_n), wp<int>(pl.live_rlen), wp<int>(pl.live_rows)));
    // (the weight gradients of a live-row step must walk the same block list - only the grouped launch takes one: ADVICE round 4)
    rows_live = dead_ok && b.training && g_live_rows && g_wgrad_group && live16() && (H % 64) == 0 && (I % 64) == 0 && (int64_t)Tk * I * 2 < 0xFFFFFF00ll;
    // Round 6, opt-in (realise_batch.eval_live_rows): an EVALUATION forward over the live rows as well.  The reference's evaluation reads
    // the predictions of a sentence's real tokens only (run.py:262-270 cuts them at `lengths`; the loss counts loss_masks rows), so the
    // transformer stacks need not compute the padding rows behind a sentence's last attended / loss position: their logits rows come
    // out finite and meaningless, every other row and the loss bit-identical to the dense forward's (same kernels per row).
    if (!b.training && !dead_ok && b.eval_live_rows && g_live_rows && live16() && b.masks != nullptr && (Tk % 64) == 0 && (H % 64) == 0 &&
        (I % 64) == 0 && (int64_t)Tk * I * 2 < 0xFFFFFF00ll) {
      RL_TRY(row_liveness(st, b.masks, b.loss_masks, pl.B, pl.S, wp<uint8_t>(pl.row_live), wp<int>(pl.live_t64), wp<int>(pl.live_t32), wp<int>(pl.live_t16), wp<int>(pl.live_n), wp<int>(pl.live_rlen), wp<int>(pl.live_rows)));
      dead_ok = true; rows_live = true;      // (the tables exist; no backward follows an evaluation forward)
    }
    const T* bert_h = nullptr;
    const bool ovl = cfg.model_type == 1 && g_branch_overlap && branches_ok();
    hipStream_t s_pho = ovl ? bst[0] : st, s_glyph = ovl ? bst[1] : st;
    // operand copies refreshed on the branch streams (refresh_shadows): with the branch streams in use below, each branch is already
    // ordered behind its own copies and the caller's stream meets the classifier / output-stack copies at the join; otherwise wait here
    if (!ovl) RL_TRY(wait_shadows(st));
    if (ovl) RL_TRY(fork(st));                                    // bert | pinyin GRU + pho_model | glyph ResNet
    // Enqueue order (g_fwd_order): the host needs ~0.6 ms to enqueue the 150 launches of the bert stack; enqueued first (0) the two
    // shorter branches reach the GPU only after that, enqueued last (1) they start at once and bert joins them ~0.4 ms later.
    if (!(ovl && g_fwd_order == 1)) RL_TRY(stack_forward(st, 0, L.bert, sh_bert, pl.bert, last.src_idx, nullptr, 0, &bert_h));
    const T* top = bert_h;
    if (cfg.model_type == 1) {
      RL_TRY(wait_opt(s_pho, opt_groups));                          // (pipelined sweep: pinyin / output stacks and the GRU are one piece)
      RL_TRY(gru_forward(s_pho));
      const T* pho_h = nullptr;
      RL_TRY(stack_forward(s_pho, 1, L.pho, sh_pho, pl.pho, nullptr, wp<T>(pl.gru_out), 0, &pho_h));
      const T* res = nullptr;
      RL_TRY(resnet_forward(s_glyph, last.src_idx, &res));
      {
        LnFwdArgs<T> ln; ln.rows = Tk; ln.H = H; ln.x = res; ln.row_index = wp<int>(pl.gu_inv);      // token t reads its glyph's row
        ln.gamma = pp(L.res_ln_g); ln.beta = pp(L.res_ln_b); ln.eps = cfg.ln_eps;
        ln.y = wp<T>(pl.res_h); ln.xhat = bwd_follows() ? wp<T>(pl.res_xhat) : nullptr; ln.rstd = wp<float>(pl.res_rstd);
        RL_TRY(ln_fwd<T>(s_glyph, ln));
      }
      if (ovl && g_fwd_order == 1) { RL_TRY(stack_forward(st, 0, L.bert, sh_bert, pl.bert, last.src_idx, nullptr, 0, &bert_h)); top = bert_h; }
      if (ovl) RL_TRY(join(st));
      RL_TRY(sync_optimizer(st));                                   // every piece of a pipelined sweep is now in front of this stream
      RL_TRY(gate_fwd<T>(st, gate_args()));
      RL_TRY(stack_forward(st, 2, L.outb, sh_out, pl.outb, nullptr, wp<T>(pl.fused), 1, &top));
    }
    RL_TRY(sync_optimizer(st));
    const DropParams dfin = site(5000, cfg.hidden_dropout);
    const T* cls_in = top;
    if (dfin.thresh) { RL_TRY(dropout_apply<T>(st, top, wp<T>(pl.out_d), Tk, H, dfin)); cls_in = wp<T>(pl.out_d); }
    if (b.logits_out == nullptr) {
      // K13 (round 6; models.py:859-869 as the training loop consumes it, run.py:191 takes the loss alone): no [B*S, V] logits.  The
      // classifier runs over the rows that enter the loss only - listed first, the classifier input gathered to match - and writes
      // their logits rows, compacted and at the gradient's row pitch, INTO the gradient buffer; the cross-entropy kernel reads each
      // row once into registers and stores the row's gradient over it.  Same fp32 accumulators, same bf16 rounding, same row kernel
      // as the two-buffer form: loss and every gradient bit-identical (tests/test_round6_gpu.py).
      if (sizeof(T) != 2 || !b.tgt_idx || !b.loss_masks || !b.loss_out || !b.want_dlogits || !g_cls_compact || Tk > 65536) return RL_ERR_ARG;
      cls_compact = true;
      CeCompact cc; cc.act_idx = wp<int>(pl.cls_act); cc.inv = wp<int>(pl.cls_inv); cc.n_act = wp<int>(pl.cls_nact);
      cc.phase = 1;
      RL_TRY(ce_loss<T>(st, nullptr, Vp, b.tgt_idx, b.loss_masks, Tk, V, b.loss_out, wp<float>(pl.count), wp<T>(pl.dlogits),
                        wp<float>(pl.loss_internal), Vp, cc));
      RL_TRY(gather_rows<T>(st, cls_in, wp<int>(pl.cls_act), wp<int>(pl.cls_nact), Tk, H, wp<T>(pl.cls_xc)));
      EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = wp<T>(pl.dlogits); ep.ldo = Vp; ep.bias = pp(L.cls_b);
      RL_TRY(gemm_nt<T>(st, wp<T>(pl.cls_xc), H, sp<T>(sh_cls_w), H, Tk, V, H, ep, wp<int>(pl.cls_nact)));
      cc.phase = 2; cc.logits_compact = 1;
      RL_TRY(ce_loss<T>(st, wp<T>(pl.dlogits), Vp, b.tgt_idx, b.loss_masks, Tk, V, b.loss_out, wp<float>(pl.count), wp<T>(pl.dlogits),
                        wp<float>(pl.loss_internal), Vp, cc));
      have_fwd = b.training != 0;
      return RL_OK;
    }
    {  // tied vocabulary classifier (models.py:859)
      EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = (T*)b.logits_out; ep.ldo = V; ep.bias = pp(L.cls_b);
      // logits_f32_out (round 6): the fp32 logits the reference returns (models.py:859) written by the classifier's own epilogue next
      // to the compute-dtype ones the loss reads - no cast pass over [B*S, V]; a shape the persistent kernel does not take is cast here
      int rc = RL_ERR_ARG;
      if (b.logits_f32_out != nullptr && sizeof(T) == 2) {
        EpiParams<T> e2 = ep; e2.out_f32 = b.logits_f32_out; e2.ldo_f32 = V;
        rc = gemm_nt<T>(st, cls_in, H, sp<T>(sh_cls_w), H, Tk, V, H, e2);
        if (rc != RL_OK && rc != RL_ERR_ARG) return rc;
      }
      if (rc != RL_OK) {
        RL_TRY(gemm_nt<T>(st, cls_in, H, sp<T>(sh_cls_w), H, Tk, V, H, ep));
        if (b.logits_f32_out != nullptr) RL_TRY(cast_to_f32<T>(st, (const T*)b.logits_out, b.logits_f32_out, (int64_t)Tk * V));
      }
    }
    if (b.tgt_idx != nullptr) {
      if (!b.loss_masks || !b.loss_out) return RL_ERR_ARG;
      // Backward of the classifier over the rows that enter the loss only (loss_masks: no [CLS] / [SEP] / padding - 60 % of the
      // rows of a SIGHAN-shaped batch): the gradient rows are written compacted, the classifier input is gathered to match; the
      // other rows' gradients are exact zeros in the dense form, so the weight / bias / data gradients are the same sums.
      // (the padding rows the backward skips: row_liveness at the top of this call)
      cls_compact = g_cls_compact && b.want_dlogits && Tk <= 65536;
      CeCompact cc;
      if (cls_compact) { cc.act_idx = wp<int>(pl.cls_act); cc.inv = wp<int>(pl.cls_inv); cc.n_act = wp<int>(pl.cls_nact); }
      RL_TRY(ce_loss<T>(st, (const T*)b.logits_out, V, b.tgt_idx, b.loss_masks, Tk, V, b.loss_out, wp<float>(pl.count),
                        b.want_dlogits ? wp<T>(pl.dlogits) : nullptr, wp<float>(pl.loss_internal), Vp, cc));
      if (cls_compact) RL_TRY(gather_rows<T>(st, cls_in, wp<int>(pl.cls_act), wp<int>(pl.cls_nact), Tk, H, wp<T>(pl.cls_xc)));
    }
    have_fwd = b.training && b.tgt_idx != nullptr && b.want_dlogits;
    return RL_OK;
  }

  // ---------------------------------------------------------------- glyph-only entry points (BASELINE configs[3])
  // CharResNet alone (src/char_cnn.py:46-55) on the B*S glyph stacks selected by src_idx (src/models.py:829-836):
  // res_out[t, :] = resnet(char_images_multifonts[src_idx[t]]), before resnet_layernorm.
  bool have_glyph_fwd = false;
  int glyph_forward(hipStream_t st, const int64_t* ids, int B, int S, int training, void* res_out) override {
    if (!sh || !ws || cfg.model_type != 1 || !ids || !res_out || B < 1 || S < 1) return RL_ERR_ARG;
    RL_TRY(sync_optimizer(st));
    if (pl.B != B || pl.S != S || pl.Tp != -1) {
      RL_TRY(install_plan(st, make_plan(B, S, -1)));
    }
    last = realise_batch();
    last.B = B; last.S = S; last.Tp = 1; last.training = training;
    RL_TRY(wait_shadows(st));
    RL_TRY(sanitize_ids(st, ids, (int64_t)B * S, V, wp<int64_t>(pl.ids_clean), id_flag));
    last.src_idx = wp<int64_t>(pl.ids_clean);
    have_fwd = false; have_glyph_fwd = false;
    const T* res = nullptr;
    RL_TRY(resnet_forward(st, last.src_idx, &res));
    RL_TRY(gather_rows<T>(st, res, wp<int>(pl.gu_inv), B * S, H, (T*)res_out));
    have_glyph_fwd = training != 0;
    return RL_OK;
  }
  // accumulates the gradients of the 15 conv / BatchNorm parameter tensors for d_res [B*S, 768] (per token)
  int glyph_backward(hipStream_t st, const void* d_res) override {
    if (!have_glyph_fwd || !d_res) { fprintf(stderr, "[realise_hip] glyph_backward without a training glyph_forward\n"); return RL_ERR_ARG; }
    RL_TRY(sync_optimizer(st));
    RL_TRY(begin_gradient_pass(st, true));
    const int Tk = pl.B * pl.S;
    RL_TRY(segment_sum<T>(st, (const T*)d_res, wp<int>(pl.gu_inv), Tk, H, wp<float>(pl.seg_acc), wp<T>(pl.r_dout), wp<int>(pl.gu_bounds)));
    RL_TRY(resnet_backward(st, wp<T>(pl.r_dout)));
    return RL_OK;
  }

  // ---------------------------------------------------------------- fresh gradients (zero_grad() without a 680 MB memset)
  // After zero_grad() the module only tells the engine that the gradient arena holds nothing (set_grads_fresh).  The next
  // backward then (a) zero-fills, in ONE launch, everything EXCEPT the Linear weight gradients of the transformer layers - 80 % of
  // the arena -, and (b) lets the grouped weight-gradient GEMM store those instead of adding to them: no fill of those bytes and no
  // read-modify-write in its epilogue.  Accumulating backwards (no zero_grad() in between) run exactly as before.
  bool grads_fresh = false, pass_overwrite = false, fill_built = false;
  int n_fill = 0;
  void set_grads_fresh(int fresh) override { grads_fresh = fresh != 0; }
  int begin_gradient_pass(hipStream_t st, bool glyph_only) {
    pass_overwrite = false;
    if (!grads_fresh) return RL_OK;
    grads_fresh = false;
    const int64_t total = L.arena_elems[AR_TRAIN];
    if (glyph_only || !g_wgrad_group || !sh) return fill_f32(st, G, 0.0f, total);
    if (!fill_built) {
      std::vector<std::pair<int64_t, int64_t>> skip;          // [begin, end) of the tensors the grouped launches overwrite
      auto add_stack = [&](const StackOff& so) {
        for (const LayerOff& o : so.layers) {
          skip.push_back({o.qkv_w, o.qkv_w + 3LL * H * H}); skip.push_back({o.ao_w, o.ao_w + (int64_t)H * H});
          skip.push_back({o.in_w, o.in_w + (int64_t)I * H}); skip.push_back({o.out_w, o.out_w + (int64_t)H * I});
        }
      };
      add_stack(L.bert);
      if (cfg.model_type == 1) { add_stack(L.pho); add_stack(L.outb); }
      std::sort(skip.begin(), skip.end());
      fill_host.clear();
      int64_t at = 0;
      auto emit = [&](int64_t b, int64_t e) {
        for (int64_t o = b; o < e; o += (1 << 20)) fill_host.push_back(FillChunk{o, (int32_t)std::min<int64_t>(1 << 20, e - o), 0});
      };
      for (auto& r : skip) { if (r.first > at) emit(at, r.first); at = std::max(at, r.second); }
      if (at < total) emit(at, total);
      if ((int)fill_host.size() > FILL_MAX) return fill_f32(st, G, 0.0f, total);
      if (hipMemcpyAsync(sh + sh_fill, fill_host.data(), fill_host.size() * sizeof(FillChunk), hipMemcpyHostToDevice, st) != hipSuccess) return RL_ERR_LAUNCH;
      n_fill = (int)fill_host.size(); fill_built = true;
    }
    RL_TRY(zero_chunks(st, G, (const FillChunk*)(sh + sh_fill), n_fill));
    pass_overwrite = true;
    return RL_OK;
  }
  std::vector<FillChunk> fill_host;

  // ---------------------------------------------------------------- backward, in bucket-sized stages
  int n_stages() const { return (int)L.buckets.size(); }

  // d loss arriving at backward() as a DEVICE scalar (nullptr: 1): the head's three gradients are linear in it, so it is applied where
  // they leave the head - the classifier's weight / bias gradient epilogues (TnEpi::alpha_dev) and the scatter that writes d(top
  // hidden) - instead of a read-modify-write pass over the ~200 MB of cross-entropy gradient rows
  const float* loss_grad = nullptr;
  void set_loss_grad(const float* g) override { loss_grad = g; }
  int stage_head(hipStream_t st) {      // classifier + final dropout ; leaves d(top hidden) in gA
    const int Tk = pl.B * pl.S;
    const T* dl = wp<T>(pl.dlogits);
    T* gA = wp<T>(pl.gA);
    const DropParams dfin = site(5000, cfg.hidden_dropout);
    const T* top = cfg.model_type == 1 ? wp<T>(pl.outb.layers.back().y2) : wp<T>(pl.bert.layers.back().y2);
    const T* cls_in = dfin.thresh ? wp<T>(pl.out_d) : top;
    if (cls_compact) {
      const int* n_act = wp<int>(pl.cls_nact);
      { TnEpi te; te.slab = wp<float>(pl.tn_slab); te.slab_elems = TN_SLAB_ELEMS; te.colsum = gp(L.cls_b); te.out = gp(L.cls_w); te.ldo = H;
        te.alpha_dev = loss_grad;
        RL_TRY(gemm_tn<T>(st, dl, Vp, wp<T>(pl.cls_xc), H, Tk, V, H, te, n_act)); }
      if constexpr (sizeof(T) == 2) {
        const int ns = g_cls_splitk;
        if (ns >= 2 && pl.cls_slab != 0 && Tk >= 1024 &&
            gemm_nt8_splitk(st, dl, Vp, sp<T>(sh_cls_wT), Vp, Tk, H, Vp, ns, wp<float>(pl.cls_slab), (int64_t)Tk * H, n_act) == RL_OK)
          return scatter_rows_drop_slab<T>(st, wp<float>(pl.cls_slab), ns, (int64_t)Tk * H, wp<int>(pl.cls_inv), Tk, H, gA, dfin, loss_grad);
      }
      { EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = wp<T>(pl.cls_gc); ep.ldo = H; ep.m_dev = n_act;
        RL_TRY(gemm_nt<T>(st, dl, Vp, sp<T>(sh_cls_wT), Vp, Tk, H, Vp, ep)); }
      return scatter_rows_drop<T>(st, wp<T>(pl.cls_gc), wp<int>(pl.cls_inv), Tk, H, gA, dfin, loss_grad);      // rows outside the loss: zero; + the final dropout's map
    }
    { TnEpi te; te.slab = wp<float>(pl.tn_slab); te.slab_elems = TN_SLAB_ELEMS; te.colsum = gp(L.cls_b); te.out = gp(L.cls_w); te.ldo = H; te.alpha_dev = loss_grad;
      RL_TRY(gemm_tn<T>(st, dl, Vp, cls_in, H, Tk, V, H, te)); }
    { EpiParams<T> ep; ep.mode = EPI_STORE; ep.out = gA; ep.ldo = H;
      RL_TRY(gemm_nt<T>(st, dl, Vp, sp<T>(sh_cls_wT), Vp, Tk, H, Vp, ep)); }      // K = Vp: the padding columns are exact zeros on both sides
    if (loss_grad != nullptr) RL_TRY(scale_by_dev<T>(st, gA, (int64_t)Tk * H, loss_grad));      // (dense fallback: a pass over [T, H], not over the logits)
    if (dfin.thresh) RL_TRY(dropout_apply<T>(st, gA, gA, Tk, H, dfin));
    return RL_OK;
  }

  int run_stage(hipStream_t st, int s) {
    const int Tk = pl.B * pl.S;
    T* gA = wp<T>(pl.gA);
    if (cfg.model_type == 0) {
      // stage g (0..groups-1): bert layer group g (stage 0 also runs the head); last stage: embeddings
      const int groups = L.bert_groups;
      if (s < groups) {
        if (s == 0) RL_TRY(stage_head(st));
        int hi = cfg.bert_layers - 1 - 4 * s;
        int lo = hi - 3 > 0 ? hi - 3 : 0;
        RL_TRY(layers_backward(st, 0, L.bert, sh_bert, pl.bert, hi, lo, gA));
      } else {
        RL_TRY(emb_backward(st, 0, L.bert, pl.bert, last.src_idx, 0, gA, wp<T>(pl.sc[0].gB)));
      }
      return RL_OK;
    }
    switch (s) {
      case 0: RL_TRY(stage_out_block(st)); break;
      case 1: RL_TRY(stage_gate(st)); RL_TRY(stage_glyph(st)); break;
      case 2: RL_TRY(stage_pho(st)); break;
      default: RL_TRY(stage_bert(st, s - 3)); break;
    }
    return RL_OK;
  }
  int stage_out_block(hipStream_t st) {      // head + output_block; leaves d fused in gB (set 0)
    T* gA = wp<T>(pl.gA);
    RL_TRY(stage_head(st));
    RL_TRY(layers_backward(st, 2, L.outb, sh_out, pl.outb, cfg.out_layers - 1, 0, gA));
    RL_TRY(emb_backward(st, 2, L.outb, pl.outb, nullptr, 1, gA, wp<T>(pl.sc[0].gB)));
    return RL_OK;
  }
  int stage_gate(hipStream_t st) {           // d fused -> X1 (d bert), X2 (d pho), X3 (d res) + gate_net gradients
    GateArgs<T> g = gate_args();
    g.dfused = wp<T>(pl.sc[0].gB); g.dbert = wp<T>(pl.X1); g.dpho = wp<T>(pl.X2); g.dres = wp<T>(pl.X3); g.dz = wp<float>(pl.dz);
    g.dW = gp(L.gate_w); g.dbias = gp(L.gate_b); g.row_live = live_rows();
    return gate_bwd<T>(st, g);
  }
  int stage_glyph(hipStream_t st) {          // resnet LayerNorm, segment sum over the tokens of a glyph, glyph ResNet
    const int Tk = pl.B * pl.S;
    const typename Plan::Scratch& sc = pl.sc[cs];
    LnBwdArgs<T> ln; ln.slots = ln_region(LN_FOLD_MAX); ln.rows = Tk; ln.H = H; ln.dy = wp<T>(pl.X3); ln.xhat = wp<T>(pl.res_xhat); ln.rstd = wp<float>(pl.res_rstd);
    ln.row_live = live_rows();
    ln.gamma = pp(L.res_ln_g); ln.dx = wp<T>(sc.gE); ln.dgamma = gp(L.res_ln_g); ln.dbeta = gp(L.res_ln_b);
    RL_TRY(ln_bwd<T>(st, ln));
    RL_TRY(segment_sum<T>(st, wp<T>(sc.gE), wp<int>(pl.gu_inv), Tk, H, wp<float>(pl.seg_acc), wp<T>(pl.r_dout), wp<int>(pl.gu_bounds)));
    return resnet_backward(st, wp<T>(pl.r_dout));
  }
  int stage_pho(hipStream_t st) {            // pho_model + GRU
    T* g2 = wp<T>(pl.X2);
    T* gE = wp<T>(pl.sc[cs].gE);
    RL_TRY(layers_backward(st, 1, L.pho, sh_pho, pl.pho, cfg.pho_layers - 1, 0, g2));
    RL_TRY(emb_backward(st, 1, L.pho, pl.pho, nullptr, 0, g2, gE));       // gE = d gru_out (original order)
    return gru_backward(st, gE);
  }
  int stage_bert(hipStream_t st, int g) {    // bert layer group g, or (g == groups) its embeddings
    T* g1 = wp<T>(pl.X1);
    if (g < L.bert_groups) {
      int hi = cfg.bert_layers - 1 - 4 * g;
      int lo = hi - 3 > 0 ? hi - 3 : 0;
      return layers_backward(st, 0, L.bert, sh_bert, pl.bert, hi, lo, g1);
    }
    return emb_backward(st, 0, L.bert, pl.bert, last.src_idx, 0, g1, wp<T>(pl.sc[0].gB));
  }

  bool branch_mode = false;
  // Whole backward pass with a "bucket final" signal per gradient bucket: evs[i] (caller-owned hipEvent_t) is recorded once every
  // kernel that writes into bucket i has been enqueued, on the stream that runs last for that bucket - so a data-parallel caller can
  // start the all-reduce of bucket i on its own stream (hipStreamWaitEvent) while the three model branches and the deferred weight
  // gradients keep running.  The per-bucket form of backward() would serialise the branches and join the weight-gradient stream at
  // every bucket boundary.
  std::vector<hipEvent_t> ev_sig;       // one per bucket (any number of bert groups)
  int signal_bucket(void* ev, hipStream_t s, bool through_side, int slot) {
    hipEvent_t e = (hipEvent_t)ev;
    if ((int)ev_sig.size() <= slot) ev_sig.resize(slot + 1, nullptr);
    if (through_side && g_wgrad_overlap && side_ok()) {      // the bert branch: its weight gradients finish on the side stream
      if (ev_sig[slot] == nullptr && hipEventCreateWithFlags(&ev_sig[slot], hipEventDisableTiming) != hipSuccess) return RL_ERR_LAUNCH;
      if (hipEventRecord(ev_sig[slot], s) != hipSuccess || hipStreamWaitEvent(side, ev_sig[slot], 0) != hipSuccess ||
          hipEventRecord(e, side) != hipSuccess) return RL_ERR_LAUNCH;
      return RL_OK;
    }
    return hipEventRecord(e, s) == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
  }
  int backward_signalled(hipStream_t st, void* const* evs, int n_events) override {
    if (!have_fwd) { fprintf(stderr, "[realise_hip] backward without a training forward (tgt_idx + want_dlogits)\n"); return RL_ERR_ARG; }
    const int n = n_stages();
    if (n_events != n || evs == nullptr) return RL_ERR_ARG;
    for (int i = 0; i < n; ++i) if (evs[i] == nullptr) return RL_ERR_ARG;
    RL_TRY(begin_gradient_pass(st, false));
    cs = 0;
    if (cfg.model_type == 1 && g_branch_overlap && branches_ok()) {
      branch_mode = true;
      int rc = stage_out_block(st);
      if (rc == RL_OK) rc = signal_bucket(evs[0], st, true, 0);
      if (rc == RL_OK) rc = stage_gate(st);
      if (rc == RL_OK) rc = fork(st);
      if (rc == RL_OK) { cs = 2; rc = stage_glyph(bst[1]); }
      if (rc == RL_OK) rc = signal_bucket(evs[1], bst[1], false, 1);          // gate (before the fork) + glyph ResNet
      if (rc == RL_OK) { cs = 1; rc = stage_pho(bst[0]); }
      if (rc == RL_OK) rc = signal_bucket(evs[2], bst[0], false, 2);
      cs = 0;
      for (int g = 0; rc == RL_OK && g <= L.bert_groups; ++g) {
        rc = stage_bert(st, g);
        if (rc == RL_OK) rc = signal_bucket(evs[3 + g], st, true, 3 + g);
      }
      const int rj = join(st);
      const int rs = join_side(st);
      branch_mode = false;
      return rc != RL_OK ? rc : (rj != RL_OK ? rj : rs);
    }
    for (int s = 0; s < n; ++s) {
      RL_TRY(run_stage(st, s));
      RL_TRY(join_side(st));
      RL_TRY(signal_bucket(evs[s], st, false, s));
    }
    return RL_OK;
  }
  int backward(hipStream_t st, int first, int last_stage) override {
    if (!have_fwd) { fprintf(stderr, "[realise_hip] backward without a training forward (tgt_idx + want_dlogits)\n"); return RL_ERR_ARG; }
    const int n = n_stages();
    if (last_stage < 0) last_stage = n - 1;
    if (first < 0 || last_stage >= n || first > last_stage) return RL_ERR_ARG;
    if (first == 0) RL_TRY(begin_gradient_pass(st, false));           // (a bucket-by-bucket caller starts its pass with stage 0)
    cs = 0;
    // whole pass in one call (no per-bucket gradient exchange in between): the three branches behind the gate run concurrently
    if (cfg.model_type == 1 && first == 0 && last_stage == n - 1 && g_branch_overlap && branches_ok()) {
      branch_mode = true;
      int rc = stage_out_block(st);
      if (rc == RL_OK) rc = stage_gate(st);
      if (rc == RL_OK) rc = fork(st);
      if (rc == RL_OK) { cs = 2; rc = stage_glyph(bst[1]); }
      if (rc == RL_OK) { cs = 1; rc = stage_pho(bst[0]); }
      cs = 0;
      for (int g = 0; rc == RL_OK && g <= L.bert_groups; ++g) rc = stage_bert(st, g);
      const int rj = join(st);
      const int rs = join_side(st);
      branch_mode = false;
      return rc != RL_OK ? rc : (rj != RL_OK ? rj : rs);
    }
    for (int s = first; s <= last_stage; ++s) RL_TRY(run_stage(st, s));
    return join_side(st);          // the caller's stream owns every gradient of these stages once this returns
  }
};

EngineBase* make_engine(const realise_config& c, float* p, float* g, float* pu, float* fz, float* bf, int64_t* bi) {
  if (c.hidden % 64 || c.hidden / c.heads != 64 || c.hidden > 1024 || (c.intermediate % 8) || (c.vocab % 8)) return nullptr;
  if (c.model_type == 1 && c.hidden != 768) return nullptr;     // CharResNet output is 768 wide (char_cnn.py:44)
  if (c.dtype == REALISE_BF16) return new Engine<bf16_t>(c, p, g, pu, fz, bf, bi);
  if (c.dtype == REALISE_F32) return new Engine<float>(c, p, g, pu, fz, bf, bi);
  return nullptr;
}

}  // namespace rl
