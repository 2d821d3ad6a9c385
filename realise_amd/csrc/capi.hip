// extern "C" surface of librealise_hip.so (declared in include/realise_hip.h; diagnostics in include/realise_hip_debug.h).
#include <math.h>
#include <string.h>

#include "attention.h"
#include "engine.h"
#include "gemm.h"
#include "layout.h"
#include "ops.h"
#include "prof.h"
#include "../../include/realise_hip_debug.h"

using namespace rl;
#define RL_TRY(expr) do { const int _rc = (expr); if (_rc != RL_OK) return _rc; } while (0)

struct realise_engine { EngineBase* impl; };

namespace {
template <typename T> EpiParams<T> to_epi(const realise_epilogue* e) {
  EpiParams<T> p;
  if (!e) return p;
  p.mode = e->mode; p.accumulate = e->accumulate; p.out = (T*)e->out; p.ldo = e->ldo; p.out2 = (T*)e->out2;
  p.bias = e->bias; p.aux = (const T*)e->aux; p.ldaux = e->ldaux; p.alpha = e->alpha;
  p.drop_seed = e->drop_seed; p.drop_thresh = e->drop_thresh; p.drop_scale = e->drop_scale;
  return p;
}
template <typename T> ConvLoader<T> to_geom(const realise_conv_geom* g) {
  ConvLoader<T> c;
  c.src = (const T*)g->src; c.img_index = g->img_index; c.rows = g->rows; c.Hr = g->Hr; c.Wr = g->Wr; c.Hs = g->Hs; c.Ws = g->Ws;
  c.C = g->C; c.KH = g->KH; c.KW = g->KW; c.stride = g->stride; c.pad = g->pad; c.mode = g->mode;
  c.finalize();
  return c;
}
}  // namespace

template <typename T>
static int conv_dgrad_s2(hipStream_t st, const realise_conv_geom* a, const T* B, int64_t ldb, int Cin, const realise_epilogue* ep) {
  if (a->mode != 1 || a->stride != 2 || a->Hr != a->Wr || (a->Hr & (a->Hr - 1)) || a->Hr < 2 || a->img_index != nullptr) return RL_ERR_ARG;
  int hsh = 0; while ((1 << hsh) < a->Hr) ++hsh;
  const int nimg = a->rows / (a->Hr * a->Wr);
  for (int c = 0; c < 4; ++c) {
    int first = 0;
    const int nt = conv_s2_class(a->KH, a->KW, a->pad, c, &first, nullptr);
    if (nt == 0) { if (!ep->accumulate) return RL_ERR_ARG; continue; }      // pixels no tap reaches: only meaningful when accumulating
    ConvLoader<T> g = to_geom<T>(a);
    g.rows = nimg * (a->Hr / 2) * (a->Wr / 2);
    g.par = c;
    int ntaps = nt;
    if (a->Hr == 2 && a->Hs == 1 && a->KH == 3 && a->KW == 3 && a->pad == 1) {      // one reachable tap per pixel of a 2x2 map
      g.tap_sel = ((c >> 1) + 1) * 3 + (c & 1) + 1;
      first = conv_s2_slot(3, 3, 1, (c >> 1) + 1, (c & 1) + 1); ntaps = 1;
    }
    EpiParams<T> e = to_epi<T>(ep);
    e.rm_hw_shift = 2 * hsh; e.rm_w_shift = hsh; e.rm_par = c;
    const int rc = gemm_nt_conv<T>(st, g, B + (int64_t)first * a->C, ldb, g.rows, Cin, ntaps * a->C, e);
    if (rc != RL_OK) return rc;
  }
  return RL_OK;
}

template <typename T>
static int tn_grouped(hipStream_t st, int n, const realise_tn_problem* pr, int P, const int* live = nullptr, const int* n_live = nullptr,
                      int list_rows = 0, int overwrite = 0) {
  if (n < 1 || n > TN_GROUP_MAX || pr == nullptr) return RL_ERR_ARG;
  TnGroupProblem<T> g[TN_GROUP_MAX];
  for (int k = 0; k < n; ++k) {
    g[k].A = (const T*)pr[k].A; g[k].lda = pr[k].lda; g[k].B = (const T*)pr[k].B; g[k].ldb = pr[k].ldb;
    g[k].I = pr[k].I; g[k].J = pr[k].J; g[k].out = pr[k].out; g[k].ldo = pr[k].ldo; g[k].colsum = pr[k].colsum;
  }
  return gemm_tn_group<T>(st, n, g, P, 1.0f, overwrite, live, n_live, list_rows);
}

extern "C" {

const char* realise_version(void) { return RL_PROBES ? "realise_hip 0.3 (gfx950) +probes" : "realise_hip 0.3 (gfx950)"; }

int realise_gemm_nt(void* stream, int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                    const realise_epilogue* ep) {
  hipStream_t st = (hipStream_t)stream;
  if (!ep || !ep->out) return RL_ERR_ARG;
  if (dtype == REALISE_BF16) return gemm_nt<bf16_t>(st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, M, N, K, to_epi<bf16_t>(ep));
  if (dtype == REALISE_F32) return gemm_nt<float>(st, (const float*)A, lda, (const float*)B, ldb, M, N, K, to_epi<float>(ep));
  return RL_ERR_ARG;
}
int realise_conv_nt(void* stream, int dtype, const realise_conv_geom* a, const void* B, int64_t ldb, int M, int N, int K,
                    const realise_epilogue* ep) {
  hipStream_t st = (hipStream_t)stream;
  if (!ep || !ep->out || !a) return RL_ERR_ARG;
  if (dtype == REALISE_BF16) return gemm_nt_conv<bf16_t>(st, to_geom<bf16_t>(a), (const bf16_t*)B, ldb, M, N, K, to_epi<bf16_t>(ep));
  if (dtype == REALISE_F32) return gemm_nt_conv<float>(st, to_geom<float>(a), (const float*)B, ldb, M, N, K, to_epi<float>(ep));
  return RL_ERR_ARG;
}
int realise_conv_dgrad_s2(void* stream, int dtype, const realise_conv_geom* a, const void* B_classes, int64_t ldb, int Cin,
                          const realise_epilogue* ep) {
  hipStream_t st = (hipStream_t)stream;
  if (!ep || !ep->out || !a) return RL_ERR_ARG;
  if (dtype == REALISE_BF16) return conv_dgrad_s2<bf16_t>(st, a, (const bf16_t*)B_classes, ldb, Cin, ep);
  if (dtype == REALISE_F32) return conv_dgrad_s2<float>(st, a, (const float*)B_classes, ldb, Cin, ep);
  return RL_ERR_ARG;
}
int realise_gemm_tn(void* stream, int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int P, int I, int J,
                    float* out, int64_t ldo, float* scratch, int64_t scratch_elems, float* colsum_out) {
  hipStream_t st = (hipStream_t)stream;
  TnEpi te; te.out = out; te.ldo = ldo; te.slab = scratch; te.slab_elems = scratch_elems; te.colsum = colsum_out;
  if (dtype == REALISE_BF16) return gemm_tn<bf16_t>(st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, P, I, J, te);
  if (dtype == REALISE_F32) return gemm_tn<float>(st, (const float*)A, lda, (const float*)B, ldb, P, I, J, te);
  return RL_ERR_ARG;
}
int realise_gemm_tn_grouped(void* stream, int dtype, int n, const realise_tn_problem* problems, int P) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == REALISE_BF16) return tn_grouped<bf16_t>(st, n, problems, P);
  if (dtype == REALISE_F32) return tn_grouped<float>(st, n, problems, P);
  return RL_ERR_ARG;
}
int realise_gemm_nt_rows(void* stream, int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                         const realise_epilogue* ep, const int* rows_dev) {
  hipStream_t st = (hipStream_t)stream;
  if (!ep || !ep->out || !rows_dev) return RL_ERR_ARG;
  if (dtype == REALISE_BF16) return gemm_nt<bf16_t>(st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, M, N, K, to_epi<bf16_t>(ep), rows_dev);
  if (dtype == REALISE_F32) return gemm_nt<float>(st, (const float*)A, lda, (const float*)B, ldb, M, N, K, to_epi<float>(ep), rows_dev);
  return RL_ERR_ARG;
}
int realise_gemm_nt_live(void* stream, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                         const realise_epilogue* ep, const int* live_list, const int* live_count) {
  if (!ep || !ep->out || !live_list || !live_count) return RL_ERR_ARG;
  EpiParams<bf16_t> e = to_epi<bf16_t>(ep);
  e.live_list = live_list; e.live_count = live_count;
  return gemm_nt8_live((hipStream_t)stream, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, M, N, K, e);
}
int realise_gemm_nt_live_rows(void* stream, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                              const realise_epilogue* ep, const int* row_list, const int* row_count) {
  if (!ep || !ep->out || !row_list || !row_count) return RL_ERR_ARG;
  EpiParams<bf16_t> e = to_epi<bf16_t>(ep);
  e.live_list = row_list; e.live_count = row_count; e.live_unit = 1;
  return gemm_nt8_live((hipStream_t)stream, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, M, N, K, e);
}
int realise_gemm_nt_streamk(void* stream, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                            const realise_epilogue* ep, const int* live_list, const int* live_count, float* part, int* flags, int tag, int* timeout) {
#if RL_PROBES
  if (!ep || !ep->out || !part || !flags) return RL_ERR_ARG;
  EpiParams<bf16_t> e = to_epi<bf16_t>(ep);
  e.live_list = live_list; e.live_count = live_count;
  e.sk_part = part; e.sk_flag = flags; e.sk_tag = tag; e.sk_timeout = timeout;
  return gemm_nt8s((hipStream_t)stream, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, M, N, K, e);
#else
  // the stream-K kernel (measured slower on every layer shape) ships in the probe build only
  (void)stream; (void)A; (void)lda; (void)B; (void)ldb; (void)M; (void)N; (void)K; (void)ep; (void)live_list; (void)live_count; (void)part; (void)flags; (void)tag; (void)timeout;
  return RL_ERR_ARG;
#endif
}
int realise_gemm_nt_splitk(void* stream, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int nsplit,
                           float* slab, int64_t slab_stride, const int* m_dev) {
  return gemm_nt8_splitk((hipStream_t)stream, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, M, N, K, nsplit, slab, slab_stride, m_dev);
}
int realise_gemm_tn_grouped_live(void* stream, int dtype, int n, const realise_tn_problem* problems, int P, const int* live,
                                 const int* n_live, int list_rows, int overwrite) {
  hipStream_t st = (hipStream_t)stream;
  if (live == nullptr || n_live == nullptr) return RL_ERR_ARG;
  if (dtype == REALISE_BF16) return tn_grouped<bf16_t>(st, n, problems, P, live, n_live, list_rows, overwrite);
  if (dtype == REALISE_F32) return tn_grouped<float>(st, n, problems, P, live, n_live, list_rows, overwrite);
  return RL_ERR_ARG;
}
int realise_conv_tn(void* stream, int dtype, const void* A, int64_t lda, const realise_conv_geom* b, int P, int Co, int Ci,
                    float* out, float* scratch, int64_t scratch_elems) {
  hipStream_t st = (hipStream_t)stream;
  if (!b) return RL_ERR_ARG;
  TnEpi te; te.mode = TN_CONVW; te.out = out; te.Cin = Ci; te.Cpad = b->C; te.KHW = b->KH * b->KW;
  te.slab = scratch; te.slab_elems = scratch_elems;
  const int J = b->KH * b->KW * b->C;
  if (dtype == REALISE_BF16) return gemm_tn_conv<bf16_t>(st, (const bf16_t*)A, lda, to_geom<bf16_t>(b), P, Co, J, te);
  if (dtype == REALISE_F32) return gemm_tn_conv<float>(st, (const float*)A, lda, to_geom<float>(b), P, Co, J, te);
  return RL_ERR_ARG;
}
void realise_set_tn_transpose_read(int enable) { set_tn_transpose_read(enable); }
void realise_set_nt_allow_n96(int on) { set_nt_allow_n96(on); }
void realise_set_nt_probe(int mode) { set_nt_probe(mode); }
void realise_set_nt_variant(int v) { set_nt_variant(v); }
void realise_set_nt_group_m(int g) { set_nt8_group_m(g); }
void realise_set_ln(int key, int value) {
  if (key == 0) set_ln_fast(value); else if (key == 1) set_ln_bwd_blocks(value); else if (key == 2) set_bn_fast(value); else if (key == 3) set_bn_chunks(value); else if (key == 4) set_ce_fast(value); else if (key == 5) set_ln_v2(value); else if (key == 6) set_adamw_reg(value);
}
void realise_set_engine(int key, int value) { if (key == 0) set_fwd_order(value); else if (key >= 1 && key <= 3) set_stream_priority(key - 1, value); else if (key == 4) set_cls_compact(value); else if (key == 5) set_skip_dead(value); else if (key == 6) set_cls_splitk(value); else if (key == 7) set_tn_group8(value); else if (key == 8) set_ln_fuse(value); else if (key == 9) set_gru_fuse(value); else if (key == 10) set_live_rows(value); else if (key == 11) set_streamk(value); else if (key == 12) set_streamk_min(value); else if (key == 13) set_glyph_fuse(value); else if (key == 14) set_bn_fold(value); else if (key == 15) set_opt_pipe(value); }
void realise_set_nt8p(int key, int value) { if (key == 0) set_nt8p_order(value); else if (key == 1) set_nt8p_wgs(value); else if (key == 2) set_nt8_single_round(value); else if (key == 3) set_nt8_live_gc(value); else if (key == 4) set_nt8_epi_pre(value); else if (key == 5) set_nt8_live_big(value); else if (key == 6) set_tn_jmajor(value); else if (key == 7) set_nt8_cu_pair(value); else if (key == 8) set_nt8_l2_prefetch(value); }
void realise_set_tn_probe(int mode) { set_tn_probe(mode); }
void realise_set_attn_probe(int mode) { set_attn_probe(mode); }
void realise_set_tn_split(int n) { set_tn_split(n); }
void realise_set_wgrad_group(int on) { set_wgrad_group(on); }
void realise_set_dgrad_parity(int on) { set_dgrad_parity(on); }
void realise_set_tn_variant(int v) { set_tn_variant(v); }
void realise_set_tn_group_ring(int on) { set_tn_group_ring(on); }
void realise_set_conv_c64(int on) { set_conv_c64(on); }
void realise_set_nt_wide_epilogue(int on) { set_nt_wide_epilogue(on); }
void realise_set_glyph_dedup(int on) { set_glyph_dedup(on); }
void realise_set_wgrad_overlap(int on) { set_wgrad_overlap(on); }
void realise_set_branch_overlap(int on) { set_branch_overlap(on); }

int realise_attention_fwd(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ldq, const float* mask_add,
                          void* ctx, int64_t ldc, float* lse, int B, int nh, int S, uint32_t drop_seed, uint32_t drop_thresh,
                          float drop_scale) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == REALISE_BF16)
    return attn_fwd<bf16_t>(st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, mask_add, (bf16_t*)ctx, ldc, lse, B, nh, S,
                            drop_seed, drop_thresh, drop_scale);
  if (dtype == REALISE_F32)
    return attn_fwd<float>(st, (const float*)q, (const float*)k, (const float*)v, ldq, mask_add, (float*)ctx, ldc, lse, B, nh, S,
                           drop_seed, drop_thresh, drop_scale);
  return RL_ERR_ARG;
}
int realise_attention_bwd(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ldq, const float* mask_add,
                          const void* ctx, const void* dctx, int64_t ldc, const float* lse, float* rowdot, void* dq, void* dk,
                          void* dv, int64_t ldd, int B, int nh, int S, uint32_t drop_seed, uint32_t drop_thresh, float drop_scale) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == REALISE_BF16)
    return attn_bwd<bf16_t>(st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, mask_add, (const bf16_t*)ctx,
                            (const bf16_t*)dctx, ldc, lse, rowdot, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, ldd, B, nh, S, drop_seed,
                            drop_thresh, drop_scale);
  if (dtype == REALISE_F32)
    return attn_bwd<float>(st, (const float*)q, (const float*)k, (const float*)v, ldq, mask_add, (const float*)ctx,
                           (const float*)dctx, ldc, lse, rowdot, (float*)dq, (float*)dk, (float*)dv, ldd, B, nh, S, drop_seed,
                           drop_thresh, drop_scale);
  return RL_ERR_ARG;
}
int realise_mask_to_additive(void* stream, const int64_t* masks, float* out, int n) {
  return mask_to_additive((hipStream_t)stream, masks, out, n);
}

int realise_layernorm_fwd_live(void* stream, const void* x, const float* gamma, const float* beta, float eps, void* y, void* xhat, float* rstd,
                               const uint8_t* row_live, int rows, int H) {
  if (row_live == nullptr || (rows % 16) != 0) return RL_ERR_ARG;
  LnFwdArgs<bf16_t> a; a.rows = rows; a.H = H; a.x = (const bf16_t*)x; a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.y = (bf16_t*)y; a.xhat = (bf16_t*)xhat; a.rstd = rstd; a.row_live = row_live;
  return ln_fwd<bf16_t>((hipStream_t)stream, a);
}
int realise_layernorm_fwd(void* stream, int dtype, const void* x, const float* gamma, const float* beta, float eps, void* y,
                          void* xhat, float* rstd, int rows, int H) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == REALISE_BF16) {
    LnFwdArgs<bf16_t> a; a.rows = rows; a.H = H; a.x = (const bf16_t*)x; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.y = (bf16_t*)y; a.xhat = (bf16_t*)xhat; a.rstd = rstd;
    return ln_fwd<bf16_t>(st, a);
  }
  if (dtype == REALISE_F32) {
    LnFwdArgs<float> a; a.rows = rows; a.H = H; a.x = (const float*)x; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.y = (float*)y; a.xhat = (float*)xhat; a.rstd = rstd;
    return ln_fwd<float>(st, a);
  }
  return RL_ERR_ARG;
}
int realise_layernorm_bwd(void* stream, int dtype, const void* dy, const void* xhat, const float* rstd, const float* gamma,
                          void* dx, float* dgamma, float* dbeta, int rows, int H) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == REALISE_BF16) {
    LnBwdArgs<bf16_t> a; a.rows = rows; a.H = H; a.dy = (const bf16_t*)dy; a.xhat = (const bf16_t*)xhat; a.rstd = rstd;
    a.gamma = gamma; a.dx = (bf16_t*)dx; a.dgamma = dgamma; a.dbeta = dbeta;
    return ln_bwd<bf16_t>(st, a);
  }
  if (dtype == REALISE_F32) {
    LnBwdArgs<float> a; a.rows = rows; a.H = H; a.dy = (const float*)dy; a.xhat = (const float*)xhat; a.rstd = rstd;
    a.gamma = gamma; a.dx = (float*)dx; a.dgamma = dgamma; a.dbeta = dbeta;
    return ln_bwd<float>(st, a);
  }
  return RL_ERR_ARG;
}
int realise_layernorm_bwd_ex(void* stream, const void* dy, const void* xhat, const float* rstd, const float* gamma, void* dx, void* dx_drop,
                             uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, float* dgamma, float* dbeta, float* slots, int rows, int H) {
  LnBwdArgs<bf16_t> a; a.rows = rows; a.H = H; a.dy = (const bf16_t*)dy; a.xhat = (const bf16_t*)xhat; a.rstd = rstd;
  a.gamma = gamma; a.dx = (bf16_t*)dx; a.dx_drop = (bf16_t*)dx_drop; a.out_drop.seed = drop_seed; a.out_drop.thresh = drop_thresh;
  a.out_drop.scale = drop_scale; a.dgamma = dgamma; a.dbeta = dbeta; a.slots = slots;
  return ln_bwd<bf16_t>((hipStream_t)stream, a);
}
int realise_layernorm_bwd_live(void* stream, const void* dy, const void* xhat, const float* rstd, const float* gamma, void* dx, void* dx_drop,
                               uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, float* dgamma, float* dbeta, float* slots,
                               const uint8_t* row_live, int rows, int H) {
  LnBwdArgs<bf16_t> a; a.rows = rows; a.H = H; a.dy = (const bf16_t*)dy; a.xhat = (const bf16_t*)xhat; a.rstd = rstd;
  a.gamma = gamma; a.dx = (bf16_t*)dx; a.dx_drop = (bf16_t*)dx_drop; a.out_drop.seed = drop_seed; a.out_drop.thresh = drop_thresh;
  a.out_drop.scale = drop_scale; a.dgamma = dgamma; a.dbeta = dbeta; a.slots = slots; a.row_live = row_live;
  return ln_bwd<bf16_t>((hipStream_t)stream, a);
}
int realise_build_pho(void* stream, const int64_t* src_idx, int T, const int64_t* table, const int32_t* vlens, int V, int Tw,
                      int64_t* pho_idx, int32_t* perm, int32_t* lens_sorted, int32_t* n_alive_dev) {
  if (!src_idx || !table || !vlens || !pho_idx || !perm || !lens_sorted || !n_alive_dev) return RL_ERR_ARG;
  return pho_prepare((hipStream_t)stream, src_idx, T, table, vlens, V, Tw, pho_idx, perm, lens_sorted, n_alive_dev);
}
int realise_argmax(void* stream, int dtype, const void* logits, int64_t ld, int rows, int V, int64_t* ids) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1) return argmax_rows<bf16_t>(st, (const bf16_t*)logits, ld, rows, V, ids);
  if (dtype == 0) return argmax_rows<float>(st, (const float*)logits, ld, rows, V, ids);
  return RL_ERR_ARG;
}
int realise_masked_ce(void* stream, int dtype, const void* logits, int64_t ld, const int64_t* labels, const int64_t* loss_mask,
                      int rows, int V, float* loss_out, float* count_scratch, void* dlogits) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == REALISE_BF16)
    return ce_loss<bf16_t>(st, (const bf16_t*)logits, ld, labels, loss_mask, rows, V, loss_out, count_scratch, (bf16_t*)dlogits);
  if (dtype == REALISE_F32)
    return ce_loss<float>(st, (const float*)logits, ld, labels, loss_mask, rows, V, loss_out, count_scratch, (float*)dlogits);
  return RL_ERR_ARG;
}

// ---- remaining fine-grained operators of the path (per-op parity tests call these) ------------------------------------------------
}  // extern "C"
namespace {
template <typename T> GruStepArgs<T> to_gru(const realise_gru_step* g) {
  GruStepArgs<T> a;
  a.n_alive = g->n_alive; a.H = g->H; a.Tp = g->Tp; a.t = g->t; a.table = g->table; a.pho_idx = g->pho_idx; a.perm = g->perm; a.lens = g->lens;
  a.gh = (const T*)g->gh; a.b_hh = g->b_hh; a.h_prev = (const T*)g->h_prev; a.h_new = (T*)g->h_new; a.rzn = (T*)g->rzn; a.out = (T*)g->out;
  a.dout = (const T*)g->dout; a.dh = (T*)g->dh; a.dgi = (T*)g->dgi; a.dgh = (T*)g->dgh; a.onehot = (T*)g->onehot;
  return a;
}
template <typename T> GateArgs<T> to_gate(const realise_gate* g) {
  GateArgs<T> a;
  a.B = g->B; a.S = g->S; a.H = g->H; a.bert = (const T*)g->bert; a.pho = (const T*)g->pho; a.res = (const T*)g->res; a.masks = g->masks;
  a.W = g->W; a.bias = g->bias; a.mean = g->mean; a.msum = g->msum; a.g = g->g; a.fused = (T*)g->fused; a.dfused = (const T*)g->dfused;
  a.dbert = (T*)g->dbert; a.dpho = (T*)g->dpho; a.dres = (T*)g->dres; a.dz = g->dz; a.dW = g->dW; a.dbias = g->dbias;
  return a;
}
template <typename T>
int bn_fwd_t(hipStream_t st, const T* x, int P, int C, const float* gamma, const float* beta, float eps, float momentum, float* rmean, float* rvar,
             int64_t* nbt, int training, int relu, T* y, float* save_mean, float* save_rstd, float* scratch) {
  float* sums = scratch; float* sq = scratch + C; float* scale = scratch + 2 * C; float* shift = scratch + 3 * C;
  if (training) {
    RL_TRY(fill_f32(st, scratch, 0.0f, 2 * C));
    RL_TRY(col_sum<T>(st, x, P, C, sums));
    RL_TRY(bn_finalize_mean(st, sums, C, P, save_mean));
    RL_TRY(col_sumsq_centered<T>(st, x, P, C, save_mean, sq));
    RL_TRY(bn_finalize_train(st, save_mean, sq, C, P, gamma, beta, eps, momentum, rmean, rvar, save_rstd, scale, shift, nbt));
  } else {
    RL_TRY(bn_finalize_eval(st, C, gamma, beta, eps, rmean, rvar, scale, shift));
  }
  return bn_apply<T>(st, x, scale, shift, (const T*)nullptr, nullptr, nullptr, y, P, C, relu);
}
template <typename T>
int bn_bwd_t(hipStream_t st, const T* dy, const T* relu_src, const T* x, const float* mean, const float* rstd, const float* gamma, int P, int C,
             T* dx, float* dgamma, float* dbeta, float* scratch) {
  RL_TRY(fill_f32(st, scratch, 0.0f, 2 * C));
  RL_TRY(bn_bwd_reduce<T>(st, dy, relu_src, x, mean, rstd, P, C, scratch));
  return bn_bwd_apply<T>(st, dy, relu_src, x, mean, rstd, gamma, scratch, P, C, dx, dgamma, dbeta);
}
}  // namespace
extern "C" {
#define RL_BY_DTYPE(expr_bf16, expr_f32) do { if (dtype == REALISE_BF16) return (expr_bf16); if (dtype == REALISE_F32) return (expr_f32); return RL_ERR_ARG; } while (0)
int realise_gru_step_fwd(void* stream, int dtype, const realise_gru_step* a) {
  if (!a) return RL_ERR_ARG;
  RL_BY_DTYPE(gru_step_fwd<bf16_t>((hipStream_t)stream, to_gru<bf16_t>(a)), gru_step_fwd<float>((hipStream_t)stream, to_gru<float>(a)));
}
int realise_gru_step_bwd(void* stream, int dtype, const realise_gru_step* a) {
  if (!a) return RL_ERR_ARG;
  RL_BY_DTYPE(gru_step_bwd<bf16_t>((hipStream_t)stream, to_gru<bf16_t>(a)), gru_step_bwd<float>((hipStream_t)stream, to_gru<float>(a)));
}
int realise_gate_fwd(void* stream, int dtype, const realise_gate* a) {
  if (!a) return RL_ERR_ARG;
  RL_BY_DTYPE(gate_fwd<bf16_t>((hipStream_t)stream, to_gate<bf16_t>(a)), gate_fwd<float>((hipStream_t)stream, to_gate<float>(a)));
}
int realise_gate_bwd(void* stream, int dtype, const realise_gate* a) {
  if (!a) return RL_ERR_ARG;
  RL_BY_DTYPE(gate_bwd<bf16_t>((hipStream_t)stream, to_gate<bf16_t>(a)), gate_bwd<float>((hipStream_t)stream, to_gate<float>(a)));
}
int realise_batchnorm_fwd(void* stream, int dtype, const void* x, int P, int C, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, int training, int relu, void* y,
                          float* save_mean, float* save_rstd, float* scratch) {
  if (!x || !y || !scratch || P < 1 || C < 4 || (C & 3)) return RL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RL_BY_DTYPE(bn_fwd_t<bf16_t>(st, (const bf16_t*)x, P, C, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, training, relu,
                               (bf16_t*)y, save_mean, save_rstd, scratch),
              bn_fwd_t<float>(st, (const float*)x, P, C, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, training, relu,
                              (float*)y, save_mean, save_rstd, scratch));
}
int realise_batchnorm_bwd(void* stream, int dtype, const void* dy, const void* relu_src, const void* x, const float* save_mean, const float* save_rstd,
                          const float* gamma, int P, int C, void* dx, float* dgamma, float* dbeta, float* scratch) {
  if (!dy || !x || !dx || !scratch || P < 1 || C < 4 || (C & 3)) return RL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RL_BY_DTYPE(bn_bwd_t<bf16_t>(st, (const bf16_t*)dy, (const bf16_t*)relu_src, (const bf16_t*)x, save_mean, save_rstd, gamma, P, C, (bf16_t*)dx, dgamma, dbeta, scratch),
              bn_bwd_t<float>(st, (const float*)dy, (const float*)relu_src, (const float*)x, save_mean, save_rstd, gamma, P, C, (float*)dx, dgamma, dbeta, scratch));
}
// The BatchNorm reductions / maps exactly as the engine's glyph branch calls them (bf16; include/realise_hip_debug.h)
int realise_batchnorm_stats_ex(void* stream, const void* x, int P, int C, int hw, const float* counts, int n_stat, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd,
                               float* scale, float* shift, float* sq_scratch, float* slots) {
  if (!x || !mean || !rstd || !scale || !shift || !sq_scratch || !slots || !running_mean || !running_var || P < 1 || C < 4 || (C & 3) || hw < 1) return RL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RowBound rb; rb.counts = counts; rb.hw = hw; rb.slots = slots;
  const int n = n_stat > 0 ? n_stat : P;
  if (bn_stats_train16(st, (const bf16_t*)x, P, C, n, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift, num_batches_tracked, rb) == RL_OK)
    return RL_OK;
  RL_TRY(col_sum<bf16_t>(st, (const bf16_t*)x, P, C, mean, rb, 1.0f / (float)n));
  RL_TRY(col_sumsq_centered<bf16_t>(st, (const bf16_t*)x, P, C, mean, sq_scratch, rb));
  return bn_finalize_train(st, mean, sq_scratch, C, n, gamma, beta, eps, momentum, running_mean, running_var, rstd, scale, shift, num_batches_tracked);
}
int realise_batchnorm_bwd_ex(void* stream, const void* dy, const void* relu_src, int P, int C, int hw, const float* counts, int n_stat,
                             const void* xa, const float* mean_a, const float* rstd_a, const float* gamma_a, void* dxa, float* dgamma_a, float* dbeta_a,
                             const void* xb, const float* mean_b, const float* rstd_b, const float* gamma_b, void* dxb, float* dgamma_b, float* dbeta_b,
                             float* sums, float* slots) {
  if (!dy || !xa || !dxa || !sums || !slots || P < 1 || C < 4 || (C & 3) || hw < 1) return RL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RowBound rb; rb.counts = counts; rb.hw = hw; rb.slots = slots;
  const bf16_t* g = (const bf16_t*)dy; const bf16_t* o = (const bf16_t*)relu_src;
  const int ns = n_stat > 0 ? n_stat : P;
  if (xb != nullptr) {
    if (bn_bwd_reduce2(st, g, o, (const bf16_t*)xa, mean_a, rstd_a, (const bf16_t*)xb, mean_b, rstd_b, P, C, sums, rb) == RL_OK)
      return bn_bwd_apply2(st, g, o, (const bf16_t*)xa, mean_a, rstd_a, gamma_a, (bf16_t*)dxa, dgamma_a, dbeta_a, (const bf16_t*)xb, mean_b, rstd_b, gamma_b,
                           (bf16_t*)dxb, dgamma_b, dbeta_b, sums, P, C, rb, ns);
    RL_TRY(bn_bwd_reduce<bf16_t>(st, g, o, (const bf16_t*)xb, mean_b, rstd_b, P, C, sums, rb));
    RL_TRY(bn_bwd_apply<bf16_t>(st, g, o, (const bf16_t*)xb, mean_b, rstd_b, gamma_b, sums, P, C, (bf16_t*)dxb, dgamma_b, dbeta_b, rb, ns));
  }
  RL_TRY(bn_bwd_reduce<bf16_t>(st, g, o, (const bf16_t*)xa, mean_a, rstd_a, P, C, sums, rb));
  return bn_bwd_apply<bf16_t>(st, g, o, (const bf16_t*)xa, mean_a, rstd_a, gamma_a, sums, P, C, (bf16_t*)dxa, dgamma_a, dbeta_a, rb, ns);
}
int realise_embedding_bwd(void* stream, int dtype, const void* de, const int64_t* ids, int B, int S, int H, float* word_grad, float* pos_grad,
                          int pos_zero, float* type_grad) {
  hipStream_t st = (hipStream_t)stream;
  RL_BY_DTYPE(embed_bwd<bf16_t>(st, (const bf16_t*)de, ids, B, S, H, word_grad, pos_grad, pos_zero, type_grad),
              embed_bwd<float>(st, (const float*)de, ids, B, S, H, word_grad, pos_grad, pos_zero, type_grad));
}
int realise_glyph_unique(void* stream, const int64_t* ids, int T, int V, int32_t* first_scratch, int32_t* flag_scratch, int64_t* uniq_ids, float* counts,
                         int32_t* inv, int32_t* bounds, int nhw, const int32_t* hw) {
  if (!ids || T < 1 || V < 1 || nhw < 0 || nhw > 8 || (nhw > 0 && !hw)) return RL_ERR_ARG;
  HwList h; h.n = nhw;
  for (int k = 0; k < nhw; ++k) h.v[k] = hw[k];
  return glyph_unique((hipStream_t)stream, ids, T, V, (int*)first_scratch, (int*)flag_scratch, uniq_ids, counts, (int*)inv, (int*)bounds, h);
}
int realise_segment_sum(void* stream, int dtype, const void* x, const int32_t* inv, int T, int C, float* acc, void* out, const int32_t* nuniq_dev) {
  hipStream_t st = (hipStream_t)stream;
  RL_BY_DTYPE(segment_sum<bf16_t>(st, (const bf16_t*)x, (const int*)inv, T, C, acc, (bf16_t*)out, (const int*)nuniq_dev),
              segment_sum<float>(st, (const float*)x, (const int*)inv, T, C, acc, (float*)out, (const int*)nuniq_dev));
}

// ---- layout --------------------------------------------------------------------------------------
int realise_layout_count(const realise_config* cfg) { return cfg ? (int)build_layout(*cfg).tensors.size() : -1; }
int realise_layout_entry(const realise_config* cfg, int index, char* name, int name_cap, int32_t* arena, int64_t* offset,
                         int32_t* ndim, int64_t* shape4) {
  if (!cfg) return RL_ERR_ARG;
  const Layout L = build_layout(*cfg);
  if (index < 0 || index >= (int)L.tensors.size()) return RL_ERR_ARG;
  const TensorInfo& t = L.tensors[index];
  if ((int)t.name.size() + 1 > name_cap) return RL_ERR_ARG;
  strcpy(name, t.name.c_str());
  *arena = t.arena; *offset = t.offset; *ndim = t.ndim;
  for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
  return RL_OK;
}
int64_t realise_arena_elems(const realise_config* cfg, int arena) {
  if (!cfg || arena < 0 || arena >= AR_COUNT) return -1;
  return build_layout(*cfg).arena_elems[arena];
}
int realise_bucket_count(const realise_config* cfg) { return cfg ? (int)build_layout(*cfg).buckets.size() : -1; }
int realise_bucket_bounds(const realise_config* cfg, int bucket, int64_t* begin, int64_t* end) {
  if (!cfg) return RL_ERR_ARG;
  const Layout L = build_layout(*cfg);
  if (bucket < 0 || bucket >= (int)L.buckets.size()) return RL_ERR_ARG;
  *begin = L.buckets[bucket].first; *end = L.buckets[bucket].second;
  return RL_OK;
}

// ---- engine ----------------------------------------------------------------------------------------
realise_engine* realise_engine_create(const realise_config* cfg, float* params, float* grads, float* unused_params, float* frozen,
                                      float* buffers_f32, int64_t* buffers_i64) {
  if (!cfg) return nullptr;
  EngineBase* impl = make_engine(*cfg, params, grads, unused_params, frozen, buffers_f32, buffers_i64);
  if (!impl) return nullptr;
  realise_engine* e = new realise_engine;
  e->impl = impl;
  return e;
}
void realise_engine_destroy(realise_engine* e) { if (e) { delete e->impl; delete e; } }
int64_t realise_engine_shadow_bytes(const realise_engine* e) { return e ? e->impl->shadow_bytes() : -1; }
int64_t realise_engine_workspace_bytes(const realise_engine* e, int B, int S, int Tp) { return e ? e->impl->workspace_bytes(B, S, Tp) : -1; }
int realise_engine_bind(realise_engine* e, void* shadow, void* workspace, int64_t workspace_bytes) {
  return e ? e->impl->bind(shadow, workspace, workspace_bytes) : RL_ERR_ARG;
}
int realise_debug_tn8_supported(int64_t lda, int64_t ldb, int P, int I, int J, int64_t ldo) {
  TnEpi te; te.ldo = ldo;
  return tn8_supported(lda, ldb, P, I, J, te) ? 1 : 0;
}
int realise_debug_tn_list_lds(int64_t entries) { return tn_list_lds_bytes(entries); }
int64_t realise_engine_plan_installs(const realise_engine* e) { return e ? e->impl->plan_install_count() : -1; }
void realise_engine_forget_workspace(realise_engine* e, void* workspace) { if (e) e->impl->forget_workspace(workspace); }
int realise_engine_refresh_shadows(realise_engine* e, void* stream) { return e ? e->impl->refresh_shadows((hipStream_t)stream) : RL_ERR_ARG; }
int realise_engine_refresh_shadows_ex(realise_engine* e, void* stream, int linear_current) {
  return e ? e->impl->refresh_shadows_ex((hipStream_t)stream, linear_current) : RL_ERR_ARG;
}
void realise_engine_invalidate_frozen(realise_engine* e) { if (e) e->impl->invalidate_frozen(); }
void realise_engine_set_grads_fresh(realise_engine* e, int fresh) { if (e) e->impl->set_grads_fresh(fresh); }
void realise_engine_set_id_flag(realise_engine* e, int32_t* flag) { if (e) e->impl->set_id_flag((int*)flag); }
void realise_engine_set_loss_grad(realise_engine* e, const float* grad_dev) { if (e) e->impl->set_loss_grad(grad_dev); }
int realise_engine_forward(realise_engine* e, void* stream, const realise_batch* batch) {
  return (e && batch) ? e->impl->forward((hipStream_t)stream, *batch) : RL_ERR_ARG;
}
int realise_engine_backward(realise_engine* e, void* stream, int first_bucket, int last_bucket) {
  return e ? e->impl->backward((hipStream_t)stream, first_bucket, last_bucket) : RL_ERR_ARG;
}
int realise_engine_backward_signalled(realise_engine* e, void* stream, void* const* bucket_events, int n_events) {
  return e ? e->impl->backward_signalled((hipStream_t)stream, bucket_events, n_events) : RL_ERR_ARG;
}
int realise_engine_glyph_forward(realise_engine* e, void* stream, const int64_t* src_idx, int B, int S, int training, void* res_out) {
  return e ? e->impl->glyph_forward((hipStream_t)stream, src_idx, B, S, training, res_out) : RL_ERR_ARG;
}
int realise_engine_glyph_backward(realise_engine* e, void* stream, const void* d_res) {
  return e ? e->impl->glyph_backward((hipStream_t)stream, d_res) : RL_ERR_ARG;
}
int realise_engine_tap(realise_engine* e, const char* name, void** ptr, int64_t* numel) {
  return (e && name && ptr && numel) ? e->impl->get_tap(name, ptr, numel) : RL_ERR_ARG;
}

// ---- optimizer ---------------------------------------------------------------------------------------
int realise_sumsq(void* stream, const float* g, int64_t n, float* out_accum) { return sumsq_accum((hipStream_t)stream, g, n, out_accum); }
int realise_adamw(void* stream, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, int correct_bias, const float* grad_norm_sq, float max_grad_norm) {
  float bc1 = 1.0f, bc2 = 1.0f;
  if (correct_bias) {
    bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  }
  return adamw_flat((hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_norm_sq, max_grad_norm);
}
int realise_clip_scale(void* stream, float* g, int64_t n, const float* grad_norm_sq, float max_grad_norm) { return clip_scale((hipStream_t)stream, g, n, grad_norm_sq, max_grad_norm); }
static bool make_adamw_groups(const realise_adamw_group* groups, int n_groups, int64_t step, AdamwGroups& gs) {
  if (!groups || n_groups < 1 || n_groups > ADAMW_MAX_GROUPS) return false;
  gs.n = n_groups;
  for (int i = 0; i < n_groups; ++i) {
    const realise_adamw_group& h = groups[i];
    double bc1 = 1.0, bc2 = 1.0;
    if (h.correct_bias) { bc1 = 1.0 - pow((double)h.beta1, (double)step); bc2 = 1.0 - pow((double)h.beta2, (double)step); }
    gs.g[i] = AdamwGroup{h.lr, h.beta1, h.beta2, h.eps, h.weight_decay, (float)(h.lr * sqrt(bc2) / bc1)};
  }
  return true;
}
int realise_adamw_grouped(void* stream, float* p, const float* g, float* m, float* v, int64_t n, const uint8_t* group_of_block64,
                          const realise_adamw_group* groups, int n_groups, int64_t step, const float* grad_norm_sq, float max_grad_norm) {
  AdamwGroups gs;
  if (!make_adamw_groups(groups, n_groups, step, gs)) return RL_ERR_ARG;
  return adamw_grouped((hipStream_t)stream, p, g, m, v, n, group_of_block64, gs, grad_norm_sq, max_grad_norm);
}
int realise_engine_adamw(realise_engine* e, void* stream, float* m, float* v, const uint8_t* group_of_block64,
                         const realise_adamw_group* groups, int n_groups, int64_t step, const float* grad_norm_sq, float max_grad_norm) {
  AdamwGroups gs;
  if (!e || !make_adamw_groups(groups, n_groups, step, gs)) return RL_ERR_ARG;
  return e->impl->adamw_step((hipStream_t)stream, m, v, group_of_block64, gs, grad_norm_sq, max_grad_norm, 0);
}
int realise_engine_adamw_pipelined(realise_engine* e, void* stream, float* m, float* v, const uint8_t* group_of_block64,
                                   const realise_adamw_group* groups, int n_groups, int64_t step, const float* grad_norm_sq, float max_grad_norm) {
  AdamwGroups gs;
  if (!e || !make_adamw_groups(groups, n_groups, step, gs)) return RL_ERR_ARG;
  return e->impl->adamw_step((hipStream_t)stream, m, v, group_of_block64, gs, grad_norm_sq, max_grad_norm, 1);
}
int realise_engine_sync_optimizer(realise_engine* e, void* stream) { return e ? e->impl->sync_optimizer((hipStream_t)stream) : RL_ERR_ARG; }
int realise_profile_enable(int max_launches) { return prof_enable(max_launches); }
void realise_profile_pause(int paused) { prof_pause(paused); }
void realise_profile_mode(int attached) { prof_set_mode(attached); }
void realise_profile_disable(void) { prof_disable(); }
int realise_profile_dump(int kernel_family, int max_records, float* ms_out, double* work_out) { return prof_dump(kernel_family, max_records, ms_out, work_out); }
int realise_profile_dump_ex(int kernel_family, int max_records, float* ms_out, double* work_out, double* work_exec_out) {
  return prof_dump(kernel_family, max_records, ms_out, work_out, work_exec_out);
}
int realise_profile_read_ex(int kernel_family, long long* count, double* total_ms, double* total_work, double* total_work_executed) {
  if (kernel_family < 0 || kernel_family >= PK_COUNT || !count || !total_ms || !total_work || !total_work_executed) return RL_ERR_ARG;
  return prof_read(kernel_family, count, total_ms, total_work, total_work_executed);
}
int realise_profile_read(int kernel_family, long long* count, double* total_ms, double* total_work) {
  if (kernel_family < 0 || kernel_family >= PK_COUNT || !count || !total_ms || !total_work) return RL_ERR_ARG;
  return prof_read(kernel_family, count, total_ms, total_work);
}
int realise_cast_to_f32(void* stream, int dtype, const void* src, float* dst, int64_t n) {
  if (!src || !dst) return RL_ERR_ARG;
  if (dtype == REALISE_BF16) return cast_to_f32<bf16_t>((hipStream_t)stream, (const bf16_t*)src, dst, n);
  if (dtype == REALISE_F32) return cast_to_f32<float>((hipStream_t)stream, (const float*)src, dst, n);
  return RL_ERR_ARG;
}
int realise_fill_f32(void* stream, float* p, float value, int64_t n) { return fill_f32((hipStream_t)stream, p, value, n); }

}  // extern "C"
