// Fused attention launchers (attention.hip).
#pragma once
#include "common.h"

namespace rl {
void set_attn_probe(int mode);
// q/k/v: token-major [B*S][ldq] matrices (head h at columns h*64..h*64+63); mask_add: [B][S] fp32
// additive mask ((1-m) * -10000); ctx: [B*S][ldc]; lse: [B][nh][S] fp32 row log-sum-exp.
template <typename T>
int attn_fwd(hipStream_t st, const T* q, const T* k, const T* v, int64_t ldq, const float* mask_add, T* ctx, int64_t ldc,
             float* lse, int B, int nh, int S, uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, const int* rlen = nullptr);
// rlen (optional, B ints; live-row training steps): rows >= rlen[b] of sentence b are padding - masked as keys (their probabilities
// are exact zeros either way), never read as queries.  With it the K / V rows beyond rlen[b] are not read (they may hold stale
// values) and the 32-query blocks that start at or beyond it are not computed: their ctx / lse rows are left as they are.
// rowdot: [B][nh][S] fp32 scratch; dq/dk/dv: token-major [B*S][ldd], fully overwritten for every head.
template <typename T>
int attn_bwd(hipStream_t st, const T* q, const T* k, const T* v, int64_t ldq, const float* mask_add, const T* ctx,
             const T* dctx, int64_t ldc, const float* lse, float* rowdot, T* dq, T* dk, T* dv, int64_t ldd,
             int B, int nh, int S, uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, const int* rlen = nullptr);
// rlen (optional, B ints): rows >= rlen[b] of sentence b are padding whose dctx rows count as exact zeros and whose keys are masked -
// the kernels do not visit them (their dq / dk / dv rows are stored as zeros), and nothing read from such a row (q, k, v, ctx, dctx,
// lse) enters a result: after a live-row forward those rows hold stale values
}  // namespace rl
