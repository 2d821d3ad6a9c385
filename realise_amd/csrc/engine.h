// Engine interface shared by engine.hip (implementation) and capi.hip (C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/realise_hip.h"

namespace rl {
struct AdamwGroups;
void set_wgrad_overlap(int on);
void set_wgrad_group(int on);     // the four wgrad GEMMs of a transformer layer as one grouped launch (default on)
void set_dgrad_parity(int on);    // stride-2 conv data gradients by input-pixel parity classes (default on)
void set_fwd_order(int o);        // enqueue order of the forward branches: 0 bert first, 1 bert last
void set_skip_dead(int on);        // backward skips padding rows (default 1)
void set_cls_compact(int on);      // classifier backward over the loss rows only (default 1)
void set_ln_fuse(int on);          // dense + dropout + residual + LayerNorm as one launch (default 0: see engine.hip)
void set_gru_fuse(int on);         // GRU time step as one launch: recurrent GEMM + gate math (default 1)
void set_glyph_fuse(int on);       // K7: glyph lookup fused into block 1's forward conv loaders (default 1)
void set_opt_pipe(int on);         // realise_engine_adamw_pipelined as such (1, default) or as the plain sweep on the caller's stream (0)
void set_bn_fold(int on);          // K9 (evaluation): BatchNorm on running statistics applied in the convolutions' epilogues (default 1)
void set_streamk(int v);           // 1: layer GEMMs on the stream-K 256 x 192 kernel (gemm_nt8s.hip; measured slower: default 0)
void set_streamk_min(int n);       // the least K-tiles per workgroup of a launch that select it (default 10)
void set_live_rows(int on);        // bf16 training steps: layer GEMMs / attention forward over the live 16-row blocks only (default 1)
void set_cls_splitk(int n);        // K-ranges of the classifier's data gradient (default 3, 0 / 1: one launch over the whole K)
void set_stream_priority(int which, int pri);   // 0 pinyin branch, 1 glyph branch, 2 weight-gradient stream; -1 high, 0 default, +1 low (read at stream creation)
void set_branch_overlap(int on);  // bert | pho | glyph branches on three streams inside forward / whole-pass backward (default on)   // weight-gradient GEMMs of the BERT layers on an engine-owned side stream (default off)

struct EngineBase {
  virtual ~EngineBase() {}
  virtual int64_t shadow_bytes() const = 0;
  virtual int64_t workspace_bytes(int B, int S, int Tp) = 0;
  virtual int bind(void* shadow, void* workspace, int64_t bytes) = 0;
  virtual void forget_workspace(void* workspace) = 0;
  virtual int64_t plan_install_count() const = 0;
  virtual int refresh_shadows(hipStream_t st) = 0;
  virtual int refresh_shadows_ex(hipStream_t st, int skip_linear) = 0;
  virtual int adamw_step(hipStream_t st, float* m, float* v, const uint8_t* group_of_block, const struct AdamwGroups& gs, const float* norm_sq,
                         float max_norm, int pipelined) = 0;
  virtual int sync_optimizer(hipStream_t st) = 0;
  virtual void invalidate_frozen() = 0;
  virtual void set_id_flag(int* flag) = 0;
  virtual void set_grads_fresh(int fresh) = 0;
  virtual void set_loss_grad(const float* grad_dev) = 0;
  virtual int forward(hipStream_t st, const realise_batch& b) = 0;
  virtual int backward(hipStream_t st, int first, int last) = 0;
  virtual int backward_signalled(hipStream_t st, void* const* bucket_events, int n_events) = 0;
  virtual int glyph_forward(hipStream_t st, const int64_t* ids, int B, int S, int training, void* res_out) = 0;
  virtual int glyph_backward(hipStream_t st, const void* d_res) = 0;
  virtual int get_tap(const char* name, void** ptr, int64_t* numel) = 0;
};
EngineBase* make_engine(const realise_config& c, float* p, float* g, float* pu, float* fz, float* bf, int64_t* bi);
}  // namespace rl
