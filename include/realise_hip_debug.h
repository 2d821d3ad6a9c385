/* realise_hip_debug.h - diagnostics of librealise_hip.so: A/B knobs, probe modes, per-launch timing hooks.
 *
 * NOT part of the operator boundary (include/realise_hip.h): nothing here changes results unless its comment says so, the
 * defaults are the production configuration, and a drop-in user never needs to call any of it.  tools/*.cpp, bench.py
 * (roofline sampling) and the A/B tests do.
 */
#ifndef REALISE_HIP_DEBUG_H
#define REALISE_HIP_DEBUG_H
#include "realise_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* 1: ds_read_b64_tr_b16 transposed operand reads in the TN kernel (bf16), 0: 16-bit LDS gathers. */
void realise_set_tn_transpose_read(int enable);
/* A/B knob: allow the 128x96 NT tile chosen by the chip-balance heuristic (default 1) */
void realise_set_nt_allow_n96(int on);
/* Diagnostics only (tools/nt_probe.cpp; results are WRONG when != 0): 1 every tile fetches tile 0's operands (cache-hot),
 * 2 no operand fetches, 3 no MFMA work - separates the memory, issue and compute shares of the NT kernel's time. */
void realise_set_nt_probe(int mode);
/* Diagnostics: force an experimental NT tile shape for dense bf16 GEMMs (0 = production heuristic). */
void realise_set_nt_variant(int v);
/* step engine: key 0 = enqueue order of the three forward branches (0: bert stack first, 1: the shorter pinyin / glyph branches first);
 * keys 1 / 2 / 3 = priority class of the pinyin-branch / glyph-branch / weight-gradient stream (-1 highest, 0 device default, +1 lowest),
 * read when the engine creates the stream, i.e. to be set before the first forward; key 4 = classifier backward over the rows that
 * enter the loss only (1, default) or over all rows (0); key 5 = the backward skips the rows of padding tokens, whose gradient rows are
 * exact zeros (LayerNorm backward rows, blocks of the weight-gradient reductions; 1 default: live 16-row blocks packed four to a
 * reduction tile in bf16, 2: whole 64-row tiles only - bit-identical to 0 -, 0 off); key 6 = K-ranges of the split-K classifier data
 * gradient (bf16; default 3, 0 / 1 = one launch over the whole K = 21184); key 7 = grouped weight gradients of the transformer layers on
 * the 4-wave 128 x 128 kernel, two workgroups per CU (0, default) or on the 8-wave 256 x 128 kernel, one tile per CU (1: measured, slower); key 8 = BertSelfOutput /
 * BertOutput as GEMM + LayerNorm launches (0, default) or as one launch (dense + bias + dropout + residual + LayerNorm: 1 - correct, measured
 * 0.3 ms/step slower; 2 = the same without the cross-tile hand-off, diagnostics only: wrong statistics); key 9 = a GRU
 * time step as one launch (recurrent GEMM with the gate math in its epilogue: 1, default) or as GEMM + gate kernel (0); key 10 = bf16
 * training steps run the layer GEMMs of the transformer stacks (forward and data gradients) and the attention forward over the live
 * 16-row blocks of the padded batch only (1, default: loss, live-row logits and gradients bit-identical to the dense step; the rows
 * behind a sentence's last attended / loss position keep stale activations) or over all rows (0); key 11 = the layer GEMMs with enough
 * K-tiles to share out (qkv, FFN-up / -down and their data gradients) on the stream-K 256 x 192 kernel (1) or on the 128 x 192 two-per-CU
 * kernels (0, default: measured faster on every layer shape, and their live-row results are bit-identical to the dense step's - a tile that
 * stream-K cuts sums its K range in two or three chains); key 12 = the least K-tiles per workgroup of a launch that select the stream-K
 * kernel under key 11 (default 10; 0: every shape it supports; keys 11 / 12 act in the probe build only - round 6); key 13 = K7, the
 * glyph lookup fused into the loaders of block 1's forward convolutions (1, default: no gathered image batch in the forward; a
 * training step gathers it at the head of the backward for the two weight-gradient reductions) or gather_images + dense loaders (0);
 * key 14 = K9 in evaluation mode: BatchNorm on its running statistics applied in the glyph convolutions' epilogues (1, default: per
 * block three launches, shortcut first, its normalised output added in the second convolution's epilogue) or as separate scale / shift
 * and apply kernels over the raw convolution outputs (0, the round-5 form); key 15 = realise_engine_adamw_pipelined runs pipelined (1,
 * default) or as the plain sweep on the caller's stream (0) */
void realise_set_engine(int key, int value);
/* realise_gemm_tn_grouped over a list of live reduction blocks, as the engine's backward calls it: live[k] (device, ascending) = index
 * of the k-th block of `list_rows` rows that holds anything but exact zeros in the A operands, *n_live (device) = how many; the other
 * blocks are not read.  list_rows = 64 (bf16) / 32 (fp32): whole reduction tiles; 16 (bf16): four live blocks form a reduction tile.
 * overwrite != 0: out = result instead of out += result. */
/* realise_gemm_nt bounded by a DEVICE-side row count, as the GRU steps of a device-built batch launch it: rows at or beyond
 * *rows_dev contribute zeros (an accumulating epilogue leaves them unchanged, a storing one writes bias-only rows in the last live
 * tile and nothing beyond it). */
int realise_gemm_nt_rows(void* stream, int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                         const realise_epilogue* ep, const int* rows_dev);
/* realise_gemm_nt (bf16, K % 64 == 0, M % 16 == 0) over a device-side LIST of 16-row blocks, as a training step launches the layer GEMMs of
 * a padded batch: live_list[k] = index of the k-th listed block (ascending), *live_count = how many.  Rows of listed blocks are computed
 * exactly as the dense launch computes them (bit-identical) and read / written at their original positions; rows of the other blocks
 * are neither read nor written. */
int realise_gemm_nt_live(void* stream, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                         const realise_epilogue* ep, const int* live_list, const int* live_count);
/* Round 6: the same over a device-side list of ROWS (row-granular packing, what a training step launches by default): row_list[k] =
 * index of the k-th listed row (ascending), *row_count = how many; tile t of the 128 x 192 kernel works on rows row_list[128 t ..
 * 128 t + 127], so no tile row is spent on rows that merely complete a 16-row block.  Listed rows: bit-identical to the dense launch. */
int realise_gemm_nt_live_rows(void* stream, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                              const realise_epilogue* ep, const int* row_list, const int* row_count);
/* The stream-K form of the layer GEMMs (bf16, K % 128 == 0; gemm_nt8s.hip): ONE round of 256 workgroups over 256 x 192 tiles, the
 * (tile, K-tile) space cut into 256 equal ranges, tiles that a cut splits folded in-kernel in workgroup order.  live_list / live_count
 * as realise_gemm_nt_live, or both NULL (all rows).  part: exchange buffer of 256 * 24 * 512 * 16 bytes; flags: 256 x 64 ints (a flag every 256 bytes), zero before the
 * first launch; tag: never 0, different from the previous launch's on the same buffers; timeout (nullable): set to 1 if a workgroup gave
 * up waiting for a partial tile.  Reproducible bit for bit for given (shape, live count); differs in the last bits from
 * realise_gemm_nt / realise_gemm_nt_live where a tile's K range is cut (two or three accumulation chains instead of one). */
int realise_gemm_nt_streamk(void* stream, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                            const realise_epilogue* ep, const int* live_list, const int* live_count, float* part, int* flags, int tag,
                            int* timeout);
/* Split-K form of the 8-wave NT GEMM as the classifier's data gradient uses it (bf16, K % 64 == 0): slab[s][m][n] (fp32, row pitch N,
 * plane pitch slab_stride floats) = A[m, K-range s] . B[n, K-range s]^T; rows at or beyond *m_dev (device, nullable) are not computed. */
int realise_gemm_nt_splitk(void* stream, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int nsplit,
                           float* slab, int64_t slab_stride, const int* m_dev);
int realise_gemm_tn_grouped_live(void* stream, int dtype, int n, const realise_tn_problem* problems, int P, const int* live,
                                 const int* n_live, int list_rows, int overwrite);
/* LayerNorm backward exactly as the engine calls it (bf16): optional second output dx_drop = dx * dropout mask, per-workgroup
 * [dgamma | dbeta] records in `slots` (8 MiB scratch) folded in a fixed order.  tools/ln_probe.py times it. */
int realise_layernorm_bwd_ex(void* stream, const void* dy, const void* xhat, const float* rstd, const float* gamma, void* dx, void* dx_drop,
                             uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, float* dgamma, float* dbeta, float* slots, int rows, int H);
/* realise_layernorm_fwd (bf16, rows % 16 == 0) as a live-row training step launches it: row_live = one byte per row (0 = padding).  A
 * 16-row block without a live row is not visited - its y / xhat / rstd rows keep what they held -, a block with one is computed whole. */
int realise_layernorm_fwd_live(void* stream, const void* x, const float* gamma, const float* beta, float eps, void* y, void* xhat, float* rstd,
                               const uint8_t* row_live, int rows, int H);
/* the same with the row-liveness bytes of a padded batch (row_live[r] == 0: dy[r] is an exact zero - the row is not read, its outputs are zeros) */
int realise_layernorm_bwd_live(void* stream, const void* dy, const void* xhat, const float* rstd, const float* gamma, void* dx, void* dx_drop,
                               uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, float* dgamma, float* dbeta, float* slots,
                               const uint8_t* row_live, int rows, int H);
/* LayerNorm kernels: key 0 = bf16 fast path (half a wave per row, 16-byte accesses; default 1), key 1 = workgroups of its backward (default 512);
 * BatchNorm kernels: key 2 = bf16 fast paths (1, default: 16-byte accesses, paired bn2 + shortcut backward, one-pass training
 * statistics; 3: the same with two-pass statistics; 0: generic kernels), key 3 = row chunks of their
 * column reductions (default 1024) ; key 4 = bf16 masked cross-entropy with the logits row read once into registers
 * (1, default) or the generic three-walk kernel (0); key 5 = round-5 LayerNorm kernels (1, default: row reductions through DPP,
 * backward with four rows of a wave in flight, no barrier before the first row, one-barrier epilogue, 256 workgroups; 0: the round-4
 * kernels) - key 1 sets the workgroups of whichever backward is active */
void realise_set_ln(int key, int value);
/* BatchNorm training statistics and backward exactly as the engine's glyph branch runs them (bf16, NHWC viewed as [P, C];
 * char_cnn.py:15-32): per-row-chunk partial records in `slots` (1 MiB scratch) folded in a fixed order; `counts` (nullable) =
 * multiplicity of each image of `hw` pixels (glyph dedup), n_stat = the true sample count of the statistics (0: P).
 * stats: mean / rstd / scale / shift [C], running buffers updated (unbiased variance), num_batches_tracked += 1; sq_scratch: C floats.
 * bwd: dx = gamma rstd (g - w sum(g) / n - xhat w sum(g xhat) / n) with g = dy masked by relu_src > 0; dgamma / dbeta are ADDED to;
 * xb != NULL: a second normalisation sharing dy and the mask (bn2 + shortcut BN of a BasicBlock; one pass reads dy and the mask for
 * both); sums: 4C floats of scratch.  tools/bn_probe.py times them; tests/test_ops_gpu.py checks fast against generic paths. */
int realise_batchnorm_stats_ex(void* stream, const void* x, int P, int C, int hw, const float* counts, int n_stat, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd,
                               float* scale, float* shift, float* sq_scratch, float* slots);
int realise_batchnorm_bwd_ex(void* stream, const void* dy, const void* relu_src, int P, int C, int hw, const float* counts, int n_stat,
                             const void* xa, const float* mean_a, const float* rstd_a, const float* gamma_a, void* dxa, float* dgamma_a, float* dbeta_a,
                             const void* xb, const float* mean_b, const float* rstd_b, const float* gamma_b, void* dxb, float* dgamma_b, float* dbeta_b,
                             float* sums, float* slots);
/* persistent NT kernel (gemm_nt8p.hip): key 0 = tile walk (1 default: every XCD owns a band of tile rows, 0: chunked tile ids),
 * key 1 = workgroups launched (default 256 = one per CU); key 2 = one-round outputs (at most one 128 x 192 tile per CU) on the
 * three-stage one-per-CU shape (1) or the two-per-CU shape (0, default: faster inside a step) of gemm_nt8.hip; key 3 = column groups of
 * the XCD split of the live-row layer GEMMs (0 default: from the shape - 2 for the wide K = 768 outputs, 1 otherwise; 1 / 2 / 4 / 8 forced);
 * key 4 = the 8-wave kernels add alpha / bias to the accumulators before the epilogue's transposes (1, default: one bias fetch per wave and no
 * per-item waits in the bias-only epilogues) or per item (0, the round-4 form) - same bits */
void realise_set_nt8p(int key, int value);
void realise_set_nt_group_m(int g);        /* tile order of the 8-wave NT GEMM: 0 row-major, g: g tile rows per column step (L2 blocking) */
/* Diagnostics for the TN kernel: 2 no operand fetches, 3 no MFMA work, 4 skip the slab fold pass. */
void realise_set_tn_probe(int mode);
/* Diagnostics: force the number of reduction splits of the TN kernel (0 = heuristic). */
void realise_set_tn_split(int n);
/* weight-gradient kernel selection: 0 production (8-wave 256x128-tile kernel for the big dense bf16 shapes), 9 force the 4-wave kernel */
void realise_set_tn_variant(int v);
void realise_set_conv_c64(int on);         /* 1 (default): the 64->64 channel 3x3 conv on 16x16 maps (forward, input and weight gradient) runs the LDS-resident kernels */
void realise_set_tn_group_ring(int on);    /* grouped weight gradients: 0 (default) two LDS stages of 64-row K-tiles, 1 four stages of 32 rows (measured 11 % slower) */
/* Diagnostics for the attention forward kernel: 1 stop after operand staging, 2 skip the softmax (results WRONG). */
void realise_set_attn_probe(int mode);
/* A/B knob: 1 (default) the NT epilogue goes through a per-wave LDS transpose so every global access is 16 B per lane over
 * whole 128-byte row segments; 0 stores the MFMA fragments directly (8 B per lane). Identical results. */
void realise_set_nt_wide_epilogue(int on);
/* A/B knob: 1 (default) run the glyph ResNet once per distinct token id with multiplicity-weighted BatchNorm;
 * 0 run it densely over all B*S tokens like the reference (identical results) */
void realise_set_glyph_dedup(int on);
/* A/B knob: 1 (default) = the four weight-gradient GEMMs of each BERT layer run on an engine-owned side stream, overlapped with the
 * data-gradient chain on the caller's stream (joined before realise_engine_backward returns); 0 everything in order on the
 * caller's stream.  Identical results; +2 % throughput measured, per-kernel timings become overlap-dependent. */
void realise_set_wgrad_overlap(int on);
/* 1 (default): the three branches of SpellBertPho2ResArch3.forward that are independent between the inputs and the gate
 * (src/models.py:816 bert | :818-827 pinyin GRU + pho_model | :829-838 glyph ResNet), and their backward passes behind the
 * gate, run on three HIP streams (the caller's + two engine-owned), forked / joined with events inside the engine call; the
 * caller's stream owns every result when the call returns.  0: everything in order on the caller's stream.  Identical results. */
void realise_set_branch_overlap(int on);
void realise_set_dgrad_parity(int on);     /* 1 (default): stride-2 conv data gradients as four input-pixel parity-class GEMMs (no stride-miss taps) */
void realise_set_wgrad_group(int on);      /* 1 (default): the four weight gradients of a transformer layer as one grouped launch */

/* Per-launch timing of the MFMA kernel families with HIP events on the launch stream (bench.py roofline).
 * family: 0 gemm_nt, 1 conv_nt (implicit im2col), 2 gemm_tn, 3 conv_tn, 4 attention fwd, 5 attention bwd.
 * total_work = algorithmic FLOPs (2*M*N*K per GEMM launch). */
int realise_profile_enable(int max_launches);
/* 1 (default): the event pair is attached to the kernel dispatch (its own begin / end timestamps, what rocprofv3's kernel
 * trace reports); 0: two event markers recorded around the launch (adds the marker packets' processing to every sample) */
void realise_profile_mode(int attached);
void realise_profile_disable(void);
/* 1: stop bracketing launches but keep what was recorded, 0: resume (an event pair costs ~4 us of stream time per launch, so
 * bench.py samples every 10th timed step instead of all of them) */
void realise_profile_pause(int paused);
int realise_profile_read(int kernel_family, long long* count, double* total_ms, double* total_work);
/* per-launch records of one family in launch order (host arrays of max_records entries); returns the number written */
int realise_profile_dump(int kernel_family, int max_records, float* ms_out, double* work_out);
/* the same with the EXECUTED work next to the booked (nominal) one: launches bounded by a device-side row count (the loss rows of the
 * classifier's gradients, the live blocks of a padded batch in the weight-gradient reductions, the GRU's alive sequences) book
 * 2.M.N.K for the nominal M and execute fewer rows; the counters are read from the device when the records are read (valid while
 * the batch shape does not change between the sampled steps and the read) */
int realise_profile_read_ex(int kernel_family, long long* count, double* total_ms, double* total_work, double* total_work_executed);
int realise_profile_dump_ex(int kernel_family, int max_records, float* ms_out, double* work_out, double* work_executed_out);

/* Host-side support predicates of the weight-gradient kernels, exported so that a CPU test can pin them (a comment
 * once swallowed half of the first one's conditions).  realise_debug_tn8_supported: 1 when the 8-wave TN kernel takes the problem
 * (plain epilogue, no live-block list), 0 when the caller must fall back to the 4-wave kernel.  realise_debug_tn_list_lds: bytes of
 * LDS a listed TN launch reserves for `entries` live blocks, -1 when the list does not fit (the engine then runs dense reductions). */
int realise_debug_tn8_supported(int64_t lda, int64_t ldb, int P, int I, int J, int64_t ldo);
int realise_debug_tn_list_lds(int64_t entries);

/* how many workspace plans the engine has installed (= whole-workspace zero fills) since it was created: a loop that alternates
 * batch shapes over per-shape workspace buffers (realise_engine_forget_workspace in realise_hip.h) must count one per shape */
int64_t realise_engine_plan_installs(const realise_engine* e);


#ifdef __cplusplus
}
#endif
#endif /* REALISE_HIP_DEBUG_H */
