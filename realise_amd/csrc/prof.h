// Optional per-launch timing of the MFMA kernel families with HIP events recorded on the launch
// stream (bench.py's `roofline` object).  Disabled by default: zero cost when off.
//
// Two timing modes (prof_set_mode): 1 (default) = the two events are ATTACHED to the kernel dispatch (hipExtLaunchKernelGGL start /
// stop events: the dispatch packet's own begin / end timestamps, the quantity rocprofv3's kernel trace reports); 0 = the events are
// recorded around the launch as separate stream markers (round 1/2 form; adds the two marker packets' processing, ~1-4 us, to
// every bracketed kernel).  Launch sites inside a ProfScope go through RL_LAUNCH so that mode 1 can hand them the events.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace rl {
enum ProfKernel { PK_GEMM_NT = 0, PK_CONV_NT = 1, PK_GEMM_TN = 2, PK_CONV_TN = 3, PK_ATTN_FWD = 4, PK_ATTN_BWD = 5, PK_COUNT = 6 };
void prof_begin(hipStream_t st, int kid, double work);
// Executed work of the scope that is open (no-op otherwise): launches bounded by a DEVICE-side row count (glyph dedup, loss rows,
// live blocks of padded batches) book `work` for the nominal row count; the executed figure is work_per_row x min(rows_max,
// *rows_dev rounded up to `quantum`), the counter being read when the records are (valid while the batch shape does not change
// between the sampled launch and the read - bench.py's fixed synthetic batch).
void prof_set_exec(const int* rows_dev, double work_per_row, int quantum, int rows_max);
void prof_end(hipStream_t st);
int prof_enable(int max_launches);
void prof_set_mode(int attached);  // 1: events attached to the dispatch (default), 0: event markers around the launch
bool prof_take(hipEvent_t* e0, hipEvent_t* e1);   // mode 1, inside an open scope: the scope's event pair (at most once per scope)
void prof_disable();
void prof_pause(int paused);      // keep the collected records, stop / resume bracketing launches (sampled steps)
int prof_read(int kid, long long* count, double* total_ms, double* total_work, double* total_work_exec = nullptr);
int prof_dump(int kid, int max, float* ms_out, double* work_out, double* work_exec_out = nullptr);   // per-launch records of one family, in launch order
struct ProfScope {
  hipStream_t st;
  ProfScope(hipStream_t s, int kid, double work) : st(s) { prof_begin(s, kid, work); }
  ~ProfScope() { prof_end(st); }
};
}  // namespace rl

#define RL_LAUNCH(kern, grid, block, lds, st, ...)                                                          \
  do {                                                                                                      \
    hipEvent_t rl_e0_, rl_e1_;                                                                              \
    if (rl::prof_take(&rl_e0_, &rl_e1_)) hipExtLaunchKernelGGL(kern, grid, block, lds, st, rl_e0_, rl_e1_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                       \
  } while (0)
