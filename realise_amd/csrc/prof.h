// Optional per-launch timing of the MFMA kernel families with HIP events recorded on the launch
// stream (bench.py's `roofline` object).  Disabled by default: zero cost when off.
#pragma once
#include <hip/hip_runtime.h>

namespace rl {
enum ProfKernel { PK_GEMM_NT = 0, PK_CONV_NT = 1, PK_GEMM_TN = 2, PK_CONV_TN = 3, PK_ATTN_FWD = 4, PK_ATTN_BWD = 5, PK_COUNT = 6 };
void prof_begin(hipStream_t st, int kid, double work);
void prof_end(hipStream_t st);
int prof_enable(int max_launches);
void prof_disable();
void prof_pause(int paused);      // keep the collected records, stop / resume bracketing launches (sampled steps)
int prof_read(int kid, long long* count, double* total_ms, double* total_work);
int prof_dump(int kid, int max, float* ms_out, double* work_out);   // per-launch records of one family, in launch order
struct ProfScope {
  hipStream_t st;
  ProfScope(hipStream_t s, int kid, double work) : st(s) { prof_begin(s, kid, work); }
  ~ProfScope() { prof_end(st); }
};
}  // namespace rl
