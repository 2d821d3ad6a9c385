#include "prof.h"
#include <map>
#include <vector>

namespace rl {
namespace {
struct Rec { int kid; double work; size_t e0; const int* rows_dev; double per_row; int quantum, rows_max; };
bool g_on = false;
std::vector<hipEvent_t> g_ev;
std::vector<Rec> g_recs;
size_t g_used = 0;
bool g_open = false;
bool g_paused = false;
int g_mode = 1;
bool g_taken = false;
}  // namespace

void prof_set_mode(int attached) { g_mode = attached ? 1 : 0; }
bool prof_take(hipEvent_t* e0, hipEvent_t* e1) {
  if (!g_open || g_mode != 1 || g_taken) return false;
  *e0 = g_ev[g_used]; *e1 = g_ev[g_used + 1];
  g_taken = true;
  return true;
}

int prof_enable(int max_launches) {
  prof_disable();
  g_ev.resize((size_t)max_launches * 2);
  for (auto& e : g_ev)
    if (hipEventCreate(&e) != hipSuccess) return 2;
  g_recs.clear();
  g_recs.reserve(max_launches);
  g_used = 0;
  g_on = true;
  g_paused = false;
  return 0;
}
void prof_disable() {
  for (auto& e : g_ev) (void)hipEventDestroy(e);
  g_ev.clear();
  g_recs.clear();
  g_used = 0;
  g_on = false;
  g_open = false;
}
void prof_pause(int paused) { g_paused = paused != 0; }
void prof_begin(hipStream_t st, int kid, double work) {
  g_open = false;
  if (!g_on || g_paused || g_used + 2 > g_ev.size()) return;
  if (g_mode == 0) (void)hipEventRecord(g_ev[g_used], st);
  g_recs.push_back({kid, work, g_used, nullptr, 0.0, 1, 0});
  g_open = true;
  g_taken = false;
}
void prof_end(hipStream_t st) {
  if (!g_open) return;
  g_open = false;
  if (g_mode == 0) (void)hipEventRecord(g_ev[g_used + 1], st);
  else if (!g_taken) { g_recs.pop_back(); return; }     // no launch inside the scope picked the events up: nothing was timed
  g_used += 2;
}
void prof_set_exec(const int* rows_dev, double work_per_row, int quantum, int rows_max) {
  if (!g_open || g_recs.empty() || rows_dev == nullptr) return;
  Rec& r = g_recs.back();
  r.rows_dev = rows_dev; r.per_row = work_per_row; r.quantum = quantum > 0 ? quantum : 1; r.rows_max = rows_max;
}
namespace {
// executed work of a record: the device counter is read once per distinct address (after the caller synchronised on the last event)
double exec_work(const Rec& r, std::map<const int*, int>& cache) {
  if (r.rows_dev == nullptr) return r.work;
  auto it = cache.find(r.rows_dev);
  if (it == cache.end()) {
    int v = 0;
    if (hipMemcpy(&v, r.rows_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) v = r.rows_max;
    it = cache.emplace(r.rows_dev, v).first;
  }
  long long rows = ((long long)it->second + r.quantum - 1) / r.quantum * r.quantum;
  if (rows > r.rows_max) rows = r.rows_max;
  if (rows < 0) rows = 0;
  return r.per_row * (double)rows;
}
}  // namespace
int prof_read(int kid, long long* count, double* total_ms, double* total_work, double* total_work_exec) {
  *count = 0; *total_ms = 0.0; *total_work = 0.0;
  if (total_work_exec) *total_work_exec = 0.0;
  if (!g_on) return 1;
  if (g_used >= 2) (void)hipEventSynchronize(g_ev[g_used - 1]);
  std::map<const int*, int> cache;
  for (const Rec& r : g_recs) {
    if (r.kid != kid || r.e0 + 2 > g_used) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_ev[r.e0], g_ev[r.e0 + 1]) != hipSuccess) continue;
    *count += 1; *total_ms += ms; *total_work += r.work;
    if (total_work_exec) *total_work_exec += exec_work(r, cache);
  }
  return 0;
}
int prof_dump(int kid, int max, float* ms_out, double* work_out, double* work_exec_out) {
  if (!g_on) return 0;
  if (g_used >= 2) (void)hipEventSynchronize(g_ev[g_used - 1]);
  std::map<const int*, int> cache;
  int n = 0;
  for (const Rec& r : g_recs) {
    if (r.kid != kid || r.e0 + 2 > g_used || n >= max) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_ev[r.e0], g_ev[r.e0 + 1]) != hipSuccess) continue;
    ms_out[n] = ms; work_out[n] = r.work;
    if (work_exec_out) work_exec_out[n] = exec_work(r, cache);
    ++n;
  }
  return n;
}
}  // namespace rl
