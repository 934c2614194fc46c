// Error plumbing + ABI version for libomp355.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace {
thread_local char g_err[512] = "";

// ---- measurement hooks (bench.py roofline legs): hipEvent brackets around eagerly launched kernels of one class ----
struct ProfClass {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t used = 0;
  double work = 0.0;   // flops (GEMM / MLP) or algorithmic bytes (cross-attention) of the bracketed launches
};
ProfClass g_prof[OMP_PROF_NCLASS];
int g_prof_mask = 0;
std::mutex g_prof_mu;
}

bool omp_prof_active(int cls) { return (g_prof_mask >> cls) & 1; }

int omp_prof_begin(int cls, hipStream_t st, double work) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfClass& c = g_prof[cls];
  if (c.used == c.ev.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    c.ev.emplace_back(a, b);
  }
  c.work += work;
  (void)hipEventRecord(c.ev[c.used].first, st);
  return (int)c.used++;
}

void omp_prof_end(int cls, int slot, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (slot >= 0 && (size_t)slot < g_prof[cls].ev.size()) (void)hipEventRecord(g_prof[cls].ev[slot].second, st);
}

extern "C" int omp_prof_enable(int mask) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_mask = mask & ((1 << OMP_PROF_NCLASS) - 1);
  for (int c = 0; c < OMP_PROF_NCLASS; ++c) { g_prof[c].used = 0; g_prof[c].work = 0.0; }
  return OMP_OK;
}

extern "C" int omp_prof_read_class(int cls, double* total_ms, int64_t* count, double* work) {
  OMP_CHECK_ARG(cls >= 0 && cls < OMP_PROF_NCLASS, "omp_prof_read_class: bad class %d", cls);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfClass& c = g_prof[cls];
  double tot = 0.0;
  for (size_t i = 0; i < c.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(c.ev[i].second) != hipSuccess || hipEventElapsedTime(&ms, c.ev[i].first, c.ev[i].second) != hipSuccess) {
      omp_set_error("omp_prof_read_class: event query failed");
      return OMP_ERR_LAUNCH;
    }
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (count) *count = (int64_t)c.used;
  if (work) *work = c.work;
  return OMP_OK;
}

extern "C" int omp_prof_read(double* total_ms, int64_t* count) { return omp_prof_read_class(OMP_PROF_CROSS, total_ms, count, nullptr); }

void omp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* omp_last_error(void) { return g_err; }
extern "C" int omp_abi_version(void) { return OMP_ABI_VERSION; }
