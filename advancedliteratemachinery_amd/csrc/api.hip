// Error plumbing, the context object (all mutable library state) and the ABI version of libomp355.
#include <stdarg.h>
#include <stdio.h>

#include <atomic>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "common.h"

// ---- measurement hooks (bench.py roofline legs): hipEvent brackets around eagerly launched kernels of one class ----
struct OmpProfClass {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t used = 0;
  double work = 0.0;   // flops (GEMM / MLP) of the bracketed launches
  double bytes = 0.0;  // their algorithmic HBM bytes
  double roof_s = 0.0; // sum over launches of max(flops / 2.5 PF, bytes / 8 TB/s): the time the launches take on their own rooflines
};

namespace {
thread_local char g_err[512] = "";
thread_local omp_ctx* t_ctx = nullptr;   // this thread's context; nullptr = the process default
thread_local uint64_t t_epoch = 0;       // value of g_epoch when t_ctx was last validated
std::mutex g_prof_mu;                    // the event lists may be appended to from several lane threads of one context
// Registry of live contexts: omp_ctx_destroy on one thread must not leave ANOTHER thread (a pipeline lane worker) with a dangling
// thread-local pointer.  Every destroy bumps g_epoch; a thread whose cached epoch is stale re-validates its pointer against the
// registry (one atomic load on the fast path) and falls back to the default context if its context is gone.
std::mutex g_ctx_mu;
std::set<omp_ctx*> g_live;
std::atomic<uint64_t> g_epoch{1};

omp_ctx& default_ctx() {
  static omp_ctx* c = [] {
    omp_ctx* x = new omp_ctx();
    x->prof = new OmpProfClass[OMP_PROF_NCLASS];
    return x;
  }();
  return *c;
}
}  // namespace

omp_ctx& omp_cur() {
  if (t_ctx != nullptr) {
    const uint64_t e = g_epoch.load(std::memory_order_acquire);
    if (e != t_epoch) {
      std::lock_guard<std::mutex> lk(g_ctx_mu);
      if (g_live.find(t_ctx) == g_live.end()) t_ctx = nullptr;   // destroyed elsewhere: back to the default context
      t_epoch = e;
    }
  }
  return t_ctx != nullptr ? *t_ctx : default_ctx();
}

int omp_device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cus[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

extern "C" int omp_ctx_create(omp_ctx** out) {
  OMP_CHECK_ARG(out != nullptr, "omp_ctx_create: null pointer");
  omp_ctx* c = new omp_ctx();
  c->prof = new OmpProfClass[OMP_PROF_NCLASS];
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_live.insert(c);
  }
  *out = c;
  return OMP_OK;
}

extern "C" int omp_ctx_destroy(omp_ctx* c) {
  if (c == nullptr) return OMP_OK;
  OMP_CHECK_ARG(c != &default_ctx(), "omp_ctx_destroy: the default context is not destroyable");
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (g_live.erase(c) == 0) {
      omp_set_error("omp_ctx_destroy: not a live context handle");
      return OMP_ERR_INVALID;
    }
    g_epoch.fetch_add(1, std::memory_order_acq_rel);   // other threads drop their pointer on their next call
  }
  if (t_ctx == c) t_ctx = nullptr;
  for (int i = 0; i < OMP_MAX_GRAPH_SLOTS; ++i) {
    if (c->slots[i].exec) (void)hipGraphExecDestroy(c->slots[i].exec);
    if (c->slots[i].graph) (void)hipGraphDestroy(c->slots[i].graph);
    if (c->slots[i].exec_n) (void)hipGraphExecDestroy(c->slots[i].exec_n);
    if (c->slots[i].graph_n) (void)hipGraphDestroy(c->slots[i].graph_n);
  }
  for (int k = 0; k < OMP_PROF_NCLASS; ++k)
    for (auto& e : c->prof[k].ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  delete[] c->prof;
  delete c;
  return OMP_OK;
}

extern "C" int omp_ctx_make_current(omp_ctx* c) {
  if (c == nullptr || c == &default_ctx()) { t_ctx = nullptr; return OMP_OK; }
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  if (g_live.find(c) == g_live.end()) {   // a stale handle (destroyed, or never a context) must not become current
    omp_set_error("omp_ctx_make_current: not a live context handle");
    return OMP_ERR_INVALID;
  }
  t_ctx = c;
  t_epoch = g_epoch.load(std::memory_order_acquire);
  return OMP_OK;
}

extern "C" omp_ctx* omp_ctx_current(void) { return &omp_cur(); }

bool omp_prof_active(int cls) { return (omp_cur().prof_mask >> cls) & 1; }

int omp_prof_begin(int cls, hipStream_t st, double work, double bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  OmpProfClass& c = omp_cur().prof[cls];
  if (c.used == c.ev.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    c.ev.emplace_back(a, b);
  }
  c.work += work;
  c.bytes += bytes;
  { const double tm = work / 2.5e15, tb = bytes / 8.0e12; c.roof_s += tm > tb ? tm : tb; }
  (void)hipEventRecord(c.ev[c.used].first, st);
  return (int)c.used++;
}

void omp_prof_end(int cls, int slot, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  OmpProfClass& c = omp_cur().prof[cls];
  if (slot >= 0 && (size_t)slot < c.ev.size()) (void)hipEventRecord(c.ev[slot].second, st);
}

extern "C" int omp_prof_enable(int mask) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  omp_ctx& x = omp_cur();
  x.prof_mask = mask & ((1 << OMP_PROF_NCLASS) - 1);
  for (int c = 0; c < OMP_PROF_NCLASS; ++c) { x.prof[c].used = 0; x.prof[c].work = 0.0; x.prof[c].bytes = 0.0; x.prof[c].roof_s = 0.0; }
  return OMP_OK;
}

extern "C" int omp_prof_read_class(int cls, double* total_ms, int64_t* count, double* work) {
  OMP_CHECK_ARG(cls >= 0 && cls < OMP_PROF_NCLASS, "omp_prof_read_class: bad class %d", cls);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  OmpProfClass& c = omp_cur().prof[cls];
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

extern "C" int omp_prof_read_roofline(int cls, double* bytes, double* roofline_seconds) {
  OMP_CHECK_ARG(cls >= 0 && cls < OMP_PROF_NCLASS, "omp_prof_read_roofline: bad class %d", cls);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const OmpProfClass& c = omp_cur().prof[cls];
  if (bytes) *bytes = c.bytes;
  if (roofline_seconds) *roofline_seconds = c.roof_s;
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

// ---- HIP streams restricted to a subset of the compute units -----------------------------------------------------
// The hot path alternates matrix-core-bound phases (Swin encoder) and HBM-bound phases (the decoders' cross-attention
// streams 8.4 MB of K / V^T per image, layer and step).  Two engine calls in flight overlap them only if the HBM-bound
// kernels do not take every CU: their streams can be created on a CU subset here (engine/pipeline.py).
namespace {
__global__ void where_kernel(int32_t* out) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);    // XCC_ID
    const unsigned hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);     // HW_ID
    out[blockIdx.x * 2] = (int32_t)xcc;
    out[blockIdx.x * 2 + 1] = (int32_t)hw;
  }
  // stay resident long enough for the grid to spread over every CU the stream may use
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 200000ull) {}
}
}  // namespace

extern "C" int omp_stream_create_cu_mask(const uint32_t* mask, int n_words, omp_stream_t* out) {
  OMP_CHECK_ARG(mask && out && n_words > 0, "omp_stream_create_cu_mask: bad arguments");
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask);
  if (e != hipSuccess) { omp_set_error("omp_stream_create_cu_mask: %s", hipGetErrorString(e)); return OMP_ERR_LAUNCH; }
  *out = (omp_stream_t)st;
  return OMP_OK;
}

extern "C" int omp_stream_destroy(omp_stream_t s) {
  if (s != nullptr && hipStreamDestroy((hipStream_t)s) != hipSuccess) { omp_set_error("omp_stream_destroy failed"); return OMP_ERR_LAUNCH; }
  return OMP_OK;
}

extern "C" int omp_debug_where(int32_t* out, int n_workgroups, omp_stream_t s) {
  OMP_CHECK_ARG(out && n_workgroups > 0, "omp_debug_where: bad arguments");
  hipLaunchKernelGGL(where_kernel, dim3(n_workgroups), dim3(64), 0, (hipStream_t)s, out);
  OMP_CHECK_LAUNCH("omp_debug_where");
  return OMP_OK;
}
