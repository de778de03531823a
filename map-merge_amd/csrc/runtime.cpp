// runtime.cpp -- the host-only runtime of the library: the reference-counted device-memory pool, a context's waits
// (stream_wait: poll + nap instead of spinning), its pinned arena, the deferred device-side error checks, the kernel
// profiling scopes and the bookkeeping of the chained scans.  No kernel lives here, so this file -- with capi.cpp,
// host_pipeline.cpp and linalg.cpp -- is what the host sanitizer build compiles for real (tests/host_san: a fake HIP runtime
// and a fake device layer stand in for everything below it; ThreadSanitizer / AddressSanitizer + UBSan).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execinfo.h>
#include <map>
#include <memory>
#include <mutex>
#include <sys/prctl.h>
#include <thread>

#include "types.hpp"

namespace mm3d {

// ---------------------------------------------------------------- pool / context
size_t Pool::size_class(size_t bytes)
{
  if (bytes <= 256) return 256;
  size_t p = 256;
  while (p * 2 <= bytes) p *= 2;        // p <= bytes < 2p
  if (bytes == p) return p;
  size_t step = p / 4;
  return p + ((bytes - p + step - 1) / step) * step;
}

void *Pool::alloc(size_t bytes)
{
  std::lock_guard<std::mutex> lk(mu_);
  size_t cls = size_class(bytes);
  auto it = free_.find(cls);
  void *p = nullptr;
  if (it != free_.end() && !it->second.empty()) {
    p = it->second.back();
    it->second.pop_back();
  } else {
    hipError_t e = hipMalloc(&p, cls);
    if (e != hipSuccess) {
      trim_locked();
      e = hipMalloc(&p, cls);
      if (e != hipSuccess) throw Error(MM3D_ENOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    }
  }
  live_[p] = cls;
  return p;
}

void Pool::release(void *p)
{
  std::lock_guard<std::mutex> lk(mu_);
  auto it = live_.find(p);
  if (it == live_.end()) return;
  free_[it->second].push_back(p);
  live_.erase(it);
}

void Pool::trim()
{
  std::lock_guard<std::mutex> lk(mu_);
  trim_locked();
}

void Pool::trim_locked()
{
  for (auto &kv : free_)
    for (void *p : kv.second) (void)hipFree(p);
  free_.clear();
}

hipError_t stream_wait(hipStream_t stream)
{
  static const bool spin_only = [] { const char *e = getenv("MM3D_WAIT"); return e && std::string(e) == "spin"; }();
  if (spin_only) return hipStreamSynchronize(stream);
  static const long spin_us = [] { const char *e = getenv("MM3D_WAIT_SPIN_US"); return e ? atol(e) : 200L; }();
  // A nap of 10 us lasts 60 with Linux's default timer slack of 50 us, so a thread that starts napping asks for 1 us --
  // and gives the caller's thread its own slack back before it returns: worker 0 is the application's thread, and a
  // library has no business changing how that thread's later sleeps, selects and futex waits are rounded.
  const auto t0 = std::chrono::steady_clock::now();
  long nap_us = 5;
  long old_slack = -1;                                           // >= 0: changed, to be restored
  hipError_t e;
  for (;;) {
    e = hipStreamQuery(stream);
    if (e != hipErrorNotReady) break;
    const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    if (waited < spin_us) continue;                              // most readbacks of a size are over by now
    if (old_slack < 0) {
      old_slack = (long)prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL);
      if (old_slack < 0) old_slack = 50000;                      // the kernel's default, should the query fail
      (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
    }
    std::this_thread::sleep_for(std::chrono::microseconds(nap_us));
    if (nap_us < 60) nap_us += 5;                                // naps of 5 ... 60 us
  }
  if (old_slack >= 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old_slack, 0UL, 0UL, 0UL);
  return e;
}

// MM3D_WAIT_TRACE=1: every host wait is counted under the address it was called from; the table goes to stderr when the
// process ends (scripts/one_map_latency.py reads it: which function makes a map wait, how often, how long).
namespace {
struct WaitSites {
  std::mutex mu;
  std::map<void *, std::pair<long long, long long>> sites;       // caller -> (waits, ns)
  ~WaitSites()
  {
    for (auto &kv : sites) {
      void *a = kv.first;
      char **sym = backtrace_symbols(&a, 1);
      fprintf(stderr, "wait site %-90s %8lld waits %10.3f ms\n", sym ? sym[0] : "?", kv.second.first, kv.second.second * 1e-6);
      free(sym);
    }
  }
};
WaitSites *wait_sites()
{
  static std::unique_ptr<WaitSites> w(getenv("MM3D_WAIT_TRACE") ? new WaitSites() : nullptr);
  return w.get();
}
}  // namespace

void Context::sync()
{
  const auto t0 = std::chrono::steady_clock::now();
  MM3D_HIP(stream_wait(stream));
  ++waits;
  if (WaitSites *w = wait_sites()) {
    void *bt[3] = {nullptr, nullptr, nullptr};
    const int got = backtrace(bt, 3);
    const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> lk(w->mu);
    auto &e = w->sites[got >= 2 ? bt[1] : nullptr];
    e.first += 1; e.second += ns;
  }
  wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  if (!deferred.empty()) {
    std::vector<Deferred> d;
    d.swap(deferred);
    for (const Deferred &e : d)
      if (*e.flag) throw Error(e.status, e.what);
  }
}

void *Context::pin(size_t bytes)
{
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes > pinned_bytes) {
    if (pinned) { sync(); (void)hipHostFree(pinned); pinned = nullptr; }
    pinned_bytes = bytes * 2 < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes * 2;
    MM3D_HIP(hipHostMalloc(&pinned, pinned_bytes));
    pinned_off = 0;
  }
  if (pinned_off + bytes > pinned_bytes) {
    sync();                       // every copy that used the arena has drained
    pinned_off = 0;
  }
  void *p = (char *)pinned + pinned_off;
  pinned_off += bytes;
  return p;
}

int Context::prof_slot(const char *name)
{
  auto it = prof_index.find(name);
  if (it != prof_index.end()) return it->second;
  int s = (int)prof.size();
  prof_index[name] = s;
  prof_names.emplace_back(name);
  prof.emplace_back();
  return s;
}

void Context::prof_resolve()
{
  if (pending.empty()) return;
  MM3D_HIP(stream_wait(stream));
  for (auto &p : pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) prof[p.slot].ms += ms;
    event_pool.push_back(p.a);
    event_pool.push_back(p.b);
  }
  pending.clear();
}

KernelScope::KernelScope(Context *ctx, const char *name, double bytes, bool attached_) : c(ctx), attached(attached_)
{
  if (!c->prof_on) return;
  slot = c->prof_slot(name);
  c->prof[slot].launches++;
  c->prof[slot].bytes += bytes;
  auto get = [&]() {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e;
    MM3D_HIP(hipEventCreate(&e));
    return e;
  };
  a = get(); b = get();
  if (!attached) (void)hipEventRecord(a, c->stream);
}

KernelScope::~KernelScope()
{
  if (slot < 0) return;
  if (!attached) (void)hipEventRecord(b, c->stream);
  c->pending.push_back({slot, a, b});
  if (c->pending.size() > 8192) {
    try { c->prof_resolve(); } catch (...) {}
  }
}

// Bookkeeping of one chained-scan launch over n elements on this context: status words large enough, the next epoch, the
// tickets this launch will take (k_scan_int here, k_scan_fused in scan_fused.hpp).
ScanLaunchState scan_prepare(Context *c, size_t n)
{
  const size_t tiles = (n + kScanTile - 1) / kScanTile;
  if (tiles > c->scan_tiles_cap) {
    c->sync();                                   // earlier scans on this stream are done with the old buffers
    if (!c->scan_ticket) {
      MM3D_HIP(hipMalloc((void **)&c->scan_ticket, sizeof(unsigned)));
      MM3D_HIP(hipMemsetAsync(c->scan_ticket, 0, sizeof(unsigned), c->stream));
      c->scan_tickets_taken = 0;
    }
    // the new buffer first: if the allocation fails the old one (and its capacity) stay valid
    const size_t cap = tiles * 2 < 4096 ? 4096 : tiles * 2;
    unsigned long long *fresh = nullptr;
    MM3D_HIP(hipMalloc((void **)&fresh, cap * sizeof(unsigned long long)));
    if (c->scan_status) (void)hipFree(c->scan_status);
    c->scan_status = fresh;
    c->scan_tiles_cap = cap;
    MM3D_HIP(hipMemsetAsync(c->scan_status, 0, c->scan_tiles_cap * sizeof(unsigned long long), c->stream));
    c->scan_epoch = 0;
  }
  c->scan_epoch = (c->scan_epoch + 1) & 0x3fffffffu;
  if (c->scan_epoch == 0) {                      // the epoch wrapped: clear once, start over
    MM3D_HIP(hipMemsetAsync(c->scan_status, 0, c->scan_tiles_cap * sizeof(unsigned long long), c->stream));
    c->scan_epoch = 1;
  }
  ScanLaunchState st{c->scan_status, c->scan_ticket, c->scan_tickets_taken, c->scan_epoch, (unsigned)tiles};
  c->scan_tickets_taken += (unsigned)tiles;      // unsigned wrap-around matches the device counter's
  return st;
}

}  // namespace mm3d
