// common.hpp -- shared host-side plumbing of libmm3d (context, device buffers, launch/profiling).
//
// MI355X-native (gfx950) only.  Device memory comes from a per-context size-class pool on top of
// hipMalloc so the per-pair loop never hits the allocator; every kernel of a context runs on the
// context's own stream.
#pragma once

#include <cstdlib>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mm3d.h"

namespace mm3d {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};

#define MM3D_HIP(expr)                                                                            \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      throw ::mm3d::Error(MM3D_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + \
                                            __FILE__ + ":" + std::to_string(__LINE__) + ")");     \
  } while (0)

#define MM3D_REQUIRE(cond, msg)                                  \
  do {                                                           \
    if (!(cond)) throw ::mm3d::Error(MM3D_EINVAL, (msg));        \
  } while (0)

// ---- glibc rand() replay (TYPE_3 additive feedback; SAC-IA's getRandomIndex) -----------------
struct GlibcRand {
  uint32_t ring[31];
  int f = 3, b = 0;
  GlibcRand() { seed(1); }
  void seed(unsigned s)
  {
    if (s == 0) s = 1;
    int32_t r[31];
    r[0] = (int32_t)s;
    for (int i = 1; i < 31; ++i) {
      long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
      long word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      r[i] = (int32_t)word;
    }
    for (int i = 0; i < 31; ++i) ring[i] = (uint32_t)r[i];
    f = 3; b = 0;
    for (int i = 0; i < 310; ++i) (void)next();
  }
  int next()
  {
    ring[f] += ring[b];
    uint32_t res = ring[f] >> 1;
    f = (f + 1) % 31;
    b = (b + 1) % 31;
    return (int)res;
  }
};

// ---- boost::mt19937 (pcl::SampleConsensusModel::rnd) -----------------------------------------
struct Mt19937 {
  uint32_t s[624];
  int pos = 624;
  explicit Mt19937(uint32_t seed)
  {
    s[0] = seed;
    for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
  }
  uint32_t next()
  {
    if (pos >= 624) {
      for (int k = 0; k < 624; ++k) {
        uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
        s[k] = s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      pos = 0;
    }
    uint32_t y = s[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
};

// ---- profiling ----------------------------------------------------------------------------------
struct ProfEntry {
  double ms = 0.0;
  uint64_t launches = 0;
  double bytes = 0.0;
};

struct Context;

// ---- device memory pool -------------------------------------------------------------------------
// Thread-safe (its own mutex), and shared: every buffer holds a reference to the pool it came from, so a
// buffer may be released from another context's thread, or after its context is gone (an object that
// outlives the mm3d_ctx that made it, a worker context removed by mm3d_set_streams).
class Pool {
 public:
  void *alloc(size_t bytes);
  void release(void *p);
  void trim();
  ~Pool() { trim(); }

 private:
  static size_t size_class(size_t bytes);
  void trim_locked();
  std::mutex mu_;
  std::unordered_map<size_t, std::vector<void *>> free_;
  std::unordered_map<void *, size_t> live_;
};

// Waits for a stream without burning a core: hipStreamSynchronize spins for as long as the wait lasts (measured:
// CPU time inside the waits = wall time inside them, whatever hipDeviceScheduleBlockingSync / blocking events say
// with this runtime), and a library that keeps sixteen streams busy then holds sixteen cores.  This polls
// hipStreamQuery: a short spin for the waits that are over in microseconds, then naps that grow from 5 to 60 us (the caller's timer slack is restored on return).
// (Measured and dropped: letting the stream write a sequence number to pinned memory behind its work,
// hipStreamWriteValue32, and polling that word -- the extra packet costs more than the runtime call it saves:
// 650 against 700 map-pairs/s.  MM3D_WAIT=spin restores hipStreamSynchronize: 720 map-pairs/s on ten busy cores.)
hipError_t stream_wait(hipStream_t stream);

struct Context {
  int device = 0;
  hipStream_t stream = nullptr;
  std::shared_ptr<Pool> pool = std::make_shared<Pool>();
  std::string err;
  std::mutex mu;
  GlibcRand rnd;
  int last_icp_iterations = 0, last_icp_converged = 0;
  bool debug = false;            // mm3d_set_debug: collect counters that cost a host sync
  long long knn_fallback_rows = 0, knn_rows = 0;
  long long waits = 0, wait_ns = 0;      // host waits for the stream (sync()): how many, how long (mm3d_debug_waits)
  // pinned host arena for small asynchronous H2D / D2H copies: pin() bumps through it, so regions
  // handed out earlier stay untouched while their copies are in flight; the stream is drained
  // before the arena wraps around or is replaced
  void *pinned = nullptr;
  size_t pinned_bytes = 0, pinned_off = 0;
  // the chained scans (runtime.cpp::scan_prepare; k_scan_int in grid.hip, k_scan_fused in scan_fused.hpp): per-tile status
  // words and the ticket counter; the words carry the launch's epoch, so nothing is cleared between launches
  unsigned long long *scan_status = nullptr;
  unsigned *scan_ticket = nullptr;
  size_t scan_tiles_cap = 0;
  unsigned scan_epoch = 0, scan_tickets_taken = 0;
  // profiling
  bool prof_on = false;
  struct Pending { int slot; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
  std::vector<std::string> prof_names;
  std::vector<ProfEntry> prof;
  std::unordered_map<std::string, int> prof_index;

  void *pin(size_t bytes);
  // sync(): wait for the stream; also the moment at which device-side error flags recorded with check_later() are
  // looked at (throws Error).
  // settle(): a wait whose only purpose is that OTHER contexts may see what was just built (a cache published under
  // its lock) or that buffers go back to the pool.  A driver that keeps its objects private to this context until it
  // has drained the stream itself (estimate_maps_streams, mm3d_shard_begin: a map is published after a full sync())
  // sets private_objects, and these waits are skipped: the pool belongs to this context and hands memory out in this
  // stream's order, so whoever gets a released block next is enqueued behind its last user.
  bool private_objects = false;
  struct Deferred { const int *flag; int status; const char *what; };
  std::vector<Deferred> deferred;
  void sync();
  void settle() { if (!private_objects) sync(); }
  // flag: pinned host memory a D2H copy of a device error word has been enqueued into
  void check_later(const int *flag, int status, const char *what)
  {
    deferred.push_back(Deferred{flag, status, what});
    if (!private_objects) sync();
  }
  int prof_slot(const char *name);
  // a launch whose algorithmic bytes were still on the device when it was enqueued (MM3D_LAUNCH with 0) gets them here
  void prof_add_bytes(const char *name, double bytes) { if (prof_on) prof[prof_slot(name)].bytes += bytes; }
  void prof_resolve();
};

// A raw device pointer with DevBuf's accessor, for a region carved out of a larger buffer (several small arrays that are
// zeroed by one fill dispatch live in one DevBuf): code written against `.get()` works on either.
template <typename T>
struct DevPtr {
  T *p = nullptr;
  T *get() const { return p; }
};

template <typename T>
class DevBuf {
 public:
  DevBuf() = default;
  DevBuf(Context *c, size_t n) : pool_(c->pool), n_(n) { p_ = n ? (T *)pool_->alloc(n * sizeof(T)) : nullptr; }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : pool_(std::move(o.pool_)), p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept
  {
    if (this != &o) { reset(); pool_ = std::move(o.pool_); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
    return *this;
  }
  ~DevBuf() { reset(); }
  void reset()
  {
    if (p_) pool_->release(p_);
    p_ = nullptr; n_ = 0;
  }
  T *get() const { return p_; }
  size_t size() const { return n_; }

 private:
  std::shared_ptr<Pool> pool_;     // keeps the pool alive for as long as the buffer
  T *p_ = nullptr;
  size_t n_ = 0;
};

// RAII kernel timer (only when profiling).  attached = true: the two events ride on the kernel's own
// dispatch packet (hipExtLaunchKernelGGL start / stop events, see MM3D_LAUNCH) -- no extra packets in
// the queue; attached = false: events recorded on the stream around whatever the scope encloses
// (library calls such as rocPRIM's that launch several kernels).
struct KernelScope {
  Context *c;
  int slot = -1;
  bool attached = false;
  hipEvent_t a = nullptr, b = nullptr;
  KernelScope(Context *ctx, const char *name, double bytes, bool attached = false);
  ~KernelScope();
};

#define MM3D_LAUNCH(ctx, name, bytes, kernel, grid, block, shmem, ...)                                            \
  do {                                                                                                            \
    ::mm3d::KernelScope ks_((ctx), (name), (double)(bytes), true);                                                \
    if (ks_.a)                                                                                                    \
      hipExtLaunchKernelGGL(kernel, (grid), (block), (shmem), (ctx)->stream, ks_.a, ks_.b, 0, __VA_ARGS__);       \
    else                                                                                                          \
      hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (ctx)->stream, __VA_ARGS__);                           \
    MM3D_HIP(hipGetLastError());                                                                                  \
  } while (0)

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace mm3d
