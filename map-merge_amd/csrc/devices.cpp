// devices.cpp -- one process, several GPUs: what the devices of an mm3d_create_devices context exchange.
//
// The reference's caller is ONE process (R/src/map_merge_node.cpp:133-153 calls estimateMapsTransforms from a timer callback;
// R/src/map_merge_tool.cpp:37-38 from main), so the N-GPU form of the path has to live behind that one call:
// capi.cpp::estimate_maps_devices runs the mm3d_shard_* scheme with one host thread + stream set per device, and this file
// holds the two exchanges between the devices:
//   * the maps' bundles (filtered cloud, keypoints, descriptors, and the source-side search structures built on them) are
//     PULLED by every device that does not own the map with hipMemcpyPeerAsync -- point-to-point over xGMI, each device
//     reading from up to seven peers at once on its own streams; no collective, no staging through the host;
//   * the 104-byte pair records are all-gathered with RCCL (ncclAllGather inside one group, one communicator per device
//     from ncclCommInitAll; librccl bound on first use, see rccl() below): the one collective north_star names ("only a final RCCL gather of the pairwise
//     Eigen::Matrix4f over xGMI before the host-side pose-graph solve").  Latency-bound (120 records = 12.5 KB).
// Host code only: it also compiles, unchanged, into the host sanitizer build (tests/host_san: fake HIP runtime + fake RCCL).
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <set>

#include "types.hpp"

namespace mm3d {

// librccl is bound on first use, not at load time: the library is 570 MB of device code, and a process that works on one GPU
// (every context made by mm3d_create) should neither map it nor run its constructors -- on a cold machine that alone can take
// longer than the whole job.  mm3d_create_devices resolves the six entry points it needs: from what the process has already
// loaded (an application that links RCCL itself, PyTorch's bundled copy, the fake of tests/host_san), else from librccl.so.1.
namespace {
struct Rccl {
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;                                   // why it could not be bound (empty: bound)
};
const Rccl &rccl()
{
  static const Rccl R = [] {
    Rccl r;
    void *h = nullptr;
    if (!dlsym(RTLD_DEFAULT, "ncclCommInitAll")) {
      // MM3D_RCCL_LIB: another file name for the library (a private build; the tests' "no RCCL on this box" case)
      const char *override_name = getenv("MM3D_RCCL_LIB");
      std::string tried;
      for (const char *name : {override_name, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (!name || !*name) continue;
        // RTLD_LOCAL: a process that loads a second RCCL later (a Python test that imports torch, which ships its own) must not
        // have that one's references bound to this one's symbols -- "double free or corruption" at exit, measured
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        // (dlerror() hands its message out ONCE and clears it: one call per failure, every name's reason kept)
        const char *e = dlerror();
        tried += std::string(tried.empty() ? "" : "; ") + name + ": " + (e ? e : "?");
        if (override_name && name == override_name) break;    // an explicit name is not a hint: nothing else is tried
      }
      if (!h) { r.error = "librccl could not be loaded (" + tried + ")"; return r; }
    }
    auto sym = [&](const char *n) { void *p = h ? dlsym(h, n) : dlsym(RTLD_DEFAULT, n); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n; return p; };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return r;
  }();
  return R;
}
}  // namespace

struct DeviceSet {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;       // one per list entry; empty for the duplicate-device test hook
  bool broken = false;                 // a collective failed half-way: the communicators are not used again
};

#define MM3D_NCCL(expr)                                                                                                \
  do {                                                                                                                 \
    ncclResult_t r_ = (expr);                                                                                          \
    if (r_ != ncclSuccess)                                                                                             \
      throw ::mm3d::Error(MM3D_EDEVICE, std::string(#expr) + ": " + rccl().GetErrorString(r_) + " (" + __FILE__ + ":" + \
                                            std::to_string(__LINE__) + ")");                                           \
  } while (0)

DeviceSet *device_set_create(const int *devices, int n)
{
  std::unique_ptr<DeviceSet> ds(new DeviceSet());
  ds->devices.assign(devices, devices + n);
  const bool distinct = std::set<int>(ds->devices.begin(), ds->devices.end()).size() == ds->devices.size();
  if (!distinct) {
    // TEST HOOK: two "devices" that are the same GPU exercise the whole multi-device driver -- threads, ownership, peer
    // copies, record packing -- on a one-GPU box; RCCL refuses a device twice in one communicator (ncclInvalidUsage), so the
    // records of such a set are gathered by the same packing code through host memory.  Never silently: the list must be
    // asked for with MM3D_DEVICES_ALLOW_DUPLICATES=1.
    const char *e = getenv("MM3D_DEVICES_ALLOW_DUPLICATES");
    if (!(e && atoi(e))) throw Error(MM3D_EINVAL, "mm3d_create_devices: a device is listed twice");
    return ds.release();
  }
  // peer access both ways between every two devices: the bundle pulls then go GPU to GPU over xGMI (without it the runtime
  // stages a peer copy through host memory).  "Already enabled" is fine; no peer access at all (no link) is not an error
  // either -- the copies still work, slower.
  for (int a = 0; a < n; ++a) {
    if (hipSetDevice(devices[a]) != hipSuccess) throw Error(MM3D_EDEVICE, "mm3d_create_devices: hipSetDevice failed");
    for (int b = 0; b < n; ++b) {
      if (a == b) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw Error(MM3D_EDEVICE, "mm3d_create_devices: hipDeviceEnablePeerAccess failed");
        (void)hipGetLastError();
      }
    }
  }
  if (!rccl().error.empty()) throw Error(MM3D_EDEVICE, "mm3d_create_devices: " + rccl().error);
  ds->comms.resize((size_t)n);
  MM3D_NCCL(rccl().CommInitAll(ds->comms.data(), n, devices));
  (void)hipSetDevice(devices[0]);
  return ds.release();
}

void device_set_destroy(DeviceSet *ds)
{
  if (!ds) return;
  for (ncclComm_t c : ds->comms) (void)rccl().CommDestroy(c);
  delete ds;
}

bool device_set_has_comms(const DeviceSet *ds) { return ds && !ds->comms.empty(); }

template <class T>
static DevBuf<T> pull(Context *c, const T *src, size_t n, int src_device)
{
  DevBuf<T> d(c, n);
  if (n) MM3D_HIP(hipMemcpyPeerAsync(d.get(), c->device, src, src_device, n * sizeof(T), c->stream));
  return d;
}

mm3d_cloud *cloud_clone_from_peer(Context *c, const mm3d_cloud *src_, int src_device)
{
  auto *src = const_cast<mm3d_cloud *>(src_);
  std::lock_guard<std::recursive_mutex> lk(src->cache_mu);
  std::unique_ptr<mm3d_cloud> cl(cloud_from_device(c, pull(c, (const float4 *)src->pts.get(), src->n, src_device), src->n));
  // what the owner already knows or has built travels with the points instead of being recomputed here: the bounding box
  // (no k_bbox launch and wait), the Hilbert query copy with its keys and work items (the source role of ICP / score /
  // SAC-IA scoring: five launches and a wait per cloud otherwise), the keypoints' host copy (the rand() replay reads it)
  if (src->have_bbox) {
    cl->have_bbox = true;
    cl->n_finite = src->n_finite;
    for (int a = 0; a < 3; ++a) { cl->bmin[a] = src->bmin[a]; cl->bmax[a] = src->bmax[a]; }
  }
  if (src->hil_pts.get() && src->have_bbox) {
    cl->hil_pts = pull(c, (const float4 *)src->hil_pts.get(), src->hil_pts.size(), src_device);
    if (src->hil_keys.get()) cl->hil_keys = pull(c, (const uint32_t *)src->hil_keys.get(), src->hil_keys.size(), src_device);
    // (the items' buffer is sized by its bound; only the first n_wave_items entries mean anything)
    DevBuf<int2> items(c, (size_t)std::max(src->n_wave_items, 1));
    if (src->n_wave_items)
      MM3D_HIP(hipMemcpyPeerAsync(items.get(), c->device, src->wave_items.get(), src_device, (size_t)src->n_wave_items * sizeof(int2), c->stream));
    cl->wave_items = std::move(items);
    cl->n_wave_items = src->n_wave_items;
  }
  if (src->host.size() == src->n) cl->host = src->host;
  return cl.release();
}

mm3d_desc *desc_clone_from_peer(Context *c, const mm3d_desc *src, int src_device)
{
  std::unique_ptr<mm3d_desc> d(new mm3d_desc());
  d->n = src->n; d->dim = src->dim; d->type = src->type;
  d->data = pull(c, (const float *)src->data.get(), src->n * (size_t)src->dim, src_device);
  return d.release();
}

double gather_pair_records(DeviceSet *ds, const std::vector<mm3d_ctx *> &roots, const std::vector<std::vector<mm3d_pair_result>> &send,
                           size_t slots, std::vector<mm3d_pair_result> &out)
{
  const size_t D = roots.size();
  const size_t rec = sizeof(mm3d_pair_result), bytes = slots * rec;
  out.assign(D * slots, mm3d_pair_result{});
  if (slots == 0) return 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  if (ds->comms.empty()) {
    // the duplicate-device test hook (device_set_create): same layout, through host memory
    for (size_t d = 0; d < D; ++d) std::memcpy(&out[d * slots], send[d].data(), std::min(send[d].size(), slots) * rec);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  std::vector<DevBuf<unsigned char>> sbuf(D), rbuf(D);
  for (size_t d = 0; d < D; ++d) {
    mm3d_ctx *c = roots[d];
    MM3D_HIP(hipSetDevice(c->device));
    sbuf[d] = DevBuf<unsigned char>(c, bytes);
    rbuf[d] = DevBuf<unsigned char>(c, bytes * D);
    unsigned char *h = (unsigned char *)c->pin(bytes);
    std::memset(h, 0, bytes);
    std::memcpy(h, send[d].data(), std::min(send[d].size(), slots) * rec);
    MM3D_HIP(hipMemcpyAsync(sbuf[d].get(), h, bytes, hipMemcpyHostToDevice, c->stream));
  }
  // one thread, one group: the standard single-process form (every rank's call is enqueued on its own device's stream)
  if (ds->broken) throw Error(MM3D_EDEVICE, "gather_pair_records: an earlier collective on these communicators failed; make a new context");
  MM3D_NCCL(rccl().GroupStart());
  {
    // nothing may leave this block with the group open: the group depth is the THREAD's, and a group left at depth 1 queues
    // every later collective of the thread without ever launching it (the next call would wait for ever in sync()).  A failed
    // enqueue closes the group, restores the first device and marks the communicators unusable before it is reported.
    std::string failed;
    for (size_t d = 0; d < D && failed.empty(); ++d) {
      if (hipSetDevice(roots[d]->device) != hipSuccess) { failed = "hipSetDevice failed inside the RCCL group"; break; }
      const ncclResult_t r = rccl().AllGather(sbuf[d].get(), rbuf[d].get(), bytes, ncclChar, ds->comms[d], roots[d]->stream);
      if (r != ncclSuccess) failed = std::string("ncclAllGather: ") + rccl().GetErrorString(r);
    }
    const ncclResult_t re = rccl().GroupEnd();
    if (failed.empty() && re != ncclSuccess) failed = std::string("ncclGroupEnd: ") + rccl().GetErrorString(re);
    if (!failed.empty()) {
      ds->broken = true;
      (void)hipSetDevice(roots[0]->device);
      throw Error(MM3D_EDEVICE, "gather_pair_records: " + failed);
    }
  }
  // the host solves the pose graph once, from the first device's copy; with mm3d_set_debug every device's copy is read back
  // and must hold the same bytes
  const size_t n_read = roots[0]->debug ? D : 1;
  std::vector<unsigned char *> h(n_read);
  for (size_t d = 0; d < n_read; ++d) {
    MM3D_HIP(hipSetDevice(roots[d]->device));
    h[d] = (unsigned char *)roots[d]->pin(bytes * D);
    MM3D_HIP(hipMemcpyAsync(h[d], rbuf[d].get(), bytes * D, hipMemcpyDeviceToHost, roots[d]->stream));
  }
  for (size_t d = 0; d < D; ++d) {
    MM3D_HIP(hipSetDevice(roots[d]->device));
    roots[d]->sync();
  }
  MM3D_HIP(hipSetDevice(roots[0]->device));
  std::memcpy(out.data(), h[0], bytes * D);
  for (size_t d = 1; d < n_read; ++d)
    if (std::memcmp(h[0], h[d], bytes * D) != 0) throw Error(MM3D_EDEVICE, "gather_pair_records: the devices received different records");
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace mm3d
