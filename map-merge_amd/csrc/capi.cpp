// capi.cpp -- the extern "C" boundary of libmm3d.so (include/mm3d.h).  Nothing throws across it.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <thread>
#include <sstream>
#include <string>

#include "device_util.hpp"

using namespace mm3d;

namespace {

template <class F>
int guarded(mm3d_ctx *ctx, F &&f)
{
  if (!ctx) return MM3D_EINVAL;
  std::lock_guard<std::mutex> lock(ctx->mu);
  // error flags waiting for the next sync() belong to the call that recorded them: a call that ends with an
  // exception must not leave them (their pinned words get reused) to the next one, on this context or its helpers
  struct Clean {
    mm3d_ctx *c;
    ~Clean()
    {
      auto one = [](mm3d_ctx *r) {
        r->deferred.clear();
        r->private_objects = false;
        for (mm3d_ctx *h : r->helpers) { h->deferred.clear(); h->private_objects = false; }
      };
      one(c);
      for (mm3d_ctx *p : c->peers) one(p);
    }
  } clean{ctx};
  try {
    if (hipSetDevice(ctx->device) != hipSuccess) throw Error(MM3D_EDEVICE, "hipSetDevice failed");
    f();
    if (!ctx->deferred.empty()) ctx->sync();      // nothing recorded by this call is left unchecked
    return MM3D_OK;
  } catch (const Error &e) {
    ctx->err = e.what();
    return e.status;
  } catch (const std::bad_alloc &) {
    ctx->err = "out of host memory";
    return MM3D_ENOMEM;
  } catch (const std::exception &e) {
    ctx->err = e.what();
    return MM3D_EDEVICE;
  } catch (...) {
    ctx->err = "unknown error";
    return MM3D_EDEVICE;
  }
}

const char *kDescNames[] = {"PFH", "PFHRGB", "FPFH", "RSD", "SHOT", "SC3D"};
const char *kDescFields[] = {"pfh", "pfhrgb", "fpfh", "r_min", "shot", "shape_context"};
const int kDescDims[] = {125, 250, 33, 2, 1344, 1980};
const char *kKpNames[] = {"SIFT", "HARRIS"};
const char *kEstNames[] = {"MATCHING", "SAC_IA"};

int from_string(const char *s, const char *const *names, int n)
{
  if (!s) return MM3D_EINVAL;
  for (int i = 0; i < n; ++i)
    if (std::string(names[i]) == s) return i;
  return MM3D_EINVAL;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- enums / params
const char *mm3d_descriptor_name(int d) { return (d >= 0 && d < 6) ? kDescNames[d] : nullptr; }
int mm3d_descriptor_from_string(const char *s) { return from_string(s, kDescNames, 6); }
const char *mm3d_descriptor_field_name(int d) { return (d >= 0 && d < 6) ? kDescFields[d] : nullptr; }
int mm3d_descriptor_dim(int d) { return (d >= 0 && d < 6) ? kDescDims[d] : MM3D_EINVAL; }
const char *mm3d_keypoint_name(int k) { return (k >= 0 && k < 2) ? kKpNames[k] : nullptr; }
int mm3d_keypoint_from_string(const char *s) { return from_string(s, kKpNames, 2); }
const char *mm3d_estimation_method_name(int m) { return (m >= 0 && m < 2) ? kEstNames[m] : nullptr; }
int mm3d_estimation_method_from_string(const char *s) { return from_string(s, kEstNames, 2); }

void mm3d_params_default(mm3d_params *p)
{
  if (!p) return;
  p->resolution = 0.1;
  p->descriptor_radius = p->resolution * 8.0;
  p->outliers_min_neighbours = 50;
  p->normal_radius = p->resolution * 6.0;
  p->keypoint_type = MM3D_KP_SIFT;
  p->keypoint_threshold = 5.0;
  p->descriptor_type = MM3D_DESC_PFH;
  p->estimation_method = MM3D_EST_MATCHING;
  p->refine_transform = 1;
  p->inlier_threshold = p->resolution * 5.0;
  p->max_correspondence_distance = p->inlier_threshold * 2.0;
  p->max_iterations = 500;
  p->matching_k = 5;
  p->transform_epsilon = 1e-2;
  p->confidence_threshold = 0.0;
  p->output_resolution = 0.05;
}

// pcl::console::parse_argument semantics: the first occurrence of "--name" (argv[1..]) followed by a
// value is taken; unknown options are ignored; a bool is atoi(value) == 1.
int mm3d_params_from_command_line(int argc, const char *const *argv, mm3d_params *p)
{
  if (!p || (argc > 0 && !argv)) return MM3D_EINVAL;
  mm3d_params_default(p);
  auto find = [&](const char *name) -> const char * {
    for (int i = 1; i < argc; ++i)   // pcl::console::find_argument
      if (argv[i] && std::string(argv[i]) == name) return (i + 1 < argc) ? argv[i + 1] : nullptr;
    return nullptr;
  };
  auto get_d = [&](const char *name, double &v) { if (const char *s = find(name)) v = std::atof(s); };
  auto get_i = [&](const char *name, int &v) { if (const char *s = find(name)) v = std::atoi(s); };
  get_d("--resolution", p->resolution);
  get_d("--descriptor_radius", p->descriptor_radius);
  get_i("--outliers_min_neighbours", p->outliers_min_neighbours);
  get_d("--normal_radius", p->normal_radius);
  if (const char *s = find("--keypoint_type"); s && *s) {
    int v = mm3d_keypoint_from_string(s);
    if (v < 0) return MM3D_EINVAL;
    p->keypoint_type = v;
  }
  get_d("--keypoint_threshold", p->keypoint_threshold);
  if (const char *s = find("--descriptor_type"); s && *s) {
    int v = mm3d_descriptor_from_string(s);
    if (v < 0) return MM3D_EINVAL;
    p->descriptor_type = v;
  }
  if (const char *s = find("--estimation_method"); s && *s) {
    int v = mm3d_estimation_method_from_string(s);
    if (v < 0) return MM3D_EINVAL;
    p->estimation_method = v;
  }
  if (const char *s = find("--refine_transform")) p->refine_transform = std::atoi(s) == 1;   // parse_argument(bool&)
  get_d("--inlier_threshold", p->inlier_threshold);
  get_d("--max_correspondence_distance", p->max_correspondence_distance);
  get_i("--max_iterations", p->max_iterations);
  int matching_k = -1;
  get_i("--matching_k", matching_k);
  if (matching_k > 0) p->matching_k = (uint64_t)matching_k;
  get_d("--transform_epsilon", p->transform_epsilon);
  get_d("--confidence_threshold", p->confidence_threshold);
  get_d("--output_resolution", p->output_resolution);
  return MM3D_OK;
}

size_t mm3d_params_to_string(const mm3d_params *p, char *buf, size_t cap)
{
  if (!p) return 0;
  std::ostringstream s;
  s << "resolution: " << p->resolution << std::endl;
  s << "descriptor_radius: " << p->descriptor_radius << std::endl;
  s << "outliers_min_neighbours: " << p->outliers_min_neighbours << std::endl;
  s << "normal_radius: " << p->normal_radius << std::endl;
  s << "keypoint_type: " << (mm3d_keypoint_name(p->keypoint_type) ? mm3d_keypoint_name(p->keypoint_type) : "?") << std::endl;
  s << "keypoint_threshold: " << p->keypoint_threshold << std::endl;
  s << "descriptor_type: " << (mm3d_descriptor_name(p->descriptor_type) ? mm3d_descriptor_name(p->descriptor_type) : "?") << std::endl;
  s << "estimation_method: "
    << (mm3d_estimation_method_name(p->estimation_method) ? mm3d_estimation_method_name(p->estimation_method) : "?") << std::endl;
  s << "refine_transform: " << (p->refine_transform ? 1 : 0) << std::endl;
  s << "inlier_threshold: " << p->inlier_threshold << std::endl;
  s << "max_correspondence_distance: " << p->max_correspondence_distance << std::endl;
  s << "max_iterations: " << p->max_iterations << std::endl;
  s << "matching_k: " << p->matching_k << std::endl;
  s << "transform_epsilon: " << p->transform_epsilon << std::endl;
  s << "confidence_threshold: " << p->confidence_threshold << std::endl;
  s << "output_resolution: " << p->output_resolution << std::endl;
  const std::string str = s.str();
  if (buf && cap) {
    size_t m = std::min(cap - 1, str.size());
    std::memcpy(buf, str.data(), m);
    buf[m] = 0;
  }
  return str.size() + 1;
}

// ---------------------------------------------------------------- context
static std::string &create_error()
{
  static thread_local std::string e = "null context";
  return e;
}

int mm3d_create(int device, mm3d_ctx **out)
{
  if (!out) return MM3D_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
    create_error() = "mm3d_create: no HIP device " + std::to_string(device) + " (" + std::to_string(count) + " visible)";
    return MM3D_EDEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) { create_error() = "mm3d_create: hipSetDevice failed"; return MM3D_EDEVICE; }
  auto *c = new (std::nothrow) mm3d_ctx();
  if (!c) return MM3D_ENOMEM;
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MM3D_EDEVICE; }
  *out = c;
  return MM3D_OK;
}

// One process, several GPUs (the reference's caller is one process: R/src/map_merge_node.cpp:133-153).  The context is the
// first device's; the others' root contexts hang off it (mm3d_ctx::peers), and a DeviceSet holds the RCCL communicators.
int mm3d_create_devices(const int *devices, int n_devices, mm3d_ctx **out)
{
  if (!out) return MM3D_EINVAL;
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return MM3D_EINVAL;
  mm3d_ctx *root = nullptr;
  int st = mm3d_create(devices[0], &root);
  if (st != MM3D_OK) return st;
  for (int d = 1; d < n_devices && st == MM3D_OK; ++d) {
    mm3d_ctx *p = nullptr;
    st = mm3d_create(devices[d], &p);
    if (st == MM3D_OK) root->peers.push_back(p);
  }
  if (st == MM3D_OK) {
    try {
      root->device_set = device_set_create(devices, n_devices);
    } catch (const Error &e) {
      st = e.status;
      create_error() = e.what();
    } catch (...) {
      st = MM3D_EDEVICE;
      create_error() = "mm3d_create_devices: unknown failure";
    }
  }
  if (st != MM3D_OK) { mm3d_destroy(root); return st; }
  (void)hipSetDevice(devices[0]);
  *out = root;
  return MM3D_OK;
}

int mm3d_device_count(const mm3d_ctx *ctx) { return ctx ? (int)ctx->peers.size() + 1 : 0; }
int mm3d_device_at(const mm3d_ctx *ctx, int i)
{
  if (!ctx || i < 0 || i > (int)ctx->peers.size()) return MM3D_EINVAL;
  return i == 0 ? ctx->device : ctx->peers[(size_t)i - 1]->device;
}
int mm3d_devices_use_rccl(const mm3d_ctx *ctx) { return ctx && device_set_has_comms(ctx->device_set) ? 1 : 0; }

static void set_streams_one(mm3d_ctx *ctx, int n_streams)
{
  if (hipSetDevice(ctx->device) != hipSuccess) throw Error(MM3D_EDEVICE, "hipSetDevice failed");
  while ((int)ctx->helpers.size() + 1 > n_streams) {
    mm3d_destroy(ctx->helpers.back());
    ctx->helpers.pop_back();
  }
  while ((int)ctx->helpers.size() + 1 < n_streams) {
    mm3d_ctx *h = nullptr;
    const int st = mm3d_create(ctx->device, &h);
    if (st != MM3D_OK) throw Error(st, "mm3d_set_streams: could not create a helper context");
    ctx->helpers.push_back(h);
  }
}

int mm3d_set_streams(mm3d_ctx *ctx, int n_streams)
{
  if (n_streams < 1 || n_streams > 64) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    set_streams_one(ctx, n_streams);
    for (mm3d_ctx *p : ctx->peers) set_streams_one(p, n_streams);     // (every device of an mm3d_create_devices context)
    (void)hipSetDevice(ctx->device);
  });
}

int mm3d_get_streams(const mm3d_ctx *ctx) { return ctx ? (int)ctx->helpers.size() + 1 : 0; }

void mm3d_destroy(mm3d_ctx *ctx)
{
  if (!ctx) return;
  device_set_destroy(ctx->device_set);                  // (the communicators go before the streams they were used on)
  ctx->device_set = nullptr;
  for (mm3d_ctx *p : ctx->peers) mm3d_destroy(p);
  ctx->peers.clear();
  for (mm3d_ctx *h : ctx->helpers) mm3d_destroy(h);
  ctx->helpers.clear();
  (void)hipSetDevice(ctx->device);
  (void)stream_wait(ctx->stream);
  for (auto &p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  ctx->pool->trim();
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->scan_status) (void)hipFree(ctx->scan_status);
  if (ctx->scan_ticket) (void)hipFree(ctx->scan_ticket);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// (a null context: why the calling thread's last mm3d_create / mm3d_create_devices failed -- there is no context to ask then)
const char *mm3d_last_error(const mm3d_ctx *ctx) { return ctx ? ctx->err.c_str() : create_error().c_str(); }
int mm3d_last_icp_iterations(const mm3d_ctx *ctx) { return ctx ? ctx->last_icp_iterations : 0; }
int mm3d_last_icp_converged(const mm3d_ctx *ctx) { return ctx ? ctx->last_icp_converged : 0; }
int mm3d_last_run_stage_seconds(const mm3d_ctx *ctx, double *features_s, double *total_s)
{
  if (!ctx) return MM3D_EINVAL;
  if (features_s) *features_s = ctx->last_features_s;
  if (total_s) *total_s = ctx->last_total_s;
  return MM3D_OK;
}
int mm3d_last_run_device_seconds(const mm3d_ctx *ctx, double *exchange_s, double *pairs_s, double *gather_s)
{
  if (!ctx) return MM3D_EINVAL;
  if (exchange_s) *exchange_s = ctx->last_exchange_s;
  if (pairs_s) *pairs_s = ctx->last_pairs_s;
  if (gather_s) *gather_s = ctx->last_gather_s;
  return MM3D_OK;
}
size_t mm3d_last_run_map_sizes(const mm3d_ctx *ctx, size_t *points, size_t *keypoints, size_t capacity)
{
  if (!ctx) return 0;
  const size_t n = ctx->last_points.size();
  for (size_t i = 0; i < n && i < capacity; ++i) {
    if (points) points[i] = ctx->last_points[i];
    if (keypoints) keypoints[i] = ctx->last_keypoints[i];
  }
  return n;
}
void mm3d_set_debug(mm3d_ctx *ctx, int on)
{
  if (!ctx) return;
  ctx->debug = on != 0;
  ctx->knn_fallback_rows = ctx->knn_rows = 0;
}
long long mm3d_debug_knn_fallback_rows(mm3d_ctx *ctx) { return ctx ? ctx->knn_fallback_rows : 0; }
long long mm3d_debug_knn_rows(mm3d_ctx *ctx) { return ctx ? ctx->knn_rows : 0; }
void mm3d_debug_waits(mm3d_ctx *ctx, long long out[2])
{
  out[0] = ctx ? ctx->waits : 0;
  out[1] = ctx ? ctx->wait_ns : 0;
  if (ctx) {                      // (with the worker contexts of mm3d_set_streams, on every device: the whole library call's waits)
    for (mm3d_ctx *h : ctx->helpers) { out[0] += h->waits; out[1] += h->wait_ns; }
    for (mm3d_ctx *p : ctx->peers) {
      out[0] += p->waits; out[1] += p->wait_ns;
      for (mm3d_ctx *h : p->helpers) { out[0] += h->waits; out[1] += h->wait_ns; }
    }
  }
}
int mm3d_debug_float_chain(mm3d_ctx *ctx, const float *incr, const unsigned *hits, int n, float *out)
{
  if (n < 0 || (n && (!incr || !hits || !out))) return MM3D_EINVAL;
  return guarded(ctx, [&] { debug_float_chain(ctx, incr, hits, n, out); });
}
int mm3d_debug_libm(mm3d_ctx *ctx, int fn, const float *x, const float *y, int n, float *out)
{
  if (n < 0 || fn < 0 || fn > 5 || (n && (!x || !out || (fn == 4 && !y)))) return MM3D_EINVAL;
  return guarded(ctx, [&] { debug_libm(ctx, fn, x, y, n, out); });
}
int mm3d_debug_sift_cert_octave(mm3d_ctx *ctx, const mm3d_cloud *points, double min_scale, int octave, float *val, float *bound, size_t capacity,
                                size_t *n_out)
{
  if (!points || !n_out || octave < 0 || (capacity && (!val || !bound))) return MM3D_EINVAL;
  return guarded(ctx, [&] { *n_out = debug_sift_cert_octave(ctx, points, min_scale, octave, val, bound, capacity); });
}
void mm3d_debug_sift_cert_stats(long long out[8], int reset) { if (out) debug_sift_cert_stats(out, reset); }
void mm3d_debug_sift_cert_min(int n) { debug_sift_cert_min(n); }
float mm3d_debug_cloud_voxel_leaf(const mm3d_cloud *cloud) { return cloud ? cloud->voxel_leaf : 0.0f; }
void mm3d_debug_sacia_stats(long long out[4], int reset, int collect) { if (out) debug_sacia_stats(out, reset, collect); }
void mm3d_srand(mm3d_ctx *ctx, unsigned seed) { if (ctx) ctx->rnd.seed(seed); }
int mm3d_synchronize(mm3d_ctx *ctx) { return guarded(ctx, [&] { ctx->sync(); }); }

// ---------------------------------------------------------------- objects
int mm3d_cloud_create(mm3d_ctx *ctx, const void *points, size_t n, size_t stride, size_t rgba_offset, mm3d_cloud **out)
{
  if (!out) return MM3D_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] { *out = cloud_from_memory(ctx, points, n, stride, rgba_offset); ctx->sync(); });
}
size_t mm3d_cloud_size(const mm3d_cloud *c) { return c ? c->n : 0; }
int mm3d_cloud_download(mm3d_ctx *ctx, const mm3d_cloud *c, void *dst, size_t stride, size_t rgba_offset)
{
  if (!c || (!dst && c->n)) return MM3D_EINVAL;
  return guarded(ctx, [&] { cloud_download(ctx, c, dst, stride, rgba_offset); });
}
void mm3d_cloud_free(mm3d_ctx *ctx, mm3d_cloud *c)
{
  if (!ctx || !c) return;
  std::lock_guard<std::mutex> lock(ctx->mu);
  (void)stream_wait(ctx->stream);
  delete c;
}

size_t mm3d_normals_size(const mm3d_normals *n) { return n ? n->n : 0; }
int mm3d_normals_download(mm3d_ctx *ctx, const mm3d_normals *n, void *dst, size_t stride)
{
  if (!n || (!dst && n->n) || stride < 16 || stride % 4) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    if (!n->n) return;
    MM3D_HIP(hipMemcpy2DAsync(dst, stride, n->nrm.get(), 16, 16, n->n, hipMemcpyDefault, ctx->stream));
    ctx->sync();
  });
}
int mm3d_normals_create(mm3d_ctx *ctx, const void *normals, size_t n, size_t stride, mm3d_normals **out)
{
  if (!out || stride < 16 || stride % 4 || (!normals && n)) return MM3D_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] {
    auto *r = new mm3d_normals();
    r->n = n;
    r->nrm = DevBuf<float4>(ctx, n);
    if (n) {
      MM3D_HIP(hipMemcpy2DAsync(r->nrm.get(), 16, normals, stride, 16, n, hipMemcpyDefault, ctx->stream));
      ctx->sync();
    }
    *out = r;
  });
}
void mm3d_normals_free(mm3d_ctx *ctx, mm3d_normals *n)
{
  if (!ctx || !n) return;
  std::lock_guard<std::mutex> lock(ctx->mu);
  (void)stream_wait(ctx->stream);
  delete n;
}

size_t mm3d_desc_size(const mm3d_desc *d) { return d ? d->n : 0; }
int mm3d_desc_dim(const mm3d_desc *d) { return d ? d->dim : 0; }
int mm3d_desc_type(const mm3d_desc *d) { return d ? d->type : MM3D_EINVAL; }
int mm3d_desc_download(mm3d_ctx *ctx, const mm3d_desc *d, float *dst)
{
  if (!d || (!dst && d->n)) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    if (!d->n) return;
    MM3D_HIP(hipMemcpyAsync(dst, d->data.get(), d->n * d->dim * sizeof(float), hipMemcpyDefault, ctx->stream));
    ctx->sync();
  });
}
int mm3d_desc_download_frames(mm3d_ctx *ctx, const mm3d_desc *d, float *dst)
{
  if (!d || (!dst && d->n)) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    if (d->type != MM3D_DESC_SHOT || (d->n && !d->rf.get())) throw Error(MM3D_EINVAL, "these descriptors carry no reference frames");
    if (!d->n) return;
    MM3D_HIP(hipMemcpyAsync(dst, d->rf.get(), d->n * 9 * sizeof(float), hipMemcpyDefault, ctx->stream));
    ctx->sync();
  });
}
static mm3d_desc *desc_from_memory(mm3d_ctx *ctx, const float *data, size_t n, int descriptor_type)
{
  const int dim = mm3d_descriptor_dim(descriptor_type);
  if (dim < 0) throw Error(MM3D_EINVAL, "unknown descriptor type");
  std::unique_ptr<mm3d_desc> r(new mm3d_desc());
  r->n = n; r->dim = dim; r->type = descriptor_type;
  r->data = DevBuf<float>(ctx, n * dim);
  if (n) {
    MM3D_HIP(hipMemcpyAsync(r->data.get(), data, n * dim * sizeof(float), hipMemcpyDefault, ctx->stream));
    ctx->sync();
  }
  return r.release();
}

int mm3d_debug_desc_knn(mm3d_ctx *ctx, const float *a, size_t na, const float *b, size_t nb, int dim, int k, int *idx, float *d2)
{
  if (!a || !b || !idx || !d2 || na == 0 || nb == 0 || dim < 1 || k < 1) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    auto make = [&](const float *data, size_t n) {
      std::unique_ptr<mm3d_desc> r(new mm3d_desc());
      r->n = n; r->dim = dim; r->type = -1;
      r->data = DevBuf<float>(ctx, n * (size_t)dim);
      MM3D_HIP(hipMemcpyAsync(r->data.get(), data, n * (size_t)dim * sizeof(float), hipMemcpyDefault, ctx->stream));
      return r;
    };
    std::unique_ptr<mm3d_desc> A = make(a, na), B = make(b, nb);
    DevBuf<int> di;
    DevBuf<float> dd;
    desc_knn(ctx, A.get(), B.get(), k, di, dd);
    MM3D_HIP(hipMemcpyAsync(idx, di.get(), na * (size_t)k * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MM3D_HIP(hipMemcpyAsync(d2, dd.get(), na * (size_t)k * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
  });
}

int mm3d_desc_create(mm3d_ctx *ctx, const float *data, size_t n, int descriptor_type, mm3d_desc **out)
{
  if (!out || (!data && n)) return MM3D_EINVAL;
  *out = nullptr;
  if (mm3d_descriptor_dim(descriptor_type) < 0) return MM3D_EINVAL;
  return guarded(ctx, [&] { *out = desc_from_memory(ctx, data, n, descriptor_type); });
}
void mm3d_desc_free(mm3d_ctx *ctx, mm3d_desc *d)
{
  if (!ctx || !d) return;
  std::lock_guard<std::mutex> lock(ctx->mu);
  (void)stream_wait(ctx->stream);
  delete d;
}

// ---------------------------------------------------------------- features.h
int mm3d_downsample(mm3d_ctx *ctx, const mm3d_cloud *in, double resolution, mm3d_cloud **out)
{
  if (!in || !out) return MM3D_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] { *out = downsample(ctx, in, resolution); });
}

int mm3d_remove_outliers(mm3d_ctx *ctx, const mm3d_cloud *in, double radius, int min_neighbours, mm3d_cloud **out)
{
  if (!in || !out) return MM3D_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] { *out = remove_outliers(ctx, in, radius, min_neighbours); });
}

int mm3d_compute_normals(mm3d_ctx *ctx, const mm3d_cloud *in, double radius, mm3d_normals **out)
{
  if (!in || !out) return MM3D_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] { *out = compute_normals(ctx, in, radius); ctx->sync(); });
}

int mm3d_detect_keypoints(mm3d_ctx *ctx, const mm3d_cloud *points, const mm3d_normals *normals, int type, double threshold,
                          double radius, double resolution, mm3d_cloud **keypoints)
{
  (void)normals; (void)radius;
  if (!points || !keypoints) return MM3D_EINVAL;
  *keypoints = nullptr;
  return guarded(ctx, [&] {
    if (type == MM3D_KP_SIFT) {
      // detectKeypointsSIFT(points, resolution, 3, 3, threshold)  (features.cpp:92)
      *keypoints = detect_keypoints_sift(ctx, points, resolution, 3, 3, threshold);
    } else if (type == MM3D_KP_HARRIS) {
      // detectKeypointsHarris(points, normals, threshold, radius)  (features.cpp:94)
      if (!normals) throw Error(MM3D_EINVAL, "HARRIS keypoints need the surface normals");
      *keypoints = detect_keypoints_harris(ctx, points, normals, threshold, radius);
    } else {
      throw Error(MM3D_EINVAL, "invalid keypoint type");
    }
  });
}

int mm3d_harris_response(mm3d_ctx *ctx, const mm3d_cloud *points, const mm3d_normals *normals, double radius, float *dst)
{
  if (!points || !normals || (!dst && points->n)) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    DevBuf<float> r;
    harris_response(ctx, points, normals, radius, r);
    if (points->n) MM3D_HIP(hipMemcpyAsync(dst, r.get(), points->n * sizeof(float), hipMemcpyDefault, ctx->stream));
    ctx->sync();
  });
}

int mm3d_compute_descriptors(mm3d_ctx *ctx, const mm3d_cloud *points, const mm3d_normals *normals, mm3d_cloud *keypoints,
                             int descriptor, double feature_radius, mm3d_desc **out)
{
  if (!points || !normals || !keypoints || !out) return MM3D_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] {
    if (descriptor == MM3D_DESC_FPFH) {
      *out = compute_fpfh(ctx, points, normals, keypoints, feature_radius);
    } else if (descriptor == MM3D_DESC_PFH) {
      *out = compute_pfh(ctx, points, normals, keypoints, feature_radius);
    } else if (descriptor == MM3D_DESC_PFHRGB) {
      *out = compute_pfhrgb(ctx, points, normals, keypoints, feature_radius);
    } else if (descriptor == MM3D_DESC_RSD) {
      *out = compute_rsd(ctx, points, normals, keypoints, feature_radius);
    } else if (descriptor == MM3D_DESC_SHOT) {
      *out = compute_shot(ctx, points, normals, keypoints, feature_radius);
    } else if (descriptor == MM3D_DESC_SC3D) {
      *out = compute_sc3d(ctx, points, normals, keypoints, feature_radius);
    } else {
      throw Error(MM3D_EINVAL, "unknown descriptor type");   // dispatch_descriptors.h:63
    }
  });
}

// ---------------------------------------------------------------- matching.h
int mm3d_find_correspondences(mm3d_ctx *ctx, const mm3d_desc *source, const mm3d_desc *target, size_t k, mm3d_corr *out,
                              size_t cap, size_t *n)
{
  if (!source || !target || !n) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    // assertDescriptorsPair / dispatch by field name: both sides must be the same descriptor kind
    if (source->type != target->type) throw Error(MM3D_EINVAL, "descriptor types differ");
    std::vector<mm3d_corr> v;
    find_correspondences(ctx, source, target, k, v);
    *n = v.size();
    if (out) {
      if (cap < v.size()) throw Error(MM3D_ECAPACITY, "correspondence buffer too small");
      std::memcpy(out, v.data(), v.size() * sizeof(mm3d_corr));
    }
  });
}

int mm3d_estimate_transform_from_correspondences(mm3d_ctx *ctx, const mm3d_cloud *skp, const mm3d_cloud *tkp,
                                                 const mm3d_corr *corr, size_t n_corr, double inlier_threshold, float T[16],
                                                 mm3d_corr *inliers, size_t cap, size_t *n_inliers)
{
  if (!skp || !tkp || (!corr && n_corr) || !T) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    std::vector<mm3d_corr> inl;
    ransac_transform(ctx, skp, tkp, corr, n_corr, inlier_threshold, T, inl);
    if (n_inliers) *n_inliers = inl.size();
    if (inliers) {
      if (cap < inl.size()) throw Error(MM3D_ECAPACITY, "inlier buffer too small");
      std::memcpy(inliers, inl.data(), inl.size() * sizeof(mm3d_corr));
    }
  });
}

int mm3d_estimate_transform_from_descriptors(mm3d_ctx *ctx, const mm3d_cloud *skp, const mm3d_desc *sd, const mm3d_cloud *tkp,
                                             const mm3d_desc *td, double min_sample_distance, double max_corr_dist,
                                             int max_iterations, float T[16])
{
  if (!skp || !sd || !tkp || !td || !T) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    if (sd->type != td->type) throw Error(MM3D_EINVAL, "descriptor types differ");
    sac_ia(ctx, skp, sd, tkp, td, min_sample_distance, max_corr_dist, max_iterations, T, true);
  });
}

int mm3d_estimate_transform_icp(mm3d_ctx *ctx, const mm3d_cloud *source, const mm3d_cloud *target, const float guess[16],
                                double max_corr_dist, double outlier_rejection_threshold, int max_iterations, double eps,
                                float T[16])
{
  (void)outlier_rejection_threshold;   // setRANSACOutlierRejectionThreshold: no rejector is registered in the reference
  if (!source || !target || !guess || !T) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    IcpResult r = icp(ctx, source, target, guess, max_corr_dist, max_iterations, eps);
    std::memcpy(T, r.T, sizeof(r.T));
  });
}

int mm3d_estimate_transform(mm3d_ctx *ctx, const mm3d_cloud *sp, const mm3d_cloud *skp, const mm3d_desc *sd,
                            const mm3d_cloud *tp, const mm3d_cloud *tkp, const mm3d_desc *td, int method, int refine,
                            double inlier_threshold, double max_corr_dist, int max_iterations, size_t matching_k, double eps,
                            float T[16])
{
  if (!sp || !skp || !sd || !tp || !tkp || !td || !T) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    estimate_transform(ctx, sp, skp, sd, tp, tkp, td, method, refine, inlier_threshold, max_corr_dist, max_iterations,
                       matching_k, eps, T, true);
  });
}

int mm3d_transform_score(mm3d_ctx *ctx, const mm3d_cloud *source, const mm3d_cloud *target, const float T[16],
                         double max_distance, double *score)
{
  if (!source || !target || !T || !score) return MM3D_EINVAL;
  return guarded(ctx, [&] { *score = transform_score(ctx, source, target, T, max_distance); });
}

// ---------------------------------------------------------------- map bundles
// wait = false: the caller goes on in the same stream (map_prepare_impl) and waits once, there
static mm3d_map *map_features_impl(mm3d_ctx *ctx, const mm3d_cloud *raw, const mm3d_params *p, bool wait = true)
{
  if (p->keypoint_type != MM3D_KP_SIFT && p->keypoint_type != MM3D_KP_HARRIS) throw Error(MM3D_EINVAL, "invalid keypoint type");
  if (p->descriptor_type < 0 || p->descriptor_type >= 6) throw Error(MM3D_EINVAL, "unknown descriptor type");   // dispatch_descriptors.h:63
  std::unique_ptr<mm3d_cloud> down(downsample(ctx, raw, p->resolution));
  // NB: the outlier radius is the DESCRIPTOR radius (map_merging.cpp:219-220)
  std::unique_ptr<mm3d_cloud> filt(remove_outliers(ctx, down.get(), p->descriptor_radius, p->outliers_min_neighbours));
  down.reset();
  // computeSurfaceNormals, then detectKeypoints(points, normals, type, keypoint_threshold, normal_radius, resolution)
  // (map_merging.cpp:225-233).  SIFT does not read the normals, and its first octave builds every point's sorted
  // neighbour list over a ball that contains the normals': the two stages share that launch (sift.hip), same bits.
  std::unique_ptr<mm3d_normals> nrm;
  std::unique_ptr<mm3d_cloud> kp;
  if (p->keypoint_type == MM3D_KP_HARRIS) {
    nrm.reset(compute_normals(ctx, filt.get(), p->normal_radius));
    kp.reset(detect_keypoints_harris(ctx, filt.get(), nrm.get(), p->keypoint_threshold, p->normal_radius));
  } else {
    mm3d_normals *n_out = nullptr;
    static const bool share_grid = [] { const char *e = getenv("MM3D_SIFT_NO_SHARED_GRID"); return !(e && atoi(e)); }();   // A/B knob
    // (every descriptor searches `filt` on a grid of descriptor_radius / 2 cells: the first octave uses that one too)
    kp.reset(detect_keypoints_sift(ctx, filt.get(), p->resolution, 3, 3, p->keypoint_threshold, p->normal_radius, &n_out,
                                   share_grid ? (float)(p->descriptor_radius * 0.5) : 0.0f));
    nrm.reset(n_out);
  }
  std::unique_ptr<mm3d_desc> desc(p->descriptor_type == MM3D_DESC_PFH    ? compute_pfh(ctx, filt.get(), nrm.get(), kp.get(), p->descriptor_radius)
                                  : p->descriptor_type == MM3D_DESC_SC3D ? compute_sc3d(ctx, filt.get(), nrm.get(), kp.get(), p->descriptor_radius)
                                  : p->descriptor_type == MM3D_DESC_RSD ? compute_rsd(ctx, filt.get(), nrm.get(), kp.get(), p->descriptor_radius)
                                  : p->descriptor_type == MM3D_DESC_PFHRGB ? compute_pfhrgb(ctx, filt.get(), nrm.get(), kp.get(), p->descriptor_radius)
                                  : p->descriptor_type == MM3D_DESC_SHOT ? compute_shot(ctx, filt.get(), nrm.get(), kp.get(), p->descriptor_radius)
                                                                         : compute_fpfh(ctx, filt.get(), nrm.get(), kp.get(), p->descriptor_radius));
  if (wait) ctx->sync();
  auto *m = new mm3d_map();
  m->points = filt.release();
  m->keypoints = kp.release();
  m->desc = desc.release();
  return m;
}

int mm3d_map_features(mm3d_ctx *ctx, const mm3d_cloud *raw, const mm3d_params *params, mm3d_map **out)
{
  if (!raw || !params || !out) return MM3D_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] { *out = map_features_impl(ctx, raw, params); });
}

const mm3d_cloud *mm3d_map_points(const mm3d_map *m) { return m ? m->points : nullptr; }
const mm3d_cloud *mm3d_map_keypoints(const mm3d_map *m) { return m ? m->keypoints : nullptr; }
const mm3d_desc *mm3d_map_descriptors(const mm3d_map *m) { return m ? m->desc : nullptr; }

int mm3d_map_from_parts(mm3d_ctx *ctx, mm3d_cloud *points, mm3d_cloud *keypoints, mm3d_desc *desc, mm3d_map **out)
{
  if (!ctx || !points || !keypoints || !desc || !out) return MM3D_EINVAL;
  if (keypoints->n != desc->n) return MM3D_EINVAL;
  auto *m = new mm3d_map();
  m->points = points; m->keypoints = keypoints; m->desc = desc;
  *out = m;
  return MM3D_OK;
}

static void map_prepare_impl(mm3d_ctx *ctx, mm3d_map *m, const mm3d_params *p)
{
  prepare_pair_search(ctx, m->points, p->max_correspondence_distance, p->max_correspondence_distance);
  if (p->estimation_method == MM3D_EST_SAC_IA) prepare_sacia_target(ctx, m->keypoints, (float)p->max_correspondence_distance);
  desc_knn_prepare_target(ctx, m->desc);
  (void)cloud_host(ctx, m->keypoints, false);       // (the keypoints' host copy rides on the wait below)
  ctx->sync();                                       // everything complete, the error flags the kernels left looked at
}

int mm3d_map_prepare(mm3d_ctx *ctx, mm3d_map *m, const mm3d_params *p)
{
  if (!m || !p) return MM3D_EINVAL;
  return guarded(ctx, [&] { map_prepare_impl(ctx, m, p); });
}

void mm3d_map_free(mm3d_ctx *ctx, mm3d_map *m)
{
  if (!ctx || !m) return;
  std::lock_guard<std::mutex> lock(ctx->mu);
  (void)stream_wait(ctx->stream);
  delete m->points; delete m->keypoints; delete m->desc;
  delete m;
}

static void pair_estimate_impl(mm3d_ctx *ctx, const mm3d_map *s, const mm3d_map *t, const mm3d_params *p, bool execute,
                               mm3d_pair_result *out)
{
  std::memset(out->transform, 0, sizeof(out->transform));
  out->confidence = 0.0;
  out->icp_iterations = 0;
  out->n_correspondences = out->n_inliers = out->icp_correspondences = 0;
  // estimateTransform and transformScore of its result (R/src/map_merging.cpp:91-107) as one device
  // pipeline: the transform never visits the host in between
  double score = DBL_MAX;
  PairCounts counts;
  const int iters = estimate_pair(ctx, s->points, s->keypoints, s->desc, t->points, t->keypoints, t->desc,
                                  p->estimation_method, p->refine_transform, p->inlier_threshold,
                                  p->max_correspondence_distance, p->max_iterations, (size_t)p->matching_k,
                                  p->transform_epsilon, out->transform, execute, true, p->max_correspondence_distance, &score, &counts);
  if (!execute) return;
  out->icp_iterations = iters;
  out->n_correspondences = counts.n_correspondences;
  out->n_inliers = counts.n_inliers;
  out->icp_correspondences = counts.icp_correspondences;
  out->confidence = 1.0 / score;
}

constexpr size_t kPairBatch = 16;     // pairs whose tails and scoring share launches
// the two experiment knobs of the pair batches, validated once (a share <= 0 or not a number would divide by zero and cast
// inf to size_t; a batch cap of 0 would never claim a pair and leave the scheduler waiting for ever)
static double pair_share_knob()
{
  static const double v = [] {
    const char *e = getenv("MM3D_PAIR_SHARE");
    double s = e ? atof(e) : 0.25;
    if (!(s >= 1.0 / 64.0)) s = 1.0 / 64.0;           // (also catches NaN)
    return std::min(s, 64.0);
  }();
  return v;
}
static size_t pair_batch_knob()
{
  static const size_t v = [] {
    const char *e = getenv("MM3D_PAIR_BATCH");
    const long b = e ? atol(e) : (long)kPairBatch;
    return (size_t)std::min<long>(std::max<long>(b, 1), 32);      // (32: the largest batch ever run)
  }();
  return v;
}

// Several pairs on one context: the initial estimates one after the other (each from its own generator state),
// then every pair's ICP + score tail in lockstep, one launch per step for the whole batch (icp_score_batch).
struct PairWork { const mm3d_map *s, *t; mm3d_pair_result *out; GlibcRand rnd; };
static void pairs_estimate_batch(mm3d_ctx *ctx, PairWork *w, size_t n, const mm3d_params *p)
{
  std::vector<PairFront> fronts(n);
  std::vector<IcpScoreJob> jobs(n);
  std::vector<SacPrepared> prepared;
  for (size_t i = 0; i < n; ++i) {
    mm3d_pair_result *out = w[i].out;
    std::memset(out->transform, 0, sizeof(out->transform));
    out->confidence = 0.0;
    out->icp_iterations = 0;
    out->n_correspondences = out->n_inliers = out->icp_correspondences = 0;
    ctx->rnd = w[i].rnd;
    if (p->estimation_method == MM3D_EST_SAC_IA) {
      // argument mapping of matching.cpp:243-246: min_sample_distance := inlier_threshold
      sac_ia_replay(ctx, w[i].s->keypoints, w[i].s->desc, w[i].t->keypoints, w[i].t->desc, p->inlier_threshold, p->max_iterations, true,
                    fronts[i]);
      prepared.push_back(SacPrepared{w[i].s->keypoints, w[i].t->keypoints, w[i].s->desc, w[i].t->desc, &fronts[i]});
    } else {
      estimate_pair_front(ctx, w[i].s->keypoints, w[i].s->desc, w[i].t->keypoints, w[i].t->desc, p->estimation_method,
                          p->inlier_threshold, p->max_correspondence_distance, p->max_iterations, (size_t)p->matching_k, true, fronts[i]);
    }
  }
  if (!prepared.empty()) {
    // the sampled rows of every pair with the same target go through one descriptor search, and the hypotheses of
    // all the batch's pairs are scored by the same five launches
    std::stable_sort(prepared.begin(), prepared.end(), [](const SacPrepared &a, const SacPrepared &b) { return a.td < b.td; });
    std::vector<DevBuf<int>> nn_owners;
    std::vector<DevBuf<float>> nd_owners;
    for (size_t a = 0; a < prepared.size();) {
      size_t b = a;
      while (b < prepared.size() && prepared[b].td == prepared[a].td) ++b;
      nn_owners.emplace_back();
      nd_owners.emplace_back();
      sac_ia_knn(ctx, &prepared[a], (int)(b - a), nn_owners.back(), nd_owners.back());
      a = b;
    }
    sac_ia_finish(ctx, prepared.data(), (int)prepared.size(), p->max_correspondence_distance);
  }
  for (size_t i = 0; i < n; ++i) {
    jobs[i].src = w[i].s->points;
    jobs[i].tgt = w[i].t->points;
    jobs[i].guess_dev = fronts[i].on_device ? fronts[i].dT0.get() : nullptr;
    std::memcpy(jobs[i].guess_host, fronts[i].T0, sizeof(fronts[i].T0));
  }
  // estimateTransform's ICP and transformScore of its result (R/src/map_merging.cpp:91-107), max_distance = max_correspondence_distance
  icp_score_batch(ctx, jobs.data(), (int)n, p->refine_transform != 0, p->max_correspondence_distance, p->max_iterations, p->transform_epsilon,
                  true, p->max_correspondence_distance);
  for (size_t i = 0; i < n; ++i) {
    mm3d_pair_result *out = w[i].out;
    std::memcpy(out->transform, jobs[i].out.T, sizeof(out->transform));
    out->icp_iterations = jobs[i].out.iterations;
    out->n_correspondences = fronts[i].counts.n_correspondences;
    out->n_inliers = fronts[i].counts.n_inliers;
    out->icp_correspondences = jobs[i].out.n_corr;
    out->confidence = 1.0 / jobs[i].out.score;
  }
}

int mm3d_pair_estimate(mm3d_ctx *ctx, const mm3d_map *source, const mm3d_map *target, const mm3d_params *params, int execute,
                       mm3d_pair_result *out)
{
  if (!source || !target || !params || !out) return MM3D_EINVAL;
  return guarded(ctx, [&] { pair_estimate_impl(ctx, source, target, params, execute != 0, out); });
}

int mm3d_pairs_skip(mm3d_ctx *ctx, const mm3d_map *const *sources, const mm3d_map *const *targets, size_t n, const mm3d_params *params)
{
  if (!params || (n && (!sources || !targets))) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    for (size_t i = 0; i < n; ++i) {
      const mm3d_map *s = sources[i], *t = targets[i];
      if (!s || !t) throw Error(MM3D_EINVAL, "null map");
      if (s->keypoints->n == 0 || t->keypoints->n == 0) continue;      // not a pair (map_merging.cpp:250)
      pair_rand_replay(ctx->rnd, params->estimation_method, cloud_host(ctx, s->keypoints), params->inlier_threshold,
                       params->max_iterations);
    }
  });
}

int mm3d_global_transforms(const mm3d_pair_result *pairs, size_t n_pairs, double confidence_threshold, size_t n_clouds,
                           float *out_T, size_t *n_out)
{
  if ((!pairs && n_pairs) || !out_T || !n_out) return MM3D_EINVAL;
  try {
    return global_transforms(pairs, n_pairs, confidence_threshold, n_clouds, out_T, n_out);
  } catch (...) {
    return MM3D_ENOMEM;
  }
}

// estimateMapsTransforms over the context's streams (mm3d_set_streams).  The reference's two loops
// (map_merging.cpp:212-242 per cloud, :256-269 per pair) are dealt to S workers, one context (HIP
// stream + memory pool) and one host thread each: about 3/4 of them extract features, every worker
// then claims pairs in the reference's order and waits until both maps of its pair exist.  A map is
// prepared (map_prepare_impl) before it is published, so pairs only read it.  The reference's single
// rand() stream is kept by replay: every worker starts from the caller's generator state and replays
// the draws of the pairs it does not execute (execute = false, host only), so each pair sees exactly
// the state the sequential loop would give it; worker 0 (the caller's own context) replays to the
// end, which leaves the caller's generator where the sequential loop would.
static void estimate_maps_streams(mm3d_ctx *ctx, const mm3d_cloud_view *clouds, size_t n, const mm3d_params *params, float *out_T,
                                  size_t *n_out, mm3d_pair_result *pairs_out, size_t *n_pairs_out)
{
  std::vector<mm3d_ctx *> cs{ctx};
  cs.insert(cs.end(), ctx->helpers.begin(), ctx->helpers.end());
  const size_t S = cs.size();
  // many small maps (at least two per worker): every worker extracts features first, and the pairs then start in
  // batches; otherwise (few large maps) 3/4 of the workers do, and the others begin with the pairs of the first maps
  // (measured on 16 x 500 k points with 16 workers, map pairs/s at 6 / 8 / 10 / 12 / 16 feature workers:
  // 740 / 741 / 736 / 765 / 741;
  // and three quarters of the maps with 8 x 2 M points: 93.0 pairs/s with 6 feature workers, 91.2 with 12 = all 8 maps at
  // once; the dense indoor variant 59.0 / 53.9)
  size_t F = (S <= 4 || n >= 2 * S) ? S : std::max<size_t>(4, std::min(S * 3 / 4, (n * 3 + 3) / 4));
  if (const char *e = std::getenv("MM3D_FEATURE_WORKERS")) {     // tuning knob: how many workers start on features
    const long v = std::atol(e);
    if (v >= 1) F = std::min<size_t>(S, (size_t)v);
  }
  std::vector<std::pair<size_t, size_t>> all;
  for (size_t i = 0; i + 1 < n; ++i)
    for (size_t j = i + 1; j < n; ++j) all.emplace_back(i, j);
  std::vector<mm3d_map *> maps(n, nullptr);
  struct MapsGuard {                                    // the maps go when the call ends, whichever way
    std::vector<mm3d_map *> &m;
    ~MapsGuard()
    {
      for (mm3d_map *x : m)
        if (x) { delete x->points; delete x->keypoints; delete x->desc; delete x; }
    }
  } maps_guard{maps};
  std::vector<mm3d_pair_result> rec(all.size());
  std::vector<char> ready(n, 0), done(all.size(), 0);
  std::mutex mu;
  std::condition_variable cv;
  size_t next_map = 0;
  bool abort = false;
  std::exception_ptr first_error;
  const GlibcRand rnd0 = ctx->rnd;
  const auto t_start = std::chrono::steady_clock::now();
  auto since_start = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  ctx->last_points.assign(n, 0);
  ctx->last_keypoints.assign(n, 0);
  ctx->last_features_s = ctx->last_total_s = 0.0;

  // The reference's single rand() stream, without serialising the pairs on it: the draws a pair consumes
  // depend on its SOURCE keypoints only (pair_rand_replay), given that its target has a keypoint at all.
  // state_at[q] = the generator before pair q; it is advanced pair by pair (once, under rng_mu) as far as
  // a worker needs it, taking "is pair q live" from the maps that exist and ASSUMING a target that is
  // still being computed will have keypoints.  A pair can therefore start as soon as its own two maps
  // and the sources of the rows before it exist -- not only after the last map.  Every assumption is
  // checked once all maps exist; a wrong one (a map without keypoints, e.g. an untextured cloud) makes
  // the call redo the pair loop sequentially, which is the reference's loop.
  const size_t P = all.size();
  std::vector<GlibcRand> state_at(P + 1, rnd0);
  std::vector<char> assumed_live(P, 0), claimed(P, 0);
  size_t known_upto = 0;                                // state_at[0 .. known_upto] are final
  std::mutex rng_mu;
  auto wait_ready = [&](size_t i) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return abort || ready[i]; });
    if (abort) throw Error(MM3D_EDEVICE, "aborted");
  };
  auto is_ready = [&](size_t i) { std::lock_guard<std::mutex> lk(mu); return ready[i] != 0; };
  auto advance_states = [&](size_t p) {                 // make state_at[p] final
    std::lock_guard<std::mutex> rl(rng_mu);
    while (known_upto < p) {
      const size_t q = known_upto, a = all[q].first, b = all[q].second;
      wait_ready(a);
      bool live = maps[a]->keypoints->n > 0;
      if (live) {
        if (is_ready(b)) live = maps[b]->keypoints->n > 0;
        else assumed_live[q] = 1;
      }
      GlibcRand r = state_at[q];
      if (live) pair_rand_replay(r, params->estimation_method, cloud_host(cs[0], maps[a]->keypoints), params->inlier_threshold,
                                 params->max_iterations);
      state_at[q + 1] = r;
      known_upto = q + 1;
    }
  };
  // the next pairs to work on: the first unclaimed ones, in the reference's order, whose two maps and all
  // earlier sources exist (maps finish roughly in index order, so that is rarely a restriction).  A worker takes
  // its share of what can start right now, up to kPairBatch pairs: their ICP / score tails then run as one batch
  // (many small maps: thousands of pairs are ready at once and a launch per pair leaves the chip idle), while a
  // job whose pairs trickle in behind the feature stage keeps dealing them out one by one.
  auto claim_pairs = [&](std::vector<size_t> &out) -> bool {
    out.clear();
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      if (abort) return false;
      bool any_left = false;
      size_t prefix = 0, avail = 0;
      while (prefix < n && ready[prefix]) ++prefix;      // maps [0, prefix) exist
      for (size_t q = 0; q < P; ++q) {
        if (claimed[q]) continue;
        any_left = true;
        if (all[q].first < prefix && ready[all[q].second]) ++avail;
      }
      if (!any_left) return false;
      if (avail) {
        // a batch shares its TARGET (the pairs (i, t) of one t): one descriptor search for the sampled rows of all
        // its sources, and one target grid under every search of the batch
        // How large a batch: round 4 measured take = avail / (share * S) on the headline (16 streams, 120 pairs trickling in behind
        // the feature stage): share 4 / 2 / 1 / 0.5 / 0.25 / 0.125 -> 989 / 990 / 1004 / 1013 / 1021 / 1022 map-pairs/s.  The pair
        // stage's kernels are latency-bound and only four run at a time (hardware queues), so a launch that serves four pairs
        // costs little more queue time than one that serves one; with share 2 most batches were a single pair.
        const double share = pair_share_knob();
        const size_t cap = pair_batch_knob();              // (experiment knob: 8 / 16 / 32 the same)
        const size_t take = std::min(cap, std::max<size_t>(1, (size_t)((double)avail / (share * (double)S))));
        size_t target = n;
        for (size_t q = 0; q < P && out.size() < take; ++q)
          if (!claimed[q] && all[q].first < prefix && ready[all[q].second] && (target == n || all[q].second == target)) {
            target = all[q].second;
            claimed[q] = 1;
            out.push_back(q);
          }
        return true;
      }
      cv.wait(lk);
    }
  };
  auto worker = [&](size_t w) {
    mm3d_ctx *c = cs[w];
    try {
      if (hipSetDevice(c->device) != hipSuccess) throw Error(MM3D_EDEVICE, "hipSetDevice failed");
      while (w < F) {
        size_t i;
        {
          std::lock_guard<std::mutex> lk(mu);
          if (abort || next_map >= n) break;
          i = next_map++;
        }
        // a null / empty map (robot subscribed but no message yet) counts as "no keypoints"
        std::unique_ptr<mm3d_cloud> raw(cloud_from_memory(c, clouds[i].points, clouds[i].points ? clouds[i].n : 0,
                                                          clouds[i].stride ? clouds[i].stride : 16,
                                                          clouds[i].stride ? clouds[i].rgba_offset : 12));
        // owned here until it is published: a throw from map_prepare_impl must not strand the map's buffers
        struct MapFree {
          void operator()(mm3d_map *x) const { delete x->points; delete x->keypoints; delete x->desc; delete x; }
        };
        // the map is this worker's alone until it is published: no waits for other contexts' sake while it is built
        // (Context::settle), one full wait -- which also looks at the recorded error flags -- before it is published
        c->private_objects = true;
        std::unique_ptr<mm3d_map, MapFree> held(map_features_impl(c, raw.get(), params, false));
        map_prepare_impl(c, held.get(), params);        // (ends in that wait)
        raw.reset();
        c->private_objects = false;
        {
          std::lock_guard<std::mutex> lk(mu);
          mm3d_map *m = held.release();
          maps[i] = m;
          ready[i] = 1;
          ctx->last_points[i] = m->points->n;
          ctx->last_keypoints[i] = m->keypoints->n;
          ctx->last_features_s = std::max(ctx->last_features_s, since_start());
        }
        cv.notify_all();
      }
      std::vector<size_t> mine;
      std::vector<PairWork> work;
      while (claim_pairs(mine)) {
        work.clear();
        for (size_t p : mine) {
          const mm3d_map *ms = maps[all[p].first], *mt = maps[all[p].second];
          if (ms->keypoints->n > 0 && mt->keypoints->n > 0) {
            advance_states(p);
            rec[p].source_idx = all[p].first;
            rec[p].target_idx = all[p].second;
            work.push_back(PairWork{ms, mt, &rec[p], state_at[p]});
          }
        }
        if (!work.empty()) pairs_estimate_batch(c, work.data(), work.size(), params);
        for (size_t p : mine)
          if (maps[all[p].first]->keypoints->n > 0 && maps[all[p].second]->keypoints->n > 0) done[p] = 1;
      }
      // (every map and every batch of pairs ended in a wait that brought its results to the host: nothing is in flight here
      // unless a kernel left an error flag to be looked at)
      if (!c->deferred.empty()) c->sync();
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!first_error) first_error = std::current_exception();
      abort = true;
      cv.notify_all();
    }
  };
  std::vector<std::thread> threads;
  for (size_t w = 1; w < S; ++w) threads.emplace_back(worker, w);
  worker(0);
  for (auto &t : threads) t.join();
  // every stream has been synchronised by its worker (or the run was aborted): the maps can go
  for (size_t w = 0; w < S; ++w) (void)stream_wait(cs[w]->stream);
  if (first_error) std::rethrow_exception(first_error);
  // all maps exist now: finish the generator states and check what was assumed about late targets
  advance_states(P);
  bool assumptions_hold = true;
  for (size_t q = 0; q < P; ++q)
    if (assumed_live[q] && maps[all[q].second]->keypoints->n == 0) assumptions_hold = false;
  if (assumptions_hold) {
    ctx->rnd = state_at[P];                             // where the sequential loop leaves the generator
  } else {
    // a target turned out to have no keypoints: the states after that pair were positioned wrongly.
    // Redo the pair loop the reference's way, on the caller's stream.
    ctx->rnd = rnd0;
    std::fill(done.begin(), done.end(), 0);
    for (size_t q = 0; q < P; ++q) {
      const mm3d_map *ms = maps[all[q].first], *mt = maps[all[q].second];
      if (ms->keypoints->n == 0 || mt->keypoints->n == 0) continue;
      pair_estimate_impl(ctx, ms, mt, params, true, &rec[q]);
      rec[q].source_idx = all[q].first;
      rec[q].target_idx = all[q].second;
      done[q] = 1;
    }
    ctx->sync();
  }
  std::vector<mm3d_pair_result> pairs;
  for (size_t p = 0; p < all.size(); ++p)
    if (done[p]) pairs.push_back(rec[p]);
  if (pairs_out) std::memcpy(pairs_out, pairs.data(), pairs.size() * sizeof(mm3d_pair_result));
  if (n_pairs_out) *n_pairs_out = pairs.size();
  const int st = global_transforms(pairs.data(), pairs.size(), params->confidence_threshold, n, out_T, n_out);
  if (st != MM3D_OK) throw Error(st, "computeGlobalTransforms failed");
  ctx->last_total_s = since_start();
}

// ---------------------------------------------------------------- the same job on N processes (one per GPU)
// The N > 1 driver, inside the library like the N = 1 one (estimate_maps_streams): the caller (bench.py,
// one process per GPU) only moves bytes between ranks -- one all-gather of the maps' feature bundles, one
// all-gather of the pair records.  A rank extracts the features of the maps it owns (on its streams),
// receives the other maps' bundles, and estimates the pairs whose TARGET it owns, so each rank builds
// target-side search structures (grids, distance transforms, k-NN operands) for n / world maps only.
// Owners zig-zag over the ranks (0 1 .. w-1 w-1 .. 1 0 0 1 ..): target j has j pairs, and j and its mirror
// image share a rank, which evens the pair counts out.
struct mm3d_shard {
  mm3d_ctx *ctx = nullptr;
  int rank = 0, world = 1;
  size_t n = 0;
  mm3d_params params{};
  std::vector<mm3d_map *> maps;
  ~mm3d_shard()
  {
    for (mm3d_map *x : maps)
      if (x) { delete x->points; delete x->keypoints; delete x->desc; delete x; }
  }
};

int mm3d_shard_map_owner(size_t map, int world)
{
  if (world <= 1) return 0;
  const size_t j = map % (2 * (size_t)world);
  return (int)(j < (size_t)world ? j : 2 * (size_t)world - 1 - j);
}

}  // extern "C" (a template needs C++ linkage)
// run fn(worker index, context, failed) on the context's streams (the caller's thread is worker 0); the first
// exception is rethrown.  `failed` is set when any worker has thrown: the others stop taking work.
template <class Fn>
static void on_streams(mm3d_ctx *ctx, Fn &&fn)
{
  std::vector<mm3d_ctx *> cs{ctx};
  cs.insert(cs.end(), ctx->helpers.begin(), ctx->helpers.end());
  std::mutex mu;
  std::exception_ptr first_error;
  std::atomic<bool> failed{false};
  auto body = [&](size_t w) {
    try {
      if (hipSetDevice(cs[w]->device) != hipSuccess) throw Error(MM3D_EDEVICE, "hipSetDevice failed");
      fn(w, cs[w], failed);
      cs[w]->sync();
    } catch (...) {
      failed.store(true);
      cs[w]->private_objects = false;
      std::lock_guard<std::mutex> lk(mu);
      if (!first_error) first_error = std::current_exception();
    }
  };
  std::vector<std::thread> threads;
  for (size_t w = 1; w < cs.size(); ++w) threads.emplace_back(body, w);
  body(0);
  for (auto &t : threads) t.join();
  for (mm3d_ctx *c : cs) (void)stream_wait(c->stream);
  if (first_error) std::rethrow_exception(first_error);
}
extern "C" {

// (no lock, no device selection: the callers -- mm3d_shard_begin under guarded(), estimate_maps_devices on a device's own thread -- did both)
static mm3d_shard *shard_begin_impl(mm3d_ctx *ctx, const mm3d_cloud_view *clouds, size_t n, const mm3d_params *params, int rank, int world)
{
  std::unique_ptr<mm3d_shard> sh(new mm3d_shard());
  sh->ctx = ctx; sh->rank = rank; sh->world = world; sh->n = n; sh->params = *params;
  sh->maps.assign(n, nullptr);
  std::vector<size_t> mine;
  for (size_t i = 0; i < n; ++i)
    if (mm3d_shard_map_owner(i, world) == rank) mine.push_back(i);
  std::atomic<size_t> next{0};
  on_streams(ctx, [&](size_t, mm3d_ctx *c, const std::atomic<bool> &failed) {
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= mine.size() || failed.load()) break;
      const size_t i = mine[k];
      std::unique_ptr<mm3d_cloud> raw(cloud_from_memory(c, clouds[i].points, clouds[i].points ? clouds[i].n : 0,
                                                        clouds[i].stride ? clouds[i].stride : 16,
                                                        clouds[i].stride ? clouds[i].rgba_offset : 12));
      // nobody else sees the map before on_streams has drained every stream: no waits for other contexts' sake
      c->private_objects = true;
      mm3d_map *m = map_features_impl(c, raw.get(), params, false);
      sh->maps[i] = m;                         // (distinct slots: no lock needed; the shard owns it from here)
      map_prepare_impl(c, m, params);          // this rank is the map's target-side owner; ends in a full wait, which also
      c->private_objects = false;              // looks at the error flags the kernels left
    }
  });
  return sh.release();
}

int mm3d_shard_begin(mm3d_ctx *ctx, const mm3d_cloud_view *clouds, size_t n, const mm3d_params *params, int rank, int world,
                     mm3d_shard **out)
{
  if (!ctx || !params || !out || (n && !clouds) || world < 1 || rank < 0 || rank >= world) return MM3D_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] { *out = shard_begin_impl(ctx, clouds, n, params, rank, world); });
}

int mm3d_shard_bundle_sizes(const mm3d_shard *sh, uint64_t *n_points, uint64_t *n_keypoints)
{
  if (!sh || !n_points || !n_keypoints) return MM3D_EINVAL;
  for (size_t i = 0; i < sh->n; ++i) {
    const bool own = sh->maps[i] && mm3d_shard_map_owner(i, sh->world) == sh->rank;
    n_points[i] = own ? sh->maps[i]->points->n : 0;
    n_keypoints[i] = own ? sh->maps[i]->keypoints->n : 0;
  }
  return MM3D_OK;
}

// A map's bundle (round 5: the source-side structures travel with it).  What a rank does with another rank's map is the SOURCE
// role: ICP / score / SAC-IA scoring read the cloud in its Hilbert query order through its work items, the rand() replay reads
// the keypoints on the host.  Until round 5 a rank rebuilt those orders from the points it had received (two Hilbert sorts and a
// wait per map: 2.3 ms per rank and step at N = 8, with fourteen foreign maps); now the owner -- who has them -- sends them:
//   header (256 B) | points 16 B x P | keypoints 16 B x K | descriptors 4 B x dim x K |
//   points in Hilbert order 16 B x P | their work items 8 B x (P / 64 + 16 386) | the same two for the keypoints
// Every part starts 16-byte aligned and is as large as P and K allow (the sizes are all a receiver knows before the exchange);
// the header says how much of the Hilbert parts is meant, and carries the bounding boxes.  The order a pair's reductions run in
// is then the owner's, i.e. the one-process run's, by construction.
namespace {
struct BundleHeader {
  uint64_t magic, n_points, n_keypoints;
  uint64_t p_finite, p_items, k_finite, k_items;
  uint32_t p_have, k_have;                   // bounding box + Hilbert copy + items are in the bundle
  float p_bmin[3], p_bmax[3], k_bmin[3], k_bmax[3];
  unsigned char pad[256 - 7 * 8 - 2 * 4 - 12 * 4];
};
static_assert(sizeof(BundleHeader) == 256, "bundle header");
constexpr uint64_t kBundleMagic = 0x6d6d33642d623032ull;          // "mm3d-b02"
// A stack object (a bundle header) is the source / destination of an asynchronous copy: nothing may unwind the frame while
// that copy can still be in flight.  Armed until the function's own wait.
struct DrainOnUnwind {
  Context *c;
  bool armed = true;
  ~DrainOnUnwind() { if (armed) (void)stream_wait(c->stream); }
};
struct BundleLayout {
  size_t pts, kp, desc, p_hil, p_items, k_hil, k_items, total, p_item_cap, k_item_cap;
};
size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }
BundleLayout bundle_layout(uint64_t P, uint64_t K, int dim)
{
  BundleLayout L;
  L.p_item_cap = (size_t)P / 64 + 16384 + 2;                        // cloud_hilbert's bound (grid.hip)
  L.k_item_cap = (size_t)K / 64 + 16384 + 2;
  L.pts = sizeof(BundleHeader);
  L.kp = L.pts + (size_t)P * 16;
  L.desc = L.kp + (size_t)K * 16;
  L.p_hil = up16(L.desc + (size_t)K * (size_t)(dim > 0 ? dim : 0) * 4);
  L.p_items = L.p_hil + (size_t)P * 16;
  L.k_hil = up16(L.p_items + L.p_item_cap * sizeof(int2));
  L.k_items = L.k_hil + (size_t)K * 16;
  L.total = up16(L.k_items + L.k_item_cap * sizeof(int2));
  return L;
}
}  // namespace

size_t mm3d_shard_bundle_bytes(uint64_t n_points, uint64_t n_keypoints, int descriptor_type)
{
  return bundle_layout(n_points, n_keypoints, mm3d_descriptor_dim(descriptor_type)).total;
}

int mm3d_shard_pack(mm3d_shard *sh, size_t map, void *dst)
{
  if (!sh || map >= sh->n || !sh->maps[map] || !dst) return MM3D_EINVAL;
  mm3d_ctx *ctx = sh->ctx;
  return guarded(ctx, [&] {
    const mm3d_map *m = sh->maps[map];
    char *d = static_cast<char *>(dst);
    const BundleLayout L = bundle_layout(m->points->n, m->keypoints->n, m->desc->dim);
    // the query orders exist on the owner as soon as it has played the source role once; a map that has not is ordered now
    if (m->points->n) cloud_hilbert(ctx, m->points);
    if (m->keypoints->n) cloud_hilbert(ctx, m->keypoints);
    // (ordinary memory: the pinned arena may wrap under the copies below, and 256 bytes need no pinning)
    BundleHeader header;
    BundleHeader *h = &header;
    std::memset(h, 0, sizeof(*h));
    h->magic = kBundleMagic; h->n_points = m->points->n; h->n_keypoints = m->keypoints->n;
    auto side = [&](const mm3d_cloud *cl, uint64_t &fin, uint64_t &items, uint32_t &have, float *bmin, float *bmax, size_t off_hil,
                    size_t off_items, size_t item_cap) {
      have = (cl->n && cl->have_bbox && cl->hil_pts.get() && (size_t)cl->n_wave_items <= item_cap) ? 1u : 0u;
      if (!have) return;
      fin = cl->n_finite; items = (uint64_t)cl->n_wave_items;
      for (int a = 0; a < 3; ++a) { bmin[a] = cl->bmin[a]; bmax[a] = cl->bmax[a]; }
      if (cl->n_finite) MM3D_HIP(hipMemcpyAsync(d + off_hil, cl->hil_pts.get(), cl->n_finite * 16, hipMemcpyDefault, ctx->stream));
      if (cl->n_wave_items)
        MM3D_HIP(hipMemcpyAsync(d + off_items, cl->wave_items.get(), (size_t)cl->n_wave_items * sizeof(int2), hipMemcpyDefault, ctx->stream));
    };
    DrainOnUnwind drain{ctx};                 // (`header` is read by the copy queued below)
    side(m->points, h->p_finite, h->p_items, h->p_have, h->p_bmin, h->p_bmax, L.p_hil, L.p_items, L.p_item_cap);
    side(m->keypoints, h->k_finite, h->k_items, h->k_have, h->k_bmin, h->k_bmax, L.k_hil, L.k_items, L.k_item_cap);
    MM3D_HIP(hipMemcpyAsync(d, h, sizeof(*h), hipMemcpyDefault, ctx->stream));
    if (m->points->n) MM3D_HIP(hipMemcpyAsync(d + L.pts, m->points->pts.get(), m->points->n * 16, hipMemcpyDefault, ctx->stream));
    if (m->keypoints->n) MM3D_HIP(hipMemcpyAsync(d + L.kp, m->keypoints->pts.get(), m->keypoints->n * 16, hipMemcpyDefault, ctx->stream));
    if (m->desc->n) MM3D_HIP(hipMemcpyAsync(d + L.desc, m->desc->data.get(), m->desc->n * (size_t)m->desc->dim * 4, hipMemcpyDefault, ctx->stream));
    ctx->sync();
    drain.armed = false;
  });
}

// one received bundle -> a map in the source role, on context c (copies, a short wait for the 256-byte header and ONE for the rest; no kernel unless the owner sent no orders)
static mm3d_map *map_from_bundle(mm3d_ctx *c, const void *src, uint64_t n_points, uint64_t n_keypoints, int descriptor_type)
{
  const char *s = static_cast<const char *>(src);
  const int dim = mm3d_descriptor_dim(descriptor_type);
  const BundleLayout L = bundle_layout(n_points, n_keypoints, dim);
  // (into ordinary memory: the pinned arena may wrap under cloud_host() below, and 256 bytes need no pinning)
  BundleHeader header;
  BundleHeader *h = &header;
  std::memset(h, 0, sizeof(*h));
  // the header first, blocking (256 bytes), and checked BEFORE the large copies are queued at sizes the caller supplied
  if (s) {
    MM3D_HIP(hipMemcpyAsync(h, s, sizeof(*h), hipMemcpyDefault, c->stream));
    DrainOnUnwind drain{c};
    c->sync();
    drain.armed = false;
    if (h->magic != kBundleMagic || h->n_points != n_points || h->n_keypoints != n_keypoints)
      throw Error(MM3D_EINVAL, "mm3d_shard_unpack: not a bundle of this library version, or the sizes do not match it");
  }
  std::unique_ptr<mm3d_cloud> pts(cloud_from_memory(c, n_points ? s + L.pts : nullptr, n_points, 16, 12));
  std::unique_ptr<mm3d_cloud> kp(cloud_from_memory(c, n_keypoints ? s + L.kp : nullptr, n_keypoints, 16, 12));
  std::unique_ptr<mm3d_desc> desc(desc_from_memory(c, reinterpret_cast<const float *>(s ? s + L.desc : nullptr), n_keypoints, descriptor_type));
  // the Hilbert parts at their full size (how much of them is meant is in the header, which arrives with the same wait)
  struct Side { DevBuf<float4> hil; DevBuf<int2> items; };
  auto grab = [&](uint64_t n, size_t off_hil, size_t off_items, size_t item_cap) {
    Side sd;
    if (!n || !s) return sd;
    sd.hil = DevBuf<float4>(c, n);
    sd.items = DevBuf<int2>(c, item_cap);
    MM3D_HIP(hipMemcpyAsync(sd.hil.get(), s + off_hil, (size_t)n * 16, hipMemcpyDefault, c->stream));
    MM3D_HIP(hipMemcpyAsync(sd.items.get(), s + off_items, item_cap * sizeof(int2), hipMemcpyDefault, c->stream));
    return sd;
  };
  Side ps = grab(n_points, L.p_hil, L.p_items, L.p_item_cap), ks = grab(n_keypoints, L.k_hil, L.k_items, L.k_item_cap);
  (void)cloud_host(c, kp.get());                      // (the host copy of the keypoints: this is the wait)
  c->sync();
  auto adopt = [&](mm3d_cloud *cl, Side &sd, uint32_t have, uint64_t fin, uint64_t items, const float *bmin, const float *bmax, size_t item_cap) {
    if (!have || !cl->n || fin > cl->n || items > item_cap) return;
    std::lock_guard<std::recursive_mutex> lk(cl->cache_mu);
    cl->have_bbox = true;
    cl->n_finite = (size_t)fin;
    for (int a = 0; a < 3; ++a) { cl->bmin[a] = bmin[a]; cl->bmax[a] = bmax[a]; }
    cl->hil_pts = std::move(sd.hil);
    cl->wave_items = std::move(sd.items);
    cl->n_wave_items = (int)items;
  };
  adopt(pts.get(), ps, h->p_have, h->p_finite, h->p_items, h->p_bmin, h->p_bmax, L.p_item_cap);
  adopt(kp.get(), ks, h->k_have, h->k_finite, h->k_items, h->k_bmin, h->k_bmax, L.k_item_cap);
  // (an owner that sent no orders -- an empty or all-NaN cloud -- leaves them to be built here, as before round 5)
  if (pts->n) cloud_hilbert(c, pts.get());
  if (kp->n) cloud_hilbert(c, kp.get());
  c->sync();
  auto *m = new mm3d_map();
  m->points = pts.release(); m->keypoints = kp.release(); m->desc = desc.release();
  return m;
}

int mm3d_shard_unpack(mm3d_shard *sh, size_t map, const void *src, uint64_t n_points, uint64_t n_keypoints)
{
  if (!sh || map >= sh->n || (!src && (n_points || n_keypoints))) return MM3D_EINVAL;
  if (sh->maps[map]) return MM3D_OK;               // an owned map is already here
  mm3d_ctx *ctx = sh->ctx;
  return guarded(ctx, [&] {
    // source role only: the query orders of ICP / score and of SAC-IA's scoring, and the host copy of the
    // keypoints that the rand() replay reads; target-side structures are the owner's business
    sh->maps[map] = map_from_bundle(ctx, src, n_points, n_keypoints, sh->params.descriptor_type);
  });
}

// every map another rank owns, on the context's streams (at 8 ranks that is 14 of 16 maps per rank)
int mm3d_shard_unpack_many(mm3d_shard *sh, size_t count, const size_t *maps, const void *const *srcs, const uint64_t *n_points,
                           const uint64_t *n_keypoints)
{
  if (!sh || (count && (!maps || !srcs || !n_points || !n_keypoints))) return MM3D_EINVAL;
  mm3d_ctx *ctx = sh->ctx;
  return guarded(ctx, [&] {
    for (size_t k = 0; k < count; ++k)
      if (maps[k] >= sh->n || (!srcs[k] && (n_points[k] || n_keypoints[k]))) throw Error(MM3D_EINVAL, "mm3d_shard_unpack_many: bad item");
    std::atomic<size_t> next{0};
    on_streams(ctx, [&](size_t, mm3d_ctx *c, const std::atomic<bool> &failed) {
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= count || failed.load()) break;
        const size_t i = maps[k];
        if (sh->maps[i]) continue;                    // an owned map is already here
        c->private_objects = true;                    // nobody sees the map before this worker's waits
        mm3d_map *m = map_from_bundle(c, srcs[k], n_points[k], n_keypoints[k], sh->params.descriptor_type);   // source role only, as in mm3d_shard_unpack
        c->private_objects = false;
        sh->maps[i] = m;                              // distinct slots
      }
    });
  });
}

static void shard_pairs_impl(mm3d_shard *sh, mm3d_pair_result *pairs, unsigned char *mine, size_t capacity, size_t *n_pairs)
{
  mm3d_ctx *ctx = sh->ctx;
  for (size_t i = 0; i < sh->n; ++i)
    if (!sh->maps[i]) throw Error(MM3D_EINVAL, "mm3d_shard_pairs: a map has neither been computed here nor unpacked");
  const mm3d_params *params = &sh->params;
  // the live pairs in the reference's order, and the generator state before each of them (the draws of a pair
  // depend on its source keypoints only: every rank replays the whole stream on the host, ~30 us per pair)
  std::vector<std::pair<size_t, size_t>> live;
  for (size_t i = 0; i + 1 < sh->n; ++i)
    for (size_t j = i + 1; j < sh->n; ++j)
      if (sh->maps[i]->keypoints->n > 0 && sh->maps[j]->keypoints->n > 0) live.emplace_back(i, j);
  const size_t P = live.size();
  *n_pairs = P;
  if (P > capacity) throw Error(MM3D_ECAPACITY, "mm3d_shard_pairs: room for every live pair is needed");
  // state_at[q] = the generator before pair q, advanced on demand (under rng_mu) as far as a worker needs it:
  // the first pairs start at once, the table's tail (~30 us of host work per pair) is filled in while they run
  std::vector<GlibcRand> state_at(P + 1, ctx->rnd);
  size_t known_upto = 0;
  std::mutex rng_mu;
  std::vector<const std::vector<float4> *> src_kp(sh->n, nullptr);
  for (size_t i = 0; i < sh->n; ++i) src_kp[i] = &cloud_host(ctx, sh->maps[i]->keypoints);   // (cached at prepare / unpack time)
  auto advance_states = [&](size_t upto) {
    std::lock_guard<std::mutex> lk(rng_mu);
    while (known_upto < upto) {
      GlibcRand r = state_at[known_upto];
      pair_rand_replay(r, params->estimation_method, *src_kp[live[known_upto].first], params->inlier_threshold, params->max_iterations);
      state_at[++known_upto] = r;
    }
  };
  std::vector<size_t> todo;
  for (size_t q = 0; q < P; ++q) {
    std::memset(&pairs[q], 0, sizeof(mm3d_pair_result));
    pairs[q].source_idx = live[q].first;
    pairs[q].target_idx = live[q].second;
    mine[q] = mm3d_shard_map_owner(live[q].second, sh->world) == sh->rank ? 1 : 0;
    if (mine[q]) todo.push_back(q);
  }
  // batches of pairs with the same target (pairs_estimate_batch), at most kPairBatch of them and not so many that
  // a stream runs dry: every map exists already, so the whole list can be cut up front
  const size_t S = ctx->helpers.size() + 1;
  const double share = pair_share_knob();                  // (as claim_pairs above)
  const size_t take = std::min(pair_batch_knob(), std::max<size_t>(1, (size_t)((double)todo.size() / (share * (double)S))));
  std::stable_sort(todo.begin(), todo.end(), [&](size_t a, size_t b) { return live[a].second < live[b].second; });
  std::vector<std::pair<size_t, size_t>> batches;           // [first, last) into todo
  for (size_t a = 0; a < todo.size();) {
    size_t b = a + 1;
    while (b < todo.size() && b - a < take && live[todo[b]].second == live[todo[a]].second) ++b;
    batches.emplace_back(a, b);
    a = b;
  }
  std::atomic<size_t> next{0};
  on_streams(ctx, [&](size_t, mm3d_ctx *c, const std::atomic<bool> &failed) {
    std::vector<PairWork> work;
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= batches.size() || failed.load()) break;
      work.clear();
      for (size_t e = batches[k].first; e < batches[k].second; ++e) {
        const size_t q = todo[e];
        advance_states(q);
        work.push_back(PairWork{sh->maps[live[q].first], sh->maps[live[q].second], &pairs[q], state_at[q]});
      }
      pairs_estimate_batch(c, work.data(), work.size(), params);
    }
  });
  advance_states(P);
  ctx->rnd = state_at[P];                       // where the reference's sequential loop leaves the generator
}

int mm3d_shard_pairs(mm3d_shard *sh, mm3d_pair_result *pairs, unsigned char *mine, size_t capacity, size_t *n_pairs)
{
  if (!sh || !n_pairs || !pairs || !mine) return MM3D_EINVAL;
  return guarded(sh->ctx, [&] { shard_pairs_impl(sh, pairs, mine, capacity, n_pairs); });
}

void mm3d_shard_end(mm3d_shard *sh)
{
  if (!sh) return;
  mm3d_ctx *ctx = sh->ctx;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)stream_wait(ctx->stream);
    for (mm3d_ctx *h : ctx->helpers) (void)stream_wait(h->stream);
  }
  delete sh;
}

// the reference's two loops on ONE stream, in the reference's order (mm3d_set_streams(ctx, 1), the default)
static void estimate_maps_sequential(mm3d_ctx *ctx, const mm3d_cloud_view *clouds, size_t n, const mm3d_params *params, float *out_T,
                                     size_t *n_out, mm3d_pair_result *pairs_out, size_t *n_pairs_out)
{
  const auto t_start = std::chrono::steady_clock::now();
  auto since_start = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  ctx->last_points.assign(n, 0);
  ctx->last_keypoints.assign(n, 0);
  std::vector<std::unique_ptr<mm3d_map, std::function<void(mm3d_map *)>>> maps;
  auto del = [](mm3d_map *m) { if (m) { delete m->points; delete m->keypoints; delete m->desc; delete m; } };
  for (size_t i = 0; i < n; ++i) {
    // a null / empty map (robot subscribed but no message yet) counts as "no keypoints"
    std::unique_ptr<mm3d_cloud> raw(cloud_from_memory(ctx, clouds[i].points, clouds[i].points ? clouds[i].n : 0,
                                                      clouds[i].stride ? clouds[i].stride : 16,
                                                      clouds[i].stride ? clouds[i].rgba_offset : 12));
    maps.emplace_back(map_features_impl(ctx, raw.get(), params), del);
    map_prepare_impl(ctx, maps.back().get(), params);   // search structures and k-NN target operands, once per map
    ctx->last_points[i] = maps.back()->points->n;
    ctx->last_keypoints[i] = maps.back()->keypoints->n;
  }
  ctx->last_features_s = since_start();
  std::vector<mm3d_pair_result> pairs;
  for (size_t i = 0; i + 1 < n; ++i)
    for (size_t j = i + 1; j < n; ++j)
      if (maps[i]->keypoints->n > 0 && maps[j]->keypoints->n > 0) {
        mm3d_pair_result r;
        std::memset(&r, 0, sizeof(r));
        r.source_idx = i; r.target_idx = j;
        pairs.push_back(r);
      }
  for (auto &r : pairs) pair_estimate_impl(ctx, maps[r.source_idx].get(), maps[r.target_idx].get(), params, true, &r);
  if (pairs_out) std::memcpy(pairs_out, pairs.data(), pairs.size() * sizeof(mm3d_pair_result));
  if (n_pairs_out) *n_pairs_out = pairs.size();
  int st = global_transforms(pairs.data(), pairs.size(), params->confidence_threshold, n, out_T, n_out);
  if (st != MM3D_OK) throw Error(st, "computeGlobalTransforms failed");
  ctx->last_total_s = since_start();
}

// ---------------------------------------------------------------- the same job on N devices of ONE process
// estimateMapsTransforms behind the reference's own entry point on a device list (mm3d_create_devices): the reference's
// caller is one process -- a ROS timer callback, R/src/map_merge_node.cpp:133-153 -- and cannot be relaunched under torchrun.
// One host thread per device drives that device's root context and its streams through the mm3d_shard_* scheme:
//   1. features of the maps the device owns (zig-zag ownership, shard_begin_impl) incl. their target-side structures;
//   2. when every device is done, each PULLS the other maps' bundles and source-side structures from their owners with
//      hipMemcpyPeerAsync (devices.cpp::cloud_clone_from_peer), dealt to its streams -- xGMI is point to point, every
//      device reads from up to seven peers at once; nothing is recomputed (the multi-process form re-builds the Hilbert
//      orders from the bundles: 2.3 ms per rank at N = 8);
//   3. the pairs whose TARGET the device owns (shard_pairs_impl), every device replaying the reference's single rand() stream;
//   4. ONE RCCL all-gather of the 104-byte pair records (devices.cpp::gather_pair_records), then the pose graph on the host.
// Same bits as one device: the ownership only decides where a map or a pair is computed.
namespace {
// a barrier the device threads can leave through a failure: whoever throws releases the others, who then throw too
struct FailBarrier {
  std::mutex mu;
  std::condition_variable cv;
  size_t n, waiting = 0, generation = 0;
  bool failed = false;
  explicit FailBarrier(size_t n_) : n(n_) {}
  void wait()
  {
    std::unique_lock<std::mutex> lk(mu);
    if (failed) throw Error(MM3D_EDEVICE, "another device failed");
    const size_t gen = generation;
    if (++waiting == n) { waiting = 0; ++generation; cv.notify_all(); return; }
    cv.wait(lk, [&] { return failed || generation != gen; });
    if (failed) throw Error(MM3D_EDEVICE, "another device failed");
  }
  void fail()
  {
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    cv.notify_all();
  }
};
}  // namespace

// ---- one process, several devices --------------------------------------------------------------------------------------
// What every device of a run shares on the host (one process: one address space).
struct DevicesRun {
  std::vector<mm3d_ctx *> roots;
  size_t D = 0, n = 0;
  std::vector<std::unique_ptr<mm3d_shard>> sh;            // per device: its own maps and its copies of the others'
  std::vector<std::vector<mm3d_pair_result>> rec;         // per device, by live-pair number
  std::vector<std::vector<unsigned char>> mine;
  std::vector<size_t> np;
  std::vector<double> t_feat, t_exch, t_pairs;
  std::chrono::steady_clock::time_point t_start;
  double since_start() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); }
};

// Round 5's form: three lock-step stages -- every device's features, barrier, every device pulls every other map, every
// device replays the WHOLE rand() stream for itself and runs its pairs, barrier.  Kept as the fallback of the pipelined form
// below (a map without keypoints falsifies its assumptions) and as its A/B (MM3D_DEVICES_STAGED=1).
static void devices_run_staged(mm3d_ctx *ctx, DevicesRun &R, const mm3d_cloud_view *clouds, const mm3d_params *params, size_t max_pairs)
{
  std::vector<mm3d_ctx *> &roots = R.roots;
  const size_t D = R.D, n = R.n;
  auto &sh = R.sh; auto &rec = R.rec; auto &mine = R.mine; auto &np = R.np;
  auto &t_feat = R.t_feat; auto &t_exch = R.t_exch; auto &t_pairs = R.t_pairs;
  auto since_start = [&] { return R.since_start(); };
  FailBarrier bar(D);
  std::mutex err_mu;
  std::exception_ptr first_error;
  for (size_t d = 1; d < D; ++d) roots[d]->rnd = ctx->rnd;        // every device replays the one rand() stream from the caller's state
  auto body = [&](size_t d) {
    mm3d_ctx *root = roots[d];
    // (the peers are reached only through this call, which holds the first context's lock: theirs is taken for the helpers'
    // sake of invariants only -- nothing else can be using them)
    std::unique_lock<std::mutex> peer_lock;
    if (d > 0) peer_lock = std::unique_lock<std::mutex>(root->mu);
    try {
      if (hipSetDevice(root->device) != hipSuccess) throw Error(MM3D_EDEVICE, "hipSetDevice failed");
      sh[d].reset(shard_begin_impl(root, clouds, n, params, (int)d, (int)D));
      t_feat[d] = since_start();
      bar.wait();                                         // every owner's maps exist and its streams are drained
      // the other devices' maps: bundle + source-side structures straight from the owner's memory, dealt to this device's streams
      std::vector<size_t> theirs;
      for (size_t i = 0; i < n; ++i)
        if (!sh[d]->maps[i]) theirs.push_back(i);
      std::atomic<size_t> next{0};
      on_streams(root, [&](size_t, mm3d_ctx *c, const std::atomic<bool> &failed) {
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= theirs.size() || failed.load()) break;
          const size_t i = theirs[k];
          const size_t o = (size_t)mm3d_shard_map_owner(i, (int)D);
          const mm3d_map *src = sh[o]->maps[i];
          const int src_dev = roots[o]->device;
          c->private_objects = true;
          std::unique_ptr<mm3d_cloud> pts(cloud_clone_from_peer(c, src->points, src_dev));
          std::unique_ptr<mm3d_cloud> kp(cloud_clone_from_peer(c, src->keypoints, src_dev));
          std::unique_ptr<mm3d_desc> desc(desc_clone_from_peer(c, src->desc, src_dev));
          // source role only (as mm3d_shard_unpack): whatever of the query orders / host copy did not come with the clone
          if (pts->n) cloud_hilbert(c, pts.get());
          if (kp->n) cloud_hilbert(c, kp.get());
          (void)cloud_host(c, kp.get());
          c->private_objects = false;
          c->sync();
          auto *m = new mm3d_map();
          m->points = pts.release(); m->keypoints = kp.release(); m->desc = desc.release();
          sh[d]->maps[i] = m;                             // distinct slots; nobody reads another device's non-owned slots
        }
      });
      t_exch[d] = since_start();
      shard_pairs_impl(sh[d].get(), rec[d].data(), mine[d].data(), max_pairs, &np[d]);
      t_pairs[d] = since_start();
      static const bool dbg = [] { const char *e = getenv("MM3D_DEVICES_DEBUG"); return e && atoi(e); }();
      if (dbg) fprintf(stderr, "mm3d devices (staged): dev%zu features %.2f ms, pulls done %.2f ms, pairs done %.2f ms (its own replay of every pair inside)\n", d,
                       1e3 * t_feat[d], 1e3 * t_exch[d], 1e3 * t_pairs[d]);
      // an owner's maps are read by its peers' pulls: nobody leaves (and nothing is freed) before everybody has pulled
      bar.wait();
    } catch (...) {
      bar.fail();
      std::lock_guard<std::mutex> lk(err_mu);
      if (!first_error) first_error = std::current_exception();
    }
  };
  {
    std::vector<std::thread> threads;
    for (size_t d = 1; d < D; ++d) threads.emplace_back(body, d);
    body(0);
    for (auto &t : threads) t.join();
  }
  (void)hipSetDevice(ctx->device);
  if (first_error) {
    for (size_t d = 0; d < D; ++d) {                      // drain before the shards (and their maps) go
      (void)hipSetDevice(roots[d]->device);
      (void)stream_wait(roots[d]->stream);
      for (mm3d_ctx *h : roots[d]->helpers) (void)stream_wait(h->stream);
    }
    (void)hipSetDevice(ctx->device);
    std::rethrow_exception(first_error);
  }
}

// Round 6: the same split -- features by owner, pairs by target owner, peer copies in between -- WITHOUT the lock-step and
// WITHOUT D private replays of the rand() stream:
//   * ONE table of generator states (state_at[q] = the state before pair q of the reference's loop), filled once by one host
//     thread as the sources' keypoints appear (the draws of a pair depend on its source keypoints only) and read by every
//     device.  Before, each device replayed all n (n - 1) / 2 pairs itself: 8.5 us x 2 016 pairs = 17 ms of serial host work
//     per device on 64 x 50 k maps, the size of a device's whole pair stage at N = 8 ("host RNG replay dominates").
//   * per-map readiness: a map is published (a flag under the run's mutex, behind its owner's full stream wait) the moment its
//     owner has finished it; any device pulls it then (hipMemcpyPeerAsync on its own stream) and starts a pair as soon as the
//     pair's two maps are on the device and the pair's state is in the table.  Only the end of the call waits for everybody
//     (an owner's maps are read by its peers' pulls until then).
// The table assumes that a target which does not exist yet will have keypoints (as estimate_maps_streams does); the
// assumptions are checked when every map exists.  Returns false when one was wrong: the caller runs the staged form.
static bool devices_run_pipelined(mm3d_ctx *ctx, DevicesRun &R, const mm3d_cloud_view *clouds, const mm3d_params *params)
{
  std::vector<mm3d_ctx *> &roots = R.roots;
  const size_t D = R.D, n = R.n;
  std::vector<std::pair<size_t, size_t>> all;
  for (size_t i = 0; i + 1 < n; ++i)
    for (size_t j = i + 1; j < n; ++j) all.emplace_back(i, j);
  const size_t P = all.size();
  for (size_t d = 0; d < D; ++d) {
    R.sh[d].reset(new mm3d_shard());
    R.sh[d]->ctx = roots[d]; R.sh[d]->rank = (int)d; R.sh[d]->world = (int)D; R.sh[d]->n = n; R.sh[d]->params = *params;
    R.sh[d]->maps.assign(n, nullptr);
  }
  std::mutex mu;                                          // guards everything below but the table
  std::condition_variable cv;
  std::vector<char> ready(n, 0);                          // map i is published by its owner
  std::vector<std::vector<char>> pull_claimed(D, std::vector<char>(n, 0)), have(D, std::vector<char>(n, 0));
  std::vector<char> claimed(P, 0);
  std::vector<std::vector<size_t>> own_maps(D), todo(D);  // per device: the maps it owns; the pairs whose target it owns
  std::vector<size_t> next_own(D, 0);
  for (size_t i = 0; i < n; ++i) own_maps[(size_t)mm3d_shard_map_owner(i, (int)D)].push_back(i);
  for (size_t q = 0; q < P; ++q) todo[(size_t)mm3d_shard_map_owner(all[q].second, (int)D)].push_back(q);
  bool abort = false;
  std::exception_ptr first_error;
  std::vector<mm3d_pair_result> rec_all(P);
  std::vector<char> done(P, 0);
  // the table
  std::vector<GlibcRand> state_at(P + 1, ctx->rnd);
  std::vector<char> assumed_live(P, 0);
  std::atomic<size_t> known_upto{0};                      // state_at[0 .. known_upto] are final
  double fill_busy_s = 0.0, fill_done_s = 0.0;            // (the filler thread's alone until it is joined)
  auto fill_table = [&] {
    try {
      for (size_t q = 0; q < P; ++q) {
        const size_t a = all[q].first, b = all[q].second;
        const mm3d_map *ma = nullptr;
        bool live = false;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return abort || ready[a]; });
          if (abort) return;
          ma = R.sh[(size_t)mm3d_shard_map_owner(a, (int)D)]->maps[a];
          live = ma->keypoints->n > 0;
          if (live) {
            if (ready[b]) live = R.sh[(size_t)mm3d_shard_map_owner(b, (int)D)]->maps[b]->keypoints->n > 0;
            else assumed_live[q] = 1;
          }
        }
        GlibcRand r = state_at[q];
        const auto tb = std::chrono::steady_clock::now();
        // (the host copy of an owner's keypoints was made when the map was prepared: no device is touched here)
        if (live) pair_rand_replay(r, params->estimation_method, cloud_host(roots[(size_t)mm3d_shard_map_owner(a, (int)D)], ma->keypoints),
                                   params->inlier_threshold, params->max_iterations);
        fill_busy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count();
        fill_done_s = R.since_start();
        state_at[q + 1] = r;
        known_upto.store(q + 1, std::memory_order_release);
        if ((q & 7) == 7 || q + 1 == P) { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
      }
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!first_error) first_error = std::current_exception();
      abort = true;
      cv.notify_all();
    }
  };
  struct MapFree {
    void operator()(mm3d_map *x) const { delete x->points; delete x->keypoints; delete x->desc; delete x; }
  };
  auto device_body = [&](size_t d) {
    mm3d_ctx *root = roots[d];
    std::unique_lock<std::mutex> peer_lock;
    if (d > 0) peer_lock = std::unique_lock<std::mutex>(root->mu);
    const size_t S = root->helpers.size() + 1;
    try {
      if (hipSetDevice(root->device) != hipSuccess) throw Error(MM3D_EDEVICE, "hipSetDevice failed");
      on_streams(root, [&](size_t, mm3d_ctx *c, const std::atomic<bool> &failed) {
        // 1. this device's own maps, in index order
        for (;;) {
          size_t i;
          {
            std::lock_guard<std::mutex> lk(mu);
            if (abort || failed.load() || next_own[d] >= own_maps[d].size()) break;
            i = own_maps[d][next_own[d]++];
          }
          std::unique_ptr<mm3d_cloud> raw(cloud_from_memory(c, clouds[i].points, clouds[i].points ? clouds[i].n : 0,
                                                            clouds[i].stride ? clouds[i].stride : 16,
                                                            clouds[i].stride ? clouds[i].rgba_offset : 12));
          c->private_objects = true;
          std::unique_ptr<mm3d_map, MapFree> held(map_features_impl(c, raw.get(), params, false));
          map_prepare_impl(c, held.get(), params);          // this device is the map's target-side owner; ends in a full wait:
          raw.reset();                                      // the map is complete in this device's memory BEFORE anybody is told
          c->private_objects = false;
          {
            std::lock_guard<std::mutex> lk(mu);
            R.sh[d]->maps[i] = held.release();
            have[d][i] = 1;
            ready[i] = 1;
            R.t_feat[d] = std::max(R.t_feat[d], R.since_start());
          }
          cv.notify_all();
        }
        // 2. pulls and pairs, whatever can start
        std::vector<size_t> batch;
        std::vector<PairWork> work;
        for (;;) {
          size_t pull = n;
          batch.clear();
          {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
              if (abort || failed.load()) return;
              // a published map this device does not hold yet: first, it unlocks pairs
              for (size_t i = 0; i < n && pull == n; ++i)
                if (ready[i] && !have[d][i] && !pull_claimed[d][i]) pull = i;
              if (pull != n) { pull_claimed[d][pull] = 1; break; }
              // pairs of this device whose two maps are here and whose state is in the table: a batch shares its target
              const size_t known = known_upto.load(std::memory_order_acquire);
              size_t avail = 0, left = 0;
              for (size_t q : todo[d]) {
                if (claimed[q]) continue;
                ++left;
                if (q <= known && have[d][all[q].first] && have[d][all[q].second]) ++avail;
              }
              if (avail) {
                const size_t take = std::min(pair_batch_knob(), std::max<size_t>(1, (size_t)((double)avail / (pair_share_knob() * (double)S))));
                size_t target = n;
                for (size_t q : todo[d]) {
                  if (batch.size() >= take) break;
                  if (claimed[q] || q > known || !have[d][all[q].first] || !have[d][all[q].second]) continue;
                  if (target != n && all[q].second != target) continue;
                  target = all[q].second;
                  claimed[q] = 1;
                  batch.push_back(q);
                }
                break;
              }
              bool pulls_left = false;
              for (size_t i = 0; i < n; ++i) pulls_left = pulls_left || (!have[d][i] && !pull_claimed[d][i]);
              if (!left && !pulls_left) return;           // nothing more for this worker, ever
              cv.wait(lk);
            }
          }
          if (pull != n) {
            const size_t o = (size_t)mm3d_shard_map_owner(pull, (int)D);
            const mm3d_map *src = R.sh[o]->maps[pull];    // (published: complete, and not freed before every thread has joined)
            const int src_dev = roots[o]->device;
            c->private_objects = true;
            std::unique_ptr<mm3d_cloud> pts(cloud_clone_from_peer(c, src->points, src_dev));
            std::unique_ptr<mm3d_cloud> kp(cloud_clone_from_peer(c, src->keypoints, src_dev));
            std::unique_ptr<mm3d_desc> desc(desc_clone_from_peer(c, src->desc, src_dev));
            if (pts->n) cloud_hilbert(c, pts.get());
            if (kp->n) cloud_hilbert(c, kp.get());
            (void)cloud_host(c, kp.get());
            c->private_objects = false;
            c->sync();
            auto *m = new mm3d_map();
            m->points = pts.release(); m->keypoints = kp.release(); m->desc = desc.release();
            {
              std::lock_guard<std::mutex> lk(mu);
              R.sh[d]->maps[pull] = m;
              have[d][pull] = 1;
              R.t_exch[d] = std::max(R.t_exch[d], R.since_start());
            }
            cv.notify_all();
            continue;
          }
          work.clear();
          for (size_t q : batch) {
            const mm3d_map *ms = R.sh[d]->maps[all[q].first], *mt = R.sh[d]->maps[all[q].second];
            if (ms->keypoints->n > 0 && mt->keypoints->n > 0) {
              rec_all[q].source_idx = all[q].first;
              rec_all[q].target_idx = all[q].second;
              work.push_back(PairWork{ms, mt, &rec_all[q], state_at[q]});
              done[q] = 1;                                // (distinct q per worker; read after the joins)
            }
          }
          if (!work.empty()) pairs_estimate_batch(c, work.data(), work.size(), params);
          { std::lock_guard<std::mutex> lk(mu); R.t_pairs[d] = std::max(R.t_pairs[d], R.since_start()); }
        }
      });
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!first_error) first_error = std::current_exception();
      abort = true;
      cv.notify_all();
    }
  };
  {
    std::thread filler(fill_table);
    std::vector<std::thread> threads;
    for (size_t d = 1; d < D; ++d) threads.emplace_back(device_body, d);
    device_body(0);
    for (auto &t : threads) t.join();
    { std::lock_guard<std::mutex> lk(mu); if (first_error) abort = true; }
    cv.notify_all();
    filler.join();
  }
  (void)hipSetDevice(ctx->device);
  if (first_error) {
    for (size_t d = 0; d < D; ++d) {                      // drain before the shards (and their maps) go
      (void)hipSetDevice(roots[d]->device);
      (void)stream_wait(roots[d]->stream);
      for (mm3d_ctx *h : roots[d]->helpers) (void)stream_wait(h->stream);
    }
    (void)hipSetDevice(ctx->device);
    std::rethrow_exception(first_error);
  }
  static const bool dbg = [] { const char *e = getenv("MM3D_DEVICES_DEBUG"); return e && atoi(e); }();
  if (dbg) {
    fprintf(stderr, "mm3d devices (pipelined): %zu devices, %zu maps, %zu pairs; ONE rand() table: %.2f ms of replay on one host thread, complete %.2f ms into the call "
            "(0 ms of replay on the devices' threads);", D, n, P, 1e3 * fill_busy_s, 1e3 * fill_done_s);
    for (size_t d = 0; d < D; ++d) fprintf(stderr, " dev%zu last map %.2f last pull %.2f last pair %.2f ms;", d, 1e3 * R.t_feat[d], 1e3 * R.t_exch[d], 1e3 * R.t_pairs[d]);
    fprintf(stderr, "\n");
  }
  // every map exists: were the table's assumptions right?
  for (size_t q = 0; q < P; ++q)
    if (assumed_live[q] && R.sh[0]->maps[all[q].second]->keypoints->n == 0) return false;
  ctx->rnd = state_at[P];                                 // where the reference's sequential loop leaves the generator
  // the live pairs in the reference's order, per executing device (what the gather sends)
  size_t nl = 0;
  for (size_t q = 0; q < P; ++q) {
    if (!done[q]) continue;
    const size_t d = (size_t)mm3d_shard_map_owner(all[q].second, (int)D);
    for (size_t e = 0; e < D; ++e) {
      R.rec[e][nl] = mm3d_pair_result{};
      R.rec[e][nl].source_idx = all[q].first; R.rec[e][nl].target_idx = all[q].second;
      R.mine[e][nl] = e == d ? 1 : 0;
    }
    R.rec[d][nl] = rec_all[q];
    ++nl;
  }
  for (size_t d = 0; d < D; ++d) R.np[d] = nl;
  return true;
}

static void estimate_maps_devices(mm3d_ctx *ctx, const mm3d_cloud_view *clouds, size_t n, const mm3d_params *params, float *out_T,
                                  size_t *n_out, mm3d_pair_result *pairs_out, size_t *n_pairs_out)
{
  std::vector<mm3d_ctx *> roots{ctx};
  roots.insert(roots.end(), ctx->peers.begin(), ctx->peers.end());
  const size_t D = roots.size();
  const size_t max_pairs = n * (n - 1) / 2;
  if (D == 1) {
    // a list of one device: nothing to shard, so the job runs as on a plain context (pipelined over the streams, not in
    // barriered stages) -- and its pair records still travel through the communicator's all-gather (a world of one), so that
    // the collective of the path is exercised wherever a device list is used
    std::vector<mm3d_pair_result> local(std::max<size_t>(max_pairs, 1));
    size_t np = 0;
    if (!ctx->helpers.empty()) estimate_maps_streams(ctx, clouds, n, params, out_T, n_out, local.data(), &np);
    else estimate_maps_sequential(ctx, clouds, n, params, out_T, n_out, local.data(), &np);
    const double t_before = ctx->last_total_s;
    std::vector<std::vector<mm3d_pair_result>> send(1);
    send[0].assign(local.begin(), local.begin() + (ptrdiff_t)np);
    std::vector<mm3d_pair_result> gathered;
    ctx->last_gather_s = gather_pair_records(ctx->device_set, roots, send, np, gathered);
    ctx->last_exchange_s = ctx->last_features_s;
    ctx->last_pairs_s = t_before;
    if (pairs_out && np) std::memcpy(pairs_out, gathered.data(), np * sizeof(mm3d_pair_result));
    if (n_pairs_out) *n_pairs_out = np;
    const int st = global_transforms(gathered.data(), np, params->confidence_threshold, n, out_T, n_out);   // (from what the gather delivered)
    if (st != MM3D_OK) throw Error(st, "computeGlobalTransforms failed");
    ctx->last_total_s = t_before + ctx->last_gather_s;
    return;
  }
  DevicesRun R;
  R.roots = roots; R.D = D; R.n = n;
  R.sh.resize(D);
  R.rec.assign(D, std::vector<mm3d_pair_result>(max_pairs));
  R.mine.assign(D, std::vector<unsigned char>(max_pairs, 0));
  R.np.assign(D, 0);
  R.t_feat.assign(D, 0.0); R.t_exch.assign(D, 0.0); R.t_pairs.assign(D, 0.0);
  R.t_start = std::chrono::steady_clock::now();
  static const bool staged_only = [] { const char *e = getenv("MM3D_DEVICES_STAGED"); return e && atoi(e); }();
  const GlibcRand rnd0 = ctx->rnd;
  bool ran = false;
  if (!staged_only) {
    ran = devices_run_pipelined(ctx, R, clouds, params);
    if (!ran) {                                           // a map without keypoints: the table was positioned wrongly after it
      for (size_t d = 0; d < D; ++d) { (void)hipSetDevice(roots[d]->device); R.sh[d].reset(); }
      (void)hipSetDevice(ctx->device);
      ctx->rnd = rnd0;
    }
  }
  if (!ran) devices_run_staged(ctx, R, clouds, params, max_pairs);
  auto &sh = R.sh; auto &rec = R.rec; auto &mine = R.mine; auto &np = R.np;
  auto &t_feat = R.t_feat; auto &t_exch = R.t_exch; auto &t_pairs = R.t_pairs;
  auto since_start = [&] { return R.since_start(); };
  ctx->last_points.assign(n, 0);
  ctx->last_keypoints.assign(n, 0);
  for (size_t i = 0; i < n; ++i) {
    ctx->last_points[i] = sh[0]->maps[i]->points->n;
    ctx->last_keypoints[i] = sh[0]->maps[i]->keypoints->n;
  }
  ctx->last_features_s = *std::max_element(t_feat.begin(), t_feat.end());
  ctx->last_exchange_s = *std::max_element(t_exch.begin(), t_exch.end());
  ctx->last_pairs_s = *std::max_element(t_pairs.begin(), t_pairs.end());
  // the gather: rank d sends the records of its own pairs, in pair order, padded to the largest rank's count
  const size_t P = np[0];
  for (size_t d = 1; d < D; ++d)
    if (np[d] != P) throw Error(MM3D_EDEVICE, "estimate_maps_devices: the devices disagree on the live pairs");
  std::vector<std::vector<mm3d_pair_result>> send(D);
  std::vector<std::vector<size_t>> which(D);
  for (size_t d = 0; d < D; ++d)
    for (size_t q = 0; q < P; ++q)
      if (mine[d][q]) { send[d].push_back(rec[d][q]); which[d].push_back(q); }
  size_t slots = 0;
  for (size_t d = 0; d < D; ++d) slots = std::max(slots, send[d].size());
  std::vector<mm3d_pair_result> gathered;
  ctx->last_gather_s = gather_pair_records(ctx->device_set, roots, send, slots, gathered);
  std::vector<mm3d_pair_result> pairs(P);
  std::vector<char> seen(P, 0);
  for (size_t d = 0; d < D; ++d)
    for (size_t k = 0; k < which[d].size(); ++k) {
      pairs[which[d][k]] = gathered[d * slots + k];
      seen[which[d][k]] = 1;
    }
  for (size_t q = 0; q < P; ++q)
    if (!seen[q]) throw Error(MM3D_EDEVICE, "estimate_maps_devices: a pair has no owner");
  // the shards (maps on every device) go now; every stream was drained by its device's thread
  for (size_t d = 0; d < D; ++d) {
    (void)hipSetDevice(roots[d]->device);
    sh[d].reset();
  }
  (void)hipSetDevice(ctx->device);
  if (pairs_out) std::memcpy(pairs_out, pairs.data(), pairs.size() * sizeof(mm3d_pair_result));
  if (n_pairs_out) *n_pairs_out = pairs.size();
  const int st = global_transforms(pairs.data(), pairs.size(), params->confidence_threshold, n, out_T, n_out);
  if (st != MM3D_OK) throw Error(st, "computeGlobalTransforms failed");
  ctx->last_total_s = since_start();
}

// ---------------------------------------------------------------- map_merging.h
int mm3d_estimate_maps_transforms(mm3d_ctx *ctx, const mm3d_cloud_view *clouds, size_t n, const mm3d_params *params,
                                  float *out_T, size_t *n_out, mm3d_pair_result *pairs_out, size_t *n_pairs_out)
{
  if (!params || !n_out || (n && (!clouds || !out_T))) return MM3D_EINVAL;
  if (n_pairs_out) *n_pairs_out = 0;
  *n_out = 0;
  if (n == 0) return MM3D_OK;                       // {} -> {}  (map_merging.cpp:192-194)
  if (n == 1) {                                     // one cloud -> {Identity}, the cloud is not touched (:195-197)
    std::memset(out_T, 0, sizeof(float) * 16);
    out_T[0] = out_T[5] = out_T[10] = out_T[15] = 1.0f;
    *n_out = 1;
    return MM3D_OK;
  }
  return guarded(ctx, [&] {
    if (ctx->device_set) {                             // a device list (mm3d_create_devices), even of one device
      estimate_maps_devices(ctx, clouds, n, params, out_T, n_out, pairs_out, n_pairs_out);
      return;
    }
    if (!ctx->helpers.empty()) {
      estimate_maps_streams(ctx, clouds, n, params, out_T, n_out, pairs_out, n_pairs_out);
      return;
    }
    estimate_maps_sequential(ctx, clouds, n, params, out_T, n_out, pairs_out, n_pairs_out);
  });
}

int mm3d_compose_maps(mm3d_ctx *ctx, const mm3d_cloud *const *clouds, size_t n, const float *transforms, size_t n_transforms,
                      double resolution, mm3d_cloud **out)
{
  if (!out) return MM3D_EINVAL;
  *out = nullptr;
  if (n == 0) return MM3D_OK;                       // nullptr (map_merging.cpp:281-283)
  if (n != n_transforms) {                          // the reference throws (map_merging.cpp:285-288)
    if (ctx) ctx->err = "composeMaps: clouds and transforms size must be the same.";
    return MM3D_EINVAL;
  }
  if (!clouds || !transforms) return MM3D_EINVAL;
  return guarded(ctx, [&] {
    std::unique_ptr<mm3d_cloud> cat(transform_concat(ctx, clouds, n, transforms));
    *out = downsample(ctx, cat.get(), resolution);
  });
}

// ---------------------------------------------------------------- measurement
int mm3d_profile_enable(mm3d_ctx *ctx, int on)
{
  return guarded(ctx, [&] {
    ctx->prof_resolve();
    ctx->prof_on = on != 0;
    for (mm3d_ctx *h : ctx->helpers) { h->prof_resolve(); h->prof_on = on != 0; }   // mm3d_set_streams helpers
    for (mm3d_ctx *p : ctx->peers) {                                                // mm3d_create_devices: the other devices
      (void)hipSetDevice(p->device);
      p->prof_resolve();
      p->prof_on = on != 0;
      for (mm3d_ctx *h : p->helpers) { h->prof_resolve(); h->prof_on = on != 0; }
    }
    (void)hipSetDevice(ctx->device);
  });
}
void mm3d_profile_reset(mm3d_ctx *ctx)
{
  if (!ctx) return;
  std::lock_guard<std::mutex> lock(ctx->mu);
  auto reset_one = [](mm3d_ctx *r) {
    (void)hipSetDevice(r->device);
    try { r->prof_resolve(); } catch (...) {}
    for (auto &e : r->prof) e = ProfEntry();
    for (mm3d_ctx *h : r->helpers) {
      try { h->prof_resolve(); } catch (...) {}
      for (auto &e : h->prof) e = ProfEntry();
    }
  };
  reset_one(ctx);
  for (mm3d_ctx *p : ctx->peers) reset_one(p);
  (void)hipSetDevice(ctx->device);
}
int mm3d_profile_count(mm3d_ctx *ctx)
{
  if (!ctx) return 0;
  std::lock_guard<std::mutex> lock(ctx->mu);
  try { ctx->prof_resolve(); } catch (...) {}
  // fold what the helper streams (and, for a device list, the other devices' contexts) recorded into this context's
  // table (summed over streams and devices)
  auto fold = [&](mm3d_ctx *h) {
    (void)hipSetDevice(h->device);
    try { h->prof_resolve(); } catch (...) {}
    for (size_t i = 0; i < h->prof.size(); ++i) {
      const int s = ctx->prof_slot(h->prof_names[i].c_str());
      ctx->prof[s].ms += h->prof[i].ms;
      ctx->prof[s].launches += h->prof[i].launches;
      ctx->prof[s].bytes += h->prof[i].bytes;
      h->prof[i] = ProfEntry();
    }
  };
  for (mm3d_ctx *h : ctx->helpers) fold(h);
  for (mm3d_ctx *p : ctx->peers) {
    fold(p);
    for (mm3d_ctx *h : p->helpers) fold(h);
  }
  (void)hipSetDevice(ctx->device);
  return (int)ctx->prof.size();
}
int mm3d_profile_entry(mm3d_ctx *ctx, int i, const char **name, double *total_ms, uint64_t *launches, double *bytes)
{
  if (!ctx) return MM3D_EINVAL;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (i < 0 || i >= (int)ctx->prof.size()) return MM3D_EINVAL;
  if (name) *name = ctx->prof_names[i].c_str();
  if (total_ms) *total_ms = ctx->prof[i].ms;
  if (launches) *launches = ctx->prof[i].launches;
  if (bytes) *bytes = ctx->prof[i].bytes;
  return MM3D_OK;
}

}  // extern "C"
