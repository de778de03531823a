// fake_device.cpp -- TEST INFRASTRUCTURE: host stand-ins for every stage entry point that lives in a .hip file (types.hpp),
// so that the library's HOST code -- the stream scheduler of mm3d_estimate_maps_transforms, the shard driver, the rand()
// state table, the SAC-IA / RANSAC replays, the pose graph, the pool and the waits (capi.cpp, host_pipeline.cpp, linalg.cpp,
// runtime.cpp, compiled for real) -- runs under ThreadSanitizer and AddressSanitizer + UBSan without a GPU
// (tests/test_host_sanitizers.py).  The stages compute cheap deterministic placeholders of the right
// shapes; their numbers mean nothing, only that they depend on nothing but their inputs -- which lets the driver check that
// sixteen worker threads give the bits of one.  "Device" memory is host memory (fake_hip.cpp).  Nothing here is product code.
#include <thread>
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstring>

#include "types.hpp"

namespace mm3d {

static uint32_t mix(uint32_t h, uint32_t v) { h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2); return h; }
static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static uint32_t point_hash(const float4 &p) { return mix(mix(mix(17u, bits(p.x)), bits(p.y)), bits(p.z)); }

// ---- grid.hip --------------------------------------------------------------------------------------------------------
mm3d_cloud *cloud_from_device(Context *, DevBuf<float4> &&pts, size_t n)
{
  auto *cl = new mm3d_cloud();
  cl->pts = std::move(pts);
  cl->n = n;
  cl->n_finite = n;
  return cl;
}
mm3d_cloud *cloud_from_memory(Context *c, const void *src, size_t n, size_t stride, size_t rgba_off)
{
  MM3D_REQUIRE(stride >= 16 && stride % 4 == 0 && rgba_off % 4 == 0 && rgba_off >= 12 && rgba_off + 4 <= stride,
               "mm3d_cloud_create: stride/rgba_offset do not describe an x,y,z,rgba record");
  DevBuf<float4> pts(c, n);
  if (n) MM3D_REQUIRE(src != nullptr, "mm3d_cloud_create: null points with n > 0");
  for (size_t i = 0; i < n; ++i) {
    const unsigned char *p = (const unsigned char *)src + i * stride;
    float4 o;
    std::memcpy(&o.x, p, 12);
    std::memcpy(&o.w, p + rgba_off, 4);
    pts.get()[i] = o;
  }
  return cloud_from_device(c, std::move(pts), n);
}
void cloud_download(Context *, const mm3d_cloud *cl, void *dst, size_t stride, size_t rgba_off)
{
  for (size_t i = 0; i < cl->n; ++i) {
    unsigned char *p = (unsigned char *)dst + i * stride;
    std::memset(p, 0, stride);
    std::memcpy(p, &cl->pts.get()[i].x, 12);
    std::memcpy(p + rgba_off, &cl->pts.get()[i].w, 4);
  }
}
const std::vector<float4> &cloud_host(Context *c, const mm3d_cloud *cl, bool)
{
  auto *m = const_cast<mm3d_cloud *>(cl);
  std::lock_guard<std::recursive_mutex> lk(m->cache_mu);
  if (m->host.size() != m->n) {
    m->host.resize(m->n);
    if (m->n) MM3D_HIP(hipMemcpyAsync(m->host.data(), m->pts.get(), m->n * 16, hipMemcpyDeviceToHost, c->stream));
    c->sync();
  }
  return m->host;
}
void cloud_hilbert(Context *c, const mm3d_cloud *cl_, float)
{
  auto *cl = const_cast<mm3d_cloud *>(cl_);
  std::lock_guard<std::recursive_mutex> lk(cl->cache_mu);
  if (cl->hil_pts.get() || cl->n == 0) return;
  cl->hil_pts = DevBuf<float4>(c, cl->n);
  std::memcpy(cl->hil_pts.get(), cl->pts.get(), cl->n * 16);
  cl->n_wave_items = (int)((cl->n + 63) / 64);
}

// ---- filters / normals / keypoints / descriptors ---------------------------------------------------------------------
static mm3d_cloud *every(Context *c, const mm3d_cloud *in, size_t step, size_t phase)
{
  size_t m = 0;
  for (size_t i = phase; i < in->n; i += step) ++m;
  DevBuf<float4> out(c, m);
  size_t k = 0;
  for (size_t i = phase; i < in->n; i += step) out.get()[k++] = in->pts.get()[i];
  return cloud_from_device(c, std::move(out), m);
}
mm3d_cloud *downsample(Context *c, const mm3d_cloud *in, double resolution) { return every(c, in, resolution > 0.2 ? 3 : 2, 0); }
bool downsample_is_identity(Context *, const mm3d_cloud *, double) { return false; }
mm3d_cloud *remove_outliers(Context *c, const mm3d_cloud *in, double, int) { return every(c, in, 1, 0); }
mm3d_cloud *transform_concat(Context *c, const mm3d_cloud *const *clouds, size_t n, const float *T)
{
  size_t total = 0;
  auto zero = [&](size_t i) { for (int k = 0; k < 16; ++k) if (T[i * 16 + k] != 0.0f) return false; return true; };
  for (size_t i = 0; i < n; ++i) if (clouds[i] && !zero(i)) total += clouds[i]->n;
  DevBuf<float4> out(c, total);
  size_t k = 0;
  for (size_t i = 0; i < n; ++i) {
    if (!clouds[i] || zero(i)) continue;
    const float *M = T + i * 16;
    for (size_t j = 0; j < clouds[i]->n; ++j) {
      const float4 p = clouds[i]->pts.get()[j];
      out.get()[k++] = make_float4(M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
                                   M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14], p.w);
    }
  }
  return cloud_from_device(c, std::move(out), total);
}
mm3d_normals *compute_normals(Context *c, const mm3d_cloud *in, double)
{
  auto *r = new mm3d_normals();
  r->n = in->n;
  r->nrm = DevBuf<float4>(c, in->n);
  for (size_t i = 0; i < in->n; ++i) r->nrm.get()[i] = make_float4(0.f, 0.f, 1.f, 0.01f);
  return r;
}
void normals_of_items(Context *, const mm3d_cloud *, const Grid &, double, const int *, const int *, int, float4 *) {}
mm3d_cloud *detect_keypoints_sift(Context *c, const mm3d_cloud *points, double, int, int, double, double normals_radius, mm3d_normals **normals_out, float)
{
  if (normals_out) *normals_out = compute_normals(c, points, normals_radius);
  // TEST KNOB: the map whose filtered cloud has this many points is LATE (its owner publishes it long after the others)
  if (const char *e = getenv("MM3D_FAKE_LATE_POINTS"))
    if ((size_t)atol(e) == points->n) std::this_thread::sleep_for(std::chrono::milliseconds(60));
  mm3d_cloud *kp = every(c, points, 23, 5);
  for (size_t i = 0; i < kp->n; ++i) kp->pts.get()[i].w = 0.0f;
  return kp;
}
mm3d_cloud *detect_keypoints_harris(Context *c, const mm3d_cloud *points, const mm3d_normals *, double, double) { return every(c, points, 31, 3); }
void harris_response(Context *c, const mm3d_cloud *points, const mm3d_normals *, double, DevBuf<float> &out)
{
  out = DevBuf<float>(c, points->n);
  for (size_t i = 0; i < points->n; ++i) out.get()[i] = (float)(point_hash(points->pts.get()[i]) & 1023u) / 1024.0f;
}
static mm3d_desc *fake_desc(Context *c, const mm3d_cloud *kp, int type, int dim)
{
  auto *d = new mm3d_desc();
  d->n = kp->n; d->dim = dim; d->type = type;
  d->data = DevBuf<float>(c, kp->n * (size_t)dim);
  for (size_t i = 0; i < kp->n; ++i) {
    uint32_t h = point_hash(kp->pts.get()[i]);
    for (int k = 0; k < dim; ++k) { h = mix(h, (uint32_t)k); d->data.get()[i * dim + k] = (float)(h & 255u) / 16.0f; }
  }
  if (type == MM3D_DESC_SHOT) { d->rf = DevBuf<float>(c, kp->n * 9); std::memset(d->rf.get(), 0, kp->n * 9 * sizeof(float)); }
  return d;
}
mm3d_desc *compute_fpfh(Context *c, const mm3d_cloud *, const mm3d_normals *, mm3d_cloud *kp, double) { return fake_desc(c, kp, MM3D_DESC_FPFH, 33); }
mm3d_desc *compute_pfh(Context *c, const mm3d_cloud *, const mm3d_normals *, mm3d_cloud *kp, double) { return fake_desc(c, kp, MM3D_DESC_PFH, 125); }
mm3d_desc *compute_pfhrgb(Context *c, const mm3d_cloud *, const mm3d_normals *, mm3d_cloud *kp, double) { return fake_desc(c, kp, MM3D_DESC_PFHRGB, 250); }
mm3d_desc *compute_rsd(Context *c, const mm3d_cloud *, const mm3d_normals *, mm3d_cloud *kp, double) { return fake_desc(c, kp, MM3D_DESC_RSD, 2); }
mm3d_desc *compute_sc3d(Context *c, const mm3d_cloud *, const mm3d_normals *, mm3d_cloud *kp, double) { return fake_desc(c, kp, MM3D_DESC_SC3D, 1980); }
mm3d_desc *compute_shot(Context *c, const mm3d_cloud *, const mm3d_normals *, mm3d_cloud *kp, double) { return fake_desc(c, kp, MM3D_DESC_SHOT, 1344); }
void debug_libm(Context *, int, const float *x, const float *, int n, float *out) { for (int i = 0; i < n; ++i) out[i] = x[i]; }
size_t debug_sift_cert_octave(Context *, const mm3d_cloud *, double, int, float *, float *, size_t) { return 0; }
void debug_sift_cert_stats(long long *out, int) { for (int i = 0; i < 8; ++i) out[i] = 0; }
void debug_sacia_stats(long long out[4], int, int) { for (int i = 0; i < 4; ++i) out[i] = 0; }
void debug_sift_cert_min(int) {}
void debug_float_chain(Context *, const float *incr, const unsigned *hits, int n, float *out)
{
  for (int i = 0; i < n; ++i) { float v = 0.f; for (unsigned k = 0; k < hits[i]; ++k) v += incr[i]; out[i] = v; }
}

// ---- descriptor k-NN: exact, on the host ----------------------------------------------------------------------------------
static void knn_row(const float *a, const mm3d_desc *B, int k, int *idx, float *d2)
{
  std::vector<std::pair<float, int>> all(B->n);
  for (size_t j = 0; j < B->n; ++j) {
    float s = 0.f;
    for (int t = 0; t < B->dim; ++t) { const float d = a[t] - B->data.get()[j * B->dim + t]; s += d * d; }
    all[j] = {s, (int)j};
  }
  const size_t kk = std::min<size_t>((size_t)k, all.size());
  std::partial_sort(all.begin(), all.begin() + kk, all.end());
  for (int t = 0; t < k; ++t) { idx[t] = t < (int)kk ? all[t].second : -1; d2[t] = t < (int)kk ? all[t].first : INFINITY; }
}
void desc_knn(Context *c, const mm3d_desc *A, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2)
{
  idx = DevBuf<int>(c, A->n * (size_t)k);
  d2 = DevBuf<float>(c, A->n * (size_t)k);
  for (size_t i = 0; i < A->n; ++i) knn_row(A->data.get() + i * A->dim, B, k, idx.get() + i * k, d2.get() + i * k);
}
void desc_knn_prepare_target(Context *, const mm3d_desc *) {}
void desc_knn_rows_multi(Context *c, const KnnRows *srcs, int n_srcs, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2)
{
  size_t rows = 0;
  for (int s = 0; s < n_srcs; ++s) rows += (size_t)srcs[s].n_rows;
  idx = DevBuf<int>(c, rows * (size_t)k);
  d2 = DevBuf<float>(c, rows * (size_t)k);
  size_t r = 0;
  for (int s = 0; s < n_srcs; ++s)
    for (int i = 0; i < srcs[s].n_rows; ++i, ++r)
      knn_row(srcs[s].A->data.get() + (size_t)srcs[s].rows_dev[i] * srcs[s].A->dim, B, k, idx.get() + r * k, d2.get() + r * k);
}
void desc_knn_rows(Context *c, const mm3d_desc *A, const int *rows_dev, int n_rows, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2)
{
  const KnnRows s{A, rows_dev, n_rows};
  desc_knn_rows_multi(c, &s, 1, B, k, idx, d2);
}

// ---- registration ----------------------------------------------------------------------------------------------------------
void ransac_count(Context *, const float4 *src_kp, const float4 *tgt_kp, const int *idx_src, const int *idx_tgt, int n_corr, const float *T_all,
                  int H, double thr2, int *counts)
{
  for (int h = 0; h < H; ++h) {
    const float *M = T_all + (size_t)h * 16;
    int cnt = 0;
    for (int i = 0; i < n_corr; ++i) {
      const float4 p = src_kp[idx_src[i]], q = tgt_kp[idx_tgt[i]];
      const float x = M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12] - q.x, y = M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13] - q.y,
                  z = M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14] - q.z;
      cnt += (double)(x * x + y * y + z * z) <= thr2 ? 1 : 0;
    }
    counts[h] = cnt;
  }
}
void sacia_score_batch(Context *, const SacPair *pairs, int n, int H, float)
{
  for (int i = 0; i < n; ++i) {
    // a model that depends on the replayed samples and on the k-NN table: a translation between the first sample and its pick
    float *T = pairs[i].T_best;
    std::memset(T, 0, 16 * sizeof(float));
    T[0] = T[5] = T[10] = T[15] = 1.0f;
    if (H > 0 && pairs[i].src_kp->n && pairs[i].tgt_kp->n) {
      const int s = pairs[i].samp[0];
      const int t = pairs[i].nn[pairs[i].corr_ref[0]];
      if (s >= 0 && (size_t)s < pairs[i].src_kp->n && t >= 0 && (size_t)t < pairs[i].tgt_kp->n) {
        const float4 a = pairs[i].src_kp->pts.get()[s], b = pairs[i].tgt_kp->pts.get()[t];
        T[12] = b.x - a.x; T[13] = b.y - a.y; T[14] = b.z - a.z;
      }
    }
  }
}
void prepare_pair_search(Context *, const mm3d_cloud *, double, double) {}
void prepare_sacia_target(Context *, const mm3d_cloud *, float) {}
static PairTail fake_tail(const mm3d_cloud *src, const mm3d_cloud *tgt, const float *guess, bool run_icp, bool want_score)
{
  PairTail r{};
  std::memcpy(r.T, guess, 16 * sizeof(float));
  r.iterations = run_icp ? 1 + (int)((src->n + tgt->n) % 3) : 0;
  r.converged = 1;
  r.n_corr = (int)std::min(src->n, tgt->n);
  r.score = want_score ? 0.25 + 1e-6 * (double)((src->n * 31 + tgt->n) % 1000) : 0.0;
  return r;
}
void icp_score_batch(Context *, IcpScoreJob *jobs, int n_jobs, bool run_icp, double, int, double, bool want_score, double)
{
  for (int i = 0; i < n_jobs; ++i) {
    jobs[i].out = fake_tail(jobs[i].src, jobs[i].tgt, jobs[i].guess_dev ? jobs[i].guess_dev : jobs[i].guess_host, run_icp, want_score);
    jobs[i].closed = true;
  }
}
PairTail icp_score(Context *, const mm3d_cloud *src, const mm3d_cloud *tgt, const float *guess_dev, const float guess_host[16], bool run_icp, double,
                   int, double, bool want_score, double)
{
  return fake_tail(src, tgt, guess_dev ? guess_dev : guess_host, run_icp, want_score);
}
IcpResult icp(Context *, const mm3d_cloud *src, const mm3d_cloud *tgt, const float guess[16], double, int, double)
{
  const PairTail t = fake_tail(src, tgt, guess, true, false);
  IcpResult r{};
  std::memcpy(r.T, t.T, sizeof(r.T));
  r.iterations = t.iterations; r.converged = t.converged;
  return r;
}
double transform_score(Context *, const mm3d_cloud *src, const mm3d_cloud *tgt, const float T[16], double) { return fake_tail(src, tgt, T, false, true).score; }

}  // namespace mm3d
