// host_pipeline.cpp -- the sequential, RNG-driven host logic of R/src/matching.cpp and the pose
// graph of R/src/graph.cpp + R/src/map_merging.cpp:137-186, driving the gfx950 kernels.
//
// RANSAC (pcl::RandomSampleConsensus) and SAC-IA are sequential loops whose SAMPLE stream depends
// only on the random generator and on geometry of the source keypoints, never on hypothesis
// scores.  So the host replays the exact sample stream (boost::mt19937 seed 12345 / glibc rand()),
// the hypotheses are built (RANSAC: on the host, SAC-IA: on the device) and scored on the device
// all in one launch, and the host replays the accept / early-termination logic over the returned
// scores: identical to the sequential loop.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <list>
#include <queue>

#include "device_util.hpp"

namespace mm3d {

namespace {

inline void xform_host(const float *T, float x, float y, float z, float out[3])
{
  out[0] = T[0] * x + T[4] * y + T[8] * z + T[12];
  out[1] = T[1] * x + T[5] * y + T[9] * z + T[13];
  out[2] = T[2] * x + T[6] * y + T[10] * z + T[14];
}

template <typename T>
void upload(Context *c, DevBuf<T> &d, const std::vector<T> &h)
{
  d = DevBuf<T>(c, h.size() ? h.size() : 1);
  if (!h.empty()) MM3D_HIP(hipMemcpyAsync(d.get(), h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
}

template <typename T>
void download(Context *c, const T *d, std::vector<T> &h, size_t n)
{
  h.resize(n);
  if (n) {
    MM3D_HIP(hipMemcpyAsync(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    c->sync();
  }
}

}  // namespace

// ---------------------------------------------------------------- findFeatureCorrespondences
// R/src/matching.cpp:31-93: for every source descriptor, the first of its k nearest targets whose
// own k nearest sources contain it; at most one correspondence per source, in source order.
size_t find_correspondences(Context *c, const mm3d_desc *s, const mm3d_desc *t, size_t k_, std::vector<mm3d_corr> &out)
{
  out.clear();
  const int ns = (int)s->n, nt = (int)t->n;
  // any positive k (a size_t in the reference, map_merging.cpp:43-47); more neighbours than rows cannot exist
  const int k = (int)std::min<size_t>(k_, (size_t)std::max(ns, nt));
  if (ns == 0 || nt == 0 || k <= 0) return 0;
  DevBuf<int> fi, bi;
  DevBuf<float> fd, bd;
  desc_knn(c, s, t, k, fi, fd);
  desc_knn(c, t, s, k, bi, bd);
  std::vector<int> hfi, hbi;
  std::vector<float> hfd;
  download(c, fi.get(), hfi, (size_t)ns * k);
  download(c, fd.get(), hfd, (size_t)ns * k);
  download(c, bi.get(), hbi, (size_t)nt * k);
  out.reserve(ns);
  for (int i = 0; i < ns; ++i) {
    // the reference reads k_indices[j] for all j < k even when fewer came back (latent OOB,
    // matching.cpp:70-71); we stop at the entries that exist
    for (int j = 0; j < k; ++j) {
      const int match = hfi[(size_t)i * k + j];
      if (match < 0) break;
      bool found = false;
      for (int b = 0; b < k; ++b)
        if (hbi[(size_t)match * k + b] == i) { found = true; break; }
      if (found) {
        out.push_back({i, match, hfd[(size_t)i * k + j]});
        break;
      }
    }
  }
  return out.size();
}

// ---------------------------------------------------------------- RANSAC + SVD on inliers
// R/src/matching.cpp:110-140
size_t ransac_transform(Context *c, const mm3d_cloud *skp_, const mm3d_cloud *tkp_, const mm3d_corr *corr,
                        size_t n_corr_, double inlier_threshold, float T[16], std::vector<mm3d_corr> &inliers)
{
  std::memset(T, 0, sizeof(float) * 16);
  inliers.clear();
  const int n_corr = (int)n_corr_;
  const int max_iterations = 1000;      // CorrespondenceRejectorSampleConsensus default
  const double probability = 0.99;
  if (n_corr < 3) return 0;             // getSamples fails -> computeModel false -> Identity -> zero
  const std::vector<float4> &skp = cloud_host(c, skp_);
  const std::vector<float4> &tkp = cloud_host(c, tkp_);
  std::vector<int> indices(n_corr), indices_tgt(n_corr), shuffled(n_corr);
  int max_src = 0;
  for (int i = 0; i < n_corr; ++i) {
    MM3D_REQUIRE(corr[i].index_query >= 0 && (size_t)corr[i].index_query < skp.size() && corr[i].index_match >= 0 &&
                     (size_t)corr[i].index_match < tkp.size(),
                 "correspondence index out of range");
    indices[i] = corr[i].index_query; indices_tgt[i] = corr[i].index_match; shuffled[i] = indices[i];
    max_src = std::max(max_src, indices[i]);
  }
  std::vector<int> tgt_of_src(max_src + 1, -1), pos_of_src(max_src + 1, -1);
  for (int i = 0; i < n_corr; ++i) { tgt_of_src[indices[i]] = indices_tgt[i]; pos_of_src[indices[i]] = i; }

  // computeSampleDistanceThreshold(cloud, indices): float raw-moment covariance -> eigen33
  double sample_dist_thresh;
  {
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < n_corr; ++j) {
      const float4 &p = skp[indices[j]];
      a[0] += p.x * p.x; a[1] += p.x * p.y; a[2] += p.x * p.z;
      a[3] += p.y * p.y; a[4] += p.y * p.z; a[5] += p.z * p.z;
      a[6] += p.x; a[7] += p.y; a[8] += p.z;
    }
    const float fc = (float)n_corr;
    for (int i = 0; i < 9; ++i) a[i] /= fc;
    float cov[9];
    cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
    cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev[3];
    eigen33_values(cov, ev);
    sample_dist_thresh = (double)(sqrtf(ev[0]) + sqrtf(ev[1]) + sqrtf(ev[2])) / 3.0;
    sample_dist_thresh *= sample_dist_thresh;
  }

  // pre-draw the hypothesis stream (at most max_iterations + 1 iterations can execute)
  Mt19937 rng(12345u);
  auto rnd = [&]() { return (int)(rng.next() >> 1); };   // uniform_int<>(0, INT_MAX)
  const int Hmax = max_iterations + 1;
  std::vector<float> T_all;
  T_all.reserve((size_t)Hmax * 16);
  std::vector<int> sels;
  int H = 0;
  for (; H < Hmax; ++H) {
    int sel[3];
    bool ok = false;
    for (int chk = 0; chk < 1000; ++chk) {     // max_sample_checks_
      for (int i = 0; i < 3; ++i) {
        const int j = i + (rnd() % (n_corr - i));
        std::swap(shuffled[i], shuffled[j]);
      }
      sel[0] = shuffled[0]; sel[1] = shuffled[1]; sel[2] = shuffled[2];
      const float4 &p0 = skp[sel[0]], &p1 = skp[sel[1]], &p2 = skp[sel[2]];
      const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;
      const float bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
      const float cx = p2.x - p1.x, cy = p2.y - p1.y, cz = p2.z - p1.z;
      const float na = ax * ax + ay * ay + az * az, nb = bx * bx + by * by + bz * bz, nc = cx * cx + cy * cy + cz * cz;
      if ((double)na > sample_dist_thresh && (double)nb > sample_dist_thresh && (double)nc > sample_dist_thresh) { ok = true; break; }
    }
    if (!ok) break;   // "No samples could be selected": the sequential loop stops at this iteration
    double s[9], d[9], Td[16];
    for (int i = 0; i < 3; ++i) {
      const float4 &p = skp[sel[i]], &q = tkp[tgt_of_src[sel[i]]];
      s[i * 3] = p.x; s[i * 3 + 1] = p.y; s[i * 3 + 2] = p.z;
      d[i * 3] = q.x; d[i * 3 + 1] = q.y; d[i * 3 + 2] = q.z;
    }
    umeyama_f64(s, d, 3, Td);
    for (int i = 0; i < 16; ++i) T_all.push_back((float)Td[i]);
    sels.insert(sels.end(), sel, sel + 3);
  }
  if (H == 0) return 0;

  // score every hypothesis on the device
  DevBuf<int> d_is, d_it, d_counts(c, H);
  DevBuf<float> d_T;
  upload(c, d_is, indices);
  upload(c, d_it, indices_tgt);
  upload(c, d_T, T_all);
  const double thr2 = inlier_threshold * inlier_threshold;
  ransac_count(c, skp_->pts.get(), tkp_->pts.get(), d_is.get(), d_it.get(), n_corr, d_T.get(), H, thr2, d_counts.get());
  std::vector<int> counts;
  download(c, d_counts.get(), counts, (size_t)H);

  // replay RandomSampleConsensus::computeModel over the scores
  int iterations = 0, n_best = -INT_MAX, best_h = -1;
  double k = 1.0;
  const double log_probability = std::log(1.0 - probability);
  const double one_over_indices = 1.0 / (double)n_corr;
  while (iterations < k) {
    if (iterations >= H) break;     // sample selection failed at this iteration
    const int cnt = counts[iterations];
    if (cnt > n_best) {
      n_best = cnt;
      best_h = iterations;
      const double w = (double)n_best * one_over_indices;
      double p_no_outliers = 1.0 - std::pow(w, 3.0);
      p_no_outliers = std::max(DBL_EPSILON, p_no_outliers);
      p_no_outliers = std::min(1.0 - DBL_EPSILON, p_no_outliers);
      k = log_probability / std::log(p_no_outliers);
    }
    ++iterations;
    if (iterations > max_iterations) break;
  }
  if (best_h < 0) return 0;
  const float *bT = &T_all[(size_t)best_h * 16];
  // selectWithinDistance -> remaining correspondences in index order
  for (int i = 0; i < n_corr; ++i) {
    float p[3];
    const float4 &s = skp[indices[i]], &t = tkp[indices_tgt[i]];
    xform_host(bT, s.x, s.y, s.z, p);
    const float dx = p[0] - t.x, dy = p[1] - t.y, dz = p[2] - t.z;
    const float d = dx * dx + dy * dy + dz * dz;
    if ((double)d < thr2) inliers.push_back(corr[pos_of_src[indices[i]]]);
  }
  bool fail = inliers.size() < 3;
  if (!fail) {
    // ransac.getBestTransformation().isIdentity()  (Eigen isIdentity, float precision 1e-5)
    bool is_id = true;
    for (int cc = 0; cc < 4 && is_id; ++cc)
      for (int r = 0; r < 4; ++r) {
        const float v = bT[cc * 4 + r];
        if (r == cc) { if (!(std::fabs(v - 1.0f) <= 1e-5f * std::fmin(std::fabs(v), 1.0f))) { is_id = false; break; } }
        else { if (!(std::fabs(v) <= 1e-5f)) { is_id = false; break; } }
      }
    fail = is_id;
  }
  if (fail) { inliers.clear(); return 0; }
  std::vector<float> s(inliers.size() * 3), d(inliers.size() * 3);
  for (size_t i = 0; i < inliers.size(); ++i) {
    const float4 &p = skp[inliers[i].index_query], &q = tkp[inliers[i].index_match];
    s[i * 3] = p.x; s[i * 3 + 1] = p.y; s[i * 3 + 2] = p.z;
    d[i * 3] = q.x; d[i * 3 + 1] = q.y; d[i * 3 + 2] = q.z;
  }
  umeyama_f32(s.data(), d.data(), (int)inliers.size(), T);
  return inliers.size();
}

// The rand() stream of one SampleConsensusInitialAlignment::computeTransformation: H iterations of
// selectSamples (3 draws + rejections by the minimum sample distance, which halves after 3 * ns
// failures) and findSimilarFeatures (one draw per sample).  It depends on the SOURCE keypoints only
// (a target with at least one descriptor is assumed), which is what lets the stream scheduler of
// mm3d_estimate_maps_transforms position the generator for a pair before that pair's target exists.
// "sqrtf(d2) < msd" without the square root: the smallest float t with sqrtf(t) >= msd, so that sqrtf(d2) < msd <=> d2 < t
// (sqrtf is correctly rounded and monotone; a NaN d2 fails both tests)
static float sq_threshold(float msd)
{
  if (!(msd > 0.0f)) return 0.0f;                       // nothing is nearer than a distance <= 0
  float t = msd * msd;
  while (sqrtf(t) >= msd && t > 0.0f) t = std::nextafterf(t, 0.0f);
  while (sqrtf(t) < msd) t = std::nextafterf(t, INFINITY);
  return t;
}

// The replay of SampleConsensusInitialAlignment's random stream for one pair (selectSamples + findSimilarFeatures' picks,
// ia_ransac.hpp): it is sequential by nature -- a pair starts where the previous one left the generator -- and every rank /
// device / worker of a sharded job needs it up to its last pair, so its speed is on the critical path of many small maps
// (2 016 pairs per step) and of a rank that owns few pairs.  Round 5: the generator's state in locals and its two ring
// indices wrapped by a compare instead of `% 31`, the distance test on the squared distance against an exactly equivalent
// threshold: 18 -> 8.5 us per pair (scripts/micro/replay_bench.cpp), the same state after every pair.
void sac_ia_draws(GlibcRand &rnd, const std::vector<float4> &skp, int ns, float min_sample_distance, int H, int kk, int *samp,
                  int *pick)
{
  const int nr_samples = 3, k_corr = 10;
  uint32_t *ring = rnd.ring;
  int f = rnd.f, b = rnd.b;
  auto next = [&]() {                                   // GlibcRand::next
    ring[f] += ring[b];
    const uint32_t res = ring[f] >> 1;
    if (++f == 31) f = 0;
    if (++b == 31) b = 0;
    return (int)res;
  };
  // getRandomIndex(n) = int(n * (rand() / (RAND_MAX + 1.0))): the division by 2^31 is an exact scaling
  auto get_random_index = [&](int n) { return (int)(n * (next() * (1.0 / 2147483648.0))); };
  float thr = sq_threshold(min_sample_distance);
  int scratch[3];
  for (int it = 0; it < H; ++it) {
    int *sample = samp ? &samp[(size_t)it * 3] : scratch;
    // selectSamples
    {
      int cnt = 0, without = 0;
      const int max_without = 3 * ns;
      while (cnt < nr_samples) {
        const int si = get_random_index(ns);
        bool valid = true;
        const float4 a = skp[si];
        for (int i = 0; i < cnt; ++i) {
          const float4 &o = skp[sample[i]];
          const float dx = a.x - o.x, dy = a.y - o.y, dz = a.z - o.z;
          const float d2 = dx * dx + dy * dy + dz * dz;         // euclideanDistance = sqrt of this, compared with `<`
          if (si == sample[i] || d2 < thr) { valid = false; break; }
        }
        if (valid) { sample[cnt++] = si; without = 0; }
        else ++without;
        if (without >= max_without) { min_sample_distance *= 0.5f; thr = sq_threshold(min_sample_distance); without = 0; }
      }
    }
    // findSimilarFeatures
    for (int i = 0; i < nr_samples; ++i) {
      int rc = get_random_index(k_corr);
      if (rc >= kk) rc = kk - 1;     // the reference indexes past the resized result when nt < 10 (UB)
      if (pick) pick[(size_t)it * 3 + i] = rc;
    }
  }
  rnd.f = f; rnd.b = b;
}

// the draws estimate_pair(method, source, non-empty target) consumes, without touching the device or the target
void pair_rand_replay(GlibcRand &rnd, int method, const std::vector<float4> &skp_host, double inlier_threshold, int max_iterations)
{
  if (method != MM3D_EST_SAC_IA) return;            // RANSAC seeds its own mt19937 per call
  const int ns = (int)skp_host.size();
  if (ns < 3) return;
  sac_ia_draws(rnd, skp_host, ns, (float)inlier_threshold, max_iterations > 0 ? max_iterations : 0, 10, nullptr, nullptr);
}

void sac_ia_replay(Context *c, const mm3d_cloud *skp_, const mm3d_desc *sd, const mm3d_cloud *tkp_, const mm3d_desc *td,
                   double min_sample_distance_d, int max_iterations, bool execute, PairFront &f)
{
  std::memset(f.T0, 0, sizeof(f.T0));
  f.T0[0] = f.T0[5] = f.T0[10] = f.T0[15] = 1.0f;   // final_transformation_ = guess = Identity
  f.on_device = false;
  f.sac_h = 0;
  f.sac_rows = 0;
  f.sac_nn_ptr = nullptr;
  const int ns = (int)skp_->n, nt = (int)tkp_->n;
  const int nr_samples = 3, k_corr = 10;
  if (ns < nr_samples || nt < 1) return;
  MM3D_REQUIRE(sd->n == (size_t)ns && td->n == (size_t)nt, "SAC-IA: keypoints and descriptors differ in size");
  float min_sample_distance = (float)min_sample_distance_d;
  const std::vector<float4> &skp = cloud_host(c, skp_);
  const int kk = std::min(k_corr, nt);
  const int H = max_iterations > 0 ? max_iterations : 0;
  // The host only replays the sample stream: the rand() draws and the distance tests on source
  // keypoints.  Nothing in that stream depends on the descriptor search (findSimilarFeatures draws
  // one rand() per sample whatever the neighbours are), so the whole stream is replayed FIRST, the
  // k-NN is then computed for the sampled rows only (<= 3 * 500 of ~16k), and the device looks the
  // replayed picks up in its own table and builds the 500 three-point Umeyama models.
  std::vector<int> samp((size_t)H * 3), pick((size_t)H * 3);
  sac_ia_draws(c->rnd, skp, ns, min_sample_distance, H, kk, samp.data(), pick.data());
  if (!execute || H == 0) return;
  // distinct sampled rows -> position in the subset table
  std::vector<int> rows, row_pos((size_t)ns, -1), corr_ref((size_t)H * 3);
  for (size_t e = 0; e < samp.size(); ++e) {
    int &pos = row_pos[samp[e]];
    if (pos < 0) { pos = (int)rows.size(); rows.push_back(samp[e]); }
    corr_ref[e] = pos * k_corr + pick[e];
  }
  // one pinned upload for the three index lists: samp | corr_ref | rows
  const size_t n3 = (size_t)H * 3, nr = rows.size();
  int *hp = (int *)c->pin((2 * n3 + nr) * sizeof(int));
  std::memcpy(hp, samp.data(), n3 * sizeof(int));
  std::memcpy(hp + n3, corr_ref.data(), n3 * sizeof(int));
  std::memcpy(hp + 2 * n3, rows.data(), nr * sizeof(int));
  f.sac_idx = DevBuf<int>(c, 2 * n3 + nr);
  MM3D_HIP(hipMemcpyAsync(f.sac_idx.get(), hp, (2 * n3 + nr) * sizeof(int), hipMemcpyHostToDevice, c->stream));
  f.dT0 = DevBuf<float>(c, 16);
  f.sac_rows = (int)nr;
  f.sac_h = H;
}

void sac_ia_knn(Context *c, SacPrepared *same_target, int n, DevBuf<int> &nn_owner, DevBuf<float> &nd_owner)
{
  const int k_corr = 10;
  std::vector<KnnRows> srcs;
  std::vector<int> who;
  for (int i = 0; i < n; ++i) {
    PairFront &f = *same_target[i].front;
    if (f.sac_h == 0) continue;
    MM3D_REQUIRE(same_target[i].td == same_target[0].td, "sac_ia_knn: the pairs of a group share their target");
    srcs.push_back(KnnRows{same_target[i].sd, f.sac_idx.get() + 2 * (size_t)f.sac_h * 3, f.sac_rows});
    who.push_back(i);
  }
  if (srcs.empty()) return;
  desc_knn_rows_multi(c, srcs.data(), (int)srcs.size(), same_target[0].td, k_corr, nn_owner, nd_owner);
  size_t off = 0;
  for (size_t j = 0; j < who.size(); ++j) {
    same_target[who[j]].front->sac_nn_ptr = nn_owner.get() + off * k_corr;
    off += (size_t)srcs[j].n_rows;
  }
}

void sac_ia_finish(Context *c, SacPrepared *pairs, int n, double max_corr_dist)
{
  std::vector<SacPair> jobs;
  int H = 0;
  for (int i = 0; i < n; ++i) {
    PairFront &f = *pairs[i].front;
    if (f.sac_h == 0) continue;
    MM3D_REQUIRE(H == 0 || H == f.sac_h, "SAC-IA batch: pairs differ in their iteration count");
    H = f.sac_h;
    const size_t n3 = (size_t)H * 3;
    MM3D_REQUIRE(f.sac_nn_ptr != nullptr, "SAC-IA: the k-NN step has not run");
    jobs.push_back(SacPair{pairs[i].skp, pairs[i].tkp, f.sac_idx.get(), f.sac_idx.get() + n3, f.sac_nn_ptr, f.dT0.get()});
    f.on_device = true;              // the caller keeps going on the device (ICP reads the winner there)
  }
  sacia_score_batch(c, jobs.data(), (int)jobs.size(), H, (float)max_corr_dist);
}

// ---------------------------------------------------------------- SAC-IA
// R/src/matching.cpp:142-194 -> pcl::SampleConsensusInitialAlignment (nr_samples 3,
// k_correspondences 10, TruncatedError(max_correspondence_distance)).
bool sac_ia(Context *c, const mm3d_cloud *skp_, const mm3d_desc *sd, const mm3d_cloud *tkp_, const mm3d_desc *td,
            double min_sample_distance_d, double max_corr_dist, int max_iterations, float T[16], bool execute,
            DevBuf<float> *T_dev)
{
  PairFront f;
  sac_ia_replay(c, skp_, sd, tkp_, td, min_sample_distance_d, max_iterations, execute, f);
  std::memcpy(T, f.T0, sizeof(f.T0));
  if (f.sac_h == 0) return false;
  SacPrepared one{skp_, tkp_, sd, td, &f};
  sac_ia_knn(c, &one, 1, f.sac_nn, f.sac_nd);
  sac_ia_finish(c, &one, 1, max_corr_dist);
  if (T_dev) {
    *T_dev = std::move(f.dT0);
    return true;
  }
  float *hT = (float *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(hT, f.dT0.get(), sizeof(float) * 16, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  std::memcpy(T, hT, sizeof(float) * 16);
  return false;
}

// ---------------------------------------------------------------- estimateTransform
// R/src/matching.cpp:223-257
// the part of estimateTransform before ICP: the initial estimate, on the host (T0) or on the device (dT0)
void estimate_pair_front(Context *c, const mm3d_cloud *skp, const mm3d_desc *sd, const mm3d_cloud *tkp, const mm3d_desc *td, int method,
                         double inlier_threshold, double max_corr_dist, int max_iterations, size_t matching_k, bool execute, PairFront &f)
{
  std::memset(f.T0, 0, sizeof(f.T0));
  f.on_device = false;
  f.counts = PairCounts();
  if (method == MM3D_EST_MATCHING) {
    if (execute) {
      std::vector<mm3d_corr> corr, inl;
      find_correspondences(c, sd, td, matching_k, corr);
      ransac_transform(c, skp, tkp, corr.data(), corr.size(), inlier_threshold, f.T0, inl);
      f.counts.n_correspondences = (int)corr.size();
      f.counts.n_inliers = (int)inl.size();
    }
  } else if (method == MM3D_EST_SAC_IA) {
    // argument mapping of matching.cpp:243-246: min_sample_distance := inlier_threshold
    f.on_device = sac_ia(c, skp, sd, tkp, td, inlier_threshold, max_corr_dist, max_iterations, f.T0, execute, &f.dT0);
  } else {
    throw Error(MM3D_EINVAL, "unknown estimation method");
  }
}

int estimate_pair(Context *c, const mm3d_cloud *sp, const mm3d_cloud *skp, const mm3d_desc *sd, const mm3d_cloud *tp,
                  const mm3d_cloud *tkp, const mm3d_desc *td, int method, int refine, double inlier_threshold,
                  double max_corr_dist, int max_iterations, size_t matching_k, double eps, float T[16], bool execute,
                  bool want_score, double score_max_distance, double *score, PairCounts *counts)
{
  PairFront f;
  estimate_pair_front(c, skp, sd, tkp, td, method, inlier_threshold, max_corr_dist, max_iterations, matching_k, execute, f);
  if (counts) *counts = f.counts;
  if (!execute) { std::memset(T, 0, sizeof(float) * 16); return 0; }
  // no guard in the reference: ICP also runs from a zero matrix (and returns zero)
  const PairTail r = icp_score(c, sp, tp, f.on_device ? f.dT0.get() : nullptr, f.T0, refine != 0, max_corr_dist, max_iterations, eps,
                               want_score, score_max_distance);
  std::memcpy(T, r.T, sizeof(r.T));
  if (score) *score = r.score;
  if (counts) counts->icp_correspondences = r.n_corr;
  return r.iterations;
}

int estimate_transform(Context *c, const mm3d_cloud *sp, const mm3d_cloud *skp, const mm3d_desc *sd, const mm3d_cloud *tp,
                       const mm3d_cloud *tkp, const mm3d_desc *td, int method, int refine, double inlier_threshold,
                       double max_corr_dist, int max_iterations, size_t matching_k, double eps, float T[16], bool execute)
{
  return estimate_pair(c, sp, skp, sd, tp, tkp, td, method, refine, inlier_threshold, max_corr_dist, max_iterations, matching_k,
                       eps, T, execute, false, 0.0, nullptr);
}

// ---------------------------------------------------------------- pose graph (host only)
namespace {

struct DisjointSets {
  std::vector<size_t> parent, size, rank;
  explicit DisjointSets(size_t n) : parent(n), size(n, 1), rank(n, 0)
  {
    for (size_t i = 0; i < n; ++i) parent[i] = i;
  }
  size_t find(size_t elem)
  {
    size_t set = elem;
    while (set != parent[set]) set = parent[set];
    while (elem != parent[elem]) { size_t next = parent[elem]; parent[elem] = set; elem = next; }
    return set;
  }
  size_t merge(size_t s1, size_t s2)
  {
    if (rank[s1] < rank[s2]) { parent[s1] = s2; size[s2] += size[s1]; return s2; }
    if (rank[s2] < rank[s1]) { parent[s2] = s1; size[s1] += size[s2]; return s1; }
    parent[s1] = s2; rank[s2]++; size[s2] += size[s1];
    return s2;
  }
};

size_t nodes_in(const std::vector<mm3d_pair_result> &e)
{
  size_t m = 0;
  for (const auto &p : e) m = std::max({m, (size_t)p.source_idx + 1, (size_t)p.target_idx + 1});
  return m;
}

}  // namespace

// computeGlobalTransforms (R/src/map_merging.cpp:153-186): largestConnectedComponent
// (graph.cpp:64-102, including the sub-threshold edge leak of its second loop),
// findMaxSpanningTree (graph.cpp:104-175), BFS from centres[0] chaining
// global[to] = global[from] * getTransform(from, to).
int global_transforms(const mm3d_pair_result *pairs_, size_t n_pairs, double thr, size_t n_clouds, float *out, size_t *n_out)
{
  std::vector<mm3d_pair_result> pairs(pairs_, pairs_ + n_pairs);
  const size_t nodes_count = nodes_in(pairs);
  if (nodes_count == 0) {
    // no pair survived: the reference indexes an empty centre list (UB).  Defined here: n_clouds
    // zero matrices (nothing could be estimated).
    std::memset(out, 0, sizeof(float) * 16 * n_clouds);
    *n_out = n_clouds;
    return MM3D_OK;
  }
  if (nodes_count > n_clouds) return MM3D_EINVAL;
  // largest connected component
  std::vector<mm3d_pair_result> comp;
  {
    DisjointSets sets(nodes_count);
    for (const auto &e : pairs) {
      if (e.confidence < thr) continue;
      size_t a = sets.find(e.source_idx), b = sets.find(e.target_idx);
      if (a != b) sets.merge(a, b);
    }
    size_t max_comp = 0;
    for (size_t i = 1; i < nodes_count; ++i)
      if (sets.size[i] > sets.size[max_comp]) max_comp = i;
    for (const auto &e : pairs)
      if (sets.find(e.source_idx) == max_comp) comp.push_back(e);
  }
  std::memset(out, 0, sizeof(float) * 16 * nodes_count);
  *n_out = nodes_count;
  const size_t nn = nodes_in(comp);
  if (nn == 0) return MM3D_OK;   // reference: UB; nothing reachable
  // maximum spanning tree (Kruskal).  std::sort is unstable in the reference; equal weights keep
  // pair order here.
  struct Edge { size_t from, to; double w; size_t ord; };
  std::vector<Edge> edges;
  for (size_t i = 0; i < comp.size(); ++i) edges.push_back({(size_t)comp[i].source_idx, (size_t)comp[i].target_idx, comp[i].confidence, i});
  std::stable_sort(edges.begin(), edges.end(), [](const Edge &a, const Edge &b) { return a.w > b.w; });
  DisjointSets sets(nn);
  std::vector<std::list<size_t>> tree(nn);
  std::vector<size_t> powers(nn, 0);
  for (const auto &e : edges) {
    size_t a = sets.find(e.from), b = sets.find(e.to);
    if (a != b) {
      sets.merge(a, b);
      tree[e.from].push_back(e.to);
      tree[e.to].push_back(e.from);
      powers[e.from]++; powers[e.to]++;
    }
  }
  auto bfs = [&](size_t from, auto &&body) {
    std::vector<bool> was(nn, false);
    std::queue<size_t> q;
    was[from] = true; q.push(from);
    while (!q.empty()) {
      size_t v = q.front(); q.pop();
      for (size_t to : tree[v])
        if (!was[to]) { body(v, to); was[to] = true; q.push(to); }
    }
  };
  std::vector<size_t> max_d(nn, 0), cur;
  for (size_t leaf = 0; leaf < nn; ++leaf) {
    if (powers[leaf] != 1) continue;
    cur.assign(nn, 0);
    bfs(leaf, [&](size_t f, size_t t) { cur[t] = cur[f] + 1; });
    for (size_t j = 0; j < nn; ++j) max_d[j] = std::max(max_d[j], cur[j]);
  }
  size_t min_max = max_d[0];
  for (size_t i = 1; i < nn; ++i) min_max = std::min(min_max, max_d[i]);
  size_t ref = 0;
  for (size_t i = 0; i < nn; ++i)
    if (max_d[i] == min_max) { ref = i; break; }   // centres[0]
  float *G = out;
  std::memset(&G[ref * 16], 0, sizeof(float) * 16);
  G[ref * 16 + 0] = G[ref * 16 + 5] = G[ref * 16 + 10] = G[ref * 16 + 15] = 1.0f;
  auto get_transform = [&](size_t from, size_t to, float *o) {
    for (const auto &e : comp) {
      if (e.source_idx == from && e.target_idx == to) { mat4_inverse(e.transform, o); return; }
      if (e.source_idx == to && e.target_idx == from) { std::memcpy(o, e.transform, sizeof(float) * 16); return; }
    }
    std::memset(o, 0, sizeof(float) * 16);
  };
  bfs(ref, [&](size_t f, size_t t) {
    float Tft[16];
    get_transform(f, t, Tft);
    mat4_mul(&G[f * 16], Tft, &G[t * 16]);
  });
  return MM3D_OK;
}

}  // namespace mm3d
