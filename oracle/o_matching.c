/*
 * o_matching.c -- R/src/matching.cpp restated (TEST INFRASTRUCTURE).
 *
 * findFeatureCorrespondences            R/src/matching.cpp:31-108
 * estimateTransformFromCorrespondences  R/src/matching.cpp:110-140
 *    -> CorrespondenceRejectorSampleConsensus -> RandomSampleConsensus over
 *       SampleConsensusModelRegistration (PCL 1.8.1 sample_consensus/impl/ransac.hpp,
 *       sac_model.h, impl/sac_model_registration.hpp), then TransformationEstimationSVD.
 * estimateTransformFromDescriptorsSets  R/src/matching.cpp:142-194
 *    -> SampleConsensusInitialAlignment (registration/impl/ia_ransac.hpp)
 * estimateTransformICP                  R/src/matching.cpp:196-221
 *    -> IterativeClosestPoint (registration/impl/icp.hpp, correspondence_estimation.hpp,
 *       default_convergence_criteria.hpp)
 * estimateTransform                     R/src/matching.cpp:223-257
 * transformScore                        R/src/matching.cpp:259-268
 *    -> TransformationValidationEuclidean::validateTransformation
 */
#include "mm3d_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

void mo_eigen33_values(const float cov[9], float evals[3]);
void mo_mean_cov(const mo_point *pts, const int *idx, int cnt, float cov[9], float centroid[3]);

#define M(T, r, c) (T)[(c) * 4 + (r)]

static inline void xform_point(const float T[16], float x, float y, float z, float out[3])
{
  /* pcl::transformPointCloud / TransformationValidationEuclidean: row . (x,y,z) + t, left to right */
  out[0] = M(T, 0, 0) * x + M(T, 0, 1) * y + M(T, 0, 2) * z + M(T, 0, 3);
  out[1] = M(T, 1, 0) * x + M(T, 1, 1) * y + M(T, 1, 2) * z + M(T, 1, 3);
  out[2] = M(T, 2, 0) * x + M(T, 2, 1) * y + M(T, 2, 2) * z + M(T, 2, 3);
}

/* ------------------------------------------------------------------------------------------ */
/* descriptor k-NN: FLANN L2_Simple (result += diff*diff, sequential over dims), sorted (d2,idx) */
void mo_desc_knn(const float *a, int na, const float *b, int nb, int dim, int k, int *idx,
                 float *d2)
{
#pragma omp parallel for schedule(dynamic, 32) num_threads(mo_get_threads()) if (na >= 64)
  for (int i = 0; i < na; ++i) {
    int m = 0;
    int *ti = &idx[(size_t)i * k];
    float *td = &d2[(size_t)i * k];
    const float *av = &a[(size_t)i * dim];
    for (int j = 0; j < nb; ++j) {
      const float *bv = &b[(size_t)j * dim];
      float r = 0.0f;
      for (int d = 0; d < dim; ++d) { float df = av[d] - bv[d]; r += df * df; }
      int pos = m;
      if (m == k) {
        if (!(r < td[k - 1])) continue;   /* ties keep the lower index (j ascending) */
        pos = k - 1;
      } else {
        ++m;
      }
      while (pos > 0 && td[pos - 1] > r) { td[pos] = td[pos - 1]; ti[pos] = ti[pos - 1]; --pos; }
      td[pos] = r; ti[pos] = j;
    }
    for (int j = m; j < k; ++j) { ti[j] = -1; td[j] = INFINITY; }
  }
}

int mo_find_correspondences(const float *ds, int ns, const float *dt, int nt, int dim, size_t k_,
                            mo_corr *out)
{
  int k = (int)k_;
  if (ns <= 0 || nt <= 0 || k <= 0) return 0;
  int *fi = (int *)malloc(sizeof(int) * (size_t)ns * k), *bi = (int *)malloc(sizeof(int) * (size_t)nt * k);
  float *fd = (float *)malloc(sizeof(float) * (size_t)ns * k), *bd = (float *)malloc(sizeof(float) * (size_t)nt * k);
  mo_desc_knn(ds, ns, dt, nt, dim, k, fi, fd);   /* source -> target */
  mo_desc_knn(dt, nt, ds, ns, dim, k, bi, bd);   /* target -> source */
  int n = 0;
  for (int i = 0; i < ns; ++i) {
    /* the reference indexes k_indices[j] for j < k even when fewer than k came back (latent OOB,
     * R/src/matching.cpp:70-71); the oracle stops at the number actually returned. */
    for (int j = 0; j < k; ++j) {
      int match = fi[(size_t)i * k + j];
      if (match < 0) break;
      float dist = fd[(size_t)i * k + j];
      int found = 0;
      for (int b = 0; b < k; ++b)
        if (bi[(size_t)match * k + b] == i) { found = 1; break; }
      if (found) {
        out[n].index_query = i; out[n].index_match = match; out[n].distance = dist;
        ++n;
        break;
      }
    }
  }
  free(fi); free(bi); free(fd); free(bd);
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* RANSAC */
static inline int ransac_rnd(void) { return (int)(mo_mt19937_next() >> 1); } /* uniform_int<>(0,INT_MAX) */


static void model_from_samples(const mo_point *src, const int *samples, const mo_point *tgt,
                               const int *tgt_of, float T[16])
{
  /* estimateRigidTransformationSVD: double 3xN + pcl::umeyama, cast to float rows */
  double s[9], d[9], Td[16];
  for (int i = 0; i < 3; ++i) {
    const mo_point *p = &src[samples[i]], *q = &tgt[tgt_of[i]];
    s[i * 3] = p->x; s[i * 3 + 1] = p->y; s[i * 3 + 2] = p->z;
    d[i * 3] = q->x; d[i * 3 + 1] = q->y; d[i * 3 + 2] = q->z;
  }
  mo_umeyama_f64(s, d, 3, Td);
  for (int i = 0; i < 16; ++i) T[i] = (float)Td[i];
}

int mo_ransac(const mo_point *src_kp, int ns, const mo_point *tgt_kp, int nt, const mo_corr *corr,
              int n_corr, double inlier_threshold, float T[16], mo_corr *inliers, int *iters_out,
              int *best_count_out)
{
  (void)ns; (void)nt;
  memset(T, 0, sizeof(float) * 16);
  if (iters_out) *iters_out = 0;
  if (best_count_out) *best_count_out = 0;
  const int max_iterations = 1000;     /* CorrespondenceRejectorSampleConsensus default */
  const double probability = 0.99;
  if (n_corr < 3) return 0;            /* getSamples: too few indices -> computeModel fails -> Identity */

  int *indices = (int *)malloc(sizeof(int) * (size_t)n_corr), *indices_tgt = (int *)malloc(sizeof(int) * (size_t)n_corr);
  int *shuffled = (int *)malloc(sizeof(int) * (size_t)n_corr);
  for (int i = 0; i < n_corr; ++i) {
    indices[i] = corr[i].index_query; indices_tgt[i] = corr[i].index_match; shuffled[i] = indices[i];
  }
  /* correspondences_ map: source index -> target index (later entries overwrite) */
  int max_src = 0;
  for (int i = 0; i < n_corr; ++i) if (indices[i] > max_src) max_src = indices[i];
  int *tgt_of_src = (int *)malloc(sizeof(int) * (size_t)(max_src + 1));
  int *pos_of_src = (int *)malloc(sizeof(int) * (size_t)(max_src + 1));
  for (int i = 0; i < n_corr; ++i) { tgt_of_src[indices[i]] = indices_tgt[i]; pos_of_src[indices[i]] = i; }

  /* computeSampleDistanceThreshold(cloud, indices) */
  float cov[9], centroid[3], ev[3];
  mo_mean_cov(src_kp, indices, n_corr, cov, centroid);
  mo_eigen33_values(cov, ev);
  /* eigen_values.array().sqrt().sum() is float; "/ 3.0" promotes to double */
  double sample_dist_thresh = (double)(sqrtf(ev[0]) + sqrtf(ev[1]) + sqrtf(ev[2])) / 3.0;
  sample_dist_thresh *= sample_dist_thresh;

  mo_mt19937_seed(12345u);
  int iterations = 0;
  int n_best = -INT_MAX;
  double k = 1.0;
  const double log_probability = log(1.0 - probability);
  const double one_over_indices = 1.0 / (double)n_corr;
  int best_sel[3] = {-1, -1, -1};
  float best_T[16];
  int have_model = 0;
  const double thr2d = inlier_threshold * inlier_threshold;
  while (iterations < k) {
    /* getSamples: up to max_sample_checks_ = 1000 draws until isSampleGood */
    int sel[3], ok = 0;
    for (int chk = 0; chk < 1000; ++chk) {
      for (int i = 0; i < 3; ++i) {
        int j = i + (ransac_rnd() % (n_corr - i));
        int t = shuffled[i]; shuffled[i] = shuffled[j]; shuffled[j] = t;
      }
      sel[0] = shuffled[0]; sel[1] = shuffled[1]; sel[2] = shuffled[2];
      const mo_point *p0 = &src_kp[sel[0]], *p1 = &src_kp[sel[1]], *p2 = &src_kp[sel[2]];
      float a[3] = {p1->x - p0->x, p1->y - p0->y, p1->z - p0->z};
      float b[3] = {p2->x - p0->x, p2->y - p0->y, p2->z - p0->z};
      float c[3] = {p2->x - p1->x, p2->y - p1->y, p2->z - p1->z};
      float na = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
      float nb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
      float nc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      if ((double)na > sample_dist_thresh && (double)nb > sample_dist_thresh && (double)nc > sample_dist_thresh) { ok = 1; break; }
    }
    if (!ok) break;   /* selection.empty(): "No samples could be selected!" */
    int tg[3] = {tgt_of_src[sel[0]], tgt_of_src[sel[1]], tgt_of_src[sel[2]]};
    float Tm[16];
    model_from_samples(src_kp, sel, tgt_kp, tg, Tm);
    int cnt = 0;
    for (int i = 0; i < n_corr; ++i) {
      float p[3];
      const mo_point *s = &src_kp[indices[i]], *t = &tgt_kp[indices_tgt[i]];
      xform_point(Tm, s->x, s->y, s->z, p);
      float dx = p[0] - t->x, dy = p[1] - t->y, dz = p[2] - t->z;
      float d = dx * dx + dy * dy + dz * dz;
      if ((double)d < thr2d) ++cnt;
    }
    if (cnt > n_best) {
      n_best = cnt;
      memcpy(best_sel, sel, sizeof(sel));
      memcpy(best_T, Tm, sizeof(Tm));
      have_model = 1;
      double w = (double)n_best * one_over_indices;
      double p_no_outliers = 1.0 - pow(w, 3.0);
      if (p_no_outliers < DBL_EPSILON) p_no_outliers = DBL_EPSILON;
      if (p_no_outliers > 1.0 - DBL_EPSILON) p_no_outliers = 1.0 - DBL_EPSILON;
      k = log_probability / log(p_no_outliers);
    }
    ++iterations;
    if (iterations > max_iterations) break;
  }
  if (iters_out) *iters_out = iterations;
  int n_in = 0;
  if (have_model) {
    /* selectWithinDistance -> inlier source indices in index order */
    for (int i = 0; i < n_corr; ++i) {
      float p[3];
      const mo_point *s = &src_kp[indices[i]], *t = &tgt_kp[indices_tgt[i]];
      xform_point(best_T, s->x, s->y, s->z, p);
      float dx = p[0] - t->x, dy = p[1] - t->y, dz = p[2] - t->z;
      float d = dx * dx + dy * dy + dz * dz;
      if ((double)d < thr2d) inliers[n_in++] = corr[pos_of_src[indices[i]]];
    }
  }
  if (best_count_out) *best_count_out = have_model ? n_best : 0;
  int fail = !have_model || n_in < 3;
  if (!fail) {
    /* getBestTransformation().isIdentity() (Eigen isIdentity, prec 1e-5 float) => failure */
    int is_id = 1;
    for (int c = 0; c < 4 && is_id; ++c)
      for (int r = 0; r < 4; ++r) {
        float v = M(best_T, r, c);
        if (r == c) { if (!(fabsf(v - 1.0f) <= 1e-5f * fminf(fabsf(v), 1.0f))) { is_id = 0; break; } }
        else { if (!(fabsf(v) <= 1e-5f)) { is_id = 0; break; } }
      }
    if (is_id) fail = 1;
  }
  if (fail) {
    n_in = 0;                     /* result.setZero(); inliers->clear(); */
  } else {
    float *s = (float *)malloc(sizeof(float) * 3 * (size_t)n_in), *d = (float *)malloc(sizeof(float) * 3 * (size_t)n_in);
    for (int i = 0; i < n_in; ++i) {
      const mo_point *p = &src_kp[inliers[i].index_query], *q = &tgt_kp[inliers[i].index_match];
      s[i * 3] = p->x; s[i * 3 + 1] = p->y; s[i * 3 + 2] = p->z;
      d[i * 3] = q->x; d[i * 3 + 1] = q->y; d[i * 3 + 2] = q->z;
    }
    mo_umeyama_f32(s, d, n_in, T);
    free(s); free(d);
  }
  (void)best_sel;
  free(indices); free(indices_tgt); free(shuffled); free(tgt_of_src); free(pos_of_src);
  return n_in;
}

/* ------------------------------------------------------------------------------------------ */
/* SAC-IA */
static inline int get_random_index(int n) { return (int)(n * (mo_rand() / (2147483647 + 1.0))); }

static float *g_sacia_sink = NULL;
static double *g_sacia_sink64 = NULL;
static int g_sacia_sink_cap = 0;
void mo_sac_ia_error_sink(float *sink, double *sink64, int capacity)
{
  g_sacia_sink = sink; g_sacia_sink64 = sink64; g_sacia_sink_cap = sink ? capacity : 0;
}

void mo_sac_ia(const mo_point *src_kp, const float *src_desc, int ns, const mo_point *tgt_kp,
               const float *tgt_desc, int nt, int dim, double min_sample_distance_d,
               double max_correspondence_distance, int max_iterations, float T[16],
               int *best_iter_out, float *best_err_out)
{
  /* final_transformation_ = guess = Identity if nothing can be done */
  memset(T, 0, sizeof(float) * 16);
  T[0] = T[5] = T[10] = T[15] = 1.0f;
  if (best_iter_out) *best_iter_out = -1;
  if (best_err_out) *best_err_out = 0.0f;
  const int nr_samples = 3, k_corr = 10;
  if (ns < nr_samples || nt < 1) return;   /* selectSamples errors out / empty target */
  float min_sample_distance = (float)min_sample_distance_d;   /* setMinSampleDistance(float) */
  const float corr_thresh = (float)max_correspondence_distance; /* TruncatedError(float(corr_dist_threshold_)) */

  /* feature_tree_: 10-NN of every source feature among target features */
  int kk = k_corr < nt ? k_corr : nt;
  /* findSimilarFeatures searches the feature tree once per drawn sample (at most 3 x max_iterations rows), not
   * for every source feature: rows are searched when a sample first needs them */
  int *nn = (int *)malloc(sizeof(int) * (size_t)ns * k_corr);
  float *nd = (float *)malloc(sizeof(float) * (size_t)k_corr);
  unsigned char *have_nn = (unsigned char *)calloc((size_t)ns, 1);

  /* target keypoint tree for the error metric; a bounded search suffices because the truncated
   * error is 1 whenever d2 > threshold (the un-squared max_correspondence_distance) */
  float sac_cell = sqrtf(corr_thresh);
  if (!(sac_cell > 0.25f)) sac_cell = 0.25f;
  mo_grid *g = mo_grid_build(tgt_kp, nt, sac_cell);

  float lowest_error = 0.0f;
  float best[16];
  memcpy(best, T, sizeof(best));
  float *err_of = (float *)malloc(sizeof(float) * (size_t)ns);
  for (int it = 0; it < max_iterations; ++it) {
    int sample[3], corr_idx[3];
    /* selectSamples */
    {
      int cnt = 0, without = 0;
      const int max_without = 3 * ns;
      while (cnt < nr_samples) {
        int si = get_random_index(ns);
        int valid = 1;
        for (int i = 0; i < cnt; ++i) {
          const mo_point *a = &src_kp[si], *b = &src_kp[sample[i]];
          float dx = a->x - b->x, dy = a->y - b->y, dz = a->z - b->z;
          float dist = sqrtf(dx * dx + dy * dy + dz * dz);   /* euclideanDistance */
          if (si == sample[i] || dist < min_sample_distance) { valid = 0; break; }
        }
        if (valid) { sample[cnt++] = si; without = 0; }
        else ++without;
        if (without >= max_without) { min_sample_distance *= 0.5f; without = 0; }
      }
    }
    /* findSimilarFeatures */
    for (int i = 0; i < nr_samples; ++i) {
      int rc = get_random_index(k_corr);
      if (rc >= kk) rc = kk - 1;   /* reference reads past the resized result (UB) when nt < 10 */
      if (!have_nn[sample[i]]) {
        mo_desc_knn(&src_desc[(size_t)sample[i] * dim], 1, tgt_desc, nt, dim, k_corr, &nn[(size_t)sample[i] * k_corr], nd);
        have_nn[sample[i]] = 1;
      }
      corr_idx[i] = nn[(size_t)sample[i] * k_corr + rc];
    }
    /* TransformationEstimationSVD (float) on the 3 pairs */
    float s[9], d[9], Tm[16];
    for (int i = 0; i < 3; ++i) {
      const mo_point *p = &src_kp[sample[i]], *q = &tgt_kp[corr_idx[i]];
      s[i * 3] = p->x; s[i * 3 + 1] = p->y; s[i * 3 + 2] = p->z;
      d[i * 3] = q->x; d[i * 3 + 1] = q->y; d[i * 3 + 2] = q->z;
    }
    mo_umeyama_f32(s, d, 3, Tm);
    /* computeErrorMetric over all source keypoints: the searches are independent (threads, baseline B2),
     * the float sum runs in keypoint order */
    /* (a few thousand short searches per hypothesis: more than 16 threads only add wake-up time) */
#pragma omp parallel for schedule(static) num_threads(mo_get_threads() < 16 ? mo_get_threads() : 16) if (ns >= 2048)
    for (int i = 0; i < ns; ++i) {
      float p[3];
      xform_point(Tm, src_kp[i].x, src_kp[i].y, src_kp[i].z, p);
      int ni; float nd2;
      int found = mo_knn_search(g, p[0], p[1], p[2], 1, corr_thresh, &ni, &nd2);
      err_of[i] = (found && nd2 <= corr_thresh) ? nd2 / corr_thresh : 1.0f;
    }
    float error = 0.0f;
    for (int i = 0; i < ns; ++i) error += err_of[i];
    if (it < g_sacia_sink_cap) {
      double e64 = 0.0;
      for (int i = 0; i < ns; ++i) e64 += (double)err_of[i];
      g_sacia_sink[it] = error;
      if (g_sacia_sink64) g_sacia_sink64[it] = e64;
    }
    if (it == 0 || error < lowest_error) {
      lowest_error = error;
      memcpy(best, Tm, sizeof(best));
      if (best_iter_out) *best_iter_out = it;
    }
  }
  memcpy(T, best, sizeof(best));
  if (best_err_out) *best_err_out = lowest_error;
  free(nn); free(nd); free(err_of); free(have_nn);
  mo_grid_free(g);
}

/* ------------------------------------------------------------------------------------------ */
/* integer observables of the most recent mo_estimate_transform (what registration_visualisation.cpp:129-130
 * prints for MATCHING, plus the ICP trace); process-global like the rand() state */
static mo_pair_trace g_trace;
void mo_last_pair_trace(mo_pair_trace *out) { *out = g_trace; }
/* the initial estimate (RANSAC | SAC-IA) mo_estimate_transform handed to ICP most recently, and the last-iteration
 * correspondence count of the most recent mo_icp_double_sums: what the exact-arithmetic yardstick of a whole job needs */
static float g_last_init[16];
static int g_dbl_corr = 0;
void mo_last_pair_init(float T[16]) { memcpy(T, g_last_init, sizeof(g_last_init)); }
int mo_last_double_sums_correspondences(void) { return g_dbl_corr; }

/* ICP */
void mo_icp(const mo_point *src, int ns, const mo_point *tgt, int nt, const float guess[16],
            double max_correspondence_distance, double outlier_rejection_threshold,
            int max_iterations, double transformation_epsilon, float T[16], int *iters_out)
{
  (void)outlier_rejection_threshold; /* setRANSACOutlierRejectionThreshold: no rejector registered */
  float final_T[16];
  memset(final_T, 0, sizeof(final_T));
  final_T[0] = final_T[5] = final_T[10] = final_T[15] = 1.0f;
  int iters = 0;
  if (ns > 0 && nt > 0) {
    /* pcl::transformPointCloud(source, transformed, initial_guess) */
    float *cur = (float *)malloc(sizeof(float) * 3 * (size_t)ns);
    for (int i = 0; i < ns; ++i) xform_point(guess, src[i].x, src[i].y, src[i].z, &cur[i * 3]);
    const double max_dist_sqr = max_correspondence_distance * max_correspondence_distance;
    /* bounded 1-NN: anything farther than max_dist is rejected anyway */
    float bound = (float)(max_dist_sqr * (1.0 + 1e-6));
    float cell = (float)(max_correspondence_distance * 0.25);
    if (!(cell > 0.0f)) cell = 0.25f;
    mo_grid *g = mo_grid_build(tgt, nt, cell);
    float *cs = (float *)malloc(sizeof(float) * 3 * (size_t)ns), *cd = (float *)malloc(sizeof(float) * 3 * (size_t)ns);
    float *cdist = (float *)malloc(sizeof(float) * (size_t)ns);
    int *nn_of = (int *)malloc(sizeof(int) * (size_t)ns);       /* per source point: its match, or -1 */
    float *nd_of = (float *)malloc(sizeof(float) * (size_t)ns);
    double prev_mse = DBL_MAX;
    const double rot_thresh = 1.0 - transformation_epsilon, trans_thresh = transformation_epsilon;
    const double mse_abs = 1e-12;
    int converged = 0;
    do {
      /* the searches are independent (threads, baseline B2); the correspondences are collected in
       * source order like the sequential loop */
#pragma omp parallel for schedule(dynamic, 1024) num_threads(mo_get_threads())
      for (int i = 0; i < ns; ++i) {
        int ni; float d2;
        int found = mo_knn_search(g, cur[i * 3], cur[i * 3 + 1], cur[i * 3 + 2], 1, bound, &ni, &d2);
        nn_of[i] = (!found || (double)d2 > max_dist_sqr) ? -1 : ni;
        nd_of[i] = d2;
      }
      int cnt = 0;
      for (int i = 0; i < ns; ++i) {
        const int ni = nn_of[i];
        const float d2 = nd_of[i];
        if (ni < 0) continue;
        cs[cnt * 3] = cur[i * 3]; cs[cnt * 3 + 1] = cur[i * 3 + 1]; cs[cnt * 3 + 2] = cur[i * 3 + 2];
        cd[cnt * 3] = tgt[ni].x; cd[cnt * 3 + 1] = tgt[ni].y; cd[cnt * 3 + 2] = tgt[ni].z;
        cdist[cnt] = d2;
        ++cnt;
      }
      g_trace.icp_correspondences = cnt;
      if (cnt < 3) { converged = 0; break; }   /* min_number_correspondences_ */
      float Tinc[16];
      mo_umeyama_f32(cs, cd, cnt, Tinc);
#pragma omp parallel for schedule(static) num_threads(mo_get_threads())
      for (int i = 0; i < ns; ++i) {
        float p[3];
        xform_point(Tinc, cur[i * 3], cur[i * 3 + 1], cur[i * 3 + 2], p);
        cur[i * 3] = p[0]; cur[i * 3 + 1] = p[1]; cur[i * 3 + 2] = p[2];
      }
      mo_mat4_mul(Tinc, final_T, final_T);
      ++iters;
      /* DefaultConvergenceCriteria::hasConverged */
      if (iters >= max_iterations) { converged = 1; break; }
      double cos_angle = 0.5 * ((double)M(Tinc, 0, 0) + (double)M(Tinc, 1, 1) + (double)M(Tinc, 2, 2) - 1.0);
      double translation_sqr = (double)M(Tinc, 0, 3) * M(Tinc, 0, 3) + (double)M(Tinc, 1, 3) * M(Tinc, 1, 3) +
                               (double)M(Tinc, 2, 3) * M(Tinc, 2, 3);
      if (cos_angle >= rot_thresh && translation_sqr <= trans_thresh) { converged = 1; break; }
      double mse = 0.0;
      for (int i = 0; i < cnt; ++i) mse += cdist[i];
      mse /= (double)cnt;
      if (fabs(mse - prev_mse) < mse_abs) { converged = 1; break; }
      /* relative MSE test disabled: euclidean_fitness_epsilon_ = -DBL_MAX */
      prev_mse = mse;
    } while (!converged);
    free(cur); free(cs); free(cd); free(cdist); free(nn_of); free(nd_of);
    mo_grid_free(g);
  }
  if (iters_out) *iters_out = iters;
  g_trace.icp_iterations = iters;
  mo_mat4_mul(final_T, guess, T);   /* icp.getFinalTransformation() * initial_guess */
}

/* The same ICP with its sums in double and ONE cumulative transform applied to the original source points --
 * what exact arithmetic gives, and how the device formulates it (documented deviation).
 * NOT the reference's arithmetic: pcl::IterativeClosestPoint re-transforms the float cloud every iteration and
 * TransformationEstimationSVD sums in float, sequentially; beyond ~1 M points those float sums carry millimetres of
 * their own rounding noise (a sum of a million coordinates of ~15 m passes 2^24, where a float's ulp is 1).  This
 * variant lets a check tell "the device differs from the CPU path" from "the CPU path's float sums are noisy". */
void mo_icp_double_sums(const mo_point *src, int ns, const mo_point *tgt, int nt, const float guess[16],
                        double max_correspondence_distance, int max_iterations, double transformation_epsilon,
                        float T[16], int *iters_out)
{
  double Tc[16];
  for (int i = 0; i < 16; ++i) Tc[i] = (double)guess[i];
  int iters = 0;
  if (ns > 0 && nt > 0) {
    const double max_dist_sqr = max_correspondence_distance * max_correspondence_distance;
    float bound = (float)(max_dist_sqr * (1.0 + 1e-6));
    float cell = (float)(max_correspondence_distance * 0.25);
    if (!(cell > 0.0f)) cell = 0.25f;
    mo_grid *g = mo_grid_build(tgt, nt, cell);
    float *cur = (float *)malloc(sizeof(float) * 3 * (size_t)ns);
    double *cs = (double *)malloc(sizeof(double) * 3 * (size_t)ns), *cd = (double *)malloc(sizeof(double) * 3 * (size_t)ns);
    int *nn_of = (int *)malloc(sizeof(int) * (size_t)ns);
    float *nd_of = (float *)malloc(sizeof(float) * (size_t)ns);
    double prev_mse = DBL_MAX;
    const double rot_thresh = 1.0 - transformation_epsilon, trans_thresh = transformation_epsilon;
    int converged = 0;
    do {
#pragma omp parallel for schedule(dynamic, 1024) num_threads(mo_get_threads())
      for (int i = 0; i < ns; ++i) {
        const double x = src[i].x, y = src[i].y, z = src[i].z;
        for (int r = 0; r < 3; ++r) cur[i * 3 + r] = (float)(Tc[r] * x + Tc[4 + r] * y + Tc[8 + r] * z + Tc[12 + r]);
        int ni; float d2;
        int found = mo_knn_search(g, cur[i * 3], cur[i * 3 + 1], cur[i * 3 + 2], 1, bound, &ni, &d2);
        nn_of[i] = (!found || (double)d2 > max_dist_sqr) ? -1 : ni;
        nd_of[i] = d2;
      }
      int cnt = 0;
      double mse = 0.0;
      for (int i = 0; i < ns; ++i) {
        const int ni = nn_of[i];
        if (ni < 0) continue;
        cs[cnt * 3] = cur[i * 3]; cs[cnt * 3 + 1] = cur[i * 3 + 1]; cs[cnt * 3 + 2] = cur[i * 3 + 2];
        cd[cnt * 3] = tgt[ni].x; cd[cnt * 3 + 1] = tgt[ni].y; cd[cnt * 3 + 2] = tgt[ni].z;
        mse += nd_of[i];
        ++cnt;
      }
      g_dbl_corr = cnt;
      if (cnt < 3) break;
      double Tinc[16], Tn[16];
      mo_umeyama_f64(cs, cd, cnt, Tinc);
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
          double a = 0.0;
          for (int k = 0; k < 4; ++k) a += Tinc[k * 4 + r] * Tc[c * 4 + k];
          Tn[c * 4 + r] = a;
        }
      memcpy(Tc, Tn, sizeof(Tc));
      ++iters;
      if (iters >= max_iterations) break;
      const double cos_angle = 0.5 * (Tinc[0] + Tinc[5] + Tinc[10] - 1.0);
      const double translation_sqr = Tinc[12] * Tinc[12] + Tinc[13] * Tinc[13] + Tinc[14] * Tinc[14];
      if (cos_angle >= rot_thresh && translation_sqr <= trans_thresh) converged = 1;
      mse /= (double)cnt;
      if (fabs(mse - prev_mse) < 1e-12) converged = 1;
      prev_mse = mse;
    } while (!converged);
    free(cur); free(cs); free(cd); free(nn_of); free(nd_of);
    mo_grid_free(g);
  }
  if (iters_out) *iters_out = iters;
  for (int i = 0; i < 16; ++i) T[i] = (float)Tc[i];
}

/* ------------------------------------------------------------------------------------------ */
double mo_transform_score(const mo_point *src, int ns, const mo_point *tgt, int nt,
                          const float T[16], double max_distance)
{
  if (ns <= 0 || nt <= 0) return DBL_MAX;
  float bound = (float)(max_distance * (1.0 + 1e-6));
  float cell = (float)(sqrt(max_distance) * 0.25);
  if (!(cell > 0.0f)) cell = 0.25f;
  mo_grid *g = mo_grid_build(tgt, nt, cell);
  double fitness = 0.0;
  int nr = 0;
  float *nd_of = (float *)malloc(sizeof(float) * (size_t)ns);   /* per source point: its d2, or -1 */
#pragma omp parallel for schedule(dynamic, 1024) num_threads(mo_get_threads())
  for (int i = 0; i < ns; ++i) {
    float p[3];
    xform_point(T, src[i].x, src[i].y, src[i].z, p);
    int ni; float d2;
    int found = mo_knn_search(g, p[0], p[1], p[2], 1, bound, &ni, &d2);
    nd_of[i] = found ? d2 : -1.0f;
  }
  for (int i = 0; i < ns; ++i) {               /* the double sum in source order */
    const float d2 = nd_of[i];
    if (d2 < 0.0f) continue;
    if ((double)d2 > max_distance) continue;   /* squared distance vs un-squared max_range_ */
    fitness += d2;
    nr++;
  }
  free(nd_of);
  mo_grid_free(g);
  return nr > 0 ? fitness / nr : DBL_MAX;
}

/* ------------------------------------------------------------------------------------------ */
void mo_estimate_transform(const mo_point *src, int ns, const mo_point *src_kp, const float *src_desc,
                           int nsk, const mo_point *tgt, int nt, const mo_point *tgt_kp,
                           const float *tgt_desc, int ntk, int dim, int method, int refine,
                           double inlier_threshold, double max_correspondence_distance,
                           int max_iterations, size_t matching_k, double transform_epsilon, float T[16])
{
  float T0[16];
  memset(&g_trace, 0, sizeof(g_trace));
  if (method == 0) {
    mo_corr *corr = (mo_corr *)malloc(sizeof(mo_corr) * (size_t)(nsk > 0 ? nsk : 1));
    mo_corr *inl = (mo_corr *)malloc(sizeof(mo_corr) * (size_t)(nsk > 0 ? nsk : 1));
    int nc = mo_find_correspondences(src_desc, nsk, tgt_desc, ntk, dim, matching_k, corr);
    int ni = mo_ransac(src_kp, nsk, tgt_kp, ntk, corr, nc, inlier_threshold, T0, inl, NULL, NULL);
    g_trace.n_correspondences = nc;
    g_trace.n_inliers = ni;
    free(corr); free(inl);
  } else {
    /* note the argument mapping at R/src/matching.cpp:243-246: min_sample_distance := inlier_threshold */
    mo_sac_ia(src_kp, src_desc, nsk, tgt_kp, tgt_desc, ntk, dim, inlier_threshold,
              max_correspondence_distance, max_iterations, T0, NULL, NULL);
  }
  memcpy(g_last_init, T0, sizeof(T0));
  if (refine) {
    /* ICP also runs on a zero initial transform (no guard at R/src/matching.cpp:250) */
    mo_icp(src, ns, tgt, nt, T0, max_correspondence_distance, inlier_threshold, max_iterations,
           transform_epsilon, T, NULL);
  } else {
    memcpy(T, T0, sizeof(T0));
  }
}
