/*
 * o_fpfh.c -- computeLocalDescriptors(FPFH) restated (TEST INFRASTRUCTURE).
 *
 * R/src/features.cpp:99-150 with the FPFH row of R/src/dispatch_descriptors.h:40
 *   (pcl::FPFHEstimation<PointXYZRGB, Normal, FPFHSignature33>, setRadiusSearch,
 *    setSearchSurface(points), setInputNormals, setInputCloud(keypoints)), then pruning of
 *    descriptors with any non-finite bin together with their keypoints (:118-143).
 * PCL 1.8.1 features/impl/fpfh.hpp: computeFeature, computeSPFHSignatures,
 *   computePointSPFHSignature, weightPointSPFHSignature; features/src/pfh.cpp
 *   pcl::computePairFeatures.  FPFHEstimation::computePairFeatures returns true
 *   unconditionally, so a degenerate pair (f=0,0,0) is still binned.
 */
#include "mm3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NB 11

static void pair_features(const mo_point *p1, const mo_normal *n1, const mo_point *p2,
                          const mo_normal *n2, float *f1, float *f2, float *f3, float *f4)
{
  float d[3] = {p2->x - p1->x, p2->y - p1->y, p2->z - p1->z};
  *f4 = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (*f4 == 0.0f) { *f1 = *f2 = *f3 = *f4 = 0.0f; return; }
  float a[3] = {n1->nx, n1->ny, n1->nz}, b[3] = {n2->nx, n2->ny, n2->nz};
  float angle1 = (a[0] * d[0] + a[1] * d[1] + a[2] * d[2]) / *f4;
  float angle2 = (b[0] * d[0] + b[1] * d[1] + b[2] * d[2]) / *f4;
  if (acos(fabs(angle1)) > acos(fabs(angle2))) {
    /* switch p1 and p2 */
    float t;
    for (int i = 0; i < 3; ++i) { t = a[i]; a[i] = b[i]; b[i] = t; d[i] *= -1.0f; }
    *f3 = -angle2;
  } else {
    *f3 = angle1;
  }
  /* v = d x n1 */
  float v[3] = {d[1] * a[2] - d[2] * a[1], d[2] * a[0] - d[0] * a[2], d[0] * a[1] - d[1] * a[0]};
  float v_norm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (v_norm == 0.0f) { *f1 = *f2 = *f3 = *f4 = 0.0f; return; }
  v[0] /= v_norm; v[1] /= v_norm; v[2] /= v_norm;
  /* w = n1 x v */
  float w[3] = {a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0]};
  *f2 = v[0] * b[0] + v[1] * b[1] + v[2] * b[2];
  *f1 = atan2f(w[0] * b[0] + w[1] * b[1] + w[2] * b[2], a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
}

/* Test hook: pcl::computePairFeatures on n pairs, out[5 i ..] = {f1, f2, f3, f4, branch}, branch = 1 when the "switch p1 and
 * p2" branch was taken, 0 when not, 2 on the f4 == 0 exit.  tests/test_oracle_cpu.py uses it to show that the features of
 * (p1, p2) and (p2, p1) are the same bits whenever exactly one of the two calls switches (the device computes such a pair
 * once and votes into both histograms, csrc/fpfh.hip::k_spfh) -- and that they differ on ties, which it evaluates twice. */
void mo_pair_features(const mo_point *p1, const mo_normal *n1, const mo_point *p2, const mo_normal *n2, int n, float *out)
{
  for (int i = 0; i < n; ++i) {
    float *o = out + 5 * (size_t)i;
    pair_features(&p1[i], &n1[i], &p2[i], &n2[i], &o[0], &o[1], &o[2], &o[3]);
    float d[3] = {p2[i].x - p1[i].x, p2[i].y - p1[i].y, p2[i].z - p1[i].z};
    float f4 = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (f4 == 0.0f) { o[4] = 2.0f; continue; }
    float angle1 = (n1[i].nx * d[0] + n1[i].ny * d[1] + n1[i].nz * d[2]) / f4;
    float angle2 = (n2[i].nx * d[0] + n2[i].ny * d[1] + n2[i].nz * d[2]) / f4;
    o[4] = (acos(fabs(angle1)) > acos(fabs(angle2))) ? 1.0f : 0.0f;
  }
}

/* static_cast<int>(floor(x)) with x86 cvttsd2si semantics for NaN/out-of-range (INT_MIN) */
static inline int floor_to_int(double x)
{
  double f = floor(x);
  if (!(f >= -2147483648.0 && f <= 2147483647.0)) return (-2147483647 - 1);
  return (int)f;
}

static void point_spfh(const mo_point *cloud, const mo_normal *normals, int p_idx, const int *nbr,
                       int cnt, float *hist /* 33 */)
{
  const float d_pi = 1.0f / (2.0f * (float)M_PI);
  float hist_incr = 100.0f / (float)(cnt - 1);
  for (int j = 0; j < cnt; ++j) {
    if (p_idx == nbr[j]) continue;
    float f1, f2, f3, f4;
    pair_features(&cloud[p_idx], &normals[p_idx], &cloud[nbr[j]], &normals[nbr[j]], &f1, &f2, &f3, &f4);
    int h = floor_to_int(NB * ((f1 + M_PI) * d_pi));
    if (h < 0) h = 0;
    if (h >= NB) h = NB - 1;
    hist[h] += hist_incr;
    h = floor_to_int(NB * ((f2 + 1.0) * 0.5));
    if (h < 0) h = 0;
    if (h >= NB) h = NB - 1;
    hist[NB + h] += hist_incr;
    h = floor_to_int(NB * ((f3 + 1.0) * 0.5));
    if (h < 0) h = 0;
    if (h >= NB) h = NB - 1;
    hist[2 * NB + h] += hist_incr;
  }
}

int mo_fpfh_raw(const mo_point *surface, const mo_normal *normals, int n,
                const mo_point *keypoints, int n_kp, double radius, float *desc,
                int *support_idx, float *spfh_out)
{
  mo_grid *g = mo_grid_build(surface, n, (float)(radius * 0.5));
  const float r2 = (float)(radius * radius);
  /* every loop below is over independent keypoints / support points: baseline B2 runs them on
   * mo_get_threads() threads, each with its own search buffers (SEARCH_BUFFERS) */
#define SEARCH_BUFFERS                                                               \
  int cap = 4096;                                                                    \
  int *idx = (int *)malloc(sizeof(int) * (size_t)cap);                               \
  float *d2 = (float *)malloc(sizeof(float) * (size_t)cap)
#define SEARCH(qx, qy, qz, cntvar)                                                   \
  do {                                                                               \
    cntvar = mo_radius_search(g, qx, qy, qz, r2, idx, d2, cap);                      \
    if (cntvar > cap) {                                                              \
      cap = cntvar * 2;                                                              \
      idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);                          \
      d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);                        \
      cntvar = mo_radius_search(g, qx, qy, qz, r2, idx, d2, cap);                    \
    }                                                                                \
  } while (0)

  /* computeSPFHSignatures: std::set of all neighbours of all keypoints (surface != input) */
  unsigned char *in_set = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
#pragma omp parallel num_threads(mo_get_threads())
  {
    SEARCH_BUFFERS;
#pragma omp for schedule(dynamic, 64)
    for (int k = 0; k < n_kp; ++k) {
      int cnt;
      SEARCH(keypoints[k].x, keypoints[k].y, keypoints[k].z, cnt);
      for (int j = 0; j < cnt; ++j) {
#pragma omp atomic write
        in_set[idx[j]] = 1;
      }
    }
    free(idx); free(d2);
  }
  int *lookup = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int ns = 0;
  for (int i = 0; i < n; ++i) if (in_set[i]) lookup[i] = ns++; else lookup[i] = -1;
  float *spfh = (float *)calloc((size_t)(ns > 0 ? ns : 1) * 33, sizeof(float));
#pragma omp parallel num_threads(mo_get_threads())
  {
    SEARCH_BUFFERS;
#pragma omp for schedule(dynamic, 256)
    for (int i = 0; i < n; ++i) {
      if (!in_set[i]) continue;
      if (support_idx) support_idx[lookup[i]] = i;
      int cnt;
      SEARCH(surface[i].x, surface[i].y, surface[i].z, cnt);
      if (cnt == 0) continue;
      point_spfh(surface, normals, i, idx, cnt, &spfh[(size_t)lookup[i] * 33]);
    }
    free(idx); free(d2);
  }
  if (spfh_out) memcpy(spfh_out, spfh, sizeof(float) * (size_t)ns * 33);

  /* computeFeature: weightPointSPFHSignature per keypoint */
#pragma omp parallel num_threads(mo_get_threads())
  {
  SEARCH_BUFFERS;
#pragma omp for schedule(dynamic, 64)
  for (int k = 0; k < n_kp; ++k) {
    float *out = &desc[(size_t)k * 33];
    int cnt;
    SEARCH(keypoints[k].x, keypoints[k].y, keypoints[k].z, cnt);
    if (cnt == 0) {
      for (int b = 0; b < 33; ++b) out[b] = NAN;
      continue;
    }
    double sum[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < 33; ++b) out[b] = 0.0f;
    for (int j = 0; j < cnt; ++j) {
      if (d2[j] == 0) continue;            /* minus the query point itself */
      float weight = 1.0f / d2[j];
      const float *h = &spfh[(size_t)lookup[idx[j]] * 33];
      for (int f = 0; f < 3; ++f)
        for (int b = 0; b < NB; ++b) {
          float val = h[f * NB + b] * weight;
          sum[f] += val;
          out[f * NB + b] += val;
        }
    }
    for (int f = 0; f < 3; ++f) {
      if (sum[f] != 0) sum[f] = 100.0 / sum[f];
      for (int b = 0; b < NB; ++b) out[f * NB + b] *= (float)sum[f];
    }
  }
  free(idx); free(d2);
  }
#undef SEARCH
#undef SEARCH_BUFFERS
  free(in_set); free(lookup); free(spfh);
  mo_grid_free(g);
  return ns;
}

/*
 * computeLocalDescriptors(PFH): the PFH row of R/src/dispatch_descriptors.h:38 -- the reference's
 * DEFAULT descriptor (map_merging.h:33) -- pcl::PFHEstimation<PointXYZRGB, Normal, PFHSignature125>.
 * PCL 1.8.1 features/impl/pfh.hpp: computeFeature (no cache: use_cache_ defaults to false) and
 * computePointPFHSignature: every pair (i, j < i) of the keypoint's radius neighbours (sorted by
 * distance, so p1 = the farther one), 5 x 5 x 5 bins over (f1, f2, f3), each hit adds
 * 100 / (n (n-1) / 2) with the pair count taken in size_t arithmetic.  Like FPFH's, the class'
 * computePairFeatures wrapper returns true unconditionally, so degenerate pairs are binned.
 */
#define PFH_SPLIT 5
int mo_pfh_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints,
               int n_kp, double radius, float *desc /* n_kp x 125 */)
{
  mo_grid *g = mo_grid_build(surface, n, (float)(radius * 0.5));
  const float r2 = (float)(radius * radius);
  const float d_pi = 1.0f / (2.0f * (float)M_PI);
  int cap = 4096;
  int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)cap);
  for (int k = 0; k < n_kp; ++k) {
    float *out = &desc[(size_t)k * 125];
    int cnt = mo_radius_search(g, keypoints[k].x, keypoints[k].y, keypoints[k].z, r2, idx, d2, cap);
    if (cnt > cap) {
      cap = cnt * 2;
      idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
      d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);
      cnt = mo_radius_search(g, keypoints[k].x, keypoints[k].y, keypoints[k].z, r2, idx, d2, cap);
    }
    if (cnt == 0) {
      for (int b = 0; b < 125; ++b) out[b] = NAN;
      continue;
    }
    for (int b = 0; b < 125; ++b) out[b] = 0.0f;
    const float hist_incr = 100.0f / (float)((size_t)cnt * ((size_t)cnt - 1) / 2);
    for (int i = 0; i < cnt; ++i)
      for (int j = 0; j < i; ++j) {
        float f1, f2, f3, f4;
        pair_features(&surface[idx[i]], &normals[idx[i]], &surface[idx[j]], &normals[idx[j]], &f1, &f2, &f3, &f4);
        int h1 = floor_to_int(PFH_SPLIT * ((f1 + M_PI) * d_pi));
        if (h1 < 0) h1 = 0;
        if (h1 >= PFH_SPLIT) h1 = PFH_SPLIT - 1;
        int h2 = floor_to_int(PFH_SPLIT * ((f2 + 1.0) * 0.5));
        if (h2 < 0) h2 = 0;
        if (h2 >= PFH_SPLIT) h2 = PFH_SPLIT - 1;
        int h3 = floor_to_int(PFH_SPLIT * ((f3 + 1.0) * 0.5));
        if (h3 < 0) h3 = 0;
        if (h3 >= PFH_SPLIT) h3 = PFH_SPLIT - 1;
        out[h1 + PFH_SPLIT * h2 + PFH_SPLIT * PFH_SPLIT * h3] += hist_incr;
      }
  }
  free(idx); free(d2);
  mo_grid_free(g);
  return n_kp;
}

/*
 * computeLocalDescriptors(PFHRGB): dispatch_descriptors.h:39 = pcl::PFHRGBEstimation<PointXYZRGB, Normal,
 * PFHRGBSignature250>.  PCL 1.8.1 features/impl/pfhrgb.hpp (computeFeature, computePointPFHRGBSignature) and
 * features/src/pfh.cpp pcl::computeRGBPairFeatures.  Unlike PFH:
 *   - EVERY ordered pair (i, j != i) of the neighbours is binned (so each 125-bin half sums to 200);
 *   - the Darboux frame is always built on the first point (no angle comparison / swap), f3 = angle1;
 *   - f5..f7 are colour ratios computed with INTEGER division (Eigen::Vector4i colours:
 *     static_cast<float>(c1 / c2), 1 when c2 == 0), folded into [-1, 1] by f > 1 -> -1 / f;
 *   - computeFeature has no "no neighbours" branch: such a keypoint keeps an all-zero (valid) row.
 * The class' computeRGBPairFeatures wrapper returns true unconditionally, so degenerate pairs are binned.
 */
static void rgb_pair_features(const mo_point *p1, const mo_normal *n1, const mo_point *p2, const mo_normal *n2, float f[7])
{
  float d[3] = {p2->x - p1->x, p2->y - p1->y, p2->z - p1->z};
  f[3] = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (f[3] == 0.0f) { for (int i = 0; i < 7; ++i) f[i] = 0.0f; return; }
  const float a[3] = {n1->nx, n1->ny, n1->nz}, b[3] = {n2->nx, n2->ny, n2->nz};
  f[2] = (a[0] * d[0] + a[1] * d[1] + a[2] * d[2]) / f[3];
  float v[3] = {d[1] * a[2] - d[2] * a[1], d[2] * a[0] - d[0] * a[2], d[0] * a[1] - d[1] * a[0]};
  const float v_norm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (v_norm == 0.0f) { for (int i = 0; i < 7; ++i) f[i] = 0.0f; return; }
  v[0] /= v_norm; v[1] /= v_norm; v[2] /= v_norm;
  const float w[3] = {a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0]};
  f[1] = v[0] * b[0] + v[1] * b[1] + v[2] * b[2];
  f[0] = atan2f(w[0] * b[0] + w[1] * b[1] + w[2] * b[2], a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
  for (int c = 0; c < 3; ++c) {
    const int sh = 16 - 8 * c;                                   /* r, g, b */
    const int c1 = (int)((p1->rgba >> sh) & 0xff), c2 = (int)((p2->rgba >> sh) & 0xff);
    float r = (c2 != 0) ? (float)(c1 / c2) : 1.0f;
    if (r > 1.0f) r = -1.0f / r;
    f[4 + c] = r;
  }
}

static inline int split_bin(double x)
{
  int h = floor_to_int(x);
  if (h < 0) h = 0;
  if (h >= PFH_SPLIT) h = PFH_SPLIT - 1;
  return h;
}

int mo_pfhrgb_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints,
                  int n_kp, double radius, float *desc /* n_kp x 250 */)
{
  mo_grid *g = mo_grid_build(surface, n, (float)(radius * 0.5));
  const float r2 = (float)(radius * radius);
  const float d_pi = 1.0f / (2.0f * (float)M_PI);
  int cap = 4096;
  int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)cap);
  for (int k = 0; k < n_kp; ++k) {
    float *out = &desc[(size_t)k * 250];
    int cnt = mo_radius_search(g, keypoints[k].x, keypoints[k].y, keypoints[k].z, r2, idx, d2, cap);
    if (cnt > cap) {
      cap = cnt * 2;
      idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
      d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);
      cnt = mo_radius_search(g, keypoints[k].x, keypoints[k].y, keypoints[k].z, r2, idx, d2, cap);
    }
    for (int b = 0; b < 250; ++b) out[b] = 0.0f;
    const float hist_incr = 100.0f / (float)((size_t)cnt * ((size_t)cnt - 1) / 2);
    for (int i = 0; i < cnt; ++i)
      for (int j = 0; j < cnt; ++j) {
        if (i == j) continue;
        float f[7];
        rgb_pair_features(&surface[idx[i]], &normals[idx[i]], &surface[idx[j]], &normals[idx[j]], f);
        const int h1 = split_bin(PFH_SPLIT * ((f[0] + M_PI) * d_pi));
        const int h2 = split_bin(PFH_SPLIT * ((f[1] + 1.0) * 0.5));
        const int h3 = split_bin(PFH_SPLIT * ((f[2] + 1.0) * 0.5));
        out[h1 + PFH_SPLIT * h2 + PFH_SPLIT * PFH_SPLIT * h3] += hist_incr;
        const int h5 = split_bin(PFH_SPLIT * ((f[4] + 1.0) * 0.5));
        const int h6 = split_bin(PFH_SPLIT * ((f[5] + 1.0) * 0.5));
        const int h7 = split_bin(PFH_SPLIT * ((f[6] + 1.0) * 0.5));
        out[125 + h5 + PFH_SPLIT * h6 + PFH_SPLIT * PFH_SPLIT * h7] += hist_incr;
      }
  }
  free(idx); free(d2);
  mo_grid_free(g);
  return n_kp;
}

int mo_descriptors_pfhrgb(const mo_point *surface, const mo_normal *normals, int n,
                          mo_point *keypoints, int n_kp, double radius, float *desc)
{
  if (n_kp <= 0) return 0;
  mo_pfhrgb_raw(surface, normals, n, keypoints, n_kp, radius, desc);
  /* DefaultPointRepresentation<PFHRGBSignature250>::isValid: all bins finite; prune both (features.cpp:118-143) */
  int m = 0;
  for (int k = 0; k < n_kp; ++k) {
    int valid = 1;
    for (int b = 0; b < 250; ++b) if (!isfinite(desc[(size_t)k * 250 + b])) { valid = 0; break; }
    if (!valid) continue;
    if (m != k) {
      memmove(&desc[(size_t)m * 250], &desc[(size_t)k * 250], sizeof(float) * 250);
      keypoints[m] = keypoints[k];
    }
    ++m;
  }
  return m;
}

/*
 * computeLocalDescriptors(RSD): dispatch_descriptors.h:43 = pcl::RSDEstimation<PointXYZRGB, Normal,
 * PrincipalRadiiRSD> (field "r_min"; the point representation reads its 2 floats r_min, r_max).
 * PCL 1.8.1 features/impl/rsd.hpp: computeFeature -> pcl::computeRSD (surface, normals, nn_indices,
 * search_radius, nr_subdiv = 5, plane_radius = 0.2, out, false): the FIRST neighbour is the reference
 * point; for every other neighbour the angle between the two normal LINES (acos, folded to [0, pi/2])
 * and the distance go into 5 distance bins that keep the minimum and maximum angle; the radii are the
 * least-squares slopes d ~ r * angle of the two envelopes, capped at plane_radius, scaled by 1.1 / 0.9
 * and ordered.  Fewer than 2 neighbours: (0, 0).  Restatement choices: the neighbours come sorted by
 * (distance, index), so the reference point is the NEAREST surface point (PCL's unsorted tree hands
 * over whatever FLANN visits first); sqrt of the float sum of squares is the float overload
 * (pcl_macros.h includes <math.h>); a distance that lands in bin 5 (dist == max_dist after rounding:
 * out of bounds in PCL) is skipped.
 */
int mo_rsd_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints, int n_kp,
               double radius, float *desc /* n_kp x 2: r_min, r_max */)
{
  const int nr_subdiv = 5;
  const double plane_radius = 0.2, max_dist = radius;
  mo_grid *g = mo_grid_build(surface, n, (float)(radius * 0.5));
  const float r2 = (float)(radius * radius);
  int cap = 4096;
  int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)cap);
  for (int k = 0; k < n_kp; ++k) {
    float *out = &desc[(size_t)k * 2];
    int cnt = mo_radius_search(g, keypoints[k].x, keypoints[k].y, keypoints[k].z, r2, idx, d2, cap);
    if (cnt > cap) {
      cap = cnt * 2;
      idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
      d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);
      cnt = mo_radius_search(g, keypoints[k].x, keypoints[k].y, keypoints[k].z, r2, idx, d2, cap);
    }
    if (cnt < 2) { out[0] = out[1] = 0.0f; continue; }
    double lo[5], hi[5];
    lo[0] = hi[0] = 0.0;
    for (int di = 1; di < nr_subdiv; ++di) { lo[di] = +1.7976931348623157e308; hi[di] = -1.7976931348623157e308; }
    const mo_point *p0 = &surface[idx[0]];
    const mo_normal *n0 = &normals[idx[0]];
    for (int i = 1; i < cnt; ++i) {
      const mo_point *p = &surface[idx[i]];
      const mo_normal *nv = &normals[idx[i]];
      double cosine = (double)(nv->nx * n0->nx + nv->ny * n0->ny + nv->nz * n0->nz);
      if (cosine > 1) cosine = 1;
      if (cosine < -1) cosine = -1;
      double angle = acos(cosine);
      if (angle > M_PI / 2) angle = M_PI - angle;
      const double dist = (double)sqrtf((p->x - p0->x) * (p->x - p0->x) + (p->y - p0->y) * (p->y - p0->y) + (p->z - p0->z) * (p->z - p0->z));
      if (dist > max_dist) continue;
      const int bin_d = floor_to_int(nr_subdiv * dist / max_dist);
      if (bin_d < 0 || bin_d >= nr_subdiv) continue;
      if (lo[bin_d] > angle) lo[bin_d] = angle;
      if (hi[bin_d] < angle) hi[bin_d] = angle;
    }
    double Amint_Amin = 0, Amint_d = 0, Amaxt_Amax = 0, Amaxt_d = 0;
    for (int di = 0; di < nr_subdiv; ++di)
      if (hi[di] >= 0) {
        const double p_min = lo[di], p_max = hi[di];
        const double f = (di + 0.5) * max_dist / nr_subdiv;
        Amint_Amin += p_min * p_min;
        Amint_d += p_min * f;
        Amaxt_Amax += p_max * p_max;
        Amaxt_d += p_max * f;
      }
    float min_radius = Amint_Amin == 0.0f ? (float)plane_radius : (float)fmin(Amint_d / Amint_Amin, plane_radius);
    float max_radius = Amaxt_Amax == 0.0f ? (float)plane_radius : (float)fmin(Amaxt_d / Amaxt_Amax, plane_radius);
    min_radius *= 1.1f;
    max_radius *= 0.9f;
    if (min_radius < max_radius) { out[0] = min_radius; out[1] = max_radius; }
    else { out[1] = min_radius; out[0] = max_radius; }
  }
  free(idx); free(d2);
  mo_grid_free(g);
  return n_kp;
}

int mo_descriptors_rsd(const mo_point *surface, const mo_normal *normals, int n, mo_point *keypoints, int n_kp,
                       double radius, float *desc)
{
  if (n_kp <= 0) return 0;
  mo_rsd_raw(surface, normals, n, keypoints, n_kp, radius, desc);
  /* DefaultPointRepresentation<PrincipalRadiiRSD>: 2 floats, both finite */
  int m = 0;
  for (int k = 0; k < n_kp; ++k) {
    if (!isfinite(desc[(size_t)k * 2]) || !isfinite(desc[(size_t)k * 2 + 1])) continue;
    if (m != k) { desc[(size_t)m * 2] = desc[(size_t)k * 2]; desc[(size_t)m * 2 + 1] = desc[(size_t)k * 2 + 1]; keypoints[m] = keypoints[k]; }
    ++m;
  }
  return m;
}

int mo_descriptors_pfh(const mo_point *surface, const mo_normal *normals, int n,
                       mo_point *keypoints, int n_kp, double radius, float *desc)
{
  if (n_kp <= 0) return 0;
  mo_pfh_raw(surface, normals, n, keypoints, n_kp, radius, desc);
  /* DefaultPointRepresentation<PFHSignature125>::isValid: all bins finite; prune both (features.cpp:118-143) */
  int m = 0;
  for (int k = 0; k < n_kp; ++k) {
    int valid = 1;
    for (int b = 0; b < 125; ++b) if (!isfinite(desc[(size_t)k * 125 + b])) { valid = 0; break; }
    if (!valid) continue;
    if (m != k) {
      memmove(&desc[(size_t)m * 125], &desc[(size_t)k * 125], sizeof(float) * 125);
      keypoints[m] = keypoints[k];
    }
    ++m;
  }
  return m;
}

int mo_descriptors_fpfh(const mo_point *surface, const mo_normal *normals, int n,
                        mo_point *keypoints, int n_kp, double radius, float *desc)
{
  if (n_kp <= 0) return 0;
  mo_fpfh_raw(surface, normals, n, keypoints, n_kp, radius, desc, NULL, NULL);
  /* DefaultPointRepresentation<FPFHSignature33>::isValid: all 33 floats finite; prune both */
  int m = 0;
  for (int k = 0; k < n_kp; ++k) {
    int valid = 1;
    for (int b = 0; b < 33; ++b) if (!isfinite(desc[(size_t)k * 33 + b])) { valid = 0; break; }
    if (!valid) continue;
    if (m != k) {
      memmove(&desc[(size_t)m * 33], &desc[(size_t)k * 33], sizeof(float) * 33);
      keypoints[m] = keypoints[k];
    }
    ++m;
  }
  return m;
}
