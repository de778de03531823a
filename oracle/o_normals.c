/*
 * o_normals.c -- computeSurfaceNormals restated (TEST INFRASTRUCTURE).
 *
 * R/src/features.cpp:168-179 -> pcl::NormalEstimation<PointXYZRGB, Normal>
 *   PCL 1.8.1 features/impl/normal_3d.hpp computeFeature (dense path),
 *   features/normal_3d.h computePointNormal / flipNormalTowardsViewpoint,
 *   common/impl/centroid.hpp computeMeanAndCovarianceMatrix (float raw moments,
 *   neighbour order = FLANN's sorted radius result),
 *   common/impl/eigen.hpp eigen33 / computeRoots / computeRoots2,
 *   features/impl/feature.hpp solvePlaneParameters.
 */
#include "mm3d_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

/* closed-form roots of the characteristic cubic of a symmetric 3x3 (float) */
static void compute_roots2(float b, float c, float roots[3])
{
  roots[0] = 0.0f;
  float d = (float)(b * b - 4.0 * c);
  if (d < 0.0) d = 0.0f;
  float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

static void compute_roots(const float m[9] /* row-major symmetric */, float roots[3])
{
#define A(r, c) m[(r) * 3 + (c)]
  float c0 = A(0, 0) * A(1, 1) * A(2, 2) + 2.0f * A(0, 1) * A(0, 2) * A(1, 2) -
             A(0, 0) * A(1, 2) * A(1, 2) - A(1, 1) * A(0, 2) * A(0, 2) - A(2, 2) * A(0, 1) * A(0, 1);
  float c1 = A(0, 0) * A(1, 1) - A(0, 1) * A(0, 1) + A(0, 0) * A(2, 2) - A(0, 2) * A(0, 2) +
             A(1, 1) * A(2, 2) - A(1, 2) * A(1, 2);
  float c2 = A(0, 0) + A(1, 1) + A(2, 2);
#undef A
  if (fabsf(c0) < FLT_EPSILON) {
    compute_roots2(c2, c1, roots);
  } else {
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = sqrtf(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    float rho = sqrtf(-a_over_3);
    float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
    float cos_theta = cosf(theta);
    float sin_theta = sinf(theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    if (roots[1] >= roots[2]) {
      t = roots[1]; roots[1] = roots[2]; roots[2] = t;
      if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    }
    if (roots[0] <= 0.0f) compute_roots2(c2, c1, roots);
  }
}

static inline void cross3(const float *a, const float *b, float *o)
{
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* smallest eigenpair: pcl::eigen33(mat, eigenvalue, eigenvector) */
void mo_eigen33_smallest(const float cov[9], float *eigenvalue, float vec[3])
{
  float scale = 0.0f;
  for (int i = 0; i < 9; ++i) { float a = fabsf(cov[i]); if (a > scale) scale = a; }
  if (scale <= FLT_MIN) scale = 1.0f;
  float m[9];
  for (int i = 0; i < 9; ++i) m[i] = cov[i] / scale;
  float roots[3];
  compute_roots(m, roots);
  *eigenvalue = roots[0] * scale;
  m[0] -= roots[0]; m[4] -= roots[0]; m[8] -= roots[0];
  float v1[3], v2[3], v3[3];
  cross3(&m[0], &m[3], v1);
  cross3(&m[0], &m[6], v2);
  cross3(&m[3], &m[6], v3);
  float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const float *v; float l;
  if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
  else { v = v3; l = l3; }
  float s = sqrtf(l);
  vec[0] = v[0] / s; vec[1] = v[1] / s; vec[2] = v[2] / s;
}

/* all three eigenvalues ascending (used by the RANSAC sample-distance threshold) */
void mo_eigen33_values(const float cov[9], float evals[3])
{
  float scale = 0.0f;
  for (int i = 0; i < 9; ++i) { float a = fabsf(cov[i]); if (a > scale) scale = a; }
  if (scale <= FLT_MIN) scale = 1.0f;
  float m[9];
  for (int i = 0; i < 9; ++i) m[i] = cov[i] / scale;
  compute_roots(m, evals);
  for (int i = 0; i < 3; ++i) evals[i] *= scale;
}

/* computeMeanAndCovarianceMatrix(cloud, indices, cov, centroid): float raw moments */
void mo_mean_cov(const mo_point *pts, const int *idx, int cnt, float cov[9], float centroid[3])
{
  float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < cnt; ++j) {
    const mo_point *p = &pts[idx ? idx[j] : j];
    a[0] += p->x * p->x; a[1] += p->x * p->y; a[2] += p->x * p->z;
    a[3] += p->y * p->y; a[4] += p->y * p->z; a[5] += p->z * p->z;
    a[6] += p->x; a[7] += p->y; a[8] += p->z;
  }
  float fc = (float)cnt;
  for (int i = 0; i < 9; ++i) a[i] /= fc;
  centroid[0] = a[6]; centroid[1] = a[7]; centroid[2] = a[8];
  cov[0] = a[0] - a[6] * a[6];
  cov[1] = a[1] - a[6] * a[7];
  cov[2] = a[2] - a[6] * a[8];
  cov[4] = a[3] - a[7] * a[7];
  cov[5] = a[4] - a[7] * a[8];
  cov[8] = a[5] - a[8] * a[8];
  cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
}

void mo_normals(const mo_point *in, int n, double radius, mo_normal *out)
{
  if (n <= 0) return;
  mo_grid *g = mo_grid_build(in, n, (float)(radius * 0.5));
  const float r2 = (float)(radius * radius);   /* KdTreeFLANN::radiusSearch: float(radius*radius) */
#pragma omp parallel num_threads(mo_get_threads())
  {
  int cap = 4096;
  int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)cap);
#pragma omp for schedule(dynamic, 1024)
  for (int i = 0; i < n; ++i) {
    int cnt = mo_radius_search(g, in[i].x, in[i].y, in[i].z, r2, idx, d2, cap);
    if (cnt > cap) {
      cap = cnt * 2;
      idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
      d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);
      cnt = mo_radius_search(g, in[i].x, in[i].y, in[i].z, r2, idx, d2, cap);
    }
    mo_normal *o = &out[i];
    if (cnt < 3) {     /* searchForNeighbors()==0 or computePointNormal's indices.size() < 3 */
      o->nx = o->ny = o->nz = o->curvature = NAN;
      continue;
    }
    float cov[9], centroid[3], ev, vec[3];
    mo_mean_cov(in, idx, cnt, cov, centroid);
    mo_eigen33_smallest(cov, &ev, vec);
    float eig_sum = cov[0] + cov[4] + cov[8];
    o->curvature = (eig_sum != 0.0f) ? fabsf(ev / eig_sum) : 0.0f;
    /* flipNormalTowardsViewpoint(point, 0,0,0, ...) */
    float vx = 0.0f - in[i].x, vy = 0.0f - in[i].y, vz = 0.0f - in[i].z;
    float cos_theta = vx * vec[0] + vy * vec[1] + vz * vec[2];
    if (cos_theta < 0.0f) { vec[0] *= -1.0f; vec[1] *= -1.0f; vec[2] *= -1.0f; }
    o->nx = vec[0]; o->ny = vec[1]; o->nz = vec[2];
  }
  free(idx); free(d2);
  }
  mo_grid_free(g);
}
