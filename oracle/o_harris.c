/*
 * o_harris.c -- detectKeypoints(HARRIS) restated (TEST INFRASTRUCTURE).
 *
 * R/src/features.cpp:64-83: pcl::HarrisKeypoint3D<PointXYZRGB, PointXYZI> with setNormals(normals),
 *   setNonMaxSupression(true), setRefine(true), setThreshold(float(threshold)), setRadius(float(radius));
 *   the result is copied with pcl::copyPointCloud (xyz only, rgb = 0).  R/src/map_merging.cpp:231-233 passes
 *   params.keypoint_threshold and params.normal_radius.
 * PCL 1.8.1 keypoints/impl/harris_3d.hpp: detectKeypoints, responseHarris (method HARRIS, the default),
 *   calculateNormalCovar (the __SSE__ branch: the sums are DIVIDED by float(count)), refineCorners;
 *   common/eigen.h invert3x3SymMatrix.
 *
 * Restatement choices: neighbours come sorted by (distance, index) (the Keypoint base class builds its
 * KdTree with sorted = false, so PCL's float sums follow FLANN's traversal order, which is not
 * reproducible); 3-term products / sums are evaluated left to right; keypoints are emitted in index
 * order (PCL's loop is an OpenMP parallel for with a critical push_back).
 */
#include "mm3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int *idx; float *d2; int cap; } nb_buf;

static int search(const mo_grid *g, float x, float y, float z, float r2, nb_buf *b)
{
  int cnt = mo_radius_search(g, x, y, z, r2, b->idx, b->d2, b->cap);
  if (cnt > b->cap) {
    b->cap = cnt * 2;
    b->idx = (int *)realloc(b->idx, sizeof(int) * (size_t)b->cap);
    b->d2 = (float *)realloc(b->d2, sizeof(float) * (size_t)b->cap);
    cnt = mo_radius_search(g, x, y, z, r2, b->idx, b->d2, b->cap);
  }
  return cnt;
}

/* responseHarris for every point: 0.04 + det(C) - 0.04 tr(C)^2 of the mean outer product of the
 * neighbours' normals; 0 for a non-finite point or a zero trace */
void mo_harris_response(const mo_point *in, const mo_normal *normals, int n, double radius, float *response)
{
  const double sr = (double)(float)radius;          /* setRadius(float(radius)) */
  const float r2 = (float)(sr * sr);
  mo_grid *g = mo_grid_build(in, n, (float)(sr * 0.5));
  nb_buf b = {(int *)malloc(sizeof(int) * 4096), (float *)malloc(sizeof(float) * 4096), 4096};
  for (int i = 0; i < n; ++i) {
    response[i] = 0.0f;
    if (!(isfinite(in[i].x) && isfinite(in[i].y) && isfinite(in[i].z))) continue;
    const int cnt = search(g, in[i].x, in[i].y, in[i].z, r2, &b);
    float xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    unsigned count = 0;
    for (int k = 0; k < cnt; ++k) {
      const mo_normal *nv = &normals[b.idx[k]];
      if (!isfinite(nv->nx)) continue;
      xx += nv->nx * nv->nx; xy += nv->nx * nv->ny; xz += nv->nx * nv->nz;
      yy += nv->ny * nv->ny; yz += nv->ny * nv->nz;
      zz += nv->nz * nv->nz;
      ++count;
    }
    if (count > 0) {
      const float c = (float)count;
      xx /= c; xy /= c; xz /= c; yy /= c; yz /= c; zz /= c;
    } else {
      xx = xy = xz = yy = yz = zz = 0.0f;
    }
    const float trace = xx + yy + zz;
    if (trace != 0) {
      const float det = xx * yy * zz + 2.0f * xy * xz * yz - xz * xz * yy - xy * xy * zz - yz * yz * xx;
      response[i] = 0.04f + det - 0.04f * trace * trace;
    }
  }
  free(b.idx); free(b.d2);
  mo_grid_free(g);
}

/* refineCorners for one corner (position updated in place); returns the iterations used */
static int refine_corner(const mo_grid *g, const mo_point *in, const mo_normal *normals, float r2, nb_buf *b, float c[3])
{
  unsigned iterations = 0;
  float diff;
  do {
    float N[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Np[3] = {0, 0, 0};
    const float old[3] = {c[0], c[1], c[2]};
    const int cnt = search(g, c[0], c[1], c[2], r2, b);
    for (int k = 0; k < cnt; ++k) {
      const mo_normal *nv = &normals[b->idx[k]];
      if (!isfinite(nv->nx)) continue;
      const float nn[3] = {nv->nx, nv->ny, nv->nz};
      const mo_point *p = &in[b->idx[k]];
      for (int r = 0; r < 3; ++r) {
        const float t0 = nn[r] * nn[0], t1 = nn[r] * nn[1], t2 = nn[r] * nn[2];
        N[r][0] += t0; N[r][1] += t1; N[r][2] += t2;
        Np[r] += t0 * p->x + t1 * p->y + t2 * p->z;
      }
    }
    /* invert3x3SymMatrix: a b c / b d e / c e f = coeff 0 1 2 / 1 4 5 / 2 5 8 */
    const float a = N[0][0], bb = N[0][1], cc = N[0][2], d = N[1][1], e = N[1][2], f = N[2][2];
    const float fd_ee = d * f - e * e;
    const float ce_bf = cc * e - bb * f;
    const float be_cd = bb * e - cc * d;
    const float det = a * fd_ee + bb * ce_bf + cc * be_cd;
    if (det != 0) {
      float I[3][3];
      I[0][0] = fd_ee; I[0][1] = I[1][0] = ce_bf; I[0][2] = I[2][0] = be_cd;
      I[1][1] = a * f - cc * cc;
      I[1][2] = I[2][1] = bb * cc - a * e;
      I[2][2] = a * d - bb * bb;
      for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) I[r][q] /= det;
      for (int r = 0; r < 3; ++r) c[r] = I[r][0] * Np[0] + I[r][1] * Np[1] + I[r][2] * Np[2];
    }
    const float dx = c[0] - old[0], dy = c[1] - old[1], dz = c[2] - old[2];
    diff = dx * dx + dy * dy + dz * dz;
  } while (diff > 1e-6 && ++iterations < 10);
  return (int)iterations;
}

/* detectKeypoints(HARRIS).  Keypoints (refined xyz, rgba = 0) malloc'ed into *out; optional outputs:
 * the kept point indices (malloc'ed) and the unrefined responses of all points (n floats, caller's). */
int mo_keypoints_harris(const mo_point *in, const mo_normal *normals, int n, double threshold, double radius,
                        mo_point **out, int **kept_idx, float *response_out)
{
  float *response = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  mo_harris_response(in, normals, n, radius, response);
  if (response_out) memcpy(response_out, response, sizeof(float) * (size_t)n);
  const double sr = (double)(float)radius;
  const float r2 = (float)(sr * sr);
  const float thr = (float)threshold;
  mo_grid *g = mo_grid_build(in, n, (float)(sr * 0.5));
  nb_buf b = {(int *)malloc(sizeof(int) * 4096), (float *)malloc(sizeof(float) * 4096), 4096};
  mo_point *kp = (mo_point *)malloc(sizeof(mo_point) * (size_t)(n > 0 ? n : 1));
  int *kept = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int nk = 0;
  for (int i = 0; i < n; ++i) {
    if (!(isfinite(in[i].x) && isfinite(in[i].y) && isfinite(in[i].z)) || !isfinite(response[i]) || response[i] < thr)
      continue;
    const int cnt = search(g, in[i].x, in[i].y, in[i].z, r2, &b);
    int is_maxima = 1;
    for (int k = 0; k < cnt; ++k)
      if (response[i] < response[b.idx[k]]) { is_maxima = 0; break; }
    if (!is_maxima) continue;
    kp[nk].x = in[i].x; kp[nk].y = in[i].y; kp[nk].z = in[i].z; kp[nk].rgba = 0;
    kept[nk] = i;
    ++nk;
  }
  for (int k = 0; k < nk; ++k) {
    float c[3] = {kp[k].x, kp[k].y, kp[k].z};
    refine_corner(g, in, normals, r2, &b, c);
    kp[k].x = c[0]; kp[k].y = c[1]; kp[k].z = c[2];
  }
  free(b.idx); free(b.d2); free(response);
  mo_grid_free(g);
  *out = kp;
  if (kept_idx) *kept_idx = kept; else free(kept);
  return nk;
}
