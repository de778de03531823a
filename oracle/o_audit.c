/*
 * o_audit.c -- exposure census for the audit list of DESIGN.md section 4 (TEST / EVIDENCE INFRASTRUCTURE).
 *
 * The oracle restates PCL 1.8.1 from memory; five recalled details cannot be checked without PCL (rows 1, 2, 10, 11, 13 of
 * the audit list).  What CAN be measured without it is how much of a workload's output would move if such a detail were
 * recalled wrongly -- scripts/audit_exposure.py counts that on the BASELINE workloads with the hooks below
 * (and oracle/audit_sort.cpp for libstdc++'s std::sort).
 *   R/src/features.cpp:19-40, 171-176 (VoxelGrid, radius searches), R/src/matching.cpp:50-75 (descriptor k-NN),
 *   :204-220 (ICP -> TransformationEstimationSVD -> Eigen::umeyama).
 */
#include "mm3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Row 1: radiusSearch's order among EQUAL squared distances (the oracle: ascending index; FLANN: its leaf walk).
 * out[0] neighbourhoods, [1] neighbourhoods that hold at least one pair of neighbours at exactly the same float d2,
 * [2] such adjacent pairs in total, [3] neighbours in total. */
void mo_audit_radius_ties(const mo_point *pts, int n, double radius, long long out[4])
{
  const float r2 = (float)(radius * radius);
  mo_grid *g = mo_grid_build(pts, n, (float)(radius * 0.5));
  long long nb = 0, tied_nb = 0, tied_pairs = 0, total = 0;
#pragma omp parallel num_threads(mo_get_threads()) reduction(+ : nb, tied_nb, tied_pairs, total)
  {
    int cap = 4096;
    int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
    float *d2 = (float *)malloc(sizeof(float) * (size_t)cap);
#pragma omp for schedule(dynamic, 512)
    for (int i = 0; i < n; ++i) {
      int cnt = mo_radius_search(g, pts[i].x, pts[i].y, pts[i].z, r2, idx, d2, cap);
      if (cnt > cap) {
        cap = cnt * 2;
        idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
        d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);
        cnt = mo_radius_search(g, pts[i].x, pts[i].y, pts[i].z, r2, idx, d2, cap);
      }
      int t = 0;
      for (int j = 1; j < cnt; ++j) t += d2[j] == d2[j - 1];
      ++nb; total += cnt; tied_pairs += t; tied_nb += t > 0;
    }
    free(idx); free(d2);
  }
  mo_grid_free(g);
  out[0] = nb; out[1] = tied_nb; out[2] = tied_pairs; out[3] = total;
}

/* Row 10: FLANN's `L2` functor (four-way unrolled: result += d0 d0 + d1 d1 + d2 d2 + d3 d3 per group of four, then the
 * tail one by one) in place of `L2_Simple`'s "result += diff * diff" -- PCL's KdTreeFLANN is recalled to use L2_Simple. */
void mo_audit_desc_knn_unrolled(const float *a, int na, const float *b, int nb, int dim, int k, int *idx, float *d2)
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
      int d = 0;
      for (; d + 4 <= dim; d += 4) {
        const float d0 = av[d] - bv[d], d1 = av[d + 1] - bv[d + 1], d2_ = av[d + 2] - bv[d + 2], d3 = av[d + 3] - bv[d + 3];
        r += d0 * d0 + d1 * d1 + d2_ * d2_ + d3 * d3;
      }
      for (; d < dim; ++d) { const float df = av[d] - bv[d]; r += df * df; }
      int pos = m;
      if (m == k) {
        if (!(r < td[k - 1])) continue;
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

/* Row 13: Eigen::umeyama's float sums under three summation orders -- 0 sequential (the oracle's and the device's),
 * 1 pairwise (recursive halves), 2 eight interleaved partial sums added at the end (a vectorised reduction). */
static float sum_order(const float *v, int n, int order)
{
  if (n <= 0) return 0.0f;
  if (order == 0) { float s = 0.0f; for (int i = 0; i < n; ++i) s += v[i]; return s; }
  if (order == 1) {
    if (n <= 8) { float s = 0.0f; for (int i = 0; i < n; ++i) s += v[i]; return s; }
    const int h = n / 2;
    return sum_order(v, h, 1) + sum_order(v + h, n - h, 1);
  }
  float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int i = 0;
  for (; i + 8 <= n; i += 8)
    for (int l = 0; l < 8; ++l) p[l] += v[i + l];
  float s = ((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7]));
  for (; i < n; ++i) s += v[i];
  return s;
}

void mo_umeyama_f64(const double *src, const double *dst, int n, double T[16]);

/* T (column-major 4x4) of the n pairs src[i] -> dst[i] with every float sum taken in the given order.  The 3x3 SVD and
 * the rest are mo_umeyama_f32's (through a three-point problem is not possible: the core is static there, so the means
 * and sigma are formed here and handed to the same algebra via mo_umeyama_core_f32 in o_linalg.c). */
void mo_umeyama_core_f32(const float sg[9], const float sm[3], const float dm[3], float one_over_n, float T[16]);

void mo_audit_umeyama_order(const float *src, const float *dst, int n, int order, float T[16])
{
  float *tmp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  float sm[3], dm[3];
  const float one_over_n = 1.0f / (float)n;
  for (int a = 0; a < 3; ++a) {
    for (int i = 0; i < n; ++i) tmp[i] = src[i * 3 + a];
    sm[a] = sum_order(tmp, n, order) * one_over_n;
    for (int i = 0; i < n; ++i) tmp[i] = dst[i * 3 + a];
    dm[a] = sum_order(tmp, n, order) * one_over_n;
  }
  float sg[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      for (int i = 0; i < n; ++i) tmp[i] = (dst[i * 3 + r] - dm[r]) * (src[i * 3 + c] - sm[c]);
      sg[r * 3 + c] = sum_order(tmp, n, order);
    }
  free(tmp);
  mo_umeyama_core_f32(sg, sm, dm, one_over_n, T);
}

/* the correspondences of ONE ICP iteration (R/src/matching.cpp:204-220 -> icp.hpp: every source point transformed by
 * `guess`, its nearest target point kept when d2 <= max_corr^2): src_out / dst_out hold 3 floats per kept pair. */
int mo_audit_icp_correspondences(const mo_point *src, int ns, const mo_point *tgt, int nt, const float guess[16], double max_corr,
                                 float *src_out, float *dst_out)
{
  mo_grid *g = mo_grid_build(tgt, nt, (float)(max_corr * 0.5));
  const float max2 = (float)(max_corr * max_corr);
  int n = 0;
  for (int i = 0; i < ns; ++i) {
    const float x = src[i].x, y = src[i].y, z = src[i].z;
    const float px = ((guess[0] * x + guess[4] * y) + guess[8] * z) + guess[12];
    const float py = ((guess[1] * x + guess[5] * y) + guess[9] * z) + guess[13];
    const float pz = ((guess[2] * x + guess[6] * y) + guess[10] * z) + guess[14];
    int id; float d2;
    if (mo_knn_search(g, px, py, pz, 1, max2, &id, &d2) < 1 || !(d2 <= max2)) continue;
    src_out[n * 3] = px; src_out[n * 3 + 1] = py; src_out[n * 3 + 2] = pz;
    dst_out[n * 3] = tgt[id].x; dst_out[n * 3 + 1] = tgt[id].y; dst_out[n * 3 + 2] = tgt[id].z;
    ++n;
  }
  mo_grid_free(g);
  return n;
}
