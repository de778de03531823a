/*
 * o_filters.c -- downSample / removeOutliers restated (TEST INFRASTRUCTURE).
 *
 * downSample      R/src/features.cpp:17-27  -> pcl::VoxelGrid<PointXYZRGB>
 *                 (PCL 1.8.1 filters/impl/voxel_grid.hpp applyFilter,
 *                  common/impl/centroid.hpp CentroidPoint / AccumulatorXYZ / AccumulatorRGBA)
 * removeOutliers  R/src/features.cpp:31-43  -> pcl::RadiusOutlierRemoval
 *                 (PCL 1.8.1 filters/impl/radius_outlier_removal.hpp applyFilterIndices,
 *                  dense fast path: k = min_pts+1 nearest, keep iff the k-th d2 <= r*r)
 */
#include "mm3d_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t idx; int pt; } vox_key;
static int vox_cmp(const void *a, const void *b)
{
  const vox_key *x = (const vox_key *)a, *y = (const vox_key *)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  /* std::sort is not stable in the reference, so the order inside a voxel is unspecified there;
   * the oracle fixes it to ascending input index. */
  return (x->pt > y->pt) - (x->pt < y->pt);
}

int mo_downsample(const mo_point *in, int n, double resolution, mo_point *out)
{
  if (n <= 0) return 0;
  const float leaf = (float)resolution;         /* setLeafSize(float(resolution), ...) */
  const float inv = 1.0f / leaf;                /* inverse_leaf_size_ = Array4f::Ones() / leaf_size_ */
  /* getMinMax3D over finite points (cloud treated as !is_dense: non-finite points are skipped) */
  float mn[3], mx[3];
  int any = 0;
  for (int i = 0; i < n; ++i) {
    if (!isfinite(in[i].x) || !isfinite(in[i].y) || !isfinite(in[i].z)) continue;
    float v[3] = {in[i].x, in[i].y, in[i].z};
    for (int a = 0; a < 3; ++a) {
      if (!any || v[a] < mn[a]) mn[a] = v[a];
      if (!any || v[a] > mx[a]) mx[a] = v[a];
    }
    any = 1;
  }
  if (!any) return 0;
  /* overflow guard: int64 dx = (max-min)*inv + 1 ...; too many voxels => output = input */
  int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1,
          dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) {
    memcpy(out, in, sizeof(mo_point) * (size_t)n);
    return n;
  }
  int min_b[3], max_b[3], div_b[3];
  for (int a = 0; a < 3; ++a) {
    min_b[a] = (int)floorf(mn[a] * inv);
    max_b[a] = (int)floorf(mx[a] * inv);
    div_b[a] = max_b[a] - min_b[a] + 1;
  }
  const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
  vox_key *keys = (vox_key *)malloc(sizeof(vox_key) * (size_t)n);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (!isfinite(in[i].x) || !isfinite(in[i].y) || !isfinite(in[i].z)) continue;
    int ijk0 = (int)(floorf(in[i].x * inv) - (float)min_b[0]);
    int ijk1 = (int)(floorf(in[i].y * inv) - (float)min_b[1]);
    int ijk2 = (int)(floorf(in[i].z * inv) - (float)min_b[2]);
    keys[m].idx = (uint32_t)(ijk0 + ijk1 * mul1 + ijk2 * mul2);
    keys[m].pt = i;
    ++m;
  }
  qsort(keys, (size_t)m, sizeof(vox_key), vox_cmp);
  int nout = 0;
  for (int b = 0; b < m;) {
    int e = b + 1;
    while (e < m && keys[e].idx == keys[b].idx) ++e;
    /* CentroidPoint<PointXYZRGB>: float sums of x,y,z and of r,g,b,a; get(): xyz/n, channels
     * truncated to uint32 then packed a<<24|r<<16|g<<8|b. */
    float sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0, sa = 0;
    for (int j = b; j < e; ++j) {
      const mo_point *p = &in[keys[j].pt];
      sx += p->x; sy += p->y; sz += p->z;
      sr += (float)((p->rgba >> 16) & 255u);
      sg += (float)((p->rgba >> 8) & 255u);
      sb += (float)(p->rgba & 255u);
      sa += (float)((p->rgba >> 24) & 255u);
    }
    float cnt = (float)(e - b);
    mo_point *o = &out[nout++];
    o->x = sx / cnt; o->y = sy / cnt; o->z = sz / cnt;
    o->rgba = ((uint32_t)(sa / cnt) << 24) | ((uint32_t)(sr / cnt) << 16) |
              ((uint32_t)(sg / cnt) << 8) | (uint32_t)(sb / cnt);
    b = e;
  }
  free(keys);
  return nout;
}

int mo_remove_outliers(const mo_point *in, int n, double radius, int min_neighbors,
                       mo_point *out)
{
  if (n <= 0) return 0;
  mo_grid *g = mo_grid_build(in, n, (float)(radius * 0.5));
  const int mean_k = min_neighbors + 1;             /* k includes the query point */
  const double nn_dists_max = radius * radius;      /* double, compared with float d2 */
  unsigned char *keep_flag = (unsigned char *)malloc((size_t)n);
#pragma omp parallel num_threads(mo_get_threads())
  {
    int *idx = (int *)malloc(sizeof(int) * (size_t)(mean_k > 0 ? mean_k : 1));
    float *d2 = (float *)malloc(sizeof(float) * (size_t)(mean_k > 0 ? mean_k : 1));
#pragma omp for schedule(dynamic, 1024)
    for (int i = 0; i < n; ++i) {
      int keep;
      if (mean_k <= 0) {
        keep = 1;
      } else {
        int k = mo_knn_search(g, in[i].x, in[i].y, in[i].z, mean_k, INFINITY, idx, d2);
        if (k == mean_k) keep = !(nn_dists_max < (double)d2[k - 1]);
        else keep = 0;
      }
      keep_flag[i] = (unsigned char)keep;
    }
    free(idx); free(d2);
  }
  int nout = 0;
  for (int i = 0; i < n; ++i)           /* input order preserved */
    if (keep_flag[i]) out[nout++] = in[i];
  free(keep_flag);
  mo_grid_free(g);
  return nout;
}
