/*
 * o_sift.c -- detectKeypoints(SIFT) restated (TEST INFRASTRUCTURE).
 *
 * R/src/features.cpp:45-62,85-96:
 *   pcl::SIFTKeypoint<PointXYZRGB, PointWithScale>; setScales(resolution, 3, 3);
 *   setMinimumContrast(threshold); result copied with pcl::copyPointCloud (xyz only, rgb = 0).
 * PCL 1.8.1 keypoints/impl/sift_keypoint.hpp: detectKeypoints, detectKeypointsForOctave,
 *   computeScaleSpace, findScaleSpaceExtrema; keypoints/sift_keypoint.h
 *   SIFTKeypointFieldSelector<PointXYZRGB>: (299*r + 587*g + 114*b) / 1000.0f.
 */
#include "mm3d_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float intensity(const mo_point *p)
{
  int r = (int)((p->rgba >> 16) & 255u), g = (int)((p->rgba >> 8) & 255u), b = (int)(p->rgba & 255u);
  return (float)(299 * r + 587 * g + 114 * b) / 1000.0f;
}

typedef struct { mo_point *pts; float *scales; int n, cap; } kp_vec;
static void kp_push(kp_vec *v, float x, float y, float z, float scale)
{
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 1024;
    v->pts = (mo_point *)realloc(v->pts, sizeof(mo_point) * (size_t)v->cap);
    v->scales = (float *)realloc(v->scales, sizeof(float) * (size_t)v->cap);
  }
  v->pts[v->n].x = x; v->pts[v->n].y = y; v->pts[v->n].z = z; v->pts[v->n].rgba = 0;
  v->scales[v->n] = scale;
  v->n++;
}

static void detect_octave(const mo_point *cloud, int n, float base_scale, int nr_scales_per_octave,
                          float min_contrast, kp_vec *out)
{
  const int ns = nr_scales_per_octave + 3;
  float *scales = (float *)malloc(sizeof(float) * (size_t)ns);
  for (int i = 0; i < ns; ++i)
    scales[i] = base_scale * powf(2.0f, (1.0f * (float)i - 1.0f) / (float)nr_scales_per_octave);
  const int nd = ns - 1;                       /* DoG columns */
  float *dog = (float *)malloc(sizeof(float) * (size_t)n * (size_t)nd);
  const float max_radius = 3.0f * scales[ns - 1];
  /* tree.radiusSearch(i_point, max_radius): KdTreeFLANN casts radius*radius (double) to float */
  const float r2 = (float)((double)max_radius * (double)max_radius);
  mo_grid *g = mo_grid_build(cloud, n, max_radius * 0.5f);

  /* computeScaleSpace */
#pragma omp parallel num_threads(mo_get_threads())
  {
  int cap = 4096;
  int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)cap);
#pragma omp for schedule(dynamic, 512)
  for (int i = 0; i < n; ++i) {
    int cnt = mo_radius_search(g, cloud[i].x, cloud[i].y, cloud[i].z, r2, idx, d2, cap);
    if (cnt > cap) {
      cap = cnt * 2;
      idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
      d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);
      cnt = mo_radius_search(g, cloud[i].x, cloud[i].y, cloud[i].z, r2, idx, d2, cap);
    }
    float filter_response = 0.0f, previous_filter_response;
    for (int s = 0; s < ns; ++s) {
      float sigma_sqr = powf(scales[s], 2.0f);
      float numerator = 0.0f, denominator = 0.0f;
      for (int j = 0; j < cnt; ++j) {
        float value = intensity(&cloud[idx[j]]);
        float dist_sqr = d2[j];
        if (dist_sqr <= 9 * sigma_sqr) {
          float w = expf(-0.5f * dist_sqr / sigma_sqr);
          numerator += value * w;
          denominator += w;
        } else {
          break;   /* sorted: beyond 3 sigma */
        }
      }
      previous_filter_response = filter_response;
      filter_response = numerator / denominator;
      if (s > 0) dog[(size_t)i * nd + (s - 1)] = filter_response - previous_filter_response;
    }
  }
  free(idx); free(d2);
  }
  /* findScaleSpaceExtrema: the tests run per point (in parallel for baseline B2), the keypoints are
   * emitted afterwards in point order, scale order -- the order of the sequential loop */
  const int k = 25;
  unsigned char *is_kp = (unsigned char *)calloc((size_t)n * (size_t)nd + 1, 1);
#pragma omp parallel num_threads(mo_get_threads())
  {
  int nn_idx[25]; float nn_d2[25];
  float *min_val = (float *)malloc(sizeof(float) * (size_t)nd), *max_val = (float *)malloc(sizeof(float) * (size_t)nd);
#pragma omp for schedule(dynamic, 512)
  for (int i = 0; i < n; ++i) {
    int nr_nn = mo_knn_search(g, cloud[i].x, cloud[i].y, cloud[i].z, k, INFINITY, nn_idx, nn_d2);
    for (int s = 0; s < nd; ++s) {
      min_val[s] = FLT_MAX; max_val[s] = -FLT_MAX;
      for (int j = 0; j < nr_nn; ++j) {
        float d = dog[(size_t)nn_idx[j] * nd + s];
        /* std::min/std::max: NaN in d never replaces the running value */
        min_val[s] = (d < min_val[s]) ? d : min_val[s];
        max_val[s] = (max_val[s] < d) ? d : max_val[s];
      }
    }
    for (int s = 1; s < nd - 1; ++s) {
      float val = dog[(size_t)i * nd + s];
      if (fabs(val) >= min_contrast) {
        /* equality at the point's own scale, STRICT against the adjacent scales (sift_keypoint.hpp, findScaleSpaceExtrema:
         * "(val == min_val[i_scale]) && (val < min_val[i_scale - 1]) && (val < min_val[i_scale + 1])"; rounds 1 - 4 had
         * <= / >= here: the same keypoints unless two DoG values tie exactly -- DESIGN.md section 4, audit table) */
        if ((val == min_val[s]) && (val < min_val[s - 1]) && (val < min_val[s + 1]))
          is_kp[(size_t)i * nd + s] = 1;
        else if ((val == max_val[s]) && (val > max_val[s - 1]) && (val > max_val[s + 1]))
          is_kp[(size_t)i * nd + s] = 1;
      }
    }
  }
  free(min_val); free(max_val);
  }
  for (int i = 0; i < n; ++i)
    for (int s = 1; s < nd - 1; ++s)
      if (is_kp[(size_t)i * nd + s]) kp_push(out, cloud[i].x, cloud[i].y, cloud[i].z, scales[s]);
  free(is_kp); free(dog); free(scales);
  mo_grid_free(g);
}

int mo_keypoints_sift(const mo_point *in, int n, double min_scale, int nr_octaves,
                      int nr_scales_per_octave, double min_contrast, mo_point **out,
                      float **scales_out)
{
  kp_vec kv = {0, 0, 0, 0};
  mo_point *cloud = (mo_point *)malloc(sizeof(mo_point) * (size_t)(n > 0 ? n : 1));
  memcpy(cloud, in, sizeof(mo_point) * (size_t)(n > 0 ? n : 0));
  int cn = n;
  float scale = (float)min_scale;
  for (int oct = 0; oct < nr_octaves; ++oct) {
    const float s = 1.0f * scale;
    mo_point *tmp = (mo_point *)malloc(sizeof(mo_point) * (size_t)(cn > 0 ? cn : 1));
    int tn = mo_downsample(cloud, cn, (double)s, tmp);
    free(cloud); cloud = tmp; cn = tn;
    if (cn < 25) break;
    detect_octave(cloud, cn, scale, nr_scales_per_octave, (float)min_contrast, &kv);
    scale *= 2;
  }
  free(cloud);
  *out = kv.pts;
  if (scales_out) *scales_out = kv.scales; else free(kv.scales);
  return kv.n;
}

/* TEST / EVIDENCE HOOK (scripts/sift_price.py, tests/test_sift_bound.py): the scale space of ONE octave laid open.
 * For octave `octave` of detectKeypoints(SIFT) on `in`: the octave's cloud (*cloud_out, *n_out, malloc'ed), and per point
 *   dog[5]      the float DoG values of computeScaleSpace (the CPU path's bits),
 *   resp_d[6]   the six responses evaluated in DOUBLE (weights exp() of the exact quotient, double sums): the "real value"
 *               the error bound of the certified device path is stated against,
 *   cnt[6]      the number of neighbours inside 3 sigma of every scale,
 *   knn[25]     the 25 nearest neighbours in nearestKSearch's order (-1 padded).
 * Returns 0, or -1 when the octave does not exist (fewer than 25 points). */
int mo_sift_octave_debug(const mo_point *in, int n, double min_scale, int octave, int nr_scales_per_octave,
                         mo_point **cloud_out, int *n_out, float **dog_out, double **resp_out, int **cnt_out, int **knn_out)
{
  mo_point *cloud = (mo_point *)malloc(sizeof(mo_point) * (size_t)(n > 0 ? n : 1));
  memcpy(cloud, in, sizeof(mo_point) * (size_t)(n > 0 ? n : 0));
  int cn = n;
  float scale = (float)min_scale;
  for (int oct = 0; oct <= octave; ++oct) {
    const float s = 1.0f * scale;
    mo_point *tmp = (mo_point *)malloc(sizeof(mo_point) * (size_t)(cn > 0 ? cn : 1));
    int tn = mo_downsample(cloud, cn, (double)s, tmp);
    free(cloud); cloud = tmp; cn = tn;
    if (cn < 25) { free(cloud); return -1; }
    if (oct < octave) scale *= 2;
  }
  const int ns = nr_scales_per_octave + 3, nd = ns - 1;
  float scales[16];
  for (int i = 0; i < ns; ++i) scales[i] = scale * powf(2.0f, (1.0f * (float)i - 1.0f) / (float)nr_scales_per_octave);
  const float max_radius = 3.0f * scales[ns - 1];
  const float r2 = (float)((double)max_radius * (double)max_radius);
  mo_grid *g = mo_grid_build(cloud, cn, max_radius * 0.5f);
  float *dog = (float *)malloc(sizeof(float) * (size_t)cn * (size_t)nd);
  double *resp = (double *)malloc(sizeof(double) * (size_t)cn * (size_t)ns);
  int *cnts = (int *)malloc(sizeof(int) * (size_t)cn * (size_t)ns);
  int *knn = (int *)malloc(sizeof(int) * (size_t)cn * 25);
#pragma omp parallel num_threads(mo_get_threads())
  {
  int cap = 4096;
  int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)cap);
#pragma omp for schedule(dynamic, 512)
  for (int i = 0; i < cn; ++i) {
    int cnt = mo_radius_search(g, cloud[i].x, cloud[i].y, cloud[i].z, r2, idx, d2, cap);
    if (cnt > cap) {
      cap = cnt * 2;
      idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
      d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);
      cnt = mo_radius_search(g, cloud[i].x, cloud[i].y, cloud[i].z, r2, idx, d2, cap);
    }
    float filter_response = 0.0f, previous_filter_response;
    for (int s = 0; s < ns; ++s) {
      float sigma_sqr = powf(scales[s], 2.0f);
      float numerator = 0.0f, denominator = 0.0f;
      double nd_ = 0.0, dd_ = 0.0;
      int m = 0;
      for (int j = 0; j < cnt; ++j) {
        float value = intensity(&cloud[idx[j]]);
        float dist_sqr = d2[j];
        if (dist_sqr <= 9 * sigma_sqr) {
          float w = expf(-0.5f * dist_sqr / sigma_sqr);
          numerator += value * w;
          denominator += w;
          double wd = exp(-0.5 * (double)dist_sqr / (double)sigma_sqr);
          nd_ += (double)value * wd;
          dd_ += wd;
          ++m;
        } else {
          break;
        }
      }
      previous_filter_response = filter_response;
      filter_response = numerator / denominator;
      if (s > 0) dog[(size_t)i * nd + (s - 1)] = filter_response - previous_filter_response;
      resp[(size_t)i * ns + s] = nd_ / dd_;
      cnts[(size_t)i * ns + s] = m;
    }
    int nn_idx[25]; float nn_d2[25];
    int nr_nn = mo_knn_search(g, cloud[i].x, cloud[i].y, cloud[i].z, 25, INFINITY, nn_idx, nn_d2);
    for (int j = 0; j < 25; ++j) knn[(size_t)i * 25 + j] = j < nr_nn ? nn_idx[j] : -1;
  }
  free(idx); free(d2);
  }
  mo_grid_free(g);
  *cloud_out = cloud; *n_out = cn; *dog_out = dog; *resp_out = resp; *cnt_out = cnts; *knn_out = knn;
  return 0;
}
