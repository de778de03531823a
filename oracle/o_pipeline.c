/*
 * o_pipeline.c -- R/src/map_merging.cpp restated (TEST INFRASTRUCTURE).
 *
 * MapMergingParams defaults  R/include/map_merge_3d/map_merging.h:28-44
 * estimateMapsTransforms     R/src/map_merging.cpp:188-275
 * composeMaps                R/src/map_merging.cpp:277-305
 *
 * Every row of the descriptor dispatch table and both keypoint detectors are restated; values outside the
 * enums make mo_estimate_maps_transforms return -3.
 */
#include "mm3d_oracle.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

void mo_params_default(mo_params *p)
{
  p->resolution = 0.1;
  p->descriptor_radius = p->resolution * 8.0;
  p->outliers_min_neighbours = 50;
  p->normal_radius = p->resolution * 6.0;
  p->keypoint_type = 0;       /* SIFT */
  p->keypoint_threshold = 5.0;
  p->descriptor_type = 0;     /* PFH */
  p->estimation_method = 0;   /* MATCHING */
  p->refine_transform = 1;
  p->inlier_threshold = p->resolution * 5.0;
  p->max_correspondence_distance = p->inlier_threshold * 2.0;
  p->max_iterations = 500;
  p->matching_k = 5;
  p->transform_epsilon = 1e-2;
  p->confidence_threshold = 0.0;
  p->output_resolution = 0.05;
}

static mo_pair_trace *g_traces = NULL;
static int g_n_traces = 0;
/* per-pair traces of the most recent mo_estimate_maps_transforms, in pair order; returns how many exist */
int mo_last_run_traces(mo_pair_trace *out, int cap)
{
  for (int i = 0; i < g_n_traces && i < cap; ++i) out[i] = g_traces[i];
  return g_n_traces;
}

static int g_yardstick = 0;
static float *g_exact_T = NULL;
static int *g_exact_it = NULL, *g_exact_corr = NULL;
static int g_n_exact = 0;
void mo_set_exact_yardstick(int on) { g_yardstick = on; }
int mo_last_run_exact(float *T, int *iters, int *corr, int cap)
{
  for (int i = 0; i < g_n_exact && i < cap; ++i) {
    memcpy(T + (size_t)i * 16, g_exact_T + (size_t)i * 16, sizeof(float) * 16);
    iters[i] = g_exact_it[i];
    corr[i] = g_exact_corr[i];
  }
  return g_n_exact;
}

static void identity16(float *T)
{
  memset(T, 0, sizeof(float) * 16);
  T[0] = T[5] = T[10] = T[15] = 1.0f;
}

int mo_estimate_maps_transforms(const mo_point *const *clouds, const int *sizes, int n_clouds,
                                const mo_params *params, float *out_T, mo_estimate *pair_out,
                                int *n_pairs_out)
{
  if (n_pairs_out) *n_pairs_out = 0;
  if (n_clouds == 0) return 0;
  if (n_clouds == 1) { identity16(out_T); return 1; }
  if ((params->descriptor_type != 2 && params->descriptor_type != 0 && params->descriptor_type != 4 &&
       params->descriptor_type != 1 && params->descriptor_type != 3 && params->descriptor_type != 5) ||
      (params->keypoint_type != 0 && params->keypoint_type != 1))
    return -3;
  const int dim = params->descriptor_type == 0 ? 125 : params->descriptor_type == 4 ? 1344 : params->descriptor_type == 1 ? 250 : params->descriptor_type == 3 ? 2 : params->descriptor_type == 5 ? 1980 : 33;

  mo_point **resized = (mo_point **)calloc((size_t)n_clouds, sizeof(mo_point *));
  int *rn = (int *)calloc((size_t)n_clouds, sizeof(int));
  mo_normal **normals = (mo_normal **)calloc((size_t)n_clouds, sizeof(mo_normal *));
  mo_point **kps = (mo_point **)calloc((size_t)n_clouds, sizeof(mo_point *));
  int *kn = (int *)calloc((size_t)n_clouds, sizeof(int));
  float **desc = (float **)calloc((size_t)n_clouds, sizeof(float *));

  for (int i = 0; i < n_clouds; ++i) {
    int n = sizes[i];
    mo_point *a = (mo_point *)malloc(sizeof(mo_point) * (size_t)(n > 0 ? n : 1));
    int na = mo_downsample(clouds[i], n, params->resolution, a);
    mo_point *b = (mo_point *)malloc(sizeof(mo_point) * (size_t)(na > 0 ? na : 1));
    /* NB: the outlier radius is the DESCRIPTOR radius (R/src/map_merging.cpp:219-220) */
    int nb = mo_remove_outliers(a, na, params->descriptor_radius, params->outliers_min_neighbours, b);
    free(a);
    resized[i] = b; rn[i] = nb;
    normals[i] = (mo_normal *)malloc(sizeof(mo_normal) * (size_t)(nb > 0 ? nb : 1));
    mo_normals(b, nb, params->normal_radius, normals[i]);
    /* detectKeypoints(points, normals, type, keypoint_threshold, normal_radius, resolution): map_merging.cpp:231-233 */
    kn[i] = params->keypoint_type == 1
                ? mo_keypoints_harris(b, normals[i], nb, params->keypoint_threshold, params->normal_radius, &kps[i], NULL, NULL)
                : mo_keypoints_sift(b, nb, params->resolution, 3, 3, params->keypoint_threshold, &kps[i], NULL);
    desc[i] = (float *)malloc(sizeof(float) * (size_t)dim * (size_t)(kn[i] > 0 ? kn[i] : 1));
    kn[i] = dim == 125    ? mo_descriptors_pfh(b, normals[i], nb, kps[i], kn[i], params->descriptor_radius, desc[i])
            : dim == 1980 ? mo_descriptors_sc3d(b, normals[i], nb, kps[i], kn[i], params->descriptor_radius, desc[i])
            : dim == 2    ? mo_descriptors_rsd(b, normals[i], nb, kps[i], kn[i], params->descriptor_radius, desc[i])
            : dim == 250  ? mo_descriptors_pfhrgb(b, normals[i], nb, kps[i], kn[i], params->descriptor_radius, desc[i])
            : dim == 1344 ? mo_descriptors_shot(b, normals[i], nb, kps[i], kn[i], params->descriptor_radius, desc[i])
                          : mo_descriptors_fpfh(b, normals[i], nb, kps[i], kn[i], params->descriptor_radius, desc[i]);
  }

  int max_pairs = n_clouds * (n_clouds - 1) / 2;
  mo_estimate *pairs = (mo_estimate *)malloc(sizeof(mo_estimate) * (size_t)(max_pairs > 0 ? max_pairs : 1));
  int np = 0;
  for (int i = 0; i < n_clouds - 1; ++i)
    for (int j = i + 1; j < n_clouds; ++j)
      if (kn[i] > 0 && kn[j] > 0) {
        pairs[np].source_idx = (size_t)i; pairs[np].target_idx = (size_t)j; pairs[np].confidence = 0.0;
        memset(pairs[np].transform, 0, sizeof(float) * 16);
        ++np;
      }
  free(g_traces);
  g_traces = (mo_pair_trace *)calloc((size_t)(np > 0 ? np : 1), sizeof(mo_pair_trace));
  g_n_traces = np;
  free(g_exact_T); free(g_exact_it); free(g_exact_corr);
  g_exact_T = NULL; g_exact_it = NULL; g_exact_corr = NULL; g_n_exact = 0;
  if (g_yardstick && params->refine_transform) {
    g_exact_T = (float *)calloc((size_t)(np > 0 ? np : 1) * 16, sizeof(float));
    g_exact_it = (int *)calloc((size_t)(np > 0 ? np : 1), sizeof(int));
    g_exact_corr = (int *)calloc((size_t)(np > 0 ? np : 1), sizeof(int));
    g_n_exact = np;
  }
  for (int p = 0; p < np; ++p) {
    int i = (int)pairs[p].source_idx, j = (int)pairs[p].target_idx;
    mo_estimate_transform(resized[i], rn[i], kps[i], desc[i], kn[i], resized[j], rn[j], kps[j], desc[j],
                          kn[j], dim, params->estimation_method, params->refine_transform,
                          params->inlier_threshold, params->max_correspondence_distance,
                          params->max_iterations, (size_t)params->matching_k,
                          params->transform_epsilon, pairs[p].transform);
    mo_last_pair_trace(&g_traces[p]);
    if (g_n_exact) {                               /* test yardstick only: never feeds the results */
      float T0[16];
      mo_last_pair_init(T0);
      mo_icp_double_sums(resized[i], rn[i], resized[j], rn[j], T0, params->max_correspondence_distance,
                         params->max_iterations, params->transform_epsilon, g_exact_T + (size_t)p * 16, &g_exact_it[p]);
      g_exact_corr[p] = mo_last_double_sums_correspondences();
    }
    pairs[p].confidence = 1.0 / mo_transform_score(resized[i], rn[i], resized[j], rn[j], pairs[p].transform,
                                                   params->max_correspondence_distance);
  }
  if (pair_out) memcpy(pair_out, pairs, sizeof(mo_estimate) * (size_t)np);
  if (n_pairs_out) *n_pairs_out = np;
  int nodes = mo_global_transforms(pairs, np, params->confidence_threshold, out_T, n_clouds);

  for (int i = 0; i < n_clouds; ++i) { free(resized[i]); free(normals[i]); free(kps[i]); free(desc[i]); }
  free(resized); free(rn); free(normals); free(kps); free(kn); free(desc); free(pairs);
  return nodes;
}

static int is_zero16(const float *T)
{
  for (int i = 0; i < 16; ++i) if (!(T[i] == 0.0f)) return 0;   /* Eigen isZero(): |x| <= prec*0 */
  return 1;
}

int mo_compose_maps(const mo_point *const *clouds, const int *sizes, int n_clouds,
                    const float *transforms, int n_transforms, double resolution, mo_point **out)
{
  *out = NULL;
  if (n_clouds == 0) return -1;
  if (n_clouds != n_transforms) return -2;
  size_t total = 0;
  for (int i = 0; i < n_clouds; ++i) total += (size_t)sizes[i];
  mo_point *cat = (mo_point *)malloc(sizeof(mo_point) * (total ? total : 1));
  size_t m = 0;
  for (int i = 0; i < n_clouds; ++i) {
    const float *T = &transforms[(size_t)i * 16];
    if (is_zero16(T)) continue;
    for (int k = 0; k < sizes[i]; ++k) {
      const mo_point *p = &clouds[i][k];
      mo_point q = *p;
      q.x = T[0] * p->x + T[4] * p->y + T[8] * p->z + T[12];
      q.y = T[1] * p->x + T[5] * p->y + T[9] * p->z + T[13];
      q.z = T[2] * p->x + T[6] * p->y + T[10] * p->z + T[14];
      cat[m++] = q;
    }
  }
  mo_point *res = (mo_point *)malloc(sizeof(mo_point) * (m ? m : 1));
  int n = mo_downsample(cat, (int)m, resolution, res);
  free(cat);
  *out = res;
  return n;
}
