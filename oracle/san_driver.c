/* san_driver.c -- TEST INFRASTRUCTURE: drives every entry point of the CPU oracle over a small synthetic scene so that
 * an AddressSanitizer + UndefinedBehaviourSanitizer build (make -C oracle san -> oracle/_san/oracle_san) sees every
 * function run: all six descriptors, both keypoint detectors, both estimation methods, ICP (float and double sums),
 * the pose graph, composeMaps, the degenerate inputs of the reference's gtests (R/test/test_map_merging.cpp:9-40) and
 * the OpenMP loops on two threads.  SURVEY.md section 5 (sanitizers on the CPU restatement); run by
 * tests/test_oracle_cpu.py::test_oracle_under_sanitizers.  Exit code 0 = no finding (the sanitizers abort otherwise). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mm3d_oracle.h"

static uint32_t g_lcg = 12345u;
static float urand(void)
{
  g_lcg = g_lcg * 1664525u + 1013904223u;
  return (float)(g_lcg >> 8) / 16777216.0f;
}

/* a textured ground with a step, a wall and a box: enough geometry and colour for keypoints of both kinds */
static int make_scene(mo_point *p, int n, float ox, float oy, float yaw)
{
  const float c = cosf(yaw), s = sinf(yaw);
  for (int i = 0; i < n; ++i) {
    float x = urand() * 14.0f - 7.0f, y = urand() * 14.0f - 7.0f, z;
    const float pick = urand();
    if (pick < 0.6f) z = (x > 1.5f ? 0.4f : 0.0f) + 0.02f * sinf(3.0f * x) * cosf(2.0f * y);
    else if (pick < 0.8f) { z = urand() * 2.5f; y = 3.0f + 0.01f * urand(); }
    else { z = urand() * 1.2f; x = -3.0f + (urand() < 0.5f ? 0.0f : 1.5f); y = -2.0f + urand() * 1.5f; }
    const float wx = x + ox, wy = y + oy;
    const int chk = (((int)floorf(wx * 0.9f) + (int)floorf(wy * 0.9f)) & 1);
    const uint32_t r = (uint32_t)(chk ? 200 + (int)(urand() * 40) : 40 + (int)(urand() * 40));
    const uint32_t g = (uint32_t)(80 + (int)(120.0f * urand())), b = (uint32_t)(z > 0.3f ? 220 : 30);
    p[i].x = c * x - s * y + 20.0f;
    p[i].y = s * x + c * y - 10.0f;
    p[i].z = z + 0.005f * urand();
    p[i].rgba = 0xff000000u | (r << 16) | (g << 8) | b;
  }
  return n;
}

static int finite16(const float *T)
{
  for (int i = 0; i < 16; ++i) if (!isfinite(T[i])) return 0;
  return 1;
}

int main(void)
{
  enum { N = 9000, MAPS = 3 };
  mo_point *clouds[MAPS];
  int sizes[MAPS];
  for (int m = 0; m < MAPS; ++m) {
    clouds[m] = (mo_point *)malloc(sizeof(mo_point) * N);
    sizes[m] = make_scene(clouds[m], N, 0.7f * (float)m, 0.4f * (float)m, 0.15f * (float)m);
  }
  int failures = 0;
  /* every descriptor x a method x a keypoint type through the whole job */
  static const int combos[][3] = {{2, 1, 0}, {0, 0, 0}, {1, 0, 0}, {3, 1, 0}, {4, 1, 0}, {5, 0, 0}, {2, 0, 1}};
  for (unsigned k = 0; k < sizeof(combos) / sizeof(combos[0]); ++k) {
    mo_params p;
    mo_params_default(&p);
    p.descriptor_type = combos[k][0];
    p.estimation_method = combos[k][1];
    p.keypoint_type = combos[k][2];
    p.refine_transform = 1;
    p.max_iterations = 60;
    if (p.keypoint_type == 1) p.keypoint_threshold = 0.001;
    mo_set_threads(k & 1 ? 2 : 1);
    mo_set_exact_yardstick(k == 0);
    mo_srand(1);
    float T[MAPS * 16];
    mo_estimate est[MAPS * (MAPS - 1) / 2];
    int np = 0;
    const int nodes = mo_estimate_maps_transforms((const mo_point *const *)clouds, sizes, MAPS, &p, T, est, &np);
    mo_pair_trace tr[MAPS * (MAPS - 1) / 2];
    const int nt = mo_last_run_traces(tr, MAPS * (MAPS - 1) / 2);
    printf("descriptor %d method %d keypoints %d: %d nodes, %d pairs, %d traces\n", p.descriptor_type, p.estimation_method,
           p.keypoint_type, nodes, np, nt);
    if (nodes < 0 || nt != np) ++failures;
    for (int i = 0; i < nodes && i < MAPS; ++i) if (!finite16(T + i * 16)) ++failures;
    if (k == 0) {
      float Te[3 * 16]; int it[3], co[3];
      if (mo_last_run_exact(Te, it, co, 3) != np) ++failures;
      mo_point *merged = NULL;
      const int nm = mo_compose_maps((const mo_point *const *)clouds, sizes, MAPS, T, nodes < MAPS ? nodes : MAPS, 0.05, &merged);
      printf("composeMaps: %d points\n", nm);
      mo_free(merged);
    }
  }
  mo_set_exact_yardstick(0);
  mo_set_threads(1);
  /* degenerate inputs (R/test/test_map_merging.cpp:9-40): no clouds, one cloud, an empty cloud among the clouds */
  {
    mo_params p;
    mo_params_default(&p);
    float T[MAPS * 16];
    int np = -1;
    if (mo_estimate_maps_transforms(NULL, NULL, 0, &p, T, NULL, &np) != 0) ++failures;
    if (mo_estimate_maps_transforms((const mo_point *const *)clouds, sizes, 1, &p, T, NULL, &np) != 1) ++failures;
    int sz2[2] = {sizes[0], 0};
    const mo_point *two[2] = {clouds[0], clouds[1]};
    p.descriptor_type = 2; p.estimation_method = 1;
    const int nodes = mo_estimate_maps_transforms(two, sz2, 2, &p, T, NULL, &np);
    printf("one empty cloud: %d nodes, %d pairs\n", nodes, np);
    mo_point *merged = NULL;
    if (mo_compose_maps(NULL, NULL, 0, NULL, 0, 0.05, &merged) != -1) ++failures;
    if (mo_compose_maps(two, sz2, 2, T, 1, 0.05, &merged) != -2) ++failures;
  }
  /* the searches at their edges: k larger than the cloud, a radius that holds everything, a far query */
  {
    mo_grid *g = mo_grid_build(clouds[0], 50, 0.3f);
    int idx[64]; float d2[64];
    const int a = mo_knn_search(g, 20.0f, -10.0f, 0.0f, 64, INFINITY, idx, d2);
    const int b = mo_radius_search(g, 20.0f, -10.0f, 0.0f, 1.0e6f, idx, d2, 64);
    const int c = mo_knn_search(g, 1.0e6f, 1.0e6f, 1.0e6f, 1, 1.0f, idx, d2);
    printf("edge searches: knn %d of 50, radius %d of 50, far %d\n", a, b, c);
    if (a != 50 || b != 50 || c != 0) ++failures;
    mo_grid_free(g);
  }
  for (int m = 0; m < MAPS; ++m) free(clouds[m]);
  printf(failures ? "FAILED: %d checks\n" : "sanitizer driver ok (%d failed checks)\n", failures);
  return failures ? 1 : 0;
}
