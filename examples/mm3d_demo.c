/*
 * mm3d_demo.c -- the C ABI from plain C99: what a cgo / JNI / N-API stub would do.
 *
 *   gcc -std=c99 -D_DEFAULT_SOURCE -O2 -Iinclude examples/mm3d_demo.c -Lmap-merge_amd -lmm3d -Wl,-rpath,$PWD/map-merge_amd -lm -o mm3d_demo
 *   ./mm3d_demo            # needs an MI355X
 *
 * Builds two overlapping synthetic clouds (a bumpy textured floor seen from two poses), runs
 * estimateMapsTransforms (R/src/map_merging.cpp:188-275) through mm3d_estimate_maps_transforms and
 * prints the recovered relative transform next to the ground truth.  Exit code 0 when the call
 * chain succeeded and returned finite numbers (how good the alignment is depends on the scene).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mm3d.h"

typedef struct { float x, y, z; unsigned rgba; } point16;   /* packed 16-byte record (stride 16, rgba at 12) */

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static float frand(unsigned *s) { return (float)lcg(s) / 16777216.0f; }

static float hash2(int i, int j)
{
  unsigned h = (unsigned)i * 374761393u + (unsigned)j * 668265263u;
  h = (h ^ (h >> 13)) * 1274126177u;
  return (float)((h ^ (h >> 16)) & 0xFFFFu) / 65535.0f;
}
/* bilinear value noise on a lattice of pitch `cell` */
static float vnoise(float x, float y, float cell)
{
  const float u = x / cell, v = y / cell;
  const int i = (int)floorf(u), j = (int)floorf(v);
  const float fu = u - (float)i, fv = v - (float)j;
  const float a = hash2(i, j), b = hash2(i + 1, j), c = hash2(i, j + 1), d = hash2(i + 1, j + 1);
  return (a * (1 - fu) + b * fu) * (1 - fv) + (c * (1 - fu) + d * fu) * fv;
}
/* rolling ground with a few kerbs: no symmetry, no period */
static float height(float x, float y)
{
  float h = 1.2f * vnoise(x, y, 3.0f) + 0.3f * vnoise(x + 7.0f, y - 3.0f, 0.9f);
  if (vnoise(x, y, 2.0f) > 0.62f) h += 0.4f;
  return h;
}
static unsigned colour(float x, float y)
{
  /* grey texture with 0.3 m and 1 m features: SIFT's contrast threshold (5.0 on 0..255) must fire */
  int v = (int)(255.0f * (0.6f * vnoise(x, y, 0.3f) + 0.4f * vnoise(x, y, 1.0f)));
  if (v < 0) v = 0;
  if (v > 255) v = 255;
  return 0xFF000000u | ((unsigned)v << 16) | ((unsigned)v << 8) | (unsigned)v;
}

int main(void)
{
  const size_t n = 60000;
  const float yaw = 0.25f, tx = 1.5f, ty = -0.8f;
  const char *method = getenv("MM3D_DEMO_METHOD");   /* cloud 1 = world seen from a second pose */
  point16 *a = (point16 *)malloc(n * sizeof(point16)), *b = (point16 *)malloc(n * sizeof(point16));
  unsigned seed = 12345u;
  for (size_t i = 0; i < n; ++i) {
    float x = 12.0f * frand(&seed), y = 12.0f * frand(&seed);
    a[i].x = x; a[i].y = y; a[i].z = height(x, y); a[i].rgba = colour(x, y);
    float u = 2.0f + 12.0f * frand(&seed), v = 1.0f + 12.0f * frand(&seed);   /* shifted window: ~75 % overlap */
    float wz = height(u, v);
    /* world -> pose 1 frame: p1 = R(-yaw) (p - t) */
    float dx = u - tx, dy = v - ty;
    b[i].x = cosf(yaw) * dx + sinf(yaw) * dy; b[i].y = -sinf(yaw) * dx + cosf(yaw) * dy; b[i].z = wz; b[i].rgba = colour(u, v);
  }
  mm3d_ctx *ctx = NULL;
  /* MM3D_DEMO_DEVICES=<n>: the same call on the first n GPUs of this ONE process (mm3d_create_devices: the job is sharded inside
   * the library, the pair records travel through an RCCL all-gather); otherwise one GPU */
  const char *nd = getenv("MM3D_DEMO_DEVICES");
  if (nd && atoi(nd) > 0) {
    int devs[64], k = atoi(nd) > 64 ? 64 : atoi(nd);
    for (int i = 0; i < k; ++i) devs[i] = i;
    if (mm3d_create_devices(devs, k, &ctx) != MM3D_OK) { fprintf(stderr, "mm3d_create_devices(%d GPUs) failed\n", k); return 2; }
    printf("device list of %d, pair records through %s\n", mm3d_device_count(ctx), mm3d_devices_use_rccl(ctx) ? "ncclAllGather" : "host memory");
  } else if (mm3d_create(0, &ctx) != MM3D_OK) { fprintf(stderr, "mm3d_create failed (no GPU?)\n"); return 2; }
  mm3d_params p;
  mm3d_params_default(&p);
  p.descriptor_type = MM3D_DESC_FPFH;
  /* reciprocal matching + RANSAC (the reference's default method); MM3D_DEMO_METHOD=SAC_IA for the other one,
   * which with its 500 random triples out of 10-NN picks often misses on a gentle scene like this */
  p.estimation_method = (method && strcmp(method, "SAC_IA") == 0) ? MM3D_EST_SAC_IA : MM3D_EST_MATCHING;
  mm3d_cloud_view views[2] = {{a, n, sizeof(point16), 12}, {b, n, sizeof(point16), 12}};
  float T[2][16];
  size_t n_out = 0, n_pairs = 1;
  mm3d_pair_result pairs[1];
  int rc = mm3d_estimate_maps_transforms(ctx, views, 2, &p, &T[0][0], &n_out, pairs, &n_pairs);
  if (rc != MM3D_OK) { fprintf(stderr, "estimate failed: %s\n", mm3d_last_error(ctx)); return 3; }
  printf("pairs estimated: %zu, confidence %.3f, ICP iterations %d\n", n_pairs, pairs[0].confidence, pairs[0].icp_iterations);
  /* pairs[0].transform maps cloud 0 (source) into cloud 1's frame: p1 = R(-yaw) (p0 - t) */
  const float *M = pairs[0].transform;   /* column-major like Eigen */
  const float ex = M[12] - (-(cosf(yaw) * tx + sinf(yaw) * ty)), ey = M[13] - (sinf(yaw) * tx - cosf(yaw) * ty), ez = M[14];
  const float err = sqrtf(ex * ex + ey * ey + ez * ez);
  printf("recovered yaw %.4f (truth %.4f), translation error %.3f m\n", atan2f(M[4], M[0]), yaw, err);
  mm3d_destroy(ctx);
  free(a); free(b);
  return (err == err && n_pairs == 1) ? 0 : 1;
}
