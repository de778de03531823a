/*
 * o_search.c -- exact radius / k-NN search for the CPU oracle (TEST INFRASTRUCTURE).
 *
 * Stands in for pcl::search::KdTree -> pcl::KdTreeFLANN -> FLANN
 * KDTreeSingleIndex (exact: checks=-1, eps=0, sorted results), which the
 * reference uses at R/src/features.cpp:34,171,50,105 and
 * R/src/matching.cpp:204,263.  Only the RESULT contract is restated:
 *   - squared distances accumulated as FLANN's L2_Simple does
 *     (result = 0; result += d*d for x, y, z in that order, float);
 *   - radius search keeps dist < r*r (strict; FLANN RadiusResultSet::addPoint);
 *   - results sorted ascending by distance; ties (unspecified in FLANN) are
 *     broken by ascending point index so the oracle is deterministic.
 * The index itself is a uniform grid; which index finds the neighbours does
 * not change an exact search's result.
 */
#include "mm3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif
static int g_threads = 1;
void mo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int mo_get_threads(void)
{
#ifdef _OPENMP
  return g_threads;
#else
  return 1;
#endif
}

struct mo_grid {
  const mo_point *pts;
  int n;
  double minx, miny, minz, cell, inv;
  int dx, dy, dz;
  int *cell_start; /* dx*dy*dz + 1 */
  int *order;      /* point indices sorted by cell, ascending index inside a cell */
};

static inline float dist2(const mo_point *p, float qx, float qy, float qz)
{
  float r = 0.0f, d;
  d = qx - p->x; r += d * d;
  d = qy - p->y; r += d * d;
  d = qz - p->z; r += d * d;
  return r;
}

static inline int cell_coord(double v, double mn, double inv, int dim)
{
  int c = (int)floor((v - mn) * inv);
  if (c < 0) c = 0;
  if (c >= dim) c = dim - 1;
  return c;
}

mo_grid *mo_grid_build(const mo_point *pts, int n, float cell)
{
  mo_grid *g = (mo_grid *)calloc(1, sizeof(mo_grid));
  g->pts = pts; g->n = n; g->cell = cell;
  double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    double v[3] = {pts[i].x, pts[i].y, pts[i].z};
    for (int a = 0; a < 3; ++a) {
      if (i == 0 || v[a] < mn[a]) mn[a] = v[a];
      if (i == 0 || v[a] > mx[a]) mx[a] = v[a];
    }
  }
  /* keep the table bounded: grow the cell when the box is huge */
  for (;;) {
    g->inv = 1.0 / g->cell;
    double ex = floor((mx[0] - mn[0]) * g->inv) + 1, ey = floor((mx[1] - mn[1]) * g->inv) + 1,
           ez = floor((mx[2] - mn[2]) * g->inv) + 1;
    if (ex * ey * ez <= 64e6) { g->dx = (int)ex; g->dy = (int)ey; g->dz = (int)ez; break; }
    g->cell *= 1.5;
  }
  if (n == 0) { g->dx = g->dy = g->dz = 1; }
  g->minx = mn[0]; g->miny = mn[1]; g->minz = mn[2];
  size_t nc = (size_t)g->dx * g->dy * g->dz;
  g->cell_start = (int *)calloc(nc + 1, sizeof(int));
  g->order = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int *cid = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    int cx = cell_coord(pts[i].x, g->minx, g->inv, g->dx);
    int cy = cell_coord(pts[i].y, g->miny, g->inv, g->dy);
    int cz = cell_coord(pts[i].z, g->minz, g->inv, g->dz);
    cid[i] = (cz * g->dy + cy) * g->dx + cx;
    g->cell_start[cid[i] + 1]++;
  }
  for (size_t c = 0; c < nc; ++c) g->cell_start[c + 1] += g->cell_start[c];
  int *fill = (int *)malloc(sizeof(int) * nc);
  memcpy(fill, g->cell_start, sizeof(int) * nc);
  for (int i = 0; i < n; ++i) g->order[fill[cid[i]]++] = i;
  free(fill); free(cid);
  return g;
}

void mo_grid_free(mo_grid *g)
{
  if (!g) return;
  free(g->cell_start); free(g->order); free(g);
}

typedef struct { float d2; int idx; } cand;
static int cand_cmp(const void *a, const void *b)
{
  const cand *x = (const cand *)a, *y = (const cand *)b;
  if (x->d2 < y->d2) return -1;
  if (x->d2 > y->d2) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* one scratch list per thread (baseline B2 searches from several OpenMP threads) */
static _Thread_local cand *g_scratch = NULL;
static _Thread_local int g_scratch_cap = 0;
static cand *scratch(int need)
{
  if (need > g_scratch_cap) {
    g_scratch_cap = need * 2 + 1024;
    g_scratch = (cand *)realloc(g_scratch, sizeof(cand) * (size_t)g_scratch_cap);
  }
  return g_scratch;
}

int mo_radius_search(const mo_grid *g, float qx, float qy, float qz, float r2,
                     int *idx, float *d2, int cap)
{
  if (g->n == 0) return 0;
  double r = sqrt((double)r2) * (1.0 + 1e-6) + 1e-9;
  int x0 = (int)floor((qx - r - g->minx) * g->inv), x1 = (int)floor((qx + r - g->minx) * g->inv);
  int y0 = (int)floor((qy - r - g->miny) * g->inv), y1 = (int)floor((qy + r - g->miny) * g->inv);
  int z0 = (int)floor((qz - r - g->minz) * g->inv), z1 = (int)floor((qz + r - g->minz) * g->inv);
  if (x0 < 0) x0 = 0;
  if (y0 < 0) y0 = 0;
  if (z0 < 0) z0 = 0;
  if (x1 >= g->dx) x1 = g->dx - 1;
  if (y1 >= g->dy) y1 = g->dy - 1;
  if (z1 >= g->dz) z1 = g->dz - 1;
  int cnt = 0;
  cand *sc = scratch(1024);
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      if (x0 > x1) continue;
      size_t row = ((size_t)z * g->dy + y) * g->dx;
      int b = g->cell_start[row + x0], e = g->cell_start[row + x1 + 1];
      for (int j = b; j < e; ++j) {
        int i = g->order[j];
        float d = dist2(&g->pts[i], qx, qy, qz);
        if (d < r2) {
          if (cnt >= g_scratch_cap) sc = scratch(cnt + 1);
          sc[cnt].d2 = d; sc[cnt].idx = i; ++cnt;
        }
      }
    }
  qsort(sc, (size_t)cnt, sizeof(cand), cand_cmp);
  int m = cnt < cap ? cnt : cap;
  for (int j = 0; j < m; ++j) { idx[j] = sc[j].idx; d2[j] = sc[j].d2; }
  return cnt;
}

/* insert into a sorted (d2,idx) list of at most k */
static inline void topk_insert(cand *top, int *m, int k, float d, int i)
{
  int pos = *m;
  if (pos == k) {
    const cand *w = &top[k - 1];
    if (d > w->d2 || (d == w->d2 && i > w->idx)) return;
    pos = k - 1;
  } else {
    (*m)++;
  }
  while (pos > 0 && (top[pos - 1].d2 > d || (top[pos - 1].d2 == d && top[pos - 1].idx > i))) {
    top[pos] = top[pos - 1];
    --pos;
  }
  top[pos].d2 = d; top[pos].idx = i;
}

int mo_knn_search(const mo_grid *g, float qx, float qy, float qz, int k,
                  float max_d2, int *idx, float *d2)
{
  if (g->n == 0 || k <= 0) return 0;
  if (k > g->n) k = g->n;
  cand *top = (cand *)alloca(sizeof(cand) * (size_t)k);
  int m = 0;
  /* query cell (unclamped: the query may lie outside the box) */
  int cx = (int)floor((qx - g->minx) * g->inv), cy = (int)floor((qy - g->miny) * g->inv),
      cz = (int)floor((qz - g->minz) * g->inv);
  int maxring;
  {
    int a = cx > g->dx - 1 - cx ? cx : g->dx - 1 - cx;
    int b = cy > g->dy - 1 - cy ? cy : g->dy - 1 - cy;
    int c = cz > g->dz - 1 - cz ? cz : g->dz - 1 - cz;
    maxring = a > b ? a : b; if (c > maxring) maxring = c;
    if (maxring < 0) maxring = 0;
  }
  double rmax = isinf(max_d2) ? INFINITY : sqrt((double)max_d2) * (1.0 + 1e-6) + 1e-9;
  for (int ring = 0; ring <= maxring; ++ring) {
    /* everything outside rings 0..ring-1 is at least (ring-1)*cell away (conservative by one
     * cell against rounding of the cell assignment) */
    if (ring >= 2) {
      double guard = (double)(ring - 1) * g->cell;
      if (guard > rmax) break;
      if (m == k) {
        double w = sqrt((double)top[k - 1].d2);
        if (w < guard * (1.0 - 1e-6)) break;
      }
    }
    int z0 = cz - ring, z1 = cz + ring, y0 = cy - ring, y1 = cy + ring, x0 = cx - ring, x1 = cx + ring;
    for (int z = z0; z <= z1; ++z) {
      if (z < 0 || z >= g->dz) continue;
      for (int y = y0; y <= y1; ++y) {
        if (y < 0 || y >= g->dy) continue;
        int shell = (z == z0 || z == z1 || y == y0 || y == y1);
        for (int pass = 0; pass < (shell ? 1 : 2); ++pass) {
          int xa, xb;
          if (shell) { xa = x0; xb = x1; }
          else if (pass == 0) { xa = x0; xb = x0; }
          else { xa = x1; xb = x1; if (ring == 0) continue; }
          if (xa < 0) xa = 0;
          if (xb >= g->dx) xb = g->dx - 1;
          if (xa > xb) continue;
          size_t row = ((size_t)z * g->dy + y) * g->dx;
          int b = g->cell_start[row + xa], e = g->cell_start[row + xb + 1];
          for (int j = b; j < e; ++j) {
            int i = g->order[j];
            float d = dist2(&g->pts[i], qx, qy, qz);
            if (d <= max_d2) topk_insert(top, &m, k, d, i);
          }
        }
      }
    }
  }
  for (int j = 0; j < m; ++j) { idx[j] = top[j].idx; d2[j] = top[j].d2; }
  return m;
}

void mo_free(void *p) { free(p); }
