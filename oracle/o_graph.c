/*
 * o_graph.c -- pose graph restated (TEST INFRASTRUCTURE).
 *
 * R/src/graph.cpp:7-15    numberOfNodesInEstimates
 * R/src/graph.cpp:17-57   DisjointSets (union by rank, path compression)
 * R/src/graph.cpp:64-102  largestConnectedComponent  (note: the second loop re-checks only
 *                         membership of source_idx, not confidence -> sub-threshold edges whose
 *                         source lies in the largest component leak back in)
 * R/src/graph.cpp:104-175 findMaxSpanningTree (Kruskal on std::sort(greater<GraphEdge>), leaves,
 *                         BFS eccentricities from every leaf, centres = min of max distance)
 * R/src/map_merging.cpp:137-186 getTransform / computeGlobalTransforms
 *
 * std::sort is not stable; the order of equal-weight edges is unspecified in the reference. The
 * oracle uses a stable order (original pair order among equals).
 */
#include "mm3d_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct { size_t *parent, *size, *rank; size_t n; } dsets;
static void ds_init(dsets *d, size_t n)
{
  d->n = n;
  d->parent = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));
  d->size = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));
  d->rank = (size_t *)calloc(n ? n : 1, sizeof(size_t));
  for (size_t i = 0; i < n; ++i) { d->parent[i] = i; d->size[i] = 1; }
}
static void ds_free(dsets *d) { free(d->parent); free(d->size); free(d->rank); }
static size_t ds_find(dsets *d, size_t elem)
{
  size_t set = elem;
  while (set != d->parent[set]) set = d->parent[set];
  while (elem != d->parent[elem]) { size_t next = d->parent[elem]; d->parent[elem] = set; elem = next; }
  return set;
}
static size_t ds_merge(dsets *d, size_t s1, size_t s2)
{
  if (d->rank[s1] < d->rank[s2]) { d->parent[s1] = s2; d->size[s2] += d->size[s1]; return s2; }
  if (d->rank[s2] < d->rank[s1]) { d->parent[s2] = s1; d->size[s1] += d->size[s2]; return s1; }
  d->parent[s1] = s2; d->rank[s2]++; d->size[s2] += d->size[s1];
  return s2;
}

static size_t num_nodes(const mo_estimate *p, int n)
{
  size_t m = 0;
  for (int i = 0; i < n; ++i) {
    if (p[i].source_idx + 1 > m) m = p[i].source_idx + 1;
    if (p[i].target_idx + 1 > m) m = p[i].target_idx + 1;
  }
  return m;
}

int mo_largest_component(const mo_estimate *pairs, int n_pairs, double thr, int *kept)
{
  size_t nn = num_nodes(pairs, n_pairs);
  for (int i = 0; i < n_pairs; ++i) kept[i] = 0;
  if (nn == 0) return 0;
  dsets c; ds_init(&c, nn);
  for (int i = 0; i < n_pairs; ++i) {
    if (pairs[i].confidence < thr) continue;
    size_t a = ds_find(&c, pairs[i].source_idx), b = ds_find(&c, pairs[i].target_idx);
    if (a != b) ds_merge(&c, a, b);
  }
  size_t max_comp = 0;                    /* first maximum of comps.size (std::max_element) */
  for (size_t i = 1; i < nn; ++i) if (c.size[i] > c.size[max_comp]) max_comp = i;
  int cnt = 0;
  for (int i = 0; i < n_pairs; ++i)
    if (ds_find(&c, pairs[i].source_idx) == max_comp) { kept[i] = 1; ++cnt; }
  ds_free(&c);
  return cnt;
}

typedef struct { size_t from, to; double w; int ord; } gedge;
static int edge_cmp_desc(const void *a, const void *b)
{
  const gedge *x = (const gedge *)a, *y = (const gedge *)b;
  if (x->w > y->w) return -1;
  if (x->w < y->w) return 1;
  return (x->ord > y->ord) - (x->ord < y->ord);
}

/* adjacency lists in insertion order (std::list push_back) */
typedef struct { size_t *to; int *start, *cnt; } adj;

static void bfs_dist(const size_t *adj_to, const int *adj_start, const int *adj_cnt, size_t nn,
                     size_t from, size_t *dist)
{
  unsigned char *was = (unsigned char *)calloc(nn, 1);
  size_t *queue = (size_t *)malloc(sizeof(size_t) * nn);
  size_t qh = 0, qt = 0;
  was[from] = 1; queue[qt++] = from;
  while (qh < qt) {
    size_t v = queue[qh++];
    for (int e = 0; e < adj_cnt[v]; ++e) {
      size_t to = adj_to[adj_start[v] + e];
      if (!was[to]) { dist[to] = dist[v] + 1; was[to] = 1; queue[qt++] = to; }
    }
  }
  free(was); free(queue);
}

/* builds the spanning tree adjacency (tree edges added from->to and to->from in Kruskal order);
 * returns centres count */
static int span_tree(const mo_estimate *pairs, int n, size_t nn, size_t **adj_to_out,
                     int **adj_start_out, int **adj_cnt_out, size_t centers[2])
{
  gedge *edges = (gedge *)malloc(sizeof(gedge) * (size_t)(n ? n : 1));
  for (int i = 0; i < n; ++i) {
    edges[i].from = pairs[i].source_idx; edges[i].to = pairs[i].target_idx;
    edges[i].w = pairs[i].confidence; edges[i].ord = i;
  }
  qsort(edges, (size_t)n, sizeof(gedge), edge_cmp_desc);
  dsets c; ds_init(&c, nn);
  /* tree edge list in order, then bucket per vertex preserving order */
  size_t *tf = (size_t *)malloc(sizeof(size_t) * 2 * (nn ? nn : 1)), *tt = (size_t *)malloc(sizeof(size_t) * 2 * (nn ? nn : 1));
  int nt = 0;
  int *powers = (int *)calloc(nn ? nn : 1, sizeof(int));
  for (int i = 0; i < n; ++i) {
    size_t a = ds_find(&c, edges[i].from), b = ds_find(&c, edges[i].to);
    if (a != b) {
      ds_merge(&c, a, b);
      tf[nt] = edges[i].from; tt[nt] = edges[i].to; ++nt;
      tf[nt] = edges[i].to; tt[nt] = edges[i].from; ++nt;
      powers[edges[i].from]++; powers[edges[i].to]++;
    }
  }
  int *start = (int *)calloc(nn + 1, sizeof(int)), *cnt = (int *)calloc(nn ? nn : 1, sizeof(int));
  for (int e = 0; e < nt; ++e) start[tf[e] + 1]++;
  for (size_t v = 0; v < nn; ++v) start[v + 1] += start[v];
  size_t *to = (size_t *)malloc(sizeof(size_t) * (size_t)(nt ? nt : 1));
  for (int e = 0; e < nt; ++e) { to[start[tf[e]] + cnt[tf[e]]] = tt[e]; cnt[tf[e]]++; }

  size_t *max_d = (size_t *)calloc(nn ? nn : 1, sizeof(size_t)), *cur = (size_t *)malloc(sizeof(size_t) * (nn ? nn : 1));
  for (size_t v = 0; v < nn; ++v) {
    if (powers[v] != 1) continue;
    memset(cur, 0, sizeof(size_t) * nn);
    bfs_dist(to, start, cnt, nn, v, cur);
    for (size_t j = 0; j < nn; ++j) if (cur[j] > max_d[j]) max_d[j] = cur[j];
  }
  int nc = 0;
  if (nn > 0) {
    size_t mm = max_d[0];
    for (size_t i = 1; i < nn; ++i) if (mm > max_d[i]) mm = max_d[i];
    for (size_t i = 0; i < nn; ++i) if (max_d[i] == mm) { if (nc < 2) centers[nc] = i; ++nc; }
  }
  free(edges); ds_free(&c); free(tf); free(tt); free(powers); free(max_d); free(cur);
  *adj_to_out = to; *adj_start_out = start; *adj_cnt_out = cnt;
  return nc;
}

int mo_max_spanning_tree_centers(const mo_estimate *pairs, int n_pairs, size_t centers[2])
{
  size_t nn = num_nodes(pairs, n_pairs);
  size_t *to; int *start, *cnt;
  int nc = span_tree(pairs, n_pairs, nn, &to, &start, &cnt, centers);
  free(to); free(start); free(cnt);
  return nc;
}

static void get_transform(const mo_estimate *comp, int n, size_t from, size_t to, float out[16])
{
  for (int i = 0; i < n; ++i) {
    if (comp[i].source_idx == from && comp[i].target_idx == to) { mo_mat4_inverse(comp[i].transform, out); return; }
    if (comp[i].source_idx == to && comp[i].target_idx == from) { memcpy(out, comp[i].transform, sizeof(float) * 16); return; }
  }
  memset(out, 0, sizeof(float) * 16);
}

int mo_global_transforms(const mo_estimate *pairs, int n_pairs, double confidence_threshold,
                         float *out, int out_cap_nodes)
{
  size_t nodes_count = num_nodes(pairs, n_pairs);
  if (nodes_count == 0) return 0;     /* reference: span_tree_centers[0] on an empty vector (UB) */
  if ((int)nodes_count > out_cap_nodes) return -1;
  int *kept = (int *)malloc(sizeof(int) * (size_t)n_pairs);
  int nk = mo_largest_component(pairs, n_pairs, confidence_threshold, kept);
  mo_estimate *comp = (mo_estimate *)malloc(sizeof(mo_estimate) * (size_t)(nk ? nk : 1));
  int m = 0;
  for (int i = 0; i < n_pairs; ++i) if (kept[i]) comp[m++] = pairs[i];
  free(kept);
  memset(out, 0, sizeof(float) * 16 * nodes_count);
  /* the tree is built over numberOfNodesInEstimates(component) nodes */
  size_t nn = num_nodes(comp, nk);
  size_t centers[2] = {0, 0};
  size_t *to; int *start, *cnt;
  int nc = span_tree(comp, nk, nn, &to, &start, &cnt, centers);
  if (nc > 0) {
    size_t ref = centers[0];
    float *G = out;
    memset(&G[ref * 16], 0, sizeof(float) * 16);
    G[ref * 16 + 0] = G[ref * 16 + 5] = G[ref * 16 + 10] = G[ref * 16 + 15] = 1.0f;
    /* walkBreadthFirst from the centre, chaining global[to] = global[from] * getTransform(from,to) */
    unsigned char *was = (unsigned char *)calloc(nn, 1);
    size_t *queue = (size_t *)malloc(sizeof(size_t) * nn);
    size_t qh = 0, qt = 0;
    was[ref] = 1; queue[qt++] = ref;
    while (qh < qt) {
      size_t v = queue[qh++];
      for (int e = 0; e < cnt[v]; ++e) {
        size_t t = to[start[v] + e];
        if (was[t]) continue;
        float Tft[16];
        get_transform(comp, nk, v, t, Tft);
        mo_mat4_mul(&G[v * 16], Tft, &G[t * 16]);
        was[t] = 1; queue[qt++] = t;
      }
    }
    free(was); free(queue);
  }
  free(to); free(start); free(cnt); free(comp);
  return (int)nodes_count;
}
