// normals.hip -- computeSurfaceNormals on gfx950 (K3 in SURVEY 2.2).
//
// R/src/features.cpp:168-179: pcl::NormalEstimation<PointXYZRGB, Normal>, radius search,
// viewpoint (0,0,0).  Per point: neighbours with d2 < float(r*r) (self included); fewer than 3
// => NaN; computeMeanAndCovarianceMatrix (float raw moments accumulated over the neighbours IN THE
// ORDER radiusSearch returns them: (distance, index)) -> pcl::eigen33 smallest eigenpair ->
// flipNormalTowardsViewpoint; curvature = |lambda0 / trace|.
//
// The nine raw-moment sums are float chains, so the kernel builds every point's neighbour list in
// that order first (sorted_nb.hpp) and then runs the chains one per lane: 16 points per group, four
// lanes per point, each lane owns two or three of the nine accumulators and walks the point's sorted
// list.  The closed-form eigen solve uses the restated glibc atan2f / cosf / sinf (libm_exact.hpp), so
// the normals are the CPU path's, bit for bit.  Algorithmic traffic: 28 B / point (12 B xyz in, 16 B
// normal out); the lists are L2-resident scratch.
#include "sorted_nb.hpp"
#include "snb_lds.hpp"

namespace mm3d {

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_normals(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g, const float4 *__restrict__ pts,
          float radius, float r2, SnScratch sc, float4 *__restrict__ out /* by original index */)
{
  __shared__ SnLds lds[4];
  __shared__ float sums[4][kSnG][9];
  __shared__ int cnts[4][kSnG];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SnLds &L = lds[wave];
  const size_t slot = (size_t)blockIdx.x * 4 + wave;
  unsigned long long *tmp = sc.tmp + slot * kSnEntries;
  float4 *fin = (float4 *)sc.fin + slot * kSnEntries;
  const int n_units = sn_unit_count(sc, n_items);
  for (;;) {
    const int unit = sn_claim_unit(sc.unit_ctr, n_units, lane);
    if (unit < 0) break;
    const int2 it = items[sn_unit_item(sc, unit)];
    int first = (unit & 3) * kSnG;                       // this quarter's points of the item
    int left = min(kSnG, it.y - first);
    while (left > 0) {
      const int p = lane >> 2, sub = lane & 3;
      const float4 q = q_pts[it.x + first + (p < left ? p : 0)];
      const int fit = sn_build_lists<float4>(g, L, q.x, q.y, q.z, left, radius, r2, pts, tmp, fin, sc.error, lane,
                                             [](float, unsigned, const float4 &pt) { return pt; });
      SN_TICK(t_chain);
      // chains: lane (p, sub) owns accumulators sub, sub + 4, sub + 8 of a = {xx, xy, xz, yy, yz, zz, x, y, z}
      if (p < fit) {
        const int base = L.list_off[p], m = L.list_off[p + 1] - base;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        // eight list entries are requested at a time (the list is L2-resident scratch: one round trip per window)
        for (int e0 = 0; e0 < m; e0 += 8) {
          float4 cw[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) cw[u] = fin[base + min(e0 + u, m - 1)];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (e0 + u < m) {
              const float4 c = cw[u];
              // u * v per accumulator, rounded, then added (the CPU path's a[k] += p.u * p.v)
              const float u0 = sub == 3 ? c.y : c.x;                                  // xx, xy, xz | yy
              const float v0 = sub == 0 ? c.x : (sub == 2 ? c.z : c.y);
              const float u1 = sub == 0 ? c.y : (sub == 1 ? c.z : (sub == 2 ? c.x : c.y));   // yz, zz, x, y
              const float v1 = sub <= 1 ? c.z : 1.0f;
              a0 = __fadd_rn(a0, __fmul_rn(u0, v0));
              a1 = __fadd_rn(a1, __fmul_rn(u1, v1));
              if (sub == 0) a2 = __fadd_rn(a2, c.z);                                  // z
            }
          }
        }
        sums[wave][p][sub] = a0;
        sums[wave][p][sub + 4] = a1;
        if (sub == 0) { sums[wave][p][8] = a2; cnts[wave][p] = m; }
      }
      SN_TOCK(5, t_chain);
      wave_lds_fence();
      // one lane per point: covariance, eigen33, flip (features/normal_3d.h computePointNormal)
      if (lane < fit) {
        const float4 pq = q_pts[it.x + first + lane];
        out[__float_as_int(pq.w)] = normal_from_moments(sums[wave][lane], cnts[wave][lane], pq);
      }
      wave_lds_fence();
      first += fit;
      left -= fit;
    }
  }
}


// ---- the same sums on LDS-resident neighbour lists (snb_lds.hpp) ------------------------------------------
// 8 points per wave, eight lanes per point; lane (p, sub) owns accumulator sub of a = {xx, xy, xz, yy, yz, zz,
// x, y} (and every lane sums z; sub 0's copy is read) and walks the point's list: 16-bit slots into the block's
// tile, every read an LDS read.  Bit for bit k_normals' chains.
using NormalsCfg = SnbCfg<8, 1792, 1024, 256, 128, false>;      // lists of ~70 (up to 256), 2 blocks of 8 waves per CU
// Dense clouds (round 5; BASELINE configs[3], 8 x 2 M indoor points at resolution 0.05: the normals' ball of 0.6 m holds 500 -
// 2 000 neighbours and an item's box thousands of candidates): the small configuration left 45 % of the items to the lists in
// global memory, after working them itself -- 47 ms of normals per map.  One block per CU around the largest tile that leaves
// room for 4 096-entry arenas (a single list must fit its wave's arena); longer lists than the hit buffer go in distance bands.
using NormalsCfgLarge = SnbCfg<8, 3584, 4096, 512, 128, false>;

template <class Cfg>
__global__ void __launch_bounds__(512)
k_normals_lds(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g, float radius, float r2, SnbCtl *ctl,
              int *__restrict__ ov_items, float4 *__restrict__ out /* by original index */)
{
  __shared__ SnbLds<Cfg> S;
  __shared__ float sums[Cfg::kWaves][Cfg::kQ][9];
  __shared__ int cnts[Cfg::kWaves][Cfg::kQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SnbWave<Cfg> &W = S.w[wave];
  snb_run<Cfg>(
      g, S, q_pts, items, n_items, radius, r2, ctl, ov_items, [](const float4 &) { return 0.0f; },
      [&](int fit, const float4 &, const float4 &pq) {
        const int p = lane >> 3, sub = lane & 7;
        if (p < fit) {
          const int base = W.list_off[p], m = W.list_off[p + 1] - base;
          // the factors of this lane's product, picked once: a += u v; v = 1 for the plain sums x and y
          const float *up = sub <= 2 ? S.tx : (sub <= 4 ? S.ty : (sub == 5 ? S.tz : (sub == 6 ? S.tx : S.ty)));
          const float *vp = sub == 0 ? S.tx : ((sub == 1 || sub == 3) ? S.ty : S.tz);
          const bool v1 = sub >= 6;
          float a0 = 0.f, a2 = 0.f;
          int e0 = 0;
          for (; e0 + 4 <= m; e0 += 4) {
            unsigned sl[4];
            float fu[4], fv[4], fz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) sl[u] = W.arena[base + e0 + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) { fu[u] = up[sl[u]]; fv[u] = vp[sl[u]]; fz[u] = S.tz[sl[u]]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              // u * v, rounded, then added (the CPU path's a[k] += p.u * p.v; x * 1 is x)
              a0 = __fadd_rn(a0, __fmul_rn(fu[u], v1 ? 1.0f : fv[u]));
              a2 = __fadd_rn(a2, fz[u]);
            }
          }
          for (; e0 < m; ++e0) {
            const unsigned sl = W.arena[base + e0];
            a0 = __fadd_rn(a0, __fmul_rn(up[sl], v1 ? 1.0f : vp[sl]));
            a2 = __fadd_rn(a2, S.tz[sl]);
          }
          sums[wave][p][sub] = a0;
          if (sub == 0) { sums[wave][p][8] = a2; cnts[wave][p] = m; }
        }
        wave_lds_fence();
        // one lane per point: covariance, eigen33, flip (features/normal_3d.h computePointNormal)
        if (lane < fit) out[__float_as_int(pq.w)] = normal_from_moments(sums[wave][lane], cnts[wave][lane], pq);
        wave_lds_fence();
      });
}

#ifdef MM3D_SN_STATS
extern "C" void mm3d_debug_sn_stats(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sn_stats), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sn_stats), z, sizeof(z)); }
}
#endif

__global__ void k_fill_nan(float4 *out, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float q = __uint_as_float(0x7fc00000u); out[i] = make_float4(q, q, q, q); }
}

// How many candidates would the boxes of the cloud's work items hold?  A sample of up to 256 items, one thread each: the item's
// bounding box grown by the radius, the points of its cells counted from the cell table.  out[0] = sum of the counts, out[1] =
// sampled items, out[2] = items whose box holds more than `cap` candidates.  (~30 us; decides which LDS configuration
// compute_normals launches first.)
__global__ void __launch_bounds__(256)
k_item_box_probe(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g, float radius, int cap, int *__restrict__ out)
{
  const int n_sample = min(n_items, 256);
  const int t = threadIdx.x;
  if (t >= n_sample) return;
  const int2 it = items[(int)((long long)t * n_items / n_sample)];
  float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
  for (int k = 0; k < it.y; ++k) {
    const float4 p = q_pts[it.x + k];
    lx = fminf(lx, p.x); hx = fmaxf(hx, p.x);
    ly = fminf(ly, p.y); hy = fmaxf(hy, p.y);
    lz = fminf(lz, p.z); hz = fmaxf(hz, p.z);
  }
  const float ri = radius * 1.0001f + 1e-4f;
  const int x0 = max(cell_floor(lx - ri, g.minx, g.inv), 0), x1 = min(cell_floor(hx + ri, g.minx, g.inv), g.dx - 1);
  const int y0 = max(cell_floor(ly - ri, g.miny, g.inv), 0), y1 = min(cell_floor(hy + ri, g.miny, g.inv), g.dy - 1);
  const int z0 = max(cell_floor(lz - ri, g.minz, g.inv), 0), z1 = min(cell_floor(hz + ri, g.minz, g.inv), g.dz - 1);
  int cnt = 0;
  if (x0 <= x1)
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const int row = (z * g.dy + y) * g.dx;
        cnt += g.cell_start[row + x1 + 1] - g.cell_start[row + x0];
      }
  atomicAdd(&out[0], cnt);
  atomicAdd(&out[1], 1);
  if (cnt > cap) atomicAdd(&out[2], 1);
}

// The normals of the work items on a device list (ov_items[0 .. *ov_count_dev), n_overflow = that count as the host read
// it) with the lists in global memory (k_normals): what compute_normals runs for the items its LDS launch could not hold,
// and what the fused scale-space + normals launch of sift.hip leaves behind.  Any grid of the cloud serves.
void normals_of_items(Context *c, const mm3d_cloud *in, const Grid &g, double radius, const int *ov_items, const int *ov_count_dev, int n_overflow,
                      float4 *out)
{
  const float r2 = (float)(radius * radius);
  // (as many blocks as there are overflow items, up to 1024: a cloud that is dense everywhere is all overflow)
  SnLaunch<float4> sn(c, n_overflow * 4, in->n, 4, 1024u);
  SnScratch sc{sn.tmp.get(), sn.fin.get(), sn.ctr.get(), sn.error(), ov_items, ov_count_dev};
  MM3D_LAUNCH(c, "normals_radius_big", 0.0, k_normals, dim3(sn.blocks), dim3(256), 0, (const float4 *)in->hil_pts.get(),
              (const int2 *)in->wave_items.get(), in->n_wave_items, g.view(), (const float4 *)in->pts.get(), (float)radius, r2, sc, out);
  int *h = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(h, sn.error(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  // (the scratch goes back to this context's pool; whoever gets it next is enqueued behind the kernel)
  c->check_later(h, MM3D_EUNSUPPORTED, "computeSurfaceNormals: a point has more than 16384 neighbours within the radius");
}

mm3d_normals *compute_normals(Context *c, const mm3d_cloud *in, double radius)
{
  auto *res = new mm3d_normals();
  res->n = in->n;
  res->nrm = DevBuf<float4>(c, in->n);
  if (in->n == 0) return res;
  const Grid &g = cloud_grid(c, in, (float)(radius * 0.5));
  if ((size_t)g.n != in->n)   // non-finite inputs have no normal
    MM3D_LAUNCH(c, "fill_nan", 0, k_fill_nan, dim3(div_up(in->n, 256)), dim3(256), 0, res->nrm.get(), in->n);
  const float r2 = (float)(radius * radius);   // KdTreeFLANN::radiusSearch: float(radius*radius)
  if (g.n) {
    cloud_hilbert(c, in);
    const int n_items = in->n_wave_items;
    // the LDS path; the items it could not hold (dense spots: normally none) are counted, and only if the count is
    // not zero the launch with the lists in global memory follows.  The look at the count costs a host wait here; an
    // empty launch of that kernel cost more: it queues for LDS behind the other streams' kernels (0.4 ms of stream time
    // on the 16-stream bench) and holds its hardware queue meanwhile.
    // Which configuration: a probe counts the candidates in the boxes of a sample of the items.  Where a third of them
    // would not fit the small tile even in two parts -- or the mean box is nearly a tile -- the cloud is dense and the large
    // configuration (one block per CU, 4 096-entry arenas) works it at once; the small one would work every such item and
    // then leave it to the lists in global memory.
    DevBuf<int> probe(c, 4);
    MM3D_HIP(hipMemsetAsync(probe.get(), 0, 4 * sizeof(int), c->stream));
    MM3D_LAUNCH(c, "normals_probe", 0.0, k_item_box_probe, dim3(1), dim3(256), 0, (const float4 *)in->hil_pts.get(), (const int2 *)in->wave_items.get(), n_items,
                g.view(), (float)radius, 2 * NormalsCfg::kTileCap, probe.get());
    int *hp = (int *)c->pin(64);
    MM3D_HIP(hipMemcpyAsync(hp, probe.get(), 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    static const int force_cfg = [] { const char *e = getenv("MM3D_NORMALS_CFG"); return e ? atoi(e) : 0; }();   // A/B knob: 1 small, 2 large
    const double mean_box = hp[1] > 0 ? (double)hp[0] / hp[1] : 0.0;
    const bool dense = force_cfg ? force_cfg == 2 : (hp[1] > 0 && (3 * hp[2] >= hp[1] || mean_box > 0.8 * NormalsCfg::kTileCap));
    int *ho = (int *)c->pin(64);
    auto run = [&](auto cfg_tag) {
      using Cfg = typename decltype(cfg_tag)::type;
      SnbLaunch<Cfg> sl(c, n_items, sizeof(float) * 64 * 10 + 256);
      SnbCtl *ctl = sl.ctl_dev();
      MM3D_LAUNCH(c, "normals_radius", g.n * 28.0, k_normals_lds<Cfg>, dim3(sl.blocks), dim3(64 * Cfg::kWaves), 0, (const float4 *)in->hil_pts.get(),
                  (const int2 *)in->wave_items.get(), n_items, g.view(), (float)radius, r2, ctl, sl.ov_items.get(), res->nrm.get());
      MM3D_HIP(hipMemcpyAsync(ho, &ctl->ov_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      c->sync();
      if (getenv("MM3D_SNB_DEBUG"))
        fprintf(stderr, "normals: n=%d items=%d blocks=%u mean box %.0f, %d of %d sampled boxes over two small tiles -> %s configuration, overflow items=%d\n", g.n, n_items,
                sl.blocks, mean_box, hp[2], hp[1], dense ? "large" : "small", ho[0]);
      if (ho[0] > 0) normals_of_items(c, in, g, radius, sl.ov_items.get(), &ctl->ov_count, ho[0], res->nrm.get());
    };
    struct SmallTag { using type = NormalsCfg; };
    struct LargeTag { using type = NormalsCfgLarge; };
    if (dense) run(LargeTag{});
    else run(SmallTag{});
  }
  return res;
}

#ifdef MM3D_SNB_STATS
extern "C" void mm3d_debug_snb_stats_normals(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_snb_stats), sizeof(unsigned long long) * 32);
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_snb_stats), z, sizeof(z)); }
}
#endif

}  // namespace mm3d
