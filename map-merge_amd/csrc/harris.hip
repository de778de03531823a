// harris.hip -- detectKeypoints(HARRIS) on gfx950.
//
// R/src/features.cpp:64-83: pcl::HarrisKeypoint3D<PointXYZRGB, PointXYZI>, setNormals(normals),
// non-maximum suppression on, refinement on, threshold = keypoint_threshold, radius = normal_radius
// (R/src/map_merging.cpp:231-233); copyPointCloud keeps xyz only.
// PCL 1.8.1 keypoints/impl/harris_3d.hpp:
//   responseHarris        per point: C = mean outer product of the neighbours' normals (the __SSE__ branch of
//                         calculateNormalCovar: sums divided by float(count)); response = 0.04 + det C - 0.04 tr(C)^2
//   detectKeypoints       keep points with response >= threshold that no neighbour within the radius beats
//   refineCorners         <= 10 steps of  x <- (sum n n^T)^-1 (sum n n^T p)  over the neighbours of x
// Kernels: the response sums the neighbours' normal products in radiusSearch's (distance, index) order like the
// CPU loop does -- sorted neighbour lists in LDS (snb_lds.hpp, the machinery of normals.hip), six chains per point
// -- so the responses, and with them the set of corners, are the CPU restatement's bit for bit; one thread per
// point for the suppression; one WAVE per corner for the refinement, which sorts the neighbours by (distance,
// index) and runs the twelve float sums as sequential chains in that order.
#include "sorted_nb.hpp"
#include "snb_lds.hpp"

namespace mm3d {

constexpr int kHarrisCap = 2048;   // neighbour keys of one corner held in LDS (a power of two)

template <class F>
__device__ __forceinline__ void harris_for_each(const GridView &g, float qx, float qy, float qz, float r, F &&f)
{
  const float ri = r * 1.0001f + 1e-4f;
  if (cell_floor(qx + ri, g.minx, g.inv) < 0 || cell_floor(qx - ri, g.minx, g.inv) > g.dx - 1) return;
  const int x0 = clampi(cell_floor(qx - ri, g.minx, g.inv), 0, g.dx - 1), x1 = clampi(cell_floor(qx + ri, g.minx, g.inv), 0, g.dx - 1);
  int y0 = cell_floor(qy - ri, g.miny, g.inv), y1 = cell_floor(qy + ri, g.miny, g.inv);
  int z0 = cell_floor(qz - ri, g.minz, g.inv), z1 = cell_floor(qz + ri, g.minz, g.inv);
  y0 = y0 < 0 ? 0 : y0; z0 = z0 < 0 ? 0 : z0;
  y1 = y1 > g.dy - 1 ? g.dy - 1 : y1; z1 = z1 > g.dz - 1 ? g.dz - 1 : z1;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      const int row = (z * g.dy + y) * g.dx;
      const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
      for (int j = b; j < e; ++j)
        if (!f(j)) return;
    }
}

// what a point's response is once its six sums and the number of finite normals are known
__device__ __forceinline__ float harris_response_of(float xx, float xy, float xz, float yy, float yz, float zz, unsigned count)
{
  if (count > 0) {
    const float c = (float)count;
    xx /= c; xy /= c; xz /= c; yy /= c; yz /= c; zz /= c;
  }
  float r = 0.0f;
  const float trace = xx + yy + zz;
  if (trace != 0) {
    const float det = xx * yy * zz + 2.0f * xy * xz * yz - xz * xz * yy - xy * xy * zz - yz * yz * xx;
    r = 0.04f + det - 0.04f * trace * trace;
  }
  return r;
}

// responseHarris, by ORIGINAL index (non-finite points keep 0).  Eight lanes per point: lane sub < 6 owns the sum
// sub of {xx, xy, xz, yy, yz, zz} and walks the point's sorted list; a neighbour whose normal is not finite is
// skipped (and not counted), like in the CPU loop.
using HarrisCfg = SnbCfg<8, 1792, 1024, 256, 128, false>;

__global__ void __launch_bounds__(512)
k_harris_response_lds(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g, const float4 *__restrict__ nrm,
                      float radius, float r2, SnbCtl *ctl, int *__restrict__ ov_items, float *__restrict__ resp /* by original index */)
{
  using Cfg = HarrisCfg;
  __shared__ SnbLds<Cfg> S;
  __shared__ float sums[Cfg::kWaves][Cfg::kQ][6];
  __shared__ unsigned cnts[Cfg::kWaves][Cfg::kQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SnbWave<Cfg> &W = S.w[wave];
  snb_run<Cfg>(
      g, S, q_pts, items, n_items, radius, r2, ctl, ov_items, [](const float4 &) { return 0.0f; },
      [&](int fit, const float4 &, const float4 &qw) {
        const int p = lane >> 3, sub = lane & 7;
        if (p < fit) {
          const int base = W.list_off[p], m = W.list_off[p + 1] - base;
          float acc = 0.0f;
          unsigned count = 0;
          for (int e = 0; e < m; ++e) {
            const float4 nv = nrm[S.tw[W.arena[base + e]]];
            if (isfinite(nv.x)) {
              const float u = sub <= 2 ? nv.x : (sub <= 4 ? nv.y : nv.z);
              const float v = sub == 0 ? nv.x : ((sub == 1 || sub == 3) ? nv.y : nv.z);
              acc = __fadd_rn(acc, __fmul_rn(u, v));
              ++count;
            }
          }
          if (sub < 6) sums[wave][p][sub] = acc;
          if (sub == 0) cnts[wave][p] = count;
        }
        wave_lds_fence();
        if (lane < fit) {
          const float *a = sums[wave][lane];
          resp[__float_as_int(qw.w)] = harris_response_of(a[0], a[1], a[2], a[3], a[4], a[5], cnts[wave][lane]);
        }
        wave_lds_fence();
      });
}

// the same for the work items the LDS path could not hold (sorted_nb.hpp: lists in global scratch)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_harris_response_big(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g, const float4 *__restrict__ pts,
                      const float4 *__restrict__ nrm, float radius, float r2, SnScratch sc, float *__restrict__ resp)
{
  __shared__ SnLds lds[4];
  __shared__ float sums[4][kSnG][6];
  __shared__ unsigned cnts[4][kSnG];
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
    int first = (unit & 3) * kSnG;
    int left = min(kSnG, it.y - first);
    while (left > 0) {
      const int p = lane >> 2, sub = lane & 3;
      const float4 q = q_pts[it.x + first + (p < left ? p : 0)];
      const int fit = sn_build_lists<float4>(g, L, q.x, q.y, q.z, left, radius, r2, pts, tmp, fin, sc.error, lane,
                                             [&](float, unsigned idx, const float4 &) { return nrm[idx]; });
      if (p < fit) {
        const int base = L.list_off[p], m = L.list_off[p + 1] - base;
        // lane sub owns sums sub and sub + 4 (sub < 2) of {xx, xy, xz, yy, yz, zz}
        float a0 = 0.0f, a1 = 0.0f;
        unsigned count = 0;
        for (int e = 0; e < m; ++e) {
          const float4 nv = fin[base + e];
          if (isfinite(nv.x)) {
            const float u0 = sub <= 2 ? nv.x : nv.y, v0 = sub == 0 ? nv.x : ((sub == 1 || sub == 3) ? nv.y : nv.z);
            a0 = __fadd_rn(a0, __fmul_rn(u0, v0));
            const float u1 = sub == 0 ? nv.y : nv.z;
            a1 = __fadd_rn(a1, __fmul_rn(u1, nv.z));
            ++count;
          }
        }
        sums[wave][p][sub] = a0;
        if (sub < 2) sums[wave][p][sub + 4] = a1;
        if (sub == 0) cnts[wave][p] = count;
      }
      wave_lds_fence();
      if (lane < fit) {
        const float *a = sums[wave][lane];
        resp[__float_as_int(q_pts[it.x + first + lane].w)] = harris_response_of(a[0], a[1], a[2], a[3], a[4], a[5], cnts[wave][lane]);
      }
      wave_lds_fence();
      first += fit;
      left -= fit;
    }
  }
}

__global__ void k_harris_to_sorted(const float4 *__restrict__ sorted, const float *__restrict__ resp, int n, float *__restrict__ resp_sorted)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) resp_sorted[j] = resp[__float_as_int(sorted[j].w)];
}

// the responses of a cloud by original index (zero where the point is not finite): the LDS path, then -- if it
// counted any -- the launch over the items it could not hold
static void harris_response_launch(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, const Grid &g, float sr, float r2,
                                   float *resp /* n floats, zeroed */)
{
  cloud_hilbert(c, points);
  const int n_items = points->n_wave_items;
  SnbLaunch<HarrisCfg> sl(c, n_items, sizeof(float) * 64 * 7 + 256);
  SnbCtl *ctl = sl.ctl_dev();
  MM3D_LAUNCH(c, "harris_response", g.n * 32.0, k_harris_response_lds, dim3(sl.blocks), dim3(64 * HarrisCfg::kWaves), 0,
              (const float4 *)points->hil_pts.get(), (const int2 *)points->wave_items.get(), n_items, g.view(), (const float4 *)normals->nrm.get(),
              sr, r2, ctl, sl.ov_items.get(), resp);
  // (the fallback launch only when the LDS path counted items it could not hold: normals.hip has the reason)
  int *ho = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(ho, &ctl->ov_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  if (ho[0] == 0) return;
  SnLaunch<float4> sn(c, ho[0] * 4, points->n, 4, 1024u);
  SnScratch sc{sn.tmp.get(), sn.fin.get(), ctl->fb_ctr, &ctl->error, sl.ov_items.get(), &ctl->ov_count};
  MM3D_LAUNCH(c, "harris_response_big", 0.0, k_harris_response_big, dim3(sn.blocks), dim3(256), 0, (const float4 *)points->hil_pts.get(),
              (const int2 *)points->wave_items.get(), n_items, g.view(), (const float4 *)points->pts.get(), (const float4 *)normals->nrm.get(), sr, r2,
              sc, resp);
  int *h = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(h, &ctl->error, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->check_later(h, MM3D_EUNSUPPORTED, "detectKeypoints(HARRIS): a point has more than 16384 neighbours within the radius");
}

// non-maximum suppression: flag (by ORIGINAL index, so that the compaction emits keypoints in index order)
__global__ void __launch_bounds__(256)
k_harris_nonmax(GridView g, const float *__restrict__ resp_sorted, float radius, float r2, float threshold, int *__restrict__ flags)
{
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int i = bid * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  const float ri = resp_sorted[i];
  if (!isfinite(ri) || ri < threshold) return;
  const float4 q = g.pts[i];
  bool is_max = true;
  harris_for_each(g, q.x, q.y, q.z, radius, [&](int j) {
    const float4 p = g.pts[j];
    if (dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2 && ri < resp_sorted[j]) { is_max = false; return false; }
    return true;
  });
  if (is_max) flags[__float_as_int(q.w)] = 1;
}

__global__ void k_harris_emit(const float4 *__restrict__ pts, const int *__restrict__ flags, const int *__restrict__ pos, int n,
                              float4 *__restrict__ out, int *__restrict__ kept)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) { const float4 p = pts[i]; out[pos[i]] = make_float4(p.x, p.y, p.z, 0.0f); kept[pos[i]] = i; }
}

__global__ void k_harris_restore(const float4 *__restrict__ pts, const int *__restrict__ kept, const int *__restrict__ rows, int n_rows,
                                 float4 *__restrict__ corners)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_rows) { const float4 p = pts[kept[rows[t]]]; corners[rows[t]] = make_float4(p.x, p.y, p.z, 0.0f); }
}

// refineCorners: one wave per corner.  rows / scratch: the corners whose neighbourhood overflowed the LDS keys.
__global__ void __launch_bounds__(64)
k_harris_refine(float4 *__restrict__ corners, int nc, GridView g, const float4 *__restrict__ pts /* original order */,
                const float4 *__restrict__ nrm, float radius, float r2, const int *__restrict__ rows,
                unsigned long long *__restrict__ scratch, int cap, int *__restrict__ overflow /* [0] count, [1..] ids, [nc + 1] max */)
{
  __shared__ unsigned long long s_keys[kHarrisCap];
  __shared__ float4 s_n[64], s_p[64];
  __shared__ int s_m;
  const int lane = threadIdx.x;
  const int k = rows ? rows[blockIdx.x] : (int)blockIdx.x;
  unsigned long long *keys = rows ? scratch + (size_t)blockIdx.x * cap : s_keys;
  float4 c = corners[k];
  // lane -> which sum it owns: 0..8 = N[r][q] (r = lane / 3, q = lane % 3), 9..11 = Np[r]
  const int cr = lane < 9 ? lane / 3 : lane - 9, cq = lane % 3;
  unsigned iterations = 0;
  float diff;
  do {
    if (lane == 0) s_m = 0;
    __syncthreads();
    const float ri = radius * 1.0001f + 1e-4f;
    if (!(cell_floor(c.x + ri, g.minx, g.inv) < 0 || cell_floor(c.x - ri, g.minx, g.inv) > g.dx - 1)) {
      const int x0 = clampi(cell_floor(c.x - ri, g.minx, g.inv), 0, g.dx - 1), x1 = clampi(cell_floor(c.x + ri, g.minx, g.inv), 0, g.dx - 1);
      int y0 = cell_floor(c.y - ri, g.miny, g.inv), y1 = cell_floor(c.y + ri, g.miny, g.inv);
      int z0 = cell_floor(c.z - ri, g.minz, g.inv), z1 = cell_floor(c.z + ri, g.minz, g.inv);
      y0 = y0 < 0 ? 0 : y0; z0 = z0 < 0 ? 0 : z0;
      y1 = y1 > g.dy - 1 ? g.dy - 1 : y1; z1 = z1 > g.dz - 1 ? g.dz - 1 : z1;
      for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
          const int row = (z * g.dy + y) * g.dx;
          const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
          for (int j = b + lane; j < e; j += 64) {
            const float4 p = g.pts[j];
            const float d2 = dist2(c.x, c.y, c.z, p.x, p.y, p.z);
            if (d2 < r2) {
              const int slot = atomicAdd(&s_m, 1);
              if (slot < cap) keys[slot] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(p.w);
            }
          }
        }
    }
    __syncthreads();
    const int m = s_m;
    if (m > cap) {
      if (lane == 0) {
        if (!rows) { const int o = atomicAdd(&overflow[0], 1); overflow[1 + o] = k; }
        atomicMax(&overflow[nc + 1], m);
        if (rows) atomicAdd(&overflow[nc + 2], 1);       // still too large in the scratch pass
      }
      return;                                            // the corner keeps its position of the last full step
    }
    int n2 = 1;
    while (n2 < m) n2 <<= 1;
    for (int i = m + lane; i < n2; i += 64) keys[i] = ~0ull;
    __syncthreads();
    for (int k2 = 2; k2 <= n2; k2 <<= 1)
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int t = lane; t < (n2 >> 1); t += 64) {
          const int i = 2 * t - (t & (j - 1)), l = i + j;
          const bool up = (i & k2) == 0;
          const unsigned long long a = keys[i], b = keys[l];
          if ((a > b) == up) { keys[i] = b; keys[l] = a; }
        }
        __syncthreads();
      }
    // the twelve sums, in neighbour order
    float chain = 0.0f;
    for (int b0 = 0; b0 < m; b0 += 64) {
      const int i = b0 + lane;
      bool ok = false;
      if (i < m) {
        const unsigned oi = (unsigned)(keys[i] & 0xffffffffull);
        const float4 n = nrm[oi];
        ok = isfinite(n.x);
        s_n[lane] = n;
        s_p[lane] = pts[oi];
      }
      const unsigned long long mask = ballot(ok);
      __syncthreads();
      if (lane < 12) {
        const int bn = min(64, m - b0);
        for (int e = 0; e < bn; ++e)
          if ((mask >> e) & 1ull) {
            const float4 n = s_n[e], p = s_p[e];
            const float nr = cr == 0 ? n.x : (cr == 1 ? n.y : n.z);
            if (lane < 9) {
              const float nq = cq == 0 ? n.x : (cq == 1 ? n.y : n.z);
              chain += nr * nq;
            } else {
              const float t0 = nr * n.x, t1 = nr * n.y, t2 = nr * n.z;
              chain += t0 * p.x + t1 * p.y + t2 * p.z;
            }
          }
      }
      __syncthreads();
    }
    // invert3x3SymMatrix (common/eigen.h) on the accumulated matrix, x <- N^-1 Np
    const float a = __shfl(chain, 0, 64), bb = __shfl(chain, 1, 64), cc = __shfl(chain, 2, 64);
    const float d = __shfl(chain, 4, 64), e = __shfl(chain, 5, 64), f = __shfl(chain, 8, 64);
    const float np0 = __shfl(chain, 9, 64), np1 = __shfl(chain, 10, 64), np2 = __shfl(chain, 11, 64);
    const float fd_ee = d * f - e * e;
    const float ce_bf = cc * e - bb * f;
    const float be_cd = bb * e - cc * d;
    const float det = a * fd_ee + bb * ce_bf + cc * be_cd;
    const float4 old = c;
    if (det != 0) {
      const float i00 = fd_ee / det, i01 = ce_bf / det, i02 = be_cd / det;
      const float i11 = (a * f - cc * cc) / det, i12 = (bb * cc - a * e) / det, i22 = (a * d - bb * bb) / det;
      c.x = i00 * np0 + i01 * np1 + i02 * np2;
      c.y = i01 * np0 + i11 * np1 + i12 * np2;
      c.z = i02 * np0 + i12 * np1 + i22 * np2;
    }
    const float ex = c.x - old.x, ey = c.y - old.y, ez = c.z - old.z;
    diff = ex * ex + ey * ey + ez * ez;
    if (lane == 0) corners[k] = c;
  } while ((double)diff > 1e-6 && ++iterations < 10);
}

// the response of every point (original order; 0 where the point is not finite), for tests and tools
void harris_response(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, double radius, DevBuf<float> &out)
{
  MM3D_REQUIRE(normals->n == points->n, "detectKeypoints: normals and points differ in size");
  const int n = (int)points->n;
  out = DevBuf<float>(c, (size_t)(n > 0 ? n : 1));
  if (n == 0) return;
  MM3D_HIP(hipMemsetAsync(out.get(), 0, (size_t)n * sizeof(float), c->stream));
  const double sr = (double)(float)radius;             // setRadius(float(radius))
  const float r2 = (float)(sr * sr);
  const Grid &g = cloud_grid(c, points, (float)(sr * 0.5));
  if (g.n == 0) return;
  harris_response_launch(c, points, normals, g, (float)sr, r2, out.get());
  c->settle();
}

mm3d_cloud *detect_keypoints_harris(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, double threshold, double radius)
{
  MM3D_REQUIRE(normals->n == points->n, "detectKeypoints: normals and points differ in size");
  const int n = (int)points->n;
  const double sr = (double)(float)radius;
  const float r2 = (float)(sr * sr);
  if (n == 0) return cloud_from_device(c, DevBuf<float4>(c, 0), 0);
  const Grid &g = cloud_grid(c, points, (float)(sr * 0.5));
  if (g.n == 0) return cloud_from_device(c, DevBuf<float4>(c, 0), 0);
  DevBuf<float> ro(c, (size_t)n), rs(c, g.n);
  MM3D_HIP(hipMemsetAsync(ro.get(), 0, (size_t)n * sizeof(float), c->stream));
  harris_response_launch(c, points, normals, g, (float)sr, r2, ro.get());
  MM3D_LAUNCH(c, "harris_pack", g.n * 12.0, k_harris_to_sorted, dim3(div_up(g.n, 256)), dim3(256), 0, (const float4 *)g.sorted.get(),
              (const float *)ro.get(), g.n, rs.get());
  DevBuf<int> flags(c, (size_t)n + 1), pos(c, (size_t)n + 1);
  MM3D_HIP(hipMemsetAsync(flags.get(), 0, ((size_t)n + 1) * sizeof(int), c->stream));
  MM3D_LAUNCH(c, "harris_nonmax", g.n * 20.0, k_harris_nonmax, dim3(div_up(g.n, 256)), dim3(256), 0, g.view(), (const float *)rs.get(),
              (float)sr, r2, (float)threshold, flags.get());
  exclusive_scan_int(c, flags.get(), pos.get(), (size_t)n + 1);
  int *h = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(h, pos.get() + n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  const int nc = h[0];
  DevBuf<float4> corners(c, (size_t)nc);
  if (nc == 0) return cloud_from_device(c, std::move(corners), 0);
  DevBuf<int> kept(c, (size_t)nc);
  MM3D_LAUNCH(c, "harris_emit", n * 24.0, k_harris_emit, dim3(div_up(n, 256)), dim3(256), 0, (const float4 *)points->pts.get(),
              (const int *)flags.get(), (const int *)pos.get(), n, corners.get(), kept.get());
  DevBuf<int> overflow(c, (size_t)nc + 3);
  MM3D_HIP(hipMemsetAsync(overflow.get(), 0, ((size_t)nc + 3) * sizeof(int), c->stream));
  MM3D_LAUNCH(c, "harris_refine", nc * 10.0 * 120.0 * 40.0, k_harris_refine, dim3(nc), dim3(64), 0, corners.get(), nc, g.view(),
              (const float4 *)points->pts.get(), (const float4 *)normals->nrm.get(), (float)sr, r2, (const int *)nullptr,
              (unsigned long long *)nullptr, kHarrisCap, overflow.get());
  MM3D_HIP(hipMemcpyAsync(h, overflow.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MM3D_HIP(hipMemcpyAsync(h + 1, overflow.get() + nc + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  if (h[0] > 0) {
    // neighbourhoods beyond the LDS keys: those corners restart from their point with the keys in global
    // scratch (twice the largest count seen, since a corner moves while it is refined)
    const int n_over = h[0];
    int cap = kHarrisCap;
    while (cap < 2 * h[1]) cap <<= 1;
    if ((double)n_over * cap * 8.0 > 8e9) throw Error(MM3D_EUNSUPPORTED, "HARRIS: neighbourhoods too large for the refinement (reduce normal_radius)");
    DevBuf<unsigned long long> scratch(c, (size_t)n_over * cap);
    MM3D_LAUNCH(c, "harris_emit", n_over * 32.0, k_harris_restore, dim3(div_up(n_over, 64)), dim3(64), 0, (const float4 *)points->pts.get(),
                (const int *)kept.get(), (const int *)(overflow.get() + 1), n_over, corners.get());
    MM3D_LAUNCH(c, "harris_refine", n_over * 10.0 * cap * 40.0, k_harris_refine, dim3(n_over), dim3(64), 0, corners.get(), nc, g.view(),
                (const float4 *)points->pts.get(), (const float4 *)normals->nrm.get(), (float)sr, r2, (const int *)(overflow.get() + 1),
                scratch.get(), cap, overflow.get());
    MM3D_HIP(hipMemcpyAsync(h, overflow.get() + nc + 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (h[0] > 0) throw Error(MM3D_EUNSUPPORTED, "HARRIS: neighbourhoods too large for the refinement (reduce normal_radius)");
  }
  c->sync();
  return cloud_from_device(c, std::move(corners), (size_t)nc);
}

}  // namespace mm3d
