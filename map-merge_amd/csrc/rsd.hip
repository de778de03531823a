// rsd.hip -- computeLocalDescriptors(RSD) on gfx950.
//
// R/src/dispatch_descriptors.h:43: pcl::RSDEstimation<PointXYZRGB, Normal, PrincipalRadiiRSD> (field
// "r_min"; matching reads the two floats r_min, r_max).  PCL 1.8.1 features/impl/rsd.hpp: computeFeature
// -> pcl::computeRSD with nr_subdiv = 5, plane_radius = 0.2, max_dist = the search radius: the first
// neighbour is the reference point; every other neighbour contributes (angle between the two normal
// lines, distance) to 5 distance bins that keep the smallest and the largest angle; the radii are the
// least-squares slopes of the two envelopes.  The reference point is the nearest neighbour, as in the
// CPU restatement (oracle/o_fpfh.c::mo_rsd_raw); everything after that choice is order free (minima
// and maxima), so one wave per keypoint reduces it with shuffles.
// Algorithmic bytes: 32 B per gathered neighbour (point + normal) + 8 B per row.
#include "device_util.hpp"

namespace mm3d {

constexpr int kRsdBins = 5;

__global__ void __launch_bounds__(256)
k_rsd(const float4 *__restrict__ kp, int nk, GridView g, const float4 *__restrict__ nrm /* original order */, float radius_f, double radius,
      float r2, float *__restrict__ desc /* [nk][2] */)
{
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (k >= nk) return;                                     // wave-uniform
  const float4 q = kp[k];
  const bool finite = isfinite(q.x) && isfinite(q.y) && isfinite(q.z);
  const float ri = radius_f * 1.0001f + 1e-4f;
  int x0 = 0, x1 = -1, y0 = 0, y1 = -1, z0 = 0, z1 = -1;
  if (finite && !(cell_floor(q.x + ri, g.minx, g.inv) < 0 || cell_floor(q.x - ri, g.minx, g.inv) > g.dx - 1)) {
    x0 = clampi(cell_floor(q.x - ri, g.minx, g.inv), 0, g.dx - 1); x1 = clampi(cell_floor(q.x + ri, g.minx, g.inv), 0, g.dx - 1);
    y0 = max(cell_floor(q.y - ri, g.miny, g.inv), 0); y1 = min(cell_floor(q.y + ri, g.miny, g.inv), g.dy - 1);
    z0 = max(cell_floor(q.z - ri, g.minz, g.inv), 0); z1 = min(cell_floor(q.z + ri, g.minz, g.inv), g.dz - 1);
  }
  // pass 1: the neighbour count and the nearest neighbour (ties to the lower index) = the reference point
  unsigned long long best = ~0ull;
  int cnt = 0;
  unsigned best_j = 0;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      const int row = (z * g.dy + y) * g.dx;
      const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
      for (int j = b + lane; j < e; j += 64) {
        const float4 p = g.pts[j];
        const float d2 = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
        if (d2 < r2) {
          ++cnt;
          const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(p.w);
          if (key < best) { best = key; best_j = (unsigned)j; }
        }
      }
    }
  cnt = __shfl(wave_sum(cnt), 0, 64);
  unsigned long long wbest = best;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const unsigned long long o = __shfl_xor(wbest, s, 64);
    wbest = o < wbest ? o : wbest;
  }
  float *out = desc + (size_t)k * 2;
  if (cnt < 2) {
    if (lane == 0) { out[0] = 0.0f; out[1] = 0.0f; }
    return;
  }
  const int owner = __ffsll((long long)ballot(best == wbest)) - 1;
  const unsigned ref_j = __shfl(best_j, owner, 64);
  const float4 p0 = g.pts[ref_j];
  const float4 n0 = nrm[(unsigned)(wbest & 0xffffffffull)];
  // pass 2: per distance bin, the smallest and the largest angle
  double lo[kRsdBins], hi[kRsdBins];
#pragma unroll
  for (int d = 0; d < kRsdBins; ++d) { lo[d] = 1.7976931348623157e308; hi[d] = -1.7976931348623157e308; }
  const double max_dist = radius;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      const int row = (z * g.dy + y) * g.dx;
      const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
      for (int j = b + lane; j < e; j += 64) {
        const float4 p = g.pts[j];
        if (!(dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2) || (unsigned)j == ref_j) continue;
        const float4 nv = nrm[__float_as_int(p.w)];
        double cosine = (double)(nv.x * n0.x + nv.y * n0.y + nv.z * n0.z);
        if (cosine > 1) cosine = 1;
        if (cosine < -1) cosine = -1;
        double angle = acos(cosine);
        if (angle > 3.14159265358979323846 / 2) angle = 3.14159265358979323846 - angle;
        const double dist = (double)sqrtf((p.x - p0.x) * (p.x - p0.x) + (p.y - p0.y) * (p.y - p0.y) + (p.z - p0.z) * (p.z - p0.z));
        if (dist > max_dist) continue;
        const double fb = floor(kRsdBins * dist / max_dist);
        if (!(fb >= 0.0 && fb < (double)kRsdBins)) continue;           // NaN, or dist == max_dist after rounding
        const int bin_d = (int)fb;
#pragma unroll
        for (int d = 0; d < kRsdBins; ++d)
          if (d == bin_d) {
            if (lo[d] > angle) lo[d] = angle;
            if (hi[d] < angle) hi[d] = angle;
          }
      }
    }
#pragma unroll
  for (int d = 0; d < kRsdBins; ++d)
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const double ol = __shfl_xor(lo[d], s, 64), oh = __shfl_xor(hi[d], s, 64);
      lo[d] = ol < lo[d] ? ol : lo[d];      // NaN angles never enter (comparisons with NaN are false), as on the CPU
      hi[d] = oh > hi[d] ? oh : hi[d];
    }
  if (lane != 0) return;
  // bin 0 starts at (0, 0) in PCL (min_max_angle_by_dist[0] = {0, 0})
  lo[0] = lo[0] < 0.0 ? lo[0] : 0.0;
  hi[0] = hi[0] > 0.0 ? hi[0] : 0.0;
  double Amint_Amin = 0, Amint_d = 0, Amaxt_Amax = 0, Amaxt_d = 0;
#pragma unroll
  for (int di = 0; di < kRsdBins; ++di)
    if (hi[di] >= 0) {
      const double p_min = lo[di], p_max = hi[di];
      const double f = (di + 0.5) * max_dist / kRsdBins;
      Amint_Amin += p_min * p_min;
      Amint_d += p_min * f;
      Amaxt_Amax += p_max * p_max;
      Amaxt_d += p_max * f;
    }
  const double plane_radius = 0.2;
  float min_radius = Amint_Amin == 0.0 ? (float)plane_radius : (float)fmin(Amint_d / Amint_Amin, plane_radius);
  float max_radius = Amaxt_Amax == 0.0 ? (float)plane_radius : (float)fmin(Amaxt_d / Amaxt_Amax, plane_radius);
  min_radius *= 1.1f;
  max_radius *= 0.9f;
  if (min_radius < max_radius) { out[0] = min_radius; out[1] = max_radius; }
  else { out[1] = min_radius; out[0] = max_radius; }
}

mm3d_desc *compute_rsd(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, mm3d_cloud *keypoints, double radius)
{
  MM3D_REQUIRE(normals->n == points->n, "computeLocalDescriptors: normals and points differ in size");
  auto *res = new mm3d_desc();
  res->dim = 2;
  res->type = MM3D_DESC_RSD;
  const int nk = (int)keypoints->n;
  res->n = (size_t)nk;
  res->data = DevBuf<float>(c, (size_t)nk * 2);
  if (nk == 0) return res;
  // every row is finite by construction (radii are capped), so nothing is pruned (features.cpp:118-143 finds no invalid row)
  const Grid &g = cloud_grid(c, points, (float)(radius * 0.5));
  if (g.n == 0) {
    MM3D_HIP(hipMemsetAsync(res->data.get(), 0, (size_t)nk * 2 * sizeof(float), c->stream));
    c->sync();
    return res;
  }
  MM3D_LAUNCH(c, "rsd", nk * (200.0 * 32.0 + 8.0), k_rsd, dim3(div_up(nk, 4)), dim3(256), 0, (const float4 *)keypoints->pts.get(), nk, g.view(),
              (const float4 *)normals->nrm.get(), (float)radius, radius, (float)(radius * radius), res->data.get());
  c->sync();
  return res;
}

}  // namespace mm3d
