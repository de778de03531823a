// normals.hip -- computeSurfaceNormals on gfx950 (K3 in SURVEY 2.2).
//
// R/src/features.cpp:168-179: pcl::NormalEstimation<PointXYZRGB, Normal>, radius search,
// viewpoint (0,0,0).  Per point: neighbours with d2 < float(r*r) (self included); fewer than 3
// => NaN; 3x3 covariance -> pcl::eigen33 smallest eigenpair -> flip towards the viewpoint;
// curvature = |lambda0 / trace|.
//
// The covariance is accumulated about the QUERY point (|d| <= r), which is the same matrix as
// PCL's raw-moment form E[xx^T] - E[x]E[x]^T by translation invariance but does not lose the
// eigenvalue to cancellation far from the origin.  Algorithmic traffic: 28 B / point
// (12 B xyz in, 16 B normal out); the neighbourhood walk itself is L1/L2 reuse.
#include "device_util.hpp"

namespace mm3d {

__global__ void __launch_bounds__(256)
k_normals(GridView g, float radius, float r2, float4 *__restrict__ out /* by original index */)
{
  unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  int i = bid * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  const float4 q = g.pts[i];
  int cnt = 0;
  float sx = 0.f, sy = 0.f, sz = 0.f, sxx = 0.f, sxy = 0.f, sxz = 0.f, syy = 0.f, syz = 0.f, szz = 0.f;
  for_each_candidate(g, q.x, q.y, q.z, radius, [&](const float4 &p) {
    float d2 = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
    if (d2 < r2) {
      float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
      ++cnt;
      sx += dx; sy += dy; sz += dz;
      sxx = fmaf(dx, dx, sxx); sxy = fmaf(dx, dy, sxy); sxz = fmaf(dx, dz, sxz);
      syy = fmaf(dy, dy, syy); syz = fmaf(dy, dz, syz); szz = fmaf(dz, dz, szz);
    }
    return true;
  });
  float4 o;
  if (cnt < 3) {
    o.x = o.y = o.z = o.w = __uint_as_float(0x7fc00000u);
  } else {
    const float inv = 1.0f / (float)cnt;
    const float mx = sx * inv, my = sy * inv, mz = sz * inv;
    const float cxx = sxx * inv - mx * mx, cxy = sxy * inv - mx * my, cxz = sxz * inv - mx * mz;
    const float cyy = syy * inv - my * my, cyz = syz * inv - my * mz, czz = szz * inv - mz * mz;
    float ev, v[3];
    eigen33_smallest(cxx, cxy, cxz, cyy, cyz, czz, &ev, v);
    const float eig_sum = cxx + cyy + czz;
    o.w = (eig_sum != 0.0f) ? fabsf(ev / eig_sum) : 0.0f;
    // flipNormalTowardsViewpoint(point, 0, 0, 0)
    const float vx = 0.0f - q.x, vy = 0.0f - q.y, vz = 0.0f - q.z;
    const float cos_theta = vx * v[0] + vy * v[1] + vz * v[2];
    const float s = (cos_theta < 0.0f) ? -1.0f : 1.0f;
    o.x = v[0] * s; o.y = v[1] * s; o.z = v[2] * s;
  }
  out[__float_as_int(q.w)] = o;
}

__global__ void k_fill_nan(float4 *out, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float q = __uint_as_float(0x7fc00000u); out[i] = make_float4(q, q, q, q); }
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
  if (g.n)
    MM3D_LAUNCH(c, "normals_radius", g.n * 28.0, k_normals, dim3(div_up(g.n, 256)), dim3(256), 0, g.view(),
                (float)radius, r2, res->nrm.get());
  return res;
}

}  // namespace mm3d
