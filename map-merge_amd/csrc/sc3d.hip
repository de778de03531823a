// sc3d.hip -- computeLocalDescriptors(SC3D) on gfx950.
//
// R/src/dispatch_descriptors.h:47: pcl::ShapeContext3DEstimation<PointXYZRGB, Normal, ShapeContext1980>
// (field "shape_context"), configured by R/src/features.cpp:105-109.  PCL 1.8.1 features/impl/3dsc.hpp:
//   initCompute   log-spaced radii (min_radius 0.1 .. search radius, 15 shells), 11 elevation and 12 azimuth
//                 divisions, and the 1 / cbrt(bin volume) table
//   computePoint  frame = normal of the nearest surface point + a RANDOM tangent direction (three draws of a
//                 boost::mt19937 seeded with 12345 per keypoint that has a neighbour, made orthogonal to the
//                 normal); every neighbour votes (1 / local point density) / cbrt(volume) into its (azimuth,
//                 elevation, radius) bin; local density = surface points within 0.2 m of the neighbour
// The random draws are generated on the host in keypoint order and handed to the wave of each keypoint
// through a prefix sum over "has a neighbour"; one WAVE per keypoint sorts its neighbours by (distance,
// index) like shot.hip, so every bin receives its float votes in the CPU restatement's order (lane =
// bin % 64 owns the bin).  Neighbourhoods beyond the LDS key capacity rerun with the keys in global scratch.
// Algorithmic bytes: 36 B per gathered neighbour (point, density, nearest normal) + 7920 B per row.
#include <cmath>

#include "device_util.hpp"

namespace mm3d {

constexpr int kScAz = 12, kScEl = 11, kScRad = 15;
constexpr int kScDim = kScAz * kScEl * kScRad;        // 1980
constexpr int kScCap = 1024;                          // neighbour keys held in LDS (a power of two)
constexpr int kScTab = (kScRad + 1) + (kScEl + 1) + (kScAz + 1);   // radii, theta and phi divisions; then the volume table

// local point density of every surface point: neighbours within point_density_radius (itself included)
__global__ void __launch_bounds__(256) k_sc3d_density(GridView g, float radius, float r2, int *__restrict__ density /* original order */)
{
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int i = bid * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  const float4 q = g.pts[i];
  int cnt = 0;
  for_each_candidate(g, q.x, q.y, q.z, radius, [&](const float4 &p) {
    cnt += dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2 ? 1 : 0;
    return true;
  });
  density[__float_as_int(q.w)] = cnt;
}

// 1 where the keypoint is finite and has a neighbour within the radius (it then consumes three random draws)
__global__ void __launch_bounds__(256) k_sc3d_has_neighbour(const float4 *__restrict__ kp, int nk, GridView g, float radius, float r2,
                                                           int *__restrict__ flag /* [nk + 1] */)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nk) return;
  int f = 0;
  if (k < nk) {
    const float4 q = kp[k];
    if (isfinite(q.x) && isfinite(q.y) && isfinite(q.z))
      for_each_candidate(g, q.x, q.y, q.z, radius, [&](const float4 &p) {
        if (dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2) { f = 1; return false; }
        return true;
      });
  }
  flag[k] = f;
}

__global__ void __launch_bounds__(64)
k_sc3d(const float4 *__restrict__ kp, int nk, GridView g, const float4 *__restrict__ pts /* original order */, const float4 *__restrict__ nrm,
       const int *__restrict__ density, const float *__restrict__ tab /* kScTab divisions, then kScDim volumes */,
       const float *__restrict__ rnd /* 3 per keypoint with a neighbour */, const int *__restrict__ rnd_pos, float radius_f, float r2,
       const int *__restrict__ rows, unsigned long long *__restrict__ scratch, int cap, float *__restrict__ desc /* [nk][1980] */,
       int *__restrict__ valid, int *__restrict__ overflow /* [0] count, [1..] keypoint ids, [nk + 1] max count */)
{
  __shared__ unsigned long long s_keys[kScCap];
  __shared__ float s_hist[kScDim];
  __shared__ int s_bin[64];
  __shared__ float s_w[64];
  __shared__ int s_m;
  const int lane = threadIdx.x;
  const int k = rows ? rows[blockIdx.x] : (int)blockIdx.x;
  unsigned long long *keys = rows ? scratch + (size_t)blockIdx.x * cap : s_keys;
  float *out = desc + (size_t)k * kScDim;
  const float qnan = __uint_as_float(0x7fc00000u);
  auto give_up = [&]() {
    for (int b = lane; b < kScDim; b += 64) out[b] = qnan;
    if (lane == 0) valid[k] = 0;
  };
  if (lane == 0) s_m = 0;
  __syncthreads();
  const float4 q = kp[k];
  if (!isfinite(q.x) || !isfinite(q.y) || !isfinite(q.z)) { give_up(); return; }
  const float ri = radius_f * 1.0001f + 1e-4f;
  if (!(cell_floor(q.x + ri, g.minx, g.inv) < 0 || cell_floor(q.x - ri, g.minx, g.inv) > g.dx - 1)) {
    const int x0 = clampi(cell_floor(q.x - ri, g.minx, g.inv), 0, g.dx - 1), x1 = clampi(cell_floor(q.x + ri, g.minx, g.inv), 0, g.dx - 1);
    int y0 = cell_floor(q.y - ri, g.miny, g.inv), y1 = cell_floor(q.y + ri, g.miny, g.inv);
    int z0 = cell_floor(q.z - ri, g.minz, g.inv), z1 = cell_floor(q.z + ri, g.minz, g.inv);
    y0 = y0 < 0 ? 0 : y0; z0 = z0 < 0 ? 0 : z0;
    y1 = y1 > g.dy - 1 ? g.dy - 1 : y1; z1 = z1 > g.dz - 1 ? g.dz - 1 : z1;
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const int row = (z * g.dy + y) * g.dx;
        const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
        for (int j = b + lane; j < e; j += 64) {
          const float4 p = g.pts[j];
          const float d2 = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
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
    if (lane == 0 && !rows) {
      const int o = atomicAdd(&overflow[0], 1);
      overflow[1 + o] = k;
      atomicMax(&overflow[nk + 1], m);
    }
    return;
  }
  if (m == 0) { give_up(); return; }
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
  for (int b = lane; b < kScDim; b += 64) s_hist[b] = 0.0f;

  // frame: the nearest neighbour's normal, and a random direction made orthogonal to it
  const float4 nm = nrm[(unsigned)(keys[0] & 0xffffffffull)];
  const float *rv = rnd + 3 * (size_t)rnd_pos[k];
  float xa0 = rv[0], xa1 = rv[1], xa2 = rv[2];
  const float kFltMin = 1.17549435e-38f;
  if (!(fabsf(nm.z - 0.0f) < kFltMin)) xa2 = -(nm.x * xa0 + nm.y * xa1) / nm.z;
  else if (!(fabsf(nm.y - 0.0f) < kFltMin)) xa1 = -(nm.x * xa0 + nm.z * xa2) / nm.y;
  else if (!(fabsf(nm.x - 0.0f) < kFltMin)) xa0 = -(nm.y * xa1 + nm.z * xa2) / nm.x;
  {
    const float nn = sqrtf(xa0 * xa0 + xa1 * xa1 + xa2 * xa2);
    xa0 /= nn; xa1 /= nn; xa2 /= nn;
  }
  const float *radii = tab, *theta_div = tab + (kScRad + 1), *phi_div = tab + (kScRad + 1) + (kScEl + 1), *lut = tab + kScTab;
  __syncthreads();
  for (int b0 = 0; b0 < m; b0 += 64) {
    const int i = b0 + lane;
    int bin = -1;
    float w = 0.0f;
    if (i < m) {
      const unsigned long long key = keys[i];
      const float d2 = __uint_as_float((unsigned)(key >> 32));
      if (!(fabsf(d2 - 0.0f) < kFltMin)) {
        const unsigned oi = (unsigned)(key & 0xffffffffull);
        const float4 p = pts[oi];
        const float r = sqrtf(d2);
        const float po0 = p.x - q.x, po1 = p.y - q.y, po2 = p.z - q.z;
        const float lambda = nm.x * po0 + nm.y * po1 + nm.z * po2;
        float pr0 = p.x - lambda * nm.x, pr1 = p.y - lambda * nm.y, pr2 = p.z - lambda * nm.z;
        pr0 -= q.x; pr1 -= q.y; pr2 -= q.z;
        {
          const float nn = sqrtf(pr0 * pr0 + pr1 * pr1 + pr2 * pr2);
          pr0 /= nn; pr1 /= nn; pr2 /= nn;
        }
        const float c0 = xa1 * pr2 - xa2 * pr1, c1 = xa2 * pr0 - xa0 * pr2, c2 = xa0 * pr1 - xa1 * pr0;
        const float cross_norm = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
        float phi = lm::atan2f_glibc(cross_norm, xa0 * pr0 + xa1 * pr1 + xa2 * pr2) * 57.29578f;
        phi = (c0 * nm.x + c1 * nm.y + c2 * nm.z) < 0.f ? (360.0f - phi) : phi;
        float n0 = po0, n1 = po1, n2v = po2;
        {
          const float nn = sqrtf(n0 * n0 + n1 * n1 + n2v * n2v);
          n0 /= nn; n1 /= nn; n2v /= nn;
        }
        float theta = nm.x * n0 + nm.y * n1 + nm.z * n2v;
        theta = acosf(fminf(1.0f, fmaxf(-1.0f, theta))) * 57.29578f;
        int j = 0, kk = 0, l = 0;
        for (int rad = kScRad; rad >= 1; --rad) if (r <= radii[rad]) j = rad - 1;          // = the first division that holds r
        for (int ang = kScEl; ang >= 1; --ang) if (theta <= theta_div[ang]) kk = ang - 1;
        for (int ang = kScAz; ang >= 1; --ang) if (phi <= phi_div[ang]) l = ang - 1;
        const int dens = density[oi];
        if (dens != 0) {
          bin = (l * kScEl * kScRad) + (kk * kScRad) + j;
          w = (1.0f / (float)dens) * lut[bin];
        }
      }
    }
    s_bin[lane] = bin;
    s_w[lane] = w;
    __syncthreads();
    const int bn = min(64, m - b0);
    for (int e = 0; e < bn; ++e) {
      const int b = s_bin[e];
      if (b >= 0 && (b & 63) == lane) s_hist[b] += s_w[e];
    }
    __syncthreads();
  }
  bool fin = true;
  for (int b = lane; b < kScDim; b += 64) {
    const float v = s_hist[b];
    fin = fin && isfinite(v);
    out[b] = v;
  }
  const bool all_fin = __all(fin);
  if (lane == 0) valid[k] = all_fin ? 1 : 0;
}

__global__ void k_sc3d_compact(const float *__restrict__ in, const int *__restrict__ flags, const int *__restrict__ pos, int n, int dim,
                               float *__restrict__ out)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n * dim) return;
  const int r = (int)(e / dim), cidx = (int)(e % dim);
  if (flags[r]) out[(size_t)pos[r] * dim + cidx] = in[e];
}

// ShapeContext3DEstimation::initCompute (the host's libm, like PCL)
static void sc3d_tables(double search_radius, float *tab)
{
  const double min_radius = 0.1;
  float *radii = tab, *theta_div = tab + (kScRad + 1), *phi_div = theta_div + (kScEl + 1), *lut = tab + kScTab;
  const float azimuth_interval = 360.0f / static_cast<float>(kScAz);
  const float elevation_interval = 180.0f / static_cast<float>(kScEl);
  for (int j = 0; j < kScRad + 1; ++j)
    radii[j] = static_cast<float>(exp(log(min_radius) + ((static_cast<float>(j) / static_cast<float>(kScRad)) * log(search_radius / min_radius))));
  for (int k = 0; k < kScEl + 1; ++k) theta_div[k] = static_cast<float>(k) * elevation_interval;
  for (int l = 0; l < kScAz + 1; ++l) phi_div[l] = static_cast<float>(l) * azimuth_interval;
  const float integr_phi = phi_div[1] * 0.017453293f - phi_div[0] * 0.017453293f;
  const float e = 1.0f / 3.0f;
  for (int j = 0; j < kScRad; ++j) {
    const float integr_r = (radii[j + 1] * radii[j + 1] * radii[j + 1] / 3.0f) - (radii[j] * radii[j] * radii[j] / 3.0f);
    for (int k = 0; k < kScEl; ++k) {
      const float integr_theta = cosf(theta_div[k] * 0.017453293f) - cosf(theta_div[k + 1] * 0.017453293f);
      const float V = integr_phi * integr_theta * integr_r;
      for (int l = 0; l < kScAz; ++l) lut[(l * kScEl * kScRad) + k * kScRad + j] = 1.0f / powf(V, e);
    }
  }
}

mm3d_desc *compute_sc3d(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, mm3d_cloud *keypoints, double radius)
{
  MM3D_REQUIRE(normals->n == points->n, "computeLocalDescriptors: normals and points differ in size");
  // initCompute: "search_radius_ must be GREATER than min_radius_" -> compute() leaves an empty output, every keypoint goes
  if (radius < 0.1) throw Error(MM3D_EINVAL, "SC3D: descriptor_radius must not be below min_radius (0.1)");
  auto *res = new mm3d_desc();
  res->dim = kScDim;
  res->type = MM3D_DESC_SC3D;
  const int nk = (int)keypoints->n;
  if (nk == 0) { res->n = 0; res->data = DevBuf<float>(c, 0); return res; }
  auto drop_all = [&]() {
    res->n = 0; res->data = DevBuf<float>(c, 0);
    keypoints->pts = DevBuf<float4>(c, 0); keypoints->n = 0; keypoints->grids.clear(); keypoints->host.clear();
    keypoints->reset_caches();
  };
  const float r2 = (float)(radius * radius);
  const Grid &g = cloud_grid(c, points, (float)(radius * 0.5));
  if (g.n == 0) { drop_all(); return res; }
  const double density_radius = 0.2;
  const Grid &gd = cloud_grid(c, points, (float)(density_radius * 0.5));
  const int n = (int)points->n;
  DevBuf<int> density(c, (size_t)n);
  MM3D_HIP(hipMemsetAsync(density.get(), 0, (size_t)n * sizeof(int), c->stream));
  MM3D_LAUNCH(c, "sc3d_density", gd.n * 16.0, k_sc3d_density, dim3(div_up(gd.n, 256)), dim3(256), 0, gd.view(), (float)density_radius,
              (float)(density_radius * density_radius), density.get());
  // tables and random draws (host), in one upload
  DevBuf<float> tab(c, (size_t)kScTab + kScDim), rnd(c, (size_t)nk * 3);
  {
    float *h = (float *)c->pin(((size_t)kScTab + kScDim) * sizeof(float));
    sc3d_tables(radius, h);
    MM3D_HIP(hipMemcpyAsync(tab.get(), h, ((size_t)kScTab + kScDim) * sizeof(float), hipMemcpyHostToDevice, c->stream));
    std::vector<float> hr((size_t)nk * 3);
    Mt19937 gen(12345u);                                // ShapeContext3DEstimation(random = false)
    for (auto &v : hr) v = static_cast<float>(static_cast<double>(gen.next()) * (1.0 / 4294967296.0));   // boost::uniform_01<mt19937>
    MM3D_HIP(hipMemcpyAsync(rnd.get(), hr.data(), hr.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    c->sync();                                          // hr goes out of scope
  }
  DevBuf<int> hasn(c, (size_t)nk + 1), rpos(c, (size_t)nk + 1);
  MM3D_LAUNCH(c, "sc3d_density", nk * 64.0, k_sc3d_has_neighbour, dim3(div_up((size_t)nk + 1, 256)), dim3(256), 0,
              (const float4 *)keypoints->pts.get(), nk, g.view(), (float)radius, r2, hasn.get());
  exclusive_scan_int(c, hasn.get(), rpos.get(), (size_t)nk + 1);
  DevBuf<float> raw(c, (size_t)nk * kScDim);
  DevBuf<int> valid(c, (size_t)nk + 1), overflow(c, (size_t)nk + 2);
  MM3D_HIP(hipMemsetAsync(valid.get(), 0, ((size_t)nk + 1) * sizeof(int), c->stream));
  MM3D_HIP(hipMemsetAsync(overflow.get(), 0, ((size_t)nk + 2) * sizeof(int), c->stream));
  MM3D_LAUNCH(c, "sc3d", nk * (200.0 * 36.0 + 7920.0), k_sc3d, dim3(nk), dim3(64), 0, (const float4 *)keypoints->pts.get(), nk, g.view(),
              (const float4 *)points->pts.get(), (const float4 *)normals->nrm.get(), (const int *)density.get(), (const float *)tab.get(),
              (const float *)rnd.get(), (const int *)rpos.get(), (float)radius, r2, (const int *)nullptr, (unsigned long long *)nullptr, kScCap,
              raw.get(), valid.get(), overflow.get());
  int *h = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(h, overflow.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MM3D_HIP(hipMemcpyAsync(h + 1, overflow.get() + nk + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  if (h[0] > 0) {
    const int n_over = h[0];
    int cap = kScCap;
    while (cap < h[1]) cap <<= 1;
    if ((double)n_over * cap * 8.0 > 8e9) throw Error(MM3D_EUNSUPPORTED, "SC3D: neighbourhoods too large for the scratch pass (reduce descriptor_radius)");
    DevBuf<unsigned long long> scratch(c, (size_t)n_over * cap);
    MM3D_LAUNCH(c, "sc3d", n_over * (cap * 36.0 + 7920.0), k_sc3d, dim3(n_over), dim3(64), 0, (const float4 *)keypoints->pts.get(), nk, g.view(),
                (const float4 *)points->pts.get(), (const float4 *)normals->nrm.get(), (const int *)density.get(), (const float *)tab.get(),
                (const float *)rnd.get(), (const int *)rpos.get(), (float)radius, r2, (const int *)(overflow.get() + 1), scratch.get(), cap,
                raw.get(), valid.get(), overflow.get());
    c->sync();
  }
  DevBuf<int> vpos(c, (size_t)nk + 1);
  exclusive_scan_int(c, valid.get(), vpos.get(), (size_t)nk + 1);
  MM3D_HIP(hipMemcpyAsync(h, vpos.get() + nk, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  const int nv = h[0];
  res->n = (size_t)nv;
  if (nv == nk) {
    res->data = std::move(raw);
  } else {
    res->data = DevBuf<float>(c, (size_t)nv * kScDim);
    DevBuf<float4> kp2(c, nv);
    if (nv) {
      MM3D_LAUNCH(c, "compact_rows", nk * 15840.0, k_sc3d_compact, dim3(div_up((size_t)nk * kScDim, 256)), dim3(256), 0,
                  (const float *)raw.get(), (const int *)valid.get(), (const int *)vpos.get(), nk, kScDim, res->data.get());
      MM3D_LAUNCH(c, "compact_rows", nk * 32.0, k_sc3d_compact, dim3(div_up((size_t)nk * 4, 256)), dim3(256), 0,
                  (const float *)keypoints->pts.get(), (const int *)valid.get(), (const int *)vpos.get(), nk, 4, (float *)kp2.get());
    }
    c->sync();
    keypoints->pts = std::move(kp2);
    keypoints->n = (size_t)nv;
    keypoints->grids.clear();
    keypoints->host.clear();
    keypoints->reset_caches();
  }
  c->sync();
  return res;
}

}  // namespace mm3d
