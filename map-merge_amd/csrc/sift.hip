// sift.hip -- detectKeypoints(SIFT) on gfx950 (K4 in SURVEY 2.2).
//
// R/src/features.cpp:45-62,85-96: pcl::SIFTKeypoint<PointXYZRGB, PointWithScale>,
// setScales(resolution, 3 octaves, 3 scales per octave), setMinimumContrast(threshold); the result
// is copied with pcl::copyPointCloud, i.e. only x,y,z survive (rgb = 0).
// Per octave: VoxelGrid(leaf = scale) of the previous octave's cloud -> DoG scale space over a
// radius search of 3*sigma_max -> extrema over the 25 nearest neighbours and 3 adjacent scales.
#include <cfloat>

#include "device_util.hpp"

namespace mm3d {

constexpr int kScales = 6;      // nr_scales_per_octave (3) + 3
constexpr int kDog = 5;
constexpr int kKnn = 25;

struct SiftScales {
  float sigma_sqr[kScales];
  float thr9[kScales];          // 9 * sigma_sqr
};

__device__ __forceinline__ float intensity_of(float w)
{
  const unsigned c = __float_as_uint(w);
  const int r = (int)((c >> 16) & 255u), g = (int)((c >> 8) & 255u), b = (int)(c & 255u);
  return (float)(299 * r + 587 * g + 114 * b) / 1000.0f;
}

// intensity per sorted grid entry (so the DoG walk reads 4 B instead of decoding rgba each time)
__global__ void k_sift_intensity(const float4 *__restrict__ sorted, const float4 *__restrict__ pts, int n,
                                 float *__restrict__ val)
{
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  val[j] = intensity_of(pts[__float_as_int(sorted[j].w)].w);
}

// computeScaleSpace: Gaussian-weighted mean intensity at 6 scales -> 5 differences
__global__ void __launch_bounds__(256)
k_sift_dog(GridView g, const float *__restrict__ val, float radius, float r2, SiftScales sc,
           float *__restrict__ dog /* [n][5] by original index */)
{
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int i = bid * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  const float4 q = g.pts[i];
  float num[kScales], den[kScales];
#pragma unroll
  for (int s = 0; s < kScales; ++s) { num[s] = 0.f; den[s] = 0.f; }
  const float ri = radius * 1.0001f + 1e-4f;
  const int x0 = clampi(cell_floor(q.x - ri, g.minx, g.inv), 0, g.dx - 1), x1 = clampi(cell_floor(q.x + ri, g.minx, g.inv), 0, g.dx - 1);
  const int y0 = clampi(cell_floor(q.y - ri, g.miny, g.inv), 0, g.dy - 1), y1 = clampi(cell_floor(q.y + ri, g.miny, g.inv), 0, g.dy - 1);
  const int z0 = clampi(cell_floor(q.z - ri, g.minz, g.inv), 0, g.dz - 1), z1 = clampi(cell_floor(q.z + ri, g.minz, g.inv), 0, g.dz - 1);
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      const int row = (z * g.dy + y) * g.dx;
      const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
      for (int j = b; j < e; ++j) {
        const float4 p = g.pts[j];
        const float d2 = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
        if (d2 < r2) {
          const float v = val[j];
#pragma unroll
          for (int s = 0; s < kScales; ++s) {
            if (d2 <= sc.thr9[s]) {
              const float w = expf(-0.5f * d2 / sc.sigma_sqr[s]);
              num[s] += v * w;
              den[s] += w;
            }
          }
        }
      }
    }
  float prev = num[0] / den[0];
  float *o = dog + (size_t)__float_as_int(q.w) * kDog;
#pragma unroll
  for (int s = 1; s < kScales; ++s) {
    const float cur = num[s] / den[s];
    o[s - 1] = cur - prev;
    prev = cur;
  }
}

// findScaleSpaceExtrema: exact 25-NN (self included) by ring expansion, then min/max of the DoG
// over that set at every scale.  The per-thread sorted candidate list lives in LDS, slot-major so
// that lane l touches bank l.
template <int BD>
__global__ void __launch_bounds__(BD)
k_sift_extrema(GridView g, const float *__restrict__ dog, float min_contrast, int *__restrict__ flags /* [n*3] */)
{
  __shared__ float s_d[kKnn][BD];
  __shared__ int s_i[kKnn][BD];
  const int t = threadIdx.x;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int i = bid * BD + t;
  if (i >= g.n) return;
  const float4 q = g.pts[i];
  const int self = __float_as_int(q.w);
  const int kk = g.n < kKnn ? g.n : kKnn;
  int m = 0;
  const int cx = cell_floor(q.x, g.minx, g.inv), cy = cell_floor(q.y, g.miny, g.inv), cz = cell_floor(q.z, g.minz, g.inv);
  int maxring = max(max(max(cx, g.dx - 1 - cx), max(cy, g.dy - 1 - cy)), max(cz, g.dz - 1 - cz));
  for (int ring = 0; ring <= maxring; ++ring) {
    if (ring >= 2 && m == kk) {
      const float guard = (float)(ring - 1) * g.cell;
      if (s_d[kk - 1][t] <= guard * guard * 0.99999f) break;
    }
    const int z0 = cz - ring, z1 = cz + ring, y0 = cy - ring, y1 = cy + ring, xa = cx - ring, xb = cx + ring;
    for (int z = z0 < 0 ? 0 : z0; z <= (z1 >= g.dz ? g.dz - 1 : z1); ++z) {
      const bool zs = (z == z0 || z == z1);
      for (int y = y0 < 0 ? 0 : y0; y <= (y1 >= g.dy ? g.dy - 1 : y1); ++y) {
        const bool shell = zs || y == y0 || y == y1;
        const int row = (z * g.dy + y) * g.dx;
        const int npass = (shell || ring == 0) ? 1 : 2;
        for (int pass = 0; pass < npass; ++pass) {
          int lo, hi;
          if (npass == 1) { lo = xa; hi = xb; }
          else if (pass == 0) { lo = xa; hi = xa; }
          else { lo = xb; hi = xb; }
          lo = lo < 0 ? 0 : lo;
          hi = hi >= g.dx ? g.dx - 1 : hi;
          if (lo > hi) continue;
          const int b = g.cell_start[row + lo], e = g.cell_start[row + hi + 1];
          for (int j = b; j < e; ++j) {
            const float4 p = g.pts[j];
            const float d = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
            const int oi = __float_as_int(p.w);
            int pos = m;
            if (m == kk) {
              const float wd = s_d[kk - 1][t];
              if (d > wd || (d == wd && oi > s_i[kk - 1][t])) continue;
              pos = kk - 1;
            } else {
              ++m;
            }
            while (pos > 0) {
              const float pd = s_d[pos - 1][t];
              const int pi = s_i[pos - 1][t];
              if (pd > d || (pd == d && pi > oi)) { s_d[pos][t] = pd; s_i[pos][t] = pi; --pos; }
              else break;
            }
            s_d[pos][t] = d; s_i[pos][t] = oi;
          }
        }
      }
    }
  }
  float mn[kDog], mx[kDog];
#pragma unroll
  for (int s = 0; s < kDog; ++s) { mn[s] = FLT_MAX; mx[s] = -FLT_MAX; }
  for (int k = 0; k < m; ++k) {
    const float *d = dog + (size_t)s_i[k][t] * kDog;
#pragma unroll
    for (int s = 0; s < kDog; ++s) {
      const float v = d[s];
      mn[s] = (v < mn[s]) ? v : mn[s];     // std::min(a, b): b < a ? b : a
      mx[s] = (mx[s] < v) ? v : mx[s];     // std::max(a, b): a < b ? b : a
    }
  }
  const float *dv = dog + (size_t)self * kDog;
#pragma unroll
  for (int s = 1; s < kDog - 1; ++s) {
    const float v = dv[s];
    int f = 0;
    if (fabsf(v) >= min_contrast) {
      if ((v == mn[s]) && (v <= mn[s - 1]) && (v <= mn[s + 1])) f = 1;
      else if ((v == mx[s]) && (v >= mx[s - 1]) && (v >= mx[s + 1])) f = 1;
    }
    flags[(size_t)self * 3 + (s - 1)] = f;
  }
}

__global__ void k_sift_emit(const float4 *__restrict__ pts, const int *__restrict__ flags, const int *__restrict__ pos,
                            size_t n3, float4 *__restrict__ out)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n3) return;
  if (flags[e]) {
    const float4 p = pts[e / 3];
    out[pos[e]] = make_float4(p.x, p.y, p.z, 0.0f);   // copyPointCloud(PointWithScale -> PointXYZRGB): rgb = 0
  }
}

mm3d_cloud *detect_keypoints_sift(Context *c, const mm3d_cloud *points, double min_scale, int nr_octaves,
                                  int nr_scales, double min_contrast)
{
  MM3D_REQUIRE(nr_scales == 3, "SIFT: only nr_scales_per_octave == 3 (the reference's setting) is built");
  std::vector<DevBuf<float4>> parts;
  std::vector<size_t> part_n;
  std::unique_ptr<mm3d_cloud> cur;
  const mm3d_cloud *input = points;
  float scale = (float)min_scale;
  for (int oct = 0; oct < nr_octaves; ++oct) {
    std::unique_ptr<mm3d_cloud> next(downsample(c, input, (double)scale));
    cur = std::move(next);
    input = cur.get();
    if (cur->n < 25) break;
    float scales[kScales];
    for (int i = 0; i < kScales; ++i)
      scales[i] = scale * powf(2.0f, (1.0f * (float)i - 1.0f) / (float)nr_scales);
    SiftScales sc;
    for (int i = 0; i < kScales; ++i) { sc.sigma_sqr[i] = powf(scales[i], 2.0f); sc.thr9[i] = 9 * sc.sigma_sqr[i]; }
    const float max_radius = 3.0f * scales[kScales - 1];
    const float r2 = (float)((double)max_radius * (double)max_radius);
    const int n = (int)cur->n;
    // scale space on a grid with cell = r/2
    const Grid &gr = cloud_grid(c, cur.get(), max_radius * 0.5f);
    DevBuf<float> val(c, gr.n);
    DevBuf<float> dog(c, (size_t)n * kDog);
    MM3D_LAUNCH(c, "sift_intensity", gr.n * 24.0, k_sift_intensity, dim3(div_up(gr.n, 256)), dim3(256), 0, gr.sorted.get(),
                cur->pts.get(), gr.n, val.get());
    MM3D_LAUNCH(c, "sift_dog", gr.n * 36.0, k_sift_dog, dim3(div_up(gr.n, 256)), dim3(256), 0, gr.view(), val.get(),
                max_radius, r2, sc, dog.get());
    // extrema on a finer grid (25 neighbours lie within ~3 leaf sizes on a surface)
    const Grid &gk = cloud_grid(c, cur.get(), 3.0f * scale);
    DevBuf<int> flags(c, (size_t)n * 3 + 1);
    MM3D_HIP(hipMemsetAsync(flags.get(), 0, ((size_t)n * 3 + 1) * sizeof(int), c->stream));
    MM3D_LAUNCH(c, "sift_extrema", gk.n * 48.0, (k_sift_extrema<128>), dim3(div_up(gk.n, 128)), dim3(128), 0, gk.view(),
                (const float *)dog.get(), (float)min_contrast, flags.get());
    DevBuf<int> pos(c, (size_t)n * 3 + 1);
    exclusive_scan_int(c, flags.get(), pos.get(), (size_t)n * 3 + 1);
    int *h = (int *)c->pin(64);
    MM3D_HIP(hipMemcpyAsync(h, pos.get() + (size_t)n * 3, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    const size_t nk = (size_t)h[0];
    DevBuf<float4> kp(c, nk);
    if (nk)
      MM3D_LAUNCH(c, "sift_emit", n * 3 * 8.0, k_sift_emit, dim3(div_up((size_t)n * 3, 256)), dim3(256), 0, cur->pts.get(),
                  flags.get(), pos.get(), (size_t)n * 3, kp.get());
    c->sync();
    parts.emplace_back(std::move(kp));
    part_n.push_back(nk);
    scale *= 2;
  }
  size_t total = 0;
  for (size_t v : part_n) total += v;
  DevBuf<float4> all(c, total);
  size_t off = 0;
  for (size_t i = 0; i < parts.size(); ++i) {
    if (part_n[i])
      MM3D_HIP(hipMemcpyAsync(all.get() + off, parts[i].get(), part_n[i] * 16, hipMemcpyDeviceToDevice, c->stream));
    off += part_n[i];
  }
  c->sync();
  return cloud_from_device(c, std::move(all), total);
}

}  // namespace mm3d
