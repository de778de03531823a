// filters.hip -- downSample / removeOutliers / composeMaps' transform+concatenate on gfx950.
//
// downSample      R/src/features.cpp:17-27  (pcl::VoxelGrid<PointXYZRGB>, cubic leaf)
// removeOutliers  R/src/features.cpp:31-43  (pcl::RadiusOutlierRemoval, dense k-NN form:
//                 keep p iff its (min_pts+1)-th nearest neighbour, self included, has d2 <= r*r)
// composeMaps     R/src/map_merging.cpp:292-299 (transformPointCloud + operator+=)
//
// Both filters are HBM/L2-bound integer+float work (voxel 16 B read per raw point +
// 16 B written per voxel; outlier filter 13 B per point); neither is reshaped into a GEMM.
#include <algorithm>
#include <climits>

#include "device_util.hpp"
#include "scan_fused.hpp"

namespace mm3d {

// ---------------------------------------------------------------- voxel grid
// key = i + j*div_x + k*div_x*div_y exactly as VoxelGrid computes it (float floor, int32)
__global__ void k_voxel_keys(const float4 *__restrict__ pts, int n, float inv, int minbx, int minby, int minbz,
                             int mul1, int mul2, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  vals[i] = (uint32_t)i;
  if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { keys[i] = 0xFFFFFFFFu; return; }
  int ijk0 = (int)(floorf(__fmul_rn(p.x, inv)) - (float)minbx);
  int ijk1 = (int)(floorf(__fmul_rn(p.y, inv)) - (float)minby);
  int ijk2 = (int)(floorf(__fmul_rn(p.z, inv)) - (float)minbz);
  keys[i] = (uint32_t)(ijk0 + ijk1 * mul1 + ijk2 * mul2);
}

// a voxel starts where the sorted key changes (non-finite points carry the key 0xFFFFFFFF and sort last); the fused scan
// numbers the voxels and writes every voxel's first position (scan_fused.hpp: one launch for heads + scan + starts)
struct VoxelHeadLoad {
  const uint32_t *keys; int n;
  __device__ __forceinline__ int operator()(size_t j) const
  {
    if (j >= (size_t)n) return 0;
    const uint32_t k = keys[j];
    return (k != 0xFFFFFFFFu && (j == 0 || keys[j - 1] != k)) ? 1 : 0;
  }
};
struct VoxelStartStore {
  int n; int *starts; int *n_voxels;
  __device__ __forceinline__ void operator()(size_t j, int prefix, int v) const
  {
    if (j == (size_t)n) { n_voxels[0] = prefix; n_voxels[1] = 0; return; }      // ([1]: k_voxel_centroid's "a centroid strayed" word)
    if (v) starts[prefix] = (int)j;
  }
  __device__ __forceinline__ void done() const {}
};

// One thread per voxel walks its (stable-sorted, i.e. ascending input index) members and sums in
// float in that order: bit-identical to CentroidPoint on the CPU restatement.
__global__ void __launch_bounds__(256)
k_voxel_centroid(const float4 *__restrict__ pts, const uint32_t *__restrict__ order, const int *__restrict__ starts,
                 int *__restrict__ nvox_dev /* [1]: set when a centroid lies farther than two leaves from its voxel's first member */,
                 const int *__restrict__ unsorted, int nvalid, float two_leaves, float4 *__restrict__ out, unsigned *__restrict__ box)
{
  // (the grid is sized by the bound -- one thread per finite input point --: the number of voxels is still on the device
  // when this is launched; the centroids' bounding box is reduced on the way out, scan_fused.hpp::BoxAcc)
  // `unsorted`: the counting sort's "a bin was too long, nothing was placed" word -- `order` is not valid then, and the host,
  // which sees that word at the same wait as the count, sorts by radix and launches this again (unsorted = nullptr)
  const int nvox = (unsorted && *unsorted) ? 0 : *nvox_dev;
  BoxAcc acc;
  // (grid-stride over the voxels with a capped grid, and every block leaves its box in its own slot: blocks that finish
  // together and meet at seven shared words pay ~50 ns per atomic, one after the other -- 1 750 blocks did, 8 x the launch)
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += gridDim.x * blockDim.x) {
    int b = starts[v], e = (v + 1 < nvox) ? starts[v + 1] : nvalid;
    float sx = 0.f, sy = 0.f, sz = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, sa = 0.f;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    for (int j = b; j < e; ++j) {
      float4 p = pts[order[j]];
      unsigned c = __float_as_uint(p.w);
      if (j == b) { fx = p.x; fy = p.y; fz = p.z; }
      sx += p.x; sy += p.y; sz += p.z;
      sr += (float)((c >> 16) & 255u);
      sg += (float)((c >> 8) & 255u);
      sb += (float)(c & 255u);
      sa += (float)((c >> 24) & 255u);
    }
    float cnt = (float)(e - b);
    float4 o;
    o.x = sx / cnt; o.y = sy / cnt; o.z = sz / cnt;
    unsigned rgba = ((unsigned)(sa / cnt) << 24) | ((unsigned)(sr / cnt) << 16) | ((unsigned)(sg / cnt) << 8) |
                    (unsigned)(sb / cnt);
    o.w = __uint_as_float(rgba);
    out[v] = o;
    acc.add(o);
    // The true mean lies inside the voxel, within one leaf of every member; the float sums of a voxel with very many points far
    // from the origin can carry it away.  mm3d_cloud::voxel_leaf promises the grids that it has not gone far.
    if (!(fabsf(o.x - fx) <= two_leaves && fabsf(o.y - fy) <= two_leaves && fabsf(o.z - fz) <= two_leaves)) nvox_dev[1] = 1;
  }
  acc.flush_slot(box);
}

// Does VoxelGrid(leaf = resolution) return this cloud unchanged, bit for bit?  True when every point is finite, the voxel
// keys ascend strictly in input order (one point per voxel, and the output's voxel order is the input's order) and no
// coordinate is -0.0f (the centroid 0.f + (-0.0f) is +0.0f).  That is the normal case of SIFT's first octave: the
// reference calls detectKeypoints with min_scale = resolution on a cloud downSample(resolution) produced
// (R/src/map_merging.cpp:231-233), whose points lie one to a voxel of the same global lattice.
__global__ void k_voxel_identity(const float4 *__restrict__ pts, int n, float inv, int minbx, int minby, int minbz, int mul1, int mul2,
                                 int *__restrict__ not_identity)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto key_of = [&](const float4 &p, bool &ok) {
    ok = isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && __float_as_uint(p.x) != 0x80000000u && __float_as_uint(p.y) != 0x80000000u &&
         __float_as_uint(p.z) != 0x80000000u;
    const int ijk0 = (int)(floorf(__fmul_rn(p.x, inv)) - (float)minbx);
    const int ijk1 = (int)(floorf(__fmul_rn(p.y, inv)) - (float)minby);
    const int ijk2 = (int)(floorf(__fmul_rn(p.z, inv)) - (float)minbz);
    return (uint32_t)(ijk0 + ijk1 * mul1 + ijk2 * mul2);
  };
  bool ok, ok_prev = true;
  const uint32_t k = key_of(pts[i], ok);
  uint32_t kp = 0;
  if (i > 0) kp = key_of(pts[i - 1], ok_prev);
  if (!ok || (i > 0 && !(kp < k))) *not_identity = 1;      // (benign race: every writer stores 1)
}

struct VoxelSetup { bool overflow; float inv; int min_b[3], div_b[3]; };
static VoxelSetup voxel_setup(const mm3d_cloud *in, float leaf)
{
  VoxelSetup v;
  v.inv = 1.0f / leaf;
  const float *mn = in->bmin, *mx = in->bmax;
  // VoxelGrid's overflow guard: too many voxels => the input is returned unchanged
  auto i64 = [](float x) -> int64_t {
    if (!(x > -9.2e18f && x < 9.2e18f)) return INT64_MAX / 4;
    return (int64_t)x;
  };
  const int64_t dx = i64((mx[0] - mn[0]) * v.inv) + 1, dy = i64((mx[1] - mn[1]) * v.inv) + 1, dz = i64((mx[2] - mn[2]) * v.inv) + 1;
  const long double prod = (long double)dx * (long double)dy * (long double)dz;
  v.overflow = prod > (long double)INT32_MAX || !(leaf > 0.0f);
  for (int a = 0; a < 3; ++a) {
    v.min_b[a] = v.div_b[a] = 0;
    if (v.overflow) continue;
    v.min_b[a] = (int)std::floor(mn[a] * v.inv);
    const int max_b = (int)std::floor(mx[a] * v.inv);
    v.div_b[a] = max_b - v.min_b[a] + 1;
  }
  return v;
}

bool downsample_is_identity(Context *c, const mm3d_cloud *in_, double resolution)
{
  auto *in = const_cast<mm3d_cloud *>(in_);
  cloud_bbox(c, in);
  const int n = (int)in->n;
  if (n == 0 || in->n_finite != in->n) return false;
  const VoxelSetup v = voxel_setup(in, (float)resolution);
  if (v.overflow) return false;
  DevBuf<int> flag(c, 1);
  MM3D_HIP(hipMemsetAsync(flag.get(), 0, sizeof(int), c->stream));
  MM3D_LAUNCH(c, "voxel_identity", n * 16.0, k_voxel_identity, dim3(div_up(n, 256)), dim3(256), 0, in->pts.get(), n, v.inv, v.min_b[0], v.min_b[1],
              v.min_b[2], v.div_b[0], v.div_b[0] * v.div_b[1], flag.get());
  int *h = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(h, flag.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  return h[0] == 0;
}

mm3d_cloud *downsample(Context *c, const mm3d_cloud *in_, double resolution)
{
  auto *in = const_cast<mm3d_cloud *>(in_);
  cloud_bbox(c, in);
  const int n = (int)in->n;
  if (n == 0 || in->n_finite == 0) return cloud_from_device(c, DevBuf<float4>(c, 0), 0);
  const float leaf = (float)resolution;
  const VoxelSetup vs = voxel_setup(in, leaf);
  const float inv = vs.inv;
  if (vs.overflow) {
    DevBuf<float4> copy(c, in->n);
    MM3D_HIP(hipMemcpyAsync(copy.get(), in->pts.get(), in->n * 16, hipMemcpyDeviceToDevice, c->stream));
    return cloud_from_device(c, std::move(copy), in->n);
  }
  const int *min_b = vs.min_b, *div_b = vs.div_b;
  const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
  DevBuf<uint32_t> keys(c, n), vals(c, n), keys2(c, n), vals2(c, n);
  MM3D_LAUNCH(c, "voxel_keys", n * 24.0, k_voxel_keys, dim3(div_up(n, 256)), dim3(256), 0, in->pts.get(), n, inv,
              min_b[0], min_b[1], min_b[2], mul1, mul2, keys.get(), vals.get());
  // pcl::VoxelGrid sorts (voxel index, point) pairs; the order inside a voxel is the input order here (stable),
  // by counting sort (grid.hip), or by rocPRIM's radix sort when some bin of the counting sort is very long
  DevBuf<int> too_long(c, 1);
  counting_sort_pairs_u32(c, keys.get(), n, (uint64_t)div_b[0] * (uint64_t)div_b[1] * (uint64_t)div_b[2], keys2.get(), vals2.get(),
                          too_long.get(), in->n_finite == in->n);
  // One wait for the whole filter: the centroid launch is sized by its bound and reads the voxel count on the device, so the
  // count, the counting sort's "bin too long" word and the centroids' bounding box come back together (the new cloud needs no
  // k_bbox launch and no wait of its own).
  const size_t bound = in->n_finite;                            // a voxel holds at least one finite point
  DevBuf<int> starts(c, (size_t)n + 1);
  // (240 blocks: their slots come back in ONE 7.7 KB copy.  Round 4's 512 blocks made it 16.5 KB, and the HIP runtime hands a
  // device-to-host copy of more than 16 KB to an SDMA engine instead of its blit kernel: every kernel trace since showed 6 - 8 ms
  // without a kernel behind k_voxel_centroid at every step start, all feature workers parked at this wait -- DESIGN.md section 6)
  static const unsigned cblocks_cap = [] { const char *e = getenv("MM3D_VOXEL_BLOCKS"); return e ? (unsigned)std::max(1, atoi(e)) : 240u; }();
  const unsigned cblocks = std::min<unsigned>(div_up(bound, 256), cblocks_cap);
  DevBuf<unsigned> ctl(c, 16 + 8 * (size_t)cblocks);             // [0] voxels, [16 + 8 b ..] block b's box of centroids
  DevBuf<float4> out(c, bound);
  const size_t ctl_bytes = (16 + 8 * (size_t)cblocks) * sizeof(unsigned);
  unsigned *h = (unsigned *)c->pin(64 + ctl_bytes);
  for (int attempt = 0; attempt < 2; ++attempt) {
    scan_fused(c, "voxel_starts", n * 12.0, (size_t)n + 1, VoxelHeadLoad{keys2.get(), n}, VoxelStartStore{n, starts.get(), (int *)ctl.get()});
    // 16 B read per raw point + 16 B written per voxel
    MM3D_LAUNCH(c, "voxel_centroid", in->n_finite * 32.0, k_voxel_centroid, dim3(cblocks), dim3(256), 0, in->pts.get(),
                (const uint32_t *)vals2.get(), (const int *)starts.get(), (int *)ctl.get(), attempt == 0 ? (const int *)too_long.get() : (const int *)nullptr, (int)in->n_finite, 2.0f * leaf,
                out.get(), ctl.get() + 16);
    MM3D_HIP(hipMemcpyAsync(h + 16, ctl.get(), ctl_bytes, hipMemcpyDeviceToHost, c->stream));
    MM3D_HIP(hipMemcpyAsync(h + 15, too_long.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (attempt == 1 || !h[15]) break;
    // (a bin of the counting sort was too long and nothing was placed: the launches above found no voxel; radix sort, once more)
    sort_pairs_u32(c, keys.get(), keys2.get(), vals.get(), vals2.get(), n, 32);
  }
  const size_t nvox = h[16];
  // The output was allocated at its bound (one wait for the whole filter).  A large cloud that shrank to less than half of it
  // (2 M raw points -> 200 k voxels would keep 32 MB for the cloud's lifetime, times sixteen streams) moves to a buffer
  // of its size; small or well-filled ones stay (a copy dispatch per SIFT octave would cost the headline more than the
  // memory is worth: 406 k -> 100 k points is 6.5 MB for the octave's lifetime).
  if (bound >= ((size_t)1 << 20) && nvox * 2 < bound) {
    DevBuf<float4> fit(c, nvox);
    if (nvox) MM3D_HIP(hipMemcpyAsync(fit.get(), out.get(), nvox * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    c->settle();                                      // (`out` returns to the pool behind the copy)
    out = std::move(fit);
  }
  mm3d_cloud *res = cloud_from_device(c, std::move(out), nvox);
  unsigned box[8];
  box_of_slots(h + 32, cblocks, box);
  cloud_set_bbox(res, box);
  if (h[17] == 0) res->voxel_leaf = leaf;              // every centroid stayed with its voxel (types.hpp: what cloud_grid makes of it)
  c->settle();
  return res;
}

// ---------------------------------------------------------------- radius outlier removal
// One thread per (cell-sorted) query; counts candidates with d2 <= thr and stops at need.
__global__ void __launch_bounds__(256)
k_radius_count(GridView g, float radius, float thr, int need, int *__restrict__ keep /* by original index */)
{
  unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  int i = bid * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  float4 q = g.pts[i];
  int cnt = 0;
  for_each_candidate(g, q.x, q.y, q.z, radius, [&](const float4 &p) {
    float d = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
    cnt += (d <= thr) ? 1 : 0;
    return cnt < need;
  });
  keep[__float_as_int(q.w)] = cnt >= need ? 1 : 0;
}

mm3d_cloud *remove_outliers(Context *c, const mm3d_cloud *in, double radius, int min_neighbours)
{
  const size_t n = in->n;
  if (n == 0) return cloud_from_device(c, DevBuf<float4>(c, 0), 0);
  // largest float whose double value does not exceed r*r: (double)d2 > r*r  <=>  d2 > thr
  const double r2 = radius * radius;
  float thr = (float)r2;
  if ((double)thr > r2) thr = std::nextafterf(thr, -INFINITY);
  const int need = min_neighbours + 1;
  DevBuf<int> keep(c, n + 1);
  if (need <= 0) {
    // k == 0 neighbours requested: everything passes
    std::vector<int> ones(n + 1, 1);
    MM3D_HIP(hipMemcpyAsync(keep.get(), ones.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->sync();
  } else {
    MM3D_HIP(hipMemsetAsync(keep.get(), 0, (n + 1) * sizeof(int), c->stream));
    const Grid &g = cloud_grid(c, in, (float)(radius * 0.5));
    if (g.n) {
      // 12 B read + 1 B mask per point (we read 16 and write a 4 B flag: 20 B)
      MM3D_LAUNCH(c, "radius_outlier_count", g.n * 13.0, k_radius_count, dim3(div_up(g.n, 256)), dim3(256), 0,
                  g.view(), (float)radius, thr, need, keep.get());
    }
  }
  DevBuf<float4> out;
  unsigned box[7];
  size_t m = compact_points(c, in->pts.get(), keep.get(), n, out, box);     // (the kept points' bounding box comes back with the count)
  if (n >= ((size_t)1 << 20) && m * 2 < n) {           // as in downsample: a large, mostly empty bound-sized buffer is not kept
    DevBuf<float4> fit(c, m);
    if (m) MM3D_HIP(hipMemcpyAsync(fit.get(), out.get(), m * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    c->settle();
    out = std::move(fit);
  }
  mm3d_cloud *res = cloud_from_device(c, std::move(out), m);
  cloud_set_bbox(res, box);
  res->voxel_leaf = in->voxel_leaf;                    // a subset of a voxel grid's centroids
  return res;
}

// ---------------------------------------------------------------- composeMaps: transform + concat
__global__ void k_transform_into(const float4 *__restrict__ in, size_t n, const float *__restrict__ T16,
                                 float4 *__restrict__ out)
{
  __shared__ float T[16];
  if (threadIdx.x < 16) T[threadIdx.x] = T16[threadIdx.x];
  __syncthreads();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = in[i];
  float3 r = xform(T, p.x, p.y, p.z);
  out[i] = make_float4(r.x, r.y, r.z, p.w);
}

mm3d_cloud *transform_concat(Context *c, const mm3d_cloud *const *clouds, size_t n, const float *T)
{
  size_t total = 0;
  for (size_t i = 0; i < n; ++i) {
    bool zero = true;
    for (int k = 0; k < 16; ++k) zero = zero && (T[i * 16 + k] == 0.0f);
    if (!zero && clouds[i]) total += clouds[i]->n;
  }
  DevBuf<float4> out(c, total);
  DevBuf<float> dT(c, n * 16 + 16);
  if (n) MM3D_HIP(hipMemcpyAsync(dT.get(), T, n * 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  size_t off = 0;
  for (size_t i = 0; i < n; ++i) {
    bool zero = true;
    for (int k = 0; k < 16; ++k) zero = zero && (T[i * 16 + k] == 0.0f);
    if (zero || !clouds[i] || clouds[i]->n == 0) continue;
    MM3D_LAUNCH(c, "transform_cloud", clouds[i]->n * 32.0, k_transform_into, dim3(div_up(clouds[i]->n, 256)), dim3(256), 0,
                clouds[i]->pts.get(), clouds[i]->n, dT.get() + i * 16, out.get() + off);
    off += clouds[i]->n;
  }
  c->sync();
  return cloud_from_device(c, std::move(out), total);
}

}  // namespace mm3d
