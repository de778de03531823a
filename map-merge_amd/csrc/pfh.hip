// pfh.hip -- computeLocalDescriptors(PFH) on gfx950: the reference's DEFAULT descriptor.
//
// R/src/features.cpp:99-150 with the PFH row of R/src/dispatch_descriptors.h:38:
// pcl::PFHEstimation<PointXYZRGB, Normal, PFHSignature125> with setRadiusSearch(feature_radius),
// setSearchSurface(points), setInputNormals(normals), setInputCloud(keypoints); descriptors with a
// non-finite bin are pruned together with their keypoints (features.cpp:118-143).
//   per keypoint: its radius neighbours N (|N| = m ~ 200), then EVERY pair (i, j < i) of them
//   (m (m-1) / 2 ~ 20 000 Darboux pair features, pcl::computePairFeatures) binned 5 x 5 x 5, each
//   hit adding 100 / (m (m-1) / 2).
//
// One block per keypoint.  The neighbours (position, normal, distance key) are gathered into LDS
// once, the pairs are dealt round-robin to the 256 threads, and -- as in SPFH -- every hit adds the
// SAME float, so bins are counted in integers (LDS atomics, order free) and the float chain
// "0 + incr + incr + ..." is replayed once per bin at the end.  A pair's roles (p1 = the later one in
// the distance-sorted neighbour list) only matter when |angle1| == |angle2| exactly; they are
// reproduced from the (distance, index) keys.
// Algorithmic work: m (m-1) / 2 pair features per keypoint (~300 VALU instructions each); bytes: 32 B
// per gathered neighbour + 500 B per descriptor.
#include "device_util.hpp"

namespace mm3d {

constexpr int kPfhSplit = 5;
constexpr int kPfhDim = kPfhSplit * kPfhSplit * kPfhSplit;
constexpr int kPfhCap = 1024;     // neighbours held in LDS; larger neighbourhoods go through global scratch

struct PfhNb {
  float x, y, z, d2;              // position, squared distance to the keypoint
  float nx, ny, nz;               // normal
  int idx;                        // original index (tie-break of the sorted neighbour order)
};

// pcl::computePairFeatures (features/src/pfh.cpp) -- same text as fpfh.hip
__device__ __forceinline__ void pfh_pair_features(const PfhNb &p1, const PfhNb &p2, float &f1, float &f2, float &f3)
{
  float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
  const float f4 = sqrtf(dx * dx + dy * dy + dz * dz);
  if (f4 == 0.0f) { f1 = f2 = f3 = 0.0f; return; }
  float ax = p1.nx, ay = p1.ny, az = p1.nz, bx = p2.nx, by = p2.ny, bz = p2.nz;
  const float angle1 = (ax * dx + ay * dy + az * dz) / f4;
  const float angle2 = (bx * dx + by * dy + bz * dz) / f4;
  // acos(fabs(angle1)) > acos(fabs(angle2)) in double: device_util.hpp::acos_abs_greater
  if (acos_abs_greater(angle1, angle2)) {
    float t;
    t = ax; ax = bx; bx = t; t = ay; ay = by; by = t; t = az; az = bz; bz = t;
    dx *= -1.0f; dy *= -1.0f; dz *= -1.0f;
    f3 = -angle2;
  } else {
    f3 = angle1;
  }
  float vx = dy * az - dz * ay, vy = dz * ax - dx * az, vz = dx * ay - dy * ax;
  const float v_norm = sqrtf(vx * vx + vy * vy + vz * vz);
  if (v_norm == 0.0f) { f1 = f2 = f3 = 0.0f; return; }
  vx /= v_norm; vy /= v_norm; vz /= v_norm;
  const float wx = ay * vz - az * vy, wy = az * vx - ax * vz, wz = ax * vy - ay * vx;
  f2 = vx * bx + vy * by + vz * bz;
  f1 = lm::atan2f_glibc(wx * bx + wy * by + wz * bz, ax * bx + ay * by + az * bz);   // glibc's atan2f, bit for bit (libm_exact.hpp)
}

// pcl::computeRGBPairFeatures (features/src/pfh.cpp), PFHRGB: the frame always sits on p1 (no swap),
// f3 = angle1, and f5..f7 are colour ratios taken with INTEGER division (Eigen::Vector4i colours),
// folded into [-1, 1] by f > 1 -> -1 / f.  The neighbour's rgba travels in PfhNb::d2.
__device__ __forceinline__ void pfhrgb_pair_features(const PfhNb &p1, const PfhNb &p2, float &f1, float &f2, float &f3, float &f5,
                                                     float &f6, float &f7)
{
  const float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
  const float f4 = sqrtf(dx * dx + dy * dy + dz * dz);
  if (f4 == 0.0f) { f1 = f2 = f3 = f5 = f6 = f7 = 0.0f; return; }
  const float ax = p1.nx, ay = p1.ny, az = p1.nz, bx = p2.nx, by = p2.ny, bz = p2.nz;
  f3 = (ax * dx + ay * dy + az * dz) / f4;
  float vx = dy * az - dz * ay, vy = dz * ax - dx * az, vz = dx * ay - dy * ax;
  const float v_norm = sqrtf(vx * vx + vy * vy + vz * vz);
  if (v_norm == 0.0f) { f1 = f2 = f3 = f5 = f6 = f7 = 0.0f; return; }
  vx /= v_norm; vy /= v_norm; vz /= v_norm;
  const float wx = ay * vz - az * vy, wy = az * vx - ax * vz, wz = ax * vy - ay * vx;
  f2 = vx * bx + vy * by + vz * bz;
  f1 = lm::atan2f_glibc(wx * bx + wy * by + wz * bz, ax * bx + ay * by + az * bz);   // glibc's atan2f, bit for bit (libm_exact.hpp)
  const unsigned c1 = __float_as_uint(p1.d2), c2 = __float_as_uint(p2.d2);
  float r[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sh = 16 - 8 * c;
    const int a1 = (int)((c1 >> sh) & 0xffu), a2 = (int)((c2 >> sh) & 0xffu);
    float q = (a2 != 0) ? (float)(a1 / a2) : 1.0f;
    if (q > 1.0f) q = -1.0f / q;
    r[c] = q;
  }
  f5 = r[0]; f6 = r[1]; f7 = r[2];
}

// static_cast<int>(floor(x)) with x86 semantics for NaN / out of range (INT_MIN -> clamped to 0)
__device__ __forceinline__ int pfh_bin(double x)
{
  const double f = floor(x);
  int h = (f >= -2147483648.0 && f <= 2147483647.0) ? (int)f : (-2147483647 - 1);
  h = h < 0 ? 0 : h;
  return h >= kPfhSplit ? kPfhSplit - 1 : h;
}

// rows: optional list of keypoints to process (the ones whose neighbourhood overflowed LDS), with
// their neighbours then kept in `scratch` (cap_scratch entries per row)
// kRgb: PFHRGBSignature250 (pfhrgb.hpp): every ORDERED pair (i, j != i), 125 geometry + 125 colour bins,
// no "no neighbours" branch (the row stays zero and valid); pts = the cloud in original order (rgba)
template <bool kRgb>
__global__ void __launch_bounds__(256)
k_pfh(const float4 *__restrict__ kp, int nk, GridView g, const float4 *__restrict__ nrm /* original order */,
      const float4 *__restrict__ pts, float radius,
      float r2, const int *__restrict__ rows, PfhNb *__restrict__ scratch, int cap, float *__restrict__ desc /* [nk][125] */,
      int *__restrict__ valid, int *__restrict__ overflow /* [0] count, [1..] keypoint ids, [nk + 1] max count */)
{
  __shared__ PfhNb s_nb[kPfhCap];
  constexpr int kBins = kRgb ? 2 * kPfhDim : kPfhDim;
  __shared__ unsigned s_hist[kBins];
  __shared__ int s_m;
  const int k = rows ? rows[blockIdx.x] : (int)blockIdx.x;
  PfhNb *nb = rows ? scratch + (size_t)blockIdx.x * cap : s_nb;
  const int tid = threadIdx.x;
  if (tid == 0) s_m = 0;
  if (tid < kBins) s_hist[tid] = 0u;
  __syncthreads();
  const float4 q = kp[k];
  // 1. gather the radius neighbours (any order; the sorted order is carried by the keys)
  const float ri = radius * 1.0001f + 1e-4f;
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
        for (int j = b + tid; j < e; j += 256) {
          const float4 p = g.pts[j];
          const float d2 = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
          if (d2 < r2) {
            const int slot = atomicAdd(&s_m, 1);
            if (slot < cap) {
              const int oi = __float_as_int(p.w);
              const float4 n = nrm[oi];
              PfhNb v;
              v.x = p.x; v.y = p.y; v.z = p.z; v.d2 = kRgb ? pts[oi].w : d2; v.nx = n.x; v.ny = n.y; v.nz = n.z; v.idx = oi;
              nb[slot] = v;
            }
          }
        }
      }
  }
  __syncthreads();
  const int m = s_m;
  if (m > cap) {
    // does not fit: leave it to the scratch pass (first pass only; the scratch pass is sized to fit)
    if (tid == 0 && !rows) {
      const int o = atomicAdd(&overflow[0], 1);
      overflow[1 + o] = k;
      atomicMax(&overflow[nk + 1], m);
    }
    return;
  }
  float *out = desc + (size_t)k * kBins;
  if (m == 0 && !kRgb) {
    if (tid < kPfhDim) out[tid] = __uint_as_float(0x7fc00000u);   // searchForNeighbors == 0: NaN row, pruned below
    if (tid == 0) valid[k] = 0;
    return;
  }
  if (kRgb) {
    // all ordered pairs: m (m - 1) of them, row i pairs with every j != i
    const float d_pi = 1.0f / (2.0f * 3.14159274f);
    const long long total = (long long)m * (m - 1);
    for (long long t = tid; t < total; t += 256) {
      const int i = (int)(t / (m - 1)), c = (int)(t % (m - 1));
      const int j = c < i ? c : c + 1;
      float f1, f2, f3, f5, f6, f7;
      pfhrgb_pair_features(nb[i], nb[j], f1, f2, f3, f5, f6, f7);
      const int h1 = pfh_bin(kPfhSplit * (((double)f1 + 3.14159265358979323846) * (double)d_pi));
      const int h2 = pfh_bin(kPfhSplit * (((double)f2 + 1.0) * 0.5));
      const int h3 = pfh_bin(kPfhSplit * (((double)f3 + 1.0) * 0.5));
      atomicAdd(&s_hist[h1 + kPfhSplit * h2 + kPfhSplit * kPfhSplit * h3], 1u);
      const int h5 = pfh_bin(kPfhSplit * (((double)f5 + 1.0) * 0.5));
      const int h6 = pfh_bin(kPfhSplit * (((double)f6 + 1.0) * 0.5));
      const int h7 = pfh_bin(kPfhSplit * (((double)f7 + 1.0) * 0.5));
      atomicAdd(&s_hist[kPfhDim + h5 + kPfhSplit * h6 + kPfhSplit * kPfhSplit * h7], 1u);
    }
    __syncthreads();
    if (tid < kBins) {
      const unsigned long long pairs = (unsigned long long)m * (unsigned long long)(m - 1) / 2ull;
      const float hist_incr = 100.0f / (float)pairs;
      const unsigned hits = s_hist[tid];
      out[tid] = float_chain_sum(hist_incr, hits);
    }
    if (tid == 0) valid[k] = 1;
    return;
  }
  // 2. all pairs (i, j < i).  Rows i and m-1-i together hold exactly m-1 pairs, so "super rows" of
  // equal length are dealt to the threads by one division per pair.
  const int half = (m + 1) / 2, len = m - 1;
  const float d_pi = 1.0f / (2.0f * 3.14159274f);
  const long long total = (long long)half * len;
  for (long long t = tid; t < total; t += 256) {
    const int s = (int)(t / len), c = (int)(t % len);
    int i, j;
    if (c < s) { i = s; j = c; }
    else {
      i = m - 1 - s; j = c - s;
      if (i == s) continue;                    // odd m: the middle row pairs with itself, count it once
    }
    const PfhNb a = nb[i], b = nb[j];
    // p1 = the later one in the (distance, index)-sorted neighbour list
    const bool a_later = a.d2 > b.d2 || (a.d2 == b.d2 && a.idx > b.idx);
    float f1, f2, f3;
    if (a_later) pfh_pair_features(a, b, f1, f2, f3);
    else pfh_pair_features(b, a, f1, f2, f3);
    const int h1 = pfh_bin(kPfhSplit * (((double)f1 + 3.14159265358979323846) * (double)d_pi));
    const int h2 = pfh_bin(kPfhSplit * (((double)f2 + 1.0) * 0.5));
    const int h3 = pfh_bin(kPfhSplit * (((double)f3 + 1.0) * 0.5));
    atomicAdd(&s_hist[h1 + kPfhSplit * h2 + kPfhSplit * kPfhSplit * h3], 1u);
  }
  __syncthreads();
  // 3. replay the float chain per bin
  if (tid < kPfhDim) {
    const unsigned long long pairs = (unsigned long long)m * (unsigned long long)(m - 1) / 2ull;
    const float hist_incr = 100.0f / (float)pairs;
    const unsigned hits = s_hist[tid];
    out[tid] = float_chain_sum(hist_incr, hits);
  }
  if (tid == 0) valid[k] = 1;
}

__global__ void k_pfh_compact(const float *__restrict__ in, const int *__restrict__ flags, const int *__restrict__ pos, int n,
                              int dim, float *__restrict__ out)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n * dim) return;
  const int r = (int)(e / dim), cidx = (int)(e % dim);
  if (flags[r]) out[(size_t)pos[r] * dim + cidx] = in[e];
}

template <bool kRgb>
static mm3d_desc *compute_pfh_impl(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, mm3d_cloud *keypoints, double radius)
{
  MM3D_REQUIRE(normals->n == points->n, "computeLocalDescriptors: normals and points differ in size");
  constexpr int kDim = kRgb ? 2 * kPfhDim : kPfhDim;
  auto *res = new mm3d_desc();
  res->dim = kDim;
  res->type = kRgb ? MM3D_DESC_PFHRGB : MM3D_DESC_PFH;
  const int nk = (int)keypoints->n;
  if (nk == 0) { res->n = 0; res->data = DevBuf<float>(c, 0); return res; }
  const float r2 = (float)(radius * radius);
  const Grid &g = cloud_grid(c, points, (float)(radius * 0.5));
  auto drop_all = [&]() {
    res->n = 0; res->data = DevBuf<float>(c, 0);
    keypoints->pts = DevBuf<float4>(c, 0); keypoints->n = 0; keypoints->grids.clear(); keypoints->host.clear();
    keypoints->reset_caches();
  };
  if (g.n == 0 && !kRgb) { drop_all(); return res; }   // no surface: every PFH descriptor is NaN and gets pruned
  DevBuf<float> raw(c, (size_t)nk * kDim);
  if (g.n == 0) {                                       // PFHRGB keeps all-zero rows
    MM3D_HIP(hipMemsetAsync(raw.get(), 0, (size_t)nk * kDim * sizeof(float), c->stream));
    res->n = (size_t)nk; res->data = std::move(raw);
    c->sync();
    return res;
  }
  DevBuf<int> valid(c, (size_t)nk + 1), overflow(c, (size_t)nk + 2);
  MM3D_HIP(hipMemsetAsync(valid.get(), 0, ((size_t)nk + 1) * sizeof(int), c->stream));
  MM3D_HIP(hipMemsetAsync(overflow.get(), 0, ((size_t)nk + 2) * sizeof(int), c->stream));
  MM3D_LAUNCH(c, "pfh", nk * (200.0 * 32.0 + kDim * 4.0), (k_pfh<kRgb>), dim3(nk), dim3(256), 0, (const float4 *)keypoints->pts.get(), nk,
              g.view(), (const float4 *)normals->nrm.get(), (const float4 *)points->pts.get(), (float)radius, r2, (const int *)nullptr,
              (PfhNb *)nullptr, kPfhCap, raw.get(), valid.get(), overflow.get());
  int *h = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(h, overflow.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MM3D_HIP(hipMemcpyAsync(h + 1, overflow.get() + nk + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  if (h[0] > 0) {
    // neighbourhoods beyond the LDS capacity: the same kernel with the neighbour list in global scratch
    const int n_over = h[0], max_m = h[1];
    const double bytes = (double)n_over * max_m * sizeof(PfhNb);
    if (bytes > 8e9) throw Error(MM3D_EUNSUPPORTED, "PFH: neighbourhoods too large for the scratch pass (reduce descriptor_radius)");
    DevBuf<PfhNb> scratch(c, (size_t)n_over * max_m);
    MM3D_LAUNCH(c, "pfh", n_over * (max_m * 32.0 + kDim * 4.0), (k_pfh<kRgb>), dim3(n_over), dim3(256), 0,
                (const float4 *)keypoints->pts.get(), nk, g.view(), (const float4 *)normals->nrm.get(), (const float4 *)points->pts.get(),
                (float)radius, r2, (const int *)(overflow.get() + 1), scratch.get(), max_m, raw.get(), valid.get(), overflow.get());
    c->sync();
  }
  // prune invalid descriptors and the same keypoints (features.cpp:118-143)
  DevBuf<int> vpos(c, (size_t)nk + 1);
  exclusive_scan_int(c, valid.get(), vpos.get(), (size_t)nk + 1);
  MM3D_HIP(hipMemcpyAsync(h, vpos.get() + nk, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  const int nv = h[0];
  res->n = (size_t)nv;
  if (nv == nk) {
    res->data = std::move(raw);
  } else {
    res->data = DevBuf<float>(c, (size_t)nv * kDim);
    DevBuf<float4> kp2(c, nv);
    if (nv) {
      MM3D_LAUNCH(c, "compact_rows", nk * kDim * 8.0, k_pfh_compact, dim3(div_up((size_t)nk * kDim, 256)), dim3(256), 0,
                  (const float *)raw.get(), (const int *)valid.get(), (const int *)vpos.get(), nk, kDim, res->data.get());
      MM3D_LAUNCH(c, "compact_rows", nk * 32.0, k_pfh_compact, dim3(div_up((size_t)nk * 4, 256)), dim3(256), 0,
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

// test hook for float_chain_sum (device_util.hpp): out[i] = the float chain 0 + incr[i] + ... (hits[i] times)
__global__ void k_float_chain(const float *__restrict__ incr, const unsigned *__restrict__ hits, int n, float *__restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = float_chain_sum(incr[i], hits[i]);
}

void debug_float_chain(Context *c, const float *incr_host, const unsigned *hits_host, int n, float *out_host)
{
  if (n <= 0) return;
  DevBuf<float> di(c, n), dout(c, n);
  DevBuf<unsigned> dh(c, n);
  MM3D_HIP(hipMemcpyAsync(di.get(), incr_host, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  MM3D_HIP(hipMemcpyAsync(dh.get(), hits_host, n * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
  MM3D_LAUNCH(c, "debug", n * 12.0, k_float_chain, dim3(div_up(n, 64)), dim3(64), 0, (const float *)di.get(), (const unsigned *)dh.get(), n,
              dout.get());
  MM3D_HIP(hipMemcpyAsync(out_host, dout.get(), n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  c->sync();
}

mm3d_desc *compute_pfh(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, mm3d_cloud *keypoints, double radius)
{
  return compute_pfh_impl<false>(c, points, normals, keypoints, radius);
}

mm3d_desc *compute_pfhrgb(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, mm3d_cloud *keypoints, double radius)
{
  return compute_pfh_impl<true>(c, points, normals, keypoints, radius);
}

}  // namespace mm3d
