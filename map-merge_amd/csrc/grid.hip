// grid.hip -- context plumbing, cloud objects and the device-built uniform grid (K1 in SURVEY 2.2).
//
// The grid replaces every kd-tree the reference builds through PCL (normals, outlier filter,
// FPFH, SIFT, ICP target, SAC-IA target, transformScore target: R/src/features.cpp:34,50,105,171,
// R/src/matching.cpp:159,204,263).  It is built once per (cloud, cell size) and cached on the
// cloud; PCL rebuilds its tree per call and per pair.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "device_util.hpp"
#include "scan_fused.hpp"

namespace mm3d {

// (the pool, the context's waits and profiling scopes and the scan bookkeeping are host-only code: runtime.cpp)

// ---------------------------------------------------------------- rocPRIM-backed primitives
// Exclusive prefix sum of ints in ONE launch (30 per map: cell tables, compactions, work items): a chained scan
// with decoupled look-back.  A block takes a ticket (tiles are therefore started in order, whatever the
// dispatcher does), scans its tile of 4096 elements, publishes its aggregate, adds up its predecessors'
// published aggregates / prefixes (spinning on the few that are not there yet) and publishes its inclusive
// prefix.  A status word = epoch << 34 | state << 32 | value: the launch's epoch makes words of earlier launches
// read as "not there yet", and tickets count on from launch to launch, so nothing is cleared in between.
// The words carry their payload themselves and nothing else is read from another block, so the atomics are
// relaxed: an agent-scope release / acquire per tile would write back / invalidate caches on this multi-XCD part.
__global__ void __launch_bounds__(256)
k_scan_int(const int *__restrict__ in, int *__restrict__ out, size_t n, unsigned long long *status, unsigned *ticket,
           unsigned ticket_base, unsigned epoch)
{
  __shared__ unsigned s_tile;
  __shared__ int s_wave[4];
  __shared__ int s_prefix;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
  __syncthreads();
  const size_t tile = s_tile;
  const size_t base = tile * kScanTile + (size_t)threadIdx.x * kScanItems;
  int v[kScanItems];
  int sum = 0;
  const bool whole = base + kScanItems <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0;   // 16-byte loads where the thread's run is complete
  if (whole) {
#pragma unroll
    for (int k = 0; k < kScanItems; k += 4) {
      const int4 q = *reinterpret_cast<const int4 *>(in + base + k);
      v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] = base + k < n ? in[base + k] : 0;
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) sum += v[k];
  // block-wide exclusive scan of the per-thread sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = sum;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int t = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += t;
  }
  if (lane == kWave - 1) s_wave[wave] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
  const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  if (wave == 0) {
    // look-back by one wave: 64 predecessors per round, nearest first; the nearest published inclusive prefix ends it
    const unsigned long long tag = (unsigned long long)epoch << 34;
    if (lane == 0 && tile > 0)
      __hip_atomic_store(&status[tile], tag | (1ull << 32) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // aggregate
    int prefix = 0;
    for (long long hi = (long long)tile - 1; hi >= 0; hi -= kWave) {
      const long long t = hi - lane;
      unsigned long long w = 2ull << 32;                // lanes before tile 0: an empty prefix
      if (t >= 0) {
        do w = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((w >> 34) != epoch || ((w >> 32) & 3u) == 0u);
      }
      const unsigned long long is_prefix = ballot(((w >> 32) & 3u) == 2u);
      const int first = __ffsll((long long)is_prefix) - 1;          // nearest predecessor that carries a prefix (-1: none)
      const int take = (first < 0 || lane <= first) ? (int)(unsigned)(w & 0xffffffffull) : 0;
      prefix += wave_sum(take);
      if (first >= 0) break;
    }
    prefix = __shfl(prefix, 0, kWave);
    if (lane == 0) {
      __hip_atomic_store(&status[tile], tag | (2ull << 32) | (unsigned)(prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = prefix;
    }
  }
  __syncthreads();
  int run = s_prefix + wave_off + incl - sum;
  if (whole && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
#pragma unroll
    for (int k = 0; k < kScanItems; k += 4) {
      int4 q;
      q.x = run; run += v[k];
      q.y = run; run += v[k + 1];
      q.z = run; run += v[k + 2];
      q.w = run; run += v[k + 3];
      *reinterpret_cast<int4 *>(out + base + k) = q;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (base + k < n) out[base + k] = run;
      run += v[k];
    }
  }
}

void exclusive_scan_int(Context *c, const int *in, int *out, size_t n)
{
  if (n == 0) return;
  const ScanLaunchState st = scan_prepare(c, n);
  MM3D_LAUNCH(c, "scan_int", (double)n * 8.0, k_scan_int, dim3(st.tiles), dim3(256), 0, in, out, n, st.status, st.ticket, st.ticket_base, st.epoch);
}

void sort_pairs_u32(Context *c, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                    size_t n, int end_bit)
{
  if (n == 0) return;
  size_t tmp_bytes = 0;
  MM3D_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, n, 0, end_bit, c->stream));
  DevBuf<char> tmp(c, tmp_bytes ? tmp_bytes : 1);
  KernelScope ks(c, "rocprim_radix_sort_pairs", (double)n * 16.0);
  MM3D_HIP(rocprim::radix_sort_pairs(tmp.get(), tmp_bytes, kin, kout, vin, vout, n, 0, end_bit, c->stream));
}

// ---------------------------------------------------------------- cloud objects
__global__ void k_repack(const unsigned char *__restrict__ src, size_t n, size_t stride, size_t rgba_off,
                         float4 *__restrict__ dst)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char *p = src + i * stride;
  const float *f = reinterpret_cast<const float *>(p);
  float4 o;
  o.x = f[0]; o.y = f[1]; o.z = f[2];
  o.w = __uint_as_float(*reinterpret_cast<const unsigned *>(p + rgba_off));
  dst[i] = o;
}

__global__ void k_unpack(const float4 *__restrict__ src, size_t n, size_t stride, size_t rgba_off,
                         unsigned char *__restrict__ dst)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned char *p = dst + i * stride;
  float4 v = src[i];
  float *f = reinterpret_cast<float *>(p);
  f[0] = v.x; f[1] = v.y; f[2] = v.z;
  if (rgba_off != 12 && stride >= 16) f[3] = 1.0f;   // PCL's data[3]
  *reinterpret_cast<unsigned *>(p + rgba_off) = __float_as_uint(v.w);
}

mm3d_cloud *cloud_from_device(Context *c, DevBuf<float4> &&pts, size_t n)
{
  auto *cl = new mm3d_cloud();
  cl->pts = std::move(pts);
  cl->n = n;
  return cl;
}

mm3d_cloud *cloud_from_memory(Context *c, const void *src, size_t n, size_t stride, size_t rgba_off)
{
  MM3D_REQUIRE(stride >= 16 && stride % 4 == 0 && rgba_off % 4 == 0 && rgba_off >= 12 && rgba_off + 4 <= stride,
               "mm3d_cloud_create: stride/rgba_offset do not describe an x,y,z,rgba record");
  DevBuf<float4> pts(c, n);
  if (n) {
    MM3D_REQUIRE(src != nullptr, "mm3d_cloud_create: null points with n > 0");
    if (stride == 16 && rgba_off == 12) {
      MM3D_HIP(hipMemcpyAsync(pts.get(), src, n * 16, hipMemcpyDefault, c->stream));
    } else {
      DevBuf<unsigned char> stage(c, n * stride);
      MM3D_HIP(hipMemcpyAsync(stage.get(), src, n * stride, hipMemcpyDefault, c->stream));
      MM3D_LAUNCH(c, "repack", 0, k_repack, dim3(div_up(n, 256)), dim3(256), 0, stage.get(), n, stride, rgba_off, pts.get());
      c->settle();   // stage goes back to the pool only after the kernel is done with it
    }
  }
  return cloud_from_device(c, std::move(pts), n);
}

void cloud_download(Context *c, const mm3d_cloud *cl, void *dst, size_t stride, size_t rgba_off)
{
  MM3D_REQUIRE(stride >= 16 && stride % 4 == 0 && rgba_off % 4 == 0 && rgba_off >= 12 && rgba_off + 4 <= stride,
               "mm3d_cloud_download: stride/rgba_offset do not describe an x,y,z,rgba record");
  if (cl->n == 0) return;
  if (stride == 16 && rgba_off == 12) {
    MM3D_HIP(hipMemcpyAsync(dst, cl->pts.get(), cl->n * 16, hipMemcpyDefault, c->stream));
    c->sync();
    return;
  }
  DevBuf<unsigned char> stage(c, cl->n * stride);
  MM3D_HIP(hipMemsetAsync(stage.get(), 0, cl->n * stride, c->stream));
  MM3D_LAUNCH(c, "unpack", 0, k_unpack, dim3(div_up(cl->n, 256)), dim3(256), 0, cl->pts.get(), cl->n, stride, rgba_off, stage.get());
  MM3D_HIP(hipMemcpyAsync(dst, stage.get(), cl->n * stride, hipMemcpyDefault, c->stream));
  c->sync();
}

const std::vector<float4> &cloud_host(Context *c, const mm3d_cloud *cl, bool wait)
{
  auto *m = const_cast<mm3d_cloud *>(cl);
  std::lock_guard<std::recursive_mutex> lk(m->cache_mu);
  if (m->host.size() != m->n) {
    m->host.resize(m->n);
    if (m->n) {
      MM3D_HIP(hipMemcpyAsync(m->host.data(), m->pts.get(), m->n * 16, hipMemcpyDeviceToHost, c->stream));
      if (wait) c->sync();
    }
  }
  return m->host;
}

// ---------------------------------------------------------------- bounding box (finite points)
__global__ void k_bbox(const float4 *__restrict__ pts, size_t n, unsigned *__restrict__ out /* 6 ord + count */)
{
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int cnt = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 p = pts[i];
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
      mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
      ++cnt;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = fminf(mn[a], __shfl_down(mn[a], o, kWave));
      mx[a] = fmaxf(mx[a], __shfl_down(mx[a], o, kWave));
    }
  cnt = wave_sum(cnt);
  // one set of atomics per block (the seven words are shared by the whole grid)
  __shared__ float smn[4][3], smx[4][3];
  __shared__ int scnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    for (int a = 0; a < 3; ++a) { smn[wave][a] = mn[a]; smx[wave][a] = mx[a]; }
    scnt[wave] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      for (int a = 0; a < 3; ++a) { smn[0][a] = fminf(smn[0][a], smn[w][a]); smx[0][a] = fmaxf(smx[0][a], smx[w][a]); }
      scnt[0] += scnt[w];
    }
    for (int a = 0; a < 3; ++a) {
      atomicMin(&out[a], f2ord(smn[0][a]));
      atomicMax(&out[3 + a], f2ord(smx[0][a]));
    }
    atomicAdd(&out[6], (unsigned)scnt[0]);
  }
}

void cloud_bbox(Context *c, mm3d_cloud *cl)
{
  std::lock_guard<std::recursive_mutex> lk(cl->cache_mu);
  if (cl->have_bbox) return;
  cl->have_bbox = true;
  cl->n_finite = 0;
  if (cl->n == 0) return;
  DevBuf<unsigned> d(c, 8);
  unsigned init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
  unsigned *h = (unsigned *)c->pin(64);
  memcpy(h, init, sizeof(init));
  MM3D_HIP(hipMemcpyAsync(d.get(), h, sizeof(init), hipMemcpyHostToDevice, c->stream));
  unsigned blocks = std::min<unsigned>(div_up(cl->n, 256 * 8), 512);   // 256 threads, k_bbox assumes 4 waves
  MM3D_LAUNCH(c, "bbox", cl->n * 16.0, k_bbox, dim3(blocks), dim3(256), 0, cl->pts.get(), cl->n, d.get());
  MM3D_HIP(hipMemcpyAsync(h, d.get(), sizeof(init), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  cl->n_finite = h[6];
  if (cl->n_finite) {
    for (int a = 0; a < 3; ++a) { cl->bmin[a] = ord2f(h[a]); cl->bmax[a] = ord2f(h[3 + a]); }
  }
}

// ---------------------------------------------------------------- grid build
__global__ void k_cell_keys(const float4 *__restrict__ pts, int n, float minx, float miny, float minz, float inv,
                            int dx, int dy, int dz, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                            int *__restrict__ counts, uint32_t invalid_key)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { keys[i] = invalid_key; vals[i] = 0u; return; }
  int cx = clampi(cell_floor(p.x, minx, inv), 0, dx - 1);
  int cy = clampi(cell_floor(p.y, miny, inv), 0, dy - 1);
  int cz = clampi(cell_floor(p.z, minz, inv), 0, dz - 1);
  uint32_t key = (uint32_t)((cz * dy + cy) * dx + cx);
  keys[i] = key;
  vals[i] = (uint32_t)atomicAdd(&counts[key], 1);   // arrival rank in the cell (k_cell_scatter)
}

// Cell sort without a radix sort.  The keys are dense cell numbers, so a counting sort needs only
// the histogram the grid wants anyway: (1) k_cell_keys also keeps each point's arrival rank in its
// cell (atomicAdd's return value: arbitrary order), (2) after the scan every point is dropped into an
// arbitrary slot of its cell, (3) every point finds its STABLE place by counting the points of its
// cell with a smaller index -- a handful of L2-resident reads -- and is written there.  The result
// is exactly the stable sort by (cell, original index), with three small kernels and the scan.
// Cells longer than kCellSortMax (degenerate clouds) leave that to the rocPRIM path.
constexpr int kCellSortMax = 4096;
constexpr int kCellSortSmall = 32768;     // clouds up to this size never leave the counting sort (cloud_grid)

__global__ void k_cell_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ ranks, const int *__restrict__ cell_start,
                               int n, uint32_t invalid_key, int max_len, int *__restrict__ slots, int *__restrict__ too_long)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t key = keys[i];
  if (key == invalid_key) return;
  const int b = cell_start[key];
  slots[b + (int)ranks[i]] = i;
  if (cell_start[key + 1] - b > max_len) *too_long = 1;
}

__global__ void k_cell_place(const float4 *__restrict__ pts, const uint32_t *__restrict__ keys, const int *__restrict__ cell_start,
                             const int *__restrict__ slots, int n, uint32_t invalid_key, const int *__restrict__ too_long,
                             float4 *__restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || *too_long) return;
  const uint32_t key = keys[i];
  if (key == invalid_key) return;
  const int b = cell_start[key], e = cell_start[key + 1];
  int r = 0;
  for (int j = b; j < e; ++j) r += slots[j] < i ? 1 : 0;
  float4 p = pts[i];
  p.w = __int_as_float(i);
  out[b + r] = p;
}

__global__ void k_iota_u32(uint32_t *__restrict__ v, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

__global__ void k_gather_sorted(const float4 *__restrict__ pts, const uint32_t *__restrict__ order, int n,
                                float4 *__restrict__ out)
{
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t i = order[j];
  float4 p = pts[i];
  p.w = __int_as_float((int)i);
  out[j] = p;
}

const Grid &cloud_grid(Context *c, const mm3d_cloud *cl_, float cell)
{
  auto *cl = const_cast<mm3d_cloud *>(cl_);
  MM3D_REQUIRE(cell > 0.f && std::isfinite(cell), "grid cell size must be positive");
  std::lock_guard<std::recursive_mutex> lk(cl->cache_mu);
  int key = (int)std::lround((double)cell * 1e4);
  auto it = cl->grids.find(key);
  if (it != cl->grids.end()) return *it->second;
  cloud_bbox(c, cl);
  auto g = std::make_unique<Grid>();
  g->cell = cell;
  const size_t nfin = cl->n_finite;
  if (nfin == 0) {
    g->n = 0;
    g->cell_start = DevBuf<int>(c, 2);
    MM3D_HIP(hipMemsetAsync(g->cell_start.get(), 0, 8, c->stream));
    g->sorted = DevBuf<float4>(c, 1);
  } else {
    // bounded table: grow the cell when the box is huge (correctness does not depend on the size)
    for (;;) {
      double inv = 1.0 / (double)g->cell;
      double ex = std::floor(((double)cl->bmax[0] - cl->bmin[0]) * inv) + 2;
      double ey = std::floor(((double)cl->bmax[1] - cl->bmin[1]) * inv) + 2;
      double ez = std::floor(((double)cl->bmax[2] - cl->bmin[2]) * inv) + 2;
      if (ex * ey * ez <= 2.0e8) { g->dims[0] = (int)ex; g->dims[1] = (int)ey; g->dims[2] = (int)ez; break; }
      g->cell *= 1.5f;
    }
    for (int a = 0; a < 3; ++a) g->mn[a] = cl->bmin[a];
    const size_t ncell = (size_t)g->dims[0] * g->dims[1] * g->dims[2];
    const int n = (int)cl->n;
    // (the cell counts and the "a cell is too long for the counting sort" word share a buffer: one fill dispatch, not two --
    // in the 16-stream runs every dispatch, however small, waits in line behind the other streams' kernels)
    DevBuf<int> counts(c, ncell + 2);
    MM3D_HIP(hipMemsetAsync(counts.get(), 0, (ncell + 2) * sizeof(int), c->stream));
    int *const too_long = counts.get() + ncell + 1;
    DevBuf<uint32_t> keys(c, n), ranks(c, n);
    const uint32_t invalid = (uint32_t)ncell;   // sorts after every real cell
    MM3D_LAUNCH(c, "grid_cell_keys", n * 24.0, k_cell_keys, dim3(div_up(n, 256)), dim3(256), 0, cl->pts.get(), n,
                g->mn[0], g->mn[1], g->mn[2], 1.0f / g->cell, g->dims[0], g->dims[1], g->dims[2], keys.get(),
                ranks.get(), counts.get(), invalid);
    g->cell_start = DevBuf<int>(c, ncell + 1);
    exclusive_scan_int(c, counts.get(), g->cell_start.get(), ncell + 1);
    g->sorted = DevBuf<float4>(c, nfin);
    DevBuf<int> slots(c, nfin);
    // (a small cloud -- a map's keypoints -- is placed by counting whatever its cells hold: n^2 slot reads at worst, a
    // millisecond at 32 k points, and no question to ask)
    const bool small = n <= kCellSortSmall;
    MM3D_LAUNCH(c, "grid_cell_sort", n * 16.0, k_cell_scatter, dim3(div_up(n, 256)), dim3(256), 0, (const uint32_t *)keys.get(),
                (const uint32_t *)ranks.get(), (const int *)g->cell_start.get(), n, invalid, small ? n : kCellSortMax, slots.get(), too_long);
    MM3D_LAUNCH(c, "grid_cell_sort", n * 40.0, k_cell_place, dim3(div_up(n, 256)), dim3(256), 0, cl->pts.get(),
                (const uint32_t *)keys.get(), (const int *)g->cell_start.get(), (const int *)slots.get(), n, invalid,
                (const int *)too_long, g->sorted.get());
    // Can a cell have outgrown the counting sort?  Not when the cloud is a voxel grid's output and the cell spans few leaves
    // (types.hpp: voxel_leaf), nor when the cloud is small -- every grid of the map pipeline: the answer is known without asking
    // the device, which saves the wait (six per map); the word is still copied and looked at with the next wait, an
    // invariant rather than a case.
    // A caller's raw cloud can hold thousands of points in a cell, and its grid waits for the answer.
    bool bounded = small;
    if (!bounded && cl->voxel_leaf > 0.f) {
      const double per_axis = std::floor((double)g->cell / (double)cl->voxel_leaf) + 6.0;
      bounded = per_axis * per_axis * per_axis <= (double)kCellSortMax;
    }
    int *h_long = (int *)c->pin(64);
    *h_long = 0;
    MM3D_HIP(hipMemcpyAsync(h_long, too_long, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (bounded) c->check_later(h_long, MM3D_EDEVICE, "grid: a cell of a voxel-filtered cloud holds more points than its leaf allows");
    else c->sync();
    if (!bounded && *h_long) {
      // a cell with thousands of points: stable radix sort of (cell, index) instead
      DevBuf<uint32_t> vals(c, n), keys2(c, n), vals2(c, n);
      MM3D_LAUNCH(c, "grid_cell_keys", n * 8.0, k_iota_u32, dim3(div_up(n, 256)), dim3(256), 0, vals.get(), n);
      int bits = 1;
      while (((size_t)1 << bits) <= ncell) ++bits;
      sort_pairs_u32(c, keys.get(), keys2.get(), vals.get(), vals2.get(), n, bits);
      MM3D_LAUNCH(c, "grid_gather", nfin * 36.0, k_gather_sorted, dim3(div_up(nfin, 256)), dim3(256), 0, cl->pts.get(),
                  vals2.get(), (int)nfin, g->sorted.get());
    }
    g->n = (int)nfin;
    c->settle();   // temporaries return to the pool after the stream is done with them
  }
  auto &ref = *g;
  cl->grids[key] = std::move(g);
  return ref;
}

// ---------------------------------------------------------------- Chebyshev distance transform
// dt[c] = distance in cells (max-norm) from cell c to the nearest occupied cell, 255 if > R.
// The max-norm separates exactly: f1 = 1-D distance along x, f2 = min_dy max(f1, |dy|), f3 likewise
// along z.  A 1-NN query reads ONE byte to learn which ring of cells to start at, or that nothing
// lies within range at all (most source points of a non-overlapping map pair).
__global__ void k_dt_x(const int *__restrict__ cell_start, int dx, int dy, int dz, int R, unsigned char *__restrict__ out)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nc = (size_t)dx * dy * dz;
  if (c >= nc) return;
  const int x = (int)(c % dx);
  const size_t row = c - x;
  int best = 255;
  for (int d = 0; d <= R && best == 255; ++d) {
    const int xl = x - d, xr = x + d;
    if (xl >= 0 && cell_start[row + xl + 1] > cell_start[row + xl]) best = d;
    else if (xr < dx && cell_start[row + xr + 1] > cell_start[row + xr]) best = d;
  }
  out[c] = (unsigned char)best;
}

__global__ void k_dt_axis(const unsigned char *__restrict__ in, int dx, int dy, int dz, int axis, int R,
                          unsigned char *__restrict__ out)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nc = (size_t)dx * dy * dz;
  if (c >= nc) return;
  const int y = (int)((c / dx) % dy), z = (int)(c / ((size_t)dx * dy));
  const int pos = axis == 1 ? y : z, lim = axis == 1 ? dy : dz;
  const size_t stride = axis == 1 ? (size_t)dx : (size_t)dx * dy;
  int best = 255;
  for (int d = -R; d <= R; ++d) {
    const int q = pos + d;
    if (q < 0 || q >= lim) continue;
    const int v = in[c + (long long)d * (long long)stride];
    const int a = d < 0 ? -d : d;
    const int m = v > a ? v : a;
    best = m < best ? m : best;
  }
  out[c] = (unsigned char)(best > R ? 255 : best);
}

void grid_ensure_dt(Context *c, const Grid &g_, int R)
{
  Grid &g = const_cast<Grid &>(g_);
  std::lock_guard<std::mutex> lk(g.cache_mu);
  if (R > 250) R = 250;
  if (g.dt.get() && g.dt_cap >= R) return;
  const size_t nc = (size_t)g.dims[0] * g.dims[1] * g.dims[2];
  DevBuf<unsigned char> a(c, nc), b(c, nc);
  const unsigned blocks = div_up(nc, 256);
  MM3D_LAUNCH(c, "grid_dt", nc * 5.0, k_dt_x, dim3(blocks), dim3(256), 0, (const int *)g.cell_start.get(), g.dims[0], g.dims[1],
              g.dims[2], R, a.get());
  MM3D_LAUNCH(c, "grid_dt", nc * 2.0, k_dt_axis, dim3(blocks), dim3(256), 0, (const unsigned char *)a.get(), g.dims[0], g.dims[1],
              g.dims[2], 1, R, b.get());
  MM3D_LAUNCH(c, "grid_dt", nc * 2.0, k_dt_axis, dim3(blocks), dim3(256), 0, (const unsigned char *)b.get(), g.dims[0], g.dims[1],
              g.dims[2], 2, R, a.get());
  g.dt = std::move(a);
  g.dt_cap = R;
  c->settle();      // another context may read the table as soon as the lock is gone
}

// ---------------------------------------------------------------- merged neighbourhood lists
// For a search whose radius is fixed and whose queries are many (SAC-IA scores 500 x K_s transformed
// keypoints against K_t targets), the stencil walk is paid once per CELL instead of once per query:
// every cell gets the concatenation of the points of its (2R+1)^3 block, in stencil order.  A query
// then reads one contiguous span; an empty span means "nothing within R cells".
__global__ void k_nb_count(const int *__restrict__ cell_start, int dx, int dy, int dz, int R, int *__restrict__ counts)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nc = (size_t)dx * dy * dz;
  if (c > nc) return;
  if (c == nc) { counts[c] = 0; return; }
  const int x = (int)(c % dx), y = (int)((c / dx) % dy), z = (int)(c / ((size_t)dx * dy));
  const int x0 = max(x - R, 0), x1 = min(x + R, dx - 1);
  int total = 0;
  for (int zz = max(z - R, 0); zz <= min(z + R, dz - 1); ++zz)
    for (int yy = max(y - R, 0); yy <= min(y + R, dy - 1); ++yy) {
      const int row = (zz * dy + yy) * dx;
      total += cell_start[row + x1 + 1] - cell_start[row + x0];
    }
  counts[c] = total;
}

__global__ void k_nb_fill(const int *__restrict__ cell_start, const float4 *__restrict__ sorted, int dx, int dy, int dz, int R,
                          const int *__restrict__ nb_start, float4 *__restrict__ nb_pts)
{
  // one wave per cell: lanes stride over each row span
  const size_t c = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t nc = (size_t)dx * dy * dz;
  if (c >= nc) return;
  const int lane = threadIdx.x & 63;
  int out = nb_start[c];
  if (nb_start[c + 1] == out) return;
  const int x = (int)(c % dx), y = (int)((c / dx) % dy), z = (int)(c / ((size_t)dx * dy));
  const int x0 = max(x - R, 0), x1 = min(x + R, dx - 1);
  for (int zz = max(z - R, 0); zz <= min(z + R, dz - 1); ++zz)
    for (int yy = max(y - R, 0); yy <= min(y + R, dy - 1); ++yy) {
      const int row = (zz * dy + yy) * dx;
      const int b = cell_start[row + x0], e = cell_start[row + x1 + 1];
      for (int j = b + lane; j < e; j += 64) {
        const float4 p = sorted[j];
        nb_pts[(size_t)(out + (j - b))] = make_float4(p.x, p.y, p.z, 0.0f);
      }
      out += e - b;
    }
}

// Lists of up to kNbSortCap entries: .w = distance from the cell centre, entries reordered to ascend in it
// (rank by counting in LDS; ties by stencil position, so the order is a function of the input alone).
constexpr int kNbSortCap = 256;       // 16 KB of LDS per block: a launch that asks for more waits for the neighbourhood kernels of other streams
__global__ void __launch_bounds__(256)
k_nb_sort(float minx, float miny, float minz, float cell, int dx, int dy, int dz, const int *__restrict__ nb_start,
          float4 *__restrict__ nb_pts)
{
  __shared__ float4 ent[4][kNbSortCap];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t c = (size_t)blockIdx.x * 4 + w;
  const size_t nc = (size_t)dx * dy * dz;
  if (c >= nc) return;
  const int b = nb_start[c], n = nb_start[c + 1] - b;
  if (n == 0) return;
  if (n > kNbSortCap) return;
  const int x = (int)(c % dx), y = (int)((c / dx) % dy), z = (int)(c / ((size_t)dx * dy));
  const float ox = minx + ((float)x + 0.5f) * cell, oy = miny + ((float)y + 0.5f) * cell, oz = minz + ((float)z + 0.5f) * cell;
  for (int j = lane; j < n; j += 64) {
    float4 p = nb_pts[(size_t)b + j];
    const float ax = p.x - ox, ay = p.y - oy, az = p.z - oz;
    p.w = sqrtf(ax * ax + ay * ay + az * az);
    ent[w][j] = p;
  }
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < n; j += 64) {
    const float4 p = ent[w][j];
    int rank = 0;
    for (int k = 0; k < n; ++k) {
      const float o = ent[w][k].w;
      rank += (o < p.w || (o == p.w && k < j)) ? 1 : 0;
    }
    nb_pts[(size_t)b + rank] = p;
  }
}

void grid_ensure_nblists(Context *c, const Grid &g_, int R)
{
  Grid &g = const_cast<Grid &>(g_);
  std::lock_guard<std::mutex> lk(g.cache_mu);
  if (g.nb_start.get() && g.nb_R == R) return;
  const size_t nc = (size_t)g.dims[0] * g.dims[1] * g.dims[2];
  DevBuf<int> counts(c, nc + 1);
  MM3D_LAUNCH(c, "grid_nblists", nc * 8.0, k_nb_count, dim3(div_up(nc + 1, 256)), dim3(256), 0, (const int *)g.cell_start.get(),
              g.dims[0], g.dims[1], g.dims[2], R, counts.get());
  g.nb_start = DevBuf<int>(c, nc + 1);
  exclusive_scan_int(c, counts.get(), g.nb_start.get(), nc + 1);
  int *h = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(h, g.nb_start.get() + nc, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  const size_t total = (size_t)h[0];
  g.nb_pts = DevBuf<float4>(c, total ? total : 1);
  if (total) {
    MM3D_LAUNCH(c, "grid_nblists", total * 32.0, k_nb_fill, dim3(div_up(nc, 4)), dim3(256), 0, (const int *)g.cell_start.get(),
                (const float4 *)g.sorted.get(), g.dims[0], g.dims[1], g.dims[2], R, (const int *)g.nb_start.get(), g.nb_pts.get());
    MM3D_LAUNCH(c, "grid_nblists", total * 32.0, k_nb_sort, dim3(div_up(nc, 4)), dim3(256), 0, g.mn[0], g.mn[1], g.mn[2], g.cell,
                g.dims[0], g.dims[1], g.dims[2], (const int *)g.nb_start.get(), g.nb_pts.get());
  }
  g.nb_R = R;
  c->settle();
}

// ---------------------------------------------------------------- Hilbert order + wave work items
// Queries that run 64 to a wave should form a compact patch, so that the cells they can reach are a
// small box (wave_stream_box).  Points are ordered by the Hilbert index of their (x, y) column with
// z below it; a work item is a run of at most 64 consecutive points that never straddles a
// 2 m x 2 m column block (8 x 8 Hilbert cells): the curve may leave the occupied area and re-enter
// far away, but never inside one block.
// (counts / ranks != null: the counting sort's first step -- a point's arrival rank in its Hilbert column -- in the same launch)
__global__ void k_hilbert_keys(const float4 *__restrict__ pts, int n, float minx, float miny, float minz, float inv,
                               uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, int *__restrict__ counts, uint32_t *__restrict__ ranks)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  vals[i] = (uint32_t)i;
  if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { keys[i] = 0xFFFFFFFFu; return; }
  unsigned x = (unsigned)clampi(cell_floor(p.x, minx, inv), 0, 1023);
  unsigned y = (unsigned)clampi(cell_floor(p.y, miny, inv), 0, 1023);
  const unsigned z = (unsigned)clampi(cell_floor(p.z, minz, inv), 0, 1023);
  unsigned d = 0;
  for (unsigned sft = 512; sft > 0; sft >>= 1) {
    const unsigned rx = (x & sft) ? 1u : 0u, ry = (y & sft) ? 1u : 0u;
    d += sft * sft * ((3u * rx) ^ ry);
    if (ry == 0) {
      if (rx == 1) { x = 1023u - x; y = 1023u - y; }
      const unsigned t = x; x = y; y = t;
    }
  }
  const uint32_t key = (d << 10) | z;
  keys[i] = key;
  if (counts) ranks[i] = (uint32_t)atomicAdd(&counts[key >> 10], 1);
}

__global__ void k_hilbert_gather(const float4 *__restrict__ pts, const uint32_t *__restrict__ order, int n, float4 *__restrict__ out)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t i = order[j];
  float4 p = pts[i];
  p.w = __int_as_float((int)i);
  out[j] = p;
}

// a work item starts at the first point, at every change of the column block (key >> 16) and every 64 points.
// `inert` (device word or null): when it is set the keys were never written (the counting sort gave up, k_hil_place
// returned early) -- no head is seen and no item is stored, so the scan cannot write past the items' bound n / 64 + 16386
// on recycled pool memory that shows a head at every j
struct ItemHeadLoad {
  const uint32_t *keys; int n; const int *inert;
  __device__ __forceinline__ bool head(size_t j) const { return j < (size_t)n && (j == 0 || (keys[j] >> 16) != (keys[j - 1] >> 16) || (j & 63) == 0); }
  __device__ __forceinline__ int operator()(size_t j) const { return (inert && *inert) ? 0 : (head(j) ? 1 : 0); }
};
struct ItemFillStore {
  const uint32_t *keys; int n; int2 *items; int *n_items; const int *inert;
  __device__ __forceinline__ void operator()(size_t j, int prefix, int v) const
  {
    if (j == (size_t)n) { *n_items = prefix; return; }        // (the scan runs over n + 1 elements: the last one carries the total)
    if (!v) return;
    const ItemHeadLoad h{keys, n, nullptr};
    int cnt = 1;
    while (cnt < 64 && j + cnt < (size_t)n && !h.head(j + cnt)) ++cnt;
    items[prefix] = make_int2((int)j, cnt);
  }
  __device__ __forceinline__ void done() const {}
};

// ---------------------------------------------------------------- counting sort of (key, index) pairs
// Stable sort of sparse 32-bit keys whose equal-or-near values are few (voxel indices, Hilbert keys): bin = key /
// divisor with at most 2^22 bins; a point's arrival rank in its bin comes with the histogram (one atomic), a scan
// gives the bins' starts, and inside its bin (tens of points) every point counts the smaller (key, index) pairs.
// Four small launches; the result is the order rocPRIM's stable radix sort of the keys gives.  A bin longer than
// kCountSortBinMax sets *too_long and nothing is written: the caller falls back to the radix sort.
constexpr int kCountSortBinMax = 1024;
__global__ void k_cs_count(const uint32_t *__restrict__ keys, int n, uint32_t divisor, int *__restrict__ counts, uint32_t *__restrict__ ranks)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t key = keys[i];
  if (key == 0xFFFFFFFFu) return;               // non-finite point: stays out (the output's tail keeps its fill value)
  ranks[i] = (uint32_t)atomicAdd(&counts[key / divisor], 1);
}
__global__ void k_cs_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ ranks, const int *__restrict__ bin_start, int n,
                             uint32_t divisor, int *__restrict__ slots, int *__restrict__ too_long)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t key = keys[i];
  if (key == 0xFFFFFFFFu) return;
  const uint32_t bin = key / divisor;
  const int b = bin_start[bin];
  slots[b + (int)ranks[i]] = i;
  if (bin_start[bin + 1] - b > kCountSortBinMax) *too_long = 1;
}
__global__ void k_cs_place(const uint32_t *__restrict__ keys, const int *__restrict__ bin_start, const int *__restrict__ slots, int n,
                           uint32_t divisor, const int *__restrict__ too_long, uint32_t *__restrict__ keys_out, uint32_t *__restrict__ idx_out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || *too_long) return;
  const uint32_t key = keys[i];
  if (key == 0xFFFFFFFFu) return;
  const uint32_t bin = key / divisor;
  const int b = bin_start[bin], e = bin_start[bin + 1];
  int r = 0;
  for (int j = b; j < e; ++j) {
    const int o = slots[j];
    const uint32_t ko = keys[o];
    r += (ko < key || (ko == key && o < i)) ? 1 : 0;
  }
  keys_out[b + r] = key;
  idx_out[b + r] = (uint32_t)i;
}

// keys_out / idx_out: n entries; entries beyond the finite keys read 0xFFFFFFFF / are unspecified.  too_long (device)
// must be checked by the caller after its next synchronisation.
// no_invalid_keys: the caller knows that no key is 0xFFFFFFFF (every point finite) -- every output entry is written and the fill of
// keys_out (n words: the largest of a map's fills) is left out.
void counting_sort_pairs_u32(Context *c, const uint32_t *keys, int n, uint64_t key_range, uint32_t *keys_out, uint32_t *idx_out,
                             int *too_long, bool no_invalid_keys)
{
  uint32_t divisor = (uint32_t)((key_range + ((uint64_t)1 << 22) - 1) >> 22);
  if (divisor == 0) divisor = 1;
  const int nbins = (int)(key_range / divisor) + 1;
  DevBuf<int> counts(c, (size_t)nbins + 1), bin_start(c, (size_t)nbins + 1), slots(c, (size_t)n);
  DevBuf<uint32_t> ranks(c, (size_t)n);
  MM3D_HIP(hipMemsetAsync(counts.get(), 0, ((size_t)nbins + 1) * sizeof(int), c->stream));
  MM3D_HIP(hipMemsetAsync(too_long, 0, sizeof(int), c->stream));
  if (!no_invalid_keys) MM3D_HIP(hipMemsetAsync(keys_out, 0xFF, (size_t)n * sizeof(uint32_t), c->stream));
  MM3D_LAUNCH(c, "count_sort", n * 12.0, k_cs_count, dim3(div_up(n, 256)), dim3(256), 0, keys, n, divisor, counts.get(), ranks.get());
  exclusive_scan_int(c, counts.get(), bin_start.get(), (size_t)nbins + 1);
  MM3D_LAUNCH(c, "count_sort", n * 16.0, k_cs_scatter, dim3(div_up(n, 256)), dim3(256), 0, keys, (const uint32_t *)ranks.get(),
              (const int *)bin_start.get(), n, divisor, slots.get(), too_long);
  MM3D_LAUNCH(c, "count_sort", n * 24.0, k_cs_place, dim3(div_up(n, 256)), dim3(256), 0, keys, (const int *)bin_start.get(),
              (const int *)slots.get(), n, divisor, (const int *)too_long, keys_out, idx_out);
  // (the temporaries return to the pool when this function ends: the pool is stream-ordered per context, and the
  // next user of these blocks runs on the same stream)
}

// Hilbert order by counting sort on the column index (key >> 10, at most 2^20 columns): a point's arrival rank in
// its column comes with the histogram (k_hilbert_keys computes both), a scan gives the columns' starts, and inside its column (a handful of
// points; a wall: a hundred) every point counts the smaller (key, index) pairs -- the order a stable radix sort
// of the keys gives, in four small launches instead of rocPRIM's seven.
constexpr int kHilColumns = 1 << 20;
constexpr int kHilColumnMax = 1024;      // longer columns (degenerate clouds) leave it to the radix sort
__global__ void k_hil_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ ranks, const int *__restrict__ col_start, int n,
                              int *__restrict__ slots, int *__restrict__ too_long)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t key = keys[i];
  if (key == 0xFFFFFFFFu) return;
  const int b = col_start[key >> 10];
  slots[b + (int)ranks[i]] = i;
  if (col_start[(key >> 10) + 1] - b > kHilColumnMax) *too_long = 1;
}
__global__ void k_hil_place(const float4 *__restrict__ pts, const uint32_t *__restrict__ keys, const int *__restrict__ col_start,
                            const int *__restrict__ slots, int n, const int *__restrict__ too_long, float4 *__restrict__ out,
                            uint32_t *__restrict__ out_keys)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || *too_long) return;
  const uint32_t key = keys[i];
  if (key == 0xFFFFFFFFu) return;
  const int b = col_start[key >> 10], e = col_start[(key >> 10) + 1];
  int r = 0;
  for (int j = b; j < e; ++j) {
    const int o = slots[j];
    const uint32_t ko = keys[o];
    r += (ko < key || (ko == key && o < i)) ? 1 : 0;
  }
  float4 p = pts[i];
  p.w = __int_as_float(i);
  out[b + r] = p;
  out_keys[b + r] = key;
}

void cloud_hilbert(Context *c, const mm3d_cloud *cl_, float min_cell)
{
  auto *cl = const_cast<mm3d_cloud *>(cl_);
  std::lock_guard<std::recursive_mutex> lk(cl->cache_mu);
  cloud_bbox(c, cl);
  const int n = (int)cl->n_finite;
  if (cl->hil_pts.get() || n == 0) return;
  const int total = (int)cl->n;
  // cell so that the larger box side spans at most 1024 cells, but never finer than 0.25 m
  float ext = 0.f;
  for (int a = 0; a < 3; ++a) ext = std::fmax(ext, cl->bmax[a] - cl->bmin[a]);
  const float cell = std::fmax(std::fmax(0.25f, min_cell), ext / 1023.0f);
  DevBuf<uint32_t> keys(c, total), vals(c, total), keys2(c, total);
  cl->hil_pts = DevBuf<float4>(c, n);
  DevBuf<int> counts(c, kHilColumns + 2), col_start(c, kHilColumns + 1), slots(c, n);
  DevBuf<uint32_t> ranks(c, total);
  MM3D_HIP(hipMemsetAsync(counts.get(), 0, (kHilColumns + 2) * sizeof(int), c->stream));   // (counts and the too_long word: one fill)
  const DevPtr<int> too_long{counts.get() + kHilColumns + 1};
  MM3D_LAUNCH(c, "hilbert_keys", total * 36.0, k_hilbert_keys, dim3(div_up(total, 256)), dim3(256), 0, cl->pts.get(), total,
              cl->bmin[0], cl->bmin[1], cl->bmin[2], 1.0f / cell, keys.get(), vals.get(), counts.get(), ranks.get());
  exclusive_scan_int(c, counts.get(), col_start.get(), kHilColumns + 1);
  MM3D_LAUNCH(c, "hilbert_sort", total * 16.0, k_hil_scatter, dim3(div_up(total, 256)), dim3(256), 0, (const uint32_t *)keys.get(),
              (const uint32_t *)ranks.get(), (const int *)col_start.get(), total, slots.get(), too_long.get());
  MM3D_LAUNCH(c, "hilbert_sort", total * 44.0, k_hil_place, dim3(div_up(total, 256)), dim3(256), 0, cl->pts.get(),
              (const uint32_t *)keys.get(), (const int *)col_start.get(), (const int *)slots.get(), total, (const int *)too_long.get(),
              cl->hil_pts.get(), keys2.get());
  // Work items: flag the heads, number them, write {first point, count} per head -- one launch (scan_fused.hpp) instead of
  // three.  The item array is sized by its bound (a head every 64 points plus one per column block: key >> 16 < 2^14); the
  // count comes back with the sort's too_long word in the one host wait of this function.
  const size_t max_items = (size_t)n / 64 + 16384 + 2;
  cl->wave_items = DevBuf<int2>(c, max_items);
  DevBuf<int> n_items_dev(c, 1);
  int *h = (int *)c->pin(64);
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int *inert = attempt == 0 ? (const int *)too_long.get() : nullptr;      // (attempt 1: the radix sort always writes the keys)
    scan_fused(c, "hilbert_items", n * 20.0, (size_t)n + 1, ItemHeadLoad{keys2.get(), n, inert},
               ItemFillStore{keys2.get(), n, cl->wave_items.get(), n_items_dev.get(), inert});
    MM3D_HIP(hipMemcpyAsync(h, n_items_dev.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
    MM3D_HIP(hipMemcpyAsync(h + 1, too_long.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (attempt == 1 || !h[1]) break;
    // a column with more than a thousand points (the counting sort gave up): stable radix sort instead, then once more
    DevBuf<uint32_t> vals2(c, total);
    sort_pairs_u32(c, keys.get(), keys2.get(), vals.get(), vals2.get(), total, 32);
    MM3D_LAUNCH(c, "hilbert_gather", n * 36.0, k_hilbert_gather, dim3(div_up(n, 256)), dim3(256), 0, cl->pts.get(),
                (const uint32_t *)vals2.get(), n, cl->hil_pts.get());
    c->settle();                                    // vals2 goes back to the pool after the gather
  }
  cl->n_wave_items = h[0];
  cl->hil_keys = std::move(keys2);
  c->settle();
}

// ---------------------------------------------------------------- ordered compaction
// Ordered compaction, the count and the bounding box of what is kept in ONE launch and one wait (scan_fused.hpp): the output is
// sized by its bound n, the kept points -- all finite: the callers' flags are only set for finite points -- are scattered by
// the scan's own store, and their box rides along (box_out: 7 words as cloud_bbox reads them, or null).
struct CompactLoad {
  const int *flags; size_t n;
  __device__ __forceinline__ int operator()(size_t j) const { return j < n ? (flags[j] ? 1 : 0) : 0; }
};
struct CompactStore {
  const float4 *in; size_t n; float4 *out; int *total; unsigned *box;
  BoxAcc acc;
  __device__ __forceinline__ void operator()(size_t j, int prefix, int v)
  {
    if (j == n) { *total = prefix; return; }
    if (!v) return;
    const float4 p = in[j];
    out[prefix] = p;
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) acc.add(p);     // (min_neighbours < 0 keeps every point, finite or not)
  }
  __device__ __forceinline__ void done() { acc.flush(box); }
};

size_t compact_points(Context *c, const float4 *in, const int *flags, size_t n, DevBuf<float4> &out, unsigned *box_host)
{
  if (n == 0) { out = DevBuf<float4>(c, 0); return 0; }
  out = DevBuf<float4>(c, n);
  DevBuf<unsigned> ctl(c, 16);                       // [0] kept count, [8..14] box
  unsigned *h = (unsigned *)c->pin(64);
  std::memset(h, 0, 64);
  std::memcpy(h + 8, kBoxInit, sizeof(kBoxInit));
  MM3D_HIP(hipMemcpyAsync(ctl.get(), h, 64, hipMemcpyHostToDevice, c->stream));
  scan_fused(c, "compact", n * 24.0, n + 1, CompactLoad{flags, n}, CompactStore{in, n, out.get(), (int *)ctl.get(), ctl.get() + 8, BoxAcc()});
  unsigned *r = (unsigned *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(r, ctl.get(), 64, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  if (box_host) std::memcpy(box_host, r + 8, 7 * sizeof(unsigned));
  return (size_t)r[0];
}

// a cloud whose producer reduced its bounding box (scan_fused.hpp::BoxAcc): no k_bbox launch, no wait
void cloud_set_bbox(mm3d_cloud *cl, const unsigned box[7])
{
  std::lock_guard<std::recursive_mutex> lk(cl->cache_mu);
  cl->have_bbox = true;
  cl->n_finite = box[6];
  if (cl->n_finite)
    for (int a = 0; a < 3; ++a) { cl->bmin[a] = ord2f(box[a]); cl->bmax[a] = ord2f(box[3 + a]); }
}

// compute units of a device (the persistent grids of snb_lds.hpp are sized from it)
int snb_cu_count(int device)
{
  static std::mutex mu;
  static std::map<int, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(device);
  if (it != cache.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
  cache[device] = n;
  return n;
}

}  // namespace mm3d
