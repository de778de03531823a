// device_util.hpp -- device-side helpers shared by the gfx950 kernels.
//
// Arithmetic policy: the library is compiled with -ffp-contract=off, so a*b+c is two roundings
// exactly like the CPU path it must agree with; kernels that are not bit-critical say fmaf()
// explicitly.  Wave = 64 lanes (CDNA4); nothing here assumes 32.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "libm_exact.hpp"
#include "types.hpp"

namespace mm3d {

constexpr int kWave = 64;
constexpr int kXcds = 8;   // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only)

// XCD-aware block remap: give every XCD one contiguous slice of the (spatially sorted) work so
// that each private 4 MiB L2 caches one region of the grid instead of all of it.  Bijective for
// any grid size (cdna_hip_programming.md T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks)
{
  unsigned q = nblocks / kXcds, r = nblocks % kXcds;
  unsigned xcd = bid % kXcds, k = bid / kXcds;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

__device__ __forceinline__ unsigned f2ord(float f)
{
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(unsigned u)
{
  unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __uint_as_float(v);
#else
  memcpy(&f, &v, 4);
#endif
  return f;
}

// FLANN L2_Simple order: ((dx*dx + dy*dy) + dz*dz), no contraction
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz)
{
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// pcl::transformPointCloud order: ((m0*x + m1*y) + m2*z) + m3, per row; T column-major
__device__ __forceinline__ float3 xform(const float *T, float x, float y, float z)
{
  float3 r;
  r.x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], x), __fmul_rn(T[4], y)), __fmul_rn(T[8], z)), T[12]);
  r.y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[1], x), __fmul_rn(T[5], y)), __fmul_rn(T[9], z)), T[13]);
  r.z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[2], x), __fmul_rn(T[6], y)), __fmul_rn(T[10], z)), T[14]);
  return r;
}

__device__ __forceinline__ int cell_floor(float v, float mn, float inv)
{
  float f = floorf((v - mn) * inv);
  f = fminf(fmaxf(f, -1048576.0f), 1048576.0f);
  return (int)f;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Visit every point of the grid that can lie within `r` of q: (2s+1)^2 contiguous row spans.
// f(const float4 &p) -> bool; returning false stops the walk early.
template <class F>
__device__ __forceinline__ void for_each_candidate(const GridView &g, float qx, float qy, float qz, float r, F &&f)
{
  const float ri = r * 1.0001f + 1e-4f;   // keeps the cell range conservative under float rounding
  int x0 = clampi(cell_floor(qx - ri, g.minx, g.inv), 0, g.dx - 1);
  int x1 = clampi(cell_floor(qx + ri, g.minx, g.inv), 0, g.dx - 1);
  int y0 = cell_floor(qy - ri, g.miny, g.inv), y1 = cell_floor(qy + ri, g.miny, g.inv);
  int z0 = cell_floor(qz - ri, g.minz, g.inv), z1 = cell_floor(qz + ri, g.minz, g.inv);
  if (cell_floor(qx + ri, g.minx, g.inv) < 0 || cell_floor(qx - ri, g.minx, g.inv) > g.dx - 1) return;
  y0 = y0 < 0 ? 0 : y0; z0 = z0 < 0 ? 0 : z0;
  y1 = y1 > g.dy - 1 ? g.dy - 1 : y1; z1 = z1 > g.dz - 1 ? g.dz - 1 : z1;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      const int row = (z * g.dy + y) * g.dx;
      const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
      for (int j = b; j < e; ++j)
        if (!f(g.pts[j])) return;
    }
}

// "v = 0; repeat hits times: v += incr" in float, bit for bit, without walking the whole chain.  Inside one
// binade of v every addition but possibly the first moves v by the same multiple of its ulp (round-to-nearest
// of v + incr = v + q ulp + r: r </> ulp/2 always rounds the same way, and a tie r = ulp/2 alternates at most
// once before the parity of v / ulp settles).  So after two equal consecutive steps inside one binade the
// remaining steps of that binade are taken at once (exactly: everything is a small multiple of the ulp, done
// in double), and only the few additions around each power of two are really executed.
__device__ __forceinline__ float float_chain_sum(float incr, unsigned hits)
{
  float v = 0.0f;
  unsigned h = hits;
  if (!(incr > 0.0f) || !(incr < INFINITY)) {           // not a regular chain: just run it
    for (; h > 0; --h) v += incr;
    return v;
  }
  while (h > 0) {
    if (h < 8) { v += incr; --h; continue; }
    const float v1 = v + incr, v2 = v1 + incr, v3 = v2 + incr;
    h -= 3;
    v = v3;
    const unsigned e1 = __float_as_uint(v1) >> 23, e3 = __float_as_uint(v3) >> 23;
    const float s = v3 - v2;
    if (s == 0.0f && v2 == v1) return v3;                // incr has fallen below half an ulp of v: the sum has saturated
    if (e1 != e3 || (v2 - v1) != s || !(s > 0.0f)) continue;
    // v1, v2, v3 share a binade and two equal steps s: all further steps below the next power of two equal s
    const float top = __uint_as_float((e3 + 1u) << 23);
    const double room = ((double)top - (double)v3) / (double)s;
    if (room > 2.0) {
      double n = floor(room) - 1.0;                      // stay strictly inside the binade
      if (n > (double)h) n = (double)h;
      v = (float)((double)v3 + n * (double)s);
      h -= (unsigned)n;
    }
  }
  return v;
}

// acos(fabs(a1)) > acos(fabs(a2)) as the CPU path evaluates it (pcl::computePairFeatures: the arguments are floats, the
// arc cosines glibc's double ones).  acos is decreasing, so this is |a1| < |a2| -- except where it is not: an argument
// above 1 (a rounded-up cosine) gives NaN and the comparison is false; and below 2^-28 two different floats can have
// the same double arc cosine (acos(x) = RN(pi/2 - x) there, and the double nearest pi/2 has an ulp of 2.2e-16), in which
// case the CPU path sees a tie.  Checked against this image's libm on 14 M pairs incl. adjacent floats down to 2^-70.
__device__ __forceinline__ bool acos_abs_greater(float a1, float a2)
{
  const float x1 = fabsf(a1), x2 = fabsf(a2);
  if (!(x1 <= 1.0f) || !(x2 <= 1.0f)) return false;
  if (x1 >= 0x1p-28f || x2 >= 0x1p-28f) return x1 < x2;
  const double hi = 0x1.921fb54442d18p+0, lo = 0x1.1a62633145c07p-54;   // pi/2 = hi + lo
  return (hi + (lo - (double)x1)) > (hi + (lo - (double)x2));
}

// Lane mask of a predicate.  HIP's ballot() takes an int: a bool argument is first materialised as 0 / 1 (v_cndmask) and
// compared with zero again (v_cmp_ne) -- two VALU instructions per call in the inner loops that vote on every step; the
// builtin takes the i1 as it is (the comparison's own lane mask).
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ---- wave / block reductions -------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  return v;
}

__device__ __forceinline__ int wave_min_int(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ int wave_max_int(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, kWave));
  return v;
}

__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  return v;
}

// LDS written by some lanes of a wave and read by others of the SAME wave: order the accesses for
// the compiler; the hardware executes one wave's DS operations in order.
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- wave-cooperative neighbourhood staging --------------------------------------------------------
// A wave whose 64 queries form a compact patch streams the box of grid cells those queries can
// reach through its private LDS slab: row headers one per lane, exclusive scan of the span lengths,
// then TILE points at a time with coalesced 16-byte gathers (all of a lane's loads are issued before
// its first LDS store, so a tile costs one memory round trip).  `scan(cnt)` is called with the tile
// in s_pts[0..cnt) (and NX extra float4 per point in s_x[e*TILE + slot], produced by load_x from the
// sorted index): every lane then reads the SAME LDS address, i.e. broadcast reads, no global
// traffic.  Must be called by all 64 lanes with wave-uniform box arguments.
// `keep(point)` drops candidates while they are staged (ballot + prefix compaction, relative order
// preserved): the box is made of whole cells, a fixed-radius search only needs the points inside
// the patch's bounding box grown by the radius -- about a quarter fewer candidates for every lane.
struct KeepAll {
  __device__ __forceinline__ bool operator()(const float4 &) const { return true; }
};
struct KeepInBox {
  float lx, hx, ly, hy, lz, hz;
  __device__ __forceinline__ bool operator()(const float4 &p) const
  {
    return p.x >= lx && p.x <= hx && p.y >= ly && p.y <= hy && p.z >= lz && p.z <= hz;
  }
};
// ... or, tighter: inside the box [l, h] grown by r with its edges and corners ROUNDED -- the points within r of the box.  A
// candidate within r of a query inside [l, h] is within r of the box, so nothing is lost; what goes is the grown box's corner
// regions (a tenth of it in the plane, more in space), which every query of the item would otherwise test and reject.
struct KeepNearBox {
  float lx, hx, ly, hy, lz, hz, r2;
  __device__ __forceinline__ bool operator()(const float4 &p) const
  {
    const float dx = fmaxf(fmaxf(lx - p.x, p.x - hx), 0.0f), dy = fmaxf(fmaxf(ly - p.y, p.y - hy), 0.0f), dz = fmaxf(fmaxf(lz - p.z, p.z - hz), 0.0f);
    return dx * dx + dy * dy + dz * dz <= r2;
  }
};
template <int TILE, int NX, class LoadX, class Scan, class Keep = KeepAll>
__device__ __forceinline__ void wave_stream_box(const GridView &g, int x0, int x1, int y0, int y1, int z0, int z1,
                                                float4 *s_pts, float4 *s_x, int *s_off, int *s_beg, int lane,
                                                LoadX &&load_x, Scan &&scan, Keep keep = Keep())
{
  constexpr int PER = TILE / kWave;
  constexpr int NXA = NX > 0 ? NX : 1;
  const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
  const int nrows = (x0 <= x1 && ny > 0 && nz > 0) ? ny * nz : 0;
  for (int r0 = 0; r0 < nrows; r0 += kWave) {
    const int r = r0 + lane;
    int b = 0, len = 0;
    if (r < nrows) {
      const int z = z0 + r / ny, y = y0 + r % ny;
      const int row = (z * g.dy + y) * g.dx;
      b = g.cell_start[row + x0];
      len = g.cell_start[row + x1 + 1] - b;
    }
    int incl = len;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int t = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += t;
    }
    const int total = __shfl(incl, kWave - 1, kWave);
    wave_lds_fence();                 // readers of the previous tile / offsets are done
    s_off[lane] = incl - len;
    s_beg[lane] = b;
    wave_lds_fence();
    for (int t0 = 0; t0 < total; t0 += TILE) {
      const int cnt = min(TILE, total - t0);
      float4 st[PER];
      float4 sx[PER][NXA];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int s = lane + u * kWave;
        const int slot = t0 + (s < cnt ? s : 0);
        int lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
          if (s_off[lo + step] <= slot) lo += step;   // offsets are non-decreasing; empty rows collapse
        const int j = s_beg[lo] + (slot - s_off[lo]);
        st[u] = g.pts[j];
        if (NX > 0) load_x(j, sx[u]);
      }
      int kept = 0;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int s = lane + u * kWave;
        const bool k = s < cnt && keep(st[u]);
        const unsigned long long m = ballot(k);
        if (k) {
          const int d = kept + __popcll(m & ((1ull << lane) - 1ull));
          s_pts[d] = st[u];
#pragma unroll
          for (int e = 0; e < NX; ++e) s_x[e * TILE + d] = sx[u][e];
        }
        kept += __popcll(m);
      }
      wave_lds_fence();
      // a scan that also takes a bool is told whether this tile IS the whole box (one header round, one
      // tile): a caller that needs a second look at the same candidates can then take it from LDS
      if constexpr (std::is_invocable_v<Scan, int, bool>) scan(kept, nrows <= kWave && total <= TILE);
      else scan(kept);
      wave_lds_fence();
    }
  }
}

// Hit compaction for a staged tile.  The cheap part (is candidate k within r2 of MY query?) runs
// for all candidates with broadcast LDS reads and leaves a per-lane bitset; the expensive part then
// runs only over each lane's own set bits, in increasing k (the staged order), so the wave's trip
// count is the maximum hit count over its lanes instead of the tile size.
template <int TILE>
__device__ __forceinline__ void tile_hit_mask(const float4 *sp, int cnt, float qx, float qy, float qz, float r2, bool live,
                                              unsigned (&m)[TILE / 32])
{
#pragma unroll
  for (int gq = 0; gq < TILE / 32; ++gq) {
    unsigned bits = 0;
    if (gq * 32 < cnt) {                      // wave-uniform
#pragma unroll 4
      for (int b = 0; b < 32; ++b) {
        const float4 p = sp[gq * 32 + b];     // slots >= cnt hold stale points: masked off below
        bits |= (dist2(qx, qy, qz, p.x, p.y, p.z) < r2) ? (1u << b) : 0u;
      }
      const int rem = cnt - gq * 32;
      if (rem < 32) bits &= (1u << rem) - 1u;
    }
    m[gq] = live ? bits : 0u;
  }
}

template <int G, class F>
__device__ __forceinline__ void for_each_hit(unsigned (&m)[G], F &&f)
{
  for (;;) {
    int k = -1;
#pragma unroll
    for (int gq = 0; gq < G; ++gq)
      if (k < 0 && m[gq]) {
        k = gq * 32 + __ffs((int)m[gq]) - 1;
        m[gq] &= m[gq] - 1u;
      }
    if (k < 0) break;
    f(k);
  }
}

// ---- pcl::eigen33 (common/impl/eigen.hpp) ------------------------------------------------------
__host__ __device__ inline void compute_roots2(float b, float c, float *roots)
{
  roots[0] = 0.0f;
  float d = (float)((double)(b * b) - 4.0 * (double)c);
  if (d < 0.0f) d = 0.0f;
  float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

__host__ __device__ inline void compute_roots(const float *m /* row-major symmetric 3x3 */, float *roots)
{
  float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] -
             m[8] * m[1] * m[1];
  float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  float c2 = m[0] + m[4] + m[8];
  if (fabsf(c0) < 1.1920929e-07f) {
    compute_roots2(c2, c1, roots);
  } else {
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = 1.7320508f;
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    float rho = sqrtf(-a_over_3);
    // glibc's atan2f / cosf / sinf on both sides: the host calls its libm, the device the restatement of
    // exactly those functions (libm_exact.hpp), so the roots -- and the normals -- agree bit for bit
#if defined(__HIP_DEVICE_COMPILE__)
    float theta = lm::atan2f_glibc(sqrtf(-q), half_b) * s_inv3;
    float cos_theta = lm::cosf_glibc(theta);
    float sin_theta = lm::sinf_glibc(theta);
#else
    float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
    float cos_theta = cosf(theta);
    float sin_theta = sinf(theta);
#endif
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    if (roots[1] >= roots[2]) {
      t = roots[1]; roots[1] = roots[2]; roots[2] = t;
      if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    }
    if (roots[0] <= 0.0f) compute_roots2(c2, c1, roots);
  }
}

// smallest eigenpair of a symmetric 3x3 given as {xx,xy,xz,yy,yz,zz}
__host__ __device__ inline void eigen33_smallest(float xx, float xy, float xz, float yy, float yz, float zz,
                                                 float *eigenvalue, float *v)
{
  float scale = fmaxf(fmaxf(fmaxf(fabsf(xx), fabsf(xy)), fmaxf(fabsf(xz), fabsf(yy))), fmaxf(fabsf(yz), fabsf(zz)));
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float m[9] = {xx / scale, xy / scale, xz / scale, xy / scale, yy / scale, yz / scale,
                xz / scale, yz / scale, zz / scale};
  float roots[3];
  compute_roots(m, roots);
  *eigenvalue = roots[0] * scale;
  m[0] -= roots[0]; m[4] -= roots[0]; m[8] -= roots[0];
  float v1[3] = {m[1] * m[5] - m[2] * m[4], m[2] * m[3] - m[0] * m[5], m[0] * m[4] - m[1] * m[3]};
  float v2[3] = {m[1] * m[8] - m[2] * m[7], m[2] * m[6] - m[0] * m[8], m[0] * m[7] - m[1] * m[6]};
  float v3[3] = {m[4] * m[8] - m[5] * m[7], m[5] * m[6] - m[3] * m[8], m[3] * m[7] - m[4] * m[6]};
  float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const bool c1 = (l1 >= l2 && l1 >= l3);
  const bool c2 = !c1 && (l2 >= l1 && l2 >= l3);
  const float sx = c1 ? v1[0] : (c2 ? v2[0] : v3[0]);
  const float sy = c1 ? v1[1] : (c2 ? v2[1] : v3[1]);
  const float sz = c1 ? v1[2] : (c2 ? v2[2] : v3[2]);
  const float n = sqrtf(c1 ? l1 : (c2 ? l2 : l3));
  v[0] = sx / n; v[1] = sy / n; v[2] = sz / n;
}

// computePointNormal + flipNormalTowardsViewpoint(point, 0, 0, 0) (features/normal_3d.h) from the nine raw-moment sums
// a = {xx, xy, xz, yy, yz, zz, x, y, z} over cnt neighbours: covariance E[x x^T] - E[x] E[x]^T, smallest eigenpair,
// curvature |lambda0 / trace|; fewer than 3 neighbours: NaN.  One source for every normals kernel (normals.hip, sift.hip).
__device__ __forceinline__ float4 normal_from_moments(const float *sums, int cnt, const float4 &pq)
{
  float4 o;
  if (cnt < 3) {
    o.x = o.y = o.z = o.w = __uint_as_float(0x7fc00000u);
    return o;
  }
  float a[9];
  const float fc = (float)cnt;
#pragma unroll
  for (int k = 0; k < 9; ++k) a[k] = sums[k] / fc;
  const float cxx = a[0] - a[6] * a[6], cxy = a[1] - a[6] * a[7], cxz = a[2] - a[6] * a[8];
  const float cyy = a[3] - a[7] * a[7], cyz = a[4] - a[7] * a[8], czz = a[5] - a[8] * a[8];
  float ev, v[3];
  eigen33_smallest(cxx, cxy, cxz, cyy, cyz, czz, &ev, v);
  const float eig_sum = cxx + cyy + czz;
  o.w = (eig_sum != 0.0f) ? fabsf(ev / eig_sum) : 0.0f;
  const float vx = 0.0f - pq.x, vy = 0.0f - pq.y, vz = 0.0f - pq.z;
  const float cos_theta = vx * v[0] + vy * v[1] + vz * v[2];
  if (cos_theta < 0.0f) { v[0] *= -1.0f; v[1] *= -1.0f; v[2] *= -1.0f; }
  o.x = v[0]; o.y = v[1]; o.z = v[2];
  return o;
}

}  // namespace mm3d
