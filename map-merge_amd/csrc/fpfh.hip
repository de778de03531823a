// fpfh.hip -- computeLocalDescriptors(FPFH) on gfx950 (K5/K6 in SURVEY 2.2).
//
// R/src/features.cpp:99-150 with R/src/dispatch_descriptors.h:40: pcl::FPFHEstimation with
// setRadiusSearch(feature_radius), setSearchSurface(points), setInputNormals(normals),
// setInputCloud(keypoints); descriptors with a non-finite bin are pruned together with their
// keypoints (features.cpp:118-143).
//   1. support set S = union of the keypoints' radius neighbourhoods (std::set order = index order)
//   2. SPFH for every s in S over its own neighbourhood: Darboux pair features -> 3 x 11 bins,
//      each hit adds 100/(|N(s)|-1)   (pfh.cpp computePairFeatures; FPFH's wrapper always bins)
//   3. per keypoint: sum of SPFH rows weighted by 1/d2 (d2 == 0 skipped), each 11-bin block scaled
//      to sum 100.
// Algorithmic traffic: SPFH 156 B per support point; weighting 132 B per gathered row.
#include "sorted_nb.hpp"

namespace mm3d {

constexpr int kBins = 11;
constexpr int kDim = 33;

// 1. mark the support set (by original index): one wave per keypoint, lanes stride over the
// candidate row spans (coalesced 16-byte loads)
__global__ void __launch_bounds__(256)
k_fpfh_mark(const float4 *__restrict__ kp, int nk, GridView g, float radius, float r2, int *__restrict__ in_set)
{
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (k >= nk) return;
  const int lane = threadIdx.x & 63;
  const float4 q = kp[k];
  const float ri = radius * 1.0001f + 1e-4f;
  if (cell_floor(q.x + ri, g.minx, g.inv) < 0 || cell_floor(q.x - ri, g.minx, g.inv) > g.dx - 1) return;
  const int x0 = clampi(cell_floor(q.x - ri, g.minx, g.inv), 0, g.dx - 1), x1 = clampi(cell_floor(q.x + ri, g.minx, g.inv), 0, g.dx - 1);
  int y0 = cell_floor(q.y - ri, g.miny, g.inv), y1 = cell_floor(q.y + ri, g.miny, g.inv);
  int z0 = cell_floor(q.z - ri, g.minz, g.inv), z1 = cell_floor(q.z + ri, g.minz, g.inv);
  y0 = y0 < 0 ? 0 : y0; z0 = z0 < 0 ? 0 : z0;
  y1 = y1 > g.dy - 1 ? g.dy - 1 : y1; z1 = z1 > g.dz - 1 ? g.dz - 1 : z1;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      const int row = (z * g.dy + y) * g.dx;
      const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
      for (int j = b + lane; j < e; j += kWave) {
        const float4 p = g.pts[j];
        if (dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2) in_set[__float_as_int(p.w)] = 1;
      }
    }
}

// hil_pos (optional): Hilbert position by original index; a support point's normal then carries that position in .w (-1 for
// the others), which is how k_spfh recognises a candidate that is one of its own block's points
__global__ void k_fpfh_support(const float4 *__restrict__ sorted, int n, const int *__restrict__ in_set,
                               const int *__restrict__ pos, int *__restrict__ row_of_sorted /* by sorted position, -1 if none */,
                               const float4 *__restrict__ nrm, const int *__restrict__ hil_pos, float4 *__restrict__ nrm_sorted)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int oi = __float_as_int(sorted[j].w);
  float4 v = nrm[oi];
  const bool in = in_set[oi] != 0;
  if (hil_pos) v.w = __int_as_float(in ? hil_pos[oi] : -1);
  nrm_sorted[j] = v;
  row_of_sorted[j] = in ? pos[oi] : -1;
}
__global__ void k_hil_inverse(const float4 *__restrict__ hil_pts, int n, int *__restrict__ hil_pos)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hil_pos[__float_as_int(hil_pts[i].w)] = i;
}

// pcl::computePairFeatures (features/src/pfh.cpp); returns f1,f2,f3 (all 0 on the degenerate exits).
// `symmetric` (optional): would the call with the two points exchanged return the same bits?  It works on the negated
// difference vector, so its angles are (-angle2, -angle1): whenever exactly one of the two calls takes the "switch p1 and p2"
// branch, both continue with the same source point, difference vector and normals.  On a tie neither switches and the
// results differ (tests/test_oracle_cpu.py::test_pair_features_are_symmetric_under_the_swap_except_on_ties).
__device__ __forceinline__ void pair_features(const float4 &p1, const float4 &n1, const float4 &p2, const float4 &n2,
                                              float &f1, float &f2, float &f3, bool *symmetric = nullptr)
{
  float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
  const float f4 = sqrtf(dx * dx + dy * dy + dz * dz);
  if (symmetric) *symmetric = true;
  if (f4 == 0.0f) { f1 = f2 = f3 = 0.0f; return; }
  float ax = n1.x, ay = n1.y, az = n1.z, bx = n2.x, by = n2.y, bz = n2.z;
  const float angle1 = (ax * dx + ay * dy + az * dz) / f4;
  const float angle2 = (bx * dx + by * dy + bz * dz) / f4;
  // acos(fabs(angle1)) > acos(fabs(angle2)) evaluated in double on the CPU: device_util.hpp::acos_abs_greater
  const bool sw = acos_abs_greater(angle1, angle2);
  if (symmetric) *symmetric = sw || acos_abs_greater(angle2, angle1);
  if (sw) {
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

// static_cast<int>(floor(x)) with x86 semantics for NaN / out of range (INT_MIN -> clamped to 0)
__device__ __forceinline__ int bin_of(double x)
{
  const double f = floor(x);
  int h = (f >= -2147483648.0 && f <= 2147483647.0) ? (int)f : (-2147483647 - 1);
  h = h < 0 ? 0 : h;
  return h >= kBins ? kBins - 1 : h;
}

// ---- the three BINS of a pair, certified (round 6) ------------------------------------------------------------------
// A pair's features only matter through floor(11 (f1 + pi) / (2 pi)), floor(11 (f2 + 1) / 2), floor(11 (f3 + 1) / 2) and
// the "switch p1 and p2" decision acos|a1| > acos|a2|.  pair_features above reproduces the CPU path's floats -- restated
// glibc atan2f, two square roots, five IEEE divisions, double bins: ~300 instructions.  This evaluates the same geometry
// in plain f32 (v_rsq / v_rcp, fma dot products, a degree-15 odd polynomial for the arc tangent: ~150 instructions) TOGETHER
// WITH a bound on how far each feature can lie from the CPU path's float, and answers only when no feature is within its
// bound of a bin edge, the two angles are not within theirs of a tie, and nothing is near a degenerate exit.  Everything
// else -- about one pair in 10^4 -- takes pair_features.  The bins it answers with are the CPU path's.
//
// Bounds (u = 2^-24; both chains against the REAL function of the same float inputs d = p2 - p1, n1, n2 -- the subtraction
// is the same IEEE operation on both sides; |n1|, |n2| within 1 % of 1 is required; kappa = |d| |n1| / |d x n1| >= 1):
//   a_k = n_k . d / |d|      CPU: 3 roundings in the dot product (<= 3 u |n||d|), sqrt of a 3-rounding sum 2.5 u, division 1 u
//                            -> 6.5 u;  here: fma dot 3 u, v_rsq (1 ulp) of a 3 u sum 3.5 u, product 1 u -> 7.5 u;  sum 14 u
//   v = d x n1               each component within 2 u |d||n1| on both sides: |dv| <= 3.5 u |d||n1| = 3.5 u kappa |v|
//   f2 = v^ . n2             CPU: v^ within (7 kappa + 3.5) u, dot 3 u -> (10.5 kappa + 3) u;  here (7 kappa + 7.5) u
//   y = n2 . (n1 x v^), x = n1 . n2     CPU within 17 kappa u and 3 u;  here within (7 kappa + 11) u and 3 u
//   f1 = atan2(y, x)         a perturbation delta of (x, y) with delta <= rho / 2, rho = |(x, y)|, turns the angle by at most
//                            1.6 delta / rho;  glibc's atan2f within 1 ulp (<= 5 u);  the polynomial arc tangent below within
//                            10 u in all (fit 0.04 u, float evaluation, the two folds; tests/test_gpu_parity.py checks it)
// The constants below are TWICE these sums.  In bin units t = 11 (f + c) / w the float evaluation of t adds <= 22 u.
__device__ __forceinline__ float atan_poly01(float z)      // atan(z) on [0, 1]: z P(z^2), |error| < 4e-8 before rounding
{
  const float w = z * z;
  float p = -0.004054537974298f;
  p = fmaf(p, w, 0.02186284214258194f);
  p = fmaf(p, w, -0.055912140756845474f);
  p = fmaf(p, w, 0.09642181545495987f);
  p = fmaf(p, w, -0.13908623158931732f);
  p = fmaf(p, w, 0.19946563243865967f);
  p = fmaf(p, w, -0.33329859375953674f);
  p = fmaf(p, w, 0.9999993443489075f);
  return p * z;
}
__device__ __forceinline__ float atan2_fast(float y, float x)
{
  const float ay = fabsf(y), ax = fabsf(x);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float r = atan_poly01(mn * __builtin_amdgcn_rcpf(mx));
  r = ay > ax ? 1.57079637f - r : r;
  r = x < 0.0f ? 3.14159274f - r : r;
  return copysignf(r, y);
}

__device__ __forceinline__ bool pair_bins_fast(const float4 &p1, const float4 &n1, const float4 &p2, const float4 &n2, int &h1, int &h2, int &h3)
{
  constexpr float U = 0x1p-24f;
  const float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
  const float s = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  const float inv = __builtin_amdgcn_rsqf(s);
  const float A2 = fmaf(n1.z, n1.z, fmaf(n1.y, n1.y, n1.x * n1.x)), B2 = fmaf(n2.z, n2.z, fmaf(n2.y, n2.y, n2.x * n2.x));
  const float a1 = fmaf(n1.z, dz, fmaf(n1.y, dy, n1.x * dx)) * inv, a2 = fmaf(n2.z, dz, fmaf(n2.y, dy, n2.x * dx)) * inv;
  const float m1 = fabsf(a1), m2 = fabsf(a2);
  // (every condition is of the form "value > bound": a NaN anywhere answers false)
  bool ok = s > 1e-12f && s < 1e12f && A2 > 0.98f && A2 < 1.02f && B2 > 0.98f && B2 < 1.02f;
  ok = ok && fabsf(m1 - m2) > 64.0f * U && fmaxf(m1, m2) < 1.0f - 1e-5f;
  const bool sw = m1 < m2;                       // acos|a1| > acos|a2|: certain, by the margin
  const float ax = sw ? n2.x : n1.x, ay = sw ? n2.y : n1.y, az = sw ? n2.z : n1.z;
  const float bx = sw ? n1.x : n2.x, by = sw ? n1.y : n2.y, bz = sw ? n1.z : n2.z;
  const float ex = sw ? -dx : dx, ey = sw ? -dy : dy, ez = sw ? -dz : dz;
  const float f3 = sw ? -a2 : a1;
  const float vx = fmaf(ey, az, -(ez * ay)), vy = fmaf(ez, ax, -(ex * az)), vz = fmaf(ex, ay, -(ey * ax));
  const float v2 = fmaf(vz, vz, fmaf(vy, vy, vx * vx));
  ok = ok && v2 > s * 1e-6f;                     // kappa < 1000: far from the "v_norm == 0" exit
  const float rv = __builtin_amdgcn_rsqf(v2);
  const float kappa = (s * inv) * rv * 1.02f;    // |d| |n1| / |v|, |n1| <= 1.01
  const float f2 = fmaf(vz, bz, fmaf(vy, by, vx * bx)) * rv;
  const float wx = fmaf(ay, vz, -(az * vy)), wy = fmaf(az, vx, -(ax * vz)), wz = fmaf(ax, vy, -(ay * vx));
  const float y = fmaf(wz, bz, fmaf(wy, by, wx * bx)) * rv;
  const float x = fmaf(az, bz, fmaf(ay, by, ax * bx));
  const float f1 = atan2_fast(y, x);
  const float irho = __builtin_amdgcn_rsqf(fmaf(y, y, x * x));
  // bin coordinates and their bounds
  const float t1 = fmaf(f1, 1.75070429f, 5.5f);                    // 11 d_pi f1 + 11 d_pi pi, d_pi = float(1 / (2 pi_f))
  const float t2 = fmaf(f2, 5.5f, 5.5f), t3 = fmaf(f3, 5.5f, 5.5f);
  const float e1 = ((80.0f * kappa + 40.0f) * irho + 30.0f) * (1.76f * U) + 24.0f * U;
  const float e2 = (220.0f * kappa + 144.0f) * U;
  const float e3 = 192.0f * U;
  const float g1 = floorf(t1), g2 = floorf(t2), g3 = floorf(t3);
  const float r1 = t1 - g1, r2 = t2 - g2, r3 = t3 - g3;
  ok = ok && fminf(r1, 1.0f - r1) > e1 && fminf(r2, 1.0f - r2) > e2 && fminf(r3, 1.0f - r3) > e3;
  h1 = min(max((int)g1, 0), kBins - 1);
  h2 = min(max((int)g2, 0), kBins - 1);
  h3 = min(max((int)g3, 0), kBins - 1);
  return ok;
}

// 2. SPFH, wave-cooperative: a wave owns one compact patch of the surface (<= 64 points in Hilbert
// order, the lanes that belong to the support set are live), streams the box of cells those lanes
// can reach through LDS together with the candidates' normals, and bins the pairs inside every live
// lane's radius.  Every hit of one point adds the SAME float 100/(|N|-1), so a bin's value depends
// only on its hit count: hits are counted in integers (LDS) in whatever order they are processed, and
// the float chain "0 + incr + incr + ..." of the CPU loop is replayed once per bin at the end, bit for bit.
//
// The expensive part is the pair feature (~300 VALU instructions: atan2f, two square roots, five
// IEEE divisions, three double bin computations), so (a) its lanes must be full and (b) it should run
// once per PAIR, not once per histogram:
//  (a) per tile the cheap in-radius test leaves a per-lane bitset; the (owner lane, candidate) hits of the
//      whole wave are numbered by a prefix sum, written to an LDS pool and dealt to the 64 lanes: a lane
//      works on ANY point's hit (the owner's position and normal are in LDS) and bumps the owner's counters
//      with LDS atomics.  Trip count = total hits / 64 instead of the largest per-lane hit count.
//  (b) computePairFeatures(p, q) and (q, p) return the same bits unless the pair is a tie of its two angles
//      (pair_features above), and a quarter of a point's neighbours are among the 256 consecutive points of
//      its BLOCK (kSpfhWaves waves, whose points and counters share the block's LDS).  A candidate that is a
//      live point of this block carries its block offset; of the two hits (o, c) and (c, o) only the one with
//      the lower offset as owner is pooled -- by whichever wave owns o --, and its lane votes into BOTH
//      histograms (a tie is evaluated the second way as well).  The neighbour count |N| of each point still
//      comes from its own in-radius test.  The waves stream their own boxes and never wait for each other
//      (a first version staged one box per block: 1 340 candidates to test per point instead of ~600, which
//      cost what the shared pairs saved).
constexpr int kSpfhTile = 64;
#ifndef MM3D_SPFH_POOL
#define MM3D_SPFH_POOL 1536
#endif
constexpr int kSpfhPool = MM3D_SPFH_POOL;
#ifndef MM3D_SPFH_WAVES
#define MM3D_SPFH_WAVES 4
#endif
constexpr int kSpfhWaves = MM3D_SPFH_WAVES;
#ifdef MM3D_SPFH_STATS
__device__ unsigned long long g_spfh_stats[16];  // 0 waves, 1 candidates tested per wave, 2 in-radius hits, 3 pooled hits, 4 second votes, 5 ties, 6 live points,
                                                 // 7 pooled hits the certified bins refused, 8 (-DMM3D_SPFH_VERIFY) certified bins that differ from the exact ones
#define MM3D_SPFH_STAT(i_, v_) atomicAdd(&g_spfh_stats[i_], (unsigned long long)(v_))
#else
#define MM3D_SPFH_STAT(i_, v_)
#endif
#ifndef MM3D_SPFH_FAST
#define MM3D_SPFH_FAST 1
#endif
constexpr bool kSpfhFast = MM3D_SPFH_FAST != 0;   // 0: every pair through pair_features (the A/B)
constexpr int kSpfhT = 64 * kSpfhWaves;       // points per block
constexpr int kSpfhNone = 0xFFFF;             // a candidate that is not a live point of the block: above every offset
__global__ void __launch_bounds__(kSpfhT)
k_spfh(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g,
       const float4 *__restrict__ nrm /* original order */, const float4 *__restrict__ nrm_sorted /* .w = Hilbert position of a support point, -1 */,
       const int *__restrict__ in_set, const int *__restrict__ pos, float radius, float r2, float *__restrict__ spfh /* [ns][33] */,
       int *__restrict__ error)
{
  __shared__ float4 s_pts[kSpfhWaves][kSpfhTile];           // a wave's staged tile (.w of a point: its block offset if it is a live point of this block)
  __shared__ float4 s_nrm[kSpfhWaves][kSpfhTile];
  __shared__ int s_off[kSpfhWaves][64];
  __shared__ int s_beg[kSpfhWaves][64];
  __shared__ float4 s_q[kSpfhT], s_nq[kSpfhT];              // the block's points and normals by block offset
  // hit counters, two 16-bit counts to a word (offsets 2w and 2w + 1): half the LDS of one word per point, which is
  // what bounds the blocks per CU; a point with more than 65535 neighbours is reported, not miscounted
  __shared__ unsigned hist[kDim][kSpfhT / 2];
  __shared__ unsigned short s_pool[kSpfhWaves][kSpfhPool];
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int item0 = (int)bid * kSpfhWaves;
  const int blk_first = items[item0].x;                      // (the grid has no block without an item)
  const int last_item = min(item0 + kSpfhWaves, n_items) - 1;
  const int blk_count = items[last_item].x + items[last_item].y - blk_first;   // items are consecutive and hold <= 64 points
  const int2 it = item0 + wave < n_items ? items[item0 + wave] : make_int2(blk_first, 0);
  const bool valid = lane < it.y;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f), nq = make_float4(0.f, 0.f, 0.f, 0.f);
  int self = 0;
  bool live = false;
  const int wave_off0 = it.x - blk_first, my_off = wave_off0 + lane;
  if (valid) {
    q = q_pts[it.x + lane];
    self = __float_as_int(q.w);
    live = in_set[self] != 0;
    if (live) nq = nrm[self];
    s_q[my_off] = q;
    s_nq[my_off] = nq;
  }
  for (int i = tid; i < kDim * (kSpfhT / 2); i += kSpfhT) (&hist[0][0])[i] = 0u;
  __syncthreads();                                            // the block's points and zeroed counters are in LDS
  int cnt = 0;
  if (ballot(live)) {                                         // wave-uniform: a patch without a support point has nothing to bin
    if (lane == 0) MM3D_SPFH_STAT(0, 1);
    if (live) MM3D_SPFH_STAT(6, 1);
    const float ri = radius * 1.0001f + 1e-4f;
    const float lx = wave_min_f(live ? q.x : INFINITY), hx = wave_max_f(live ? q.x : -INFINITY);
    const float ly = wave_min_f(live ? q.y : INFINITY), hy = wave_max_f(live ? q.y : -INFINITY);
    const float lz = wave_min_f(live ? q.z : INFINITY), hz = wave_max_f(live ? q.z : -INFINITY);
    const int x0 = max(cell_floor(lx - ri, g.minx, g.inv), 0), x1 = min(cell_floor(hx + ri, g.minx, g.inv), g.dx - 1);
    const int y0 = max(cell_floor(ly - ri, g.miny, g.inv), 0), y1 = min(cell_floor(hy + ri, g.miny, g.inv), g.dy - 1);
    const int z0 = max(cell_floor(lz - ri, g.minz, g.inv), 0), z1 = min(cell_floor(hz + ri, g.minz, g.inv), g.dz - 1);
    const float d_pi = 1.0f / (2.0f * 3.14159274f);
    const float4 *sp = s_pts[wave];
    const float4 *sn = s_nrm[wave];
    unsigned short *pool = s_pool[wave];
    wave_stream_box<kSpfhTile, 1>(
        g, x0, x1, y0, y1, z0, z1, s_pts[wave], s_nrm[wave], s_off[wave], s_beg[wave], lane,
        [&](int j, float4 (&out)[1]) { out[0] = nrm_sorted[j]; },
        [&](int n_tile) {
          if (lane == 0) MM3D_SPFH_STAT(1, n_tile);
          // the staged normal's .w is the candidate's Hilbert position if it is a support point (-1 otherwise): a live
          // point of this block becomes its block offset in the POINT's .w (the original index is not needed any more:
          // "is it me" is an offset test), everything else kSpfhNone
          if (lane < n_tile) {
            const int hp = __float_as_int(s_nrm[wave][lane].w), off = hp - blk_first;
            s_pts[wave][lane].w = __int_as_float((hp >= 0 && off >= 0 && off < blk_count) ? off : kSpfhNone);
          }
          wave_lds_fence();
          unsigned all[kSpfhTile / 32], todo[kSpfhTile / 32];
#pragma unroll
          for (int gq = 0; gq < kSpfhTile / 32; ++gq) {
            unsigned bits = 0, skip = 0;
            if (gq * 32 < n_tile) {                           // wave-uniform
#pragma unroll 4
              for (int b = 0; b < 32; ++b) {
                const float4 p = sp[gq * 32 + b];             // slots >= n_tile hold stale points: masked off below
                bits |= (dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2) ? (1u << b) : 0u;
                // the pair belongs to the end point with the lower block offset (== : the point itself, "p_idx == indices[idx]")
                skip |= (__float_as_int(p.w) <= my_off) ? (1u << b) : 0u;
              }
              const int rem = n_tile - gq * 32;
              if (rem < 32) bits &= (1u << rem) - 1u;
            }
            all[gq] = live ? bits : 0u;
            todo[gq] = all[gq] & ~skip;
          }
          cnt += __popc(all[0]) + __popc(all[1]);
          const int mine = __popc(todo[0]) + __popc(todo[1]);
          MM3D_SPFH_STAT(2, __popc(all[0]) + __popc(all[1]));
          MM3D_SPFH_STAT(3, mine);
          // number the wave's hits: exclusive prefix of the per-lane counts
          int incl = mine;
#pragma unroll
          for (int o = 1; o < kWave; o <<= 1) {
            const int t = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += t;
          }
          const int total = __shfl(incl, kWave - 1, kWave);
          const int first = incl - mine;
          for (int base = 0; base < total; base += kSpfhPool) {   // one chunk unless nearly every lane hits every candidate
            {
              unsigned m0 = todo[0], m1 = todo[1];
              int r = first;
              while (m0 | m1) {
                int k;
                if (m0) { k = __ffs((int)m0) - 1; m0 &= m0 - 1u; }
                else { k = 32 + __ffs((int)m1) - 1; m1 &= m1 - 1u; }
                if (r >= base && r < base + kSpfhPool) pool[r - base] = (unsigned short)((lane << 6) | k);
                ++r;
              }
            }
            wave_lds_fence();
            const int n = min(kSpfhPool, total - base);
            // Lane l takes the entries l * per .. l * per + per - 1: the pool is numbered owner by owner, so 64 CONSECUTIVE
            // entries are mostly one owner's hits, whose votes all land in the two banks of that owner's counter column;
            // entries `per` apart belong to different owners and spread over the banks.
            const int per = (n + kWave - 1) / kWave;
            for (int i = 0; i < per; ++i) {
              const int e = lane * per + i;
              const bool act = e < n;
              const unsigned ent = act ? pool[e] : 0u;
              const int o = wave_off0 + (int)(ent >> 6), k = (int)(ent & 63u);
              const float4 qo = s_q[o], no = s_nq[o];
              const float4 p = sp[k], np = sn[k];
              const int c = __float_as_int(p.w);
              // the certified bins where they can be had (nearly always); a lane that is refused one evaluates the CPU path's
              // floats (the wave waits for it: about one iteration in a hundred)
              int h1, h2, h3;
              bool sym = true;                       // (certified: the angles are clear of a tie, so the exchanged call gives the same bits)
              bool ok = true;
              if (kSpfhFast) ok = pair_bins_fast(qo, no, p, np, h1, h2, h3);
              else ok = false;
              MM3D_SPFH_STAT(7, (act && !ok) ? 1 : 0);
#ifdef MM3D_SPFH_VERIFY
              const bool cert = ok;
              const int c1 = h1, c2 = h2, c3 = h3;
              ok = false;
#endif
              if (ballot(act && !ok)) {
                if (act && !ok) {
                  float f1, f2, f3;
                  pair_features(qo, no, p, np, f1, f2, f3, &sym);
                  h1 = bin_of(kBins * (((double)f1 + 3.14159265358979323846) * (double)d_pi));
                  h2 = bin_of(kBins * (((double)f2 + 1.0) * 0.5));
                  h3 = bin_of(kBins * (((double)f3 + 1.0) * 0.5));
                }
              }
#ifdef MM3D_SPFH_VERIFY
              if (act && cert && (c1 != h1 || c2 != h2 || c3 != h3 || !sym)) MM3D_SPFH_STAT(8, 1);
#endif
              if (act) {
                const unsigned one = 1u << ((o & 1) << 4);
                atomicAdd(&hist[h1][o >> 1], one);
                atomicAdd(&hist[kBins + h2][o >> 1], one);
                atomicAdd(&hist[2 * kBins + h3][o >> 1], one);
                if (c != kSpfhNone) {                          // the same pair seen from the candidate, a point of this block
                  MM3D_SPFH_STAT(4, 1);
                  if (!sym) {
                    MM3D_SPFH_STAT(5, 1);
                    float f1, f2, f3;
                    pair_features(p, np, qo, no, f1, f2, f3);
                    h1 = bin_of(kBins * (((double)f1 + 3.14159265358979323846) * (double)d_pi));
                    h2 = bin_of(kBins * (((double)f2 + 1.0) * 0.5));
                    h3 = bin_of(kBins * (((double)f3 + 1.0) * 0.5));
                  }
                  const unsigned onec = 1u << ((c & 1) << 4);
                  atomicAdd(&hist[h1][c >> 1], onec);
                  atomicAdd(&hist[kBins + h2][c >> 1], onec);
                  atomicAdd(&hist[2 * kBins + h3][c >> 1], onec);
                }
              }
            }
            wave_lds_fence();
          }
        },
        // only points within the radius of the patch's bounding box can be in range
        KeepNearBox{lx, hx, ly, hy, lz, hz, ri * ri});
  }
  __syncthreads();                                            // all votes are in (other waves vote into this wave's columns)
  if (!live) return;
  if (cnt > 65535) { atomicExch(error, 1); return; }
  const float hist_incr = 100.0f / (float)(cnt - 1);
  float *o = spfh + (size_t)pos[self] * kDim;
  for (int b = 0; b < kDim; ++b) {
    const unsigned hits = (hist[b][my_off >> 1] >> ((my_off & 1) << 4)) & 0xffffu;
    float v = 0.0f;
    for (unsigned i = 0; i < hits; ++i) v += hist_incr;
    o[b] = v;
  }
}

// 3. weighting (FPFHEstimation::weightPointSPFHSignature): per keypoint, over its neighbours IN radiusSearch's
// ORDER ((distance, index), d2 == 0 skipped): val = spfh[neighbour][bin] * (1 / d2); the bin takes "+= val" in
// float, its 11-bin block's sum takes "+= val" in double, neighbour-major, bin-minor.  Both are chains, so the
// neighbour lists are built in that order first (sorted_nb.hpp, payload = (d2, SPFH row)); a lane then owns one
// (keypoint, 11-bin block) and walks the keypoint's list: 16 keypoints x 3 blocks = 48 chains per wave.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_fpfh_weight(const float4 *__restrict__ q_pts /* keypoints, Hilbert order, .w = keypoint index */, const int2 *__restrict__ items,
              int n_items, GridView g, const float4 *__restrict__ surface /* the points in original order */,
              const int *__restrict__ row_of /* support row by original point index */,
              const float *__restrict__ spfh, float radius, float r2, SnScratch scr, float *__restrict__ desc /* [nk][33] */,
              int *__restrict__ valid)
{
  __shared__ SnLds lds[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SnLds &L = lds[wave];
  const size_t slot = (size_t)blockIdx.x * 4 + wave;
  unsigned long long *tmp = scr.tmp + slot * kSnEntries;
  float2 *fin = (float2 *)scr.fin + slot * kSnEntries;
  const int n_units = sn_unit_count(scr, n_items);
  for (;;) {
    const int unit = sn_claim_unit(scr.unit_ctr, n_units, lane);
    if (unit < 0) break;
    const int2 it = items[sn_unit_item(scr, unit)];
    int first = (unit & 3) * kSnG;
    int left = min(kSnG, it.y - first);
    while (left > 0) {
      const int pq = lane >> 2;
      const float4 q = q_pts[it.x + first + (pq < left ? pq : 0)];
      const int fit = sn_build_lists<float2>(g, L, q.x, q.y, q.z, left, radius, r2, surface, tmp, fin, scr.error, lane,
                                             [&](float d2, unsigned idx, const float4 &) { return make_float2(d2, __int_as_float(row_of[idx])); });
      // chains: lane = keypoint * 3 + block
      const int p = lane / 3, f = lane - p * 3;
      if (p < fit) {
        const int base = L.list_off[p], m = L.list_off[p + 1] - base;
        float out[kBins];
#pragma unroll
        for (int b = 0; b < kBins; ++b) out[b] = 0.0f;
        double sum = 0.0;
        // four list entries and their SPFH blocks are requested at a time
        for (int e0 = 0; e0 < m; e0 += 4) {
          float2 ent[4];
          float hv[4][kBins];
#pragma unroll
          for (int u = 0; u < 4; ++u) ent[u] = fin[base + min(e0 + u, m - 1)];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float *h = spfh + (size_t)__float_as_int(ent[u].y) * kDim + f * kBins;
#pragma unroll
            for (int b = 0; b < kBins; ++b) hv[u][b] = h[b];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (e0 + u >= m || ent[u].x == 0.0f) continue;  // "minus the query point itself"
            const float weight = 1.0f / ent[u].x;
#pragma unroll
            for (int b = 0; b < kBins; ++b) {
              const float val = __fmul_rn(hv[u][b], weight);
              sum += (double)val;
              out[b] = __fadd_rn(out[b], val);
            }
          }
        }
        const int k = __float_as_int(q_pts[it.x + first + p].w);
        float *o = desc + (size_t)k * kDim + f * kBins;
        if (m == 0) {
#pragma unroll
          for (int b = 0; b < kBins; ++b) o[b] = __uint_as_float(0x7fc00000u);
        } else {
          if (sum != 0.0) sum = 100.0 / sum;
          const float sc = (float)sum;
#pragma unroll
          for (int b = 0; b < kBins; ++b) o[b] = __fmul_rn(out[b], sc);
        }
        if (f == 0) valid[k] = m != 0 ? 1 : 0;
      }
      wave_lds_fence();
      first += fit;
      left -= fit;
    }
  }
}

__global__ void k_compact_rows(const float *__restrict__ in, const int *__restrict__ flags, const int *__restrict__ pos,
                               int n, int dim, float *__restrict__ out)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n * dim) return;
  const int r = (int)(e / dim), cidx = (int)(e % dim);
  if (flags[r]) out[(size_t)pos[r] * dim + cidx] = in[e];
}

mm3d_desc *compute_fpfh(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, mm3d_cloud *keypoints,
                        double radius)
{
  MM3D_REQUIRE(normals->n == points->n, "computeLocalDescriptors: normals and points differ in size");
  auto *res = new mm3d_desc();
  res->dim = kDim;
  res->type = MM3D_DESC_FPFH;
  const int nk = (int)keypoints->n;
  if (nk == 0) { res->n = 0; res->data = DevBuf<float>(c, 0); return res; }
  const float r2 = (float)(radius * radius);
  const Grid &g = cloud_grid(c, points, (float)(radius * 0.5));
  const int n = (int)points->n;
  DevBuf<float> raw(c, (size_t)nk * kDim);
  DevBuf<int> valid(c, (size_t)nk + 1);
  MM3D_HIP(hipMemsetAsync(valid.get(), 0, ((size_t)nk + 1) * sizeof(int), c->stream));
  if (g.n == 0) {
    // no surface: every descriptor is NaN and gets pruned
    res->n = 0; res->data = DevBuf<float>(c, 0);
    keypoints->pts = DevBuf<float4>(c, 0); keypoints->n = 0; keypoints->grids.clear(); keypoints->host.clear();
    keypoints->reset_caches();
    return res;
  }
  DevBuf<int> in_set(c, (size_t)n + 1), pos(c, (size_t)n + 1);
  MM3D_HIP(hipMemsetAsync(in_set.get(), 0, ((size_t)n + 1) * sizeof(int), c->stream));
  MM3D_LAUNCH(c, "fpfh_mark", nk * 16.0, k_fpfh_mark, dim3(div_up(nk, 4)), dim3(256), 0, keypoints->pts.get(), nk, g.view(),
              (float)radius, r2, in_set.get());
  exclusive_scan_int(c, in_set.get(), pos.get(), (size_t)n + 1);
  // The size of the support set (the points within the radius of some keypoint: six in ten on the headline) stays on the
  // device: the SPFH rows are sized by their bound, one per point, and the number comes back with the pruning's wait below --
  // one wait per map less than asking for it here (round 6); the kernels' algorithmic bytes are entered then.
  int *h = (int *)c->pin(64);
  DevBuf<int> row_of(c, g.n);
  DevBuf<float4> nrm_sorted(c, g.n);
  DevBuf<int> hil_pos;
  {
    cloud_hilbert(c, points);                            // query order + wave work items (shared with ICP / score)
    hil_pos = DevBuf<int>(c, (size_t)n + 1);
    MM3D_LAUNCH(c, "fpfh_support", points->n_finite * 20.0, k_hil_inverse, dim3(div_up(points->n_finite, 256)), dim3(256), 0,
                (const float4 *)points->hil_pts.get(), (int)points->n_finite, hil_pos.get());
  }
  MM3D_LAUNCH(c, "fpfh_support", g.n * 44.0, k_fpfh_support, dim3(div_up(g.n, 256)), dim3(256), 0, g.sorted.get(), g.n,
              in_set.get(), pos.get(), row_of.get(), normals->nrm.get(), (const int *)hil_pos.get(), nrm_sorted.get());
  DevBuf<float> spfh(c, (size_t)g.n * kDim);
  int *hse = nullptr;                                   // the SPFH kernel's error word, looked at with the weighting's below
  const int n_items = points->n_wave_items;
  if (n_items > 0) {
    DevBuf<int> spfh_err(c, 1);
    MM3D_HIP(hipMemsetAsync(spfh_err.get(), 0, sizeof(int), c->stream));
    MM3D_LAUNCH(c, "spfh", 0.0, k_spfh, dim3(div_up(n_items, kSpfhWaves)), dim3(kSpfhT), 0, (const float4 *)points->hil_pts.get(),
                (const int2 *)points->wave_items.get(), n_items, g.view(), (const float4 *)normals->nrm.get(),
                (const float4 *)nrm_sorted.get(), (const int *)in_set.get(), (const int *)pos.get(), (float)radius, r2, spfh.get(), spfh_err.get());
    hse = (int *)c->pin(64);
    MM3D_HIP(hipMemcpyAsync(hse, spfh_err.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  }
  {
    // row of the support set by ORIGINAL point index (what a sorted list entry carries)
    cloud_hilbert(c, keypoints);
    const int nki = keypoints->n_wave_items;
    SnLaunch<float2> sn(c, nki * 4, points->n);
    SnScratch scr{sn.tmp.get(), sn.fin.get(), sn.ctr.get(), sn.error()};
    if (keypoints->n_finite)
      MM3D_LAUNCH(c, "fpfh_weight", nk * 132.0, k_fpfh_weight, dim3(sn.blocks), dim3(256), 0,
                  (const float4 *)keypoints->hil_pts.get(), (const int2 *)keypoints->wave_items.get(), nki, g.view(),
                  (const float4 *)points->pts.get(), (const int *)pos.get(), (const float *)spfh.get(), (float)radius, r2, scr, raw.get(), valid.get());
    int *he = (int *)c->pin(64);
    MM3D_HIP(hipMemcpyAsync(he, sn.error(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (hse) c->check_later(hse, MM3D_EUNSUPPORTED, "computeLocalDescriptors(FPFH): a point has more than 65535 neighbours within the radius");
    c->check_later(he, MM3D_EUNSUPPORTED, "computeLocalDescriptors(FPFH): a keypoint has more than 16384 neighbours within the radius");
  }
  // prune invalid descriptors and the same keypoints (features.cpp:118-143)
  DevBuf<int> vpos(c, (size_t)nk + 1);
  exclusive_scan_int(c, valid.get(), vpos.get(), (size_t)nk + 1);
  MM3D_HIP(hipMemcpyAsync(h, vpos.get() + nk, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MM3D_HIP(hipMemcpyAsync(h + 1, pos.get() + n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  const int nv = h[0], ns = h[1];
  if (n_items > 0) c->prof_add_bytes("spfh", ns * 156.0);          // per point of the support set
  if (keypoints->n_finite) c->prof_add_bytes("fpfh_weight", (double)ns * 132.0);
  res->n = (size_t)nv;
  if (nv == nk) {
    res->data = std::move(raw);
  } else {
    res->data = DevBuf<float>(c, (size_t)nv * kDim);
    DevBuf<float4> kp2(c, nv);
    if (nv) {
      MM3D_LAUNCH(c, "compact_rows", nk * 264.0, k_compact_rows, dim3(div_up((size_t)nk * kDim, 256)), dim3(256), 0,
                  (const float *)raw.get(), (const int *)valid.get(), (const int *)vpos.get(), nk, kDim, res->data.get());
      MM3D_LAUNCH(c, "compact_rows", nk * 32.0, k_compact_rows, dim3(div_up((size_t)nk * 4, 256)), dim3(256), 0,
                  (const float *)keypoints->pts.get(), (const int *)valid.get(), (const int *)vpos.get(), nk, 4,
                  (float *)kp2.get());
    }
    c->settle();
    keypoints->pts = std::move(kp2);
    keypoints->n = (size_t)nv;
    keypoints->grids.clear();
    keypoints->host.clear();
    keypoints->reset_caches();
  }
  c->settle();
  return res;
}

}  // namespace mm3d

#ifdef MM3D_SPFH_STATS
extern "C" void mm3d_debug_spfh_stats(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mm3d::g_spfh_stats), sizeof(unsigned long long) * 16);     // (out: 16 words)
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mm3d::g_spfh_stats), z, sizeof(z)); }
}
#endif
