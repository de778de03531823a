// sift_cert.hpp -- detectKeypoints(SIFT): the keypoint DECISION certified, not the DoG bits reproduced (round 6).
//
// R/src/features.cpp:45-62: the reference copies the keypoints' x, y, z and drops the scale, so what SIFTKeypoint's scale
// space decides per octave is an INDEX SET: point i is a keypoint at scale s iff |DoG(i, s)| >= min_contrast and DoG(i, s)
// is the minimum (maximum) over the point's 25 nearest neighbours and the three adjacent scales.  No DoG float leaves the
// stage.  The sorted-list kernels (k_sift_dog_lds) reproduce the CPU path's float sums bit for bit -- neighbour lists in
// (distance, index) order, glibc's expf in double, ordered additions: 8 wave-instructions per list entry.  Here, for the
// later octaves (the first keeps its lists: the fused normals need them):
//
//   1. k_sift_dog_fast: ONE UNSORTED pass over the staged tile per query.  The set of neighbours the CPU loop sums at a
//      scale is order free -- its `break` is a threshold on the squared distance, d2 <= 9 sigma^2, and d2 is computed
//      here with FLANN's own operations -- so the six (numerator, denominator) pairs are summed in any order with
//      v_exp_f32 weights, together with the neighbour COUNT of every scale.  Out: val*(i, s) and a RIGOROUS bound
//      B(i, s) >= |CPU path's float DoG - val*| (derivation below).
//   2. k_sift_pack_iv: intervals [val* - B, val* + B] per point and scale; contrast classes: certainly below the
//      contrast (no candidate), certainly above (live), open.
//   3. k_sift_reject: a (point, scale) that is no extremum has many neighbours below AND above it, and ONE certain violator
//      per side decides "no" if it is among the 25 nearest -- which every point inside the widest ball that the unsorted pass
//      counted at most 25 points in is.  97 % of the candidates end here.  What is left -- the keypoints themselves, near misses,
//      open comparisons -- gets the exact 25-nearest test of k_sift_extrema (nearest certain / possible violator as (distance,
//      index) keys, then the number of closer points) on intervals: k_sift_extrema_one, one wave per point inside the narrowest
//      counted ball with at least 25 points (a point whose 3 sigma_max ball holds fewer doubles its radius until the ball does):
//      a neighbour CERTAINLY below decides "no", no neighbour POSSIBLY below-or-equal decides "yes", anything else is open --
//      the point and the neighbours whose comparison is open are marked.
//   4. the marked points (0.05 - 0.1 % of an octave on the headline maps, scripts/sift_price.py) get the CPU path's exact
//      DoG floats from k_sift_dog_lds on single-query items; their intervals collapse to points and the test of the open
//      points is taken again: every comparison that was open is now exact against exact, every other one stays decided
//      (the exact value lies inside the old interval), so nothing is open afterwards.
//
// The bound.  u = 2^-24.  A response is R = N / D, N = sum_j I_j w_j, D = sum_j w_j over the n neighbours with d2_j <= T,
// all terms >= 0 (intensities are in [0, 255]).  Real weight: w^_j = exp(-0.5 d2_j / sigma^2) with the FLOAT sigma^2.
//   CPU path (PCL 1.8.1 sift_keypoint.hpp computeScaleSpace, restated in oracle/o_sift.c):
//     x_j = RN(-0.5 d2_j / sigma^2): one rounding, |x| <= 4.5  -> e^x within 4.5 u (1 + u) of w^_j (relative);
//     expf: within 1 ulp <= 2 u;  value * w: 1 u;  n - 1 sequential float additions: gamma_(n-1) <= (n - 1) u (1 + n u);
//     the final division 1 u.     |R_cpu - R^| <= R^ (2 n + 13.2) u (1 + second order).
//   Device: p = RN(d2 * c), c = RN(-0.5 log2(e) / sigma^2): two roundings of a number of magnitude <= 6.5 -> 2^p within
//     ln 2 * 13.0001 u = 9.02 u of w^; v_exp_f32 within 2 ulp (checked exhaustively over the range by
//     tests/test_gpu_sift_cert.py against double) = 4 u; fmaf / add chains of at most kmax terms per wave-partial, the W
//     partials added in float: gamma_(kmax + W); the division 1 u.   |R* - R^| <= R^ (2 (kmax + W) + 28) u (1 + ...).
//   DoG = RN(R_(s+1) - R_s) on both sides: + 2 * 255 u.
//   B_resp = R* (2 n + 2 (kmax + W) + 48) u (1 + 2^-6)   (R^ <= R* (1 + ...); the second-order terms; n < 2^17)
//   B_dog(s) = (B_resp(s + 1) + B_resp(s) + 2^-14) (1 + 2^-18)
// tests/test_gpu_sift_cert.py checks |oracle float DoG - val*| <= B on every point of its scenes, and the run itself
// checks it on every marked point (their exact values are computed anyway): a violation sends the octave to the
// sorted-list path and is counted (mm3d_debug_sift_cert_stats).
#pragma once

#include "snb_lds.hpp"

namespace mm3d {

constexpr int kCertScales = 6, kCertDog = 5, kCertKnn = 25;

struct SfScales {
  float T[kCertScales];      // effective thresholds: neighbour of scale s iff d2 <= T[s]  (= min(9 sigma^2, pred(r2)))
  float c[kCertScales];      // RN(-0.5 log2(e) / sigma^2)
  float TA, TB;              // T[0] / 2, T[0] / 4: two more neighbour counts for the reject radius (k_sift_reject)
};

template <int WAVES_, int TILE_CAP_>
struct SfCfg {
  static constexpr int kWaves = WAVES_;
  static constexpr int kTileCap = TILE_CAP_;
  static constexpr int kVals = 3 * kCertScales + 2;        // per query and wave: six numerators, denominators, counts + two small-ball counts
  static_assert(WAVES_ == 8, "the reduction deals its eight sums to eight waves");
  static_assert(TILE_CAP_ * 16 >= WAVES_ * (3 * kCertScales + 2) * 64 * 4, "the partial sums reuse the tile's memory");
};
#ifndef MM3D_SF_TILE
#define MM3D_SF_TILE 2816
#endif
#ifndef MM3D_SF_BLOCKS
#define MM3D_SF_BLOCKS 2
#endif
using SfCfgDefault = SfCfg<8, MM3D_SF_TILE>;

template <class Cfg>
struct alignas(16) SfLds {
  float4 tile[Cfg::kTileCap + 4];            // staged candidates: x, y, z, intensity (then: the waves' partial sums)
  int off[Cfg::kWaves][64], beg[Cfg::kWaves][64];
  float resp[kCertScales][64], bres[kCertScales][64];
  int ntot[kCertScales + 2][64];             // neighbours inside T[0..5], TA, TB
  int n_tile[1];
  int item;
};

__device__ __forceinline__ float sf_intensity(float w)
{
  const unsigned c = __float_as_uint(w);
  const int r = (int)((c >> 16) & 255u), g = (int)((c >> 8) & 255u), b = (int)(c & 255u);
  return lm::fdiv_const((float)(299 * r + 587 * g + 114 * b), 1000.0f, 0.001f);     // (sift.hip::intensity_of: the CPU path's float)
}

// the certified scale space of one octave: val* and B for every point (by original index, [n][5]), the reject radius and the
// search radius (rlo2, rup2).  A block owns one work item (<= 64 queries, lane = query); the box of cells the item can reach
// is STREAMED through the tile: staged up to the tile's capacity (coalesced gathers, every wave a share of the slots), scanned
// by the eight waves (each a share of the candidates, broadcast LDS reads), staged again -- an unsorted pass needs no
// candidate twice, so no box is too large for it (round 6, second half: a first version staged the whole box at once and gave
// the items of a dense cloud -- 8 x 2 M indoor points: most of them -- up to the exact path, 56 map-pairs/s against 76).
template <class Cfg>
__global__ void __launch_bounds__(64 * Cfg::kWaves, MM3D_SF_BLOCKS * Cfg::kWaves / 4)
k_sift_dog_fast(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g /* .w = original index */,
                const float4 *__restrict__ pts /* original order: rgba */, float radius, SfScales sc, SnbCtl *ctl, int *__restrict__ ov_items,
                float *__restrict__ dogv, float *__restrict__ dogb, float *__restrict__ rlo2, float *__restrict__ rup2,
                unsigned char *__restrict__ need_exact)
{
  __shared__ SfLds<Cfg> S;
  constexpr int W = Cfg::kWaves, T = 64 * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float ri = radius * 1.0001f + 1e-4f;
  (void)ov_items; (void)need_exact;
  for (;;) {
    if (threadIdx.x == 0) { S.item = snb_claim_item(ctl->item_ctr, n_items); S.n_tile[0] = 0; }
    __syncthreads();
    if (S.item < 0) break;
    const int2 it = items[S.item];
    const bool live = lane < it.y;
    const float4 q = live ? q_pts[it.x + lane] : make_float4(kSnbFar, kSnbFar, kSnbFar, 0.0f);
    const float lx = snb_min_f_dpp(live ? q.x : INFINITY), hx = snb_max_f_dpp(live ? q.x : -INFINITY);
    const float ly = snb_min_f_dpp(live ? q.y : INFINITY), hy = snb_max_f_dpp(live ? q.y : -INFINITY);
    const float lz = snb_min_f_dpp(live ? q.z : INFINITY), hz = snb_max_f_dpp(live ? q.z : -INFINITY);
    const int x0 = max(cell_floor(lx - ri, g.minx, g.inv), 0), x1 = min(cell_floor(hx + ri, g.minx, g.inv), g.dx - 1);
    const int y0 = max(cell_floor(ly - ri, g.miny, g.inv), 0), y1 = min(cell_floor(hy + ri, g.miny, g.inv), g.dy - 1);
    const int z0 = max(cell_floor(lz - ri, g.minz, g.inv), 0), z1 = min(cell_floor(hz + ri, g.minz, g.inv), g.dz - 1);
    const KeepNearBox keep{lx, hx, ly, hy, lz, hz, ri * ri};
    const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    const int nrows = (x0 <= x1 && ny > 0 && nz > 0) ? ny * nz : 0;
    float num[kCertScales], den[kCertScales];
    int cnt[kCertScales], cnt_a = 0, cnt_b = 0;
#pragma unroll
    for (int s = 0; s < kCertScales; ++s) { num[s] = 0.0f; den[s] = 0.0f; cnt[s] = 0; }
    // the pass over what the tile holds: lane = query, the wave's share of the candidates four at a time
    auto scan_tile = [&]() {
      __syncthreads();                               // the staged candidates and their number are visible
      const int n_tile = S.n_tile[0];
      if (threadIdx.x < 4) S.tile[n_tile + threadIdx.x] = make_float4(kSnbFar, kSnbFar, kSnbFar, 0.0f);   // pad to a multiple of four
      __syncthreads();
      const int n4 = (n_tile + 3) >> 2;            // groups of four
      const int per = (n4 + W - 1) / W;
      const int g0 = wave * per, g1 = min(n4, g0 + per);
      for (int gi = g0; gi < g1; ++gi) {
        float4 p[4];
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = S.tile[4 * gi + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = dist2(q.x, q.y, q.z, p[k].x, p[k].y, p[k].z);
        const float dmin = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
        // the supports are nested: once no lane has one of the four inside a scale, none has inside a narrower one
        bool go = true;
#pragma unroll
        for (int s = kCertScales - 1; s >= 0; --s) {
          go = go && ballot(dmin <= sc.T[s]) != 0ull;      // wave-uniform
          if (go) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const bool m = d[k] <= sc.T[s];
              const float e = __builtin_amdgcn_exp2f(d[k] * sc.c[s]);
              const float w = m ? e : 0.0f;
              num[s] = fmaf(p[k].w, w, num[s]);
              den[s] = den[s] + w;
              cnt[s] += m ? 1 : 0;
              if (s == 0) { cnt_a += d[k] <= sc.TA ? 1 : 0; cnt_b += d[k] <= sc.TB ? 1 : 0; }
            }
          }
        }
      }
      __syncthreads();                               // every wave is done with the tile
      if (threadIdx.x == 0) S.n_tile[0] = 0;
      __syncthreads();
    };
    // stage, scan whenever the next slice might not fit (an upper bound that ignores the near-box filter: block-uniform
    // without looking at the counter)
    int *w_off = S.off[wave], *w_beg = S.beg[wave];
    int ub = 0;
    for (int r0 = 0; r0 < nrows; r0 += kWave) {
      const int r = r0 + lane;
      int b = 0, len = 0;
      if (r < nrows) {
        const int z = z0 + r / ny, y = y0 + r % ny;
        const int row = (z * g.dy + y) * g.dx;
        b = g.cell_start[row + x0];
        len = g.cell_start[row + x1 + 1] - b;
      }
      const int incl = snb_scan_dpp(len);
      const int total = __builtin_amdgcn_readlane(incl, 63);
      wave_lds_fence();
      w_off[lane] = incl - len;
      w_beg[lane] = b;
      wave_lds_fence();
      for (int tb = 0; tb < total; tb += T) {
        const int m = min(T, total - tb);
        if (ub + m > Cfg::kTileCap) { scan_tile(); ub = 0; }
        ub += m;
        const int sl = tb + wave * kWave + lane;
        const bool in = sl < total;
        const int slot = in ? sl : 0;
        int lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
          if (w_off[lo + step] <= slot) lo += step;
        const float4 cnd = g.pts[w_beg[lo] + (slot - w_off[lo])];
        const bool k = in && keep(cnd);
        float pv = 0.0f;
        if (k) pv = sf_intensity(pts[__float_as_int(cnd.w)].w);
        const unsigned long long mk = ballot(k);
        if (mk) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&S.n_tile[0], __popcll(mk));
          base = __builtin_amdgcn_readfirstlane(base);
          const int d = snb_mbcnt(mk, base);
          if (k) S.tile[d] = make_float4(cnd.x, cnd.y, cnd.z, pv);
        }
      }
    }
    scan_tile();
    // the waves' partial sums meet in the tile's memory
    float *red = reinterpret_cast<float *>(S.tile);
    int *redi = reinterpret_cast<int *>(S.tile);
#pragma unroll
    for (int s = 0; s < kCertScales; ++s) {
      red[(wave * Cfg::kVals + s) * 64 + lane] = num[s];
      red[(wave * Cfg::kVals + kCertScales + s) * 64 + lane] = den[s];
      redi[(wave * Cfg::kVals + 2 * kCertScales + s) * 64 + lane] = cnt[s];
    }
    redi[(wave * Cfg::kVals + 3 * kCertScales) * 64 + lane] = cnt_a;
    redi[(wave * Cfg::kVals + 3 * kCertScales + 1) * 64 + lane] = cnt_b;
    __syncthreads();
    if (wave < kCertScales) {                    // wave s: the response of scale s and its bound, per query
      const int s = wave;
      float N = 0.0f, D = 0.0f;
      int n = 0, kmax = 0;
#pragma unroll
      for (int w2 = 0; w2 < W; ++w2) {
        N += red[(w2 * Cfg::kVals + s) * 64 + lane];
        D += red[(w2 * Cfg::kVals + kCertScales + s) * 64 + lane];
        const int c2 = redi[(w2 * Cfg::kVals + 2 * kCertScales + s) * 64 + lane];
        n += c2;
        kmax = max(kmax, c2);
      }
      const float R = N / D;
      const float e = (float)(2 * n + 2 * (kmax + W) + 48) * 0x1p-24f * 1.015625f;
      S.resp[s][lane] = R;
      S.bres[s][lane] = R * e;
      S.ntot[s][lane] = n;
    } else {                                     // waves 6, 7: the two small-ball counts
      int n = 0;
#pragma unroll
      for (int w2 = 0; w2 < W; ++w2) n += redi[(w2 * Cfg::kVals + 3 * kCertScales + (wave - kCertScales)) * 64 + lane];
      S.ntot[wave][lane] = n;
    }
    __syncthreads();
    if (wave == 0 && live) {
      const int self = __float_as_int(q.w);
#pragma unroll
      for (int s = 0; s < kCertDog; ++s) {
        const float v = S.resp[s + 1][lane] - S.resp[s][lane];
        const float b = (S.bres[s + 1][lane] + S.bres[s][lane] + 0x1p-14f) * (1.0f + 0x1p-18f);
        dogv[(size_t)self * kCertDog + s] = v;
        dogb[(size_t)self * kCertDog + s] = b;
      }
      // the reject radius: the widest of five balls that holds at most 25 points (the point itself included) -- every
      // point inside it is one of the 25 nearest, whatever their order (k_sift_reject); 0: even the smallest holds more
      float r = 0.0f;
      if (S.ntot[kCertScales + 1][lane] <= kCertKnn) r = sc.TB;
      if (S.ntot[kCertScales][lane] <= kCertKnn) r = sc.TA;
#pragma unroll
      for (int s = 0; s < 3; ++s)
        if (S.ntot[s][lane] <= kCertKnn) r = sc.T[s];
      rlo2[self] = r;
      // ... and the search radius: the narrowest of the eight balls that holds at least 25 points -- the 25 nearest all lie
      // inside it (k_sift_extrema_one); +inf: even the 3 sigma_max ball holds fewer (the search doubles the radius)
      float ru = INFINITY;
#pragma unroll
      for (int s = kCertScales - 1; s >= 0; --s)
        if (S.ntot[s][lane] >= kCertKnn) ru = sc.T[s];
      if (S.ntot[kCertScales][lane] >= kCertKnn) ru = sc.TA;
      if (S.ntot[kCertScales + 1][lane] >= kCertKnn) ru = sc.TB;
      rup2[self] = ru;
    }
    __syncthreads();                               // (the next item's claim rewrites S.item, its staging the tile)
  }
}

// the next float below / above x (NaN and the infinity on that side stay; -0 and +0 count as one zero)
__device__ __forceinline__ float cert_pred(float x)
{
  const unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u || u == 0xff800000u) return x;          // NaN, -inf
  if ((u & 0x7fffffffu) == 0u) return __uint_as_float(0x80000001u);            // +-0 -> the smallest negative number
  return __uint_as_float((u & 0x80000000u) ? u + 1u : u - 1u);
}
__device__ __forceinline__ float cert_succ(float x) { return -cert_pred(-x); }

// [lo, hi] contains the CPU path's float; b == 0: the value IS the CPU path's float
__device__ __forceinline__ void cert_interval(float v, float b, float &lo, float &hi)
{
  if (b == 0.0f) { lo = v; hi = v; return; }
  lo = cert_pred(v - b);
  hi = cert_succ(v + b);
}

// What a point contributes to its neighbours' extremum tests, and its own contrast classes.
// findScaleSpaceExtrema asks of a minimum at scale s: val == min over the 25 nearest at s, val < min at s - 1, val < min at
// s + 1: the point's own scale with equality, the ADJACENT scales strictly.  A neighbour therefore spoils a minimum val
// when its DoG at s is below val, or its DoG at s - 1 or s + 1 is below OR EQUAL -- and "a <= v" is "pred(a) < v" (no float
// lies between a and the one below it): one value per (neighbour, scale) and one strict comparison per (query, neighbour),
// computed once per point instead of once per pair.  (Rounds 1 - 4 compared the adjacent scales with <=, >=: the same
// keypoints unless two DoG values tie exactly.)  With intervals there are two such values per (scale, side):
//   mnhi_s = min(hi_s, pred(min(hi_(s-1), hi_(s+1)))):  mnhi < lo_p  -> the neighbour CERTAINLY spoils the minimum
//   mnlo_s = min(lo_s, pred(min(lo_(s-1), lo_(s+1)))):  mnlo < hi_p  -> it POSSIBLY does
// (maxima mirrored: mxlo > hi_p certainly, mxhi > lo_p possibly).  With b = 0 the two coincide.
// dogx: [3][n] float4 = (mnhi1, mnhi2, mnhi3, mxlo1), (mxlo2, mxlo3, mnlo1, mnlo2), (mnlo3, mxhi1, mxhi2, mxhi3) -- the certain
// values first: k_sift_reject reads one row and a half;
// cls: bit s = |DoG(s + 1)| may reach the contrast (candidate), bit 3 + s = it certainly does (live).
// dogb == nullptr: dogv holds the CPU path's floats themselves (the octaves on the sorted lists, sift.hip).
__device__ __forceinline__ void cert_pack_point(const float *__restrict__ dogv, const float *__restrict__ dogb, int i, int n, float min_contrast,
                                                float4 *__restrict__ dogx, unsigned char *__restrict__ cls)
{
  float lo[kCertDog], hi[kCertDog];
#pragma unroll
  for (int s = 0; s < kCertDog; ++s) cert_interval(dogv[(size_t)i * kCertDog + s], dogb ? dogb[(size_t)i * kCertDog + s] : 0.0f, lo[s], hi[s]);
  float mnhi[3], mnlo[3], mxlo[3], mxhi[3];
  unsigned c = 0;
#pragma unroll
  for (int s = 1; s <= 3; ++s) {
    mnhi[s - 1] = fminf(hi[s], cert_pred(fminf(hi[s - 1], hi[s + 1])));
    mnlo[s - 1] = fminf(lo[s], cert_pred(fminf(lo[s - 1], lo[s + 1])));
    mxlo[s - 1] = fmaxf(lo[s], cert_succ(fmaxf(lo[s - 1], lo[s + 1])));
    mxhi[s - 1] = fmaxf(hi[s], cert_succ(fmaxf(hi[s - 1], hi[s + 1])));
    if (hi[s] >= min_contrast || lo[s] <= -min_contrast) c |= 1u << (s - 1);
    if (lo[s] >= min_contrast || hi[s] <= -min_contrast) c |= 8u << (s - 1);
  }
  dogx[i] = make_float4(mnhi[0], mnhi[1], mnhi[2], mxlo[0]);
  dogx[(size_t)n + i] = make_float4(mxlo[1], mxlo[2], mnlo[0], mnlo[1]);
  dogx[2 * (size_t)n + i] = make_float4(mnlo[2], mxhi[0], mxhi[1], mxhi[2]);
  cls[i] = (unsigned char)c;
}

// counters of one octave's certified run (device words, copied to the host at the octave's sync)
struct CertCounters {
  int n_marked;        // points that take the exact path
  int n_open;          // points whose test was open after the first pass
  int still_open;      // ... after the second (must be 0)
  int violations;      // marked points whose exact DoG lies outside [val* - B, val* + B] (must be 0)
  int pad[4];         // [0] (point, scale) candidates k_sift_reject decided, [1] candidates it left to the search
};

// The cheap way out for most candidates.  A (point, scale) that is no extremum usually has MANY neighbours below and above
// it; to decide "no" ONE certain violator per side is enough, provided it is one of the 25 nearest -- and every point inside
// the ball rlo2(p) is (the unsorted pass counted at most 25 points in it).  So: one unsorted look at the candidates within
// the item's largest reject radius, no keys, no counts: per (query, candidate) the distance, the ball test and six
// comparisons.  A scale with a certain violator on both sides loses its candidate bit in cls; what survives (the keypoints,
// the near misses and everything open: a few per cent) goes to k_sift_extrema_one.
struct SrCfg {
  static constexpr int kWaves = 4;
  static constexpr int kTileCap = 640;      // (an item is a few hundred candidates; five blocks per CU hide the staging round trips)
};
struct alignas(16) SrLds {
  float4 ta[SrCfg::kTileCap + 4];            // x, y, z, -
  float4 tb[SrCfg::kTileCap + 4];            // mnhi1, mnhi2, mnhi3, mxlo1
  float2 tc[SrCfg::kTileCap + 4];            // mxlo2, mxlo3
  int off[SrCfg::kWaves][64], beg[SrCfg::kWaves][64];
  float4 qpts[64];
  unsigned rej[64];
  int n_tile[2];
  int item;
};

__global__ void __launch_bounds__(64 * SrCfg::kWaves)
k_sift_reject(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g /* .w = original index */,
              const float *__restrict__ rlo2, const float *__restrict__ dogv, const float *__restrict__ dogb, const float4 *__restrict__ dogx, int n_pts,
              unsigned char *__restrict__ cls, SnbCtl *ctl, CertCounters *__restrict__ ctr)
{
  __shared__ SrLds S;
  constexpr int W = SrCfg::kWaves, T = 64 * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int epoch = 0;
  for (;; ++epoch) {
    if (threadIdx.x == 0) { S.item = snb_claim_item(ctl->item_ctr, n_items); S.n_tile[epoch & 1] = 0; }
    if (threadIdx.x < 64) S.rej[threadIdx.x] = 0u;
    __syncthreads();
    if (S.item < 0) break;
    const int2 it = items[S.item];
    const bool live = lane < it.y;
    const float4 qa = q_pts[it.x + (live ? lane : 0)];
    const int self = __float_as_int(qa.w);
    const unsigned c0 = live ? cls[self] : 0u;
    const float r2q = (live && (c0 & 7u)) ? rlo2[self] : 0.0f;      // (a point without a candidate scale asks for nothing)
    const float r2max = snb_max_f_dpp(r2q);
    if (r2max > 0.0f) {                            // block-uniform
      // own intervals; the box of the queries that ask, grown by the largest reject radius
      float lo_p[3], hi_p[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        lo_p[s] = 0.0f; hi_p[s] = 0.0f;
        if (live) cert_interval(dogv[(size_t)self * kCertDog + s + 1], dogb[(size_t)self * kCertDog + s + 1], lo_p[s], hi_p[s]);
      }
      const bool ask = r2q > 0.0f;
      const float lx = snb_min_f_dpp(ask ? qa.x : INFINITY), hx = snb_max_f_dpp(ask ? qa.x : -INFINITY);
      const float ly = snb_min_f_dpp(ask ? qa.y : INFINITY), hy = snb_max_f_dpp(ask ? qa.y : -INFINITY);
      const float lz = snb_min_f_dpp(ask ? qa.z : INFINITY), hz = snb_max_f_dpp(ask ? qa.z : -INFINITY);
      const float ri = sqrtf(r2max) * 1.0001f + 1e-4f;
      const int x0 = max(cell_floor(lx - ri, g.minx, g.inv), 0), x1 = min(cell_floor(hx + ri, g.minx, g.inv), g.dx - 1);
      const int y0 = max(cell_floor(ly - ri, g.miny, g.inv), 0), y1 = min(cell_floor(hy + ri, g.miny, g.inv), g.dy - 1);
      const int z0 = max(cell_floor(lz - ri, g.minz, g.inv), 0), z1 = min(cell_floor(hz + ri, g.minz, g.inv), g.dz - 1);
      const KeepNearBox keep{lx, hx, ly, hy, lz, hz, ri * ri};
      int *n_tile = &S.n_tile[epoch & 1];
      const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
      const int nrows = (x0 <= x1 && ny > 0 && nz > 0) ? ny * nz : 0;
      int *w_off = S.off[wave], *w_beg = S.beg[wave];
      for (int r0 = 0; r0 < nrows; r0 += kWave) {
        const int r = r0 + lane;
        int b = 0, len = 0;
        if (r < nrows) {
          const int z = z0 + r / ny, y = y0 + r % ny;
          const int row = (z * g.dy + y) * g.dx;
          b = g.cell_start[row + x0];
          len = g.cell_start[row + x1 + 1] - b;
        }
        const int incl = snb_scan_dpp(len);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        wave_lds_fence();
        w_off[lane] = incl - len;
        w_beg[lane] = b;
        wave_lds_fence();
        for (int t0 = wave * kWave; t0 < total; t0 += T) {
          const int sl = t0 + lane;
          const bool in = sl < total;
          const int slot = in ? sl : t0;
          int lo = 0;
#pragma unroll
          for (int step = 32; step > 0; step >>= 1)
            if (w_off[lo + step] <= slot) lo += step;
          const float4 cnd = g.pts[w_beg[lo] + (slot - w_off[lo])];
          const bool k = in && keep(cnd);
          float4 x0v = make_float4(0.f, 0.f, 0.f, 0.f), x1v = x0v;
          if (k) {
            const int o = __float_as_int(cnd.w);
            x0v = dogx[o]; x1v = dogx[(size_t)n_pts + o];
          }
          const unsigned long long m = ballot(k);
          if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(n_tile, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            const int d = snb_mbcnt(m, base);
            if (k && d < SrCfg::kTileCap) {
              S.ta[d] = make_float4(cnd.x, cnd.y, cnd.z, 0.0f);
              S.tb[d] = x0v;                                         // mnhi1..3, mxlo1
              S.tc[d] = make_float2(x1v.x, x1v.y);                   // mxlo2, mxlo3
            }
          }
        }
      }
      __syncthreads();
      const int nt = S.n_tile[epoch & 1];
      if (nt <= SrCfg::kTileCap) {                 // (a box the tile cannot hold: nothing is rejected here, the search decides)
        if (threadIdx.x < 4) { S.ta[nt + threadIdx.x] = make_float4(kSnbFar, kSnbFar, kSnbFar, 0.0f); }
        __syncthreads();
        const int n4 = (nt + 3) >> 2, per = (n4 + W - 1) / W;
        const int g0 = wave * per, g1 = min(n4, g0 + per);
        bool mn[3] = {false, false, false}, mx[3] = {false, false, false};
        for (int gi = g0; gi < g1; ++gi) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 a = S.ta[4 * gi + k], b = S.tb[4 * gi + k];
            const float2 c2 = S.tc[4 * gi + k];
            const bool in = dist2(qa.x, qa.y, qa.z, a.x, a.y, a.z) <= r2q;      // (the padding is far; r2q = 0 admits the point itself only)
            mn[0] = mn[0] || (in && b.x < lo_p[0]); mn[1] = mn[1] || (in && b.y < lo_p[1]); mn[2] = mn[2] || (in && b.z < lo_p[2]);
            mx[0] = mx[0] || (in && b.w > hi_p[0]); mx[1] = mx[1] || (in && c2.x > hi_p[1]); mx[2] = mx[2] || (in && c2.y > hi_p[2]);
          }
        }
        unsigned bits = 0;
#pragma unroll
        for (int s = 0; s < 3; ++s) bits |= (mn[s] ? 1u << s : 0u) | (mx[s] ? 8u << s : 0u);
        if (bits && ask) atomicOr(&S.rej[lane], bits);
      }
      __syncthreads();
      if (wave == 0 && live && (c0 & 7u)) {
        const unsigned b = S.rej[lane];
        unsigned c = c0;
        int gone = 0, kept = 0;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          if (!(c0 & (1u << s))) continue;
          if ((b & (1u << s)) && (b & (8u << s))) { c &= ~((1u << s) | (8u << s)); ++gone; }
          else ++kept;
        }
        if (c != c0) cls[self] = (unsigned char)c;
        if (gone) atomicAdd(&ctr->pad[0], gone);
        if (kept) atomicAdd(&ctr->pad[1], kept);
      }
    }
    __syncthreads();
  }
}

__global__ void k_sift_pack_iv(const float *__restrict__ dogv, const float *__restrict__ dogb, int n, float min_contrast,
                               float4 *__restrict__ dogx, unsigned char *__restrict__ cls)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cert_pack_point(dogv, dogb, i, n, min_contrast, dogx, cls);
}


// the marked points' exact DoG floats (`dog`, written by k_sift_dog_lds) replace their intervals; the bound is checked
__global__ void k_sift_pack_marked(const int *__restrict__ ids, const int *__restrict__ n_ids, const float *__restrict__ dog,
                                   float *__restrict__ dogv, float *__restrict__ dogb, int n, float min_contrast,
                                   float4 *__restrict__ dogx, unsigned char *__restrict__ cls, CertCounters *__restrict__ ctr)
{
  const int total = *n_ids;           // (known to the device only: a modest grid walks the list)
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < total; k += gridDim.x * blockDim.x) {
    const int i = ids[k];
    bool bad = false;
#pragma unroll
    for (int s = 0; s < kCertDog; ++s) {
      const float ex = dog[(size_t)i * kCertDog + s];
      float lo, hi;
      cert_interval(dogv[(size_t)i * kCertDog + s], dogb[(size_t)i * kCertDog + s], lo, hi);
      if (!(ex >= lo && ex <= hi)) bad = true;
      dogv[(size_t)i * kCertDog + s] = ex;
      dogb[(size_t)i * kCertDog + s] = 0.0f;
    }
    if (bad) atomicAdd(&ctr->violations, 1);
    cert_pack_point(dogv, dogb, i, n, min_contrast, dogx, cls);
  }
}

// the points with mark[i] != 0 as single-query work items: q[k] = (x, y, z, index), items[k] = (k, 1), ids[k] = index,
// ident[k] = k (k_sift_dog_lds' sub_items)
__global__ void k_sift_collect(const unsigned char *__restrict__ mark, const float4 *__restrict__ pts, int n, float4 *__restrict__ q,
                               int2 *__restrict__ items, int *__restrict__ ids, int *__restrict__ ident, int *__restrict__ count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !mark[i]) return;
  const int k = atomicAdd(count, 1);
  const float4 p = pts[i];
  q[k] = make_float4(p.x, p.y, p.z, __int_as_float(i));
  items[k] = make_int2(k, 1);
  ids[k] = i;
  ident[k] = k;
}

// the points with sel[i] & mask, as the id list of k_sift_extrema_one
__global__ void k_sift_collect_ids(const unsigned char *__restrict__ sel, unsigned mask, int n, int *__restrict__ ids, int *__restrict__ n_ids)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (sel[i] & mask)) ids[atomicAdd(n_ids, 1)] = i;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
    const unsigned long long w = ((unsigned long long)hi << 32) | lo;
    v = w < v ? w : v;
  }
  return v;
}

// findScaleSpaceExtrema on intervals for ONE point per wave, the lanes over the candidates.  What is left after
// k_sift_reject is a few thousand scattered points per octave (the keypoints themselves, near misses, open comparisons);
// compact runs of them do not exist any more, and a block that grows rings around a single point is a long chain of
// passes.  Here the radius is KNOWN: the unsorted pass counted at least 25 points inside rup2(p), so the 25 nearest all lie
// in that ball and one look at the cells that cover it is enough -- nearest certain / possible violator per (scale, side)
// as (distance, index) keys over the ball's points (every lane its share, one wave reduction), then the number of points
// closer than each.  Per side: a certain violator among the 25 nearest -> not an extremum; else a possible one among them ->
// OPEN; else an extremum.  A (point, scale) is decided when both sides say no, or the contrast is certain and a side says yes;
// what is open marks the point (open_p, need_exact) and, in a third look at the same candidates, every neighbour whose
// comparison on an open side is possible but not certain (need_exact).  kFinal: nothing may be open (counted).
// A point whose 3 sigma_max ball holds fewer than 25 points (rup2 = +inf: the border of a sparse cloud) first doubles its
// radius until the ball does.  No LDS tile: a candidate is read by one lane, once per pass.
// The octaves on the sorted lists (sift.hip) use the kFinal form on exact values for the few points whose list is shorter
// than 25: dogb == nullptr (every interval is a point, nothing can be open), rup2 == nullptr (every ball is grown).
template <bool kFinal>
__global__ void __launch_bounds__(256)
k_sift_extrema_one(const int *__restrict__ ids, const int *__restrict__ n_ids_dev, const float4 *__restrict__ pts /* original order */, GridView g /* .w = original index */,
                   const float *__restrict__ rup2, float r2_max /* the 3 sigma_max ball */, const float4 *__restrict__ dogx /* [3][n_pts] */, int n_pts,
                   const float *__restrict__ dogv,
                   const float *__restrict__ dogb, const unsigned char *__restrict__ cls, int *__restrict__ flags /* [n*3] */,
                   unsigned char *__restrict__ need_exact, unsigned char *__restrict__ open_p, CertCounters *__restrict__ ctr)
{
  __shared__ int s_off[4][65];
  __shared__ int s_beg[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_ids = *n_ids_dev;
  const int n_waves = (int)gridDim.x * 4;
  int *w_off = s_off[wave], *w_beg = s_beg[wave];
  for (int k = (int)blockIdx.x * 4 + wave; k < n_ids; k += n_waves) {
    const int self = ids[k];
    const float4 q = pts[self];
    float r2q = rup2 ? rup2[self] : INFINITY;
    const unsigned c = cls[self];
    const unsigned cand = c & 7u, live = (c >> 3) & 7u;
    if (!cand) continue;                         // wave-uniform
    if (!(r2q < INFINITY)) {                     // wave-uniform: grow the ball until it holds the 25 nearest
      const int kk0 = g.n < kCertKnn ? g.n : kCertKnn;
      float r2 = r2_max;
      for (int grow = 0; grow < 48; ++grow) {
        r2 *= 4.0f;
        const float rg = sqrtf(r2) * 1.0001f + 1e-4f;
        const int bx0 = max(cell_floor(q.x - rg, g.minx, g.inv), 0), bx1 = min(cell_floor(q.x + rg, g.minx, g.inv), g.dx - 1);
        const int by0 = max(cell_floor(q.y - rg, g.miny, g.inv), 0), by1 = min(cell_floor(q.y + rg, g.miny, g.inv), g.dy - 1);
        const int bz0 = max(cell_floor(q.z - rg, g.minz, g.inv), 0), bz1 = min(cell_floor(q.z + rg, g.minz, g.inv), g.dz - 1);
        if (bx0 == 0 && by0 == 0 && bz0 == 0 && bx1 == g.dx - 1 && by1 == g.dy - 1 && bz1 == g.dz - 1) { r2 = 3.0e38f; break; }   // the whole cloud
        int cnt = 0;
        const int bny = by1 - by0 + 1, brows = (bx0 <= bx1 && bny > 0 && bz1 >= bz0) ? bny * (bz1 - bz0 + 1) : 0;
        for (int r = 0; r < brows; ++r) {        // (rare points: one row at a time, the lanes over its span)
          const int z = bz0 + r / bny, y = by0 + r % bny;
          const int row = (z * g.dy + y) * g.dx;
          const int b = g.cell_start[row + bx0], e = g.cell_start[row + bx1 + 1];
          for (int j = b + lane; j < e; j += kWave) {
            const float4 cd = g.pts[j];
            cnt += dist2(q.x, q.y, q.z, cd.x, cd.y, cd.z) <= r2 ? 1 : 0;
          }
        }
        cnt = wave_sum(cnt);
        cnt = __shfl(cnt, 0, 64);
        if (cnt >= kk0) break;
      }
      r2q = r2;
    }
    float lo[kCertDog], hi[kCertDog];
#pragma unroll
    for (int s = 0; s < kCertDog; ++s) cert_interval(dogv[(size_t)self * kCertDog + s], dogb ? dogb[(size_t)self * kCertDog + s] : 0.0f, lo[s], hi[s]);
    float lo_p[3], hi_p[3];
    const unsigned long long key_self = (unsigned long long)(unsigned)self;
    unsigned long long vc_min[3], vp_min[3], vc_max[3], vp_max[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      lo_p[s] = lo[s + 1]; hi_p[s] = hi[s + 1];
      // the point itself at the adjacent scales (distance 0)
      vc_min[s] = cert_pred(fminf(hi[s], hi[s + 2])) < lo_p[s] ? key_self : ~0ull;
      vp_min[s] = cert_pred(fminf(lo[s], lo[s + 2])) < hi_p[s] ? key_self : ~0ull;
      vc_max[s] = cert_succ(fmaxf(lo[s], lo[s + 2])) > hi_p[s] ? key_self : ~0ull;
      vp_max[s] = cert_succ(fmaxf(hi[s], hi[s + 2])) > lo_p[s] ? key_self : ~0ull;
    }
    // the rows of cells that cover the ball, flattened (as wave_stream_box: headers one per lane, a scan, then every lane
    // finds the row of its slot by bisection)
    const float ri = sqrtf(r2q) * 1.0001f + 1e-4f;
    const int x0 = max(cell_floor(q.x - ri, g.minx, g.inv), 0), x1 = min(cell_floor(q.x + ri, g.minx, g.inv), g.dx - 1);
    const int y0 = max(cell_floor(q.y - ri, g.miny, g.inv), 0), y1 = min(cell_floor(q.y + ri, g.miny, g.inv), g.dy - 1);
    const int z0 = max(cell_floor(q.z - ri, g.minz, g.inv), 0), z1 = min(cell_floor(q.z + ri, g.minz, g.inv), g.dz - 1);
    const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    const int nrows = (x0 <= x1 && ny > 0 && nz > 0) ? ny * nz : 0;
    int cc_min[3] = {0, 0, 0}, cp_min[3] = {0, 0, 0}, cc_max[3] = {0, 0, 0}, cp_max[3] = {0, 0, 0}, cg = 0;
    unsigned open_min = 0, open_max = 0;
    // pass 0: violators; pass 1: counts; pass 2 (open points of the first run only): mark the open neighbours
    for (int pass = 0; pass < 3; ++pass) {
      for (int r0 = 0; r0 < nrows; r0 += kWave) {
        const int r = r0 + lane;
        int b = 0, len = 0;
        if (r < nrows) {
          const int z = z0 + r / ny, y = y0 + r % ny;
          const int row = (z * g.dy + y) * g.dx;
          b = g.cell_start[row + x0];
          len = g.cell_start[row + x1 + 1] - b;
        }
        const int incl = snb_scan_dpp(len);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        wave_lds_fence();
        w_off[lane] = incl - len;
        w_beg[lane] = b;
        if (lane == 0) w_off[64] = 0x7fffffff;
        wave_lds_fence();
        for (int t0 = 0; t0 < total; t0 += kWave) {
          const int slot = t0 + lane;
          if (slot >= total) continue;
          int lo_r = 0;
#pragma unroll
          for (int step = 32; step > 0; step >>= 1)
            if (w_off[lo_r + step] <= slot) lo_r += step;
          const float4 cd = g.pts[w_beg[lo_r] + (slot - w_off[lo_r])];
          const float d2 = dist2(q.x, q.y, q.z, cd.x, cd.y, cd.z);
          if (!(d2 <= r2q)) continue;
          const unsigned idx = __float_as_uint(cd.w);
          const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | idx;
          if (pass == 1) {
            ++cg;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
              cc_min[s] += key < vc_min[s] ? 1 : 0; cp_min[s] += key < vp_min[s] ? 1 : 0;
              cc_max[s] += key < vc_max[s] ? 1 : 0; cp_max[s] += key < vp_max[s] ? 1 : 0;
            }
            continue;
          }
          if (idx == (unsigned)self) continue;
          const float4 a = dogx[idx], bq = dogx[(size_t)n_pts + idx], dq = dogx[2 * (size_t)n_pts + idx];
          const float mnhi[3] = {a.x, a.y, a.z}, mxlo[3] = {a.w, bq.x, bq.y};
          const float mnlo[3] = {bq.z, bq.w, dq.x}, mxhi[3] = {dq.y, dq.z, dq.w};
          if (pass == 0) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
              if (mnhi[s] < lo_p[s] && key < vc_min[s]) vc_min[s] = key;
              if (mnlo[s] < hi_p[s] && key < vp_min[s]) vp_min[s] = key;
              if (mxlo[s] > hi_p[s] && key < vc_max[s]) vc_max[s] = key;
              if (mxhi[s] > lo_p[s] && key < vp_max[s]) vp_max[s] = key;
            }
          } else {
            bool hit = false;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
              if ((open_min & (1u << s)) && mnlo[s] < hi_p[s] && !(mnhi[s] < lo_p[s])) hit = true;
              if ((open_max & (1u << s)) && mxhi[s] > lo_p[s] && !(mxlo[s] > hi_p[s])) hit = true;
            }
            if (hit) need_exact[idx] = 1;
          }
        }
        wave_lds_fence();
      }
      if (pass == 0) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          vc_min[s] = wave_min_u64(vc_min[s]); vp_min[s] = wave_min_u64(vp_min[s]);
          vc_max[s] = wave_min_u64(vc_max[s]); vp_max[s] = wave_min_u64(vp_max[s]);
        }
        continue;
      }
      if (pass == 2) break;
      // after the counts: the decisions (wave-uniform)
      cg = wave_sum(cg);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        cc_min[s] = wave_sum(cc_min[s]); cp_min[s] = wave_sum(cp_min[s]);
        cc_max[s] = wave_sum(cc_max[s]); cp_max[s] = wave_sum(cp_max[s]);
      }
      cg = __shfl(cg, 0, 64);
      const int kk = g.n < kCertKnn ? g.n : kCertKnn;
      const bool proven = cg >= kk;              // (the unsorted pass counted them with the same test: always)
      bool any_open = !proven;
      int fl[3] = {0, 0, 0};
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (!(cand & (1u << s))) continue;
        const int ccn = __shfl(cc_min[s], 0, 64), cpn = __shfl(cp_min[s], 0, 64), ccx = __shfl(cc_max[s], 0, 64), cpx = __shfl(cp_max[s], 0, 64);
        // per side: 0 no (a certain violator among the 25 nearest), 2 open (a possible one among them), 1 yes
        const int st_min = (vc_min[s] != ~0ull && ccn < kk) ? 0 : ((vp_min[s] != ~0ull && cpn < kk) ? 2 : 1);
        const int st_max = (vc_max[s] != ~0ull && ccx < kk) ? 0 : ((vp_max[s] != ~0ull && cpx < kk) ? 2 : 1);
        const bool is_live = live & (1u << s);
        const bool no = st_min == 0 && st_max == 0, yes = is_live && (st_min == 1 || st_max == 1);
        fl[s] = yes ? 1 : 0;
        if (!no && !yes) {
          any_open = true;
          if (st_min == 2) open_min |= 1u << s;
          if (st_max == 2) open_max |= 1u << s;
        }
      }
      if (!any_open || kFinal) {
        if (lane == 0) {
#pragma unroll
          for (int s = 0; s < 3; ++s)
            if (cand & (1u << s)) flags[(size_t)self * 3 + s] = fl[s];
          if (any_open) atomicAdd(&ctr->still_open, 1);
        }
        break;
      }
      if (lane == 0) { need_exact[self] = 1; open_p[self] = 1; atomicAdd(&ctr->n_open, 1); }
      if (!(open_min | open_max)) break;         // only the contrast is open: no neighbour to mark
    }
  }
}

}  // namespace mm3d
