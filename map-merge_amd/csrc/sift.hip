// sift.hip -- detectKeypoints(SIFT) on gfx950 (K4 in SURVEY 2.2).
//
// R/src/features.cpp:45-62,85-96: pcl::SIFTKeypoint<PointXYZRGB, PointWithScale>,
// setScales(resolution, 3 octaves, 3 scales per octave), setMinimumContrast(threshold); the result
// is copied with pcl::copyPointCloud, i.e. only x,y,z survive (rgb = 0).
// Per octave: VoxelGrid(leaf = scale) of the previous octave's cloud -> DoG scale space over a
// radius search of 3*sigma_max -> extrema over the 25 nearest neighbours and 3 adjacent scales.
//
// Both kernels are wave-cooperative (device_util.hpp::wave_stream_box): a wave owns one compact
// patch of <= 64 points (Hilbert order of the octave cloud), streams the box of grid cells its
// lanes can reach through LDS with coalesced loads, and every lane filters the staged candidates
// by its own distance test.
//
// The 25-NN extremum test never materialises the 25 neighbours.  A point fails "minimum at scale
// s" iff some neighbour q among its 25 nearest has DoG(q, s-1|s|s+1) < val: so it suffices to find
// the NEAREST such violator (one min-reduction over (distance, index) keys) and count how many
// points are closer than it (>= 25 <=> the violator is not among the 25 nearest).
#include <atomic>
#include <cfloat>
#include <functional>
#include <type_traits>
#include <memory>

#include "sorted_nb.hpp"
#include "snb_lds.hpp"
#include "sift_cert.hpp"

namespace mm3d {

constexpr int kScales = 6;      // nr_scales_per_octave (3) + 3
constexpr int kDog = 5;
constexpr int kKnn = 25;

struct SiftScales {
  float sigma_sqr[kScales];
  float thr9[kScales];          // 9 * sigma_sqr
  float rcp[kScales];           // RN(1 / sigma_sqr), for the correctly rounded division by a constant (lm::fdiv_const)
};

__device__ __forceinline__ float intensity_of(float w)
{
  const unsigned c = __float_as_uint(w);
  const int r = (int)((c >> 16) & 255u), g = (int)((c >> 8) & 255u), b = (int)(c & 255u);
  // / 1000.0f correctly rounded without the division macro: exact for every one of the 255 001 possible
  // numerators (checked against exact rational arithmetic), lm::fdiv_const's five operations
  return lm::fdiv_const((float)(299 * r + 587 * g + 114 * b), 1000.0f, 0.001f);
}

// computeScaleSpace: Gaussian-weighted mean intensity at 6 scales -> 5 differences.
// SIFTKeypoint::computeScaleSpace sums "value * w" and "w" over the neighbours IN radiusSearch's ORDER
// ((distance, index); the loop even leaves with `break` at the first neighbour beyond 3 sigma), in float:
// the sums are chains, so every point's neighbour list is built in that order first (sorted_nb.hpp) as
// (d2, intensity) pairs, and the chains then run one per lane: 16 points per group, four lanes per point;
// the supports are nested (3 sigma grows with the scale), so the lanes of a point take the scales
// {5}, {4}, {3 then 2}, {1 then 0} -- about equal numbers of neighbours each.  w = expf(-0.5 d2 / sigma2)
// is glibc's expf restated (libm_exact.hpp, its 2^(i/32) table in LDS).
__global__ void __launch_bounds__(256)
k_sift_dog(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g /* .w = original index */,
           const float4 *__restrict__ pts /* original order: rgba */, float radius, float r2, SiftScales sc, SnScratch scr,
           float *__restrict__ dog /* [n][5] by original index */)
{
  __shared__ SnLds lds[4];
  __shared__ float resp[4][kSnG][kScales];
  __shared__ uint64_t s_tab[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 32) lm::exp2f_tab_copy(s_tab, threadIdx.x);
  __syncthreads();
  SnLds &L = lds[wave];
  const size_t slot = (size_t)blockIdx.x * 4 + wave;
  unsigned long long *tmp = scr.tmp + slot * kSnEntries;
  float2 *fin = (float2 *)scr.fin + slot * kSnEntries;
  const int n_units = sn_unit_count(scr, n_items);
  const int p = lane >> 2, sub = lane & 3;
  // this lane's scales: first, then (for sub 2, 3) a second, narrower one
  const int sA = sub == 0 ? 5 : (sub == 1 ? 4 : (sub == 2 ? 3 : 1));
  const int sB = sub == 2 ? 2 : (sub == 3 ? 0 : -1);
  const float sigA = sc.sigma_sqr[sA], thrA = sc.thr9[sA], rcpA = sc.rcp[sA];
  const float sigB = sB >= 0 ? sc.sigma_sqr[sB] : 1.0f, thrB = sB >= 0 ? sc.thr9[sB] : -1.0f, rcpB = sB >= 0 ? sc.rcp[sB] : 1.0f;
  for (;;) {
    const int unit = sn_claim_unit(scr.unit_ctr, n_units, lane);
    if (unit < 0) break;
    const int2 it = items[sn_unit_item(scr, unit)];
    int first = (unit & 3) * kSnG;
    int left = min(kSnG, it.y - first);
    while (left > 0) {
      const float4 q = q_pts[it.x + first + (p < left ? p : 0)];
      const int fit = sn_build_lists<float2>(g, L, q.x, q.y, q.z, left, radius, r2, pts, tmp, fin, scr.error, lane,
                                             [](float d2, unsigned, const float4 &pt) { return make_float2(d2, intensity_of(pt.w)); });
      SN_TICK(t_chain);
      {
        const bool mine = p < fit;
        const int base = mine ? L.list_off[p] : 0, m = mine ? L.list_off[p + 1] - base : 0;
        int s = sA, e = 0;
        float sig = sigA, thr = thrA, rcp = rcpA, num = 0.0f, den = 0.0f;
        bool active = mine;
        while (ballot(active)) {
          if (active) {
            // eight list entries are requested at a time (L2-resident scratch: one round trip per window);
            // past the end of the list they read as "beyond 3 sigma"
            float2 ent[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ent[u] = (e + u < m) ? fin[base + e + u] : make_float2(INFINITY, 0.0f);
            bool brk = false;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (!brk) {
                if (ent[u].x <= thr) {
                  // -0.5f * d2 / sigma_sqr with the division correctly rounded (where fdiv_const's remainders would
                  // underflow, |x| < 2^-96, expf is 1.0f whatever the quotient's last bit); x is in [-4.5, 0]
                  const float w = lm::expf_glibc_t<false>(lm::fdiv_const(-0.5f * ent[u].x, sig, rcp), [&](unsigned i) { return s_tab[i]; });
                  num = __fadd_rn(num, __fmul_rn(ent[u].y, w));
                  den = __fadd_rn(den, w);
                } else {
                  brk = true;
                }
              }
            }
            e += 8;
            if (brk) {                           // beyond 3 sigma (the CPU loop's break) or end of the list
              resp[wave][p][s] = num / den;
              if (s == sA && sB >= 0) { s = sB; sig = sigB; thr = thrB; rcp = rcpB; e = 0; num = 0.0f; den = 0.0f; }
              else active = false;
            }
          }
        }
      }
      SN_TOCK(5, t_chain);
      wave_lds_fence();
      if (lane < fit) {
        const float4 pq = q_pts[it.x + first + lane];
        float *o = dog + (size_t)__float_as_int(pq.w) * kDog;
        float prev = resp[wave][lane][0];
#pragma unroll
        for (int s = 1; s < kScales; ++s) {
          const float cur = resp[wave][lane][s];
          o[s - 1] = cur - prev;
          prev = cur;
        }
      }
      wave_lds_fence();
      first += fit;
      left -= fit;
    }
  }
}


// ---- computeScaleSpace on LDS-resident neighbour lists (snb_lds.hpp) ------------------------------------
// The same sums as k_sift_dog, bit for bit, with the lists as 16-bit tile slots in LDS and the candidates'
// intensities staged beside the tile.  Eight lanes per query (8 queries per wave) form two quads that take the
// scales {5, 2, 1} and {4, 3, 0}; lane `sub4` of a quad takes the list entries e = sub4 (mod 4) and computes
// their Gaussian weights for the quad's scales (all lanes busy: the supports are nested, a lane that walked one
// scale alone idled through a third of the steps), and the additions of a (query, scale) then run in list order
// through the quad: entry 4t, 4t+1, 4t+2, 4t+3 come from lanes 0..3 by DPP quad broadcasts, every lane of the
// quad carrying the same running sums.  An entry beyond 3 sigma (or past the end) contributes +0.0f, which
// changes no sum: the CPU loop's `break`.
// (Measured and dropped: one query at a time on the whole wave, the weights of 64 entries through LDS to twelve
// adding lanes -- no arena, four waves per SIMD for every octave, but three times the add instructions:
// 4.7 ms for the three octaves of a 500 k map against 3.4 ms.)
template <int J>
__device__ __forceinline__ float quad_bcast(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), J * 0x55, 0xf, 0xf, false));
}

// The LDS of a block is a budget split between the tile (20 B per staged candidate, shared) and the eight waves' arenas
// (2 B per list entry).  Round 4 moved it towards the arenas: a wave's consumer pass costs the same for three lists as for
// eight, and with the round-3 sizes (1792 / 1024 and 3584 / 2560) the arena, not the eight-query budget, ended most rounds
// (1.28, 1.42 and 1.86 rounds per item in the three octaves).  The tile's capacity was mostly unused (716 / 1246 / 1191
// candidates per item on average), but not everywhere: what a smaller tile cannot hold even in eight parts goes to the repair
// launch (SiftCfgDense below, ~0.2 - 0.3 ms each), and over the sixteen headline maps the split that wins is 2816 / 3328, not
// the 2304 / 4160 that is fastest on a map without dense spots (one stream, ms per step for octaves 1 + 2 + repairs: 41.8 with
// 3584 / 2560, 38.9 with 2816 / 3328, 39.5 with 2560 / 3840, 41.2 with 2304 / 4160 where 14 of 16 maps need a repair launch;
// profiles/r04_sift_config_ab.txt).  First octave: hit buffer 384 -> 256 (longer lists: distance bands), arena 1024 -> 1408:
// 1.23 -> 1.18 ms.
#ifndef MM3D_SIFT_STEPWISE
#define MM3D_SIFT_STEPWISE 0
#endif
#ifndef MM3D_SIFT_SMALL
#define MM3D_SIFT_SMALL 8, 1792, 1408, 256, 128
#endif
using SiftCfgSmall = SnbCfg<MM3D_SIFT_SMALL, true>;     // first octave: lists of ~110 (longer than 256: distance bands), 2 blocks of 8 waves per CU
#ifndef MM3D_SIFT_LARGE
#define MM3D_SIFT_LARGE 8, 2816, 3328, 768, 256
#endif
using SiftCfgLarge = SnbCfg<MM3D_SIFT_LARGE, true>;     // later octaves: lists of 300-900 (longer ones in bands), 1 block of 8 waves per CU
// The repair configuration: the items a launch of the two above could not hold (dense spots: a tile of more candidates than
// theirs even in eight parts) get the largest tile that fits a CU, one block per item, before anything is left to the lists in
// global memory (k_sift_dog: 0.46 ms per launch on the headline maps, against 0.1 - 0.2 for this one)
#ifndef MM3D_SIFT_DENSE
#define MM3D_SIFT_DENSE 8, 3584, 2560, 768, 256
#endif
using SiftCfgDense = SnbCfg<MM3D_SIFT_DENSE, true>;
// Later octaves (round 4): ONE block of SIXTEEN waves per CU around one tile.  A wave issues at most one instruction every
// ~6 cycles (scripts/micro/valu_rate.hip), so the eight waves of SiftCfgLarge left two thirds of every SIMD's issue slots
// empty; LDS is what limits the waves, so the per-wave share shrinks instead: four lists per round (16 lanes per query)
// instead of eight, a 512-entry hit buffer (longer lists in distance bands), 128 buckets.
#ifndef MM3D_SIFT_LARGE16
#define MM3D_SIFT_LARGE16 16, 2816, 1408, 512, 128
#endif
using SiftCfgLarge16 = SnbCfg<MM3D_SIFT_LARGE16, true>;

// kNormals: the launch also computes the surface normals of its queries (computeSurfaceNormals, R/src/features.cpp:168-179)
// from the SAME lists: with normal_radius <= 3 sigma_max the normals' neighbours of a point are the prefix d2 < nr2 of
// the list built here, in the same (distance, index) order, so the nine raw-moment chains of k_normals_lds (normals.hip)
// run over that prefix and the separate stage-and-sort launch of the normals disappears.  Same bits: the same entries
// in the same order through the same arithmetic.
// (second launch bound: waves per SIMD the registers must leave room for -- two blocks per CU with the small tile, one with
// the large; without it the fused variant took 162 VGPRs, one block per CU)
template <class Cfg, bool kNormals>
__global__ void __launch_bounds__(64 * Cfg::kWaves, (Cfg::kTileCap <= 2048 ? 2 : 1) * Cfg::kWaves / 4)
k_sift_dog_lds(const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items, GridView g /* .w = original index */,
               const float4 *__restrict__ pts /* original order: rgba */, float radius, float r2, SiftScales sc, SnbCtl *ctl,
               int *__restrict__ ov_items, float *__restrict__ dog /* [n][5] by original index */,
               int *__restrict__ knn /* [n][kKnn] by original index */, unsigned char *__restrict__ knn_ok /* [n], zeroed */,
               const int *__restrict__ sub_items, const int *__restrict__ sub_count, float nr2, float4 *__restrict__ nrm_out /* by original index */)
{
  __shared__ SnbLds<Cfg> S;
  __shared__ float resp[Cfg::kWaves][Cfg::kQ][kNormals ? 9 : kScales];      // scale responses, then (kNormals) the nine moment sums
  __shared__ int cnts[kNormals ? Cfg::kWaves : 1][Cfg::kQ];
  __shared__ uint64_t s_tab[32];
  constexpr int LPQ = Cfg::kLpq;
  static_assert(LPQ == 8 || LPQ == 16, "two or four quads per query");
  // LPQ 8: two quads per query take the scales {5, 2, 1} and {4, 3, 0}; LPQ 16: four quads take {5}, {4, 0}, {3, 1}, {2}
  // (the supports are nested: about equal numbers of list entries per quad either way)
  constexpr int NS = LPQ == 8 ? 3 : 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 32) lm::exp2f_tab_copy(s_tab, threadIdx.x);       // (snb_run's first barrier publishes it)
  const int sub4 = lane & 3, quad = (lane >> 2) & (LPQ / 4 - 1);
  // this quad's scales, widest first (-1: the quad has no scale in that slot)
  float sig[NS], thr[NS], rcp[NS];
  int ss[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    int sk;
    if (LPQ == 8) {
      const int a = k == 0 ? 5 : (k == 1 ? 2 : 1), b = k == 0 ? 4 : (k == 1 ? 3 : 0);
      sk = quad ? b : a;
    } else {
      sk = k == 0 ? (quad == 0 ? 5 : (quad == 1 ? 4 : (quad == 2 ? 3 : 2))) : (quad == 1 ? 0 : (quad == 2 ? 1 : -1));
    }
    ss[k] = sk;
    sig[k] = sk >= 0 ? sc.sigma_sqr[sk] : 1.0f;
    thr[k] = sk >= 0 ? sc.thr9[sk] : -1.0f;       // (no squared distance is <= -1: the slot never takes part)
    rcp[k] = sk >= 0 ? sc.rcp[sk] : 1.0f;
  }
  SnbWave<Cfg> &W = S.w[wave];
  snb_run<Cfg>(
      g, S, q_pts, items, n_items, radius, r2, ctl, ov_items,
      [&](const float4 &c) { return intensity_of(pts[__float_as_int(c.w)].w); },
      [&](int fit, const float4 &q, const float4 &pq) {
        const int p = lane / LPQ;
        const bool mine = p < fit;
        const int base = mine ? W.list_off[p] : 0, m = mine ? W.list_off[p + 1] - base : 0;
        float num[NS], den[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) { num[k] = 0.0f; den[k] = 0.0f; }
        // four quad steps (sixteen list entries of a query) per iteration: their slots, points and intensities
        // are requested together, the four weights of a scale are independent instruction streams (no divergent
        // branch: a lane outside 3 sigma computes a weight nobody uses), and only the additions are a chain
        for (int t0 = 0;; t0 += 4) {
          bool valid[4];
          unsigned slot[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = 4 * (t0 + j) + sub4;
            valid[j] = e < m;
            slot[j] = valid[j] ? (unsigned)W.arena[base + e] : 0u;
          }
          if (!ballot(valid[0])) break;          // wave-uniform: every list is exhausted
          float d2[4], val[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            val[j] = S.pay[slot[j]];
            d2[j] = dist2(q.x, q.y, q.z, S.tx[slot[j]], S.ty[slot[j]], S.tz[slot[j]]);
          }
#pragma unroll
          for (int k = 0; k < NS; ++k) {
            // the lists are sorted: if no lane's first entry of the four is inside 3 sigma, none of the others is
            if (ballot(valid[0] && d2[0] <= thr[k])) {
#if MM3D_SIFT_STEPWISE
              // (round 5) ... and the same holds step by step: quad step j of this iteration is worked only while some lane's
              // entry of that step is inside -- the weights of the steps behind a scale's support (a third of the evaluations
              // of the narrow scales, whose whole support is one or two iterations) are not computed at all
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const bool in = valid[j] && d2[j] <= thr[k];
                if (j > 0 && !ballot(in)) break;
                const float wj = lm::expf_glibc_t<false>(lm::fdiv_const(-0.5f * d2[j], sig[k], rcp[k]), [&](unsigned i) { return s_tab[i]; });
                const float w1 = in ? wj : 0.0f, vw1 = in ? __fmul_rn(val[j], wj) : 0.0f;
                num[k] = __fadd_rn(num[k], quad_bcast<0>(vw1)); den[k] = __fadd_rn(den[k], quad_bcast<0>(w1));
                num[k] = __fadd_rn(num[k], quad_bcast<1>(vw1)); den[k] = __fadd_rn(den[k], quad_bcast<1>(w1));
                num[k] = __fadd_rn(num[k], quad_bcast<2>(vw1)); den[k] = __fadd_rn(den[k], quad_bcast<2>(w1));
                num[k] = __fadd_rn(num[k], quad_bcast<3>(vw1)); den[k] = __fadd_rn(den[k], quad_bcast<3>(w1));
              }
#else
              float w[4], vw[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                // -0.5f * d2 / sigma_sqr, the division correctly rounded (k_sift_dog has the argument)
                const float wj = lm::expf_glibc_t<false>(lm::fdiv_const(-0.5f * d2[j], sig[k], rcp[k]), [&](unsigned i) { return s_tab[i]; });
                w[j] = (valid[j] && d2[j] <= thr[k]) ? wj : 0.0f;
                vw[j] = (valid[j] && d2[j] <= thr[k]) ? __fmul_rn(val[j], wj) : 0.0f;
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                num[k] = __fadd_rn(num[k], quad_bcast<0>(vw[j])); den[k] = __fadd_rn(den[k], quad_bcast<0>(w[j]));
                num[k] = __fadd_rn(num[k], quad_bcast<1>(vw[j])); den[k] = __fadd_rn(den[k], quad_bcast<1>(w[j]));
                num[k] = __fadd_rn(num[k], quad_bcast<2>(vw[j])); den[k] = __fadd_rn(den[k], quad_bcast<2>(w[j]));
                num[k] = __fadd_rn(num[k], quad_bcast<3>(vw[j])); den[k] = __fadd_rn(den[k], quad_bcast<3>(w[j]));
              }
#endif
            }
          }
        }
        if (mine && sub4 == 0) {
#pragma unroll
          for (int k = 0; k < NS; ++k)
            if (ss[k] >= 0) resp[wave][p][ss[k]] = num[k] / den[k];
        }
        // The list's first 25 entries ARE the point's 25 nearest neighbours in nearestKSearch's (distance, index)
        // order whenever the ball holds that many: findScaleSpaceExtrema (k_sift_extrema_knn) reads them back
        // instead of searching again.
        // A shorter list is left behind as well, with its length: its entries are the point's m nearest, and a
        // violator among them already decides the test (only a point that is still an extremum candidate after
        // its m < 25 nearest has to search for the others).  0 stays for a point whose list was not built here.
        if (knn && mine && m > 0) {                   // (knn == nullptr: the certified path's exact launch wants the DoG only)
          const int self_q = __float_as_int(q.w);
          const int keep = m < kKnn ? m : kKnn;
          int *row = knn + (size_t)self_q * kKnn;
          for (int e = lane & (LPQ - 1); e < keep; e += LPQ) row[e] = (int)S.tw[W.arena[base + e]];
          if ((lane & (LPQ - 1)) == 0) knn_ok[self_q] = (unsigned char)keep;
        }
        wave_lds_fence();
        if (lane < fit) {
          float *o = dog + (size_t)__float_as_int(pq.w) * kDog;
          float prev = resp[wave][lane][0];
#pragma unroll
          for (int s = 1; s < kScales; ++s) {
            const float cur = resp[wave][lane][s];
            o[s - 1] = cur - prev;
            prev = cur;
          }
        }
        wave_lds_fence();
        if (kNormals) {
          // the normals' neighbours: the list's prefix with d2 < nr2 (the list ascends in d2; a lane looks at every eighth
          // entry and stops at its first outside, the eight lanes of the query add their counts)
          const int sub = lane & (LPQ - 1);
          int mn = 0;
          for (int e = sub; e < m; e += LPQ) {
            const unsigned sl = W.arena[base + e];
            if (!(dist2(q.x, q.y, q.z, S.tx[sl], S.ty[sl], S.tz[sl]) < nr2)) break;
            ++mn;
          }
          mn += __shfl_xor(mn, 1, 64);
          mn += __shfl_xor(mn, 2, 64);
          mn += __shfl_xor(mn, 4, 64);
          if (LPQ == 16) mn += __shfl_xor(mn, 8, 64);
          if (mine && sub < 8) {                   // (eight accumulators: with sixteen lanes per query the upper eight watch)
            // k_normals_lds' chains, bit for bit: lane sub owns accumulator sub of {xx, xy, xz, yy, yz, zz, x, y}, every lane sums z
            const float *up = sub <= 2 ? S.tx : (sub <= 4 ? S.ty : (sub == 5 ? S.tz : (sub == 6 ? S.tx : S.ty)));
            const float *vp = sub == 0 ? S.tx : ((sub == 1 || sub == 3) ? S.ty : S.tz);
            const bool v1 = sub >= 6;
            float a0 = 0.f, a2 = 0.f;
            int e0 = 0;
            for (; e0 + 4 <= mn; e0 += 4) {
              unsigned sl[4];
              float fu[4], fv[4], fz[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) sl[u] = W.arena[base + e0 + u];
#pragma unroll
              for (int u = 0; u < 4; ++u) { fu[u] = up[sl[u]]; fv[u] = vp[sl[u]]; fz[u] = S.tz[sl[u]]; }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                a0 = __fadd_rn(a0, __fmul_rn(fu[u], v1 ? 1.0f : fv[u]));
                a2 = __fadd_rn(a2, fz[u]);
              }
            }
            for (; e0 < mn; ++e0) {
              const unsigned sl = W.arena[base + e0];
              a0 = __fadd_rn(a0, __fmul_rn(up[sl], v1 ? 1.0f : vp[sl]));
              a2 = __fadd_rn(a2, S.tz[sl]);
            }
            resp[wave][p][sub] = a0;
            if (sub == 0) { resp[wave][p][8] = a2; cnts[wave][p] = mn; }
          }
          wave_lds_fence();
          if (lane < fit) nrm_out[__float_as_int(pq.w)] = normal_from_moments(resp[wave][lane], cnts[wave][lane], pq);
          wave_lds_fence();
        }
      },
      sub_items, sub_count);
}

// one octave's scale space on the LDS path.  What it could not hold (a dense spot: normally nothing) is counted in
// *h_overflow -- pinned host memory, valid after the stream's next sync -- and worked by fallback(), the launch over
// those items with the lists in global memory (sorted_nb.hpp).  The caller only launches it when the count says so:
// an empty launch of that kernel still waits for LDS the scale-space kernels of other streams hold (0.7 ms of stream
// time per octave on the 16-stream bench), and the hardware queue behind it waits with it.
struct SiftDogPending {
  int *h_overflow = nullptr;
  std::function<void(int)> fallback;       // argument: *h_overflow
};

// nrm != nullptr: the launches also write the normals of the cloud's points (radius nrad, nr2 = float(nrad^2) <= r2) -- the
// fused first octave of detect_keypoints_sift
template <class Cfg>
static SiftDogPending sift_dog_octave(Context *c, int oct, const mm3d_cloud *cur, const Grid &gr, int n_items, float max_radius, float r2,
                                      const SiftScales &sc, float *dog, int *knn, unsigned char *knn_ok, double nrad = 0.0, float4 *nrm = nullptr)
{
  const float nr2 = (float)(nrad * nrad);      // KdTreeFLANN::radiusSearch: float(radius * radius), as compute_normals
  const size_t extra_lds = sizeof(float) * 64 * (nrm ? 9 : kScales) + (nrm ? sizeof(int) * 64 : 0) + 256;
  auto sl = std::make_shared<SnbLaunch<Cfg>>(c, n_items, extra_lds);
  static const bool per_octave_names = getenv("MM3D_SNB_DEBUG") != nullptr;    // (read once, not per launch)
  static const char *const oct_names[4] = {"sift_dog_oct0", "sift_dog_oct1", "sift_dog_oct2", "sift_dog_oct3+"};
  // (one profile name for the three octaves, the fused first one included: the counters under profiles/ are per kernel symbol)
  const char *name = per_octave_names ? oct_names[std::min(oct, 3)] : "sift_dog";
  // algorithmic bytes: 36 B per point (SURVEY 8d's scale-space figure) + the normals' 28 B when they come out of the same launch
  bool launched = false;
  if constexpr (Cfg::kLpq == 8) {                // (the fused variant exists for the eight-wave configurations only)
    if (nrm) {
      MM3D_LAUNCH(c, name, gr.n * (36.0 + 28.0), (k_sift_dog_lds<Cfg, true>), dim3(sl->blocks), dim3(64 * Cfg::kWaves), 0, (const float4 *)cur->hil_pts.get(),
                  (const int2 *)cur->wave_items.get(), n_items, gr.view(), (const float4 *)cur->pts.get(), max_radius, r2, sc, sl->ctl_dev(),
                  sl->ov_items.get(), dog, knn, knn_ok, (const int *)nullptr, (const int *)nullptr, nr2, nrm);
      launched = true;
    }
  }
  if (nrm && !launched) throw Error(MM3D_EINVAL, "sift_dog_octave: fused normals asked of a configuration that has none");
  if (!launched)
    MM3D_LAUNCH(c, name, gr.n * 36.0, (k_sift_dog_lds<Cfg, false>), dim3(sl->blocks), dim3(64 * Cfg::kWaves), 0, (const float4 *)cur->hil_pts.get(),
                (const int2 *)cur->wave_items.get(), n_items, gr.view(), (const float4 *)cur->pts.get(), max_radius, r2, sc, sl->ctl_dev(),
                sl->ov_items.get(), dog, knn, knn_ok, (const int *)nullptr, (const int *)nullptr, 0.0f, (float4 *)nullptr);
  SiftDogPending pend;
  pend.h_overflow = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(pend.h_overflow, &sl->ctl_dev()->ov_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  const Grid *grp = &gr;
  pend.fallback = [c, cur, grp, n_items, max_radius, r2, sc, dog, knn, knn_ok, sl, nrad, nr2, nrm](int n_overflow) {
    SnbCtl *ctl = sl->ctl_dev();
    int *ov = sl->ov_items.get();
    // a few overflow items (dense spots of an outdoor map) go through the repair configuration -- one block per item, the
    // largest tile a CU holds -- before anything is left to the global-memory lists; a cloud that is dense everywhere (thousands of items) goes to those directly (measured on
    // 8 x 2 M indoor points: the large configuration is no faster per neighbour there, and it runs one block per CU)
    std::shared_ptr<SnbLaunch<SiftCfgDense>> sl2;
    static const int dense_max = [] { const char *e = getenv("MM3D_SIFT_DENSE_MAX"); return e ? atoi(e) : 256; }();   // A/B knob
    if (!std::is_same<Cfg, SiftCfgDense>::value && n_overflow <= dense_max) {
      const size_t extra2 = sizeof(float) * 64 * (nrm ? 9 : kScales) + (nrm ? sizeof(int) * 64 : 0) + 256;
      sl2 = std::make_shared<SnbLaunch<SiftCfgDense>>(c, n_items, extra2);
      const dim3 grid2(std::min(sl2->blocks, (unsigned)n_overflow)), block2(64 * SiftCfgDense::kWaves);
      if (nrm)
        MM3D_LAUNCH(c, "sift_dog_dense", 0.0, (k_sift_dog_lds<SiftCfgDense, true>), grid2, block2, 0,
                    (const float4 *)cur->hil_pts.get(), (const int2 *)cur->wave_items.get(), n_items, grp->view(), (const float4 *)cur->pts.get(),
                    max_radius, r2, sc, sl2->ctl_dev(), sl2->ov_items.get(), dog, knn, knn_ok, (const int *)sl->ov_items.get(),
                    (const int *)&sl->ctl_dev()->ov_count, nr2, nrm);
      else
        MM3D_LAUNCH(c, "sift_dog_dense", 0.0, (k_sift_dog_lds<SiftCfgDense, false>), grid2, block2, 0,
                    (const float4 *)cur->hil_pts.get(), (const int2 *)cur->wave_items.get(), n_items, grp->view(), (const float4 *)cur->pts.get(),
                    max_radius, r2, sc, sl2->ctl_dev(), sl2->ov_items.get(), dog, knn, knn_ok, (const int *)sl->ov_items.get(),
                    (const int *)&sl->ctl_dev()->ov_count, 0.0f, (float4 *)nullptr);
      ctl = sl2->ctl_dev();
      ov = sl2->ov_items.get();
    }
    SnLaunch<float2> sn(c, n_overflow * 4, cur->n, 4, 1024u);        // (at most n_overflow items are left: a block each)
    SnScratch scr{sn.tmp.get(), sn.fin.get(), ctl->fb_ctr, &ctl->error, ov, &ctl->ov_count};
    MM3D_LAUNCH(c, "sift_dog_big", 0.0, k_sift_dog, dim3(sn.blocks), dim3(256), 0, (const float4 *)cur->hil_pts.get(),
                (const int2 *)cur->wave_items.get(), n_items, grp->view(), (const float4 *)cur->pts.get(), max_radius, r2, sc, scr, dog);
    int *he = (int *)c->pin(64);
    MM3D_HIP(hipMemcpyAsync(he, &ctl->error, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    c->check_later(he, MM3D_EUNSUPPORTED, "detectKeypoints(SIFT): a point has more than 16384 neighbours within 3 sigma");
    // the same items' normals, with the lists in global memory like their scale space (an item the large LDS configuration
    // has worked gets its normals there; what is left on `ov` did not)
    if (nrm) normals_of_items(c, cur, *grp, nrad, ov, &ctl->ov_count, n_overflow, nrm);
  };
  return pend;
}


#ifdef MM3D_SN_STATS
extern "C" void mm3d_debug_sn_stats_sift(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sn_stats), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sn_stats), z, sizeof(z)); }
}
#endif

// findScaleSpaceExtrema from the neighbours the scale-space kernel left behind (knn, knn_ok = how many: 25, or
// fewer when the 3 sigma_max ball holds fewer, or 0 when the point's list was not built there): one thread per point
// of the octave, in Hilbert order.  A point is a minimum at scale s iff none of its 25 nearest (itself included)
// has a DoG below its own at s-1, s or s+1; a maximum likewise (dogx: what a neighbour contributes per scale and side,
// sift_cert.hpp::cert_pack_point on the exact values; its first row and a half are read here).  With fewer than 25
// neighbours at hand a violator among them still decides "no"; a point they leave undecided goes onto the list of
// k_sift_extrema_one, which searches the grid for its 25 nearest (borders and sparse places: a handful per map).
__global__ void __launch_bounds__(256)
k_sift_extrema_knn(const float4 *__restrict__ hil, int nh, int n, const float *__restrict__ dog, const float4 *__restrict__ dogx,
                   const int *__restrict__ knn, const unsigned char *__restrict__ knn_ok, float min_contrast, int *__restrict__ flags /* [n*3] */,
                   int *__restrict__ search_ids /* [n] */, int *__restrict__ n_search /* zeroed */)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nh) return;
  const int self = __float_as_int(hil[j].w);
  float v[3];
  unsigned live = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    v[s] = dog[(size_t)self * kDog + s + 1];
    if (fabsf(v[s]) >= min_contrast) live |= 1u << s;
  }
  if (!live) return;
  const int cnt = knn_ok[self];
  bool is_min[3] = {true, true, true}, is_max[3] = {true, true, true};
  const int *row = knn + (size_t)self * kKnn;
  for (int e = 0; e < cnt; ++e) {
    const int nb = row[e];
    const float4 a = dogx[nb], b = dogx[n + nb];
    const float mn[3] = {a.x, a.y, a.z};
    const float mx[3] = {a.w, b.x, b.y};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      is_min[s] = is_min[s] && !(mn[s] < v[s]);
      is_max[s] = is_max[s] && !(mx[s] > v[s]);
    }
  }
  if (cnt < kKnn) {
    bool open = false;                         // still a candidate at some scale that passes the contrast test
#pragma unroll
    for (int s = 0; s < 3; ++s) open = open || ((live & (1u << s)) && (is_min[s] || is_max[s]));
    if (open) { search_ids[atomicAdd(n_search, 1)] = self; return; }
  }
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (live & (1u << s)) flags[(size_t)self * 3 + s] = (is_min[s] || is_max[s]) ? 1 : 0;
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

// ---- the certified octave (sift_cert.hpp) -----------------------------------------------------------------
// process-wide statistics (mm3d_debug_sift_cert_stats): 0 octaves on the certified path, 1 their points, 2 points that took
// the exact path, 3 points whose test was open after the first pass, 4 octaves sent back to the sorted-list path,
// 5 bound violations, 6 points still open after the second pass, 7 items the fast kernel's tile could not hold
static std::atomic<long long> g_cert_stats[8];
static std::atomic<int> g_cert_min_override{-1};
void debug_sift_cert_min(int n) { g_cert_min_override.store(n); }

void debug_sift_cert_stats(long long *out, int reset)
{
  for (int i = 0; i < 8; ++i) { out[i] = g_cert_stats[i].load(); if (reset) g_cert_stats[i].store(0); }
}

static SfScales cert_scales(const SiftScales &sc, float r2)
{
  SfScales fs;
  const float r2p = nextafterf(r2, 0.0f);          // radiusSearch keeps d2 < r2: d2 <= pred(r2)
  for (int i = 0; i < kScales; ++i) {
    fs.T[i] = fminf(sc.thr9[i], r2p);
    fs.c[i] = (float)(-0.5 * 1.4426950408889634074 / (double)sc.sigma_sqr[i]);
  }
  fs.TA = 0.5f * fs.T[0];
  fs.TB = 0.25f * fs.T[0];
  return fs;
}

template <class Cfg>
struct SfLaunch {
  DevBuf<int> ctl, ov_items;
  unsigned blocks = 0;
  SfLaunch(Context *c, int n_items)
  {
    const unsigned cap = (unsigned)snb_cu_count(c->device) * (unsigned)MM3D_SF_BLOCKS;
    blocks = (unsigned)std::max(1, std::min<int>(n_items, (int)cap));
    ctl = DevBuf<int>(c, sizeof(SnbCtl) / sizeof(int));
    ov_items = DevBuf<int>(c, (size_t)std::max(n_items, 1));
    MM3D_HIP(hipMemsetAsync(ctl.get(), 0, sizeof(SnbCtl), c->stream));
  }
  SnbCtl *ctl_dev() const { return reinterpret_cast<SnbCtl *>(ctl.get()); }
};

static void sift_dog_fast_launch(Context *c, const mm3d_cloud *cur, const Grid &gr, int n_items, float max_radius, const SfScales &fs,
                                 SfLaunch<SfCfgDefault> &fl, float *dogv, float *dogb, float *rlo2, float *rup2, unsigned char *need_exact)
{
  MM3D_LAUNCH(c, "sift_dog_fast", gr.n * 36.0, (k_sift_dog_fast<SfCfgDefault>), dim3(fl.blocks), dim3(64 * SfCfgDefault::kWaves), 0,
              (const float4 *)cur->hil_pts.get(), (const int2 *)cur->wave_items.get(), n_items, gr.view(), (const float4 *)cur->pts.get(), max_radius, fs,
              fl.ctl_dev(), fl.ov_items.get(), dogv, dogb, rlo2, rup2, need_exact);
}

// One octave's keypoint flags by certified decisions.  flags ([3 n + 1] ints, zeroed) receives them; returns false when the
// octave has to be taken by the sorted-list path instead (flags may then hold partial results: the caller clears them).
static bool sift_octave_certified(Context *c, const mm3d_cloud *cur, const Grid &gr, int n_items, float max_radius, float r2,
                                  const SiftScales &sc, float min_contrast, int *flags, int *pos, int *h_keypoints /* pinned */)
{
  const int n = (int)cur->n;
  const SfScales fs = cert_scales(sc, r2);
  DevBuf<float> dogv(c, (size_t)n * kDog), dogb(c, (size_t)n * kDog), dog(c, (size_t)n * kDog);
  DevBuf<float4> dogx(c, (size_t)n * 3);
  DevBuf<unsigned char> cls(c, (size_t)n);
  // zeroed: CertCounters | list lengths: marked, -, candidates, wide candidates, open, wide open, -, - | need_exact [n] | open_p [n]
  const size_t z_head = sizeof(CertCounters) + 8 * sizeof(int);
  DevBuf<unsigned char> zeroed(c, z_head + 2 * (size_t)n);
  MM3D_HIP(hipMemsetAsync(zeroed.get(), 0, z_head + 2 * (size_t)n, c->stream));
  CertCounters *ctr = reinterpret_cast<CertCounters *>(zeroed.get());
  int *n_marked_dev = reinterpret_cast<int *>(zeroed.get() + sizeof(CertCounters));
  unsigned char *need_exact = zeroed.get() + z_head, *open_p = need_exact + n;
  // 1. the unsorted pass
  SfLaunch<SfCfgDefault> fl(c, n_items);
  DevBuf<float> rlo2(c, (size_t)n);
  DevBuf<float> rup2(c, (size_t)n);
  sift_dog_fast_launch(c, cur, gr, n_items, max_radius, fs, fl, dogv.get(), dogb.get(), rlo2.get(), rup2.get(), need_exact);
  // 2. intervals and contrast classes
  MM3D_LAUNCH(c, "sift_pack", n * 100.0, k_sift_pack_iv, dim3(div_up(n, 256)), dim3(256), 0, (const float *)dogv.get(), (const float *)dogb.get(), n,
              min_contrast, dogx.get(), cls.get());
  // 2b. most candidates leave here: a certain violator on both sides inside the ball that holds at most 25 points
  static const bool use_reject = [] { const char *e = getenv("MM3D_SIFT_NO_REJECT"); return !(e && atoi(e)); }();   // A/B knob
  DevBuf<int> rctl(c, sizeof(SnbCtl) / sizeof(int));
  if (use_reject) {
    MM3D_HIP(hipMemsetAsync(rctl.get(), 0, sizeof(SnbCtl), c->stream));
    const unsigned rblocks = (unsigned)std::max(1, std::min<int>(n_items, snb_cu_count(c->device) * 5));
    MM3D_LAUNCH(c, "sift_reject", 0.0, k_sift_reject, dim3(rblocks), dim3(64 * SrCfg::kWaves), 0, (const float4 *)cur->hil_pts.get(),
                (const int2 *)cur->wave_items.get(), n_items, gr.view(), (const float *)rlo2.get(), (const float *)dogv.get(), (const float *)dogb.get(),
                (const float4 *)dogx.get(), n, cls.get(), reinterpret_cast<SnbCtl *>(rctl.get()), ctr);
  }
  // 3. what is left: one wave per point, inside the ball the unsorted pass counted the 25 nearest in
  DevBuf<int> cids(c, (size_t)n), oids2(c, (size_t)n);
  int *n_cids = n_marked_dev + 2, *n_oids = n_marked_dev + 4;
  const unsigned one_blocks = (unsigned)snb_cu_count(c->device) * 4u;
  MM3D_LAUNCH(c, "sift_collect", n * 5.0, k_sift_collect_ids, dim3(div_up(n, 256)), dim3(256), 0, (const unsigned char *)cls.get(), 7u, n, cids.get(), n_cids);
  MM3D_LAUNCH(c, "sift_extrema_one", 0.0, k_sift_extrema_one<false>, dim3(one_blocks), dim3(256), 0, (const int *)cids.get(), (const int *)n_cids,
              (const float4 *)cur->pts.get(), gr.view(), (const float *)rup2.get(), fs.T[kScales - 1], (const float4 *)dogx.get(), n, (const float *)dogv.get(),
              (const float *)dogb.get(), (const unsigned char *)cls.get(), flags, need_exact, open_p, ctr);
  // 4. the marked points: exact DoG floats from the sorted lists (single-query items, the number known to the device only)
  DevBuf<float4> mq(c, (size_t)n);
  DevBuf<int2> mitems(c, (size_t)n);
  DevBuf<int> mids(c, (size_t)n), mident(c, (size_t)n);
  MM3D_LAUNCH(c, "sift_collect", n * 1.0, k_sift_collect, dim3(div_up(n, 256)), dim3(256), 0, (const unsigned char *)need_exact,
              (const float4 *)cur->pts.get(), n, mq.get(), mitems.get(), mids.get(), mident.get(), n_marked_dev);
  const size_t extra_lds = sizeof(float) * 64 * kScales + 256;
  SnbLaunch<SiftCfgLarge> sl(c, n, extra_lds);
  // (a few hundred single-query items, a block each: 64 blocks of four items measured slower inside the step -- 410 against 270 us per
  // launch -- although every block of this configuration waits for a whole CU's LDS)
  const unsigned exact_blocks = std::min(sl.blocks, 256u);
  MM3D_LAUNCH(c, "sift_dog_exact", 0.0, (k_sift_dog_lds<SiftCfgLarge, false>), dim3(exact_blocks), dim3(64 * SiftCfgLarge::kWaves), 0,
              (const float4 *)mq.get(), (const int2 *)mitems.get(), 0, gr.view(), (const float4 *)cur->pts.get(), max_radius, r2, sc, sl.ctl_dev(),
              sl.ov_items.get(), dog.get(), (int *)nullptr, (unsigned char *)nullptr, (const int *)mident.get(), (const int *)n_marked_dev, 0.0f,
              (float4 *)nullptr);
  MM3D_LAUNCH(c, "sift_pack", 0.0, k_sift_pack_marked, dim3(std::min(div_up(n, 256), 64u)), dim3(256), 0, (const int *)mids.get(), (const int *)n_marked_dev,
              (const float *)dog.get(), dogv.get(), dogb.get(), n, min_contrast, dogx.get(), cls.get(), ctr);
  // 5. the open points again, on the collapsed intervals
  MM3D_LAUNCH(c, "sift_collect", n * 1.0, k_sift_collect_ids, dim3(div_up(n, 256)), dim3(256), 0, (const unsigned char *)open_p, 1u, n, oids2.get(), n_oids);
  MM3D_LAUNCH(c, "sift_extrema_fin", 0.0, k_sift_extrema_one<true>, dim3(64), dim3(256), 0, (const int *)oids2.get(), (const int *)n_oids,
              (const float4 *)cur->pts.get(), gr.view(), (const float *)rup2.get(), fs.T[kScales - 1], (const float4 *)dogx.get(), n, (const float *)dogv.get(),
              (const float *)dogb.get(), (const unsigned char *)cls.get(), flags, need_exact, open_p, ctr);
  int *h = (int *)c->pin(64);      // [0..7] CertCounters, [8] marked, [9] exact launch's overflow, [10] fast launch's overflow items
  MM3D_HIP(hipMemcpyAsync(h, ctr, sizeof(CertCounters), hipMemcpyDeviceToHost, c->stream));
  MM3D_HIP(hipMemcpyAsync(h + 8, n_marked_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MM3D_HIP(hipMemcpyAsync(h + 9, &sl.ctl_dev()->ov_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MM3D_HIP(hipMemcpyAsync(h + 10, &fl.ctl_dev()->ov_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  // the keypoint positions and their number ride on the same wait (thrown away when the octave is sent back)
  exclusive_scan_int(c, flags, pos, (size_t)n * 3 + 1);
  MM3D_HIP(hipMemcpyAsync(h_keypoints, pos + (size_t)n * 3, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  const CertCounters *hc = reinterpret_cast<const CertCounters *>(h);
  g_cert_stats[0] += 1; g_cert_stats[1] += n; g_cert_stats[2] += h[8]; g_cert_stats[3] += hc->n_open;
  g_cert_stats[5] += hc->violations; g_cert_stats[6] += hc->still_open; g_cert_stats[7] += h[10];
  static const bool snb_debug = getenv("MM3D_SNB_DEBUG") != nullptr;
  if (snb_debug)
    fprintf(stderr, "sift certified octave: n=%d items=%d rejected=%d to-search=%d marked=%d open=%d still_open=%d violations=%d exact overflow=%d fast overflow items=%d\n", n, n_items,
            hc->pad[0], hc->pad[1], h[8], hc->n_open, hc->still_open, hc->violations, h[9], h[10]);
  const bool ok = hc->still_open == 0 && hc->violations == 0 && h[9] == 0;
  if (!ok) g_cert_stats[4] += 1;
  return ok;
}


// test hook: val* and B of one octave (0-based) of detectKeypoints(SIFT) on `points`, downloaded; returns the octave
// cloud's size (0: the octave does not exist).  Nothing is written when the size exceeds `capacity`.
size_t debug_sift_cert_octave(Context *c, const mm3d_cloud *points, double min_scale, int octave, float *val_host, float *bound_host, size_t capacity)
{
  std::unique_ptr<mm3d_cloud> cur;
  const mm3d_cloud *input = points;
  float scale = (float)min_scale;
  for (int oct = 0; oct <= octave; ++oct) {
    std::unique_ptr<mm3d_cloud> next(downsample(c, input, (double)scale));
    cur = std::move(next);
    input = cur.get();
    if (cur->n < 25) return 0;
    if (oct < octave) scale *= 2;
  }
  const mm3d_cloud *oc = cur.get();
  float scales[kScales];
  for (int i = 0; i < kScales; ++i) scales[i] = scale * powf(2.0f, (1.0f * (float)i - 1.0f) / 3.0f);
  SiftScales sc;
  for (int i = 0; i < kScales; ++i) {
    sc.sigma_sqr[i] = powf(scales[i], 2.0f);
    sc.thr9[i] = 9 * sc.sigma_sqr[i];
    sc.rcp[i] = (float)(1.0 / (double)sc.sigma_sqr[i]);
  }
  const float max_radius = 3.0f * scales[kScales - 1];
  const float r2 = (float)((double)max_radius * (double)max_radius);
  const size_t n = oc->n;
  if (n > capacity) return n;
  cloud_hilbert(c, oc, 2.5f * scale);
  const Grid &gr = cloud_grid(c, oc, max_radius * 0.5f);
  DevBuf<float> dogv(c, n * kDog), dogb(c, n * kDog);
  DevBuf<unsigned char> mark(c, n);
  MM3D_HIP(hipMemsetAsync(mark.get(), 0, n, c->stream));
  SfLaunch<SfCfgDefault> fl(c, oc->n_wave_items);
  const SfScales fs = cert_scales(sc, r2);
  DevBuf<float> rlo2(c, n), rup2(c, n);
  sift_dog_fast_launch(c, oc, gr, oc->n_wave_items, max_radius, fs, fl, dogv.get(), dogb.get(), rlo2.get(), rup2.get(), mark.get());
  MM3D_HIP(hipMemcpyAsync(val_host, dogv.get(), n * kDog * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  MM3D_HIP(hipMemcpyAsync(bound_host, dogb.get(), n * kDog * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  return n;
}

// normals_radius > 0 and normals_out: the caller also wants computeSurfaceNormals(points, normals_radius) (it is about to
// describe the keypoints).  When the first octave works on `points` itself and normal_radius <= 3 sigma_max, the normals
// come out of the first octave's scale-space launch (k_sift_dog_lds<., true>); otherwise compute_normals runs as usual.
// Either way *normals_out holds the same bits.
mm3d_cloud *detect_keypoints_sift(Context *c, const mm3d_cloud *points, double min_scale, int nr_octaves,
                                  int nr_scales, double min_contrast, double normals_radius, mm3d_normals **normals_out, float grid_cell_hint)
{
  std::unique_ptr<mm3d_normals> nrm_res;
  static const bool fuse_normals = [] { const char *e = getenv("MM3D_SIFT_NO_FUSED_NORMALS"); return !(e && atoi(e)); }();   // A/B knob
  MM3D_REQUIRE(nr_scales == 3, "SIFT: only nr_scales_per_octave == 3 (the reference's setting) is built");
  std::vector<DevBuf<float4>> parts;
  std::vector<size_t> part_n;
  std::unique_ptr<mm3d_cloud> cur;
  const mm3d_cloud *input = points;
  float scale = (float)min_scale;
  static const bool try_identity = [] { const char *e = getenv("MM3D_SIFT_NO_IDENTITY"); return !(e && atoi(e)); }();   // A/B knob
  for (int oct = 0; oct < nr_octaves; ++oct) {
    // The octave's cloud: VoxelGrid(leaf = scale) of the previous octave's.  In the first octave of the reference's call
    // (min_scale = resolution on a cloud downSample(resolution) made) that filter returns its input bit for bit; one
    // small launch checks it, and the octave then works on `points` itself -- no voxel chain, and the Hilbert order and
    // work items the normals built on that cloud are reused instead of being built again on a copy.
    const mm3d_cloud *octave_cloud = nullptr;
    if (oct == 0 && try_identity && downsample_is_identity(c, input, (double)scale)) {
      octave_cloud = input;
    } else {
      std::unique_ptr<mm3d_cloud> next(downsample(c, input, (double)scale));
      cur = std::move(next);
      octave_cloud = cur.get();
    }
    input = octave_cloud;
    if (octave_cloud->n < 25) break;
    float scales[kScales];
    for (int i = 0; i < kScales; ++i)
      scales[i] = scale * powf(2.0f, (1.0f * (float)i - 1.0f) / (float)nr_scales);
    SiftScales sc;
    for (int i = 0; i < kScales; ++i) {
      sc.sigma_sqr[i] = powf(scales[i], 2.0f);
      sc.thr9[i] = 9 * sc.sigma_sqr[i];
      sc.rcp[i] = (float)(1.0 / (double)sc.sigma_sqr[i]);
    }
    const float max_radius = 3.0f * scales[kScales - 1];
    const float r2 = (float)((double)max_radius * (double)max_radius);
    const int n = (int)octave_cloud->n;
    // query order + wave work items.  An item never straddles a block of 8 x 8 Hilbert cells; with the default 0.25 m cell
    // the later octaves' sparser clouds (leaf 0.2 m, 0.4 m) would fill such a block with 50 and 34 points instead of 64,
    // and every item pays its staging and its end-of-item barrier whatever it holds: the cell grows with the leaf
    // (2.5 leaves: 0.25 m in the first octave, as everywhere else)
    static const float hil_factor = [] { const char *e = getenv("MM3D_SIFT_HIL_FACTOR"); return e ? (float)atof(e) : 2.5f; }();
    // (only on the octave clouds SIFT owns: the caller's `points` keeps the default cell, so the query order its later users --
    // FPFH blocks, ICP and score reductions -- inherit does not depend on whether SIFT ran first; at the reference's resolution
    // 0.1 the two cells are the same 0.25 m)
    cloud_hilbert(c, octave_cloud, octave_cloud == points ? 0.25f : hil_factor * scale);
    const int n_items = octave_cloud->n_wave_items;
    // scale space on a grid with cell = r/2 -- or, in the first octave on `points` itself, on the grid the caller is about
    // to build on that cloud anyway (the descriptors' radius / 2) when its cell is close to that: one grid build less per
    // map; the cell size only sizes the staged boxes, no result depends on it
    float grid_cell = max_radius * 0.5f;
    if (octave_cloud == points && grid_cell_hint >= 0.8f * grid_cell && grid_cell_hint <= 1.25f * grid_cell) grid_cell = grid_cell_hint;
    const Grid &gr = cloud_grid(c, octave_cloud, grid_cell);
    DevBuf<float> dog(c, (size_t)n * kDog);
    DevBuf<int> knn(c, (size_t)n * kKnn);
    // everything the octave wants zeroed lies in ONE buffer -- knn_ok [n bytes, padded to 4] | flags [3 n + 1 ints] | n_search
    // [1 int] | the searching kernel's counters -- and is cleared by one fill dispatch (a second extremum test clears from
    // flags on)
    const size_t z_knn = ((size_t)n + 3) & ~(size_t)3, z_flags = ((size_t)n * 3 + 1) * sizeof(int);
    const size_t z_tail = sizeof(int) + sizeof(CertCounters);
    DevBuf<unsigned char> zeroed(c, z_knn + z_flags + z_tail);
    MM3D_HIP(hipMemsetAsync(zeroed.get(), 0, z_knn + z_flags + z_tail, c->stream));
    unsigned char *const knn_ok_p = zeroed.get();
    int *const flags_p = reinterpret_cast<int *>(zeroed.get() + z_knn);
    int *const n_search_p = flags_p + (size_t)n * 3 + 1;
    CertCounters *const search_ctr = reinterpret_cast<CertCounters *>(n_search_p + 1);
    bool zero_again = false;                             // (the first extremum test finds its words cleared by the fill above)
    const DevPtr<unsigned char> knn_ok{knn_ok_p};
    // the normals ride on the first octave when it works on `points` itself and their ball is inside the scale space's
    const bool fused = oct == 0 && normals_out && fuse_normals && octave_cloud == points && normals_radius > 0.0 &&
                       (float)(normals_radius * normals_radius) <= r2 && octave_cloud->n_finite == octave_cloud->n;
    if (fused) {
      nrm_res.reset(new mm3d_normals());
      nrm_res->n = points->n;
      nrm_res->nrm = DevBuf<float4>(c, points->n);
    }
    const DevPtr<int> flags{flags_p}, n_search{n_search_p};
    DevBuf<int> pos(c, (size_t)n * 3 + 1);
    const int nh = (int)octave_cloud->n_finite;
    int *h = (int *)c->pin(64);                          // [0] keypoints, [1] points for the searching kernel, [2] points it left open (none)
    // Later octaves (round 6): the keypoint DECISION is certified from an unsorted scale space with an error bound, and only
    // the points it leaves open get the sorted lists (sift_cert.hpp).  The first octave keeps its lists -- the fused normals
    // ride on them; where they do not (a normals ball wider than the first octave's 3 sigma_max: the dense indoor workload at
    // resolution 0.05) the first octave is certified too.  MM3D_SIFT_CERT=0: every octave on the sorted lists (the A/B);
    // MM3D_SIFT_CERT=1: the later octaves only.
    static const int cert_mode = [] { const char *e = getenv("MM3D_SIFT_CERT"); return e ? atoi(e) : 2; }();
    // (a small octave is cheaper on the sorted lists: the certified path is sixteen launches against five, and below a few
    // ten thousand points every launch is latency -- measured interleaved, octaves under 15 000 points on the lists: 2 x 10 k 232 against
    // 220 map-pairs/s, 64 x 50 k 12 650 against 12 320, 4 x 200 k and the headline unchanged; MM3D_SIFT_CERT_MIN moves the line)
    static const int cert_min_env = [] { const char *e = getenv("MM3D_SIFT_CERT_MIN"); return e ? atoi(e) : 15000; }();
    const int cert_min = g_cert_min_override.load() >= 0 ? g_cert_min_override.load() : cert_min_env;     // (test hook: debug_sift_cert_min)
    bool certified = false;
    if (cert_mode > 0 && (oct >= 1 || (cert_mode >= 2 && !fused)) && octave_cloud->n_finite == octave_cloud->n && n >= cert_min) {
      certified = sift_octave_certified(c, octave_cloud, gr, n_items, max_radius, r2, sc, (float)min_contrast, flags.get(), pos.get(), h);
      if (!certified) zero_again = true;                   // (the flags hold the abandoned run's)
    }
    if (!certified) {
    // (measured, round 4: the sixteen-wave configuration is bit-equal and NOT faster -- octave 1 1.74 against 1.69 ms, octave 2
    // 0.88 against 0.78 -- every phase's ticks per query double with the waves: the kernel is bound by VALU issue, not by
    // latency; it stays behind this knob)
    static const bool large16 = [] { const char *e = getenv("MM3D_SIFT_LARGE16"); return e && atoi(e); }();
    SiftDogPending pend = oct == 0 ? sift_dog_octave<SiftCfgSmall>(c, oct, octave_cloud, gr, n_items, max_radius, r2, sc, dog.get(), knn.get(), knn_ok.get(),
                                                                   fused ? normals_radius : 0.0, fused ? nrm_res->nrm.get() : nullptr)
                                   : (large16 ? sift_dog_octave<SiftCfgLarge16>(c, oct, octave_cloud, gr, n_items, max_radius, r2, sc, dog.get(), knn.get(), knn_ok.get())
                                              : sift_dog_octave<SiftCfgLarge>(c, oct, octave_cloud, gr, n_items, max_radius, r2, sc, dog.get(), knn.get(), knn_ok.get()));
    // The extremum test: the points whose list held 25 neighbours read them back (k_sift_extrema_knn, nearly all
    // of them); the others -- borders and sparse places, where the 25 nearest reach beyond 3 sigma_max: a handful per map --
    // are listed by that kernel and searched for in the same grid, a wave per point (k_sift_extrema_one on the exact values:
    // every interval a point, every ball grown until it holds 25).  Until round 6 that second kernel was a second PASS: the
    // first one counted the points, the host looked at the count at the octave's wait, and two maps in three then ran a
    // dozen launches for their two or three stray points and waited once more.  The scale-space items left to the
    // global-memory lists are still COUNTED by the first pass and served after the octave's wait, the test taken again (a
    // launch that finds nothing to do is not free here: it queues for LDS behind the other streams' kernels).
    const Grid &gk = gr;
    DevBuf<float4> dogx(c, (size_t)n * 3);
    DevBuf<unsigned char> cls(c, (size_t)n);
    DevBuf<int> search_ids(c, (size_t)n);
    auto extremum_test = [&]() {
      MM3D_LAUNCH(c, "sift_pack", n * 69.0, k_sift_pack_iv, dim3(div_up(n, 256)), dim3(256), 0, (const float *)dog.get(), (const float *)nullptr, n,
                  (float)min_contrast, dogx.get(), cls.get());
      if (zero_again) MM3D_HIP(hipMemsetAsync(flags.get(), 0, z_flags + z_tail, c->stream));
      zero_again = true;
      MM3D_LAUNCH(c, "sift_extrema_knn", nh * 16.0 + n * 0.25 * (kKnn * 36.0 + 20.0), k_sift_extrema_knn, dim3(div_up(nh, 256)), dim3(256), 0,
                  (const float4 *)octave_cloud->hil_pts.get(), nh, n, (const float *)dog.get(), (const float4 *)dogx.get(), (const int *)knn.get(),
                  (const unsigned char *)knn_ok.get(), (float)min_contrast, flags.get(), search_ids.get(), n_search.get());
      MM3D_LAUNCH(c, "sift_extrema_one", 0.0, k_sift_extrema_one<true>, dim3(64), dim3(256), 0, (const int *)search_ids.get(), (const int *)n_search.get(),
                  (const float4 *)octave_cloud->pts.get(), gk.view(), (const float *)nullptr, r2, (const float4 *)dogx.get(), n, (const float *)dog.get(),
                  (const float *)nullptr, (const unsigned char *)cls.get(), flags.get(), (unsigned char *)nullptr, (unsigned char *)nullptr, search_ctr);
      exclusive_scan_int(c, flags.get(), pos.get(), (size_t)n * 3 + 1);
      MM3D_HIP(hipMemcpyAsync(h, pos.get() + (size_t)n * 3, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      MM3D_HIP(hipMemcpyAsync(h + 1, n_search.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
      MM3D_HIP(hipMemcpyAsync(h + 2, &search_ctr->still_open, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      c->sync();
    };
    extremum_test();
#ifdef MM3D_SNB_STATS
    {   // instrumentation build: the phase ticks of this octave alone (scripts/snb_stats.py prints the sum over the octaves)
      unsigned long long v[32];
      (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_snb_stats), sizeof(v));
      const double q = (double)std::max<unsigned long long>(v[11], 1), it = (double)std::max<unsigned long long>(v[0], 1);
      fprintf(stderr, "snb octave %d: n=%d items=%llu queries=%llu hits/query %.1f staged/item %.1f rounds/item %.2f | ticks per query: A %.0f B %.0f C %.0f D %.0f E %.0f "
              "consume %.0f | per item per wave: claim %.0f stage %.0f stage_barrier %.0f end_barrier %.0f | total/query %.0f; parts: %llu items in %llu parts, extra bands %llu\n",
              oct, n, v[0], v[11], v[12] / q, v[13] / it, v[14] / it, v[4] / q, v[5] / q, v[6] / q, v[7] / q, v[8] / q, v[9] / q, v[1] / it, v[2] / it, v[3] / it, v[10] / it,
              v[15] / q, v[17], v[18], v[19]);
      unsigned long long z[32] = {0};
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_snb_stats), z, sizeof(z));
    }
#endif
    static const bool snb_debug = getenv("MM3D_SNB_DEBUG") != nullptr;
    if (snb_debug)
      fprintf(stderr, "sift_dog: n=%d items=%d overflow items=%d, points for the searching extremum kernel %d\n", gr.n, n_items, *pend.h_overflow, h[1]);
    if (*pend.h_overflow > 0) {
      // the items the first pass left out get their scale space now (and, from the large LDS configuration, their
      // neighbour lists): the test is taken again on the complete values
      pend.fallback(*pend.h_overflow);
      extremum_test();
    }
    MM3D_REQUIRE(h[2] == 0, "SIFT: the extremum search left a point undecided on exact values");
    }
    const size_t nk = (size_t)h[0];
    DevBuf<float4> kp(c, nk);
    if (nk)
      MM3D_LAUNCH(c, "sift_emit", n * 3 * 8.0, k_sift_emit, dim3(div_up((size_t)n * 3, 256)), dim3(256), 0, octave_cloud->pts.get(),
                  flags.get(), pos.get(), (size_t)n * 3, kp.get());
    c->settle();
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
  c->settle();
  if (normals_out) {
    // not fused (another min_scale than the cloud's resolution, a wider normals ball, fewer than 25 points, ...): the usual launch
    if (!nrm_res) nrm_res.reset(compute_normals(c, points, normals_radius));
    *normals_out = nrm_res.release();
  }
  return cloud_from_device(c, std::move(all), total);
}

#ifdef MM3D_SNB_STATS
extern "C" void mm3d_debug_snb_stats_sift(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_snb_stats), sizeof(unsigned long long) * 32);
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_snb_stats), z, sizeof(z)); }
}
#endif

}  // namespace mm3d
