// sorted_nb.hpp -- radius neighbourhoods in the CPU path's order, for float sums that must come out
// bit for bit.
//
// PCL's feature estimators walk the result of KdTreeFLANN::radiusSearch, which FLANN returns sorted by
// (squared distance, index), and accumulate in float as they go: computeMeanAndCovarianceMatrix
// (normals), SIFTKeypoint::computeScaleSpace (Gaussian sums with a `break` that relies on the order),
// FPFHEstimation::weightPointSPFHSignature.  A float sum is only reproducible in its own order, so the
// kernels that feed such sums build every query's neighbour list in exactly that order first.
//
// One wave works on a GROUP of up to kSnG query points that lie close together (a quarter of a
// Hilbert work item).  The box of grid cells the group can reach is streamed through LDS twice, 256
// candidates at a time with coalesced loads (wave_stream_box):
//   pass 1  every (query, candidate) pair inside the radius bumps the query's histogram over kSnNB
//           equal-width buckets of the squared distance (LDS atomics, two 16-bit counters per word);
//           a wave-wide scan turns the counts into bucket starts,
//   pass 2  the same pairs again: the atomic's return value is the pair's slot inside its bucket; the
//           64-bit key (distance bits << 32 | original index) goes to the wave's scratch list,
//   rank    inside a bucket (a handful of keys) every key counts the smaller keys of its bucket and
//           lands at its final place, as the payload the consumer's chains read; the keys of several
//           queries at a time are pulled into LDS (the tile's memory) for that, so the inner loop
//           never waits for global memory.
// The lists live in a per-wave global scratch region that is rewritten for every group (a few tens
// of KB that stay in L2 / Infinity Cache); LDS holds the tile and the histograms (8.6 KB per wave).
// The consumer then runs its float chains, one chain per lane, over the sorted payloads.
//
// Kernels are persistent: a fixed number of blocks, each wave claims units (work item, quarter) from
// the counter of its XCD's contiguous slice of the Hilbert order, so one XCD's L2 sees one region.
#pragma once

#include "device_util.hpp"

namespace mm3d {

constexpr int kSnG = 16;            // query points per group
constexpr int kSnNB = 128;          // distance buckets per query
constexpr int kSnTile = 256;        // staged candidates per tile
constexpr int kSnEntries = 16384;   // scratch list entries per wave (sum over the group's queries)

// entries per rank batch: their 8-byte keys and their payloads share the tile's 4 KB (256 with 8-byte
// payloads, 128 with 16-byte ones)
template <class Payload>
constexpr int sn_batch_cap() { return ((kSnTile * 16) / (8 + (int)sizeof(Payload))) / 64 * 64; }

struct SnLds {
  float4 tile[kSnTile];             // staged candidates; during the rank step: a batch of keys and payloads
  float4 q[kSnG];                   // the group's queries
  int off[64], beg[64];
  unsigned cnt[kSnG][kSnNB / 2];    // pass 1: counts (2 x u16); then bucket starts; after pass 2: bucket ENDS
  int list_off[kSnG + 1];
};

// start of bucket b = end of bucket b - 1 (cnt holds the ends once pass 2 has filled every bucket)
__device__ __forceinline__ int sn_bucket_start(const unsigned *ends, int b)
{
  if (b == 0) return 0;
  const unsigned w = ends[(b - 1) >> 1];
  return (int)(((b - 1) & 1) ? (w >> 16) : (w & 0xffffu));
}
__device__ __forceinline__ int sn_bucket_end(const unsigned *ends, int b)
{
  const unsigned w = ends[b >> 1];
  return (int)((b & 1) ? (w >> 16) : (w & 0xffffu));
}

// instrumentation build (-DMM3D_SN_STATS): 100 MHz ticks per phase, summed over waves (lane 0)
#ifdef MM3D_SN_STATS
__device__ unsigned long long g_sn_stats[16];   // 0 groups, 1 pass 1, 2 prefix, 3 pass 2, 4 rank, 5 chains (consumer), 6 queries, 7 list entries, 8 staged candidates
#define SN_TICK(var_) const long long var_ = wall_clock64()
#define SN_TOCK(i_, from_) do { if (lane == 0) atomicAdd(&g_sn_stats[i_], (unsigned long long)(wall_clock64() - (from_))); } while (0)
#define SN_COUNT(i_, v_) do { if (lane == 0) atomicAdd(&g_sn_stats[i_], (unsigned long long)(v_)); } while (0)
#else
#define SN_TICK(var_)
#define SN_TOCK(i_, from_)
#define SN_COUNT(i_, v_)
#endif

struct SnScratch {
  unsigned long long *tmp;          // [waves][kSnEntries] keys in bucket order
  void *fin;                        // [waves][kSnEntries] payloads in final order
  int *unit_ctr;                    // [kXcds] next unit of every XCD slice
  int *error;                       // set when one query alone has more than kSnEntries neighbours
  // fallback launches (snb_lds.hpp): only the work items the LDS path could not hold, *ov_count of them
  const int *ov_items = nullptr;
  const int *ov_count = nullptr;
};

// the launch's units (work item, quarter) and the item a unit belongs to
__device__ __forceinline__ int sn_unit_count(const SnScratch &s, int n_items) { return (s.ov_count ? *s.ov_count : n_items) * 4; }
__device__ __forceinline__ int sn_unit_item(const SnScratch &s, int unit) { return s.ov_items ? s.ov_items[unit >> 2] : (unit >> 2); }

__device__ __forceinline__ float sn_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__device__ __forceinline__ int sn_bucket(float d2, float bscale)
{
  const int b = (int)(d2 * bscale);
  return b < kSnNB - 1 ? b : kSnNB - 1;
}

// Claims the next unit: this block's XCD slice first, then the other slices (a grid of fewer than kXcds
// blocks, uneven slices); -1 when every slice is done.  Wave-uniform.
__device__ __forceinline__ int sn_claim_unit(int *unit_ctr, int n_units, int lane)
{
  const int q = n_units / kXcds, r = n_units % kXcds;
  for (int k = 0; k < kXcds; ++k) {
    const int xcd = (int)((blockIdx.x + k) % kXcds);
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int count = xcd < r ? q + 1 : q;
    if (count == 0) continue;
    int u = count;
    if (lane == 0 && *(volatile int *)&unit_ctr[xcd] < count) u = atomicAdd(&unit_ctr[xcd], 1);
    u = __builtin_amdgcn_readfirstlane(u);
    if (u < count) return first + u;
  }
  return -1;
}

// Builds the sorted neighbour lists of the group's queries.  Lane l carries query l / 4 in (qx, qy, qz)
// (4 lanes per query); g <= kSnG queries are live.  On return L.list_off[p] .. L.list_off[p + 1] is query
// p's range in `fin` (payloads in (d2, index) order) for p < the returned count g' <= g (g' < g only when
// the scratch region cannot hold the whole group: the caller runs the rest as the next group).
// opts = the surface in original order (the lists name points by original index < 2^28);
// finish(d2, original index, point) -> Payload is called once per list entry.
template <class Payload, class Finish>
__device__ __forceinline__ int sn_build_lists(const GridView &g, SnLds &L, float qx, float qy, float qz, int n_q, float radius, float r2,
                                              const float4 *__restrict__ opts, unsigned long long *tmp, Payload *fin, int *error, int lane,
                                              Finish &&finish)
{
  const float ri = radius * 1.0001f + 1e-4f;
  const bool live = (lane >> 2) < n_q;
  const float lx = wave_min_f(live ? qx : INFINITY), hx = wave_max_f(live ? qx : -INFINITY);
  const float ly = wave_min_f(live ? qy : INFINITY), hy = wave_max_f(live ? qy : -INFINITY);
  const float lz = wave_min_f(live ? qz : INFINITY), hz = wave_max_f(live ? qz : -INFINITY);
  const int x0 = max(cell_floor(lx - ri, g.minx, g.inv), 0), x1 = min(cell_floor(hx + ri, g.minx, g.inv), g.dx - 1);
  const int y0 = max(cell_floor(ly - ri, g.miny, g.inv), 0), y1 = min(cell_floor(hy + ri, g.miny, g.inv), g.dy - 1);
  const int z0 = max(cell_floor(lz - ri, g.minz, g.inv), 0), z1 = min(cell_floor(hz + ri, g.minz, g.inv), g.dz - 1);
  const KeepNearBox keep{lx, hx, ly, hy, lz, hz, ri * ri};
  const float bscale = (float)kSnNB / r2;
  for (int p = 0; p < kSnG; ++p) L.cnt[p][lane] = 0u;
  if ((lane & 3) == 0) L.q[lane >> 2] = make_float4(qx, qy, qz, 0.0f);
  wave_lds_fence();
  SN_TICK(t_p1);
  SN_COUNT(0, 1);
  SN_COUNT(6, n_q);
  // pass 1: histograms
  wave_stream_box<kSnTile, 0>(g, x0, x1, y0, y1, z0, z1, L.tile, (float4 *)nullptr, L.off, L.beg, lane, [](int, float4 (&)[1]) {},
                              [&](int n) {
                                SN_COUNT(8, n);
                                float4 c[kSnTile / kWave];
#pragma unroll
                                for (int u = 0; u < kSnTile / kWave; ++u) c[u] = L.tile[lane + u * kWave];
                                for (int p = 0; p < n_q; ++p) {
                                  const float px = sn_readlane(qx, p * 4), py = sn_readlane(qy, p * 4), pz = sn_readlane(qz, p * 4);
#pragma unroll
                                  for (int u = 0; u < kSnTile / kWave; ++u) {
                                    if (u * kWave >= n) break;                    // wave-uniform: slots beyond the tile's fill
                                    const float d2 = dist2(px, py, pz, c[u].x, c[u].y, c[u].z);
                                    if (lane + u * kWave < n && d2 < r2) {
                                      const int b = sn_bucket(d2, bscale);
                                      atomicAdd(&L.cnt[p][b >> 1], (b & 1) ? 0x10000u : 1u);
                                    }
                                  }
                                }
                              },
                              keep);
  wave_lds_fence();
  SN_TOCK(1, t_p1);
  SN_TICK(t_pre);
  // counts -> bucket starts; list offsets
  int total = 0, fit = 0;
  for (int p = 0; p < n_q; ++p) {
    const unsigned w = L.cnt[p][lane];
    const int c0 = (int)(w & 0xffffu), c1 = (int)(w >> 16);
    int incl = c0 + c1;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int t = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += t;
    }
    const int excl = incl - c0 - c1;
    const int m = __builtin_amdgcn_readlane(incl, kWave - 1);
    const bool ok = fit == p && total + m <= kSnEntries && m <= 0xffff;
    if (ok) {
      L.cnt[p][lane] = (unsigned)excl | ((unsigned)(excl + c0) << 16);
      if (lane == 0) L.list_off[p] = total;
      total += m;
      fit = p + 1;
    }
  }
  if (lane == 0) L.list_off[fit] = total;
  if (fit == 0 && n_q > 0) {                   // one query alone overflows the scratch list: reported, skipped
    if (lane == 0) *error = 1;
    if (lane == 0) L.list_off[1] = 0;
    wave_lds_fence();
    return 1;
  }
  wave_lds_fence();
  SN_TOCK(2, t_pre);
  SN_COUNT(7, total);
  SN_TICK(t_p2);
  // pass 2: keys into their buckets
  wave_stream_box<kSnTile, 0>(g, x0, x1, y0, y1, z0, z1, L.tile, (float4 *)nullptr, L.off, L.beg, lane, [](int, float4 (&)[1]) {},
                              [&](int n) {
                                float4 c[kSnTile / kWave];
#pragma unroll
                                for (int u = 0; u < kSnTile / kWave; ++u) c[u] = L.tile[lane + u * kWave];
                                for (int p = 0; p < fit; ++p) {
                                  const float px = sn_readlane(qx, p * 4), py = sn_readlane(qy, p * 4), pz = sn_readlane(qz, p * 4);
                                  const int base = L.list_off[p];
#pragma unroll
                                  for (int u = 0; u < kSnTile / kWave; ++u) {
                                    if (u * kWave >= n) break;
                                    const float d2 = dist2(px, py, pz, c[u].x, c[u].y, c[u].z);
                                    if (lane + u * kWave < n && d2 < r2) {
                                      const int b = sn_bucket(d2, bscale);
                                      const unsigned old = atomicAdd(&L.cnt[p][b >> 1], (b & 1) ? 0x10000u : 1u);
                                      const int at = (int)((b & 1) ? (old >> 16) : (old & 0xffffu));
                                      // only the original index goes to the scratch list (4 bytes): the rank step fetches
                                      // the point again and recomputes the distance
                                      reinterpret_cast<unsigned *>(tmp)[base + at] = __float_as_uint(c[u].w);
                                    }
                                  }
                                }
                              },
                              keep);
  wave_lds_fence();
  SN_TOCK(3, t_p2);
  SN_TICK(t_rank);
  // rank inside the buckets -> final order, one batch of whole buckets (at most kCap entries, normally one
  // query's whole list) at a time, entirely in LDS: the batch's indices come out of `tmp` with coalesced loads,
  // every entry fetches its point (L2: the cloud is a few MB), recomputes its distance and builds its payload;
  // the keys are ranked inside their buckets, the payloads land at their final places in LDS, and the batch
  // goes to `fin` with coalesced full-line stores.  (Scratch traffic is what bounds these kernels: with 8-byte
  // keys in `tmp` and payloads scattered straight into `fin` they spent most of their time on it.)
  constexpr int kCap = sn_batch_cap<Payload>();
  constexpr int kRU = kCap / kWave;
  unsigned long long *kbuf = reinterpret_cast<unsigned long long *>(L.tile);
  Payload *obuf = reinterpret_cast<Payload *>(kbuf + kCap);
  const unsigned *tmp32 = reinterpret_cast<const unsigned *>(tmp);
  for (int p = 0; p < fit; ++p) {
    const int lo = L.list_off[p], hi = L.list_off[p + 1];
    const float4 qp = L.q[p];
    int s = lo;
    while (s < hi) {
      int e_end = hi;
      if (hi - s > kCap) {
        // the largest bucket boundary within kCap of s (lane l holds the ends of buckets 2l and 2l + 1)
        const unsigned w = L.cnt[p][lane];
        const int target = s - lo + kCap, e0 = (int)(w & 0xffffu), e1 = (int)(w >> 16);
        const int cut = wave_max_int(e1 <= target ? e1 : (e0 <= target ? e0 : 0));
        e_end = lo + cut;
      }
      if (e_end > s) {
        const int n = e_end - s;
        unsigned long long key[kRU];
        Payload pay[kRU];
#pragma unroll
        for (int u = 0; u < kRU; ++u) {
          const int i = lane + u * kWave;
          const unsigned idx = tmp32[s + (i < n ? i : 0)];
          const float4 pt = opts[idx];
          const float d2 = dist2(qp.x, qp.y, qp.z, pt.x, pt.y, pt.z);
          key[u] = ((unsigned long long)__float_as_uint(d2) << 32) | idx;
          pay[u] = finish(d2, idx, pt);
          if (i < n) kbuf[i] = key[u];
        }
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < kRU; ++u) {
          const int i = lane + u * kWave;
          if (i < n) {
            const int b = sn_bucket(__uint_as_float((unsigned)(key[u] >> 32)), bscale);
            const int bs = sn_bucket_start(L.cnt[p], b) - (s - lo), be = sn_bucket_end(L.cnt[p], b) - (s - lo);
            int r = 0;
            for (int j = bs; j < be; ++j) r += kbuf[j] < key[u] ? 1 : 0;
            obuf[bs + r] = pay[u];
          }
        }
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < kRU; ++u) {
          const int i = lane + u * kWave;
          if (i < n) fin[s + i] = obuf[i];
        }
        wave_lds_fence();
        s = e_end;
      } else {
        // one bucket alone holds more than kCap entries (hundreds of neighbours at the same distance): rank it
        // against global memory
        const unsigned w = L.cnt[p][lane];
        const int e0 = (int)(w & 0xffffu), e1 = (int)(w >> 16), from = s - lo;
        const int b_end = lo + wave_min_int(e0 > from ? e0 : (e1 > from ? e1 : 0x7fffffff));
        for (int i = s + lane; i < b_end; i += kWave) {
          const unsigned idx = tmp32[i];
          const float4 pt = opts[idx];
          const float d2 = dist2(qp.x, qp.y, qp.z, pt.x, pt.y, pt.z);
          const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | idx;
          int r = 0;
          for (int j = s; j < b_end; ++j) {
            const unsigned oj = tmp32[j];
            const float4 pj = opts[oj];
            const unsigned long long kj = ((unsigned long long)__float_as_uint(dist2(qp.x, qp.y, qp.z, pj.x, pj.y, pj.z)) << 32) | oj;
            r += kj < key ? 1 : 0;
          }
          fin[s + r] = finish(d2, idx, pt);
        }
        s = b_end;
      }
    }
  }
  wave_lds_fence();
  SN_TOCK(4, t_rank);
  return fit;
}

// host side: scratch for a persistent launch of `blocks` blocks of 4 waves
template <class Payload>
struct SnLaunch {
  DevBuf<unsigned long long> tmp;
  DevBuf<Payload> fin;
  DevBuf<int> ctr;                   // kXcds unit counters + the error flag
  unsigned blocks = 0;
  SnLaunch(Context *c, int n_units, size_t n_surface_points, int waves_per_block = 4, unsigned max_blocks = 1024u)
  {
    if (n_surface_points >= ((size_t)1 << 28))
      throw Error(MM3D_EUNSUPPORTED, "sorted neighbour lists: clouds of 2^28 points or more are not supported");
    // four blocks of 8.6 KB x 4 per CU fit the LDS; never more blocks than there are units
    const unsigned want = div_up((size_t)n_units, (size_t)waves_per_block);
    blocks = want < max_blocks ? (want ? want : 1u) : max_blocks;
    tmp = DevBuf<unsigned long long>(c, (size_t)blocks * waves_per_block * kSnEntries);
    fin = DevBuf<Payload>(c, (size_t)blocks * waves_per_block * kSnEntries);
    ctr = DevBuf<int>(c, kXcds + 1);
    MM3D_HIP(hipMemsetAsync(ctr.get(), 0, (kXcds + 1) * sizeof(int), c->stream));
  }
  int *error() const { return ctr.get() + kXcds; }
};

}  // namespace mm3d
