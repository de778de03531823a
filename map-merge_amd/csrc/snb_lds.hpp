// snb_lds.hpp -- radius neighbourhoods in the CPU path's (distance, index) order, built entirely in LDS.
//
// Why the order matters is in sorted_nb.hpp (PCL's estimators add floats while they walk
// KdTreeFLANN::radiusSearch's result: R/src/features.cpp:50-56 SIFT, :105-113 FPFH, :171-176 normals).
// That file keeps the lists in a global scratch region; round 2 measured what that costs (sift_dog moved
// 200 x its algorithmic bytes, 60 % of the wave cycles waited).  Here nothing leaves the CU:
//
//   * a BLOCK owns one Hilbert work item (<= 64 queries that form a compact patch).  The box of grid cells
//     the whole item can reach is staged ONCE into the block's LDS tile (coalesced 16-byte gathers, every
//     wave a share of the slots; x, y, z and the original index as separate arrays), optionally with one
//     float of payload per candidate;
//   * each wave then takes 64 / WAVES of the item's queries, ONE QUERY AT A TIME with the query in scalar
//     registers and the lanes over the candidates:
//       A  distance test of every staged candidate, two per lane and step with packed arithmetic; hits are
//          compacted (ballot + mbcnt) as (d2, tile slot);
//       B  every hit bumps its distance bucket (LDS atomic, the return value is its arrival rank);
//       C  one wave scan (DPP) turns the counts into bucket starts;
//       D  hits go to their bucket;
//       E  inside its bucket (a handful of hits) every hit counts the smaller (d2, original index) keys
//          and lands at its final place: a list of 16-bit tile slots in the wave's arena;
//   * when the arena is full (or the wave's queries are done) the consumer runs its float chains over the
//     lists -- slots index the tile, so positions / payloads come out of LDS too.
//
// What does not fit is made to fit first: an item whose box holds more candidates than the tile is worked in
// parts (runs of its queries, smaller boxes), a ball with more neighbours than the hit buffer in distance bands.
// Only a dense spot beyond that (one query's own box larger than the tile, a list longer than the arena) sends the
// item to an overflow list.  The list is counted (SnbCtl::ov_count); the caller reads the count at its next host sync
// and only then launches what serves it -- the global-scratch kernels of sorted_nb.hpp, for SIFT's first octave the
// large LDS configuration first (sub_items) -- because a launch that finds an empty list is not free on a GPU busy with
// other streams: it queues for its LDS at the head of a hardware queue.  Results are the same bits on either path
// (the order is total).
//
// The blocks run at two or three waves per SIMD (LDS), and measured (scripts/snb_stats.py, a build with
// -DMM3D_SNB_STATS) they are bound by instruction issue, not by latency: every phase is written to spend few
// instructions -- padded arrays and +inf sentinels instead of bounds tests, 32-bit counters instead of packed
// halves, values kept in registers between the phases.
#pragma once

#include <type_traits>

#include "device_util.hpp"

namespace mm3d {

// instrumentation build (-DMM3D_SNB_STATS): shader-clock ticks per phase, summed over waves
#ifdef MM3D_SNB_STATS
__device__ unsigned long long g_snb_stats[32];
// 0 items, 1 claim + box, 2 stage, 3 stage barrier, 4 A, 5 B, 6 C, 7 D, 8 E, 9 consume, 10 end barrier, 11 queries, 12 hits, 13 staged, 14 rounds, 15 total
// (accumulated in registers, one atomic per counter and wave when the kernel ends: atomics inside the loops would
// queue in front of the loads they are meant to time)
struct SnbStats { long long v[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; };
#define SNB_TICK(var_) const long long var_ = clock64()
#define SNB_TOCK(i_, from_) do { snb_st.v[i_] += clock64() - (from_); } while (0)
#define SNB_COUNT(i_, v_) do { snb_st.v[i_] += (v_); } while (0)
#define SNB_FLUSH() do { if ((threadIdx.x & 63) == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_snb_stats[i_], (unsigned long long)snb_st.v[i_]); } } while (0)
// 16 longest item (ticks), 17 items worked in parts, 18 parts of those, 19 bands beyond the first, 20 longest wave total
#define SNB_MAX(i_, v_) do { if ((threadIdx.x & 63) == 0) atomicMax(&g_snb_stats[i_], (unsigned long long)(v_)); } while (0)
#define SNB_ADD(i_, v_) do { if (threadIdx.x == 0) atomicAdd(&g_snb_stats[i_], (unsigned long long)(v_)); } while (0)
#else
struct SnbStats {};
#define SNB_TICK(var_)
#define SNB_TOCK(i_, from_)
#define SNB_COUNT(i_, v_)
#define SNB_FLUSH()
#define SNB_MAX(i_, v_)
#define SNB_ADD(i_, v_)
#endif

template <int WAVES_, int TILE_CAP_, int ARENA_, int HIT_CAP_, int NB_, bool PAY_>
struct SnbCfg {
  static constexpr int kWaves = WAVES_;          // waves per block
  static constexpr int kQ = 64 / WAVES_;         // queries per wave
  static constexpr int kLpq = WAVES_;            // lanes per query in the consumers (64 / kQ)
  static constexpr int kTileCap = TILE_CAP_;     // staged candidates per item
  static constexpr int kArena = ARENA_;          // list entries (16-bit slots) per wave
  static constexpr int kHitCap = HIT_CAP_;       // longest list
  static constexpr int kNB = NB_;                // distance buckets per query
  static constexpr bool kPay = PAY_;
  static_assert(64 % WAVES_ == 0 && (WAVES_ == 4 || WAVES_ == 8 || WAVES_ == 16), "4, 8 or 16 waves per block");
  static_assert(TILE_CAP_ % 256 == 0 && TILE_CAP_ <= 65536, "the tile is read 256 candidates at a time; slots are 16 bits");
  static_assert(HIT_CAP_ % 128 == 0 && NB_ % 64 == 0 && (ARENA_ == 0 || ARENA_ >= HIT_CAP_), "sizes (kArena = 0: per-query consumers, snb_run_each)");
  static_assert(HIT_CAP_ + 8 >= 128, "the staging offsets live in d2buf");
};

constexpr float kSnbFar = 3.0e18f;               // coordinate of the padding candidates: never within any radius

template <class Cfg>
struct alignas(16) SnbWave {
  unsigned short arena[Cfg::kArena ? Cfg::kArena : 8];   // sorted lists (tile slots), back to back
  float d2buf[Cfg::kHitCap + 8];                 // the current query's hits: d2 in arrival, then in bucket order, +inf behind the last
  unsigned short sbuf[Cfg::kHitCap + 8];         // their tile slots, same order (the last entry of both: a dump for masked stores)
  unsigned hist[Cfg::kNB + 4];                   // bucket counts, then bucket starts; [kNB] = total
  int list_off[Cfg::kQ + 1];
  int qidx[Cfg::kQ];                             // which of the part's queries the lists belong to
  // staging (before any list is built): the row offsets / first points of the current 64 rows live in d2buf
  __device__ __forceinline__ int *off() { return reinterpret_cast<int *>(d2buf); }
  __device__ __forceinline__ int *beg() { return reinterpret_cast<int *>(d2buf) + 64; }
};

template <class Cfg>
struct alignas(16) SnbLds {
  float tx[Cfg::kTileCap], ty[Cfg::kTileCap], tz[Cfg::kTileCap];   // staged candidates
  unsigned tw[Cfg::kTileCap];                                       // their original indices
  float pay[Cfg::kPay ? Cfg::kTileCap : 4];
  SnbWave<Cfg> w[Cfg::kWaves];
  float4 qpts[64];                               // the queries of the part in work (.w = original index)
  int n_tile[2];                                 // staged candidates of the current staging / the next one, in turn (by epoch)
  int item, overflow;
  int next_q[2];                                 // the waves take the part's queries one at a time; two counters in turn:
                                                 // a wave may still be claiming from one part's when the next part's is reset.
                                                 // n_tile likewise: a wave that is late to read this staging's count must not
                                                 // see thread 0 clear it for the next staging (an item worked in parts re-stages
                                                 // at once), so the next staging counts in the other word
};

// device control block of one launch (zeroed before it)
struct SnbCtl {
  int item_ctr[kXcds];      // next item of every XCD slice
  int fb_ctr[kXcds];        // the fallback launch's unit counters (sorted_nb.hpp)
  int ov_count;             // items left to the fallback launch
  int error;                // the fallback's "one query alone overflows the scratch" word
};

// ---- wave-wide scan / reductions on the DPP network ------------------------------------------------------
// Row shifts inside the rows of 16 lanes, then the two row broadcasts: six dependent VALU operations, ~80
// cycles, where six ds_bpermute shuffles take ~440 (scripts/micro/lat.hip checks them against a sequential
// loop and times both).
__device__ __forceinline__ int snb_scan_dpp(int v)                    // inclusive sum
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2, 3
  return v;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int snb_dpp_i(int v)                       // lanes without a source keep their own value
{
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ int snb_max_dpp(int v)                     // wave-uniform result
{
  v = max(v, snb_dpp_i<0x111, 0xf>(v));
  v = max(v, snb_dpp_i<0x112, 0xf>(v));
  v = max(v, snb_dpp_i<0x114, 0xf>(v));
  v = max(v, snb_dpp_i<0x118, 0xf>(v));
  v = max(v, snb_dpp_i<0x142, 0xa>(v));
  v = max(v, snb_dpp_i<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float snb_dpp_f(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float snb_min_f_dpp(float v)               // wave-uniform result
{
  v = fminf(v, snb_dpp_f<0x111, 0xf>(v));
  v = fminf(v, snb_dpp_f<0x112, 0xf>(v));
  v = fminf(v, snb_dpp_f<0x114, 0xf>(v));
  v = fminf(v, snb_dpp_f<0x118, 0xf>(v));
  v = fminf(v, snb_dpp_f<0x142, 0xa>(v));
  v = fminf(v, snb_dpp_f<0x143, 0xc>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float snb_max_f_dpp(float v)
{
  v = fmaxf(v, snb_dpp_f<0x111, 0xf>(v));
  v = fmaxf(v, snb_dpp_f<0x112, 0xf>(v));
  v = fmaxf(v, snb_dpp_f<0x114, 0xf>(v));
  v = fmaxf(v, snb_dpp_f<0x118, 0xf>(v));
  v = fmaxf(v, snb_dpp_f<0x142, 0xa>(v));
  v = fmaxf(v, snb_dpp_f<0x143, 0xc>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ int snb_mbcnt(unsigned long long m, int base)   // base + number of set bits of m below this lane
{
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
}

// Next item for this block: its XCD's contiguous slice of the Hilbert order first (one XCD's L2 sees one
// region), then whatever other slices still hold (small grids, uneven slices).  One thread calls it.
__device__ __forceinline__ int snb_claim_item(int *ctr, int n_items)
{
  const int q = n_items / kXcds, r = n_items % kXcds;
  for (int k = 0; k < kXcds; ++k) {
    const int xcd = (int)((blockIdx.x + k) % kXcds);
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int count = xcd < r ? q + 1 : q;
    if (count == 0) continue;
    if (*(volatile int *)&ctr[xcd] >= count) continue;       // exhausted (no need to bump it further)
    const int u = atomicAdd(&ctr[xcd], 1);
    if (u < count) return first + u;
  }
  return -1;
}

// Stages the candidates of the cell box [x0..x1] x [y0..y1] x [z0..z1] that pass keep() into the tile (and
// load_pay(candidate) into S.pay); every wave reads the row headers (64 rows at a time), the waves share the
// slots.  Called by all threads of the block; *n_tile (a word of S.n_tile) must be 0 and visible; the caller syncs afterwards.
// The tile's order is whatever order the waves arrive in: the lists are sorted by a total order, so it
// never shows in a result.
template <class Cfg, class Keep, class LoadPay>
__device__ __forceinline__ void snb_stage(const GridView &g, SnbLds<Cfg> &S, int *n_tile, int x0, int x1, int y0, int y1, int z0, int z1, int lane, int wave,
                                          Keep keep, LoadPay &&load_pay)
{
  constexpr int T = 64 * Cfg::kWaves;
  SnbWave<Cfg> &W = S.w[wave];
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
    const int incl = snb_scan_dpp(len);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int *w_off = W.off(), *w_beg = W.beg();
    wave_lds_fence();
    w_off[lane] = incl - len;
    w_beg[lane] = b;
    wave_lds_fence();
    for (int t0 = wave * kWave; t0 < total; t0 += T) {
      const int s = t0 + lane;
      const bool in = s < total;
      const int slot = in ? s : t0;
      int lo = 0;
#pragma unroll
      for (int step = 32; step > 0; step >>= 1)
        if (w_off[lo + step] <= slot) lo += step;
      const float4 c = g.pts[w_beg[lo] + (slot - w_off[lo])];
      const bool k = in && keep(c);
      float pv = 0.0f;
      if (Cfg::kPay && k) pv = load_pay(c);
      const unsigned long long m = ballot(k);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(n_tile, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const int d = snb_mbcnt(m, base);
        if (k && d < Cfg::kTileCap) {
          S.tx[d] = c.x; S.ty[d] = c.y; S.tz[d] = c.z; S.tw[d] = __float_as_uint(c.w);
          if (Cfg::kPay) S.pay[d] = pv;
        }
      }
    }
  }
}

typedef float snb_v2f __attribute__((ext_vector_type(2)));

template <int NB>
__device__ __forceinline__ int snb_bucket(float d2, float bscale)
{
  const int b = (int)(d2 * bscale);
  return b < NB - 1 ? b : NB - 1;
}

// Sorts ONE query's neighbours with squared distance in [lo2, hi2) (a "band" of the ball; [0, r2) = all of it):
// returns their number nh, or -1 when there are more than kHitCap, -2 when there are more than `room` (known
// after the distance tests; nothing usable is left behind in either case).
//   kInPlace:  the sorted (d2, tile slot) pairs are left in W.d2buf[0 .. nh) / W.sbuf[0 .. nh);
//   otherwise: the sorted tile slots go to list_out[0 .. nh) (the caller made sure they fit).
// (px, py, pz) is wave-uniform; n_pad = staged candidates rounded up to 128 (the padding is at kSnbFar).
template <class Cfg, bool kInPlace>
__device__ __forceinline__ int snb_sort_one(SnbLds<Cfg> &S, SnbWave<Cfg> &W, float px, float py, float pz, float lo2, float hi2, int n_pad, int lane,
                                            unsigned short *list_out, int room, SnbStats &snb_st)
{
  constexpr int R = Cfg::kHitCap / 64;           // rows of 64 hits
  constexpr int WPL = Cfg::kNB / 64;             // histogram words per lane
  constexpr int NB = Cfg::kNB;
  const float bscale = (float)NB / (hi2 - lo2);
  // the histogram is cleared here so that the stores are long done when phase B needs them
#pragma unroll
  for (int k = 0; k < WPL; ++k) W.hist[k * kWave + lane] = 0u;
  if (lane < 4) W.hist[NB + lane] = 0u;
  // A: hits, compacted; a lane tests candidates 2 l and 2 l + 1 of every 128 (FLANN's L2_Simple order
  // ((dx dx + dy dy) + dz dz), two at a time: the packed operations round each half like the scalar ones)
  SNB_TICK(t_a);
  int nh = 0;
  // (the whole ball -- lo2 = 0, every query whose list fits the hit buffer -- needs no lower test: d2 >= 0 always holds;
  // two compares and two mask operations less per 128 candidates than a band)
  auto phase_a = [&](auto banded_tag) {
    constexpr bool kBanded = decltype(banded_tag)::value;
    // Two steps of 128 candidates per iteration, straight-line: a lane that has no hit stores to the dump entry
    // instead of branching around the store, so the two steps' instruction streams interleave (a wave is a chain
    // of dependent operations; what it gains here it gains in latency, not in instruction count).
    constexpr int kDump = Cfg::kHitCap + 7;
    const snb_v2f vx = {px, px}, vy = {py, py}, vz = {pz, pz};
    for (int c0 = 0; c0 < n_pad; c0 += 4 * kWave) {
      snb_v2f d2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = c0 + u * 2 * kWave + 2 * lane;
        const snb_v2f cx = *reinterpret_cast<const snb_v2f *>(&S.tx[s]);
        const snb_v2f cy = *reinterpret_cast<const snb_v2f *>(&S.ty[s]);
        const snb_v2f cz = *reinterpret_cast<const snb_v2f *>(&S.tz[s]);
        const snb_v2f dx = vx - cx, dy = vy - cy, dz = vz - cz;
        d2[u] = (dx * dx + dy * dy) + dz * dz;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = c0 + u * 2 * kWave + 2 * lane;
        const bool h0 = d2[u].x < hi2 && (!kBanded || d2[u].x >= lo2), h1 = d2[u].y < hi2 && (!kBanded || d2[u].y >= lo2);
        const unsigned long long m0 = ballot(h0), m1 = ballot(h1);
        // (positions past the buffer are folded onto its last entry: nh > kHitCap is looked at after the loop)
        const int p0 = h0 ? min(snb_mbcnt(m0, nh), Cfg::kHitCap - 1) : kDump;
        nh += __popcll(m0);
        const int p1 = h1 ? min(snb_mbcnt(m1, nh), Cfg::kHitCap - 1) : kDump;
        nh += __popcll(m1);
        W.d2buf[p0] = d2[u].x; W.sbuf[p0] = (unsigned short)s;
        W.d2buf[p1] = d2[u].y; W.sbuf[p1] = (unsigned short)(s + 1);
      }
    }
  };
  if (lo2 > 0.0f) phase_a(std::true_type());       // wave-uniform
  else phase_a(std::false_type());
  wave_lds_fence();
  SNB_TOCK(4, t_a);
  if (nh > Cfg::kHitCap) return -1;              // wave-uniform
  if (nh > room) return -2;
  SNB_COUNT(12, nh);
  SNB_TICK(t_b);
  // B: bucket counts; a hit's arrival rank comes back with the atomic.  Rows of 64 hits, two rows in flight.
  float d2r[R];
  unsigned slot[R];
  int bkt[R], arr[R];
#pragma unroll
  for (int u0 = 0; u0 < R; u0 += 2) {
#pragma unroll
    for (int u = u0; u < u0 + 2; ++u) { d2r[u] = 0.0f; slot[u] = 0u; bkt[u] = 0; arr[u] = 0; }
    if (u0 * kWave < nh) {                       // wave-uniform
#pragma unroll
      for (int u = u0; u < u0 + 2; ++u) {
        const int h = u * kWave + lane;          // (rows past the hits read stale entries: never used)
        d2r[u] = W.d2buf[h];
        slot[u] = W.sbuf[h];
      }
#pragma unroll
      for (int u = u0; u < u0 + 2; ++u) {
        const int h = u * kWave + lane;
        bkt[u] = h < nh ? snb_bucket<NB>(d2r[u] - lo2, bscale) : NB + 1;     // (rows past the hits count in a spare word)
        arr[u] = (int)atomicAdd(&W.hist[bkt[u]], 1u);
      }
    }
  }
  wave_lds_fence();
  SNB_TOCK(5, t_b);
  SNB_TICK(t_c);
  // C: counts -> starts (lane l owns buckets [WPL l, WPL (l + 1)))
  int mb;
  {
    unsigned wd[WPL];
#pragma unroll
    for (int k = 0; k < WPL; ++k) wd[k] = W.hist[lane * WPL + k];
    int sum = 0, mx = 0;
#pragma unroll
    for (int k = 0; k < WPL; ++k) {
      const int cnt = (int)wd[k];
      wd[k] = (unsigned)sum;
      sum += cnt;
      mx = max(mx, cnt);
    }
    const int incl = snb_scan_dpp(sum);
    const unsigned ex = (unsigned)(incl - sum);
#pragma unroll
    for (int k = 0; k < WPL; ++k) W.hist[lane * WPL + k] = wd[k] + ex;
    if (lane == kWave - 1) W.hist[NB] = (unsigned)incl;         // start of the bucket past the last = nh
    mb = snb_max_dpp(mx);
  }
  wave_lds_fence();
  SNB_TOCK(6, t_c);
  SNB_TICK(t_d);
  // D: into the buckets (every lane holds its hits in registers: the buffers are rewritten in place), and
  // +inf behind the last hit: phase E reads a few entries past a bucket's end without testing
#pragma unroll
  for (int u0 = 0; u0 < R; u0 += 2) {
    if (u0 * kWave < nh) {
      int st[2];
#pragma unroll
      for (int u = u0; u < u0 + 2; ++u) st[u - u0] = (int)W.hist[bkt[u]];
#pragma unroll
      for (int u = u0; u < u0 + 2; ++u) {
        const int h = u * kWave + lane;
        const int pos = h < nh ? st[u - u0] + arr[u] : Cfg::kHitCap + 7;
        W.d2buf[pos] = d2r[u];
        W.sbuf[pos] = (unsigned short)slot[u];
      }
    }
  }
  wave_lds_fence();
  if (lane < 8) W.d2buf[nh + lane] = INFINITY;   // (after the dump stores: the last of these is the dump entry when nh = kHitCap)
  wave_lds_fence();
  SNB_TOCK(7, t_d);
  SNB_TICK(t_e);
  // E: rank inside the bucket by (d2, original index).  An entry behind the bucket's end has a larger d2 (its
  // bucket is a later one) or is +inf, so the first four bucket mates are compared without looking at the
  // bucket's length; equal distances are rare and take the slow branch.
  int fin[R];
#pragma unroll
  for (int u0 = 0; u0 < R; u0 += 2) {
#pragma unroll
    for (int u = u0; u < u0 + 2; ++u) fin[u] = 0;
    if (u0 * kWave < nh) {
      int bs[2], be[2];
#pragma unroll
      for (int u = u0; u < u0 + 2; ++u) {
        const int h = u * kWave + lane;
        d2r[u] = W.d2buf[h < nh ? h : nh];       // rows past the hits look at the +inf entry
        slot[u] = W.sbuf[h < nh ? h : 0];
      }
#pragma unroll
      for (int u = u0; u < u0 + 2; ++u) {
        const int b = snb_bucket<NB>(d2r[u] - lo2, bscale);
        bs[u - u0] = (int)W.hist[b];
        be[u - u0] = (int)W.hist[b + 1];
      }
      int r[2] = {0, 0}, eq[2] = {0, 0};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float dj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dj[j] = W.d2buf[bs[u] + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r[u] += dj[j] < d2r[u0 + u] ? 1 : 0;
          eq[u] += dj[j] == d2r[u0 + u] ? 1 : 0;
        }
      }
      for (int j = 4; j < mb; ++j) {             // wave-uniform: some bucket of this query holds more than four
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float dj = W.d2buf[min(bs[u] + j, nh)];
          r[u] += dj < d2r[u0 + u] ? 1 : 0;
          eq[u] += dj == d2r[u0 + u] ? 1 : 0;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int h = (u0 + u) * kWave + lane;
        const bool tie = eq[u] > 1 && h < nh;    // (the hit itself is one of the equal ones)
        if (ballot(tie)) {
          if (tie) {
            const unsigned idx = S.tw[slot[u0 + u]];
            for (int jj = bs[u]; jj < be[u]; ++jj)
              if (jj != h && W.d2buf[jj] == d2r[u0 + u]) r[u] += S.tw[W.sbuf[jj]] < idx ? 1 : 0;
          }
        }
        fin[u0 + u] = bs[u] + r[u];
      }
    }
  }
  wave_lds_fence();
#pragma unroll
  for (int u = 0; u < R; ++u) {
    if (u * kWave < nh) {
      const int h = u * kWave + lane;
      if (kInPlace) {
        const int pos = h < nh ? fin[u] : Cfg::kHitCap + 7;
        W.d2buf[pos] = d2r[u]; W.sbuf[pos] = (unsigned short)slot[u];
      } else if (h < nh) {
        list_out[fin[u]] = (unsigned short)slot[u];
      }
    }
  }
  wave_lds_fence();
  SNB_TOCK(8, t_e);
  return nh;
}

// Builds the sorted lists of up to kQ more of the part's n_queries queries (S.qpts), taken from the block's counter.
// Returns how many lists were built (0: the part is done): list p belongs to query W.qidx[p] and is
// W.list_off[p] .. W.list_off[p + 1] of W.arena (tile slots in (d2, original index) order).  n_pad = n_tile rounded
// up to 256: the tile is padded with candidates at kSnbFar.
// A ball with more than kHitCap neighbours is sorted in nb distance bands [r2 b / nb, r2 (b + 1) / nb), nb a power
// of two that doubles until every band fits (the list is the bands one after the other); a list that does not fit
// the arena even alone sets *overflow (the item goes to the fallback launch) and is left empty.
constexpr int kSnbMaxBands = 16;
#ifndef MM3D_SNB_MAX_PARTS
#define MM3D_SNB_MAX_PARTS 8
#endif
constexpr int kSnbMaxParts = MM3D_SNB_MAX_PARTS;
template <class Cfg>
__device__ __forceinline__ int snb_build(SnbLds<Cfg> &S, SnbWave<Cfg> &W, int n_queries, int *next_q, float r2, int n_pad, int lane, int *pending,
                                         int *budget, int *overflow, SnbStats &snb_st)
{
  int total = 0, fit = 0, nb = 1;
  if (lane == 0) W.list_off[0] = 0;
  for (int p = 0; p < Cfg::kQ; ++p) {
    // the next query of the part: the one this wave could not fit last round, else the block's counter.  The waves
    // take queries one at a time, so a wave whose lists are long simply takes fewer; a wave takes at most kQ of a
    // part (its budget): the consumer's pass costs the same for one list as for kQ, a ninth query would buy a
    // whole pass for itself.
    int qi = *pending;
    *pending = -1;
    // (some wave found a list this configuration cannot hold: the whole item goes to the overflow list and is computed again
    // there, so nothing more of it is worked here -- on a cloud that is dense everywhere that is nearly every item)
    if (*(volatile int *)overflow) break;
    if (qi < 0) {
      if (*budget <= 0) break;
      --*budget;
      if (lane == 0) qi = atomicAdd(next_q, 1);
      qi = __builtin_amdgcn_readfirstlane(qi);
    }
    if (qi >= n_queries) break;
    const float4 qp = S.qpts[qi];
    const float px = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qp.x)));
    const float py = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qp.y)));
    const float pz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qp.z)));
    // (a band's length is known after its distance tests; one that finds too little room left in the arena is not
    // sorted: the query then opens the next round)
    int len = 0, status = 0;                       // 0 sorted, 1 no room in the arena, 2 too dense even in kSnbMaxBands bands
    for (;;) {
      len = 0;
      status = 0;
      for (int b = 0; b < nb; ++b) {
        // band edges r2 * (b / nb): nb is a power of two, so the quotient is exact and edge(nb) is r2 itself
        const float lo2 = r2 * ((float)b / (float)nb), hi2 = r2 * ((float)(b + 1) / (float)nb);
        const int nh = snb_sort_one<Cfg, false>(S, W, px, py, pz, lo2, hi2, n_pad, lane, W.arena + total + len, Cfg::kArena - total - len, snb_st);
        if (nh == -2) { status = 1; break; }
        if (nh < 0) { status = 3; break; }
        len += nh;
      }
      if (status != 3) break;
      if (nb == kSnbMaxBands) { status = 2; break; }
      nb *= 2;                                     // (kept for the wave's later queries of this round: they are neighbours)

    }
    SNB_COUNT(11, 1);
    if (status == 1 && p > 0) { *pending = qi; break; }   // the arena is full: this query opens the next round
    if (lane == 0) W.qidx[p] = qi;
    if (status != 0) {                             // alone and still too long, or a dense spot: left to the fallback launch
      if (lane == 0) { *overflow = 1; W.list_off[p + 1] = total; }
      fit = p + 1;
      wave_lds_fence();
      continue;
    }
    total += len;
    fit = p + 1;
    if (lane == 0) W.list_off[p + 1] = total;
    wave_lds_fence();
  }
  return fit;
}

// Stages the box of the queries q_pts[first .. first + count) (count <= 64) into the block's tile.  Called by all
// threads; begins and ends with a block barrier (the previous tile's readers are done, the new tile is visible).
// Returns the number of staged candidates (block-uniform); more than kTileCap means "does not fit".
template <class Cfg, class LoadPay>
__device__ __forceinline__ int snb_stage_queries(const GridView &g, SnbLds<Cfg> &S, const float4 *__restrict__ q_pts, int first, int count, float ri,
                                                 int epoch, LoadPay &&load_pay, SnbStats &snb_st)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) { S.n_tile[epoch & 1] = 0; S.next_q[epoch & 1] = 0; }
  __syncthreads();
  // an earlier part of this item put it on the overflow list (snb_build): block-uniform here -- every wave reads the word
  // between this barrier and the next, and it is only written while lists are built, behind the next barrier
  if (S.overflow) return -1;
  SNB_TICK(t_stage);
  // the box (every wave computes it from the same queries)
  const bool live = lane < count;
  const float4 qa = q_pts[first + (live ? lane : 0)];
  if (wave == 0 && live) S.qpts[lane] = qa;
  const float lx = snb_min_f_dpp(live ? qa.x : INFINITY), hx = snb_max_f_dpp(live ? qa.x : -INFINITY);
  const float ly = snb_min_f_dpp(live ? qa.y : INFINITY), hy = snb_max_f_dpp(live ? qa.y : -INFINITY);
  const float lz = snb_min_f_dpp(live ? qa.z : INFINITY), hz = snb_max_f_dpp(live ? qa.z : -INFINITY);
  const int x0 = max(cell_floor(lx - ri, g.minx, g.inv), 0), x1 = min(cell_floor(hx + ri, g.minx, g.inv), g.dx - 1);
  const int y0 = max(cell_floor(ly - ri, g.miny, g.inv), 0), y1 = min(cell_floor(hy + ri, g.miny, g.inv), g.dy - 1);
  const int z0 = max(cell_floor(lz - ri, g.minz, g.inv), 0), z1 = min(cell_floor(hz + ri, g.minz, g.inv), g.dz - 1);
  snb_stage<Cfg>(g, S, &S.n_tile[epoch & 1], x0, x1, y0, y1, z0, z1, lane, wave, KeepNearBox{lx, hx, ly, hy, lz, hz, ri * ri}, load_pay);
  SNB_TOCK(2, t_stage);
  SNB_TICK(t_sb);
  __syncthreads();
  SNB_TOCK(3, t_sb);
  return S.n_tile[epoch & 1];
}

// Per-item driver: claims items, stages their boxes and hands each wave's lists to consume(fit, q, qw) -- fit lists;
// lane l is given the query of list l / kLpq in q (its list is W.list_off[l / kLpq] .. [l / kLpq + 1] when
// l / kLpq < fit) and, for the lanes l < fit that write a result per query, the query of list l in qw (.w = the
// query's original index).  An item whose box holds more candidates than the
// tile is worked in 2, 4, ... parts (runs of its queries: smaller boxes); what still does not fit -- one query's
// box alone, or a list of more than kSnbMaxBands * kHitCap neighbours -- sends the item to ov_items / ctl->ov_count.
template <class Cfg, class LoadPay, class Consume>
__device__ __forceinline__ void snb_run(const GridView &g, SnbLds<Cfg> &S, const float4 *__restrict__ q_pts, const int2 *__restrict__ items, int n_items,
                                        float radius, float r2, SnbCtl *ctl, int *__restrict__ ov_items, LoadPay &&load_pay, Consume &&consume,
                                        const int *__restrict__ sub_items = nullptr, const int *__restrict__ sub_count = nullptr)
{
  // sub_items: this launch works the *sub_count items an earlier launch (a configuration with a smaller tile) could
  // not hold -- its overflow list -- instead of all n_items
  if (sub_items) n_items = *sub_count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SnbWave<Cfg> &W = S.w[wave];
  const float ri = radius * 1.0001f + 1e-4f;
  SnbStats snb_st;
  SNB_TICK(t_all);
  int epoch = 0;                                 // stagings so far (block-uniform)
  for (;;) {
    SNB_TICK(t_claim);
    if (threadIdx.x == 0) {
      S.item = snb_claim_item(ctl->item_ctr, n_items);
      S.overflow = 0;
    }
    __syncthreads();
    if (S.item < 0) break;
    const int item = sub_items ? sub_items[S.item] : S.item;
    const int2 it = items[item];
    SNB_TOCK(1, t_claim);
    SNB_COUNT(0, 1);
    int parts = 1;
    for (int part = 0; part < parts; ++epoch) {
      const int lo = (int)((long long)it.y * part / parts), hi = (int)((long long)it.y * (part + 1) / parts);
      if (hi == lo) { ++part; continue; }
      const int n_tile = snb_stage_queries<Cfg>(g, S, q_pts, it.x + lo, hi - lo, ri, epoch, load_pay, snb_st);
      if (n_tile < 0) break;                       // abandoned (block-uniform)
      SNB_COUNT(13, n_tile);
      if (n_tile > Cfg::kTileCap) {                // block-uniform
        // Halving the run of queries shrinks the box only by the patch's share of it (the radius margin stays), and
        // every part is a staging of its own with fewer queries per wave: beyond kSnbMaxParts the item is a dense
        // spot for the fallback launch (measured: an item worked down to single queries took 3.5 ms of a 1.8 ms kernel)
        if (hi - lo == 1 || parts >= kSnbMaxParts) {
          // (the item as a whole goes on the overflow list: what its other parts would still compute here is computed
          // again there, so they are not worked -- a cloud that is dense everywhere is all such items)
          if (threadIdx.x == 0) S.overflow = 1;
          break;
        } else {                                   // the same queries again, in two halves
          parts *= 2;
          part *= 2;
        }
        continue;
      }
      // pad the tile to a multiple of 256 with candidates nobody reaches (every wave writes the same values and
      // needs only its own writes)
      const int n_pad = (n_tile + 255) & ~255;
      for (int s = n_tile + lane; s < n_pad; s += kWave) { S.tx[s] = kSnbFar; S.ty[s] = kSnbFar; S.tz[s] = kSnbFar; }
      wave_lds_fence();
      int pending = -1, budget = Cfg::kQ;
      for (;;) {
        const int fit = snb_build<Cfg>(S, W, hi - lo, &S.next_q[epoch & 1], r2, n_pad, lane, &pending, &budget, &S.overflow, snb_st);
        if (fit == 0) break;
        SNB_COUNT(14, 1);
        SNB_TICK(t_cons);
        const int pl = lane / Cfg::kLpq;
        consume(fit, S.qpts[W.qidx[pl < fit ? pl : 0]], S.qpts[W.qidx[lane < fit ? lane : 0]]);
        wave_lds_fence();
        SNB_TOCK(9, t_cons);
      }
      ++part;
    }
    SNB_TICK(t_eb);
    __syncthreads();
    SNB_TOCK(10, t_eb);
    if (threadIdx.x == 0 && S.overflow) ov_items[atomicAdd(&ctl->ov_count, 1)] = item;
#ifdef MM3D_SNB_STATS
    SNB_MAX(16, clock64() - t_claim);
    if (parts > 1) { SNB_ADD(17, 1); SNB_ADD(18, parts); }
#endif
  }
  SNB_TOCK(15, t_all);
#ifdef MM3D_SNB_STATS
  SNB_MAX(20, clock64() - t_all);
#endif
  SNB_FLUSH();
}

// host side
int snb_cu_count(int device);                    // grid.hip

template <class Cfg>
struct SnbLaunch {
  DevBuf<int> ctl;                   // SnbCtl
  DevBuf<int> ov_items;
  unsigned blocks = 0;
  // extra_lds: what the kernel declares in LDS besides SnbLds<Cfg>
  SnbLaunch(Context *c, int n_items, size_t extra_lds)
  {
    const size_t lds = sizeof(SnbLds<Cfg>) + extra_lds;
    unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)(32 / Cfg::kWaves), (size_t)163840 / lds));
    if (const char *e = getenv("MM3D_SNB_PER_CU")) per_cu = std::min<unsigned>(per_cu, (unsigned)std::max(1, atoi(e)));   // experiment: leave LDS to other streams' kernels
    const unsigned cap = (unsigned)snb_cu_count(c->device) * per_cu;
    blocks = (unsigned)std::max(1, std::min<int>(n_items, (int)cap));
    ctl = DevBuf<int>(c, sizeof(SnbCtl) / sizeof(int));
    ov_items = DevBuf<int>(c, (size_t)std::max(n_items, 1));
    MM3D_HIP(hipMemsetAsync(ctl.get(), 0, sizeof(SnbCtl), c->stream));
  }
  SnbCtl *ctl_dev() const { return reinterpret_cast<SnbCtl *>(ctl.get()); }
};

}  // namespace mm3d
