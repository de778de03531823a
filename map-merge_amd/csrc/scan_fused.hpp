// scan_fused.hpp -- transform + exclusive prefix sum + scatter in ONE launch.
//
// Several steps of the path are "flag the elements that start something, number the flagged ones, write a record per
// flagged element" (the work items of a Hilbert-ordered cloud, the first point of every voxel): three launches each, and in
// the 16-stream runs every launch, however small, waits in line behind the other streams' kernels.  This is the chained
// scan of grid.hip::k_scan_int (tickets, decoupled look-back over status words that carry the launch's epoch) with the
// input computed by `load(i)` and the result handed to `store(i, exclusive prefix, value)` instead of two arrays.
#pragma once

#include "device_util.hpp"

namespace mm3d {

// Bounding box of the points a kernel produces, reduced on the way out: every thread folds its points into (mn, mx, cnt) and
// calls flush() once, all threads of its 256-thread block together; the block's box lands in box[0..6] (the ordered-uint
// encoding and word order of grid.hip::k_bbox / cloud_bbox: min x y z, max x y z, count; initialised to FFFFFFFF x 3, 0 x 4).
// A cloud made this way carries its box and needs no k_bbox launch and no wait of its own.
struct BoxAcc {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int cnt = 0;
  __device__ __forceinline__ void add(const float4 &p)
  {
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    ++cnt;
  }
  // Called by every thread of a 256-thread block (four whole waves).  One reduction per block, and a block only touches a
  // word of the shared box when it would move it: a first version with one set of unconditional atomics per WAVE made the
  // 500 k-point centroid launch 30 x slower (55 000 atomics on one cache line).
  template <bool kSlot>
  __device__ __forceinline__ void flush_impl(unsigned *box)
  {
    __shared__ float s_mn[4][3], s_mx[4][3];
    __shared__ int s_cnt[4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
      for (int o = 32; o > 0; o >>= 1) {
        mn[a] = fminf(mn[a], __shfl_down(mn[a], o, kWave));
        mx[a] = fmaxf(mx[a], __shfl_down(mx[a], o, kWave));
      }
    const int total = wave_sum(cnt);
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
      s_cnt[wave] = total;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int n = 0;
      for (int w = 0; w < 4; ++w) n += s_cnt[w];
      if (kSlot) box[6] = (unsigned)n;
      if (n > 0 || kSlot) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          float lo = s_mn[0][a], hi = s_mx[0][a];
          for (int w = 1; w < 4; ++w) { lo = fminf(lo, s_mn[w][a]); hi = fmaxf(hi, s_mx[w][a]); }
          const unsigned ol = n > 0 ? f2ord(lo) : 0xFFFFFFFFu, oh = n > 0 ? f2ord(hi) : 0u;
          if (kSlot) { box[a] = ol; box[3 + a] = oh; continue; }
          if (ol < __hip_atomic_load(&box[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&box[a], ol);
          if (oh > __hip_atomic_load(&box[3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&box[3 + a], oh);
        }
        if (!kSlot) atomicAdd(&box[6], (unsigned)n);
      }
    }
  }
  // flush() into shared words (above) suits launches whose blocks finish one after the other (the chained scan: ~100 blocks,
  // most of which find the box already wider than theirs).  A launch whose blocks all finish together pays ~50 ns per atomic on
  // the one cache line, one after the other; flush_slot() instead leaves each block's box in its own eight words, plain stores,
  // and whoever reads them (the host, next to the count it waits for anyway: box_of_slots) folds them.
  __device__ __forceinline__ void flush(unsigned *box) { flush_impl<false>(box); }
  __device__ __forceinline__ void flush_slot(unsigned *slots) { flush_impl<true>(slots + 8 * (size_t)blockIdx.x); }
};
// host: fold `blocks` slots (eight words each, as flush_slot leaves them) into box[0..6]
inline void box_of_slots(const unsigned *slots, unsigned blocks, unsigned *box)
{
  for (int a = 0; a < 3; ++a) { box[a] = 0xFFFFFFFFu; box[3 + a] = 0u; }
  box[6] = 0u;
  for (unsigned b = 0; b < blocks; ++b) {
    const unsigned *s = slots + 8 * (size_t)b;
    for (int a = 0; a < 3; ++a) { box[a] = s[a] < box[a] ? s[a] : box[a]; box[3 + a] = s[3 + a] > box[3 + a] ? s[3 + a] : box[3 + a]; }
    box[6] += s[6];
  }
}
constexpr unsigned kBoxInit[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};

template <class Load, class Store>
__global__ void __launch_bounds__(256)
k_scan_fused(size_t n, unsigned long long *status, unsigned *ticket, unsigned ticket_base, unsigned epoch, Load load, Store store)
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
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) v[k] = base + k < n ? load(base + k) : 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) sum += v[k];
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
    // (relaxed atomics: the words carry their payload themselves, grid.hip has the measurement)
    const unsigned long long tag = (unsigned long long)epoch << 34;
    if (lane == 0 && tile > 0)
      __hip_atomic_store(&status[tile], tag | (1ull << 32) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int prefix = 0;
    for (long long hi = (long long)tile - 1; hi >= 0; hi -= kWave) {
      const long long t = hi - lane;
      unsigned long long w = 2ull << 32;                // lanes before tile 0: an empty prefix
      if (t >= 0) {
        do w = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((w >> 34) != epoch || ((w >> 32) & 3u) == 0u);
      }
      const unsigned long long is_prefix = ballot(((w >> 32) & 3u) == 2u);
      const int first = __ffsll((long long)is_prefix) - 1;
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
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) store(base + k, run, v[k]);
    run += v[k];
  }
  store.done();        // called by EVERY thread of the grid once, after its elements (a store that reduces something flushes here)
}

// load: size_t -> int (device callable), store: (size_t index, int exclusive_prefix, int value) -> void, plus done()
template <class Load, class Store>
void scan_fused(Context *c, const char *name, double bytes, size_t n, Load load, Store store)
{
  if (n == 0) return;
  const ScanLaunchState st = scan_prepare(c, n);
  MM3D_LAUNCH(c, name, bytes, (k_scan_fused<Load, Store>), dim3(st.tiles), dim3(256), 0, n, st.status, st.ticket, st.ticket_base, st.epoch, load, store);
}

}  // namespace mm3d
