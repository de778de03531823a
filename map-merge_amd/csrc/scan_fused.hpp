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
}

// load: size_t -> int (device callable), store: (size_t index, int exclusive_prefix, int value) -> void
template <class Load, class Store>
void scan_fused(Context *c, const char *name, double bytes, size_t n, Load load, Store store)
{
  if (n == 0) return;
  const ScanLaunchState st = scan_prepare(c, n);
  MM3D_LAUNCH(c, name, bytes, (k_scan_fused<Load, Store>), dim3(st.tiles), dim3(256), 0, n, st.status, st.ticket, st.ticket_base, st.epoch, load, store);
}

}  // namespace mm3d
