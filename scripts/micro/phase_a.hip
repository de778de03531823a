// Latency of the distance-test + compaction step of snb_lds.hpp (phase A) on one CU, in a few variants.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off phase_a.hip -o phase_a
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kTile = 1024, kHitCap = 256;
__device__ __forceinline__ int mbcnt(unsigned long long m, int base)
{
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
}
// VAR 0: branches around the stores (v3); 1: dump-slot stores, two steps unrolled; 2: no stores at all (count only);
// 3: per-lane hit bitmask, no compaction in the loop; 4: as 0 with the next step's loads issued first
template <int VAR>
__global__ void k(int iters, int n_pad, float r2, int *out, long long *cyc)
{
  __shared__ float tx[kTile], ty[kTile], tz[kTile];
  __shared__ float d2buf[16][kHitCap + 8];
  __shared__ unsigned short sbuf[16][kHitCap + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kTile; i += blockDim.x) {
    unsigned h = i * 2654435761u;
    tx[i] = (h & 1023) * (2.2f / 1024); ty[i] = ((h >> 10) & 1023) * (2.2f / 1024); tz[i] = ((h >> 20) & 255) * (0.6f / 256);
  }
  __syncthreads();
  float *db = d2buf[wave];
  unsigned short *sb = sbuf[wave];
  int acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const float px = 0.6f + (it & 7) * 0.1f, py = 0.7f + ((it >> 3) & 7) * 0.1f, pz = 0.3f;
    const v2f vx = {px, px}, vy = {py, py}, vz = {pz, pz};
    int nh = 0;
    unsigned mask = 0;
    if (VAR == 4) {
      int s = 2 * lane;
      v2f cx = *(const v2f *)&tx[s], cy = *(const v2f *)&ty[s], cz = *(const v2f *)&tz[s];
      for (int c0 = 0; c0 < n_pad; c0 += 128) {
        const int sn = (c0 + 128 < n_pad ? c0 + 128 : c0) + 2 * lane;
        const v2f nx = *(const v2f *)&tx[sn], ny = *(const v2f *)&ty[sn], nz = *(const v2f *)&tz[sn];
        const v2f dx = vx - cx, dy = vy - cy, dz = vz - cz;
        const v2f d2 = (dx * dx + dy * dy) + dz * dz;
        const bool h0 = d2.x < r2, h1 = d2.y < r2;
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
        const int p0 = min(mbcnt(m0, nh), kHitCap - 1); nh += __popcll(m0);
        const int p1 = min(mbcnt(m1, nh), kHitCap - 1); nh += __popcll(m1);
        if (h0) { db[p0] = d2.x; sb[p0] = (unsigned short)(c0 + 2 * lane); }
        if (h1) { db[p1] = d2.y; sb[p1] = (unsigned short)(c0 + 2 * lane + 1); }
        cx = nx; cy = ny; cz = nz;
      }
    } else if (VAR == 1) {
      for (int c0 = 0; c0 < n_pad; c0 += 256) {
        v2f d2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int s = c0 + u * 128 + 2 * lane;
          const v2f cx = *(const v2f *)&tx[s], cy = *(const v2f *)&ty[s], cz = *(const v2f *)&tz[s];
          const v2f dx = vx - cx, dy = vy - cy, dz = vz - cz;
          d2[u] = (dx * dx + dy * dy) + dz * dz;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int s = c0 + u * 128 + 2 * lane;
          const bool h0 = d2[u].x < r2, h1 = d2[u].y < r2;
          const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
          const int p0 = h0 ? min(mbcnt(m0, nh), kHitCap - 1) : kHitCap + 7; nh += __popcll(m0);
          const int p1 = h1 ? min(mbcnt(m1, nh), kHitCap - 1) : kHitCap + 7; nh += __popcll(m1);
          db[p0] = d2[u].x; sb[p0] = (unsigned short)s;
          db[p1] = d2[u].y; sb[p1] = (unsigned short)(s + 1);
        }
      }
    } else {
      for (int c0 = 0; c0 < n_pad; c0 += 128) {
        const int s = c0 + 2 * lane;
        const v2f cx = *(const v2f *)&tx[s], cy = *(const v2f *)&ty[s], cz = *(const v2f *)&tz[s];
        const v2f dx = vx - cx, dy = vy - cy, dz = vz - cz;
        const v2f d2 = (dx * dx + dy * dy) + dz * dz;
        const bool h0 = d2.x < r2, h1 = d2.y < r2;
        if (VAR == 3) {
          mask = (mask << 2) | (h0 ? 1u : 0u) | (h1 ? 2u : 0u);
        } else {
          const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
          const int p0 = min(mbcnt(m0, nh), kHitCap - 1); nh += __popcll(m0);
          const int p1 = min(mbcnt(m1, nh), kHitCap - 1); nh += __popcll(m1);
          if (VAR == 0) {
            if (h0) { db[p0] = d2.x; sb[p0] = (unsigned short)s; }
            if (h1) { db[p1] = d2.y; sb[p1] = (unsigned short)(s + 1); }
          } else {
            acc += p0 + p1;
          }
        }
      }
    }
    acc += nh + __popc(mask);
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (int)db[lane] + sb[lane];
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int VAR> void run(const char *name)
{
  int *out; long long *cyc;
  (void)hipMalloc(&out, 4 * 1024 * 16); (void)hipMalloc(&cyc, 8);
  const int iters = 2000, n_pad = 512;
  for (int waves : {1, 4, 8, 16}) {
    hipLaunchKernelGGL(k<VAR>, dim3(1), dim3(64 * waves), 0, 0, iters, n_pad, 0.36f, out, cyc);
    hipLaunchKernelGGL(k<VAR>, dim3(1), dim3(64 * waves), 0, 0, iters, n_pad, 0.36f, out, cyc);
    (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_step = (double)h / iters / (n_pad / 128);
    printf("%-52s %2d waves (%d/SIMD): %6.1f cycles per 128-candidate step per wave = %6.1f per SIMD\n", name, waves, (waves + 3) / 4, per_step,
           per_step / ((waves + 3) / 4));
  }
}
int main()
{
  run<0>("0 branches around the stores");
  run<1>("1 dump-slot stores, two steps unrolled");
  run<2>("2 no stores (positions only)");
  run<3>("3 per-lane bitmask, no compaction");
  run<4>("4 as 0, next step's loads first");
  return 0;
}
