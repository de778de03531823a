// Host-side launch / sync throughput of the HIP runtime from T threads on T streams (what bounds a job of many
// small maps).  hipcc --offload-arch=gfx950 -O2 -o launch_rate launch_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 1000) *p = 1; }
__global__ void k_spin(int *p, int n)
{
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < n) {}
  if (p && threadIdx.x == 1000) *p = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
  for (int T : {1, 2, 4, 8, 16}) {
    for (int mode = 0; mode < 6; ++mode) {
      // 0: launches only, sync at end; 1: sync after every 4 launches; 2: 4 launches + 4-byte pageable D2H + sync
      // 3: 4 launches of a 20 us kernel + sync; 4: the same with a 100 us kernel; 5: 100 us kernels of 1024 blocks x 256
      const int iters = mode >= 4 ? 300 : 2000;
      std::vector<std::thread> th;
      std::vector<hipStream_t> st(T);
      std::vector<int *> d(T);
      for (int t = 0; t < T; ++t) { hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking); hipMalloc(&d[t], 64); }
      hipDeviceSynchronize();
      double t0 = now();
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
          hipSetDevice(0);
          int h = 0;
          for (int i = 0; i < iters; ++i) {
            for (int j = 0; j < 4; ++j) {
              if (mode == 3) k_spin<<<64, 64, 0, st[t]>>>(d[t], 2000);
              else if (mode == 4) k_spin<<<64, 64, 0, st[t]>>>(d[t], 10000);
              else if (mode == 5) k_spin<<<1024, 256, 0, st[t]>>>(d[t], 10000);
              else k_empty<<<1, 64, 0, st[t]>>>(d[t]);
            }
            if (mode == 2) hipMemcpyAsync(&h, d[t], 4, hipMemcpyDeviceToHost, st[t]);
            if (mode >= 1) hipStreamSynchronize(st[t]);
          }
          hipStreamSynchronize(st[t]);
        });
      for (auto &x : th) x.join();
      double dt = now() - t0;
      printf("threads %2d mode %d: %.1f k launches/s total, %.2f us per group of 4 per thread\n", T, mode, T * iters * 4 / dt / 1e3,
             dt / iters * 1e6);
      for (int t = 0; t < T; ++t) { hipStreamDestroy(st[t]); hipFree(d[t]); }
    }
  }
  return 0;
}
