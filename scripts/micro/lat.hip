// Calibration for the LDS sorted-neighbour kernels: dependent-issue latency of one wave (f32 / f64 VALU,
// LDS read, LDS atomic with return, ds_bpermute shuffles vs DPP), and a check of the DPP wave scan / max.
// hipcc --offload-arch=gfx950 -O3 lat.hip -o lat && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int scan_dpp(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}
__device__ __forceinline__ int max_dpp(int v)
{
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(v, 63);
}
template <int J> __device__ __forceinline__ float quad_bcast(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), J * 0x55, 0xf, 0xf, false));
}

__global__ void k_check(const int *in, int *scan, int *mx, float *qb)
{
  const int l = threadIdx.x;
  scan[l] = scan_dpp(in[l]);
  mx[l] = max_dpp(in[l]);
  const float f = (float)in[l];
  qb[l * 4 + 0] = quad_bcast<0>(f); qb[l * 4 + 1] = quad_bcast<1>(f); qb[l * 4 + 2] = quad_bcast<2>(f); qb[l * 4 + 3] = quad_bcast<3>(f);
}

// mode 0 f32 fma chain, 1 f64 fma chain, 2 LDS read chain, 3 LDS atomic-return chain, 4 shfl_up chain, 5 dpp scan chain, 6 f32 add with quad dpp
template <int MODE>
__global__ void k_lat(int iters, float *out, long long *cyc)
{
  __shared__ int lds[1024];
  const int l = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (i * 17 + 5) & 1023;
  __syncthreads();
  float f = 1.0f + l * 1e-3f; double d = 1.0 + l * 1e-3; int x = l;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) f = __builtin_fmaf(f, 1.0000001f, 1e-7f);
      if (MODE == 1) d = __builtin_fma(d, 1.0000001, 1e-7);
      if (MODE == 2) x = lds[x];
      if (MODE == 3) x = atomicAdd(&lds[x & 1023], 1) & 1023;
      if (MODE == 4) x += __shfl_up(x, 1, 64);
      if (MODE == 5) x = scan_dpp(x) & 1023;
      if (MODE == 6) f = f + quad_bcast<1>(f * 0.5f);
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = f + (float)d + (float)x;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves)
{
  float *out; long long *cyc;
  hipMalloc(&out, 4 * 64 * 16 * 1024); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k_lat<MODE>, dim3(1), dim3(64 * waves), 0, 0, iters, out, cyc);
  hipLaunchKernelGGL(k_lat<MODE>, dim3(1), dim3(64 * waves), 0, 0, iters, out, cyc);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s waves/block %2d (=%d per SIMD): %.1f clock64 ticks per op (per wave)\n", name, waves, (waves + 3) / 4, (double)h / (iters * 16.0));
  hipFree(out); hipFree(cyc);
}

int main()
{
  int hin[64], hs[64], hm[64]; float hq[256];
  for (int i = 0; i < 64; ++i) hin[i] = (i * 7 + 3) % 11;
  int *din, *ds, *dm; float *dq;
  hipMalloc(&din, 256); hipMalloc(&ds, 256); hipMalloc(&dm, 256); hipMalloc(&dq, 1024);
  hipMemcpy(din, hin, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, din, ds, dm, dq);
  hipMemcpy(hs, ds, 256, hipMemcpyDeviceToHost); hipMemcpy(hm, dm, 256, hipMemcpyDeviceToHost); hipMemcpy(hq, dq, 1024, hipMemcpyDeviceToHost);
  int acc = 0, bad = 0, mx = 0;
  for (int i = 0; i < 64; ++i) { acc += hin[i]; mx = hin[i] > mx ? hin[i] : mx; if (hs[i] != acc) ++bad; }
  for (int i = 0; i < 64; ++i) if (hm[i] != mx) ++bad;
  for (int i = 0; i < 64; ++i) for (int j = 0; j < 4; ++j) if (hq[i * 4 + j] != (float)hin[(i & ~3) + j]) ++bad;
  printf("dpp scan / max / quad_bcast check: %s (%d bad)\n", bad ? "FAILED" : "ok", bad);
  int dev_clock = 0; hipDeviceGetAttribute(&dev_clock, hipDeviceAttributeWallClockRate, 0);
  for (int w : {1, 4, 8, 16}) {
    run<0>("f32 fma dependent chain", w);
    run<1>("f64 fma dependent chain", w);
    run<2>("LDS read dependent chain", w);
    run<3>("LDS atomic-return chain", w);
    run<4>("shfl_up (ds_bpermute) chain", w);
    run<5>("dpp scan (6 dpp adds)", w);
    run<6>("f32 mul + add with quad dpp", w);
  }
  return bad ? 1 : 0;
}
