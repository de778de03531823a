// Throughput of the SIFT chain's building blocks on one CU: glibc expf restated (libm_exact.hpp), the
// correctly rounded division by a constant, and the quad-ordered DPP additions, for 1..4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../map-merge_amd/csrc expf_rate.hip -o expf_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "libm_exact.hpp"
using namespace mm3d;

template <int J> __device__ __forceinline__ float quad_bcast(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), J * 0x55, 0xf, 0xf, false));
}

// MODE 0: 4 independent expf per iteration; 1: 4 fdiv_const + 4 expf; 2: the whole block (fdiv + expf + select + 32 ordered adds);
// 3: only the 32 ordered adds; 4: 4 expf via hardware v_exp_f32 (reference point, not exact)
template <int MODE>
__global__ void k(int iters, float *out, long long *cyc, float sig, float rcp)
{
  __shared__ uint64_t s_tab[32];
  if (threadIdx.x < 32) lm::exp2f_tab_copy(s_tab, threadIdx.x);
  __syncthreads();
  const int l = threadIdx.x;
  float d2[4] = {0.1f + l * 1e-3f, 0.2f + l * 1e-3f, 0.3f + l * 1e-3f, 0.4f + l * 1e-3f};
  float num = 0.f, den = 0.f, acc = 0.f;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    float w[4], vw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = -0.5f * d2[j];
      if (MODE == 1 || MODE == 2) x = lm::fdiv_const(x, sig, rcp);
      if (MODE == 4) w[j] = __expf(x);
      else if (MODE == 3) w[j] = x;
      else w[j] = lm::expf_glibc_t<false>(x, [&](unsigned t) { return s_tab[t]; });
      vw[j] = w[j] * 0.7f;
      d2[j] += 1e-4f;
    }
    if (MODE == 2 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        num = __fadd_rn(num, quad_bcast<0>(vw[j])); den = __fadd_rn(den, quad_bcast<0>(w[j]));
        num = __fadd_rn(num, quad_bcast<1>(vw[j])); den = __fadd_rn(den, quad_bcast<1>(w[j]));
        num = __fadd_rn(num, quad_bcast<2>(vw[j])); den = __fadd_rn(den, quad_bcast<2>(w[j]));
        num = __fadd_rn(num, quad_bcast<3>(vw[j])); den = __fadd_rn(den, quad_bcast<3>(w[j]));
      }
    } else {
      acc += w[0] + w[1] + w[2] + w[3];
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den + acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE> void run(const char *name)
{
  float *out; long long *cyc;
  (void)hipMalloc(&out, 4 * 1024 * 16); (void)hipMalloc(&cyc, 8);
  const int iters = 4000;
  for (int waves : {1, 4, 8, 16}) {
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, iters, out, cyc, 0.0123f, 1.0f / 0.0123f);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, iters, out, cyc, 0.0123f, 1.0f / 0.0123f);
    (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_iter = (double)h / iters;
    printf("%-44s %2d waves (%d/SIMD): %7.1f cycles per iteration per wave = %6.1f per SIMD\n", name, waves, (waves + 3) / 4, per_iter,
           per_iter / ((waves + 3) / 4));
  }
}

int main()
{
  run<0>("4 x expf_glibc");
  run<1>("4 x (fdiv_const + expf_glibc)");
  run<2>("block: 4 x (fdiv + expf) + 32 ordered adds");
  run<3>("32 ordered dpp adds only");
  run<4>("4 x hardware __expf");
  return 0;
}
