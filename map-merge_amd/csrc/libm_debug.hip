// libm_debug.hip -- test hook: evaluates the restated glibc functions (libm_exact.hpp) on the device, so the
// tests can hold the device's results against the host's libm argument by argument.
#include "device_util.hpp"

namespace mm3d {

__global__ void k_debug_libm(int fn, const float *__restrict__ x, const float *__restrict__ y, int n, float *__restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r;
  switch (fn) {
    case 0: r = lm::expf_glibc(x[i]); break;
    case 1: r = lm::atanf_glibc(x[i]); break;
    case 2: r = lm::sinf_glibc(x[i]); break;
    case 3: r = lm::cosf_glibc(x[i]); break;
    case 5: r = __builtin_amdgcn_exp2f(x[i]); break;       // v_exp_f32 as the certified SIFT pass uses it (sift_cert.hpp)
    default: r = lm::atan2f_glibc(y[i], x[i]); break;
  }
  out[i] = r;
}

void debug_libm(Context *c, int fn, const float *x_host, const float *y_host, int n, float *out_host)
{
  if (n <= 0) return;
  DevBuf<float> x(c, n), y(c, n), o(c, n);
  MM3D_HIP(hipMemcpyAsync(x.get(), x_host, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  if (y_host) MM3D_HIP(hipMemcpyAsync(y.get(), y_host, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  MM3D_LAUNCH(c, "debug_libm", 0, k_debug_libm, dim3(div_up(n, 256)), dim3(256), 0, fn, (const float *)x.get(), (const float *)y.get(), n, o.get());
  MM3D_HIP(hipMemcpyAsync(out_host, o.get(), (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  c->sync();
}

}  // namespace mm3d
