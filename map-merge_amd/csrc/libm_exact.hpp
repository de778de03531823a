// libm_exact.hpp -- glibc's expf / atanf / atan2f / sinf / cosf restated as ONE host + device source.
//
// The CPU path of the reference (PCL on glibc) rounds these functions the way glibc's libm does; the
// device's own libm (ROCm ocml) differs from it by an ulp on a fraction of the inputs, and one such ulp
// moves a SIFT extremum, a Darboux angle across a histogram bin edge or an eigenvector of a normal --
// after which every RNG-driven stage downstream deals different samples.  These restatements follow
// the algorithms of glibc 2.35 (the libc of this image, which the CPU oracle links):
//
//   expf          sysdeps/ieee754/flt-32/e_expf.c + e_exp2f_data.c (Szabolcs Nagy's table + cubic in
//                 double; N = 32 table entries).  x86_64 selects the FMA build of that file at load
//                 time (sysdeps/x86_64/fpu/multiarch/e_expf.c), in which the compiler contracts the
//                 polynomial's a*b+c steps: they are spelled fma() here.
//   sinf, cosf    sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h, s_sincosf_data.c (double
//                 polynomials after a fast reduction by pi/2; also an FMA build on x86_64).  Only the
//                 |x| < 120 paths are restated: the callers' arguments are angles in [-pi, pi].
//   atanf, atan2f sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c (fdlibm, pure float arithmetic, no
//                 multiarch build: every operation is a separately rounded float operation).
//
// The library is compiled with -ffp-contract=off, so the only fused operations are the explicit
// fma() calls.  scripts/libm_sweep.cpp compares every function with the host's libm over ALL 2^32
// float arguments (atan2f: a dense sample of pairs); tests/test_libm_exact.py runs a strided sweep on
// the CPU and a device-vs-host sweep through mm3d_debug_libm on the GPU.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define MM3D_HD __host__ __device__ __forceinline__
#else
#define MM3D_HD inline
#endif

namespace mm3d {
namespace lm {

MM3D_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
MM3D_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
MM3D_HD uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
MM3D_HD double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
MM3D_HD float fabsf_(float x) { return u2f(f2u(x) & 0x7fffffffu); }

// ---- expf ------------------------------------------------------------------------------------------
// tab[i] = bits(2^(i/32)) - (i << 47)   (e_exp2f_data.c; scripts/libm_sweep.cpp regenerates and checks it)
MM3D_HD uint64_t exp2f_tab(unsigned i)
{
  constexpr uint64_t T[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
      0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
      0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
      0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
      0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
      0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
      0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
  return T[i & 31u];
}

// the table as data, for kernels that keep it in LDS (a per-lane index into a constant array would be a
// global load per call)
MM3D_HD void exp2f_tab_copy(uint64_t *dst, int i) { dst[i] = exp2f_tab((unsigned)i); }

// kCheck = false: the caller guarantees |x| < 88 (no overflow / underflow / NaN handling needed)
template <bool kCheck = true, class Tab>
MM3D_HD float expf_glibc_t(float x, Tab &&tab)
{
  const double xd = (double)x;
  if (kCheck) {
    const uint32_t abstop = (f2u(x) >> 20) & 0x7ffu;
    if (abstop >= (0x42b00000u >> 20)) {                     // |x| >= 88 or NaN
      if (f2u(x) == 0xff800000u) return 0.0f;                // -inf
      if (abstop >= (0x7f800000u >> 20)) return x + x;       // +inf, NaN
      if (x > 88.72283172607421875f) return INFINITY;        // 0x1.62e42ep6f: overflow
      if (x < -103.972076416015625f) return 0.0f;            // -0x1.9fe368p6f: underflow to zero
    }
  }
  constexpr double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
  constexpr double Shift = 0x1.8p+52;
  constexpr double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
  constexpr double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
  constexpr double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
  double z = InvLn2N * xd;
  double kd = z + Shift;
  const uint64_t ki = d2u(kd);
  kd -= Shift;
  const double r = fma(InvLn2N, xd, -kd);                    // the FMA build contracts z - kd with z's product
  uint64_t t = tab((unsigned)(ki & 31u));
  t += ki << (52 - 5);
  const double s = u2d(t);
  z = fma(C0, r, C1);
  const double r2 = r * r;
  double y = fma(C2, r, 1.0);
  y = fma(z, r2, y);
  y = y * s;
  return (float)y;
}

MM3D_HD float expf_glibc(float x)
{
  return expf_glibc_t<true>(x, [](unsigned i) { return exp2f_tab(i); });
}

// ---- a / b, correctly rounded, for a divisor known in advance --------------------------------------
// With rcp = RN(1 / b) prepared on the host: q0 = RN(a * rcp) is within 2 ulp of a / b, one Newton step on
// the exact remainder (fma) makes it faithful, a second one rounds it correctly (Markstein, "Computation of
// elementary functions on the IBM RISC System/6000 processor", Theorem: the final fma rounds correctly when the
// reciprocal is the correctly rounded one and the quotient estimate is within one ulp).  Five operations
// instead of the hardware's division macro (ten).  Valid away from overflow / underflow of a / b and of the
// remainders, which is what fdiv_const_ok() checks for the callers' ranges; tests/test_libm_exact.py sweeps it
// against IEEE division.
MM3D_HD float fdiv_const(float a, float b, float rcp)
{
  float q = a * rcp;
  float r = fmaf(-b, q, a);
  q = fmaf(r, rcp, q);
  r = fmaf(-b, q, a);
  return fmaf(r, rcp, q);
}

// ---- sinf / cosf (|x| < 120) -----------------------------------------------------------------------
MM3D_HD float sinf_poly(double x, double x2, bool neg_table, int n)
{
  // __sincosf_table[0] / [1]: the second table is the first with the cosine coefficients negated
  const double sg = neg_table ? -1.0 : 1.0;
  const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5;
  const double c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    const double x3 = x * x2;
    const double s1_ = fma(x2, s3, s2);
    const double x7 = x3 * x2;
    const double s = fma(x3, s1, x);
    return (float)fma(x7, s1_, s);
  }
  const double x4 = x2 * x2;
  const double c2_ = fma(x2, c4, c3);
  const double c1_ = fma(x2, c1, c0);
  const double x6 = x4 * x2;
  const double c = fma(x4, c2, c1_);
  return (float)fma(x6, c2_, c);
}

MM3D_HD double sincos_reduce_fast(double x, int *np)
{
  constexpr double hpi_inv = 0x1.45F306DC9C883p+23;   // 2/pi * 2^24: the quadrant ends up in bits 24..31
  constexpr double hpi = 0x1.921FB54442D18p0;
  const double r = x * hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return fma(-(double)n, hpi, x);
}

MM3D_HD float sinf_glibc(float y)
{
  double x = (double)y;
  const uint32_t top = (f2u(y) >> 20) & 0x7ffu;
  if (top < (0x3f490fdbu >> 20)) {                 // |y| < pi/4 (compared on the top 12 bits)
    const double s = x * x;
    if (top < (0x39800000u >> 20)) return y;       // |y| < 2^-12
    return sinf_poly(x, s, false, 0);
  }
  if (top < (0x42f00000u >> 20)) {                 // |y| < 120
    int n;
    x = sincos_reduce_fast(x, &n);
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;   // sign[n & 3] = {1, -1, -1, 1}
    return sinf_poly(x * sgn, x * x, (n & 2) != 0, n);
  }
  return sinf(y);                                  // not restated: the callers never get here
}

MM3D_HD float cosf_glibc(float y)
{
  double x = (double)y;
  const uint32_t top = (f2u(y) >> 20) & 0x7ffu;
  if (top < (0x3f490fdbu >> 20)) {
    const double s = x * x;
    if (top < (0x39800000u >> 20)) return 1.0f;
    return sinf_poly(x, s, false, 1);
  }
  if (top < (0x42f00000u >> 20)) {
    int n;
    x = sincos_reduce_fast(x, &n);
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    // cosf: p = table[(n >> 1) & 1] after n -> n + 1 ... expressed on the original n:
    // s = sign[n & 3]; if (n & 2) p = table[1]; return sinf_poly(x * s, x * x, p, n ^ 1)
    return sinf_poly(x * sgn, x * x, (n & 2) != 0, n ^ 1);
  }
  return cosf(y);
}

// ---- atanf / atan2f (fdlibm float) -----------------------------------------------------------------
MM3D_HD float atanf_glibc(float x)
{
  // the decimal literals of s_atanf.c (its hex comments are not all exact)
  const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                        6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
  const int32_t hx = (int32_t)f2u(x);
  const int32_t ix = hx & 0x7fffffff;
  if (ix >= 0x4c000000) {                          // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;             // NaN
    return hx > 0 ? 1.5707962513e+00f + 7.5497894159e-08f : -1.5707962513e+00f - 7.5497894159e-08f;
  }
  if (ix < 0x31000000) return x;                   // |x| < 2^-29
  // s_atanf.c reduces by range -- (2x-1)/(2+x), (x-1)/(x+1), (x-1.5)/(1+1.5x), -1/x, or x itself below 0.4375 --
  // in four branches; here the numerator and the denominator are picked by range and divided ONCE (a wave's lanes
  // fall into all the ranges: every branch would run, each with a division of its own).  x / 1 is x, and the
  // function is odd in every step (rounding is symmetric), so the work is done on |x| and the sign put back.
  const float ax = fabsf_(x);
  const bool r_lo = ix < 0x3ee00000, r0 = ix < 0x3f300000, r1 = ix < 0x3f980000, r2 = ix < 0x401c0000;
  const float num = r_lo ? ax : (r0 ? 2.0f * ax - 1.0f : (r1 ? ax - 1.0f : (r2 ? ax - 1.5f : -1.0f)));
  const float den = r_lo ? 1.0f : (r0 ? 2.0f + ax : (r1 ? ax + 1.0f : (r2 ? 1.0f + 1.5f * ax : ax)));
  const float hi = r0 ? 4.6364760399e-01f : (r1 ? 7.8539812565e-01f : (r2 ? 9.8279368877e-01f : 1.5707962513e+00f));
  const float lo = r0 ? 5.0121582440e-09f : (r1 ? 3.7748947079e-08f : (r2 ? 3.4473217170e-08f : 7.5497894159e-08f));
  const float t = num / den;
  const float z = t * t;
  const float w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  const float ts = t * (s1 + s2);
  const float r = r_lo ? t - ts : hi - ((ts - lo) - t);
  return hx < 0 ? -r : r;
}

// the zeros, infinities, NaNs and x == 1 of e_atan2f.c, in its own order
MM3D_HD float atan2f_glibc_special(float y, float x)
{
  const float tiny = 1.0e-30f;
  const float pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f;
  const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;          // NaN
  if (hx == 0x3f800000) return atanf_glibc(y);                   // x = 1
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);             // 2 * sign(x) + sign(y)
  if (iy == 0) {
    switch (m) {
      case 0: case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    }
    switch (m) {
      case 0: return 0.0f;
      case 1: return -0.0f;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;               // y = +-inf, x finite
}

MM3D_HD float atan2f_glibc(float y, float x)
{
  const float pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  // one test for every special operand (zero, infinity, NaN on either side, x == 1): ix - 1 wraps for zero
  if ((uint32_t)(ix - 1) >= 0x7f7fffffu || (uint32_t)(iy - 1) >= 0x7f7fffffu || hx == 0x3f800000) return atan2f_glibc_special(y, x);
  const int32_t k = (iy - ix) >> 23;
  float z = atanf_glibc(fabsf_(y / x));
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                          // |y / x| > 2^60
  else if (hx < 0 && k < -60) z = 0.0f;                           // |y| / x < -2^60
  // quadrants: z | -z | pi - (z - pi_lo) | (z - pi_lo) - pi; the last is the third negated (z - pi_lo never equals pi)
  const float q = hx < 0 ? pi - (z - pi_lo) : z;
  return u2f(f2u(q) ^ ((uint32_t)hy & 0x80000000u));
}

}  // namespace lm
}  // namespace mm3d
