/*
 * o_libm.c -- the host libm's expf / atanf / sinf / cosf / atan2f over arrays (TEST INFRASTRUCTURE).
 *
 * The CPU path of the reference (PCL) reaches these functions through libm; the device path uses
 * restatements of them (map-merge_amd/csrc/libm_exact.hpp).  The tests evaluate both on the same
 * arguments and compare bits.  (numpy's float32 ufuncs are its own SIMD code, not libm.)
 */
#include "mm3d_oracle.h"

#include <math.h>

void mo_libm_eval(int fn, const float *x, const float *y, int n, float *out)
{
  for (int i = 0; i < n; ++i) {
    switch (fn) {
      case 0: out[i] = expf(x[i]); break;
      case 1: out[i] = atanf(x[i]); break;
      case 2: out[i] = sinf(x[i]); break;
      case 3: out[i] = cosf(x[i]); break;
      default: out[i] = atan2f(y[i], x[i]); break;
    }
  }
}
