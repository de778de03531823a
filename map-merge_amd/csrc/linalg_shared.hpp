// linalg_shared.hpp -- the Umeyama core (3x3 Jacobi SVD in double) as ONE host+device source.
//
// The same instruction sequence runs on the host (RANSAC models, inlier refit) and on the device
// (ICP finalize, SAC-IA hypothesis models).  With -ffp-contract=off and IEEE sqrt/div on both sides
// the results are bit-identical, which is what lets hypotheses be built on the device and still
// match the sequential CPU loop.
//
// Umeyama = pcl::umeyama == Eigen::umeyama(src, dst, with_scaling = false) behind
// pcl::registration::TransformationEstimationSVD (R/src/matching.cpp:135-137, SAC-IA, ICP).
#pragma once

#include <hip/hip_runtime.h>

namespace mm3d {

// 3x3 SVD by one-sided Jacobi (Hestenes), double; A row-major; singular values descending
__host__ __device__ inline void svd3_shared(const double *A, double *U, double *S, double *V)
{
  double B[9];
  double frob2 = 0.0;
  for (int i = 0; i < 9; ++i) { B[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; frob2 += A[i] * A[i]; }
  // Eigen's JacobiSVD leaves a 2x2 block alone once its off-diagonals are <= 2 eps * (largest
  // diagonal entry); for the one-sided form that is |col_p . col_q| <= 2 eps |A| max(|col_p|, |col_q|).
  // (A threshold relative to |col_p| |col_q| never settles on the rank-2 matrices three-point
  // samples produce: the null column is pure rounding noise.)
  const double thr = 2.0 * 2.220446049250313e-16 * sqrt(frob2);
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int k = 0; k < 3; ++k) {
      const int p = (k == 2) ? 1 : 0, q = (k == 0) ? 1 : 2;
      double alpha = 0, beta = 0, gamma = 0;
      for (int i = 0; i < 3; ++i) {
        alpha += B[i * 3 + p] * B[i * 3 + p];
        beta += B[i * 3 + q] * B[i * 3 + q];
        gamma += B[i * 3 + p] * B[i * 3 + q];
      }
      if (gamma == 0.0 || fabs(gamma) <= thr * sqrt(alpha > beta ? alpha : beta)) continue;
      rotated = 1;
      const double zeta = (beta - alpha) / (2.0 * gamma);
      const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
      for (int i = 0; i < 3; ++i) {
        const double bp = B[i * 3 + p], bq = B[i * 3 + q];
        B[i * 3 + p] = c * bp - s * bq;
        B[i * 3 + q] = s * bp + c * bq;
        const double vp = V[i * 3 + p], vq = V[i * 3 + q];
        V[i * 3 + p] = c * vp - s * vq;
        V[i * 3 + q] = s * vp + c * vq;
      }
    }
    if (!rotated) break;
  }
  double nrm[3];
  for (int j = 0; j < 3; ++j) nrm[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
  // stable descending order of the three norms (same comparisons as a selection sort a<b)
  int o0 = 0, o1 = 1, o2 = 2, t;
  if (nrm[o1] > nrm[o0]) { t = o0; o0 = o1; o1 = t; }
  if (nrm[o2] > nrm[o0]) { t = o0; o0 = o2; o2 = t; }
  if (nrm[o2] > nrm[o1]) { t = o1; o1 = o2; o2 = t; }
  const int ord[3] = {o0, o1, o2};
  double Vs[9];
  for (int j = 0; j < 3; ++j) {
    S[j] = nrm[ord[j]];
    for (int i = 0; i < 3; ++i) {
      Vs[i * 3 + j] = V[i * 3 + ord[j]];
      U[i * 3 + j] = (S[j] > 0.0) ? B[i * 3 + ord[j]] / S[j] : 0.0;
    }
  }
  for (int i = 0; i < 9; ++i) V[i] = Vs[i];
  const double tiny = 1e-14 * (S[0] > 0 ? S[0] : 1.0);
  if (S[0] <= 0.0) {
    for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  if (S[1] <= tiny) {
    const double u0[3] = {U[0], U[3], U[6]};
    const int m = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    const double d = u0[m];
    const double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
    const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    U[1] = v[0] / n; U[4] = v[1] / n; U[7] = v[2] / n;
  }
  if (S[2] <= tiny) {
    const double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
    U[2] = a[1] * b[2] - a[2] * b[1];
    U[5] = a[2] * b[0] - a[0] * b[2];
    U[8] = a[0] * b[1] - a[1] * b[0];
  }
}

__host__ __device__ inline double det3_shared(const double *M)
{
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// Eq. (39)-(43) of Umeyama as pcl::umeyama writes them; prec = NumTraits<Scalar>::dummy_precision()
__host__ __device__ inline void umeyama_core_shared(const double *sigma, const double *sm, const double *dm, double prec,
                                                    double *R, double *t)
{
  double U[9], S[3], V[9];
  svd3_shared(sigma, U, S, V);
  double Sd[3] = {1.0, 1.0, 1.0};
  if (det3_shared(sigma) < 0) Sd[2] = -1.0;
  int rank = 0;
  for (int i = 0; i < 3; ++i)
    if (!(fabs(S[i]) <= fabs(S[0]) * prec)) ++rank;
  if (rank == 2) Sd[2] = (det3_shared(U) * det3_shared(V) > 0) ? 1.0 : -1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += U[i * 3 + k] * Sd[k] * V[j * 3 + k];
      R[i * 3 + j] = acc;
    }
  for (int i = 0; i < 3; ++i)
    t[i] = dm[i] - (R[i * 3 + 0] * sm[0] + R[i * 3 + 1] * sm[1] + R[i * 3 + 2] * sm[2]);
}

// float instantiation (TransformationEstimationSVD<PointXYZRGB, PointXYZRGB, float>): means, demeaning
// and sigma accumulate in float, sequentially; src/dst are n x 3, T column-major 4x4
__host__ __device__ inline void umeyama_f32_shared(const float *src, const float *dst, int n, float *T)
{
  float sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) { sm[a] += src[i * 3 + a]; dm[a] += dst[i * 3 + a]; }
  const float one_over_n = 1.0f / (float)n;
  for (int a = 0; a < 3; ++a) { sm[a] *= one_over_n; dm[a] *= one_over_n; }
  float sg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const float s[3] = {src[i * 3] - sm[0], src[i * 3 + 1] - sm[1], src[i * 3 + 2] - sm[2]};
    const float d[3] = {dst[i * 3] - dm[0], dst[i * 3 + 1] - dm[1], dst[i * 3 + 2] - dm[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) sg[r * 3 + c] += d[r] * s[c];
  }
  double sigma[9], smd[3], dmd[3], R[9], t[3];
  for (int i = 0; i < 9; ++i) sigma[i] = (double)(sg[i] * one_over_n);
  for (int a = 0; a < 3; ++a) { smd[a] = sm[a]; dmd[a] = dm[a]; }
  umeyama_core_shared(sigma, smd, dmd, 1e-5, R, t);
  for (int i = 0; i < 16; ++i) T[i] = 0.0f;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[c * 4 + r] = (float)R[r * 3 + c];
    T[12 + r] = (float)t[r];
  }
  T[15] = 1.0f;
}

}  // namespace mm3d
