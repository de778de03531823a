// linalg.cpp -- small dense host helpers of the registration host code.
//
// Umeyama (pcl::umeyama == Eigen::umeyama without scaling) behind
// pcl::registration::TransformationEstimationSVD (R/src/matching.cpp:135-137, and inside SAC-IA),
// in float, and behind SampleConsensusModelRegistration::estimateRigidTransformationSVD in double;
// Eigen::Matrix4f::inverse() / operator* used by computeGlobalTransforms (R/src/map_merging.cpp:137-186).
#include <cmath>
#include <cstring>

#include "device_util.hpp"

namespace mm3d {

namespace {

// 3x3 SVD by one-sided Jacobi (double); A row-major; singular values descending
void svd3(const double A[9], double U[9], double S[3], double V[9])
{
  double B[9];
  std::memcpy(B, A, sizeof(B));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int k = 0; k < 3; ++k) {
      const int p = P[k], q = Q[k];
      double alpha = 0, beta = 0, gamma = 0;
      for (int i = 0; i < 3; ++i) {
        alpha += B[i * 3 + p] * B[i * 3 + p];
        beta += B[i * 3 + q] * B[i * 3 + q];
        gamma += B[i * 3 + p] * B[i * 3 + q];
      }
      if (gamma == 0.0 || std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta)) continue;
      rotated = 1;
      const double zeta = (beta - alpha) / (2.0 * gamma);
      const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
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
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) nrm[j] = std::sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (nrm[ord[b]] > nrm[ord[a]]) std::swap(ord[a], ord[b]);
  double Vs[9];
  for (int j = 0; j < 3; ++j) {
    S[j] = nrm[ord[j]];
    for (int i = 0; i < 3; ++i) {
      Vs[i * 3 + j] = V[i * 3 + ord[j]];
      U[i * 3 + j] = (S[j] > 0.0) ? B[i * 3 + ord[j]] / S[j] : 0.0;
    }
  }
  std::memcpy(V, Vs, sizeof(Vs));
  const double tiny = 1e-14 * (S[0] > 0 ? S[0] : 1.0);
  if (S[0] <= 0.0) {
    for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  if (S[1] <= tiny) {
    const double u0[3] = {U[0], U[3], U[6]};
    const int m = std::fabs(u0[0]) < std::fabs(u0[1]) ? (std::fabs(u0[0]) < std::fabs(u0[2]) ? 0 : 2)
                                                      : (std::fabs(u0[1]) < std::fabs(u0[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    const double d = u0[m];
    const double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
    const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    U[1] = v[0] / n; U[4] = v[1] / n; U[7] = v[2] / n;
  }
  if (S[2] <= tiny) {
    const double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
    U[2] = a[1] * b[2] - a[2] * b[1];
    U[5] = a[2] * b[0] - a[0] * b[2];
    U[8] = a[0] * b[1] - a[1] * b[0];
  }
}

double det3(const double M[9])
{
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// Eq. (39)-(43) of Umeyama as pcl::umeyama writes them; prec = NumTraits<Scalar>::dummy_precision()
void umeyama_core(const double sigma[9], const double sm[3], const double dm[3], double prec, double R[9], double t[3])
{
  double U[9], S[3], V[9];
  svd3(sigma, U, S, V);
  double Sd[3] = {1.0, 1.0, 1.0};
  if (det3(sigma) < 0) Sd[2] = -1.0;
  int rank = 0;
  for (int i = 0; i < 3; ++i)
    if (!(std::fabs(S[i]) <= std::fabs(S[0]) * prec)) ++rank;
  if (rank == 2) Sd[2] = (det3(U) * det3(V) > 0) ? 1.0 : -1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += U[i * 3 + k] * Sd[k] * V[j * 3 + k];
      R[i * 3 + j] = acc;
    }
  for (int i = 0; i < 3; ++i)
    t[i] = dm[i] - (R[i * 3 + 0] * sm[0] + R[i * 3 + 1] * sm[1] + R[i * 3 + 2] * sm[2]);
}

}  // namespace

void umeyama_f32(const float *src, const float *dst, int n, float T[16])
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
  umeyama_core(sigma, smd, dmd, 1e-5, R, t);
  std::memset(T, 0, sizeof(float) * 16);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[c * 4 + r] = (float)R[r * 3 + c];
    T[12 + r] = (float)t[r];
  }
  T[15] = 1.0f;
}

void umeyama_f64(const double *src, const double *dst, int n, double T[16])
{
  double sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) { sm[a] += src[i * 3 + a]; dm[a] += dst[i * 3 + a]; }
  const double one_over_n = 1.0 / (double)n;
  for (int a = 0; a < 3; ++a) { sm[a] *= one_over_n; dm[a] *= one_over_n; }
  double sigma[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double s[3] = {src[i * 3] - sm[0], src[i * 3 + 1] - sm[1], src[i * 3 + 2] - sm[2]};
    const double d[3] = {dst[i * 3] - dm[0], dst[i * 3 + 1] - dm[1], dst[i * 3 + 2] - dm[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) sigma[r * 3 + c] += d[r] * s[c];
  }
  for (int i = 0; i < 9; ++i) sigma[i] *= one_over_n;
  double R[9], t[3];
  umeyama_core(sigma, sm, dm, 1e-12, R, t);
  std::memset(T, 0, sizeof(double) * 16);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[c * 4 + r] = R[r * 3 + c];
    T[12 + r] = t[r];
  }
  T[15] = 1.0;
}

void mat4_mul(const float A[16], const float B[16], float out[16])
{
  float r[16];
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 4; ++i) {
      float acc = 0.0f;
      for (int k = 0; k < 4; ++k) acc += A[k * 4 + i] * B[c * 4 + k];
      r[c * 4 + i] = acc;
    }
  std::memcpy(out, r, sizeof(r));
}

void mat4_inverse(const float A[16], float out[16])
{
  // cofactor expansion; a singular input yields inf/NaN like Eigen's inverse()
  const float *m = A;
  float inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const float invdet = 1.0f / det;
  for (int i = 0; i < 16; ++i) out[i] = inv[i] * invdet;
}

void eigen33_values(const float cov[9], float evals[3])
{
  float scale = 0.0f;
  for (int i = 0; i < 9; ++i) scale = std::fmax(scale, std::fabs(cov[i]));
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float m[9];
  for (int i = 0; i < 9; ++i) m[i] = cov[i] / scale;
  compute_roots(m, evals);
  for (int i = 0; i < 3; ++i) evals[i] *= scale;
}

}  // namespace mm3d
