// linalg.cpp -- small dense host helpers of the registration host code.
//
// Umeyama (pcl::umeyama == Eigen::umeyama without scaling) behind
// pcl::registration::TransformationEstimationSVD (R/src/matching.cpp:135-137, and inside SAC-IA),
// in float, and behind SampleConsensusModelRegistration::estimateRigidTransformationSVD in double;
// Eigen::Matrix4f::inverse() / operator* used by computeGlobalTransforms (R/src/map_merging.cpp:137-186).
#include <cmath>
#include <cstring>

#include "device_util.hpp"
#include "linalg_shared.hpp"

namespace mm3d {

void umeyama_f32(const float *src, const float *dst, int n, float T[16]) { umeyama_f32_shared(src, dst, n, T); }

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
  umeyama_core_shared(sigma, sm, dm, 1e-12, R, t);
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
