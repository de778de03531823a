/*
 * o_linalg.c -- small dense helpers for the CPU oracle (TEST INFRASTRUCTURE).
 *
 * umeyama   -> pcl::registration::TransformationEstimationSVD (use_umeyama_ = true)
 *              -> pcl::umeyama == Eigen::umeyama(src, dst, with_scaling = false)
 *              (Eigen/src/Geometry/Umeyama.h; JacobiSVD<Matrix3>, FullU|FullV).
 *              Used at R/src/matching.cpp:135-137 (float) and inside ICP / SAC-IA (float) and
 *              SampleConsensusModelRegistration::estimateRigidTransformationSVD (double).
 * mat4 inverse -> Eigen::Matrix4f::inverse() at R/src/map_merging.cpp:143 (cofactor form).
 * RNGs     -> boost::mt19937 (seed 12345) behind pcl::SampleConsensusModel::rnd(), and glibc
 *              rand() behind SampleConsensusInitialAlignment::getRandomIndex.
 */
#include "mm3d_oracle.h"

#include <math.h>
#include <string.h>

/* ---- 3x3 SVD, one-sided Jacobi (Hestenes), double.  A row-major. ---------- */
static void svd3(const double A[9], double U[9], double S[3], double V[9])
{
  double B[9];
  memcpy(B, A, sizeof(B));
  double frob2 = 0.0;
  for (int i = 0; i < 9; ++i) { V[i] = (i % 4 == 0) ? 1.0 : 0.0; frob2 += A[i] * A[i]; }
  /* Eigen's JacobiSVD stops rotating a 2x2 block once its off-diagonals are <= 2 eps * (largest
   * diagonal entry); one-sided equivalent: |col_p . col_q| <= 2 eps |A| max(|col_p|, |col_q|) */
  const double thr = 2.0 * 2.220446049250313e-16 * sqrt(frob2);
  static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int k = 0; k < 3; ++k) {
      int p = P[k], q = Q[k];
      double alpha = 0, beta = 0, gamma = 0;
      for (int i = 0; i < 3; ++i) {
        alpha += B[i * 3 + p] * B[i * 3 + p];
        beta += B[i * 3 + q] * B[i * 3 + q];
        gamma += B[i * 3 + p] * B[i * 3 + q];
      }
      if (gamma == 0.0 || fabs(gamma) <= thr * sqrt(alpha > beta ? alpha : beta)) continue;
      rotated = 1;
      double zeta = (beta - alpha) / (2.0 * gamma);
      double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
      for (int i = 0; i < 3; ++i) {
        double bp = B[i * 3 + p], bq = B[i * 3 + q];
        B[i * 3 + p] = c * bp - s * bq;
        B[i * 3 + q] = s * bp + c * bq;
        double vp = V[i * 3 + p], vq = V[i * 3 + q];
        V[i * 3 + p] = c * vp - s * vq;
        V[i * 3 + q] = s * vp + c * vq;
      }
    }
    if (!rotated) break;
  }
  double nrm[3];
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j)
    nrm[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (nrm[ord[b]] > nrm[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
  double Vs[9];
  for (int j = 0; j < 3; ++j) {
    S[j] = nrm[ord[j]];
    for (int i = 0; i < 3; ++i) {
      Vs[i * 3 + j] = V[i * 3 + ord[j]];
      U[i * 3 + j] = (S[j] > 0.0) ? B[i * 3 + ord[j]] / S[j] : 0.0;
    }
  }
  memcpy(V, Vs, sizeof(Vs));
  /* complete U to an orthonormal basis where singular values vanish */
  const double tiny = 1e-14 * (S[0] > 0 ? S[0] : 1.0);
  if (S[0] <= 0.0) {
    for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  if (S[1] <= tiny) {
    /* pick any unit vector orthogonal to u0 */
    double u0[3] = {U[0], U[3], U[6]};
    int m = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
    double e[3] = {0, 0, 0}; e[m] = 1.0;
    double d = u0[m];
    double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
    double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    U[1] = v[0] / n; U[4] = v[1] / n; U[7] = v[2] / n;
  }
  if (S[2] <= tiny) {
    double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
    U[2] = a[1] * b[2] - a[2] * b[1];
    U[5] = a[2] * b[0] - a[0] * b[2];
    U[8] = a[0] * b[1] - a[1] * b[0];
  }
}

static double det3(const double M[9])
{
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* Eigen::umeyama core: sigma (row-major 3x3 = 1/n * dst_demean * src_demean^T), means.
 * prec = NumTraits<Scalar>::dummy_precision() (1e-5 float, 1e-12 double) for the rank test. */
static void umeyama_core(const double sigma[9], const double src_mean[3], const double dst_mean[3],
                         double prec, double R[9], double t[3])
{
  double U[9], S[3], V[9];
  svd3(sigma, U, S, V);
  double Sd[3] = {1.0, 1.0, 1.0};
  if (det3(sigma) < 0) Sd[2] = -1.0;
  int rank = 0;
  for (int i = 0; i < 3; ++i)
    if (!(fabs(S[i]) <= fabs(S[0]) * prec)) ++rank;
  if (rank == 2) {
    if (det3(U) * det3(V) > 0) { Sd[0] = Sd[1] = Sd[2] = 1.0; }
    else { Sd[0] = Sd[1] = 1.0; Sd[2] = -1.0; }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += U[i * 3 + k] * Sd[k] * V[j * 3 + k];
      R[i * 3 + j] = acc;
    }
  for (int i = 0; i < 3; ++i)
    t[i] = dst_mean[i] - (R[i * 3 + 0] * src_mean[0] + R[i * 3 + 1] * src_mean[1] + R[i * 3 + 2] * src_mean[2]);
}

void mo_umeyama_core_f32(const float sg[9], const float sm[3], const float dm[3], float one_over_n, float T[16]);

/* float instantiation: means, demeaning and sigma accumulate in float, sequentially */
void mo_umeyama_f32(const float *src, const float *dst, int n, float T[16])
{
  float sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) { sm[a] += src[i * 3 + a]; dm[a] += dst[i * 3 + a]; }
  const float one_over_n = 1.0f / (float)n;
  for (int a = 0; a < 3; ++a) { sm[a] *= one_over_n; dm[a] *= one_over_n; }
  float sg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    float s[3] = {src[i * 3] - sm[0], src[i * 3 + 1] - sm[1], src[i * 3 + 2] - sm[2]};
    float d[3] = {dst[i * 3] - dm[0], dst[i * 3 + 1] - dm[1], dst[i * 3 + 2] - dm[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) sg[r * 3 + c] += d[r] * s[c];
  }
  mo_umeyama_core_f32(sg, sm, dm, one_over_n, T);
}

/* the algebra behind the float sums (also used by o_audit.c, which forms the sums in other orders) */
void mo_umeyama_core_f32(const float sg[9], const float sm[3], const float dm[3], float one_over_n, float T[16])
{
  double sigma[9], smd[3], dmd[3], R[9], t[3];
  for (int i = 0; i < 9; ++i) sigma[i] = (double)(sg[i] * one_over_n);
  for (int a = 0; a < 3; ++a) { smd[a] = sm[a]; dmd[a] = dm[a]; }
  umeyama_core(sigma, smd, dmd, 1e-5, R, t);
  memset(T, 0, sizeof(float) * 16);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[c * 4 + r] = (float)R[r * 3 + c];
    T[12 + r] = (float)t[r];
  }
  T[15] = 1.0f;
}

void mo_umeyama_f64(const double *src, const double *dst, int n, double T[16])
{
  double sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) { sm[a] += src[i * 3 + a]; dm[a] += dst[i * 3 + a]; }
  const double one_over_n = 1.0 / (double)n;
  for (int a = 0; a < 3; ++a) { sm[a] *= one_over_n; dm[a] *= one_over_n; }
  double sigma[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    double s[3] = {src[i * 3] - sm[0], src[i * 3 + 1] - sm[1], src[i * 3 + 2] - sm[2]};
    double d[3] = {dst[i * 3] - dm[0], dst[i * 3 + 1] - dm[1], dst[i * 3 + 2] - dm[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) sigma[r * 3 + c] += d[r] * s[c];
  }
  for (int i = 0; i < 9; ++i) sigma[i] *= one_over_n;
  double R[9], t[3];
  umeyama_core(sigma, sm, dm, 1e-12, R, t);
  memset(T, 0, sizeof(double) * 16);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[c * 4 + r] = R[r * 3 + c];
    T[12 + r] = t[r];
  }
  T[15] = 1.0;
}

/* ---- 4x4 float, column-major ------------------------------------------------ */
void mo_mat4_mul(const float A[16], const float B[16], float out[16])
{
  float r[16];
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 4; ++i) {
      float acc = 0.0f;
      for (int k = 0; k < 4; ++k) acc += A[k * 4 + i] * B[c * 4 + k];
      r[c * 4 + i] = acc;
    }
  memcpy(out, r, sizeof(r));
}

void mo_mat4_inverse(const float A[16], float out[16])
{
  /* cofactor expansion; a singular input yields inf/NaN as Eigen's inverse() does */
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
  float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  float invdet = 1.0f / det;
  for (int i = 0; i < 16; ++i) out[i] = inv[i] * invdet;
}

/* ---- boost::mt19937 --------------------------------------------------------- */
static uint32_t mt_state[624];
static int mt_pos = 625;
void mo_mt19937_seed(uint32_t seed)
{
  mt_state[0] = seed;
  for (int i = 1; i < 624; ++i)
    mt_state[i] = 1812433253u * (mt_state[i - 1] ^ (mt_state[i - 1] >> 30)) + (uint32_t)i;
  mt_pos = 624;
}
uint32_t mo_mt19937_next(void)
{
  if (mt_pos >= 624) {
    if (mt_pos == 625) mo_mt19937_seed(5489u);
    for (int k = 0; k < 624; ++k) {
      uint32_t y = (mt_state[k] & 0x80000000u) | (mt_state[(k + 1) % 624] & 0x7fffffffu);
      mt_state[k] = mt_state[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    mt_pos = 0;
  }
  uint32_t y = mt_state[mt_pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* ---- glibc rand(): TYPE_3 additive feedback generator (r[i] = r[i-3] + r[i-31]) ------------- */
static int32_t gl_r[34];
static uint32_t gl_ring[31];
static int gl_f, gl_b, gl_init = 0;
void mo_srand(unsigned seed)
{
  if (seed == 0) seed = 1;
  gl_r[0] = (int32_t)seed;
  for (int i = 1; i < 31; ++i) {
    /* r[i] = (16807 * r[i-1]) % 2147483647, computed as glibc does (Schrage) */
    long hi = gl_r[i - 1] / 127773, lo = gl_r[i - 1] % 127773;
    long word = 16807 * lo - 2836 * hi;
    if (word < 0) word += 2147483647;
    gl_r[i] = (int32_t)word;
  }
  for (int i = 0; i < 31; ++i) gl_ring[i] = (uint32_t)gl_r[i];
  gl_f = 3; gl_b = 0;
  gl_init = 1;
  for (int i = 0; i < 310; ++i) (void)mo_rand();
}
int mo_rand(void)
{
  if (!gl_init) mo_srand(1);
  gl_ring[gl_f] += gl_ring[gl_b];
  uint32_t result = gl_ring[gl_f] >> 1;
  gl_f = (gl_f + 1) % 31;
  gl_b = (gl_b + 1) % 31;
  return (int)result;
}
