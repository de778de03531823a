/*
 * o_shot.c -- computeLocalDescriptors(SHOT) restated (TEST INFRASTRUCTURE).
 *
 * R/src/dispatch_descriptors.h:46 binds Descriptor::SHOT to
 *   pcl::SHOTColorEstimation<PointXYZRGB, Normal, SHOT1344>  (the 352-bin shape-only row at :45 is
 *   commented out), configured by R/src/features.cpp:105-109: setRadiusSearch(feature_radius),
 *   setSearchSurface(points), setInputNormals(normals), setInputCloud(keypoints).
 * PCL 1.8.1:
 *   features/impl/shot_lrf.hpp  SHOTLocalReferenceFrameEstimation::getLocalRF  (default frames,
 *       same radius, same surface: SHOTEstimationBase::initCompute)
 *   features/impl/shot.hpp      SHOTColorEstimation::computeFeature / computePointSHOT /
 *       interpolateDoubleChannel / RGB2CIELAB, SHOTEstimationBase::createBinDistanceShape /
 *       normalizeHistogram
 * Layout of a row: 32 volumes x 11 shape slots (352) then 32 volumes x 31 colour slots (992);
 * volume index bits: [azimuth octant 3][outer shell 1][z > 0 1]; rf = x, y, z axes (9 floats).
 *
 * Restatement choices (unknowable from the call sites, documented so the device can agree):
 *   - neighbour order: Feature::initCompute builds its KdTree with sorted = false, so PCL walks
 *     the neighbours in FLANN's traversal order; here they come sorted by (distance, index) like
 *     every other search of this oracle.  The float accumulation of the bins follows that order.
 *   - pcl_macros.h includes <math.h>, so sqrt()/fabs() of float arguments are the float overloads
 *     (libstdc++ >= 6); acos/atan2/floor act on doubles.
 *   - 3- and 4-element dot products are summed left to right (Eigen's SIMD reduction order depends
 *     on the build flags).
 *   - Eigen::SelfAdjointEigenSolver<Matrix3d> is replaced by a cyclic Jacobi iteration in double;
 *     the signs of the axes are fixed afterwards by the disambiguation step as in PCL.
 *   - RGB2CIELAB indexes sXYZ_LUT[int(v * 4000)] with v that can reach 1.0 (one past the end of
 *     the table: UB in PCL); the index is clamped to 3999 here.
 *   - the keypoint's own colour is what the keypoint cloud carries: R/src/features.cpp:57-60
 *     copies only x, y, z out of the SIFT result, so it is (0, 0, 0) in the reference pipeline.
 */
#include "mm3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SHOT_SHAPE_BINS 10
#define SHOT_COLOR_BINS 30
#define SHOT_SECTORS 32
#define SHOT_DIM 1344
#define SHOT_COLOR_OFFSET (SHOT_SECTORS * (SHOT_SHAPE_BINS + 1))

#define PST_RAD_45 0.78539816339744830961566084581988
#define PST_RAD_90 1.5707963267948966192313216916398
#define PST_RAD_135 2.3561944901923449288469825374596
#define PST_RAD_PI_7_8 2.7488935718910690836548129603691

static float s_rgb_lut[256];
static float s_xyz_lut[4000];
static int s_lut_ready = 0;

static void lab_luts(void)
{
  if (s_lut_ready) return;
  for (int i = 0; i < 256; ++i) {
    float f = (float)i / 255.0f;
    if (f > 0.04045) s_rgb_lut[i] = powf((f + 0.055f) / 1.055f, 2.4f);
    else s_rgb_lut[i] = f / 12.92f;
  }
  for (int i = 0; i < 4000; ++i) {
    float f = (float)i / 4000.0f;
    if (f > 0.008856) s_xyz_lut[i] = powf(f, 0.3333f);
    else s_xyz_lut[i] = (float)((7.787 * f) + (16.0 / 116.0));
  }
  s_lut_ready = 1;
}

/* SHOTColorEstimation::RGB2CIELAB followed by the /100, /120, /120 of computePointSHOT */
void mo_shot_rgb2lab(unsigned char R, unsigned char G, unsigned char B, float lab[3])
{
  lab_luts();
  float fr = s_rgb_lut[R], fg = s_rgb_lut[G], fb = s_rgb_lut[B];
  const float x = fr * 0.412453f + fg * 0.357580f + fb * 0.180423f;
  const float y = fr * 0.212671f + fg * 0.715160f + fb * 0.072169f;
  const float z = fr * 0.019334f + fg * 0.119193f + fb * 0.950227f;
  float vx = x / 0.95047f, vy = y, vz = z / 1.08883f;
  int ix = (int)(vx * 4000), iy = (int)(vy * 4000), iz = (int)(vz * 4000);
  if (ix > 3999) ix = 3999;
  if (iy > 3999) iy = 3999;
  if (iz > 3999) iz = 3999;
  vx = s_xyz_lut[ix]; vy = s_xyz_lut[iy]; vz = s_xyz_lut[iz];
  float L = 116.0f * vy - 16.0f;
  if (L > 100) L = 100.0f;
  float A = 500.0f * (vx - vy);
  if (A > 120) A = 120.0f; else if (A < -120) A = -120.0f;
  float B2 = 200.0f * (vy - vz);
  if (B2 > 120) B2 = 120.0f; else if (B2 < -120) B2 = -120.0f;
  lab[0] = L / 100.0f; lab[1] = A / 120.0f; lab[2] = B2 / 120.0f;
}

/* Eigen-decomposition of a symmetric 3x3 (double): cyclic Jacobi; eigenvalues ascending in w,
 * eigenvectors in the COLUMNS of V.  The device carries the same text (csrc/shot.hip). */
static void sym_eig3(double a[3][3], double w[3], double V[3][3])
{
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double dg = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (!(off > 4.93e-32 * (dg + 2.0 * off))) break;   /* off-diagonal norm <= eps * |A|_F, or NaN */
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[p][q];
        if (apq == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        const int r = 3 - p - q;
        const double app = a[p][p], aqq = a[q][q], apr = a[p][r], aqr = a[q][r];
        a[p][p] = app - t * apq;
        a[q][q] = aqq + t * apq;
        a[p][q] = a[q][p] = 0.0;
        a[p][r] = a[r][p] = c * apr - s * aqr;
        a[q][r] = a[r][q] = s * apr + c * aqr;
        for (int k = 0; k < 3; ++k) {
          const double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq;
          V[k][q] = s * vp + c * vq;
        }
      }
  }
  w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
  /* ascending, stable */
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (w[j + 1] < w[j]) {
        double t = w[j]; w[j] = w[j + 1]; w[j + 1] = t;
        for (int k = 0; k < 3; ++k) { t = V[k][j]; V[k][j] = V[k][j + 1]; V[k][j + 1] = t; }
      }
}

/* shot_lrf.hpp getLocalRF.  Returns 1 and rf (rows x, y, z) or 0 and NaN. */
static int shot_lrf(const mo_point *surface, const mo_point *c, const int *idx, const float *d2, int cnt,
                    double radius, double *vij /* cnt x 3 scratch */, float rf[9])
{
  double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double sum = 0.0;
  int valid = 0;
  for (int i = 0; i < cnt; ++i) {
    const mo_point *pt = &surface[idx[i]];
    if (pt->x == c->x && pt->y == c->y && pt->z == c->z) continue;
    double *v = &vij[(size_t)valid * 3];
    v[0] = (double)(pt->x - c->x); v[1] = (double)(pt->y - c->y); v[2] = (double)(pt->z - c->z);
    const double distance = radius - (double)sqrtf(d2[i]);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) cov[a][b] += distance * (v[a] * v[b]);
    sum += distance;
    ++valid;
  }
  if (valid < 5) { for (int i = 0; i < 9; ++i) rf[i] = NAN; return 0; }
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a][b] /= sum;
  double w[3], V[3][3];
  sym_eig3(cov, w, V);
  if (!isfinite(w[0]) || !isfinite(w[1]) || !isfinite(w[2])) { for (int i = 0; i < 9; ++i) rf[i] = NAN; return 0; }
  double v1[3] = {V[0][2], V[1][2], V[2][2]};   /* largest eigenvalue: x axis */
  double v3[3] = {V[0][0], V[1][0], V[2][0]};   /* smallest: z axis */
  int plus_x = 0, plus_z = 0;
  for (int i = 0; i < valid; ++i) {
    const double *v = &vij[(size_t)i * 3];
    if (v[0] * v1[0] + v[1] * v1[1] + v[2] * v1[2] >= 0) ++plus_x;
    if (v[0] * v3[0] + v[1] * v3[1] + v[2] * v3[2] >= 0) ++plus_z;
  }
  double *axes[2] = {v1, v3};
  int plus[2] = {plus_x, plus_z};
  for (int a = 0; a < 2; ++a) {
    int p = 2 * plus[a] - valid;
    double *ax = axes[a];
    if (p == 0) {
      const int points = 5, median = valid / 2;
      for (int i = -points / 2; i <= points / 2; ++i) {
        const double *v = &vij[(size_t)(median - i) * 3];
        if (v[0] * ax[0] + v[1] * ax[1] + v[2] * ax[2] > 0) ++p;
      }
      if (p < points / 2 + 1) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; }
    } else if (p < 0) {
      ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2];
    }
  }
  for (int k = 0; k < 3; ++k) { rf[k] = (float)v1[k]; rf[6 + k] = (float)v3[k]; }
  /* y = z x x in float (rf.row(2).cross(rf.row(0))) */
  rf[3] = rf[7] * rf[2] - rf[8] * rf[1];
  rf[4] = rf[8] * rf[0] - rf[6] * rf[2];
  rf[5] = rf[6] * rf[1] - rf[7] * rf[0];
  return 1;
}

/* computePointSHOT (shape + colour) for one keypoint with a valid frame and >= 5 neighbours */
static void point_shot(const mo_point *surface, const mo_normal *normals, const mo_point *c, const float rf[9],
                       const int *idx, const float *d2, int cnt, double radius, float *shot)
{
  const double radius3_4 = (radius * 3) / 4, radius1_4 = radius / 4, radius1_2 = radius / 2;
  for (int b = 0; b < SHOT_DIM; ++b) shot[b] = 0.0f;
  float lab_ref[3];
  mo_shot_rgb2lab((c->rgba >> 16) & 0xff, (c->rgba >> 8) & 0xff, c->rgba & 0xff, lab_ref);
  for (int i = 0; i < cnt; ++i) {
    const mo_point *pt = &surface[idx[i]];
    const mo_normal *nv = &normals[idx[i]];
    /* createBinDistanceShape */
    if (!isfinite(nv->nx) || !isfinite(nv->ny) || !isfinite(nv->nz)) continue;
    double cosine = (double)(nv->nx * rf[6] + nv->ny * rf[7] + nv->nz * rf[8]);
    if (cosine > 1.0) cosine = 1.0;
    if (cosine < -1.0) cosine = -1.0;
    double bin_shape = ((1.0 + cosine) * SHOT_SHAPE_BINS) / 2;
    /* colour bin distance */
    float lab[3];
    mo_shot_rgb2lab((pt->rgba >> 16) & 0xff, (pt->rgba >> 8) & 0xff, pt->rgba & 0xff, lab);
    double color_distance =
        (double)((fabsf(lab_ref[0] - lab[0]) + ((fabsf(lab_ref[1] - lab[1]) + fabsf(lab_ref[2] - lab[2])) / 2)) / 3);
    if (color_distance > 1.0) color_distance = 1.0;
    if (color_distance < 0.0) color_distance = 0.0;
    double bin_color = color_distance * SHOT_COLOR_BINS;

    /* interpolateDoubleChannel */
    const float dx = pt->x - c->x, dy = pt->y - c->y, dz = pt->z - c->z;
    const double distance = (double)sqrtf(d2[i]);
    if (fabs(distance - 0.0) < 1e-15) continue;
    double x_ref = (double)(dx * rf[0] + dy * rf[1] + dz * rf[2]);
    double y_ref = (double)(dx * rf[3] + dy * rf[4] + dz * rf[5]);
    double z_ref = (double)(dx * rf[6] + dy * rf[7] + dz * rf[8]);
    if (fabs(y_ref) < 1e-30) y_ref = 0;
    if (fabs(x_ref) < 1e-30) x_ref = 0;
    if (fabs(z_ref) < 1e-30) z_ref = 0;
    const int bit4 = ((y_ref > 0) || ((y_ref == 0.0) && (x_ref < 0))) ? 1 : 0;
    const int bit3 = ((x_ref > 0) || ((x_ref == 0.0) && (y_ref > 0))) ? !bit4 : bit4;
    int desc_index = (bit4 << 3) + (bit3 << 2);
    desc_index = desc_index << 1;
    if ((x_ref * y_ref > 0) || (x_ref == 0.0)) desc_index += (fabs(x_ref) >= fabs(y_ref)) ? 0 : 4;
    else desc_index += (fabs(x_ref) > fabs(y_ref)) ? 4 : 0;
    desc_index += z_ref > 0 ? 1 : 0;
    desc_index += (distance > radius1_2) ? 2 : 0;

    const int step_shape = (int)floor(bin_shape + 0.5);
    const int step_color = (int)floor(bin_color + 0.5);
    const int vol_shape = desc_index * (SHOT_SHAPE_BINS + 1);
    const int vol_color = SHOT_COLOR_OFFSET + desc_index * (SHOT_COLOR_BINS + 1);
    bin_shape -= step_shape;
    bin_color -= step_color;
    double w_shape = 1 - fabs(bin_shape), w_color = 1 - fabs(bin_color);
    if (bin_shape > 0) shot[vol_shape + ((step_shape + 1) % SHOT_SHAPE_BINS)] += (float)bin_shape;
    else shot[vol_shape + ((step_shape - 1 + SHOT_SHAPE_BINS) % SHOT_SHAPE_BINS)] -= (float)bin_shape;
    if (bin_color > 0) shot[vol_color + ((step_color + 1) % SHOT_COLOR_BINS)] += (float)bin_color;
    else shot[vol_color + ((step_color - 1 + SHOT_COLOR_BINS) % SHOT_COLOR_BINS)] -= (float)bin_color;

    /* radial shells */
    if (distance > radius1_2) {
      const double rd = (distance - radius3_4) / radius1_2;
      if (distance > radius3_4) { w_shape += 1 - rd; w_color += 1 - rd; }
      else {
        w_shape += 1 + rd; w_color += 1 + rd;
        shot[(desc_index - 2) * (SHOT_SHAPE_BINS + 1) + step_shape] -= (float)rd;
        shot[SHOT_COLOR_OFFSET + (desc_index - 2) * (SHOT_COLOR_BINS + 1) + step_color] -= (float)rd;
      }
    } else {
      const double rd = (distance - radius1_4) / radius1_2;
      if (distance < radius1_4) { w_shape += 1 + rd; w_color += 1 + rd; }
      else {
        w_shape += 1 - rd; w_color += 1 - rd;
        shot[(desc_index + 2) * (SHOT_SHAPE_BINS + 1) + step_shape] += (float)rd;
        shot[SHOT_COLOR_OFFSET + (desc_index + 2) * (SHOT_COLOR_BINS + 1) + step_color] += (float)rd;
      }
    }

    /* inclination */
    double inc_cos = z_ref / distance;
    if (inc_cos < -1.0) inc_cos = -1.0;
    if (inc_cos > 1.0) inc_cos = 1.0;
    const double inclination = acos(inc_cos);
    if (inclination > PST_RAD_90 || (fabs(inclination - PST_RAD_90) < 1e-30 && z_ref <= 0)) {
      const double id = (inclination - PST_RAD_135) / PST_RAD_90;
      if (inclination > PST_RAD_135) { w_shape += 1 - id; w_color += 1 - id; }
      else {
        w_shape += 1 + id; w_color += 1 + id;
        shot[(desc_index + 1) * (SHOT_SHAPE_BINS + 1) + step_shape] -= (float)id;
        shot[SHOT_COLOR_OFFSET + (desc_index + 1) * (SHOT_COLOR_BINS + 1) + step_color] -= (float)id;
      }
    } else {
      const double id = (inclination - PST_RAD_45) / PST_RAD_90;
      if (inclination < PST_RAD_45) { w_shape += 1 + id; w_color += 1 + id; }
      else {
        w_shape += 1 - id; w_color += 1 - id;
        shot[(desc_index - 1) * (SHOT_SHAPE_BINS + 1) + step_shape] += (float)id;
        shot[SHOT_COLOR_OFFSET + (desc_index - 1) * (SHOT_COLOR_BINS + 1) + step_color] += (float)id;
      }
    }

    /* azimuth */
    if (y_ref != 0.0 || x_ref != 0.0) {
      const double azimuth = atan2(y_ref, x_ref);
      const int sel = desc_index >> 2;
      double ad = (azimuth - (-PST_RAD_PI_7_8 + PST_RAD_45 * sel)) / PST_RAD_45;
      ad = fmax(-0.5, fmin(ad, 0.5));
      if (ad > 0) {
        w_shape += 1 - ad; w_color += 1 - ad;
        const int ii = (desc_index + 4) % SHOT_SECTORS;
        shot[ii * (SHOT_SHAPE_BINS + 1) + step_shape] += (float)ad;
        shot[SHOT_COLOR_OFFSET + ii * (SHOT_COLOR_BINS + 1) + step_color] += (float)ad;
      } else {
        const int ii = (desc_index - 4 + SHOT_SECTORS) % SHOT_SECTORS;
        w_shape += 1 + ad; w_color += 1 + ad;
        shot[ii * (SHOT_SHAPE_BINS + 1) + step_shape] -= (float)ad;
        shot[SHOT_COLOR_OFFSET + ii * (SHOT_COLOR_BINS + 1) + step_color] -= (float)ad;
      }
    }
    shot[vol_shape + step_shape] += (float)w_shape;
    shot[vol_color + step_color] += (float)w_color;
  }
  /* normalizeHistogram */
  double acc = 0;
  for (int j = 0; j < SHOT_DIM; ++j) acc += shot[j] * shot[j];
  acc = sqrt(acc);
  for (int j = 0; j < SHOT_DIM; ++j) shot[j] /= (float)acc;
}

int mo_shot_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints, int n_kp,
                double radius, float *desc /* n_kp x 1344 */, float *rf_out /* n_kp x 9 or NULL */)
{
  mo_grid *g = mo_grid_build(surface, n, (float)(radius * 0.5));
  /* Feature::searchForNeighbors -> radiusSearch(double radius): FLANN compares against
   * static_cast<float>(radius * radius) */
  const float r2 = (float)(radius * radius);
  int cap = 4096;
  int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)cap);
  double *vij = (double *)malloc(sizeof(double) * 3 * (size_t)cap);
  for (int k = 0; k < n_kp; ++k) {
    float *out = &desc[(size_t)k * SHOT_DIM];
    float rf[9];
    const mo_point *c = &keypoints[k];
    int cnt = 0;
    const int finite = isfinite(c->x) && isfinite(c->y) && isfinite(c->z);
    if (finite) {
      cnt = mo_radius_search(g, c->x, c->y, c->z, r2, idx, d2, cap);
      if (cnt > cap) {
        cap = cnt * 2;
        idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
        d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);
        vij = (double *)realloc(vij, sizeof(double) * 3 * (size_t)cap);
        cnt = mo_radius_search(g, c->x, c->y, c->z, r2, idx, d2, cap);
      }
    }
    const int have_rf = finite && shot_lrf(surface, c, idx, d2, cnt, radius, vij, rf);
    if (!have_rf || cnt == 0) {
      /* computeFeature: NaN descriptor AND NaN rf */
      for (int b = 0; b < SHOT_DIM; ++b) out[b] = NAN;
      if (rf_out) for (int b = 0; b < 9; ++b) rf_out[(size_t)k * 9 + b] = NAN;
      continue;
    }
    if (rf_out) memcpy(&rf_out[(size_t)k * 9], rf, sizeof(rf));
    if (cnt < 5) {   /* computePointSHOT: too few neighbours -> NaN descriptor, rf kept */
      for (int b = 0; b < SHOT_DIM; ++b) out[b] = NAN;
      continue;
    }
    point_shot(surface, normals, c, rf, idx, d2, cnt, radius, out);
  }
  free(idx); free(d2); free(vij);
  mo_grid_free(g);
  return n_kp;
}

int mo_descriptors_shot(const mo_point *surface, const mo_normal *normals, int n, mo_point *keypoints, int n_kp,
                        double radius, float *desc)
{
  if (n_kp <= 0) return 0;
  mo_shot_raw(surface, normals, n, keypoints, n_kp, radius, desc, NULL);
  /* DefaultPointRepresentation<SHOT1344>::isValid: the 1344 descriptor floats finite (rf is not
   * part of the representation); prune descriptors and keypoints (features.cpp:118-143) */
  int m = 0;
  for (int k = 0; k < n_kp; ++k) {
    int valid = 1;
    for (int b = 0; b < SHOT_DIM; ++b) if (!isfinite(desc[(size_t)k * SHOT_DIM + b])) { valid = 0; break; }
    if (!valid) continue;
    if (m != k) {
      memmove(&desc[(size_t)m * SHOT_DIM], &desc[(size_t)k * SHOT_DIM], sizeof(float) * SHOT_DIM);
      keypoints[m] = keypoints[k];
    }
    ++m;
  }
  return m;
}
