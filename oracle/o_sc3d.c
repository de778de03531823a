/*
 * o_sc3d.c -- computeLocalDescriptors(SC3D) restated (TEST INFRASTRUCTURE).
 *
 * R/src/dispatch_descriptors.h:47 binds Descriptor::SC3D to
 *   pcl::ShapeContext3DEstimation<PointXYZRGB, Normal, ShapeContext1980> (field "shape_context"),
 *   configured by R/src/features.cpp:105-109 (setRadiusSearch(feature_radius), surface, normals, keypoints).
 * PCL 1.8.1 features/impl/3dsc.hpp (initCompute, computePoint, computeFeature), features/3dsc.h
 *   (defaults: azimuth 12, elevation 11, radius 15 bins, min_radius 0.1, point_density_radius 0.2;
 *   rnd() = boost::uniform_01<boost::mt19937> seeded with 12345 -> double u32 * 2^-32),
 *   common/geometry.h project(), common/utils.h equal() (eps = numeric_limits<float>::min()).
 * Row layout: bin (l azimuth, k elevation, j radius) at l * 11 * 15 + k * 15 + j; rf is zeroed by PCL.
 *
 * Restatement choices: neighbours sorted by (distance, index) (PCL's tree is unsorted: the order of
 * the float "+= w" per bin follows FLANN's traversal there; the nearest neighbour -- PCL takes the
 * first minimum of the distances -- is the same point either way up to exact distance ties); vector
 * sums left to right; Eigen's normalize() = divide by sqrt(squared norm); pcl::deg2rad / rad2deg are the
 * float overloads of common/impl/angles.hpp (alpha * 0.017453293f, alpha * 57.29578f).
 */
#include "mm3d_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SC_AZ 12
#define SC_EL 11
#define SC_RAD 15
#define SC_DIM (SC_AZ * SC_EL * SC_RAD)

static inline float deg2rad_f(float a) { return a * 0.017453293f; }   /* pcl::deg2rad(float): alpha * 0.017453293f */
static inline float rad2deg_f(float a) { return a * 57.29578f; }      /* pcl::rad2deg(float): alpha * 57.29578f */

/* the tables of ShapeContext3DEstimation::initCompute */
void mo_sc3d_tables(double search_radius, float radii[SC_RAD + 1], float theta_div[SC_EL + 1], float phi_div[SC_AZ + 1],
                    float volume_lut[SC_DIM])
{
  const double min_radius = 0.1;
  const float azimuth_interval = 360.0f / (float)SC_AZ;
  const float elevation_interval = 180.0f / (float)SC_EL;
  for (int j = 0; j < SC_RAD + 1; ++j)
    radii[j] = (float)exp(log(min_radius) + (((float)j / (float)SC_RAD) * log(search_radius / min_radius)));
  for (int k = 0; k < SC_EL + 1; ++k) theta_div[k] = (float)k * elevation_interval;
  for (int l = 0; l < SC_AZ + 1; ++l) phi_div[l] = (float)l * azimuth_interval;
  const float integr_phi = deg2rad_f(phi_div[1]) - deg2rad_f(phi_div[0]);
  const float e = 1.0f / 3.0f;
  for (int j = 0; j < SC_RAD; ++j) {
    const float integr_r = (radii[j + 1] * radii[j + 1] * radii[j + 1] / 3.0f) - (radii[j] * radii[j] * radii[j] / 3.0f);
    for (int k = 0; k < SC_EL; ++k) {
      const float integr_theta = cosf(deg2rad_f(theta_div[k])) - cosf(deg2rad_f(theta_div[k + 1]));
      const float V = integr_phi * integr_theta * integr_r;
      for (int l = 0; l < SC_AZ; ++l) volume_lut[(l * SC_EL * SC_RAD) + k * SC_RAD + j] = 1.0f / powf(V, e);
    }
  }
}

int mo_sc3d_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints, int n_kp,
                double radius, float *desc /* n_kp x 1980 */)
{
  float radii[SC_RAD + 1], theta_div[SC_EL + 1], phi_div[SC_AZ + 1];
  float *lut = (float *)malloc(sizeof(float) * SC_DIM);
  mo_sc3d_tables(radius, radii, theta_div, phi_div, lut);
  const double density_radius = 0.2;
  mo_grid *g = mo_grid_build(surface, n, (float)(radius * 0.5));
  mo_grid *gd = mo_grid_build(surface, n, (float)(density_radius * 0.5));
  const float r2 = (float)(radius * radius);
  const float rd2 = (float)(density_radius * density_radius);
  int cap = 4096;
  int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)cap);
  int dcap = 4096;
  int *didx = (int *)malloc(sizeof(int) * (size_t)dcap);
  float *dd2 = (float *)malloc(sizeof(float) * (size_t)dcap);
  /* the local point density of every surface point is needed over and over: count once, on demand */
  int *density = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) density[i] = -1;
  mo_mt19937_seed(12345u);
  for (int kq = 0; kq < n_kp; ++kq) {
    float *out = &desc[(size_t)kq * SC_DIM];
    const mo_point *c = &keypoints[kq];
    if (!(isfinite(c->x) && isfinite(c->y) && isfinite(c->z))) {
      for (int b = 0; b < SC_DIM; ++b) out[b] = NAN;
      continue;
    }
    int cnt = mo_radius_search(g, c->x, c->y, c->z, r2, idx, d2, cap);
    if (cnt > cap) {
      cap = cnt * 2;
      idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
      d2 = (float *)realloc(d2, sizeof(float) * (size_t)cap);
      cnt = mo_radius_search(g, c->x, c->y, c->z, r2, idx, d2, cap);
    }
    if (cnt == 0) {
      for (int b = 0; b < SC_DIM; ++b) out[b] = NAN;
      continue;
    }
    for (int b = 0; b < SC_DIM; ++b) out[b] = 0.0f;
    /* first minimum of the distances = element 0 of the sorted list */
    const mo_normal *nm = &normals[idx[0]];
    const float normal[3] = {nm->nx, nm->ny, nm->nz};
    float x_axis[3];
    x_axis[0] = (float)((double)mo_mt19937_next() * (1.0 / 4294967296.0));
    x_axis[1] = (float)((double)mo_mt19937_next() * (1.0 / 4294967296.0));
    x_axis[2] = (float)((double)mo_mt19937_next() * (1.0 / 4294967296.0));
    if (!(fabsf(normal[2] - 0.0f) < FLT_MIN)) x_axis[2] = -(normal[0] * x_axis[0] + normal[1] * x_axis[1]) / normal[2];
    else if (!(fabsf(normal[1] - 0.0f) < FLT_MIN)) x_axis[1] = -(normal[0] * x_axis[0] + normal[2] * x_axis[2]) / normal[1];
    else if (!(fabsf(normal[0] - 0.0f) < FLT_MIN)) x_axis[0] = -(normal[1] * x_axis[1] + normal[2] * x_axis[2]) / normal[0];
    {
      const float nn = sqrtf(x_axis[0] * x_axis[0] + x_axis[1] * x_axis[1] + x_axis[2] * x_axis[2]);
      x_axis[0] /= nn; x_axis[1] /= nn; x_axis[2] /= nn;
    }
    for (int ne = 0; ne < cnt; ++ne) {
      if (fabsf(d2[ne] - 0.0f) < FLT_MIN) continue;
      const mo_point *p = &surface[idx[ne]];
      const float r = sqrtf(d2[ne]);
      /* geometry::project(neighbour, origin, normal, proj); proj -= origin; normalize */
      const float po[3] = {p->x - c->x, p->y - c->y, p->z - c->z};
      const float lambda = normal[0] * po[0] + normal[1] * po[1] + normal[2] * po[2];
      float proj[3] = {p->x - lambda * normal[0], p->y - lambda * normal[1], p->z - lambda * normal[2]};
      proj[0] -= c->x; proj[1] -= c->y; proj[2] -= c->z;
      {
        const float nn = sqrtf(proj[0] * proj[0] + proj[1] * proj[1] + proj[2] * proj[2]);
        proj[0] /= nn; proj[1] /= nn; proj[2] /= nn;
      }
      const float cross[3] = {x_axis[1] * proj[2] - x_axis[2] * proj[1], x_axis[2] * proj[0] - x_axis[0] * proj[2],
                              x_axis[0] * proj[1] - x_axis[1] * proj[0]};
      const float cross_norm = sqrtf(cross[0] * cross[0] + cross[1] * cross[1] + cross[2] * cross[2]);
      float phi = rad2deg_f(atan2f(cross_norm, x_axis[0] * proj[0] + x_axis[1] * proj[1] + x_axis[2] * proj[2]));
      phi = (cross[0] * normal[0] + cross[1] * normal[1] + cross[2] * normal[2]) < 0.f ? (360.0f - phi) : phi;
      float no[3] = {po[0], po[1], po[2]};
      {
        const float nn = sqrtf(no[0] * no[0] + no[1] * no[1] + no[2] * no[2]);
        no[0] /= nn; no[1] /= nn; no[2] /= nn;
      }
      float theta = normal[0] * no[0] + normal[1] * no[1] + normal[2] * no[2];
      theta = rad2deg_f(acosf(fminf(1.0f, fmaxf(-1.0f, theta))));
      int j = 0, k = 0, l = 0;
      for (int rad = 1; rad < SC_RAD + 1; ++rad) if (r <= radii[rad]) { j = rad - 1; break; }
      for (int ang = 1; ang < SC_EL + 1; ++ang) if (theta <= theta_div[ang]) { k = ang - 1; break; }
      for (int ang = 1; ang < SC_AZ + 1; ++ang) if (phi <= phi_div[ang]) { l = ang - 1; break; }
      if (density[idx[ne]] < 0) {
        int dc = mo_radius_search(gd, p->x, p->y, p->z, rd2, didx, dd2, dcap);
        if (dc > dcap) {      /* only the count matters */
          dcap = dc * 2;
          didx = (int *)realloc(didx, sizeof(int) * (size_t)dcap);
          dd2 = (float *)realloc(dd2, sizeof(float) * (size_t)dcap);
        }
        density[idx[ne]] = dc;
      }
      const int point_density = density[idx[ne]];
      if (point_density == 0) continue;
      const float w = (1.0f / (float)point_density) * lut[(l * SC_EL * SC_RAD) + (k * SC_RAD) + j];
      out[(l * SC_EL * SC_RAD) + (k * SC_RAD) + j] += w;
    }
  }
  free(idx); free(d2); free(didx); free(dd2); free(density); free(lut);
  mo_grid_free(g); mo_grid_free(gd);
  return n_kp;
}

int mo_descriptors_sc3d(const mo_point *surface, const mo_normal *normals, int n, mo_point *keypoints, int n_kp,
                        double radius, float *desc)
{
  if (n_kp <= 0) return 0;
  mo_sc3d_raw(surface, normals, n, keypoints, n_kp, radius, desc);
  int m = 0;
  for (int k = 0; k < n_kp; ++k) {
    int valid = 1;
    for (int b = 0; b < SC_DIM; ++b) if (!isfinite(desc[(size_t)k * SC_DIM + b])) { valid = 0; break; }
    if (!valid) continue;
    if (m != k) {
      memmove(&desc[(size_t)m * SC_DIM], &desc[(size_t)k * SC_DIM], sizeof(float) * SC_DIM);
      keypoints[m] = keypoints[k];
    }
    ++m;
  }
  return m;
}
