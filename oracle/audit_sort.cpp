// audit_sort.cpp -- audit row 2 (TEST / EVIDENCE INFRASTRUCTURE): what libstdc++'s std::sort does to the order inside a voxel.
//
// pcl::VoxelGrid::applyFilter (PCL 1.8.1 filters/impl/voxel_grid.hpp; R/src/features.cpp:19-24) sorts a vector of
// (voxel index, point index) records with std::sort and an operator< that looks at the voxel index ONLY, then adds every
// voxel's points in the sorted order, in float.  std::sort is an introsort: records with equal keys come out in an order
// that depends on the whole array.  The oracle (o_filters.c) and the device take the points of a voxel in INPUT order.
// This file performs the sort the way PCL does, with the libstdc++ of this image, and counts the voxels whose centroid
// (float x, y, z or the packed rgba) differs between the two orders.  (PCL 1.8.1 was built against GCC 7's libstdc++ on
// Bionic; std::sort's algorithm -- median-of-three introsort, insertion sort below 16 -- has not changed since GCC 4.x.)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

struct Pt { float x, y, z; uint32_t rgba; };
struct Rec {
  unsigned int idx, cloud_point_index;
  bool operator<(const Rec &p) const { return idx < p.idx; }
};

static void centroid(const Pt *in, const Rec *b, const Rec *e, float out[3], uint32_t *rgba)
{
  float sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0, sa = 0;
  for (const Rec *j = b; j < e; ++j) {
    const Pt &p = in[j->cloud_point_index];
    sx += p.x; sy += p.y; sz += p.z;
    sr += (float)((p.rgba >> 16) & 255u); sg += (float)((p.rgba >> 8) & 255u); sb += (float)(p.rgba & 255u); sa += (float)((p.rgba >> 24) & 255u);
  }
  const float cnt = (float)(e - b);
  out[0] = sx / cnt; out[1] = sy / cnt; out[2] = sz / cnt;
  *rgba = ((uint32_t)(sa / cnt) << 24) | ((uint32_t)(sr / cnt) << 16) | ((uint32_t)(sg / cnt) << 8) | (uint32_t)(sb / cnt);
}

// out[0] voxels, [1] voxels with >= 3 points (a float sum of three or more depends on the order), [2] voxels with >= 17,
// [3] voxels whose std::sort order is not the input order, [4] of those, voxels whose centroid xyz BITS differ,
// [5] ... whose packed rgba differs, [6] points
extern "C" void ma_voxel_sort(const Pt *in, int n, double resolution, long long out[7])
{
  for (int i = 0; i < 7; ++i) out[i] = 0;
  const float leaf = (float)resolution, inv = 1.0f / leaf;
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  bool any = false;
  for (int i = 0; i < n; ++i) {
    if (!std::isfinite(in[i].x) || !std::isfinite(in[i].y) || !std::isfinite(in[i].z)) continue;
    const float v[3] = {in[i].x, in[i].y, in[i].z};
    for (int a = 0; a < 3; ++a) { if (!any || v[a] < mn[a]) mn[a] = v[a]; if (!any || v[a] > mx[a]) mx[a] = v[a]; }
    any = true;
  }
  if (!any) return;
  int min_b[3], div_b[3];
  for (int a = 0; a < 3; ++a) { min_b[a] = (int)floorf(mn[a] * inv); div_b[a] = (int)floorf(mx[a] * inv) - min_b[a] + 1; }
  const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
  std::vector<Rec> rec;
  rec.reserve((size_t)n);
  for (int i = 0; i < n; ++i) {
    if (!std::isfinite(in[i].x) || !std::isfinite(in[i].y) || !std::isfinite(in[i].z)) continue;
    const int i0 = (int)(floorf(in[i].x * inv) - (float)min_b[0]), i1 = (int)(floorf(in[i].y * inv) - (float)min_b[1]),
              i2 = (int)(floorf(in[i].z * inv) - (float)min_b[2]);
    rec.push_back(Rec{(unsigned)(i0 + i1 * mul1 + i2 * mul2), (unsigned)i});
  }
  std::vector<Rec> stable = rec;
  std::sort(rec.begin(), rec.end(), std::less<Rec>());                 // PCL's call
  std::stable_sort(stable.begin(), stable.end(), std::less<Rec>());    // the oracle's order: input order inside a voxel
  out[6] = (long long)rec.size();
  for (size_t b = 0; b < rec.size();) {
    size_t e = b + 1;
    while (e < rec.size() && rec[e].idx == rec[b].idx) ++e;
    ++out[0];
    if (e - b >= 3) ++out[1];
    if (e - b >= 17) ++out[2];
    bool same = true;
    for (size_t j = b; j < e; ++j) same = same && rec[j].cloud_point_index == stable[j].cloud_point_index;
    if (!same) {
      ++out[3];
      float c1[3], c2[3];
      uint32_t r1, r2;
      centroid(in, &rec[b], &rec[e - 1] + 1, c1, &r1);
      centroid(in, &stable[b], &stable[e - 1] + 1, c2, &r2);
      if (std::memcmp(c1, c2, sizeof(c1)) != 0) ++out[4];
      if (r1 != r2) ++out[5];
    }
    b = e;
  }
}

// pcl::VoxelGrid as o_filters.c::mo_downsample restates it, but with the points of a voxel taken in the order THIS libstdc++'s
// std::sort leaves them in (PCL's own call) instead of input order: what the first stage of the path would hand on if PCL 1.8.1's
// libstdc++ orders equal keys like this one.  scripts/audit_exposure.py runs the rest of the oracle on both clouds and counts
// what changes downstream.  out has room for n points; returns the number of voxels.
extern "C" int ma_downsample_stdsort(const Pt *in, int n, double resolution, Pt *out)
{
  const float leaf = (float)resolution, inv = 1.0f / leaf;
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  bool any = false;
  for (int i = 0; i < n; ++i) {
    if (!std::isfinite(in[i].x) || !std::isfinite(in[i].y) || !std::isfinite(in[i].z)) continue;
    const float v[3] = {in[i].x, in[i].y, in[i].z};
    for (int a = 0; a < 3; ++a) { if (!any || v[a] < mn[a]) mn[a] = v[a]; if (!any || v[a] > mx[a]) mx[a] = v[a]; }
    any = true;
  }
  if (!any) return 0;
  int min_b[3], div_b[3];
  for (int a = 0; a < 3; ++a) { min_b[a] = (int)floorf(mn[a] * inv); div_b[a] = (int)floorf(mx[a] * inv) - min_b[a] + 1; }
  const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
  std::vector<Rec> rec;
  rec.reserve((size_t)n);
  for (int i = 0; i < n; ++i) {
    if (!std::isfinite(in[i].x) || !std::isfinite(in[i].y) || !std::isfinite(in[i].z)) continue;
    const int i0 = (int)(floorf(in[i].x * inv) - (float)min_b[0]), i1 = (int)(floorf(in[i].y * inv) - (float)min_b[1]),
              i2 = (int)(floorf(in[i].z * inv) - (float)min_b[2]);
    rec.push_back(Rec{(unsigned)(i0 + i1 * mul1 + i2 * mul2), (unsigned)i});
  }
  std::sort(rec.begin(), rec.end(), std::less<Rec>());
  int nout = 0;
  for (size_t b = 0; b < rec.size();) {
    size_t e = b + 1;
    while (e < rec.size() && rec[e].idx == rec[b].idx) ++e;
    float c[3];
    uint32_t rgba;
    centroid(in, &rec[b], &rec[e - 1] + 1, c, &rgba);
    out[nout].x = c[0]; out[nout].y = c[1]; out[nout].z = c[2]; out[nout].rgba = rgba;
    ++nout;
    b = e;
  }
  return nout;
}
