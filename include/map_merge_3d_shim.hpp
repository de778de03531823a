// map_merge_3d_shim.hpp -- the reference-side binding a maintainer adds to use libmm3d.so.
//
// Drop-in for the static library `map_merging` (R/CMakeLists.txt:67-74): it defines the SAME free
// functions with the SAME signatures as R/include/map_merge_3d/{features,matching,map_merging}.h
// and forwards each to the C ABI of include/mm3d.h.  map_merge_node.cpp, map_merge_tool.cpp and
// registration_visualisation.cpp compile unchanged against the reference's own headers and link
// this translation unit + libmm3d.so instead of features.cpp / matching.cpp / map_merging.cpp /
// graph.cpp.  Needs PCL and ROS headers (the reference's headers include them), so it is compiled
// only where they exist; this image has neither, hence the __has_include guard.
//
//   g++ -std=c++14 -I<ref>/include -Iinclude -DMM3D_SHIM_IMPLEMENTATION -c mm3d_shim.cpp   (a .cpp that
//   includes this header once), then link with -lmm3d.
#pragma once

#if defined(__has_include)
#if __has_include(<pcl/point_cloud.h>) && __has_include(<map_merge_3d/map_merging.h>)
#define MM3D_SHIM_AVAILABLE 1
#endif
#endif

#ifdef MM3D_SHIM_AVAILABLE

#include <map_merge_3d/map_merging.h>
#include <pcl/conversions.h>

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>

#include "mm3d.h"

namespace map_merge_3d
{
namespace mm3d_shim
{
// One engine per process, created on first use (the reference functions are stateless free
// functions; the context only caches device memory and carries the rand() replay).
inline mm3d_ctx *ctx()
{
  static mm3d_ctx *c = [] {
    mm3d_ctx *p = nullptr;
    if (mm3d_create(0, &p) != MM3D_OK) throw std::runtime_error("mm3d: no MI355X device");
    // estimateMapsTransforms deals its per-cloud and per-pair loops to 16 HIP streams inside the
    // library (same bits as one stream, about twice the throughput); MM3D_STREAMS overrides
    const char *s = std::getenv("MM3D_STREAMS");
    const int n = s ? std::atoi(s) : 16;
    (void)mm3d_set_streams(p, n >= 1 && n <= 64 ? n : 16);
    return p;
  }();
  return c;
}
inline void check(int st)
{
  if (st != MM3D_OK) throw std::runtime_error(std::string("mm3d: ") + mm3d_last_error(ctx()));
}
// pcl::PointXYZRGB is 32 bytes: x,y,z,pad | rgba,pad,pad,pad  -> stride 32, rgba_offset 16
inline mm3d_cloud *upload(const PointCloud &c)
{
  mm3d_cloud *h = nullptr;
  check(mm3d_cloud_create(ctx(), c.points.data(), c.points.size(), sizeof(PointT), offsetof(PointT, rgba), &h));
  return h;
}
inline PointCloudPtr download(mm3d_cloud *h)
{
  PointCloudPtr out(new PointCloud);
  out->points.resize(mm3d_cloud_size(h));
  out->width = static_cast<uint32_t>(out->points.size());
  out->height = 1;
  out->is_dense = true;
  check(mm3d_cloud_download(ctx(), h, out->points.data(), sizeof(PointT), offsetof(PointT, rgba)));
  return out;
}
inline mm3d_normals *upload(const SurfaceNormals &n)
{
  mm3d_normals *h = nullptr;
  // pcl::Normal is 32 bytes: nx,ny,nz,pad | curvature,... : repack to 16-byte records first
  std::vector<float> tmp(n.points.size() * 4);
  for (size_t i = 0; i < n.points.size(); ++i) {
    tmp[4 * i] = n.points[i].normal_x; tmp[4 * i + 1] = n.points[i].normal_y;
    tmp[4 * i + 2] = n.points[i].normal_z; tmp[4 * i + 3] = n.points[i].curvature;
  }
  check(mm3d_normals_create(ctx(), tmp.data(), n.points.size(), 16, &h));
  return h;
}
// PCLPointCloud2 <-> mm3d_desc: fields[0].name selects the descriptor (dispatch_descriptors.h:104-111)
inline int descriptor_from_field(const std::string &name)
{
  for (int d = 0; d < 6; ++d)
    if (name == mm3d_descriptor_field_name(d)) return d;
  throw std::runtime_error("unknown descriptor type");
}
inline mm3d_desc *upload(const LocalDescriptors &d)
{
  if (d.fields.empty()) throw std::runtime_error("descriptors must contain at least one field with descriptors.");
  const int type = descriptor_from_field(d.fields[0].name);
  const int dim = mm3d_descriptor_dim(type);
  const size_t n = static_cast<size_t>(d.width) * d.height;
  std::vector<float> tmp(n * dim);
  for (size_t i = 0; i < n; ++i)
    std::memcpy(&tmp[i * dim], &d.data[i * d.point_step + d.fields[0].offset], sizeof(float) * dim);
  mm3d_desc *h = nullptr;
  check(mm3d_desc_create(ctx(), tmp.data(), n, type, &h));
  return h;
}
inline Eigen::Matrix4f to_eigen(const float *T)
{
  return Eigen::Map<const Eigen::Matrix4f>(T);   // both column-major
}
inline mm3d_params to_params(const MapMergingParams &p)
{
  mm3d_params q;
  q.resolution = p.resolution; q.descriptor_radius = p.descriptor_radius;
  q.outliers_min_neighbours = p.outliers_min_neighbours; q.normal_radius = p.normal_radius;
  q.keypoint_type = static_cast<int>(p.keypoint_type); q.keypoint_threshold = p.keypoint_threshold;
  q.descriptor_type = static_cast<int>(p.descriptor_type); q.estimation_method = static_cast<int>(p.estimation_method);
  q.refine_transform = p.refine_transform; q.inlier_threshold = p.inlier_threshold;
  q.max_correspondence_distance = p.max_correspondence_distance; q.max_iterations = p.max_iterations;
  q.matching_k = p.matching_k; q.transform_epsilon = p.transform_epsilon;
  q.confidence_threshold = p.confidence_threshold; q.output_resolution = p.output_resolution;
  return q;
}
}  // namespace mm3d_shim

#ifdef MM3D_SHIM_IMPLEMENTATION

// R/include/map_merge_3d/features.h:34
PointCloudPtr downSample(const PointCloudConstPtr &input, double resolution)
{
  using namespace mm3d_shim;
  mm3d_cloud *in = upload(*input), *out = nullptr;
  check(mm3d_downsample(ctx(), in, resolution, &out));
  PointCloudPtr r = download(out);
  mm3d_cloud_free(ctx(), in); mm3d_cloud_free(ctx(), out);
  return r;
}

// features.h:45
PointCloudPtr removeOutliers(const PointCloudConstPtr &input, double radius, int min_neighbors)
{
  using namespace mm3d_shim;
  mm3d_cloud *in = upload(*input), *out = nullptr;
  check(mm3d_remove_outliers(ctx(), in, radius, min_neighbors, &out));
  PointCloudPtr r = download(out);
  mm3d_cloud_free(ctx(), in); mm3d_cloud_free(ctx(), out);
  return r;
}

// features.h:97
SurfaceNormalsPtr computeSurfaceNormals(const PointCloudConstPtr &input, double radius)
{
  using namespace mm3d_shim;
  mm3d_cloud *in = upload(*input);
  mm3d_normals *n = nullptr;
  check(mm3d_compute_normals(ctx(), in, radius, &n));
  std::vector<float> tmp(mm3d_normals_size(n) * 4);
  check(mm3d_normals_download(ctx(), n, tmp.data(), 16));
  SurfaceNormalsPtr out(new SurfaceNormals);
  out->points.resize(tmp.size() / 4);
  out->width = static_cast<uint32_t>(out->points.size()); out->height = 1; out->is_dense = true;
  for (size_t i = 0; i < out->points.size(); ++i) {
    out->points[i].normal_x = tmp[4 * i]; out->points[i].normal_y = tmp[4 * i + 1];
    out->points[i].normal_z = tmp[4 * i + 2]; out->points[i].curvature = tmp[4 * i + 3];
    if (!std::isfinite(tmp[4 * i])) out->is_dense = false;
  }
  mm3d_cloud_free(ctx(), in); mm3d_normals_free(ctx(), n);
  return out;
}

// features.h:65
PointCloudPtr detectKeypoints(const PointCloudConstPtr &points, const SurfaceNormalsPtr &normals, Keypoint type,
                              double threshold, double radius, double resolution)
{
  using namespace mm3d_shim;
  mm3d_cloud *in = upload(*points), *kp = nullptr;
  mm3d_normals *n = normals ? upload(*normals) : nullptr;
  check(mm3d_detect_keypoints(ctx(), in, n, static_cast<int>(type), threshold, radius, resolution, &kp));
  PointCloudPtr r = download(kp);
  mm3d_cloud_free(ctx(), in); mm3d_cloud_free(ctx(), kp);
  if (n) mm3d_normals_free(ctx(), n);
  return r;
}

// features.h:83 -- prunes `keypoints` in place like the reference (features.cpp:137-141)
LocalDescriptorsPtr computeLocalDescriptors(const PointCloudConstPtr &points, const SurfaceNormalsPtr &normals,
                                            const PointCloudPtr &keypoints, Descriptor descriptor, double feature_radius)
{
  using namespace mm3d_shim;
  mm3d_cloud *in = upload(*points), *kp = upload(*keypoints);
  mm3d_normals *n = upload(*normals);
  mm3d_desc *d = nullptr;
  check(mm3d_compute_descriptors(ctx(), in, n, kp, static_cast<int>(descriptor), feature_radius, &d));
  *keypoints = *download(kp);
  const int dim = mm3d_desc_dim(d);
  const size_t cnt = mm3d_desc_size(d);
  LocalDescriptorsPtr out(new LocalDescriptors);
  out->fields.resize(1);
  out->fields[0].name = mm3d_descriptor_field_name(static_cast<int>(descriptor));
  out->fields[0].offset = 0; out->fields[0].datatype = pcl::PCLPointField::FLOAT32; out->fields[0].count = dim;
  out->point_step = sizeof(float) * dim; out->width = static_cast<uint32_t>(cnt); out->height = 1;
  out->row_step = out->point_step * out->width; out->is_dense = true;
  out->data.resize(out->row_step);
  check(mm3d_desc_download(ctx(), d, reinterpret_cast<float *>(out->data.data())));
  mm3d_cloud_free(ctx(), in); mm3d_cloud_free(ctx(), kp); mm3d_normals_free(ctx(), n); mm3d_desc_free(ctx(), d);
  return out;
}

// matching.h:26
CorrespondencesPtr findFeatureCorrespondences(const LocalDescriptorsPtr &source_descriptors,
                                              const LocalDescriptorsPtr &target_descriptors, size_t k)
{
  using namespace mm3d_shim;
  mm3d_desc *s = upload(*source_descriptors), *t = upload(*target_descriptors);
  size_t n = 0;
  check(mm3d_find_correspondences(ctx(), s, t, k, nullptr, 0, &n));
  std::vector<mm3d_corr> buf(n ? n : 1);
  check(mm3d_find_correspondences(ctx(), s, t, k, buf.data(), buf.size(), &n));
  CorrespondencesPtr out(new Correspondences);
  for (size_t i = 0; i < n; ++i) out->emplace_back(buf[i].index_query, buf[i].index_match, buf[i].distance);
  mm3d_desc_free(ctx(), s); mm3d_desc_free(ctx(), t);
  return out;
}

// matching.h:44
Eigen::Matrix4f estimateTransformFromCorrespondences(const PointCloudPtr &source_keypoints, const PointCloudPtr &target_keypoints,
                                                     const CorrespondencesPtr &correspondences, CorrespondencesPtr &inliers,
                                                     double inlier_threshold)
{
  using namespace mm3d_shim;
  mm3d_cloud *s = upload(*source_keypoints), *t = upload(*target_keypoints);
  std::vector<mm3d_corr> c(correspondences->size()), inl(correspondences->size() + 1);
  for (size_t i = 0; i < c.size(); ++i) c[i] = {(*correspondences)[i].index_query, (*correspondences)[i].index_match, (*correspondences)[i].distance};
  float T[16]; size_t n = 0;
  check(mm3d_estimate_transform_from_correspondences(ctx(), s, t, c.data(), c.size(), inlier_threshold, T, inl.data(), inl.size(), &n));
  inliers.reset(new Correspondences);
  for (size_t i = 0; i < n; ++i) inliers->emplace_back(inl[i].index_query, inl[i].index_match, inl[i].distance);
  mm3d_cloud_free(ctx(), s); mm3d_cloud_free(ctx(), t);
  return to_eigen(T);
}

// matching.h:68
Eigen::Matrix4f estimateTransformFromDescriptorsSets(const PointCloudPtr &source_keypoints, const LocalDescriptorsPtr &source_descriptors,
                                                     const PointCloudPtr &target_keypoints, const LocalDescriptorsPtr &target_descriptors,
                                                     double min_sample_distance, double max_correspondence_distance, int max_iterations)
{
  using namespace mm3d_shim;
  mm3d_cloud *s = upload(*source_keypoints), *t = upload(*target_keypoints);
  mm3d_desc *sd = upload(*source_descriptors), *td = upload(*target_descriptors);
  float T[16];
  check(mm3d_estimate_transform_from_descriptors(ctx(), s, sd, t, td, min_sample_distance, max_correspondence_distance, max_iterations, T));
  mm3d_cloud_free(ctx(), s); mm3d_cloud_free(ctx(), t); mm3d_desc_free(ctx(), sd); mm3d_desc_free(ctx(), td);
  return to_eigen(T);
}

// matching.h:94
Eigen::Matrix4f estimateTransformICP(const PointCloudPtr &source_points, const PointCloudPtr &target_points,
                                     const Eigen::Matrix4f &initial_guess, double max_correspondence_distance,
                                     double outlier_rejection_threshold, int max_iterations, double transformation_epsilon)
{
  using namespace mm3d_shim;
  mm3d_cloud *s = upload(*source_points), *t = upload(*target_points);
  float T[16];
  check(mm3d_estimate_transform_icp(ctx(), s, t, initial_guess.data(), max_correspondence_distance, outlier_rejection_threshold,
                                    max_iterations, transformation_epsilon, T));
  mm3d_cloud_free(ctx(), s); mm3d_cloud_free(ctx(), t);
  return to_eigen(T);
}

// matching.h:129
Eigen::Matrix4f estimateTransform(const PointCloudPtr &source_points, const PointCloudPtr &source_keypoints,
                                  const LocalDescriptorsPtr &source_descriptors, const PointCloudPtr &target_points,
                                  const PointCloudPtr &target_keypoints, const LocalDescriptorsPtr &target_descriptors,
                                  EstimationMethod method, bool refine, double inlier_threshold, double max_correspondence_distance,
                                  int max_iterations, size_t matching_k, double transform_epsilon)
{
  using namespace mm3d_shim;
  mm3d_cloud *sp = upload(*source_points), *sk = upload(*source_keypoints), *tp = upload(*target_points), *tk = upload(*target_keypoints);
  mm3d_desc *sd = upload(*source_descriptors), *td = upload(*target_descriptors);
  float T[16];
  check(mm3d_estimate_transform(ctx(), sp, sk, sd, tp, tk, td, static_cast<int>(method), refine, inlier_threshold,
                                max_correspondence_distance, max_iterations, matching_k, transform_epsilon, T));
  for (mm3d_cloud *c : {sp, sk, tp, tk}) mm3d_cloud_free(ctx(), c);
  mm3d_desc_free(ctx(), sd); mm3d_desc_free(ctx(), td);
  return to_eigen(T);
}

// matching.h:150
double transformScore(const PointCloudPtr &source_points, const PointCloudPtr &target_points, const Eigen::Matrix4f &transform,
                      double max_distance)
{
  using namespace mm3d_shim;
  mm3d_cloud *s = upload(*source_points), *t = upload(*target_points);
  double score = 0;
  check(mm3d_transform_score(ctx(), s, t, transform.data(), max_distance, &score));
  mm3d_cloud_free(ctx(), s); mm3d_cloud_free(ctx(), t);
  return score;
}

// map_merging.h:85 -- the whole hot path in one call; clouds go to the device once
std::vector<Eigen::Matrix4f> estimateMapsTransforms(const std::vector<PointCloudConstPtr> &clouds, const MapMergingParams &params)
{
  using namespace mm3d_shim;
  std::vector<mm3d_cloud_view> views(clouds.size());
  for (size_t i = 0; i < clouds.size(); ++i) {
    // a robot that is subscribed but has no map yet hands over nullptr (map_merge_node.cpp:171)
    views[i].points = clouds[i] ? clouds[i]->points.data() : nullptr;
    views[i].n = clouds[i] ? clouds[i]->points.size() : 0;
    views[i].stride = sizeof(PointT);
    views[i].rgba_offset = offsetof(PointT, rgba);
  }
  std::vector<float> out(16 * (clouds.size() ? clouds.size() : 1));
  size_t n_out = 0;
  const mm3d_params p = to_params(params);
  check(mm3d_estimate_maps_transforms(ctx(), views.data(), views.size(), &p, out.data(), &n_out, nullptr, nullptr));
  std::vector<Eigen::Matrix4f> result(n_out);
  for (size_t i = 0; i < n_out; ++i) result[i] = to_eigen(&out[16 * i]);
  return result;
}

// map_merging.h:99
PointCloudPtr composeMaps(const std::vector<PointCloudConstPtr> &clouds, const std::vector<Eigen::Matrix4f> &transforms, double resolution)
{
  using namespace mm3d_shim;
  if (clouds.empty()) return nullptr;
  if (clouds.size() != transforms.size())
    throw new std::runtime_error("composeMaps: clouds and transforms size must be the same.");   // a pointer, like the reference
  std::vector<mm3d_cloud *> h(clouds.size());
  std::vector<float> T(16 * clouds.size());
  for (size_t i = 0; i < clouds.size(); ++i) {
    h[i] = upload(*clouds[i]);
    std::memcpy(&T[16 * i], transforms[i].data(), sizeof(float) * 16);
  }
  mm3d_cloud *out = nullptr;
  check(mm3d_compose_maps(ctx(), h.data(), h.size(), T.data(), transforms.size(), resolution, &out));
  PointCloudPtr r = download(out);
  for (mm3d_cloud *c : h) mm3d_cloud_free(ctx(), c);
  mm3d_cloud_free(ctx(), out);
  return r;
}

#endif  // MM3D_SHIM_IMPLEMENTATION
}  // namespace map_merge_3d

#endif  // MM3D_SHIM_AVAILABLE
