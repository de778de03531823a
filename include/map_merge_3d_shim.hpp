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

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "mm3d.h"

namespace map_merge_3d
{
namespace mm3d_shim
{
// MM3D_DEVICES="0,1,2,3,4,5,6,7" (or "all"): the node's ONE process uses that many GPUs -- estimateMapsTransforms is then
// sharded over them inside the library, the pair records gathered by one RCCL all-gather (mm3d_create_devices); no source
// change in the node.  Otherwise one GPU: MM3D_DEVICE (default 0).
inline mm3d_ctx *make_ctx(int streams, bool may_use_device_list = false)
{
  mm3d_ctx *p = nullptr;
  const char *list = may_use_device_list ? std::getenv("MM3D_DEVICES") : nullptr;
  if (list && *list) {
    std::vector<int> devs;
    if (std::string(list) == "all") {
      for (int d = 0; d < 64; ++d) {                       // as many as exist: creation fails at the first that does not
        mm3d_ctx *probe = nullptr;
        if (mm3d_create(d, &probe) != MM3D_OK) break;
        mm3d_destroy(probe);
        devs.push_back(d);
      }
    } else {
      for (const char *q = list; *q;) {
        devs.push_back(std::atoi(q));
        while (*q && *q != ',') ++q;
        if (*q == ',') ++q;
      }
    }
    if (devs.empty() || mm3d_create_devices(devs.data(), (int)devs.size(), &p) != MM3D_OK)
      throw std::runtime_error("mm3d: MM3D_DEVICES does not name usable MI355X devices");
  } else {
    const char *d = std::getenv("MM3D_DEVICE");
    if (mm3d_create(d ? std::atoi(d) : 0, &p) != MM3D_OK) throw std::runtime_error("mm3d: no MI355X device");
  }
  (void)mm3d_set_streams(p, streams);
  return p;
}
// Two engines per process, created on first use (the reference functions are stateless free
// functions; a context only caches device memory and carries the rand() replay).  A context
// serialises its calls, and the node runs mapCompositing (0.3 Hz) beside transformsEstimation
// (0.01 Hz, seconds long) under a MultiThreadedSpinner (R/src/map_merge_node.cpp:32-40,264-265),
// so composeMaps has a context of its own and never queues behind an estimation.
inline mm3d_ctx *ctx()
{
  // estimateMapsTransforms deals its per-cloud and per-pair loops to 16 HIP streams inside the
  // library (same bits as one stream, about twice the throughput); MM3D_STREAMS overrides
  static mm3d_ctx *c = [] {
    const char *s = std::getenv("MM3D_STREAMS");
    const int n = s ? std::atoi(s) : 16;
    return make_ctx(n >= 1 && n <= 64 ? n : 16, true);      // (the estimation engine is the one that may span several GPUs)
  }();
  return c;
}
inline mm3d_ctx *compose_ctx()
{
  static mm3d_ctx *c = make_ctx(1);
  return c;
}
inline void check(mm3d_ctx *c, int st)
{
  if (st != MM3D_OK) throw std::runtime_error(std::string("mm3d: ") + mm3d_last_error(c));
}
inline void check(int st) { check(ctx(), st); }
// pcl::PointXYZRGB is 32 bytes: x,y,z,pad | rgba,pad,pad,pad  -> stride 32, rgba_offset 16
inline mm3d_cloud *upload(const PointCloud &c, mm3d_ctx *on = nullptr)
{
  mm3d_ctx *e = on ? on : ctx();
  mm3d_cloud *h = nullptr;
  check(e, mm3d_cloud_create(e, c.points.data(), c.points.size(), sizeof(PointT), offsetof(PointT, rgba), &h));
  return h;
}
inline PointCloudPtr download(mm3d_cloud *h, mm3d_ctx *on = nullptr)
{
  mm3d_ctx *e = on ? on : ctx();
  PointCloudPtr out(new PointCloud);
  out->points.resize(mm3d_cloud_size(h));
  out->width = static_cast<uint32_t>(out->points.size());
  out->height = 1;
  out->is_dense = true;
  check(e, mm3d_cloud_download(e, h, out->points.data(), sizeof(PointT), offsetof(PointT, rgba)));
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
inline MapMergingParams from_params(const mm3d_params &q)
{
  MapMergingParams p;
  p.resolution = q.resolution; p.descriptor_radius = q.descriptor_radius;
  p.outliers_min_neighbours = q.outliers_min_neighbours; p.normal_radius = q.normal_radius;
  p.keypoint_type = static_cast<Keypoint>(q.keypoint_type); p.keypoint_threshold = q.keypoint_threshold;
  p.descriptor_type = static_cast<Descriptor>(q.descriptor_type);
  p.estimation_method = static_cast<EstimationMethod>(q.estimation_method);
  p.refine_transform = q.refine_transform != 0; p.inlier_threshold = q.inlier_threshold;
  p.max_correspondence_distance = q.max_correspondence_distance; p.max_iterations = q.max_iterations;
  p.matching_k = static_cast<size_t>(q.matching_k); p.transform_epsilon = q.transform_epsilon;
  p.confidence_threshold = q.confidence_threshold; p.output_resolution = q.output_resolution;
  return p;
}
// a ROS parameter that names an enum value: absent or empty keeps the default, an unknown
// name throws like enums::from_string (R/include/map_merge_3d/enum.h:58)
inline void enum_param(const ros::NodeHandle &n, const char *key, int (*parse)(const char *), int *value)
{
  std::string s;
  n.getParam(key, s);
  if (s.empty()) return;
  const int v = parse(s.c_str());
  if (v < 0) throw std::runtime_error("string is not a valid enum value: " + s);
  *value = v;
}
}  // namespace mm3d_shim

#ifdef MM3D_SHIM_IMPLEMENTATION

// map_merging.h:53, R/src/map_merging.cpp:10-54 -- the parsing itself lives behind the C ABI
// (mm3d_params_from_command_line) so that non-C++ callers get the same behaviour
MapMergingParams MapMergingParams::fromCommandLine(int argc, char **argv)
{
  mm3d_params q;
  if (mm3d_params_from_command_line(argc, argv, &q) != MM3D_OK)
    throw std::runtime_error("string is not a valid enum value");   // enums::from_string, enum.h:58
  return mm3d_shim::from_params(q);
}

// map_merging.h:60, R/src/map_merging.cpp:56-98 -- the same keys read from the node's parameters;
// a missing key keeps the default, matching_k is applied only when positive
MapMergingParams MapMergingParams::fromROSNode(const ros::NodeHandle &n)
{
  mm3d_params q;
  mm3d_params_default(&q);
  bool refine = q.refine_transform != 0;
  int k = -1;
  n.getParam("resolution", q.resolution);
  n.getParam("descriptor_radius", q.descriptor_radius);
  n.getParam("outliers_min_neighbours", q.outliers_min_neighbours);
  n.getParam("normal_radius", q.normal_radius);
  mm3d_shim::enum_param(n, "keypoint_type", mm3d_keypoint_from_string, &q.keypoint_type);
  n.getParam("keypoint_threshold", q.keypoint_threshold);
  mm3d_shim::enum_param(n, "descriptor_type", mm3d_descriptor_from_string, &q.descriptor_type);
  mm3d_shim::enum_param(n, "estimation_method", mm3d_estimation_method_from_string, &q.estimation_method);
  n.getParam("refine_transform", refine);
  n.getParam("inlier_threshold", q.inlier_threshold);
  n.getParam("max_correspondence_distance", q.max_correspondence_distance);
  n.getParam("max_iterations", q.max_iterations);
  n.getParam("matching_k", k);
  n.getParam("transform_epsilon", q.transform_epsilon);
  n.getParam("confidence_threshold", q.confidence_threshold);
  n.getParam("output_resolution", q.output_resolution);
  q.refine_transform = refine;
  if (k > 0) q.matching_k = static_cast<uint64_t>(k);
  return mm3d_shim::from_params(q);
}

// map_merging.h:62, R/src/map_merging.cpp:100-123 -- "name: value" lines, mm3d_params_to_string
std::ostream &operator<<(std::ostream &stream, const MapMergingParams &params)
{
  const mm3d_params q = mm3d_shim::to_params(params);
  std::vector<char> text(mm3d_params_to_string(&q, nullptr, 0));   // size includes the terminating NUL
  mm3d_params_to_string(&q, text.data(), text.size());
  return stream << text.data();
}

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
  // R/src/map_merging.cpp:192-197: nothing to estimate, the clouds are not touched (and no device is needed)
  if (clouds.empty()) return {};
  if (clouds.size() == 1) return {Eigen::Matrix4f::Identity()};
  std::vector<mm3d_cloud_view> views(clouds.size());
  for (size_t i = 0; i < clouds.size(); ++i) {
    // a robot that is subscribed but has no map yet hands over nullptr (map_merge_node.cpp:171)
    views[i].points = clouds[i] ? clouds[i]->points.data() : nullptr;
    views[i].n = clouds[i] ? clouds[i]->points.size() : 0;
    views[i].stride = sizeof(PointT);
    views[i].rgba_offset = offsetof(PointT, rgba);
  }
  std::vector<float> out(16 * clouds.size());
  size_t n_out = 0;
  const mm3d_params p = to_params(params);
  check(mm3d_estimate_maps_transforms(ctx(), views.data(), views.size(), &p, out.data(), &n_out, nullptr, nullptr));
  std::vector<Eigen::Matrix4f> result(n_out);
  for (size_t i = 0; i < n_out; ++i) result[i] = to_eigen(&out[16 * i]);
  return result;
}

// map_merging.h:99 -- on the compositing context (see mm3d_shim::compose_ctx)
PointCloudPtr composeMaps(const std::vector<PointCloudConstPtr> &clouds, const std::vector<Eigen::Matrix4f> &transforms, double resolution)
{
  using namespace mm3d_shim;
  if (clouds.empty()) return nullptr;
  if (clouds.size() != transforms.size())
    throw new std::runtime_error("composeMaps: clouds and transforms size must be the same.");   // a pointer, like the reference
  size_t total = 0;
  for (const auto &c : clouds) total += c ? c->points.size() : 0;
  if (total == 0) {                                // nothing to transform or voxelise: an empty cloud, no device needed
    PointCloudPtr empty(new PointCloud);
    empty->is_dense = true;
    return empty;
  }
  mm3d_ctx *e = compose_ctx();
  const PointCloud none;
  std::vector<mm3d_cloud *> h(clouds.size());
  std::vector<float> T(16 * clouds.size());
  for (size_t i = 0; i < clouds.size(); ++i) {
    h[i] = upload(clouds[i] ? *clouds[i] : none, e);
    std::memcpy(&T[16 * i], transforms[i].data(), sizeof(float) * 16);
  }
  mm3d_cloud *out = nullptr;
  check(e, mm3d_compose_maps(e, h.data(), h.size(), T.data(), transforms.size(), resolution, &out));
  PointCloudPtr r = download(out, e);
  for (mm3d_cloud *c : h) mm3d_cloud_free(e, c);
  mm3d_cloud_free(e, out);
  return r;
}

#endif  // MM3D_SHIM_IMPLEMENTATION
}  // namespace map_merge_3d

#endif  // MM3D_SHIM_AVAILABLE
