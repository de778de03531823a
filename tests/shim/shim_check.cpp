// shim_check.cpp -- TEST: the reference-side binding (include/map_merge_3d_shim.hpp) as ONE translation
// unit, compiled against the reference's own public headers (-I R/include) and the stand-ins of
// tests/shim/mock (this image has no PCL / ROS / Eigen), linked against libmm3d.so.
//
//   shim_check cpu           the reference's five gtest cases (R/test/test_map_merging.cpp:9-40) and the
//                            MapMergingParams members the shim has to define because INTEGRATION.md
//                            removes R/src/map_merging.cpp (fromCommandLine :10-54, fromROSNode :56-98,
//                            operator<< :100-123); needs no device
//   shim_check gpu IN OUT    every free function of features.h / matching.h / map_merging.h through the
//                            shim on the clouds in IN; results to OUT for tests/test_gpu_shim.py, which
//                            holds them against direct calls of the C ABI
#define MM3D_SHIM_IMPLEMENTATION
#include <map_merge_3d_shim.hpp>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

using namespace map_merge_3d;
using Eigen::Matrix4f;

static int failures = 0;
#define EXPECT(cond)                                                              \
  do {                                                                            \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

static int run_cpu()
{
  // R/test/test_map_merging.cpp:9-13
  EXPECT(estimateMapsTransforms({}, MapMergingParams()).empty());
  {  // :15-21
    std::vector<Matrix4f> r = estimateMapsTransforms({PointCloudConstPtr(new PointCloud)}, MapMergingParams());
    EXPECT(r.size() == 1);
    EXPECT(r.size() == 1 && r[0] == Matrix4f::Identity());
  }
  EXPECT(composeMaps({}, {}, 0.0) == nullptr);   // :23-27
  {  // :29-32, EXPECT_ANY_THROW
    bool thrown = false;
    try { composeMaps({nullptr}, {}, 0.0); } catch (...) { thrown = true; }
    EXPECT(thrown);
  }
  {  // :34-40
    PointCloudPtr r = composeMaps({PointCloudConstPtr(new PointCloud)}, {Matrix4f::Identity()}, 0.0);
    EXPECT(r != nullptr);
    EXPECT(r && r->size() == 0);
  }
  {  // defaults of map_merging.h:28-44 survive the round trip through mm3d_params
    const MapMergingParams d, p = MapMergingParams::fromCommandLine(0, nullptr);
    EXPECT(p.resolution == d.resolution && p.descriptor_radius == d.descriptor_radius && p.normal_radius == d.normal_radius);
    EXPECT(p.outliers_min_neighbours == d.outliers_min_neighbours && p.keypoint_type == d.keypoint_type);
    EXPECT(p.keypoint_threshold == d.keypoint_threshold && p.descriptor_type == d.descriptor_type);
    EXPECT(p.estimation_method == d.estimation_method && p.refine_transform == d.refine_transform);
    EXPECT(p.inlier_threshold == d.inlier_threshold && p.max_correspondence_distance == d.max_correspondence_distance);
    EXPECT(p.max_iterations == d.max_iterations && p.matching_k == d.matching_k && p.transform_epsilon == d.transform_epsilon);
    EXPECT(p.confidence_threshold == d.confidence_threshold && p.output_resolution == d.output_resolution);
  }
  {
    const char *argv[] = {"tool", "--resolution", "0.2", "--descriptor_type", "FPFH", "--estimation_method", "SAC_IA",
                          "--matching_k", "-3", "--refine_transform", "0", "--max_iterations", "77", "a.pcd"};
    const MapMergingParams p = MapMergingParams::fromCommandLine(14, const_cast<char **>(argv));
    EXPECT(p.resolution == 0.2);
    EXPECT(p.descriptor_radius == 0.1 * 8.0);      // dependent defaults are fixed at construction (map_merging.h:30)
    EXPECT(p.descriptor_type == Descriptor::FPFH && p.estimation_method == EstimationMethod::SAC_IA);
    EXPECT(p.matching_k == 5 && !p.refine_transform && p.max_iterations == 77);
    bool thrown = false;
    const char *bad[] = {"tool", "--keypoint_type", "FAST"};
    try { MapMergingParams::fromCommandLine(3, const_cast<char **>(bad)); } catch (const std::runtime_error &) { thrown = true; }
    EXPECT(thrown);
  }
  {
    ros::NodeHandle n;
    n.values = {{"resolution", "0.05"}, {"keypoint_type", "HARRIS"}, {"descriptor_type", "SHOT"}, {"matching_k", "9"},
                {"refine_transform", "false"}, {"confidence_threshold", "1.5"}};
    const MapMergingParams p = MapMergingParams::fromROSNode(n);
    EXPECT(p.resolution == 0.05 && p.keypoint_type == Keypoint::HARRIS && p.descriptor_type == Descriptor::SHOT);
    EXPECT(p.matching_k == 9 && !p.refine_transform && p.confidence_threshold == 1.5);
    EXPECT(p.estimation_method == EstimationMethod::MATCHING && p.max_iterations == 500);
    n.values = {{"estimation_method", "nope"}};
    bool thrown = false;
    try { MapMergingParams::fromROSNode(n); } catch (const std::runtime_error &) { thrown = true; }
    EXPECT(thrown);
  }
  {
    std::ostringstream s;
    s << MapMergingParams();
    const std::string t = s.str();
    EXPECT(t.find("resolution: 0.1\n") == 0);
    EXPECT(t.find("descriptor_type: PFH\n") != std::string::npos && t.find("estimation_method: MATCHING\n") != std::string::npos);
    EXPECT(t.find("matching_k: 5\n") != std::string::npos && t.size() > 0 && t.back() == '\n' && t.find('\0') == std::string::npos);
  }
  return failures;
}

// ---- gpu mode: little-endian records, read and written by tests/test_gpu_shim.py -----------------
static void put(std::ofstream &f, const void *p, size_t n) { f.write(static_cast<const char *>(p), static_cast<std::streamsize>(n)); }
static void put_u64(std::ofstream &f, uint64_t v) { put(f, &v, 8); }
static void put_cloud(std::ofstream &f, const PointCloud &c)
{
  put_u64(f, c.points.size());
  for (const PointT &p : c.points) { put(f, &p.x, 12); put(f, &p.rgba, 4); }
}
static void put_T(std::ofstream &f, const Matrix4f &T) { put(f, T.data(), 64); }

static int run_gpu(const char *in_path, const char *out_path)
{
  std::ifstream in(in_path, std::ios::binary);
  uint64_t n_clouds = 0;
  in.read(reinterpret_cast<char *>(&n_clouds), 8);
  std::vector<PointCloudConstPtr> clouds;
  for (uint64_t i = 0; i < n_clouds; ++i) {
    uint64_t n = 0;
    in.read(reinterpret_cast<char *>(&n), 8);
    PointCloudPtr c(new PointCloud);
    c->points.resize(n);
    for (PointT &p : c->points) { in.read(reinterpret_cast<char *>(&p.x), 12); in.read(reinterpret_cast<char *>(&p.rgba), 4); }
    c->width = static_cast<uint32_t>(n); c->height = 1;
    clouds.push_back(c);
  }
  if (!in || clouds.size() < 2) { std::printf("bad input file\n"); return 2; }
  const char *argv[] = {"shim_check", "--descriptor_type", "FPFH", "--estimation_method", "MATCHING"};
  const MapMergingParams params = MapMergingParams::fromCommandLine(5, const_cast<char **>(argv));
  std::ofstream out(out_path, std::ios::binary);

  // the per-cloud loop of R/src/map_merging.cpp:212-242, one reference function at a time
  std::vector<PointCloudPtr> pts(2), kps(2);
  std::vector<LocalDescriptorsPtr> desc(2);
  for (int i = 0; i < 2; ++i) {
    PointCloudPtr d = downSample(clouds[i], params.resolution);
    pts[i] = removeOutliers(d, params.descriptor_radius, params.outliers_min_neighbours);
    SurfaceNormalsPtr nrm = computeSurfaceNormals(pts[i], params.normal_radius);
    kps[i] = detectKeypoints(pts[i], nrm, params.keypoint_type, params.keypoint_threshold, params.normal_radius, params.resolution);
    desc[i] = computeLocalDescriptors(pts[i], nrm, kps[i], params.descriptor_type, params.descriptor_radius);
    put_cloud(out, *d); put_cloud(out, *pts[i]);
    put_u64(out, nrm->points.size());
    for (const NormalT &q : nrm->points) { put(out, &q.normal_x, 12); put(out, &q.curvature, 4); }
    put_cloud(out, *kps[i]);
    put_u64(out, desc[i]->width); put_u64(out, desc[i]->point_step);
    put(out, desc[i]->data.data(), desc[i]->data.size());
    EXPECT(desc[i]->fields.size() == 1 && desc[i]->fields[0].name == "fpfh" && desc[i]->fields[0].count == 33);
  }
  // the per-pair body, :256-269
  CorrespondencesPtr corr = findFeatureCorrespondences(desc[0], desc[1], params.matching_k);
  put_u64(out, corr->size());
  for (const auto &c : *corr) { put(out, &c.index_query, 4); put(out, &c.index_match, 4); put(out, &c.distance, 4); }
  CorrespondencesPtr inliers;
  const Matrix4f T_ransac = estimateTransformFromCorrespondences(kps[0], kps[1], corr, inliers, params.inlier_threshold);
  put_T(out, T_ransac); put_u64(out, inliers->size());
  const Matrix4f T_icp = estimateTransformICP(pts[0], pts[1], T_ransac, params.max_correspondence_distance, params.inlier_threshold,
                                              params.max_iterations, params.transform_epsilon);
  put_T(out, T_icp);
  const Matrix4f T_est = estimateTransform(pts[0], kps[0], desc[0], pts[1], kps[1], desc[1], params.estimation_method,
                                           params.refine_transform, params.inlier_threshold, params.max_correspondence_distance,
                                           params.max_iterations, params.matching_k, params.transform_epsilon);
  put_T(out, T_est);
  const double score = transformScore(pts[0], pts[1], T_est, params.max_correspondence_distance);
  put(out, &score, 8);
  const Matrix4f T_sac = estimateTransformFromDescriptorsSets(kps[0], desc[0], kps[1], desc[1], params.inlier_threshold,
                                                              params.max_correspondence_distance, params.max_iterations);
  put_T(out, T_sac);
  // the entry point itself and the compositing that follows it in map_merge_tool.cpp:37-49
  const std::vector<Matrix4f> Ts = estimateMapsTransforms(clouds, params);
  put_u64(out, Ts.size());
  for (const Matrix4f &T : Ts) put_T(out, T);
  std::vector<PointCloudConstPtr> used(clouds.begin(), clouds.begin() + static_cast<long>(Ts.size()));
  PointCloudPtr merged = composeMaps(used, Ts, params.output_resolution);
  put_cloud(out, *merged);
  EXPECT(out.good());
  return failures;
}

int main(int argc, char **argv)
{
  try {
    if (argc >= 2 && std::string(argv[1]) == "cpu") {
      const int f = run_cpu();
      std::printf(f ? "shim_check cpu: %d failure(s)\n" : "shim_check cpu: ok\n", f);
      return f ? 1 : 0;
    }
    if (argc >= 4 && std::string(argv[1]) == "gpu") {
      const int f = run_gpu(argv[2], argv[3]);
      std::printf(f ? "shim_check gpu: %d failure(s)\n" : "shim_check gpu: ok\n", f);
      return f ? 1 : 0;
    }
  } catch (const std::exception &e) {
    std::printf("shim_check: exception: %s\n", e.what());
    return 3;
  }
  std::printf("usage: shim_check cpu | shim_check gpu IN OUT\n");
  return 2;
}
