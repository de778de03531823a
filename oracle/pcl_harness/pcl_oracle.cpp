// pcl_oracle.cpp -- TEST INFRASTRUCTURE (baseline B3 of SURVEY 8c(iii) / 8d): the real PCL behind the
// reference's call sites, for pinning the C restatement in oracle/*.c wherever PCL exists.
//
// Not part of the product and not built in this image: PCL, Eigen, FLANN and boost are absent here and on
// the GPU box, so `build.sh` beside this file prints "PCL absent -- oracle =
// restatement" and builds nothing.  On a machine with PCL >= 1.8 it builds `pcl_oracle`, and
// tests/test_oracle_cpu.py::test_oracle_against_real_pcl then holds every stage of the restatement
// against it.
//
// It calls the SAME PCL classes with the SAME setters, in the same order, as the reference:
//   downSample            R/src/features.cpp:17-27      pcl::VoxelGrid
//   removeOutliers        R/src/features.cpp:31-43      pcl::RadiusOutlierRemoval
//   computeSurfaceNormals R/src/features.cpp:168-179    pcl::NormalEstimation
//   detectKeypoints SIFT  R/src/features.cpp:45-62      pcl::SIFTKeypoint<PointXYZRGB, PointWithScale>, 3 octaves x 3 scales
//   FPFH descriptors      R/src/features.cpp:99-150 + dispatch_descriptors.h:40   pcl::FPFHEstimation, invalid rows pruned
//   reciprocal matching   R/src/matching.cpp:31-92      two pcl::search::KdTree<FPFHSignature33>, k nearest each way
//   RANSAC + SVD          R/src/matching.cpp:110-140    CorrespondenceRejectorSampleConsensus, TransformationEstimationSVD
//   SAC-IA                R/src/matching.cpp:142-174    SampleConsensusInitialAlignment (min sample distance, max
//                                                       correspondence distance, max iterations as mapped at :243-246)
//   ICP                   R/src/matching.cpp:196-221    IterativeClosestPoint on the pre-transformed source
//   transformScore        R/src/matching.cpp:259-268    TransformationValidationEuclidean
//
//   pcl_oracle IN OUT     IN: u64 n_clouds, then per cloud u64 n and n records {float x, y, z; u32 rgba}
//                         OUT: the same little-endian layout tests/shim/shim_check writes in gpu mode, for
//                         clouds 0 and 1 with the reference's default parameters except FPFH + both methods
#include <pcl/common/transforms.h>
#include <pcl/features/fpfh.h>
#include <pcl/features/normal_3d.h>
#include <pcl/filters/extract_indices.h>
#include <pcl/filters/radius_outlier_removal.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/keypoints/sift_keypoint.h>
#include <pcl/point_cloud.h>
#include <pcl/point_representation.h>
#include <pcl/point_types.h>
#include <pcl/registration/correspondence_rejection_sample_consensus.h>
#include <pcl/registration/ia_ransac.h>
#include <pcl/registration/icp.h>
#include <pcl/registration/transformation_estimation_svd.h>
#include <pcl/registration/transformation_validation_euclidean.h>
#include <pcl/search/kdtree.h>

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <vector>

typedef pcl::PointXYZRGB P;
typedef pcl::PointCloud<P> Cloud;
typedef pcl::PointCloud<pcl::Normal> Normals;
typedef pcl::PointCloud<pcl::FPFHSignature33> Fpfh;

// MapMergingParams defaults (R/include/map_merge_3d/map_merging.h:28-44)
static const double kResolution = 0.1, kDescriptorRadius = 0.8, kNormalRadius = 0.6, kKeypointThreshold = 5.0;
static const int kMinNeighbours = 50, kMaxIterations = 500;
static const double kInlierThreshold = 0.5, kMaxCorrespondenceDistance = 1.0, kTransformEpsilon = 1e-2;
static const size_t kMatchingK = 5;

static void put(std::ofstream &f, const void *p, size_t n) { f.write(static_cast<const char *>(p), static_cast<std::streamsize>(n)); }
static void put_u64(std::ofstream &f, uint64_t v) { put(f, &v, 8); }
static void put_cloud(std::ofstream &f, const Cloud &c)
{
  put_u64(f, c.size());
  for (const P &p : c.points) { put(f, &p.x, 12); put(f, &p.rgba, 4); }
}
static void put_T(std::ofstream &f, const Eigen::Matrix4f &T) { put(f, T.data(), 64); }

int main(int argc, char **argv)
{
  if (argc < 3) { std::printf("usage: pcl_oracle IN OUT\n"); return 2; }
  std::ifstream in(argv[1], std::ios::binary);
  uint64_t n_clouds = 0;
  in.read(reinterpret_cast<char *>(&n_clouds), 8);
  std::vector<Cloud::Ptr> raw;
  for (uint64_t i = 0; i < n_clouds; ++i) {
    uint64_t n = 0;
    in.read(reinterpret_cast<char *>(&n), 8);
    Cloud::Ptr c(new Cloud);
    c->points.resize(n);
    for (P &p : c->points) { in.read(reinterpret_cast<char *>(&p.x), 12); in.read(reinterpret_cast<char *>(&p.rgba), 4); }
    c->width = static_cast<uint32_t>(n); c->height = 1; c->is_dense = true;
    raw.push_back(c);
  }
  if (!in || raw.size() < 2) { std::printf("bad input\n"); return 2; }
  std::ofstream out(argv[2], std::ios::binary);

  Cloud::Ptr pts[2], kps[2];
  Fpfh::Ptr desc[2];
  for (int i = 0; i < 2; ++i) {
    Cloud::Ptr down(new Cloud);
    {
      pcl::VoxelGrid<P> f;
      f.setLeafSize(float(kResolution), float(kResolution), float(kResolution));
      f.setInputCloud(raw[i]);
      f.filter(*down);
    }
    pts[i].reset(new Cloud);
    {
      pcl::RadiusOutlierRemoval<P> f;
      f.setInputCloud(down);
      f.setRadiusSearch(kDescriptorRadius);            // the reference passes descriptor_radius (map_merging.cpp:219-220)
      f.setMinNeighborsInRadius(kMinNeighbours);
      f.filter(*pts[i]);
    }
    Normals::Ptr nrm(new Normals);
    {
      pcl::NormalEstimation<P, pcl::Normal> e;
      e.setRadiusSearch(kNormalRadius);
      e.setInputCloud(pts[i]);
      e.compute(*nrm);
    }
    kps[i].reset(new Cloud);
    {
      pcl::SIFTKeypoint<P, pcl::PointWithScale> d;
      d.setScales(float(kResolution), 3, 3);
      d.setMinimumContrast(float(kKeypointThreshold));
      d.setInputCloud(pts[i]);
      pcl::PointCloud<pcl::PointWithScale> tmp;
      d.compute(tmp);
      pcl::copyPointCloud(tmp, *kps[i]);
    }
    desc[i].reset(new Fpfh);
    {
      pcl::FPFHEstimation<P, pcl::Normal, pcl::FPFHSignature33> e;
      e.setRadiusSearch(kDescriptorRadius);
      e.setSearchSurface(pts[i]);
      e.setInputNormals(nrm);
      e.setInputCloud(kps[i]);
      e.compute(*desc[i]);
      // rows with a non-finite bin go, and their keypoints with them (features.cpp:118-143)
      pcl::DefaultPointRepresentation<pcl::FPFHSignature33> rep;
      pcl::IndicesPtr bad(new std::vector<int>);
      for (size_t j = 0; j < desc[i]->size(); ++j)
        if (!rep.isValid(desc[i]->points[j])) bad->push_back(static_cast<int>(j));
      pcl::ExtractIndices<pcl::FPFHSignature33> fd;
      fd.setInputCloud(desc[i]); fd.setIndices(bad); fd.setNegative(true); fd.filter(*desc[i]);
      pcl::ExtractIndices<P> fk;
      fk.setInputCloud(kps[i]); fk.setIndices(bad); fk.setNegative(true); fk.filter(*kps[i]);
    }
    put_cloud(out, *down); put_cloud(out, *pts[i]);
    put_u64(out, nrm->size());
    for (const pcl::Normal &q : nrm->points) { put(out, &q.normal_x, 12); put(out, &q.curvature, 4); }
    put_cloud(out, *kps[i]);
    put_u64(out, desc[i]->size()); put_u64(out, 132);
    for (const pcl::FPFHSignature33 &d : desc[i]->points) put(out, d.histogram, 132);
  }

  // reciprocal k-NN matching (matching.cpp:31-92)
  pcl::CorrespondencesPtr corr(new pcl::Correspondences);
  {
    pcl::search::KdTree<pcl::FPFHSignature33> to_target, to_source;
    to_target.setInputCloud(desc[1]); to_target.setSortedResults(true);
    to_source.setInputCloud(desc[0]); to_source.setSortedResults(true);
    std::vector<int> fwd(kMatchingK), back(kMatchingK);
    std::vector<float> fwd_d(kMatchingK), back_d(kMatchingK);
    for (size_t i = 0; i < desc[0]->size(); ++i) {
      to_target.nearestKSearch(*desc[0], int(i), int(kMatchingK), fwd, fwd_d);
      bool matched = false;
      for (size_t j = 0; j < fwd.size() && !matched; ++j) {
        to_source.nearestKSearch(*desc[1], fwd[j], int(kMatchingK), back, back_d);
        for (int b : back)
          if (b == int(i)) { corr->emplace_back(int(i), fwd[j], fwd_d[j]); matched = true; break; }
      }
    }
  }
  put_u64(out, corr->size());
  for (const auto &c : *corr) { put(out, &c.index_query, 4); put(out, &c.index_match, 4); put(out, &c.distance, 4); }

  // RANSAC on the matches, SVD on the inliers (matching.cpp:110-140)
  Eigen::Matrix4f T_ransac;
  pcl::Correspondences inliers;
  {
    pcl::registration::CorrespondenceRejectorSampleConsensus<P> r;
    r.setInputSource(kps[0]); r.setInputTarget(kps[1]);
    r.setInputCorrespondences(corr);
    r.setInlierThreshold(kInlierThreshold);
    r.getCorrespondences(inliers);
    if (r.getBestTransformation().isIdentity()) { T_ransac.setZero(); inliers.clear(); }
    else {
      pcl::registration::TransformationEstimationSVD<P, P> svd;
      svd.estimateRigidTransformation(*kps[0], *kps[1], inliers, T_ransac);
    }
  }
  put_T(out, T_ransac); put_u64(out, inliers.size());

  auto icp = [&](const Eigen::Matrix4f &guess) {
    pcl::IterativeClosestPoint<P, P> e;
    e.setMaxCorrespondenceDistance(kMaxCorrespondenceDistance);
    e.setRANSACOutlierRejectionThreshold(kInlierThreshold);
    e.setTransformationEpsilon(kTransformEpsilon);
    e.setMaximumIterations(kMaxIterations);
    Cloud::Ptr moved(new Cloud);
    pcl::transformPointCloud(*pts[0], *moved, guess);
    e.setInputSource(moved); e.setInputTarget(pts[1]);
    Cloud ignored;
    e.align(ignored);
    return Eigen::Matrix4f(e.getFinalTransformation() * guess);
  };
  const Eigen::Matrix4f T_icp = icp(T_ransac);
  put_T(out, T_icp);
  put_T(out, T_icp);                                   // estimateTransform(MATCHING, refine) is exactly the two steps above
  double score;
  {
    pcl::registration::TransformationValidationEuclidean<P, P> v;
    v.setMaxRange(kMaxCorrespondenceDistance);
    score = v.validateTransformation(pts[0], pts[1], T_icp);
  }
  put(out, &score, 8);

  // SAC-IA (matching.cpp:142-174; the process has drawn no rand() before: glibc seed 1)
  Eigen::Matrix4f T_sac;
  {
    pcl::SampleConsensusInitialAlignment<P, P, pcl::FPFHSignature33> e;
    e.setMinSampleDistance(float(kInlierThreshold));
    e.setMaxCorrespondenceDistance(kMaxCorrespondenceDistance);
    e.setMaximumIterations(kMaxIterations);
    e.setInputSource(kps[0]); e.setSourceFeatures(desc[0]);
    e.setInputTarget(kps[1]); e.setTargetFeatures(desc[1]);
    Cloud ignored;
    e.align(ignored);
    T_sac = e.getFinalTransformation();
  }
  put_T(out, T_sac);
  return out.good() ? 0 : 1;
}
