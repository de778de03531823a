// test stand-in, see ../README.md
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace pcl
{
template <typename PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  uint32_t width = 0, height = 0;
  bool is_dense = true;
  size_t size() const { return points.size(); }
};
}  // namespace pcl
