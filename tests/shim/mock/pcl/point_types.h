// test stand-in, see ../README.md: the memory layout of pcl::PointXYZRGB / pcl::Normal (32 bytes each)
#pragma once
#include <cstdint>
#include <Eigen/Core>
namespace pcl
{
struct alignas(16) PointXYZRGB {
  float x, y, z, pad_;
  union {
    struct { uint8_t b, g, r, a; };
    float rgb;
    uint32_t rgba;
  };
  uint32_t pad2_[3];
};
struct alignas(16) Normal {
  float normal_x, normal_y, normal_z, pad_;
  float curvature;
  float pad2_[3];
};
static_assert(sizeof(PointXYZRGB) == 32 && sizeof(Normal) == 32, "PCL point layouts");
}  // namespace pcl
