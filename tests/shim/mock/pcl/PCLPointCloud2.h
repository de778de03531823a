// test stand-in, see ../README.md
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
namespace pcl
{
struct PCLPointField {
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;
  uint32_t count = 0;
  enum PointFieldTypes { INT8 = 1, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 };
};
struct PCLPointCloud2 {
  typedef std::shared_ptr<PCLPointCloud2> Ptr;
  typedef std::shared_ptr<const PCLPointCloud2> ConstPtr;
  uint32_t height = 0, width = 0;
  std::vector<PCLPointField> fields;
  uint8_t is_bigendian = 0;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  uint8_t is_dense = 0;
};
}  // namespace pcl
