// test stand-in, see ../README.md
#pragma once
#include <memory>
#include <vector>
namespace pcl
{
struct Correspondence {
  int index_query = 0, index_match = -1;
  float distance = 0;
  Correspondence() = default;
  Correspondence(int q, int m, float d) : index_query(q), index_match(m), distance(d) {}
};
typedef std::vector<Correspondence> Correspondences;
typedef std::shared_ptr<Correspondences> CorrespondencesPtr;
}  // namespace pcl
