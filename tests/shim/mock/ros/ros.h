// test stand-in, see ../README.md: ros::NodeHandle::getParam as MapMergingParams::fromROSNode uses it
#pragma once
#include <map>
#include <string>
namespace ros
{
class NodeHandle
{
public:
  // the test fills `values` ("name" -> text) in place of a parameter server
  std::map<std::string, std::string> values;
  bool getParam(const std::string &key, double &v) const { auto i = values.find(key); if (i == values.end()) return false; v = std::stod(i->second); return true; }
  bool getParam(const std::string &key, int &v) const { auto i = values.find(key); if (i == values.end()) return false; v = std::stoi(i->second); return true; }
  bool getParam(const std::string &key, bool &v) const { auto i = values.find(key); if (i == values.end()) return false; v = i->second == "true" || i->second == "1"; return true; }
  bool getParam(const std::string &key, std::string &v) const { auto i = values.find(key); if (i == values.end()) return false; v = i->second; return true; }
};
}  // namespace ros
