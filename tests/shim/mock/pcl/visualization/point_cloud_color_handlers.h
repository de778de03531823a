// test stand-in, see ../../README.md (typedefs.h names the type; nothing on the path uses it)
#pragma once
namespace pcl { namespace visualization { template <typename PointT> class PointCloudColorHandlerCustom; } }
