// TEST INFRASTRUCTURE: stand-in for <pcl/io/pcd_io.h> (syntax check of oracle/ref_recipe/dump_fixtures.cpp)
#pragma once
#include <string>
#include <pcl/mock_eigen_extra.h>
namespace pcl { namespace io {
template <typename PointT> int loadPCDFile(const std::string&, pcl::PointCloud<PointT>&) { return 0; }
} }
