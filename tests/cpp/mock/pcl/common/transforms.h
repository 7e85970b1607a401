// TEST INFRASTRUCTURE: stand-in for <pcl/common/transforms.h>
#pragma once
#include <pcl/mock_eigen_extra.h>
namespace pcl {
template <typename PointT> void transformPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Eigen::Matrix4f&) { out = in; }
}
