// TEST INFRASTRUCTURE: stand-in for <pcl/filters/voxel_grid.h>
#pragma once
#include <pcl/mock_eigen_extra.h>
namespace pcl {
template <typename PointT> class VoxelGrid {
 public:
  void setLeafSize(float, float, float) {}
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr&) {}
  void filter(PointCloud<PointT>&) {}
  Eigen::Vector3i getMinBoxCoordinates() const { return Eigen::Vector3i(); }
  Eigen::Vector3i getMaxBoxCoordinates() const { return Eigen::Vector3i(); }
};
}
