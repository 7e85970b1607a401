// TEST INFRASTRUCTURE: stand-in for <pcl/kdtree/kdtree_flann.h>
#pragma once
#include <vector>
#include <pcl/mock_eigen_extra.h>
namespace pcl {
template <typename PointT> class KdTreeFLANN {
 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr&) {}
  int nearestKSearch(const PointT&, int, std::vector<int>&, std::vector<float>&) const { return 0; }
};
}
