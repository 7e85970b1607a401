// TEST INFRASTRUCTURE: stand-in for <pclomp/ndt_omp.h> of rsasaki0109/ndt_omp_ros2 (fork of koide3/ndt_omp) with the members
// oracle/ref_recipe/dump_fixtures.cpp uses — public setters, the protected derivative pass and voxel grid it opens up in a subclass.
#pragma once
#include <map>
#include <pcl/filters/voxel_grid.h>
namespace pclomp {
enum NeighborSearchMethod { KDTREE, DIRECT26, DIRECT7, DIRECT1 };
template <typename PointT> class VoxelGridCovariance : public pcl::VoxelGrid<PointT> {
 public:
  struct FloatVec { float v[4] = {0, 0, 0, 0}; float operator[](int k) const { return v[k]; } };   // Eigen::VectorXf centroid
  struct Leaf { int nr_points = 0; Eigen::Matrix3d cov_, icov_; FloatVec centroid; };
  const std::map<std::size_t, Leaf>& getLeaves() const { return leaves_; }
 protected:
  std::map<std::size_t, Leaf> leaves_;
};
template <typename PointSource, typename PointTarget>
class NormalDistributionsTransform : public pcl::Registration<PointSource, PointTarget> {
 public:
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using TargetGrid = VoxelGridCovariance<PointTarget>;
  void setNumThreads(int) {}
  void setResolution(float) {}
  void setStepSize(double) {}
  void setOulierRatio(double) {}
  void setNeighborhoodSearchMethod(NeighborSearchMethod) {}
  int getFinalNumIteration() const { return this->nr_iterations_; }
 protected:
  void computeTransformation(PointCloudSource&, const Eigen::Matrix4f&) override {}
  double computeDerivatives(Eigen::Matrix<double, 6, 1>&, Eigen::Matrix<double, 6, 6>&, PointCloudSource&, Eigen::Matrix<double, 6, 1>&, bool = true) { return 0; }
  void computeAngleDerivatives(Eigen::Matrix<double, 6, 1>&, bool = true) {}
  TargetGrid target_cells_;
  double gauss_d1_ = 0, gauss_d2_ = 0;
};
}  // namespace pclomp
