// TEST INFRASTRUCTURE: stand-in for <pclomp/gicp_omp.h> (pclomp::GeneralizedIterativeClosestPoint: PCL's GICP with OpenMP loops)
#pragma once
#include <memory>
#include <vector>
#include <pcl/mock_eigen_extra.h>
namespace pclomp {
template <typename PointSource, typename PointTarget>
class GeneralizedIterativeClosestPoint : public pcl::Registration<PointSource, PointTarget> {
 public:
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using MatricesVector = std::vector<Eigen::Matrix3d>;
 protected:
  void computeTransformation(PointCloudSource&, const Eigen::Matrix4f&) override {}
  std::shared_ptr<MatricesVector> input_covariances_, target_covariances_;
};
}  // namespace pclomp
