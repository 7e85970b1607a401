// TEST INFRASTRUCTURE: a stand-in for <pcl/registration/registration.h> (PCL 1.12) with exactly the members of
// pcl::Registration / pcl::PointCloud / pcl::PointXYZI / Eigen::Matrix4f that include/lidarslam_reg/gfx950_registration.hpp
// and the INTEGRATION.md snippets touch, so that they are COMPILED in this repository (PCL itself is not in the image).
// Names, signatures and virtual-ness follow PCL: setInputSource / setInputTarget virtual, align non-virtual calling the
// pure virtual computeTransformation, getFitnessScore NON-virtual.
#pragma once
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)

namespace Eigen {
struct Matrix4f {   // column-major 4x4 float, as Eigen::Matrix4f
  float m[16];
  static Matrix4f Identity() { Matrix4f r; std::memset(r.m, 0, sizeof(r.m)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
  float* data() { return m; }
  const float* data() const { return m; }
};
struct Matrix4d { double m[16]; const double* data() const { return m; } double* data() { return m; } };
template <typename M> struct Map { explicit Map(const double* p) { std::memcpy(v.m, p, sizeof(v.m)); } Matrix4d v; operator Matrix4d() const { return v; } };
struct Isometry3d { Isometry3d() = default; explicit Isometry3d(const Matrix4d& a) : mat(a) {} Matrix4d mat; };
}  // namespace Eigen

namespace pcl {
struct alignas(16) PointXYZI { float x, y, z, pad; float intensity, p1, p2, p3; };   // 32 bytes, SURVEY.md §9.9
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is a 32-byte record");

template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  std::size_t size() const { return points.size(); }
};

template <typename PointSource, typename PointTarget>
class Registration {
 public:
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  Registration() : final_transformation_(Eigen::Matrix4f::Identity()), transformation_(Eigen::Matrix4f::Identity()) {}
  virtual ~Registration() = default;
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  void setEuclideanFitnessEpsilon(double e) { euclidean_fitness_epsilon_ = e; }
  void setRANSACIterations(int n) { ransac_iterations_ = n; }
  Eigen::Matrix4f getFinalTransformation() { return final_transformation_; }
  bool hasConverged() const { return converged_; }
  double getFitnessScore(double = std::numeric_limits<double>::max()) { return -1.0; }   // NON-virtual in PCL (host FLANN search)
  void align(PointCloudSource& output) { align(output, Eigen::Matrix4f::Identity()); }
  void align(PointCloudSource& output, const Eigen::Matrix4f& guess) {
    if (input_) output.points = input_->points;      // PCL copies the source into `output` first
    converged_ = false;
    computeTransformation(output, guess);
  }

 protected:
  virtual void computeTransformation(PointCloudSource& output, const Eigen::Matrix4f& guess) = 0;
  std::string reg_name_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  int nr_iterations_ = 0, max_iterations_ = 10, ransac_iterations_ = 0;
  Eigen::Matrix4f final_transformation_, transformation_;
  double transformation_epsilon_ = 0.0, euclidean_fitness_epsilon_ = 0.0, corr_dist_threshold_ = 0.0;
  bool converged_ = false;
};
}  // namespace pcl
