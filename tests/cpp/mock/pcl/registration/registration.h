// TEST INFRASTRUCTURE: a stand-in for <pcl/registration/registration.h> (PCL 1.12) with exactly the members of
// pcl::Registration / pcl::PointCloud / pcl::PointXYZI / Eigen::Matrix4f that include/lidarslam_reg/gfx950_registration.hpp
// and the INTEGRATION.md snippets touch, so that they are COMPILED in this repository (PCL itself is not in the image).
// Names, signatures and virtual-ness follow PCL: setInputSource / setInputTarget virtual, align non-virtual calling the
// pure virtual computeTransformation, getFitnessScore NON-virtual — and align() keeps PCL's initCompute() contract
// (registration.hpp: fails without target_; rebuilds the target kd-tree whenever a new target was set unless
// force_no_recompute_), with a build counter on the stand-in kd-tree so that a test can SEE whether a binding makes the host
// build a FLANN tree over the 661k-point submap it never searches.
#pragma once
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
#define PCL_WARN(...) std::fprintf(stderr, __VA_ARGS__)

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
  const PointT& operator[](std::size_t i) const { return points[i]; }
  PointT& operator[](std::size_t i) { return points[i]; }
  void push_back(const PointT& p) { points.push_back(p); }
  void clear() { points.clear(); }
  PointCloud& operator+=(const PointCloud& o) { points.insert(points.end(), o.points.begin(), o.points.end()); return *this; }
};

using Indices = std::vector<int>;   // pcl::Indices (PCL 1.12: std::vector<index_t>, index_t = int)

namespace search {
template <typename PointT>
struct KdTree {   // pcl::search::KdTree<PointT>: setInputCloud is the O(M log M) FLANN build, nearestKSearch virtual (pcl::search::Search)
  using Ptr = std::shared_ptr<KdTree<PointT>>;
  virtual ~KdTree() = default;
  static int& builds() { static int n = 0; return n; }
  static std::size_t& points_indexed() { static std::size_t n = 0; return n; }
  static std::size_t& unindexed_searches() { static std::size_t n = 0; return n; }
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& cloud) { builds()++; points_indexed() += cloud ? cloud->size() : 0; indexed_ = true; }
  // The stand-in indexes nothing, so it can only say what PCL would have done: a search on a tree that never saw a cloud
  // (force_no_recompute and a caller that stayed on the base pointer) is a null index in FLANN — counted here, outputs untouched.
  virtual int nearestKSearch(const PointT&, int, Indices&, std::vector<float>&) const {
    if (!indexed_) unindexed_searches()++;
    return 0;
  }
 private:
  bool indexed_ = false;
};
}  // namespace search

template <typename PointSource, typename PointTarget>
class Registration {
 public:
  using KdTree = pcl::search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  Registration() : tree_(new KdTree), final_transformation_(Eigen::Matrix4f::Identity()), transformation_(Eigen::Matrix4f::Identity()) {}
  virtual ~Registration() = default;
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) {
    if (!cloud || cloud->points.empty()) { PCL_ERROR("[pcl::%s::setInputTarget] Invalid or empty point cloud dataset given!\n", reg_name_.c_str()); return; }
    target_ = cloud;
    target_cloud_updated_ = true;
  }
  // registration.h: "force_no_recompute: if set to true, this tree will NEVER be recomputed, regardless of calls to setInputTarget"
  void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false) {
    tree_ = tree;
    if (force_no_recompute) force_no_recompute_ = true;
    target_cloud_updated_ = true;
  }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  void setEuclideanFitnessEpsilon(double e) { euclidean_fitness_epsilon_ = e; }
  void setRANSACIterations(int n) { ransac_iterations_ = n; }
  Eigen::Matrix4f getFinalTransformation() { return final_transformation_; }
  bool hasConverged() const { return converged_; }
  // NON-virtual in PCL (registration.hpp): transforms input_ by final_transformation_ and asks tree_ for the nearest target
  // point of every source point, one nearestKSearch per point; the mean of the squared distances <= max_range
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double fitness_score = 0.0;
    if (!input_) return std::numeric_limits<double>::max();
    const float* M = final_transformation_.data();
    Indices nn_indices(1);
    std::vector<float> nn_dists(1);
    int nr = 0;
    for (const PointSource& p : input_->points) {
      PointSource q = p;   // pcl::transformPointCloud: xyz only
      q.x = M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12];
      q.y = M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13];
      q.z = M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14];
      tree_->nearestKSearch(q, 1, nn_indices, nn_dists);
      if (nn_dists[0] <= max_range) { fitness_score += nn_dists[0]; nr++; }
    }
    return nr > 0 ? fitness_score / nr : std::numeric_limits<double>::max();
  }
  void align(PointCloudSource& output) { align(output, Eigen::Matrix4f::Identity()); }
  void align(PointCloudSource& output, const Eigen::Matrix4f& guess) {
    if (!initCompute()) return;
    if (input_) output.points = input_->points;      // PCL copies the source into `output` first
    converged_ = false;
    final_transformation_ = transformation_ = Eigen::Matrix4f::Identity();
    computeTransformation(output, guess);
  }

 protected:
  bool initCompute() {   // pcl::Registration::initCompute (registration.hpp)
    if (!target_) { PCL_ERROR("[pcl::registration::%s::compute] No input target dataset was given!\n", reg_name_.c_str()); return false; }
    if (target_cloud_updated_ && !force_no_recompute_) {   // "Only update target kd-tree if a new target cloud was set"
      tree_->setInputCloud(target_);
      target_cloud_updated_ = false;
    }
    return input_ != nullptr;
  }
  KdTreePtr tree_;
  bool target_cloud_updated_ = true, force_no_recompute_ = false;
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
