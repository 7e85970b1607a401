// lidarslam_reg/registration.hpp — header-only C++ adapter over the C ABI (lidarslam_reg.h) with the
// exact pcl::Registration surface lidarslam_ros2's two nodes exercise:
//
//   scanmatcher/src/scanmatcher_component.cpp:105-120   construction + setters
//   scanmatcher/src/scanmatcher_component.cpp:275,307,315,329,353,356,375,376
//   graph_based_slam/src/graph_based_slam_component.cpp:64-82,181,227,230,231,244
//
// A node keeps its `registration_` member and its call sites untouched; only the concrete type
// changes.  PCL is not required to compile this header: it is templated on the
// point type (anything with float x,y,z at offset 0, e.g. pcl::PointXYZI = 32-byte records) and on a
// cloud type exposing `points` (std::vector-like) — pcl::PointCloud<PointT> satisfies both.
// INTEGRATION.md shows the few lines that derive it from pcl::Registration inside the ROS2 package.
#pragma once

#include <cfloat>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../lidarslam_reg.h"

namespace lidarslam_reg {

enum NeighborSearchMethod { KDTREE = LSR_KDTREE, DIRECT26 = LSR_DIRECT26, DIRECT7 = LSR_DIRECT7, DIRECT1 = LSR_DIRECT1 };

// Column-major 4x4 float, layout-compatible with Eigen::Matrix4f.
struct Matrix4f {
  float m[16];
  static Matrix4f Identity() {
    Matrix4f r;
    std::memset(r.m, 0, sizeof(r.m));
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f;
    return r;
  }
  float& operator()(int row, int col) { return m[col * 4 + row]; }
  float operator()(int row, int col) const { return m[col * 4 + row]; }
  const float* data() const { return m; }
  float* data() { return m; }
};

template <typename PointSource, typename PointTarget, typename CloudSource, typename CloudTarget>
class Registration {
 public:
  using PointCloudSourceConstPtr = std::shared_ptr<const CloudSource>;
  using PointCloudTargetConstPtr = std::shared_ptr<const CloudTarget>;

  virtual ~Registration() {
    if (h_) lsr_destroy(h_);
  }
  Registration(const Registration&) = delete;
  Registration& operator=(const Registration&) = delete;

  // registration_->setInputTarget(cloud)   scanmatcher_component.cpp:275,307,315
  void setInputTarget(const PointCloudTargetConstPtr& cloud) {
    target_ = cloud;
    check(lsr_set_input_target(h_, cloud->points.data(), sizeof(PointTarget), cloud->points.size()), "setInputTarget");
  }
  // registration_->setInputSource(cloud)   scanmatcher_component.cpp:329
  void setInputSource(const PointCloudSourceConstPtr& cloud) {
    input_ = cloud;
    check(lsr_set_input_source(h_, cloud->points.data(), sizeof(PointSource), cloud->points.size()), "setInputSource");
  }
  // registration_->align(output, guess)    scanmatcher_component.cpp:353
  template <typename Mat4>
  void align(CloudSource& output, const Mat4& guess) {
    alignImpl(output, guess.data());
  }
  // registration_->align(output)           graph_based_slam_component.cpp:230
  void align(CloudSource& output) { alignImpl(output, nullptr); }

  Matrix4f getFinalTransformation() const { return final_; }                  // scanmatcher_component.cpp:356
  bool hasConverged() const { return last_.converged != 0; }                   // scanmatcher_component.cpp:375
  double getFitnessScore(double max_range = DBL_MAX) {                         // graph_based_slam_component.cpp:231
    double v = 0;
    check(lsr_get_fitness_score(h_, max_range, &v), "getFitnessScore");
    return v;
  }
  void setTransformationEpsilon(double e) { check(lsr_set_f64(h_, LSR_TRANSFORMATION_EPSILON, e), "setTransformationEpsilon"); }
  void setMaximumIterations(int n) { check(lsr_set_i32(h_, LSR_MAX_ITERATIONS, n), "setMaximumIterations"); }
  void setMaxCorrespondenceDistance(double d) { check(lsr_set_f64(h_, LSR_MAX_CORRESPONDENCE_DISTANCE, d), "setMaxCorrespondenceDistance"); }
  void setEuclideanFitnessEpsilon(double e) { check(lsr_set_f64(h_, LSR_EUCLIDEAN_FITNESS_EPSILON, e), "setEuclideanFitnessEpsilon"); }
  void setRANSACIterations(int n) { check(lsr_set_i32(h_, LSR_RANSAC_ITERATIONS, n), "setRANSACIterations"); }
  int getFinalNumIteration() const { return last_.iterations; }
  const lsr_result& lastResult() const { return last_; }
  lsr_handle handle() const { return h_; }

 protected:
  explicit Registration(int method, int device = 0) {
    // the reference exits the process on an invalid configuration (scanmatcher_component.cpp:121-124);
    // a library throws from the constructor instead — nothing ever throws across the C ABI itself.
    int st = lsr_create(method, device, nullptr, &h_);
    if (st != LSR_OK) throw std::runtime_error(std::string("lsr_create: ") + lsr_status_string(st) + ": " + lsr_last_error());
    final_ = Matrix4f::Identity();
    std::memset(&last_, 0, sizeof(last_));
  }
 public:
  // ---- candidate sets: the loop of GraphBasedSlamComponent::searchLoop (graph_based_slam_component.cpp:181-231: per candidate
  // setInputTarget, align, getFitnessScore) over a SET of registration objects, staged so that the candidates overlap on the
  // device (lsr_set_input_target_batch / lsr_align_batch / lsr_get_fitness_score_batch).  Results are read from each object
  // afterwards (getFinalTransformation, hasConverged, ...).  All objects on one device, each appearing once.
  static bool setInputTargets(const std::vector<Registration*>& regs, const std::vector<PointCloudTargetConstPtr>& clouds) {
    if (regs.size() != clouds.size()) return false;
    std::vector<lsr_handle> hs;
    std::vector<const void*> ptrs;
    std::vector<size_t> counts;
    for (size_t b = 0; b < regs.size(); b++) {
      regs[b]->target_ = clouds[b];
      hs.push_back(regs[b]->h_);
      ptrs.push_back(clouds[b]->points.data());
      counts.push_back(clouds[b]->points.size());
    }
    const int st = lsr_set_input_target_batch(hs.data(), (int)hs.size(), ptrs.data(), counts.data(), sizeof(PointTarget), 0);
    if (st != LSR_OK) std::fprintf(stderr, "[lidarslam_reg::setInputTargets] %s: %s\n", lsr_status_string(st), lsr_last_error());
    return st == LSR_OK;
  }
  static bool setInputSources(const std::vector<Registration*>& regs, const std::vector<PointCloudSourceConstPtr>& clouds) {
    if (regs.size() != clouds.size()) return false;
    std::vector<lsr_handle> hs;
    std::vector<const void*> ptrs;
    std::vector<size_t> counts;
    for (size_t b = 0; b < regs.size(); b++) {
      regs[b]->input_ = clouds[b];
      hs.push_back(regs[b]->h_);
      ptrs.push_back(clouds[b]->points.data());
      counts.push_back(clouds[b]->points.size());
    }
    const int st = lsr_set_input_source_batch(hs.data(), (int)hs.size(), ptrs.data(), counts.data(), sizeof(PointSource), 0);
    if (st != LSR_OK) std::fprintf(stderr, "[lidarslam_reg::setInputSources] %s: %s\n", lsr_status_string(st), lsr_last_error());
    return st == LSR_OK;
  }
  // one shared launch chain for all members (NDT; GICP members are registered one after the other); `output` clouds are not
  // materialised (both reference callers discard them)
  static bool alignBatch(const std::vector<Registration*>& regs, const std::vector<Matrix4f>& guesses) {
    if (regs.empty() || regs.size() != guesses.size()) return false;
    std::vector<lsr_handle> hs;
    std::vector<float> g(16 * regs.size()), f(16 * regs.size());
    std::vector<lsr_result> res(regs.size());
    for (size_t b = 0; b < regs.size(); b++) {
      hs.push_back(regs[b]->h_);
      std::memcpy(g.data() + 16 * b, guesses[b].m, sizeof(float) * 16);
    }
    const int st = lsr_align_batch(hs.data(), (int)hs.size(), g.data(), f.data(), res.data());
    if (st != LSR_OK) {
      std::fprintf(stderr, "[lidarslam_reg::alignBatch] %s: %s\n", lsr_status_string(st), lsr_last_error());
      for (auto* r : regs) r->last_.converged = 0;
      return false;
    }
    for (size_t b = 0; b < regs.size(); b++) {
      std::memcpy(regs[b]->final_.m, f.data() + 16 * b, sizeof(float) * 16);
      regs[b]->last_ = res[b];
    }
    return true;
  }
  // align() + getFitnessScore() of every member in one call (graph_based_slam_component.cpp:230-231): the searches of the members
  // that finish early run under the launch chain of the others; `scores` receives one value per member
  static bool alignAndScoreBatch(const std::vector<Registration*>& regs, const std::vector<Matrix4f>& guesses, std::vector<double>& scores,
                                 double max_range = DBL_MAX) {
    if (regs.empty() || regs.size() != guesses.size()) return false;
    std::vector<lsr_handle> hs;
    std::vector<float> g(16 * regs.size()), f(16 * regs.size());
    std::vector<lsr_result> res(regs.size());
    scores.assign(regs.size(), DBL_MAX);
    for (size_t b = 0; b < regs.size(); b++) {
      hs.push_back(regs[b]->h_);
      std::memcpy(g.data() + 16 * b, guesses[b].m, sizeof(float) * 16);
    }
    const int st = lsr_align_fitness_batch(hs.data(), (int)hs.size(), g.data(), f.data(), res.data(), max_range, scores.data());
    if (st != LSR_OK) {
      std::fprintf(stderr, "[lidarslam_reg::alignAndScoreBatch] %s: %s\n", lsr_status_string(st), lsr_last_error());
      for (auto* r : regs) r->last_.converged = 0;
      return false;
    }
    for (size_t b = 0; b < regs.size(); b++) {
      std::memcpy(regs[b]->final_.m, f.data() + 16 * b, sizeof(float) * 16);
      regs[b]->last_ = res[b];
    }
    return true;
  }
  static std::vector<double> getFitnessScores(const std::vector<Registration*>& regs, double max_range = DBL_MAX) {
    std::vector<lsr_handle> hs;
    for (auto* r : regs) hs.push_back(r->h_);
    std::vector<double> out(regs.size(), DBL_MAX);
    const int st = lsr_get_fitness_score_batch(hs.data(), (int)hs.size(), max_range, out.data());
    if (st != LSR_OK) std::fprintf(stderr, "[lidarslam_reg::getFitnessScores] %s: %s\n", lsr_status_string(st), lsr_last_error());
    return out;
  }

 protected:
  void check(int st, const char* where) const {
    // PCL logs errors and carries on; mirror that: report, leave the previous pose in place.
    if (st != LSR_OK) std::fprintf(stderr, "[lidarslam_reg::%s] %s: %s\n", where, lsr_status_string(st), lsr_last_error());
  }
  void alignImpl(CloudSource& output, const float* guess) {
    const size_t n = input_ ? input_->points.size() : 0;
    // PCL copies the source into `output` and then overwrites xyz with the transformed coordinates, every other field
    // (intensity, ...) stays: the C ABI writes exactly the xyz bytes of every record, so the copy below is all it takes.
    if (n) output.points.assign(input_->points.begin(), input_->points.end()); else output.points.clear();
    int st = lsr_align(h_, guess, final_.m, &last_, n ? (void*)output.points.data() : nullptr, sizeof(output.points[0]));
    if (st != LSR_OK) {
      check(st, "align");
      last_.converged = 0;
    }
  }
  lsr_handle h_ = nullptr;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  Matrix4f final_;
  lsr_result last_;
};

// pclomp::NormalDistributionsTransform<PointSource,PointTarget>   scanmatcher_component.cpp:105-113
template <typename PointSource, typename PointTarget, typename CloudSource, typename CloudTarget>
class NormalDistributionsTransform : public Registration<PointSource, PointTarget, CloudSource, CloudTarget> {
  using Base = Registration<PointSource, PointTarget, CloudSource, CloudTarget>;

 public:
  explicit NormalDistributionsTransform(int device = 0) : Base(LSR_METHOD_NDT, device) {}
  void setResolution(float r) { this->check(lsr_set_f64(this->h_, LSR_RESOLUTION, r), "setResolution"); }
  void setStepSize(double s) { this->check(lsr_set_f64(this->h_, LSR_STEP_SIZE, s), "setStepSize"); }
  void setOulierRatio(double r) { this->check(lsr_set_f64(this->h_, LSR_OUTLIER_RATIO, r), "setOulierRatio"); }
  void setNeighborhoodSearchMethod(NeighborSearchMethod m) { this->check(lsr_set_i32(this->h_, LSR_NEIGHBORHOOD, m), "setNeighborhoodSearchMethod"); }
  void setNumThreads(int n) { this->check(lsr_set_i32(this->h_, LSR_NUM_THREADS, n), "setNumThreads"); }  // accepted, ignored
  double getTransformationProbability() const { return this->last_.score; }
};

// pclomp::GeneralizedIterativeClosestPoint<PointSource,PointTarget>   scanmatcher_component.cpp:115-120
template <typename PointSource, typename PointTarget, typename CloudSource, typename CloudTarget>
class GeneralizedIterativeClosestPoint : public Registration<PointSource, PointTarget, CloudSource, CloudTarget> {
  using Base = Registration<PointSource, PointTarget, CloudSource, CloudTarget>;

 public:
  explicit GeneralizedIterativeClosestPoint(int device = 0) : Base(LSR_METHOD_GICP, device) {}
  void setRotationEpsilon(double e) { this->check(lsr_set_f64(this->h_, LSR_ROTATION_EPSILON, e), "setRotationEpsilon"); }
  void setCorrespondenceRandomness(int k) { this->check(lsr_set_i32(this->h_, LSR_K_CORRESPONDENCES, k), "setCorrespondenceRandomness"); }
  void setMaximumOptimizerIterations(int n) { this->check(lsr_set_i32(this->h_, LSR_MAX_INNER_ITERATIONS, n), "setMaximumOptimizerIterations"); }
};

}  // namespace lidarslam_reg
