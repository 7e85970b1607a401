// gfx950_registration.hpp — the binding a lidarslam_ros2 maintainer adds next to scanmatcher_component.h /
// graph_based_slam_component.h (INTEGRATION.md §2 quotes this file): a pcl::Registration-derived class that forwards to
// the C ABI (lidarslam_reg.h), so that it fits the nodes' existing member
//     boost::shared_ptr<pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>> registration_;
// (scanmatcher/include/scanmatcher/scanmatcher_component.h:93, graph_based_slam/include/graph_based_slam/graph_based_slam_component.h:106).
// Needs PCL (<pcl/registration/registration.h>); this repository compiles it against tests/cpp/mock/pcl (a 60-line stand-in
// with the members pcl::Registration really has) in tests/test_host_cpu.py, so that it cannot drift from lidarslam_reg.h.
#pragma once
// [snippet: binding]
#include <pcl/registration/registration.h>
#include <lidarslam_reg.h>

#include <cmath>
#include <cstdlib>
#include <limits>
#include <memory>
#include <vector>

template <typename PointSource, typename PointTarget> class Gfx950Registration;

// What stands in pcl::Registration::tree_.  getFitnessScore is NOT virtual in PCL: a call that stays on the base pointer
// (graph_based_slam_component.cpp:231, scanmatcher_component.cpp:376 as the reference writes them) runs pcl::Registration's own
// loop — transform input_, one tree_->nearestKSearch per point — over a kd-tree that force_no_recompute (constructor) never
// builds.  This tree answers that loop from ONE device search (lsr_nearest_neighbors under final_transformation_): the un-edited
// call returns the device's score instead of dereferencing an empty FLANN index.  Any OTHER search through tree_ is refused loudly
// (0 neighbours, distance FLT_MAX, so that a fitness computed from it can never pass a loop-closure gate).
template <typename PointSource, typename PointTarget>
class Gfx950FitnessTree : public pcl::search::KdTree<PointTarget> {
 public:
  explicit Gfx950FitnessTree(Gfx950Registration<PointSource, PointTarget>* owner) : owner_(owner) {}
  int nearestKSearch(const PointTarget& point, int k, pcl::Indices& k_indices, std::vector<float>& k_sqr_distances) const override {
    return owner_->serveBaseClassSearch(point, k, k_indices, k_sqr_distances);
  }
 private:
  Gfx950Registration<PointSource, PointTarget>* owner_;
};

template <typename PointSource, typename PointTarget>
class Gfx950Registration : public pcl::Registration<PointSource, PointTarget> {
  using Base = pcl::Registration<PointSource, PointTarget>;
  friend class Gfx950FitnessTree<PointSource, PointTarget>;
 public:
  // wait_mode: how the calling thread waits for the device (LSR_WAIT_MODE: 0 spin, 1 yield, 2 sleep).  Default yield: frontend
  // and backend run side by side under a MultiThreadedExecutor (lidarslam/src/lidarslam.cpp:12-17) and a spinning wait pins one
  // core per running align (measured cost of yielding between polls: none); a node that owns its cores passes 0.
  explicit Gfx950Registration(lsr_method method, int device = 0, int wait_mode = 1) {
    if (lsr_create(method, device, nullptr, &h_) != LSR_OK) {          // same failure mode as an invalid
      PCL_ERROR("[gfx950] %s\n", lsr_last_error()); std::exit(1);      // registration_method: exit(1)
    }                                                                   // (scanmatcher_component.cpp:121-124)
    lsr_set_i32(h_, LSR_WAIT_MODE, wait_mode);
    // pcl::Registration::align() -> initCompute() rebuilds a FLANN kd-tree over target_ after every setInputTarget
    // (scanmatcher_component.cpp:307,353; graph_based_slam_component.cpp:227,230) — >= 100 ms of host time for the
    // 661k-point submap, next to a 0.1 ms voxel-grid build on the device, for a tree nothing here searches.
    // force_no_recompute = true: "this tree will NEVER be recomputed, regardless of calls to setInputTarget".  The tree installed
    // is the stand-in above, so that a getFitnessScore() left on the base pointer is served by the device, not by an empty index.
    this->setSearchMethodTarget(std::make_shared<Gfx950FitnessTree<PointSource, PointTarget>>(this), /*force_no_recompute=*/true);
    this->reg_name_ = method == LSR_METHOD_NDT ? "Gfx950NDT" : "Gfx950GICP";
  }
  ~Gfx950Registration() override { lsr_destroy(h_); }
  Gfx950Registration(const Gfx950Registration&) = delete;
  Gfx950Registration& operator=(const Gfx950Registration&) = delete;

  void setInputTarget(const typename Base::PointCloudTargetConstPtr& cloud) override {
    Base::setInputTarget(cloud);                                        // keeps target_ for callers that read it
    epoch_++;
    report(lsr_set_input_target(h_, cloud->points.data(), sizeof(PointTarget), cloud->size()));
  }
  void setInputSource(const typename Base::PointCloudSourceConstPtr& cloud) override {
    Base::setInputSource(cloud);
    epoch_++;
    report(lsr_set_input_source(h_, cloud->points.data(), sizeof(PointSource), cloud->size()));
  }
  // NDT-only setters the nodes call (scanmatcher_component.cpp:107-111, graph_based_slam_component.cpp:66-71)
  void setResolution(float r) { report(lsr_set_f64(h_, LSR_RESOLUTION, r)); }
  void setNeighborhoodSearchMethod(int m) { report(lsr_set_i32(h_, LSR_NEIGHBORHOOD, m)); }   // LSR_DIRECT7 == pclomp::DIRECT7 (all four pclomp methods are served, KDTREE included)
  void setNumThreads(int n) { report(lsr_set_i32(h_, LSR_NUM_THREADS, n)); }                  // accepted, ignored

  // getFitnessScore is NOT virtual in PCL: call it through this type (one device launch chain, the score comes back through a
  // mailbox).  The same call through pcl::Registration* still returns the right number (Gfx950FitnessTree), at the price of one
  // device search + PCL's per-point host loop + a warning.
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double v = std::numeric_limits<double>::max();
    report(lsr_get_fitness_score(h_, max_range, &v));
    return v;
  }
  lsr_handle handle() const { return h_; }      // for the entry points that have no pcl::Registration counterpart (§3b-3d)
  // align(output, ...) fills `output` with the transformed source like PCL does (12 bytes per point cross PCIe).  Both
  // reference callers discard it (scanmatcher_component.cpp:350-353, graph_based_slam_component.cpp:229-230): they switch
  // it off and `output` keeps the plain copy of the source pcl::Registration::align made.
  void setMaterializeOutput(bool on) { materialize_output_ = on; }

 protected:
  // pcl::Registration::align() -> computeTransformation(): the single virtual the base class needs.
  void computeTransformation(typename Base::PointCloudSource& output, const Eigen::Matrix4f& guess) override {
    // base-class setters land in members: forward them right before the run
    lsr_set_f64(h_, LSR_TRANSFORMATION_EPSILON, this->transformation_epsilon_);       // :108,119 / gbs :68,79
    lsr_set_i32(h_, LSR_MAX_ITERATIONS, this->max_iterations_);                       // gbs :66,77
    lsr_set_f64(h_, LSR_MAX_CORRESPONDENCE_DISTANCE, this->corr_dist_threshold_);     // :118 / gbs :76
    lsr_result r;
    Eigen::Matrix4f T;                                                  // column-major fp32 == the ABI layout
    // `output` already holds a copy of the source (pcl::Registration::align made it).  lsr_align writes ONLY the 12 xyz
    // bytes of every record and leaves the other fields (intensity, padding) as the copy left them — exactly what
    // pcl::transformPointCloud does to `output`.
    int st = materialize_output_ ? lsr_align(h_, guess.data(), T.data(), &r, output.points.data(), sizeof(PointSource))
                                 : lsr_align(h_, guess.data(), T.data(), &r, nullptr, 0);
    epoch_++;
    this->converged_ = (st == LSR_OK) && r.converged;
    this->nr_iterations_ = (st == LSR_OK) ? r.iterations : 0;
    if (st == LSR_OK) this->final_transformation_ = this->transformation_ = T;       // else: previous pose stays
    else report(st);
  }
  void report(int st) const { if (st != LSR_OK) PCL_ERROR("[gfx950] %s: %s\n", lsr_status_string(st), lsr_last_error()); }

  // tree_->nearestKSearch as pcl::Registration::getFitnessScore issues it: k = 1, query number i = final_transformation_ * input_[i],
  // in order.  The first query after anything changed (epoch_) runs ONE device search for every source point; the queries are
  // then answered from its result — after checking that the query really is the point the walk has reached.
  int serveBaseClassSearch(const PointTarget& q, int k, pcl::Indices& k_indices, std::vector<float>& k_sqr_distances) {
    k_indices.assign((std::size_t)(k > 0 ? k : 1), -1);
    k_sqr_distances.assign((std::size_t)(k > 0 ? k : 1), std::numeric_limits<float>::max());
    const std::size_t n = this->input_ ? this->input_->size() : 0;
    if (k == 1 && n > 0) {
      if (nn_epoch_ != epoch_) {
        nn_idx_.resize(n); nn_d2_.resize(n);
        const int st = lsr_nearest_neighbors(h_, this->final_transformation_.data(), nn_idx_.data(), nn_d2_.data());
        if (st == LSR_OK) {
          nn_epoch_ = epoch_; cursor_ = 0;
          if (!warned_) {
            warned_ = true;
            PCL_WARN("[gfx950] getFitnessScore called through pcl::Registration*: served by one device search + PCL's host loop; "
                     "call it on the derived pointer (INTEGRATION.md 2)\n");
          }
        } else {
          report(st);
        }
      }
      if (nn_epoch_ == epoch_ && nn_idx_.size() == n) {
        const float* M = this->final_transformation_.data();   // column-major
        for (int attempt = 0; attempt < 2; attempt++) {          // the walk restarts at 0 when getFitnessScore is called again
          const std::size_t i = attempt == 0 ? cursor_ : 0;
          const auto& p = this->input_->points[i];
          const float ex = M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], ey = M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
                      ez = M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14];
          const float tol = 1e-4f * (1.f + std::fabs(ex) + std::fabs(ey) + std::fabs(ez));
          if (std::fabs(q.x - ex) <= tol && std::fabs(q.y - ey) <= tol && std::fabs(q.z - ez) <= tol) {
            k_indices[0] = nn_idx_[i];
            k_sqr_distances[0] = nn_d2_[i];
            cursor_ = (i + 1) % n;
            return nn_idx_[i] >= 0 ? 1 : 0;
          }
        }
      }
    }
    if (!refused_) {
      refused_ = true;
      PCL_ERROR("[gfx950] a search through pcl::Registration::tree_ that is not getFitnessScore's walk over the registered source: "
                "this object keeps no host kd-tree (force_no_recompute); 0 neighbours returned\n");
    }
    return 0;
  }

  lsr_handle h_ = nullptr;
  bool materialize_output_ = true;
  unsigned long epoch_ = 1, nn_epoch_ = 0;       // bumped by setInputTarget / setInputSource / align; epoch of the cached search
  std::vector<int32_t> nn_idx_;
  std::vector<float> nn_d2_;
  std::size_t cursor_ = 0;
  bool warned_ = false, refused_ = false;
};
// [end snippet]
