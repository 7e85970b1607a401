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

#include <cstdlib>
#include <limits>

template <typename PointSource, typename PointTarget>
class Gfx950Registration : public pcl::Registration<PointSource, PointTarget> {
  using Base = pcl::Registration<PointSource, PointTarget>;
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
    // force_no_recompute = true: "this tree will NEVER be recomputed, regardless of calls to setInputTarget".
    this->setSearchMethodTarget(this->tree_, /*force_no_recompute=*/true);
    this->reg_name_ = method == LSR_METHOD_NDT ? "Gfx950NDT" : "Gfx950GICP";
  }
  ~Gfx950Registration() override { lsr_destroy(h_); }
  Gfx950Registration(const Gfx950Registration&) = delete;
  Gfx950Registration& operator=(const Gfx950Registration&) = delete;

  void setInputTarget(const typename Base::PointCloudTargetConstPtr& cloud) override {
    Base::setInputTarget(cloud);                                        // keeps target_ for callers that read it
    report(lsr_set_input_target(h_, cloud->points.data(), sizeof(PointTarget), cloud->size()));
  }
  void setInputSource(const typename Base::PointCloudSourceConstPtr& cloud) override {
    Base::setInputSource(cloud);
    report(lsr_set_input_source(h_, cloud->points.data(), sizeof(PointSource), cloud->size()));
  }
  // NDT-only setters the nodes call (scanmatcher_component.cpp:107-111, graph_based_slam_component.cpp:66-71)
  void setResolution(float r) { report(lsr_set_f64(h_, LSR_RESOLUTION, r)); }
  void setNeighborhoodSearchMethod(int m) { report(lsr_set_i32(h_, LSR_NEIGHBORHOOD, m)); }   // LSR_DIRECT7 == pclomp::DIRECT7
  void setNumThreads(int n) { report(lsr_set_i32(h_, LSR_NUM_THREADS, n)); }                  // accepted, ignored

  // getFitnessScore is NOT virtual in PCL and would run a FLANN kd-tree search on the host: call it through this type
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
    this->converged_ = (st == LSR_OK) && r.converged;
    this->nr_iterations_ = (st == LSR_OK) ? r.iterations : 0;
    if (st == LSR_OK) this->final_transformation_ = this->transformation_ = T;       // else: previous pose stays
    else report(st);
  }
  void report(int st) const { if (st != LSR_OK) PCL_ERROR("[gfx950] %s: %s\n", lsr_status_string(st), lsr_last_error()); }
  lsr_handle h_ = nullptr;
  bool materialize_output_ = true;
};
// [end snippet]
