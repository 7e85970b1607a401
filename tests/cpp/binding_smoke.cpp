// TEST INFRASTRUCTURE: include/lidarslam_reg/gfx950_registration.hpp — the pcl::Registration subclass a maintainer adds —
// driven through the BASE-CLASS pointer the nodes hold (scanmatcher_component.h:93), against tests/cpp/mock/pcl, whose align()
// keeps PCL's initCompute() contract.  Prints what tests/test_host_cpu.py asserts: the registration converges, align() before
// setInputTarget() fails the PCL way, and the host never builds a kd-tree over the target (KDTREE builds=0).
#include <lidarslam_reg/gfx950_registration.hpp>

#include <cmath>
#include <cstdio>
#include <limits>
#include <memory>
#include <vector>

using Cloud = pcl::PointCloud<pcl::PointXYZI>;
using Reg = Gfx950Registration<pcl::PointXYZI, pcl::PointXYZI>;

int main() {
  int ndev = 0;
  if (lsr_device_count(&ndev) != LSR_OK || ndev <= 0) { std::printf("NO_DEVICE\n"); return 0; }
  auto ndt = std::make_shared<Reg>(LSR_METHOD_NDT, /*device=*/0, /*wait_mode=*/1);
  ndt->setResolution(5.0f); ndt->setTransformationEpsilon(0.01); ndt->setNeighborhoodSearchMethod(LSR_DIRECT7);
  std::shared_ptr<pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>> registration_ = ndt;   // what the node holds
  auto tgt = std::make_shared<Cloud>();
  auto src = std::make_shared<Cloud>();
  for (int i = 0; i < 4000; i++) {
    float u = (i % 64) * 0.3f, v = (i / 64) * 0.3f;
    tgt->points.push_back({u, v, 0.02f * ((i * 7) % 5), 1.f, 0, 0, 0, 0});
    tgt->points.push_back({u, 0.05f * ((i * 3) % 7), v, 1.f, 0, 0, 0, 0});
    if (i % 3 == 0) src->points.push_back({u + 0.2f, v - 0.1f, 0.02f * ((i * 7) % 5), 1.f, (float)i, 0, 0, 0});
  }
  Cloud output;
  registration_->setInputSource(src);
  registration_->align(output);                     // no target yet: PCL's initCompute() refuses, nothing reaches the device
  std::printf("NO_TARGET converged=%d\n", (int)registration_->hasConverged());
  for (int scan = 0; scan < 3; scan++) {            // the frontend's steady state: a new target every few scans (:307), align (:353)
    registration_->setInputTarget(tgt);
    registration_->setInputSource(src);
    registration_->align(output, Eigen::Matrix4f::Identity());
  }
  const Eigen::Matrix4f T = registration_->getFinalTransformation();
  std::printf("OK converged=%d t=(%.3f %.3f %.3f) fitness=%.4f\n", (int)registration_->hasConverged(), T.m[12], T.m[13], T.m[14],
              ndt->getFitnessScore());
  std::printf("KDTREE builds=%d points_indexed=%zu\n", pcl::search::KdTree<pcl::PointXYZI>::builds(),
              pcl::search::KdTree<pcl::PointXYZI>::points_indexed());
  // the call the reference makes (graph_based_slam_component.cpp:231): NON-virtual, on the base pointer.  PCL's own loop runs,
  // over the stand-in tree the binding installed: one device search, the same score as the derived call, no empty-index search.
  const double through_base = registration_->getFitnessScore();
  const double again = registration_->getFitnessScore();               // the walk restarts; no second device search is needed
  const double derived = ndt->getFitnessScore();
  std::printf("BASE_FITNESS base=%.9g again=%.9g derived=%.9g unindexed_searches=%zu\n", through_base, again, derived,
              pcl::search::KdTree<pcl::PointXYZI>::unindexed_searches());
  // ... and after a new align the cached search is stale: the next base-pointer call searches again at the new pose
  Eigen::Matrix4f G = Eigen::Matrix4f::Identity(); G.m[12] = 0.05f;
  registration_->align(output, G);
  std::printf("BASE_FITNESS_AFTER_ALIGN base=%.9g derived=%.9g\n", registration_->getFitnessScore(), ndt->getFitnessScore());
  // a search that is NOT that walk is refused: 0 neighbours, distance FLT_MAX
  {
    pcl::Indices idx(1); std::vector<float> d2(1);
    pcl::PointXYZI far{1.0e6f, 1.0e6f, 1.0e6f, 1.f, 0, 0, 0, 0};
    struct Peek : pcl::Registration<pcl::PointXYZI, pcl::PointXYZI> { using pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>::tree_; };
    const int found = (registration_.get()->*(&Peek::tree_))->nearestKSearch(far, 1, idx, d2);
    std::printf("FOREIGN_SEARCH found=%d d2_is_max=%d\n", found, (int)(d2[0] == std::numeric_limits<float>::max()));
  }
  return 0;
}
