// Compiles the pcl::Registration-shaped adapter without PCL (stand-in point/cloud types with PCL's
// layout) and exercises the exact call sequence of ScanMatcherComponent
// (scanmatcher/src/scanmatcher_component.cpp:105-113,275,329,353,356,375).
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

#include "lidarslam_reg/registration.hpp"

struct alignas(16) PointXYZI {  // pcl::PointXYZI: 32 bytes
  float x, y, z, pad;
  float intensity, p1, p2, p3;
};
static_assert(sizeof(PointXYZI) == 32, "PCL layout");
struct Cloud {
  std::vector<PointXYZI> points;
};
using NDT = lidarslam_reg::NormalDistributionsTransform<PointXYZI, PointXYZI, Cloud, Cloud>;
using Reg = lidarslam_reg::Registration<PointXYZI, PointXYZI, Cloud, Cloud>;

int main() {
  std::shared_ptr<Reg> registration_;
  try {
    std::shared_ptr<NDT> ndt(new NDT());
    ndt->setResolution(5.0f);
    ndt->setTransformationEpsilon(0.01);
    ndt->setNeighborhoodSearchMethod(lidarslam_reg::DIRECT7);
    ndt->setNumThreads(2);
    registration_ = ndt;
  } catch (const std::exception& e) {
    std::printf("NO_DEVICE %s\n", e.what());
    return 0;  // expected on a CPU-only host: the core has no CPU path
  }
  auto tgt = std::make_shared<Cloud>();
  auto src = std::make_shared<Cloud>();
  for (int i = 0; i < 4000; i++) {
    float u = (i % 64) * 0.3f, v = (i / 64) * 0.3f;
    tgt->points.push_back({u, v, 0.02f * ((i * 7) % 5), 1.f, 0, 0, 0, 0});
    tgt->points.push_back({u, 0.05f * ((i * 3) % 7), v, 1.f, 0, 0, 0, 0});
    if (i % 3 == 0) src->points.push_back({u + 0.2f, v - 0.1f, 0.02f * ((i * 7) % 5), 1.f, (float)i, 0, 0, 0});
  }
  registration_->setInputTarget(tgt);
  registration_->setInputSource(src);
  Cloud output;
  registration_->align(output, lidarslam_reg::Matrix4f::Identity());
  auto T = registration_->getFinalTransformation();
  std::printf("OK converged=%d iters=%d t=(%.3f %.3f %.3f) fitness=%.4f out=%zu\n", (int)registration_->hasConverged(),
              registration_->getFinalNumIteration(), T(0, 3), T(1, 3), T(2, 3), registration_->getFitnessScore(), output.points.size());

  // `output` = the source with xyz moved by the final transformation, every other field kept (what PCL's align leaves)
  int fields_ok = 1;
  for (size_t i = 0; i < output.points.size(); i++) {
    const PointXYZI& a = src->points[i];
    const PointXYZI& b = output.points[i];
    const float ex = T(0, 0) * a.x + T(0, 1) * a.y + T(0, 2) * a.z + T(0, 3);
    const float ey = T(1, 0) * a.x + T(1, 1) * a.y + T(1, 2) * a.z + T(1, 3);
    const float ez = T(2, 0) * a.x + T(2, 1) * a.y + T(2, 2) * a.z + T(2, 3);
    const float err = std::fabs(b.x - ex) + std::fabs(b.y - ey) + std::fabs(b.z - ez);
    if (!(err < 1e-3f) || b.intensity != a.intensity || b.pad != a.pad) fields_ok = 0;
  }
  std::printf("OUTPUT fields_ok=%d\n", fields_ok);

  // A candidate set through the staged batch helpers: three objects, three shifted targets; every member must end where the
  // same registration ends on its own (graph_based_slam_component.cpp:181-231).
  {
    std::vector<std::shared_ptr<NDT>> own;
    std::vector<Reg*> regs;
    std::vector<std::shared_ptr<const Cloud>> targets;
    std::vector<lidarslam_reg::Matrix4f> guesses;
    float single_t[3][3];
    double single_fit[3];
    for (int c = 0; c < 3; c++) {
      auto t2 = std::make_shared<Cloud>();
      for (const auto& p : tgt->points) t2->points.push_back({p.x + 0.05f * c, p.y - 0.03f * c, p.z, 1.f, 0, 0, 0, 0});
      targets.push_back(t2);
      std::shared_ptr<NDT> n(new NDT());
      n->setResolution(5.0f); n->setTransformationEpsilon(0.01); n->setNeighborhoodSearchMethod(lidarslam_reg::DIRECT7);
      n->setInputTarget(t2); n->setInputSource(src);
      Cloud out1;
      n->align(out1, lidarslam_reg::Matrix4f::Identity());
      auto T1 = n->getFinalTransformation();
      for (int k = 0; k < 3; k++) single_t[c][k] = T1(k, 3);
      single_fit[c] = n->getFitnessScore();
      std::shared_ptr<NDT> m(new NDT());
      m->setResolution(5.0f); m->setTransformationEpsilon(0.01); m->setNeighborhoodSearchMethod(lidarslam_reg::DIRECT7);
      m->setInputSource(src);
      own.push_back(m);
      regs.push_back(m.get());
      guesses.push_back(lidarslam_reg::Matrix4f::Identity());
    }
    bool ok = Reg::setInputTargets(regs, targets) && Reg::alignBatch(regs, guesses);
    const std::vector<double> fits = Reg::getFitnessScores(regs);
    for (int c = 0; c < 3 && ok; c++) {
      auto Tb = regs[c]->getFinalTransformation();
      for (int k = 0; k < 3; k++)
        if (Tb(k, 3) != single_t[c][k]) ok = false;   // one input, one answer: the batch returns the single registration's bits
      if (!regs[c]->hasConverged() || !(std::fabs(fits[c] - single_fit[c]) <= 1e-9 * single_fit[c])) ok = false;
    }
    // the same set through the one-call form: same poses, same scores
    std::vector<double> fits2;
    ok = ok && Reg::alignAndScoreBatch(regs, guesses, fits2);
    for (int c = 0; c < 3 && ok; c++) {
      auto Tb = regs[c]->getFinalTransformation();
      for (int k = 0; k < 3; k++)
        if (Tb(k, 3) != single_t[c][k]) ok = false;
      if (!(std::fabs(fits2[c] - fits[c]) <= 1e-9 * fits[c])) ok = false;
    }
    std::printf("BATCH ok=%d\n", ok ? 1 : 0);
  }

  // searchLoop() through the C ABI: four "submaps" sharing the target cloud; the last one has
  // travelled far enough and sits next to the first, so exactly one candidate (id 0) is registered.
  std::vector<lsr_submap> sm(4);
  const double xs[4] = {0.0, 15.0, 30.0, 0.1};
  for (int i = 0; i < 4; i++) {
    sm[i] = lsr_submap{{xs[i], 0.0, 0.0}, {0.0, 0.0, 0.0, 1.0}, 15.0 * i, tgt->points.data(), tgt->points.size()};
  }
  sm[3].position[0] = 0.1;
  lsr_loop_params lp = {1.0, 20.0, 5.0, 0, 0.2f, 1, 0};
  lsr_loop_edge edge;
  int n_eval = -1;
  int st = lsr_search_loop(registration_->handle(), sm.data(), 4, sizeof(PointXYZI), 0, &lp,
                           &edge, 1, &n_eval);
  std::printf("LOOP st=%d n=%d from=%d to=%d accepted=%d\n", st, n_eval, edge.id_from, edge.id_to, edge.accepted);
  return 0;
}
