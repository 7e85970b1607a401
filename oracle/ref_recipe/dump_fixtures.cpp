// TEST INFRASTRUCTURE (oracle/ref_recipe/README.md) — runs the REFERENCE's own registration code (pclomp, from a checkout of
// rsasaki0109/ndt_omp_ros2: the submodule /root/reference/.gitmodules:1-4 pins) on the inputs export_inputs.py wrote and dumps, as
// JSON, every quantity the committed golden fixtures hold.  import_results.py turns the JSON into tests/golden/ref_*.npz.
//
//   dump_fixtures <inputs dir> <output dir>
//
// The objects are configured the way the nodes configure them (scanmatcher/src/scanmatcher_component.cpp:105-125,
// graph_based_slam/src/graph_based_slam_component.cpp:63-82): DIRECT7, step size 0.1 (PCL default), outlier ratio 0.55 (default),
// one thread count for all (results of pclomp do not depend on it: per-thread partial sums are added in index order).
// Not part of the product; compiled here only against stand-in headers (tests/cpp/mock, -fsyntax-only).
#include <pcl/io/pcd_io.h>
#include <pcl/kdtree/kdtree_flann.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/common/transforms.h>
#include <pclomp/ndt_omp.h>
#include <pclomp/gicp_omp.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

using Point = pcl::PointXYZI;
using Cloud = pcl::PointCloud<Point>;

// ---- a reader for the one JSON shape export_inputs.py writes: "key": number | "string" | [numbers]  (no dependency on a JSON library)
struct Json {
  std::string text;
  explicit Json(const std::string& path) { std::ifstream f(path); std::stringstream ss; ss << f.rdbuf(); text = ss.str(); }
  size_t find_key(const std::string& key, size_t from) const { return text.find("\"" + key + "\"", from); }
  double number(const std::string& key, size_t from = 0) const { size_t p = text.find(':', find_key(key, from)); return std::strtod(text.c_str() + p + 1, nullptr); }
  std::string str(const std::string& key, size_t from = 0) const {
    size_t p = text.find('"', text.find(':', find_key(key, from)) + 1), q = text.find('"', p + 1);
    return text.substr(p + 1, q - p - 1);
  }
  std::vector<double> array(const std::string& key, size_t from = 0) const {
    std::vector<double> v;
    size_t p = text.find('[', find_key(key, from)), q = text.find(']', p);
    const char* c = text.c_str() + p + 1;
    while (c < text.c_str() + q) { char* e; double d = std::strtod(c, &e); if (e == c) break; v.push_back(d); c = e; while (*c == ',' || *c == ' ') c++; }
    return v;
  }
};

static Eigen::Matrix4f to_matrix(const std::vector<double>& colmajor) {
  Eigen::Matrix4f M = Eigen::Matrix4f::Identity();
  for (int k = 0; k < 16; k++) M.data()[k] = (float)colmajor[k];
  return M;
}
template <typename T> static void put(FILE* f, const char* key, const T* v, size_t n, bool last = false) {
  std::fprintf(f, "\"%s\": [", key);
  for (size_t k = 0; k < n; k++) std::fprintf(f, "%s%.17g", k ? ", " : "", (double)v[k]);
  std::fprintf(f, "]%s\n", last ? "" : ",");
}

// pclomp::NormalDistributionsTransform with its protected derivative pass and voxel grid opened up
struct NdtProbe : pclomp::NormalDistributionsTransform<Point, Point> {
  using Base = pclomp::NormalDistributionsTransform<Point, Point>;
  double derivatives(const Eigen::Matrix<double, 6, 1>& p_in, Eigen::Matrix<double, 6, 1>& grad, Eigen::Matrix<double, 6, 6>& hess) {
    // computeTransformation() has initialised gauss_d1_ / gauss_d2_ (call align() once before this); the first pass of
    // computeTransformation at pose p: transform the source, angle derivatives, computeDerivatives
    Eigen::Matrix<double, 6, 1> p = p_in;
    Eigen::Translation<float, 3> t((float)p(0), (float)p(1), (float)p(2));
    Eigen::Matrix4f T = (t * Eigen::AngleAxis<float>((float)p(3), Eigen::Vector3f::UnitX()) * Eigen::AngleAxis<float>((float)p(4), Eigen::Vector3f::UnitY()) *
                         Eigen::AngleAxis<float>((float)p(5), Eigen::Vector3f::UnitZ())).matrix();
    Cloud moved;
    pcl::transformPointCloud(*this->input_, moved, T);
    this->computeAngleDerivatives(p, true);
    return this->computeDerivatives(grad, hess, moved, p, true);
  }
  const Base::TargetGrid& cells() const { return this->target_cells_; }
};
struct GicpProbe : pclomp::GeneralizedIterativeClosestPoint<Point, Point> {
  int iterations() const { return this->nr_iterations_; }   // pcl::Registration keeps the count protected
  const Eigen::Matrix3d& source_cov(size_t i) const { return (*this->input_covariances_)[i]; }
  const Eigen::Matrix3d& target_cov(size_t i) const { return (*this->target_covariances_)[i]; }
};

static Cloud::Ptr load(const std::string& dir, const std::string& name) {
  Cloud::Ptr c(new Cloud);
  if (pcl::io::loadPCDFile<Point>(dir + "/" + name, *c) != 0) { std::fprintf(stderr, "cannot read %s/%s\n", dir.c_str(), name.c_str()); std::exit(2); }
  return c;
}
static void configure(NdtProbe& ndt, double res, double eps, int max_iter) {
  ndt.setResolution((float)res);                              // scanmatcher_component.cpp:107
  ndt.setTransformationEpsilon(eps);                          // :108
  ndt.setMaximumIterations(max_iter);
  ndt.setNeighborhoodSearchMethod(pclomp::DIRECT7);           // :110
  ndt.setNumThreads(1);
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: dump_fixtures <inputs dir> <output dir>\n"); return 1; }
  const std::string in = argv[1], out = argv[2];
  std::system(("mkdir -p " + out).c_str());
  Json J(in + "/cases.json");
  FILE* f = std::fopen((out + "/results.json").c_str(), "w");
  if (!f) return 3;
  std::fprintf(f, "{\n");
  {  // ---- ndt_small (tests/golden/make_golden.py)
    const size_t at = J.find_key("ndt_small", 0);
    Cloud::Ptr tgt = load(in, J.str("target", at)), src = load(in, J.str("source", at));
    const double res = J.number("resolution", at);
    const Eigen::Matrix4f guess = to_matrix(J.array("guess_colmajor", at));
    const std::vector<double> pv = J.array("p", at);
    NdtProbe ndt;
    configure(ndt, res, 0.01, 35);
    ndt.setInputTarget(tgt);
    ndt.setInputSource(src);
    Cloud aligned;
    ndt.align(aligned, guess);
    std::fprintf(f, "\"ndt_small\": {\n");
    put(f, "final_eps001", ndt.getFinalTransformation().data(), 16);
    std::fprintf(f, "\"iters_eps001\": %d,\n", ndt.getFinalNumIteration());
    Eigen::Matrix<double, 6, 1> p, g;
    Eigen::Matrix<double, 6, 6> H;
    for (int k = 0; k < 6; k++) p(k) = pv[k];
    const double score = ndt.derivatives(p, g, H);
    std::fprintf(f, "\"score\": %.17g,\n", score);
    put(f, "grad", g.data(), 6);
    double Hrow[36];
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) Hrow[r * 6 + c] = H(r, c);
    put(f, "hess", Hrow, 36);
    std::vector<double> idx, cnt;
    for (const auto& kv : ndt.cells().getLeaves()) { idx.push_back((double)kv.first); cnt.push_back((double)kv.second.nr_points); }   // std::map: ascending leaf index
    put(f, "leaf_idx", idx.data(), idx.size());
    put(f, "leaf_n", cnt.data(), cnt.size());
    const Eigen::Vector3i mn = ndt.cells().getMinBoxCoordinates(), mx = ndt.cells().getMaxBoxCoordinates();
    const double mnv[3] = {(double)mn(0), (double)mn(1), (double)mn(2)}, mxv[3] = {(double)mx(0), (double)mx(1), (double)mx(2)};
    put(f, "min_b", mnv, 3);
    put(f, "max_b", mxv, 3);
    configure(ndt, res, 1e-6, 30);
    ndt.align(aligned, guess);
    put(f, "final_tight", ndt.getFinalTransformation().data(), 16);
    std::fprintf(f, "\"iters_tight\": %d,\n", ndt.getFinalNumIteration());
    // the fourth pclomp neighbourhood: radiusSearch(x', resolution_) on the kd-tree over the leaves' float centroids
    configure(ndt, res, 0.01, 35);
    ndt.setNeighborhoodSearchMethod(pclomp::KDTREE);
    ndt.align(aligned, guess);
    put(f, "final_kdtree", ndt.getFinalTransformation().data(), 16);
    std::fprintf(f, "\"iters_kdtree\": %d,\n", ndt.getFinalNumIteration());
    const double score_kd = ndt.derivatives(p, g, H);
    std::fprintf(f, "\"score_kdtree\": %.17g,\n", score_kd);
    put(f, "grad_kdtree", g.data(), 6);
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) Hrow[r * 6 + c] = H(r, c);
    put(f, "hess_kdtree", Hrow, 36);
    std::vector<double> cen;   // Leaf::centroid (Eigen::VectorXf) of every leaf, ascending leaf index; the leaves below min_points_per_voxel are not in the kd-tree
    for (const auto& kv : ndt.cells().getLeaves())
      for (int k = 0; k < 3; k++) cen.push_back(kv.second.nr_points >= 6 ? (double)kv.second.centroid[k] : 1e300);   // (1e300: not in the tree; JSON has no NaN)
    put(f, "leaf_centroid", cen.data(), cen.size());
    std::fprintf(f, "\"kdtree\": 1\n},\n");
  }
  {  // ---- gicp_small (tests/golden/make_golden_gicp.py): target re-filtered like the GICP frontend (scanmatcher_component.cpp:309-315)
    const size_t at = J.find_key("gicp_small", 0);
    Cloud::Ptr raw = load(in, J.str("target_raw", at)), src = load(in, J.str("source", at)), tgt(new Cloud);
    pcl::VoxelGrid<Point> vg;
    const float leaf = (float)J.number("leaf", at);
    vg.setLeafSize(leaf, leaf, leaf);
    vg.setInputCloud(raw);
    vg.filter(*tgt);
    const Eigen::Matrix4f guess = to_matrix(J.array("guess_colmajor", at));
    GicpProbe gicp;
    gicp.setMaxCorrespondenceDistance(J.number("corr_dist", at));     // scanmatcher_component.cpp:118
    gicp.setTransformationEpsilon(J.number("eps", at));               // :119
    gicp.setMaximumIterations((int)J.number("max_iterations", at));
    gicp.setInputTarget(tgt);
    gicp.setInputSource(src);
    Cloud aligned;
    gicp.align(aligned, guess);
    const size_t head = (size_t)J.number("head", at);
    std::vector<double> th, cs, ct;
    for (size_t i = 0; i < 64 && i < tgt->size(); i++) { th.push_back((*tgt)[i].x); th.push_back((*tgt)[i].y); th.push_back((*tgt)[i].z); }
    for (size_t i = 0; i < head && i < src->size(); i++) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) cs.push_back(gicp.source_cov(i)(r, c));
    for (size_t i = 0; i < head && i < tgt->size(); i++) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) ct.push_back(gicp.target_cov(i)(r, c));
    // 1-NN of the guess-moved source in the target, as pcl::KdTreeFLANN answers it (what getFitnessScore and the correspondences search)
    pcl::KdTreeFLANN<Point> tree;
    tree.setInputCloud(tgt);
    Cloud moved;
    pcl::transformPointCloud(*src, moved, guess);
    std::vector<double> nn_i, nn_d;
    std::vector<int> ki(1);
    std::vector<float> kd(1);
    for (const Point& q : moved.points) { tree.nearestKSearch(q, 1, ki, kd); nn_i.push_back(ki[0]); nn_d.push_back(kd[0]); }
    std::fprintf(f, "\"gicp_small\": {\n\"n_target\": %zu,\n", tgt->size());
    put(f, "target_head", th.data(), th.size());
    put(f, "cov_src_head", cs.data(), cs.size());
    put(f, "cov_tgt_head", ct.data(), ct.size());
    put(f, "nn_idx", nn_i.data(), nn_i.size());
    put(f, "nn_d2", nn_d.data(), nn_d.size());
    put(f, "final_bfgs", gicp.getFinalTransformation().data(), 16);
    std::fprintf(f, "\"iters_bfgs\": %d,\n\"fitness\": %.17g\n},\n", gicp.iterations(), gicp.getFitnessScore());
  }
  {  // ---- cfg 4: the 64 loop-closure candidates at the backend's settings (graph_based_slam_component.cpp:64-72, 227-231)
    const size_t at = J.find_key("cfg4", 0);
    const double res = J.number("resolution", at), eps = J.number("eps", at);
    const int mi = (int)J.number("max_iterations", at);
    std::fprintf(f, "\"cfg4\": [\n");
    size_t pos = J.find_key("candidates", at);
    bool first = true;
    for (;;) {
      const size_t t = J.find_key("target", pos + 1);
      if (t == std::string::npos) break;
      pos = t;
      Cloud::Ptr tgt = load(in, J.str("target", pos)), src = load(in, J.str("source", pos));
      NdtProbe ndt;
      configure(ndt, res, eps, mi);
      ndt.setInputTarget(tgt);
      ndt.setInputSource(src);
      Cloud aligned;
      ndt.align(aligned, to_matrix(J.array("guess_colmajor", pos)));
      const std::vector<double> truth = J.array("truth_rowmajor", pos);
      std::fprintf(f, "%s{", first ? "" : ",\n");
      put(f, "final", ndt.getFinalTransformation().data(), 16);
      put(f, "truth_rowmajor", truth.data(), truth.size());
      std::fprintf(f, "\"iterations\": %d, \"converged\": %d, \"fitness\": %.17g}", ndt.getFinalNumIteration(), (int)ndt.hasConverged(), ndt.getFitnessScore());
      first = false;
      pos = J.find_key("truth_rowmajor", pos);
    }
    std::fprintf(f, "\n],\n");
  }
  {  // ---- the frontend loop over a drive: ScanMatcherComponent::receiveCloud + ::updateMap (scanmatcher_component.cpp:296-356, 436-481)
     // with the map update applied before the next scan (tests/test_frontend_stream_gpu.py, hand-over lag 0)
    const size_t at = J.find_key("frontend_stream", 0);
    const double res = J.number("ndt_resolution", at), eps = J.number("eps", at);
    const int mi = (int)J.number("max_iterations", at), window = (int)J.number("num_targeted_cloud", at);
    const float vg_in = (float)J.number("vg_size_for_input", at), vg_map = (float)J.number("vg_size_for_map", at);
    const double rmin = J.number("scan_min_range", at), rmax = J.number("scan_max_range", at), trans_upd = J.number("trans_for_mapupdate", at);
    std::vector<Cloud::Ptr> submaps;            // filtered keyframes, pose-local
    std::vector<Eigen::Matrix4f> submap_pose;
    size_t pos = J.find_key("frames", at);
    const size_t scans_at = J.find_key("scans", at);
    for (;;) {
      const size_t t = J.find_key("frame_cloud", pos + 1);
      if (t == std::string::npos || t > scans_at) break;
      pos = t;
      submaps.push_back(load(in, J.str("frame_cloud", pos)));
      submap_pose.push_back(to_matrix(J.array("pose_colmajor", pos)));
    }
    auto assemble = [&](Cloud& target) {   // newest first, each moved by its pose (:448-464)
      target.clear();
      const int n = (int)submaps.size();
      for (int k = n - 1; k >= 0 && k >= n - window; k--) { Cloud moved; pcl::transformPointCloud(*submaps[k], moved, submap_pose[k]); target += moved; }
    };
    NdtProbe ndt;
    configure(ndt, res, eps, mi);
    Cloud::Ptr target(new Cloud);
    assemble(*target);
    ndt.setInputTarget(target);
    Eigen::Matrix4f pose = to_matrix(J.array("guess0_colmajor", at));
    float key_position[3] = {submap_pose.back().data()[12], submap_pose.back().data()[13], submap_pose.back().data()[14]};
    std::vector<int> update_at;
    std::fprintf(f, "\"frontend_stream\": {\n\"scans\": [\n");
    pos = scans_at;
    int scan_no = 0;
    bool first = true;
    for (;;) {
      const size_t t = J.find_key("scan_cloud", pos + 1);
      if (t == std::string::npos || t > J.find_key("loop_gate", 0)) break;
      pos = t;
      Cloud::Ptr raw = load(in, J.str("scan_cloud", pos)), ranged(new Cloud), filtered(new Cloud);
      for (const Point& p : raw->points) {   // the subscription's range filter (:210-218): horizontal range, open interval, in double
        const double r = std::sqrt(std::pow((double)p.x, 2.0) + std::pow((double)p.y, 2.0));
        if (rmin < r && r < rmax) ranged->push_back(p);
      }
      pcl::VoxelGrid<Point> vg;
      vg.setLeafSize(vg_in, vg_in, vg_in);
      vg.setInputCloud(ranged);
      vg.filter(*filtered);
      ndt.setInputSource(filtered);
      Cloud aligned;
      ndt.align(aligned, pose);
      pose = ndt.getFinalTransformation();
      std::fprintf(f, "%s{", first ? "" : ",\n");
      put(f, "final", pose.data(), 16);
      std::fprintf(f, "\"iterations\": %d, \"points_kept\": %zu}", ndt.getFinalNumIteration(), filtered->size());
      first = false;
      const float dx = pose.data()[12] - key_position[0], dy = pose.data()[13] - key_position[1], dz = pose.data()[14] - key_position[2];
      if (std::sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz) >= trans_upd) {   // :412-434 (Eigen::Vector3d norm), then updateMap
        for (int k = 0; k < 3; k++) key_position[k] = pose.data()[12 + k];
        update_at.push_back(scan_no);
        Cloud::Ptr key(new Cloud);
        pcl::VoxelGrid<Point> vgm;
        vgm.setLeafSize(vg_map, vg_map, vg_map);
        vgm.setInputCloud(ranged);
        vgm.filter(*key);
        submaps.push_back(key);
        submap_pose.push_back(pose);
        Cloud::Ptr next(new Cloud);
        assemble(*next);
        target = next;
        ndt.setInputTarget(target);
      }
      scan_no++;
    }
    std::fprintf(f, "\n],\n");
    std::vector<double> ua(update_at.begin(), update_at.end());
    put(f, "update_at", ua.data(), ua.size(), true);
    std::fprintf(f, "},\n");
  }
  {  // ---- the loop gate over a route: GraphBasedSlamComponent::searchLoop (graph_based_slam_component.cpp:164-252), NDT backend
    const size_t at = J.find_key("loop_gate", 0);
    const double thr = J.number("threshold_loop_closure_score", at), dist_lc = J.number("distance_loop_closure", at),
                 range = J.number("range_of_searching_loop_closure", at);
    const int num = (int)J.number("search_submap_num", at);
    const float leaf = (float)J.number("voxel_leaf_size", at);
    struct Sub { Cloud::Ptr cloud; Eigen::Matrix4f pose; double position[3]; double distance; };
    std::vector<Sub> subs;
    size_t pos = J.find_key("submaps", at);
    for (;;) {
      const size_t t = J.find_key("submap_cloud", pos + 1);
      if (t == std::string::npos) break;
      pos = t;
      Sub S;
      S.cloud = load(in, J.str("submap_cloud", pos));
      const std::vector<double> p3 = J.array("position", pos), q = J.array("orientation_xyzw", pos);
      for (int k = 0; k < 3; k++) S.position[k] = p3[k];
      // tf2::fromMsg(geometry_msgs::Pose) = Translation * Quaterniond: Eigen's toRotationMatrix (no normalisation), cast to float (:177-181)
      const double x = q[0], y = q[1], z = q[2], w = q[3];
      const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                   tyz = tz * y, tzz = tz * z;
      const double M[16] = {1 - (tyy + tzz), txy + twz, txz - twy, 0, txy - twz, 1 - (txx + tzz), tyz + twx, 0, txz + twy, tyz - twx, 1 - (txx + tyy), 0,
                            p3[0], p3[1], p3[2], 1};   // column-major
      S.pose = Eigen::Matrix4f::Identity();
      for (int k = 0; k < 16; k++) S.pose.data()[k] = (float)M[k];
      S.distance = J.number("distance", pos);
      subs.push_back(S);
    }
    const int n = (int)subs.size();
    const Sub& latest = subs[n - 1];
    Cloud::Ptr source(new Cloud);
    pcl::transformPointCloud(*latest.cloud, *source, latest.pose);   // :176-181
    int id_min = -1;
    double min_dist = 1e300;
    for (int i = 0; i < n; i++) {                                     // :188-205
      double d = 0;
      for (int k = 0; k < 3; k++) d += (latest.position[k] - subs[i].position[k]) * (latest.position[k] - subs[i].position[k]);
      d = std::sqrt(d);
      if (latest.distance - subs[i].distance > dist_lc && d < range && d < min_dist) { id_min = i; min_dist = d; }
    }
    std::fprintf(f, "\"loop_gate\": {\n");
    if (id_min >= 0) {
      Cloud::Ptr window(new Cloud), target(new Cloud);
      for (int j = 0; j <= 2 * num; j++) {                            // :209-222 (the reference guards the lower end only)
        const int idx = id_min + j - num;
        if (idx < 0 || idx >= n) continue;
        Cloud moved;
        pcl::transformPointCloud(*subs[idx].cloud, moved, subs[idx].pose);
        *window += moved;
      }
      pcl::VoxelGrid<Point> vg;
      vg.setLeafSize(leaf, leaf, leaf);
      vg.setInputCloud(window);
      vg.filter(*target);                                             // :224-226
      NdtProbe ndt;
      configure(ndt, J.number("ndt_resolution", at), J.number("eps", at), (int)J.number("max_iterations", at));
      ndt.setInputTarget(target);
      ndt.setInputSource(source);
      Cloud aligned;
      ndt.align(aligned);                                             // :230 (identity guess)
      const double fitness = ndt.getFitnessScore();                   // :231
      const double pair[2] = {(double)id_min, (double)(n - 1)};
      put(f, "pair_id", pair, 2);
      put(f, "final", ndt.getFinalTransformation().data(), 16);
      std::fprintf(f, "\"fitness\": %.17g, \"accepted\": %d, \"n_target_points\": %zu, \"iterations\": %d\n", fitness, (int)(fitness < thr), target->size(),
                   ndt.getFinalNumIteration());
    } else {
      std::fprintf(f, "\"pair_id\": [-1, -1], \"final\": [1,0,0,0,0,1,0,0,0,0,1,0,0,0,0,1], \"fitness\": 0, \"accepted\": 0, \"n_target_points\": 0, \"iterations\": 0\n");
    }
    std::fprintf(f, "}\n");
  }
  std::fprintf(f, "}\n");
  std::fclose(f);
  return 0;
}
