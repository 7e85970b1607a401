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
    std::fprintf(f, "\"iters_tight\": %d\n},\n", ndt.getFinalNumIteration());
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
    std::fprintf(f, "\n]\n");
  }
  std::fprintf(f, "}\n");
  std::fclose(f);
  return 0;
}
