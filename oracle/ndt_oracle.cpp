// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// PARITY UNPINNED.  CPU restatement (C++17 + OpenMP) of the NDT path that lidarslam_ros2
// calls through pcl::Registration (call sites: scanmatcher/src/scanmatcher_component.cpp:
// 105-113,275,307,329,353,356 and graph_based_slam/src/graph_based_slam_component.cpp:64-72,
// 181,227,230-231).  The arithmetic itself lives in the un-vendored, un-pinned submodule
// Thirdparty/ndt_omp_ros2 (/root/reference/.gitmodules:1-4; a ROS2 fork of koide3/ndt_omp)
// on top of PCL 1.12 / Eigen 3.4 (scanmatcher/package.xml:27), none present in the
// container; the reference ships no tests/golden vectors.  This file restates the
// published algorithm (Magnusson 2009 eqs 6.8-6.21; More-Thuente 1994) with ndt_omp's
// precision recipe and quirks as catalogued in SURVEY.md §9:
//   §9.1 gauss constants, §9.2 VoxelGridCovariance, §9.3 DIRECT7/1/26 lookups,
//   §9.4 Euler-XYZ parameterisation + angle derivatives (incl. the h_ang d1 "+sy" quirk),
//   §9.5 per-pair fp32 maths / fp64 accumulation / index-ordered final sum,
//   §9.6 Newton + More-Thuente line search (incl. stale h_ang in computeHessian).
// It is "a restatement of ndt_omp", never "ndt_omp".
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "linalg.h"

namespace {

struct Leaf {
  int n = 0;            // nr_points (-1 = invalidated)
  double sum[3] = {0, 0, 0};     // running sum of points (mean_ before normalisation)
  float csum[3] = {0.f, 0.f, 0.f};   // Leaf::centroid: the FLOAT running sum of the points, in cloud order ("leaf.centroid += pt")
  float centroid[3] = {0.f, 0.f, 0.f};   // ... normalised by (float)nr_points: the point the voxel-centroid kd-tree holds (KDTREE search)
  double sq[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // running sum of p p^T
  double mean[3];
  double cov[9];
  double icov[9];
  double evals[3];
};

struct Grid {
  float leaf = 1.f, inv_leaf = 1.f;
  int min_b[3] = {0, 0, 0}, max_b[3] = {0, 0, 0}, div_b[3] = {1, 1, 1}, mul[3] = {1, 1, 1};
  int min_points = 6;
  double eig_mult = 0.01;
  std::unordered_map<int, Leaf> leaves;
  bool overflow = false;
};

inline const float* P(const float* base, size_t stride_f, size_t i) { return base + i * stride_f; }

// VoxelGridCovariance::applyFilter restatement.
Grid* grid_build(const float* pts, size_t stride_f, size_t n, float leaf) {
  Grid* g = new Grid();
  g->leaf = leaf;
  g->inv_leaf = 1.0f / leaf;  // inverse_leaf_size_ = Array4f::Ones() / leaf_size_.array()
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  size_t finite = 0;
  for (size_t i = 0; i < n; i++) {
    const float* p = P(pts, stride_f, i);
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    finite++;
    for (int k = 0; k < 3; k++) {
      mn[k] = std::min(mn[k], p[k]);
      mx[k] = std::max(mx[k], p[k]);
    }
  }
  if (finite == 0) return g;
  int64_t d[3];
  for (int k = 0; k < 3; k++) d[k] = (int64_t)((mx[k] - mn[k]) * g->inv_leaf) + 1;
  if (d[0] * d[1] * d[2] > (int64_t)std::numeric_limits<int32_t>::max()) {
    g->overflow = true;  // PCL warns "Leaf size is too small ... Integer indices would overflow" and bails
    return g;
  }
  for (int k = 0; k < 3; k++) {
    g->min_b[k] = (int)std::floor(mn[k] * g->inv_leaf);
    g->max_b[k] = (int)std::floor(mx[k] * g->inv_leaf);
    g->div_b[k] = g->max_b[k] - g->min_b[k] + 1;
  }
  g->mul[0] = 1;
  g->mul[1] = g->div_b[0];
  g->mul[2] = g->div_b[0] * g->div_b[1];

  // First pass: accumulate, in point order, into the leaf map.
  for (size_t i = 0; i < n; i++) {
    const float* p = P(pts, stride_f, i);
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    int ijk[3];
    for (int k = 0; k < 3; k++)
      ijk[k] = (int)(std::floor(p[k] * g->inv_leaf) - (float)g->min_b[k]);
    int idx = ijk[0] * g->mul[0] + ijk[1] * g->mul[1] + ijk[2] * g->mul[2];
    Leaf& L = g->leaves[idx];
    double q[3] = {(double)p[0], (double)p[1], (double)p[2]};
    for (int a = 0; a < 3; a++) {
      L.sum[a] += q[a];
      L.csum[a] += p[a];   // float accumulation (Eigen::VectorXf centroid)
      for (int b = 0; b < 3; b++) L.sq[a * 3 + b] += q[a] * q[b];
    }
    L.n++;
  }

  // Second pass: mean / covariance / eigen clamp / inverse.
  for (auto& kv : g->leaves) {
    Leaf& L = kv.second;
    double nn = (double)L.n;
    for (int a = 0; a < 3; a++) L.mean[a] = L.sum[a] / nn;
    for (int a = 0; a < 3; a++) L.centroid[a] = L.csum[a] / (float)L.n;   // "leaf.centroid /= static_cast<float>(leaf.nr_points)"
    if (L.n < g->min_points) continue;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        L.cov[a * 3 + b] = (L.sq[a * 3 + b] - 2.0 * (L.sum[a] * L.mean[b])) / nn + L.mean[a] * L.mean[b];
    double f = (nn - 1.0) / nn;
    for (int a = 0; a < 9; a++) L.cov[a] *= f;
    double w[3], V[9];
    orc::sym3_eigen(L.cov, w, V);
    if (w[0] < 0 || w[1] < 0 || w[2] <= 0) {
      L.n = -1;
      continue;
    }
    double lmin = g->eig_mult * w[2];
    if (w[0] < lmin) {
      w[0] = lmin;
      if (w[1] < lmin) w[1] = lmin;
      // cov = evecs * diag(evals) * evecs^{-1}
      double Vi[9], VD[9];
      orc::mat3_inverse(V, Vi);
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) VD[a * 3 + b] = V[a * 3 + b] * w[b];
      orc::mat3_mul(VD, Vi, L.cov);
    }
    for (int a = 0; a < 3; a++) L.evals[a] = w[a];
    orc::mat3_inverse(L.cov, L.icov);
    double mxc = -std::numeric_limits<double>::infinity(), mnc = std::numeric_limits<double>::infinity();
    bool bad = false;
    for (int a = 0; a < 9; a++) {
      mxc = std::max(mxc, L.icov[a]);
      mnc = std::min(mnc, L.icov[a]);
      if (L.icov[a] != L.icov[a]) bad = true;
    }
    if (mxc == std::numeric_limits<double>::infinity() || mnc == -std::numeric_limits<double>::infinity() || bad)
      L.n = -1;
  }
  return g;
}

// Neighbourhood offsets.  search: 7 = DIRECT7, 1 = DIRECT1, 26 = DIRECT26
// (DIRECT26 in ndt_omp visits the full 3x3x3 block = 27 offsets including the centre), 0 = KDTREE.
// KDTREE (ndt_omp: target_cells_.radiusSearch(x_trans_pt, resolution_, ...) = a radius search of the kd-tree over the centroids of the
// leaves with >= min_points_per_voxel points): every leaf whose centroid lies within `resolution` of the point.  A centroid lies inside
// its own cell, so such a leaf is one of the 3 x 3 x 3 cells around the point's cell: the search is restated as those 27 cells
// filtered by the kd-tree's own test — FLANN L2_Simple<float> (dx*dx, + dy*dy, + dz*dz in float) strictly below (float)(r*r), as
// RadiusResultSet::addPoint compares.  (The kd-tree returns its hits sorted by distance; the order only decides in which order a
// point's voxel contributions are added.)
int neighbour_offsets(int search, int off[27][3]) {
  if (search == 1) {
    off[0][0] = off[0][1] = off[0][2] = 0;
    return 1;
  }
  if (search == 26 || search == 0) {
    int c = 0;
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dz = -1; dz <= 1; dz++) {
          off[c][0] = dx; off[c][1] = dy; off[c][2] = dz;
          c++;
        }
    return c;
  }
  const int o7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  for (int i = 0; i < 7; i++)
    for (int k = 0; k < 3; k++) off[i][k] = o7[i][k];
  return 7;
}

inline int neighbours(const Grid& g, const float* xt, const int off[27][3], int noff, const Leaf** out, float radius2 = -1.f) {
  int ijk[3];
  for (int k = 0; k < 3; k++) ijk[k] = (int)std::floor(xt[k] / g.leaf);  // float / float, then floor
  int cnt = 0;
  for (int o = 0; o < noff; o++) {
    bool ok = true;
    for (int k = 0; k < 3; k++) {
      int c = ijk[k] + off[o][k];
      if (c < g.min_b[k] || c > g.max_b[k]) ok = false;
    }
    if (!ok) continue;
    int idx = (ijk[0] + off[o][0] - g.min_b[0]) * g.mul[0] + (ijk[1] + off[o][1] - g.min_b[1]) * g.mul[1] +
              (ijk[2] + off[o][2] - g.min_b[2]) * g.mul[2];
    auto it = g.leaves.find(idx);
    if (it == g.leaves.end() || it->second.n < g.min_points) continue;
    if (radius2 >= 0.f) {   // KDTREE: the kd-tree's radius test on the leaf's float centroid
      const Leaf& L = it->second;
      float d = 0.f;
      for (int k = 0; k < 3; k++) {
        const float diff = xt[k] - L.centroid[k];
        d += diff * diff;
      }
      if (!(d < radius2)) continue;
    }
    out[cnt++] = &it->second;
  }
  return cnt;
}

// Angle derivative tables: double members + float copies, as ndt_omp keeps.
struct AngleDeriv {
  double j[8][3];    // a..h
  double h[15][3];   // a2,a3,b2,b3,c2,c3,d1,d2,d3,e1,e2,e3,f1,f2,f3
  float jf[8][3];
  float hf[15][3];
};

void compute_angle_derivatives(const double* p, bool compute_hessian, int d1_sign, AngleDeriv& A) {
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
  if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
  if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
  double J[8][3] = {
      {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},
      {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},
      {(-sy * cz), sy * sz, cy},
      {sx * cy * cz, (-sx * cy * sz), sx * sy},
      {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},
      {(-cy * sz), (-cy * cz), 0},
      {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},
      {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 3; c++) {
      A.j[r][c] = J[r][c];
      A.jf[r][c] = (float)J[r][c];
    }
  if (compute_hessian) {
    double H[15][3] = {
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},    // a2
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)}, // a3
        {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},                       // b2
        {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},                       // b3
        {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},           // c2
        {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},           // c3
        {(-cy * cz), (cy * sz), (d1_sign >= 0 ? sy : -sy)},                 // d1 (upstream: +sy)
        {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},                       // d2
        {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},                      // d3
        {(sy * sz), (sy * cz), 0},                                          // e1
        {(-sx * cy * sz), (-sx * cy * cz), 0},                              // e2
        {(cx * cy * sz), (cx * cy * cz), 0},                                // e3
        {(-cy * cz), (cy * sz), 0},                                         // f1
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},          // f2
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};         // f3
    for (int r = 0; r < 15; r++)
      for (int c = 0; c < 3; c++) {
        A.h[r][c] = H[r][c];
        A.hf[r][c] = (float)H[r][c];
      }
  }
}

// ---- fp32 transforms as Eigen::Affine3f would build them -------------------------------
// (Translation3f(t) * AngleAxisf(rx,X) * AngleAxisf(ry,Y) * AngleAxisf(rz,Z)).matrix(), column-major
void pose_to_matrix_f(const double* p, float* M /*col-major 4x4*/) {
  float ax = (float)p[3], ay = (float)p[4], az = (float)p[5];
  float cx = std::cos(ax), sx = std::sin(ax), cy = std::cos(ay), sy = std::sin(ay), cz = std::cos(az), sz = std::sin(az);
  float Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  float Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  float Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  float A[9], R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float s = 0;
      for (int k = 0; k < 3; k++) s += Rx[i * 3 + k] * Ry[k * 3 + j];
      A[i * 3 + j] = s;
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float s = 0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * Rz[k * 3 + j];
      R[i * 3 + j] = s;
    }
  for (int i = 0; i < 16; i++) M[i] = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[j * 4 + i] = R[i * 3 + j];
  M[12] = (float)p[0]; M[13] = (float)p[1]; M[14] = (float)p[2]; M[15] = 1.f;
}

// Eigen 3.4 MatrixBase::eulerAngles(0,1,2) on a float 3x3 (ranges [0,pi]x[-pi,pi]x[-pi,pi]).
void euler_angles_012_f(const float* M /*col-major 4x4*/, float* res) {
  auto c = [&](int r, int cc) { return M[cc * 4 + r]; };
  const int i = 0, j = 1, k = 2;  // a0=0,a1=1,a2=2 -> odd = 0
  const float PI = 3.14159265358979323846f;
  res[0] = std::atan2(c(j, k), c(k, k));
  float c2 = std::sqrt(c(i, i) * c(i, i) + c(i, j) * c(i, j));
  if (res[0] > 0.f) {  // (!odd) && res[0] > 0
    if (res[0] > 0.f) res[0] -= PI; else res[0] += PI;
    res[1] = std::atan2(-c(i, k), -c2);
  } else {
    res[1] = std::atan2(-c(i, k), c2);
  }
  float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * c(k, i) - c1 * c(j, i), c1 * c(j, j) - s1 * c(k, j));
  res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2];  // !odd
}

inline void transform_point_f(const float* M, const float* x, float* o) {
  o[0] = M[0] * x[0] + M[4] * x[1] + M[8] * x[2] + M[12];
  o[1] = M[1] * x[0] + M[5] * x[1] + M[9] * x[2] + M[13];
  o[2] = M[2] * x[0] + M[6] * x[1] + M[10] * x[2] + M[14];
}

struct NdtParams {
  double resolution;
  double step_size;
  double outlier_ratio;
  double trans_eps;
  int max_iterations;
  int search;       // 7 / 1 / 26
  int d1_sign;      // +1 upstream quirk, -1 analytic
  int num_threads;  // 0 = all
};

struct NdtResult {
  float final_transformation[16];
  int converged;
  int iterations;
  double trans_probability;
  double final_p[6];
  int n_evals;            // derivative evaluations with hessian
  int n_evals_grad;       // gradient-only evaluations
  int n_hessian_recompute;
};

// Test hook (tests/ndt_host_emu.py): the Newton / More-Thuente loop below driven by a DIFFERENT derivative evaluation — the GPU
// kernels' own fp32 operation order compiled for the host (tools/ndt_host_emu) — to measure on the CPU what that operation
// order alone does to a registration.  mode: 3 = first pass of align() (point transform = T16 as given), 1 = pass with
// Hessian at p (refresh j_ang and h_ang), 0 = gradient-only pass at p (refresh j_ang only), 2 = computeHessian at the pose of
// the last pass (current j_ang, h_ang as left by the last mode-1/3 pass; only hess is written).  Returns the score.
typedef double (*DerivCb)(void* user, const double* p, const float* T16, int mode, double* grad, double* hess);

struct Ndt {
  DerivCb cb = nullptr;
  void* cb_user = nullptr;
  bool cb_first = true;
  const Grid* g;
  const float* src;
  size_t stride_f, n;
  NdtParams prm;
  double d1, d2;
  AngleDeriv ang;
  int off[27][3];
  int noff;
  float radius2 = -1.f;   // KDTREE: (float)(resolution * resolution) as pcl::KdTreeFLANN::radiusSearch hands it to FLANN; < 0 otherwise
  std::vector<float> trans;  // transformed cloud xyz
  std::vector<double> sc, gr, he;  // per-point results (ndt_omp keeps scores[]/score_gradients[]/hessians[])
  NdtResult* res;
};

void gauss_constants(double res, double outlier, double* d1, double* d2, double* d3) {
  double c1 = 10 * (1 - outlier);
  double c2 = outlier / std::pow(res, 3);
  *d3 = -std::log(c2);
  *d1 = -std::log(c1 + c2) - *d3;
  *d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - *d3) / *d1);
}

void transform_cloud(Ndt& S, const float* M) {
  if (S.cb) return;   // the hook transforms the points itself
  S.trans.resize(S.n * 3);
#pragma omp parallel for schedule(static) num_threads(S.prm.num_threads > 0 ? S.prm.num_threads : omp_get_max_threads())
  for (long i = 0; i < (long)S.n; i++) transform_point_f(M, P(S.src, S.stride_f, i), &S.trans[3 * i]);
}

// computeDerivatives + updateDerivatives restatement: per-pair fp32,
// per-point fp64 accumulators, final sum sequential in index order.
double compute_derivatives(Ndt& S, const double* p, bool compute_hessian, double* grad, double* hess) {
  if (S.cb) {
    for (int a = 0; a < 36; a++) hess[a] = 0;
    const int mode = S.cb_first ? 3 : (compute_hessian ? 1 : 0);
    S.cb_first = false;
    return S.cb(S.cb_user, p, S.res->final_transformation, mode, grad, hess);
  }
  compute_angle_derivatives(p, compute_hessian, S.prm.d1_sign, S.ang);
  const size_t n = S.n;
  std::vector<double>&sc = S.sc, &gr = S.gr, &he = S.he;
  if (sc.size() != n) { sc.resize(n); gr.resize(n * 6); }
  if (compute_hessian && he.size() != n * 36) he.resize(n * 36);
  const float gd2 = (float)S.d2;
  const double gd1 = S.d1;
#pragma omp parallel for schedule(guided, 8) num_threads(S.prm.num_threads > 0 ? S.prm.num_threads : omp_get_max_threads())
  for (long idx = 0; idx < (long)n; idx++) {
    const float* xs = P(S.src, S.stride_f, idx);
    const float* xt = &S.trans[3 * idx];
    const Leaf* nb[27];
    int cnt = neighbours(*S.g, xt, S.off, S.noff, nb, S.radius2);
    double score_pt = 0, g_pt[6] = {0, 0, 0, 0, 0, 0}, h_pt[36];
    for (int a = 0; a < 36; a++) h_pt[a] = 0;
    // computePointDerivatives (float): x is the ORIGINAL point
    float x4[3] = {(float)(double)xs[0], (float)(double)xs[1], (float)(double)xs[2]};
    float ja[8];
    for (int r = 0; r < 8; r++) ja[r] = S.ang.jf[r][0] * x4[0] + S.ang.jf[r][1] * x4[1] + S.ang.jf[r][2] * x4[2];
    float Jm[3][6] = {{1, 0, 0, 0, ja[2], ja[5]}, {0, 1, 0, ja[0], ja[3], ja[6]}, {0, 0, 1, ja[1], ja[4], ja[7]}};
    float Hv[6][6][3];
    if (compute_hessian) {
      float hh[15];
      for (int r = 0; r < 15; r++) hh[r] = S.ang.hf[r][0] * x4[0] + S.ang.hf[r][1] * x4[1] + S.ang.hf[r][2] * x4[2];
      float va[3] = {0, hh[0], hh[1]}, vb[3] = {0, hh[2], hh[3]}, vc[3] = {0, hh[4], hh[5]};
      float vd[3] = {hh[6], hh[7], hh[8]}, ve[3] = {hh[9], hh[10], hh[11]}, vf[3] = {hh[12], hh[13], hh[14]};
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++)
          for (int k = 0; k < 3; k++) Hv[a][b][k] = 0;
      for (int k = 0; k < 3; k++) {
        Hv[3][3][k] = va[k]; Hv[4][3][k] = vb[k]; Hv[5][3][k] = vc[k];
        Hv[3][4][k] = vb[k]; Hv[4][4][k] = vd[k]; Hv[5][4][k] = ve[k];
        Hv[3][5][k] = vc[k]; Hv[4][5][k] = ve[k]; Hv[5][5][k] = vf[k];
      }
    }
    for (int c = 0; c < cnt; c++) {
      const Leaf& L = *nb[c];
      // x_trans (double) -= mean ; then cast to float
      float q[3];
      for (int k = 0; k < 3; k++) q[k] = (float)((double)xt[k] - L.mean[k]);
      float C[3][3];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) C[a][b] = (float)L.icov[a * 3 + b];
      // x_trans4 * c_inv4  (row vector times matrix)
      float qC[3];
      for (int b = 0; b < 3; b++) qC[b] = q[0] * C[0][b] + q[1] * C[1][b] + q[2] * C[2][b];
      float qCq = q[0] * qC[0] + q[1] * qC[1] + q[2] * qC[2];
      float e = std::exp(-gd2 * qCq * 0.5f);
      float score_inc = (float)(-gd1 * (double)e);
      e = gd2 * e;
      if (e > 1 || e < 0 || e != e) continue;
      e = (float)((double)e * gd1);
      // c_inv4 * point_gradient4  (3x6)
      float CJ[3][6];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 6; b++) CJ[a][b] = C[a][0] * Jm[0][b] + C[a][1] * Jm[1][b] + C[a][2] * Jm[2][b];
      float u[6];
      for (int b = 0; b < 6; b++) u[b] = q[0] * CJ[0][b] + q[1] * CJ[1][b] + q[2] * CJ[2][b];
      for (int b = 0; b < 6; b++) g_pt[b] += (double)(e * u[b]);
      if (compute_hessian) {
        float JCJ[6][6];
        for (int a = 0; a < 6; a++)
          for (int b = 0; b < 6; b++) JCJ[a][b] = Jm[0][a] * CJ[0][b] + Jm[1][a] * CJ[1][b] + Jm[2][a] * CJ[2][b];
        for (int i = 0; i < 6; i++) {
          float xCH[6];
          for (int j = 0; j < 6; j++) xCH[j] = qC[0] * Hv[i][j][0] + qC[1] * Hv[i][j][1] + qC[2] * Hv[i][j][2];
          for (int j = 0; j < 6; j++) h_pt[i * 6 + j] += (double)(e * (-gd2 * u[i] * u[j] + xCH[j] + JCJ[j][i]));
        }
      }
      score_pt += (double)score_inc;
    }
    sc[idx] = score_pt;
    for (int b = 0; b < 6; b++) gr[idx * 6 + b] = g_pt[b];
    if (compute_hessian)
      for (int a = 0; a < 36; a++) he[idx * 36 + a] = h_pt[a];
  }
  double score = 0;
  for (int b = 0; b < 6; b++) grad[b] = 0;
  for (int a = 0; a < 36; a++) hess[a] = 0;  // hessian.setZero() happens regardless of compute_hessian
  for (size_t i = 0; i < n; i++) {
    score += sc[i];
    for (int b = 0; b < 6; b++) grad[b] += gr[i * 6 + b];
    if (compute_hessian)
      for (int a = 0; a < 36; a++) hess[a] += he[i * 36 + a];
  }
  return score;
}

// computeHessian + updateHessian restatement: fp64 per pair, sequential; uses the CURRENT
// j_ang and whatever h_ang the last compute_hessian=true call left behind.
void compute_hessian_only(Ndt& S, double* hess) {
  if (S.cb) {
    double g_unused[6];
    S.cb(S.cb_user, nullptr, S.res->final_transformation, 2, g_unused, hess);
    return;
  }
  for (int a = 0; a < 36; a++) hess[a] = 0;
  for (size_t idx = 0; idx < S.n; idx++) {
    const float* xs = P(S.src, S.stride_f, idx);
    const float* xt = &S.trans[3 * idx];
    const Leaf* nb[27];
    int cnt = neighbours(*S.g, xt, S.off, S.noff, nb, S.radius2);
    if (!cnt) continue;
    double x[3] = {xs[0], xs[1], xs[2]};
    double ja[8], hh[15];
    for (int r = 0; r < 8; r++) ja[r] = x[0] * S.ang.j[r][0] + x[1] * S.ang.j[r][1] + x[2] * S.ang.j[r][2];
    for (int r = 0; r < 15; r++) hh[r] = x[0] * S.ang.h[r][0] + x[1] * S.ang.h[r][1] + x[2] * S.ang.h[r][2];
    double Jm[3][6] = {{1, 0, 0, 0, ja[2], ja[5]}, {0, 1, 0, ja[0], ja[3], ja[6]}, {0, 0, 1, ja[1], ja[4], ja[7]}};
    double Hv[6][6][3];
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++)
        for (int k = 0; k < 3; k++) Hv[a][b][k] = 0;
    double va[3] = {0, hh[0], hh[1]}, vb[3] = {0, hh[2], hh[3]}, vc[3] = {0, hh[4], hh[5]};
    double vd[3] = {hh[6], hh[7], hh[8]}, ve[3] = {hh[9], hh[10], hh[11]}, vf[3] = {hh[12], hh[13], hh[14]};
    for (int k = 0; k < 3; k++) {
      Hv[3][3][k] = va[k]; Hv[4][3][k] = vb[k]; Hv[5][3][k] = vc[k];
      Hv[3][4][k] = vb[k]; Hv[4][4][k] = vd[k]; Hv[5][4][k] = ve[k];
      Hv[3][5][k] = vc[k]; Hv[4][5][k] = ve[k]; Hv[5][5][k] = vf[k];
    }
    for (int c = 0; c < cnt; c++) {
      const Leaf& L = *nb[c];
      double q[3];
      for (int k = 0; k < 3; k++) q[k] = (double)xt[k] - L.mean[k];
      const double* C = L.icov;
      double Cq[3];
      for (int a = 0; a < 3; a++) Cq[a] = C[a * 3 + 0] * q[0] + C[a * 3 + 1] * q[1] + C[a * 3 + 2] * q[2];
      double e = S.d2 * std::exp(-S.d2 * (q[0] * Cq[0] + q[1] * Cq[1] + q[2] * Cq[2]) / 2);
      if (e > 1 || e < 0 || e != e) continue;
      e *= S.d1;
      double CJ[3][6];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 6; b++) CJ[a][b] = C[a * 3 + 0] * Jm[0][b] + C[a * 3 + 1] * Jm[1][b] + C[a * 3 + 2] * Jm[2][b];
      double u[6];
      for (int b = 0; b < 6; b++) u[b] = q[0] * CJ[0][b] + q[1] * CJ[1][b] + q[2] * CJ[2][b];
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
          double CH[3];
          for (int a = 0; a < 3; a++)
            CH[a] = C[a * 3 + 0] * Hv[i][j][0] + C[a * 3 + 1] * Hv[i][j][1] + C[a * 3 + 2] * Hv[i][j][2];
          double xCH = q[0] * CH[0] + q[1] * CH[1] + q[2] * CH[2];
          double JCJ = Jm[0][j] * CJ[0][i] + Jm[1][j] * CJ[1][i] + Jm[2][j] * CJ[2][i];
          hess[i * 6 + j] += e * (-S.d2 * u[i] * u[j] + xCH + JCJ);
        }
    }
  }
}

// ---- More-Thuente -------------------------------------------------------
inline double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
inline double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

double trial_value_selection_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t,
                                double f_t, double g_t) {
  if (f_t > f_l) {  // case 1
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    if (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {  // case 2
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    if (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (std::fabs(g_t) <= std::fabs(g_l)) {  // case 3
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double a_t_next = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
    if (a_t > a_l) return std::min(a_t + 0.66 * (a_u - a_t), a_t_next);
    return std::max(a_t + 0.66 * (a_u - a_t), a_t_next);
  } else {  // case 4
    double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
    double w = std::sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
  }
}

bool update_interval_mt(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t,
                        double f_t, double g_t) {
  if (f_t > f_l) {
    a_u = a_t; f_u = f_t; g_u = g_t;
    return false;
  } else if (g_t * (a_l - a_t) > 0) {
    a_l = a_t; f_l = f_t; g_l = g_t;
    return false;
  } else if (g_t * (a_l - a_t) < 0) {
    a_u = a_l; f_u = f_l; g_u = g_l;
    a_l = a_t; f_l = f_t; g_l = g_t;
    return false;
  }
  return true;
}

double compute_step_length_mt(Ndt& S, const double* x, double* step_dir, double step_init, double step_max,
                              double step_min, double& score, double* grad, double* hess) {
  double phi_0 = -score;
  double d_phi_0 = 0;
  for (int i = 0; i < 6; i++) d_phi_0 += grad[i] * step_dir[i];
  d_phi_0 = -d_phi_0;
  double x_t[6];
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) return 0;
    d_phi_0 *= -1;
    for (int i = 0; i < 6; i++) step_dir[i] *= -1;
  }
  const int max_step_iterations = 10;
  int step_iterations = 0;
  const double mu = 1.e-4, nu = 0.9;
  double a_l = 0, a_u = 0;
  double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu);
  double g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
  double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu);
  double g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
  bool interval_converged = (step_max - step_min) < 0, open_interval = true;
  double a_t = step_init;
  a_t = std::min(a_t, step_max);
  a_t = std::max(a_t, step_min);
  for (int i = 0; i < 6; i++) x_t[i] = x[i] + step_dir[i] * a_t;
  pose_to_matrix_f(x_t, S.res->final_transformation);
  transform_cloud(S, S.res->final_transformation);
  score = compute_derivatives(S, x_t, true, grad, hess);
  S.res->n_evals++;
  double phi_t = -score;
  double d_phi_t = 0;
  for (int i = 0; i < 6; i++) d_phi_t += grad[i] * step_dir[i];
  d_phi_t = -d_phi_t;
  double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
  double d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
  while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    if (open_interval)
      a_t = trial_value_selection_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
    else
      a_t = trial_value_selection_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_t = std::min(a_t, step_max);
    a_t = std::max(a_t, step_min);
    for (int i = 0; i < 6; i++) x_t[i] = x[i] + step_dir[i] * a_t;
    pose_to_matrix_f(x_t, S.res->final_transformation);
    transform_cloud(S, S.res->final_transformation);
    score = compute_derivatives(S, x_t, false, grad, hess);
    S.res->n_evals_grad++;
    phi_t = -score;
    d_phi_t = 0;
    for (int i = 0; i < 6; i++) d_phi_t += grad[i] * step_dir[i];
    d_phi_t = -d_phi_t;
    psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
    d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = false;
      f_l = f_l + phi_0 - mu * d_phi_0 * a_l;
      g_l = g_l + mu * d_phi_0;
      f_u = f_u + phi_0 - mu * d_phi_0 * a_u;
      g_u = g_u + mu * d_phi_0;
    }
    if (open_interval)
      interval_converged = update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
    else
      interval_converged = update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    step_iterations++;
  }
  if (step_iterations) {
    compute_hessian_only(S, hess);
    S.res->n_hessian_recompute++;
  }
  return a_t;
}

}  // namespace

extern "C" {

void* orc_grid_build(const float* pts, size_t stride_floats, size_t n, float leaf) {
  return grid_build(pts, stride_floats, n, leaf);
}
void orc_grid_free(void* g) { delete (Grid*)g; }

// info[0..2]=min_b, [3..5]=max_b, [6]=#leaves (any n), [7]=#valid (n>=6), [8]=overflow
void orc_grid_info(void* gp, int* info) {
  Grid* g = (Grid*)gp;
  for (int k = 0; k < 3; k++) {
    info[k] = g->min_b[k];
    info[3 + k] = g->max_b[k];
  }
  int valid = 0;
  for (auto& kv : g->leaves)
    if (kv.second.n >= g->min_points) valid++;
  info[6] = (int)g->leaves.size();
  info[7] = valid;
  info[8] = g->overflow ? 1 : 0;
}

// Dump all leaves sorted by linear index. Arrays sized for info[6] leaves.
int orc_grid_dump(void* gp, int* idx, int* npts, double* mean, double* cov, double* icov) {
  Grid* g = (Grid*)gp;
  std::vector<int> keys;
  keys.reserve(g->leaves.size());
  for (auto& kv : g->leaves) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end());
  int c = 0;
  for (int k : keys) {
    const Leaf& L = g->leaves[k];
    idx[c] = k;
    npts[c] = L.n;
    for (int a = 0; a < 3; a++) mean[c * 3 + a] = L.mean[a];
    for (int a = 0; a < 9; a++) {
      cov[c * 9 + a] = (L.n >= g->min_points) ? L.cov[a] : 0.0;
      icov[c * 9 + a] = (L.n >= g->min_points) ? L.icov[a] : 0.0;
    }
    c++;
  }
  return c;
}

// The float centroids (Leaf::centroid, what the voxel-centroid kd-tree of the KDTREE search holds) of all leaves, sorted by linear
// index like orc_grid_dump: centroid[3 c .. 3 c + 2].
int orc_grid_centroids(void* gp, float* centroid) {
  Grid* g = (Grid*)gp;
  std::vector<int> keys;
  keys.reserve(g->leaves.size());
  for (auto& kv : g->leaves) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end());
  int c = 0;
  for (int k : keys) {
    const Leaf& L = g->leaves[k];
    for (int a = 0; a < 3; a++) centroid[c * 3 + a] = L.centroid[a];
    c++;
  }
  return c;
}

void orc_gauss_constants(double res, double outlier, double* d1, double* d2, double* d3) {
  gauss_constants(res, outlier, d1, d2, d3);
}

void orc_pose_to_matrix(const double* p, float* M) { pose_to_matrix_f(p, M); }
void orc_matrix_to_pose(const float* M, double* p) {
  float e[3];
  euler_angles_012_f(M, e);
  p[0] = M[12]; p[1] = M[13]; p[2] = M[14];
  p[3] = e[0]; p[4] = e[1]; p[5] = e[2];
}

// One derivative evaluation at pose p.  If T16 != NULL the cloud is transformed by T16
// (col-major fp32) instead of the matrix rebuilt from p (first evaluation of align()).
// fp64_hessian != 0 evaluates the Hessian through the fp64 computeHessian path instead.
double orc_ndt_derivatives(void* gp, const float* src, size_t stride_floats, size_t n, const double* p,
                           const float* T16, int compute_hessian, int search, int d1_sign, int num_threads,
                           double resolution, double outlier_ratio, double* grad, double* hess, int fp64_hessian) {
  Ndt S;
  NdtResult R;
  std::memset(&R, 0, sizeof(R));
  S.g = (Grid*)gp; S.src = src; S.stride_f = stride_floats; S.n = n; S.res = &R;
  S.prm = NdtParams{resolution, 0.1, outlier_ratio, 0.01, 35, search, d1_sign, num_threads};
  double d3;
  gauss_constants(resolution, outlier_ratio, &S.d1, &S.d2, &d3);
  S.noff = neighbour_offsets(search, S.off);
  if (search == 0) S.radius2 = (float)((double)(float)resolution * (double)(float)resolution);   // resolution_ is a float member
  float M[16];
  if (T16) std::memcpy(M, T16, sizeof(M)); else pose_to_matrix_f(p, M);
  transform_cloud(S, M);
  double score = compute_derivatives(S, p, compute_hessian != 0, grad, hess);
  if (fp64_hessian) compute_hessian_only(S, hess);
  return score;
}

// pcl::Registration::align + NDT::computeTransformation restatement.
// trace (nullable): per Newton iteration 9 doubles {p[6], score, step_length, n_evals_so_far}.
static int ndt_align_impl(void* gp, const float* src, size_t stride_floats, size_t n, const float* guess16,
                          const NdtParams* prm, NdtResult* R, double* trace, int trace_cap, DerivCb cb, void* cb_user) {
  Ndt S;
  S.cb = cb; S.cb_user = cb_user;
  std::memset(R, 0, sizeof(*R));
  S.g = (Grid*)gp; S.src = src; S.stride_f = stride_floats; S.n = n; S.res = R; S.prm = *prm;
  double d3;
  gauss_constants(prm->resolution, prm->outlier_ratio, &S.d1, &S.d2, &d3);
  S.noff = neighbour_offsets(prm->search, S.off);
  if (prm->search == 0) S.radius2 = (float)((double)(float)prm->resolution * (double)(float)prm->resolution);
  float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float* F = R->final_transformation;
  std::memcpy(F, I16, sizeof(I16));
  bool guess_is_identity = true;
  if (guess16)
    for (int i = 0; i < 16; i++)
      if (guess16[i] != I16[i]) guess_is_identity = false;
  if (!guess_is_identity) std::memcpy(F, guess16, sizeof(I16));
  transform_cloud(S, F);  // output = guess * input (identity guess leaves the copy untouched)

  double p[6], delta_p[6], grad[6], hess[36];
  orc_matrix_to_pose(F, p);
  double score = compute_derivatives(S, p, true, grad, hess);
  R->n_evals++;
  int nr_iterations = 0;
  bool converged = false;
  while (!converged) {
    double neg_g[6];
    for (int i = 0; i < 6; i++) neg_g[i] = -grad[i];
    orc::svd6_solve(hess, neg_g, delta_p);
    double nrm = 0;
    for (int i = 0; i < 6; i++) nrm += delta_p[i] * delta_p[i];
    nrm = std::sqrt(nrm);
    if (nrm == 0 || nrm != nrm) {
      R->trans_probability = score / (double)n;
      R->converged = (nrm == nrm) ? 1 : 0;
      R->iterations = nr_iterations;
      for (int i = 0; i < 6; i++) R->final_p[i] = p[i];
      return 0;
    }
    for (int i = 0; i < 6; i++) delta_p[i] /= nrm;
    nrm = compute_step_length_mt(S, p, delta_p, nrm, prm->step_size, prm->trans_eps / 2, score, grad, hess);
    for (int i = 0; i < 6; i++) delta_p[i] *= nrm;
    for (int i = 0; i < 6; i++) p[i] += delta_p[i];
    if (trace && nr_iterations < trace_cap) {
      double* t = trace + 9 * nr_iterations;
      for (int i = 0; i < 6; i++) t[i] = p[i];
      t[6] = score; t[7] = nrm; t[8] = (double)(R->n_evals + R->n_evals_grad + R->n_hessian_recompute);
    }
    if (nr_iterations > prm->max_iterations || (nr_iterations && (std::fabs(nrm) < prm->trans_eps))) converged = true;
    nr_iterations++;
  }
  R->trans_probability = score / (double)n;
  R->converged = 1;
  R->iterations = nr_iterations;
  for (int i = 0; i < 6; i++) R->final_p[i] = p[i];
  return 0;
}

int orc_ndt_align(void* gp, const float* src, size_t stride_floats, size_t n, const float* guess16,
                  const NdtParams* prm, NdtResult* R, double* trace, int trace_cap) {
  return ndt_align_impl(gp, src, stride_floats, n, guess16, prm, R, trace, trace_cap, nullptr, nullptr);
}
// the same loop with the derivative evaluation replaced by `cb` (see DerivCb)
int orc_ndt_align_cb(void* gp, const float* src, size_t stride_floats, size_t n, const float* guess16,
                     const NdtParams* prm, NdtResult* R, double* trace, int trace_cap, DerivCb cb, void* cb_user) {
  return ndt_align_impl(gp, src, stride_floats, n, guess16, prm, R, trace, trace_cap, cb, cb_user);
}

// pcl::VoxelGrid<PointT>::applyFilter restatement (PCL 1.12 voxel_grid.hpp; call sites
// scanmatcher_component.cpp:324-328,266-269,443-447; graph_based_slam_component.cpp:224-226): leaf index as
// in VoxelGridCovariance, points sorted by leaf index, centroid per leaf accumulated in FLOAT
// (CentroidPoint's AccumulatorXYZ is Eigen::Vector3f), output in ascending leaf order.  std::sort's order
// inside a leaf is unspecified upstream; ascending point index is used here.
int orc_voxel_grid_filter_impl(const float* pts, size_t stride_f, size_t n, float leaf, float* out, int intensity_col);
int orc_voxel_grid_filter(const float* pts, size_t stride_f, size_t n, float leaf, float* out_xyz) {
  return orc_voxel_grid_filter_impl(pts, stride_f, n, leaf, out_xyz, -1);
}
// with downsample_all_data_ (PCL's default) every field is averaged: out_xyzi = 4 floats per leaf {x, y, z, intensity},
// intensity read from column `intensity_col` of the input records (AccumulatorIntensity is a float sum too)
int orc_voxel_grid_filter_xyzi(const float* pts, size_t stride_f, size_t n, float leaf, int intensity_col, float* out_xyzi) {
  return orc_voxel_grid_filter_impl(pts, stride_f, n, leaf, out_xyzi, intensity_col);
}
int orc_voxel_grid_filter_impl(const float* pts, size_t stride_f, size_t n, float leaf, float* out_xyz, int intensity_col) {
  const int ow = intensity_col >= 0 ? 4 : 3;
  const float inv = 1.0f / leaf;
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  size_t finite = 0;
  for (size_t i = 0; i < n; i++) {
    const float* p = pts + i * stride_f;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    finite++;
    for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); }
  }
  if (!finite) return 0;
  int min_b[3], div_b[3];
  for (int k = 0; k < 3; k++) {
    min_b[k] = (int)std::floor(mn[k] * inv);
    div_b[k] = (int)std::floor(mx[k] * inv) - min_b[k] + 1;
  }
  std::vector<std::pair<unsigned int, unsigned int>> iv;
  iv.reserve(finite);
  for (size_t i = 0; i < n; i++) {
    const float* p = pts + i * stride_f;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    int i0 = (int)(std::floor(p[0] * inv) - (float)min_b[0]);
    int i1 = (int)(std::floor(p[1] * inv) - (float)min_b[1]);
    int i2 = (int)(std::floor(p[2] * inv) - (float)min_b[2]);
    iv.emplace_back((unsigned int)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]), (unsigned int)i);
  }
  std::sort(iv.begin(), iv.end());
  int cnt = 0;
  size_t a = 0;
  while (a < iv.size()) {
    size_t b = a;
    float acc[4] = {0, 0, 0, 0};
    while (b < iv.size() && iv[b].first == iv[a].first) {
      const float* p = pts + (size_t)iv[b].second * stride_f;
      acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
      if (intensity_col >= 0) acc[3] += p[intensity_col];
      b++;
    }
    float m = (float)(b - a);
    out_xyz[ow * cnt] = acc[0] / m; out_xyz[ow * cnt + 1] = acc[1] / m; out_xyz[ow * cnt + 2] = acc[2] / m;
    if (intensity_col >= 0) out_xyz[ow * cnt + 3] = acc[3] / m;
    cnt++;
    a = b;
  }
  return cnt;
}

int orc_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
