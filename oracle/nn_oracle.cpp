// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// PARITY UNPINNED.  Exact nearest-neighbour / k-nearest-neighbour search over a target cloud:
// the CPU stand-in for pcl::KdTreeFLANN (PCL 1.12 + FLANN, absent here), which the reference
// reaches through pcl::Registration::getFitnessScore (call sites
// graph_based_slam/src/graph_based_slam_component.cpp:231, scanmatcher/src/
// scanmatcher_component.cpp:376) and through GICP's correspondence / 20-NN covariance steps
// (SURVEY.md §9.7, §9.8).  A kd-tree and this grid return the same neighbours (both exact);
// ties are broken by lowest index here.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct NNGrid {
  const float* pts;
  size_t stride_f, n;
  float cell, inv;
  int mn[3], dim[3];
  std::vector<int> start;   // dim0*dim1*dim2 + 1
  std::vector<int> order;   // point indices sorted by cell (stable -> ascending index inside a cell)
};

inline void cell_of(const NNGrid& g, const float* p, int* c) {
  for (int k = 0; k < 3; k++) c[k] = (int)std::floor(p[k] * g.inv) - g.mn[k];
}

NNGrid* nn_build(const float* pts, size_t stride_f, size_t n, float cell) {
  NNGrid* g = new NNGrid();
  g->pts = pts; g->stride_f = stride_f; g->n = n; g->cell = cell; g->inv = 1.0f / cell;
  int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (size_t i = 0; i < n; i++) {
    const float* p = pts + i * stride_f;
    for (int k = 0; k < 3; k++) {
      int c = (int)std::floor(p[k] * g->inv);
      mn[k] = std::min(mn[k], c);
      mx[k] = std::max(mx[k], c);
    }
  }
  if (n == 0) { for (int k = 0; k < 3; k++) { mn[k] = 0; mx[k] = 0; } }
  for (int k = 0; k < 3; k++) { g->mn[k] = mn[k]; g->dim[k] = mx[k] - mn[k] + 1; }
  size_t ncell = (size_t)g->dim[0] * g->dim[1] * g->dim[2];
  g->start.assign(ncell + 1, 0);
  std::vector<int> key(n);
  for (size_t i = 0; i < n; i++) {
    int c[3];
    cell_of(*g, pts + i * stride_f, c);
    key[i] = c[0] + g->dim[0] * (c[1] + g->dim[1] * c[2]);
    g->start[key[i] + 1]++;
  }
  for (size_t c = 0; c < ncell; c++) g->start[c + 1] += g->start[c];
  g->order.resize(n);
  std::vector<int> cur(g->start.begin(), g->start.end() - 1);
  for (size_t i = 0; i < n; i++) g->order[cur[key[i]]++] = (int)i;
  return g;
}

struct Cand { float d2; int idx; };
inline bool cand_less(const Cand& a, const Cand& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx); }

// k nearest (k>=1) by ring expansion; exact: stop once the k-th best distance <= (ring*cell - max offset)^2 bound.
void knn_query(const NNGrid& g, const float* q, int k, Cand* best /*k, sorted ascending*/) {
  for (int i = 0; i < k; i++) { best[i].d2 = std::numeric_limits<float>::infinity(); best[i].idx = -1; }
  if (g.n == 0) return;
  int c[3];
  cell_of(g, q, c);
  // distance from q to the boundary of its own cell, per axis, lower/upper
  int maxring = 0;
  for (int k2 = 0; k2 < 3; k2++) maxring = std::max(maxring, std::max(std::abs(c[k2]) + 1, std::abs(c[k2] - g.dim[k2]) + 1));
  for (int ring = 0; ring <= maxring; ring++) {
    // all cells with Chebyshev distance == ring from c
    for (int dz = -ring; dz <= ring; dz++) {
      int z = c[2] + dz;
      if (z < 0 || z >= g.dim[2]) continue;
      for (int dy = -ring; dy <= ring; dy++) {
        int y = c[1] + dy;
        if (y < 0 || y >= g.dim[1]) continue;
        bool shell_yz = (std::abs(dz) == ring) || (std::abs(dy) == ring);
        int step = shell_yz ? 1 : std::max(1, 2 * ring);
        for (int dx = -ring; dx <= ring; dx += step) {
          int x = c[0] + dx;
          if (x < 0 || x >= g.dim[0]) continue;
          size_t ci = (size_t)x + (size_t)g.dim[0] * ((size_t)y + (size_t)g.dim[1] * z);
          for (int s = g.start[ci]; s < g.start[ci + 1]; s++) {
            int pi = g.order[s];
            const float* p = g.pts + (size_t)pi * g.stride_f;
            float dx_ = p[0] - q[0], dy_ = p[1] - q[1], dz_ = p[2] - q[2];
            Cand cd{dx_ * dx_ + dy_ * dy_ + dz_ * dz_, pi};
            if (cand_less(cd, best[k - 1])) {
              int j = k - 1;
              while (j > 0 && cand_less(cd, best[j - 1])) { best[j] = best[j - 1]; j--; }
              best[j] = cd;
            }
          }
        }
      }
    }
    // Every unvisited point lies at least `ring*cell` (minus the query's offset inside its cell,
    // bounded by using the distance to the nearest face of the visited block) away.
    float lo = std::numeric_limits<float>::infinity();
    for (int k2 = 0; k2 < 3; k2++) {
      float base = (float)(c[k2] + g.mn[k2]) * g.cell;
      float dlow = q[k2] - (base - ring * g.cell);
      float dhigh = (base + (ring + 1) * g.cell) - q[k2];
      lo = std::min(lo, std::min(dlow, dhigh));
    }
    if (lo > 0 && best[k - 1].idx >= 0 && best[k - 1].d2 < lo * lo * 0.999f) break;
  }
}

inline void xform(const float* M, const float* x, float* o) {
  o[0] = M[0] * x[0] + M[4] * x[1] + M[8] * x[2] + M[12];
  o[1] = M[1] * x[0] + M[5] * x[1] + M[9] * x[2] + M[13];
  o[2] = M[2] * x[0] + M[6] * x[1] + M[10] * x[2] + M[14];
}

}  // namespace

extern "C" {

void* orc_nn_build(const float* pts, size_t stride_floats, size_t n, float cell) { return nn_build(pts, stride_floats, n, cell); }
void orc_nn_free(void* g) { delete (NNGrid*)g; }

// 1-NN of (optionally T16-transformed, col-major fp32) queries; d2 = squared distance (fp32).
void orc_nn_search(void* gp, const float* q, size_t stride_floats, size_t n, const float* T16, int* idx, float* d2,
                   int num_threads) {
  const NNGrid& g = *(NNGrid*)gp;
#pragma omp parallel for schedule(dynamic, 64) num_threads(num_threads > 0 ? num_threads : omp_get_max_threads())
  for (long i = 0; i < (long)n; i++) {
    float p[3];
    const float* s = q + i * stride_floats;
    if (T16) xform(T16, s, p); else { p[0] = s[0]; p[1] = s[1]; p[2] = s[2]; }
    Cand b;
    knn_query(g, p, 1, &b);
    idx[i] = b.idx;
    d2[i] = b.d2;
  }
}

void orc_knn_search(void* gp, const float* q, size_t stride_floats, size_t n, int k, int* idx, float* d2, int num_threads) {
  const NNGrid& g = *(NNGrid*)gp;
#pragma omp parallel for schedule(dynamic, 64) num_threads(num_threads > 0 ? num_threads : omp_get_max_threads())
  for (long i = 0; i < (long)n; i++) {
    std::vector<Cand> b(k);
    knn_query(g, q + i * stride_floats, k, b.data());
    for (int j = 0; j < k; j++) { idx[i * k + j] = b[j].idx; d2[i * k + j] = b[j].d2; }
  }
}

// pcl::Registration::getFitnessScore(max_range) restatement: d2 <= max_range.
double orc_fitness_score(void* gp, const float* src, size_t stride_floats, size_t n, const float* T16, double max_range,
                         int num_threads) {
  std::vector<int> idx(n);
  std::vector<float> d2(n);
  orc_nn_search(gp, src, stride_floats, n, T16, idx.data(), d2.data(), num_threads);
  double sum = 0;
  long nr = 0;
  for (size_t i = 0; i < n; i++)
    if (idx[i] >= 0 && (double)d2[i] <= max_range) { sum += (double)d2[i]; nr++; }
  return nr > 0 ? sum / (double)nr : std::numeric_limits<double>::max();
}

}  // extern "C"
