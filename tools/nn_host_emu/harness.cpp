// Host emulation of nn_device.hpp: the device search code compiled for the CPU (tests/test_nn_host_emu_cpu.py, tools/nn_host_emu/run.py)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <cstdint>
#include <climits>
#define __device__
#define __host__
#define __forceinline__ inline
#define __global__
#define __restrict__
#define LSR_HOST_EMU 1
using std::min; using std::max; using std::isfinite; using std::abs;
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
struct int2 { int x, y; };
static inline int2 make_int2(int a, int b) { int2 r; r.x = a; r.y = b; return r; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
struct float4 { float x, y, z, w; };
// minimal stand-ins for what nn_device.hpp needs from common.hpp
namespace lsr { template <typename T> struct DevBuf { T* p = nullptr; };
struct DeviceCloud { float* x() const { return nullptr; } float* y() const { return nullptr; } float* z() const { return nullptr; } };
struct HashGridDev { float cell; int org[3]; int cdim[3]; DevBuf<int> coarse_block, block_off, fine_start, order; DevBuf<float4> packed; }; }
#define LSR_COMMON_HPP_STUB
struct Counters { long ranges, candidates, fine_probes, phase2_queries, coarse_blocks, offers_taken, shifts, rows_pruned; } g_cnt;
#define LSR_NN_COUNT(what, n) (g_cnt.what += (n))
#include "nn_device_emu.hpp"
using namespace lsr::nnd;
extern "C" void get_counters(long* out) { memcpy(out, &g_cnt, sizeof(g_cnt)); memset(&g_cnt, 0, sizeof(g_cnt)); }
extern "C" long run_knn(float cell, const int* org, const int* cdim, const int* coarse_block, const int* block_off, const int* fine_start,
             const float* sx, const float* sy, const float* sz, const int* order, int n_pts, const float* qx, const float* qy, const float* qz, int nq, int k, int fine_rings, int* out_idx, float* out_d2) {
  NNGridView G; G.cell = cell; G.inv_cell = 1.0f / cell;
  for (int a = 0; a < 3; a++) { G.org[a] = org[a]; G.cdim[a] = cdim[a]; }
  G.coarse_block = coarse_block; G.block_off = block_off; G.fine_start = fine_start; float4* pk = (float4*)malloc(sizeof(float4) * (size_t)n_pts);
  for (int t = 0; t < n_pts; t++) { pk[t].x = sx[t]; pk[t].y = sy[t]; pk[t].z = sz[t]; memcpy(&pk[t].w, &order[t], 4); }
  G.p = pk;
  void* lds = malloc(BestK::lds_bytes(k));
  for (int i = 0; i < nq; i++) {
    BestK c; c.init(lds, 0, k);
    nn_query(G, qx[i], qy[i], qz[i], fine_rings, INFINITY, c, -1);
    c.finalize();
    for (int j = 0; j < k; j++) { out_idx[i * k + j] = c.index(j); out_d2[i * k + j] = c.dist(j); }
  }
  return 0;
}

static NNGridView make_grid(float cell, const int* org, const int* cdim, const int* coarse_block, const int* block_off, const int* fine_start,
                            const float* sx, const float* sy, const float* sz, const int* order, int n_pts) {
  NNGridView G; G.cell = cell; G.inv_cell = 1.0f / cell;
  for (int a = 0; a < 3; a++) { G.org[a] = org[a]; G.cdim[a] = cdim[a]; }
  G.coarse_block = coarse_block; G.block_off = block_off; G.fine_start = fine_start;
  float4* pk = (float4*)malloc(sizeof(float4) * (size_t)(n_pts > 0 ? n_pts : 1));
  for (int t = 0; t < n_pts; t++) { pk[t].x = sx[t]; pk[t].y = sy[t]; pk[t].z = sz[t]; memcpy(&pk[t].w, &order[t], 4); }
  G.p = pk;
  return G;
}

// exact 1-NN by the per-thread walk (row pruning + x clipping included), with an optional distance gate
extern "C" long run_nn1(float cell, const int* org, const int* cdim, const int* coarse_block, const int* block_off, const int* fine_start,
             const float* sx, const float* sy, const float* sz, const int* order, int n_pts, const float* qx, const float* qy, const float* qz, int nq,
             int fine_rings, float max_d2, int* out_idx, float* out_d2) {
  NNGridView G = make_grid(cell, org, cdim, coarse_block, block_off, fine_start, sx, sy, sz, order, n_pts);
  for (int i = 0; i < nq; i++) {
    Best1 c; c.init();
    nn_query(G, qx[i], qy[i], qz[i], fine_rings, max_d2, c, -1);
    out_idx[i] = c.idx; out_d2[i] = c.d2;
  }
  free((void*)G.p);
  return 0;
}

// the segments fine_segment() hands out for shell r of one query: out[2 * slot] = beg, out[2 * slot + 1] = len for the
// 4 (2r+1)^2 slots.  full != 0: with pruning / clipping against `worst`.
extern "C" long run_shell_segments(float cell, const int* org, const int* cdim, const int* coarse_block, const int* block_off, const int* fine_start,
             const float* sx, const float* sy, const float* sz, const int* order, int n_pts, float qx, float qy, float qz, int r, int full, float worst,
             float max_d2, int* out) {
  NNGridView G = make_grid(cell, org, cdim, coarse_block, block_off, fine_start, sx, sy, sz, order, n_pts);
  const int fq[3] = {(int)floorf(qx * G.inv_cell) - G.org[0], (int)floorf(qy * G.inv_cell) - G.org[1], (int)floorf(qz * G.inv_cell) - G.org[2]};
  const int fdim[3] = {G.cdim[0] * 8, G.cdim[1] * 8, G.cdim[2] * 8};
  const float q[3] = {qx, qy, qz};
  const int w = 2 * r + 1, n_slots = 4 * w * w;
  for (int s = 0; s < n_slots; s++) {
    const FineSeg seg = fine_segment(G, fq, fdim, q, r, s, full != 0, worst, max_d2);
    out[2 * s] = seg.beg; out[2 * s + 1] = seg.len;
  }
  free((void*)G.p);
  return n_slots;
}

// ball_cell_range() of the seeded correspondence search: out[0..2] = lo, out[3..5] = hi; returns 1 when the box is usable
extern "C" long run_ball_range(float cell, const int* org, const int* cdim, float qx, float qy, float qz, float d2, int max_cells, int* out) {
  NNGridView G; G.cell = cell; G.inv_cell = 1.0f / cell;
  for (int a = 0; a < 3; a++) { G.org[a] = org[a]; G.cdim[a] = cdim[a]; }
  G.coarse_block = nullptr; G.block_off = nullptr; G.fine_start = nullptr; G.p = nullptr;
  const float q[3] = {qx, qy, qz};
  return ball_cell_range(G, q, d2, max_cells, out, out + 3) ? 1 : 0;
}
