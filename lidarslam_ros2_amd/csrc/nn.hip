// Kd-tree-free EXACT nearest-neighbour search for gfx950 (K8 fitness score, K5/K6 of GICP).
// Replaces pcl::KdTreeFLANN behind pcl::Registration::getFitnessScore
// (graph_based_slam/src/graph_based_slam_component.cpp:231, scanmatcher/src/scanmatcher_component.cpp:376)
// and behind GICP's 1-NN correspondences / 20-NN covariances.
//
// Structure: a two-level blocked voxel grid.  Points are radix-sorted by (coarse cell, fine cell);
// coarse cells (8x8x8 fine cells) are a dense int32 map to a block id, each block owns 513 fine-cell
// start offsets.  A query first scans the (2R+1)^3 fine cells around it; if the k-th best distance is
// not yet provably minimal it expands over coarse shells, pruning cells by box distance, until the
// bound holds — so results equal a kd-tree's (ties broken by lowest point index), including for far
// outliers, without ever touching a tree.  Distances use the reference's fp32 arithmetic
// ((dx*dx + dy*dy) + dz*dz, no FMA contraction) so the CPU oracle and the GPU agree bit for bit.
#include "handle.hpp"
#include "nn_device.hpp"
#include "sort.hpp"

namespace lsr {

namespace {

using namespace nnd;

// ---- build kernels -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nn_key_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ z, int n, float inv_cell, int o0, int o1, int o2,
                                                     int c0, int c1, unsigned int sentinel, unsigned int* __restrict__ key,
                                                     int* __restrict__ val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int k = sentinel;  // non-finite points: one past the last (coarse, fine) key, sorts last
  const float px = x[i], py = y[i], pz = z[i];
  if (isfinite(px) && isfinite(py) && isfinite(pz)) {
    const int fx = (int)floorf(px * inv_cell) - o0, fy = (int)floorf(py * inv_cell) - o1, fz = (int)floorf(pz * inv_cell) - o2;
    const unsigned int coarse = (unsigned int)((fx >> 3) + c0 * ((fy >> 3) + c1 * (fz >> 3)));
    k = coarse * FINE_PER_BLOCK + (unsigned int)((fx & 7) | ((fy & 7) << 3) | ((fz & 7) << 6));
  }
  key[i] = k;
  if (val) val[i] = i;   // (the hand-written sort numbers the values itself)
}

__global__ __launch_bounds__(256) void nn_gather_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ z, const int* __restrict__ order, int n,
                                                        float4* __restrict__ packed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = order[i];
  packed[i] = make_float4(x[o], y[o], z[o], __int_as_float(o));
}

__global__ __launch_bounds__(256) void nn_coarse_key_kernel(const unsigned int* __restrict__ key_sorted, int n, unsigned int sentinel,
                                                            unsigned int* __restrict__ ckey) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned int k = key_sorted[i];
  ckey[i] = (k == sentinel) ? 0xFFFFFFFFu : k / FINE_PER_BLOCK;
}

// one wave per occupied coarse cell: 513 fine-cell starts by counting (binary search over the sorted keys)
__global__ __launch_bounds__(256) void nn_fine_table_kernel(const unsigned int* __restrict__ key_sorted,
                                                            const unsigned int* __restrict__ run_key, const int* __restrict__ run_off,
                                                            const int* __restrict__ run_cnt, int n_runs,
                                                            int* __restrict__ coarse_block, int* __restrict__ block_off,
                                                            int* __restrict__ fine_start) {
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (b >= n_runs) return;
  const unsigned int ck = run_key[b];
  if (ck == 0xFFFFFFFFu) return;  // run of non-finite points (always last)
  const int beg = run_off[b], end = run_cnt ? beg + run_cnt[b] : run_off[b + 1];   // (sorted_runs_table closes with run_off[n_runs] = n)
  if (lane == 0) {
    coarse_block[ck] = b;
    block_off[b] = beg;
    block_off[b + 1] = end;  // the next block (if any) rewrites the same value
  }
  for (int f = lane; f <= FINE_PER_BLOCK; f += 64) {
    // first sorted position in [beg,end) whose fine id >= f
    const unsigned int want = ck * FINE_PER_BLOCK + (unsigned int)f;
    int lo = beg, hi = end;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (key_sorted[mid] < want) lo = mid + 1; else hi = mid;
    }
    fine_start[(size_t)b * FINE_STRIDE + f] = (f == FINE_PER_BLOCK) ? end : lo;
  }
}

// ---- bucket build (dense key spaces): histogram of the (coarse, fine) keys by integer atomics, exclusive scan, scatter through
// the same counters — no sort.  The order of the points INSIDE a fine cell is whatever the atomics made it; every consumer
// ranks candidates by (distance, original index), so results do not depend on it.
__global__ __launch_bounds__(256) void nnb_hist_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                       int n, float inv_cell, int o0, int o1, int o2, int c0, int c1, unsigned int sentinel,
                                                       unsigned int* __restrict__ key, int* __restrict__ hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int k = sentinel;
  const float px = x[i], py = y[i], pz = z[i];
  if (isfinite(px) && isfinite(py) && isfinite(pz)) {
    const int fx = (int)floorf(px * inv_cell) - o0, fy = (int)floorf(py * inv_cell) - o1, fz = (int)floorf(pz * inv_cell) - o2;
    const unsigned int coarse = (unsigned int)((fx >> 3) + c0 * ((fy >> 3) + c1 * (fz >> 3)));
    k = coarse * FINE_PER_BLOCK + (unsigned int)((fx & 7) | ((fy & 7) << 3) | ((fz & 7) << 6));
    atomicAdd(&hist[k], 1);
  }
  key[i] = k;
}

// position = start of the point's fine cell + what is left of the cell's counter (filled from the back)
__global__ __launch_bounds__(256) void nnb_scatter_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                          int n, const unsigned int* __restrict__ key, unsigned int sentinel,
                                                          const int* __restrict__ start, int* __restrict__ hist, float4* __restrict__ packed,
                                                          int* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned int k = key[i];
  if (k == sentinel) return;  // non-finite points are not part of the grid
  const int pos = start[k] + atomicSub(&hist[k], 1) - 1;
  packed[pos] = make_float4(x[i], y[i], z[i], __int_as_float(i));
  order[pos] = i;
}

__global__ __launch_bounds__(256) void nnb_flag_kernel(const int* __restrict__ start, int ccells, int* __restrict__ flag) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ccells) return;
  flag[c] = (start[(size_t)(c + 1) * FINE_PER_BLOCK] - start[(size_t)c * FINE_PER_BLOCK]) > 0 ? 1 : 0;
}

// one wave per coarse cell: block id (rank among the occupied cells, in cell order) and its 513 fine-cell starts
__global__ __launch_bounds__(256) void nnb_table_kernel(const int* __restrict__ start, const int* __restrict__ flag, const int* __restrict__ rank,
                                                        int ccells, int* __restrict__ coarse_block, int* __restrict__ block_off,
                                                        int* __restrict__ fine_start) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (c >= ccells) return;
  if (!flag[c]) {
    if (lane == 0) coarse_block[c] = -1;
    return;
  }
  const int b = rank[c];
  const int* st = start + (size_t)c * FINE_PER_BLOCK;
  if (lane == 0) {
    coarse_block[c] = b;
    block_off[b] = st[0];
    block_off[b + 1] = st[FINE_PER_BLOCK];  // the next block (if any) rewrites the same value
  }
  for (int f = lane; f <= FINE_PER_BLOCK; f += 64) fine_start[(size_t)b * FINE_STRIDE + f] = st[f];
}


// ---- neighbour grid as a REFINEMENT of the NDT voxel grid ------------------------------------------------------------------
// An NDT target built by the counting-sort builder (grid_dense.hip) already holds its points in voxel order.  With the fine
// cell edge = leaf / 8 and the origin = 8 * min_b the coarse cells of the neighbour grid ARE the NDT voxels — floor(p * (8 /
// leaf)) >> 3 == floor(p / leaf) exactly, scaling by 8 is exact in binary floating point — so the grid only has to order the
// points of every voxel by their fine cell: one workgroup per voxel, an LDS histogram over the 512 fine cells, no global
// histogram, no device-wide scan, no second pass over the unsorted cloud (round 2 built a fresh grid over the 661k-point
// window of every loop-closure candidate: ~125 us of kernels to answer one 30k-query fitness search).
struct RefineMember {
  const float *sx, *sy, *sz; const int* sidx; const unsigned int *start, *rank;
  int ncells; float inv_cell; int o0, o1, o2;
  int *coarse_block, *block_off, *fine_start; float4* packed; int* order;
};
struct RefineGroup { RefineMember m[LSR_GROUP]; };
constexpr int NN_REFINE_KEEP = 6;         // points a thread keeps in registers between the two passes over its voxel (1 536 per voxel)
constexpr int NN_REFINE_THREADS = 256;    // (1024 threads per voxel were measured slower on 64 x 661k-point windows: 258 vs 214 us per group of 16)
__device__ __forceinline__ void nn_refine_body(const RefineMember& M, const int cell) {
  __shared__ unsigned int s_cnt[FINE_PER_BLOCK], s_off[FINE_PER_BLOCK + 1], s_part[256];
  const int tid = threadIdx.x;
  const unsigned int beg = M.start[cell], end = M.start[cell + 1];
  if (end == beg) {
    if (tid == 0) M.coarse_block[cell] = -1;
    return;
  }
  const int b = (int)M.rank[cell];
  for (int k = tid; k < FINE_PER_BLOCK; k += NN_REFINE_THREADS) s_cnt[k] = 0u;
  __syncthreads();
  // A voxel of the 661k-point submap holds ~1 000 points: four or five per thread.  The first NN_REFINE_KEEP of a thread stay in
  // registers between the counting pass and the scatter (coordinates, original index, fine cell): the scatter reads nothing twice
  // (the second trip over the planes was a quarter of this kernel's traffic).  The atomics are issued in the same order as before —
  // per thread in ascending point index —, so the layout of a fine cell does not change.
  float kx[NN_REFINE_KEEP], ky[NN_REFINE_KEEP], kz[NN_REFINE_KEEP];
  int ki[NN_REFINE_KEEP], kf[NN_REFINE_KEEP];
#pragma unroll
  for (int u = 0; u < NN_REFINE_KEEP; u++) {
    const unsigned int j = beg + tid + u * NN_REFINE_THREADS;
    kf[u] = -1;
    if (j < end) {
      kx[u] = M.sx[j]; ky[u] = M.sy[j]; kz[u] = M.sz[j]; ki[u] = M.sidx[j];
      const int fx = (int)floorf(kx[u] * M.inv_cell) - M.o0, fy = (int)floorf(ky[u] * M.inv_cell) - M.o1, fz = (int)floorf(kz[u] * M.inv_cell) - M.o2;
      kf[u] = (fx & 7) | ((fy & 7) << 3) | ((fz & 7) << 6);
      atomicAdd(&s_cnt[kf[u]], 1u);
    }
  }
  for (unsigned int j = beg + tid + NN_REFINE_KEEP * NN_REFINE_THREADS; j < end; j += NN_REFINE_THREADS) {
    const int fx = (int)floorf(M.sx[j] * M.inv_cell) - M.o0, fy = (int)floorf(M.sy[j] * M.inv_cell) - M.o1, fz = (int)floorf(M.sz[j] * M.inv_cell) - M.o2;
    atomicAdd(&s_cnt[(fx & 7) | ((fy & 7) << 3) | ((fz & 7) << 6)], 1u);
  }
  __syncthreads();
  // exclusive scan of the 512 counts: two per thread, a wave scan over the 64 pair sums of a wave, the four wave totals through
  // LDS — one barrier (a Hillis-Steele scan over the 256 pair sums took seventeen)
  static_assert(NN_REFINE_THREADS == 256, "four waves, two counters per thread");
  const unsigned int c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
  unsigned int incl = c0 + c1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int v = __shfl_up(incl, d, 64);
    if ((tid & 63) >= d) incl += v;
  }
  if ((tid & 63) == 63) s_part[tid >> 6] = incl;
  __syncthreads();
  {
    unsigned int before = 0u;
    for (int v = 0; v < (tid >> 6); v++) before += s_part[v];
    incl += before;
  }
  {
    const unsigned int base = incl - (c0 + c1);
    s_off[2 * tid] = base;
    s_off[2 * tid + 1] = base + c0;
    if (tid == 255) s_off[FINE_PER_BLOCK] = incl;
    s_cnt[2 * tid] = 0u;       // becomes the cursor of the scatter
    s_cnt[2 * tid + 1] = 0u;
  }
  __syncthreads();
  for (int f = tid; f <= FINE_PER_BLOCK; f += NN_REFINE_THREADS) M.fine_start[(size_t)b * FINE_STRIDE + f] = (int)(beg + s_off[f]);
  if (tid == 0) {
    M.coarse_block[cell] = b;
    M.block_off[b] = (int)beg;
    M.block_off[b + 1] = (int)end;  // the next occupied voxel (if any) rewrites the same value
  }
#pragma unroll
  for (int u = 0; u < NN_REFINE_KEEP; u++) {
    if (kf[u] >= 0) {
      const unsigned int pos = beg + s_off[kf[u]] + atomicAdd(&s_cnt[kf[u]], 1u);
      M.packed[pos] = make_float4(kx[u], ky[u], kz[u], __int_as_float(ki[u]));
      M.order[pos] = ki[u];
    }
  }
  for (unsigned int j = beg + tid + NN_REFINE_KEEP * NN_REFINE_THREADS; j < end; j += NN_REFINE_THREADS) {
    const float px = M.sx[j], py = M.sy[j], pz = M.sz[j];
    const int fx = (int)floorf(px * M.inv_cell) - M.o0, fy = (int)floorf(py * M.inv_cell) - M.o1, fz = (int)floorf(pz * M.inv_cell) - M.o2;
    const int fine = (fx & 7) | ((fy & 7) << 3) | ((fz & 7) << 6);
    const unsigned int pos = beg + s_off[fine] + atomicAdd(&s_cnt[fine], 1u);
    const int oi = M.sidx[j];
    M.packed[pos] = make_float4(px, py, pz, __int_as_float(oi));
    M.order[pos] = oi;
  }
}
__global__ __launch_bounds__(NN_REFINE_THREADS) void nn_refine_group_kernel(const RefineGroup g) {
  const RefineMember& M = g.m[blockIdx.y];
  if ((int)blockIdx.x >= M.ncells) return;
  nn_refine_body(M, (int)blockIdx.x);
}

// ---- query kernels -------------------------------------------------------------------------------
// One thread per query walks fine shells 0..ring_cap; a query not proven by then (far-range scan points whose nearest
// target point is metres away) goes to `work` ([0] = count, [1..] = query indices) and is finished by nn1_coop_kernel,
// one wave per query — the per-thread walk of those few queries was ~85 % of this kernel's time on a 30k-point scan
// against a 660k-point submap (every wave waits for its slowest lane's hundreds of dependent cell probes).
// `spread`: only every spread-th lane carries a query (a 30k-point scan is fewer waves than the chip has SIMDs).
__global__ __launch_bounds__(NN_THREADS) void nn1_kernel(NNGridView G, const float* __restrict__ qx, const float* __restrict__ qy,
                                                         const float* __restrict__ qz, int n, const float* __restrict__ T16,
                                                         int fine_rings, int ring_cap, int spread, float max_d2, int* __restrict__ work,
                                                         int* __restrict__ idx, float* __restrict__ d2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / spread;
  if (i >= n || (t % spread) != 0) return;
  const float x = qx[i], y = qy[i], z = qz[i];
  float tx = x, ty = y, tz = z;
  if (T16) {
    tx = xform_rn(T16[0], T16[4], T16[8], T16[12], x, y, z);
    ty = xform_rn(T16[1], T16[5], T16[9], T16[13], x, y, z);
    tz = xform_rn(T16[2], T16[6], T16[10], T16[14], x, y, z);
  }
  Best1 c;
  c.init();
  if (!nn_query(G, tx, ty, tz, fine_rings, max_d2, c, -1, ring_cap)) {
    work[1 + atomicAdd(work, 1)] = i;
    return;
  }
  idx[i] = c.idx;
  d2[i] = c.d2;
}

// The same first stage on four lanes per query (nn1_query_quad): a quarter of the dependent-load chain per lane.
__device__ __forceinline__ void nn1_quad_body(const NNGridView& G, const float* __restrict__ qx, const float* __restrict__ qy,
                                              const float* __restrict__ qz, int n, const float* __restrict__ T16, int fine_rings, int ring_cap,
                                              float max_d2, int* __restrict__ work, int* __restrict__ idx, float* __restrict__ d2, const int t) {
  const int i = t >> 2, sub = t & 3;
  if (i >= n) return;   // n * 4 threads: a quad is never split by this test
  const float x = qx[i], y = qy[i], z = qz[i];
  float tx = x, ty = y, tz = z;
  if (T16) {
    tx = xform_rn(T16[0], T16[4], T16[8], T16[12], x, y, z);
    ty = xform_rn(T16[1], T16[5], T16[9], T16[13], x, y, z);
    tz = xform_rn(T16[2], T16[6], T16[10], T16[14], x, y, z);
  }
  Best1 c;
  c.init();
  const bool proven = nn1_query_quad(G, tx, ty, tz, fine_rings, max_d2, c, ring_cap, sub);
  if (sub != 0) return;
  if (!proven) {
    work[1 + atomicAdd(work, 1)] = i;
    return;
  }
  idx[i] = c.idx;
  d2[i] = c.d2;
}
__global__ __launch_bounds__(NN_THREADS) void nn1_quad_kernel(NNGridView G, const float* __restrict__ qx, const float* __restrict__ qy,
                                                              const float* __restrict__ qz, int n, const float* __restrict__ T16,
                                                              int fine_rings, int ring_cap, float max_d2, int* __restrict__ work,
                                                              int* __restrict__ idx, float* __restrict__ d2) {
  nn1_quad_body(G, qx, qy, qz, n, T16, fine_rings, ring_cap, max_d2, work, idx, d2, blockIdx.x * blockDim.x + threadIdx.x);
}

// One wave per query (coop_search over the fine grid, coarse cells when needed): the form used for scan-sized query sets.
__device__ __forceinline__ void nn1_wave_body(const NNGridView& G, const float* __restrict__ qx, const float* __restrict__ qy,
                                              const float* __restrict__ qz, int n, const float* __restrict__ T16, int fine_rings,
                                              float max_d2, int* __restrict__ idx, float* __restrict__ d2, const int wave, const int n_waves) {
  const int lane = threadIdx.x & 63;
  for (int i = wave; i < n; i += n_waves) {
    const float x = qx[i], y = qy[i], z = qz[i];
    float tx = x, ty = y, tz = z;
    if (T16) {
      tx = xform_rn(T16[0], T16[4], T16[8], T16[12], x, y, z);
      ty = xform_rn(T16[1], T16[5], T16[9], T16[13], x, y, z);
      tz = xform_rn(T16[2], T16[6], T16[10], T16[14], x, y, z);
    }
    CoopList mine;
    mine.d = INFINITY;
    mine.i = INT_MAX;
    coop_search<true>(G, tx, ty, tz, 1, fine_rings, max_d2, -1, mine);
    if (lane == 0) {
      const bool found = mine.i != INT_MAX;
      idx[i] = found ? mine.i : -1;
      d2[i] = found ? mine.d : INFINITY;
    }
  }
}
__global__ __launch_bounds__(256) void nn1_wave_kernel(NNGridView G, const float* __restrict__ qx, const float* __restrict__ qy,
                                                       const float* __restrict__ qz, int n, const float* __restrict__ T16, int fine_rings,
                                                       float max_d2, int* __restrict__ idx, float* __restrict__ d2) {
  nn1_wave_body(G, qx, qy, qz, n, T16, fine_rings, max_d2, idx, d2, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, (gridDim.x * blockDim.x) >> 6);
}

// getFitnessScore of a GROUP of candidates: blockIdx.y selects the member, whose grid view, clouds and final transformation
// travel in the kernel arguments (no per-member upload of the 4x4)
struct FitMember {
  NNGridView G;
  const float *qx, *qy, *qz; int n, blocks;
  float T16[16];
  float max_d2; double max_range;
  int* idx; float* d2; double* part; int* work;
  BuildMailbox* mb; unsigned int token; int empty; int fine_rings;
  int ball_cells;   // widest ball (fine cells per axis) the sixteen-lane search reads itself; wider ones go to the work list
};
constexpr int FIT_GROUP = 12;
constexpr int NN_FITNESS_FINE_RINGS = 0;   // measured on 64 candidate windows (fitness stage): 2.54 ms with 1, 2.39 ms with 0
struct FitGroup { FitMember m[FIT_GROUP]; };
static_assert(sizeof(FitGroup) <= 3800, "a group's parameters must fit the kernel argument segment");
__global__ __launch_bounds__(256) void nn1_wave_group_kernel(const FitGroup g) {
  const FitMember& M = g.m[blockIdx.y];
  if ((int)blockIdx.x >= M.blocks || M.empty) return;
  nn1_wave_body(M.G, M.qx, M.qy, M.qz, M.n, M.T16, M.fine_rings, M.max_d2, M.idx, M.d2, (blockIdx.x * 256 + threadIdx.x) >> 6, (M.blocks * 256) >> 6);
}

// tail of nn1_kernel: one wave per deferred query; same (distance, index) order, same fp32 distances => same answer
__device__ __forceinline__ void nn1_coop_body(const NNGridView& G, const float* __restrict__ qx, const float* __restrict__ qy,
                                              const float* __restrict__ qz, const float* __restrict__ T16, float max_d2,
                                              const int* __restrict__ work, int* __restrict__ idx, float* __restrict__ d2, const int wave,
                                              const int n_waves) {
  const int lane = threadIdx.x & 63;
  const int n_work = work[0];
  for (int w = wave; w < n_work; w += n_waves) {
    const int i = work[1 + w];
    const float x = qx[i], y = qy[i], z = qz[i];
    float tx = x, ty = y, tz = z;
    if (T16) {
      tx = xform_rn(T16[0], T16[4], T16[8], T16[12], x, y, z);
      ty = xform_rn(T16[1], T16[5], T16[9], T16[13], x, y, z);
      tz = xform_rn(T16[2], T16[6], T16[10], T16[14], x, y, z);
    }
    CoopList mine;
    coop_knn(G, tx, ty, tz, 1, max_d2, -1, mine);
    if (lane == 0) {
      const bool found = mine.i != INT_MAX;
      idx[i] = found ? mine.i : -1;
      d2[i] = found ? mine.d : INFINITY;
    }
  }
}
__global__ __launch_bounds__(256) void nn1_coop_kernel(NNGridView G, const float* __restrict__ qx, const float* __restrict__ qy,
                                                       const float* __restrict__ qz, const float* __restrict__ T16, float max_d2,
                                                       const int* __restrict__ work, int* __restrict__ idx, float* __restrict__ d2) {
  nn1_coop_body(G, qx, qy, qz, T16, max_d2, work, idx, d2, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, (gridDim.x * blockDim.x) >> 6);
}

__global__ __launch_bounds__(NN_THREADS) void knn_kernel(NNGridView G, const float* __restrict__ qx, const float* __restrict__ qy,
                                                         const float* __restrict__ qz, int n, int k, int fine_rings,
                                                         int* __restrict__ idx, float* __restrict__ d2) {
  extern __shared__ unsigned char smem[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  BestK c;
  c.init(smem, threadIdx.x, k);
  if (i < n) {
    nn_query(G, qx[i], qy[i], qz[i], fine_rings, INFINITY, c, -1);
    c.finalize();
    for (int s = 0; s < k; s++) {
      idx[(size_t)i * k + s] = c.index(s);
      d2[(size_t)i * k + s] = c.dist(s);
    }
  }
}

// deterministic two-stage reduction of {sum d2, count} over pairs with d2 <= max_range
__device__ __forceinline__ void fitness_partial_body(const int* __restrict__ idx, const float* __restrict__ d2, int n, double max_range,
                                                     double* __restrict__ part, const int blk, const int nblk) {
  __shared__ double s_sum[256], s_cnt[256];
  double sum = 0, cnt = 0;
  for (int i = blk * 256 + threadIdx.x; i < n; i += nblk * 256) {
    if (idx[i] >= 0 && (double)d2[i] <= max_range) { sum += (double)d2[i]; cnt += 1.0; }
  }
  s_sum[threadIdx.x] = sum;
  s_cnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { s_sum[threadIdx.x] += s_sum[threadIdx.x + s]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blk] = s_sum[0]; part[2 * blk + 1] = s_cnt[0]; }
}
__global__ __launch_bounds__(256) void fitness_partial_kernel(const int* __restrict__ idx, const float* __restrict__ d2, int n,
                                                              double max_range, double* __restrict__ part) {
  fitness_partial_body(idx, d2, n, max_range, part, (int)blockIdx.x, (int)gridDim.x);
}

// second stage (one workgroup, fixed order): {sum, count} of the 256 partials straight into the host mailbox
__device__ __forceinline__ void fitness_final_body(const double* __restrict__ part, BuildMailbox* __restrict__ mb, unsigned int token) {
  __shared__ double s_sum[256], s_cnt[256];
  s_sum[threadIdx.x] = part[2 * threadIdx.x];
  s_cnt[threadIdx.x] = part[2 * threadIdx.x + 1];
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0, cnt = 0;
    for (int b = 0; b < 256; b++) { sum += s_sum[b]; cnt += s_cnt[b]; }   // the order the host used to add them in
    mb->fit_sum = sum;
    mb->fit_cnt = cnt;
    __threadfence_system();
    __hip_atomic_store(&mb->fit_token, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(256) void fitness_final_kernel(const double* __restrict__ part, BuildMailbox* __restrict__ mb, unsigned int token) {
  fitness_final_body(part, mb, token);
}
// alternative form (LSR_FIT_GROUP_FORM=1): capped walk on four lanes per query + wave-cooperative tail for the queries it defers —
// the same answers, and on paper a sixth of the instructions per query once millions of queries fill the chip; measured
// slower than one wave per query (the deferred tail alone costs as much as it saves)
__global__ void fit_zero_work_group_kernel(const FitGroup g) { g.m[threadIdx.x].work[0] = 0; }   // empty deferred-query lists

// getFitnessScore's search on SIXTEEN lanes per query, seeded by the query's own fine cell (round 4).
// A registered scan lies on the submap: the nearest target point of a query is almost always a few centimetres away, and the
// fine cell the query falls into (leaf / 8 = 0.625 m at the reference's resolution) almost always holds a point.  So:
//   1. the group scans the query's OWN cell (one dependent lookup: coarse map -> fine table -> the cell's points) — if it is empty,
//      the 3 x 3 x 3 cells around it — and keeps the best (distance, index): a real point, hence an upper bound d on the answer;
//   2. every point at distance <= d lies in a fine cell that touches the ball of radius d around the query (the cell index is a
//      monotone map; the reach is padded against rounding: ball_cell_range): those cells — one to eight for a registered scan, not the
//      27 of a shell — are ALL that is left to read, one lane per (row, coarse segment), candidates laid end to end, 16 per round
//      (the structure of gicp_corr_ball_kernel, whose seed is the previous outer iteration's neighbour);
//   3. a query with no point within a cell of it, or whose ball spans more than FIT_BALL_CELLS cells on an axis, goes on the
//      member's work list for the general one-wave-per-query search (nn1_list_group_kernel): exact all the same.
// Same candidates compared in the same total order (distance, index), same fp32 distances => the answer of every other search
// form, bit for bit (tests/test_nn_gpu.py, test_loop_closure_gpu.py hold them to each other and to the oracle).
constexpr int FIT_BALL_CELLS = 5;
__device__ __forceinline__ void nn1_ball_body(const NNGridView& G, const float* __restrict__ qx, const float* __restrict__ qy,
                                              const float* __restrict__ qz, int n, const float* __restrict__ T16, int* __restrict__ work,
                                              int* __restrict__ idx, float* __restrict__ d2, const int t, const int ball_cells) {
  const int i = t >> 4, gl = t & 15;
  if (i >= n) return;   // n * 16 threads: a group is never split by this test
  const float x = qx[i], y = qy[i], z = qz[i];
  const float q[3] = {xform_rn(T16[0], T16[4], T16[8], T16[12], x, y, z), xform_rn(T16[1], T16[5], T16[9], T16[13], x, y, z),
                      xform_rn(T16[2], T16[6], T16[10], T16[14], x, y, z)};
  if (!(isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]))) {   // no neighbour, as every other form answers
    if (gl == 0) { idx[i] = -1; d2[i] = INFINITY; }
    return;
  }
  float bd = INFINITY;
  int bi = INT_MAX;
  bool general = false;
  // ---- 1. a seed: the best point of the query's own fine cell; if that cell is empty, of the 3 x 3 x 3 cells around it
  const float ff[3] = {floorf(q[0] * G.inv_cell), floorf(q[1] * G.inv_cell), floorf(q[2] * G.inv_cell)};
  int fq[3] = {0, 0, 0};
  if (!(fabsf(ff[0]) < 1.0e9f && fabsf(ff[1]) < 1.0e9f && fabsf(ff[2]) < 1.0e9f)) general = true;
  bool inside = !general;   // the query's own cell is a cell of the grid
  if (!general) {
    for (int a = 0; a < 3; a++) {
      fq[a] = (int)ff[a] - G.org[a];
      if (fq[a] < -1 || fq[a] > G.cdim[a] * 8) general = true;   // more than a cell outside the grid: the general search
      if (fq[a] < 0 || fq[a] >= G.cdim[a] * 8) inside = false;
    }
  }
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  bool own_cell_done = false;
  if (!general) {
    if (inside) scan_cell_group16(G, q, fq, gl, bd, bi);
    own_cell_done = (bi != INT_MAX);
    if (bi == INT_MAX) {
      for (int a = 0; a < 3; a++) { lo[a] = max(fq[a] - 1, 0); hi[a] = min(fq[a] + 1, G.cdim[a] * 8 - 1); }
      scan_cells_group16(G, q, lo, hi, gl, bd, bi);
    }
    if (bi == INT_MAX) general = true;   // nothing within a cell of the query: rare for a registered scan
  }
  // ---- 2. the cells of the ball of radius sqrt(bd) around the query hold every point that could beat the seed
  if (!general && !ball_cell_range(G, q, bd, ball_cells, lo, hi)) general = true;
  if (general) {
    if (gl == 0) work[1 + atomicAdd(work, 1)] = i;
    return;
  }
  // a ball that stays inside what step 1 has read — the query's own cell, or the 3 x 3 x 3 cells when that one was empty — needs
  // no second reading: the seed is the answer
  const int reach = own_cell_done ? 0 : 1;
  bool covered = true;
  for (int a = 0; a < 3; a++) covered = covered && lo[a] >= fq[a] - reach && hi[a] <= fq[a] + reach;
  if (!covered) scan_cells_group16(G, q, lo, hi, gl, bd, bi, own_cell_done ? fq : nullptr);   // (the own cell has been offered)
  if (gl == 0) { idx[i] = bi; d2[i] = bd; }
}
__global__ __launch_bounds__(256) void nn1_list_group_kernel(const FitGroup g) {
  const FitMember& M = g.m[blockIdx.y];
  if (M.empty) return;
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
  const int n_work = M.work[0];
  for (int w = wave; w < n_work; w += n_waves) {
    const int i = M.work[1 + w];
    const float x = M.qx[i], y = M.qy[i], z = M.qz[i];
    const float tx = xform_rn(M.T16[0], M.T16[4], M.T16[8], M.T16[12], x, y, z), ty = xform_rn(M.T16[1], M.T16[5], M.T16[9], M.T16[13], x, y, z),
                tz = xform_rn(M.T16[2], M.T16[6], M.T16[10], M.T16[14], x, y, z);
    CoopList mine;
    mine.d = INFINITY;
    mine.i = INT_MAX;
    coop_search<true>(M.G, tx, ty, tz, 1, M.fine_rings, M.max_d2, -1, mine);
    if (lane == 0) {
      const bool found = mine.i != INT_MAX;
      M.idx[i] = found ? mine.i : -1;
      M.d2[i] = found ? mine.d : INFINITY;
    }
  }
}
__global__ __launch_bounds__(256) void nn1_ball_group_kernel(const FitGroup g) {
  const FitMember& M = g.m[blockIdx.y];
  if (M.empty) return;
  nn1_ball_body(M.G, M.qx, M.qy, M.qz, M.n, M.T16, M.work, M.idx, M.d2, blockIdx.x * 256 + threadIdx.x, M.ball_cells);
}
__global__ __launch_bounds__(NN_THREADS) void nn1_quad_group_kernel(const FitGroup g) {
  const FitMember& M = g.m[blockIdx.y];
  if (M.empty) return;
  nn1_quad_body(M.G, M.qx, M.qy, M.qz, M.n, M.T16, 1, 2, M.max_d2, M.work, M.idx, M.d2, blockIdx.x * NN_THREADS + threadIdx.x);
}
__global__ __launch_bounds__(256) void nn1_coop_group_kernel(const FitGroup g) {
  const FitMember& M = g.m[blockIdx.y];
  if (M.empty) return;
  nn1_coop_body(M.G, M.qx, M.qy, M.qz, M.T16, M.max_d2, M.work, M.idx, M.d2, (blockIdx.x * 256 + threadIdx.x) >> 6, (gridDim.x * 256) >> 6);
}
__global__ __launch_bounds__(256) void fitness_partial_group_kernel(const FitGroup g) {
  const FitMember& M = g.m[blockIdx.y];
  fitness_partial_body(M.idx, M.d2, M.empty ? 0 : M.n, M.max_range, M.part, (int)blockIdx.x, 256);
}
__global__ __launch_bounds__(256) void fitness_final_group_kernel(const FitGroup g) {
  const FitMember& M = g.m[blockIdx.x];
  fitness_final_body(M.part, M.mb, M.token);
}

}  // namespace

// ---- N2: submap assembly — pcl::transformPointCloud per keyframe + concatenation, on the device ----------
// scanmatcher/src/scanmatcher_component.cpp:449-464 (frontend target = newest frame + previous submaps, each
// moved by its pose); graph_based_slam/src/graph_based_slam_component.cpp:208-222 (loop candidate window).
// fp32 arithmetic in the reference's order ((m00*x + m01*y) + m02*z) + m03, no FMA contraction (this TU).
namespace {
__global__ __launch_bounds__(256) void transform_append_kernel(const unsigned char* __restrict__ aos, size_t stride, int n,
                                                               const float* __restrict__ T16, float* __restrict__ ox,
                                                               float* __restrict__ oy, float* __restrict__ oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = (const float*)(aos + (size_t)i * stride);
  const float x = p[0], y = p[1], z = p[2];
  ox[i] = xform_rn(T16[0], T16[4], T16[8], T16[12], x, y, z);
  oy[i] = xform_rn(T16[1], T16[5], T16[9], T16[13], x, y, z);
  oz[i] = xform_rn(T16[2], T16[6], T16[10], T16[14], x, y, z);
}
}  // namespace

int transform_append(const void* d_aos, size_t stride_bytes, size_t n, const float* d_T16, DeviceCloud& out, size_t offset,
                     hipStream_t stream) {
  if (n == 0) return LSR_OK;
  hipLaunchKernelGGL(transform_append_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     (const unsigned char*)d_aos, stride_bytes, (int)n, d_T16, out.x() + offset, out.y() + offset, out.z() + offset);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

float nn_pick_cell(size_t, const lsr_handle_s*) { return 0.5f; }

// LSR_NN_SORT=rocprim: rounds 1-5's rocPRIM sort / run_length_encode / scans as the A/B cross-check of the hand-written ones
static bool nn_use_rocprim() {
  static const bool v = [] { const char* e = getenv("LSR_NN_SORT"); return e && strcmp(e, "rocprim") == 0; }();
  return v;
}

int nn_build_hash(const DeviceCloud& cloud, float cell, HashGridDev& grid, BuildScratch& sc, hipStream_t stream) {
  const int n = (int)cloud.n;
  grid.cell = cell;
  grid.n = cloud.n;
  grid.n_blocks = 0;
  for (int k = 0; k < 3; k++) { grid.org[k] = 0; grid.cdim[k] = 0; }
  if (n == 0) return LSR_OK;
  float mn[3], mx[3];
  unsigned int n_finite = 0;
  int st = cloud_bbox(cloud, mn, mx, &n_finite, sc, stream);
  if (st) return st;
  if (n_finite == 0) return LSR_OK;
  const float inv = 1.0f / cell;
  size_t ccells = 1;
  for (int k = 0; k < 3; k++) {
    const int f0 = (int)floorf(mn[k] * inv), f1 = (int)floorf(mx[k] * inv);
    const int o = (f0 >= 0) ? (f0 & ~7) : -(((-f0) + 7) & ~7);  // origin aligned down to a multiple of 8
    grid.org[k] = o;
    grid.cdim[k] = ((f1 - o) >> 3) + 1;
    ccells *= (size_t)grid.cdim[k];
  }
  if (ccells * FINE_PER_BLOCK >= 0xFFFFFFFFull) {
    set_last_error("cloud extent too large for the NN grid at this cell size");
    return LSR_ERR_INDEX_OVERFLOW;
  }
  const size_t nkeys = ccells * FINE_PER_BLOCK;
  if (nkeys <= NN_BUCKET_MAX_KEYS && !sc.force_sort_path) {
    // ---- bucket build: no sort, no host round trip after the bounding box
    const size_t blocks_cap = std::min(ccells, (size_t)n);  // occupied coarse cells <= points
    if ((st = sc.words.reserve(32 + (size_t)n + 2 * (nkeys + 1) + 2 * ccells + 64))) return st;
    unsigned int* key = sc.words.p + 32;
    int* hist = (int*)(key + n);
    int* start = hist + (nkeys + 1);
    int* flag = start + (nkeys + 1);
    int* rank = flag + ccells;
    if ((st = grid.order.reserve(n))) return st;
    if ((st = grid.packed.reserve(n))) return st;
    if ((st = grid.coarse_block.reserve(ccells))) return st;
    if ((st = grid.block_off.reserve(blocks_cap + 1))) return st;
    if ((st = grid.fine_start.reserve(blocks_cap * FINE_STRIDE))) return st;
    const int nbk = (n + 255) / 256;
    const unsigned int sentinel_b = (unsigned int)nkeys;
    LSR_HIP(hipMemsetAsync(hist, 0, (nkeys + 1) * sizeof(int), stream));
    hipLaunchKernelGGL(nnb_hist_kernel, dim3(nbk), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), n, inv, grid.org[0], grid.org[1],
                       grid.org[2], grid.cdim[0], grid.cdim[1], sentinel_b, key, hist);
    // device-wide exclusive scans: hand-written (lsd_sort.hip: block sums, one-workgroup scan of the sums, block scans); rocPRIM's
    // behind LSR_NN_SORT=rocprim as the A/B cross-check
    if ((st = nn_use_rocprim() ? exclusive_scan_i32(hist, start, nkeys + 1, sc.temp, stream) : exclusive_scan_i32_lsd(hist, start, nkeys + 1, sc.temp, stream))) return st;
    hipLaunchKernelGGL(nnb_flag_kernel, dim3((unsigned)((ccells + 255) / 256)), dim3(256), 0, stream, start, (int)ccells, flag);
    if ((st = nn_use_rocprim() ? exclusive_scan_i32(flag, rank, ccells, sc.temp, stream) : exclusive_scan_i32_lsd(flag, rank, ccells, sc.temp, stream))) return st;
    hipLaunchKernelGGL(nnb_scatter_kernel, dim3(nbk), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), n, key, sentinel_b, start, hist,
                       grid.packed.p, grid.order.p);
    hipLaunchKernelGGL(nnb_table_kernel, dim3((unsigned)((ccells + 3) / 4)), dim3(256), 0, stream, start, flag, rank, (int)ccells,
                       grid.coarse_block.p, grid.block_off.p, grid.fine_start.p);
    LSR_HIP(hipGetLastError());
    grid.n_blocks = 1;  // "not empty" (n_finite > 0); the exact count stays on the device, nobody on the host needs it
    return LSR_OK;
  }
  // scratch: key_in | key_out | val_in | ckey | run_key (n+1) | run_cnt (n+1) | run_off (n+1) | nruns | block_heads | block_base
  const size_t nbr = sorted_runs_blocks((size_t)n);
  if ((st = sc.words.reserve(32 + 7 * (size_t)n + 3 + 16 + 2 * nbr))) return st;
  unsigned int* key_in = sc.words.p + 32;
  unsigned int* key_out = key_in + n;
  int* val_in = (int*)(key_out + n);
  unsigned int* ckey = (unsigned int*)(val_in + n);
  unsigned int* run_key = ckey + n;
  int* run_cnt = (int*)(run_key + n + 1);
  int* run_off = run_cnt + n + 1;
  int* d_nruns = run_off + n + 1;
  int* block_heads = d_nruns + 16;
  int* block_base = block_heads + nbr;
  if ((st = grid.order.reserve(n))) return st;
  if ((st = grid.packed.reserve(n))) return st;
  if ((st = grid.coarse_block.reserve(ccells))) return st;
  LSR_HIP(hipMemsetAsync(grid.coarse_block.p, 0xFF, ccells * sizeof(int), stream));
  const int nb = (n + 255) / 256;
  const unsigned int sentinel = (unsigned int)(ccells * FINE_PER_BLOCK);
  const bool rocprim_path = nn_use_rocprim();
  hipLaunchKernelGGL(nn_key_kernel, dim3(nb), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), n, inv, grid.org[0],
                     grid.org[1], grid.org[2], grid.cdim[0], grid.cdim[1], sentinel, key_in, rocprim_path ? val_in : (int*)nullptr);
  int bits = 1;
  while (bits < 32 && (sentinel >> bits) != 0) bits++;  // only the bits the keys can use are sorted
  const unsigned int* ks = key_out;
  if (rocprim_path) {
    if ((st = sort_pairs_u32(key_in, key_out, val_in, grid.order.p, n, bits, sc.temp, stream))) return st;
  } else {
    // hand-written stable LSD radix sort (lsd_sort.hip); the order lands in grid.order or in val_in, whichever the pass count says
    bool in_b = false;
    if ((st = sort_pairs_u32_lsd(key_in, key_out, nullptr, val_in, grid.order.p, (size_t)n, bits, sc.temp, stream, &in_b))) return st;
    ks = in_b ? key_out : key_in;
    if (!in_b) LSR_HIP(hipMemcpyAsync(grid.order.p, val_in, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, stream));
  }
  hipLaunchKernelGGL(nn_gather_kernel, dim3(nb), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), grid.order.p, n,
                     grid.packed.p);
  hipLaunchKernelGGL(nn_coarse_key_kernel, dim3(nb), dim3(256), 0, stream, ks, n, sentinel, ckey);
  // the number of occupied coarse cells reaches the host through the mailbox (one polled word)
  int n_runs = 0;
  const int* counts = run_cnt;
  if (rocprim_path) {
    if ((st = run_length_encode_u32(ckey, n, run_key, run_cnt, d_nruns, sc.temp, stream))) return st;
    if ((st = publish_device_int(d_nruns, sc, stream, &n_runs))) return st;
    if ((st = exclusive_scan_i32(run_cnt, run_off, n_runs, sc.temp, stream))) return st;
  } else {
    unsigned int rtoken = 0;
    if ((st = sorted_runs_begin(ckey, (size_t)n, block_heads, block_base, sc, stream, &rtoken))) return st;
    if ((st = sorted_runs_table(ckey, (size_t)n, block_base, run_key, run_off, stream))) return st;
    if ((st = sorted_runs_count(sc, stream, rtoken, &n_runs))) return st;
    counts = nullptr;
  }
  if ((st = grid.block_off.reserve((size_t)n_runs + 1))) return st;
  if ((st = grid.fine_start.reserve((size_t)n_runs * FINE_STRIDE))) return st;
  hipLaunchKernelGGL(nn_fine_table_kernel, dim3((n_runs + 3) / 4), dim3(256), 0, stream, ks, run_key, run_off, counts,
                     n_runs, grid.coarse_block.p, grid.block_off.p, grid.fine_start.p);
  LSR_HIP(hipGetLastError());
  // no synchronisation here: every consumer of the grid (and of the scratch buffers) is ordered on the same stream
  grid.n_blocks = n_runs;  // a trailing run of non-finite points (if any) is never referenced by coarse_block
  return LSR_OK;
}


// Neighbour grids of `count` NDT targets from the voxel order their grid builder left behind (VoxelGridDev::has_sorted): one
// launch per group of LSR_GROUP members, nothing to wait for.
int nn_build_hash_from_grids(const VoxelGridDev* const* vgrids, HashGridDev* const* grids, int count, hipStream_t stream) {
  int st;
  for (int g0 = 0; g0 < count; g0 += LSR_GROUP) {
    RefineGroup grp;
    std::memset(&grp, 0, sizeof(grp));
    const int ng = std::min(LSR_GROUP, count - g0);
    int max_cells = 0;
    for (int k = 0; k < ng; k++) {
      const VoxelGridDev& V = *vgrids[g0 + k];
      HashGridDev& G = *grids[g0 + k];
      if (!V.has_sorted) { set_last_error("the voxel grid holds no cell-ordered points"); return LSR_ERR_INVALID_ARGUMENT; }
      const size_t n = V.sorted_n, ccells = V.ncells;
      G.cell = V.leaf / 8.0f;          // exact; 1 / cell == 8 * (1 / leaf) to the last bit
      G.n = n;
      for (int a = 0; a < 3; a++) { G.org[a] = 8 * V.min_b[a]; G.cdim[a] = V.div_b[a]; }
      const size_t blocks_cap = std::min(ccells, n);
      if ((st = G.order.reserve(n))) return st;
      if ((st = G.packed.reserve(n))) return st;
      if ((st = G.coarse_block.reserve(ccells))) return st;
      if ((st = G.block_off.reserve(blocks_cap + 1))) return st;
      if ((st = G.fine_start.reserve(blocks_cap * FINE_STRIDE))) return st;
      G.n_blocks = 1;  // "not empty": a grid with ncells > 0 holds at least one finite point
      RefineMember& M = grp.m[k];
      M.sx = V.sorted.p; M.sy = M.sx + V.sorted_pitch; M.sz = M.sy + V.sorted_pitch;
      M.sidx = V.sorted_idx.p; M.start = V.cell_start.p; M.rank = V.cell_rank.p;
      M.ncells = (int)ccells; M.inv_cell = 1.0f / G.cell;
      M.o0 = G.org[0]; M.o1 = G.org[1]; M.o2 = G.org[2];
      M.coarse_block = G.coarse_block.p; M.block_off = G.block_off.p; M.fine_start = G.fine_start.p; M.packed = G.packed.p; M.order = G.order.p;
      max_cells = std::max(max_cells, (int)ccells);
    }
    if (max_cells > 0) hipLaunchKernelGGL(nn_refine_group_kernel, dim3(max_cells, ng), dim3(NN_REFINE_THREADS), 0, stream, grp);
  }
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

bool nn_coop_enabled() {
  static const bool on = [] { const char* e = getenv("LSR_NN_COOP"); return !(e && e[0] == '0'); }();
  return on;
}

// Shells the fitness search reads before it tests the shell bound for the first time (env LSR_NN_FINE_RINGS, read once): 0 lets a
// query whose own cell already proves its neighbour stop there.
static int nn_fitness_fine_rings() {
  static const int v = [] { const char* e = getenv("LSR_NN_FINE_RINGS"); const int r = e ? atoi(e) : NN_FITNESS_FINE_RINGS; return r < 0 ? 0 : r > 4 ? 4 : r; }();
  return v;
}

int nn_search_device(const DeviceCloud& q, const float* d_T16, const HashGridDev& grid, int fine_rings, float max_d2,
                     int* d_idx, float* d_d2, hipStream_t stream, int* d_work) {
  const int n = (int)q.n;
  if (n == 0) return LSR_OK;
  if (grid.n_blocks == 0) {
    LSR_HIP(hipMemsetAsync(d_idx, 0xFF, sizeof(int) * n, stream));
    LSR_HIP(hipMemsetAsync(d_d2, 0x7F, sizeof(float) * n, stream));  // 0x7F7F7F7F ~ 3.4e38
    return LSR_OK;
  }
  // d_work (n + 1 ints) given: two-stage search — per-thread walk capped at two fine shells, wave-cooperative tail
  const int ring_cap = d_work ? 2 : -1;
  const int spread = (d_work && n <= 65536) ? 2 : 1;
  if (d_work && n <= 262144 && nn_coop_enabled()) {   // small query sets (a scan): one wave per query
    hipLaunchKernelGGL(nn1_wave_kernel, dim3((unsigned)(((long)n * 64 + 255) / 256)), dim3(256), 0, stream, make_view(grid), q.x(), q.y(),
                       q.z(), n, d_T16, fine_rings, max_d2, d_idx, d_d2);
    LSR_HIP(hipGetLastError());
    return LSR_OK;
  }
  if (d_work) LSR_HIP(hipMemsetAsync(d_work, 0, sizeof(int), stream));
  if (d_work && n <= 262144) {   // small query sets (a scan): four lanes per query
    const long threads = (long)n * 4;
    hipLaunchKernelGGL(nn1_quad_kernel, dim3((unsigned)((threads + NN_THREADS - 1) / NN_THREADS)), dim3(NN_THREADS), 0, stream,
                       make_view(grid), q.x(), q.y(), q.z(), n, d_T16, fine_rings, ring_cap, max_d2, d_work, d_idx, d_d2);
  } else {
    const long threads = (long)n * spread;
    hipLaunchKernelGGL(nn1_kernel, dim3((unsigned)((threads + NN_THREADS - 1) / NN_THREADS)), dim3(NN_THREADS), 0, stream, make_view(grid),
                       q.x(), q.y(), q.z(), n, d_T16, fine_rings, ring_cap, spread, max_d2, d_work, d_idx, d_d2);
  }
  if (d_work)
    hipLaunchKernelGGL(nn1_coop_kernel, dim3(1024), dim3(256), 0, stream, make_view(grid), q.x(), q.y(), q.z(), d_T16, max_d2, d_work,
                       d_idx, d_d2);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

int knn_search_device(const DeviceCloud& q, const HashGridDev& grid, int k, int fine_rings, int* d_idx, float* d_d2,
                      hipStream_t stream) {
  const int n = (int)q.n;
  if (n == 0) return LSR_OK;
  const size_t smem = BestK::lds_bytes(k);
  hipLaunchKernelGGL(knn_kernel, dim3((n + NN_THREADS - 1) / NN_THREADS), dim3(NN_THREADS), smem, stream, make_view(grid), q.x(),
                     q.y(), q.z(), n, k, fine_rings, d_idx, d_d2);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

static int nn_scratch(BuildScratch& sc, size_t n, int** d_idx, float** d_d2, double** d_part, int** d_work) {
  int st = sc.words.reserve(32 + 3 * n + 32);
  if (st) return st;
  *d_idx = (int*)(sc.words.p + 32);
  *d_d2 = (float*)(sc.words.p + 32 + n);
  *d_work = (int*)(sc.words.p + 32 + 2 * n);   // [0] = deferred count, [1..n] = deferred query indices
  if ((st = sc.sums.reserve(2 * 256 + 8))) return st;
  *d_part = sc.sums.p;
  return LSR_OK;
}

int nn_search_host(const DeviceCloud& source, const float* T16_host, const HashGridDev& grid, int32_t* idx, float* d2,
                   BuildScratch& sc, DevBuf<float>& d_T16, hipStream_t stream) {
  int* d_idx; float* d_d2; double* d_part; int* d_work;
  int st = nn_scratch(sc, source.n, &d_idx, &d_d2, &d_part, &d_work);
  if (st) return st;
  LSR_HIP(hipMemcpyAsync(d_T16.p, T16_host, 16 * sizeof(float), hipMemcpyHostToDevice, stream));
  if ((st = nn_search_device(source, d_T16.p, grid, 1, INFINITY, d_idx, d_d2, stream, d_work))) return st;
  LSR_HIP(hipMemcpyAsync(idx, d_idx, sizeof(int) * source.n, hipMemcpyDeviceToHost, stream));
  LSR_HIP(hipMemcpyAsync(d2, d_d2, sizeof(float) * source.n, hipMemcpyDeviceToHost, stream));
  LSR_HIP(hipStreamSynchronize(stream));
  return LSR_OK;
}

// getFitnessScore in two halves: _begin enqueues search + reduction (the result goes to the build mailbox), _end waits for
// it.  A batch of candidates runs every _begin (each on its own handle's stream) before the first _end.
int nn_fitness_begin(const DeviceCloud& source, const float* T16_host, const HashGridDev& grid, double max_range, BuildScratch& sc,
                     DevBuf<float>& d_T16, hipStream_t stream) {
  int* d_idx; float* d_d2; double* d_part; int* d_work;
  int st = nn_scratch(sc, source.n, &d_idx, &d_d2, &d_part, &d_work);
  if (st) return st;
  LSR_HIP(hipMemcpyAsync(d_T16.p, T16_host, 16 * sizeof(float), hipMemcpyHostToDevice, stream));
  const float max_d2 = (max_range >= 3.0e38) ? INFINITY : (float)max_range * 1.0001f;
  if ((st = nn_search_device(source, d_T16.p, grid, nn_fitness_fine_rings(), max_d2, d_idx, d_d2, stream, d_work))) return st;
  const int nb = 256;
  hipLaunchKernelGGL(fitness_partial_kernel, dim3(nb), dim3(256), 0, stream, d_idx, d_d2, (int)source.n, max_range, d_part);
  if ((st = sc.ensure_mailbox())) return st;
  unsigned int token = ++sc.token;
  if (token == 0) token = ++sc.token;
  hipLaunchKernelGGL(fitness_final_kernel, dim3(1), dim3(256), 0, stream, d_part, sc.d_mb, token);
  LSR_HIP(hipGetLastError());
  sc.fit_token = token;
  return LSR_OK;
}


// getFitnessScore of `count` candidates in three launches per group of FIT_GROUP (search, partial sums, final sum) on ONE
// stream; every member's result arrives in its own mailbox (nn_fitness_end collects it).
int nn_fitness_begin_group(const FitJob* jobs, int count, hipStream_t stream) {
  int st;
  for (int g0 = 0; g0 < count; g0 += FIT_GROUP) {
    FitGroup grp;
    std::memset(&grp, 0, sizeof(grp));
    const int ng = std::min(FIT_GROUP, count - g0);
    int max_blocks = 1, max_n = 0;
    for (int k = 0; k < ng; k++) {
      const FitJob& J = jobs[g0 + k];
      BuildScratch& sc = *J.sc;
      int* d_idx; float* d_d2; double* d_part; int* d_work;
      if ((st = nn_scratch(sc, J.source->n, &d_idx, &d_d2, &d_part, &d_work))) return st;

      if ((st = sc.ensure_mailbox())) return st;
      unsigned int token = ++sc.token;
      if (token == 0) token = ++sc.token;
      sc.fit_token = token;
      FitMember& M = grp.m[k];
      M.G = make_view(*J.grid);
      M.qx = J.source->x(); M.qy = J.source->y(); M.qz = J.source->z();
      M.n = (int)J.source->n;
      M.blocks = (int)(((long)M.n * 64 + 255) / 256);
      M.empty = (J.grid->n_blocks == 0 || M.n == 0) ? 1 : 0;   // no finite target point: no pair, the score is DBL_MAX (nn_fitness_end)
      for (int a = 0; a < 16; a++) M.T16[a] = J.T16[a];
      M.max_d2 = (J.max_range >= 3.0e38) ? INFINITY : (float)J.max_range * 1.0001f;
      M.max_range = J.max_range;
      M.idx = d_idx; M.d2 = d_d2; M.part = d_part; M.work = d_work;
      M.mb = sc.d_mb; M.token = token;
      M.fine_rings = nn_fitness_fine_rings();
      static const int ball_cells = [] { const char* e = getenv("LSR_FIT_BALL_CELLS"); const int v = e ? atoi(e) : FIT_BALL_CELLS; return v < 1 ? 1 : v > 8 ? 8 : v; }();
      M.ball_cells = ball_cells;
      if (!M.empty) { max_blocks = std::max(max_blocks, M.blocks); max_n = std::max(max_n, M.n); }
    }
    static const int form = [] { const char* e = getenv("LSR_FIT_GROUP_FORM"); return e ? atoi(e) : -1; }();   // 0 wave, 1 quad + tail, -1 by size
    const bool quad_form = form == 1;   // measured on 64 candidates x 30k queries: 5.85 ms against 3.54 ms for one wave per query
    const bool ball_form = form == 2 || form < 0;   // sixteen lanes per query seeded by its own fine cell + a tail for the rest (default)
    if (ball_form) {
      hipLaunchKernelGGL(fit_zero_work_group_kernel, dim3(1), dim3(ng), 0, stream, grp);
      hipLaunchKernelGGL(nn1_ball_group_kernel, dim3((unsigned)(((long)max_n * 16 + 255) / 256), ng), dim3(256), 0, stream, grp);
      // the rest (one wave per query, a stride loop over the member's work list): a member whose scan leaves the submap puts
      // thousands of queries there, each a walk of 10-30 us; a small group gives such a member more waves (round 5: a share of 8
      // with one such member spent 114 us here on 1 024 waves per member; the answers do not depend on the geometry)
      const unsigned list_wgs = ng <= 8 ? 1024u : 256u;
      hipLaunchKernelGGL(nn1_list_group_kernel, dim3(list_wgs, ng), dim3(256), 0, stream, grp);
    } else if (quad_form) {
      hipLaunchKernelGGL(fit_zero_work_group_kernel, dim3(1), dim3(ng), 0, stream, grp);
      hipLaunchKernelGGL(nn1_quad_group_kernel, dim3((unsigned)(((long)max_n * 4 + NN_THREADS - 1) / NN_THREADS), ng), dim3(NN_THREADS), 0, stream, grp);
      hipLaunchKernelGGL(nn1_coop_group_kernel, dim3(64, ng), dim3(256), 0, stream, grp);
    } else {
      hipLaunchKernelGGL(nn1_wave_group_kernel, dim3(max_blocks, ng), dim3(256), 0, stream, grp);
    }
    hipLaunchKernelGGL(fitness_partial_group_kernel, dim3(256, ng), dim3(256), 0, stream, grp);
    hipLaunchKernelGGL(fitness_final_group_kernel, dim3(ng), dim3(256), 0, stream, grp);
  }
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

int nn_fitness_end(BuildScratch& sc, hipStream_t stream, double* out) {
  if (sc.fit_token == 0) { set_last_error("fitness score collected before it was enqueued"); return LSR_ERR_HIP; }
  const unsigned int token = sc.fit_token;
  sc.fit_token = 0;
  int st = wait_mailbox_word(&sc.mb.p->fit_token, token, stream, sc.wait_mode, "fitness score");
  if (st) return st;
  const double sum = sc.mb.p->fit_sum, cnt = sc.mb.p->fit_cnt;
  *out = (cnt > 0) ? sum / cnt : 1.7976931348623157e308;  // std::numeric_limits<double>::max()
  return LSR_OK;
}

int nn_fitness_score(const DeviceCloud& source, const float* T16_host, const HashGridDev& grid, double max_range, double* out,
                     BuildScratch& sc, DevBuf<float>& d_T16, hipStream_t stream) {
  int st = nn_fitness_begin(source, T16_host, grid, max_range, sc, d_T16, stream);
  if (st) return st;
  return nn_fitness_end(sc, stream, out);
}

}  // namespace lsr
