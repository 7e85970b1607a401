// K1 / K2 for GENERAL key spaces and everything around a target build: bounding box, leaf keys ->
// hand-written LSD sort (lsd_sort.hip) -> run table -> per-leaf fp64 sums -> finalisation (covariance, eigen clamp, inverse:
// grid_device.hpp), the LDS image of the usable leaves, and the orchestration of single builds and of candidate sets.  The
// counting-sort builder for dense key spaces lives in grid_dense.hip, the derivative kernels in ndt.hip.  Reference call sites:
// scanmatcher_component.cpp:275,307; graph_based_slam_component.cpp:227.  (Split out of ndt.hip in round 6: same code.)
#include "ndt.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "build_kernels.hpp"
#include "grid_device.hpp"
#include "sort.hpp"

namespace lsr {

// ===========================================================================================
// K1 / K2: voxel-covariance grid
// ===========================================================================================
namespace {

// bbox over finite points: every workgroup reduces its share and writes ONE 32-byte record {min xyz, max xyz, #finite,
// token} straight into the host mailbox; the host folds the (<= 256) records.  No device atomics, no arrival ticket, no
// fence, no read-back copy: cross-workgroup atomics on seven addresses cost this kernel 15-25 us, the streaming
// reduction itself takes 4.
__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                   const float* __restrict__ z, int n, BuildMailbox* __restrict__ mb, unsigned int token) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  unsigned int cnt = 0;
  const int step = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * step) {  // four points per trip: 12 loads in flight
    float p[4][3];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * step;
      const bool in = i < n;
      p[u][0] = in ? x[i] : NAN; p[u][1] = in ? y[i] : NAN; p[u][2] = in ? z[i] : NAN;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (!(isfinite(p[u][0]) && isfinite(p[u][1]) && isfinite(p[u][2]))) continue;
      cnt++;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        mn[k] = fminf(mn[k], p[u][k]);
        mx[k] = fmaxf(mx[k], p[u][k]);
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], m, 64));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], m, 64));
    }
    cnt += __shfl_xor(cnt, m, 64);
  }
  __shared__ float s_mn[4][3], s_mx[4][3];
  __shared__ unsigned int s_cnt[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    for (int k = 0; k < 3; k++) { s_mn[w][k] = mn[k]; s_mx[w][k] = mx[k]; }
    s_cnt[w] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < BBOX_GRANULES) {
    const int k = threadIdx.x;
    unsigned int bits;
    if (k < 3) bits = __float_as_uint(fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k])));
    else if (k < 6) bits = __float_as_uint(fmaxf(fmaxf(s_mx[0][k - 3], s_mx[1][k - 3]), fmaxf(s_mx[2][k - 3], s_mx[3][k - 3])));
    else bits = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __hip_atomic_store(&mb->part[blockIdx.x].g[k], ((unsigned long long)token << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// K1: one wave per leaf; lanes stride the leaf's points (stable-sorted => ascending point index),
// fp64 sums, fixed butterfly order => deterministic.  sums[leaf][9] = {Sx,Sy,Sz,Sxx,Sxy,Sxz,Syy,Syz,Szz}
__global__ __launch_bounds__(256) void leaf_sum_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ z, const int* __restrict__ order,
                                                       const int* __restrict__ run_off, const int* __restrict__ run_cnt,
                                                       int n_runs, double* __restrict__ sums, const int* __restrict__ n_runs_dev /*nullable*/) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (n_runs_dev) n_runs = *n_runs_dev;   // launched over an upper bound: the host has not waited for the count
  if (wave >= n_runs) return;
  const int off = run_off[wave], cnt = run_cnt ? run_cnt[wave] : run_off[wave + 1] - off;   // (a table from sorted_runs_table closes with run_off[n_runs] = n)
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int j = lane; j < cnt; j += 64) {  // unrolled: several gathers in flight, additions stay in index order
    const int pi = order[off + j];
    const double px = (double)x[pi], py = (double)y[pi], pz = (double)z[pi];
    s[0] += px; s[1] += py; s[2] += pz;
    s[3] += px * px; s[4] += px * py; s[5] += px * pz;
    s[6] += py * py; s[7] += py * pz; s[8] += pz * pz;
  }
#pragma unroll
  for (int k = 0; k < 9; k++) s[k] = wave_sum(s[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) sums[(size_t)wave * 9 + k] = s[k];
  }
}

// K2: one thread per leaf: mean, single-pass covariance, (n-1)/n, eigenvalue clamp, inverse (leaf_finalize_dev).
__global__ __launch_bounds__(256) void leaf_finalize_kernel(const double* __restrict__ sums, const unsigned int* __restrict__ run_key,
                                                            const int* __restrict__ run_cnt /*nullable: counts from run_off*/,
                                                            const int* __restrict__ run_off, int n_runs, int min_points,
                                                            double eig_mult, float4* __restrict__ rec,
                                                            double* __restrict__ mean64, double* __restrict__ icov64,
                                                            int* __restrict__ leaf_key, int* __restrict__ leaf_n,
                                                            int* __restrict__ cell_slot, int* __restrict__ n_valid, int dense,
                                                            unsigned int sentinel, const int* __restrict__ n_runs_dev /*nullable*/) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = false;
  if (n_runs_dev) n_runs = *n_runs_dev;
  if (r < n_runs) {
    const unsigned int key = run_key[r];
    if (key == sentinel) {  // the run of non-finite points: not a leaf
      leaf_key[r] = -1;
      leaf_n[r] = 0;
    } else {
      double mean[3], icov[9];
      const int n = leaf_finalize_dev(sums + (size_t)r * 9, run_cnt ? run_cnt[r] : run_off[r + 1] - run_off[r], min_points, eig_mult, mean, icov, &valid);
      leaf_key[r] = (int)key;
      leaf_n[r] = n;
      for (int k = 0; k < 3; k++) mean64[(size_t)r * 3 + k] = mean[k];
      for (int k = 0; k < 9; k++) icov64[(size_t)r * 9 + k] = icov[k];
      const size_t ri = dense ? (size_t)key : (size_t)r;  // dense: record lives at its cell index
      leaf_record_dev(mean, icov, n, valid, rec + ri * 4);
      cell_slot[key] = valid ? (int)ri : -1;
    }
  }
  // one atomic per wave: device-scope atomics on one address are served one after the other, ~13 ns each (15 000 leaves: 0.2 ms)
  const unsigned long long m = __ballot(valid);
  if (m && (threadIdx.x & 63) == 0) atomicAdd(n_valid, __popcll(m));
}

}  // namespace

// Bounding box over the finite points of a cloud: one launch, the per-workgroup records arrive in the host mailbox and
// the host folds them (no copy, no stream synchronisation).
// Two halves, so that a batch of builds can enqueue every bounding-box pass before waiting for the first one.
int cloud_bbox_begin(const DeviceCloud& cloud, BuildScratch& sc, hipStream_t stream) {
  const int n = (int)cloud.n;
  if (n > 0 && !cloud.bbox_valid && cloud.bbox_enqueued && sc.bbox_parts > 0) return LSR_OK;   // pc2_ingest wrote the records already
  sc.bbox_parts = 0;
  if (n <= 0 || cloud.bbox_valid) return LSR_OK;
  int st = sc.ensure_mailbox();
  if (st) return st;
  unsigned int token = ++sc.token;
  if (token == 0) token = ++sc.token;
  const int nb = std::max(1, std::min((n + 1023) / 1024, BBOX_MAX_PARTS));  // four points per thread per trip
  hipLaunchKernelGGL(bbox_kernel, dim3(nb), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), n, sc.d_mb, token);
  LSR_HIP(hipGetLastError());
  sc.bbox_parts = nb;
  sc.bbox_token = token;
  return LSR_OK;
}

int cloud_bbox_end(const DeviceCloud& cloud, float* mn, float* mx, unsigned int* n_finite, BuildScratch& sc, hipStream_t stream) {
  *n_finite = 0;
  for (int k = 0; k < 3; k++) { mn[k] = 0.f; mx[k] = 0.f; }
  if ((int)cloud.n <= 0) return LSR_OK;
  if (cloud.bbox_valid) {
    *n_finite = cloud.bbox_finite;
    for (int k = 0; k < 3; k++) { mn[k] = cloud.bbox_mn[k]; mx[k] = cloud.bbox_mx[k]; }
    return LSR_OK;
  }
  const int nb = sc.bbox_parts;
  if (nb <= 0) { set_last_error("bounding box collected before it was enqueued"); return LSR_ERR_HIP; }
  const unsigned int token = sc.bbox_token;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  unsigned int cnt = 0;
  int st;
  for (int b = nb - 1; b >= 0; b--) {  // the last workgroups finish last: wait there first, the rest is usually in already
    const BboxPart& P = sc.mb.p->part[b];
    float v[6];
    for (int k = 0; k < BBOX_GRANULES; k++) {
      // one 8-byte store on the device side: value and token travel together.  Checked inline (a candidate set folds tens of
      // thousands of granules); only a granule that has not landed yet takes the polling path.
      unsigned long long g = __atomic_load_n(&P.g[k], __ATOMIC_ACQUIRE);
      if ((unsigned int)(g >> 32) != token) {
        const volatile unsigned int* halves = reinterpret_cast<const volatile unsigned int*>(&P.g[k]);   // [0] value bits, [1] token
        if ((st = wait_mailbox_word(halves + 1, token, stream, sc.wait_mode, "bounding box"))) return st;
        g = __atomic_load_n(&P.g[k], __ATOMIC_ACQUIRE);
      }
      const unsigned int bits = (unsigned int)(g & 0xFFFFFFFFull);
      if (k < 6) std::memcpy(&v[k], &bits, 4); else cnt += bits;
    }
    for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], v[k]); hi[k] = std::fmax(hi[k], v[3 + k]); }
  }
  sc.bbox_parts = 0;
  cloud.bbox_enqueued = false;
  *n_finite = cnt;
  if (cnt) for (int k = 0; k < 3; k++) { mn[k] = lo[k]; mx[k] = hi[k]; }
  cloud.bbox_valid = true;
  cloud.bbox_finite = cnt;
  for (int k = 0; k < 3; k++) { cloud.bbox_mn[k] = mn[k]; cloud.bbox_mx[k] = mx[k]; }
  return LSR_OK;
}

int cloud_bbox(const DeviceCloud& cloud, float* mn, float* mx, unsigned int* n_finite, BuildScratch& sc, hipStream_t stream) {
  int st = cloud_bbox_begin(cloud, sc, stream);
  if (st) return st;
  return cloud_bbox_end(cloud, mn, mx, n_finite, sc, stream);
}

// ---- LDS image of the valid-voxel table (NDT_TAB_LDS) ----------------------------------------------------------
// One workgroup: ordered compaction of the usable leaves (cell order => deterministic slot numbers), uint16 cell->slot
// map (0xFFFF = no usable leaf) followed by 48-byte records {mean.xyz, c00 | c01 c02 c11 c12 | c22, 0, 0, 0}.  The image is
// only written when it fits image_cap bytes.  The counts go to the host mailbox (n_valid, n_occupied, lds bytes), then
// the done token: the host learns the outcome of the whole grid build by polling one word.
namespace {
__device__ __forceinline__ void lds_pack_body(const int* __restrict__ cell_slot, const float4* __restrict__ rec,
                                              const int* __restrict__ leaf_n /*nullable: per cell*/, int ncells, int map_bytes,
                                              int image_cap, unsigned char* __restrict__ image, BuildMailbox* __restrict__ mb,
                                              unsigned int token) {
  __shared__ int s_wv[16];
  __shared__ int s_occ[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (ncells + 1023) / 1024;
  const int c0 = min(ncells, tid * per), c1 = min(ncells, c0 + per);
  int cnt = 0, occ = 0;
  for (int c = c0; c < c1; c++) {
    cnt += (cell_slot[c] >= 0);
    if (leaf_n) occ += (leaf_n[c] != 0);
  }
  // valid leaves before this thread's slice: a scan inside every wave, one barrier, the waves' totals (until round 5 a Hillis-Steele
  // scan over 1024 LDS words with twenty barriers: most of this kernel's 10 us)
  int inc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(inc, d, 64);
    if (lane >= d) inc += v;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) occ += __shfl_xor(occ, m, 64);
  if (lane == 63) s_wv[wave] = inc;
  if (lane == 0) s_occ[wave] = occ;
  __syncthreads();
  int before = inc - cnt, n_valid = 0;
  for (int w = 0; w < 16; w++) { if (w < wave) before += s_wv[w]; n_valid += s_wv[w]; }
  const long long want = (((long long)map_bytes + (long long)max(n_valid, 1) * NDT_LDS_REC_BYTES) + 1023) & ~1023ll;
  const bool fits = image != nullptr && want <= (long long)image_cap && n_valid <= 65534;
  if (fits) {
    int slot = before;
    unsigned short* map = reinterpret_cast<unsigned short*>(image);
    float4* out = reinterpret_cast<float4*>(image + map_bytes);
    for (int c = c0; c < c1; c++) {
      const int ri = cell_slot[c];
      if (ri >= 0) {
        const float4 a = rec[(size_t)ri * 4 + 0], b = rec[(size_t)ri * 4 + 1], d = rec[(size_t)ri * 4 + 2];
        out[(size_t)slot * 3 + 0] = a;
        out[(size_t)slot * 3 + 1] = b;
        out[(size_t)slot * 3 + 2] = d;   // {c22, mean_lo.xyz}
        map[c] = (unsigned short)slot;
        slot++;
      } else {
        map[c] = 0xFFFFu;
      }
    }
    // tail of the image (padding of the map to 16 bytes, of the records to 1 KiB, record 0 of an empty table): zero
    unsigned int* words = reinterpret_cast<unsigned int*>(image);
    const int w0 = (ncells * 2 + 3) / 4, w1 = map_bytes / 4;
    for (int k = w0 + tid; k < w1; k += 1024) words[k] = 0u;
    if (tid == 0 && (ncells & 1)) map[ncells] = 0u;
    const int r0 = (map_bytes + n_valid * NDT_LDS_REC_BYTES) / 4, r1 = (int)(want / 4);
    for (int k = r0 + tid; k < r1; k += 1024) words[k] = 0u;
  }
  __syncthreads();
  if (tid == 0 && mb != nullptr) {
    int o = 0;
    for (int k = 0; k < 16; k++) o += s_occ[k];
    mb->n_valid = n_valid;
    mb->n_occupied = o;
    mb->lds_bytes = fits ? (int)want : 0;
    mb->lds_map_bytes = fits ? map_bytes : 0;
    __threadfence_system();
    __hip_atomic_store(&mb->done_token, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(1024) void lds_pack_kernel(const int* __restrict__ cell_slot, const float4* __restrict__ rec,
                                                        const int* __restrict__ leaf_n, int ncells, int map_bytes, int image_cap,
                                                        unsigned char* __restrict__ image, BuildMailbox* __restrict__ mb, unsigned int token) {
  lds_pack_body(cell_slot, rec, leaf_n, ncells, map_bytes, image_cap, image, mb, token);
}
// one workgroup per member of a group of targets (batched builds, grid_dense.hip)
__global__ __launch_bounds__(1024) void lds_pack_group_kernel(const PackGroup g) {
  const PackMember& M = g.m[blockIdx.x];
  lds_pack_body(M.cell_slot, M.rec, M.leaf_n, M.ncells, M.map_bytes, M.image_cap, M.image, M.mb, M.token);
}
}  // namespace

int ndt_pack_lds_tables(VoxelGridDev* const* grids, BuildScratch* const* scs, const unsigned int* tokens, int count, hipStream_t stream) {
  for (int g0 = 0; g0 < count; g0 += LSR_GROUP) {
    PackGroup grp;
    const int ng = std::min(LSR_GROUP, count - g0);
    for (int k = 0; k < ng; k++) {
      VoxelGridDev& grid = *grids[g0 + k];
      BuildScratch& sc = *scs[g0 + k];
      grid.lds_bytes = grid.lds_map_bytes = 0;
      int st = sc.ensure_mailbox();
      if (st) return st;
      const bool may_fit = grid.ncells > 0 && grid.ncells * 2 + 16 + NDT_LDS_REC_BYTES <= (size_t)NDT_LDS_TABLE_MAX;
      if (may_fit && (st = grid.lds_image.reserve(NDT_LDS_TABLE_MAX / 16))) return st;
      PackMember& M = grp.m[k];
      M.cell_slot = grid.cell_slot.p; M.rec = grid.rec.p; M.leaf_n = grid.leaf_n.p; M.ncells = (int)grid.ncells;
      M.map_bytes = (int)((grid.ncells * 2 + 15) & ~(size_t)15); M.image_cap = (int)NDT_LDS_TABLE_MAX;
      M.image = may_fit ? reinterpret_cast<unsigned char*>(grid.lds_image.p) : (unsigned char*)nullptr;
      M.mb = sc.d_mb; M.token = tokens[g0 + k];
    }
    hipLaunchKernelGGL(lds_pack_group_kernel, dim3(ng), dim3(1024), 0, stream, grp);
  }
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

// Enqueue the pack; the outcome is read from the host mailbox by ndt_finish_grid().
int ndt_pack_lds_table(VoxelGridDev& grid, BuildScratch& sc, bool per_cell_leaf_n, unsigned int token, hipStream_t stream) {
  grid.lds_bytes = grid.lds_map_bytes = 0;
  int st = sc.ensure_mailbox();
  if (st) return st;
  const bool may_fit = grid.ncells > 0 && grid.ncells * 2 + 16 + NDT_LDS_REC_BYTES <= (size_t)NDT_LDS_TABLE_MAX;
  const int map_bytes = (int)((grid.ncells * 2 + 15) & ~(size_t)15);
  if (may_fit && (st = grid.lds_image.reserve(NDT_LDS_TABLE_MAX / 16))) return st;
  hipLaunchKernelGGL(lds_pack_kernel, dim3(1), dim3(1024), 0, stream, grid.cell_slot.p, grid.rec.p,
                     per_cell_leaf_n ? grid.leaf_n.p : (const int*)nullptr, (int)grid.ncells, map_bytes, (int)NDT_LDS_TABLE_MAX,
                     may_fit ? reinterpret_cast<unsigned char*>(grid.lds_image.p) : (unsigned char*)nullptr, sc.d_mb, token);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

int ndt_build_grid(const DeviceCloud& cloud, float leaf, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream) {
  int st = cloud_bbox_begin(cloud, sc, stream);
  if (st) return st;
  if ((st = ndt_build_grid_begin(cloud, leaf, grid, sc, stream))) return st;
  return ndt_build_grid_end(grid, sc, stream);
}

// Collect a build left pending by ndt_build_grid_begin (host poll #2).
int ndt_build_grid_end(VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream) {
  if (!sc.grid_pending) return LSR_OK;
  sc.grid_pending = false;
  int st = wait_mailbox_word(&sc.mb.p->done_token, sc.grid_token, stream, sc.wait_mode, "voxel grid build");
  if (st) return st;
  grid.n_valid = sc.mb.p->n_valid;
  grid.lds_bytes = sc.mb.p->lds_bytes;
  grid.lds_map_bytes = sc.mb.p->lds_map_bytes;
  return LSR_OK;
}

// Geometry of the voxel grid from the cloud's bounding box (host poll #1; the bounding-box pass must have been enqueued —
// cloud_bbox_begin / the batched ingest — or be cached).  *path: 0 = no finite point (empty grid), 1 = dense key space
// (counting-sort builder, grid_dense.hip), 2 = general key space (radix sort).
int ndt_grid_geometry(const DeviceCloud& cloud, float leaf, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream, int* path) {
  *path = 0;
  sc.grid_pending = false;
  const int n = (int)cloud.n;
  grid.leaf = leaf;
  grid.n_leaves = grid.n_valid = 0;
  grid.ncells = 0;
  grid.lds_bytes = grid.lds_map_bytes = 0;
  grid.has_sorted = false;
  for (int k = 0; k < 3; k++) { grid.min_b[k] = 0; grid.max_b[k] = -1; grid.div_b[k] = 0; }
  if (n == 0) return LSR_OK;
  const float inv_leaf = 1.0f / leaf;
  float mn[3], mx[3];
  unsigned int n_finite = 0;
  int st = cloud_bbox_end(cloud, mn, mx, &n_finite, sc, stream);   // host poll #1
  if (st) return st;
  if (n_finite == 0) return LSR_OK;  // no finite point: empty grid
  int64_t d[3];
  for (int k = 0; k < 3; k++) d[k] = (int64_t)((mx[k] - mn[k]) * inv_leaf) + 1;
  if (d[0] * d[1] * d[2] > (int64_t)INT32_MAX) {
    set_last_error("voxel index space exceeds int32: leaf size too small for the target extent");
    return LSR_ERR_INDEX_OVERFLOW;
  }
  for (int k = 0; k < 3; k++) {
    grid.min_b[k] = (int)floorf(mn[k] * inv_leaf);
    grid.max_b[k] = (int)floorf(mx[k] * inv_leaf);
    grid.div_b[k] = grid.max_b[k] - grid.min_b[k] + 1;
  }
  grid.ncells = (size_t)grid.div_b[0] * grid.div_b[1] * grid.div_b[2];
  *path = (grid.ncells <= (size_t)VG_DENSE_MAX_CELLS && !sc.force_sort_path) ? 1 : 2;
  return LSR_OK;
}

static unsigned int next_token(BuildScratch& sc) {
  unsigned int token = ++sc.token;
  if (token == 0) token = ++sc.token;
  return token;
}

// general key space (more than VG_DENSE_MAX_CELLS cells: cfg 5's 2 m grid over a 20-frame submap, the reference's own 1.0-2.0 m
// resolutions): leaf keys -> hand-written stable LSD radix sort (lsd_sort.hip; the points of a leaf stay in ascending index, so the
// fp64 sums of a leaf are a fixed function of the cloud) -> run heads (count per 256 keys, one-workgroup scan, run table) -> leaf sums
// -> finalise.  Returns with the grid complete.  LSR_TARGET_SORT=rocprim keeps rounds 1-5's rocPRIM radix sort + run_length_encode +
// exclusive_scan as the A/B cross-check (same leaves, same sums, bit for bit: tests/test_ndt_gpu.py).
static int ndt_build_grid_general(const DeviceCloud& cloud, float leaf, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream) {
  static const bool use_rocprim = [] { const char* e = std::getenv("LSR_TARGET_SORT"); return e && std::strcmp(e, "rocprim") == 0; }();
  DevBuf<char>& temp = sc.temp;
  DevBuf<unsigned int>& scratch = sc.words;
  DevBuf<double>& sums = sc.sums;
  const int n = (int)cloud.n;
  const float inv_leaf = 1.0f / leaf;
  const int mul1 = grid.div_b[0], mul2 = grid.div_b[0] * grid.div_b[1];
  const unsigned int token = next_token(sc);
  int st = grid.cell_slot.reserve(grid.ncells);
  if (st) return st;
  // Dense leaf records (64 B per grid cell) while the table stays <= 256 MiB; compact otherwise.
  grid.dense = grid.ncells <= ((size_t)4 << 20);
  if (grid.dense && (st = grid.rec.reserve(grid.ncells * 4))) return st;
  const unsigned int sentinel = (unsigned int)grid.ncells;  // non-finite points: one past the last leaf index
  // scratch carved from one allocation: pad[16] | key_in[n] | key_out[n] | val_in[n] | val_out[n] | run_key[n+1] | run_cnt[n+1] | run_off[n+1] |
  // nruns | nvalid | block_heads[nb] | block_base[nb]
  const size_t nb = sorted_runs_blocks((size_t)n);
  size_t words = 16 + 7 * (size_t)n + 3 + 16 + 2 * nb;
  if ((st = scratch.reserve(words))) return st;
  unsigned int* key_in = scratch.p + 16;
  unsigned int* key_out = key_in + n;
  int* val_in = (int*)(key_out + n);
  int* val_out = val_in + n;
  unsigned int* run_key = (unsigned int*)(val_out + n);
  int* run_cnt = (int*)(run_key + n + 1);
  int* run_off = run_cnt + n + 1;
  int* d_nruns = run_off + n + 1;
  int* d_nvalid = d_nruns + 1;
  int* block_heads = d_nvalid + 15;
  int* block_base = block_heads + nb;

  hipLaunchKernelGGL(leaf_key_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), n,
                     inv_leaf, grid.min_b[0], grid.min_b[1], grid.min_b[2], mul1, mul2, sentinel, key_in, use_rocprim ? val_in : (int*)nullptr,
                     reinterpret_cast<uint4*>(grid.dense ? grid.rec.p : nullptr), grid.dense ? grid.ncells * 4 : (size_t)0,   // all-ones words are NaN: empty cells answer no lookup
                     grid.cell_slot.p, grid.ncells, d_nvalid);
  // keys live in [0, ncells]: only that many radix bits are sorted
  int n_runs = 0;
  const int* order = val_out;
  const int* counts = run_cnt;
  const int* n_runs_dev = nullptr;   // non-null: the kernels behind read the run count from the device, the host learns it at the end
  unsigned int rtoken = 0;
  // runs <= min(points, cells + 1): with that bound no larger than 1 Mi the leaf buffers are sized by it and the host does not wait
  // for the count in the middle of the chain (one round trip and ~8 us of idle device less)
  const size_t run_bound = std::min((size_t)n, grid.ncells + 1);
  if (use_rocprim) {
    st = sort_pairs_u32(key_in, key_out, val_in, val_out, n, bits_for(sentinel), temp, stream);
    if (st) return st;
    st = run_length_encode_u32(key_out, n, run_key, run_cnt, d_nruns, temp, stream);
    if (st) return st;
    // through the host mailbox (one small launch + a poll: no copy engine, no stream synchronisation)
    if ((st = publish_device_int(d_nruns, sc, stream, &n_runs))) return st;
    st = exclusive_scan_i32(run_cnt, run_off, n_runs, temp, stream);
    if (st) return st;
  } else {
    bool in_b = false;
    if ((st = sort_pairs_u32_lsd(key_in, key_out, nullptr, val_in, val_out, (size_t)n, bits_for(sentinel), temp, stream, &in_b))) return st;
    const unsigned int* ks = in_b ? key_out : key_in;
    order = in_b ? val_out : val_in;
    if ((st = sorted_runs_begin(ks, (size_t)n, block_heads, block_base, sc, stream, &rtoken, nullptr, d_nruns))) return st;
    if ((st = sorted_runs_table(ks, (size_t)n, block_base, run_key, run_off, stream))) return st;   // enqueued before the count is known
    counts = nullptr;   // run r covers [run_off[r], run_off[r + 1])
    if (run_bound <= ((size_t)1 << 20)) { n_runs = (int)run_bound; n_runs_dev = d_nruns; }
    else if ((st = sorted_runs_count(sc, stream, rtoken, &n_runs))) return st;
  }

  st = sums.reserve((size_t)n_runs * 9);
  if (st) return st;
  if (!grid.dense && (st = grid.rec.reserve((size_t)n_runs * 4))) return st;
  if ((st = grid.mean64.reserve((size_t)n_runs * 3))) return st;
  if ((st = grid.icov64.reserve((size_t)n_runs * 9))) return st;
  if ((st = grid.leaf_key.reserve(n_runs))) return st;
  if ((st = grid.leaf_n.reserve(n_runs))) return st;
  hipLaunchKernelGGL(leaf_sum_kernel, dim3((n_runs + 3) / 4), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), order,
                     run_off, counts, n_runs, sums.p, n_runs_dev);
  hipLaunchKernelGGL(leaf_finalize_kernel, dim3((n_runs + 255) / 256), dim3(256), 0, stream, sums.p, run_key, counts, run_off, n_runs,
                     6, 0.01, grid.rec.p, grid.mean64.p, grid.icov64.p, grid.leaf_key.p, grid.leaf_n.p, grid.cell_slot.p,
                     d_nvalid, grid.dense ? 1 : 0, sentinel, n_runs_dev);
  LSR_HIP(hipGetLastError());
  if ((st = ndt_pack_lds_table(grid, sc, false, token, stream))) return st;
  if ((st = wait_mailbox_word(&sc.mb.p->done_token, token, stream, sc.wait_mode, "voxel grid build"))) return st;
  if (n_runs_dev && (st = sorted_runs_count(sc, stream, rtoken, &n_runs))) return st;   // published long before the pack: no wait
  grid.n_leaves = n_runs;  // includes the sentinel run if non-finite points exist (leaf_key = -1)
  grid.n_valid = sc.mb.p->n_valid;
  grid.lds_bytes = sc.mb.p->lds_bytes;
  grid.lds_map_bytes = sc.mb.p->lds_map_bytes;
  return LSR_OK;
}

// ---- KDTREE neighbourhood: the leaves' float centroids --------------------------------------------------------------------------
// pclomp::VoxelGridCovariance keeps, next to the fp64 mean_, a FLOAT Leaf::centroid ("leaf.centroid += pt" point by point in cloud
// order, then "/= static_cast<float>(nr_points)"); the centroid cloud it hands to its kd-tree — the one radiusSearch() queries for the
// KDTREE neighbourhood — is made of these.  A float running sum is a function of the order of the points, so it is rebuilt here in
// that order: stable LSD sort of the point indices by leaf key (equal keys keep ascending index = cloud order), run table, one thread
// per leaf adding its points one after the other.  Independent of which builder made the grid; runs once per target, on first use.
namespace {
__global__ __launch_bounds__(256) void leaf_centroid_seq_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                                const int* __restrict__ order, const unsigned int* __restrict__ run_key,
                                                                const int* __restrict__ run_off, const int* __restrict__ n_runs_dev,
                                                                unsigned int sentinel, const int* __restrict__ cell_slot,
                                                                float4* __restrict__ centroid) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *n_runs_dev) return;
  const unsigned int key = run_key[r];
  if (key == sentinel) return;          // the run of the non-finite points
  const int slot = cell_slot[key];
  if (slot < 0) return;                 // fewer than min_points_per_voxel points / invalid covariance: not in the kd-tree
  const int beg = run_off[r], end = run_off[r + 1];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int j = beg; j < end; j++) {
    const int i = order[j];
    sx = __fadd_rn(sx, x[i]); sy = __fadd_rn(sy, y[i]); sz = __fadd_rn(sz, z[i]);
  }
  const float n = (float)(end - beg);
  centroid[slot] = make_float4(__fdiv_rn(sx, n), __fdiv_rn(sy, n), __fdiv_rn(sz, n), 0.f);
}
}  // namespace

int ndt_build_centroids(const DeviceCloud& cloud, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream) {
  const int n = (int)cloud.n;
  if (n == 0 || grid.ncells == 0) return LSR_OK;
  const size_t n_slots = grid.dense ? grid.ncells : (size_t)std::max(grid.n_leaves, 1);
  int st = grid.centroid.reserve(n_slots);
  if (st) return st;
  const unsigned int sentinel = (unsigned int)grid.ncells;
  const size_t nb = sorted_runs_blocks((size_t)n);
  // scratch: pad[16] | key_in[n] | key_out[n] | val_in[n] | val_out[n] | run_key[n+1] | run_off[n+1] | nruns[16] | block_heads[nb] | block_base[nb]
  if ((st = sc.words.reserve(16 + 6 * (size_t)n + 2 + 16 + 2 * nb))) return st;
  unsigned int* key_in = sc.words.p + 16;
  unsigned int* key_out = key_in + n;
  int* val_in = (int*)(key_out + n);
  int* val_out = val_in + n;
  unsigned int* run_key = (unsigned int*)(val_out + n);
  int* run_off = (int*)(run_key + n + 1);
  int* d_nruns = run_off + n + 1;
  int* block_heads = d_nruns + 16;
  int* block_base = block_heads + nb;
  hipLaunchKernelGGL(leaf_key_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), n, 1.0f / grid.leaf,
                     grid.min_b[0], grid.min_b[1], grid.min_b[2], grid.div_b[0], grid.div_b[0] * grid.div_b[1], sentinel, key_in, (int*)nullptr,
                     (uint4*)nullptr, (size_t)0, (int*)nullptr, (size_t)0, (int*)nullptr);
  bool in_b = false;
  if ((st = sort_pairs_u32_lsd(key_in, key_out, nullptr, val_in, val_out, (size_t)n, bits_for(sentinel), sc.temp, stream, &in_b))) return st;
  const unsigned int* ks = in_b ? key_out : key_in;
  const int* order = in_b ? val_out : val_in;
  unsigned int rtoken = 0;
  if ((st = sorted_runs_begin(ks, (size_t)n, block_heads, block_base, sc, stream, &rtoken, nullptr, d_nruns))) return st;
  if ((st = sorted_runs_table(ks, (size_t)n, block_base, run_key, run_off, stream))) return st;
  const size_t run_bound = std::min((size_t)n, grid.ncells + 1);
  hipLaunchKernelGGL(leaf_centroid_seq_kernel, dim3((unsigned)((run_bound + 255) / 256)), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(),
                     order, run_key, run_off, d_nruns, sentinel, grid.cell_slot.p, grid.centroid.p);
  LSR_HIP(hipGetLastError());
  int n_runs = 0;
  if ((st = sorted_runs_count(sc, stream, rtoken, &n_runs))) return st;   // the run count's mailbox word is consumed
  // complete on return: a target may be shared by objects on other streams
  if (hipStreamSynchronize(stream) != hipSuccess) { set_last_error("stream error while the leaf centroids were built"); return LSR_ERR_HIP; }
  return LSR_OK;
}

// Everything up to the last enqueue.  Dense key spaces leave the build pending (sc.grid_pending): ndt_build_grid_end() collects it.
int ndt_build_grid_begin(const DeviceCloud& cloud, float leaf, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream) {
  int path = 0;
  int st = ndt_grid_geometry(cloud, leaf, grid, sc, stream, &path);
  if (st || path == 0) return st;
  if (path == 2) return ndt_build_grid_general(cloud, leaf, grid, sc, stream);
  // dense key space: hand-written counting sort, no further host round trip until the final poll (grid_dense.hip)
  const unsigned int token = next_token(sc);
  if ((st = ndt_build_grid_dense(cloud, leaf, grid, sc, stream))) return st;
  if ((st = ndt_pack_lds_table(grid, sc, true, token, stream))) return st;
  sc.grid_pending = true;
  sc.grid_token = token;
  return LSR_OK;
}

// A SET of targets (candidate windows): every member's bounding box has been enqueued (ndt_targets_ingest / cloud_bbox_begin).
// Members with a dense key space are built by the GROUP kernels of grid_dense.hip — one launch per stage for up to LSR_GROUP
// members — and left pending; the others are built one by one right here.
int ndt_targets_build_begin(TargetBuildJob* jobs, int count, hipStream_t stream) {
  int st;
  // group by group: as soon as the bounding boxes of 16 members have arrived (the host folds their records while the ingest
  // launches of the later groups are still running) their builds are enqueued — the device never waits for the host to have
  // folded the whole set
  for (int g0 = 0; g0 < count; g0 += LSR_GROUP) {
    const int g1 = std::min(count, g0 + LSR_GROUP);
    std::vector<TargetBuildJob*> dense;
    for (int b = g0; b < g1; b++) {
      TargetBuildJob& J = jobs[b];
      if ((st = ndt_grid_geometry(*J.cloud, J.leaf, *J.grid, *J.sc, stream, &J.path))) return st;
      if (J.path == 1) dense.push_back(&J);
    }
    if (dense.empty()) continue;
    std::vector<VoxelGridDev*> grids;
    std::vector<BuildScratch*> scs;
    std::vector<unsigned int> tokens;
    for (TargetBuildJob* J : dense) {
      grids.push_back(J->grid);
      scs.push_back(J->sc);
      tokens.push_back(next_token(*J->sc));
    }
    if ((st = ndt_build_grids_dense_group(dense.data(), (int)dense.size(), stream))) return st;
    if ((st = ndt_pack_lds_tables(grids.data(), scs.data(), tokens.data(), (int)dense.size(), stream))) return st;
    for (size_t k = 0; k < dense.size(); k++) { dense[k]->sc->grid_pending = true; dense[k]->sc->grid_token = tokens[k]; }
  }
  for (int b = 0; b < count; b++)
    if (jobs[b].path == 2 && (st = ndt_build_grid_general(*jobs[b].cloud, jobs[b].leaf, *jobs[b].grid, *jobs[b].sc, stream))) return st;
  return LSR_OK;
}

}  // namespace lsr
