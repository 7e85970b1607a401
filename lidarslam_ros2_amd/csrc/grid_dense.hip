// K1/K2 for DENSE key spaces (<= VG_DENSE_MAX_CELLS grid cells: every reference configuration at ndt_resolution 5.0):
// the voxel-covariance grid of pclomp::VoxelGridCovariance::filter (scanmatcher_component.cpp:275,307;
// graph_based_slam_component.cpp:227) built with a hand-written LDS-histogram counting sort instead of
// a general radix sort:
//
//   bbox / vg_ingest   bounding box as {value, token} granules straight into the host mailbox (vg_ingest: fused with the
//                      de-interleave of the PointXYZI records)                    host folds the records, no fence on the device
//   vg_hist            key per point (uint16) + per-block LDS histogram           reads 12 B/pt, writes 2 B/pt
//   vg_scan            per cell: exclusive scan over the blocks                   nblk x C words
//   vg_cellscan        one workgroup: exclusive scan over the cells -> start[], rank[] of the occupied cells
//   vg_scatter         STABLE scatter of x,y,z (+ original index) into cell order reads 14 B/pt, writes 16 B/pt
//   vg_leaf            one workgroup per cell: fp64 sums in a fixed order + leaf finalisation (K2)   reads 12 B/pt
//   lds_pack           LDS image of the usable leaves, counts into the host mailbox (ndt.hip)
//
// Every stage exists as a body + a single-target kernel + a GROUP kernel (up to LSR_GROUP = 16 targets per launch, the
// members' pointers and sizes in the kernel arguments, blockIdx.y = member): a candidate set pays one launch per stage per
// 16 targets.  The voxel-ordered points stay with the grid (sorted planes, sorted_idx, cell_start, cell_rank): the
// neighbour grid of getFitnessScore is a refinement of that order (nn.hip).  No device-to-host copy, two host polls (bbox,
// done) — against ~35 launches and three stream synchronisations of the sort-based builder (which remains for larger key
// spaces, ndt.hip).
// Determinism: integer histograms; the scatter ranks equal keys by point index (wave ballots, waves ordered through
// packed 16-bit per-wave counters); the sums of a leaf are formed in an order that depends on its point count only.
#include <chrono>
#include <thread>

#include "grid_device.hpp"
#include "ndt.hpp"

namespace lsr {

int BuildScratch::ensure_mailbox() {
  if (mb.p) return LSR_OK;
  int st = mb.reserve(1, hipHostMallocMapped | hipHostMallocCoherent);
  if (st) return st;
  std::memset(mb.p, 0, sizeof(BuildMailbox));
  LSR_HIP(hipHostGetDevicePointer((void**)&d_mb, mb.p, 0));
  return LSR_OK;
}

int wait_mailbox_word(const volatile unsigned int* word, unsigned int token, hipStream_t stream, int wait_mode, const char* what) {
  auto t0 = std::chrono::steady_clock::now();
  for (unsigned long long spins = 1;; spins++) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == token) return LSR_OK;
    if (wait_mode == WAIT_YIELD) std::this_thread::yield();
    else if (wait_mode == WAIT_SLEEP) std::this_thread::sleep_for(std::chrono::microseconds(20));
    else __builtin_ia32_pause();
    if ((spins & 0x3FFF) == 0 || wait_mode == WAIT_SLEEP) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) {
        const hipError_t e = hipStreamQuery(stream);
        set_last_error(std::string(what) + ": no answer from the device for 30 s (stream: " + hipGetErrorString(e) + ")");
        return LSR_ERR_HIP;
      }
    }
  }
}

namespace {
__global__ void publish_int_kernel(const int* __restrict__ src, BuildMailbox* __restrict__ mb, unsigned int token) {
  mb->value = *src;
  __threadfence_system();
  __hip_atomic_store(&mb->value_token, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

int publish_device_int(const int* d_value, BuildScratch& sc, hipStream_t stream, int* out) {
  int st = sc.ensure_mailbox();
  if (st) return st;
  unsigned int token = ++sc.token;
  if (token == 0) token = ++sc.token;
  hipLaunchKernelGGL(publish_int_kernel, dim3(1), dim3(1), 0, stream, d_value, sc.d_mb, token);
  LSR_HIP(hipGetLastError());
  if ((st = wait_mailbox_word(&sc.mb.p->value_token, token, stream, sc.wait_mode, "device counter"))) return st;
  *out = sc.mb.p->value;
  return LSR_OK;
}

namespace {

constexpr int VG_CHUNK = 4096;  // points per workgroup: 4 waves x 16 steps x 64 lanes
constexpr int VG_STEPS = 16;

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// key = linear leaf index exactly as VoxelGridCovariance computes it; non-finite points get the
// sentinel bin `ncells` (sorted last, never a leaf)
__device__ __forceinline__ void vg_hist_kernel_body(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                      int n, float inv_leaf, int mb0, int mb1, int mb2, int mul1, int mul2, int ncells,
                                                      unsigned short* __restrict__ keys, unsigned short* __restrict__ hist, const int blk_x) {
  extern __shared__ unsigned int s_hist[];  // [ncells + 1]
  const int C = ncells + 1, tid = threadIdx.x;
  for (int k = tid; k < C; k += 256) s_hist[k] = 0u;
  __syncthreads();
  const int base = blk_x * VG_CHUNK;
  float px[VG_CHUNK / 256], py[VG_CHUNK / 256], pz[VG_CHUNK / 256];
#pragma unroll
  for (int j = 0; j < VG_CHUNK / 256; j++) {  // every load of the chunk in flight at once
    const int i = base + j * 256 + tid;
    const bool in = i < n;
    px[j] = in ? x[i] : 0.f; py[j] = in ? y[i] : 0.f; pz[j] = in ? z[i] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < VG_CHUNK / 256; j++) {
    const int i = base + j * 256 + tid;
    if (i < n) {
      unsigned int k = (unsigned int)ncells;
      if (isfinite(px[j]) && isfinite(py[j]) && isfinite(pz[j])) {
        const int i0 = (int)(floorf(px[j] * inv_leaf) - (float)mb0);
        const int i1 = (int)(floorf(py[j] * inv_leaf) - (float)mb1);
        const int i2 = (int)(floorf(pz[j] * inv_leaf) - (float)mb2);
        k = (unsigned int)(i0 + i1 * mul1 + i2 * mul2);
        if (k >= (unsigned int)ncells) k = (unsigned int)ncells;  // cannot happen for a bbox built from the same floats
      }
      keys[i] = (unsigned short)k;
      atomicAdd(&s_hist[k], 1u);
    }
  }
  __syncthreads();
  unsigned short* row = hist + (size_t)blk_x * C;
  for (int k = tid; k < C; k += 256) row[k] = (unsigned short)s_hist[k];  // <= VG_CHUNK = 4096
}
__global__ __launch_bounds__(256) void vg_hist_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                      int n, float inv_leaf, int mb0, int mb1, int mb2, int mul1, int mul2, int ncells,
                                                      unsigned short* __restrict__ keys, unsigned short* __restrict__ hist) {
  vg_hist_kernel_body(x, y, z, n, inv_leaf, mb0, mb1, mb2, mul1, mul2, ncells, keys, hist, (int)blockIdx.x);
}

// per cell: exclusive scan of the block histograms (offset of this block's points inside the cell) + cell total.
// 256 threads = 32 cells x 8 segments of the block range: eight times the loads in flight of a thread-per-cell loop.
constexpr int VG_SCAN_SEGS = 8;
__device__ __forceinline__ void vg_scan_kernel_body(const unsigned short* __restrict__ hist, int nblk, int C,
                                                      unsigned int* __restrict__ blkoff, unsigned int* __restrict__ total, const int blk_x) {
  __shared__ unsigned int s_seg[VG_SCAN_SEGS][32];
  const int cl = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const int k = blk_x * 32 + cl;
  const int per = (nblk + VG_SCAN_SEGS - 1) / VG_SCAN_SEGS;
  const int b0 = seg * per, b1 = min(nblk, b0 + per);
  unsigned int sum = 0u;
  if (k < C) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      unsigned int c[8];
#pragma unroll
      for (int u = 0; u < 8; u++) c[u] = hist[(size_t)(b + u) * C + k];
#pragma unroll
      for (int u = 0; u < 8; u++) sum += c[u];
    }
    for (; b < b1; b++) sum += hist[(size_t)b * C + k];
  }
  s_seg[seg][cl] = sum;
  __syncthreads();
  unsigned int run = 0u;
  for (int s2 = 0; s2 < seg; s2++) run += s_seg[s2][cl];
  if (k < C) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      unsigned int c[8];
#pragma unroll
      for (int u = 0; u < 8; u++) c[u] = hist[(size_t)(b + u) * C + k];
#pragma unroll
      for (int u = 0; u < 8; u++) { blkoff[(size_t)(b + u) * C + k] = run; run += c[u]; }
    }
    for (; b < b1; b++) { const unsigned int c = hist[(size_t)b * C + k]; blkoff[(size_t)b * C + k] = run; run += c; }
    if (seg == VG_SCAN_SEGS - 1) total[k] = run;
  }
}
__global__ __launch_bounds__(256) void vg_scan_kernel(const unsigned short* __restrict__ hist, int nblk, int C,
                                                      unsigned int* __restrict__ blkoff, unsigned int* __restrict__ total) {
  vg_scan_kernel_body(hist, nblk, C, blkoff, total, (int)blockIdx.x);
}

// one workgroup: exclusive scan over the cells -> start[0..C] (start[C] = n)
__device__ __forceinline__ void vg_cellscan_kernel_body(const unsigned int* __restrict__ total, int C, unsigned int* __restrict__ start,
                                                        unsigned int* __restrict__ rank /*nullable*/, const int blk_x) {
  // counts and occupancy flags of a thread's slice, a scan inside every wave (shuffles), one barrier, the waves' totals added up by
  // every thread (until round 5: a Hillis-Steele scan over 1024 LDS words, twenty barriers of sixteen waves each: 6.2 -> 4.7 us)
  __shared__ unsigned int s_ws[16], s_wo[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (C + 1023) / 1024;
  const int c0 = tid * per, c1 = min(C, c0 + per);
  unsigned int cnt = 0u, occ = 0u;
  for (int c = c0; c < c1; c++) { const unsigned int t = total[c]; cnt += t; occ += (t != 0u); }
  unsigned int ic = cnt, io = occ;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int v = __shfl_up(ic, d, 64), w = __shfl_up(io, d, 64);
    if (lane >= d) { ic += v; io += w; }
  }
  if (lane == 63) { s_ws[wave] = ic; s_wo[wave] = io; }
  __syncthreads();
  unsigned int run = ic - cnt, rrun = io - occ, all = 0u;
  for (int w = 0; w < 16; w++) { if (w < wave) { run += s_ws[w]; rrun += s_wo[w]; } all += s_ws[w]; }
  for (int c = c0; c < c1; c++) {
    const unsigned int t = total[c];
    start[c] = run; run += t;
    if (rank) { rank[c] = rrun; rrun += (t != 0u); }   // rank of the cell among the cells that hold points (the sentinel bin counts too: it is last)
  }
  if (tid == 1023) start[C] = all;
}
__global__ __launch_bounds__(1024) void vg_cellscan_kernel(const unsigned int* __restrict__ total, int C, unsigned int* __restrict__ start,
                                                           unsigned int* __restrict__ rank) {
  vg_cellscan_kernel_body(total, C, start, rank, (int)blockIdx.x);
}

// Scatter into cell order.  Wave w of block b owns points [b*4096 + w*1024, +1024) and walks them in 16 steps of
// 64 consecutive points.  s_c[k] packs four 16-bit counters (one per wave): first the per-wave counts of key k, then
// their exclusive prefix over the waves, then — advanced by ds_add_rtn_u64 — the running offset of each wave: blocks,
// waves and steps are in point order, lanes of one step are ranked by the returning atomic.
__device__ __forceinline__ void vg_scatter_kernel_body(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                         int n, const unsigned short* __restrict__ keys, const unsigned int* __restrict__ blkoff,
                                                         const unsigned int* __restrict__ start, int C, float* __restrict__ ox,
                                                         float* __restrict__ oy, float* __restrict__ oz, int* __restrict__ oidx /*nullable*/,
                                                         const int blk_x) {
  extern __shared__ unsigned long long s_c[];  // [C]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  for (int k = tid; k < C; k += 256) s_c[k] = 0ull;
  const int base_i = blk_x * VG_CHUNK + w * (VG_CHUNK / 4) + lane;
  unsigned int key[VG_STEPS];
  float px[VG_STEPS], py[VG_STEPS], pz[VG_STEPS];
#pragma unroll
  for (int j = 0; j < VG_STEPS; j++) {
    const int i = base_i + j * 64;
    const bool in = i < n;
    key[j] = in ? (unsigned int)keys[i] : 0xFFFFu;
    px[j] = in ? x[i] : 0.f; py[j] = in ? y[i] : 0.f; pz[j] = in ? z[i] : 0.f;
  }
  __syncthreads();
  const int sh = 16 * w;
#pragma unroll
  for (int j = 0; j < VG_STEPS; j++)
    if (key[j] != 0xFFFFu) atomicAdd(&s_c[key[j]], 1ull << sh);
  __syncthreads();
  for (int k = tid; k < C; k += 256) {
    const unsigned long long v = s_c[k];
    const unsigned long long c0 = v & 0xFFFFull, c1 = (v >> 16) & 0xFFFFull, c2 = (v >> 32) & 0xFFFFull;
    s_c[k] = (c0 << 16) | ((c0 + c1) << 32) | ((c0 + c1 + c2) << 48);
  }
  // absolute position of this block's first point of every key this lane holds (one round trip for all 16 steps)
  unsigned int absb[VG_STEPS];
  const unsigned int* boff = blkoff + (size_t)blk_x * C;
#pragma unroll
  for (int j = 0; j < VG_STEPS; j++) absb[j] = (key[j] != 0xFFFFu) ? (start[key[j]] + boff[key[j]]) : 0u;
  __syncthreads();
  // One returning LDS atomic per step ranks the lanes of equal key: the LDS unit resolves same-address lanes of one
  // instruction one after another in a fixed order, so the rank of a point inside its cell is a fixed function of the
  // input — the build is bit-reproducible.  (A ballot loop over the distinct keys of a step gives strict point-index
  // order but cost 50 us for this kernel; the leaf sums do not depend on the order beyond fp64 rounding.)
#pragma unroll
  for (int j = 0; j < VG_STEPS; j++) {
    const unsigned int k = key[j];
    if (k != 0xFFFFu) {
      const unsigned long long old = atomicAdd(&s_c[k], 1ull << sh);
      const unsigned int pos = absb[j] + ((unsigned int)(old >> sh) & 0xFFFFu);
      ox[pos] = px[j]; oy[pos] = py[j]; oz[pos] = pz[j];
      if (oidx) oidx[pos] = base_i + j * 64;
    }
  }
}
__global__ __launch_bounds__(256) void vg_scatter_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                         int n, const unsigned short* __restrict__ keys, const unsigned int* __restrict__ blkoff,
                                                         const unsigned int* __restrict__ start, int C, float* __restrict__ ox,
                                                         float* __restrict__ oy, float* __restrict__ oz, int* __restrict__ oidx) {
  vg_scatter_kernel_body(x, y, z, n, keys, blkoff, start, C, ox, oy, oz, oidx, (int)blockIdx.x);
}

// K1 + K2, one WAVE per grid cell: lane l sums the cell's points l + 64*t (cell order = point order), the lanes are combined by a
// fixed butterfly, lane 0 finalises the leaf (leaf_finalize_dev).  (Round 5 measured summing the cells of >= 2048 points with the
// whole workgroup — sums defined on 256 virtual lanes so that both forms return the same bits, four accumulator sets per lane in the
// wave form, predicated batches instead of serial tails: bit-exact, and slower: 25 us against 22 us for one 661k-point target,
// 0.28 ms against 0.25 ms for the targets of an 8-candidate share.  The large cells are not what this kernel waits for.)  Dense record layout
// (record of cell c at rec[4c]); empty cells are written as zero records.
constexpr int VG_LEAF_THREADS = 256;
// SUMS_ONLY (round 6, candidate sets): the wave leaves its nine sums in icov64[9 cell ..] and the finalisation to vg_leaf_finish_body — one
// THREAD per cell in a launch of its own.  A wave whose lanes 1..63 have left still takes a full wave's issue slots for every one of the
// ~1000 serial fp64 instructions of leaf_finalize_dev; with eight targets' cells in one launch (a dozen such waves per SIMD) that tail
// was most of the kernel.  Same function on the same sums: the same records.
template <bool SUMS_ONLY = false>
__device__ __forceinline__ void vg_leaf_kernel_body(const float* __restrict__ sx, const float* __restrict__ sy,
                                                                  const float* __restrict__ sz, const unsigned int* __restrict__ start,
                                                                  int ncells, int min_points, double eig_mult, float4* __restrict__ rec,
                                                                  double* __restrict__ mean64, double* __restrict__ icov64,
                                                                  int* __restrict__ leaf_key, int* __restrict__ leaf_n,
                                                                  int* __restrict__ cell_slot, const int blk_x) {
  // one WAVE per cell, four cells per workgroup: a cell is ~1000 points (16 per lane), its finalisation one lane's serial fp64
  // work — four times as many cells in flight per compute unit as with a workgroup per cell, no LDS, no barrier
  const int cell = blk_x * (VG_LEAF_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (cell >= ncells) return;
  const unsigned int off = start[cell];
  const int cnt = (int)(start[cell + 1] - off);
  if (cnt == 0) {
    if (SUMS_ONLY) return;   // (the finishing launch writes the empty cell's record)
    if (lane == 0) {
      const double zero[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      leaf_record_dev(zero, zero, 0, false, rec + (size_t)cell * 4);   // NaN pieces: an empty cell answers no lookup
      leaf_key[cell] = -1; leaf_n[cell] = 0; cell_slot[cell] = -1;
    }
    return;
  }
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const float* bx = sx + off; const float* by = sy + off; const float* bz = sz + off;
  int j = lane;
  for (; j + 7 * 64 < cnt; j += 8 * 64) {  // eight independent loads per plane in flight
    float fx[8], fy[8], fz[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { fx[u] = bx[j + u * 64]; fy[u] = by[j + u * 64]; fz[u] = bz[j + u * 64]; }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const double px = (double)fx[u], py = (double)fy[u], pz = (double)fz[u];
      s[0] += px; s[1] += py; s[2] += pz;
      s[3] += px * px; s[4] += px * py; s[5] += px * pz;
      s[6] += py * py; s[7] += py * pz; s[8] += pz * pz;
    }
  }
  for (; j < cnt; j += 64) {
    const double px = (double)bx[j], py = (double)by[j], pz = (double)bz[j];
    s[0] += px; s[1] += py; s[2] += pz;
    s[3] += px * px; s[4] += px * py; s[5] += px * pz;
    s[6] += py * py; s[7] += py * pz; s[8] += pz * pz;
  }
#pragma unroll
  for (int k = 0; k < 9; k++) s[k] = wave_sum64(s[k]);
  if (lane != 0) return;
  if (SUMS_ONLY) {
#pragma unroll
    for (int k = 0; k < 9; k++) icov64[(size_t)cell * 9 + k] = s[k];
    return;
  }
  double mean[3], icov[9];
  bool valid;
  const int n = leaf_finalize_dev(s, cnt, min_points, eig_mult, mean, icov, &valid);
  leaf_key[cell] = cell;
  leaf_n[cell] = n;
  for (int k = 0; k < 3; k++) mean64[(size_t)cell * 3 + k] = mean[k];
  for (int k = 0; k < 9; k++) icov64[(size_t)cell * 9 + k] = icov[k];
  leaf_record_dev(mean, icov, n, valid, rec + (size_t)cell * 4);
  cell_slot[cell] = valid ? cell : -1;
}
__global__ __launch_bounds__(VG_LEAF_THREADS) void vg_leaf_kernel(const float* __restrict__ sx, const float* __restrict__ sy,
                                                                  const float* __restrict__ sz, const unsigned int* __restrict__ start,
                                                                  int ncells, int min_points, double eig_mult, float4* __restrict__ rec,
                                                                  double* __restrict__ mean64, double* __restrict__ icov64,
                                                                  int* __restrict__ leaf_key, int* __restrict__ leaf_n,
                                                                  int* __restrict__ cell_slot) {
  vg_leaf_kernel_body(sx, sy, sz, start, ncells, min_points, eig_mult, rec, mean64, icov64, leaf_key, leaf_n, cell_slot, (int)blockIdx.x);
}

// the second half of the SUMS_ONLY form: one thread per cell
__device__ __forceinline__ void vg_leaf_finish_body(const unsigned int* __restrict__ start, int ncells, int min_points, double eig_mult,
                                                    float4* __restrict__ rec, double* __restrict__ mean64, double* __restrict__ icov64,
                                                    int* __restrict__ leaf_key, int* __restrict__ leaf_n, int* __restrict__ cell_slot, const int cell) {
  if (cell >= ncells) return;
  const int cnt = (int)(start[cell + 1] - start[cell]);
  if (cnt == 0) {
    const double zero[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    leaf_record_dev(zero, zero, 0, false, rec + (size_t)cell * 4);   // NaN pieces: an empty cell answers no lookup
    leaf_key[cell] = -1; leaf_n[cell] = 0; cell_slot[cell] = -1;
    return;
  }
  double s[9], mean[3], icov[9];
#pragma unroll
  for (int k = 0; k < 9; k++) s[k] = icov64[(size_t)cell * 9 + k];
  bool valid;
  const int n = leaf_finalize_dev(s, cnt, min_points, eig_mult, mean, icov, &valid);
  leaf_key[cell] = cell;
  leaf_n[cell] = n;
  for (int k = 0; k < 3; k++) mean64[(size_t)cell * 3 + k] = mean[k];
  for (int k = 0; k < 9; k++) icov64[(size_t)cell * 9 + k] = icov[k];
  leaf_record_dev(mean, icov, n, valid, rec + (size_t)cell * 4);
  cell_slot[cell] = valid ? cell : -1;
}

// ---- the same builders over a GROUP of targets: blockIdx.y selects the member, whose parameters travel in the kernel arguments ----
struct VgMember {
  const unsigned char* aos; size_t stride;        // ingest: strided xyz records in device memory
  float *x, *y, *z;                               // SoA planes of the target cloud
  int n, nblk, ingest_blocks;                     // points, counting-sort chunks, workgroups of the ingest pass
  float inv_leaf; int mb0, mb1, mb2, mul1, mul2, ncells;
  unsigned short *keys, *hist; unsigned int *blkoff, *total, *start, *rank;
  float *sx, *sy, *sz; int* sidx;
  float4* rec; double *mean64, *icov64; int *leaf_key, *leaf_n, *cell_slot;
  BuildMailbox* mb; unsigned int bbox_token;
};
struct VgGroup { VgMember m[LSR_GROUP]; };
static_assert(sizeof(VgGroup) <= 3800, "a group's parameters must fit the kernel argument segment");

// ingest = de-interleave + bounding box in ONE pass over the strided records (a single target runs them as two kernels):
// workgroup b of a member walks points b*256 + tid + k * ingest_blocks*256, writes the SoA planes and leaves one BboxPart
// record in the member's host mailbox (same record, same host-side fold as bbox_kernel's)
__global__ __launch_bounds__(256) void vg_ingest_group_kernel(const VgGroup g) {
  const VgMember& M = g.m[blockIdx.y];
  if ((int)blockIdx.x >= M.ingest_blocks) return;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  unsigned int cnt = 0;
  const int step = M.ingest_blocks * 256;
  // records of 16-byte aligned stride (pcl::PointXYZI: 32 bytes) are read as ONE 16-byte load per point, anything else as three
  const bool wide = ((M.stride & 15) == 0) && ((reinterpret_cast<size_t>(M.aos) & 15) == 0);
  for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < M.n; i0 += 4 * step) {
    float p[4][3];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * step;
      if (i < M.n) {
        if (wide) {
          const float4 q = *reinterpret_cast<const float4*>(M.aos + (size_t)i * M.stride);
          p[u][0] = q.x; p[u][1] = q.y; p[u][2] = q.z;
        } else {
          const float* q = reinterpret_cast<const float*>(M.aos + (size_t)i * M.stride);
          p[u][0] = q[0]; p[u][1] = q[1]; p[u][2] = q[2];
        }
      } else {
        p[u][0] = p[u][1] = p[u][2] = NAN;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * step;
      if (i < M.n) { M.x[i] = p[u][0]; M.y[i] = p[u][1]; M.z[i] = p[u][2]; }
      if (!(isfinite(p[u][0]) && isfinite(p[u][1]) && isfinite(p[u][2]))) continue;
      cnt++;
#pragma unroll
      for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], p[u][k]); mx[k] = fmaxf(mx[k], p[u][k]); }
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
  if (threadIdx.x < BBOX_GRANULES) {   // seven self-validating granules, no fence (common.hpp: BboxPart)
    const int k = threadIdx.x;
    unsigned int bits;
    if (k < 3) bits = __float_as_uint(fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k])));
    else if (k < 6) bits = __float_as_uint(fmaxf(fmaxf(s_mx[0][k - 3], s_mx[1][k - 3]), fmaxf(s_mx[2][k - 3], s_mx[3][k - 3])));
    else bits = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __hip_atomic_store(&M.mb->part[blockIdx.x].g[k], ((unsigned long long)M.bbox_token << 32) | bits, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(256) void vg_hist_group_kernel(const VgGroup g) {
  const VgMember& M = g.m[blockIdx.y];
  if ((int)blockIdx.x >= M.nblk) return;
  vg_hist_kernel_body(M.x, M.y, M.z, M.n, M.inv_leaf, M.mb0, M.mb1, M.mb2, M.mul1, M.mul2, M.ncells, M.keys, M.hist, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void vg_scan_group_kernel(const VgGroup g) {
  const VgMember& M = g.m[blockIdx.y];
  if ((int)blockIdx.x >= (M.ncells + 1 + 31) / 32) return;
  vg_scan_kernel_body(M.hist, M.nblk, M.ncells + 1, M.blkoff, M.total, (int)blockIdx.x);
}
__global__ __launch_bounds__(1024) void vg_cellscan_group_kernel(const VgGroup g) {
  const VgMember& M = g.m[blockIdx.x];
  vg_cellscan_kernel_body(M.total, M.ncells + 1, M.start, M.rank, 0);
}
__global__ __launch_bounds__(256) void vg_scatter_group_kernel(const VgGroup g) {
  const VgMember& M = g.m[blockIdx.y];
  if ((int)blockIdx.x >= M.nblk) return;
  vg_scatter_kernel_body(M.x, M.y, M.z, M.n, M.keys, M.blkoff, M.start, M.ncells + 1, M.sx, M.sy, M.sz, M.sidx, (int)blockIdx.x);
}
__global__ __launch_bounds__(VG_LEAF_THREADS) void vg_leaf_group_kernel(const VgGroup g) {
  const VgMember& M = g.m[blockIdx.y];
  if ((int)blockIdx.x * (VG_LEAF_THREADS / 64) >= M.ncells) return;
  vg_leaf_kernel_body(M.sx, M.sy, M.sz, M.start, M.ncells, 6, 0.01, M.rec, M.mean64, M.icov64, M.leaf_key, M.leaf_n, M.cell_slot, (int)blockIdx.x);
}
__global__ __launch_bounds__(VG_LEAF_THREADS) void vg_leaf_sums_group_kernel(const VgGroup g) {
  const VgMember& M = g.m[blockIdx.y];
  if ((int)blockIdx.x * (VG_LEAF_THREADS / 64) >= M.ncells) return;
  vg_leaf_kernel_body<true>(M.sx, M.sy, M.sz, M.start, M.ncells, 6, 0.01, M.rec, M.mean64, M.icov64, M.leaf_key, M.leaf_n, M.cell_slot, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void vg_leaf_finish_group_kernel(const VgGroup g) {
  const VgMember& M = g.m[blockIdx.y];
  vg_leaf_finish_body(M.start, M.ncells, 6, 0.01, M.rec, M.mean64, M.icov64, M.leaf_key, M.leaf_n, M.cell_slot, (int)(blockIdx.x * 256 + threadIdx.x));
}

// ---- source ordering for the tile-staged derivative pass (NDT_TAB_TILE) -------------------------------------------------
// key of a source point = Morton code of the (2^shift x 2^shift cells) x (all z) column of the target grid its image under
// the initial guess falls into, clamped to the grid; non-finite points get the sentinel key (sorted last).  The counting
// sort itself is the grid builder's (vg_scan / vg_cellscan / vg_scatter): stable, deterministic, no host round trip.
struct SrcSortT { float t[12]; };
__device__ __forceinline__ unsigned int spread7(unsigned int v) {   // bit i of v -> bit 2 i
  v = (v | (v << 4)) & 0x0F0Fu;
  v = (v | (v << 2)) & 0x3333u;
  v = (v | (v << 1)) & 0x5555u;
  return v;
}
__global__ __launch_bounds__(256) void src_hist_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                       int n, const SrcSortT T, float leaf, int mb0, int mb1, int d0, int d1, int shift, int nkeys,
                                                       unsigned short* __restrict__ keys, unsigned short* __restrict__ hist) {
  extern __shared__ unsigned int s_hist[];  // [nkeys + 1]
  const int C = nkeys + 1, tid = threadIdx.x;
  for (int k = tid; k < C; k += 256) s_hist[k] = 0u;
  __syncthreads();
  const int base = blockIdx.x * VG_CHUNK;
#pragma unroll 4
  for (int j = 0; j < VG_CHUNK / 256; j++) {
    const int i = base + j * 256 + tid;
    if (i < n) {
      const float px = x[i], py = y[i], pz = z[i];
      unsigned int k = (unsigned int)nkeys;
      if (isfinite(px) && isfinite(py) && isfinite(pz)) {
        const float tx = fmaf(T.t[0], px, fmaf(T.t[1], py, fmaf(T.t[2], pz, T.t[3])));
        const float ty = fmaf(T.t[4], px, fmaf(T.t[5], py, fmaf(T.t[6], pz, T.t[7])));
        const float fx = fminf(fmaxf(floorf(tx / leaf) - (float)mb0, 0.f), (float)(d0 - 1));   // NaN / inf images clamp too
        const float fy = fminf(fmaxf(floorf(ty / leaf) - (float)mb1, 0.f), (float)(d1 - 1));
        const unsigned int cx = (unsigned int)(int)fx >> shift, cy = (unsigned int)(int)fy >> shift;
        k = spread7(cx) | (spread7(cy) << 1);
        if (k >= (unsigned int)nkeys) k = (unsigned int)nkeys - 1u;
      }
      keys[i] = (unsigned short)k;
      atomicAdd(&s_hist[k], 1u);
    }
  }
  __syncthreads();
  unsigned short* row = hist + (size_t)blockIdx.x * C;
  for (int k = tid; k < C; k += 256) row[k] = (unsigned short)s_hist[k];
}

}  // namespace

// Everything after the bounding box for a dense key space.  grid.min_b / max_b / div_b / ncells are set by the caller.
int ndt_build_grid_dense(const DeviceCloud& cloud, float leaf, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream) {
  const int n = (int)cloud.n;
  const int ncells = (int)grid.ncells, C = ncells + 1;
  const int nblk = (n + VG_CHUNK - 1) / VG_CHUNK;
  const float inv_leaf = 1.0f / leaf;
  int st;
  // scratch words: total[C] | blkoff[nblk*C] | hist(u16)[nblk*C] | keys(u16)[n]; the cell-ordered points, their indices, the cell
  // starts and ranks stay with the grid (the neighbour grid of getFitnessScore refines them)
  const size_t w_total = (size_t)C, w_blkoff = (size_t)nblk * C, w_hist = ((size_t)nblk * C + 1) / 2, w_keys = ((size_t)n + 1) / 2;
  if ((st = sc.words.reserve(16 + w_total + w_blkoff + w_hist + w_keys + 16))) return st;
  unsigned int* total = sc.words.p + 16;
  unsigned int* blkoff = total + w_total;
  unsigned short* hist = reinterpret_cast<unsigned short*>(blkoff + w_blkoff);
  unsigned short* keys = reinterpret_cast<unsigned short*>(blkoff + w_blkoff + w_hist);
  const size_t pitch = ((size_t)n + 63) & ~(size_t)63;
  if ((st = grid.sorted.reserve(3 * pitch))) return st;
  if ((st = grid.sorted_idx.reserve(pitch))) return st;
  if ((st = grid.cell_start.reserve((size_t)C + 1))) return st;
  if ((st = grid.cell_rank.reserve((size_t)C))) return st;
  grid.sorted_pitch = pitch; grid.sorted_n = (size_t)n; grid.has_sorted = true;
  unsigned int* start = grid.cell_start.p;
  float* sx = grid.sorted.p; float* sy = sx + pitch; float* sz = sy + pitch;
  if ((st = grid.cell_slot.reserve(grid.ncells))) return st;
  if ((st = grid.rec.reserve(grid.ncells * 4))) return st;
  if ((st = grid.mean64.reserve(grid.ncells * 3))) return st;
  if ((st = grid.icov64.reserve(grid.ncells * 9))) return st;
  if ((st = grid.leaf_key.reserve(grid.ncells))) return st;
  if ((st = grid.leaf_n.reserve(grid.ncells))) return st;
  grid.dense = true;

  static bool attr_done[64] = {};
  int dev = 0;
  LSR_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !attr_done[dev]) {
    LSR_HIP(hipFuncSetAttribute((const void*)vg_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VG_DENSE_MAX_CELLS * 4 + 4));
    LSR_HIP(hipFuncSetAttribute((const void*)vg_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VG_DENSE_MAX_CELLS * 8 + 8));
    attr_done[dev] = true;
  }
  const int mul1 = grid.div_b[0], mul2 = grid.div_b[0] * grid.div_b[1];
  hipLaunchKernelGGL(vg_hist_kernel, dim3(nblk), dim3(256), (size_t)C * 4, stream, cloud.x(), cloud.y(), cloud.z(), n, inv_leaf,
                       grid.min_b[0], grid.min_b[1], grid.min_b[2], mul1, mul2, ncells, keys, hist);
  hipLaunchKernelGGL(vg_scan_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, hist, nblk, C, blkoff, total);
  hipLaunchKernelGGL(vg_cellscan_kernel, dim3(1), dim3(1024), 0, stream, total, C, start, grid.cell_rank.p);
  hipLaunchKernelGGL(vg_scatter_kernel, dim3(nblk), dim3(256), (size_t)C * 8, stream, cloud.x(), cloud.y(), cloud.z(), n, keys, blkoff,
                     start, C, sx, sy, sz, grid.sorted_idx.p);
  hipLaunchKernelGGL(vg_leaf_kernel, dim3((ncells + VG_LEAF_THREADS / 64 - 1) / (VG_LEAF_THREADS / 64)), dim3(VG_LEAF_THREADS), 0, stream, sx, sy, sz, start, ncells, 6, 0.01, grid.rec.p,
                     grid.mean64.p, grid.icov64.p, grid.leaf_key.p, grid.leaf_n.p, grid.cell_slot.p);
  LSR_HIP(hipGetLastError());
  grid.n_leaves = ncells;  // leaf arrays are indexed by cell; empty cells carry leaf_key = -1
  return LSR_OK;
}

int ndt_sort_source(const DeviceCloud& src, const float* T12, const VoxelGridDev& grid, DeviceCloud& out, BuildScratch& sc, hipStream_t stream) {
  const int n = (int)src.n;
  int st = out.resize(src.n);
  if (st) return st;
  if (n == 0) return LSR_OK;
  // columns of 2^shift x 2^shift cells, at most 64 x 64 of them: Morton keys below 4096
  int shift = 0;
  while (((grid.div_b[0] + (1 << shift) - 1) >> shift) > 64 || ((grid.div_b[1] + (1 << shift) - 1) >> shift) > 64) shift++;
  const int nkeys = 4096, C = nkeys + 1;
  const int nblk = (n + VG_CHUNK - 1) / VG_CHUNK;
  const size_t w_total = (size_t)C, w_start = (size_t)C + 1, w_blkoff = (size_t)nblk * C, w_hist = ((size_t)nblk * C + 1) / 2, w_keys = ((size_t)n + 1) / 2;
  if ((st = sc.words.reserve(16 + w_total + w_start + w_blkoff + w_hist + w_keys + 16))) return st;
  unsigned int* total = sc.words.p + 16;
  unsigned int* start = total + w_total;
  unsigned int* blkoff = start + w_start;
  unsigned short* hist = reinterpret_cast<unsigned short*>(blkoff + w_blkoff);
  unsigned short* keys = reinterpret_cast<unsigned short*>(blkoff + w_blkoff + w_hist);
  SrcSortT T;
  for (int k = 0; k < 12; k++) T.t[k] = T12[k];
  hipLaunchKernelGGL(src_hist_kernel, dim3(nblk), dim3(256), (size_t)C * 4, stream, src.x(), src.y(), src.z(), n, T, grid.leaf, grid.min_b[0],
                     grid.min_b[1], grid.div_b[0], grid.div_b[1], shift, nkeys, keys, hist);
  hipLaunchKernelGGL(vg_scan_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, hist, nblk, C, blkoff, total);
  hipLaunchKernelGGL(vg_cellscan_kernel, dim3(1), dim3(1024), 0, stream, total, C, start, (unsigned int*)nullptr);
  hipLaunchKernelGGL(vg_scatter_kernel, dim3(nblk), dim3(256), (size_t)C * 8, stream, src.x(), src.y(), src.z(), n, keys, blkoff, start, C,
                     out.x(), out.y(), out.z(), (int*)nullptr);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

// ---- host side of the group kernels -----------------------------------------------------------------------------------
int ndt_targets_ingest(TargetBuildJob* jobs, int count, hipStream_t stream) {
  int st;
  for (int g0 = 0; g0 < count; g0 += LSR_GROUP) {
    VgGroup grp;
    std::memset(&grp, 0, sizeof(grp));
    const int ng = std::min(LSR_GROUP, count - g0);
    int max_blocks = 0;
    for (int k = 0; k < ng; k++) {
      TargetBuildJob& J = jobs[g0 + k];
      if ((st = J.cloud->resize(J.n))) return st;
      J.sc->bbox_parts = 0;
      VgMember& M = grp.m[k];
      M.n = (int)J.n;
      if (J.n == 0) continue;
      if ((st = J.sc->ensure_mailbox())) return st;
      unsigned int token = ++J.sc->token;
      if (token == 0) token = ++J.sc->token;
      M.aos = static_cast<const unsigned char*>(J.d_aos); M.stride = J.stride;
      M.x = J.cloud->x(); M.y = J.cloud->y(); M.z = J.cloud->z();
      // workgroups (= bounding-box records the host folds) per member: a full group fills the chip with far fewer each
      const size_t per_block = (ng >= 8) ? 8192 : 1024;
      M.ingest_blocks = std::max(1, std::min((int)((J.n + per_block - 1) / per_block), BBOX_MAX_PARTS));
      M.mb = J.sc->d_mb; M.bbox_token = token;
      J.sc->bbox_parts = M.ingest_blocks;
      J.sc->bbox_token = token;
      max_blocks = std::max(max_blocks, M.ingest_blocks);
    }
    if (max_blocks > 0) hipLaunchKernelGGL(vg_ingest_group_kernel, dim3(max_blocks, ng), dim3(256), 0, stream, grp);
  }
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

// counting sort + leaf sums + finalisation of `count` targets with dense key spaces (geometry set), five launches per group
int ndt_build_grids_dense_group(TargetBuildJob* const* jobs, int count, hipStream_t stream) {
  static bool attr_done[64] = {};
  int dev = 0;
  LSR_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !attr_done[dev]) {
    LSR_HIP(hipFuncSetAttribute((const void*)vg_hist_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VG_DENSE_MAX_CELLS * 4 + 4));
    LSR_HIP(hipFuncSetAttribute((const void*)vg_scatter_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VG_DENSE_MAX_CELLS * 8 + 8));
    attr_done[dev] = true;
  }
  int st;
  for (int g0 = 0; g0 < count; g0 += LSR_GROUP) {
    VgGroup grp;
    std::memset(&grp, 0, sizeof(grp));
    const int ng = std::min(LSR_GROUP, count - g0);
    int max_nblk = 0, max_cells = 0;
    for (int k = 0; k < ng; k++) {
      const TargetBuildJob& J = *jobs[g0 + k];
      VoxelGridDev& grid = *J.grid;
      BuildScratch& sc = *J.sc;
      const DeviceCloud& cloud = *J.cloud;
      const int n = (int)cloud.n;
      const int ncells = (int)grid.ncells, C = ncells + 1;
      const int nblk = (n + VG_CHUNK - 1) / VG_CHUNK;
      // the same scratch layout as the single-target builder
      const size_t w_total = (size_t)C, w_blkoff = (size_t)nblk * C, w_hist = ((size_t)nblk * C + 1) / 2, w_keys = ((size_t)n + 1) / 2;
      if ((st = sc.words.reserve(16 + w_total + w_blkoff + w_hist + w_keys + 16))) return st;
      const size_t pitch = ((size_t)n + 63) & ~(size_t)63;
      if ((st = grid.sorted.reserve(3 * pitch))) return st;
      if ((st = grid.sorted_idx.reserve(pitch))) return st;
      if ((st = grid.cell_start.reserve((size_t)C + 1))) return st;
      if ((st = grid.cell_rank.reserve((size_t)C))) return st;
      grid.sorted_pitch = pitch; grid.sorted_n = (size_t)n; grid.has_sorted = true;
      if ((st = grid.cell_slot.reserve(grid.ncells))) return st;
      if ((st = grid.rec.reserve(grid.ncells * 4))) return st;
      if ((st = grid.mean64.reserve(grid.ncells * 3))) return st;
      if ((st = grid.icov64.reserve(grid.ncells * 9))) return st;
      if ((st = grid.leaf_key.reserve(grid.ncells))) return st;
      if ((st = grid.leaf_n.reserve(grid.ncells))) return st;
      grid.dense = true;
      grid.n_leaves = ncells;
      VgMember& M = grp.m[k];
      M.x = cloud.x(); M.y = cloud.y(); M.z = cloud.z();
      M.n = n; M.nblk = nblk;
      M.inv_leaf = 1.0f / J.leaf;
      M.mb0 = grid.min_b[0]; M.mb1 = grid.min_b[1]; M.mb2 = grid.min_b[2];
      M.mul1 = grid.div_b[0]; M.mul2 = grid.div_b[0] * grid.div_b[1]; M.ncells = ncells;
      M.total = sc.words.p + 16;
      M.blkoff = M.total + w_total;
      M.hist = reinterpret_cast<unsigned short*>(M.blkoff + w_blkoff);
      M.keys = reinterpret_cast<unsigned short*>(M.blkoff + w_blkoff + w_hist);
      M.start = grid.cell_start.p; M.rank = grid.cell_rank.p;
      M.sx = grid.sorted.p; M.sy = M.sx + pitch; M.sz = M.sy + pitch; M.sidx = grid.sorted_idx.p;
      M.rec = grid.rec.p; M.mean64 = grid.mean64.p; M.icov64 = grid.icov64.p;
      M.leaf_key = grid.leaf_key.p; M.leaf_n = grid.leaf_n.p; M.cell_slot = grid.cell_slot.p;
      max_nblk = std::max(max_nblk, nblk);
      max_cells = std::max(max_cells, ncells);
    }
    const int maxC = max_cells + 1;
    hipLaunchKernelGGL(vg_hist_group_kernel, dim3(max_nblk, ng), dim3(256), (size_t)maxC * 4, stream, grp);
    hipLaunchKernelGGL(vg_scan_group_kernel, dim3((maxC + 31) / 32, ng), dim3(256), 0, stream, grp);
    hipLaunchKernelGGL(vg_cellscan_group_kernel, dim3(ng), dim3(1024), 0, stream, grp);
    hipLaunchKernelGGL(vg_scatter_group_kernel, dim3(max_nblk, ng), dim3(256), (size_t)maxC * 8, stream, grp);
    // from two targets on: the sums by one wave per cell, the finalisation by one thread per cell in a launch of its own (LSR_VG_LEAF_SPLIT=0 / 1 forces)
    static const int split_env = [] { const char* e = getenv("LSR_VG_LEAF_SPLIT"); return e ? atoi(e) : -1; }();
    if (split_env == 1 || (split_env < 0 && ng >= 2)) {
      hipLaunchKernelGGL(vg_leaf_sums_group_kernel, dim3((max_cells + VG_LEAF_THREADS / 64 - 1) / (VG_LEAF_THREADS / 64), ng), dim3(VG_LEAF_THREADS), 0, stream, grp);
      hipLaunchKernelGGL(vg_leaf_finish_group_kernel, dim3((max_cells + 255) / 256, ng), dim3(256), 0, stream, grp);
    } else {
      hipLaunchKernelGGL(vg_leaf_group_kernel, dim3((max_cells + VG_LEAF_THREADS / 64 - 1) / (VG_LEAF_THREADS / 64), ng), dim3(VG_LEAF_THREADS), 0, stream, grp);
    }
  }
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

}  // namespace lsr
