// Shared host-side plumbing for the gfx950 registration core: error propagation without
// exceptions across the C ABI, RAII device buffers, and the structs shared by host and device.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lidarslam_reg.h"

namespace lsr {

void set_last_error(const std::string& s);

#define LSR_HIP(expr)                                                                                         \
  do {                                                                                                        \
    hipError_t _e = (expr);                                                                                   \
    if (_e != hipSuccess) {                                                                                   \
      ::lsr::set_last_error(std::string(#expr) + " -> " + hipGetErrorString(_e) + " (" + __FILE__ + ":" +     \
                            std::to_string(__LINE__) + ")");                                                  \
      return LSR_ERR_HIP;                                                                                     \
    }                                                                                                         \
  } while (0)

// Makes `dev` the calling thread's current HIP device for the lifetime of the guard and restores the caller's afterwards.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Growable device allocation (never shrinks; HBM is plentiful: 288 GB per MI355X).
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int reserve(size_t n) {
    if (n <= cap) return LSR_OK;
    if (p) LSR_HIP(hipFree(p));
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    LSR_HIP(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return LSR_OK;
  }
};

// Pinned host staging buffer.
template <typename T>
struct PinBuf {
  T* p = nullptr;
  size_t cap = 0;
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() {
    if (p) (void)hipHostFree(p);
  }
  int reserve(size_t n, unsigned int flags = hipHostMallocDefault) {
    if (n <= cap) return LSR_OK;
    if (p) LSR_HIP(hipHostFree(p));
    p = nullptr;
    cap = 0;
    LSR_HIP(hipHostMalloc((void**)&p, n * sizeof(T), flags));
    cap = n;
    return LSR_OK;
  }
};

// SoA fp32 cloud resident in HBM (x[], y[], z[] planes of one allocation).
struct DeviceCloud {
  DevBuf<float> buf;
  size_t n = 0;
  bool has_i = false;        // a fourth plane carries the intensity field (PointCloud2 codec path)
  float* x() const { return buf.p; }
  float* y() const { return buf.p + pitch; }
  float* z() const { return buf.p + 2 * pitch; }
  float* i() const { return has_i ? buf.p + 3 * pitch : nullptr; }
  size_t pitch = 0;
  // bounding box of the finite points, remembered by cloud_bbox() until the next resize() (every writer of a cloud
  // resizes it first): setInputTarget's grid build and the NN-grid build of getFitnessScore share one pass and one poll
  mutable bool bbox_valid = false;
  mutable float bbox_mn[3] = {0, 0, 0}, bbox_mx[3] = {0, 0, 0};
  mutable unsigned int bbox_finite = 0;
  // the pass that wrote this cloud left its per-workgroup bounding-box records behind (pc2_ingest): in the scratch's host mailbox
  // (cloud_bbox_end collects them, no bbox launch) and in device memory (voxel_grid_filter's device-side dimensions)
  mutable bool bbox_enqueued = false;
  // fewer points than the planes were laid out for: the planes stay where they are (a kernel that was enqueued before the count
  // was known has written them at this pitch)
  int shrink(size_t count) {
    if (count > n) return LSR_ERR_INVALID_ARGUMENT;
    n = count;
    return LSR_OK;
  }
  int resize(size_t count, bool with_intensity = false) {
    bbox_valid = false;
    bbox_enqueued = false;
    size_t pt = (count + 63) & ~size_t(63);
    if (pt == 0) pt = 64;
    int st = buf.reserve(4 * pt);   // room for the intensity plane whether or not this cloud uses it
    if (st) return st;
    pitch = pt;
    n = count;
    has_i = with_intensity;
    return LSR_OK;
  }
};

// ---- target-side NDT structure (K1/K2 output) ------------------------------------------------
// One 64-byte record per leaf: {mean_hi.xyz, c00}, {c01, c02, c11, c12}, {c22, mean_lo.xyz}, {n, -, -, -}: the mean as head +
// tail fp32 (x' - mean is formed as (x' - hi) - lo, ndt_point.hpp); an unusable leaf is all-NaN and drops itself.
struct VoxelGridDev {
  float leaf = 1.f;
  int min_b[3] = {0, 0, 0}, max_b[3] = {-1, -1, -1}, div_b[3] = {0, 0, 0};
  size_t ncells = 0;
  int n_leaves = 0;        // occupied leaves (any count)
  int n_valid = 0;         // leaves usable by lookups
  bool dense = false;      // rec[] indexed by cell (dense) or by leaf slot (compact)
  DevBuf<int> cell_slot;   // dense [ncells] -> record slot or -1
  DevBuf<float4> rec;      // [n_leaves * 4]
  // LDS image of the valid-voxel table (ndt_pack_lds_table): uint16 cell->slot map (0xFFFF = none) followed by 48-byte
  // records of the usable leaves; lds_bytes == 0 when the table is too large to stage
  DevBuf<uint4> lds_image;
  int lds_map_bytes = 0, lds_bytes = 0;
  // fp64 copies for inspection/parity (mean 3, icov 9 row-major) + key + count per leaf
  DevBuf<double> mean64, icov64;
  DevBuf<int> leaf_key, leaf_n;
  // KDTREE neighbourhood only (built on first use, ndt_build_centroids): Leaf::centroid of every usable leaf — the FLOAT running sum
  // of its points in cloud order over (float) count, what the reference's voxel-centroid kd-tree holds — indexed like rec[] (by
  // cell_slot[cell]); .w unused
  DevBuf<float4> centroid;
  // what the counting-sort builder leaves behind (dense key spaces): the target's points in cell order (x | y | z planes of
  // `sorted_pitch` floats), their original indices, the start of every cell (ncells + 2 entries: [ncells] = first non-finite
  // point, [ncells + 1] = n) and the rank of every cell among the occupied ones.  getFitnessScore's neighbour grid is a
  // refinement of exactly this ordering (nn_build_hash_from_grid), so it never sorts the target a second time.
  bool has_sorted = false;
  size_t sorted_pitch = 0, sorted_n = 0;
  DevBuf<float> sorted;
  DevBuf<int> sorted_idx;
  DevBuf<unsigned int> cell_start, cell_rank;
};

// ---- NN grid over a cloud (fitness score, GICP): two-level blocked voxel grid ------------------
// Coarse cells (8x8x8 fine cells) are a dense int32 map -> block id; every occupied coarse cell owns a
// 513-entry table of fine-cell start offsets into the cell-sorted point arrays.
struct HashGridDev {
  float cell = 0.5f;       // fine cell edge [m]; coarse edge = 8 * cell
  int org[3] = {0, 0, 0};  // fine-cell coordinate of the grid origin (multiple of 8)
  int cdim[3] = {0, 0, 0}; // coarse dims
  int n_blocks = 0;        // occupied coarse cells
  size_t n = 0;
  DevBuf<int> coarse_block; // [cdim0*cdim1*cdim2] -> block id or -1
  DevBuf<int> block_off;    // [n_blocks + 1] start of each block in the sorted arrays
  DevBuf<int> fine_start;   // [n_blocks * 513] absolute start of each fine cell (+ end sentinel)
  DevBuf<float4> packed;    // points in (coarse, fine) cell order: {x, y, z, original index as int bits} — one 16-byte
                            // load per candidate, no dependent index load
  DevBuf<int> order;        // sorted position -> original index
};

// Host mailbox of the target-side builders (pinned, host-coherent memory mapped into the device): the kernels publish
// the few scalars the host needs (bounding box, leaf counts) straight into host memory and the host polls a token —
// no device-to-host copy and no stream synchronisation inside a grid build.
constexpr int BBOX_MAX_PARTS = 256;
// One workgroup's share of a bounding box, written straight into host memory as seven self-validating 8-byte granules
// {value bits (low half), token (high half)}: min xyz, max xyz, #finite.  Every granule is ONE naturally aligned 8-byte
// system-scope store — no release fence in front of a flag: a `__threadfence_system()` in every workgroup made each of them
// write back its XCD's whole L2, which the fused ingest pass (de-interleave + box) had just filled with dirty lines (measured:
// 27 us per 661k-point target inside a group of 16, against 6 + 12 us for the two passes on their own).
struct BboxPart {
  unsigned long long g[8];
};
constexpr int BBOX_GRANULES = 7;
struct BuildMailbox {
  BboxPart part[BBOX_MAX_PARTS];
  int n_valid, n_occupied;    // leaves usable by lookups / leaves holding at least one point
  int lds_bytes, lds_map_bytes;
  unsigned int done_token;    // release-stored after the counts
  // small scalars other builders hand to the host the same way (NN grid: occupied coarse cells; fitness score: sum, count)
  int value;
  unsigned int value_token;
  double fit_sum, fit_cnt;
  unsigned int fit_token;
  // voxel_grid_filter with device-side dimensions: what the host needs next to `value` (= runs of equal keys), same token
  unsigned int vg_flags;      // VG_FLAG_*
  unsigned int vg_finite;     // finite points of the input
  unsigned int vg_bits;       // bits of the largest key (the next call on this scratch plans its sort with it)
};
constexpr unsigned int VG_FLAG_OVERFLOW = 1u;   // leaf index space beyond int32 (PCL: "Leaf size is too small for the input dataset")
constexpr unsigned int VG_FLAG_REPLAN = 2u;     // the keys need more bits than the sort was planned for: run again with exact dimensions

// How a host thread waits on a mailbox word (lsr_set_i32(LSR_WAIT_MODE)): a ROS2 MultiThreadedExecutor runs two
// registration objects side by side (lidarslam/src/lidarslam.cpp:12-17), and a spinning wait pins one core per align.
enum WaitMode : int { WAIT_SPIN = 0, WAIT_YIELD = 1, WAIT_SLEEP = 2 };

// Per-handle scratch for the target-side builders.
struct BuildScratch {
  DevBuf<char> temp;
  DevBuf<unsigned int> words;
  DevBuf<double> sums;
  DevBuf<float> sorted;            // dense-grid counting sort: cell-sorted x | y | z planes
  bool force_sort_path = false;    // tests: build dense key spaces with the general (radix sort) builder too
  PinBuf<BuildMailbox> mb;         // host view
  BuildMailbox* d_mb = nullptr;    // device view of mb.p
  unsigned int token = 0;
  int wait_mode = WAIT_SPIN;
  // staged builds (a batch enqueues every stage of every member before it waits): what is in flight on this scratch
  int bbox_parts = 0;              // workgroup records of an enqueued bounding-box pass not yet collected
  unsigned int bbox_token = 0;
  bool grid_pending = false;       // a dense voxel-grid build whose result has not been read from the mailbox yet
  unsigned int grid_token = 0;
  unsigned int fit_token = 0;      // an enqueued fitness reduction (0: none)
  DevBuf<unsigned long long> bbox_dev;   // device copy of the bounding-box records of the last pc2_ingest ([BBOX_MAX_PARTS][8])
  int vg_bits_hint = 0;            // key bits of the last voxel_grid_filter on this scratch (0: none yet) ...
  float vg_hint_leaf = 0.f;        // ... and the leaf size it was for: another leaf size is another index space
  int vg_form = 0;                 // which form that filter took (LSR_VOXEL_FILTER_FORM)
  int ensure_mailbox();
};

// Poll *word until it equals token (acquire).  Returns LSR_ERR_HIP if the stream reports an error or nothing happens
// for 30 s.  Defined in grid_dense.hip.
int wait_mailbox_word(const volatile unsigned int* word, unsigned int token, hipStream_t stream, int wait_mode, const char* what);

struct TargetData {
  std::mutex build_mutex;  // lazy builds (grid, NN hash, covariances) by handles that share this target (lsr_share_target)
  DeviceCloud cloud;
  size_t n = 0;
  bool has_grid = false;
  bool has_centroids = false;   // grid.centroid belongs to the grid in place
  VoxelGridDev grid;
  float grid_leaf = 0.f;
  bool has_hash = false;
  HashGridDev hash;
  bool has_cov = false;
  DevBuf<double> cov;      // GICP: n*9
  int cov_k = 0;
  double cov_eps = 0;
};

}  // namespace lsr
