// NDT on gfx950: device-side structures and host entry points.
//   K1/K2  voxel-covariance grid build        (replaces pclomp::VoxelGridCovariance::filter)
//   K3     derivative pass                    (replaces NDT::computeDerivatives/updateDerivatives, a5/a7)
//   K4     Newton + More-Thuente controller   (replaces NDT::computeTransformation/computeStepLengthMT, a4/a6)
// K3 and K4 are ONE kernel in a "pull" arrangement: every workgroup of launch n first sums the
// per-workgroup partial rows launch n-1 left behind (fixed order) and advances the controller —
// redundantly and deterministically in every workgroup —, then evaluates its own points and leaves its
// partial row for launch n+1.  No intra-launch synchronisation between workgroups; an align() is a
// chain of identical launches with no host round trip in between.
#pragma once
#include "common.hpp"

namespace lsr {

enum NdtPhase : int {
  PH_INIT = 0,      // first derivative pass at the guess
  PH_MT_FIRST = 1,  // first pass of a line search (with Hessian)
  PH_MT_TRIAL = 2,  // More-Thuente trial (gradient only)
  PH_MT_HESS = 3,   // Hessian recomputation after trials
  PH_DIAG = 4       // lsr_ndt_derivatives: store sums and stop
};

constexpr int NDT_NRED = 32;       // doubles per partial row: [0]=score [1..6]=grad [7]=#pairs [8..28]=Hessian upper triangle
constexpr int NDT_NRED_GRAD = 8;   // entries reduced on gradient-only passes
constexpr int NDT_LANE_THREADS = 512;    // default workgroup size of the lane kernel (1024 is the other instantiation)
constexpr int NDT_MAX_BLOCKS = 1024;
// dynamic LDS the lane kernel needs next to its table: one staging tile of 16 x 68 floats per wave (canon:: in ndt.hip)
constexpr int ndt_lane_tile_bytes(int threads) { return (threads / 64) * 16 * 68 * 4; }
constexpr int NDT_LDS_REC_BYTES = 48;  // LDS-resident leaf record: {mean.xyz, c00 | c01 c02 c11 c12 | c22, -, -, -}
constexpr int NDT_LDS_TABLE_MAX = 128 * 1024;  // largest voxel-table image staged into LDS (160 KiB per CU on gfx950)
constexpr int NDT_LDS_TABLE_MAX_QUAD = 120 * 1024;  // the quad kernel keeps a 31 KiB reduction buffer next to it

// Where the derivative pass finds the leaf records (chosen per launch by the host):
// Exact, order-independent accumulation of the per-workgroup partial sums (quad kernel): every fp64 partial is split
// into NDT_NBINS signed 31-bit chunks against fixed binary quanta q_k = 2^(62 - 31 (k + 1)) and added to int64 bins with
// integer atomics — integer addition is associative, so the totals are bit-reproducible whatever order the workgroups
// finish in, and exact (no rounding until the bins are folded back into one double).  Partials must stay below 2^62 in
// magnitude; anything else (overflow, NaN) raises the poison slot and the controller sees NaN.
constexpr int NDT_NBINS = 5;
constexpr int NDT_NSHARDS = 8;                                   // one accumulator row per XCD-ish shard (blockIdx & 7)
constexpr int NDT_BANK_WORDS = NDT_NSHARDS * NDT_NBINS * 32;     // int64 words per bank (32 value slots, 29 used, 31 = poison)
constexpr int NDT_NBANKS = 3;                                    // launch seq adds to bank seq%3, reads (seq-1)%3, clears (seq+1)%3

enum NdtTableMode : int {
  NDT_TAB_DENSE = 0,    // 64-byte records per grid cell in global memory (no cell->slot indirection)
  NDT_TAB_COMPACT = 1,  // cell_slot[] -> compact 64-byte records in global memory (huge grids)
  NDT_TAB_LDS = 2,      // the whole valid-voxel table (uint16 cell->slot map + 48-byte records) staged into LDS at the
                        // head of every launch, in the shadow of the controller: gathers become ds_read_b128
  NDT_TAB_TILE = 3      // tables that do not fit LDS (ndt_resolution <= 2 m on a 20-frame submap): per workgroup and pass the box
                        // of dense-table cells its (tile-ordered) points touch is staged into LDS; quad kernel only
};
constexpr int NDT_TILE_BYTES = 32 * 1024;  // tile buffer per workgroup: 682 cells of 48 bytes

struct NdtState {
  // ---- evaluation request, read by every workgroup of the next launch
  float T[12];       // row-major 3x4 point transform
  float jang[24];    // 8x3 angular Jacobian coefficient rows a..h       
  float hang[48];    // 15x3 angular Hessian coefficient rows a2..f3 (+pad)
  int want_hessian;
  int done;
  int phase;
  int d1_sign;
  // ---- constants of this align()
  double d1, d2;
  double step_max, step_min, eps;
  int max_iter, n_points;
  // ---- Newton state
  double p[6], x_t[6], dir[6];
  double score, g[6], H[36];
  int nr_iterations, converged;
  // ---- More-Thuente state
  double phi_0, d_phi_0, a_l, f_l, g_l, a_u, f_u, g_u, a_t;
  int open_interval, interval_converged, step_iterations;
  int token;          // identifies this align() in the host mailbox (NdtMailbox)
  // ---- results
  float final_T[16];  // column-major 4x4
  double trans_probability;
  double last_pairs;
  int n_evals, pad1;
};

// Host mailbox of a single registration (pinned, host-coherent memory mapped into the device): the chain reports its
// progress and its result straight into host memory, so the host feeds launches and detects the end of an align()
// by polling two words — no device-to-host copy and no stream synchronisation inside the chain.
struct NdtMailbox {
  unsigned long long progress;  // (token << 32) | seq of the launch workgroup 0 has entered most recently
  unsigned int done;            // = token once the controller has finished; written last (release, system scope)
  int converged, nr_iterations, n_evals;
  double trans_probability, last_pairs;
  float final_T[16];            // column-major 4x4
};

// One registration problem as the kernels see it (array of these for batched launches).
struct NdtProblem {
  const float* sx;
  const float* sy;
  const float* sz;
  int n;
  int nblocks;              // quad kernel: workgroups of this problem (fixed over the chain: they stride by it)
  const int* cell_slot;
  const float4* rec;
  int min_b[3];
  int max_b[3];
  int mul1, mul2;
  float leaf;
  int lds_map_bytes;        // NDT_TAB_LDS: bytes of the uint16 cell->slot map at the start of lds_image (multiple of 16)
  const uint4* lds_image;   // NDT_TAB_LDS: [map | records], padded to a multiple of 1 KiB (one wave-wide 16-byte DMA)
  int lds_bytes;
  int tile_bytes;           // NDT_TAB_TILE: capacity of the per-workgroup LDS tile buffer (dynamic LDS of the launch)
  long long* bins;          // [NDT_NBANKS][NDT_NSHARDS][NDT_NBINS][32] int64 accumulators (zeroed before launch 0)
  NdtState* st;             // [2] double buffered by launch parity
  NdtMailbox* mailbox;      // device view of the host mailbox (single registrations fed by polling) or nullptr
  const float4* centroid;   // KDTREE neighbourhood: float centroid of the leaf in slot cell_slot[cell] (nullptr for the DIRECT methods)
  float radius2;            // KDTREE: (float)(resolution * resolution), the kd-tree's squared search radius
};

struct NdtParamsHost {
  double resolution = 1.0, step_size = 0.1, outlier_ratio = 0.55, trans_eps = 0.1;
  int max_iterations = 35;
  int neighborhood = LSR_DIRECT7;
  int d1_sign = 1;
};

void ndt_gauss_constants(double resolution, double outlier_ratio, double* d1, double* d2);

int cloud_bbox(const DeviceCloud& cloud, float* mn, float* mx, unsigned int* n_finite, BuildScratch& sc, hipStream_t stream);
int cloud_bbox_begin(const DeviceCloud& cloud, BuildScratch& sc, hipStream_t stream);   // enqueue only
int cloud_bbox_end(const DeviceCloud& cloud, float* mn, float* mx, unsigned int* n_finite, BuildScratch& sc, hipStream_t stream);

// K1/K2: build grid from the SoA cloud.  Returns once the grid is complete (host polls the build mailbox twice).
int ndt_build_grid(const DeviceCloud& cloud, float leaf, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream);
// The same in two halves (after cloud_bbox_begin): _begin waits for the bounding box and enqueues the rest, _end waits for
// the result.  A batch of targets runs every _begin before the first _end, so the builds overlap on the device.
int ndt_build_grid_begin(const DeviceCloud& cloud, float leaf, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream);
int ndt_build_grid_end(VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream);
// KDTREE neighbourhood: grid.centroid of a complete grid (stable sort of the cloud by leaf + one thread per leaf adding its points in
// cloud order, in float, as VoxelGridCovariance does for Leaf::centroid).  Returns with the array complete.
int ndt_build_centroids(const DeviceCloud& cloud, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream);

// Geometry of a launch chain (fixed for the whole align()).
struct NdtLaunchCfg {
  int batch = 1;
  int max_blocks = 1;      // grid.x (largest nblocks of the batch)
  int neighborhood = LSR_DIRECT7;
  int tab = NDT_TAB_DENSE; // NdtTableMode
  int threads = NDT_LANE_THREADS;  // lane kernel: threads per workgroup (512 / 1024); quad kernel: POINTS per workgroup (64 / 128)
  int lds_bytes = 0;       // dynamic LDS (largest lds_bytes of the batch) when tab == NDT_TAB_LDS
  int sorted = 0;          // 1: the problems read the tile-ordered copy of the source (ndt_sort_source)
  int split = 0;           // lane kernel, 512 threads: 1 = two waves per chunk (each forms one half of the per-point tree): single scans
                           // that leave the chip half empty with one lane per point (cfg 5).  Same bits.
  int quad = 0;            // 1: four lanes per source point, 512-thread workgroups (single registrations: spreads a 30k-point scan
                           // over every CU; batches whose tables need NDT_TAB_TILE); 0: the lane kernel, one lane per point
                           // (candidate sets, large single scans).  Both return the same bits (canon:: in ndt.hip).
};
constexpr int NDT_QUAD_THREADS = 512;
constexpr int NDT_QUAD_POINTS = NDT_QUAD_THREADS / 4;  // source points per workgroup pass
// Launch `count` chained derivative+controller passes.
// h_single (nullable): host copy of the problem, passed by value when batch == 1.
int ndt_launch_evals(const NdtProblem* d_probs, const NdtProblem* h_single, const NdtLaunchCfg& cfg, int seq0, int count,
                     hipStream_t stream);
// Enqueue the LDS image of the valid-voxel table (grid.lds_image) after the leaf records exist; the kernel publishes
// n_valid / lds_bytes (0 when the table does not fit NDT_LDS_TABLE_MAX) and the `token` into the host mailbox.
int ndt_pack_lds_table(VoxelGridDev& grid, BuildScratch& sc, bool per_cell_leaf_n, unsigned int token, hipStream_t stream);
// ---- batched builds (candidate sets, graph_based_slam_component.cpp:181-231 generalised): the per-member parameters of up to
// LSR_GROUP members travel in the kernel arguments of ONE launch whose grid's y (or x) index selects the member.
constexpr int LSR_GROUP = 16;
struct PackMember {
  const int* cell_slot; const float4* rec; const int* leaf_n; int ncells, map_bytes, image_cap; unsigned int token;
  unsigned char* image; BuildMailbox* mb;
};
struct PackGroup { PackMember m[LSR_GROUP]; };
// lds_pack for `count` dense grids (leaf_n per cell) in ceil(count / LSR_GROUP) launches
int ndt_pack_lds_tables(VoxelGridDev* const* grids, BuildScratch* const* scs, const unsigned int* tokens, int count, hipStream_t stream);

// One target of a batched build: device-resident strided records in, SoA cloud + voxel grid out (each member keeps its own
// scratch and host mailbox).
struct TargetBuildJob {
  const void* d_aos; size_t stride; size_t n;
  DeviceCloud* cloud; float leaf; VoxelGridDev* grid; BuildScratch* sc;
  int path;   // set by ndt_targets_build_begin: 0 empty, 1 dense key space, 2 general
};
// de-interleave + bounding box of every member in ceil(count / LSR_GROUP) launches (the boxes arrive in the members' mailboxes)
int ndt_targets_ingest(TargetBuildJob* jobs, int count, hipStream_t stream);
// grid builds of every member, dense key spaces through the group kernels; ndt_build_grid_end() per member collects them
int ndt_targets_build_begin(TargetBuildJob* jobs, int count, hipStream_t stream);
int ndt_build_grids_dense_group(TargetBuildJob* const* jobs, int count, hipStream_t stream);
int ndt_grid_geometry(const DeviceCloud& cloud, float leaf, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream, int* path);

// K1/K2 for dense key spaces (grid_dense.hip): counting sort + per-cell sums + finalisation, everything enqueued, no host
// round trip.  grid.min_b / div_b / ncells must be set.
constexpr int VG_DENSE_MAX_CELLS = 16383;   // beyond: radix sort.  (A two-wave form with 32-bit packed counters for up to 36 000 cells was
                                            // built and measured on cfg 5's 22 113-cell grid: 0.33-0.35 ms against the radix sort's 0.30 — every
                                            // 4096-point workgroup clears, writes and prefixes a 22k-entry table; removed.)
int ndt_build_grid_dense(const DeviceCloud& cloud, float leaf, VoxelGridDev& grid, BuildScratch& sc, hipStream_t stream);
// Order the source cloud by voxel tile (NDT_TAB_TILE): counting sort of the points by the Morton code of the 2^shift x 2^shift
// column of grid cells their guess-moved image falls into (grid_dense.hip).  Consecutive points of `out` are neighbours in
// space, so the cells a workgroup of the derivative pass touches form a small box.  T12: row-major 3x4 guess (host memory).
int ndt_sort_source(const DeviceCloud& src, const float* T12, const VoxelGridDev& grid, DeviceCloud& out, BuildScratch& sc, hipStream_t stream);
// One small launch that writes the initial state (passed in the kernel arguments) into both state buffers and clears the
// quad kernel's accumulator banks (d_bins nullable).
int ndt_init_single(const NdtState& st, NdtState* d_state2, long long* d_bins, hipStream_t stream);
// Start of a candidate set's launch chain: ONE launch copies the members' problem records and initial states out of the
// lead's pinned host arrays (read by the device over PCIe: ~2.3 KB per member) and clears their accumulator banks.
// src_probs / src_states: device views of the pinned arrays (hipHostGetDevicePointer); n members; states are pairs.
int ndt_init_batch(const NdtProblem* src_probs, const NdtState* src_states, NdtProblem* d_probs, NdtState* d_states, long long* d_bins,
                   int n, hipStream_t stream);
// Host: controller state at the entry of computeTransformation (guess nullable = identity).
void ndt_fill_initial_state(NdtState& st, const float* guess16, const NdtParamsHost& prm, int n_points);
// Fill a diagnostic request on the host (lsr_ndt_derivatives).
void ndt_fill_diag_state(NdtState& st, const double* p6, const float* T16, int compute_hessian, const NdtParamsHost& prm,
                         int n_points);
// Host helper: default state constants for an align.
void ndt_fill_align_constants(NdtState& st, const NdtParamsHost& prm, int n_points);

// One device int to the host through the build mailbox (a 1-thread launch + one polled word instead of a device-to-host
// copy and a stream synchronisation).  Defined in grid_dense.hip.
int publish_device_int(const int* d_value, BuildScratch& sc, hipStream_t stream, int* out);
// N4: PointCloud2 payload (float32 fields at byte offsets ox/oy/oz/oi inside point_step records; oi < 0: no intensity)
// <-> SoA planes, device to device.
int pc2_write(const DeviceCloud& in, void* d_data, int step, int ox, int oy, int oz, int oi, hipStream_t stream);
// payload -> planes + the frontend's range filter (do_range) + the bounding-box pass in one launch; the box records wait in sc (host mailbox and
// device memory) for cloud_bbox_end / voxel_grid_filter
int pc2_ingest(const void* d_data, int step, int ox, int oy, int oz, int oi, size_t n, bool do_range, double rmin, double rmax,
               DeviceCloud& out, BuildScratch& sc, hipStream_t stream);
// N1: pcl::VoxelGrid::filter on the device (centroid per leaf, leaf-index order).
int voxel_grid_filter(const DeviceCloud& cloud, float leaf, DeviceCloud& out, BuildScratch& sc, hipStream_t stream);
int interleave(const DeviceCloud& in, void* d_out, size_t stride_bytes, hipStream_t stream);

// Transform cloud by a column-major 4x4 into a strided device buffer (align()'s `output`).
int transform_to_strided(const DeviceCloud& src, const float* T16_host, void* d_out, size_t stride_bytes, hipStream_t stream);
// AoS (strided xyz) -> SoA planes, device to device.
int deinterleave(const void* d_aos, size_t stride_bytes, size_t n, DeviceCloud& out, hipStream_t stream);
// the same for a set of clouds, LSR_GROUP per launch
struct DeinterleaveJob { const void* d_aos; size_t stride; size_t n; DeviceCloud* out; };
int deinterleave_group(const DeinterleaveJob* jobs, int count, hipStream_t stream);

}  // namespace lsr
