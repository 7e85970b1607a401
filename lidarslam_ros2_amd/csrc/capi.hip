// C ABI of the registration core (include/lidarslam_reg.h).  Host-side orchestration only:
// device buffers, launch chains, result read-back.  No CPU fallback exists — every compute
// entry point runs HIP kernels on a gfx950 device or returns an error status.
#include <algorithm>
#include <chrono>
#include <functional>
#include <limits>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <numeric>
#include <thread>

#include "handle.hpp"

namespace lsr {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
}  // namespace lsr

using namespace lsr;

namespace {

#define LSR_CHECK_HANDLE(h)                      \
  if (!(h)) {                                    \
    set_last_error("null handle");               \
    return LSR_ERR_INVALID_ARGUMENT;             \
  }                                              \
  DeviceGuard _guard((h)->device);               \
  if (!_guard.ok) {                              \
    set_last_error("hipSetDevice failed");       \
    return LSR_ERR_HIP;                          \
  }                                              \
  if ((h)->dep && settle_dep(h)) return LSR_ERR_HIP;

// the object is about to be used on its own: its stream now waits for the group work it was part of (lsr::StreamDep, handle.hpp)
int settle_dep(lsr_handle h) {
  if (!h->dep) return LSR_OK;
  if (h->dep_stream != h->stream && hipStreamWaitEvent(h->stream, h->dep->ev, 0) != hipSuccess) {
    set_last_error("hipStreamWaitEvent failed (deferred dependency on a group launch)");
    return LSR_ERR_HIP;
  }
  h->dep.reset();
  h->dep_stream = nullptr;
  return LSR_OK;
}
// after group work for `members` has been enqueued on the lead's stream: one event record, the members remember it
int defer_members_behind(lsr_handle lead, lsr_handle* members, int count) {
  if (!lead->lead_dep) {
    auto d = std::make_shared<lsr::StreamDep>();
    if (hipEventCreateWithFlags(&d->ev, hipEventDisableTiming) != hipSuccess) { set_last_error("hipEventCreate failed"); return LSR_ERR_HIP; }
    lead->lead_dep = d;
  }
  bool any = false;
  for (int b = 0; b < count; b++) any = any || (members[b] != lead && members[b]->stream != lead->stream);
  if (!any) return LSR_OK;
  if (hipEventRecord(lead->lead_dep->ev, lead->stream) != hipSuccess) { set_last_error("hipEventRecord failed"); return LSR_ERR_HIP; }
  for (int b = 0; b < count; b++)
    if (members[b] != lead && members[b]->stream != lead->stream) { members[b]->dep = lead->lead_dep; members[b]->dep_stream = lead->stream; }
  return LSR_OK;
}

// Make everything `h` still has in flight on its own stream precede what is enqueued on `lead` next (group launches of a
// candidate set run on the first member's stream).  An idle stream — the usual case — costs one query and no event.
int order_lead_after(hipStream_t lead, lsr_handle h) {
  if (h->dep) {
    // everything this member has had enqueued since its last own use went to dep_stream (a group call ordered that stream behind the
    // member's own one before it started): on the same lead stream there is nothing to order, on another one the event is the order —
    // also when the member is the lead of THIS call (its own stream is `lead`: the dependency is then settled for good)
    if (h->dep_stream != lead) LSR_HIP(hipStreamWaitEvent(lead, h->dep->ev, 0));
    if (h->stream == lead) { h->dep.reset(); h->dep_stream = nullptr; }
    return LSR_OK;
  }
  if (h->stream == lead) return LSR_OK;
  if (hipStreamQuery(h->stream) == hipSuccess) return LSR_OK;
  LSR_HIP(hipEventRecord(h->ev1, h->stream));
  LSR_HIP(hipStreamWaitEvent(lead, h->ev1, 0));
  return LSR_OK;
}

int upload_cloud(lsr_handle h, const void* pts, size_t stride, size_t n, bool on_device, DeviceCloud& out) {
  if (stride < 12 || (stride % 4) != 0) {
    set_last_error("stride_bytes must be a multiple of 4 and >= 12");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  if (n > 0 && !pts) {
    set_last_error("null point pointer");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  if (n > (size_t)INT32_MAX / 2) {
    set_last_error("cloud too large");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  const void* d_aos = pts;
  if (!on_device && n > 0) {
    int st = h->staging.reserve(n * stride);
    if (st) return st;
    LSR_HIP(hipMemcpyAsync(h->staging.p, pts, n * stride, hipMemcpyHostToDevice, h->stream));
    d_aos = h->staging.p;
  }
  return deinterleave(d_aos, stride, n, out, h->stream);
}

// A PointCloud2 payload (or strided xyz records: a payload without intensity) -> SoA planes through pc2_ingest: ONE launch that also applies
// the frontend's range filter (do_range) and leaves the bounding-box records for what follows
int ingest_pc2(lsr_handle h, const void* data, size_t n, const lsr_pc2_layout* L, bool on_device, bool do_range, double rmin, double rmax,
               DeviceCloud& out) {
  if (n > 0 && !data) { set_last_error("null PointCloud2 data"); return LSR_ERR_INVALID_ARGUMENT; }
  if (n > (size_t)INT32_MAX / 2) { set_last_error("cloud too large"); return LSR_ERR_INVALID_ARGUMENT; }
  const void* d = data;
  if (!on_device && n > 0) {
    int st = h->staging.reserve(n * L->point_step);
    if (st) return st;
    LSR_HIP(hipMemcpyAsync(h->staging.p, data, n * L->point_step, hipMemcpyHostToDevice, h->stream));
    d = h->staging.p;
  }
  return pc2_ingest(d, (int)L->point_step, (int)L->offset_x, (int)L->offset_y, (int)L->offset_z, L->offset_intensity, n, do_range, rmin, rmax,
                    out, h->scratch, h->stream);
}

// Workgroups per registration.  A single registration spreads one point per thread over as many CUs as
// it can (latency); a batch wants ~4 resident workgroups per CU in total and lets every thread stride
// over several points, which amortises the reduction and the partial-row traffic (throughput).
// Compute units of the device (256 on MI355X).
int device_cus(int device) {
  static int cached[64] = {};
  if (device >= 0 && device < 64 && cached[device] > 0) return cached[device];
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) v = 256;
  if (device >= 0 && device < 64) cached[device] = v;
  return v;
}

// Workgroups per registration.  Every workgroup of a pass pays the fixed head of the launch (totals of the previous pass,
// controller step, next request: 2.5-4 us) before it evaluates a point, so a pass never launches more workgroups than the
// chip holds at once — two 512-thread workgroups of the quad kernel (128 points each) or of the lane kernel per CU, one of its
// 1024-thread workgroups — and lets each of them walk several batches of points instead.
// `threads`: THREADS per workgroup; `points`: source points a workgroup takes per trip (quad kernel: threads / 4).
constexpr int NDT_QUAD_BATCH_MAX = 1;
constexpr int NDT_LANE_SINGLE_MIN = 65536;   // source points from which a single registration uses the lane kernel
int ndt_resident_wgs(int device, int threads) {
  static const int wgs_per_cu = [] { const char* e = std::getenv("LSR_NDT_WGS_PER_CU"); const int v = e ? std::atoi(e) : 0; return (v >= 1 && v <= 8) ? v : 0; }();
  // two 512-thread workgroups per CU (128 VGPRs per lane: four waves per SIMD; the lane kernel's LDS table + staging tiles fit twice),
  // one of 1024 threads.  (Until round 5 this returned 2048 / threads = 4 for 512: more workgroups than are ever co-resident, each of
  // them paying the head.  Measured on the 64-candidate chain: 1.46 ms with 4, 1.46 ms with 2, 2.04 ms with 1.)
  // Below 512 threads (the quad kernel with 64 points per workgroup: 256 threads, LSR_NDT_WORKGROUP=64) occupancy decides again:
  // 2048 / threads workgroups fill the same four waves per SIMD (the measurement above covers 512 and 1024 only).
  const int by_size = threads >= 1024 ? 1 : threads >= 512 ? 2 : 2048 / std::max(64, threads);
  return device_cus(device) * (wgs_per_cu ? wgs_per_cu : by_size);
}
int ndt_nblocks(size_t n, int device, int batch, int threads, int points) {
  int nb = (int)((n + points - 1) / points);
  const int share = std::max(1, ndt_resident_wgs(device, threads) / std::max(1, batch));
  nb = std::max(1, std::min(nb, std::min(share, NDT_MAX_BLOCKS)));
  return nb;
}
// workgroup geometry of a launch configuration
int cfg_wg_threads(const NdtLaunchCfg& cfg) { return cfg.quad ? 4 * cfg.threads : cfg.threads; }
int cfg_wg_points(const NdtLaunchCfg& cfg) { return (!cfg.quad && cfg.split) ? cfg.threads / 2 : cfg.threads; }

// LDS a workgroup may use on this device (hipDeviceAttributeMaxSharedMemoryPerBlock; 160 KiB on gfx950): the table modes that
// stage into LDS are only chosen when their buffers fit, anything else reads the global table.
int device_lds_bytes(int device) {
  static int cached[64] = {};
  if (device >= 0 && device < 64 && cached[device] > 0) return cached[device];
  // one workgroup may use a CU's whole LDS on AMD hardware: the larger of the two attributes (runtimes differ in which one
  // reports the 160 KiB of gfx950); 64 KiB when neither answers
  int per_block = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) per_block = 0;
  if (hipDeviceGetAttribute(&per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess) per_cu = 0;
  int v = std::max(per_block, per_cu);
  if (v <= 0) v = 64 * 1024;
  if (device >= 0 && device < 64) cached[device] = v;
  return v;
}
constexpr int NDT_QUAD_STATIC_LDS = 32 * 1024;   // static LDS of the quad kernel next to its table / tile buffer (27 KiB + margin)
constexpr int NDT_LANE_STATIC_LDS = 6 * 1024;    // ... of the lane kernel (bins, state image; its staging tiles are dynamic LDS)

void fill_problem(NdtProblem& P, lsr_handle h, NdtState* d_state, long long* d_bins, const NdtLaunchCfg& cfg) {
  const VoxelGridDev& g = h->target->grid;
  const DeviceCloud& src = cfg.sorted ? h->source_sorted : h->source;   // the tile-ordered copy (always in tile mode)
  P.sx = src.x(); P.sy = src.y(); P.sz = src.z();
  P.n = (int)h->source.n;
  P.nblocks = ndt_nblocks(h->source.n, h->device, cfg.batch, cfg_wg_threads(cfg), cfg_wg_points(cfg));
  P.lds_image = g.lds_image.p;
  P.lds_map_bytes = g.lds_map_bytes;
  P.lds_bytes = g.lds_bytes;
  P.cell_slot = g.cell_slot.p;
  P.rec = g.rec.p;
  for (int k = 0; k < 3; k++) { P.min_b[k] = g.min_b[k]; P.max_b[k] = g.max_b[k]; }
  P.mul1 = g.div_b[0];
  P.mul2 = g.div_b[0] * g.div_b[1];
  P.leaf = g.leaf;
  P.tile_bytes = (cfg.tab == NDT_TAB_TILE) ? cfg.lds_bytes : 0;
  P.st = d_state;
  P.bins = d_bins;
  P.mailbox = nullptr;
  // KDTREE: the 27-cell kernels with the kd-tree's radius test (ensure_ndt_grid built the centroids); resolution_ is a float member
  const bool kd = h->ndt.neighborhood == LSR_KDTREE && h->target->has_centroids;
  P.centroid = kd ? g.centroid.p : nullptr;
  P.radius2 = kd ? (float)((double)(float)h->ndt.resolution * (double)(float)h->ndt.resolution) : 0.f;
}

// Where the derivative pass reads the leaf records, and with which kernel.  grids: the batch's targets.
//  every table fits LDS            -> NDT_TAB_LDS   (quad kernel for one registration, one-lane kernel for a batch)
//  dense tables that do not fit    -> NDT_TAB_DENSE (global gathers; L2 resident at the reference's resolutions), or on request
//                                     NDT_TAB_TILE  (quad kernel, also for batches; source ordered by voxel tile per align)
//  tables beyond 4 Mi cells        -> NDT_TAB_COMPACT (global gathers through cell_slot)
// table_mode: the lead object's LSR_NDT_TABLE_MODE override (-1 automatic).
void choose_table_mode(lsr_handle lead, lsr_handle* hs, int B, NdtLaunchCfg& cfg) {
  bool all_lds = true, all_dense = true;
  int lds_max = 0;
  for (int b = 0; b < B; b++) {
    const VoxelGridDev& g = hs[b]->target->grid;
    all_lds = all_lds && g.lds_bytes > 0;
    all_dense = all_dense && g.dense;
    lds_max = std::max(lds_max, g.lds_bytes);
  }
  const int lds_cap = device_lds_bytes(lead->device);
  const int override_mode = lead->ndt_table_mode;
  // four lanes per point for ONE registration (latency: a 30k-point scan on every CU); the lane kernel for candidate sets and
  // on request (LSR_NDT_QUAD = 0) — both return the same bits, so the choice is a matter of speed only.  Measured on cfg-4 sets of
  // 8 / 16 / 64 candidates (align stage, round 3's kernels): one lane per point 0.61 / 0.90 / 2.73 ms, four lanes 0.76 / 1.09 / 4.06 ms
  // (env LSR_NDT_QUAD_BATCH_MAX raises the batch size up to which the four-lane kernel is used; read once)
  static const int quad_batch_max = [] { const char* e = std::getenv("LSR_NDT_QUAD_BATCH_MAX"); const int v = e ? std::atoi(e) : NDT_QUAD_BATCH_MAX; return v < 1 ? 1 : v; }();
  // ... and a single scan large enough to fill the chip with one lane per point takes the lane kernel too: cfg 5 (120k points,
  // dense global table) 9.4 us per pass with 512-thread workgroups against 16.1 us through the quad kernel (round 4)
  size_t n_max = 0;
  for (int b = 0; b < B; b++) n_max = std::max(n_max, hs[b]->source.n);
  const bool want_quad_single = (B <= quad_batch_max && (lead->ndt_quad == 1 || (lead->ndt_quad < 0 && n_max < (size_t)NDT_LANE_SINGLE_MIN)));
  // 512 threads per workgroup (1024 on request).  A workgroup's waves share ONE compute unit: 1024 threads are four waves per SIMD
  // whatever the size of the set, 512 are two and twice the workgroups — cfg-4 sets of 4 / 8 / 16 / 64 members, align stage:
  // 0.48 / 0.50 / 0.71 / 1.98 ms with 1024 threads, 0.40 / 0.48 / 0.65 / 1.74 ms with 512 (two chains from 6 members on).
  // (DIRECT26's 27 neighbours per point do not fit the 128 registers a 1024-thread workgroup leaves a lane anyway.)
  const int lane_threads = (lead->ndt_threads == 512 || lead->ndt_threads == 1024) ? lead->ndt_threads : NDT_LANE_THREADS;
  const int static_lds = want_quad_single ? NDT_QUAD_STATIC_LDS : NDT_LANE_STATIC_LDS + ndt_lane_tile_bytes(lane_threads);
  const int table_cap = std::min(want_quad_single ? NDT_LDS_TABLE_MAX_QUAD : NDT_LDS_TABLE_MAX, lds_cap - static_lds);
  const bool lds_ok = all_lds && lds_max <= table_cap;
  const bool tile_ok = all_dense && lead->ndt_quad != 0 && lead->ndt_sort != 0 && NDT_TILE_BYTES + NDT_QUAD_STATIC_LDS <= lds_cap &&
                       lead->ndt.neighborhood != LSR_KDTREE;   // (the tile mode numbers cells per tile: no centroid lookup there)
  int tab;
  if (override_mode == NDT_TAB_COMPACT) tab = NDT_TAB_COMPACT;
  else if (override_mode == NDT_TAB_DENSE && all_dense) tab = NDT_TAB_DENSE;
  else if (override_mode == NDT_TAB_TILE && tile_ok) tab = NDT_TAB_TILE;
  else if (override_mode == NDT_TAB_LDS && lds_ok) tab = NDT_TAB_LDS;
  else if (lds_ok) tab = NDT_TAB_LDS;
  else tab = all_dense ? NDT_TAB_DENSE : NDT_TAB_COMPACT;   // measured: the pass is not gather bound, the tile mode's
                                                            // extra barriers cost more than its LDS gathers save — on request only
  cfg.tab = tab;
  if (std::getenv("LSR_DEBUG_TABLE")) {
    static int shown = 0;
    if (shown++ < 4) fprintf(stderr, "[lidarslam_reg] table mode %d for %d member(s): LDS image up to %d bytes (cap %d, static %d)\n", tab, B, lds_max, table_cap, static_lds);
  }
  cfg.quad = (want_quad_single || tab == NDT_TAB_TILE) ? 1 : 0;
  cfg.lds_bytes = (tab == NDT_TAB_LDS) ? lds_max : (tab == NDT_TAB_TILE ? NDT_TILE_BYTES : 0);
  if (cfg.quad) cfg.threads = (lead->ndt_threads == 64 || lead->ndt_threads == 128) ? lead->ndt_threads : NDT_QUAD_POINTS;  // POINTS per workgroup
  else cfg.threads = lane_threads;
  // two waves per chunk (ndt.hip: SPLIT) for a single scan whose points, one lane each, leave the chip half empty but whose chunk
  // pairs still fit it at once: 65 536 .. resident workgroups x 256 points (cfg 5: 120 000 points = 469 workgroups of 512 threads).
  // LSR_NDT_SPLIT = 0 / 1 (key or environment preset) forces it off / on (on: any single registration that takes the 512-thread lane kernel).
  cfg.split = 0;
  if (!cfg.quad && B == 1 && cfg.threads == 512) {
    const size_t fits = (size_t)ndt_resident_wgs(lead->device, 512) * 256;
    // measured (cfg 5, 120 000 points, res 2.0 / 1.0): 9.25 / 9.33 us per pass split against 9.14 / 9.08 us with one wave per chunk —
    // the pass is bound by its fixed chain (boundary, head read, controller: ~5.5 us), not by the point loop the split halves; the
    // barrier and the second copy of the gathers eat what the shorter per-wave chain gives.  Automatic = off; kept as an A/B form
    // (same bits: tests/test_ndt_gpu.py) with its counters in profiles/r06_pmc_cfg5.md.
    (void)fits;
    cfg.split = lead->ndt_split >= 0 ? lead->ndt_split : 0;
  }
  // source ordered by voxel tile: always for the tile mode (its boxes are small only then); for global-table gathers on request
  // (LSR_NDT_SORT = 1): neighbouring lanes then read neighbouring records
  cfg.sorted = (tab == NDT_TAB_TILE) || (lead->ndt_sort == 1 && tab != NDT_TAB_LDS);
}

// A fresh target object — or the handle's current one recycled when nobody else holds it (lsr_share_target): its device
// buffers only ever grow, so the frontend's "new target every few scans" costs no hipMalloc / hipFree (those were 250 us
// of a 350 us setInputTarget).
std::shared_ptr<TargetData> fresh_target(lsr_handle h) {
  std::shared_ptr<TargetData> t;
  auto only_mine = [&](const std::shared_ptr<TargetData>& c) {
    return c && c.use_count() == (long)((h->target == c) + (h->spare_target == c));
  };
  if (only_mine(h->target)) t = h->target;
  else if (only_mine(h->spare_target)) t = h->spare_target;
  else t = std::make_shared<TargetData>();
  h->spare_target = t;  // survives a failed setInputTarget (which resets h->target)
  t->n = 0;
  t->has_grid = t->has_hash = t->has_cov = t->has_centroids = false;
  t->grid_leaf = 0.f;
  return t;
}

bool target_is_shared(lsr_handle h) {
  return h->target && h->target.use_count() > (long)(1 + (h->spare_target == h->target));
}

int ensure_ndt_grid(lsr_handle h) {
  TargetData& t = *h->target;
  std::lock_guard<std::mutex> lock(t.build_mutex);
  float leaf = (float)h->ndt.resolution;
  // the KDTREE neighbourhood reads the leaves' float centroids: built on first use, for the grid in place
  auto centroids = [&]() -> int {
    if (h->ndt.neighborhood != LSR_KDTREE || t.has_centroids || t.grid.ncells == 0) return LSR_OK;
    const int cst = ndt_build_centroids(t.cloud, t.grid, h->scratch, h->stream);
    if (!cst) t.has_centroids = true;
    return cst;
  };
  if (t.has_grid && t.grid_leaf == leaf) return centroids();
  if (t.has_grid && target_is_shared(h)) {
    // rebuilding in place would pull the grid from under the other handles (their d1/d2 and leaf size belong to the old one)
    set_last_error("the shared target's voxel grid was built at another resolution: sharers must use one ndt_resolution");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  t.has_centroids = false;
  int st = ndt_build_grid(t.cloud, leaf, t.grid, h->scratch, h->stream);
  if (st) return st;
  t.has_grid = true;
  t.grid_leaf = leaf;
  return centroids();
}

// An NDT target whose voxel grid was built by the counting-sort builder keeps its points in voxel order: the neighbour grid is
// a refinement of that order (one launch, nn_build_hash_from_grids) instead of a second sort of the cloud.
// A/B switch (env LSR_NN_PREFETCH=0: the neighbour grids of a candidate set are built by getFitnessScore, not under the align chain)
bool nn_prefetch_enabled() {
  static const bool on = [] { const char* e = getenv("LSR_NN_PREFETCH"); return !(e && e[0] == '0'); }();
  return on;
}

// A/B switch (env LSR_NDT_WIDEN=0: the launches of a candidate set keep their first geometry)
bool lane_widen_enabled() {
  static const bool on = [] { const char* e = getenv("LSR_NDT_WIDEN"); return !(e && e[0] == '0'); }();
  return on;
}

// A/B switch (env LSR_NN_FROM_GRID=0: always build the neighbour grid from the cloud); read once
bool hash_from_grid_enabled() {
  static const bool enabled = [] { const char* e = getenv("LSR_NN_FROM_GRID"); return !(e && e[0] == '0'); }();
  return enabled;
}
bool hash_from_grid_possible(lsr_handle h) {
  if (!hash_from_grid_enabled()) return false;
  const TargetData& t = *h->target;
  return h->method == LSR_METHOD_NDT && t.has_grid && t.grid.has_sorted && t.grid.ncells > 0 && t.grid.sorted_n == t.cloud.n;
}

int ensure_target_hash(lsr_handle h) {
  TargetData& t = *h->target;
  std::lock_guard<std::mutex> lock(t.build_mutex);
  if (t.has_hash) return LSR_OK;
  int st;
  if (hash_from_grid_possible(h)) {
    const VoxelGridDev* vg[1] = {&t.grid};
    HashGridDev* hg[1] = {&t.hash};
    st = nn_build_hash_from_grids(vg, hg, 1, h->stream);
  } else {
    st = nn_build_hash(t.cloud, nn_pick_cell(t.cloud.n, h), t.hash, h->scratch, h->stream);
  }
  if (st) return st;
  t.has_hash = true;
  return LSR_OK;
}

// The host feeds the launch chain and watches the mailboxes the chain writes into host memory (NdtMailbox, one per
// registration of the batch): workgroup 0 of every launch reports its sequence number, the launch in which a
// registration's controller finishes publishes its result and raises its `done`.  Launch `seq` consumes the rows of
// launch seq-1 (ndt.hip), so an align of E derivative passes takes E+1 launches.  The host keeps a couple of launches
// queued ahead of the device and never synchronises the stream or copies state back inside the chain; after the last
// `done` at most LOW_WATER + REFILL queued launches remain, which exit at their head.
// on_poll (nullable): called between polls while the chain runs (the eager fitness dispatch of a candidate set hangs off it).
// nb_full > 0 (lane kernel, candidate sets): the launches are widened as members finish — grid.x = resident workgroups /
// members still running, at most nb_full (= one trip per lane for the largest member); the canonical sum does not depend on it.
// --- launch-chain streams -------------------------------------------------------------------------------------------------
// Two HIP streams run concurrently only when the runtime has mapped them to different hardware queues; mapped to the same
// queue their launches serialise AND every switch between them costs a signal round trip (measured: 60 us per switch, a
// three-chain set took twice as long as one chain when two chains shared a queue).  The mapping cannot be queried, so it is
// MEASURED once per stream: a kernel on stream a spins (bounded) on a flag that a kernel enqueued AFTERWARDS on stream b sets.
__global__ void chain_probe_wait(unsigned int* flag, unsigned int* seen, unsigned long long budget) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  unsigned int v = 0;
  while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 &&
         __builtin_amdgcn_s_memrealtime() - t0 < budget) __builtin_amdgcn_s_sleep(8);
  *seen = v ? 1u : 2u;
}
__global__ void chain_probe_set(unsigned int* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 1: b's kernel ran while a's was running; 0: it did not within 200 us (same hardware queue); < 0: error (negated status)
int streams_run_concurrently(hipStream_t a, hipStream_t b, unsigned int* d_words /* 2 */) {
  // the flag is cleared and BOTH streams are idle before the two kernels go out (b's kernel may overtake anything a still holds)
  if (hipMemsetAsync(d_words, 0, 8, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -LSR_ERR_HIP;
  hipLaunchKernelGGL(chain_probe_wait, dim3(1), dim3(1), 0, a, d_words, d_words + 1, 20000ull);
  hipLaunchKernelGGL(chain_probe_set, dim3(1), dim3(1), 0, b, d_words);
  unsigned int seen = 0;
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess ||
      hipMemcpy(&seen, d_words + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return -LSR_ERR_HIP;
  return seen == 1u ? 1 : 0;
}

// The auxiliary streams of a batch lead, looked for once per handle: the side stream (grid refinement and fitness searches of a
// candidate set, under the launch chain) and up to two more launch-chain streams, each VERIFIED to run concurrently with the
// lead's stream and with the ones found before it.  A stream that fails the check is kept alive until the search ends, so
// that the runtime maps the next one to another hardware queue.  When no concurrent side stream turns up an unverified one
// is used (correct, just not overlapped); chain streams are only ever verified ones.
int ensure_aux_streams(lsr_handle lead) {
  if (lead->chain_probed) return LSR_OK;
  lsr::DevBuf<unsigned int> words;
  int st;
  if ((st = words.reserve(2))) return st;
  std::vector<hipStream_t> rejected, found;
  // (Round 5 measured a lowest-priority side stream — hipStreamCreateWithPriority — for the grid refinement and the fitness searches:
  // the chain launches were disturbed just the same and the side work itself was starved: align + fitness of an 8-candidate share
  // 1.26-1.38 ms against 0.76-0.95 ms.  Not used.)
  for (int tries = 0; tries < 10 && (int)found.size() < 3; tries++) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
    bool ok = streams_run_concurrently(lead->stream, s, words.p) == 1;
    for (size_t c = 0; ok && c < found.size(); c++) ok = streams_run_concurrently(found[c], s, words.p) == 1;
    if (ok) found.push_back(s); else rejected.push_back(s);
  }
  for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
  (void)hipStreamSynchronize(lead->stream);
  // from here on a failure must not leak what was found (and must not leave half of it on the handle: the next call probes again)
  auto give_up = [&](const char* what) {
    for (hipStream_t s : found) if (s) (void)hipStreamDestroy(s);
    if (found.empty() && lead->side_stream) (void)hipStreamDestroy(lead->side_stream);
    lead->side_stream = nullptr;
    for (int c = 0; c < 3; c++) {
      lead->chain_stream[c] = nullptr;
      if (lead->chain_ev[c]) { (void)hipEventDestroy(lead->chain_ev[c]); lead->chain_ev[c] = nullptr; }
    }
    lead->n_chain_streams = 0;
    set_last_error(std::string("auxiliary streams of a batch lead: ") + what);
    return LSR_ERR_HIP;
  };
  size_t k = 0;
  // LSR_SIDE_CUS = K (experiment, VERDICT r05 #2 ii): the side stream — grid refinement, fitness searches of early finishers — is
  // confined to K compute units (hipExtStreamCreateWithCUMask: the first K bits of the mask; the driver deals the bits of a mask to the
  // XCDs in turn, so K / 8 CUs of every XCD), the launch chain keeps the rest of the chip to itself
  static const int side_cus = [] { const char* e = std::getenv("LSR_SIDE_CUS"); return e ? std::atoi(e) : 0; }();
  if (side_cus > 0) {
    const int total = device_cus(lead->device);
    const int words = (total + 31) / 32;
    std::vector<uint32_t> mask((size_t)words, 0u);
    for (int c = 0; c < std::min(side_cus, total); c++) mask[c / 32] |= 1u << (c % 32);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data()) != hipSuccess) return give_up("CU-masked side stream");
    lead->side_stream = s;
    if (std::getenv("LSR_DEBUG_STREAMS")) {
      fprintf(stderr, "[lidarslam_reg] side stream confined to %d of %d CUs\n", std::min(side_cus, total), total);
    }
  } else
  if (!found.empty()) lead->side_stream = found[k++];
  else if (hipStreamCreateWithFlags(&lead->side_stream, hipStreamNonBlocking) != hipSuccess) return give_up("no side stream");
  lead->n_chain_streams = 0;
  for (; k < found.size(); k++) {
    hipEvent_t ev = nullptr;
    if (lead->n_chain_streams < 3 && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
      lead->chain_stream[lead->n_chain_streams] = found[k]; lead->chain_ev[lead->n_chain_streams] = ev; lead->n_chain_streams++;
    } else { (void)hipStreamDestroy(found[k]); found[k] = nullptr; }
  }
  if (!lead->side_ev && hipEventCreateWithFlags(&lead->side_ev, hipEventDisableTiming) != hipSuccess) return give_up("event");
  if (!lead->side_fork_ev && hipEventCreateWithFlags(&lead->side_fork_ev, hipEventDisableTiming) != hipSuccess) return give_up("event");
  if (!lead->chain_fork_ev && hipEventCreateWithFlags(&lead->chain_fork_ev, hipEventDisableTiming) != hipSuccess) return give_up("event");
  lead->chain_probed = true;
  if (std::getenv("LSR_DEBUG_STREAMS"))
    fprintf(stderr, "[lidarslam_reg] aux streams of handle %p: %zu verified concurrent, %zu rejected, %d chain stream(s)\n", (void*)lead,
            found.size(), rejected.size(), lead->n_chain_streams);
  return LSR_OK;
}

// how many independent launch chains a set of B registrations runs as (env LSR_NDT_CHAINS = 1..3 forces it for B >= 2*chains; at most two verified chain streams exist next to the lead's own)
int ndt_chain_count(int B) {
  static const int forced = [] { const char* e = std::getenv("LSR_NDT_CHAINS"); return e ? std::atoi(e) : 0; }();
  // measured (cfg-4 sets, align stage, 1 -> 2 chains, 512-thread workgroups): 4 members +13 %, 6 members 0 %, 8 / 12 / 16 members
  // -3 / -9 / -13 %; with 1024-thread workgroups 24 / 32 / 48 / 64 members -17 / -10 / -10 / -9 %
  int n = forced > 0 ? std::min(forced, 3) : (B >= 6 ? 2 : 1);
  while (n > 1 && B < 2 * n) n--;
  return n;
}

// n_chains > 1 (small candidate sets): the members are dealt to n_chains independent launch chains, chain c covering members
// [chain_first[c], chain_first[c+1]) on streams[c].  A launch of a small set is mostly fixed latency (the launch boundary and
// the head: bank fold + controller), which the other chain's launch hides; the answer of a member does not depend on which
// launches carry it (canonical sums), so the split changes no bit.
int run_ndt_feeder(lsr_handle h, const NdtProblem* d_probs, const NdtProblem* h_probs, const NdtLaunchCfg& cfg_in, int first, int hard_cap,
                   unsigned int token, int* launches_out, const std::function<int()>* on_poll = nullptr, int nb_full = 0,
                   int n_chains = 1, const int* chain_first = nullptr, const hipStream_t* streams = nullptr) {
  constexpr int MAX_CHAINS = 4;
  struct Chain { int b0, b1, n_done, launched; bool over; hipStream_t stream; NdtLaunchCfg cfg; };
  Chain ch[MAX_CHAINS];
  const int one_first[2] = {0, cfg_in.batch};
  if (n_chains <= 1 || !chain_first || !streams) { n_chains = 1; chain_first = one_first; streams = &h->stream; }
  if (n_chains > MAX_CHAINS) { set_last_error("too many launch chains"); return LSR_ERR_INVALID_ARGUMENT; }
  const int resident = ndt_resident_wgs(h->device, cfg_wg_threads(cfg_in));
  // spin: two launches queued ahead are enough; yield / sleep give the core away between polls, so more launches are
  // kept queued to ride out the scheduler's latency (surplus launches exit at their head, ~2 us each)
  const int wait_mode = h->scratch.wait_mode;
  const int LOW_WATER = (wait_mode == WAIT_SPIN) ? 2 : (wait_mode == WAIT_YIELD ? 4 : 16), REFILL = (wait_mode == WAIT_SLEEP) ? 8 : 2;
  const int batch = cfg_in.batch;
  const NdtMailbox* mb = h->mailbox.p;
  const NdtProblem* h_single = (batch == 1) ? h_probs : nullptr;  // a single registration travels in the kernel arguments
  int st;
  for (int c = 0; c < n_chains; c++) {
    Chain& C = ch[c];
    C.b0 = chain_first[c]; C.b1 = chain_first[c + 1]; C.n_done = C.b0; C.over = false; C.stream = streams[c];
    C.cfg = cfg_in; C.cfg.batch = C.b1 - C.b0;
    C.launched = std::max(1, std::min(first, hard_cap));
    if ((st = ndt_launch_evals(d_probs + C.b0, h_single, C.cfg, 0, C.launched, C.stream))) return st;
  }
  unsigned long long last_progress = 0;
  auto t_progress = std::chrono::steady_clock::now();
  int chains_left = n_chains;
  for (unsigned long long spins = 1; chains_left > 0; spins++) {
    unsigned long long pr_sum = 0;
    for (int c = 0; c < n_chains; c++) {
      Chain& C = ch[c];
      if (C.over) continue;
      while (C.n_done < C.b1 && __atomic_load_n(&mb[C.n_done].done, __ATOMIC_ACQUIRE) == token) C.n_done++;  // [b0, n_done) have raised their flag
      if (C.n_done == C.b1) { C.over = true; chains_left--; continue; }
      // every registration of a launch advances together: the first unfinished one tells how far the device is
      const unsigned long long pr = __atomic_load_n(&mb[C.n_done].progress, __ATOMIC_RELAXED);
      pr_sum += pr;
      const int entered = ((unsigned int)(pr >> 32) == token) ? (int)(unsigned int)pr : -1;  // -1: nothing of this align yet
      if (C.launched < hard_cap && C.launched - 1 - entered < LOW_WATER) {
        const int k = std::min(REFILL, hard_cap - C.launched);
        if (nb_full > 0 && batch > 1) {
          int running = 0;   // over all chains: they share the chip
          for (int b = 0; b < batch; b++) running += (__atomic_load_n(&mb[b].done, __ATOMIC_RELAXED) != token);
          C.cfg.max_blocks = std::max(cfg_in.max_blocks, std::min(nb_full, resident / std::max(1, running)));
        }
        if ((st = ndt_launch_evals(d_probs + C.b0, h_single, C.cfg, C.launched, k, C.stream))) return st;
        C.launched += k;
        continue;
      }
      if (C.launched >= hard_cap && entered >= hard_cap - 1) {  // the last permitted launch has started: let it finish
        LSR_HIP(hipStreamSynchronize(C.stream));
        while (C.n_done < C.b1 && __atomic_load_n(&mb[C.n_done].done, __ATOMIC_ACQUIRE) == token) C.n_done++;
        if (C.n_done == C.b1) { C.over = true; chains_left--; continue; }
        set_last_error("NDT controller did not finish within the launch cap");
        return LSR_ERR_HIP;
      }
    }
    if (chains_left == 0) break;
    if ((spins & 0x3FFF) == 0 || wait_mode == WAIT_SLEEP) {  // a device that stops making progress must not hang the caller forever
      const auto now = std::chrono::steady_clock::now();
      if (pr_sum != last_progress) { last_progress = pr_sum; t_progress = now; }
      if (std::chrono::duration<double>(now - t_progress).count() > 30.0) {
        const hipError_t e = hipStreamQuery(h->stream);
        set_last_error(std::string("NDT launch chain made no progress for 30 s (stream: ") + hipGetErrorString(e) + ")");
        return LSR_ERR_HIP;
      }
    }
    if (on_poll && (spins & 0xF) == 0 && (st = (*on_poll)())) return st;
    if (wait_mode == WAIT_YIELD) std::this_thread::yield();
    else if (wait_mode == WAIT_SLEEP) std::this_thread::sleep_for(std::chrono::microseconds(20));
    else __builtin_ia32_pause();
  }
  int launched = 0;
  for (int c = 0; c < n_chains; c++) launched = std::max(launched, ch[c].launched);
  *launches_out = launched;
  return LSR_OK;
}

int ndt_hard_cap(const NdtParamsHost& p) {
  // per Newton iteration: 1 first pass + <=10 trials + 1 Hessian recomputation; max_iter+2 iterations; + initial pass
  return (p.max_iterations + 2) * 12 + 5;
}

int ndt_min_evals(const NdtParamsHost& p) {
  // with a non-positive epsilon the loop can only stop on the iteration count: at least max_iter+2 line searches
  if (p.trans_eps <= 0) return p.max_iterations + 4;  // + the finalising launch
  return 8;
}

// fitness_out (nullable): getFitnessScore(max_range) of every member as well (graph_based_slam_component.cpp:230-231 calls the two back
// to back).  For members whose neighbour grid is refined from their voxel grid the search of a member is enqueued on the side
// stream AS SOON AS ITS REGISTRATION HAS FINISHED, under the launch chain of the members still running; fitness_out[b] is NaN for
// a member that was not served that way (the caller falls back to lsr_get_fitness_score_batch for those).
int align_ndt_batch(lsr_handle* hs, int B, const float* guesses, float* finals, lsr_result* results, double* fitness_out = nullptr,
                    double max_range = 1.7976931348623157e308) {
  lsr_handle lead = hs[0];
  if (fitness_out) for (int b = 0; b < B; b++) fitness_out[b] = std::numeric_limits<double>::quiet_NaN();
  for (int b = 0; b < B; b++) {
    lsr_handle h = hs[b];
    if (!h->target || h->target->n == 0) { set_last_error("align before setInputTarget"); return LSR_ERR_NO_TARGET; }
    if (!h->has_source) { set_last_error("align before setInputSource"); return LSR_ERR_NO_SOURCE; }
    if (h->ndt.neighborhood != lead->ndt.neighborhood) { set_last_error("batched handles must share the neighbourhood method"); return LSR_ERR_INVALID_ARGUMENT; }
    int st = ensure_ndt_grid(h);
    if (st) return st;
    if (h->target->grid.ncells == 0) {  // no finite target point: there is no table a lookup could read
      set_last_error("the input target holds no finite point");
      return LSR_ERR_NO_TARGET;
    }
  }
  // The shared launches run on the lead's stream; a member on another stream may still have its setInputSource /
  // setInputTarget kernels in flight there: order the lead's stream after each of them (event + stream wait, no host wait)
  int st;
  for (int b = 1; b < B; b++)
    if ((st = order_lead_after(lead->stream, hs[b]))) return st;
  if ((st = lead->d_state.reserve(2 * (size_t)B))) return st;
  if ((st = lead->d_prob.reserve(B))) return st;
  {  // pinned, mapped: a set's init launch reads the members' records straight out of these arrays (ndt_init_batch)
    const NdtState* hs0 = lead->h_state.p; const NdtProblem* hp0 = lead->h_prob.p;
    if ((st = lead->h_state.reserve(2 * (size_t)B, hipHostMallocMapped))) return st;
    if ((st = lead->h_prob.reserve(B, hipHostMallocMapped))) return st;
    if (lead->h_state.p != hs0 || !lead->h_state_dev) LSR_HIP(hipHostGetDevicePointer((void**)&lead->h_state_dev, lead->h_state.p, 0));
    if (lead->h_prob.p != hp0 || !lead->h_prob_dev) LSR_HIP(hipHostGetDevicePointer((void**)&lead->h_prob_dev, lead->h_prob.p, 0));
  }
  // larger sets run as several independent launch chains (run_ndt_feeder), each on a stream of its own that starts behind
  // everything the lead's stream holds at this point (the members' builds and uploads) and does its own state uploads
  int n_chains = ndt_chain_count(B);
  if (B > 1 && (st = ensure_aux_streams(lead))) return st;   // once per lead handle
  n_chains = std::min(n_chains, 1 + lead->n_chain_streams);
  int chain_first[5] = {0, B, B, B, B};
  hipStream_t chain_streams[4] = {lead->stream, nullptr, nullptr, nullptr};
  // Every way out of this function after the fork joins the chain streams back into the lead's stream: whatever follows there (the
  // next align's state upload, a release of the banks) stays behind launches, sorts and uploads still queued on a chain stream.
  struct ChainJoin {
    lsr_handle lead; int n = 1; bool joined = false;
    bool join() {   // false: a stream had to be drained by the host instead
      if (joined) return true;
      joined = true;
      bool ok = true;
      for (int c = 1; c < n; c++)
        if (hipEventRecord(lead->chain_ev[c - 1], lead->chain_stream[c - 1]) != hipSuccess ||
            hipStreamWaitEvent(lead->stream, lead->chain_ev[c - 1], 0) != hipSuccess) {
          (void)hipStreamSynchronize(lead->chain_stream[c - 1]);
          ok = false;
        }
      return ok;
    }
    ~ChainJoin() { (void)join(); }
  } chain_join{lead};
  if (n_chains > 1) {
    LSR_HIP(hipEventRecord(lead->chain_fork_ev, lead->stream));
    for (int c = 0; c <= n_chains; c++) chain_first[c] = (int)((long)B * c / n_chains);
    chain_join.n = n_chains;
    for (int c = 1; c < n_chains; c++) {
      chain_streams[c] = lead->chain_stream[c - 1];
      LSR_HIP(hipStreamWaitEvent(chain_streams[c], lead->chain_fork_ev, 0));
    }
  }
  // launch geometry: where the leaf records are read from, which kernel, workgroup size (lead handle's tuning keys, 0 / -1 = automatic)
  NdtLaunchCfg cfg;
  cfg.batch = B;
  cfg.neighborhood = lead->ndt.neighborhood;
  choose_table_mode(lead, hs, B, cfg);
  // accumulator banks, one set per member, cleared before launch 0
  if ((st = lead->d_bins.reserve((size_t)B * NDT_NBANKS * NDT_BANK_WORDS))) return st;
  int max_blocks = 1, nb_full = 1;
  int min_evals = 1, hard_cap = 1;
  long pts = 0;
  for (int b = 0; b < B; b++) {
    lsr_handle h = hs[b];
    ndt_fill_initial_state(lead->h_state.p[2 * b], guesses ? guesses + 16 * b : nullptr, h->ndt, (int)h->source.n);
    lead->h_state.p[2 * b + 1] = lead->h_state.p[2 * b];
    if (cfg.sorted) {
      // order this member's source by voxel tile of its guess-moved points (4 launches on its chain's stream)
      int c = 0;
      while (c + 1 < n_chains && b >= chain_first[c + 1]) c++;
      if ((st = ndt_sort_source(h->source, lead->h_state.p[2 * b].T, h->target->grid, h->source_sorted, h->scratch, chain_streams[c]))) return st;
    }
    fill_problem(lead->h_prob.p[b], h, lead->d_state.p + 2 * b, lead->d_bins.p + (size_t)b * NDT_NBANKS * NDT_BANK_WORDS, cfg);
    max_blocks = std::max(max_blocks, lead->h_prob.p[b].nblocks);
    nb_full = std::max(nb_full, (int)((h->source.n + cfg_wg_points(cfg) - 1) / cfg_wg_points(cfg)));
    min_evals = std::max(min_evals, ndt_min_evals(h->ndt));
    hard_cap = std::max(hard_cap, ndt_hard_cap(h->ndt));
    pts += (long)h->source.n;
  }
  cfg.max_blocks = max_blocks;
  nb_full = std::min(nb_full, NDT_MAX_BLOCKS);
  // host mailboxes, one per registration (pinned, host-coherent, mapped into the device)
  if ((size_t)B > lead->mailbox.cap) {
    LSR_HIP(hipStreamSynchronize(lead->stream));  // launches still queued from the previous align report into the old block
    if ((st = lead->mailbox.reserve((size_t)B, hipHostMallocMapped | hipHostMallocCoherent))) return st;
    std::memset(lead->mailbox.p, 0, sizeof(NdtMailbox) * lead->mailbox.cap);
    LSR_HIP(hipHostGetDevicePointer((void**)&lead->d_mailbox, lead->mailbox.p, 0));
  }
  unsigned int token = ++lead->align_token;
  if (token == 0) token = ++lead->align_token;  // 0 is the mailbox's idle value
  for (int b = 0; b < B; b++) {
    lead->h_prob.p[b].mailbox = lead->d_mailbox + b;
    lead->h_state.p[2 * b].token = lead->h_state.p[2 * b + 1].token = (int)token;
  }
  const auto t0 = std::chrono::steady_clock::now();
  // profiling brackets: a set is timed from before its state uploads (every chain does its own) to the point where the lead's stream
  // has joined every chain; a single registration from after its one-launch initialisation, as before
  if (lead->profile && B > 1) LSR_HIP(hipEventRecord(lead->ev0, lead->stream));
  if (B == 1) {  // problem and state in the kernel arguments (no SDMA copy, no memset)
    if ((st = ndt_init_single(lead->h_state.p[0], lead->d_state.p, lead->d_bins.p, lead->stream))) return st;
  } else {
    for (int c = 0; c < n_chains; c++) {   // one launch per chain: problem records + initial states out of the pinned arrays, banks cleared
      const size_t b0 = (size_t)chain_first[c], nb = (size_t)(chain_first[c + 1] - chain_first[c]);
      if ((st = ndt_init_batch(lead->h_prob_dev + b0, lead->h_state_dev + 2 * b0, lead->d_prob.p + b0, lead->d_state.p + 2 * b0,
                               lead->d_bins.p + b0 * NDT_NBANKS * NDT_BANK_WORDS, (int)nb, chain_streams[c]))) return st;
    }
  }
  if (lead->profile && B == 1) LSR_HIP(hipEventRecord(lead->ev0, lead->stream));
  // A candidate set is scored right after it is registered (graph_based_slam_component.cpp:230-231): the neighbour grids that
  // getFitnessScore needs are refined from the voxel order NOW, on a side stream, under the launch chain — the chain is a
  // sequence of short dependent launches that does not fill the chip, the refinement is one wide launch per 16 targets.
  bool prefetched = false;
  if (B > 1 && nn_prefetch_enabled()) {
    std::vector<const VoxelGridDev*> vgs;
    std::vector<HashGridDev*> hgs;
    std::vector<lsr_handle> owners;
    for (int b = 0; b < B; b++) {
      lsr_handle h = hs[b];
      if (!hash_from_grid_possible(h) || target_is_shared(h) || h->target->has_hash) continue;
      bool seen = false;
      for (lsr_handle o : owners) seen = seen || (o->target == h->target);
      if (seen) continue;
      vgs.push_back(&h->target->grid); hgs.push_back(&h->target->hash); owners.push_back(h);
    }
    if (!vgs.empty()) {
      LSR_HIP(hipEventRecord(lead->side_fork_ev, lead->stream));        // the targets' builds are behind this point of the lead's stream
      LSR_HIP(hipStreamWaitEvent(lead->side_stream, lead->side_fork_ev, 0));
      if ((st = nn_build_hash_from_grids(vgs.data(), hgs.data(), (int)vgs.size(), lead->side_stream))) return st;
      LSR_HIP(hipEventRecord(lead->side_ev, lead->side_stream));
      prefetched = true;
      for (lsr_handle o : owners) o->target->has_hash = true;
    }
  }
  // eager fitness: members that can be served by the group search on the side stream (their grid is there or being refined there)
  std::vector<char> eager_ok((size_t)B, 0), eager_queued((size_t)B, 0), eager_sent((size_t)B, 0);   // eligible / finished and waiting for a group / search enqueued
  std::vector<int> eager_ready;
  bool eager = false;
  if (fitness_out && B > 1 && nn_prefetch_enabled()) {
    for (int b = 0; b < B; b++) {
      lsr_handle h = hs[b];
      bool twice = false;
      for (int a = 0; a < b; a++) twice = twice || (hs[a] == h);
      eager_ok[b] = (!twice && hash_from_grid_possible(h) && !target_is_shared(h) && h->target->has_hash) ? 1 : 0;
      eager = eager || eager_ok[b];
    }
    if (eager && !prefetched) {   // every grid was there already: the side stream still has to start behind the lead's stream
      LSR_HIP(hipEventRecord(lead->side_fork_ev, lead->stream));
      LSR_HIP(hipStreamWaitEvent(lead->side_stream, lead->side_fork_ev, 0));
    }
  }
  const NdtMailbox* MB = lead->mailbox.p;
  auto eager_dispatch = [&](bool all) -> int {
    for (int b = 0; b < B; b++)
      if (eager_ok[b] && !eager_queued[b] && __atomic_load_n(&MB[b].done, __ATOMIC_ACQUIRE) == token) { eager_queued[b] = 1; eager_ready.push_back(b); }
    // a launch group serves up to 12 members: wait for a full group while the chain is still running.  (Round 5 measured groups of
    // ONE for small sets — a share of 8 candidates, whose chain leaves the chip idle half of the time —: a search is a wide launch
    // of long-lived waves, the chain's workgroups queue behind them for wave slots and LDS, launches of 15 us took 40-80 us, and the
    // share got slower, not faster: 0.757 / 0.949 ms against 0.710 / 0.928 ms for align + fitness of two shares.  Bounding the searches
    // and the grid refinement to 256 / 512 / 1024 resident workgroups (stride loops) so that the chain always finds its slots made it
    // worse still — 1.22 / 0.99 / 0.78 ms against 0.68 for the first share: refinement + searches are ~0.4 ms of full-chip work, at a
    // fraction of the chip they outlast the 0.5 ms chain and the tail is paid at the reduced rate.)
    static const size_t group_min = [] { const char* e = std::getenv("LSR_FIT_GROUP_MIN"); const int v = e ? std::atoi(e) : 12; return (size_t)std::max(1, std::min(12, v)); }();
    while (!eager_ready.empty() && (all || eager_ready.size() >= group_min)) {
      const int n = (int)std::min<size_t>(12, eager_ready.size());
      std::vector<FitJob> jobs;
      for (int k = 0; k < n; k++) {
        const int b = eager_ready[k];
        jobs.push_back(FitJob{&hs[b]->source, MB[b].final_T, &hs[b]->target->hash, max_range, &hs[b]->scratch});
      }
      const int fst = nn_fitness_begin_group(jobs.data(), n, lead->side_stream);
      if (fst) return fst;
      for (int k = 0; k < n; k++) eager_sent[eager_ready[k]] = 1;
      eager_ready.erase(eager_ready.begin(), eager_ready.begin() + n);
    }
    return LSR_OK;
  };
  const std::function<int()> poll_hook = [&]() { return eager_dispatch(false); };
  int launches = 0;
  st = run_ndt_feeder(lead, lead->d_prob.p, lead->h_prob.p, cfg, min_evals, hard_cap, token, &launches, eager ? &poll_hook : nullptr,
                      (!cfg.quad && lane_widen_enabled()) ? nb_full : 0, n_chains, chain_first, chain_streams);
  // whatever follows on the lead's stream (the next align's uploads, a release of the banks) stays behind the chains' queued launches
  if (!chain_join.join() && !st) { set_last_error("joining a launch chain failed"); st = LSR_ERR_HIP; }
  if (eager && !st) st = eager_dispatch(true);
  if (eager) {
    (void)hipEventRecord(lead->side_ev, lead->side_stream);   // behind the last search
    for (int b = 0; b < B; b++) {   // every score travels through its member's host mailbox: polled, not synchronised for
      if (!eager_sent[b]) continue;
      double v = 0;
      const int fst = nn_fitness_end(hs[b]->scratch, lead->side_stream, &v);
      if (fst && !st) st = fst;
      if (!fst) fitness_out[b] = v;
    }
  }
  if (prefetched || eager) {
    // the side stream's work — grid refinement, searches — ends here: every later use of the grids and of the members' scratch
    // (any stream, any call) then needs no ordering against it (after the polls above only a kernel epilogue is left to wait for)
    if (hipEventSynchronize(lead->side_ev) != hipSuccess && !st) { set_last_error("side-stream work of the candidate set failed"); st = LSR_ERR_HIP; }
    if (st && prefetched) for (int b = 0; b < B; b++) if (hs[b]->target) hs[b]->target->has_hash = false;
  }
  if (st) return st;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const NdtMailbox* M = lead->mailbox.p;
  if (lead->profile) {
    // hipEvents around the launch chain exactly as production runs it; the span also holds the finalising launch and
    // the (at most four) queued launches that exit at their head, so the time per pass is slightly over-estimated
    LSR_HIP(hipEventRecord(lead->ev1, lead->stream));
    LSR_HIP(hipEventSynchronize(lead->ev1));
    float ev_ms = 0.f;
    LSR_HIP(hipEventElapsedTime(&ev_ms, lead->ev0, lead->ev1));
    int passes = 0;
    lead->prof.deriv_pairs = 0;
    for (int b = 0; b < B; b++) {
      passes = std::max(passes, M[b].n_evals);
      lead->prof.deriv_pairs += (int64_t)M[b].last_pairs;
    }
    lead->prof.deriv_ms_total += ev_ms;
    lead->prof.deriv_launches += passes;
    lead->prof.deriv_points += (int64_t)passes * pts;
  }
  for (int b = 0; b < B; b++) {
    lsr_handle h = hs[b];
    std::memcpy(h->final_T, M[b].final_T, sizeof(float) * 16);
    h->converged = M[b].converged;
    if (finals) std::memcpy(finals + 16 * b, M[b].final_T, sizeof(float) * 16);
    if (results) {
      results[b].converged = M[b].converged;
      results[b].iterations = M[b].nr_iterations;
      results[b].score = M[b].trans_probability;
      results[b].n_evaluations = M[b].n_evals;
      results[b].n_correspondences = (int)M[b].last_pairs;   // valid (point, voxel) pairs of the last derivative pass
      results[b].gpu_ms = ms;  // host clock from the state upload to the last raised flag (launches still queued are no-ops)
    }
  }
  return LSR_OK;
}

}  // namespace

extern "C" {

const char* lsr_version(void) { return "lidarslam_reg 0.1.0 (gfx950)"; }

const char* lsr_status_string(int status) {
  switch (status) {
    case LSR_OK: return "ok";
    case LSR_ERR_INVALID_ARGUMENT: return "invalid argument";
    case LSR_ERR_NO_DEVICE: return "no usable gfx950 device";
    case LSR_ERR_HIP: return "HIP runtime error";
    case LSR_ERR_NO_TARGET: return "input target not set";
    case LSR_ERR_NO_SOURCE: return "input source not set";
    case LSR_ERR_NOT_IMPLEMENTED: return "not implemented";
    case LSR_ERR_INDEX_OVERFLOW: return "voxel index overflow";
    case LSR_ERR_TOO_FEW_POINTS: return "too few points";
    default: return "unknown status";
  }
}

const char* lsr_last_error(void) { return g_last_error.c_str(); }

int lsr_device_count(int* count) {
  if (!count) return LSR_ERR_INVALID_ARGUMENT;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    *count = 0;
    return LSR_ERR_NO_DEVICE;
  }
  *count = n;
  return LSR_OK;
}

int lsr_create(int method, int device_id, void* stream, lsr_handle* out) {
  if (!out || (method != LSR_METHOD_NDT && method != LSR_METHOD_GICP)) {
    set_last_error("invalid registration method");  // scanmatcher_component.cpp:121-124
    return LSR_ERR_INVALID_ARGUMENT;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) {
    set_last_error("no HIP device available (this library has no CPU path)");
    return LSR_ERR_NO_DEVICE;
  }
  DeviceGuard guard(device_id);
  if (!guard.ok) return LSR_ERR_NO_DEVICE;
  lsr_handle h = new (std::nothrow) lsr_handle_s();
  if (!h) return LSR_ERR_HIP;
  h->method = method;
  h->device = device_id;
  // pclomp ctor defaults
  h->ndt.resolution = 1.0; h->ndt.step_size = 0.1; h->ndt.outlier_ratio = 0.55; h->ndt.trans_eps = 0.1;
  h->ndt.max_iterations = 35; h->ndt.neighborhood = LSR_DIRECT7; h->ndt.d1_sign = 1;
  // tuning defaults may be preset from the environment (A/B runs without touching the caller)
  if (const char* e = std::getenv("LSR_NDT_WORKGROUP")) {
    const int v = std::atoi(e);
    if (v == 64 || v == 128 || v == 512 || v == 1024) h->ndt_threads = v;
    else if (v != 0) fprintf(stderr, "[lidarslam_reg] LSR_NDT_WORKGROUP=%d ignored: 64 / 128 (quad kernel, points) or 512 / 1024 (lane kernel, threads)\n", v);
  }
  if (const char* e = std::getenv("LSR_NDT_TABLE_MODE")) { const int v = std::atoi(e); if (v >= -1 && v <= 3) h->ndt_table_mode = v; }
  if (const char* e = std::getenv("LSR_NDT_QUAD")) { const int v = std::atoi(e); if (v >= -1 && v <= 1) h->ndt_quad = v; }
  if (const char* e = std::getenv("LSR_NDT_SORT")) { const int v = std::atoi(e); if (v >= -1 && v <= 1) h->ndt_sort = v; }
  if (const char* e = std::getenv("LSR_NDT_SPLIT")) { const int v = std::atoi(e); if (v >= -1 && v <= 1) h->ndt_split = v; }
  if (const char* e = std::getenv("LSR_WAIT_MODE")) {   // 0 | 1 | 2 or spin | yield | sleep
    const std::string w(e);
    const int v = (w == "spin") ? 0 : (w == "yield") ? 1 : (w == "sleep") ? 2 : (w.size() == 1 && w[0] >= '0' && w[0] <= '2') ? w[0] - '0' : -1;
    if (v >= 0) h->scratch.wait_mode = v;
  }
  if (const char* e = std::getenv("LSR_GRID_BUILDER")) { h->scratch.force_sort_path = (std::atoi(e) == 1); }
  if (stream) {
    h->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
      set_last_error("hipStreamCreateWithFlags failed");
      delete h;
      return LSR_ERR_HIP;
    }
    h->own_stream = true;
  }
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess || h->d_T16.reserve(16) != LSR_OK) {
    set_last_error("handle resources could not be created");
    (void)lsr_destroy(h);  // releases whatever was created so far
    return LSR_ERR_HIP;
  }
  *out = h;
  return LSR_OK;
}

// Streams and events of one handle object (the public handle and the worker objects lsr_search_loop(top_k > 1) keeps in h->aux: a
// worker that led a batch owns a side stream, chain streams and their events just like a public handle).
static void release_handle_streams(lsr_handle h) {
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->side_stream) { (void)hipStreamSynchronize(h->side_stream); (void)hipStreamDestroy(h->side_stream); h->side_stream = nullptr; }
  if (h->side_ev) { (void)hipEventDestroy(h->side_ev); h->side_ev = nullptr; }
  if (h->chain_fork_ev) { (void)hipEventDestroy(h->chain_fork_ev); h->chain_fork_ev = nullptr; }
  for (int c = 0; c < 3; c++) {
    if (h->chain_stream[c]) { (void)hipStreamSynchronize(h->chain_stream[c]); (void)hipStreamDestroy(h->chain_stream[c]); h->chain_stream[c] = nullptr; }
    if (h->chain_ev[c]) { (void)hipEventDestroy(h->chain_ev[c]); h->chain_ev[c] = nullptr; }
  }
  h->n_chain_streams = 0;
  if (h->side_fork_ev) { (void)hipEventDestroy(h->side_fork_ev); h->side_fork_ev = nullptr; }
  if (h->ev0) { (void)hipEventDestroy(h->ev0); h->ev0 = nullptr; }
  if (h->ev1) { (void)hipEventDestroy(h->ev1); h->ev1 = nullptr; }
  if (h->own_stream && h->stream) { (void)hipStreamDestroy(h->stream); h->stream = nullptr; }
}

int lsr_destroy(lsr_handle h) {
  if (!h) return LSR_OK;
  DeviceGuard guard(h->device);
  for (auto& w : h->aux)   // the workers first: their launches may still read buffers the handle owns
    if (w) { release_handle_streams(w.get()); w->target.reset(); }
  release_handle_streams(h);
  h->target.reset();
  delete h;
  return LSR_OK;
}

int lsr_set_f64(lsr_handle h, int key, double v) {
  LSR_CHECK_HANDLE(h);
  switch (key) {
    case LSR_RESOLUTION:
      if (!(v > 0)) { set_last_error("resolution must be > 0"); return LSR_ERR_INVALID_ARGUMENT; }
      h->ndt.resolution = v;  // pclomp rebuilds the grid lazily when the resolution changes
      return LSR_OK;
    case LSR_TRANSFORMATION_EPSILON: h->ndt.trans_eps = v; h->gicp.trans_eps = v; return LSR_OK;
    case LSR_STEP_SIZE: h->ndt.step_size = v; return LSR_OK;
    case LSR_OUTLIER_RATIO: h->ndt.outlier_ratio = v; return LSR_OK;
    case LSR_MAX_CORRESPONDENCE_DISTANCE: h->gicp.max_corr_dist = v; return LSR_OK;
    case LSR_ROTATION_EPSILON: h->gicp.rot_eps = v; return LSR_OK;
    case LSR_EUCLIDEAN_FITNESS_EPSILON: h->euclidean_fitness_eps = v; return LSR_OK;
    case LSR_GICP_EPSILON: h->gicp.gicp_eps = v; h->source_cov_valid = false; if (h->target) h->target->has_cov = false; return LSR_OK;
    default: set_last_error("unknown f64 key"); return LSR_ERR_INVALID_ARGUMENT;
  }
}

int lsr_get_f64(lsr_handle h, int key, double* v) {
  LSR_CHECK_HANDLE(h);
  if (!v) return LSR_ERR_INVALID_ARGUMENT;
  switch (key) {
    case LSR_RESOLUTION: *v = h->ndt.resolution; return LSR_OK;
    case LSR_TRANSFORMATION_EPSILON: *v = (h->method == LSR_METHOD_NDT) ? h->ndt.trans_eps : h->gicp.trans_eps; return LSR_OK;
    case LSR_STEP_SIZE: *v = h->ndt.step_size; return LSR_OK;
    case LSR_OUTLIER_RATIO: *v = h->ndt.outlier_ratio; return LSR_OK;
    case LSR_MAX_CORRESPONDENCE_DISTANCE: *v = h->gicp.max_corr_dist; return LSR_OK;
    case LSR_ROTATION_EPSILON: *v = h->gicp.rot_eps; return LSR_OK;
    case LSR_EUCLIDEAN_FITNESS_EPSILON: *v = h->euclidean_fitness_eps; return LSR_OK;
    case LSR_GICP_EPSILON: *v = h->gicp.gicp_eps; return LSR_OK;
    default: set_last_error("unknown f64 key"); return LSR_ERR_INVALID_ARGUMENT;
  }
}

int lsr_set_i32(lsr_handle h, int key, int v) {
  LSR_CHECK_HANDLE(h);
  switch (key) {
    case LSR_MAX_ITERATIONS:
      if (v < 0) { set_last_error("max_iterations must be >= 0"); return LSR_ERR_INVALID_ARGUMENT; }
      h->ndt.max_iterations = v; h->gicp.max_iterations = v; return LSR_OK;
    case LSR_NEIGHBORHOOD:
      if (v < LSR_KDTREE || v > LSR_DIRECT1) { set_last_error("unknown neighbourhood"); return LSR_ERR_INVALID_ARGUMENT; }
      h->ndt.neighborhood = v; return LSR_OK;
    case LSR_NUM_THREADS: h->num_threads = v; return LSR_OK;          // CPU-only hint: accepted, ignored
    case LSR_K_CORRESPONDENCES:
      if (v < 3 || v > GICP_MAX_K) { set_last_error("k_correspondences out of range"); return LSR_ERR_INVALID_ARGUMENT; }
      h->gicp.k = v; h->source_cov_valid = false; if (h->target) h->target->has_cov = false; return LSR_OK;
    case LSR_MAX_INNER_ITERATIONS: h->gicp.max_inner = v; return LSR_OK;
    case LSR_RANSAC_ITERATIONS: h->ransac_iterations = v; return LSR_OK;  // no effect on NDT/GICP maths
    case LSR_HESSIAN_D1_SIGN: h->ndt.d1_sign = (v >= 0) ? 1 : -1; return LSR_OK;
    case LSR_PROFILE: h->profile = v ? 1 : 0; return LSR_OK;
    case LSR_NDT_WORKGROUP:
      if (v == 256) {   // rounds 1-3: the one-lane kernel's workgroup size (then the default).  That kernel is gone: automatic, said once.
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "[lidarslam_reg] LSR_NDT_WORKGROUP = 256 is a geometry of an earlier kernel: using the automatic choice (0)\n"); }
        h->ndt_threads = 0; return LSR_OK;
      }
      if (v != 0 && v != 64 && v != 128 && v != 512 && v != 1024) { set_last_error("NDT workgroup key must be 0 (auto), 64 / 128 (quad kernel: points) or 512 / 1024 (lane kernel: threads)"); return LSR_ERR_INVALID_ARGUMENT; }
      h->ndt_threads = v; return LSR_OK;
    case LSR_NDT_TABLE_MODE:
      if (v < -1 || v > 3) { set_last_error("NDT table mode must be -1 (auto), 0 dense, 1 compact, 2 LDS, 3 tile"); return LSR_ERR_INVALID_ARGUMENT; }
      h->ndt_table_mode = v; return LSR_OK;
    case LSR_NDT_QUAD:
      if (v < -1 || v > 1) { set_last_error("NDT quad mode must be -1 (auto), 0 or 1"); return LSR_ERR_INVALID_ARGUMENT; }
      h->ndt_quad = v; return LSR_OK;
    case LSR_NDT_SORT:
      if (v < -1 || v > 1) { set_last_error("NDT source ordering must be -1 (auto), 0 or 1"); return LSR_ERR_INVALID_ARGUMENT; }
      h->ndt_sort = v; return LSR_OK;
    case LSR_NDT_SPLIT:
      if (v < -1 || v > 1) { set_last_error("NDT split mode must be -1 (auto), 0 or 1"); return LSR_ERR_INVALID_ARGUMENT; }
      h->ndt_split = v; return LSR_OK;
    case LSR_GRID_BUILDER:
      if (v < 0 || v > 1) { set_last_error("grid builder must be 0 (auto) or 1 (radix-sort builder)"); return LSR_ERR_INVALID_ARGUMENT; }
      h->scratch.force_sort_path = (v == 1);
      if (h->target) h->target->has_grid = h->target->has_centroids = false;
      return LSR_OK;
    case LSR_WAIT_MODE:
      if (v < 0 || v > 2) { set_last_error("wait mode must be 0 (spin), 1 (yield) or 2 (sleep)"); return LSR_ERR_INVALID_ARGUMENT; }
      h->scratch.wait_mode = v; return LSR_OK;
    default: set_last_error("unknown i32 key"); return LSR_ERR_INVALID_ARGUMENT;
  }
}

int lsr_get_i32(lsr_handle h, int key, int* v) {
  LSR_CHECK_HANDLE(h);
  if (!v) return LSR_ERR_INVALID_ARGUMENT;
  switch (key) {
    case LSR_MAX_ITERATIONS: *v = (h->method == LSR_METHOD_NDT) ? h->ndt.max_iterations : h->gicp.max_iterations; return LSR_OK;
    case LSR_NEIGHBORHOOD: *v = h->ndt.neighborhood; return LSR_OK;
    case LSR_NUM_THREADS: *v = h->num_threads; return LSR_OK;
    case LSR_K_CORRESPONDENCES: *v = h->gicp.k; return LSR_OK;
    case LSR_MAX_INNER_ITERATIONS: *v = h->gicp.max_inner; return LSR_OK;
    case LSR_RANSAC_ITERATIONS: *v = h->ransac_iterations; return LSR_OK;
    case LSR_HESSIAN_D1_SIGN: *v = h->ndt.d1_sign; return LSR_OK;
    case LSR_PROFILE: *v = h->profile; return LSR_OK;
    case LSR_NDT_WORKGROUP: *v = h->ndt_threads; return LSR_OK;
    case LSR_NDT_TABLE_MODE: *v = h->ndt_table_mode; return LSR_OK;
    case LSR_NDT_QUAD: *v = h->ndt_quad; return LSR_OK;
    case LSR_NDT_SORT: *v = h->ndt_sort; return LSR_OK;
    case LSR_NDT_SPLIT: *v = h->ndt_split; return LSR_OK;
    case LSR_GRID_BUILDER: *v = h->scratch.force_sort_path ? 1 : 0; return LSR_OK;
    case LSR_WAIT_MODE: *v = h->scratch.wait_mode; return LSR_OK;
    case LSR_VOXEL_FILTER_FORM: *v = h->scratch.vg_form; return LSR_OK;
    default: set_last_error("unknown i32 key"); return LSR_ERR_INVALID_ARGUMENT;
  }
}

static int set_target_impl(lsr_handle h, const void* pts, size_t stride, size_t n, bool on_device) {
  LSR_CHECK_HANDLE(h);
  if (h->method == LSR_METHOD_NDT && !h->scratch.force_sort_path) {
    // a set of one: de-interleave and bounding box in one launch, the same counting-sort kernels (661k-point submap: 104 -> 94 us)
    lsr_handle one[1] = {h};
    const void* cloud[1] = {pts};
    const size_t count[1] = {n};
    return lsr_set_input_target_batch(one, 1, cloud, count, stride, on_device ? 1 : 0);
  }
  auto t = fresh_target(h);
  int st = upload_cloud(h, pts, stride, n, on_device, t->cloud);
  if (st) return st;
  t->n = n;
  h->target = t;
  if (h->method == LSR_METHOD_NDT) {
    // pclomp::NDT::setInputTarget -> init(): the voxel-covariance grid is built right here.
    st = ensure_ndt_grid(h);
    if (st) { h->target.reset(); return st; }
  } else {
    // GICP: target covariances are computed lazily in align() by the reference; the NN structure is
    // what setInputTarget pays for (kd-tree there, hash grid here).
    st = ensure_target_hash(h);
    if (st) { h->target.reset(); return st; }
  }
  LSR_HIP(hipStreamSynchronize(h->stream));
  return LSR_OK;
}

// Submap assembly + setInputTarget without a host round trip of the assembled cloud
// (scanmatcher_component.cpp:449-464,307; graph_based_slam_component.cpp:208-227)
// Move every frame by its pose and concatenate them in order into `out` (device SoA).
static int assemble_frames(lsr_handle h, int n_frames, const void* const* frames, const size_t* counts, size_t stride_bytes,
                           const float* poses16, bool on_device, DeviceCloud& out) {
  size_t total = 0, biggest = 0;
  for (int f = 0; f < n_frames; f++) {
    if (counts[f] > 0 && !frames[f]) { set_last_error("null frame pointer"); return LSR_ERR_INVALID_ARGUMENT; }
    total += counts[f];
    biggest = std::max(biggest, counts[f]);
  }
  if (total > (size_t)INT32_MAX / 2) { set_last_error("cloud too large"); return LSR_ERR_INVALID_ARGUMENT; }
  int st = out.resize(total);
  if (st) return st;
  if ((st = h->d_poses.reserve((size_t)n_frames * 16))) return st;
  LSR_HIP(hipMemcpyAsync(h->d_poses.p, poses16, sizeof(float) * 16 * n_frames, hipMemcpyHostToDevice, h->stream));
  if (!on_device && (st = h->staging.reserve(biggest * stride_bytes))) return st;
  size_t off = 0;
  for (int f = 0; f < n_frames; f++) {
    const void* d_aos = frames[f];
    if (!on_device && counts[f] > 0) {
      LSR_HIP(hipMemcpyAsync(h->staging.p, frames[f], counts[f] * stride_bytes, hipMemcpyHostToDevice, h->stream));
      d_aos = h->staging.p;
    }
    if ((st = transform_append(d_aos, stride_bytes, counts[f], h->d_poses.p + 16 * f, out, off, h->stream))) return st;
    if (!on_device) LSR_HIP(hipStreamSynchronize(h->stream));  // the staging buffer is reused by the next frame
    off += counts[f];
  }
  // poses16 is caller memory read by an asynchronous copy: it must have landed before we return to the caller
  if (on_device) LSR_HIP(hipStreamSynchronize(h->stream));
  return LSR_OK;
}

int lsr_set_input_target_frames(lsr_handle h, int n_frames, const void* const* frames, const size_t* counts, size_t stride_bytes,
                                const float* poses16, int on_device) {
  LSR_CHECK_HANDLE(h);
  if (n_frames <= 0 || !frames || !counts || !poses16 || stride_bytes < 12 || (stride_bytes % 4)) {
    set_last_error("bad frame list");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  auto t = fresh_target(h);
  int st = assemble_frames(h, n_frames, frames, counts, stride_bytes, poses16, on_device != 0, t->cloud);
  if (st) return st;
  t->n = t->cloud.n;
  h->target = t;
  st = (h->method == LSR_METHOD_NDT) ? ensure_ndt_grid(h) : ensure_target_hash(h);
  if (st) { h->target.reset(); return st; }
  LSR_HIP(hipStreamSynchronize(h->stream));
  return LSR_OK;
}

int lsr_set_input_target(lsr_handle h, const void* pts, size_t stride_bytes, size_t n) {
  return set_target_impl(h, pts, stride_bytes, n, false);
}

// setInputTarget for a set of candidates (graph_based_slam_component.cpp:181-227 runs once per candidate): the same work
// per object as lsr_set_input_target, staged so that the builds overlap on the device — every upload and bounding-box
// pass is enqueued (each on its object's stream) before the first box is waited for, every grid build before the first
// result is read.
int lsr_set_input_target_batch(lsr_handle* handles, int count, const void* const* clouds, const size_t* counts, size_t stride_bytes,
                               int on_device) {
  if (count < 0 || (count > 0 && (!handles || !clouds || !counts))) { set_last_error("bad batch arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  if (count == 0) return LSR_OK;
  for (int b = 0; b < count; b++) {
    if (!handles[b]) { set_last_error("null handle"); return LSR_ERR_INVALID_ARGUMENT; }
    if (handles[b]->device != handles[0]->device) { set_last_error("batched objects must live on one device"); return LSR_ERR_INVALID_ARGUMENT; }
    for (int a = 0; a < b; a++)
      if (handles[a] == handles[b]) { set_last_error("the same object appears twice in the batch"); return LSR_ERR_INVALID_ARGUMENT; }
  }
  DeviceGuard guard(handles[0]->device);   // for the whole call (the stages below enqueue on every member's stream)
  if (!guard.ok) { set_last_error("hipSetDevice failed"); return LSR_ERR_HIP; }
  auto fail = [&](int st) {
    for (int b = 0; b < count; b++) {   // nothing half-built stays behind
      (void)hipStreamSynchronize(handles[b]->stream);
      handles[b]->scratch.grid_pending = false;
      handles[b]->scratch.bbox_parts = 0;
      handles[b]->target.reset();
    }
    return st;
  };
  int st;
  if (stride_bytes < 12 || (stride_bytes % 4) != 0) { set_last_error("stride_bytes must be a multiple of 4 and >= 12"); return LSR_ERR_INVALID_ARGUMENT; }
  // NDT members are built by GROUP launches on the first member's stream (one launch per stage for up to 16 members: a
  // candidate set is bound by the host's launch rate otherwise); GICP members keep their own path and stream.
  hipStream_t lead_stream = handles[0]->stream;
  std::vector<TargetBuildJob> jobs;
  std::vector<int> job_of((size_t)count, -1);
  for (int b = 0; b < count; b++) {   // stage 0: uploads (+ de-interleave and bounding boxes of the NDT members in group launches)
    lsr_handle h = handles[b];
    if (counts[b] > 0 && !clouds[b]) { set_last_error("null point pointer"); return fail(LSR_ERR_INVALID_ARGUMENT); }
    if (counts[b] > (size_t)INT32_MAX / 2) { set_last_error("cloud too large"); return fail(LSR_ERR_INVALID_ARGUMENT); }
    auto t = fresh_target(h);
    t->n = counts[b];
    h->target = t;
    if (h->method != LSR_METHOD_NDT) {
      if ((st = settle_dep(h))) return fail(st);   // used on its own stream here
      if ((st = upload_cloud(h, clouds[b], stride_bytes, counts[b], on_device != 0, t->cloud))) return fail(st);
      if ((st = cloud_bbox_begin(t->cloud, h->scratch, h->stream))) return fail(st);
      continue;
    }
    if ((st = order_lead_after(lead_stream, h))) return fail(st);   // whatever this member still has in flight on its own stream comes first
    const void* d_aos = clouds[b];
    if (!on_device && counts[b] > 0) {
      if ((st = h->staging.reserve(counts[b] * stride_bytes))) return fail(st);
      if (hipMemcpyAsync(h->staging.p, clouds[b], counts[b] * stride_bytes, hipMemcpyHostToDevice, lead_stream) != hipSuccess) return fail(LSR_ERR_HIP);
      d_aos = h->staging.p;
    }
    job_of[b] = (int)jobs.size();
    jobs.push_back(TargetBuildJob{d_aos, stride_bytes, counts[b], &t->cloud, (float)h->ndt.resolution, &t->grid, &h->scratch, 0});
  }
  if (!jobs.empty() && (st = ndt_targets_ingest(jobs.data(), (int)jobs.size(), lead_stream))) return fail(st);
  // (Round 5 measured forking the neighbour-grid refinement of a small set onto the lead's side stream right after the scatter, under
  // the leaf sums and the host's work between the calls, instead of next to the first launches of the align chain: the refinement and
  // the leaf sums slow each other down — target + source stage of an 8-candidate share 0.44 ms against 0.28 ms, align 0.03-0.05 ms
  // faster, the share 0.14-0.17 ms slower.  Not done.)
  // stage 1: the rest of every build
  if (!jobs.empty() && (st = ndt_targets_build_begin(jobs.data(), (int)jobs.size(), lead_stream))) return fail(st);
  for (int b = 0; b < count; b++)
    if (handles[b]->method != LSR_METHOD_NDT && (st = ensure_target_hash(handles[b]))) return fail(st);
  for (int b = 0; b < count; b++) {   // stage 2: results
    lsr_handle h = handles[b];
    TargetData& t = *h->target;
    if (h->method == LSR_METHOD_NDT) {
      if ((st = ndt_build_grid_end(t.grid, h->scratch, lead_stream))) return fail(st);
      t.has_grid = true;
      t.has_centroids = false;
      t.grid_leaf = (float)h->ndt.resolution;
    } else if (hipStreamSynchronize(h->stream) != hipSuccess) {
      set_last_error("stream error in the target batch");
      return fail(LSR_ERR_HIP);
    }
  }
  if (hipStreamSynchronize(lead_stream) != hipSuccess) { set_last_error("stream error in the target batch"); return fail(LSR_ERR_HIP); }
  return LSR_OK;
}
int lsr_set_input_target_device(lsr_handle h, const void* dev_pts, size_t stride_bytes, size_t n) {
  return set_target_impl(h, dev_pts, stride_bytes, n, true);
}

static int set_source_impl(lsr_handle h, const void* pts, size_t stride, size_t n, bool on_device) {
  LSR_CHECK_HANDLE(h);
  int st = upload_cloud(h, pts, stride, n, on_device, h->source);
  if (st) return st;
  h->has_source = true;
  h->source_cov_valid = false;
  if (!on_device) LSR_HIP(hipStreamSynchronize(h->stream));  // the caller may reuse its host buffer
  return LSR_OK;
}

int lsr_set_input_source(lsr_handle h, const void* pts, size_t stride_bytes, size_t n) {
  return set_source_impl(h, pts, stride_bytes, n, false);
}
int lsr_set_input_source_device(lsr_handle h, const void* dev_pts, size_t stride_bytes, size_t n) {
  return set_source_impl(h, dev_pts, stride_bytes, n, true);
}

// setInputSource of every candidate of a set (graph_based_slam_component.cpp:181 per candidate): the de-interleaves of up to 16
// members share one launch on the first member's stream.
int lsr_set_input_source_batch(lsr_handle* handles, int count, const void* const* clouds, const size_t* counts, size_t stride_bytes,
                               int on_device) {
  if (count < 0 || (count > 0 && (!handles || !clouds || !counts))) { set_last_error("bad batch arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  if (count == 0) return LSR_OK;
  if (stride_bytes < 12 || (stride_bytes % 4) != 0) { set_last_error("stride_bytes must be a multiple of 4 and >= 12"); return LSR_ERR_INVALID_ARGUMENT; }
  for (int b = 0; b < count; b++) {
    if (!handles[b]) { set_last_error("null handle"); return LSR_ERR_INVALID_ARGUMENT; }
    if (handles[b]->device != handles[0]->device) { set_last_error("batched objects must live on one device"); return LSR_ERR_INVALID_ARGUMENT; }
    if (counts[b] > 0 && !clouds[b]) { set_last_error("null point pointer"); return LSR_ERR_INVALID_ARGUMENT; }
    if (counts[b] > (size_t)INT32_MAX / 2) { set_last_error("cloud too large"); return LSR_ERR_INVALID_ARGUMENT; }
    for (int a = 0; a < b; a++)
      if (handles[a] == handles[b]) { set_last_error("the same object appears twice in the batch"); return LSR_ERR_INVALID_ARGUMENT; }
  }
  DeviceGuard guard(handles[0]->device);
  if (!guard.ok) { set_last_error("hipSetDevice failed"); return LSR_ERR_HIP; }
  hipStream_t lead_stream = handles[0]->stream;
  std::vector<DeinterleaveJob> jobs((size_t)count);
  int st;
  // No member may be left claiming a source it never received: the old clouds are being overwritten, so every member loses its
  // source up front and gets it back only when the whole set has been staged and de-interleaved.  On a failure the staging
  // copies already enqueued are drained before returning (the callers may reuse their host buffers).
  for (int b = 0; b < count; b++) { handles[b]->has_source = false; handles[b]->source_cov_valid = false; }
  auto fail = [&](int status) {
    (void)hipStreamSynchronize(lead_stream);
    return status;
  };
  for (int b = 0; b < count; b++) {
    lsr_handle h = handles[b];
    if ((st = order_lead_after(lead_stream, h))) return fail(st);
    const void* d_aos = clouds[b];
    if (!on_device && counts[b] > 0) {
      if ((st = h->staging.reserve(counts[b] * stride_bytes))) return fail(st);
      if (hipMemcpyAsync(h->staging.p, clouds[b], counts[b] * stride_bytes, hipMemcpyHostToDevice, lead_stream) != hipSuccess) {
        set_last_error("staging copy of a source failed");
        return fail(LSR_ERR_HIP);
      }
      d_aos = h->staging.p;
    }
    jobs[b] = DeinterleaveJob{d_aos, stride_bytes, counts[b], &h->source};
  }
  if ((st = deinterleave_group(jobs.data(), count, lead_stream))) return fail(st);
  // every member's own stream continues after the shared launch (its next align / fitness call runs there): deferred — the members
  // remember the event, their streams wait for it when they are next used on their own (handle.hpp: StreamDep)
  if ((st = defer_members_behind(handles[0], handles, count))) return fail(st);
  if (!on_device && hipStreamSynchronize(lead_stream) != hipSuccess) { set_last_error("stream error in the source batch"); return LSR_ERR_HIP; }
  for (int b = 0; b < count; b++) handles[b]->has_source = true;
  return LSR_OK;
}

// How the filtered-source entries return.  voxel_grid_filter has polled a mailbox word written by a kernel that runs BEHIND the pass
// that read the caller's buffer (the run count; the bounding box when no point is finite): the caller's buffer — host or device — has
// been consumed and may be reused.  What can still be running is the centroid launch, which reads and writes buffers the object owns;
// everything that uses the source afterwards is enqueued on the object's stream (or orders itself behind it: order_lead_after), so
// the call does not wait for it — the align that follows in the frontend loop starts under it (env LSR_SOURCE_SYNC=1: wait, as until
// round 5).
static int finish_source_call(lsr_handle h) {
  static const bool wait = [] { const char* e = std::getenv("LSR_SOURCE_SYNC"); return e && e[0] == '1'; }();
  if (wait) LSR_HIP(hipStreamSynchronize(h->stream));
  return LSR_OK;
}

// pcl::VoxelGrid::filter + registration_->setInputSource, without the cloud leaving HBM
// (scanmatcher_component.cpp:324-329)
int lsr_set_input_source_filtered(lsr_handle h, const void* pts, size_t stride_bytes, size_t n, float leaf, int on_device,
                                  size_t* n_out) {
  LSR_CHECK_HANDLE(h);
  if (!(leaf > 0)) { set_last_error("leaf size must be > 0"); return LSR_ERR_INVALID_ARGUMENT; }
  int st = upload_cloud(h, pts, stride_bytes, n, on_device != 0, h->raw);
  if (st) return st;
  if ((st = voxel_grid_filter(h->raw, leaf, h->source, h->scratch, h->stream))) return st;
  h->has_source = true;
  h->source_cov_valid = false;
  if (n_out) *n_out = h->source.n;
  return finish_source_call(h);
}

// The frontend's whole per-scan preprocessing in one call, on the device: min-max range filter
// (scanmatcher_component.cpp:210-218) -> VoxelGrid(vg_size_for_input) (:324-328) -> setInputSource (:329).
int lsr_set_input_source_frontend(lsr_handle h, const void* pts, size_t stride_bytes, size_t n, double scan_min_range,
                                  double scan_max_range, float vg_size_for_input, int on_device, size_t* n_out) {
  LSR_CHECK_HANDLE(h);
  if (!(vg_size_for_input > 0)) { set_last_error("leaf size must be > 0"); return LSR_ERR_INVALID_ARGUMENT; }
  if (stride_bytes < 12 || (stride_bytes % 4) != 0) { set_last_error("stride_bytes must be a multiple of 4 and >= 12"); return LSR_ERR_INVALID_ARGUMENT; }
  if (n > 0 && !pts) { set_last_error("null point pointer"); return LSR_ERR_INVALID_ARGUMENT; }
  // strided xyz records are a PointCloud2 payload without an intensity field: the same one-launch ingest
  const lsr_pc2_layout rec_layout{(uint32_t)stride_bytes, 0u, 4u, 8u, -1};
  int st = ingest_pc2(h, pts, n, &rec_layout, on_device != 0, true, scan_min_range, scan_max_range, h->raw);
  if (st) return st;
  if ((st = voxel_grid_filter(h->raw, vg_size_for_input, h->source, h->scratch, h->stream))) return st;
  h->has_source = true;
  h->source_cov_valid = false;
  if (n_out) *n_out = h->source.n;
  return finish_source_call(h);
}

// pcl::VoxelGrid::filter as a stand-alone device operation (host in, host out)
int lsr_voxel_grid_filter(lsr_handle h, const void* pts, size_t stride_bytes, size_t n, float leaf, void* out_pts,
                          size_t out_stride_bytes, size_t out_capacity, size_t* n_out) {
  LSR_CHECK_HANDLE(h);
  if (!(leaf > 0) || !n_out || out_stride_bytes < 12 || (out_stride_bytes % 4)) { set_last_error("bad argument"); return LSR_ERR_INVALID_ARGUMENT; }
  int st = upload_cloud(h, pts, stride_bytes, n, false, h->raw);
  if (st) return st;
  if ((st = voxel_grid_filter(h->raw, leaf, h->filtered, h->scratch, h->stream))) return st;
  *n_out = h->filtered.n;
  if (h->filtered.n > out_capacity) { set_last_error("output buffer too small"); return LSR_ERR_INVALID_ARGUMENT; }
  if (h->filtered.n == 0) return LSR_OK;
  const size_t bytes = h->filtered.n * out_stride_bytes;
  if ((st = h->staging.reserve(bytes))) return st;
  LSR_HIP(hipMemsetAsync(h->staging.p, 0, bytes, h->stream));
  if ((st = interleave(h->filtered, h->staging.p, out_stride_bytes, h->stream))) return st;
  LSR_HIP(hipMemcpyAsync(out_pts, h->staging.p, bytes, hipMemcpyDeviceToHost, h->stream));
  LSR_HIP(hipStreamSynchronize(h->stream));
  return LSR_OK;
}

// ---- N4: PointCloud2 codec ---------------------------------------------------------------------
namespace {
int check_layout(const lsr_pc2_layout* L) {
  if (!L || L->point_step < 12 || (L->point_step % 4) != 0) { set_last_error("PointCloud2 layout: point_step must be a multiple of 4 and >= 12"); return LSR_ERR_INVALID_ARGUMENT; }
  const uint32_t offs[3] = {L->offset_x, L->offset_y, L->offset_z};
  for (uint32_t o : offs)
    if ((o % 4) != 0 || o + 4 > L->point_step) { set_last_error("PointCloud2 layout: x/y/z offsets must be 4-byte aligned and inside point_step"); return LSR_ERR_INVALID_ARGUMENT; }
  if (L->offset_intensity >= 0 && ((L->offset_intensity % 4) != 0 || (uint32_t)L->offset_intensity + 4 > L->point_step)) {
    set_last_error("PointCloud2 layout: intensity offset must be 4-byte aligned and inside point_step (or < 0 for none)");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  return LSR_OK;
}

// payload -> SoA planes (+ intensity) on the device
// SoA planes -> host payload (bytes outside the four fields are zero)
int write_pc2_host(lsr_handle h, const DeviceCloud& cloud, void* out_data, size_t capacity, const lsr_pc2_layout* L, size_t* n_out) {
  *n_out = cloud.n;
  if (cloud.n > capacity) { set_last_error("output buffer too small"); return LSR_ERR_INVALID_ARGUMENT; }
  if (cloud.n == 0) return LSR_OK;
  const size_t bytes = cloud.n * L->point_step;
  int st = h->staging.reserve(bytes);
  if (st) return st;
  LSR_HIP(hipMemsetAsync(h->staging.p, 0, bytes, h->stream));
  if ((st = pc2_write(cloud, h->staging.p, (int)L->point_step, (int)L->offset_x, (int)L->offset_y, (int)L->offset_z, L->offset_intensity, h->stream)))
    return st;
  LSR_HIP(hipMemcpyAsync(out_data, h->staging.p, bytes, hipMemcpyDeviceToHost, h->stream));
  LSR_HIP(hipStreamSynchronize(h->stream));
  return LSR_OK;
}
}  // namespace

int lsr_set_input_source_pc2(lsr_handle h, const void* data, size_t n_points, const lsr_pc2_layout* layout, double scan_min_range,
                             double scan_max_range, float vg_size_for_input, int on_device, size_t* n_out) {
  LSR_CHECK_HANDLE(h);
  int st = check_layout(layout);
  if (st) return st;
  if (!(vg_size_for_input > 0)) { set_last_error("leaf size must be > 0"); return LSR_ERR_INVALID_ARGUMENT; }
  // payload -> planes + range filter + bounding-box records in one launch; the voxel filter behind it reads the records on the device
  if ((st = ingest_pc2(h, data, n_points, layout, on_device != 0, true, scan_min_range, scan_max_range, h->raw))) return st;
  if ((st = voxel_grid_filter(h->raw, vg_size_for_input, h->source, h->scratch, h->stream))) return st;
  h->has_source = true;
  h->source_cov_valid = false;
  if (n_out) *n_out = h->source.n;
  return finish_source_call(h);
}

int lsr_get_source_pc2(lsr_handle h, void* out_data, size_t capacity_points, const lsr_pc2_layout* layout, size_t* n_out) {
  LSR_CHECK_HANDLE(h);
  int st = check_layout(layout);
  if (st) return st;
  if (!n_out || (capacity_points > 0 && !out_data)) { set_last_error("bad argument"); return LSR_ERR_INVALID_ARGUMENT; }
  if (!h->has_source) { set_last_error("no input source"); return LSR_ERR_NO_SOURCE; }
  return write_pc2_host(h, h->source, out_data, capacity_points, layout, n_out);
}

// the same into a DEVICE buffer (a keyframe kept resident in HBM: what lsr_set_input_target_frames takes with on_device != 0):
// enqueued on the handle's stream, which is synchronised before returning so that any stream may read the records
int lsr_get_source_pc2_device(lsr_handle h, void* d_out, size_t capacity_points, const lsr_pc2_layout* layout, size_t* n_out) {
  LSR_CHECK_HANDLE(h);
  int st = check_layout(layout);
  if (st) return st;
  if (!n_out || (capacity_points > 0 && !d_out)) { set_last_error("bad argument"); return LSR_ERR_INVALID_ARGUMENT; }
  if (!h->has_source) { set_last_error("no input source"); return LSR_ERR_NO_SOURCE; }
  const DeviceCloud& cloud = h->source;
  *n_out = cloud.n;
  if (cloud.n > capacity_points) { set_last_error("output buffer too small"); return LSR_ERR_INVALID_ARGUMENT; }
  if (cloud.n == 0) return LSR_OK;
  LSR_HIP(hipMemsetAsync(d_out, 0, cloud.n * layout->point_step, h->stream));
  if ((st = pc2_write(cloud, d_out, (int)layout->point_step, (int)layout->offset_x, (int)layout->offset_y, (int)layout->offset_z,
                      layout->offset_intensity, h->stream))) return st;
  LSR_HIP(hipStreamSynchronize(h->stream));
  return LSR_OK;
}

int lsr_voxel_grid_filter_pc2(lsr_handle h, const void* data, size_t n_points, const lsr_pc2_layout* in_layout, float leaf, void* out_data,
                              size_t capacity_points, const lsr_pc2_layout* out_layout, size_t* n_out) {
  LSR_CHECK_HANDLE(h);
  int st = check_layout(in_layout);
  if (st) return st;
  if ((st = check_layout(out_layout))) return st;
  if (!(leaf > 0) || !n_out) { set_last_error("bad argument"); return LSR_ERR_INVALID_ARGUMENT; }
  if ((st = ingest_pc2(h, data, n_points, in_layout, false, false, 0.0, 0.0, h->raw))) return st;
  if ((st = voxel_grid_filter(h->raw, leaf, h->filtered, h->scratch, h->stream))) return st;
  return write_pc2_host(h, h->filtered, out_data, capacity_points, out_layout, n_out);
}

int lsr_wait_stream(lsr_handle h, void* producer_stream) {
  LSR_CHECK_HANDLE(h);
  if ((hipStream_t)producer_stream == h->stream) return LSR_OK;  // same stream: already ordered
  LSR_HIP(hipEventRecord(h->ev0, (hipStream_t)producer_stream));
  LSR_HIP(hipStreamWaitEvent(h->stream, h->ev0, 0));
  return LSR_OK;
}

int lsr_share_target(lsr_handle h, lsr_handle owner) {
  LSR_CHECK_HANDLE(h);
  if (!owner || !owner->target) { set_last_error("owner has no target"); return LSR_ERR_NO_TARGET; }
  if (owner->device != h->device) { set_last_error("handles live on different devices"); return LSR_ERR_INVALID_ARGUMENT; }
  // The frontend's hand-over (scanmatcher_component.cpp:298-320: the target the map thread assembled is taken over at the start of a
  // callback): the target h held until now goes back to the owner as its SPARE when nobody else holds it, so that the owner's next
  // setInputTarget recycles its buffers (fresh_target) instead of allocating while h still reads the new one — two TargetData
  // objects then alternate for the life of the node, no hipMalloc / hipFree per map update.  The caller must not be inside another
  // call on `owner` at this moment (the hand-over happens after the map thread's job has finished).
  std::shared_ptr<TargetData> old = h->target;
  h->target = owner->target;
  if (old && old != owner->target) {
    if (h->spare_target == old) h->spare_target.reset();
    if (old.use_count() == 1 && (!owner->spare_target || owner->spare_target == owner->target)) owner->spare_target = old;
  }
  return LSR_OK;
}

int lsr_align_batch(lsr_handle* handles, int batch, const float* guesses, float* finals, lsr_result* results) {
  if (!handles || batch <= 0) { set_last_error("empty batch"); return LSR_ERR_INVALID_ARGUMENT; }
  for (int b = 0; b < batch; b++) {
    if (!handles[b]) { set_last_error("null handle in batch"); return LSR_ERR_INVALID_ARGUMENT; }
    if (handles[b]->device != handles[0]->device || handles[b]->method != handles[0]->method) {
      set_last_error("batched handles must share device and method");
      return LSR_ERR_INVALID_ARGUMENT;
    }
  }
  lsr_handle lead = handles[0];
  LSR_CHECK_HANDLE(lead);
  // NDT: the shared chain's stream is ordered after whatever the members still have in flight (align_ndt_batch), no host wait
  if (lead->method == LSR_METHOD_NDT) return align_ndt_batch(handles, batch, guesses, finals, results);
  // GICP: every registration is a chain of small dependent launches on its own object's stream; the chains of the batch are fed
  // side by side by one host loop (an object may appear only once: a chain owns its object's workspace and mailbox)
  for (int b = 0; b < batch; b++)
    for (int a = 0; a < b; a++)
      if (handles[a] == handles[b]) { set_last_error("the same object appears twice in the batch"); return LSR_ERR_INVALID_ARGUMENT; }
  for (int b = 1; b < batch; b++)
    if (settle_dep(handles[b])) return LSR_ERR_HIP;   // each member runs on its own stream
  return gicp_align_batch(handles, batch, guesses, finals, results);
}

// registration_->align() followed by registration_->getFitnessScore() for every candidate of a set
// (graph_based_slam_component.cpp:230-231): what lsr_align_batch + lsr_get_fitness_score_batch return, with the searches of the
// members that finish early running under the launch chain of the others.
int lsr_align_fitness_batch(lsr_handle* handles, int batch, const float* guesses, float* finals, lsr_result* results, double max_range,
                            double* fitness) {
  if (!handles || batch <= 0 || !fitness) { set_last_error("empty batch"); return LSR_ERR_INVALID_ARGUMENT; }
  for (int b = 0; b < batch; b++) {
    if (!handles[b]) { set_last_error("null handle in batch"); return LSR_ERR_INVALID_ARGUMENT; }
    if (handles[b]->device != handles[0]->device || handles[b]->method != handles[0]->method) {
      set_last_error("batched handles must share device and method");
      return LSR_ERR_INVALID_ARGUMENT;
    }
  }
  // an object can hold ONE result: a handle listed twice would be scored at whichever of its poses was written last
  for (int b = 0; b < batch; b++)
    for (int a = 0; a < b; a++)
      if (handles[a] == handles[b]) { set_last_error("the same object appears twice in the batch"); return LSR_ERR_INVALID_ARGUMENT; }
  lsr_handle lead = handles[0];
  LSR_CHECK_HANDLE(lead);
  if (lead->method != LSR_METHOD_NDT) {
    const int st = lsr_align_batch(handles, batch, guesses, finals, results);
    return st ? st : lsr_get_fitness_score_batch(handles, batch, max_range, fitness);
  }
  int st = align_ndt_batch(handles, batch, guesses, finals, results, fitness, max_range);
  if (st) return st;
  std::vector<lsr_handle> rest;
  std::vector<int> where;
  for (int b = 0; b < batch; b++)
    if (fitness[b] != fitness[b]) { rest.push_back(handles[b]); where.push_back(b); }   // not served under the chain
  if (!rest.empty()) {
    std::vector<double> v(rest.size());
    if ((st = lsr_get_fitness_score_batch(rest.data(), (int)rest.size(), max_range, v.data()))) return st;
    for (size_t k = 0; k < rest.size(); k++) fitness[where[k]] = v[k];
  }
  return LSR_OK;
}

int lsr_align(lsr_handle h, const float* guess, float* final_transformation, lsr_result* result, void* output_pts,
              size_t out_stride_bytes) {
  LSR_CHECK_HANDLE(h);
  int st;
  if (h->method == LSR_METHOD_NDT) {
    lsr_handle hs[1] = {h};
    st = align_ndt_batch(hs, 1, guess, final_transformation, result);
  } else {
    st = gicp_align(h, guess, final_transformation, result);
  }
  if (st) return st;
  if (output_pts) {
    if (out_stride_bytes < 12 || (out_stride_bytes % 4) != 0) { set_last_error("bad output stride"); return LSR_ERR_INVALID_ARGUMENT; }
    // PCL's align() first copies the source records into `output` and then overwrites x, y, z with the transformed
    // coordinates; every other field (intensity, ...) stays.  Same here: only the 12 xyz bytes of each record are written,
    // whatever the caller put in the rest of the record survives (12 bytes per point cross PCIe, packed).
    const size_t n = h->source.n;
    if ((st = h->staging.reserve(n * 12))) return st;
    if ((st = h->out_xyz.reserve(n * 3))) return st;
    LSR_HIP(hipMemcpyAsync(h->d_T16.p, h->final_T, 16 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if ((st = transform_to_strided(h->source, h->d_T16.p, h->staging.p, 12, h->stream))) return st;
    LSR_HIP(hipMemcpyAsync(h->out_xyz.p, h->staging.p, n * 12, hipMemcpyDeviceToHost, h->stream));
    LSR_HIP(hipStreamSynchronize(h->stream));
    unsigned char* o = static_cast<unsigned char*>(output_pts);
    for (size_t i = 0; i < n; i++) std::memcpy(o + i * out_stride_bytes, h->out_xyz.p + 3 * i, 12);
  }
  return LSR_OK;
}

int lsr_get_final_transformation(lsr_handle h, float* out16) {
  LSR_CHECK_HANDLE(h);
  if (!out16) return LSR_ERR_INVALID_ARGUMENT;
  std::memcpy(out16, h->final_T, sizeof(float) * 16);
  return LSR_OK;
}

int lsr_has_converged(lsr_handle h, int* out) {
  LSR_CHECK_HANDLE(h);
  if (!out) return LSR_ERR_INVALID_ARGUMENT;
  *out = h->converged;
  return LSR_OK;
}

int lsr_get_fitness_score(lsr_handle h, double max_range, double* out) {
  LSR_CHECK_HANDLE(h);
  if (!out) return LSR_ERR_INVALID_ARGUMENT;
  if (!h->target || h->target->n == 0) { set_last_error("getFitnessScore before setInputTarget"); return LSR_ERR_NO_TARGET; }
  if (!h->has_source) { set_last_error("getFitnessScore before setInputSource"); return LSR_ERR_NO_SOURCE; }
  int st = ensure_target_hash(h);
  if (st) return st;
  return nn_fitness_score(h->source, h->final_T, h->target->hash, max_range, out, h->scratch, h->d_T16, h->stream);
}

// getFitnessScore for a set of candidates (graph_based_slam_component.cpp:231 per candidate): every search + reduction is
// enqueued on its object's stream before the first result is waited for.
int lsr_get_fitness_score_batch(lsr_handle* handles, int count, double max_range, double* out) {
  if (count < 0 || (count > 0 && (!handles || !out))) { set_last_error("bad batch arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  int st;
  if (count == 0) return LSR_OK;
  for (int b = 0; b < count; b++) {
    lsr_handle h = handles[b];
    if (!h) { set_last_error("null handle"); return LSR_ERR_INVALID_ARGUMENT; }
    if (h->device != handles[0]->device) { set_last_error("batched objects must live on one device"); return LSR_ERR_INVALID_ARGUMENT; }
    if (!h->target || h->target->n == 0) { set_last_error("getFitnessScore before setInputTarget"); return LSR_ERR_NO_TARGET; }
    if (!h->has_source) { set_last_error("getFitnessScore before setInputSource"); return LSR_ERR_NO_SOURCE; }
    // an object's scratch, mailbox and transform buffer hold ONE reduction in flight
    for (int a = 0; a < b; a++)
      if (handles[a] == h) { set_last_error("the same object appears twice in the batch"); return LSR_ERR_INVALID_ARGUMENT; }
  }
  DeviceGuard guard(handles[0]->device);
  if (!guard.ok) { set_last_error("hipSetDevice failed"); return LSR_ERR_HIP; }
  // Members whose neighbour grid can be refined from their voxel grid (NDT, counting-sort builder) are served by GROUP launches
  // on the first member's stream: one launch for the grids, three per group for search + reduction.  The others keep the
  // staged per-member path on their own streams.
  hipStream_t lead_stream = handles[0]->stream;
  std::vector<int> grouped, single;
  std::vector<const VoxelGridDev*> vgs;
  std::vector<HashGridDev*> hgs;
  for (int b = 0; b < count; b++) {
    lsr_handle h = handles[b];
    if (hash_from_grid_possible(h) && !target_is_shared(h)) {
      grouped.push_back(b);
      if (!h->target->has_hash) { vgs.push_back(&h->target->grid); hgs.push_back(&h->target->hash); }
      if ((st = order_lead_after(lead_stream, h))) return st;   // the member's source upload / anything else in flight on its own stream
    } else {
      single.push_back(b);
    }
  }
  int first_error = LSR_OK;
  if (!vgs.empty()) {
    if ((st = nn_build_hash_from_grids(vgs.data(), hgs.data(), (int)vgs.size(), lead_stream))) return st;
    for (int b : grouped) handles[b]->target->has_hash = true;
  }
  if (!grouped.empty()) {
    std::vector<FitJob> jobs;
    for (int b : grouped) jobs.push_back(FitJob{&handles[b]->source, handles[b]->final_T, &handles[b]->target->hash, max_range, &handles[b]->scratch});
    if ((st = nn_fitness_begin_group(jobs.data(), (int)jobs.size(), lead_stream))) return st;
  }
  int begun = 0;
  for (; begun < (int)single.size(); begun++) {
    lsr_handle h = handles[single[begun]];
    if ((st = settle_dep(h))) break;   // used on its own stream here
    if ((st = ensure_target_hash(h))) break;
    if ((st = nn_fitness_begin(h->source, h->final_T, h->target->hash, max_range, h->scratch, h->d_T16, h->stream))) break;
  }
  if (begun < (int)single.size()) first_error = st;
  for (int b : grouped) {   // collect what was enqueued even after an error: no reduction stays in flight
    double v = 0;
    st = nn_fitness_end(handles[b]->scratch, lead_stream, &v);
    if (st && !first_error) first_error = st;
    out[b] = v;
  }
  for (int k = 0; k < begun; k++) {
    const int b = single[k];
    double v = 0;
    st = nn_fitness_end(handles[b]->scratch, handles[b]->stream, &v);
    if (st && !first_error) first_error = st;
    out[b] = v;
  }
  return first_error;
}

// ---- N3: loop-closure gate ---------------------------------------------------------------------
namespace {
// tf2::fromMsg(geometry_msgs::Pose) -> Eigen::Affine3d = Translation * Quaterniond (Eigen's toRotationMatrix, no
// normalisation), column-major 4x4 (graph_based_slam_component.cpp:171,177,219,236-238)
void submap_pose_matrix(const lsr_submap& s, double M[16]) {
  const double x = s.orientation[0], y = s.orientation[1], z = s.orientation[2], w = s.orientation[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
               tzz = tz * z;
  M[0] = 1 - (tyy + tzz); M[4] = txy - twz;       M[8] = txz + twy;        M[12] = s.position[0];
  M[1] = txy + twz;       M[5] = 1 - (txx + tzz); M[9] = tyz - twx;        M[13] = s.position[1];
  M[2] = txz - twy;       M[6] = tyz + twx;       M[10] = 1 - (txx + tyy); M[14] = s.position[2];
  M[3] = 0; M[7] = 0; M[11] = 0; M[15] = 1;
}
void mat4_mul(const double* A, const double* B, double* C) {  // column-major C = A * B
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      double acc = 0;
      for (int k = 0; k < 4; k++) acc += A[r + 4 * k] * B[k + 4 * c];
      C[r + 4 * c] = acc;
    }
}
void isometry_inverse(const double* M, double* I) {  // Eigen::Isometry3d::inverse(): [R^T | -R^T t]
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) I[r + 4 * c] = M[c + 4 * r];
  for (int r = 0; r < 3; r++) I[r + 12] = -(I[r] * M[12] + I[r + 4] * M[13] + I[r + 8] * M[14]);
  I[3] = I[7] = I[11] = 0; I[15] = 1;
}
}  // namespace

int lsr_search_loop(lsr_handle h, const lsr_submap* submaps, int num_submaps, size_t stride_bytes, int on_device,
                    const lsr_loop_params* params, lsr_loop_edge* edges, int edge_capacity, int* n_evaluated) {
  LSR_CHECK_HANDLE(h);
  if (!submaps || num_submaps <= 0 || !params || !n_evaluated || stride_bytes < 12 || (stride_bytes % 4) ||
      (edge_capacity > 0 && !edges) || edge_capacity < 0) {
    set_last_error("bad argument");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  if (!(params->voxel_leaf_size > 0) || params->search_submap_num < 0) {
    set_last_error("voxel_leaf_size must be > 0 and search_submap_num >= 0");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  *n_evaluated = 0;
  const lsr_submap& latest = submaps[num_submaps - 1];
  if (latest.n_points == 0 || !latest.cloud) { set_last_error("latest submap has no points"); return LSR_ERR_NO_SOURCE; }

  // source = latest submap moved by its own pose (:171-181)
  double init_M[16];
  submap_pose_matrix(latest, init_M);
  float pose_f[16];
  for (int k = 0; k < 16; k++) pose_f[k] = (float)init_M[k];
  {
    const void* fr[1] = {latest.cloud};
    const size_t cn[1] = {latest.n_points};
    int st = assemble_frames(h, 1, fr, cn, stride_bytes, pose_f, on_device != 0, h->source);
    if (st) return st;
    h->has_source = true;
    h->source_cov_valid = false;
  }

  // candidates (:188-205): enough travel since, close enough now; nearest first (ties: lower index, as the
  // strict `dist < min_dist` of the reference keeps the first minimum)
  std::vector<std::pair<double, int>> cand;
  for (int i = 0; i < num_submaps; i++) {
    const double dx = latest.position[0] - submaps[i].position[0], dy = latest.position[1] - submaps[i].position[1],
                 dz = latest.position[2] - submaps[i].position[2];
    const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (latest.distance - submaps[i].distance > params->distance_loop_closure && dist < params->range_of_searching_loop_closure)
      cand.emplace_back(dist, i);
  }
  if (cand.empty()) return LSR_OK;
  std::stable_sort(cand.begin(), cand.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
  const int k_eval = std::min({std::max(params->top_k, 1), (int)cand.size(), edge_capacity});

  // One worker registration object per candidate evaluated together.  top_k = 1 (the reference) uses `h` itself; with
  // top_k > 1 and NDT the k windows are assembled into k auxiliary objects (own target, a device copy of the source, the
  // caller's stream) and advanced in ONE shared launch chain (align_ndt_batch) instead of k chains one after another.
  const bool batched = (k_eval > 1 && h->method == LSR_METHOD_NDT);
  std::vector<lsr_handle> workers((size_t)k_eval, h);
  if (batched) {
    while ((int)h->aux.size() < k_eval) {
      std::unique_ptr<lsr_handle_s> a(new (std::nothrow) lsr_handle_s());
      if (!a) return LSR_ERR_HIP;
      a->method = h->method; a->device = h->device; a->stream = h->stream; a->own_stream = false;
      if (a->d_T16.reserve(16) != LSR_OK) return LSR_ERR_HIP;
      h->aux.push_back(std::move(a));
    }
    for (int e = 0; e < k_eval; e++) {
      lsr_handle a = h->aux[e].get();
      a->ndt = h->ndt; a->gicp = h->gicp;
      a->ndt_threads = h->ndt_threads; a->ndt_table_mode = h->ndt_table_mode; a->ndt_quad = h->ndt_quad; a->ndt_sort = h->ndt_sort;
      a->scratch.wait_mode = h->scratch.wait_mode; a->scratch.force_sort_path = h->scratch.force_sort_path;
      int st = a->source.resize(h->source.n);
      if (st) return st;
      if (h->source.n) {
        LSR_HIP(hipMemcpyAsync(a->source.x(), h->source.x(), sizeof(float) * h->source.n, hipMemcpyDeviceToDevice, h->stream));
        LSR_HIP(hipMemcpyAsync(a->source.y(), h->source.y(), sizeof(float) * h->source.n, hipMemcpyDeviceToDevice, h->stream));
        LSR_HIP(hipMemcpyAsync(a->source.z(), h->source.z(), sizeof(float) * h->source.n, hipMemcpyDeviceToDevice, h->stream));
      }
      a->has_source = true;
      a->source_cov_valid = false;
      workers[e] = a;
    }
  }

  std::vector<const void*> frames;
  std::vector<size_t> counts;
  std::vector<float> poses;
  std::vector<lsr_result> results((size_t)k_eval);
  std::vector<int> n_target((size_t)k_eval, 0);
  for (int e = 0; e < k_eval; e++) std::memset(&results[e], 0, sizeof(lsr_result));
  // ---- targets: window assembly (:207-222), voxelgrid_.filter + setInputTarget (:224-227)
  for (int e = 0; e < k_eval; e++) {
    lsr_handle w = workers[e];
    const int id_min = cand[e].second;
    // The reference only guards the lower end of the window; an index past the last submap would read out of bounds
    // there, so it is skipped here.
    frames.clear(); counts.clear(); poses.clear();
    for (int j = 0; j <= 2 * params->search_submap_num; j++) {
      const int idx = id_min + j - params->search_submap_num;
      if (idx < 0 || idx >= num_submaps) continue;
      double M[16];
      submap_pose_matrix(submaps[idx], M);
      frames.push_back(submaps[idx].cloud);
      counts.push_back(submaps[idx].n_points);
      for (int k = 0; k < 16; k++) poses.push_back((float)M[k]);
    }
    int st = assemble_frames(w, (int)frames.size(), frames.data(), counts.data(), stride_bytes, poses.data(), on_device != 0, w->raw);
    if (st) return st;
    auto t = fresh_target(w);
    if ((st = voxel_grid_filter(w->raw, params->voxel_leaf_size, t->cloud, w->scratch, w->stream))) return st;
    t->n = t->cloud.n;
    w->target = t;
    n_target[e] = (int)t->n;
    st = (w->method == LSR_METHOD_NDT) ? ensure_ndt_grid(w) : ensure_target_hash(w);
    if (st) { w->target.reset(); return st; }
    if (!batched) {  // serial: align(output) without guess (:230) right away — the next candidate re-uses `h`'s target slot
      lsr_loop_edge& E = edges[e];
      std::memset(&E, 0, sizeof(E));
      if (h->method == LSR_METHOD_NDT) {
        lsr_handle hs[1] = {h};
        st = align_ndt_batch(hs, 1, nullptr, E.final_transformation, &results[e]);
      } else {
        st = gicp_align(h, nullptr, E.final_transformation, &results[e]);
      }
      if (st) return st;
      if ((st = ensure_target_hash(h))) return st;
      double fitness = 0;   // getFitnessScore() (:231)
      if ((st = nn_fitness_score(h->source, h->final_T, h->target->hash, 1.7976931348623157e308, &fitness, h->scratch, h->d_T16, h->stream)))
        return st;
      E.fitness_score = fitness;
    }
  }
  if (batched) {
    std::vector<float> finals((size_t)k_eval * 16);
    // align + getFitnessScore of the k candidates (graph_based_slam_component.cpp:230-231): the searches of the candidates that
    // finish early run under the launch chain of the others
    std::vector<double> fit((size_t)k_eval);
    int st = align_ndt_batch(workers.data(), k_eval, nullptr, finals.data(), results.data(), fit.data(), 1.7976931348623157e308);
    if (st) return st;
    for (int e = 0; e < k_eval; e++) {
      lsr_handle w = workers[e];
      lsr_loop_edge& E = edges[e];
      std::memset(&E, 0, sizeof(E));
      std::memcpy(E.final_transformation, finals.data() + 16 * e, sizeof(float) * 16);
      double fitness = fit[e];
      if (fitness != fitness) {   // not served under the chain (its grid is not a refinement of a counting-sort voxel grid)
        if ((st = ensure_target_hash(w))) return st;
        if ((st = nn_fitness_score(w->source, w->final_T, w->target->hash, 1.7976931348623157e308, &fitness, w->scratch, w->d_T16, w->stream)))
          return st;
      }
      E.fitness_score = fitness;
    }
    // `h` reports the best candidate like a single registration would (getFinalTransformation / hasConverged)
    std::memcpy(h->final_T, edges[0].final_transformation, sizeof(float) * 16);
    h->converged = results[0].converged;
    // ... and it holds that candidate's window as its input target, exactly as after top_k = 1 (where the one candidate is
    // registered on `h` itself, graph_based_slam_component.cpp:227): a later getFitnessScore() / align() on `h` pairs the pose
    // just stored with the window it was registered against.  The worker takes `h`'s previous target object in exchange, so
    // both keep recycling one set of device buffers each.
    {
      lsr_handle w0 = workers[0];
      std::shared_ptr<TargetData> best = w0->target, old = h->target ? h->target : h->spare_target;
      h->target = best;
      h->spare_target = best;
      w0->target.reset();
      w0->spare_target = old;
    }
  }
  for (int e = 0; e < k_eval; e++) {
    const int id_min = cand[e].second;
    lsr_loop_edge& E = edges[e];
    E.id_from = id_min;
    E.id_to = num_submaps - 1;
    E.converged = results[e].converged;
    E.iterations = results[e].iterations;
    E.n_target_points = n_target[e];
    E.candidate_distance = cand[e].first;
    E.accepted = E.fitness_score < params->threshold_loop_closure_score ? 1 : 0;
    // relative pose of the loop edge (:236-245): from^-1 * (final * init)
    double fin[16], to[16], from[16], from_inv[16];
    for (int k = 0; k < 16; k++) fin[k] = (double)E.final_transformation[k];
    mat4_mul(fin, init_M, to);
    submap_pose_matrix(submaps[id_min], from);
    isometry_inverse(from, from_inv);
    mat4_mul(from_inv, to, E.relative_pose);
    (*n_evaluated)++;
  }
  return LSR_OK;
}

int lsr_nearest_neighbors(lsr_handle h, const float* T16, int32_t* idx, float* d2) {
  LSR_CHECK_HANDLE(h);
  if (!h->target || h->target->n == 0) return LSR_ERR_NO_TARGET;
  if (!h->has_source) return LSR_ERR_NO_SOURCE;
  int st = ensure_target_hash(h);
  if (st) return st;
  float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  return nn_search_host(h->source, T16 ? T16 : I16, h->target->hash, idx, d2, h->scratch, h->d_T16, h->stream);
}

int lsr_ndt_grid_info(lsr_handle h, int32_t* info8) {
  LSR_CHECK_HANDLE(h);
  if (!info8) return LSR_ERR_INVALID_ARGUMENT;
  if (!h->target) return LSR_ERR_NO_TARGET;
  int st = ensure_ndt_grid(h);
  if (st) return st;
  const VoxelGridDev& g = h->target->grid;
  std::vector<int> keys(g.n_leaves);
  if (g.n_leaves) LSR_HIP(hipMemcpy(keys.data(), g.leaf_key.p, sizeof(int) * g.n_leaves, hipMemcpyDeviceToHost));
  int real = 0;
  for (int k : keys) real += (k >= 0);
  for (int k = 0; k < 3; k++) { info8[k] = g.min_b[k]; info8[3 + k] = g.max_b[k]; }
  info8[6] = real;
  info8[7] = g.n_valid;
  return LSR_OK;
}

int lsr_ndt_grid_dump(lsr_handle h, int32_t* idx, int32_t* npts, double* mean, double* icov) {
  LSR_CHECK_HANDLE(h);
  if (!h->target) return LSR_ERR_NO_TARGET;
  int st = ensure_ndt_grid(h);
  if (st) return st;
  const VoxelGridDev& g = h->target->grid;
  const int L = g.n_leaves;
  if (L == 0) return LSR_OK;
  std::vector<int> keys(L), cnt(L);
  std::vector<double> m((size_t)L * 3), ic((size_t)L * 9);
  LSR_HIP(hipMemcpy(keys.data(), g.leaf_key.p, sizeof(int) * L, hipMemcpyDeviceToHost));
  LSR_HIP(hipMemcpy(cnt.data(), g.leaf_n.p, sizeof(int) * L, hipMemcpyDeviceToHost));
  LSR_HIP(hipMemcpy(m.data(), g.mean64.p, sizeof(double) * L * 3, hipMemcpyDeviceToHost));
  LSR_HIP(hipMemcpy(ic.data(), g.icov64.p, sizeof(double) * L * 9, hipMemcpyDeviceToHost));
  // leaves are already in ascending key order (radix sort); the non-finite sentinel run has key -1
  int c = 0;
  for (int r = 0; r < L; r++) {
    if (keys[r] < 0) continue;
    idx[c] = keys[r];
    npts[c] = cnt[r];
    for (int k = 0; k < 3; k++) mean[c * 3 + k] = m[(size_t)r * 3 + k];
    for (int k = 0; k < 9; k++) icov[c * 9 + k] = ic[(size_t)r * 9 + k];
    c++;
  }
  return LSR_OK;
}

int lsr_ndt_grid_centroids(lsr_handle h, float* centroid) {
  LSR_CHECK_HANDLE(h);
  if (!centroid) return LSR_ERR_INVALID_ARGUMENT;
  if (!h->target) return LSR_ERR_NO_TARGET;
  int st = ensure_ndt_grid(h);
  if (st) return st;
  TargetData& t = *h->target;
  {
    std::lock_guard<std::mutex> lock(t.build_mutex);
    if (!t.has_centroids && t.grid.ncells > 0) {
      if ((st = ndt_build_centroids(t.cloud, t.grid, h->scratch, h->stream))) return st;
      t.has_centroids = true;
    }
  }
  const VoxelGridDev& g = t.grid;
  const int L = g.n_leaves;
  if (L == 0) return LSR_OK;
  const size_t n_slots = g.dense ? g.ncells : (size_t)L;
  std::vector<int> keys(L), slot(g.ncells);
  std::vector<float> cen(n_slots * 4);
  LSR_HIP(hipMemcpy(keys.data(), g.leaf_key.p, sizeof(int) * L, hipMemcpyDeviceToHost));
  LSR_HIP(hipMemcpy(slot.data(), g.cell_slot.p, sizeof(int) * g.ncells, hipMemcpyDeviceToHost));
  LSR_HIP(hipMemcpy(cen.data(), g.centroid.p, sizeof(float) * 4 * n_slots, hipMemcpyDeviceToHost));
  int c = 0;
  for (int r = 0; r < L; r++) {
    if (keys[r] < 0) continue;
    const int s = slot[(size_t)keys[r]];
    for (int k = 0; k < 3; k++) centroid[c * 3 + k] = (s >= 0) ? cen[(size_t)s * 4 + k] : std::numeric_limits<float>::quiet_NaN();
    c++;
  }
  return LSR_OK;
}

int lsr_ndt_derivatives(lsr_handle h, const double* p6, const float* T16, int compute_hessian, double* score, double* grad,
                        double* hess) {
  LSR_CHECK_HANDLE(h);
  if (!p6) return LSR_ERR_INVALID_ARGUMENT;
  if (!h->target || h->target->n == 0) return LSR_ERR_NO_TARGET;
  if (!h->has_source) return LSR_ERR_NO_SOURCE;
  int st = ensure_ndt_grid(h);
  if (st) return st;
  if (h->target->grid.ncells == 0) { set_last_error("the input target holds no finite point"); return LSR_ERR_NO_TARGET; }
  if ((st = h->d_state.reserve(2))) return st;
  if ((st = h->h_state.reserve(2, hipHostMallocMapped))) return st;
  if ((st = h->d_prob.reserve(1))) return st;
  if ((st = h->h_prob.reserve(1, hipHostMallocMapped))) return st;
  NdtLaunchCfg cfg;
  cfg.neighborhood = h->ndt.neighborhood;
  {
    lsr_handle one[1] = {h};
    choose_table_mode(h, one, 1, cfg);
  }
  if ((st = h->d_bins.reserve((size_t)NDT_NBANKS * NDT_BANK_WORDS))) return st;
  LSR_HIP(hipMemsetAsync(h->d_bins.p, 0, sizeof(long long) * NDT_NBANKS * NDT_BANK_WORDS, h->stream));
  cfg.max_blocks = ndt_nblocks(h->source.n, h->device, 1, cfg_wg_threads(cfg), cfg_wg_points(cfg));
  ndt_fill_diag_state(h->h_state.p[0], p6, T16, compute_hessian, h->ndt, (int)h->source.n);
  h->h_state.p[1] = h->h_state.p[0];
  if (cfg.sorted && (st = ndt_sort_source(h->source, h->h_state.p[0].T, h->target->grid, h->source_sorted, h->scratch, h->stream))) return st;
  fill_problem(h->h_prob.p[0], h, h->d_state.p, h->d_bins.p, cfg);
  LSR_HIP(hipMemcpyAsync(h->d_prob.p, h->h_prob.p, sizeof(NdtProblem), hipMemcpyHostToDevice, h->stream));
  LSR_HIP(hipMemcpyAsync(h->d_state.p, h->h_state.p, 2 * sizeof(NdtState), hipMemcpyHostToDevice, h->stream));
  // launch 0 evaluates, launch 1 folds the bank into the state (PH_DIAG) -> state buffer (2 & 1) = 0
  if ((st = ndt_launch_evals(h->d_prob.p, h->h_prob.p, cfg, 0, 2, h->stream))) return st;
  LSR_HIP(hipMemcpyAsync(h->h_state.p, h->d_state.p, sizeof(NdtState), hipMemcpyDeviceToHost, h->stream));
  LSR_HIP(hipStreamSynchronize(h->stream));
  const NdtState& S = h->h_state.p[0];
  if (score) *score = S.score;
  if (grad) for (int i = 0; i < 6; i++) grad[i] = S.g[i];
  if (hess) for (int i = 0; i < 36; i++) hess[i] = compute_hessian ? S.H[i] : 0.0;
  return LSR_OK;
}

int lsr_gicp_covariances(lsr_handle h, int which, double* cov) {
  LSR_CHECK_HANDLE(h);
  if (!cov) return LSR_ERR_INVALID_ARGUMENT;
  return gicp_get_covariances(h, which, cov);
}

int lsr_get_profile(lsr_handle h, lsr_profile* out, int reset) {
  LSR_CHECK_HANDLE(h);
  if (out) *out = h->prof;
  if (reset) h->prof = lsr_profile{0, 0, 0, 0};
  return LSR_OK;
}

}  // extern "C"
