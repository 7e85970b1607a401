/*
 * lidarslam_reg.h — C ABI of the MI355X-native scan-matching registration core.
 *
 * This is the drop-in boundary for the ONE hot path of rsasaki0109/lidarslam_ros2: the
 * pcl::Registration<pcl::PointXYZI,pcl::PointXYZI> object that ScanMatcherComponent
 * (scanmatcher/include/scanmatcher/scanmatcher_component.h:93) and GraphBasedSlamComponent
 * (graph_based_slam/include/graph_based_slam/graph_based_slam_component.h:106) hold and call.
 * Every entry point below names the reference call it replaces (file:line under
 * /root/reference).  The arithmetic the reference reaches through those calls lives in the
 * un-vendored submodule Thirdparty/ndt_omp_ros2 (.gitmodules:1-4) + PCL; SURVEY.md §9 is its
 * restatement.
 *
 * Conventions
 *  - plain C, no exceptions cross this boundary; every function returns an lsr_status
 *    (0 = ok, < 0 = error) and leaves outputs untouched on error;
 *  - clouds are "strided xyz": 3 consecutive fp32 at byte offset 0 of every `stride_bytes`
 *    record (pcl::PointXYZI: stride 32; packed xyz: stride 12; xyzw: stride 16);
 *  - 4x4 transforms are COLUMN-MAJOR fp32 (Eigen::Matrix4f memory order);
 *  - a handle is thread-compatible (one caller at a time), distinct handles are re-entrant —
 *    the reference's threading contract (lidarslam/src/lidarslam.cpp:12-17);
 *  - there is NO CPU fallback: without a usable gfx950 device lsr_create fails.
 */
#ifndef LIDARSLAM_REG_H
#define LIDARSLAM_REG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lsr_handle_s* lsr_handle;

typedef enum lsr_status {
  LSR_OK = 0,
  LSR_ERR_INVALID_ARGUMENT = -1,
  LSR_ERR_NO_DEVICE = -2,       /* no gfx950 device / HIP runtime unusable */
  LSR_ERR_HIP = -3,             /* a HIP call failed; see lsr_last_error() */
  LSR_ERR_NO_TARGET = -4,       /* align/getFitnessScore before setInputTarget */
  LSR_ERR_NO_SOURCE = -5,       /* align/getFitnessScore before setInputSource */
  LSR_ERR_NOT_IMPLEMENTED = -6, /* (no entry returns it at present; until round 6 the NDT neighbourhood KDTREE did) */
  LSR_ERR_INDEX_OVERFLOW = -7,  /* voxel index space exceeds int32 (PCL: "Leaf size is too small") */
  LSR_ERR_TOO_FEW_POINTS = -8   /* GICP: cloud smaller than k_correspondences */
} lsr_status;

/* registration_method: scanmatcher_component.cpp:103-124, graph_based_slam_component.cpp:63-86 */
typedef enum lsr_method { LSR_METHOD_NDT = 0, LSR_METHOD_GICP = 1 } lsr_method;

/* pclomp::NeighborSearchMethod, selected at scanmatcher_component.cpp:110 (the reference only ever selects DIRECT7).
 * KDTREE (round 6): ndt_omp's radiusSearch(x', resolution) over the kd-tree of the leaves' float centroids, restated as the 27 cells
 * around the point's cell filtered by the kd-tree's own test (FLANN L2_Simple<float> strictly below (float)(resolution^2)) — a
 * centroid lies inside its own cell, so no leaf outside those cells can pass.  The centroids are built on first use. */
typedef enum lsr_neighborhood { LSR_KDTREE = 0, LSR_DIRECT26 = 1, LSR_DIRECT7 = 2, LSR_DIRECT1 = 3 } lsr_neighborhood;

typedef enum lsr_key {
  /* double-valued (lsr_set_f64 / lsr_get_f64) */
  LSR_RESOLUTION = 0,                 /* ndt->setResolution                 scanmatcher_component.cpp:107 */
  LSR_TRANSFORMATION_EPSILON = 1,     /* setTransformationEpsilon           scanmatcher_component.cpp:108,119 */
  LSR_STEP_SIZE = 2,                  /* NDT step_size_ (ctor default 0.1; never set by the reference) */
  LSR_OUTLIER_RATIO = 3,              /* NDT outlier_ratio_ (0.55) */
  LSR_MAX_CORRESPONDENCE_DISTANCE = 4,/* setMaxCorrespondenceDistance      scanmatcher_component.cpp:118 */
  LSR_ROTATION_EPSILON = 5,           /* GICP rotation_epsilon_ (2e-3) */
  LSR_EUCLIDEAN_FITNESS_EPSILON = 6,  /* setEuclideanFitnessEpsilon         graph_based_slam_component.cpp:80 (no effect) */
  LSR_GICP_EPSILON = 7,               /* GICP gicp_epsilon_ (1e-3) */
  /* int-valued (lsr_set_i32 / lsr_get_i32) */
  LSR_MAX_ITERATIONS = 32,            /* setMaximumIterations               graph_based_slam_component.cpp:66,77 */
  LSR_NEIGHBORHOOD = 33,              /* setNeighborhoodSearchMethod        scanmatcher_component.cpp:110 */
  LSR_NUM_THREADS = 34,               /* setNumThreads (accepted, ignored)  scanmatcher_component.cpp:111 */
  LSR_K_CORRESPONDENCES = 35,         /* GICP k_correspondences_ (20) */
  LSR_MAX_INNER_ITERATIONS = 36,      /* GICP max_inner_iterations_ (20) */
  LSR_RANSAC_ITERATIONS = 37,         /* setRANSACIterations (accepted, ignored) graph_based_slam_component.cpp:81 */
  LSR_HESSIAN_D1_SIGN = 38,           /* +1 = upstream "+sy" quirk in h_ang d1 (default), -1 = analytic */
  LSR_PROFILE = 39,                   /* 1 = bracket the derivative launch chains with hipEvents (lsr_get_profile) */
  /* tuning (NO effect on results: every NDT derivative kernel returns the same bits — the sum of a pass is defined on the
     input, csrc/ndt.hip: canon): */
  LSR_NDT_WORKGROUP = 40,             /* lane kernel: threads per workgroup (512, 1024); quad kernel: source points per
                                         workgroup (64, 128; four lanes each); 0 = automatic (512 / 128).  256 (the default of
                                         the rounds 1-3 kernel) is accepted and means automatic; as an environment preset an
                                         unknown value is reported on stderr and ignored */
  LSR_NDT_TABLE_MODE = 41,            /* where the pass reads leaf records: -1 = automatic, 0 = dense global table,
                                         1 = compact global table, 2 = whole table staged in LDS (when it fits), 3 = per
                                         workgroup the box of cells its tile-ordered points touch staged in LDS (dense tables
                                         that do not fit LDS: ndt_resolution <= 2 m on a 20-frame submap) */
  LSR_GRID_BUILDER = 42,              /* 0 = automatic (counting sort for <= 16383 grid cells, radix sort beyond), 1 = always
                                         the radix-sort builder */
  LSR_WAIT_MODE = 43,                 /* how the calling thread waits for the device inside align() / setInputTarget():
                                         0 = spin (lowest latency, pins one core per running call), 1 = sched_yield between
                                         polls, 2 = sleep 20 us between polls (a ROS2 MultiThreadedExecutor runs two
                                         registration objects side by side, lidarslam/src/lidarslam.cpp:12-17).
                                         Environment preset: LSR_WAIT_MODE=spin|yield|sleep (or 0|1|2) */
  LSR_NDT_QUAD = 44,                  /* single NDT registrations: 1 = quad kernel (four lanes per source point on every CU:
                                         lowest latency for a 30k-point scan), 0 = lane kernel (one lane per point: what candidate
                                         sets use), -1 = automatic (quad below 65 536 source points, lane from there on) */
  LSR_NDT_SORT = 45,                  /* order the source by voxel tile of its guess-moved points at the start of align():
                                         -1 = automatic (tile table mode only), 0 = never (the tile mode then falls back to the
                                         global table), 1 = also when the records are gathered from the global table */
  LSR_VOXEL_FILTER_FORM = 46,         /* read-only (lsr_get_i32): which form the last VoxelGrid filter on this object took:
                                         0 = none yet, 1 = grid dimensions worked out on the host (one wait for the bounding box,
                                         one for the leaf count), 2 = on the device (one wait: from the second
                                         lsr_set_input_source_pc2 / _frontend of an object on), 3 = the device form came back
                                         flagged (more key bits than planned, or an index overflow) and the host form ran */
  LSR_NDT_SPLIT = 47                  /* single NDT registrations on the 512-thread lane kernel: 1 = two waves per 64-point chunk (each
                                         forms one half of the per-point neighbour tree; half the serial chain per wave, twice the waves),
                                         0 = one wave per chunk, -1 = automatic (= 0: measured on BASELINE cfg 5, the split form is not
                                         faster — the pass is bound by its fixed latency chain).  Environment preset LSR_NDT_SPLIT */
} lsr_key;
/* Environment presets read when an object is created: LSR_NDT_WORKGROUP, LSR_NDT_TABLE_MODE, LSR_NDT_QUAD, LSR_GRID_BUILDER,
 * LSR_WAIT_MODE (the keys above); LSR_NDT_WIDEN=0 keeps the launches of a candidate set at their first geometry (default: widened
 * as members finish); LSR_NDT_CHAINS=1|2|3 fixes the number of independent launch chains a candidate set runs as (default: two
 * from six members on, each on a stream verified to run concurrently with the object's own; LSR_DEBUG_STREAMS=1 prints what the
 * verification found).  Diagnostic A/B switches read once per process, all with bit-identical results (LSR_VG_DEVICE_DIMS=0: the
 * VoxelGrid filter always works out its grid dimensions on the host)
 * (tests/test_gicp_gpu.py::test_search_and_chain_variants_give_identical_results): LSR_NN_COOP=0 (per-thread neighbour
 * walks instead of one wave per query), LSR_GICP_FUSED=0 (accumulate + update launch pairs instead of the fused
 * Gauss-Newton step), LSR_GICP_BALL=0 (general correspondence search on every outer iteration), LSR_GICP_CORR_FUSED=0 (seeded
 * search, general search and pair records as three launches per outer iteration instead of one), LSR_FIT_GROUP_FORM=0|1|2 (fitness
 * search of a candidate set: one wave per query / four lanes + tail / sixteen lanes seeded by the own cell + tail, the default). */

typedef struct lsr_result {
  int32_t converged;            /* hasConverged()                             scanmatcher_component.cpp:375 */
  int32_t iterations;           /* nr_iterations_ */
  double score;                 /* NDT: trans_probability_ (= score / N); GICP: final mean Mahalanobis cost */
  int32_t n_evaluations;        /* NDT: derivative passes launched; GICP: Gauss-Newton inner steps */
  int32_t n_correspondences;    /* GICP: pairs in the last outer iteration; NDT: valid (point, voxel) pairs of the last derivative pass */
  double gpu_ms;                /* HOST wall-clock time of the align call that produced this result, from the first
                                   enqueue to the result in host memory; for lsr_align_batch every member carries the
                                   time of the whole batch.  (Device-side time per derivative pass: lsr_get_profile.) */
} lsr_result;

typedef struct lsr_profile {
  double deriv_ms_total;        /* hipEvent time of the derivative-launch chains since last reset (events bracket every
                                   chunk of launches as enqueued: launch-to-launch time, dependent-launch gap included) */
  int64_t deriv_launches;       /* derivative passes that ran in those chains */
  int64_t deriv_points;         /* source points processed by those launches */
  int64_t deriv_pairs;          /* valid (point,voxel) pairs of the LAST launch (K-bar * N) */
} lsr_profile;

const char* lsr_version(void);
const char* lsr_status_string(int status);
const char* lsr_last_error(void);                       /* thread-local, human readable */
int lsr_device_count(int* count);

/* new pclomp::NormalDistributionsTransform<..>() / new pclomp::GeneralizedIterativeClosestPoint<..>()
 * scanmatcher_component.cpp:105-106,116-117; graph_based_slam_component.cpp:64-65,74-75.
 * `stream` is a hipStream_t to run on (NULL = the handle creates its own). */
int lsr_create(int method, int device_id, void* stream, lsr_handle* out);
int lsr_destroy(lsr_handle h);

int lsr_set_f64(lsr_handle h, int key, double value);
int lsr_set_i32(lsr_handle h, int key, int value);
int lsr_get_f64(lsr_handle h, int key, double* value);
int lsr_get_i32(lsr_handle h, int key, int* value);

/* registration_->setInputTarget(cloud)   scanmatcher_component.cpp:275,307,315; graph_based_slam_component.cpp:227
 * NDT: builds the voxel-covariance grid on the device.  GICP: uploads and builds the neighbour-search grid; the 20-NN
 * covariances of the target are computed by the first align() that needs them, as the reference does.
 * Host-memory (`pts` readable by the CPU) and device-memory (`pts` a HIP device pointer) forms. */
int lsr_set_input_target(lsr_handle h, const void* pts, size_t stride_bytes, size_t n);
int lsr_set_input_target_device(lsr_handle h, const void* dev_pts, size_t stride_bytes, size_t n);
/* The same for a SET of candidates — one setInputTarget per candidate submap window, graph_based_slam_component.cpp:181-227
 * (BASELINE cfg 4).  handles[b] receives clouds[b] (counts[b] records of stride_bytes; all host or all device pointers).
 * Same result per object as `count` calls of lsr_set_input_target, but the builds overlap on the device: every stage of
 * every member is enqueued (on its own object's stream) before the host waits for the first — objects that were created on
 * one shared stream get the same results without the overlap.  All members on one device.  On error no member keeps a
 * target.  An object may appear only once. */
int lsr_set_input_target_batch(lsr_handle* handles, int count, const void* const* clouds, const size_t* counts,
                               size_t stride_bytes, int on_device);
/* Ordering and lifetime of DEVICE-resident inputs (every *_device entry point and every `on_device != 0` argument): the
 * core reads them with kernels on the handle's stream (lsr_create's `stream`, or the handle's own).  (1) If another
 * stream produced the buffer, call lsr_wait_stream(h, that_stream) first — it makes the handle's stream wait for
 * everything enqueued on the producer so far (event record + hipStreamWaitEvent, no host wait); buffers produced on the
 * handle's own stream or already complete need nothing.  (2) setInputTarget-type calls return after the read has
 * completed; lsr_set_input_source_device returns with the read ENQUEUED: keep the buffer alive and unmodified until the
 * next lsr_align / lsr_align_batch / lsr_get_fitness_score on this handle has returned (or synchronise its stream).
 * (3) lsr_set_input_source_filtered / _frontend / _pc2 return when the caller's buffer (host or device) has been read and the
 * number of points kept is known; the last kernel of the filter may still be running on the handle's stream, where every later
 * call on the handle is enqueued behind it (a lsr_align right after it starts under it).  LSR_SOURCE_SYNC=1 makes them wait. */
int lsr_wait_stream(lsr_handle h, void* producer_stream);
/* Submap assembly fused with setInputTarget: frame f (strided xyz, counts[f] points) is moved by poses16[16*f ..]
 * (column-major 4x4, pcl::transformPointCloud's fp32 arithmetic) and the frames are concatenated in order — what
 * updateMap() does on the host before setInputTarget (scanmatcher_component.cpp:449-464,307) and searchLoop() for a
 * loop candidate window (graph_based_slam_component.cpp:208-227).  on_device != 0: frame pointers are HIP device
 * pointers (keyframes kept resident in HBM); the assembled target never visits the host. */
int lsr_set_input_target_frames(lsr_handle h, int n_frames, const void* const* frames, const size_t* counts, size_t stride_bytes,
                                const float* poses16, int on_device);
/* registration_->setInputSource(cloud)   scanmatcher_component.cpp:329; graph_based_slam_component.cpp:181 */
int lsr_set_input_source(lsr_handle h, const void* pts, size_t stride_bytes, size_t n);
int lsr_set_input_source_device(lsr_handle h, const void* dev_pts, size_t stride_bytes, size_t n);
/* The same for a SET of candidates (graph_based_slam_component.cpp:181 inside the candidate loop): handles[b] receives clouds[b];
 * all host or all device pointers; the uploads share launches.  Device inputs follow the lifetime rule of
 * lsr_set_input_source_device. */
int lsr_set_input_source_batch(lsr_handle* handles, int count, const void* const* clouds, const size_t* counts,
                               size_t stride_bytes, int on_device);
/* pcl::VoxelGrid<PointXYZI>::filter (centroid per occupied leaf, output ordered by leaf index) fused with
 * setInputSource: the frontend's per-scan `voxel_grid.filter(*filtered); registration_->setInputSource(filtered)`
 * (scanmatcher_component.cpp:324-329) without the filtered cloud ever leaving HBM.  on_device != 0: `pts` is a
 * HIP device pointer.  n_out (nullable) receives the number of points kept. */
int lsr_set_input_source_filtered(lsr_handle h, const void* pts, size_t stride_bytes, size_t n, float leaf, int on_device,
                                  size_t* n_out);
/* The frontend's whole per-scan preprocessing on the device: min-max range filter
 * (`scan_min_range < sqrt(x^2+y^2) < scan_max_range`, scanmatcher_component.cpp:210-218) -> VoxelGrid
 * (vg_size_for_input, :324-328) -> setInputSource (:329).  A sensor_msgs/PointCloud2 payload with x@0,y@4,z@8
 * (point_step = stride_bytes, e.g. 32 for the PointXYZI layout pcl::toROSMsg writes) can be passed as is; other field
 * offsets and the intensity field: lsr_set_input_source_pc2. */
int lsr_set_input_source_frontend(lsr_handle h, const void* pts, size_t stride_bytes, size_t n, double scan_min_range,
                                  double scan_max_range, float vg_size_for_input, int on_device, size_t* n_out);
/* ---- sensor_msgs/PointCloud2 codec ---------------------------------------------
 * A PointCloud2 `data` buffer is `width*height` records of `point_step` bytes with FLOAT32 fields at the byte offsets its
 * `fields` array names; pcl::fromROSMsg (scanmatcher_component.cpp:201-202) reads them wherever they are and
 * pcl::toROSMsg (:279,284; lidarslam_msgs/msg/SubMap.msg:4) writes pcl::PointXYZI's layout {x@0, y@4, z@8, intensity@16,
 * point_step 32}.  The binding passes the message's own offsets; offset_intensity < 0 = the message has none. */
typedef struct lsr_pc2_layout {
  uint32_t point_step;
  uint32_t offset_x, offset_y, offset_z;
  int32_t offset_intensity;
} lsr_pc2_layout;
/* fromROSMsg -> min-max range filter (:210-218) -> VoxelGrid(vg_size_for_input) (:324-328) -> setInputSource (:329) from the
 * raw message payload, on the device.  Intensity is carried the way pcl::VoxelGrid does with its default
 * downsample_all_data: the leaf mean (float accumulation, points in ascending index). */
int lsr_set_input_source_pc2(lsr_handle h, const void* data, size_t n_points, const lsr_pc2_layout* layout, double scan_min_range,
                             double scan_max_range, float vg_size_for_input, int on_device, size_t* n_out);
/* toROSMsg of the handle's current input source (e.g. the filtered scan, to publish it or to store it in a SubMap):
 * n records of layout->point_step bytes, x / y / z / intensity at the layout's offsets, every other byte zero. */
int lsr_get_source_pc2(lsr_handle h, void* out_data, size_t capacity_points, const lsr_pc2_layout* layout, size_t* n_out);
/* The same into a DEVICE buffer (d_out: capacity_points records of layout->point_step bytes in HBM): the filtered scan stays resident,
 * e.g. as the newest submap of the frontend's map window — updateMap()'s VoxelGrid(vg_size_for_map) + toROSMsg
 * (scanmatcher_component.cpp:442-446, 466-470) without the 3.5 MB round trip over PCIe; lsr_set_input_target_frames(on_device = 1)
 * takes such buffers.  Returns after the records are complete (any stream may read them). */
int lsr_get_source_pc2_device(lsr_handle h, void* d_out, size_t capacity_points, const lsr_pc2_layout* layout, size_t* n_out);
/* pcl::VoxelGrid::filter, message payload in, message payload out (map side: :266-269, 443-447;
 * graph_based_slam_component.cpp:224-226), intensity averaged per leaf like the coordinates. */
int lsr_voxel_grid_filter_pc2(lsr_handle h, const void* data, size_t n_points, const lsr_pc2_layout* in_layout, float leaf, void* out_data,
                              size_t capacity_points, const lsr_pc2_layout* out_layout, size_t* n_out);
/* The same filter as a stand-alone operation, host in / host out (map side: scanmatcher_component.cpp:266-269,
 * 443-447; graph_based_slam_component.cpp:224-226).  Writes xyz at offset 0 of each out_stride_bytes record. */
int lsr_voxel_grid_filter(lsr_handle h, const void* pts, size_t stride_bytes, size_t n, float leaf, void* out_pts,
                          size_t out_stride_bytes, size_t out_capacity, size_t* n_out);
/* Let `h` register against the target already resident in `owner` (N keyframes vs ONE submap):
 * no copy, the voxel grid / target structures are reference counted. */
int lsr_share_target(lsr_handle h, lsr_handle owner);

/* registration_->align(output, guess)    scanmatcher_component.cpp:353; graph_based_slam_component.cpp:230
 * guess: col-major 4x4 or NULL (= identity, the backend's align(output)).  final_transformation: out, 16 floats.
 * output_pts (nullable): host buffer of n_source records of out_stride_bytes.  PCL's align() copies the source into
 * `output` and then overwrites x, y, z with the transformed coordinates: here ONLY the 12 xyz bytes at offset 0 of every
 * record are written, the rest of each record is left exactly as the caller passed it (pre-fill it with the source
 * records to get PCL's result, other fields included).  Both reference callers discard `output`
 * (scanmatcher_component.cpp:350-353, graph_based_slam_component.cpp:229-230) — pass NULL to skip it. */
int lsr_align(lsr_handle h, const float* guess, float* final_transformation, lsr_result* result,
              void* output_pts, size_t out_stride_bytes);
/* B independent registrations advanced together (loop-closure candidate set / N keyframes vs submap; BASELINE.json cfg 4).
 * NDT: ONE launch chain whose grid covers all B registrations.  GICP (the stand-alone backend's configuration,
 * graph_based_slam/param/graphbasedslam.yaml:3): B launch chains side by side, each on its own object's stream, fed by one
 * host loop — objects created on one shared stream get the same results without the overlap.  All handles must live on the
 * same device and use the same method; an object may appear only once.  guesses: B*16 floats or NULL; finals: B*16 floats;
 * results: B entries. */
int lsr_align_batch(lsr_handle* handles, int batch, const float* guesses, float* finals, lsr_result* results);

/* registration_->getFinalTransformation()  scanmatcher_component.cpp:356; graph_based_slam_component.cpp:244,253 */
int lsr_get_final_transformation(lsr_handle h, float* out16);
/* registration_->hasConverged()            scanmatcher_component.cpp:375 */
int lsr_has_converged(lsr_handle h, int* out);
/* registration_->getFitnessScore(max_range = DBL_MAX)  graph_based_slam_component.cpp:231; scanmatcher_component.cpp:376 */
int lsr_get_fitness_score(lsr_handle h, double max_range, double* out);
/* align() followed by getFitnessScore(max_range) for every candidate of a set in ONE call (graph_based_slam_component.cpp:230-231
 * inside the candidate loop): finals / results as lsr_align_batch, fitness[b] as lsr_get_fitness_score_batch — with the fitness
 * search of every candidate that finishes early running under the launch chain of the ones still registering. */
int lsr_align_fitness_batch(lsr_handle* handles, int count, const float* guesses, float* finals, lsr_result* results, double max_range,
                            double* fitness);

/* registration_->getFitnessScore() of every candidate of a set (graph_based_slam_component.cpp:231 inside the candidate loop):
 * out[b] = what lsr_get_fitness_score(handles[b], max_range, ..) returns; all searches are enqueued before the first wait. */
int lsr_get_fitness_score_batch(lsr_handle* handles, int count, double max_range, double* out);

/* ---- multi-GPU sharding of a batch (BASELINE.json cfg 4) -----------------------
 * One process per GPU.  The batch of `global_count` independent registrations — the loop-closure candidate set of
 * graph_based_slam_component.cpp:188-231 generalised from the nearest candidate to all of them, or N keyframes against
 * a submap — is split by the static block partition lsr_shard_range(); every rank registers its own share with
 * lsr_align_batch (no collective on the data path) and ONE ncclAllGather of fixed 64-byte records over xGMI gives every
 * rank the whole table.  RCCL is loaded on first use; a one-rank communicator needs no RCCL. */
typedef struct lsr_comm_s* lsr_comm;
typedef struct lsr_shard_record {   /* 64 bytes */
  float T[12];                      /* final transformation, ROW-major 3x4 */
  float score;                      /* lsr_result.score */
  float iterations;
  float converged;                  /* 1 / 0 */
  float fitness;                    /* getFitnessScore() when requested, else NaN */
} lsr_shard_record;
/* first index and count of rank's share: the first (n_items % world) ranks get one extra item */
void lsr_shard_range(int n_items, int world, int rank, int* first, int* count);
/* Cost-aware plan for batches whose members differ in size (the ring gate's candidates: targets from a few thousand to
 * 661 k points): longest-processing-time-first — items by cost descending (ties: lower index), each to the least-loaded
 * rank (ties: lower rank).  cost may be NULL; without costs, or with costs the model cannot tell apart (spread within 2 % of the
 * largest), the plan is the block partition of lsr_shard_range.  owner[i] = rank of item i; order = the batch
 * regrouped rank by rank, each rank's items longest first; rank r owns order[rank_first[r] .. rank_first[r+1]).
 * Device-free and deterministic: every rank computes the same plan from the same costs. */
int lsr_shard_plan(int n_items, const double* cost, int world, int32_t* owner /* n_items */, int32_t* order /* n_items */,
                   int32_t* rank_first /* world + 1 */);
/* rank 0: 128-byte ncclUniqueId to hand to the other ranks (by whatever channel the application has) */
int lsr_comm_unique_id(void* id128);
/* every rank: ncclCommInitRank on device_id (id128 may be NULL when world == 1) */
int lsr_comm_create(const void* id128, int rank, int world, int device_id, lsr_comm* out);
int lsr_comm_destroy(lsr_comm c);
/* "N keyframes vs. one submap" across ranks: the rank `root` holds the target cloud — the submap
 * scanmatcher_component.cpp:449-464 assembles and :307 hands to registration_->setInputTarget — and every rank of the communicator
 * ends up with it as the input target of its handle `h`: a 16-byte header {points, stride} by ncclBroadcast, one ncclAllGather of a
 * ready word per rank (the records travel only if every rank can receive them), one ncclBroadcast of the records (device to device
 * over xGMI), then the same voxel grid built on every rank.  pts / stride_bytes / n / on_device are read on the root only.
 * COLLECTIVE: every rank calls it, and a rank that fails locally (bad cloud on the root, no memory, a handle on another device than
 * the communicator) still takes part in every exchange and returns its error afterwards; the other ranks return an error too
 * (LSR_ERR_NO_TARGET when the root had nothing to send) — no rank waits for one that has left.
 * ORDERING of a device-resident root cloud (on_device != 0): the records are read on the COMMUNICATOR's stream, which this call
 * orders behind the handle's stream; the caller orders the handle's stream behind the producer of the records exactly as for
 * lsr_set_input_target_device — lsr_wait_stream(h, producer_stream) or a synchronisation of its own — and keeps the buffer valid
 * until the call returns.  A one-rank communicator hands the cloud straight to lsr_set_input_target(_device).
 * Reference call it generalises: registration_->setInputTarget(targeted_cloud_ptr), one node, one object. */
int lsr_set_input_target_bcast(lsr_comm c, lsr_handle h, const void* pts, size_t stride_bytes, size_t n, int on_device, int root);
/* The pose all-gather on its own (BASELINE.json north_star: "RCCL all-gather of the 6-DoF poses"): every rank contributes `count`
 * 64-byte records — e.g. the scans of its stream, registered one after the other with lsr_align as the frontend does
 * (scanmatcher_component.cpp:353-356) — and receives all world x count of them, rank-major.  COLLECTIVE, same count on every rank. */
int lsr_comm_all_gather_records(lsr_comm c, const lsr_shard_record* local, int count, lsr_shard_record* all_records /* world x count */);
/* local_handles / local_guesses: this rank's share (local_count = lsr_shard_range count), targets and sources already
 * set; with_fitness != 0 adds getFitnessScore() per registration (graph_based_slam_component.cpp:231).
 * all_records: global_count entries, in batch order, identical on every rank. */
int lsr_align_batch_sharded(lsr_comm c, lsr_handle* local_handles, int local_count, int global_count, const float* local_guesses,
                            int with_fitness, lsr_shard_record* all_records);

/* The same with a plan from lsr_shard_plan: local_handles are order[rank_first[rank] ..] in that order (longest first).
 * all_records stays in BATCH order (record of item i at all_records[i]) on every rank. */
int lsr_align_batch_planned(lsr_comm c, lsr_handle* local_handles, int local_count, int global_count, const int32_t* order,
                            const int32_t* rank_first, const float* local_guesses, int with_fitness, lsr_shard_record* all_records);

/* ---- loop-closure gate -------------------------------------------------- */
/* One lidarslam_msgs/msg/SubMap (SubMap.msg:1-4): accumulated travel distance, geometry_msgs/Pose, and the
 * PointCloud2 payload (xyz fp32 at offset 0 of every record of stride_bytes; pose-local coordinates). */
typedef struct lsr_submap {
  double position[3];           /* pose.position x,y,z */
  double orientation[4];        /* pose.orientation x,y,z,w */
  double distance;              /* SubMap.distance */
  const void* cloud;            /* host or HIP device pointer (see on_device) */
  size_t n_points;
} lsr_submap;
/* graph_based_slam parameters used by searchLoop() (graph_based_slam_component.cpp:23-38, defaults in brackets) */
typedef struct lsr_loop_params {
  double threshold_loop_closure_score;     /* [1.0]  accept when fitness < threshold        (:233) */
  double distance_loop_closure;            /* [20.0] minimum travel since the candidate      (:195) */
  double range_of_searching_loop_closure;  /* [20.0] maximum Euclidean distance to candidate (:196) */
  int search_submap_num;                   /* [3]    half-width of the target window         (:209) */
  float voxel_leaf_size;                   /* [0.2]  voxelgrid_ leaf of the assembled target (:61,224-226) */
  int top_k;                               /* 1 = the reference (nearest candidate only); >1 = k nearest candidates */
  int reserved;
} lsr_loop_params;
typedef struct lsr_loop_edge {
  int id_from;                  /* LoopEdge.pair_id.first  = candidate submap index (:240) */
  int id_to;                    /* LoopEdge.pair_id.second = num_submaps - 1 */
  int accepted;                 /* fitness_score < threshold_loop_closure_score */
  int converged;
  int iterations;
  int n_target_points;          /* size of the filtered target window */
  double candidate_distance;    /* |latest - candidate| position distance */
  double fitness_score;         /* getFitnessScore() after align (:231) */
  double relative_pose[16];     /* from^-1 * (final * init), column-major 4x4 fp64 (:241-245) */
  float final_transformation[16];
} lsr_loop_edge;
/* GraphBasedSlamComponent::searchLoop() from `latest_submap` on (graph_based_slam_component.cpp:164-252): the latest
 * submap moved by its pose becomes the source (:171-181), candidates are gated on travelled distance and range and the
 * nearest one is picked (:188-205), its window of 2*search_submap_num+1 submaps is moved, concatenated, voxel filtered and
 * set as target (:207-227), then align() without guess (:230), getFitnessScore() (:231), the threshold (:233) and the
 * relative pose of the edge (:236-245) — clouds stay in HBM from the first transform to the fitness sum.
 * edges: up to edge_capacity entries, nearest candidate first; *n_evaluated = candidates registered (0 = no candidate).
 * Afterwards `h` answers getFinalTransformation / hasConverged for the nearest candidate and holds that candidate's window
 * as its input target — the reference's state after the loop (:227) — whatever top_k was (with top_k > 1 the k windows are
 * built on worker objects and the nearest one's is handed to `h`), so a later getFitnessScore() on `h` scores exactly the
 * pose it reports.
 * The pose-graph optimisation that follows (doPoseAdjustment, g2o) is the caller's. */
int lsr_search_loop(lsr_handle h, const lsr_submap* submaps, int num_submaps, size_t stride_bytes, int on_device,
                    const lsr_loop_params* params, lsr_loop_edge* edges, int edge_capacity, int* n_evaluated);

/* ---- inspection (parity tests / profiling; not used by the ROS nodes) ------------------- */
/* NDT voxel grid: info[0..2]=min_b, [3..5]=max_b, [6]=#leaves (any point count), [7]=#leaves usable (n>=6, valid cov) */
int lsr_ndt_grid_info(lsr_handle h, int32_t* info8);
/* Dump all leaves sorted by linear index: idx[L], npts[L] (-1 = invalidated), mean[L*3], icov[L*9] (row-major). */
int lsr_ndt_grid_dump(lsr_handle h, int32_t* idx, int32_t* npts, double* mean, double* icov);
/* The FLOAT centroids of the leaves (pclomp::VoxelGridCovariance::Leaf::centroid: the float running sum of a leaf's points in cloud
 * order over (float) count) — the points of the voxel-centroid kd-tree that the KDTREE neighbourhood searches —, in the order of
 * lsr_ndt_grid_dump: centroid[L*3]; NaN for leaves that are not in the kd-tree (fewer than 6 points, invalid covariance). */
int lsr_ndt_grid_centroids(lsr_handle h, float* centroid);
/* One derivative pass at pose p = (tx,ty,tz,rx,ry,rz); T16 (nullable, col-major) overrides the point
 * transform like the first pass of align().  grad: 6, hess: 36 (row-major). */
int lsr_ndt_derivatives(lsr_handle h, const double* p6, const float* T16, int compute_hessian, double* score,
                        double* grad, double* hess);
/* GICP per-point covariances after setInput*: which = 0 source, 1 target (regularised, as the optimiser uses them);
 * 2 source, 3 target: the k-neighbour sample covariance BEFORE the eigen-regularisation.  cov: n*9 doubles. */
int lsr_gicp_covariances(lsr_handle h, int which, double* cov);
/* 1-NN of the source transformed by T16 (nullable = identity) in the target: idx[n], d2[n]. */
int lsr_nearest_neighbors(lsr_handle h, const float* T16, int32_t* idx, float* d2);
int lsr_get_profile(lsr_handle h, lsr_profile* out, int reset);
/* Host-only self check (needs no device): the angular coefficient tables of NDT eq. 6.19 / 6.21 at pose p6 by the scalar
 * formulas (jang_ref[24], hang_ref[48]) and by the 72-entry table the device evaluates one entry per lane (jang_tab, hang_tab). */
int lsr_debug_angle_tables(const double* p6, int d1_sign, float* jang_ref, float* hang_ref, float* jang_tab, float* hang_tab);

#ifdef __cplusplus
}
#endif
#endif /* LIDARSLAM_REG_H */
