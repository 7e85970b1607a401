// TEST INFRASTRUCTURE: every code block of INTEGRATION.md that calls the C ABI, compiled against include/lidarslam_reg.h
// (and the pcl::Registration binding against tests/cpp/mock/pcl) by tests/test_host_cpu.py — the blocks between
// "[snippet: NAME]" and "[end snippet]" must appear verbatim in INTEGRATION.md, so the document cannot drift from the header.
// The mock ROS message types below carry only the members the snippets read.
#include <lidarslam_reg/gfx950_registration.hpp>

#include <array>
#include <cstdint>
#include <memory>
#include <vector>

namespace mock {
struct Point { double x, y, z; };
struct Quat { double x, y, z, w; };
struct Pose { Point position; Quat orientation; };
struct PointCloud2 { uint32_t width = 0, height = 1, point_step = 32; std::vector<uint8_t> data; };
struct SubMap { double distance = 0; Pose pose; PointCloud2 cloud; };
struct MapArray { std::vector<SubMap> submaps; };
struct LoopEdge { std::pair<int, int> pair_id; Eigen::Isometry3d relative_pose; };
struct Logger {};
}  // namespace mock
#define RCLCPP_ERROR(logger, fmt, ...) std::fprintf(stderr, fmt "\n", __VA_ARGS__)

using Reg = Gfx950Registration<pcl::PointXYZI, pcl::PointXYZI>;

// ---- §3: the construction sites ---------------------------------------------------------------------------------
std::shared_ptr<pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>> construct_frontend(double ndt_resolution, int ndt_num_threads,
                                                                                      std::shared_ptr<Reg>& gfx950_) {
  std::shared_ptr<pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>> registration_;
  // [snippet: construction]
  auto ndt = std::make_shared<Gfx950Registration<pcl::PointXYZI, pcl::PointXYZI>>(LSR_METHOD_NDT);
  ndt->setResolution(ndt_resolution); ndt->setTransformationEpsilon(0.01);
  ndt->setNeighborhoodSearchMethod(LSR_DIRECT7); if (ndt_num_threads > 0) ndt->setNumThreads(ndt_num_threads);
  ndt->setMaterializeOutput(false);     // the node discards align()'s output cloud (:350-353)
  registration_ = ndt; gfx950_ = ndt;   // keep the derived pointer: getFitnessScore is not virtual in PCL
  // [end snippet]
  return registration_;
}

double fitness_site(const std::shared_ptr<Reg>& gfx950_) {
  // [snippet: fitness]
  double fitness_score = gfx950_->getFitnessScore();   // before: registration_->getFitnessScore() (host FLANN search)
  // [end snippet]
  return fitness_score;
}

// ---- §3b: searchLoop() in one call ----------------------------------------------------------------------------------
struct Backend {
  std::shared_ptr<Reg> gfx950_;
  double threshold_loop_closure_score_ = 1.0, distance_loop_closure_ = 20.0, range_of_searching_loop_closure_ = 20.0, voxel_leaf_size_ = 0.2;
  int search_submap_num_ = 3;
  bool use_save_map_in_loop_ = true;
  std::vector<mock::LoopEdge> loop_edges_;
  mock::Logger get_logger() { return {}; }
  void doPoseAdjustment(const mock::MapArray&, bool) {}
  void searchLoop(const mock::MapArray& map_array_msg) {
    using mock::LoopEdge;
    const int num_submaps = (int)map_array_msg.submaps.size();
    // [snippet: search_loop]
    std::vector<lsr_submap> sm(num_submaps);
    for (int i = 0; i < num_submaps; i++) {
      const auto & m = map_array_msg.submaps[i];
      sm[i] = {{m.pose.position.x, m.pose.position.y, m.pose.position.z},
               {m.pose.orientation.x, m.pose.orientation.y, m.pose.orientation.z, m.pose.orientation.w},
               m.distance, m.cloud.data.data(), (size_t)m.cloud.width * m.cloud.height};
    }
    lsr_loop_params lp = {threshold_loop_closure_score_, distance_loop_closure_, range_of_searching_loop_closure_,
                          search_submap_num_, (float)voxel_leaf_size_, /*top_k=*/1, 0};
    lsr_loop_edge e; int n = 0;
    if (lsr_search_loop(gfx950_->handle(), sm.data(), num_submaps, map_array_msg.submaps[0].cloud.point_step,
                        /*on_device=*/0, &lp, &e, 1, &n) != LSR_OK) { RCLCPP_ERROR(get_logger(), "%s", lsr_last_error()); return; }
    if (n == 1 && e.accepted) {
      LoopEdge loop_edge;
      loop_edge.pair_id = {e.id_from, e.id_to};
      loop_edge.relative_pose = Eigen::Isometry3d(Eigen::Map<const Eigen::Matrix4d>(e.relative_pose));
      loop_edges_.push_back(loop_edge);
      doPoseAdjustment(map_array_msg, use_save_map_in_loop_);   // g2o, unchanged
    }
    // [end snippet]
  }
};

// ---- §3c: PointCloud2 payloads ----------------------------------------------------------------------------------------
void pc2_sites(lsr_handle h, const mock::PointCloud2* msg, uint32_t off_x, uint32_t off_y, uint32_t off_z, int32_t off_intensity,
               double scan_min_range_, double scan_max_range_, float vg_size_for_input_, const void* in, size_t n_in, float leaf, void* out,
               size_t capacity, void* d_keyframe) {
  // [snippet: pc2]
  lsr_pc2_layout L = {msg->point_step, off_x, off_y, off_z, off_intensity /* -1: none */};   // from msg->fields
  // range filter [min,max] + VoxelGrid(leaf) + setInputSource in one call, cloud stays in HBM (replaces :201-218,324-329)
  size_t n_kept;
  lsr_set_input_source_pc2(h, msg->data.data(), (size_t)msg->width * msg->height, &L, scan_min_range_, scan_max_range_,
                           vg_size_for_input_, /*on_device=*/0, &n_kept);
  // VoxelGrid of any payload -> payload (replaces pcl::VoxelGrid + toROSMsg, :279,284); intensity carried like PCL
  size_t n_out; lsr_voxel_grid_filter_pc2(h, in, n_in, &L, leaf, out, capacity, &L, &n_out);
  // the current (filtered) source as a PointCloud2 payload (toROSMsg direction)
  lsr_get_source_pc2(h, out, capacity, &L, &n_out);
  // ... or into a DEVICE buffer: a keyframe that never leaves HBM (what lsr_set_input_target_frames takes with on_device = 1)
  lsr_get_source_pc2_device(h, d_keyframe, capacity, &L, &n_out);
  // [end snippet]
}

// ---- §3d: a candidate set over the GPUs of a node ---------------------------------------------------------------------
void sharded_site(int rank, int world, int device, lsr_handle* handles, const void* const* target_ptrs, const size_t* target_counts,
                  const void* const* source_ptrs, const size_t* source_counts, const float* guesses) {
  // [snippet: sharded]
  // one process (or thread with its own device) per GPU; rank 0 creates the id and hands it to the others
  char id[128]; if (rank == 0) lsr_comm_unique_id(id);  /* broadcast `id` by whatever the application has */
  lsr_comm comm; lsr_comm_create(id, rank, world, device, &comm);          // RCCL is dlopen'ed here, not before
  int first, mine; lsr_shard_range(64, world, rank, &first, &mine);        // this rank's block of the 64 candidates
  // the k-th local object gets candidate first+k: all targets in one staged call (the grid builds overlap on the device)
  lsr_set_input_target_batch(handles, mine, target_ptrs, target_counts, sizeof(pcl::PointXYZI), /*on_device=*/0);
  lsr_set_input_source_batch(handles, mine, source_ptrs, source_counts, sizeof(pcl::PointXYZI), /*on_device=*/0);
  // one shared launch chain + lsr_get_fitness_score_batch inside, then ONE
  // ncclAllGather of 64-byte records: every rank gets all 64 results in candidate order
  std::vector<lsr_shard_record> all(64);
  lsr_align_batch_sharded(comm, handles, mine, 64, guesses, /*with_fitness=*/1, all.data());
  lsr_comm_destroy(comm);
  // [end snippet]
}

void planned_site(lsr_comm comm, int rank, int world, int n_cand, lsr_handle* handles, const size_t* target_counts_all,
                  const size_t* source_counts_all, const float* guesses) {
  // [snippet: planned]
  // candidates of different size (the ring gate: targets from a few thousand to 661 k points): a cost-aware plan instead of
  // blocks — every rank computes the same plan from the same costs; rank r owns order[rank_first[r] .. rank_first[r+1]),
  // longest first, and hands its objects over in that order
  std::vector<double> cost(n_cand);
  for (int i = 0; i < n_cand; i++) cost[i] = 6.0 * target_counts_all[i] + 34.0 * source_counts_all[i];   // point visits
  std::vector<int32_t> owner(n_cand), order(n_cand), rank_first(world + 1);
  lsr_shard_plan(n_cand, cost.data(), world, owner.data(), order.data(), rank_first.data());
  const int mine = rank_first[rank + 1] - rank_first[rank];   // local object k registers candidate order[rank_first[rank] + k]
  std::vector<lsr_shard_record> all(n_cand);                  // comes back in candidate order on every rank
  lsr_align_batch_planned(comm, handles, mine, n_cand, order.data(), rank_first.data(), guesses, /*with_fitness=*/1, all.data());
  // [end snippet]
}

int main() { return 0; }
