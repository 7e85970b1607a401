// The opaque handle behind the C ABI (one registration object = one pcl::Registration instance).
#pragma once
#include "common.hpp"
#include "gicp.hpp"
#include "ndt.hpp"
#include "nn.hpp"

namespace lsr {
// A deferred stream dependency (round 6): the group launches of a candidate set run on the FIRST member's stream; instead of making
// every other member's own stream wait for them right away (one hipStreamWaitEvent per member and call, and one query + record + wait
// per member at the next group call: ~110 us of host time per share of eight), the members only remember the event.  A member's own
// stream waits for it when the member is next used on its own (LSR_CHECK_HANDLE); a group call on the SAME lead stream needs nothing.
struct StreamDep {
  hipEvent_t ev = nullptr;
  ~StreamDep() { if (ev) (void)hipEventDestroy(ev); }
};
}  // namespace lsr

struct lsr_handle_s {
  using NdtParamsHost = lsr::NdtParamsHost; using GicpParamsHost = lsr::GicpParamsHost; using TargetData = lsr::TargetData;
  using DeviceCloud = lsr::DeviceCloud; using HashGridDev = lsr::HashGridDev; using BuildScratch = lsr::BuildScratch;
  using NdtState = lsr::NdtState; using NdtProblem = lsr::NdtProblem; using GicpWorkspace = lsr::GicpWorkspace;
  template <typename T> using DevBuf = lsr::DevBuf<T>;
  template <typename T> using PinBuf = lsr::PinBuf<T>;
  int method = LSR_METHOD_NDT;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // profiling brackets (LSR_PROFILE)
  std::shared_ptr<lsr::StreamDep> dep;        // pending: `stream` must wait for dep->ev before this object's next launch of its own
  hipStream_t dep_stream = nullptr;           // ... the (lead) stream that event was recorded on
  std::shared_ptr<lsr::StreamDep> lead_dep;   // as the lead of a set: the event its members are handed
  // side stream of a batch lead: the neighbour grids of a candidate set are refined there while the shared NDT launch chain
  // runs on `stream` (align_ndt_batch); created on first use
  hipStream_t side_stream = nullptr;
  hipEvent_t side_ev = nullptr, side_fork_ev = nullptr;
  // extra launch-chain streams of a batch lead: a small candidate set runs as several independent chains (run_ndt_feeder)
  hipStream_t chain_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t chain_ev[3] = {nullptr, nullptr, nullptr};
  hipEvent_t chain_fork_ev = nullptr;
  bool chain_probed = false;   // the streams above were looked for (each is verified to run concurrently with `stream`)
  int n_chain_streams = 0;

  NdtParamsHost ndt;
  GicpParamsHost gicp;
  double euclidean_fitness_eps = -1.7976931348623157e308;
  int num_threads = 0, ransac_iterations = 0;
  int profile = 0;
  int ndt_threads = 0;       // LSR_NDT_WORKGROUP: 0 = automatic; quad kernel: 64 / 128 points, lane kernel: 512 / 1024 threads
  int ndt_table_mode = -1;   // LSR_NDT_TABLE_MODE: -1 = automatic, else lsr::NdtTableMode
  int ndt_quad = -1;         // LSR_NDT_QUAD: -1 = automatic (single registrations: four lanes per point), 0 = lane kernel, 1 = four
  int ndt_split = -1;        // LSR_NDT_SPLIT: -1 = automatic, 0 / 1 = one / two waves per chunk in the 512-thread lane kernel (single registrations)
  int ndt_sort = -1;         // LSR_NDT_SORT: -1 = automatic (tile mode only), 0 = never, 1 = also for global-table gathers

  std::shared_ptr<TargetData> target;
  std::shared_ptr<TargetData> spare_target;  // recycled by the next setInputTarget when no other handle shares it
  DeviceCloud source;
  DeviceCloud source_sorted;  // NDT_TAB_TILE: the source ordered by voxel tile of its guess-moved points (rebuilt by every align)
  DeviceCloud raw, filtered;  // N1: unfiltered upload / stand-alone filter result
  bool has_source = false;
  bool source_cov_valid = false;
  DevBuf<double> source_cov;  // GICP
  HashGridDev source_hash;    // GICP (20-NN over the source itself)

  BuildScratch scratch;
  DevBuf<unsigned char> staging;
  PinBuf<float> out_xyz;      // align(output): packed transformed xyz on their way into the caller's records

  // NDT run-time buffers (batch-capable: the leader of a batch owns arrays for all members)
  DevBuf<NdtState> d_state;
  DevBuf<long long> d_bins;   // NDT_NBANKS banks of int64 accumulators per registration (exact chunk sums, ndt.hip: canon)
  DevBuf<NdtProblem> d_prob;
  PinBuf<NdtState> h_state;
  PinBuf<NdtProblem> h_prob;
  const NdtState* h_state_dev = nullptr;   // device views of the two pinned arrays (the chain's init launch reads them, ndt_init_batch)
  const NdtProblem* h_prob_dev = nullptr;
  PinBuf<lsr::NdtMailbox> mailbox;        // host-coherent, mapped: progress + result of a single registration
  lsr::NdtMailbox* d_mailbox = nullptr;   // device view of mailbox.p
  unsigned int align_token = 0;
  DevBuf<float> d_T16;
  DevBuf<float> d_poses;  // N2: keyframe poses

  GicpWorkspace gicp_ws;

  // worker objects of lsr_search_loop(top_k > 1): one per candidate registered in the same launch chain
  std::vector<std::unique_ptr<lsr_handle_s>> aux;

  float final_T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  int converged = 0;
  lsr_profile prof = {0, 0, 0, 0};
};

