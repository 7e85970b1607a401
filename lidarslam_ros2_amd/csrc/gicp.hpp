// GICP on gfx950 (replaces pclomp::GeneralizedIterativeClosestPoint).
#pragma once
#include "common.hpp"

struct lsr_handle_s;

namespace lsr {
constexpr int GICP_MAX_K = 32;

struct GicpParamsHost {
  double max_corr_dist = 5.0;     // corr_dist_threshold_
  double trans_eps = 5e-4;        // transformation_epsilon_
  double rot_eps = 2e-3;          // rotation_epsilon_
  double gicp_eps = 1e-3;         // gicp_epsilon_
  int max_iterations = 200;
  int max_inner = 20;
  int k = 20;
};

// Host mailbox of a GICP align (pinned, host-coherent, mapped into the device): the launch chain reports how many of
// the enqueued update launches have run and, when the OUTER loop ends, the result — no device-to-host copy, no stream
// synchronisation and no host bookkeeping per outer iteration (the host only keeps launches queued).
struct GicpMailbox {
  unsigned long long progress;   // (token << 32) | update launches of this align that have run
  unsigned int done;             // = token once the align has ended; written last (release, system scope)
  int converged, nr_iterations, last_cnt, gn_steps, pad;
  double last_cost;
  float final_T[16];             // column-major: previous_transformation_ * guess
};

struct GicpWorkspace {
  DeviceCloud out;               // guess * source ("output" of the reference's align)
  DevBuf<unsigned char> pairs;   // PairRec[n]
  DevBuf<double> buf;            // per-workgroup partial rows + Rm
  DevBuf<unsigned char> state;   // per-iteration block {GnState, T16, Rm} + counters
  PinBuf<unsigned char> pin;     // pinned host mirror of the per-iteration block
  DevBuf<int> work;              // K5: count + indices of the points deferred to the wave-cooperative search
  DevBuf<double> raw_cov;        // inspection only: sample covariances before regularisation
  DevBuf<int> last_nn;           // K6: each source point's neighbour in the previous outer iteration (search seed)
  DevBuf<float> nn_d2;           // K6: its squared distance (search kernel -> pair kernel)
  DevBuf<int> corr_work;         // K6: [0] = count, [1..] = points the seeded search hands to the general one
  DevBuf<int> count_shards;      // K6: pair counters of the one-launch correspondence pass, a cache line apart
  PinBuf<GicpMailbox> mailbox;
  GicpMailbox* d_mailbox = nullptr;
  unsigned int token = 0;        // one per align
};

int gicp_align(lsr_handle_s* h, const float* guess, float* final_T, lsr_result* res);
// B registrations side by side (each chain on its own object's stream, one host loop feeding them all)
int gicp_align_batch(lsr_handle_s* const* hs, int B, const float* guesses, float* finals, lsr_result* results);
int gicp_get_covariances(lsr_handle_s* h, int which, double* cov);
}  // namespace lsr
