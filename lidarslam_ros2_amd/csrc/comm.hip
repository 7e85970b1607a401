// Multi-GPU sharding of a batch of independent registrations at the C ABI: one process per GPU, a static
// block partition or a cost-aware longest-first plan of the batch, no collective on the data path, ONE ncclAllGather of fixed 64-byte result records over
// xGMI at the end.  RCCL is loaded lazily (dlopen): single-GPU users of the library never touch it.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "handle.hpp"

namespace {

// the five RCCL entry points used, with the types of <rccl/rccl.h> restated (ncclUniqueId = 128 opaque bytes passed by
// value, ncclComm_t = opaque pointer, ncclResult_t = int with 0 = success, ncclUint8 = 1)
struct UniqueId { char internal[128]; };
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_all_gather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream);
typedef int (*fn_broadcast)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t stream);
typedef const char* (*fn_get_error_string)(int);

struct Rccl {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_broadcast broadcast = nullptr;   // optional (lsr_set_input_target_bcast); absent in no RCCL this library has met
  fn_get_error_string get_error_string = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // LSR_RCCL_LIB names the collective library instead (any library exporting the six nccl* entry points used here: a site's own
    // RCCL build — or tests/cpp/stub_ccl.cpp, which carries the collectives of two PROCESSES sharing ONE device through a shared
    // memory segment, so that every world > 1 line below executes on a one-GPU box; RCCL itself refuses two ranks on one device)
    const char* over = std::getenv("LSR_RCCL_LIB");
    if (over && *over) {
      r.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
      if (!r.lib) { std::fprintf(stderr, "lidarslam_reg: LSR_RCCL_LIB=%s could not be loaded: %s\n", over, dlerror()); return; }
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
      }
    }
    if (!r.lib) return;
    r.get_unique_id = (fn_get_unique_id)dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.lib, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.lib, "ncclCommDestroy");
    r.all_gather = (fn_all_gather)dlsym(r.lib, "ncclAllGather");
    r.broadcast = (fn_broadcast)dlsym(r.lib, "ncclBroadcast");
    r.get_error_string = (fn_get_error_string)dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather) { dlclose(r.lib); r.lib = nullptr; }
  });
  return r.lib ? &r : nullptr;
}

int rccl_fail(const char* what, int code) {
  Rccl* r = rccl();
  lsr::set_last_error(std::string(what) + " -> " + ((r && r->get_error_string) ? r->get_error_string(code) : "RCCL error") + " (" +
                      std::to_string(code) + ")");
  return LSR_ERR_HIP;
}

}  // namespace

struct lsr_comm_s {
  void* comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;   // orders the communicator's stream behind a handle's (lsr_set_input_target_bcast, device-resident root cloud)
  // Exchange buffers, allocated when the communicator is created (COMM_PREALLOC records per rank: 64 KiB + world x 64 KiB) so
  // that the sharded calls normally allocate nothing.  d_send is kept ARMED: between calls it holds records flagged invalid
  // (converged = -1, NaN), so a call whose upload of its own records fails still contributes well-formed "this share failed"
  // records to the all-gather instead of leaving the collective.
  lsr::DevBuf<lsr_shard_record> d_send, d_recv;
  size_t armed = 0;   // records of d_send currently holding the invalid pattern
  lsr::DevBuf<unsigned char> d_cloud, d_cloud_src;   // lsr_set_input_target_bcast: the broadcast target's records on this rank (+ the root's send copy)
  lsr::DevBuf<unsigned long long> d_header;          // ... its {point count, stride} header: [0..1] send, [2..3] receive; [4] this rank's ready word, [8..8+world) all of them
};
constexpr size_t COMM_PREALLOC = 1024;

// Where a rank's own share fails (an empty or ill-posed candidate, a HIP error in the registration OR in the exchange's own
// memset / upload), the rank STILL takes part in the all-gather: its records travel flagged invalid (converged = -1, NaN pose /
// score / fitness) and the error is returned after the collective — the other ranks never wait for a rank that has already
// left (the C ABI has no timeout or abort).  The one failure that cannot join is the growth of the exchange buffers beyond
// their preallocated size (hipMalloc): it is attempted BEFORE the rank's own work, so a rank that cannot take part says so
// at once instead of after its peers have entered the collective.
static void invalid_record(lsr_shard_record& R) {
  for (int k = 0; k < 12; k++) R.T[k] = NAN;
  R.score = NAN; R.iterations = 0.f; R.converged = -1.f; R.fitness = NAN;
}
// (re)fill the first `count` records of d_send with the invalid pattern (synchronous on the communicator's stream)
static int arm_send(lsr_comm c, size_t count) {
  std::vector<lsr_shard_record> inv(count);
  for (auto& R : inv) invalid_record(R);
  LSR_HIP(hipMemcpyAsync(c->d_send.p, inv.data(), sizeof(lsr_shard_record) * count, hipMemcpyHostToDevice, c->stream));
  LSR_HIP(hipStreamSynchronize(c->stream));
  c->armed = count;
  return LSR_OK;
}

extern "C" {

void lsr_shard_range(int n_items, int world, int rank, int* first, int* count) {
  if (world < 1) world = 1;
  const int base = n_items / world, extra = n_items % world;
  const int start = rank * base + std::min(rank, extra);
  if (first) *first = start;
  if (count) *count = base + (rank < extra ? 1 : 0);
}

// Longest-processing-time-first: items by cost descending (ties: lower index first), each to the rank with the least load so
// far (ties: lower rank).  Greedy LPT is within 4/3 - 1/(3 world) of the best makespan.  Costs within 2 % of each other (or none) give
// the block partition.
int lsr_shard_plan(int n_items, const double* cost, int world, int32_t* owner, int32_t* order, int32_t* rank_first) {
  if (n_items < 0 || world < 1 || (n_items > 0 && (!owner || !order)) || !rank_first) { lsr::set_last_error("bad shard-plan arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  std::vector<int> by_cost((size_t)n_items);
  for (int i = 0; i < n_items; i++) by_cost[i] = i;
  if (cost) {
    double lo = 0.0, hi = 0.0;
    for (int i = 0; i < n_items; i++) {
      if (!(cost[i] >= 0.0) || std::isinf(cost[i])) { lsr::set_last_error("shard-plan costs must be finite and non-negative"); return LSR_ERR_INVALID_ARGUMENT; }
      lo = (i == 0) ? cost[i] : std::min(lo, cost[i]);
      hi = (i == 0) ? cost[i] : std::max(hi, cost[i]);
    }
    // Costs the model cannot tell apart (spread within 2 % of the largest: cfg 4's candidates differ by a few hundred target points in
    // 661 k) are TIES: reshuffling the set by differences below the noise of the cost model buys nothing — the pass counts of the members,
    // which the sizes do not predict, decide the shares — so the plan is then the block partition (round 6; until then equal costs gave
    // round-robin, whose shares are as arbitrary and not contiguous)
    if (hi - lo <= 0.02 * hi) cost = nullptr;
    else std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  }
  if (!cost) {   // no costs / tied costs: the block partition of lsr_shard_range
    rank_first[0] = 0;
    for (int r = 0; r < world; r++) {
      int f = 0, c = 0;
      lsr_shard_range(n_items, world, r, &f, &c);
      for (int k = f; k < f + c; k++) { owner[k] = r; order[k] = k; }
      rank_first[r + 1] = f + c;
    }
    return LSR_OK;
  }
  std::vector<double> load((size_t)world, 0.0);
  std::vector<int> count((size_t)world, 0);
  for (int k = 0; k < n_items; k++) {
    int best = 0;
    for (int r = 1; r < world; r++) if (load[r] < load[best]) best = r;
    owner[by_cost[k]] = best;
    load[best] += cost[by_cost[k]];
    count[best]++;
  }
  rank_first[0] = 0;
  for (int r = 0; r < world; r++) rank_first[r + 1] = rank_first[r] + count[r];
  std::vector<int> fill(rank_first, rank_first + world);
  for (int k = 0; k < n_items; k++) order[fill[owner[by_cost[k]]]++] = by_cost[k];   // each rank's list: longest first
  return LSR_OK;
}

int lsr_comm_unique_id(void* id128) {
  if (!id128) return LSR_ERR_INVALID_ARGUMENT;
  Rccl* r = rccl();
  if (!r) { lsr::set_last_error("librccl.so could not be loaded"); return LSR_ERR_NOT_IMPLEMENTED; }
  UniqueId id;
  const int rc = r->get_unique_id(&id);
  if (rc) return rccl_fail("ncclGetUniqueId", rc);
  std::memcpy(id128, &id, sizeof(id));
  return LSR_OK;
}

int lsr_comm_create(const void* id128, int rank, int world, int device_id, lsr_comm* out) {
  if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) { lsr::set_last_error("bad communicator arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) { lsr::set_last_error("no such HIP device"); return LSR_ERR_NO_DEVICE; }
  lsr::DeviceGuard guard(device_id);   // the caller's current device is restored on every path out of here
  if (!guard.ok) { lsr::set_last_error("hipSetDevice failed"); return LSR_ERR_HIP; }
  lsr_comm c = new (std::nothrow) lsr_comm_s();
  if (!c) return LSR_ERR_HIP;
  c->rank = rank; c->world = world; c->device = device_id;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    lsr::set_last_error("communicator stream could not be created");
    return LSR_ERR_HIP;
  }
  if (world > 1 || id128) {  // a one-rank communicator created WITHOUT an id needs no RCCL at all; with one it is a real RCCL
                             // communicator of size 1 (the collective path can then be exercised on a single GPU)
    Rccl* r = rccl();
    if (!r) { (void)hipStreamDestroy(c->stream); delete c; lsr::set_last_error("librccl.so could not be loaded"); return LSR_ERR_NOT_IMPLEMENTED; }
    UniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const int rc = r->comm_init_rank(&c->comm, world, id, rank);
    if (rc) { (void)hipStreamDestroy(c->stream); delete c; return rccl_fail("ncclCommInitRank", rc); }
    if (c->d_send.reserve(COMM_PREALLOC) || c->d_recv.reserve(COMM_PREALLOC * (size_t)world) || c->d_header.reserve(8 + (size_t)world) ||
        hipMemset(c->d_header.p, 0, sizeof(unsigned long long) * (8 + (size_t)world)) != hipSuccess ||   // a header upload that fails announces zero points
        hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess || arm_send(c, COMM_PREALLOC)) {
      (void)r->comm_destroy(c->comm); (void)hipStreamDestroy(c->stream); if (c->ev) (void)hipEventDestroy(c->ev); delete c;
      return LSR_ERR_HIP;
    }
  }
  *out = c;
  return LSR_OK;
}

int lsr_comm_destroy(lsr_comm c) {
  if (!c) return LSR_OK;
  lsr::DeviceGuard guard(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) { Rccl* r = rccl(); if (r) (void)r->comm_destroy(c->comm); }
  (void)hipStreamDestroy(c->stream);
  if (c->ev) (void)hipEventDestroy(c->ev);
  delete c;
  return LSR_OK;
}

int lsr_align_batch_planned(lsr_comm c, lsr_handle* local_handles, int local_count, int global_count, const int32_t* order,
                            const int32_t* rank_first, const float* local_guesses, int with_fitness, lsr_shard_record* all_records) {
  if (!c || global_count <= 0 || !all_records || !order || !rank_first) { lsr::set_last_error("bad sharded-batch arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  // the plan is an argument every rank passes: check it here, before anything is exchanged (a bad plan is the same on all ranks)
  if (rank_first[0] != 0 || rank_first[c->world] != global_count) { lsr::set_last_error("shard plan does not cover the batch"); return LSR_ERR_INVALID_ARGUMENT; }
  {
    std::vector<char> seen((size_t)global_count, 0);
    for (int r = 0; r < c->world; r++)
      if (rank_first[r + 1] < rank_first[r]) { lsr::set_last_error("shard plan: rank_first must not decrease"); return LSR_ERR_INVALID_ARGUMENT; }
    for (int k = 0; k < global_count; k++) {
      if (order[k] < 0 || order[k] >= global_count || seen[order[k]]) { lsr::set_last_error("shard plan: order is not a permutation of the batch"); return LSR_ERR_INVALID_ARGUMENT; }
      seen[order[k]] = 1;
    }
  }
  const int first = rank_first[c->rank], mine = rank_first[c->rank + 1] - first;
  int max_count = 1;
  for (int rk = 0; rk < c->world; rk++) max_count = std::max(max_count, rank_first[rk + 1] - rank_first[rk]);
  const bool collective = !(c->world == 1 && !c->comm);
  lsr::DeviceGuard guard(c->device);
  if (collective) {
    // everything that can keep this rank OUT of the collective happens before its own work (see invalid_record above)
    if (!rccl() || !c->comm) { lsr::set_last_error("communicator has no RCCL handle"); return LSR_ERR_NOT_IMPLEMENTED; }
    if (!guard.ok) { lsr::set_last_error("hipSetDevice failed"); return LSR_ERR_HIP; }
    int st;
    if ((size_t)max_count > c->d_send.cap || (size_t)max_count * c->world > c->d_recv.cap) {
      if ((st = c->d_send.reserve((size_t)max_count))) return st;
      if ((st = c->d_recv.reserve((size_t)max_count * c->world))) return st;
      c->armed = 0;
    }
    if (c->armed < (size_t)max_count && (st = arm_send(c, (size_t)max_count))) return st;
  }
  // ---- this rank's share: no collective on the data path.  Argument errors of THIS rank are local failures too.
  int local_status = LSR_OK;
  std::string local_error;
  std::vector<lsr_shard_record> local((size_t)std::max(mine, 1));
  for (auto& R : local) invalid_record(R);
  if (local_count < 0 || (local_count > 0 && !local_handles)) {
    local_status = LSR_ERR_INVALID_ARGUMENT; local_error = "bad sharded-batch arguments";
  } else if (mine != local_count) {
    local_status = LSR_ERR_INVALID_ARGUMENT; local_error = "local_count does not match this rank's share of the batch (lsr_shard_range / lsr_shard_plan)";
  } else if (local_count > 0) {
    std::vector<float> finals((size_t)local_count * 16);
    std::vector<lsr_result> res((size_t)local_count);
    std::vector<double> fit((size_t)local_count, (double)NAN);
    // with fitness: one call — the searches of the candidates that finish early run under the launch chain of the others
    int st = with_fitness ? lsr_align_fitness_batch(local_handles, local_count, local_guesses, finals.data(), res.data(),
                                                    1.7976931348623157e308, fit.data())
                          : lsr_align_batch(local_handles, local_count, local_guesses, finals.data(), res.data());
    if (st) {
      local_status = st; local_error = lsr_last_error();
    } else {
      for (int b = 0; b < local_count; b++) {
        lsr_shard_record& R = local[b];
        const float* M = finals.data() + 16 * b;  // column-major 4x4 -> row-major 3x4
        for (int r = 0; r < 3; r++) for (int col = 0; col < 4; col++) R.T[r * 4 + col] = M[col * 4 + r];
        R.score = (float)res[b].score;
        R.iterations = (float)res[b].iterations;
        R.converged = res[b].converged ? 1.f : 0.f;
        R.fitness = with_fitness ? (float)fit[b] : NAN;
      }
    }
  }
  auto finish = [&](int collective_status) {
    if (local_status) { lsr::set_last_error(local_error); return local_status; }
    return collective_status;
  };
  if (!collective) {
    for (int k = 0; k < global_count; k++) all_records[order[k]] = local[k];
    return finish(LSR_OK);
  }
  // ---- ONE all-gather of fixed-size blocks (padded to the largest share): 64 B x 64 candidates = 4 KiB, latency bound.
  // d_send holds the invalid pattern; this rank's records replace it only if its share succeeded AND the upload works — a
  // failing upload becomes this rank's local failure, and the collective is entered all the same.
  Rccl* r = rccl();
  if (!local_status && mine > 0) {
    const hipError_t e = hipMemcpyAsync(c->d_send.p, local.data(), sizeof(lsr_shard_record) * (size_t)mine, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { local_status = LSR_ERR_HIP; local_error = std::string("upload of the shard records failed: ") + hipGetErrorString(e); }
    else c->armed = 0;
  }
  const int rc = r->all_gather(c->d_send.p, c->d_recv.p, sizeof(lsr_shard_record) * (size_t)max_count, /*ncclUint8*/ 1, c->comm, c->stream);
  if (rc) return rccl_fail("ncclAllGather", rc);
  std::vector<lsr_shard_record> table((size_t)max_count * c->world);
  LSR_HIP(hipMemcpyAsync(table.data(), c->d_recv.p, sizeof(lsr_shard_record) * table.size(), hipMemcpyDeviceToHost, c->stream));
  LSR_HIP(hipStreamSynchronize(c->stream));
  bool remote_invalid = false;
  for (int rk = 0; rk < c->world; rk++) {
    const int f = rank_first[rk], n = rank_first[rk + 1] - f;
    for (int k = 0; k < n; k++) {
      const lsr_shard_record& R = table[(size_t)rk * max_count + k];
      all_records[order[f + k]] = R;
      remote_invalid = remote_invalid || (rk != c->rank && R.converged < 0.f);
    }
  }
  (void)arm_send(c, (size_t)max_count);   // leave d_send armed for the next call (best effort: the next call re-checks)
  if (!local_status && remote_invalid) {   // the table is complete, but some other rank's share failed: say so
    lsr::set_last_error("another rank's share of the batch failed: its records are flagged converged = -1");
    return LSR_ERR_HIP;
  }
  return finish(LSR_OK);
}

// "N keyframes vs. one submap" across ranks (ncclBroadcast of the target, voxel table built redundantly per rank): the
// root holds the submap (scanmatcher_component.cpp:449-464 assembles it; :307 hands it to the registration object), every rank
// registers its own share of the scans against it.  Every rank of the communicator calls it, and every rank that has entered the
// exchange goes through ALL of its collectives whatever fails locally on the way (the C ABI has no abort):
//   1. ncclBroadcast of a 16-byte header {points, stride} — a root that cannot offer its cloud (ill-formed arguments, no staging
//      memory) announces zero points and every rank returns LSR_ERR_NO_TARGET alike;
//   2. every rank reserves room for the records, then ONE ncclAllGather of a ready word per rank: the records travel only if every
//      rank can receive them (a rank without memory, or whose handle lives on another device than the communicator, says so here
//      and all ranks return an error instead of some of them waiting in step 3 for a rank that has left);
//   3. ncclBroadcast of the records, device to device over xGMI, on the communicator's stream — ordered BEHIND the handle's stream
//      on the root when the records are device resident (the caller orders the handle's stream behind the producer with
//      lsr_wait_stream, as for every device input; the communicator's stream then waits for an event recorded there);
//   4. every rank builds the same voxel grid from the same bytes (lsr_set_input_target_device on the handle).
// A one-rank communicator hands the cloud straight through.
int lsr_set_input_target_bcast(lsr_comm c, lsr_handle h, const void* pts, size_t stride_bytes, size_t n, int on_device, int root) {
  if (!c || !h || root < 0 || root >= c->world) { lsr::set_last_error("bad broadcast-target arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  const bool is_root = (c->rank == root);
  // one rank: nothing to exchange, with or without an RCCL communicator behind it (ncclBroadcast on a communicator of ONE rank —
  // in place or out of place — left RCCL of ROCm 7.2 with a double free at ncclCommDestroy: round 5, tests/test_multigpu_gpu.py)
  if (c->world == 1) {
    if ((n > 0 && !pts) || stride_bytes < 12 || (stride_bytes % 4) != 0) { lsr::set_last_error("broadcast target: the cloud is ill-formed (null pointer or bad stride)"); return LSR_ERR_INVALID_ARGUMENT; }
    return on_device ? lsr_set_input_target_device(h, pts, stride_bytes, n) : lsr_set_input_target(h, pts, stride_bytes, n);
  }
  // ---- what is the same on every rank of a job (the library, the communicator's kind) may return at once
  Rccl* r = rccl();
  if (!r || !c->comm || !r->broadcast) { lsr::set_last_error("communicator has no RCCL broadcast"); return LSR_ERR_NOT_IMPLEMENTED; }
  lsr::DeviceGuard guard(c->device);
  if (!guard.ok) { lsr::set_last_error("hipSetDevice failed"); return LSR_ERR_HIP; }
  // ---- from here on every exit is behind the last collective.  Local failures are remembered and reported afterwards.
  int local_status = LSR_OK;
  std::string local_error;
  auto fail = [&](int st, const std::string& what) { if (!local_status) { local_status = st; local_error = what; } };
  if (h->device != c->device) fail(LSR_ERR_INVALID_ARGUMENT, "broadcast target: the handle lives on another device than the communicator");
  unsigned long long header[2] = {0ull, 12ull};
  const void* send = nullptr;
  if (is_root) {
    if ((n > 0 && !pts) || n == 0 || stride_bytes < 12 || (stride_bytes % 4) != 0) {
      fail(LSR_ERR_INVALID_ARGUMENT, "broadcast target: the root's cloud is empty or ill-formed (null pointer or bad stride)");
    } else if (on_device) {
      // device-resident records go out from where they are, once the handle's stream (which the caller has ordered behind their
      // producer) has reached this point
      hipError_t e = hipEventRecord(c->ev, h->stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev, 0);
      if (e != hipSuccess) fail(LSR_ERR_HIP, std::string("broadcast target: stream ordering failed: ") + hipGetErrorString(e));
      else send = pts;
    } else {
      const size_t bytes = n * stride_bytes;
      if (c->d_cloud_src.reserve(bytes)) fail(LSR_ERR_HIP, "broadcast target: no device memory to stage the root's cloud");
      else {
        const hipError_t e = hipMemcpyAsync(c->d_cloud_src.p, pts, bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) fail(LSR_ERR_HIP, std::string("broadcast target: staging the root's cloud failed: ") + hipGetErrorString(e));
        else send = c->d_cloud_src.p;
      }
    }
    if (!local_status) { header[0] = (unsigned long long)n; header[1] = (unsigned long long)stride_bytes; }
    // (a root that failed announces zero points; so does one whose header upload fails: the send words then keep what the last call —
    // or the zero fill at creation — left there only if that was zero, so they are cleared first)
    if (hipMemcpyAsync(c->d_header.p, header, sizeof(header), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
      fail(LSR_ERR_HIP, "broadcast target: the header could not be uploaded");
      (void)hipMemsetAsync(c->d_header.p, 0, sizeof(header), c->stream);
    }
  }
  // 1. header (send and receive buffers are kept apart: out-of-place collectives throughout)
  int rc = r->broadcast(c->d_header.p, c->d_header.p + 2, sizeof(header), /*ncclUint8*/ 1, root, c->comm, c->stream);
  if (rc) return rccl_fail("ncclBroadcast (header)", rc);   // the collective library itself failed: nothing more can be promised
  LSR_HIP(hipMemcpyAsync(header, c->d_header.p + 2, sizeof(header), hipMemcpyDeviceToHost, c->stream));
  LSR_HIP(hipStreamSynchronize(c->stream));
  const size_t count = (size_t)header[0], stride = (size_t)header[1], bytes = count * stride;
  if (count == 0) {   // every rank reads the same header: all leave here together
    if (local_status) { lsr::set_last_error(local_error); return local_status; }
    lsr::set_last_error("broadcast target: the root announced an empty cloud");
    return LSR_ERR_NO_TARGET;
  }
  // 2. can everybody receive?
  if (!local_status && c->d_cloud.reserve(bytes)) fail(LSR_ERR_HIP, "broadcast target: no device memory for the records on this rank");
  unsigned long long ready = local_status ? 0ull : 1ull;
  (void)hipMemcpyAsync(c->d_header.p + 4, &ready, sizeof(ready), hipMemcpyHostToDevice, c->stream);
  rc = r->all_gather(c->d_header.p + 4, c->d_header.p + 8, sizeof(ready), /*ncclUint8*/ 1, c->comm, c->stream);
  if (rc) return rccl_fail("ncclAllGather (ready words)", rc);
  std::vector<unsigned long long> all_ready((size_t)c->world, 0ull);
  LSR_HIP(hipMemcpyAsync(all_ready.data(), c->d_header.p + 8, sizeof(ready) * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
  LSR_HIP(hipStreamSynchronize(c->stream));
  int not_ready = -1;
  for (int rk = 0; rk < c->world; rk++) if (!all_ready[rk] && not_ready < 0) not_ready = rk;
  if (not_ready >= 0) {   // the same table on every rank: all leave here together
    if (local_status) { lsr::set_last_error(local_error); return local_status; }
    lsr::set_last_error("broadcast target: rank " + std::to_string(not_ready) + " cannot receive the records; nothing was sent");
    return LSR_ERR_HIP;
  }
  // 3. the records
  rc = r->broadcast(is_root ? send : (const void*)c->d_cloud.p, c->d_cloud.p, bytes, /*ncclUint8*/ 1, root, c->comm, c->stream);
  if (rc) return rccl_fail("ncclBroadcast (cloud)", rc);
  LSR_HIP(hipStreamSynchronize(c->stream));   // the handle reads the records on ITS stream
  // 4. the grid
  return lsr_set_input_target_device(h, c->d_cloud.p, stride, count);
}

// One all-gather of `count` 64-byte records per rank (the pose all-gather of north_star on its own: the scans of a stream a rank
// registered one after the other, a share registered through other entries).  all_records: world x count, rank-major.
int lsr_comm_all_gather_records(lsr_comm c, const lsr_shard_record* local, int count, lsr_shard_record* all_records) {
  if (!c || count <= 0 || !local || !all_records) { lsr::set_last_error("bad record all-gather arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  if (c->world == 1 && !c->comm) { std::memcpy(all_records, local, sizeof(lsr_shard_record) * (size_t)count); return LSR_OK; }
  Rccl* r = rccl();
  if (!r || !c->comm) { lsr::set_last_error("communicator has no RCCL handle"); return LSR_ERR_NOT_IMPLEMENTED; }
  lsr::DeviceGuard guard(c->device);
  if (!guard.ok) { lsr::set_last_error("hipSetDevice failed"); return LSR_ERR_HIP; }
  int st;
  if ((size_t)count > c->d_send.cap || (size_t)count * c->world > c->d_recv.cap) {   // before the collective (see invalid_record)
    if ((st = c->d_send.reserve((size_t)count))) return st;
    if ((st = c->d_recv.reserve((size_t)count * c->world))) return st;
  }
  c->armed = 0;
  int local_status = LSR_OK;
  std::string local_error;
  const hipError_t e = hipMemcpyAsync(c->d_send.p, local, sizeof(lsr_shard_record) * (size_t)count, hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) { local_status = LSR_ERR_HIP; local_error = std::string("upload of the records failed: ") + hipGetErrorString(e); }
  const int rc = r->all_gather(c->d_send.p, c->d_recv.p, sizeof(lsr_shard_record) * (size_t)count, /*ncclUint8*/ 1, c->comm, c->stream);
  if (rc) return rccl_fail("ncclAllGather", rc);
  LSR_HIP(hipMemcpyAsync(all_records, c->d_recv.p, sizeof(lsr_shard_record) * (size_t)count * c->world, hipMemcpyDeviceToHost, c->stream));
  LSR_HIP(hipStreamSynchronize(c->stream));
  if (local_status) { lsr::set_last_error(local_error); return local_status; }
  return LSR_OK;
}

// the static block partition is the plan { order = identity, rank_first = lsr_shard_range }
int lsr_align_batch_sharded(lsr_comm c, lsr_handle* local_handles, int local_count, int global_count, const float* local_guesses,
                            int with_fitness, lsr_shard_record* all_records) {
  if (!c || global_count <= 0 || !all_records) { lsr::set_last_error("bad sharded-batch arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  std::vector<int32_t> order((size_t)global_count), rank_first((size_t)c->world + 1);
  for (int k = 0; k < global_count; k++) order[k] = k;
  for (int r = 0; r <= c->world; r++) { int f = global_count, n = 0; if (r < c->world) lsr_shard_range(global_count, c->world, r, &f, &n); rank_first[r] = f; }
  return lsr_align_batch_planned(c, local_handles, local_count, global_count, order.data(), rank_first.data(), local_guesses, with_fitness,
                                 all_records);
}

}  // extern "C"
