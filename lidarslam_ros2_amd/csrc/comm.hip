// Multi-GPU sharding of a batch of independent registrations at the C ABI: one process per GPU, a static
// block partition or a cost-aware longest-first plan of the batch, no collective on the data path, ONE ncclAllGather of fixed 64-byte result records over
// xGMI at the end.  RCCL is loaded lazily (dlopen): single-GPU users of the library never touch it.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
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
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
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
  // Exchange buffers, allocated when the communicator is created (COMM_PREALLOC records per rank: 64 KiB + world x 64 KiB) so
  // that the sharded calls normally allocate nothing.  d_send is kept ARMED: between calls it holds records flagged invalid
  // (converged = -1, NaN), so a call whose upload of its own records fails still contributes well-formed "this share failed"
  // records to the all-gather instead of leaving the collective.
  lsr::DevBuf<lsr_shard_record> d_send, d_recv;
  size_t armed = 0;   // records of d_send currently holding the invalid pattern
  lsr::DevBuf<unsigned char> d_cloud, d_cloud_src;   // lsr_set_input_target_bcast: the broadcast target's records on this rank (+ the root's send copy)
  lsr::DevBuf<unsigned long long> d_header;          // ... and its {point count, stride} header: [0..1] send, [2..3] receive
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
// far (ties: lower rank).  Greedy LPT is within 4/3 - 1/(3 world) of the best makespan; with equal costs it is round-robin.
int lsr_shard_plan(int n_items, const double* cost, int world, int32_t* owner, int32_t* order, int32_t* rank_first) {
  if (n_items < 0 || world < 1 || (n_items > 0 && (!owner || !order)) || !rank_first) { lsr::set_last_error("bad shard-plan arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  std::vector<int> by_cost((size_t)n_items);
  for (int i = 0; i < n_items; i++) by_cost[i] = i;
  if (cost) {
    for (int i = 0; i < n_items; i++)
      if (!(cost[i] >= 0.0) || std::isinf(cost[i])) { lsr::set_last_error("shard-plan costs must be finite and non-negative"); return LSR_ERR_INVALID_ARGUMENT; }
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  }
  std::vector<double> load((size_t)world, 0.0);
  std::vector<int> count((size_t)world, 0);
  for (int k = 0; k < n_items; k++) {
    int best = 0;
    if (cost) {
      for (int r = 1; r < world; r++) if (load[r] < load[best]) best = r;
    } else {
      best = k % world;
    }
    owner[by_cost[k]] = best;
    load[best] += cost ? cost[by_cost[k]] : 1.0;
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
    if (c->d_send.reserve(COMM_PREALLOC) || c->d_recv.reserve(COMM_PREALLOC * (size_t)world) || arm_send(c, COMM_PREALLOC)) {
      (void)r->comm_destroy(c->comm); (void)hipStreamDestroy(c->stream); delete c;
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
// registers its own share of the scans against it.  Two broadcasts on the communicator's stream — a 16-byte header {points, stride},
// then the records, device to device over xGMI — and every rank builds the same voxel grid from the same bytes (lsr_set_input_target
// on the handle).  A one-rank communicator hands the cloud straight through.  Collective: every rank of the communicator calls it.
int lsr_set_input_target_bcast(lsr_comm c, lsr_handle h, const void* pts, size_t stride_bytes, size_t n, int on_device, int root) {
  if (!c || !h || root < 0 || root >= c->world) { lsr::set_last_error("bad broadcast-target arguments"); return LSR_ERR_INVALID_ARGUMENT; }
  const bool is_root = (c->rank == root);
  if (is_root && ((n > 0 && !pts) || stride_bytes < 12 || (stride_bytes % 4) != 0)) {
    // the root still has to enter the collective: it announces an empty cloud, every rank then fails alike
    n = 0; stride_bytes = 12; pts = nullptr;
    lsr::set_last_error("broadcast target: the root's cloud is ill-formed (null pointer or bad stride)");
  }
  // one rank: nothing to exchange, with or without an RCCL communicator behind it (ncclBroadcast on a communicator of ONE rank —
  // in place or out of place — left RCCL of ROCm 7.2 with a double free at ncclCommDestroy: round 5, tests/test_multigpu_gpu.py)
  if (c->world == 1)
    return on_device ? lsr_set_input_target_device(h, pts, stride_bytes, n) : lsr_set_input_target(h, pts, stride_bytes, n);
  Rccl* r = rccl();
  if (!r || !c->comm || !r->broadcast) { lsr::set_last_error("communicator has no RCCL broadcast"); return LSR_ERR_NOT_IMPLEMENTED; }
  lsr::DeviceGuard guard(c->device);
  if (!guard.ok) { lsr::set_last_error("hipSetDevice failed"); return LSR_ERR_HIP; }
  int st;
  // send and receive buffers are kept apart (out-of-place broadcasts: an in-place broadcast on a communicator of one rank left
  // RCCL 2.x of ROCm 7 with a double free at ncclCommDestroy)
  if ((st = c->d_header.reserve(4))) return st;
  unsigned long long header[2] = {(unsigned long long)n, (unsigned long long)stride_bytes};
  if (is_root) LSR_HIP(hipMemcpyAsync(c->d_header.p, header, sizeof(header), hipMemcpyHostToDevice, c->stream));
  int rc = r->broadcast(c->d_header.p, c->d_header.p + 2, sizeof(header), /*ncclUint8*/ 1, root, c->comm, c->stream);
  if (rc) return rccl_fail("ncclBroadcast (header)", rc);
  LSR_HIP(hipMemcpyAsync(header, c->d_header.p + 2, sizeof(header), hipMemcpyDeviceToHost, c->stream));
  LSR_HIP(hipStreamSynchronize(c->stream));
  const size_t count = (size_t)header[0], stride = (size_t)header[1], bytes = count * stride;
  if (count == 0) { lsr::set_last_error("broadcast target: the root announced an empty cloud"); return LSR_ERR_NO_TARGET; }
  if ((st = c->d_cloud.reserve(bytes))) return st;   // (a rank that cannot allocate leaves its peers in the second broadcast: there is no abort in the C ABI)
  const void* send = c->d_cloud.p;   // ranks other than the root: the send pointer is not read
  if (is_root) {
    if (on_device) {
      send = pts;   // device-resident records go out from where they are
    } else {
      if ((st = c->d_cloud_src.reserve(bytes))) return st;
      LSR_HIP(hipMemcpyAsync(c->d_cloud_src.p, pts, bytes, hipMemcpyHostToDevice, c->stream));
      send = c->d_cloud_src.p;
    }
  }
  rc = r->broadcast(send, c->d_cloud.p, bytes, /*ncclUint8*/ 1, root, c->comm, c->stream);
  if (rc) return rccl_fail("ncclBroadcast (cloud)", rc);
  LSR_HIP(hipStreamSynchronize(c->stream));   // the handle reads the records on ITS stream
  return lsr_set_input_target_device(h, c->d_cloud.p, stride, count);
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
