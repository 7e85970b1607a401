// TEST INFRASTRUCTURE: the C-level multi-GPU path (csrc/comm.hip) with world = 2 — two PROCESSES, no torch, no MPI:
// fork(), rank 0 creates the unique id (lsr_comm_unique_id) and hands it to rank 1 over a pipe, both create a communicator
// (lsr_comm_create) and walk through EVERY world > 1 entry of comm.hip:
//   A  lsr_align_batch_sharded (block partition, with fitness): both ranks end with the SAME table, equal bit for bit to what one
//      process computes for all six candidates;
//   B  lsr_align_batch_planned (longest-first plan from lsr_shard_plan over members of different size): the same table again;
//   C  lsr_set_input_target_bcast: root 0 with a host cloud, root 1 with a DEVICE-resident cloud produced on another stream
//      (lsr_wait_stream orders the handle, the call orders the communicator's stream: ADVICE r05 medium) — every rank registers a
//      scan against the broadcast submap and lands on the pose of the table;
//   D  lsr_comm_all_gather_records: the pose all-gather on its own;
//   E  a share that fails on ONE rank (a member without a source) still joins the all-gather: that rank returns its own error, the
//      other rank returns LSR_ERR_HIP with its own records valid and the failed rank's flagged converged = -1 — and the
//      communicator is usable afterwards;
//   F  a root that has nothing to broadcast: every rank returns an error from the SAME collective (nobody is left waiting), and the
//      next broadcast works.
// Generalises graph_based_slam_component.cpp:188-231.  With two devices each rank takes its own and the collectives
// are RCCL's; on a ONE-device box both ranks share device 0 and the collectives come from tests/cpp/stub_ccl.cpp through
// LSR_RCCL_LIB (RCCL refuses two ranks on one device) — comm.hip runs the same lines either way.
// Prints "TWO_RANK ok=1 ..." on success, "SKIP" when the box has one device and no LSR_RCCL_LIB.
// Everything that touches HIP happens AFTER the fork (a forked child cannot inherit a HIP context).
#include <hip/hip_runtime.h>
#include <lidarslam_reg.h>

#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Pt { float x, y, z, pad, i, p1, p2, p3; };   // pcl::PointXYZI layout

// candidate c: two crossing planes + a scan of them displaced by a few centimetres; `every` thins the scan (members of different size)
static void make_case(int c, std::vector<Pt>& tgt, std::vector<Pt>& src, int every = 3) {
  tgt.clear(); src.clear();
  for (int i = 0; i < 6000; i++) {
    const float u = (i % 80) * 0.3f, v = (i / 80) * 0.3f;
    tgt.push_back({u + 0.04f * c, v, 0.02f * ((i * 7) % 5), 1.f, 0, 0, 0, 0});
    tgt.push_back({u, 0.05f * ((i * 3) % 7), v - 0.03f * c, 1.f, 0, 0, 0, 0});
    if (i % every == 0) src.push_back({u + 0.15f + 0.01f * c, v - 0.1f, 0.02f * ((i * 7) % 5), 1.f, (float)i, 0, 0, 0});
  }
}
static int thinning(int c) { return 2 + (c * 5) % 4; }   // 2..5: source sizes 3000 .. 1200

static lsr_handle make_ndt(int device) {
  lsr_handle h = nullptr;
  if (lsr_create(LSR_METHOD_NDT, device, nullptr, &h) != LSR_OK) return nullptr;
  lsr_set_f64(h, LSR_RESOLUTION, 5.0); lsr_set_f64(h, LSR_TRANSFORMATION_EPSILON, 0.01);
  lsr_set_i32(h, LSR_NEIGHBORHOOD, LSR_DIRECT7); lsr_set_i32(h, LSR_MAX_ITERATIONS, 35);
  return h;
}

struct Share {
  std::vector<lsr_handle> hs;
  ~Share() { for (lsr_handle h : hs) lsr_destroy(h); }
};

// handles for the candidates `items` (targets and sources set); skip_source >= 0: that member gets no source (scenario E)
static int build_share(Share& S, int device, const std::vector<int>& items, int skip_source = -1) {
  const int count = (int)items.size();
  std::vector<std::vector<Pt>> tg((size_t)count), sr((size_t)count);
  std::vector<const void*> tp, sp; std::vector<size_t> tc, sc;
  for (int k = 0; k < count; k++) {
    S.hs.push_back(make_ndt(device));
    if (!S.hs.back()) return 10;
    make_case(items[k], tg[k], sr[k], thinning(items[k]));
    tp.push_back(tg[k].data()); tc.push_back(tg[k].size()); sp.push_back(sr[k].data()); sc.push_back(sr[k].size());
  }
  if (count && lsr_set_input_target_batch(S.hs.data(), count, tp.data(), tc.data(), sizeof(Pt), 0) != LSR_OK) return 11;
  if (skip_source < 0) {
    if (count && lsr_set_input_source_batch(S.hs.data(), count, sp.data(), sc.data(), sizeof(Pt), 0) != LSR_OK) return 12;
  } else {
    for (int k = 0; k < count; k++)
      if (k != skip_source && lsr_set_input_source(S.hs[k], sp[k], sizeof(Pt), sc[k]) != LSR_OK) return 12;
  }
  return 0;
}

static std::vector<int> block_items(int total, int world, int rank) {
  int first = 0, n = 0;
  lsr_shard_range(total, world, rank, &first, &n);
  std::vector<int> v;
  for (int k = 0; k < n; k++) v.push_back(first + k);
  return v;
}

struct Plan { std::vector<int32_t> owner, order, rank_first; };
static Plan make_plan(int total, int world) {
  Plan P; P.owner.resize(total); P.order.resize(total); P.rank_first.resize(world + 1);
  std::vector<double> cost((size_t)total);
  for (int c = 0; c < total; c++) cost[c] = 6000.0 / thinning(c);   // the source size: what a pass costs
  if (lsr_shard_plan(total, cost.data(), world, P.owner.data(), P.order.data(), P.rank_first.data()) != LSR_OK) P.order.clear();
  return P;
}

static bool pose_matches(const float* T_colmajor, const lsr_shard_record& R) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++)
      if (T_colmajor[c * 4 + r] != R.T[r * 4 + c]) return false;
  return true;
}

struct Report {   // what rank 1 hands to rank 0 at the end
  int rc;
  int scen_e_local_status, scen_f_status;
  lsr_shard_record block[6], planned[6], gathered[6], failed[6];
};

// the whole walk of one rank; `comm` has world 2
static int run_rank(lsr_comm comm, int rank, int device, Report& rep) {
  const int total = 6, world = 2;
  // ---- A: block partition
  {
    Share S;
    int st = build_share(S, device, block_items(total, world, rank));
    if (st) return st;
    st = lsr_align_batch_sharded(comm, S.hs.data(), (int)S.hs.size(), total, nullptr, 1, rep.block);
    if (st != LSR_OK) { std::fprintf(stderr, "[rank %d] A sharded: %s\n", rank, lsr_last_error()); return 20; }
  }
  // ---- B: longest-first plan (same on both ranks: device-free and deterministic)
  {
    Plan P = make_plan(total, world);
    if (P.order.empty()) return 21;
    std::vector<int> items(P.order.begin() + P.rank_first[rank], P.order.begin() + P.rank_first[rank + 1]);
    Share S;
    int st = build_share(S, device, items);
    if (st) return st;
    st = lsr_align_batch_planned(comm, S.hs.data(), (int)S.hs.size(), total, P.order.data(), P.rank_first.data(), nullptr, 1, rep.planned);
    if (st != LSR_OK) { std::fprintf(stderr, "[rank %d] B planned: %s\n", rank, lsr_last_error()); return 22; }
  }
  // ---- C: the shared submap.  Root 0 sends candidate 0's target from the HOST; root 1 sends candidate 4's from DEVICE memory that a
  // copy on ANOTHER stream is still filling when the call is made.
  for (int round = 0; round < 2; round++) {
    const int root = round, cand = round == 0 ? 0 : 4;
    std::vector<Pt> t0, s0;
    make_case(cand, t0, s0, thinning(cand));
    lsr_handle b = make_ndt(device);
    if (!b) return 30;
    void* d_cloud = nullptr; hipStream_t producer = nullptr; Pt* pinned = nullptr;
    int bs;
    if (rank != root) {
      bs = lsr_set_input_target_bcast(comm, b, nullptr, 0, 0, 0, root);
    } else if (round == 0) {
      bs = lsr_set_input_target_bcast(comm, b, t0.data(), sizeof(Pt), t0.size(), /*on_device=*/0, root);
    } else {
      const size_t bytes = t0.size() * sizeof(Pt);
      if (hipSetDevice(device) != hipSuccess || hipMalloc(&d_cloud, bytes) != hipSuccess || hipStreamCreateWithFlags(&producer, hipStreamNonBlocking) != hipSuccess ||
          hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault) != hipSuccess) return 31;
      std::memcpy(pinned, t0.data(), bytes);
      if (hipMemsetAsync(d_cloud, 0xff, bytes, producer) != hipSuccess) return 31;          // NaNs first ...
      if (hipMemcpyAsync(d_cloud, pinned, bytes, hipMemcpyHostToDevice, producer) != hipSuccess) return 31;   // ... then the records
      if (lsr_wait_stream(b, producer) != LSR_OK) return 32;   // the caller's half of the ordering contract (lidarslam_reg.h)
      bs = lsr_set_input_target_bcast(comm, b, d_cloud, sizeof(Pt), t0.size(), /*on_device=*/1, root);
    }
    float T[16]; lsr_result res;
    if (bs == LSR_OK) bs = lsr_set_input_source(b, s0.data(), sizeof(Pt), s0.size());
    if (bs == LSR_OK) bs = lsr_align(b, nullptr, T, &res, nullptr, 0);
    lsr_destroy(b);
    if (producer) { (void)hipStreamSynchronize(producer); (void)hipStreamDestroy(producer); }
    if (d_cloud) (void)hipFree(d_cloud);
    if (pinned) (void)hipHostFree(pinned);
    if (bs != LSR_OK) { std::fprintf(stderr, "[rank %d] C broadcast target (root %d): %s\n", rank, root, lsr_last_error()); return 33; }
    if (!pose_matches(T, rep.block[cand])) { std::fprintf(stderr, "[rank %d] C broadcast target (root %d): pose differs from the table\n", rank, root); return 34; }
  }
  // ---- D: the record all-gather on its own
  {
    lsr_shard_record mine[3];
    std::memset(mine, 0, sizeof(mine));
    for (int k = 0; k < 3; k++) { mine[k].T[0] = (float)(100 * rank + k); mine[k].score = -1.5f * (rank + 1); mine[k].iterations = (float)k; mine[k].converged = 1.f; mine[k].fitness = 0.25f * k; }
    if (lsr_comm_all_gather_records(comm, mine, 3, rep.gathered) != LSR_OK) { std::fprintf(stderr, "[rank %d] D all-gather: %s\n", rank, lsr_last_error()); return 40; }
    for (int r = 0; r < world; r++)
      for (int k = 0; k < 3; k++) {
        const lsr_shard_record& R = rep.gathered[r * 3 + k];
        if (R.T[0] != (float)(100 * r + k) || R.score != -1.5f * (r + 1) || R.iterations != (float)k || R.fitness != 0.25f * k) return 41;
      }
  }
  // ---- E: rank 1's share fails (its second member has no source); both ranks still meet in the all-gather
  {
    Share S;
    int st = build_share(S, device, block_items(total, world, rank), rank == 1 ? 1 : -1);
    if (st) return st;
    st = lsr_align_batch_sharded(comm, S.hs.data(), (int)S.hs.size(), total, nullptr, 1, rep.failed);
    rep.scen_e_local_status = st;
    if (st == LSR_OK) { std::fprintf(stderr, "[rank %d] E: a share with a failed member returned LSR_OK\n", rank); return 50; }
    for (int k = 0; k < total; k++) {
      const bool of_rank1 = k >= 3;
      if (of_rank1 && !(rep.failed[k].converged == -1.f && std::isnan(rep.failed[k].T[0]))) { std::fprintf(stderr, "[rank %d] E: record %d of the failed share is not flagged\n", rank, k); return 51; }
      if (!of_rank1 && rank == 0 && std::memcmp(&rep.failed[k], &rep.block[k], sizeof(lsr_shard_record)) != 0) { std::fprintf(stderr, "[rank %d] E: record %d of the healthy share changed\n", rank, k); return 52; }
    }
    // ... and the communicator is usable afterwards
    Share S2;
    lsr_shard_record again[6];
    st = build_share(S2, device, block_items(total, world, rank));
    if (st) return st;
    if (lsr_align_batch_sharded(comm, S2.hs.data(), (int)S2.hs.size(), total, nullptr, 1, again) != LSR_OK) return 53;
    if (std::memcmp(again, rep.block, sizeof(again)) != 0) { std::fprintf(stderr, "[rank %d] E: the table after a failed call differs\n", rank); return 54; }
  }
  // ---- F: the root has nothing to send; everybody comes back with an error, then a good broadcast works
  {
    lsr_handle b = make_ndt(device);
    if (!b) return 60;
    const int st = lsr_set_input_target_bcast(comm, b, nullptr, sizeof(Pt), 0, 0, /*root=*/1);
    rep.scen_f_status = st;
    if (st == LSR_OK) return 61;
    if (rank == 0 && st != LSR_ERR_NO_TARGET) { std::fprintf(stderr, "[rank 0] F: expected LSR_ERR_NO_TARGET, got %d (%s)\n", st, lsr_last_error()); return 62; }
    std::vector<Pt> t0, s0;
    make_case(2, t0, s0, thinning(2));
    int bs = lsr_set_input_target_bcast(comm, b, rank == 1 ? t0.data() : nullptr, sizeof(Pt), rank == 1 ? t0.size() : 0, 0, /*root=*/1);
    float T[16]; lsr_result res;
    if (bs == LSR_OK) bs = lsr_set_input_source(b, s0.data(), sizeof(Pt), s0.size());
    if (bs == LSR_OK) bs = lsr_align(b, nullptr, T, &res, nullptr, 0);
    lsr_destroy(b);
    if (bs != LSR_OK || !pose_matches(T, rep.block[2])) { std::fprintf(stderr, "[rank %d] F: broadcast after a refused one failed: %s\n", rank, lsr_last_error()); return 63; }
  }
  return 0;
}

int main() {
  const int total = 6, world = 2;
  int id_pipe[2], res_pipe[2];
  if (pipe(id_pipe) || pipe(res_pipe)) return 2;
  const pid_t pid = fork();
  const int rank = pid == 0 ? 1 : 0;
  int ndev = 0;
  const char* over = std::getenv("LSR_RCCL_LIB");
  const bool share_device = !(lsr_device_count(&ndev) == LSR_OK && ndev >= 2);
  if (ndev < 1 || (share_device && !(over && *over))) {
    if (rank == 0) { int s; waitpid(pid, &s, 0); std::printf("SKIP devices=%d and no LSR_RCCL_LIB\n", ndev); }
    return 0;
  }
  const int device = share_device ? 0 : rank;
  char id[128];
  if (rank == 0) {
    if (lsr_comm_unique_id(id) != LSR_OK) { std::printf("TWO_RANK ok=0 unique_id: %s\n", lsr_last_error()); return 1; }
    if (write(id_pipe[1], id, sizeof(id)) != (ssize_t)sizeof(id)) return 3;
  } else if (read(id_pipe[0], id, sizeof(id)) != (ssize_t)sizeof(id)) {
    return 3;
  }
  lsr_comm comm = nullptr;
  Report rep;
  std::memset(&rep, 0, sizeof(rep));
  int rc = lsr_comm_create(id, rank, world, device, &comm);
  if (rc != LSR_OK) std::fprintf(stderr, "[rank %d] lsr_comm_create: %s\n", rank, lsr_last_error());
  if (rc == LSR_OK) rc = run_rank(comm, rank, device, rep);
  if (comm) lsr_comm_destroy(comm);   // destroyed normally, on both ranks
  rep.rc = rc;
  if (rank == 1) {   // hand the report to rank 0 and leave
    (void)!write(res_pipe[1], &rep, sizeof(rep));
    return rc;
  }
  Report r1;
  bool ok = rc == 0 && read(res_pipe[0], &r1, sizeof(r1)) == (ssize_t)sizeof(r1) && r1.rc == 0;
  int status = 0; waitpid(pid, &status, 0);
  ok = ok && WIFEXITED(status) && WEXITSTATUS(status) == 0;
  // both ranks hold the same tables ...
  ok = ok && std::memcmp(rep.block, r1.block, sizeof(rep.block)) == 0 && std::memcmp(rep.planned, r1.planned, sizeof(rep.planned)) == 0 &&
       std::memcmp(rep.gathered, r1.gathered, sizeof(rep.gathered)) == 0;
  // ... the plan does not change a bit (a registration's answer does not depend on the set it runs in) ...
  ok = ok && std::memcmp(rep.block, rep.planned, sizeof(rep.block)) == 0;
  // ... and they are what ONE process computes for the whole set through a one-rank communicator
  lsr_shard_record ref[6];
  {
    Share S;
    lsr_comm one = nullptr;
    std::vector<int> all;
    for (int c = 0; c < total; c++) all.push_back(c);
    bool r_ok = build_share(S, 0, all) == 0 && lsr_comm_create(nullptr, 0, 1, 0, &one) == LSR_OK &&
                lsr_align_batch_sharded(one, S.hs.data(), total, total, nullptr, 1, ref) == LSR_OK;
    if (one) lsr_comm_destroy(one);
    ok = ok && r_ok && std::memcmp(rep.block, ref, sizeof(ref)) == 0;
  }
  // E: rank 1 saw its own error, rank 0 the "another rank failed" one
  ok = ok && r1.scen_e_local_status != LSR_OK && rep.scen_e_local_status == LSR_ERR_HIP;
  ok = ok && r1.scen_f_status != LSR_OK && rep.scen_f_status == LSR_ERR_NO_TARGET;
  int conv = 0;
  for (const auto& R : rep.block) conv += R.converged == 1.f;
  std::printf("TWO_RANK ok=%d converged=%d/%d fitness0=%.5f devices=%d collectives=%s\n", ok ? 1 : 0, conv, total, rep.block[0].fitness, ndev,
              (over && *over) ? "stub (LSR_RCCL_LIB)" : "rccl");
  return ok ? 0 : 1;
}
