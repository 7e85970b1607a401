// TEST INFRASTRUCTURE: the C-level multi-GPU path (csrc/comm.hip) with world = 2 — two PROCESSES, one GPU each, no torch, no MPI:
// fork(), rank 0 creates the RCCL unique id (lsr_comm_unique_id) and hands it to rank 1 over a pipe, both create a communicator
// (lsr_comm_create), register their block of a 6-candidate set (lsr_align_batch_sharded, with fitness) and must end with the
// SAME table, equal to what one process computes for all six on one GPU.  Generalises graph_based_slam_component.cpp:190-231
// (SURVEY.md §8e).  Prints "TWO_RANK ok=1 ..." on success, "SKIP" when the box has fewer than two devices.
// Everything that touches HIP happens AFTER the fork (a forked child cannot inherit a HIP context).
#include <lidarslam_reg.h>

#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

struct Pt { float x, y, z, pad, i, p1, p2, p3; };   // pcl::PointXYZI layout

static void make_case(int c, std::vector<Pt>& tgt, std::vector<Pt>& src) {
  tgt.clear(); src.clear();
  for (int i = 0; i < 6000; i++) {
    const float u = (i % 80) * 0.3f, v = (i / 80) * 0.3f;
    tgt.push_back({u + 0.04f * c, v, 0.02f * ((i * 7) % 5), 1.f, 0, 0, 0, 0});
    tgt.push_back({u, 0.05f * ((i * 3) % 7), v - 0.03f * c, 1.f, 0, 0, 0, 0});
    if (i % 3 == 0) src.push_back({u + 0.15f + 0.01f * c, v - 0.1f, 0.02f * ((i * 7) % 5), 1.f, (float)i, 0, 0, 0});
  }
}

static lsr_handle make_ndt(int device) {
  lsr_handle h = nullptr;
  if (lsr_create(LSR_METHOD_NDT, device, nullptr, &h) != LSR_OK) return nullptr;
  lsr_set_f64(h, LSR_RESOLUTION, 5.0); lsr_set_f64(h, LSR_TRANSFORMATION_EPSILON, 0.01);
  lsr_set_i32(h, LSR_NEIGHBORHOOD, LSR_DIRECT7); lsr_set_i32(h, LSR_MAX_ITERATIONS, 35);
  return h;
}

// registers candidates [first, first + count) on `device`; comm == nullptr: plain batch (the single-process reference)
static int run_share(lsr_comm comm, int device, int first, int count, int total, std::vector<lsr_shard_record>& table) {
  std::vector<lsr_handle> hs;
  std::vector<std::vector<Pt>> tg((size_t)count), sr((size_t)count);
  std::vector<const void*> tp, sp; std::vector<size_t> tc, sc;
  for (int k = 0; k < count; k++) {
    hs.push_back(make_ndt(device));
    if (!hs.back()) return 10;
    make_case(first + k, tg[k], sr[k]);
    tp.push_back(tg[k].data()); tc.push_back(tg[k].size()); sp.push_back(sr[k].data()); sc.push_back(sr[k].size());
  }
  if (count && lsr_set_input_target_batch(hs.data(), count, tp.data(), tc.data(), sizeof(Pt), 0) != LSR_OK) return 11;
  if (count && lsr_set_input_source_batch(hs.data(), count, sp.data(), sc.data(), sizeof(Pt), 0) != LSR_OK) return 12;
  table.assign((size_t)total, lsr_shard_record{});
  int st;
  if (comm) {
    st = lsr_align_batch_sharded(comm, hs.data(), count, total, nullptr, 1, table.data());
  } else {
    lsr_comm one = nullptr;
    if (lsr_comm_create(nullptr, 0, 1, device, &one) != LSR_OK) return 13;
    st = lsr_align_batch_sharded(one, hs.data(), count, total, nullptr, 1, table.data());
    lsr_comm_destroy(one);
  }
  for (lsr_handle h : hs) lsr_destroy(h);
  if (st != LSR_OK) { std::fprintf(stderr, "share failed: %s\n", lsr_last_error()); return 14; }
  if (comm) {
    // "N keyframes vs. one submap" across ranks (lsr_set_input_target_bcast): rank 0 (the rank whose block starts at 0) holds the
    // target of candidate 0, every rank ends up with it and registers candidate 0's source against it: the pose of record 0
    std::vector<Pt> t0, s0;
    make_case(0, t0, s0);
    const bool root = (first == 0);
    lsr_handle b = make_ndt(device);
    if (!b) return 15;
    int bs = lsr_set_input_target_bcast(comm, b, root ? t0.data() : nullptr, sizeof(Pt), root ? t0.size() : 0, /*on_device=*/0, /*root=*/0);
    float T[16];
    lsr_result res;
    if (bs == LSR_OK) bs = lsr_set_input_source(b, s0.data(), sizeof(Pt), s0.size());
    if (bs == LSR_OK) bs = lsr_align(b, nullptr, T, &res, nullptr, 0);
    lsr_destroy(b);
    if (bs != LSR_OK) { std::fprintf(stderr, "broadcast target failed: %s\n", lsr_last_error()); return 16; }
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++)
        if (std::fabs(T[c * 4 + r] - table[0].T[r * 4 + c]) > 1e-6f) { std::fprintf(stderr, "broadcast target: pose differs\n"); return 17; }
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
  if (lsr_device_count(&ndev) != LSR_OK || ndev < 2) {
    if (rank == 0) { int s; waitpid(pid, &s, 0); std::printf("SKIP devices=%d\n", ndev); }
    return 0;
  }
  char id[128];
  if (rank == 0) {
    if (lsr_comm_unique_id(id) != LSR_OK) { std::printf("TWO_RANK ok=0 unique_id: %s\n", lsr_last_error()); return 1; }
    if (write(id_pipe[1], id, sizeof(id)) != (ssize_t)sizeof(id)) return 3;
  } else if (read(id_pipe[0], id, sizeof(id)) != (ssize_t)sizeof(id)) {
    return 3;
  }
  lsr_comm comm = nullptr;
  int rc = lsr_comm_create(id, rank, world, /*device=*/rank, &comm);
  int first = 0, mine = 0;
  lsr_shard_range(total, world, rank, &first, &mine);
  std::vector<lsr_shard_record> table;
  if (rc == LSR_OK) rc = run_share(comm, rank, first, mine, total, table);
  if (comm) lsr_comm_destroy(comm);
  if (rank == 1) {   // hand the table to rank 0 and leave
    table.resize((size_t)total);
    const int ok = rc == 0;
    (void)!write(res_pipe[1], &ok, sizeof(ok));
    (void)!write(res_pipe[1], table.data(), sizeof(lsr_shard_record) * total);
    return rc;
  }
  int ok1 = 0;
  std::vector<lsr_shard_record> t1((size_t)total);
  bool ok = rc == 0 && read(res_pipe[0], &ok1, sizeof(ok1)) == (ssize_t)sizeof(ok1) && ok1 &&
            read(res_pipe[0], t1.data(), sizeof(lsr_shard_record) * total) == (ssize_t)(sizeof(lsr_shard_record) * total);
  int status = 0; waitpid(pid, &status, 0);
  ok = ok && WIFEXITED(status) && WEXITSTATUS(status) == 0;
  // both ranks hold the same table ...
  ok = ok && std::memcmp(table.data(), t1.data(), sizeof(lsr_shard_record) * total) == 0;
  // ... and it is what one GPU computes for the whole set (one input, one answer: bit for bit)
  std::vector<lsr_shard_record> ref;
  const int rr = run_share(nullptr, 0, 0, total, total, ref);
  ok = ok && rr == 0 && std::memcmp(table.data(), ref.data(), sizeof(lsr_shard_record) * total) == 0;
  int conv = 0;
  for (const auto& R : table) conv += R.converged == 1.f;
  std::printf("TWO_RANK ok=%d converged=%d/%d fitness0=%.5f\n", ok ? 1 : 0, conv, total, table[0].fitness);
  return ok ? 0 : 1;
}
