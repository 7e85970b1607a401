// Micro-benchmark (round 4): what does the kernel BOUNDARY of the NDT launch chain cost with the host out of the
// picture, and is there a cheaper way to carry the 29 sums across it?
//
// One "pass" = one launch of 235 workgroups x 512 threads with 100 KiB of dynamic LDS (one workgroup per CU, as
// ndt_eval_quad_kernel with its voxel table).  Kernel bodies:
//   empty     nothing                                              -> the bare dependent-launch boundary
//   exchange  head: every workgroup folds 8 shards x NW words the previous launch left (what the chain does today);
//             tail: NW agent-scope atomics into this launch's bank
//   ticket    tail: NW atomics, release fence, a 4-byte arrival ticket; the workgroup that draws the last ticket folds the shards
//             and writes a 1 KiB "state" (stand-in for controller + request); head: every workgroup reads that 1 KiB only
//             (VERDICT r03 #3b: the controller at the tail of the previous launch)
// Launch methods:
//   eager     hipLaunchKernelGGL in a host loop (what tools/micro/sync_probe.hip timed in round 3)
//   graph     the same launches captured once into a hipGraph of 64 kernel nodes, replayed: the host enqueues one graph per 64
//             passes, so its per-launch enqueue rate cannot be what is measured
// Two clocks: hipEvents around the whole chain (launch to launch), and s_memrealtime stamps inside the kernels — first entry /
// last exit over the workgroups of a launch — which split a pass into "kernel span" and "gap to the next launch's first wave".
//   hipcc --offload-arch=gfx950 -O3 -o boundary_probe boundary_probe.hip && ./boundary_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int THREADS = 512, NWG = 235, S = 8, DYN_LDS = 100 * 1024, PASSES = 2048, NODES = 64;

struct Args {
  unsigned long long* bins;     // [2 banks][S][NW]
  unsigned long long* state;    // [2][128] 1 KiB "controller state"
  unsigned int* tickets;        // [PASSES]
  unsigned long long* stamps;   // [PASSES][NWG][2]: entry, exit of every workgroup (wall_clock64 ticks, 100 MHz; plain stores — an
                                // atomicMin / atomicMax per workgroup on one address cost 3.7 us per pass by itself)
  unsigned long long* sink;
  int nw, mode, stamp;
};

__global__ __launch_bounds__(THREADS) void pass_kernel(const Args a, const int seq) {
  extern __shared__ unsigned char dyn[];
  __shared__ unsigned long long s_sum[256];
  __shared__ unsigned int s_ticket;
  const int tid = threadIdx.x;
  if (a.stamp && tid == 0) a.stamps[((size_t)seq * NWG + blockIdx.x) * 2] = (unsigned long long)wall_clock64();
  if (a.mode == 1) {          // exchange: head folds the shards of the previous bank
    const unsigned long long* prev = a.bins + (size_t)((seq + 1) & 1) * S * a.nw;
    unsigned long long acc = 0;
    if (tid < a.nw)
      for (int s = 0; s < S; s++) acc += prev[(size_t)s * a.nw + tid];
    if (tid < a.nw) s_sum[tid] = acc;
    __syncthreads();
  } else if (a.mode == 2) {   // ticket: head reads the 1 KiB state the last workgroup of the previous launch wrote
    if (tid < 128) s_sum[tid] = __hip_atomic_load(&a.state[((seq + 1) & 1) * 128 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  if (a.mode >= 1) {
    unsigned long long* bank = a.bins + (size_t)(seq & 1) * S * a.nw;
    if (tid < a.nw) atomicAdd(&bank[(size_t)(blockIdx.x % S) * a.nw + tid], (unsigned long long)(tid + 1) + (s_sum[tid] & 1));
    if (blockIdx.x == 0 && tid < a.nw) a.sink[tid] = s_sum[tid];
  }
  if (a.mode == 2) {
    // no release fence (it writes back the XCD's L2: 30 us per pass with 235 workgroups): every word of the exchange is an
    // agent-scope atomic, so it is enough that this workgroup's atomics have been performed before its ticket is drawn
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(&a.tickets[seq], 1u);
    __syncthreads();
    if (s_ticket == (unsigned int)gridDim.x - 1) {   // last to arrive: fold + "controller" + state for the next launch
      const unsigned long long* bank = a.bins + (size_t)(seq & 1) * S * a.nw;
      unsigned long long acc = 0;
      if (tid < a.nw)
        for (int s = 0; s < S; s++) acc += __hip_atomic_load(&bank[(size_t)s * a.nw + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid < 128) a.state[(seq & 1) * 128 + tid] = acc + seq;
    }
  }
  if (a.stamp && tid == 0) a.stamps[((size_t)seq * NWG + blockIdx.x) * 2 + 1] = (unsigned long long)wall_clock64();
  if (dyn[0] == 77 && a.nw < 0) a.sink[0] = 1;   // keeps the dynamic LDS allocation alive
}

__global__ void fill_kernel(unsigned long long* p, size_t n, unsigned long long v) {
  for (size_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) p[k] = v;
}

int main() {
  Args a;
  const int NWMAX = 160;
  CK(hipMalloc(&a.bins, sizeof(unsigned long long) * 2 * S * NWMAX));
  CK(hipMalloc(&a.state, sizeof(unsigned long long) * 256));
  CK(hipMalloc(&a.tickets, sizeof(unsigned int) * PASSES));
  CK(hipMalloc(&a.stamps, sizeof(unsigned long long) * 2 * PASSES * NWG));
  CK(hipMalloc(&a.sink, sizeof(unsigned long long) * 1024));
  CK(hipFuncSetAttribute((const void*)pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<unsigned long long> h((size_t)2 * PASSES), hw((size_t)2 * PASSES * NWG);
  const char* mode_name[3] = {"empty", "exchange", "ticket"};
  printf("%-9s %-6s %5s %6s | %8s | %9s %9s %9s\n", "body", "launch", "words", "stamps", "events", "span", "gap", "span+gap");
  printf("%-9s %-6s %5s %6s | %8s | %9s %9s %9s   (us per pass; span = first wave in .. last wave out, gap = last out .. next first in; medians)\n", "", "", "", "", "", "", "", "");
  for (int mode = 0; mode < 3; mode++)
    for (int nw : {40, 145}) {
      if (mode == 0 && nw != 40) continue;
      for (int graph = 0; graph < 2; graph++)
        for (int stamp = 0; stamp < 2; stamp++) {
          a.nw = nw; a.mode = mode; a.stamp = stamp;
          hipGraph_t g = nullptr;
          hipGraphExec_t ge = nullptr;
          float best = 1e30f;
          for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, st, a.bins, (size_t)2 * S * NWMAX, 0ull);
            hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, st, (unsigned long long*)a.tickets, (size_t)PASSES / 2, 0ull);
            CK(hipStreamSynchronize(st));
            if (graph) {
              // one graph per 64 passes; seq is a kernel argument, so every block of 64 is its own captured graph (captured outside the timed region)
              std::vector<hipGraphExec_t> execs;
              for (int b = 0; b < PASSES / NODES; b++) {
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < NODES; k++) hipLaunchKernelGGL(pass_kernel, dim3(NWG), dim3(THREADS), DYN_LDS, st, a, b * NODES + k);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphDestroy(g));
                execs.push_back(ge);
              }
              CK(hipEventRecord(e0, st));
              for (hipGraphExec_t x : execs) CK(hipGraphLaunch(x, st));
              CK(hipEventRecord(e1, st));
              CK(hipEventSynchronize(e1));
              for (hipGraphExec_t x : execs) CK(hipGraphExecDestroy(x));
            } else {
              CK(hipEventRecord(e0, st));
              for (int k = 0; k < PASSES; k++) hipLaunchKernelGGL(pass_kernel, dim3(NWG), dim3(THREADS), DYN_LDS, st, a, k);
              CK(hipEventRecord(e1, st));
              CK(hipEventSynchronize(e1));
            }
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
          }
          double span = 0, gap = 0;
          if (stamp) {
            CK(hipMemcpy(hw.data(), a.stamps, sizeof(unsigned long long) * 2 * PASSES * NWG, hipMemcpyDeviceToHost));
            for (int k = 0; k < PASSES; k++) {   // first wave in, last wave out over the workgroups of launch k
              unsigned long long lo = ~0ull, hi = 0ull;
              for (int b = 0; b < NWG; b++) { lo = std::min(lo, hw[((size_t)k * NWG + b) * 2]); hi = std::max(hi, hw[((size_t)k * NWG + b) * 2 + 1]); }
              h[2 * k] = lo; h[2 * k + 1] = hi;
            }
            std::vector<double> sp, gp;
            for (int k = 64; k + 1 < PASSES; k++) {
              sp.push_back((double)(h[2 * k + 1] - h[2 * k]) * 0.01);
              gp.push_back((double)((long long)h[2 * (k + 1)] - (long long)h[2 * k + 1]) * 0.01);
            }
            std::sort(sp.begin(), sp.end()); std::sort(gp.begin(), gp.end());
            span = sp[sp.size() / 2]; gap = gp[gp.size() / 2];
          }
          if (stamp) printf("%-9s %-6s %5d %6s | %8.2f | %9.2f %9.2f %9.2f\n", mode_name[mode], graph ? "graph" : "eager", nw, "yes", 1e3 * best / PASSES, span, gap, span + gap);
          else printf("%-9s %-6s %5d %6s | %8.2f |\n", mode_name[mode], graph ? "graph" : "eager", nw, "no", 1e3 * best / PASSES);
          fflush(stdout);
        }
    }
  return 0;
}
