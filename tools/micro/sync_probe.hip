// Micro-benchmark: what does ONE all-to-all exchange of a few hundred bytes of sums between ~235 co-resident workgroups cost
// inside a launch, next to the dependent kernel boundary the NDT launch chain pays now (tail atomics -> boundary -> head read)?
//
//   chain      P dependent launches: every workgroup adds NV x NB chunk words into bins[shard] with agent-scope atomics; the
//              next launch's head reads all shards (what ndt_eval_quad_kernel does today).
//   counted    ONE launch, P passes: the atomics carry an arrival count in the high bits of the word they add to
//              (word += chunk + 2^40); every workgroup polls the NW x S words with sc1 loads until each word's count since
//              the previous pass equals the number of workgroups of its shard.  No fence, no flag, no separate barrier.
//   counter    ONE launch, P passes: atomics, then release fence + one monotonic arrival counter + poll + acquire fence, then
//              read the bins (the textbook grid barrier, for reference).
//
//   hipcc --offload-arch=gfx950 -O3 -o sync_probe sync_probe.hip && ./sync_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int THREADS = 512;
constexpr unsigned long long ONE = 1ull << 40;
constexpr int DYN_LDS = 100 * 1024;   // one workgroup per CU, as the NDT kernel with its voxel table in LDS

__device__ __forceinline__ unsigned long long load_sc1(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- chain: one pass per launch
__global__ __launch_bounds__(THREADS) void chain_pass(unsigned long long* bins, int nw, int S, int stride_w, int seq,
                                                      unsigned long long* sink) {
  __shared__ unsigned long long s_sum[512];
  const int tid = threadIdx.x;
  // head: fold the S shards of the previous launch's bank
  const unsigned long long* prev = bins + (size_t)((seq + 1) & 1) * S * nw * stride_w;
  unsigned long long acc = 0;
  if (tid < nw)
    for (int s = 0; s < S; s++) acc += prev[((size_t)s * nw + tid) * stride_w];
  if (tid < nw) s_sum[tid] = acc;
  __syncthreads();
  // (points phase would be here)
  unsigned long long* bank = bins + (size_t)(seq & 1) * S * nw * stride_w;
  const int shard = blockIdx.x % S;
  if (tid < nw) atomicAdd(&bank[((size_t)shard * nw + tid) * stride_w], (unsigned long long)(tid + 1));
  if (blockIdx.x == 0 && tid < nw) {   // clear what the launch after next adds to... (two banks: cleared by the reader side)
    sink[tid] = s_sum[tid];
  }
}
__global__ void clear_bank(unsigned long long* bins, size_t words) {
  for (size_t k = blockIdx.x * blockDim.x + threadIdx.x; k < words; k += (size_t)gridDim.x * blockDim.x) bins[k] = 0;
}

// ---- counted: persistent, count in the word
__global__ __launch_bounds__(THREADS) void counted_passes(unsigned long long* words, int nw, int S, int stride_w, int passes, int work_sleep,
                                                         int* error, long long* stamps) {
  __shared__ unsigned long long s_sum[1024];
  __shared__ int s_fail;
  const int tid = threadIdx.x, nblocks = gridDim.x;
  const int shard = blockIdx.x % S;
  if (tid == 0) s_fail = 0;
  __syncthreads();
  // this lane polls word (ps, pv)
  const int ps = tid / nw, pv = tid % nw;
  const bool poller = tid < nw * S;
  const unsigned long long expect_cnt = poller ? (unsigned long long)((nblocks - ps + S - 1) / S) : 0ull;   // workgroups b with b % S == ps
  const unsigned long long* pw = words + ((size_t)(poller ? ps : 0) * nw + (poller ? pv : 0)) * stride_w;
  unsigned long long prev = 0;
  const long long t0 = wall_clock64();
  for (int p = 0; p < passes; p++) {
    for (int k = 0; k < work_sleep; k++) __builtin_amdgcn_s_sleep(8);   // stands in for the points phase
    if (tid < nw) atomicAdd(&words[((size_t)shard * nw + tid) * stride_w], ONE + (unsigned long long)(tid + 1));
    if (poller) {
      int spins = 0;
      unsigned long long d;
      for (;;) {
        d = load_sc1(pw) - prev;
        if ((d >> 40) == expect_cnt) break;
        if (++spins > (1 << 16)) { atomicAdd(error, 1); s_fail = 1; break; }   // bounded: a stuck exchange ends the launch
      }
      prev += d;
      const unsigned long long sum = d & (ONE - 1);
      if (sum != expect_cnt * (unsigned long long)(pv + 1)) atomicAdd(error, 1);
      s_sum[tid] = sum;
    }
    __syncthreads();
    if (s_fail) return;
    // fold the shards (what the controller would read)
    if (tid < nw) {
      unsigned long long a = 0;
      for (int s = 0; s < S; s++) a += s_sum[s * nw + tid];
      if (a != (unsigned long long)nblocks * (tid + 1)) atomicAdd(error, 1);
    }
    __syncthreads();
  }
  if (tid == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = wall_clock64(); }
}

// ---- counter: persistent, atomics into banks + arrival counter
__global__ __launch_bounds__(THREADS) void counter_passes(unsigned long long* bins, unsigned int* counter, int nw, int S, int stride_w, int passes,
                                                         int* error, long long* stamps) {
  __shared__ unsigned long long s_sum[512];
  __shared__ int s_fail;
  const int tid = threadIdx.x, nblocks = gridDim.x;
  const int shard = blockIdx.x % S;
  if (tid == 0) s_fail = 0;
  __syncthreads();
  const long long t0 = wall_clock64();
  unsigned long long prev = 0;   // running sum per value (no clearing: monotone)
  for (int p = 0; p < passes; p++) {
    if (tid < nw) atomicAdd(&bins[((size_t)shard * nw + tid) * stride_w], (unsigned long long)(tid + 1));
    __builtin_amdgcn_s_waitcnt(0);   // the atomics have been acknowledged
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int want = (unsigned int)nblocks * (unsigned int)(p + 1);
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want)
        if (++spins > (1 << 16)) { atomicAdd(error, 1); s_fail = 1; break; }
    }
    __syncthreads();
    if (s_fail) return;
    if (tid < nw) {
      unsigned long long a = 0;
      for (int s = 0; s < S; s++) a += load_sc1(&bins[((size_t)s * nw + tid) * stride_w]);
      const unsigned long long d = a - prev;
      prev = a;
      if (d != (unsigned long long)nblocks * (tid + 1)) atomicAdd(error, 1);
      s_sum[tid] = d;
    }
    __syncthreads();
  }
  if (tid == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = wall_clock64(); }
}

int main() {
  const int passes = 2000;
  const size_t max_words = (size_t)2 * 8 * 160 * 32;
  unsigned long long *bins, *sink;
  unsigned int* counter;
  int* error;
  long long* stamps;
  CK(hipMalloc(&bins, sizeof(unsigned long long) * max_words));
  CK(hipMalloc(&sink, sizeof(unsigned long long) * 1024));
  CK(hipMalloc(&counter, 256));
  CK(hipMalloc(&error, 256));
  CK(hipMalloc(&stamps, sizeof(long long) * 2 * 1024));
  CK(hipFuncSetAttribute((const void*)chain_pass, hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS));
  CK(hipFuncSetAttribute((const void*)counted_passes, hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS));
  CK(hipFuncSetAttribute((const void*)counter_passes, hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int nwg : {235, 256}) {
    for (int nw : {40, 145}) {          // gradient-only pass (8 values x 5 bins) / pass with Hessian (29 x 5)
      // chain
      for (int S : {8}) {
        hipLaunchKernelGGL(clear_bank, dim3(64), dim3(256), 0, 0, bins, max_words);
        for (int rep = 0; rep < 2; rep++) {
          CK(hipEventRecord(e0, 0));
          for (int p = 0; p < passes; p++) hipLaunchKernelGGL(chain_pass, dim3(nwg), dim3(THREADS), DYN_LDS, 0, bins, nw, S, 1, p, sink);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep) printf("chain    wgs %3d words %3d shards %d           : %.2f us per pass (launch to launch)\n", nwg, nw, S, 1e3 * ms / passes);
        }
      }
      for (int stride_w : {1, 4, 16})      // words 8 / 32 / 128 bytes apart
        for (int S : {1, 2, 3}) {
          if (nw * S > THREADS) continue;
          for (int sleep : {0, 10}) {   // 10 x s_sleep(8) ~ 2.1 us of 'points phase'
            hipLaunchKernelGGL(clear_bank, dim3(64), dim3(256), 0, 0, bins, max_words);
            CK(hipMemset(error, 0, 4));
            for (int rep = 0; rep < 2; rep++) {
              if (rep) hipLaunchKernelGGL(clear_bank, dim3(64), dim3(256), 0, 0, bins, max_words);
              CK(hipEventRecord(e0, 0));
              hipLaunchKernelGGL(counted_passes, dim3(nwg), dim3(THREADS), DYN_LDS, 0, bins, nw, S, stride_w, passes, sleep, error, stamps);
              CK(hipEventRecord(e1, 0));
              CK(hipEventSynchronize(e1));
              float ms = 0;
              CK(hipEventElapsedTime(&ms, e0, e1));
              int herr = 0;
              CK(hipMemcpy(&herr, error, 4, hipMemcpyDeviceToHost));
              if (rep) printf("counted  wgs %3d words %3d shards %d stride %3d B sleep %d: %.2f us per pass, errors %d\n", nwg, nw, S, 8 * stride_w, sleep,
                              1e3 * ms / passes, herr);
            }
          }
        }
      for (int S : {8}) {
        hipLaunchKernelGGL(clear_bank, dim3(64), dim3(256), 0, 0, bins, max_words);
        CK(hipMemset(error, 0, 4));
        for (int rep = 0; rep < 2; rep++) {
          hipLaunchKernelGGL(clear_bank, dim3(64), dim3(256), 0, 0, bins, max_words);
          CK(hipMemset(counter, 0, 4));
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(counter_passes, dim3(nwg), dim3(THREADS), DYN_LDS, 0, bins, counter, nw, S, 1, passes, error, stamps);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          int herr = 0;
          CK(hipMemcpy(&herr, error, 4, hipMemcpyDeviceToHost));
          if (rep) printf("counter  wgs %3d words %3d shards %d           : %.2f us per pass, errors %d\n", nwg, nw, S, 1e3 * ms / passes, herr);
        }
      }
    }
  }
  return 0;
}
