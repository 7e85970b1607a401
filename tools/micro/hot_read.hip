// Micro-benchmark: how long does the head of a kernel wait for a few KB that EVERY workgroup reads (state + sums of the
// previous launch), as a function of the number of workgroups, of how the data was produced (plain stores by one
// workgroup / device-scope atomics by all) and of replication (workgroup b reads copy b % R)?
//   hipcc --offload-arch=gfx950 -O3 -o hot_read hot_read.hip && ./hot_read
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// producer: mode 0 = workgroup 0 stores R copies of `words` 8-byte words; mode 1 = every workgroup adds 1 to every word of
// copy (b % R) with device-scope atomics (and workgroup 0 also stores nothing)
__global__ void produce(unsigned long long* data, int words, int R, int mode, int seq) {
  if (mode == 0) {
    if (blockIdx.x == 0)
      for (int k = threadIdx.x; k < words * R; k += blockDim.x) data[k] = (unsigned long long)seq * 1000 + k;
  } else {
    unsigned long long* d = data + (size_t)(blockIdx.x % R) * words;
    for (int k = threadIdx.x; k < words; k += blockDim.x) atomicAdd(&d[k], 1ull);
  }
}

// consumer: every workgroup reads ALL `words` of its copy (16-byte loads), then stamps
__global__ void consume(const unsigned long long* data, int words, int R, long long* stamps, unsigned long long* sink) {
  __shared__ unsigned long long s_acc[1024];
  const long long t0 = wall_clock64();
  const ulonglong2* d = reinterpret_cast<const ulonglong2*>(data + (size_t)(blockIdx.x % R) * words);
  unsigned long long a = 0;
  for (int k = threadIdx.x; k < words / 2; k += blockDim.x) { const ulonglong2 v = d[k]; a += v.x + v.y; }
  s_acc[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long t1 = wall_clock64();
    stamps[blockIdx.x * 2] = t0;
    stamps[blockIdx.x * 2 + 1] = t1;
    unsigned long long t = 0;
    for (int k = 0; k < (int)blockDim.x; k++) t += s_acc[k];
    sink[blockIdx.x] = t;
  }
}

int main() {
  const int maxwg = 1024, maxwords = 4096, maxR = 64;
  unsigned long long *data, *sink;
  long long* stamps;
  CK(hipMalloc(&data, sizeof(unsigned long long) * maxwords * maxR));
  CK(hipMalloc(&sink, sizeof(unsigned long long) * maxwg));
  CK(hipMalloc(&stamps, sizeof(long long) * 2 * maxwg));
  CK(hipMemset(data, 0, sizeof(unsigned long long) * maxwords * maxR));
  std::vector<long long> h(2 * maxwg);
  const int wgs[] = {118, 235, 469}, threads[] = {256, 512}, words_l[] = {128, 1408}, Rs[] = {1, 8, 32};
  for (int mode = 0; mode < 2; mode++)
    for (int words : words_l)
      for (int nt : threads)
        for (int nwg : wgs)
          for (int R : Rs) {
            std::vector<double> med;
            for (int rep = 0; rep < 60; rep++) {
              hipLaunchKernelGGL(produce, dim3(nwg), dim3(nt), 0, 0, data, words, R, mode, rep);
              hipLaunchKernelGGL(consume, dim3(nwg), dim3(nt), 0, 0, data, words, R, stamps, sink);
              if (rep < 10) continue;
              CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * nwg, hipMemcpyDeviceToHost));
              std::vector<double> d(nwg);
              for (int b = 0; b < nwg; b++) d[b] = 10.0 * (double)(h[2 * b + 1] - h[2 * b]);
              std::sort(d.begin(), d.end());
              med.push_back(d[nwg / 2]);
            }
            std::sort(med.begin(), med.end());
            printf("producer %s  %5d B  %3d threads  %3d WGs  R=%2d : per-WG entry->data in LDS median %.0f ns\n", mode ? "atomics" : "stores ",
                   words * 8, nt, nwg, R, med[med.size() / 2]);
          }
  return 0;
}
