// Micro-benchmark: per-launch cost of a chain of small dependent kernels vs LDS / scratch / grid size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDS_KB, bool SCRATCH>
__global__ __launch_bounds__(256) void k(float* out, const int* idx, int n) {
  __shared__ float lds[LDS_KB * 256 + 1];
  int t = threadIdx.x + blockIdx.x * 256;
  float v = 0.f;
  if (SCRATCH) {
    float arr[32];
    for (int i = 0; i < 32; i++) arr[i] = (float)(t + i);
    v = arr[idx[t & 31] & 31];   // dynamic index -> scratch
  }
  lds[threadIdx.x] = v + 1.f;
  __syncthreads();
  if (t < n) out[t] = lds[(threadIdx.x + 1) & 255] + out[t];
}

template <int LDS_KB, bool SCRATCH>
double run(int blocks, float* out, int* idx, int n, hipStream_t s, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<LDS_KB, SCRATCH>), dim3(blocks), dim3(256), 0, s, out, idx, n);
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<LDS_KB, SCRATCH>), dim3(blocks), dim3(256), 0, s, out, idx, n);
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / reps;
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int n = 1024 * 256; float* out; int* idx;
  CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&idx, 32 * 4)); CK(hipMemset(out, 0, n * 4)); CK(hipMemset(idx, 0, 128));
  int reps = 300;
  for (int blocks : {1, 118, 512}) {
    printf("blocks %4d: lds1 %.2f us | lds60 %.2f us | lds1+scratch %.2f us | lds60+scratch %.2f us\n", blocks,
           run<1, false>(blocks, out, idx, n, s, reps), run<60, false>(blocks, out, idx, n, s, reps),
           run<1, true>(blocks, out, idx, n, s, reps), run<60, true>(blocks, out, idx, n, s, reps));
  }
  // graph replay of the same chain
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<1, false>), dim3(118), dim3(256), 0, s, out, idx, n);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEventRecord(a, s); CK(hipGraphLaunch(ge, s)); hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("graph of %d x (118 blocks, lds1): %.2f us per kernel\n", reps, ms * 1e3 / reps);
  return 0;
}
