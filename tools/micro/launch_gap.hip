// Micro-benchmark: GPU-side gap between DEPENDENT launches of a kernel that is long enough (~8 us busy wait)
// for the host to run ahead: stream launches vs hipGraph replay.  gap = per-launch time - in-kernel time.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void spin(long long* stamps, int slot, int ticks, float* sink) {
  const long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  long long t = t0;
  float v = (float)threadIdx.x;
  while (t - t0 < ticks) { v = v * 1.0001f + 1.f; t = __builtin_amdgcn_s_memrealtime(); }
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[2 * slot] = t0; stamps[2 * slot + 1] = t; }
  if (v == 123.456f) sink[0] = v;
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int reps = 200;
  long long* d_st; float* sink; CK(hipMalloc(&d_st, sizeof(long long) * 2 * reps)); CK(hipMalloc(&sink, 4));
  static long long h[2 * reps];
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int ticks : {100, 400, 800}) {
    for (int mode = 0; mode < 2; mode++) {
      hipGraph_t g; hipGraphExec_t ge = nullptr;
      if (mode == 1) {
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(spin, dim3(118), dim3(256), 0, s, d_st, i, ticks, sink);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      }
      for (int rep = 0; rep < 2; rep++) {   // second pass is the measured one
        CK(hipStreamSynchronize(s));
        hipEventRecord(a, s);
        if (mode == 0) for (int i = 0; i < reps; i++) hipLaunchKernelGGL(spin, dim3(118), dim3(256), 0, s, d_st, i, ticks, sink);
        else CK(hipGraphLaunch(ge, s));
        hipEventRecord(b, s);
        CK(hipEventSynchronize(b));
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      CK(hipMemcpy(h, d_st, sizeof(h), hipMemcpyDeviceToHost));
      double in_k = 0, gap = 0;
      for (int i = 0; i < reps; i++) in_k += (double)(h[2 * i + 1] - h[2 * i]) * 10.0;
      for (int i = 1; i < reps; i++) gap += (double)(h[2 * i] - h[2 * i - 1]) * 10.0;
      printf("spin %4.1f us  %-6s: %.2f us per launch (events) | in-kernel %.2f us | end->next start gap %.2f us\n", ticks * 0.01,
             mode ? "graph" : "stream", ms * 1e3 / reps, in_k / reps * 1e-3, gap / (reps - 1) * 1e-3);
    }
  }
  return 0;
}
