// Kernels and helpers shared by the target-side builders (grid_build.hip) and the cloud codecs / voxel filter (cloud_codec.hip):
// each translation unit compiles its own copy (internal linkage).
#pragma once
#include "common.hpp"

namespace lsr {
namespace {

// key = linear leaf index exactly as VoxelGridCovariance computes it:
// ijk = (int)(floor(p * inv_leaf) - (float)min_b)
__global__ __launch_bounds__(256) void leaf_key_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ z, int n, float inv_leaf, int mb0, int mb1,
                                                       int mb2, int mul1, int mul2, unsigned int sentinel,
                                                       unsigned int* __restrict__ key, int* __restrict__ val,
                                                       uint4* __restrict__ fill_a, size_t n_a, int* __restrict__ fill_b, size_t n_b,
                                                       int* __restrict__ zero_word) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  // the all-ones fills of the sort-based builder (NaN leaf records of empty cells, cell_slot = -1) and its counter ride along:
  // three memset launches less
  for (size_t k = (size_t)i; k < n_a; k += (size_t)gridDim.x * blockDim.x) fill_a[k] = make_uint4(~0u, ~0u, ~0u, ~0u);
  for (size_t k = (size_t)i; k < n_b; k += (size_t)gridDim.x * blockDim.x) fill_b[k] = -1;
  if (i == 0 && zero_word) *zero_word = 0;
  if (i >= n) return;
  float px = x[i], py = y[i], pz = z[i];
  unsigned int k = sentinel;  // non-finite points: one past the last cell, sorts last
  if (isfinite(px) && isfinite(py) && isfinite(pz)) {
    int i0 = (int)(floorf(px * inv_leaf) - (float)mb0);
    int i1 = (int)(floorf(py * inv_leaf) - (float)mb1);
    int i2 = (int)(floorf(pz * inv_leaf) - (float)mb2);
    k = (unsigned int)(i0 + i1 * mul1 + i2 * mul2);
  }
  key[i] = k;
  if (val) val[i] = i;   // (the hand-written sort numbers the values itself)
}

inline int bits_for(unsigned int max_key) {  // radix bits needed to order keys in [0, max_key]
  int b = 1;
  while (b < 32 && (max_key >> b) != 0) b++;
  return b;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

}  // namespace
}  // namespace lsr
