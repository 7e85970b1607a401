// Device-wide sort / run-length / scan primitives used by the target-side builders (voxel grid
// K1, NN hash grid).  These are bulk utilities, not the registration hot loop; they are the one
// place the core leans on rocPRIM (header-only, ships with ROCm) instead of hand-written kernels.
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "sort.hpp"

namespace lsr {

int sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const int* vals_in, int* vals_out, size_t n,
                   int end_bit, DevBuf<char>& temp, hipStream_t stream) {
  if (n == 0) return LSR_OK;
  size_t bytes = 0;
  LSR_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, stream));
  int st = temp.reserve(bytes + 16);
  if (st) return st;
  LSR_HIP(rocprim::radix_sort_pairs((void*)temp.p, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, stream));
  return LSR_OK;
}

int run_length_encode_u32(const uint32_t* keys_sorted, size_t n, uint32_t* unique_out, int* counts_out,
                          int* num_runs_out, DevBuf<char>& temp, hipStream_t stream) {
  if (n == 0) {
    LSR_HIP(hipMemsetAsync(num_runs_out, 0, sizeof(int), stream));
    return LSR_OK;
  }
  size_t bytes = 0;
  LSR_HIP(rocprim::run_length_encode(nullptr, bytes, keys_sorted, (unsigned int)n, unique_out, counts_out,
                                     num_runs_out, stream));
  int st = temp.reserve(bytes + 16);
  if (st) return st;
  LSR_HIP(rocprim::run_length_encode((void*)temp.p, bytes, keys_sorted, (unsigned int)n, unique_out, counts_out,
                                     num_runs_out, stream));
  return LSR_OK;
}

int exclusive_scan_i32(const int* in, int* out, size_t n, DevBuf<char>& temp, hipStream_t stream) {
  if (n == 0) return LSR_OK;
  size_t bytes = 0;
  LSR_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int>(), stream));
  int st = temp.reserve(bytes + 16);
  if (st) return st;
  LSR_HIP(rocprim::exclusive_scan((void*)temp.p, bytes, in, out, 0, n, rocprim::plus<int>(), stream));
  return LSR_OK;
}

}  // namespace lsr
