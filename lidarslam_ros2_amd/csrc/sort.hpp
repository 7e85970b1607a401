#pragma once
#include "common.hpp"

namespace lsr {
// Stable LSD radix sort of (key, value) pairs on bits [0, end_bit).
int sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const int* vals_in, int* vals_out, size_t n,
                   int end_bit, DevBuf<char>& temp, hipStream_t stream);
// Runs of equal keys: unique_out[r], counts_out[r], *num_runs_out (device int).
int run_length_encode_u32(const uint32_t* keys_sorted, size_t n, uint32_t* unique_out, int* counts_out,
                          int* num_runs_out, DevBuf<char>& temp, hipStream_t stream);
int exclusive_scan_i32(const int* in, int* out, size_t n, DevBuf<char>& temp, hipStream_t stream);
}  // namespace lsr
