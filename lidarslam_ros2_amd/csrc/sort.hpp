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

// ---- hand-written stable LSD radix sort + run finder (lsd_sort.hip) -------------------------------------------------------
// Sorts n pairs on key bits [0, end_bit) in ceil(end_bit / 11) passes, ping-ponging between (key_a, val_a) and (key_b, val_b);
// val_a == nullptr: the values are 0..n-1 (no iota pass), val_a_buf then serves as the "a" side from the second pass on.
// *result_in_b says where the sorted pairs ended up.  temp holds the histogram tables.
struct BuildScratch;
int sort_pairs_u32_lsd(unsigned int* key_a, unsigned int* key_b, int* val_a, int* val_a_buf, int* val_b, size_t n, int end_bit,
                       DevBuf<char>& temp, hipStream_t stream, bool* result_in_b, bool first_hist_done = false);
// The first pass's histogram table as that call will lay it out in temp: digit-major uint16 rows hist[digit * row_pitch + workgroup],
// workgroup = 2048 consecutive keys (key i of workgroup b: b * 2048 + i), digit = key & mask.  A kernel that produces the keys may fill
// it and pass first_hist_done = true (only when `usable`).
struct LsdFirstHist { unsigned short* hist = nullptr; int row_pitch = 0, C = 0, nblk = 0; unsigned int mask = 0; bool usable = false; };
int lsd_first_hist_plan(size_t n, int end_bit, DevBuf<char>& temp, LsdFirstHist* out);
// Runs of equal keys of a sorted sequence: heads counted per 256-key block (block_heads), scanned into block_base; the number of
// runs travels to the host through sc's mailbox (sorted_runs_count polls it).  sorted_runs_blocks(n) = ints each table needs.
size_t sorted_runs_blocks(size_t n);
int sorted_runs_begin(const unsigned int* keys_sorted, size_t n, int* block_heads, int* block_base, BuildScratch& sc, hipStream_t stream,
                      unsigned int* token_out, const unsigned int* dims_dev = nullptr,   // dims_dev: see leaf_key_dims_kernel (ndt.hip)
                      int* total_dev = nullptr);   // total_dev: the number of runs left in device memory too
int sorted_runs_count(BuildScratch& sc, hipStream_t stream, unsigned int token, int* n_runs);
// The runs as a table (enqueue behind sorted_runs_begin; needs no host wait): run r in key order has key run_key[r] and covers the
// sorted positions [run_off[r], run_off[r + 1]); run_off[number of runs] = n.  run_key / run_off: n + 1 entries.
int sorted_runs_table(const unsigned int* keys_sorted, size_t n, const int* block_base, unsigned int* run_key, int* run_off, hipStream_t stream);
// Exclusive scan of an int array of any length (hand-written, three launches); in != out.
int exclusive_scan_i32_lsd(const int* in, int* out, size_t n, DevBuf<char>& temp, hipStream_t stream);
// pcl::VoxelGrid's centroid of every run (float sums, ascending point index; all fields): out[r] for run r in key order; the run
// of `sentinel` (always last) is skipped
int sorted_runs_centroids(const unsigned int* keys_sorted, const int* order, size_t n, const int* block_base, unsigned int sentinel,
                          const float* x, const float* y, const float* z, const float* w, float* ox, float* oy, float* oz, float* ow,
                          hipStream_t stream, const unsigned int* sentinel_dev = nullptr);   // sentinel_dev: the sentinel lives on the device
}  // namespace lsr
