// Stable LSD radix sort of (uint32 key, int32 value) pairs for the sizes this library sorts — a 147k-point scan before
// pcl::VoxelGrid (N1: scanmatcher_component.cpp:324-328, every scan), a map-side cloud (:443-447), the keys of a target whose voxel
// index space is too large for one counting sort — built from the stages of the dense voxel-grid builder (grid_dense.hip): per
// workgroup LDS histogram of the digit -> per digit exclusive scan over the workgroups -> stable scatter (which scans the digit
// totals itself).
// Three passes of 10 bits order a 28-bit leaf index (three launches each); rocPRIM's sort, which this replaces on these paths, spends a dozen
// dependent launches of 5-10 us each on the same 147k keys (profiles/r04_rocprofv3_bench_stats.md: merge_sort_block_merge x 523).
//
// Stability does not lean on the order in which the LDS unit serves the lanes of one atomic instruction (the dense builder's
// scatter does; its consumers only need a FIXED order): the lanes of a wave that hold the same digit find each other with one
// ballot per digit bit, rank themselves by lane number, and ONE lane per group draws the group's offset from the wave's packed
// 16-bit counter — blocks, waves, steps and lanes are all in input order, so equal keys keep their input order, which is what
// makes the second pass correct and what pcl::VoxelGrid's float centroid (points of a leaf summed in ascending index) needs.
//
// The tail of N1 lives here too: the heads of the runs of equal keys are counted per workgroup, scanned by one workgroup (which
// also reports the number of runs to the host mailbox), and every head then sums its run — three launches instead of rocPRIM's
// run_length_encode + exclusive_scan + a publishing launch.
#include "handle.hpp"
#include "sort.hpp"

namespace lsr {

namespace {

constexpr int RS_THREADS = 256;

// TRANSPOSED: the counters are written digit-major, hist[d * row_pitch + workgroup] (row_pitch = workgroups rounded up to 8), so that
// the fused scatter's thread reads the whole row of its digit with a few 16-byte loads
template <int STEPS, bool TRANSPOSED>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const unsigned int* __restrict__ keys, int n, int shift, unsigned int mask,
                                                             int C, unsigned short* __restrict__ hist, int row_pitch) {
  extern __shared__ unsigned int s_hist[];  // [C]
  const int tid = threadIdx.x;
  for (int k = tid; k < C; k += RS_THREADS) s_hist[k] = 0u;
  __syncthreads();
  const int base = blockIdx.x * (RS_THREADS * STEPS);
  unsigned int kk[STEPS];
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    const int i = base + j * RS_THREADS + tid;
    kk[j] = (i < n) ? keys[i] : 0u;
  }
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    const int i = base + j * RS_THREADS + tid;
    if (i < n) atomicAdd(&s_hist[(kk[j] >> shift) & mask], 1u);
  }
  __syncthreads();
  if (TRANSPOSED) {
    for (int k = tid; k < C; k += RS_THREADS) hist[(size_t)k * row_pitch + blockIdx.x] = (unsigned short)s_hist[k];
  } else {
    unsigned short* row = hist + (size_t)blockIdx.x * C;
    for (int k = tid; k < C; k += RS_THREADS) row[k] = (unsigned short)s_hist[k];   // <= RS_THREADS * STEPS <= 4096
  }
}

// per digit: exclusive scan of the workgroup histograms + digit total, from DIGIT-MAJOR rows (hist[d * row_pitch + b], what rs_hist_kernel<S, true> writes), one WAVE per digit: 64 workgroup
// counts per step through a wave scan — C waves instead of C / 32 workgroups walking nblk rows one after the other (a 2.5 M-key sort:
// 611 rows; the row walk took 32 us per pass, more than the scatter).  blkoff stays workgroup-major for the scatter's coalesced read.
__global__ __launch_bounds__(256) void rs_scan_wave_kernel(const unsigned short* __restrict__ hist, int row_pitch, int nblk, int C,
                                                           unsigned int* __restrict__ blkoff, unsigned int* __restrict__ total) {
  const int d = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (d >= C) return;
  const unsigned short* row = hist + (size_t)d * row_pitch;
  unsigned int run = 0u;
  for (int b0 = 0; b0 < nblk; b0 += 64) {
    const int b = b0 + lane;
    const unsigned int v = (b < nblk) ? row[b] : 0u;
    unsigned int inc = v;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
      const unsigned int x = __shfl_up(inc, k, 64);
      if (lane >= k) inc += x;
    }
    if (b < nblk) blkoff[(size_t)b * C + d] = run + inc - v;
    run += __shfl(inc, 63, 64);
  }
  if (lane == 0) total[d] = run;
}

// Stable scatter of one pass.  Wave w of workgroup b owns the points [b * chunk + w * chunk / 4, + chunk / 4) and walks them in
// STEPS steps of 64 consecutive points; s_c[d] packs four 16-bit counters (one per wave): per-wave counts of digit d, then their
// exclusive prefix over the waves, then the running offset of each wave.
// Where a workgroup's points of digit d start in the output = (number of keys with a smaller digit) + (keys of digit d in the
// workgroups before this one).  FUSED (few workgroups: a scan): every workgroup forms both sums ITSELF from the histogram rows
// of the pass — nblk x C counters, read coalesced, thread t owning the digits t, t + 256, ... — so a pass is two launches
// (histogram, scatter).  Otherwise (a map: hundreds of rows) rs_scan has prefixed the rows (blkoff) and totalled the digits and the
// workgroup only scans the C totals.
constexpr int RS_MAX_BITS = 11, RS_MAX_C = 1 << RS_MAX_BITS, RS_OWN = RS_MAX_C / RS_THREADS;
constexpr int RS_FUSED_MAX_BLOCKS = 256;
template <int STEPS, bool FUSED>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const unsigned int* __restrict__ keys, const int* __restrict__ vals, int n,
                                                                int shift, unsigned int mask, int bits,
                                                                const unsigned int* __restrict__ blkoff, const unsigned int* __restrict__ total,
                                                                const unsigned short* __restrict__ hist, int row_pitch, int nblk,
                                                                int C, unsigned int* __restrict__ keys_out, int* __restrict__ vals_out) {
  extern __shared__ unsigned long long s_c[];  // [C]
  __shared__ unsigned int s_start[RS_MAX_C];   // absolute output position of this workgroup's first key of every digit
  __shared__ unsigned int s_wsum[RS_THREADS / 64];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  for (int k = tid; k < C; k += RS_THREADS) s_c[k] = 0ull;
  {
    unsigned int tot[RS_OWN], below[RS_OWN];   // of the digits u * 256 + tid
#pragma unroll
    for (int u = 0; u < RS_OWN; u++) { tot[u] = 0u; below[u] = 0u; }
    if (FUSED) {
      const int me = (int)blockIdx.x;
      const int own = (C + RS_THREADS - 1) / RS_THREADS;   // digit slices this workgroup's threads own: uniform
      const int nq = row_pitch / 8;                        // 16-byte pieces per digit-major row (row_pitch is a multiple of 32: no tail)
#pragma unroll
      for (int u0 = 0; u0 < RS_OWN; u0 += 2) {
        if (u0 < own) {   // uniform: the rows of two slices are read side by side, eight unconditional 16-byte loads in flight
          const int d0 = u0 * RS_THREADS + tid, d1 = d0 + RS_THREADS;
          const bool v0 = d0 < C, v1 = (u0 + 1 < own) && d1 < C;
          const uint4* r0 = reinterpret_cast<const uint4*>(hist + (size_t)(v0 ? d0 : 0) * row_pitch);
          const uint4* r1 = reinterpret_cast<const uint4*>(hist + (size_t)(v1 ? d1 : 0) * row_pitch);
          unsigned int t0 = 0u, t1 = 0u, l0 = 0u, l1 = 0u;
          for (int b8 = 0; b8 < nq; b8 += 4) {
            uint4 q0[4], q1[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { q0[k] = r0[b8 + k]; q1[k] = r1[b8 + k]; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const unsigned int w0[4] = {q0[k].x, q0[k].y, q0[k].z, q0[k].w}, w1[4] = {q1[k].x, q1[k].y, q1[k].z, q1[k].w};
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const int b = (b8 + k) * 8 + 2 * e;
                const bool in0 = b < nblk, in1 = b + 1 < nblk, be0 = b < me, be1 = b + 1 < me;   // the row padding holds no count
                const unsigned int a_lo = in0 ? (w0[e] & 0xFFFFu) : 0u, a_hi = in1 ? (w0[e] >> 16) : 0u;
                const unsigned int c_lo = in0 ? (w1[e] & 0xFFFFu) : 0u, c_hi = in1 ? (w1[e] >> 16) : 0u;
                t0 += a_lo + a_hi; t1 += c_lo + c_hi;
                l0 += (be0 ? a_lo : 0u) + (be1 ? a_hi : 0u);
                l1 += (be0 ? c_lo : 0u) + (be1 ? c_hi : 0u);
              }
            }
          }
          tot[u0] = v0 ? t0 : 0u; below[u0] = v0 ? l0 : 0u;
          if (u0 + 1 < RS_OWN) { tot[u0 + 1] = v1 ? t1 : 0u; below[u0 + 1] = v1 ? l1 : 0u; }
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < RS_OWN; u++) {
        const int d = u * RS_THREADS + tid;
        if (d < C) { tot[u] = total[d]; below[u] = blkoff[(size_t)blockIdx.x * C + d]; }
      }
    }
    // exclusive scan of tot[] in digit order: slice u = digits [256 u, 256 u + 255], one workgroup scan per slice, carry across slices
    unsigned int carry = 0u;
    for (int u = 0; u < RS_OWN && u * RS_THREADS < C; u++) {
      unsigned int inc = tot[u];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned int x = __shfl_up(inc, d, 64);
        if (lane >= d) inc += x;
      }
      __syncthreads();   // s_wsum of the previous slice has been read
      if (lane == 63) s_wsum[w] = inc;
      __syncthreads();
      unsigned int run = carry + inc - tot[u];
      for (int k = 0; k < w; k++) run += s_wsum[k];
      const int d = u * RS_THREADS + tid;
      if (d < C) s_start[d] = run + below[u];
      carry += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    }
  }
  const int base_i = blockIdx.x * (RS_THREADS * STEPS) + w * (64 * STEPS) + lane;
  unsigned int key[STEPS];
  int val[STEPS];
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    const int i = base_i + j * 64;
    const bool in = i < n;
    key[j] = in ? keys[i] : 0u;
    val[j] = in ? (vals ? vals[i] : i) : 0;
  }
  __syncthreads();
  const int sh = 16 * w;
#pragma unroll
  for (int j = 0; j < STEPS; j++)
    if (base_i + j * 64 < n) atomicAdd(&s_c[(key[j] >> shift) & mask], 1ull << sh);
  __syncthreads();
  for (int k = tid; k < C; k += RS_THREADS) {
    const unsigned long long v = s_c[k];
    const unsigned long long c0 = v & 0xFFFFull, c1 = (v >> 16) & 0xFFFFull, c2 = (v >> 32) & 0xFFFFull;
    s_c[k] = (c0 << 16) | ((c0 + c1) << 32) | ((c0 + c1 + c2) << 48);
  }
  unsigned int absb[STEPS];
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    const unsigned int d = (key[j] >> shift) & mask;
    absb[j] = (base_i + j * 64 < n) ? s_start[d] : 0u;
  }
  __syncthreads();
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));   // lanes below this one
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    const bool in = base_i + j * 64 < n;
    const unsigned int d = (key[j] >> shift) & mask;
    // the lanes of this step that hold the same digit: one ballot per digit bit
    unsigned long long peers = __ballot(in);
    for (int b = 0; b < bits; b++) {
      const bool one = ((d >> b) & 1u) != 0u;
      const unsigned long long bal = __ballot(one);
      peers &= one ? bal : ~bal;
    }
    if (in) {
      const int rank = __popcll(peers & lt), cnt = __popcll(peers);
      const int leader = __ffsll((long long)peers) - 1;
      unsigned int off = 0u;
      if (rank == 0) {
        const unsigned long long old = atomicAdd(&s_c[d], (unsigned long long)cnt << sh);
        off = (unsigned int)(old >> sh) & 0xFFFFu;
      }
      off = (unsigned int)__shfl((int)off, leader, 64);
      const unsigned int pos = absb[j] + off + (unsigned int)rank;
      keys_out[pos] = key[j];
      vals_out[pos] = val[j];
    }
  }
}

// ---- runs of equal keys in the sorted sequence ---------------------------------------------------------------------------
constexpr int RUN_CHUNK = 256;   // keys per workgroup: one per thread (a head's loop over its run is the long pole: keep it one per lane)
__device__ __forceinline__ bool is_head(const unsigned int* __restrict__ keys, int i) { return i == 0 || keys[i] != keys[i - 1]; }

__global__ __launch_bounds__(256) void rs_heads_count_kernel(const unsigned int* __restrict__ keys, int n, int* __restrict__ block_heads) {
  __shared__ int s_w[4];
  const int i = blockIdx.x * RUN_CHUNK + threadIdx.x;
  int c = (i < n && is_head(keys, i)) ? 1 : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_heads[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// one workgroup: exclusive scan of the per-workgroup head counts; the total goes to the host mailbox (value + token)
// dims (nullable): {sentinel, finite points, VG_FLAG_*, key bits} left by leaf_key_dims_kernel — handed to the host with the count
__global__ __launch_bounds__(1024) void rs_heads_scan_kernel(const int* __restrict__ block_heads, int nblocks, int* __restrict__ block_base,
                                                             BuildMailbox* __restrict__ mb, unsigned int token,
                                                             const unsigned int* __restrict__ dims, int* __restrict__ total_dev /*nullable*/) {
  __shared__ int s_w[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nblocks + 1023) / 1024;
  const int b0 = tid * per, b1 = min(nblocks, b0 + per);
  int cnt = 0;
  for (int b = b0; b < b1; b++) cnt += block_heads[b];
  int inc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(inc, d, 64);
    if (lane >= d) inc += v;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wave; w++) wbase += s_w[w];
  int run = wbase + inc - cnt;
  for (int b = b0; b < b1; b++) { const int t = block_heads[b]; block_base[b] = run; run += t; }
  if (tid == 1023) {
    if (total_dev) *total_dev = run;   // for kernels enqueued behind this one that must not wait for the host to learn the count
    mb->value = run;   // == total: thread 1023 owns the last (possibly empty) slice
    if (dims) { mb->vg_finite = dims[1]; mb->vg_flags = dims[2]; mb->vg_bits = dims[3]; }
    __threadfence_system();
    __hip_atomic_store(&mb->value_token, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// every head sums its run: FLOAT accumulators, points in ascending index (the sort is stable) — the very additions
// pcl::CentroidPoint performs; output ordered by key; the run of the sentinel key (non-finite points, always last) is dropped
__global__ __launch_bounds__(256) void rs_centroid_kernel(const unsigned int* __restrict__ keys, const int* __restrict__ order, int n,
                                                          const int* __restrict__ block_base, unsigned int sentinel_arg,
                                                          const unsigned int* __restrict__ sentinel_dev /*nullable: dims[0]*/,
                                                          const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                          const float* __restrict__ w /*nullable*/, float* __restrict__ ox,
                                                          float* __restrict__ oy, float* __restrict__ oz, float* __restrict__ ow) {
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * RUN_CHUNK + threadIdx.x;
  const bool head = (i < n) && is_head(keys, i);
  const unsigned long long heads = __ballot(head);
  if (lane == 0) s_w[wave] = __popcll(heads);
  __syncthreads();
  if (!head) return;
  int r = block_base[blockIdx.x] + __popcll(heads & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
  for (int k = 0; k < wave; k++) r += s_w[k];
  const unsigned int k = keys[i];
  const unsigned int sentinel = sentinel_dev ? *sentinel_dev : sentinel_arg;
  if (k == sentinel) return;
  float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
  int j = i;
  // the additions are a chain by definition (float, in index order); the gathers that feed them are not: four points in flight
  for (; j + 3 < n && keys[j + 3] == k; j += 4) {   // sorted keys: keys[j + 3] == k implies the three before it
    const int p0 = order[j], p1 = order[j + 1], p2 = order[j + 2], p3 = order[j + 3];
    const float x0 = x[p0], y0 = y[p0], z0 = z[p0], x1 = x[p1], y1 = y[p1], z1 = z[p1];
    const float x2 = x[p2], y2 = y[p2], z2 = z[p2], x3 = x[p3], y3 = y[p3], z3 = z[p3];
    sx += x0; sy += y0; sz += z0; sx += x1; sy += y1; sz += z1; sx += x2; sy += y2; sz += z2; sx += x3; sy += y3; sz += z3;
    if (w) { const float w0 = w[p0], w1 = w[p1], w2 = w[p2], w3 = w[p3]; sw += w0; sw += w1; sw += w2; sw += w3; }
  }
  for (; j < n && keys[j] == k; j++) {
    const int pi = order[j];
    sx += x[pi]; sy += y[pi]; sz += z[pi];
    if (w) sw += w[pi];
  }
  const float m = (float)(j - i);
  ox[r] = sx / m; oy[r] = sy / m; oz[r] = sz / m;
  if (ow) ow[r] = w ? sw / m : 0.f;
}

// The runs as a table (what rocPRIM's run_length_encode + exclusive_scan gave the sort-based builders): run r in key order has
// key run_key[r] and covers the sorted positions [run_off[r], run_off[r + 1]); run_off[number of runs] = n closes the table.
__global__ __launch_bounds__(256) void rs_runs_table_kernel(const unsigned int* __restrict__ keys, int n, const int* __restrict__ block_base,
                                                            unsigned int* __restrict__ run_key, int* __restrict__ run_off) {
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * RUN_CHUNK + threadIdx.x;
  const bool head = (i < n) && is_head(keys, i);
  const unsigned long long heads = __ballot(head);
  if (lane == 0) s_w[wave] = __popcll(heads);
  __syncthreads();
  if (i >= n) return;
  int r = block_base[blockIdx.x] + __popcll(heads & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));   // heads before this position
  for (int k = 0; k < wave; k++) r += s_w[k];
  if (head) { run_key[r] = keys[i]; run_off[r] = i; }
  if (i == n - 1) run_off[r + (head ? 1 : 0)] = n;   // r counts the heads BEFORE i: the table has r (+ 1 if i is a head itself) runs
}

// ---- exclusive scan of an int array of any length: three launches (sums of 4096-element blocks; one workgroup scans the sums;
// every block scans its elements behind its base) — the device-wide scan of the bucket-built neighbour grid (nn.hip)
constexpr int XS_THREADS = 256, XS_ITEMS = 16, XS_CHUNK = XS_THREADS * XS_ITEMS;
__global__ __launch_bounds__(XS_THREADS) void xs_reduce_kernel(const int* __restrict__ in, size_t n, int* __restrict__ block_sum) {
  __shared__ int s_w[XS_THREADS / 64];
  const size_t base = (size_t)blockIdx.x * XS_CHUNK;
  int s = 0;
#pragma unroll
  for (int j = 0; j < XS_ITEMS; j++) {   // coalesced: consecutive threads read consecutive words
    const size_t i = base + (size_t)j * XS_THREADS + threadIdx.x;
    s += (i < n) ? in[i] : 0;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) block_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(1024) void xs_scan_sums_kernel(int* __restrict__ block_sum, int nblocks) {   // in place: sums -> exclusive bases
  __shared__ int s_w[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nblocks + 1023) / 1024;
  const int b0 = tid * per, b1 = min(nblocks, b0 + per);
  int cnt = 0;
  for (int b = b0; b < b1; b++) cnt += block_sum[b];
  int inc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(inc, d, 64);
    if (lane >= d) inc += v;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int run = inc - cnt;
  for (int w = 0; w < wave; w++) run += s_w[w];
  for (int b = b0; b < b1; b++) { const int t = block_sum[b]; block_sum[b] = run; run += t; }
}
__global__ __launch_bounds__(XS_THREADS) void xs_apply_kernel(const int* __restrict__ in, size_t n, const int* __restrict__ block_base,
                                                              int* __restrict__ out) {
  __shared__ int s_w[XS_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // thread t owns the XS_ITEMS CONSECUTIVE elements [base + t * XS_ITEMS, ...): a thread-local running sum, one wave scan of the
  // thread totals, one exchange between the four waves (the loads are 64-byte runs per thread: four 16-byte loads)
  const size_t first = (size_t)blockIdx.x * XS_CHUNK + (size_t)tid * XS_ITEMS;
  int v[XS_ITEMS];
  if (first + XS_ITEMS <= n && ((reinterpret_cast<size_t>(in) & 15) == 0)) {
    const int4* q = reinterpret_cast<const int4*>(in + first);
#pragma unroll
    for (int j = 0; j < XS_ITEMS / 4; j++) { const int4 t = q[j]; v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w; }
  } else {
#pragma unroll
    for (int j = 0; j < XS_ITEMS; j++) v[j] = (first + j < n) ? in[first + j] : 0;
  }
  int tot = 0;
#pragma unroll
  for (int j = 0; j < XS_ITEMS; j++) tot += v[j];
  int inc = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int x = __shfl_up(inc, d, 64);
    if (lane >= d) inc += x;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int run = block_base[blockIdx.x] + inc - tot;
  for (int w = 0; w < wave; w++) run += s_w[w];
  if (first + XS_ITEMS <= n && ((reinterpret_cast<size_t>(out) & 15) == 0)) {
    int4* q = reinterpret_cast<int4*>(out + first);
#pragma unroll
    for (int j = 0; j < XS_ITEMS / 4; j++) {
      int4 t;
      t.x = run; run += v[4 * j]; t.y = run; run += v[4 * j + 1]; t.z = run; run += v[4 * j + 2]; t.w = run; run += v[4 * j + 3];
      q[j] = t;
    }
  } else {
#pragma unroll
    for (int j = 0; j < XS_ITEMS; j++) { if (first + j < n) out[first + j] = run; run += v[j]; }
  }
}

// ... and for arrays of at most one block (the occupied-coarse-cell flags of a scan's neighbour grid: ~1000 entries) the three steps in ONE
// launch: two launches and their gaps less on a chain of a dozen short launches (GICP setInputSource)
__global__ __launch_bounds__(XS_THREADS) void xs_single_block_kernel(const int* __restrict__ in, int n, int* __restrict__ out) {
  __shared__ int s_w[XS_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int first = tid * XS_ITEMS;
  int v[XS_ITEMS];
  int tot = 0;
#pragma unroll
  for (int j = 0; j < XS_ITEMS; j++) { v[j] = (first + j < n) ? in[first + j] : 0; tot += v[j]; }
  int inc = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int x = __shfl_up(inc, d, 64);
    if (lane >= d) inc += x;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int run = inc - tot;
  for (int w = 0; w < wave; w++) run += s_w[w];
#pragma unroll
  for (int j = 0; j < XS_ITEMS; j++) { if (first + j < n) out[first + j] = run; run += v[j]; }
}

struct LsdPlan { int passes, bits, steps, nblk, C; size_t table_bytes; };
LsdPlan lsd_plan(size_t n, int end_bit) {
  LsdPlan P;
  end_bit = std::max(1, std::min(32, end_bit));
  // digits of at most 11 bits: every pass is three launches (histogram, scan over the workgroups, scatter) whose table work —
  // zeroing, prefixing and scanning C counters per workgroup — stays small next to the points (14-bit digits, 16 384 counters per
  // 1 024-point workgroup, were measured first: 33 us per pass of a 147k-point scan, most of it tables)
  P.passes = (end_bit + RS_MAX_BITS - 1) / RS_MAX_BITS;
  P.bits = (end_bit + P.passes - 1) / P.passes;
  P.C = 1 << P.bits;
  P.steps = (n <= 600000) ? 8 : 16;   // a scan: 2048-point workgroups (72 for a 147k-point scan: every workgroup reads the whole table of
                                      // the pass; 1024-point workgroups were measured: 16 us per scatter, 9 of them reading 144 rows)
  const int chunk = RS_THREADS * P.steps;
  P.nblk = (int)((n + chunk - 1) / chunk);
  // [total C | blkoff nblk*C] u32, [hist nblk*C] u16
  P.table_bytes = ((size_t)P.C + (size_t)P.nblk * P.C) * 4 + (size_t)(P.nblk + 32) * P.C * 2 + 64;
  return P;
}

}  // namespace

static unsigned short* lsd_hist_table(const LsdPlan& P, char* temp) {
  unsigned int* total = reinterpret_cast<unsigned int*>(temp);
  unsigned int* blkoff = total + P.C;
  return reinterpret_cast<unsigned short*>((reinterpret_cast<uintptr_t>(blkoff + (size_t)P.nblk * P.C) + 15) & ~(uintptr_t)15);
}

// Where the histogram of the FIRST pass of sort_pairs_u32_lsd(n, end_bit, temp) lives and what it looks like, for a producer of the
// keys that counts their first digit itself (leaf_key_dims_hist_kernel, ndt.hip: one launch less per sort).  usable only for the fused
// form on 2048-key workgroups — what a scan is sorted with.
int lsd_first_hist_plan(size_t n, int end_bit, DevBuf<char>& temp, LsdFirstHist* out) {
  *out = LsdFirstHist{};
  if (n == 0 || n > (size_t)INT32_MAX / 2) return LSR_OK;
  const LsdPlan P = lsd_plan(n, end_bit);
  int st = temp.reserve(P.table_bytes);
  if (st) return st;
  out->hist = lsd_hist_table(P, temp.p);
  out->row_pitch = (P.nblk + 31) & ~31;
  out->C = P.C; out->nblk = P.nblk; out->mask = (unsigned int)(P.C - 1);
  out->usable = P.nblk <= RS_FUSED_MAX_BLOCKS && P.steps == 8;
  return LSR_OK;
}

int sort_pairs_u32_lsd(unsigned int* key_a, unsigned int* key_b, int* val_a /*nullable: iota*/, int* val_a_buf, int* val_b, size_t n, int end_bit,
                       DevBuf<char>& temp, hipStream_t stream, bool* result_in_b, bool first_hist_done) {
  *result_in_b = false;
  if (n == 0) return LSR_OK;
  if (n > (size_t)INT32_MAX / 2) { set_last_error("lsd sort: too many keys"); return LSR_ERR_INVALID_ARGUMENT; }
  const LsdPlan P = lsd_plan(n, end_bit);
  int st = temp.reserve(P.table_bytes);
  if (st) return st;
  unsigned int* total = reinterpret_cast<unsigned int*>(temp.p);
  unsigned int* blkoff = total + P.C;
  unsigned short* hist = lsd_hist_table(P, temp.p);
  const unsigned int mask = (unsigned int)(P.C - 1);
  const unsigned int* kin = key_a;
  const int* vin = val_a;
  unsigned int* kout = key_b;
  int* vout = val_b;
  const bool fused = P.nblk <= RS_FUSED_MAX_BLOCKS;
  const int row_pitch = (P.nblk + 31) & ~31;   // digit-major histogram rows of the fused form (the padding is never read as a count)

  for (int p = 0; p < P.passes; p++) {
    const int shift = p * P.bits;
#define LSR_RS_HIST(S, T) \
  hipLaunchKernelGGL((rs_hist_kernel<S, T>), dim3(P.nblk), dim3(RS_THREADS), (size_t)P.C * 4, stream, kin, (int)n, shift, mask, P.C, hist, row_pitch)
    if (p == 0 && first_hist_done && fused && P.steps == 8) { /* the producer of the keys has counted the first digit (lsd_first_hist_plan) */ }
    else if (P.steps == 8) LSR_RS_HIST(8, true);    // digit-major rows in both forms
    else LSR_RS_HIST(16, true);
#undef LSR_RS_HIST
    if (!fused) hipLaunchKernelGGL(rs_scan_wave_kernel, dim3((P.C * 64 + 255) / 256), dim3(256), 0, stream, hist, row_pitch, P.nblk, P.C, blkoff, total);
#define LSR_RS_SCATTER(S, F)                                                                                                             \
  hipLaunchKernelGGL((rs_scatter_kernel<S, F>), dim3(P.nblk), dim3(RS_THREADS), (size_t)P.C * 8, stream, kin, vin, (int)n, shift, mask, P.bits, \
                     blkoff, total, hist, row_pitch, P.nblk, P.C, kout, vout)
    if (P.steps == 8) { if (fused) LSR_RS_SCATTER(8, true); else LSR_RS_SCATTER(8, false); }
    else { if (fused) LSR_RS_SCATTER(16, true); else LSR_RS_SCATTER(16, false); }
#undef LSR_RS_SCATTER
    // ping-pong: the next pass reads what this one wrote
    const bool wrote_b = (kout == key_b);
    kin = kout; vin = vout;
    kout = wrote_b ? key_a : key_b;
    vout = wrote_b ? val_a_buf : val_b;
    *result_in_b = wrote_b;
  }
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

int sorted_runs_begin(const unsigned int* keys_sorted, size_t n, int* block_heads, int* block_base, BuildScratch& sc, hipStream_t stream,
                      unsigned int* token_out, const unsigned int* dims_dev, int* total_dev) {
  int st = sc.ensure_mailbox();
  if (st) return st;
  unsigned int token = ++sc.token;
  if (token == 0) token = ++sc.token;
  const int nblocks = (int)((n + RUN_CHUNK - 1) / RUN_CHUNK);
  hipLaunchKernelGGL(rs_heads_count_kernel, dim3(nblocks), dim3(256), 0, stream, keys_sorted, (int)n, block_heads);
  // (one launch for the two — the workgroup that draws the last ticket scans the counts — was measured: 15.3 us against 4.6 + 4.6;
  // 576 atomics on one address and a device-scope fence per workgroup cost more than the launch they save)
  hipLaunchKernelGGL(rs_heads_scan_kernel, dim3(1), dim3(1024), 0, stream, block_heads, nblocks, block_base, sc.d_mb, token, dims_dev, total_dev);
  LSR_HIP(hipGetLastError());
  *token_out = token;
  return LSR_OK;
}

int sorted_runs_count(BuildScratch& sc, hipStream_t stream, unsigned int token, int* n_runs) {
  int st = wait_mailbox_word(&sc.mb.p->value_token, token, stream, sc.wait_mode, "run count");
  if (st) return st;
  *n_runs = sc.mb.p->value;
  return LSR_OK;
}

int sorted_runs_centroids(const unsigned int* keys_sorted, const int* order, size_t n, const int* block_base, unsigned int sentinel,
                          const float* x, const float* y, const float* z, const float* w, float* ox, float* oy, float* oz, float* ow,
                          hipStream_t stream, const unsigned int* sentinel_dev) {
  const int nblocks = (int)((n + RUN_CHUNK - 1) / RUN_CHUNK);
  hipLaunchKernelGGL(rs_centroid_kernel, dim3(nblocks), dim3(256), 0, stream, keys_sorted, order, (int)n, block_base, sentinel, sentinel_dev,
                     x, y, z, w, ox, oy, oz, ow);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

size_t sorted_runs_blocks(size_t n) { return (n + RUN_CHUNK - 1) / RUN_CHUNK; }

int sorted_runs_table(const unsigned int* keys_sorted, size_t n, const int* block_base, unsigned int* run_key, int* run_off, hipStream_t stream) {
  if (n == 0) return LSR_OK;
  const int nblocks = (int)((n + RUN_CHUNK - 1) / RUN_CHUNK);
  hipLaunchKernelGGL(rs_runs_table_kernel, dim3(nblocks), dim3(256), 0, stream, keys_sorted, (int)n, block_base, run_key, run_off);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

int exclusive_scan_i32_lsd(const int* in, int* out, size_t n, DevBuf<char>& temp, hipStream_t stream) {
  if (n == 0) return LSR_OK;
  const size_t nblocks = (n + XS_CHUNK - 1) / XS_CHUNK;
  if (nblocks > (size_t)INT32_MAX / 2) { set_last_error("exclusive scan: too many elements"); return LSR_ERR_INVALID_ARGUMENT; }
  if (nblocks == 1) {
    hipLaunchKernelGGL(xs_single_block_kernel, dim3(1), dim3(XS_THREADS), 0, stream, in, (int)n, out);
    LSR_HIP(hipGetLastError());
    return LSR_OK;
  }
  int st = temp.reserve(nblocks * sizeof(int) + 64);
  if (st) return st;
  int* sums = reinterpret_cast<int*>(temp.p);
  hipLaunchKernelGGL(xs_reduce_kernel, dim3((unsigned)nblocks), dim3(XS_THREADS), 0, stream, in, n, sums);
  hipLaunchKernelGGL(xs_scan_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, (int)nblocks);
  hipLaunchKernelGGL(xs_apply_kernel, dim3((unsigned)nblocks), dim3(XS_THREADS), 0, stream, in, n, sums, out);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

}  // namespace lsr
