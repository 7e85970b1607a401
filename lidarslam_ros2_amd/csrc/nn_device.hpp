// Device-side pieces of the kd-tree-free NN grid shared by nn.hip (1-NN, fitness) and gicp.hip
// (20-NN covariances, correspondences).  Translation units including this header are compiled with
// -ffp-contract=off so the fp32 distance arithmetic matches the reference bit for bit.
#pragma once
#include "common.hpp"

#ifndef LSR_NN_COUNT   // host-emulation statistics hook (tools/nn_host_emu); compiles to nothing in the product
#define LSR_NN_COUNT(what, n)
#endif

namespace lsr {
namespace nnd {

constexpr int NN_THREADS = 128;
constexpr int FINE_PER_BLOCK = 512;
constexpr int FINE_STRIDE = 513;
constexpr size_t NN_BUCKET_MAX_KEYS = (size_t)16 << 20;  // (coarse, fine) key spaces up to this size are built by bucketing, not sorting
constexpr int NN_MAX_FINE_RINGS = 4;  // fine-cell shells tried before falling back to coarse shells

struct NNGridView {
  float cell, inv_cell;
  int org[3];
  int cdim[3];
  const int* coarse_block;
  const int* block_off;
  const int* fine_start;
  const float4* p;   // cell-sorted points {x, y, z, original index as int bits}
};

__device__ __forceinline__ float dist2_rn(float qx, float qy, float qz, float px, float py, float pz) {
  const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// fp32 point transform in the reference's order: ((m0*x + m1*y) + m2*z) + m3, no contraction.
__device__ __forceinline__ float xform_rn(float a, float b, float c, float d, float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, x), __fmul_rn(b, y)), __fmul_rn(c, z)), d);
}

// ---- top-K collectors --------------------------------------------------------------------------
struct Best1 {
  float d2;
  int idx;
  __device__ __forceinline__ void init() { d2 = INFINITY; idx = -1; }
  __device__ __forceinline__ float worst() const { return d2; }
  __device__ __forceinline__ bool full() const { return idx >= 0; }
  __device__ __forceinline__ void offer(float d, int i) {
    if (d < d2 || (d == d2 && i < idx)) { d2 = d; idx = i; }
  }
  __device__ __forceinline__ void finalize() {}
};

// K-best list kept in LDS, column-major over threads ([slot][thread]) so lanes never collide.  The list is
// UNSORTED while the search runs: the current worst entry (value, index, slot) lives in registers, so rejecting a
// candidate touches no memory, and accepting one overwrites the worst slot and re-scans the k entries with
// independent LDS reads — a wave pays for an insertion whenever ANY of its 64 lanes inserts, so the insertion must
// not be a chain of dependent shifts.  finalize() sorts ascending by (distance, index) once at the end.
#ifdef LSR_HOST_EMU
#define LSR_LDS_PTR(T) T*
#define LSR_NOINLINE
#else
#define LSR_LDS_PTR(T) __attribute__((address_space(3))) T*
#define LSR_NOINLINE __attribute__((noinline))
#endif

struct BestK {
  typedef unsigned long long Key;  // (distance bits << 32) | index: distances are >= 0, so (distance, index) order == integer order
  LSR_LDS_PTR(Key) e;              // LDS base + tid, one ds_read_b64 per slot
  int k, count, wslot;
  Key wkey;                        // worst key in the list (the empty key until the list is full)
  float w;                         // its distance (inf until the list is full)
  static constexpr Key EMPTY = 0x7F800000FFFFFFFFull;  // (inf, -1)
  static __host__ __device__ __forceinline__ size_t lds_bytes(int kk) { return (size_t)kk * NN_THREADS * sizeof(Key); }
  static __device__ __forceinline__ Key key(float d, int i) { return ((Key)(unsigned int)__float_as_int(d) << 32) | (unsigned int)i; }
  __device__ __forceinline__ void init(void* lds_base, int tid, int kk) {
    e = (LSR_LDS_PTR(Key))lds_base + tid; k = kk; count = 0; w = INFINITY; wkey = EMPTY; wslot = 0;
    for (int s = 0; s < kk; s++) e[s * NN_THREADS] = EMPTY;
  }
  __device__ __forceinline__ float dist(int s) const { return __int_as_float((int)(e[s * NN_THREADS] >> 32)); }
  __device__ __forceinline__ int index(int s) const { return (int)(unsigned int)e[s * NN_THREADS]; }
  __device__ __forceinline__ float worst() const { return w; }
  __device__ __forceinline__ bool full() const { return count >= k; }
  template <int K>
  __device__ __forceinline__ void rescan_fixed() {
    Key v[K];
#pragma unroll
    for (int s = 0; s < K; s++) v[s] = e[s * NN_THREADS];   // K independent ds_read_b64 in flight
    Key bw = v[0];
    int bs = 0;
#pragma unroll
    for (int s = 1; s < K; s++) {
      const bool worse = v[s] > bw;
      bw = worse ? v[s] : bw; bs = worse ? s : bs;
    }
    wkey = bw; wslot = bs;
  }
  // One copy of this code per kernel (noinline): the candidate scan is instantiated at several call sites and an
  // inlined, unrolled rescan pushed nn_query past the instruction cache.
  __device__ LSR_NOINLINE void replace_worst(Key nk) {
    LSR_NN_COUNT(offers_taken, 1);
    e[wslot * NN_THREADS] = nk;
    if (k == 20) {   // PCL's default k_correspondences
      rescan_fixed<20>();
    } else {
      Key bw = e[0];
      int bs = 0;
#pragma unroll 4
      for (int s = 1; s < k; s++) {
        const Key v = e[s * NN_THREADS];
        const bool worse = v > bw;
        bw = worse ? v : bw; bs = worse ? s : bs;
      }
      wkey = bw; wslot = bs;
    }
    w = __int_as_float((int)(wkey >> 32));
  }
  __device__ __forceinline__ void offer(float d, int i) {
    if (!(d <= INFINITY)) return;  // NaN never enters
    const Key nk = key(d, i);
    if (count < k) {                // filling: slot `count`; the last fill goes through replace_worst to set the worst
      wslot = count;
      if (++count == k) replace_worst(nk);
      else e[wslot * NN_THREADS] = nk;
      return;
    }
    if (!(nk < wkey)) return;
    replace_worst(nk);
  }
  // ascending (distance, index) order in slots 0..count-1 (selection sort: every pass is a run of independent reads)
  __device__ LSR_NOINLINE void finalize() {
    for (int a = 0; a + 1 < count; a++) {
      const Key av = e[a * NN_THREADS];
      Key bk = av;
      int bs = a;
#pragma unroll 4
      for (int s = a + 1; s < count; s++) {
        const Key v = e[s * NN_THREADS];
        const bool better = v < bk;
        bk = better ? v : bk; bs = better ? s : bs;
      }
      if (bs != a) {
        e[bs * NN_THREADS] = av;
        e[a * NN_THREADS] = bk;
      }
    }
  }
};

// Cells [x0, x1] of an x-row that can still hold a point closer than the current worst: `rem2` = worst d^2 minus the
// row's squared y/z gap (>= 0 for a row that was not pruned).  A point p of the row beats the worst only if
// |p.x - qx| <= sqrt(rem2); cells are floorf(p.x * inv_cell) - org, a monotone map, so clipping the row to the cells of
// qx -+ reach (reach padded against the rounding of the subtraction) loses no such point.
__device__ __forceinline__ void row_clip_x(const NNGridView& G, float qx, float rem2, int& x0, int& x1) {
  const float reach = sqrtf(fmaxf(rem2, 0.f)) * 1.0001f + 4.0e-6f * (fabsf(qx) + G.cell);
  const float lo = floorf((qx - reach) * G.inv_cell), hi = floorf((qx + reach) * G.inv_cell);
  if (lo > -1.0e9f) x0 = max(x0, (int)lo - G.org[0]);
  if (hi < 1.0e9f) x1 = min(x1, (int)hi - G.org[0]);
}

// Candidates [beg, end) of the cell-sorted arrays.  A wave walks its 64 queries' ranges in lock step and every
// batch of coordinate loads costs one full memory round trip (one wave per SIMD: nothing else hides it), so the
// points are fetched NN_SCAN_BATCH at a time, one 16-byte load each ({x, y, z, original index}: no dependent index
// load for the (distance, index) tie-break).
constexpr int NN_SCAN_BATCH = 8;
template <typename Coll>
__device__ __forceinline__ void scan_range(const NNGridView& G, int beg, int end, float qx, float qy, float qz, Coll& c,
                                           int self_skip) {
  LSR_NN_COUNT(ranges, 1);
  LSR_NN_COUNT(candidates, end - beg);
  for (int s = beg; s < end; s += NN_SCAN_BATCH) {
    float4 P[NN_SCAN_BATCH];
#pragma unroll
    for (int u = 0; u < NN_SCAN_BATCH; u++) P[u] = G.p[min(s + u, end - 1)];
#pragma unroll
    for (int u = 0; u < NN_SCAN_BATCH; u++) {
      if (s + u < end) {
        const float d = dist2_rn(qx, qy, qz, P[u].x, P[u].y, P[u].z);
        const int oi = __float_as_int(P[u].w);
        if (!(d > c.worst()) && oi != self_skip) c.offer(d, oi);
      }
    }
  }
}

// Exact search for one query.  fine_rings: half-width of the first fine-cell block.
// ring_cap < 0 (default): the search always completes (fine shells, then coarse shells) and true is returned.
// ring_cap >= fine_rings: only fine shells 0..ring_cap are walked; false is returned when that was not enough to
// prove the result — the caller hands such a query to the wave-cooperative search (coop_knn) instead of letting
// one lane drag its whole wave through hundreds of dependent cell probes.
template <typename Coll>
__device__ bool nn_query(const NNGridView& G, float qx, float qy, float qz, int fine_rings, float max_d2, Coll& c,
                         int self_skip, int ring_cap = -1) {
  if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return true;
  const float fxf = floorf(qx * G.inv_cell), fyf = floorf(qy * G.inv_cell), fzf = floorf(qz * G.inv_cell);
  if (!(fabsf(fxf) < 1.0e9f && fabsf(fyf) < 1.0e9f && fabsf(fzf) < 1.0e9f)) return true;
  const int fq[3] = {(int)fxf - G.org[0], (int)fyf - G.org[1], (int)fzf - G.org[2]};  // fine coords rel. to origin
  const float q[3] = {qx, qy, qz};
  const int fdim[3] = {G.cdim[0] * 8, G.cdim[1] * 8, G.cdim[2] * 8};

  // ---- phase 1: fine-cell shells 0..NN_MAX_FINE_RINGS around the query; after shell R every point outside
  //      the (2R+1)^3 block is at least `lo` away, so the search stops as soon as the k-th best is closer.
  //      (`fine_rings` = shells scanned before the first bound test: 1 for 1-NN, 2 for 20-NN.)
  bool any_fine = true;
  for (int k = 0; k < 3; k++)
    if (fq[k] + NN_MAX_FINE_RINGS < 0 || fq[k] - NN_MAX_FINE_RINGS >= fdim[k]) any_fine = false;
  fine_rings = min(max(fine_rings, 0), NN_MAX_FINE_RINGS);
  const int last_ring = (ring_cap >= 0) ? max(fine_rings, min(ring_cap, NN_MAX_FINE_RINGS)) : NN_MAX_FINE_RINGS;
  if (ring_cap >= 0 && !any_fine) return false;
  if (any_fine) {
    // Walked as x-ROWS of fine cells: consecutive fine cells of one coarse block are contiguous in the sorted arrays,
    // so a row of 2r+1 cells is one or two ranges (one coarse-map load + two fine-table loads each) instead of
    // 2r+1 dependent probes.  Shell by shell, NEAREST FIRST (full rows on the y/z faces, the two end cells of the
    // inner rows): a list that fills with near candidates rejects most later ones in registers, and insertions are
    // what a wave pays for most (scanning the whole block in raster order doubled this kernel's time).
    for (int r = 0; r <= last_ring; r++) {
      const bool whole_block = false;
      for (int dz = -r; dz <= r; dz++) {
        const int z = fq[2] + dz;
        if (z < 0 || z >= fdim[2]) continue;
        for (int dy = -r; dy <= r; dy++) {
          const int y = fq[1] + dy;
          if (y < 0 || y >= fdim[1]) continue;
          const bool full_row = whole_block || (abs(dz) == r) || (abs(dy) == r);
          // Row pruning: every point of this row is at least `gyz` away in the y/z plane; once the list is full and its
          // worst entry is strictly closer, the row cannot change the answer (ties included: the bound is strict).
          // The slack covers the rounding of floorf(p * inv_cell) against (index * cell) for cells that are not powers of 2.
          float gyz2 = 0.f;
          if (r > 0) {
            const float ylo = (float)(y + G.org[1]) * G.cell, zlo = (float)(z + G.org[2]) * G.cell;
            const float gy = fmaxf(fmaxf(ylo - q[1], q[1] - (ylo + G.cell)) - 2.0e-6f * (fabsf(q[1]) + G.cell), 0.f);
            const float gz = fmaxf(fmaxf(zlo - q[2], q[2] - (zlo + G.cell)) - 2.0e-6f * (fabsf(q[2]) + G.cell), 0.f);
            gyz2 = (gy * gy + gz * gz) * 0.9999f;
            if ((c.full() && gyz2 > c.worst()) || gyz2 > max_d2) { LSR_NN_COUNT(rows_pruned, 1); continue; }
          }
          const int cbase = G.cdim[0] * ((y >> 3) + G.cdim[1] * (z >> 3));
          const int fyz = ((y & 7) << 3) | ((z & 7) << 6);
          for (int part = 0; part < (full_row ? 1 : 2); part++) {
            const int xs = full_row ? fq[0] - r : (part ? fq[0] + r : fq[0] - r);
            int x0 = max(xs, 0), x1 = min(full_row ? fq[0] + r : xs, fdim[0] - 1);
            if (r > 0 && c.full()) row_clip_x(G, qx, c.worst() - gyz2, x0, x1);   // cells of the row that can still matter
            for (int cx = x0 >> 3; cx <= (x1 >> 3); cx++) {   // empty when x0 > x1
              LSR_NN_COUNT(fine_probes, 1);
              const int blk = G.coarse_block[cbase + cx];
              if (blk < 0) continue;
              const int xa = max(x0, cx * 8) & 7, xb = min(x1, cx * 8 + 7) & 7;
              const int* fs = G.fine_start + (size_t)blk * FINE_STRIDE + fyz;
              const int beg = fs[xa], end = fs[xb + 1];
              if (beg < end) scan_range(G, beg, end, qx, qy, qz, c, self_skip);
            }
          }
        }
      }
      float lo = INFINITY;
      for (int k = 0; k < 3; k++) {
        const float base = (float)(fq[k] + G.org[k]) * G.cell;
        lo = fminf(lo, fminf(q[k] - (base - (float)r * G.cell), (base + (float)(r + 1) * G.cell) - q[k]));
      }
      lo = fmaxf(lo, 0.f);
      const float lo2 = lo * lo * 0.9999f;
      if (r >= fine_rings && ((c.full() && c.worst() <= lo2) || lo2 > max_d2)) return true;
    }
  }
  if (ring_cap >= 0) return false;
  fine_rings = NN_MAX_FINE_RINGS;  // what phase 2 must not offer again
  LSR_NN_COUNT(phase2_queries, 1);

  // ---- phase 2: coarse shells with box-distance pruning
  const float C = G.cell * 8.f;
  int cq[3];
  for (int k = 0; k < 3; k++) cq[k] = (fq[k] >= 0) ? (fq[k] >> 3) : -(((-fq[k]) + 7) >> 3);
  int rmax = 0;
  for (int k = 0; k < 3; k++) rmax = max(rmax, max(cq[k], G.cdim[k] - 1 - cq[k]));
  for (int r = 0; r <= rmax; r++) {
    for (int dz = -r; dz <= r; dz++) {
      const int z = cq[2] + dz;
      if (z < 0 || z >= G.cdim[2]) continue;
      for (int dy = -r; dy <= r; dy++) {
        const int y = cq[1] + dy;
        if (y < 0 || y >= G.cdim[1]) continue;
        const bool shell_yz = (abs(dz) == r) || (abs(dy) == r);
        const int step = shell_yz ? 1 : max(1, 2 * r);
        for (int dx = -r; dx <= r; dx += step) {
          const int x = cq[0] + dx;
          if (x < 0 || x >= G.cdim[0]) continue;
          const int blk = G.coarse_block[x + G.cdim[0] * (y + G.cdim[1] * z)];
          if (blk < 0) continue;
          // squared distance from q to the coarse cell's box
          float bd2 = 0.f;
          const int cc[3] = {x, y, z};
          for (int k = 0; k < 3; k++) {
            const float b0 = (float)(cc[k] * 8 + G.org[k]) * G.cell, b1 = b0 + C;
            const float dd = fmaxf(fmaxf(b0 - q[k], q[k] - b1), 0.f);
            bd2 += dd * dd;
          }
          bd2 *= 0.9999f;
          if ((c.full() && bd2 > c.worst()) || bd2 > max_d2) continue;
          LSR_NN_COUNT(coarse_blocks, 1);
          // fine cells already visited in phase 1 must not be offered twice (a k-best list would keep
          // the duplicate): walk the block cell by cell where it overlaps the phase-1 box
          bool overlap = any_fine;
          for (int k = 0; k < 3; k++)
            if (cc[k] * 8 + 7 < fq[k] - fine_rings || cc[k] * 8 > fq[k] + fine_rings) overlap = false;
          if (!overlap) {
            scan_range(G, G.block_off[blk], G.block_off[blk + 1], qx, qy, qz, c, self_skip);
          } else {
            const int* fs = G.fine_start + (size_t)blk * FINE_STRIDE;
            for (int f = 0; f < FINE_PER_BLOCK; f++) {
              const int beg = fs[f], end = fs[f + 1];
              if (beg == end) continue;
              const int gx = x * 8 + (f & 7), gy = y * 8 + ((f >> 3) & 7), gz = z * 8 + (f >> 6);
              const bool seen = (abs(gx - fq[0]) <= fine_rings) && (abs(gy - fq[1]) <= fine_rings) && (abs(gz - fq[2]) <= fine_rings);
              if (!seen) scan_range(G, beg, end, qx, qy, qz, c, self_skip);
            }
          }
        }
      }
    }
    float loc = INFINITY;
    for (int k = 0; k < 3; k++) {
      const float b0 = (float)((cq[k] - r) * 8 + G.org[k]) * G.cell;
      const float b1 = (float)((cq[k] + r + 1) * 8 + G.org[k]) * G.cell;
      loc = fminf(loc, fminf(q[k] - b0, b1 - q[k]));
    }
    loc = fmaxf(loc, 0.f);
    const float loc2 = loc * loc * 0.9999f;
    if ((c.full() && c.worst() <= loc2) || loc2 > max_d2) return true;
  }
  return true;
}

// Fine cells [lo, hi] (per axis, clamped to the grid) that the ball of squared radius d2 around q touches: every point p with
// |p - q|^2 <= d2 has its cell in that box (the cell index floorf(p * inv_cell) is a monotone map of the coordinate, and the
// reach is padded against the rounding of q -+ reach).  false when some axis spans more than max_cells cells or the
// numbers are out of range — the caller then uses the general search.  An empty box (lo > hi) is a valid answer.
__device__ __forceinline__ bool ball_cell_range(const NNGridView& G, const float* q, float d2, int max_cells, int* lo, int* hi) {
  const float root = sqrtf(d2) * 1.0001f;
  for (int k = 0; k < 3; k++) {
    const float reach = root + 4.0e-6f * (fabsf(q[k]) + G.cell);
    const float fl = floorf((q[k] - reach) * G.inv_cell), fh = floorf((q[k] + reach) * G.inv_cell);
    if (!(fabsf(fl) < 1.0e9f && fabsf(fh) < 1.0e9f)) return false;
    lo[k] = max((int)fl - G.org[k], 0);
    hi[k] = min((int)fh - G.org[k], G.cdim[k] * 8 - 1);
    if (hi[k] - lo[k] > max_cells - 1) return false;
  }
  return true;
}

#ifndef LSR_HOST_EMU
// ---- sixteen lanes per query (fitness search, GICP correspondences): a query group is one DPP row, so its scans, broadcasts and
// reductions are row DPP moves — no LDS crossbar (ds_bpermute), no lane-address arithmetic.
template <int CTRL, bool ZERO_FILL>
__device__ __forceinline__ int row16_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, ZERO_FILL); }
template <int CTRL>
__device__ __forceinline__ float row16_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
constexpr int DPP_ROW_SHR = 0x110, DPP_ROW_ROR = 0x120, DPP_ROW_NEWBCAST = 0x150;

// the group's best (distance, index) in every one of its sixteen lanes: rotations by 8, 4, 2, 1 (the minimum of a total order
// does not depend on the order in which it meets its operands)
__device__ __forceinline__ void row16_best(float& bd, int& bi) {
#define LSR_ROW_BEST_STEP(N)                                              \
  {                                                                       \
    const float od = row16_f<DPP_ROW_ROR + N>(bd);                        \
    const int oi = row16_i<DPP_ROW_ROR + N, false>(bi);                   \
    if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }           \
  }
  LSR_ROW_BEST_STEP(8) LSR_ROW_BEST_STEP(4) LSR_ROW_BEST_STEP(2) LSR_ROW_BEST_STEP(1)
#undef LSR_ROW_BEST_STEP
}

// Every point of ONE fine cell fq (inside the grid) offered to the group's best, sixteen at a time; then row16_best.
__device__ __forceinline__ void scan_cell_group16(const NNGridView& G, const float* q, const int* fq, const int gl, float& bd, int& bi) {
  const int blk = G.coarse_block[G.cdim[0] * ((fq[1] >> 3) + G.cdim[1] * (fq[2] >> 3)) + (fq[0] >> 3)];
  if (blk >= 0) {
    const int* fs = G.fine_start + (size_t)blk * FINE_STRIDE + (((fq[1] & 7) << 3) | ((fq[2] & 7) << 6)) + (fq[0] & 7);
    const int end = fs[1];
    for (int f = fs[0] + gl; f < end; f += 32) {   // two points per lane and trip, both loads in flight
      const bool in1 = f + 16 < end;
      const float4 p0 = G.p[f];
      float4 p1 = p0;
      if (in1) p1 = G.p[f + 16];
      {
        const float d = dist2_rn(q[0], q[1], q[2], p0.x, p0.y, p0.z);
        const int oi = __float_as_int(p0.w);
        if (d < bd || (d == bd && oi < bi)) { bd = d; bi = oi; }
      }
      if (in1) {
        const float d = dist2_rn(q[0], q[1], q[2], p1.x, p1.y, p1.z);
        const int oi = __float_as_int(p1.w);
        if (d < bd || (d == bd && oi < bi)) { bd = d; bi = oi; }
      }
    }
  }
  row16_best(bd, bi);
}

// Every point of the fine cells [lo, hi] (<= 8 per axis, clamped to the grid by the caller) offered to the group's best: one lane
// per (row, coarse segment) — an x-range of <= 8 cells touches at most two coarse cells, and inside one it is contiguous in
// memory —, the occupied slots read one after the other by the sixteen lanes; then row16_best.  Called by all sixteen lanes.
// `skip` (optional): a fine cell the caller has offered already (the query's own cell, scan_cell_group16) — its points are left out.
__device__ __forceinline__ void scan_cells_group16(const NNGridView& G, const float* q, const int* lo, const int* hi, const int gl, float& bd, int& bi,
                                                   const int* skip = nullptr) {
  const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;   // >= 1 unless the box misses the grid (then no slot is valid)
  const int n_slots = (ny > 0 && nz > 0 && hi[0] >= lo[0]) ? ny * nz * 2 : 0;
  const unsigned int ny_magic = 65536u / (unsigned int)max(ny, 1) + 1u;
  for (int s0 = 0; s0 < n_slots; s0 += 16) {
    const int slot = s0 + gl;
    int beg = 0, len = 0;
    int beg2 = 0, len2 = 0;   // the part of the slot's row behind a skipped cell
    if (slot < n_slots) {
      const int cseg = slot & 1, row = slot >> 1;
      const int dz = (int)(((unsigned int)row * ny_magic) >> 16);   // row / ny (row < 128, ny <= 8)
      const int y = lo[1] + (row - dz * ny), z = lo[2] + dz;
      const int cx = (lo[0] >> 3) + cseg;
      if (cx <= (hi[0] >> 3)) {
        const int blk = G.coarse_block[G.cdim[0] * ((y >> 3) + G.cdim[1] * (z >> 3)) + cx];
        if (blk >= 0) {
          const int xa = max(lo[0], cx * 8) & 7, xb = min(hi[0], cx * 8 + 7) & 7;
          const int* fs = G.fine_start + (size_t)blk * FINE_STRIDE + (((y & 7) << 3) | ((z & 7) << 6));
          beg = fs[xa];
          len = fs[xb + 1] - beg;
          if (skip && y == skip[1] && z == skip[2] && cx == (skip[0] >> 3)) {
            const int sx = skip[0] & 7;
            if (sx >= xa && sx <= xb) {
              len = fs[sx] - beg;
              beg2 = fs[sx + 1];
              len2 = fs[xb + 1] - beg2;
            }
          }
        }
      }
    }
    // the occupied slots one after the other (a ball of a few centimetres touches one to four rows: two or three slots hold points),
    // each read by the sixteen lanes two points per lane and trip, both loads in flight: a trip starts with its loads.  (Rounds 3-5
    // laid the group's candidates end to end and found the slot of every flat position by a four-step search over the row's
    // exclusive offsets: eight dependent LDS-crossbar shuffles before a trip's loads could be issued; 182 -> 156 us for the searches
    // of a share of 8.)  The minimum over a total order does not care in which order it meets the candidates.
    const unsigned long long wave_mask = __ballot((len | len2) > 0);
    unsigned int todo = (unsigned int)(wave_mask >> (__lane_id() & 48)) & 0xFFFFu;   // this group's sixteen bits
    auto offer = [&](const int sb, const int sn) {
      for (int f = gl; f < sn; f += 32) {
        const bool in1 = f + 16 < sn;
        const float4 p0 = G.p[sb + f];
        float4 p1 = p0;
        if (in1) p1 = G.p[sb + f + 16];
        {
          const float d = dist2_rn(q[0], q[1], q[2], p0.x, p0.y, p0.z);
          const int oi = __float_as_int(p0.w);
          if (d < bd || (d == bd && oi < bi)) { bd = d; bi = oi; }
        }
        if (in1) {
          const float d = dist2_rn(q[0], q[1], q[2], p1.x, p1.y, p1.z);
          const int oi = __float_as_int(p1.w);
          if (d < bd || (d == bd && oi < bi)) { bd = d; bi = oi; }
        }
      }
    };
    while (todo) {
      const int k = __builtin_ctz(todo);
      todo &= todo - 1;
      offer(__shfl(beg, k, 16), __shfl(len, k, 16));
      if (skip) offer(__shfl(beg2, k, 16), __shfl(len2, k, 16));
    }
  }
  row16_best(bd, bi);
}
#endif  // LSR_HOST_EMU

// ---- one (row, coarse segment) of a fine shell, for the wave-cooperative searches below (pure: also compiled by the host
// emulation, tests/test_nn_host_emu_cpu.py checks that the slots of a shell cover its cells exactly once and that pruning
// and clipping never drop a point that could matter)
struct FineSeg {
  int beg, len;
};

// Slot `slot` of shell r: slot = ((row * 2 + part) * 2 + cseg); row = (dz + r) * (2r+1) + (dy + r); part = the two end
// cells of an inner row (full rows use part 0 only); cseg = the (up to two) coarse cells an x-range of <= 9 cells touches.
__device__ __forceinline__ FineSeg fine_segment(const NNGridView& G, const int* fq, const int* fdim, const float* q, int r, int slot,
                                                bool full, float worst, float max_d2) {
  FineSeg out;
  out.beg = 0;
  out.len = 0;
  const int w = 2 * r + 1;
  const unsigned magic = (r == 0) ? 65536u : (r == 1) ? 21846u : (r == 2) ? 13108u : (r == 3) ? 9363u : 7282u;  // ceil(2^16 / w)
  const int cseg = slot & 1, part = (slot >> 1) & 1, row = slot >> 2;
  if (row >= w * w) return out;
  const int rz = (int)(((unsigned)row * magic) >> 16);
  const int dz = rz - r, dy = row - rz * w - r;
  const int z = fq[2] + dz, y = fq[1] + dy;
  if (z < 0 || z >= fdim[2] || y < 0 || y >= fdim[1]) return out;
  const bool full_row = (abs(dz) == r) || (abs(dy) == r);
  if (full_row && part) return out;
  float gyz2 = 0.f;
  if (r > 0) {   // row pruning + x clip, exactly as in nn_query
    const float ylo = (float)(y + G.org[1]) * G.cell, zlo = (float)(z + G.org[2]) * G.cell;
    const float gy = fmaxf(fmaxf(ylo - q[1], q[1] - (ylo + G.cell)) - 2.0e-6f * (fabsf(q[1]) + G.cell), 0.f);
    const float gz = fmaxf(fmaxf(zlo - q[2], q[2] - (zlo + G.cell)) - 2.0e-6f * (fabsf(q[2]) + G.cell), 0.f);
    gyz2 = (gy * gy + gz * gz) * 0.9999f;
    if ((full && gyz2 > worst) || gyz2 > max_d2) return out;
  }
  const int xs = full_row ? fq[0] - r : (part ? fq[0] + r : fq[0] - r);
  int x0 = max(xs, 0), x1 = min(full_row ? fq[0] + r : xs, fdim[0] - 1);
  if (r > 0 && full) row_clip_x(G, q[0], worst - gyz2, x0, x1);
  if (x0 > x1) return out;
  const int cx = (x0 >> 3) + cseg;
  if (cx > (x1 >> 3)) return out;
  const int blk = G.coarse_block[G.cdim[0] * ((y >> 3) + G.cdim[1] * (z >> 3)) + cx];
  if (blk < 0) return out;
  const int xa = max(x0, cx * 8) & 7, xb = min(x1, cx * 8 + 7) & 7;
  const int* fs = G.fine_start + (size_t)blk * FINE_STRIDE + (((y & 7) << 3) | ((z & 7) << 6));
  out.beg = fs[xa];
  out.len = fs[xb + 1] - out.beg;
  return out;
}

#ifndef LSR_HOST_EMU   // (wave intrinsics: not part of the host emulation in tools/nn_host_emu)
// ---- exact 1-NN on FOUR lanes per query -------------------------------------------------------------------------
// Lanes 4p..4p+3 of a wave carry the same query (`sub` = lane & 3).  The search of one query is a chain of dependent
// loads (coarse map -> fine table -> candidates) and a scan is fewer waves than the chip has SIMDs, so the time of the
// per-thread walk is the length of that chain; here the own cell's candidates are split in quarters and the rows of
// the later shells are dealt round-robin to the four lanes, which merge their bests (same (distance, index) order)
// after every shell, so the shell bound is tested on the merged result and all four leave together.
// Same candidates, same fp32 distances, a total order => the answer of nn_query.  Fine shells 0..ring_cap only:
// false = not proven (the caller defers the query to coop_knn).  `c` may come in seeded (identically on the four lanes).
__device__ __forceinline__ void best1_merge_quad(Best1& c) {
#pragma unroll
  for (int m = 1; m <= 2; m <<= 1) {
    const float od = __shfl_xor(c.d2, m, 64);
    const int oi = __shfl_xor(c.idx, m, 64);
    if (oi >= 0 && (od < c.d2 || (od == c.d2 && oi < c.idx) || c.idx < 0)) { c.d2 = od; c.idx = oi; }
  }
}

__device__ bool nn1_query_quad(const NNGridView& G, float qx, float qy, float qz, int fine_rings, float max_d2, Best1& c,
                               int ring_cap, int sub) {
  if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return true;
  const float fxf = floorf(qx * G.inv_cell), fyf = floorf(qy * G.inv_cell), fzf = floorf(qz * G.inv_cell);
  if (!(fabsf(fxf) < 1.0e9f && fabsf(fyf) < 1.0e9f && fabsf(fzf) < 1.0e9f)) return true;
  const int fq[3] = {(int)fxf - G.org[0], (int)fyf - G.org[1], (int)fzf - G.org[2]};
  const float q[3] = {qx, qy, qz};
  const int fdim[3] = {G.cdim[0] * 8, G.cdim[1] * 8, G.cdim[2] * 8};
  for (int k = 0; k < 3; k++)
    if (fq[k] + NN_MAX_FINE_RINGS < 0 || fq[k] - NN_MAX_FINE_RINGS >= fdim[k]) return false;
  fine_rings = min(max(fine_rings, 0), NN_MAX_FINE_RINGS);
  const int last_ring = max(fine_rings, min(ring_cap, NN_MAX_FINE_RINGS));
  for (int r = 0; r <= last_ring; r++) {
    int dealt = 0;   // rows of this shell seen so far (identical on the four lanes)
    for (int dz = -r; dz <= r; dz++) {
      const int z = fq[2] + dz;
      if (z < 0 || z >= fdim[2]) continue;
      for (int dy = -r; dy <= r; dy++) {
        const int y = fq[1] + dy;
        if (y < 0 || y >= fdim[1]) continue;
        const bool full_row = (abs(dz) == r) || (abs(dy) == r);
        for (int part = 0; part < (full_row ? 1 : 2); part++) {
          const bool take = (r == 0) || ((dealt++ & 3) == sub);
          if (!take) continue;
          float gyz2 = 0.f;
          if (r > 0) {   // row pruning, as in nn_query
            const float ylo = (float)(y + G.org[1]) * G.cell, zlo = (float)(z + G.org[2]) * G.cell;
            const float gy = fmaxf(fmaxf(ylo - q[1], q[1] - (ylo + G.cell)) - 2.0e-6f * (fabsf(q[1]) + G.cell), 0.f);
            const float gz = fmaxf(fmaxf(zlo - q[2], q[2] - (zlo + G.cell)) - 2.0e-6f * (fabsf(q[2]) + G.cell), 0.f);
            gyz2 = (gy * gy + gz * gz) * 0.9999f;
            if ((c.full() && gyz2 > c.worst()) || gyz2 > max_d2) continue;
          }
          const int cbase = G.cdim[0] * ((y >> 3) + G.cdim[1] * (z >> 3));
          const int fyz = ((y & 7) << 3) | ((z & 7) << 6);
          const int xs = full_row ? fq[0] - r : (part ? fq[0] + r : fq[0] - r);
          int x0 = max(xs, 0), x1 = min(full_row ? fq[0] + r : xs, fdim[0] - 1);
          if (r > 0 && c.full()) row_clip_x(G, qx, c.worst() - gyz2, x0, x1);
          for (int cx = x0 >> 3; cx <= (x1 >> 3); cx++) {
            const int blk = G.coarse_block[cbase + cx];
            if (blk < 0) continue;
            const int xa = max(x0, cx * 8) & 7, xb = min(x1, cx * 8 + 7) & 7;
            const int* fs = G.fine_start + (size_t)blk * FINE_STRIDE + fyz;
            int beg = fs[xa], end = fs[xb + 1];
            if (r == 0) {   // the own cell: a quarter of its candidates per lane
              const int quarter = (end - beg + 3) >> 2;
              beg += sub * quarter;
              end = min(end, beg + quarter);
            }
            if (beg < end) scan_range(G, beg, end, qx, qy, qz, c, -1);
          }
        }
      }
    }
    best1_merge_quad(c);
    float lo = INFINITY;
    for (int k = 0; k < 3; k++) {
      const float base = (float)(fq[k] + G.org[k]) * G.cell;
      lo = fminf(lo, fminf(q[k] - (base - (float)r * G.cell), (base + (float)(r + 1) * G.cell) - q[k]));
    }
    lo = fmaxf(lo, 0.f);
    const float lo2 = lo * lo * 0.9999f;
    if (r >= fine_rings && ((c.full() && c.worst() <= lo2) || lo2 > max_d2)) return true;
  }
  return false;
}

// ---- wave-cooperative exact k-NN (k <= 64) for the queries the per-thread walk gave up on ---------------------
// One 64-lane wave per query.  The k best (distance, index) pairs live one per lane, sorted ascending in lanes
// 0..k-1 (empty = (inf, INT_MAX)); coarse cells are visited shell by shell with the same box-distance pruning and
// termination bound as nn_query's phase 2, but every cell's points are read 64 at a time, coalesced, and a
// candidate enters the list with one ballot + one shuffle.  Same (distance, index) ordering as BestK, same fp32
// distance arithmetic: the result is identical to nn_query's.
struct CoopList {
  float d;
  int i;
};

// bound2: an upper bound on the k-th best distance known to the caller (the k-th entry of the list the fine shells left behind),
// INFINITY if none: coarse cells and candidates beyond it are skipped before the list has filled up again — the walk starts
// afresh, but it no longer reads whole 8 x 8 x 8 blocks that cannot matter.
__device__ __forceinline__ void coop_knn(const NNGridView& G, float qx, float qy, float qz, int k, float max_d2, int self_skip,
                                         CoopList& mine, const float bound2 = INFINITY) {
  const int lane = threadIdx.x & 63;
  mine.d = INFINITY;
  mine.i = INT_MAX;
  if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return;
  const float fxf = floorf(qx * G.inv_cell), fyf = floorf(qy * G.inv_cell), fzf = floorf(qz * G.inv_cell);
  if (!(fabsf(fxf) < 1.0e9f && fabsf(fyf) < 1.0e9f && fabsf(fzf) < 1.0e9f)) return;
  const int fq[3] = {(int)fxf - G.org[0], (int)fyf - G.org[1], (int)fzf - G.org[2]};
  const float q[3] = {qx, qy, qz};
  float worst = INFINITY;
  int worst_i = INT_MAX;
  const float C = G.cell * 8.f;
  int cq[3];
  for (int a = 0; a < 3; a++) cq[a] = (fq[a] >= 0) ? (fq[a] >> 3) : -(((-fq[a]) + 7) >> 3);
  int rmax = 0;
  for (int a = 0; a < 3; a++) rmax = max(rmax, max(cq[a], G.cdim[a] - 1 - cq[a]));
  for (int r = 0; r <= rmax; r++) {
    for (int dz = -r; dz <= r; dz++) {
      const int z = cq[2] + dz;
      if (z < 0 || z >= G.cdim[2]) continue;
      for (int dy = -r; dy <= r; dy++) {
        const int y = cq[1] + dy;
        if (y < 0 || y >= G.cdim[1]) continue;
        const bool shell_yz = (abs(dz) == r) || (abs(dy) == r);
        const int step = shell_yz ? 1 : max(1, 2 * r);
        for (int dx = -r; dx <= r; dx += step) {
          const int x = cq[0] + dx;
          if (x < 0 || x >= G.cdim[0]) continue;
          const int blk = G.coarse_block[x + G.cdim[0] * (y + G.cdim[1] * z)];
          if (blk < 0) continue;
          float bd2 = 0.f;
          const int cc[3] = {x, y, z};
          for (int a = 0; a < 3; a++) {
            const float b0 = (float)(cc[a] * 8 + G.org[a]) * G.cell, b1 = b0 + C;
            const float dd = fmaxf(fmaxf(b0 - q[a], q[a] - b1), 0.f);
            bd2 += dd * dd;
          }
          bd2 *= 0.9999f;
          if ((worst_i != INT_MAX && bd2 > worst) || bd2 > max_d2 || bd2 > bound2) continue;   // list full <=> last slot filled
          const int beg = G.block_off[blk], end = G.block_off[blk + 1];
          for (int s0 = beg; s0 < end; s0 += 64) {
            const int s = s0 + lane;
            const bool valid = s < end;
            const int sl = valid ? s : end - 1;
            const float4 pt = G.p[sl];
            const float d = dist2_rn(qx, qy, qz, pt.x, pt.y, pt.z);
            const int oi = __float_as_int(pt.w);
            bool qual = valid && (oi != self_skip) && (d <= bound2) && (d < worst || (d == worst && oi < worst_i));
            unsigned long long mask = __ballot(qual);
            while (mask) {
              const int src = __ffsll((long long)mask) - 1;
              mask &= mask - 1;
              const float dn = __shfl(d, src, 64);
              const int in = __shfl(oi, src, 64);
              if (!(dn < worst || (dn == worst && in < worst_i))) continue;
              const unsigned long long before = __ballot(lane < k && (mine.d < dn || (mine.d == dn && mine.i < in)));
              const int pos = __popcll(before);
              const float up_d = __shfl_up(mine.d, 1, 64);
              const int up_i = __shfl_up(mine.i, 1, 64);
              if (lane < k) {
                if (lane > pos) { mine.d = up_d; mine.i = up_i; }
                else if (lane == pos) { mine.d = dn; mine.i = in; }
              }
              worst = __shfl(mine.d, k - 1, 64);
              worst_i = __shfl(mine.i, k - 1, 64);
            }
          }
        }
      }
    }
    float loc = INFINITY;
    for (int a = 0; a < 3; a++) {
      const float b0 = (float)((cq[a] - r) * 8 + G.org[a]) * G.cell;
      const float b1 = (float)((cq[a] + r + 1) * 8 + G.org[a]) * G.cell;
      loc = fminf(loc, fminf(q[a] - b0, b1 - q[a]));
    }
    loc = fmaxf(loc, 0.f);
    const float loc2 = loc * loc * 0.9999f;
    if ((worst_i != INT_MAX && worst <= loc2) || loc2 > max_d2 || loc2 > bound2) return;
  }
}
// ---- wave-cooperative exact k-NN over the FINE grid (k <= 64; k == 1 has its own list-free form) ------------------
// One 64-lane wave per query.  The per-thread walk (nn_query) is a chain of dependent loads that one lane drags its wave
// through, and a scan is fewer such waves than the chip has SIMDs: latency bound.  Here the 64 lanes share ONE query:
// every (row, coarse segment) of a fine shell is probed by its own lane (one round of coarse-map loads, one of
// fine-table loads for the whole shell), the candidate ranges are laid end to end with a wave prefix sum and read 64 at a
// time, coalesced; a shell is a handful of round trips whatever its number of cells, and a scan is 30 000 short waves:
// throughput bound.  Same candidates, same fp32 distances, same (distance, index) order as nn_query => same answer.
// Shells 0..NN_MAX_FINE_RINGS with the same bound test; returns false when that did not prove the result (the caller
// finishes with coop_knn over the coarse cells, which starts afresh).
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// ---- 64 (distance, index) keys, one per lane, through compare-exchange networks (round 4) --------------------------------------
// key = (distance bits << 32) | index: distances are >= 0, so the integer order IS the (distance, index) order every search
// form ranks candidates by; COOP_EMPTY = (inf, INT_MAX) sorts last.
typedef unsigned long long CoopKey;
constexpr CoopKey COOP_EMPTY = 0x7F8000007FFFFFFFull;
__device__ __forceinline__ CoopKey coop_key(float d, int i) { return ((CoopKey)(unsigned int)__float_as_int(d) << 32) | (unsigned int)i; }
__device__ __forceinline__ CoopKey coop_key_xor(CoopKey v, int m) {
  const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)v, m, 64), hi = (unsigned int)__shfl_xor((int)(unsigned int)(v >> 32), m, 64);
  return ((CoopKey)hi << 32) | lo;
}
// one compare-exchange with the lane `stride` away: the lower lane of the pair keeps the smaller key if `ascending`
__device__ __forceinline__ CoopKey coop_cmpx(CoopKey v, int lane, int stride, bool ascending) {
  const CoopKey o = coop_key_xor(v, stride);
  const bool lower = (lane & stride) == 0;
  const bool take_min = lower == ascending;
  return take_min ? (o < v ? o : v) : (o > v ? o : v);
}
// the wave's 64 keys sorted DESCENDING over the lanes (bitonic sort: 21 compare-exchanges)
__device__ __forceinline__ CoopKey coop_sort_desc(CoopKey v, int lane) {
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
    const bool asc = (size == 64) ? false : ((lane & size) != 0);   // blocks alternate so that every pair of blocks is bitonic; the last pass is the final order
#pragma unroll
    for (int stride = size >> 1; stride >= 1; stride >>= 1) v = coop_cmpx(v, lane, stride, asc);
  }
  return v;
}
// a BITONIC sequence of 64 keys sorted ascending (6 compare-exchanges)
__device__ __forceinline__ CoopKey coop_merge_asc(CoopKey v, int lane) {
#pragma unroll
  for (int stride = 32; stride >= 1; stride >>= 1) v = coop_cmpx(v, lane, stride, true);
  return v;
}
constexpr int COOP_MERGE_MIN = 12;   // qualifying candidates of a 64-chunk from which the networks beat one insertion per candidate

// k-th best list one entry per lane (sorted ascending in lanes 0..k-1), as in coop_knn.  K1: `mine` is the wave's best,
// identical on every lane on return (seed it identically on every lane, or with (INFINITY, INT_MAX)).
template <bool K1>
__device__ __forceinline__ bool coop_fine_knn(const NNGridView& G, float qx, float qy, float qz, int k, int fine_rings, float max_d2,
                                              int self_skip, CoopList& mine) {
  const int lane = threadIdx.x & 63;
  if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return true;
  const float fxf = floorf(qx * G.inv_cell), fyf = floorf(qy * G.inv_cell), fzf = floorf(qz * G.inv_cell);
  if (!(fabsf(fxf) < 1.0e9f && fabsf(fyf) < 1.0e9f && fabsf(fzf) < 1.0e9f)) return true;
  const int fq[3] = {(int)fxf - G.org[0], (int)fyf - G.org[1], (int)fzf - G.org[2]};
  const float q[3] = {qx, qy, qz};
  const int fdim[3] = {G.cdim[0] * 8, G.cdim[1] * 8, G.cdim[2] * 8};
  for (int a = 0; a < 3; a++)
    if (fq[a] + NN_MAX_FINE_RINGS < 0 || fq[a] - NN_MAX_FINE_RINGS >= fdim[a]) return false;
  fine_rings = min(max(fine_rings, 0), NN_MAX_FINE_RINGS);
  // wave-uniform view of the list's worst entry
  float worst = K1 ? mine.d : INFINITY;
  int worst_i = K1 ? mine.i : INT_MAX;
  for (int r = 0; r <= NN_MAX_FINE_RINGS; r++) {
    const int w = 2 * r + 1;
    const int n_slots = 4 * w * w;
    for (int s0 = 0; s0 < n_slots; s0 += 64) {
      const bool full = worst_i != INT_MAX;
      const FineSeg seg = fine_segment(G, fq, fdim, q, r, s0 + lane, full, worst, max_d2);
      const int incl = wave_incl_scan_i(seg.len, lane);
      const int excl = incl - seg.len;
      const int total = __shfl(incl, 63, 64);
      for (int t0 = 0; t0 < total; t0 += 64) {
        const int f = t0 + lane;
        const bool valid = f < total;
        // the segment that holds flat position f: the largest lane whose exclusive offset is <= f
        int lo = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
          const int cand = lo + step;
          const int o = __shfl(excl, cand, 64);
          if (o <= f) lo = cand;
        }
        const int sb = __shfl(seg.beg, lo, 64), so = __shfl(excl, lo, 64);
        const float4 pt = G.p[valid ? sb + (f - so) : 0];
        const float d = dist2_rn(qx, qy, qz, pt.x, pt.y, pt.z);
        const int oi = __float_as_int(pt.w);
        if (K1) {
          if (valid && oi != self_skip && (d < mine.d || (d == mine.d && oi < mine.i))) { mine.d = d; mine.i = oi; }
        } else {
          const bool qual = valid && (oi != self_skip) && (d < worst || (d == worst && oi < worst_i));
          unsigned long long mask = __ballot(qual);
          if (__popcll(mask) >= COOP_MERGE_MIN) {
            // many candidates at once (the first chunk of shell 1 meets an almost empty list: ~60 of 64 qualify, and one
            // insertion costs a ballot, a popcount and four cross-lane moves): sort the chunk descending, take the lane-wise
            // minimum with the ascending list — the 64 smallest of the 128, as a bitonic sequence — and merge it ascending.
            // Same total order, so the same k entries in the same lanes as 60 insertions would leave.
            const CoopKey c = coop_sort_desc(qual ? coop_key(d, oi) : COOP_EMPTY, lane);
            const CoopKey l = (lane < k) ? coop_key(mine.d, mine.i) : COOP_EMPTY;
            CoopKey m = coop_merge_asc(c < l ? c : l, lane);
            if (lane >= k) m = COOP_EMPTY;
            mine.d = __int_as_float((int)(unsigned int)(m >> 32));
            mine.i = (int)(unsigned int)m;
            worst = __shfl(mine.d, k - 1, 64);
            worst_i = __shfl(mine.i, k - 1, 64);
            mask = 0ull;
          }
          while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float dn = __shfl(d, src, 64);
            const int in = __shfl(oi, src, 64);
            if (!(dn < worst || (dn == worst && in < worst_i))) continue;
            const unsigned long long before = __ballot(lane < k && (mine.d < dn || (mine.d == dn && mine.i < in)));
            const int pos = __popcll(before);
            const float up_d = __shfl_up(mine.d, 1, 64);
            const int up_i = __shfl_up(mine.i, 1, 64);
            if (lane < k) {
              if (lane > pos) { mine.d = up_d; mine.i = up_i; }
              else if (lane == pos) { mine.d = dn; mine.i = in; }
            }
            worst = __shfl(mine.d, k - 1, 64);
            worst_i = __shfl(mine.i, k - 1, 64);
          }
        }
      }
      if (K1) {   // merge the lanes' bests: every lane continues with the wave's best (its distance prunes the next chunk)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
          const float od = __shfl_xor(mine.d, m, 64);
          const int oi = __shfl_xor(mine.i, m, 64);
          if (od < mine.d || (od == mine.d && oi < mine.i)) { mine.d = od; mine.i = oi; }
        }
        worst = mine.d;
        worst_i = mine.i;
      }
    }
    float lo = INFINITY;
    for (int a = 0; a < 3; a++) {
      const float base = (float)(fq[a] + G.org[a]) * G.cell;
      lo = fminf(lo, fminf(q[a] - (base - (float)r * G.cell), (base + (float)(r + 1) * G.cell) - q[a]));
    }
    lo = fmaxf(lo, 0.f);
    const float lo2 = lo * lo * 0.9999f;
    if (r >= fine_rings && ((worst_i != INT_MAX && worst <= lo2) || lo2 > max_d2)) return true;
  }
  return false;
}

// Exact k-NN of one query by one wave: fine shells first, coarse cells when they do not prove the result.
template <bool K1>
__device__ __forceinline__ void coop_search(const NNGridView& G, float qx, float qy, float qz, int k, int fine_rings, float max_d2,
                                            int self_skip, CoopList& mine) {
  const CoopList seed = mine;
  if (coop_fine_knn<K1>(G, qx, qy, qz, k, fine_rings, max_d2, self_skip, mine)) return;
  // what the fine shells found bounds the answer: the k-th entry of their list (k-NN), their best (1-NN)
  const float kth_d = __shfl(mine.d, K1 ? 0 : k - 1, 64);
  const int kth_i = __shfl(mine.i, K1 ? 0 : k - 1, 64);
  const float bound2 = (kth_i != INT_MAX) ? kth_d : INFINITY;
  coop_knn(G, qx, qy, qz, k, max_d2, self_skip, mine, bound2);   // starts afresh (it visits every cell itself)
  if (K1) {
    // coop_knn keeps the list in lane 0..k-1; hand the best (and a seed that beats it) to every lane
    float d = __shfl(mine.d, 0, 64);
    int i = __shfl(mine.i, 0, 64);
    if (seed.i != INT_MAX && (seed.d < d || (seed.d == d && seed.i < i))) { d = seed.d; i = seed.i; }
    mine.d = d;
    mine.i = i;
  }
}

#endif  // LSR_HOST_EMU


inline NNGridView make_view(const HashGridDev& g) {
  NNGridView v;
  v.cell = g.cell;
  v.inv_cell = 1.0f / g.cell;
  for (int k = 0; k < 3; k++) { v.org[k] = g.org[k]; v.cdim[k] = g.cdim[k]; }
  v.coarse_block = g.coarse_block.p;
  v.block_off = g.block_off.p;
  v.fine_start = g.fine_start.p;
  v.p = g.packed.p;
  return v;
}

}  // namespace nnd
}  // namespace lsr
