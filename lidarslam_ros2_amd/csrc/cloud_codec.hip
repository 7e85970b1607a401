// The clouds on their way into and out of the registration object: PointXYZI records or a PointCloud2
// payload -> SoA planes (+ the frontend's range filter and the bounding-box records in one pass), pcl::VoxelGrid (leaf keys with the
// grid dimensions worked out on the device, stable LSD sort, run heads, float centroids in ascending point index), SoA planes ->
// records / PointCloud2 payload, pcl::transformPointCloud into strided records.  Reference call sites:
// scanmatcher_component.cpp:201-218,279-284,324-329,443-447.  (Split out of ndt.hip in round 6: same code.)
#include "ndt.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "build_kernels.hpp"
#include "grid_device.hpp"
#include "sort.hpp"

namespace lsr {

namespace {

// The same keys with the grid dimensions worked out ON THE DEVICE from the bounding-box records the ingest pass left in device memory
// (pc2_ingest): every workgroup folds the (<= 256) records itself — the arithmetic is voxel_grid_filter's host code, operation for
// operation — and workgroup 0 leaves {sentinel, finite points, VG_FLAG_*, key bits} in dims[0..3] for the kernels behind the sort.
// The host never sees the box: it enqueues key + sort + run heads + centroids without a wait in between.
// dims folded from the records of the ingest pass by ALL 256 threads of a workgroup (one barrier inside)
struct VgDims { int mb[3], dv[3]; unsigned int sentinel, flags, n_finite; int bits; };
__device__ __forceinline__ VgDims vg_fold_dims(const unsigned long long* __restrict__ parts, int nparts, float inv_leaf, int planned_bits) {
  __shared__ float s_mn[4][3], s_mx[4][3];
  __shared__ unsigned int s_cnt[4];
  const int tid = threadIdx.x, w = tid >> 6;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  unsigned int cnt = 0;
  if (tid < nparts) {
    const unsigned long long* P = parts + (size_t)tid * 8;
#pragma unroll
    for (int k = 0; k < 3; k++) { mn[k] = __uint_as_float((unsigned int)P[k]); mx[k] = __uint_as_float((unsigned int)P[3 + k]); }
    cnt = (unsigned int)P[6];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], m, 64)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], m, 64)); }
    cnt += __shfl_xor(cnt, m, 64);
  }
  if ((tid & 63) == 0) {
    for (int k = 0; k < 3; k++) { s_mn[w][k] = mn[k]; s_mx[w][k] = mx[k]; }
    s_cnt[w] = cnt;
  }
  __syncthreads();
  VgDims D;
  D.n_finite = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  for (int k = 0; k < 3; k++) { D.mb[k] = 0; D.dv[k] = 0; }
  D.sentinel = 0u; D.flags = 0u; D.bits = 1;
  if (D.n_finite != 0u) {
    long long vol = 1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float lo = fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k]));
      const float hi = fmaxf(fmaxf(s_mx[0][k], s_mx[1][k]), fmaxf(s_mx[2][k], s_mx[3][k]));
      vol *= (long long)((hi - lo) * inv_leaf) + 1;
      D.mb[k] = (int)floorf(lo * inv_leaf);
      D.dv[k] = (int)floorf(hi * inv_leaf) - D.mb[k] + 1;
    }
    if (vol > (long long)INT32_MAX) D.flags |= VG_FLAG_OVERFLOW;
    D.sentinel = (unsigned int)((long long)D.dv[0] * D.dv[1] * D.dv[2]);
    while (D.bits < 32 && (D.sentinel >> D.bits) != 0u) D.bits++;
    if (D.bits > planned_bits) D.flags |= VG_FLAG_REPLAN;
  }
  return D;
}
__device__ __forceinline__ unsigned int vg_key(const VgDims& D, float inv_leaf, float px, float py, float pz) {
  if (D.flags != 0u || !(isfinite(px) && isfinite(py) && isfinite(pz))) return D.sentinel;
  const int i0 = (int)(floorf(px * inv_leaf) - (float)D.mb[0]);
  const int i1 = (int)(floorf(py * inv_leaf) - (float)D.mb[1]);
  const int i2 = (int)(floorf(pz * inv_leaf) - (float)D.mb[2]);
  return (unsigned int)(i0 + i1 * D.dv[0] + i2 * (D.dv[0] * D.dv[1]));
}

__global__ __launch_bounds__(256) void leaf_key_dims_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ z, int n, float inv_leaf,
                                                            const unsigned long long* __restrict__ parts, int nparts, int planned_bits,
                                                            unsigned int* __restrict__ key, unsigned int* __restrict__ dims) {
  const int tid = threadIdx.x;
  // the point first: its loads are in flight while the records are folded
  const int i = blockIdx.x * 256 + tid;
  const float px = (i < n) ? x[i] : NAN, py = (i < n) ? y[i] : NAN, pz = (i < n) ? z[i] : NAN;
  const VgDims D = vg_fold_dims(parts, nparts, inv_leaf, planned_bits);
  if (blockIdx.x == 0 && tid == 0) { dims[0] = D.sentinel; dims[1] = D.n_finite; dims[2] = D.flags; dims[3] = (unsigned int)D.bits; }
  if (i < n) key[i] = vg_key(D, inv_leaf, px, py, pz);
}

// The same on the sort's 2048-key workgroups, counting the keys' FIRST digit on the way (lsd_first_hist_plan: the table the first pass
// of sort_pairs_u32_lsd reads) — the sort then starts with its scatter: one launch less, and the records are folded by an eighth of
// the workgroups.
__global__ __launch_bounds__(256) void leaf_key_dims_hist_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                 const float* __restrict__ z, int n, float inv_leaf,
                                                                 const unsigned long long* __restrict__ parts, int nparts, int planned_bits,
                                                                 unsigned int* __restrict__ key, unsigned int* __restrict__ dims,
                                                                 unsigned int mask, int C, unsigned short* __restrict__ hist, int row_pitch) {
  extern __shared__ unsigned int s_hist[];  // [C]
  const int tid = threadIdx.x;
  for (int k = tid; k < C; k += 256) s_hist[k] = 0u;
  const int base = blockIdx.x * 2048;
  float px[8], py[8], pz[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = base + j * 256 + tid;
    const bool in = i < n;
    px[j] = in ? x[i] : NAN; py[j] = in ? y[i] : NAN; pz[j] = in ? z[i] : NAN;
  }
  const VgDims D = vg_fold_dims(parts, nparts, inv_leaf, planned_bits);   // its barrier also orders the zeroing above before the counts below
  if (blockIdx.x == 0 && tid == 0) { dims[0] = D.sentinel; dims[1] = D.n_finite; dims[2] = D.flags; dims[3] = (unsigned int)D.bits; }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = base + j * 256 + tid;
    if (i < n) {
      const unsigned int k = vg_key(D, inv_leaf, px[j], py[j], pz[j]);
      key[i] = k;
      atomicAdd(&s_hist[k & mask], 1u);
    }
  }
  __syncthreads();
  for (int k = tid; k < C; k += 256) hist[(size_t)k * row_pitch + blockIdx.x] = (unsigned short)s_hist[k];
}

__global__ __launch_bounds__(256) void deinterleave_kernel(const unsigned char* __restrict__ aos, size_t stride, int n,
                                                           float* __restrict__ x, float* __restrict__ y, float* __restrict__ z) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (((stride & 15) == 0) && ((reinterpret_cast<size_t>(aos) & 15) == 0)) {   // one 16-byte load per record (pcl::PointXYZI: 32-byte stride)
    const float4 q = *reinterpret_cast<const float4*>(aos + (size_t)i * stride);
    x[i] = q.x; y[i] = q.y; z[i] = q.z;
    return;
  }
  const float* p = (const float*)(aos + (size_t)i * stride);
  x[i] = p[0]; y[i] = p[1]; z[i] = p[2];
}

struct DeintMember { const unsigned char* aos; size_t stride; int n; float *x, *y, *z; };
struct DeintGroup { DeintMember m[LSR_GROUP]; };
__global__ __launch_bounds__(256) void deinterleave_group_kernel(const DeintGroup g) {
  const DeintMember& M = g.m[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M.n) return;
  if (((M.stride & 15) == 0) && ((reinterpret_cast<size_t>(M.aos) & 15) == 0)) {   // one 16-byte load per record
    const float4 q = *reinterpret_cast<const float4*>(M.aos + (size_t)i * M.stride);
    M.x[i] = q.x; M.y[i] = q.y; M.z[i] = q.z;
    return;
  }
  const float* p = (const float*)(M.aos + (size_t)i * M.stride);
  M.x[i] = p[0]; M.y[i] = p[1]; M.z[i] = p[2];
}

__global__ __launch_bounds__(256) void transform_strided_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ z, int n, const float* __restrict__ Tdev,
                                                                unsigned char* __restrict__ out, size_t stride) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float px = x[i], py = y[i], pz = z[i];
  float* o = (float*)(out + (size_t)i * stride);
  o[0] = fmaf(Tdev[0], px, fmaf(Tdev[4], py, fmaf(Tdev[8], pz, Tdev[12])));
  o[1] = fmaf(Tdev[1], px, fmaf(Tdev[5], py, fmaf(Tdev[9], pz, Tdev[13])));
  o[2] = fmaf(Tdev[2], px, fmaf(Tdev[6], py, fmaf(Tdev[10], pz, Tdev[14])));
}

}  // namespace

int deinterleave(const void* d_aos, size_t stride_bytes, size_t n, DeviceCloud& out, hipStream_t stream) {
  int st = out.resize(n);
  if (st) return st;
  if (n == 0) return LSR_OK;
  hipLaunchKernelGGL(deinterleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     (const unsigned char*)d_aos, stride_bytes, (int)n, out.x(), out.y(), out.z());
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

int deinterleave_group(const DeinterleaveJob* jobs, int count, hipStream_t stream) {
  int st;
  for (int g0 = 0; g0 < count; g0 += LSR_GROUP) {
    DeintGroup grp;
    std::memset(&grp, 0, sizeof(grp));
    const int ng = std::min(LSR_GROUP, count - g0);
    size_t nmax = 0;
    for (int k = 0; k < ng; k++) {
      const DeinterleaveJob& J = jobs[g0 + k];
      if ((st = J.out->resize(J.n))) return st;
      grp.m[k] = DeintMember{static_cast<const unsigned char*>(J.d_aos), J.stride, (int)J.n, J.out->x(), J.out->y(), J.out->z()};
      nmax = std::max(nmax, J.n);
    }
    if (nmax > 0) hipLaunchKernelGGL(deinterleave_group_kernel, dim3((unsigned)((nmax + 255) / 256), ng), dim3(256), 0, stream, grp);
  }
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

int transform_to_strided(const DeviceCloud& src, const float* d_T16, void* d_out, size_t stride_bytes, hipStream_t stream) {
  if (src.n == 0) return LSR_OK;
  hipLaunchKernelGGL(transform_strided_kernel, dim3((unsigned)((src.n + 255) / 256)), dim3(256), 0, stream, src.x(), src.y(),
                     src.z(), (int)src.n, d_T16, (unsigned char*)d_out, stride_bytes);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

// ---- N1: pcl::VoxelGrid::filter (centroid per occupied leaf, output ordered by leaf index) -------------
// scanmatcher/src/scanmatcher_component.cpp:324-328 (every scan, vg_size_for_input), :266-269, :443-447
// (map side, vg_size_for_map), graph_based_slam/src/graph_based_slam_component.cpp:224-226.
// Same key/sort machinery as K1; one thread per leaf sums its (few) points in ascending point order in FLOAT, as
// pcl::CentroidPoint does (PCL's own order inside a leaf is whatever std::sort leaves: unspecified; ascending index is
// the oracle's choice and this kernel's) — all fields, intensity included (downsample_all_data_).
namespace {
__global__ __launch_bounds__(256) void leaf_centroid_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ z, const float* __restrict__ w /*nullable*/,
                                                            const int* __restrict__ order,
                                                            const unsigned int* __restrict__ run_key, const int* __restrict__ run_off,
                                                            const int* __restrict__ run_cnt, int n_runs, unsigned int sentinel,
                                                            float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                                                            float* __restrict__ ow) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_runs) return;
  if (run_key[r] == sentinel) return;  // the run of non-finite points (always last) is dropped
  const int off = run_off[r], cnt = run_cnt[r];
  // FLOAT accumulators, points in ascending index (stable sort): the very additions pcl::CentroidPoint performs
  // (AccumulatorXYZ / AccumulatorIntensity are float), so the centroid is bit-identical to the CPU restatement
  float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
  for (int j = 0; j < cnt; j++) {
    const int pi = order[off + j];
    sx += x[pi]; sy += y[pi]; sz += z[pi];
    if (w) sw += w[pi];
  }
  const float m = (float)cnt;
  ox[r] = sx / m; oy[r] = sy / m; oz[r] = sz / m;
  if (ow) ow[r] = w ? sw / m : 0.f;
}
}  // namespace

int voxel_grid_filter(const DeviceCloud& cloud, float leaf, DeviceCloud& out, BuildScratch& sc, hipStream_t stream) {
  const int n = (int)cloud.n;
  out.n = 0;
  if (n == 0) return out.resize(0, cloud.has_i);
  int st;
  // A/B switches, read once: LSR_VG_SORT=rocprim — the rocPRIM radix sort + run_length_encode + scan path of rounds 1-4;
  // LSR_VG_DEVICE_DIMS=0 — always work out the grid dimensions on the host
  static const bool use_rocprim = [] { const char* e = getenv("LSR_VG_SORT"); return e && e[0] == 'r'; }();
  static const bool device_dims = [] { const char* e = getenv("LSR_VG_DEVICE_DIMS"); return !(e && e[0] == '0'); }();
  const float inv_leaf = 1.0f / leaf;
  const size_t nrb = sorted_runs_blocks((size_t)n);
  if ((st = sc.words.reserve(32 + 7 * (size_t)n + 2 * nrb + 16))) return st;
  unsigned int* dims_dev = sc.words.p + 16;   // {sentinel, finite points, flags, key bits} of the device-side form
  unsigned int* key_in = sc.words.p + 32;
  unsigned int* key_out = key_in + n;
  int* val_in = (int*)(key_out + n);
  int* val_out = val_in + n;
  unsigned int* run_key = (unsigned int*)(val_out + n);
  int* run_cnt = (int*)(run_key + n);
  int* run_off = run_cnt + n;
  int* d_nruns = run_off + n;
  int* block_heads = d_nruns + 8;
  int* block_base = block_heads + nrb;

  // ---- device-side dimensions: the pass that wrote `cloud` left its bounding-box records in device memory (pc2_ingest) and an earlier
  // call on this scratch says how many key bits such a cloud needs -> key (folds the records itself), sort, run heads and centroids are
  // enqueued back to back; the host waits ONCE, for {runs, finite points, flags}.  A cloud that needs more bits than planned (or whose
  // index space overflows) comes back flagged and takes the host-side form below, which also renews the hint.
  if (!use_rocprim && device_dims && cloud.bbox_enqueued && !cloud.bbox_valid && sc.bbox_parts > 0 && sc.bbox_dev.p && sc.vg_bits_hint > 0 &&
      sc.vg_hint_leaf == leaf) {   // (an object that filters at two leaf sizes in turn — scans and keyframes — stays on the host form)
    const int planned_bits = sc.vg_bits_hint;
    LsdFirstHist fh;
    if ((st = lsd_first_hist_plan((size_t)n, planned_bits, sc.temp, &fh))) return st;
    if (fh.usable)   // the key kernel counts the first digit on the sort's own workgroups: the sort starts with its scatter
      hipLaunchKernelGGL(leaf_key_dims_hist_kernel, dim3(fh.nblk), dim3(256), (size_t)fh.C * 4, stream, cloud.x(), cloud.y(), cloud.z(), n,
                         inv_leaf, sc.bbox_dev.p, sc.bbox_parts, planned_bits, key_in, dims_dev, fh.mask, fh.C, fh.hist, fh.row_pitch);
    else
      hipLaunchKernelGGL(leaf_key_dims_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), n, inv_leaf,
                         sc.bbox_dev.p, sc.bbox_parts, planned_bits, key_in, dims_dev);
    bool in_b = false;
    if ((st = sort_pairs_u32_lsd(key_in, key_out, nullptr, val_in, val_out, (size_t)n, planned_bits, sc.temp, stream, &in_b, fh.usable))) return st;
    const unsigned int* ks = in_b ? key_out : key_in;
    const int* vs = in_b ? val_out : val_in;
    unsigned int token = 0;
    if ((st = sorted_runs_begin(ks, (size_t)n, block_heads, block_base, sc, stream, &token, dims_dev))) return st;
    // the centroids do not wait for the count: the planes are laid out for n runs, the cloud shrinks to what was found
    if ((st = out.resize((size_t)n, cloud.has_i))) return st;
    if ((st = sorted_runs_centroids(ks, vs, (size_t)n, block_base, 0u, cloud.x(), cloud.y(), cloud.z(), cloud.i(), out.x(), out.y(), out.z(),
                                    out.i(), stream, dims_dev))) return st;
    int n_runs = 0;
    if ((st = sorted_runs_count(sc, stream, token, &n_runs))) return st;
    const unsigned int flags = sc.mb.p->vg_flags, n_finite = sc.mb.p->vg_finite;
    if (flags == 0u) {
      sc.vg_form = 2;
      sc.bbox_parts = 0;            // the records are spent
      cloud.bbox_enqueued = false;
      // the plan follows the clouds: up at once (the host form does that), down only when a cloud needs clearly fewer bits — a stream
      // whose index space hovers around a power of two would otherwise be flagged every other scan; an empty cloud says nothing
      const int needed = (int)sc.mb.p->vg_bits;
      if (n_finite > 0u && needed > 0 && needed <= sc.vg_bits_hint - 3) sc.vg_bits_hint = needed;
      return out.shrink((size_t)(n_runs - ((n_finite < (unsigned int)n) ? 1 : 0)));   // minus the sentinel run
    }
    // flagged: the centroid launch may still be writing `out` — the host-side form below runs behind it on the same stream
    out.n = 0;
    sc.vg_form = 3;
  } else {
    sc.vg_form = 1;
  }

  float mn[3], mx[3];
  unsigned int n_finite = 0;
  st = cloud_bbox(cloud, mn, mx, &n_finite, sc, stream);
  if (st) return st;
  if (n_finite == 0) return out.resize(0, cloud.has_i);
  int64_t d[3];
  for (int k = 0; k < 3; k++) d[k] = (int64_t)((mx[k] - mn[k]) * inv_leaf) + 1;
  if (d[0] * d[1] * d[2] > (int64_t)INT32_MAX) {  // PCL: "Leaf size is too small for the input dataset"
    set_last_error("voxel index space exceeds int32: leaf size too small for the cloud extent");
    return LSR_ERR_INDEX_OVERFLOW;
  }
  int min_b[3], div_b[3];
  for (int k = 0; k < 3; k++) {
    min_b[k] = (int)floorf(mn[k] * inv_leaf);
    div_b[k] = (int)floorf(mx[k] * inv_leaf) - min_b[k] + 1;
  }
  const unsigned int sentinel = (unsigned int)((int64_t)div_b[0] * div_b[1] * div_b[2]);  // one past the last leaf index
  sc.vg_bits_hint = bits_for(sentinel);
  sc.vg_hint_leaf = leaf;
  hipLaunchKernelGGL(leaf_key_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), n, inv_leaf,
                     min_b[0], min_b[1], min_b[2], div_b[0], div_b[0] * div_b[1], sentinel, key_in, val_in, (uint4*)nullptr, (size_t)0,
                     (int*)nullptr, (size_t)0, (int*)nullptr);
  if (!use_rocprim) {
    // hand-written stable LSD sort (three passes for a 27-bit leaf index) + run heads + centroids: lsd_sort.hip
    bool in_b = false;
    if ((st = sort_pairs_u32_lsd(key_in, key_out, nullptr, val_in, val_out, (size_t)n, bits_for(sentinel), sc.temp, stream, &in_b))) return st;
    const unsigned int* ks = in_b ? key_out : key_in;
    const int* vs = in_b ? val_out : val_in;
    unsigned int token = 0;
    if ((st = sorted_runs_begin(ks, (size_t)n, block_heads, block_base, sc, stream, &token))) return st;
    // the centroids do not wait for the count either (planes laid out for n runs; shrunk below)
    if ((st = out.resize((size_t)n, cloud.has_i))) return st;
    if ((st = sorted_runs_centroids(ks, vs, (size_t)n, block_base, sentinel, cloud.x(), cloud.y(), cloud.z(), cloud.i(), out.x(), out.y(), out.z(),
                                    out.i(), stream))) return st;
    int n_runs = 0;
    if ((st = sorted_runs_count(sc, stream, token, &n_runs))) return st;   // host mailbox: no D2H copy, no stream sync
    return out.shrink((size_t)(n_runs - ((n_finite < (unsigned int)n) ? 1 : 0)));     // minus the sentinel run
  }
  if ((st = sort_pairs_u32(key_in, key_out, val_in, val_out, n, bits_for(sentinel), sc.temp, stream))) return st;
  if ((st = run_length_encode_u32(key_out, n, run_key, run_cnt, d_nruns, sc.temp, stream))) return st;
  int n_runs = 0;
  if ((st = publish_device_int(d_nruns, sc, stream, &n_runs))) return st;   // host mailbox: no D2H copy, no stream sync
  if ((st = exclusive_scan_i32(run_cnt, run_off, n_runs, sc.temp, stream))) return st;
  const int n_out = n_runs - ((n_finite < (unsigned int)n) ? 1 : 0);  // minus the sentinel run
  if ((st = out.resize(n_out, cloud.has_i))) return st;
  hipLaunchKernelGGL(leaf_centroid_kernel, dim3((n_runs + 255) / 256), dim3(256), 0, stream, cloud.x(), cloud.y(), cloud.z(), cloud.i(),
                     val_out, run_key, run_off, run_cnt, n_runs, sentinel, out.x(), out.y(), out.z(), out.i());
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

// SoA planes -> strided xyz records (device to device)
namespace {
__global__ __launch_bounds__(256) void interleave_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, int n, unsigned char* __restrict__ out,
                                                         size_t stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* o = (float*)(out + (size_t)i * stride);
  o[0] = x[i]; o[1] = y[i]; o[2] = z[i];
}
}  // namespace

int interleave(const DeviceCloud& in, void* d_out, size_t stride_bytes, hipStream_t stream) {
  if (in.n == 0) return LSR_OK;
  hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)((in.n + 255) / 256)), dim3(256), 0, stream, in.x(), in.y(), in.z(),
                     (int)in.n, (unsigned char*)d_out, stride_bytes);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

// ---- N4: sensor_msgs/PointCloud2 <-> SoA planes with arbitrary float32 field offsets --------------------------------
// pcl::fromROSMsg (scanmatcher_component.cpp:201-202) reads x / y / z / intensity wherever the message's fields put them;
// pcl::toROSMsg (:279,284; SubMap.msg:4) writes pcl::PointXYZI's layout.  Offsets are in bytes inside a point_step record.
namespace {
__global__ __launch_bounds__(256) void pc2_write_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                        const float* __restrict__ w, int n, unsigned char* __restrict__ data, int step,
                                                        int ox, int oy, int oz, int oi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned char* rec = data + (size_t)i * step;
  *reinterpret_cast<float*>(rec + ox) = x[i];
  *reinterpret_cast<float*>(rec + oy) = y[i];
  *reinterpret_cast<float*>(rec + oz) = z[i];
  if (oi >= 0) *reinterpret_cast<float*>(rec + oi) = w ? w[i] : 0.f;
}
}  // namespace

namespace {
// payload -> planes, the frontend's range filter and the bounding box of what is left in ONE pass (until round 5 three launches —
// read, range mask, bounding box — and two more trips over the planes).
//   range filter (N4, scanmatcher_component.cpp:210-218): keep p iff scan_min_range < sqrt(x^2 + y^2) < scan_max_range (double
//   arithmetic, as pow(p.x, 2.0) promotes); a rejected point gets a NaN x in the handle's private copy, so the voxel filter that
//   follows drops it exactly like non-finite input.
//   bounding box: the workgroup records go to the host mailbox (cloud_bbox_end folds them as it folds bbox_kernel's) and, the same
//   words, to device memory for voxel_grid_filter's device-side dimensions.
__global__ __launch_bounds__(256) void pc2_ingest_kernel(const unsigned char* __restrict__ data, int step, int ox, int oy, int oz, int oi,
                                                         int n, float* __restrict__ x, float* __restrict__ y, float* __restrict__ z,
                                                         float* __restrict__ w, int do_range, double rmin, double rmax,
                                                         unsigned long long* __restrict__ parts_dev, BuildMailbox* __restrict__ mb,
                                                         unsigned int token) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  unsigned int cnt = 0;
  const int step_pts = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * step_pts) {  // four records per trip in flight
    float p[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * step_pts;
      if (i < n) {
        const unsigned char* rec = data + (size_t)i * step;
        p[u][0] = *reinterpret_cast<const float*>(rec + ox);
        p[u][1] = *reinterpret_cast<const float*>(rec + oy);
        p[u][2] = *reinterpret_cast<const float*>(rec + oz);
        p[u][3] = (w && oi >= 0) ? *reinterpret_cast<const float*>(rec + oi) : 0.f;
      } else {
        p[u][0] = p[u][1] = p[u][2] = NAN; p[u][3] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * step_pts;
      if (i >= n) continue;
      if (do_range) {
        const double px = (double)p[u][0], py = (double)p[u][1];
        const double r = sqrt(px * px + py * py);
        if (!(rmin < r && r < rmax)) p[u][0] = __int_as_float(0x7FC00000);
      }
      x[i] = p[u][0]; y[i] = p[u][1]; z[i] = p[u][2];
      if (w) w[i] = p[u][3];
      if (!(isfinite(p[u][0]) && isfinite(p[u][1]) && isfinite(p[u][2]))) continue;
      cnt++;
#pragma unroll
      for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], p[u][k]); mx[k] = fmaxf(mx[k], p[u][k]); }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], m, 64)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], m, 64)); }
    cnt += __shfl_xor(cnt, m, 64);
  }
  __shared__ float s_mn[4][3], s_mx[4][3];
  __shared__ unsigned int s_cnt[4];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    for (int k = 0; k < 3; k++) { s_mn[wv][k] = mn[k]; s_mx[wv][k] = mx[k]; }
    s_cnt[wv] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < BBOX_GRANULES) {
    const int k = threadIdx.x;
    unsigned int bits;
    if (k < 3) bits = __float_as_uint(fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k])));
    else if (k < 6) bits = __float_as_uint(fmaxf(fmaxf(s_mx[0][k - 3], s_mx[1][k - 3]), fmaxf(s_mx[2][k - 3], s_mx[3][k - 3])));
    else bits = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const unsigned long long g = ((unsigned long long)token << 32) | bits;
    parts_dev[(size_t)blockIdx.x * 8 + k] = g;
    __hip_atomic_store(&mb->part[blockIdx.x].g[k], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace

int pc2_ingest(const void* d_data, int step, int ox, int oy, int oz, int oi, size_t n, bool do_range, double rmin, double rmax,
               DeviceCloud& out, BuildScratch& sc, hipStream_t stream) {
  int st = out.resize(n, oi >= 0);
  if (st) return st;
  if (n == 0) return LSR_OK;
  if ((st = sc.ensure_mailbox())) return st;
  if ((st = sc.bbox_dev.reserve((size_t)BBOX_MAX_PARTS * 8))) return st;
  unsigned int token = ++sc.token;
  if (token == 0) token = ++sc.token;
  const int nb = std::max(1, std::min((int)((n + 1023) / 1024), BBOX_MAX_PARTS));
  hipLaunchKernelGGL(pc2_ingest_kernel, dim3(nb), dim3(256), 0, stream, (const unsigned char*)d_data, step, ox, oy, oz, oi, (int)n, out.x(),
                     out.y(), out.z(), out.i(), do_range ? 1 : 0, rmin, rmax, sc.bbox_dev.p, sc.d_mb, token);
  LSR_HIP(hipGetLastError());
  sc.bbox_parts = nb;
  sc.bbox_token = token;
  out.bbox_enqueued = true;
  return LSR_OK;
}

int pc2_write(const DeviceCloud& in, void* d_data, int step, int ox, int oy, int oz, int oi, hipStream_t stream) {
  if (in.n == 0) return LSR_OK;
  hipLaunchKernelGGL(pc2_write_kernel, dim3((unsigned)((in.n + 255) / 256)), dim3(256), 0, stream, in.x(), in.y(), in.z(), in.i(), (int)in.n,
                     (unsigned char*)d_data, step, ox, oy, oz, oi);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

}  // namespace lsr
