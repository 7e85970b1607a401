// GICP for gfx950 (replaces pclomp::GeneralizedIterativeClosestPoint).
//   K5  gicp_knn_wave_kernel      exact 20-NN of every source (target) point, one wave per point (nn_device.hpp: coop_search)
//       gicp_cov_from_nbr_kernel  covariance of the neighbours + 3x3 symmetric eigen-decomposition + U diag(1,1,eps) U^T
//   K6  gicp_corr_ball_kernel     1-NN of (transformation_ * guess * src) in the target, seeded by the previous outer iteration's
//                                 neighbour: the cells of the ball of that radius are all it reads (16 lanes per point)
//       gicp_corr_search_kernel   the first outer iteration, and the points the seeded kernel defers: one wave per point
//       gicp_corr_pairs_kernel    M_i = (R C1_i R^T + C2_j)^-1 in fp64, packed pair records for K7
//   K7  gicp_step_kernel          one Gauss-Newton step per launch: consumes the partial rows of the previous step (fixed-order
//                                 sum, gradient test, 6x6 solve on a wave, state update, the reference's outer bookkeeping),
//                                 then accumulates 28 fp64 sums (cost, 6-gradient, 21-Hessian) at the new state
// The whole outer loop runs on the device; the host (GicpChain) keeps launches queued and polls a mailbox.
// gicp_cov_kernel / gicp_cov_coop_kernel / gicp_corr_kernel / gicp_gn_kernel / gicp_update_kernel are the per-thread and
// unfused forms of rounds 1-2: kept as independent cross-checks behind LSR_NN_COOP=0 / LSR_GICP_FUSED=0 / LSR_GICP_BALL=0
// (tests/test_gicp_gpu.py holds the production kernels to them bit for bit).
// The reference minimises the same cost with BFGS; north_star asks for Gauss-Newton accumulation, so
// the inner solver here is GN with the reference's stopping rule (|grad| < 1e-2 or max_inner
// iterations).  Same cost and correspondences => same minimiser; the oracle carries both solvers.
#include <chrono>
#include <thread>

#include "handle.hpp"
#include "nn_device.hpp"

namespace lsr {

using namespace nnd;

// A/B switch (env LSR_GICP_FUSED=0 selects the accumulate + update launch pairs); read once.
static bool gicp_fused_enabled() {
  static const bool on = [] { const char* e = getenv("LSR_GICP_FUSED"); return !(e && e[0] == '0'); }();
  return on && nn_coop_enabled();
}

// Largest ball (fine cells per axis, 3..8: an x-range of <= 8 cells touches at most two coarse cells) the seeded correspondence
// kernel searches itself; larger ones go to the general search, which costs a wave per point and a long chain of dependent
// probes: measured on cfg 3 (setInputSource + align) 3 cells 0.834 ms, 5 cells 0.734, 7 cells 0.739, 8 cells 0.750.
// env LSR_GICP_BALL_CELLS, read once.
static int gicp_ball_cells() {
  static const int v = [] { const char* e = getenv("LSR_GICP_BALL_CELLS"); const int c = e ? atoi(e) : 5; return c < 3 ? 3 : c > 8 ? 8 : c; }();
  return v;
}

namespace {

constexpr int GN_THREADS = 256;
// The pair count of a correspondence pass is summed over 64 counters a cache line apart (never reset within an align: the step
// that adopts a count remembers the total): 7 500 waves adding to ONE address took 100 us, 13 ns per atomic.
constexpr int GICP_COUNT_SHARDS = 64, GICP_SHARD_STRIDE = 16;
constexpr int GN_NRED = 28;  // [0] cost, [1..6] J^T M r, [7..27] upper triangle of J^T M J

struct PairRec {      // one candidate correspondence, written by K6, streamed by K7
  float q[3];         // target point
  int valid;
  double M[6];        // symmetric Mahalanobis matrix: 00 01 02 11 12 22
};

struct GnState {
  double x[6];        // (t, phi, theta, psi): R = Rz(psi) Ry(theta) Rx(phi)
  float T[12];        // row-major 3x4 of applyState(I, x), fp32 as the reference composes it
  float pad0[4];
  double dR[27];      // dR/dphi, dR/dtheta, dR/dpsi (row-major 3x3 each)
  double f;           // mean cost at the last evaluated x
  double gnorm;
  int m;              // number of valid pairs
  int inner_iter;
  int inner_done;
  int max_inner;
};

// Outer loop of GeneralizedIterativeClosestPoint::computeTransformation, kept ON THE DEVICE: the update
// launch that ends an inner loop also does the outer bookkeeping (transformation_ from x, the delta stop rule, the next
// Mahalanobis rotation) and tells the following launches what to do through `phase`:
//   even = the correspondence pass of outer iteration phase/2 has to run; odd = its inner Gauss-Newton loop is running.
// Every launch of the chain (correspondence / accumulation / update, enqueued by the host in a fixed pattern) reads the
// phase at its head and exits when it has nothing to do; only the single-workgroup update launch ever writes it.
struct OuterState {
  float trans[16], prev[16], G[16];   // transformation_, previous_transformation_, guess (column-major)
  double rot_eps, trans_eps;
  double last_cost;
  int nr_iterations, max_iterations, converged, outer_done;
  int phase;       // see above
  int corr_mark;   // = the (even) phase whose correspondence pass has run (written by every workgroup of that pass)
  int last_cnt, gn_steps;
  unsigned int token;
  int pad;
};

struct IterBlock {    // uploaded once per align
  GnState st;
  float T16[16];      // transformation_ (column-major) the correspondence pass moves the points by
  double Rm[9];       // rotation of transformation_ * guess, fp64
  OuterState out;
  int count;          // pairs found by the correspondence pass
  int have_partials;  // fused chain: the previous step left partial rows at the current x (to be consumed by the next step)
  int count_base;     // sharded pair counters (GICP_COUNT_SHARDS): their total when the previous outer iteration adopted it
  int pad;
};

// f6 = {cos a, sin a, cos b, sin b, cos c, sin c} of the FLOAT angles, d6 = the same of the double angles (x[3..5])
__host__ __device__ inline void gn_apply_state_trig(GnState& S, const float* f6, const double* d6) {
  const float ca = f6[0], sa = f6[1], cb = f6[2], sb = f6[3], cc = f6[4], sc = f6[5];
  // A = Rz * Ry ; R = A * Rx   (float products, as Eigen::AngleAxisf chains them)
  const float a00 = cc * cb, a01 = -sc, a02 = cc * sb;
  const float a10 = sc * cb, a11 = cc, a12 = sc * sb;
  const float a20 = -sb, a21 = 0.f, a22 = cb;
  S.T[0] = a00; S.T[1] = a01 * ca + a02 * sa; S.T[2] = -a01 * sa + a02 * ca;
  S.T[4] = a10; S.T[5] = a11 * ca + a12 * sa; S.T[6] = -a11 * sa + a12 * ca;
  S.T[8] = a20; S.T[9] = a21 * ca + a22 * sa; S.T[10] = -a21 * sa + a22 * ca;
  S.T[3] = (float)S.x[0]; S.T[7] = (float)S.x[1]; S.T[11] = (float)S.x[2];
  const double cphi = d6[0], sphi = d6[1], ct = d6[2], st = d6[3], cpsi = d6[4], spsi = d6[5];
  double* A = S.dR;
  double* B = S.dR + 9;
  double* Cc = S.dR + 18;
  A[0] = 0; A[1] = sphi * spsi + cphi * cpsi * st;   A[2] = cphi * spsi - cpsi * sphi * st;
  A[3] = 0; A[4] = -cpsi * sphi + cphi * spsi * st;  A[5] = -cphi * cpsi - sphi * spsi * st;
  A[6] = 0; A[7] = cphi * ct;                        A[8] = -ct * sphi;
  B[0] = -cpsi * st; B[1] = cpsi * ct * sphi; B[2] = cphi * cpsi * ct;
  B[3] = -spsi * st; B[4] = ct * sphi * spsi; B[5] = cphi * ct * spsi;
  B[6] = -ct;        B[7] = -sphi * st;       B[8] = -cphi * st;
  Cc[0] = -ct * spsi; Cc[1] = -cphi * cpsi - sphi * spsi * st; Cc[2] = cpsi * sphi - cphi * spsi * st;
  Cc[3] = cpsi * ct;  Cc[4] = -cphi * spsi + cpsi * sphi * st; Cc[5] = sphi * spsi + cphi * cpsi * st;
  Cc[6] = 0; Cc[7] = 0; Cc[8] = 0;
}

__host__ __device__ inline void gn_apply_state(GnState& S) {
  const float a = (float)S.x[3], b = (float)S.x[4], c = (float)S.x[5];
  const float f6[6] = {cosf(a), sinf(a), cosf(b), sinf(b), cosf(c), sinf(c)};
  const double d6[6] = {cos(S.x[3]), sin(S.x[3]), cos(S.x[4]), sin(S.x[4]), cos(S.x[5]), sin(S.x[5])};
  gn_apply_state_trig(S, f6, d6);
}

// ---- small fp64 3x3 helpers ----------------------------------------------------------------------
__device__ void sym3_eigen_k5(const double* Ain, double* w, double* V) {
  double a00 = Ain[0], a01 = Ain[1], a02 = Ain[2], a11 = Ain[4], a12 = Ain[5], a22 = Ain[8];
  double q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 32; sweep++) {
    const double off = a01 * a01 + a02 * a02 + a12 * a12;
    const double diag = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
#define LSR_ROT(app, aqq, apq, arp, arq, cp, cq)                                            \
  if (apq != 0.0) {                                                                         \
    const double theta = (aqq - app) / (2.0 * apq);                                         \
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;                                    \
    const double npp = app - t * apq, nqq = aqq + t * apq;                                  \
    const double nrp = c * arp - s * arq, nrq = s * arp + c * arq;                          \
    app = npp; aqq = nqq; apq = 0.0; arp = nrp; arq = nrq;                                  \
    for (int k = 0; k < 3; k++) {                                                           \
      const double qp = q[k * 3 + cp], qq = q[k * 3 + cq];                                  \
      q[k * 3 + cp] = c * qp - s * qq;                                                      \
      q[k * 3 + cq] = s * qp + c * qq;                                                      \
    }                                                                                       \
  }
    LSR_ROT(a00, a11, a01, a02, a12, 0, 1)
    LSR_ROT(a00, a22, a02, a01, a12, 0, 2)
    LSR_ROT(a11, a22, a12, a01, a02, 1, 2)
#undef LSR_ROT
  }
  w[0] = a00; w[1] = a11; w[2] = a22;
  for (int k = 0; k < 9; k++) V[k] = q[k];
}

// Regularised covariance of one point from the sums over its k neighbours (computeCovariances)
struct CovSums {
  double mean[3] = {0, 0, 0}, s00 = 0, s10 = 0, s11 = 0, s20 = 0, s21 = 0, s22 = 0;
  __device__ __forceinline__ void add(float x, float y, float z) {
    mean[0] += (double)x; mean[1] += (double)y; mean[2] += (double)z;
    // FLOAT products accumulated in double — the reference's `cov(0,0) += pt.x*pt.x`
    s00 += (double)(x * x);
    s10 += (double)(y * x); s11 += (double)(y * y);
    s20 += (double)(z * x); s21 += (double)(z * y); s22 += (double)(z * z);
  }
};

__device__ void cov_finish(CovSums& S, int k, double gicp_eps, double* __restrict__ out) {
  const double kk = (double)k;
  double* mean = S.mean;
  mean[0] /= kk; mean[1] /= kk; mean[2] /= kk;
  double C[9];
  C[0] = S.s00 / kk - mean[0] * mean[0];
  C[3] = S.s10 / kk - mean[1] * mean[0]; C[4] = S.s11 / kk - mean[1] * mean[1];
  C[6] = S.s20 / kk - mean[2] * mean[0]; C[7] = S.s21 / kk - mean[2] * mean[1]; C[8] = S.s22 / kk - mean[2] * mean[2];
  C[1] = C[3]; C[2] = C[6]; C[5] = C[7];
  if (gicp_eps < 0) {  // inspection (lsr_gicp_covariances which = 2 / 3): the sample covariance before regularisation
    for (int a = 0; a < 9; a++) out[a] = C[a];
    return;
  }
  double w[3], V[9];
  sym3_eigen_k5(C, w, V);
  // singular values of a symmetric matrix are |eigenvalues|: the smallest one is replaced by eps
  int small = 0;
  if (fabs(w[1]) < fabs(w[small])) small = 1;
  if (fabs(w[2]) < fabs(w[small])) small = 2;
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double s = 0;
      for (int col = 0; col < 3; col++) s += ((col == small) ? gicp_eps : 1.0) * V[a * 3 + col] * V[b * 3 + col];
      out[a * 3 + b] = s;
    }
}

// K5: exact k-NN inside the cloud's own grid + covariance regularisation (computeCovariances).
// One thread per point walks fine shells 0..ring_cap; a point whose k-th neighbour is not proven by then goes to
// `work_list` and is finished by gicp_cov_coop_kernel (one wave per point) — otherwise the rare sparse points
// would drag every wave through hundreds of dependent cell probes (that tail was ~90 % of this kernel's time).
__global__ __launch_bounds__(NN_THREADS) void gicp_cov_kernel(NNGridView G, const float* __restrict__ px, const float* __restrict__ py,
                                                              const float* __restrict__ pz, int n, int k, double gicp_eps,
                                                              int fine_rings, int ring_cap, int spread, int* __restrict__ work_count,
                                                              int* __restrict__ work_list, double* __restrict__ cov) {
  extern __shared__ unsigned char smem[];
  // `spread` (1, 2, 4): only every spread-th lane carries a point.  A wave walks the union of its lanes' paths, and a
  // 30k-point scan is fewer waves than the chip has SIMDs — thinner waves finish sooner and idle SIMDs take the rest.
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / spread;
  BestK c;
  c.init(smem, threadIdx.x, k);
  if (i >= n || (t % spread) != 0) return;
  if (!nn_query(G, px[i], py[i], pz[i], fine_rings, INFINITY, c, -1, ring_cap)) {
    work_list[atomicAdd(work_count, 1)] = i;
    return;
  }
  c.finalize();
  CovSums S;
  for (int j = 0; j < k; j++) {
    const int o = c.index(j);
    if (o < 0) continue;  // cloud smaller than k (rejected on the host); keeps the kernel safe
    S.add(px[o], py[o], pz[o]);
  }
  cov_finish(S, k, gicp_eps, cov + (size_t)i * 9);
}

// K5 tail: one wave per deferred point (k <= 64).  Same neighbours, same summation order as the per-thread kernel.
__global__ __launch_bounds__(256) void gicp_cov_coop_kernel(NNGridView G, const float* __restrict__ px, const float* __restrict__ py,
                                                            const float* __restrict__ pz, int k, double gicp_eps,
                                                            const int* __restrict__ work_count, const int* __restrict__ work_list,
                                                            double* __restrict__ cov) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int n_work = *work_count;
  for (int w = wave; w < n_work; w += n_waves) {
    const int i = work_list[w];
    CoopList mine;
    coop_knn(G, px[i], py[i], pz[i], k, INFINITY, -1, mine);
    CovSums S;
    for (int j = 0; j < k; j++) {
      const int o = __shfl(mine.i, j, 64);
      if (o == INT_MAX) continue;
      S.add(px[o], py[o], pz[o]);
    }
    if (lane == 0) cov_finish(S, k, gicp_eps, cov + (size_t)i * 9);
  }
}

// K5, wave-cooperative form (k <= 64): one wave per point finds its k neighbours over the fine grid (coop_search) and
// stores their indices, nearest first; gicp_cov_from_nbr_kernel then sums and regularises one point per thread — the same
// neighbours in the same summation order as the per-thread kernel, so the covariances are bit-identical to it.
__global__ __launch_bounds__(256) void gicp_knn_wave_kernel(NNGridView G, const float* __restrict__ px, const float* __restrict__ py,
                                                            const float* __restrict__ pz, int n, int k, int* __restrict__ nbr) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (int i = wave; i < n; i += n_waves) {
    CoopList mine;
    mine.d = INFINITY;
    mine.i = INT_MAX;
    coop_search<false>(G, px[i], py[i], pz[i], k, 1, INFINITY, -1, mine);   // bound tested from shell 1 on: dense regions never read shell 2
    if (lane < k) nbr[(size_t)i * k + lane] = (mine.i == INT_MAX) ? -1 : mine.i;
  }
}

__global__ __launch_bounds__(256) void gicp_cov_from_nbr_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                                                const float* __restrict__ pz, int n, int k, double gicp_eps,
                                                                const int* __restrict__ nbr, double* __restrict__ cov) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  CovSums S;
  for (int j = 0; j < k; j++) {
    const int o = nbr[(size_t)i * k + j];
    if (o < 0) continue;  // cloud smaller than k (rejected on the host); keeps the kernel safe
    S.add(px[o], py[o], pz[o]);
  }
  cov_finish(S, k, gicp_eps, cov + (size_t)i * 9);
}

// The start of an align in ONE launch (round 6; a copy, this kernel and a fill until then: two launches and their gaps less on a chain of
// short launches): the iteration block travels in the kernel arguments and is written to its device home by workgroup 0, the pair
// counters / work-list head are zeroed, and output = guess * input as below.
static_assert(sizeof(IterBlock) <= 3072, "IterBlock travels in the kernel arguments (4 KiB limit)");
__global__ __launch_bounds__(256) void gicp_begin_align_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n,
                                                               const IterBlock blk, IterBlock* __restrict__ d_blk, int* __restrict__ zero_words, int n_zero,
                                                               float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz) {
  if (blockIdx.x == 0) {
    const unsigned int* src = reinterpret_cast<const unsigned int*>(&blk);
    unsigned int* dst = reinterpret_cast<unsigned int*>(d_blk);
    for (int k = threadIdx.x; k < (int)(sizeof(IterBlock) / 4); k += 256) dst[k] = src[k];
    for (int k = threadIdx.x; k < n_zero; k += 256) zero_words[k] = 0;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* G16 = blk.out.G;
  const float a = x[i], b = y[i], c = z[i];
  ox[i] = xform_rn(G16[0], G16[4], G16[8], G16[12], a, b, c);
  oy[i] = xform_rn(G16[1], G16[5], G16[9], G16[13], a, b, c);
  oz[i] = xform_rn(G16[2], G16[6], G16[10], G16[14], a, b, c);
}

// output = guess * input (fp32, reference order of operations)
__global__ __launch_bounds__(256) void gicp_apply_guess_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ z, int n, const float* __restrict__ G16,
                                                               float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[i], b = y[i], c = z[i];
  ox[i] = xform_rn(G16[0], G16[4], G16[8], G16[12], a, b, c);
  oy[i] = xform_rn(G16[1], G16[5], G16[9], G16[13], a, b, c);
  oz[i] = xform_rn(G16[2], G16[6], G16[10], G16[14], a, b, c);
}

// K6: correspondences + Mahalanobis matrices.  Rm = rotation of (transformation_ * guess) in double.
__global__ __launch_bounds__(NN_THREADS) void gicp_corr_kernel(NNGridView G, const float* __restrict__ ox, const float* __restrict__ oy,
                                                               const float* __restrict__ oz, int n, const float* __restrict__ T16,
                                                               const double* __restrict__ Rm, float thr2, const double* __restrict__ C1,
                                                               const double* __restrict__ C2, const float* __restrict__ tx,
                                                               const float* __restrict__ ty, const float* __restrict__ tz,
                                                               PairRec* __restrict__ pairs, int* __restrict__ count,
                                                               OuterState* __restrict__ O, int spread, int* __restrict__ last_nn) {
  const int ph = O->phase;
  if (O->outer_done || (ph & 1)) return;  // the inner loop of this outer iteration is still running (or all is over)
  // `spread` (1, 2): only every spread-th lane carries a point.  A wave walks the union of its lanes' search paths, and a
  // 30k-point scan is fewer waves than the chip has SIMDs: thinner waves finish sooner (as in gicp_cov_kernel).
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / spread;
  if (threadIdx.x == 0) O->corr_mark = ph;  // same value from every workgroup: tells the launches behind that the pairs are fresh
  const bool mine = (i < n) && (t % spread) == 0;
  PairRec r;
  r.valid = 0;
  r.q[0] = r.q[1] = r.q[2] = 0.f;
  for (int k = 0; k < 6; k++) r.M[k] = 0.0;
  if (mine) {
  const float a = ox[i], b = oy[i], c = oz[i];
  const float qx = xform_rn(T16[0], T16[4], T16[8], T16[12], a, b, c);
  const float qy = xform_rn(T16[1], T16[5], T16[9], T16[13], a, b, c);
  const float qz = xform_rn(T16[2], T16[6], T16[10], T16[14], a, b, c);
  Best1 best;
  best.init();
  // The previous outer iteration's neighbour is offered first: between two outer iterations the cloud moves by
  // millimetres, so it is usually THE neighbour again and lets the search stop at the first shell whose bound it beats.
  // Exact all the same: Best1 orders candidates by (distance, index), whatever the order they are seen in.
  int fine_rings = 1;
  const int seed = (ph > 0) ? last_nn[i] : -1;
  if (seed >= 0) {
    best.offer(dist2_rn(qx, qy, qz, tx[seed], ty[seed], tz[seed]), seed);   // the very arithmetic of the grid scan
    fine_rings = 0;
  }
  nn_query(G, qx, qy, qz, fine_rings, thr2, best, -1);
  last_nn[i] = best.idx;
  if (best.idx >= 0 && best.d2 < thr2) {
    const int j = best.idx;
    const double* c1 = C1 + (size_t)i * 9;
    const double* c2 = C2 + (size_t)j * 9;
    double RC[9], S[9];
    for (int u = 0; u < 3; u++)
      for (int v = 0; v < 3; v++) RC[u * 3 + v] = Rm[u * 3] * c1[v] + Rm[u * 3 + 1] * c1[3 + v] + Rm[u * 3 + 2] * c1[6 + v];
    for (int u = 0; u < 3; u++)
      for (int v = 0; v < 3; v++)
        S[u * 3 + v] = RC[u * 3] * Rm[v * 3] + RC[u * 3 + 1] * Rm[v * 3 + 1] + RC[u * 3 + 2] * Rm[v * 3 + 2] + c2[u * 3 + v];
    // general 3x3 inverse by cofactors (what temp.inverse() does)
    const double k00 = S[4] * S[8] - S[5] * S[7], k01 = S[5] * S[6] - S[3] * S[8], k02 = S[3] * S[7] - S[4] * S[6];
    const double det = S[0] * k00 + S[1] * k01 + S[2] * k02;
    const double id = 1.0 / det;
    r.M[0] = k00 * id;
    r.M[1] = (S[2] * S[7] - S[1] * S[8]) * id;
    r.M[2] = (S[1] * S[5] - S[2] * S[4]) * id;
    r.M[3] = (S[0] * S[8] - S[2] * S[6]) * id;
    r.M[4] = (S[2] * S[3] - S[0] * S[5]) * id;
    r.M[5] = (S[0] * S[4] - S[1] * S[3]) * id;
    r.q[0] = tx[j]; r.q[1] = ty[j]; r.q[2] = tz[j];
    r.valid = 1;
  }
  pairs[i] = r;
  }
  // one atomic per wave (ballot + popcount) instead of one per matched point
  const unsigned long long found = __ballot(r.valid != 0);
  if (found && (threadIdx.x & 63) == (__ffsll((long long)__ballot(1)) - 1)) atomicAdd(count, __popcll(found));
}

// M = (R C1_i R^T + C2_j)^-1 in fp64 for the correspondence (i, j) at squared distance d2; an empty record when there is none
__device__ __forceinline__ PairRec empty_pair() {
  PairRec r;
  r.valid = 0;
  r.q[0] = r.q[1] = r.q[2] = 0.f;
  for (int k = 0; k < 6; k++) r.M[k] = 0.0;
  return r;
}

__device__ __forceinline__ PairRec make_pair(int i, int j, float d2, float thr2, const double* __restrict__ Rm, const double* __restrict__ C1,
                                             const double* __restrict__ C2, const float* __restrict__ tx, const float* __restrict__ ty,
                                             const float* __restrict__ tz) {
  PairRec r = empty_pair();
  if (j >= 0 && d2 < thr2) {
    const double* c1 = C1 + (size_t)i * 9;
    const double* c2 = C2 + (size_t)j * 9;
    double RC[9], S[9];
    for (int u = 0; u < 3; u++)
      for (int v = 0; v < 3; v++) RC[u * 3 + v] = Rm[u * 3] * c1[v] + Rm[u * 3 + 1] * c1[3 + v] + Rm[u * 3 + 2] * c1[6 + v];
    for (int u = 0; u < 3; u++)
      for (int v = 0; v < 3; v++)
        S[u * 3 + v] = RC[u * 3] * Rm[v * 3] + RC[u * 3 + 1] * Rm[v * 3 + 1] + RC[u * 3 + 2] * Rm[v * 3 + 2] + c2[u * 3 + v];
    // general 3x3 inverse by cofactors (what temp.inverse() does)
    const double k00 = S[4] * S[8] - S[5] * S[7], k01 = S[5] * S[6] - S[3] * S[8], k02 = S[3] * S[7] - S[4] * S[6];
    const double det = S[0] * k00 + S[1] * k01 + S[2] * k02;
    const double id = 1.0 / det;
    r.M[0] = k00 * id;
    r.M[1] = (S[2] * S[7] - S[1] * S[8]) * id;
    r.M[2] = (S[1] * S[5] - S[2] * S[4]) * id;
    r.M[3] = (S[0] * S[8] - S[2] * S[6]) * id;
    r.M[4] = (S[2] * S[3] - S[0] * S[5]) * id;
    r.M[5] = (S[0] * S[4] - S[1] * S[3]) * id;
    r.q[0] = tx[j]; r.q[1] = ty[j]; r.q[2] = tz[j];
    r.valid = 1;
  }
  return r;
}

// K6, wave-cooperative form.  Search: one wave per source point (coop_search<1-NN>), seeded with the previous outer
// iteration's neighbour.  Pairs: one thread per point builds the Mahalanobis matrix of its correspondence.
// Seeded form (outer iterations after the first), SIXTEEN lanes per point.  The previous neighbour's distance d is an upper
// bound on the answer, so the answer lies in the ball of radius d around the moved point: the fine cells that ball touches
// (at most max_cells per axis, else the point goes to `work` for the general search) are ALL the search has to read — no shells,
// no bound tests.  One row of <= max_cells cells per (y, z) pair, one lane per (row, coarse segment), the candidates of the group
// laid end to end and read 16 at a time; four points per wave.  Exact: every point at distance <= d is in one of those
// cells (the cell index is a monotone map; the reach is padded against rounding), ties included.
// work[0] = number of deferred points (zeroed by the pair kernel after use), work[1..] = their indices.
__global__ __launch_bounds__(256) void gicp_corr_ball_kernel(NNGridView G, const float* __restrict__ ox, const float* __restrict__ oy,
                                                             const float* __restrict__ oz, int n, const float* __restrict__ T16,
                                                             float thr2, const float* __restrict__ tx, const float* __restrict__ ty,
                                                             const float* __restrict__ tz, const OuterState* __restrict__ O,
                                                             int* __restrict__ last_nn, float* __restrict__ nn_d2, int* __restrict__ work,
                                                             const int max_cells) {
  const int ph = O->phase;
  if (O->outer_done || (ph & 1) || ph == 0) return;  // the first outer iteration has no seeds: the general search does it
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 4, gl = t & 15;
  if (i >= n) return;   // n * 16 threads: a group is never split by this test
  const float a = ox[i], b = oy[i], c = oz[i];
  const float q[3] = {xform_rn(T16[0], T16[4], T16[8], T16[12], a, b, c), xform_rn(T16[1], T16[5], T16[9], T16[13], a, b, c),
                      xform_rn(T16[2], T16[6], T16[10], T16[14], a, b, c)};
  const int seed = last_nn[i];
  bool general = seed < 0 || !(isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]));
  float bd = INFINITY;
  int bi = INT_MAX;
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  if (!general) {
    bd = dist2_rn(q[0], q[1], q[2], tx[seed], ty[seed], tz[seed]);
    bi = seed;
    if (!(bd < thr2)) general = true;   // the seed itself is beyond the gate: the ball would be the gate's
  }
  if (!general && !ball_cell_range(G, q, bd, max_cells, lo, hi)) general = true;   // more than max_cells (<= 8) cells on some axis
  if (general) {
    if (gl == 0) work[1 + atomicAdd(work, 1)] = i;
    return;
  }
  scan_cells_group16(G, q, lo, hi, gl, bd, bi);
  if (gl == 0) {
    last_nn[i] = bi;
    nn_d2[i] = bd;
  }
}

__global__ __launch_bounds__(256) void gicp_corr_search_kernel(NNGridView G, const float* __restrict__ ox, const float* __restrict__ oy,
                                                               const float* __restrict__ oz, int n, const float* __restrict__ T16,
                                                               float thr2, const float* __restrict__ tx, const float* __restrict__ ty,
                                                               const float* __restrict__ tz, const OuterState* __restrict__ O,
                                                               int* __restrict__ last_nn, float* __restrict__ nn_d2,
                                                               const int* __restrict__ work /* nullable */) {
  const int ph = O->phase;
  if (O->outer_done || (ph & 1)) return;  // the inner loop of this outer iteration is still running (or all is over)
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  // with a work list: the first outer iteration searches every point, later ones only what the seeded kernel deferred
  const bool listed = work && ph > 0;
  const int n_items = listed ? work[0] : n;
  for (int w = wave; w < n_items; w += n_waves) {
    const int i = listed ? work[1 + w] : w;
    const float a = ox[i], b = oy[i], c = oz[i];
    const float qx = xform_rn(T16[0], T16[4], T16[8], T16[12], a, b, c);
    const float qy = xform_rn(T16[1], T16[5], T16[9], T16[13], a, b, c);
    const float qz = xform_rn(T16[2], T16[6], T16[10], T16[14], a, b, c);
    CoopList mine;
    mine.d = INFINITY;
    mine.i = INT_MAX;
    int fine_rings = 1;
    const int seed = (ph > 0) ? last_nn[i] : -1;
    if (seed >= 0) {   // usually THE neighbour again: its distance prunes the search from the first shell on (exact all the same)
      mine.d = dist2_rn(qx, qy, qz, tx[seed], ty[seed], tz[seed]);
      mine.i = seed;
      fine_rings = 0;
    }
    coop_search<true>(G, qx, qy, qz, 1, fine_rings, thr2, -1, mine);
    if (lane == 0) {
      last_nn[i] = (mine.i == INT_MAX) ? -1 : mine.i;
      nn_d2[i] = mine.d;
    }
  }
}

__global__ __launch_bounds__(256) void gicp_corr_pairs_kernel(int n, const double* __restrict__ Rm, float thr2, const double* __restrict__ C1,
                                                              const double* __restrict__ C2, const float* __restrict__ tx,
                                                              const float* __restrict__ ty, const float* __restrict__ tz,
                                                              const int* __restrict__ last_nn, const float* __restrict__ nn_d2,
                                                              PairRec* __restrict__ pairs, int* __restrict__ count, OuterState* __restrict__ O,
                                                              int* __restrict__ work /* nullable */, int* __restrict__ count_shards = nullptr) {
  const int ph = O->phase;
  if (O->outer_done || (ph & 1)) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) O->corr_mark = ph;  // same value from every workgroup: tells the launches behind that the pairs are fresh
  if (work && i == 0) work[0] = 0;          // the deferred-point list of this pass has been consumed
  PairRec r = empty_pair();
  if (i < n) {
    r = make_pair(i, last_nn[i], nn_d2[i], thr2, Rm, C1, C2, tx, ty, tz);
    pairs[i] = r;
  }
  // one atomic per wave (ballot + popcount) instead of one per matched point
  const unsigned long long found = __ballot(r.valid != 0);
  if (found && (threadIdx.x & 63) == (__ffsll((long long)__ballot(1)) - 1))
    atomicAdd(count_shards ? count_shards + (blockIdx.x & (GICP_COUNT_SHARDS - 1)) * GICP_SHARD_STRIDE : count, __popcll(found));
}

// K6 in ONE launch per outer iteration: the seeded sixteen-lane search (gicp_corr_ball_kernel) — a point without a previous
// neighbour seeds itself from its own fine cell, as the fitness search does —, the points it
// would defer searched at once by the whole wave (the body of gicp_corr_search_kernel, one deferred point after the other), and
// the pair record of every point (gicp_corr_pairs_kernel's) written by the first lane of its group — three launches and two
// launch boundaries per outer iteration less, no work list.  Same candidates in the same order, same fp64 expressions: the
// records are those of the three-launch form bit for bit (tests/test_gicp_gpu.py; LSR_GICP_CORR_FUSED=0 selects that form).
__global__ __launch_bounds__(256) void gicp_corr_seeded_kernel(NNGridView G, const float* __restrict__ ox, const float* __restrict__ oy,
                                                               const float* __restrict__ oz, int n, const float* __restrict__ T16,
                                                               const double* __restrict__ Rm, float thr2, const float* __restrict__ tx,
                                                               const float* __restrict__ ty, const float* __restrict__ tz,
                                                               const double* __restrict__ C1, const double* __restrict__ C2,
                                                               OuterState* __restrict__ O, int* __restrict__ last_nn,
                                                               float* __restrict__ nn_d2, PairRec* __restrict__ pairs,
                                                               int* __restrict__ count_shards, const int max_cells) {
  const int ph = O->phase;
  if (O->outer_done || (ph & 1)) return;   // the inner loop of this outer iteration is still running (or all is over)
  if (threadIdx.x == 0) O->corr_mark = ph;  // same value from every workgroup: tells the launches behind that the pairs are fresh
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 4, gl = t & 15, lane = threadIdx.x & 63;
  const bool live = i < n;   // n * 16 threads: a group is never split by this test; dead groups stay for the wave-wide part
  float q[3] = {0.f, 0.f, 0.f};
  int seed = -1;
  bool general = false;
  float bd = INFINITY;
  int bi = INT_MAX;
  if (live) {
    const float a = ox[i], b = oy[i], c = oz[i];
    q[0] = xform_rn(T16[0], T16[4], T16[8], T16[12], a, b, c);
    q[1] = xform_rn(T16[1], T16[5], T16[9], T16[13], a, b, c);
    q[2] = xform_rn(T16[2], T16[6], T16[10], T16[14], a, b, c);
    seed = (ph > 0) ? last_nn[i] : -1;   // (the first outer iteration has no previous neighbours)
    general = !(isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]));
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    int seeded_reach = -1, seed_lo[3] = {0, 0, 0}, seed_hi[3] = {0, 0, 0};   // self-seeded: the cells the seed was the best of
    int fq[3] = {0, 0, 0};
    if (!general && seed >= 0) {
      bd = dist2_rn(q[0], q[1], q[2], tx[seed], ty[seed], tz[seed]);
      bi = seed;
    } else if (!general) {
      // no seed (first outer iteration; a point that had no neighbour within the gate): the best point of the query's own
      // fine cell — of the 3 x 3 x 3 cells around it if that one is empty — is a real point, hence a bound (nn.hip: nn1_ball_body)
      const float ff[3] = {floorf(q[0] * G.inv_cell), floorf(q[1] * G.inv_cell), floorf(q[2] * G.inv_cell)};
      if (!(fabsf(ff[0]) < 1.0e9f && fabsf(ff[1]) < 1.0e9f && fabsf(ff[2]) < 1.0e9f)) general = true;
      if (!general) {
        for (int a = 0; a < 3; a++) {
          fq[a] = (int)ff[a] - G.org[a];
          if (fq[a] < -1 || fq[a] > G.cdim[a] * 8) general = true;   // more than a cell outside the grid
        }
      }
      if (!general) {
        if (fq[0] >= 0 && fq[0] < G.cdim[0] * 8 && fq[1] >= 0 && fq[1] < G.cdim[1] * 8 && fq[2] >= 0 && fq[2] < G.cdim[2] * 8)
          scan_cell_group16(G, q, fq, gl, bd, bi);
        seeded_reach = (bi != INT_MAX) ? 0 : 1;
        if (bi == INT_MAX) {
          for (int a = 0; a < 3; a++) { lo[a] = max(fq[a] - 1, 0); hi[a] = min(fq[a] + 1, G.cdim[a] * 8 - 1); }
          scan_cells_group16(G, q, lo, hi, gl, bd, bi);
        }
        if (bi == INT_MAX) general = true;   // nothing within a cell of the point
        for (int a = 0; a < 3; a++) { seed_lo[a] = fq[a] - seeded_reach; seed_hi[a] = fq[a] + seeded_reach; }
      }
    }
    if (!general && !(bd < thr2)) general = true;   // the bound itself is beyond the gate: the ball would be the gate's
    if (!general && !ball_cell_range(G, q, bd, max_cells, lo, hi)) general = true;   // more than max_cells cells on some axis
    if (!general) {
      // a self-seeded point whose ball stays inside the cells its seed came from has read them already
      bool covered = seeded_reach >= 0;
      for (int a = 0; a < 3; a++) covered = covered && lo[a] >= seed_lo[a] && hi[a] <= seed_hi[a];
      if (!covered) scan_cells_group16(G, q, lo, hi, gl, bd, bi, seeded_reach == 0 ? fq : nullptr);   // (its own cell has been offered)
    }
  }
  // the points the seeded search cannot serve, one after the other, all 64 lanes on each (exactly gicp_corr_search_kernel's body)
  unsigned long long todo = __ballot(live && general && gl == 0);
  while (todo) {
    const int sl = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const float qx = __shfl(q[0], sl, 64), qy = __shfl(q[1], sl, 64), qz = __shfl(q[2], sl, 64);
    const int sd = __shfl(seed, sl, 64);
    CoopList mine;
    mine.d = INFINITY;
    mine.i = INT_MAX;
    int fine_rings = 1;
    if (sd >= 0) {
      mine.d = dist2_rn(qx, qy, qz, tx[sd], ty[sd], tz[sd]);
      mine.i = sd;
      fine_rings = 0;
    }
    coop_search<true>(G, qx, qy, qz, 1, fine_rings, thr2, -1, mine);
    const float rd = __shfl(mine.d, 0, 64);
    const int ri = __shfl(mine.i, 0, 64);
    if ((lane >> 4) == (sl >> 4)) { bd = rd; bi = (ri == INT_MAX) ? -1 : ri; }
  }
  PairRec r = empty_pair();
  if (live && gl == 0) {
    last_nn[i] = bi;
    nn_d2[i] = bd;
    r = make_pair(i, bi, bd, thr2, Rm, C1, C2, tx, ty, tz);
    pairs[i] = r;
  }
  const unsigned long long found = __ballot(r.valid != 0);
  if (found && lane == 0) atomicAdd(count_shards + (blockIdx.x & (GICP_COUNT_SHARDS - 1)) * GICP_SHARD_STRIDE, __popcll(found));
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// The GN_NRED sums of one wave through an LDS tile instead of GN_NRED x 12 shuffles: every lane parks its accumulators (fourteen
// at a time), lane (value k, half h) adds 32 of them left to right, the two halves meet in one shuffle — a fixed order.
// tile: GN_TILE_ROWS x GN_TILE_PITCH doubles private to the wave; out[k] = the wave's sum of value k (written by lane 2k).
constexpr int GN_TILE_ROWS = 14, GN_TILE_PITCH = 66;
static_assert(GN_NRED == 2 * GN_TILE_ROWS, "two rounds of GN_TILE_ROWS values");
__device__ __forceinline__ void wave_sums_tile(const double* acc, double* tile, double* out, const int lane) {
#pragma unroll
  for (int r = 0; r < 2; r++) {
#pragma unroll
    for (int k = 0; k < GN_TILE_ROWS; k++) tile[k * GN_TILE_PITCH + lane] = acc[r * GN_TILE_ROWS + k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane < 2 * GN_TILE_ROWS) {
      const int k = lane >> 1, h = lane & 1;
      const double* row = tile + k * GN_TILE_PITCH + 32 * h;
      double v = row[0];
#pragma unroll
      for (int j = 1; j < 32; j++) v += row[j];
      const double other = __shfl_xor(v, 1, 64);
      if (h == 0) out[r * GN_TILE_ROWS + k] = v + other;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// K7: one Gauss-Newton accumulation pass at the state's x.
__global__ __launch_bounds__(GN_THREADS) void gicp_gn_kernel(const float* __restrict__ ox, const float* __restrict__ oy,
                                                             const float* __restrict__ oz, int n, const PairRec* __restrict__ pairs,
                                                             const GnState* __restrict__ S, const OuterState* __restrict__ O,
                                                             double* __restrict__ partials) {
  {
    const int ph = O->phase;
    if (O->outer_done || (!(ph & 1) && O->corr_mark != ph)) return;  // nothing to accumulate until fresh pairs exist
  }
  __shared__ double s_red[GN_THREADS / 64][GN_NRED];
  __shared__ double s_tile[GN_THREADS / 64][GN_TILE_ROWS * GN_TILE_PITCH];
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = S->T[k];
  double acc[GN_NRED];
#pragma unroll
  for (int k = 0; k < GN_NRED; k++) acc[k] = 0.0;
  for (int i = blockIdx.x * GN_THREADS + threadIdx.x; i < n; i += gridDim.x * GN_THREADS) {
    const PairRec r = pairs[i];
    if (!r.valid) continue;
    const float a = ox[i], b = oy[i], c = oz[i];
    // transformation_matrix * p_src in float, residual promoted to double (OptimizationFunctorWithIndices)
    const float ppx = xform_rn(T[0], T[1], T[2], T[3], a, b, c);
    const float ppy = xform_rn(T[4], T[5], T[6], T[7], a, b, c);
    const float ppz = xform_rn(T[8], T[9], T[10], T[11], a, b, c);
    const double res[3] = {(double)(ppx - r.q[0]), (double)(ppy - r.q[1]), (double)(ppz - r.q[2])};
    const double p[3] = {(double)a, (double)b, (double)c};
    const double M00 = r.M[0], M01 = r.M[1], M02 = r.M[2], M11 = r.M[3], M12 = r.M[4], M22 = r.M[5];
    double J[3][3];  // rotation columns: dR_k * p
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double* D = S->dR + 9 * k;
#pragma unroll
      for (int u = 0; u < 3; u++) J[u][k] = D[u * 3] * p[0] + D[u * 3 + 1] * p[1] + D[u * 3 + 2] * p[2];
    }
    const double Mr0 = M00 * res[0] + M01 * res[1] + M02 * res[2];
    const double Mr1 = M01 * res[0] + M11 * res[1] + M12 * res[2];
    const double Mr2 = M02 * res[0] + M12 * res[1] + M22 * res[2];
    acc[0] += res[0] * Mr0 + res[1] * Mr1 + res[2] * Mr2;
    acc[1] += Mr0; acc[2] += Mr1; acc[3] += Mr2;
    double MJ[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      acc[4 + k] += J[0][k] * Mr0 + J[1][k] * Mr1 + J[2][k] * Mr2;
      MJ[0][k] = M00 * J[0][k] + M01 * J[1][k] + M02 * J[2][k];
      MJ[1][k] = M01 * J[0][k] + M11 * J[1][k] + M12 * J[2][k];
      MJ[2][k] = M02 * J[0][k] + M12 * J[1][k] + M22 * J[2][k];
    }
    // upper triangle of J^T M J with J = [I | Jr]: rows 0..2 (tt, tr), rows 3..5 (rr)
    acc[7] += M00; acc[8] += M01; acc[9] += M02; acc[10] += MJ[0][0]; acc[11] += MJ[0][1]; acc[12] += MJ[0][2];
    acc[13] += M11; acc[14] += M12; acc[15] += MJ[1][0]; acc[16] += MJ[1][1]; acc[17] += MJ[1][2];
    acc[18] += M22; acc[19] += MJ[2][0]; acc[20] += MJ[2][1]; acc[21] += MJ[2][2];
    acc[22] += J[0][0] * MJ[0][0] + J[1][0] * MJ[1][0] + J[2][0] * MJ[2][0];
    acc[23] += J[0][0] * MJ[0][1] + J[1][0] * MJ[1][1] + J[2][0] * MJ[2][1];
    acc[24] += J[0][0] * MJ[0][2] + J[1][0] * MJ[1][2] + J[2][0] * MJ[2][2];
    acc[25] += J[0][1] * MJ[0][1] + J[1][1] * MJ[1][1] + J[2][1] * MJ[2][1];
    acc[26] += J[0][1] * MJ[0][2] + J[1][1] * MJ[1][2] + J[2][1] * MJ[2][2];
    acc[27] += J[0][2] * MJ[0][2] + J[1][2] * MJ[1][2] + J[2][2] * MJ[2][2];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  wave_sums_tile(acc, s_tile[wid], s_red[wid], lane);
  __syncthreads();
  if (threadIdx.x < GN_NRED) {
    double v = s_red[0][threadIdx.x];
    for (int w = 1; w < GN_THREADS / 64; w++) v += s_red[w][threadIdx.x];
    partials[(size_t)blockIdx.x * 32 + threadIdx.x] = v;
  }
}

// delta = H^{-1} b, Gaussian elimination with partial pivoting (registers).  A rank-deficient J^T M J (collinear or too few
// correspondences: some motion is unobservable) has a pivot that vanishes against the matrix' scale; the full undamped
// step would be inf/NaN or astronomically large, so the step is dropped (delta = 0): the inner loop then ends on its
// iteration cap with x unchanged and align() returns a finite pose (healthy systems never come near the threshold).
__device__ void solve6_gn(const double* H, const double* b, double* x) {
  // Gauss-Jordan with partial pivoting, element for element the arithmetic of solve6_gn_wave below (the two chains must give
  // bit-identical steps: tests/test_gicp_gpu.py::test_search_and_chain_variants_give_identical_results)
  double A[6][7];
  double scale = 0.0;
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j < 6; j++) { A[i][j] = H[i * 6 + j]; scale = fmax(scale, fabs(A[i][j])); }
    A[i][6] = b[i];
  }
  bool singular = !(scale > 0.0) || !(scale < 1.0e300);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double best = fabs(A[k][k]);
    int piv = k;
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double v = fabs(A[i][k]);
      if (v > best) { best = v; piv = i; }
    }
    if (!(best > 1.0e-13 * scale)) singular = true;
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const bool sw = (piv == i);
#pragma unroll
      for (int j = 0; j < 7; j++) {
        const double a = A[k][j], c = A[i][j];
        A[k][j] = sw ? c : a;
        A[i][j] = sw ? a : c;
      }
    }
    const double inv = 1.0 / A[k][k];
    double rowk[7];
#pragma unroll
    for (int j = 0; j < 7; j++) rowk[j] = A[k][j];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      if (i == k) continue;
      const double f = A[i][k] * inv;
#pragma unroll
      for (int j = 0; j < 7; j++) A[i][j] -= f * rowk[j];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; k++) x[k] = singular ? 0.0 : A[k][6] / A[k][k];
}

// The same on ONE WAVE: lane 8 r + c holds element (r, c) of the augmented 6 x 7 matrix [H | -g], built straight from the sums
// (H = 2 sums / m, upper triangle at sums[7..27]; g = 2 sums[1..6] / m); every elimination step is a handful of cross-lane
// reads instead of ~100 dependent fp64 operations on one lane (the one-lane solve was ~2 us of every Gauss-Newton step).
// dx (LDS, 6 doubles) receives the step.
__device__ __forceinline__ void solve6_gn_wave(const double* sums, double m, double* dx) {
  const int lane = threadIdx.x & 63;
  const int r = lane >> 3, c = lane & 7;
  double v = 0.0;
  if (r < 6 && c < 6) {
    const int i = min(r, c), j = max(r, c);
    v = 2.0 * sums[7 + 6 * i - (i * (i - 1)) / 2 + (j - i)] / m;
  } else if (r < 6 && c == 6) {
    v = -(2.0 * sums[1 + r] / m);
  }
  double scale = (r < 6 && c < 6) ? fabs(v) : 0.0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) scale = fmax(scale, __shfl_xor(scale, w, 64));
  bool singular = !(scale > 0.0) || !(scale < 1.0e300);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double best = fabs(__shfl(v, k * 8 + k, 64));
    int piv = k;
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double t = fabs(__shfl(v, i * 8 + k, 64));
      if (t > best) { best = t; piv = i; }
    }
    if (!(best > 1.0e-13 * scale)) singular = true;
    const double from_piv = __shfl(v, piv * 8 + c, 64), from_k = __shfl(v, k * 8 + c, 64);
    v = (r == k) ? from_piv : ((r == piv) ? from_k : v);
    const double akk = __shfl(v, k * 8 + k, 64), akj = __shfl(v, k * 8 + c, 64), aik = __shfl(v, r * 8 + k, 64);
    const double inv = 1.0 / akk;
    const double f = aik * inv;
    if (r != k && r < 6) v -= f * akj;
  }
  const int jj = min(lane, 5);
  const double num = __shfl(v, jj * 8 + 6, 64), den = __shfl(v, jj * 8 + jj, 64);
  if (lane < 6) dx[lane] = singular ? 0.0 : num / den;
}

// column-major fp32 product (previous_transformation_ * guess)
__host__ __device__ inline void mat4_mul_cm(const float* A, const float* B, float* C) {
  float T[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      float s = 0;
      for (int k = 0; k < 4; k++) s += A[k * 4 + r] * B[c * 4 + k];
      T[c * 4 + r] = s;
    }
  for (int k = 0; k < 16; k++) C[k] = T[k];
}

// Prepare outer iteration k: Mahalanobis rotation of (transformation_ * guess), the optimiser's start x from
// transformation_, previous_transformation_ = transformation_ — what the top of the reference's while loop does.
// (everything but the trigonometry of the new x: gn_apply_state() has to follow)
__host__ __device__ inline void gicp_begin_outer_pre(IterBlock& B) {
  OuterState& O = B.out;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double v = 0;
      for (int k = 0; k < 4; k++) v += (double)O.trans[k * 4 + i] * (double)O.G[j * 4 + k];
      B.Rm[i * 3 + j] = v;
    }
  GnState& S = B.st;
  const int max_inner = S.max_inner;
  S.x[0] = O.trans[12]; S.x[1] = O.trans[13]; S.x[2] = O.trans[14];
  S.x[3] = atan2((double)O.trans[6], (double)O.trans[10]);
  S.x[4] = asin(-(double)O.trans[2]);
  S.x[5] = atan2((double)O.trans[1], (double)O.trans[0]);
  S.f = 0; S.gnorm = 0; S.m = 0; S.inner_iter = 0; S.inner_done = 0; S.max_inner = max_inner;
  for (int k = 0; k < 16; k++) { B.T16[k] = O.trans[k]; O.prev[k] = O.trans[k]; }
  B.count = 0;
}

__host__ __device__ inline void gicp_begin_outer(IterBlock& B) {
  gicp_begin_outer_pre(B);
  gn_apply_state(B.st);
}

__global__ __launch_bounds__(256) void gicp_update_kernel(IterBlock* __restrict__ B, const double* __restrict__ partials, int nblocks,
                                                          GicpMailbox* mb, unsigned int token, int launch_index) {
  GnState* S = &B->st;
  OuterState* O = &B->out;
  const unsigned long long progress = ((unsigned long long)token << 32) | (unsigned int)launch_index;
  const int ph = O->phase;
  if (O->outer_done || (!(ph & 1) && O->corr_mark != ph)) {  // all over, or no fresh pairs yet: nothing to consume
    if (threadIdx.x == 0) __hip_atomic_store(&mb->progress, progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  __shared__ double s_grp[8][32];
  __shared__ double s_sum[32];
  const int t = threadIdx.x;
  {
    const int v = t & 31, grp = t >> 5;  // 8 groups x 32 values, fixed order
    double acc = 0.0;
    if (v < GN_NRED)
      for (int b = grp; b < nblocks; b += 8) acc += partials[(size_t)b * 32 + v];
    s_grp[grp][v] = acc;
  }
  __syncthreads();
  if (t < 32) {
    double v = 0.0;
    for (int g2 = 0; g2 < 8; g2++) v += s_grp[g2][t];
    s_sum[t] = v;
  }
  __syncthreads();
  if (t != 0) return;
  bool finished = false;
  if (!(ph & 1)) {  // first step of this outer iteration: the correspondence pass has just run, adopt its pair count
    O->phase = ph + 1;
    S->m = B->count;
    if (S->m < 4) finished = true;  // the reference's NotEnoughPointsException: leave x alone, the outer loop ends
  }
  if (!finished) {
    const double m = (double)S->m;
    double g[6], H[36];
    S->f = s_sum[0] / m;
    for (int k = 0; k < 6; k++) g[k] = 2.0 * s_sum[1 + k] / m;
    int idx = 7;
    for (int i = 0; i < 6; i++)
      for (int j = i; j < 6; j++) {
        H[i * 6 + j] = H[j * 6 + i] = 2.0 * s_sum[idx] / m;
        idx++;
      }
    double gn = 0;
    for (int k = 0; k < 6; k++) gn += g[k] * g[k];
    gn = sqrt(gn);
    S->gnorm = gn;
    if (gn < 1e-2 || S->inner_iter >= S->max_inner || !(gn == gn)) {  // BFGS testGradient(1e-2) / max_inner_iterations_
      finished = true;
    } else {
      double neg[6], dx[6];
      for (int k = 0; k < 6; k++) neg[k] = -g[k];
      solve6_gn(H, neg, dx);
      for (int k = 0; k < 6; k++) S->x[k] += dx[k];
      S->inner_iter++;
      gn_apply_state(*S);
    }
  }
  if (finished) {
    // ---- the inner loop of this outer iteration has ended: the reference's outer bookkeeping
    bool stop = false;
    O->last_cnt = S->m;
    if (S->m < 4) {
      stop = true;  // NotEnoughPointsException is caught, the loop is left unconverged
    } else {
      O->gn_steps += S->inner_iter;
      O->last_cost = S->f;
      if (!(S->gnorm == S->gnorm)) {
        stop = true;  // NaN: the reference's solver exception path
      } else {
        GnState tmp = *S;  // transformation_ = applyState(identity, x)
        gn_apply_state(tmp);
        float* tr = O->trans;
        tr[0] = tmp.T[0]; tr[4] = tmp.T[1]; tr[8] = tmp.T[2];  tr[12] = tmp.T[3];
        tr[1] = tmp.T[4]; tr[5] = tmp.T[5]; tr[9] = tmp.T[6];  tr[13] = tmp.T[7];
        tr[2] = tmp.T[8]; tr[6] = tmp.T[9]; tr[10] = tmp.T[10]; tr[14] = tmp.T[11];
        tr[3] = tr[7] = tr[11] = 0.f; tr[15] = 1.f;
        double delta = 0;
        for (int k = 0; k < 4; k++)
          for (int l = 0; l < 4; l++) {
            const double ratio = (k < 3 && l < 3) ? 1. / O->rot_eps : 1. / O->trans_eps;
            const double c_delta = ratio * fabs((double)O->prev[l * 4 + k] - (double)tr[l * 4 + k]);
            if (c_delta > delta) delta = c_delta;
          }
        O->nr_iterations++;
        if (O->nr_iterations >= O->max_iterations || delta < 1) {
          O->converged = 1;
          for (int k = 0; k < 16; k++) O->prev[k] = tr[k];
          stop = true;
        }
      }
    }
    if (stop) {
      O->outer_done = 1;
      mat4_mul_cm(O->prev, O->G, mb->final_T);  // final_transformation_ = previous_transformation_ * guess
      mb->converged = O->converged;
      mb->nr_iterations = O->nr_iterations;
      mb->last_cnt = O->last_cnt;
      mb->gn_steps = O->gn_steps;
      mb->last_cost = O->last_cost;
      __threadfence_system();
      __hip_atomic_store(&mb->done, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      gicp_begin_outer(*B);
      O->phase = (O->phase | 1) + 1;  // next even phase: the correspondence pass of the next outer iteration
    }
  }
  __hip_atomic_store(&mb->progress, progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- fused Gauss-Newton chain ("pull", as the NDT chain): step s first consumes what step s-1 accumulated — every
// workgroup, redundantly and deterministically, on an LDS copy of the iteration block: fixed-order sum of the partial rows,
// gradient test, 6x6 solve, state update, and at the end of an inner loop the reference's outer bookkeeping — then
// accumulates its share of the pairs at the NEW state.  One launch per inner iteration instead of two (accumulate, update),
// and no single-workgroup launch on the critical path.  The block is double buffered by step parity (step s reads
// blk[s & 1], workgroup 0 writes blk[(s + 1) & 1]); the correspondence launches between steps s-1 and s work on blk[s & 1].
// Returns true when this launch has pairs to accumulate at the state it leaves in B.
// need_solve (nullable): instead of solving on the calling lane, raise *need_solve and return -1; the caller solves on the wave
// (solve6_gn_wave) and finishes the step with gicp_advance_take_step().
__device__ int gicp_advance(IterBlock& Bk, const double* s_sum, GicpMailbox* mb, unsigned int token, bool publish, int* need_solve) {
  GnState* S = &Bk.st;
  OuterState* O = &Bk.out;
  if (!Bk.have_partials) {
    // first step of this outer iteration: the correspondence pass has just run, adopt its pair count
    O->phase = O->phase + 1;
    S->m = Bk.count;
    if (S->m >= 4) {   // (fewer: the reference's NotEnoughPointsException — x is left alone, the outer loop ends below)
      Bk.have_partials = 1;
      return 1;         // evaluate at the start x
    }
  } else {
    const double m = (double)S->m;
    double g[6];
    S->f = s_sum[0] / m;
    for (int k = 0; k < 6; k++) g[k] = 2.0 * s_sum[1 + k] / m;
    double gn = 0;
    for (int k = 0; k < 6; k++) gn += g[k] * g[k];
    gn = sqrt(gn);
    S->gnorm = gn;
    if (!(gn < 1e-2 || S->inner_iter >= S->max_inner || !(gn == gn))) {  // else BFGS testGradient(1e-2) / max_inner_iterations_: loop over
      if (need_solve) { *need_solve = 1; return -1; }   // the wave forms H itself (solve6_gn_wave): 21 fp64 divisions less on this lane
      double H[36];
      int idx = 7;
      for (int i = 0; i < 6; i++)
        for (int j = i; j < 6; j++) {
          H[i * 6 + j] = H[j * 6 + i] = 2.0 * s_sum[idx] / m;
          idx++;
        }
      double neg[6], dx[6];
      for (int k = 0; k < 6; k++) neg[k] = -g[k];
      solve6_gn(H, neg, dx);
      for (int k = 0; k < 6; k++) S->x[k] += dx[k];
      S->inner_iter++;
      return 1 | 2;     // evaluate at the new x, whose trigonometry the caller spreads over lanes
    }
  }
  // ---- the inner loop of this outer iteration has ended: the reference's outer bookkeeping
  Bk.have_partials = 0;
  bool stop = false;
  O->last_cnt = S->m;
  if (S->m < 4) {
    stop = true;  // NotEnoughPointsException is caught, the loop is left unconverged
  } else {
    O->gn_steps += S->inner_iter;
    O->last_cost = S->f;
    if (!(S->gnorm == S->gnorm)) {
      stop = true;  // NaN: the reference's solver exception path
    } else {
      // transformation_ = applyState(identity, x): x has not moved since S->T was composed from it
      float* tr = O->trans;
      tr[0] = S->T[0]; tr[4] = S->T[1]; tr[8] = S->T[2];  tr[12] = S->T[3];
      tr[1] = S->T[4]; tr[5] = S->T[5]; tr[9] = S->T[6];  tr[13] = S->T[7];
      tr[2] = S->T[8]; tr[6] = S->T[9]; tr[10] = S->T[10]; tr[14] = S->T[11];
      tr[3] = tr[7] = tr[11] = 0.f; tr[15] = 1.f;
      double delta = 0;
      for (int k = 0; k < 4; k++)
        for (int l = 0; l < 4; l++) {
          const double ratio = (k < 3 && l < 3) ? 1. / O->rot_eps : 1. / O->trans_eps;
          const double c_delta = ratio * fabs((double)O->prev[l * 4 + k] - (double)tr[l * 4 + k]);
          if (c_delta > delta) delta = c_delta;
        }
      O->nr_iterations++;
      if (O->nr_iterations >= O->max_iterations || delta < 1) {
        O->converged = 1;
        for (int k = 0; k < 16; k++) O->prev[k] = tr[k];
        stop = true;
      }
    }
  }
  if (stop) {
    O->outer_done = 1;
    if (publish) {
      mat4_mul_cm(O->prev, O->G, mb->final_T);  // final_transformation_ = previous_transformation_ * guess
      mb->converged = O->converged;
      mb->nr_iterations = O->nr_iterations;
      mb->last_cnt = O->last_cnt;
      mb->gn_steps = O->gn_steps;
      mb->last_cost = O->last_cost;
      __threadfence_system();
      __hip_atomic_store(&mb->done, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return 0;
  }
  gicp_begin_outer_pre(Bk);
  O->phase = (O->phase | 1) + 1;  // next even phase: the correspondence pass of the next outer iteration
  return 2;                       // nothing to accumulate, but the new start x needs its trigonometry
}

__global__ __launch_bounds__(GN_THREADS) void gicp_step_kernel(IterBlock* __restrict__ blk2, int step, const float* __restrict__ ox,
                                                               const float* __restrict__ oy, const float* __restrict__ oz, int n,
                                                               const PairRec* __restrict__ pairs, double* __restrict__ partials2,
                                                               int nblocks, GicpMailbox* mb, unsigned int token, int launch_index,
                                                               const int* __restrict__ count_shards /* nullable */) {
  static_assert(sizeof(IterBlock) % 8 == 0, "IterBlock is copied as 8-byte words");
  __shared__ __attribute__((aligned(16))) unsigned long long s_raw[sizeof(IterBlock) / 8];
  __shared__ double s_grp[8][32];
  __shared__ double s_sum[32];
  __shared__ double s_red[GN_THREADS / 64][GN_NRED];
  __shared__ double s_tile[GN_THREADS / 64][GN_TILE_ROWS * GN_TILE_PITCH];
  __shared__ int s_do;
  IterBlock& Bk = *reinterpret_cast<IterBlock*>(s_raw);
  const int t = threadIdx.x;
  constexpr int NW = (int)(sizeof(IterBlock) / 8);
  {
    const unsigned long long* in = reinterpret_cast<const unsigned long long*>(blk2 + (step & 1));
    for (int w = t; w < NW; w += GN_THREADS) s_raw[w] = in[w];
  }
  __syncthreads();
  unsigned long long* out = reinterpret_cast<unsigned long long*>(blk2 + ((step + 1) & 1));
  const unsigned long long progress = ((unsigned long long)token << 32) | (unsigned int)launch_index;
  const int ph = Bk.out.phase;
  if (Bk.out.outer_done || (!(ph & 1) && Bk.out.corr_mark != ph)) {  // all over, or no fresh pairs yet: carry the block forward
    if (blockIdx.x == 0) {
      for (int w = t; w < NW; w += GN_THREADS) out[w] = s_raw[w];
      if (t == 0) __hip_atomic_store(&mb->progress, progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (Bk.have_partials) {   // rows the previous step left in the other bank: 8 groups x 32 values, fixed order
    const double* rows = partials2 + (size_t)((step + 1) & 1) * nblocks * 32;
    const int v = t & 31, grp = t >> 5;
    double acc = 0.0;
    if (v < GN_NRED)
      for (int b = grp; b < nblocks; b += 8) acc += rows[(size_t)b * 32 + v];
    s_grp[grp][v] = acc;
    __syncthreads();
    if (t < 32) {
      double v2 = 0.0;
      for (int g2 = 0; g2 < 8; g2++) v2 += s_grp[g2][t];
      s_sum[t] = v2;
    }
  }
  __syncthreads();
  __shared__ float s_f6[6];
  __shared__ double s_d6[6];
  __shared__ double s_dx[8];
  __shared__ int s_need;
  if (t < 64) {   // wave 0: the scalar bookkeeping on lane 0, the 6x6 solve on the wave (lockstep + in-order LDS: no workgroup barrier)
    int code = 0;
    if (count_shards && !Bk.have_partials) {   // the correspondence pass has just run: its pair count = the counters' total - the last one
      int total = count_shards[t * GICP_SHARD_STRIDE];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) total += __shfl_xor(total, m, 64);
      if (t == 0) { Bk.count = total - Bk.count_base; Bk.count_base = total; }
    }
    if (t == 0) { s_need = 0; code = gicp_advance(Bk, s_sum, mb, token, blockIdx.x == 0, &s_need); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (s_need) {
      solve6_gn_wave(s_sum, (double)Bk.st.m, s_dx);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (t == 0) {
        for (int k = 0; k < 6; k++) Bk.st.x[k] += s_dx[k];
        Bk.st.inner_iter++;
        code = 1 | 2;   // evaluate at the new x, whose trigonometry is spread over lanes below
      }
    }
    if (t == 0) s_do = code;
  }
  __syncthreads();
  if (s_do & 2) {   // sin/cos of the three angles: fp32 on lanes 0..2 of wave 0, fp64 on lanes 0..2 of wave 1, side by side
    if (t < 3) {
      const float ang = (float)Bk.st.x[3 + t];
      s_f6[2 * t] = cosf(ang);
      s_f6[2 * t + 1] = sinf(ang);
    } else if (t >= 64 && t < 67) {
      const double ang = Bk.st.x[3 + (t - 64)];
      s_d6[2 * (t - 64)] = cos(ang);
      s_d6[2 * (t - 64) + 1] = sin(ang);
    }
    __syncthreads();
    if (t == 0) gn_apply_state_trig(Bk.st, s_f6, s_d6);
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    for (int w = t; w < NW; w += GN_THREADS) out[w] = s_raw[w];
    if (t == 0) __hip_atomic_store(&mb->progress, progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (!(s_do & 1)) return;
  // ---- accumulate at the state just left in Bk (the body of gicp_gn_kernel)
  const GnState* S = &Bk.st;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = S->T[k];
  double acc[GN_NRED];
#pragma unroll
  for (int k = 0; k < GN_NRED; k++) acc[k] = 0.0;
  for (int i = blockIdx.x * GN_THREADS + threadIdx.x; i < n; i += gridDim.x * GN_THREADS) {
    const PairRec r = pairs[i];
    if (!r.valid) continue;
    const float a = ox[i], b = oy[i], c = oz[i];
    const float ppx = xform_rn(T[0], T[1], T[2], T[3], a, b, c);
    const float ppy = xform_rn(T[4], T[5], T[6], T[7], a, b, c);
    const float ppz = xform_rn(T[8], T[9], T[10], T[11], a, b, c);
    const double res[3] = {(double)(ppx - r.q[0]), (double)(ppy - r.q[1]), (double)(ppz - r.q[2])};
    const double p[3] = {(double)a, (double)b, (double)c};
    const double M00 = r.M[0], M01 = r.M[1], M02 = r.M[2], M11 = r.M[3], M12 = r.M[4], M22 = r.M[5];
    double J[3][3];  // rotation columns: dR_k * p
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double* D = S->dR + 9 * k;
#pragma unroll
      for (int u = 0; u < 3; u++) J[u][k] = D[u * 3] * p[0] + D[u * 3 + 1] * p[1] + D[u * 3 + 2] * p[2];
    }
    const double Mr0 = M00 * res[0] + M01 * res[1] + M02 * res[2];
    const double Mr1 = M01 * res[0] + M11 * res[1] + M12 * res[2];
    const double Mr2 = M02 * res[0] + M12 * res[1] + M22 * res[2];
    acc[0] += res[0] * Mr0 + res[1] * Mr1 + res[2] * Mr2;
    acc[1] += Mr0; acc[2] += Mr1; acc[3] += Mr2;
    double MJ[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      acc[4 + k] += J[0][k] * Mr0 + J[1][k] * Mr1 + J[2][k] * Mr2;
      MJ[0][k] = M00 * J[0][k] + M01 * J[1][k] + M02 * J[2][k];
      MJ[1][k] = M01 * J[0][k] + M11 * J[1][k] + M12 * J[2][k];
      MJ[2][k] = M02 * J[0][k] + M12 * J[1][k] + M22 * J[2][k];
    }
    acc[7] += M00; acc[8] += M01; acc[9] += M02; acc[10] += MJ[0][0]; acc[11] += MJ[0][1]; acc[12] += MJ[0][2];
    acc[13] += M11; acc[14] += M12; acc[15] += MJ[1][0]; acc[16] += MJ[1][1]; acc[17] += MJ[1][2];
    acc[18] += M22; acc[19] += MJ[2][0]; acc[20] += MJ[2][1]; acc[21] += MJ[2][2];
    acc[22] += J[0][0] * MJ[0][0] + J[1][0] * MJ[1][0] + J[2][0] * MJ[2][0];
    acc[23] += J[0][0] * MJ[0][1] + J[1][0] * MJ[1][1] + J[2][0] * MJ[2][1];
    acc[24] += J[0][0] * MJ[0][2] + J[1][0] * MJ[1][2] + J[2][0] * MJ[2][2];
    acc[25] += J[0][1] * MJ[0][1] + J[1][1] * MJ[1][1] + J[2][1] * MJ[2][1];
    acc[26] += J[0][1] * MJ[0][2] + J[1][1] * MJ[1][2] + J[2][1] * MJ[2][2];
    acc[27] += J[0][2] * MJ[0][2] + J[1][2] * MJ[1][2] + J[2][2] * MJ[2][2];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  wave_sums_tile(acc, s_tile[wid], s_red[wid], lane);
  __syncthreads();
  if (threadIdx.x < GN_NRED) {
    double v = s_red[0][threadIdx.x];
    for (int w = 1; w < GN_THREADS / 64; w++) v += s_red[w][threadIdx.x];
    partials2[(size_t)(step & 1) * nblocks * 32 + (size_t)blockIdx.x * 32 + threadIdx.x] = v;
  }
}

int compute_covariances(lsr_handle_s* h, const DeviceCloud& cloud, const HashGridDev& grid, DevBuf<double>& cov, bool raw = false) {
  const int n = (int)cloud.n;
  const double eps = raw ? -1.0 : h->gicp.gicp_eps;
  int st = cov.reserve((size_t)n * 9);
  if (st) return st;
  if (n == 0) return LSR_OK;
  const int k = h->gicp.k;
  const size_t smem = BestK::lds_bytes(k);
  // deferred-point list: [0] = count, [1..n] = point indices
  if ((st = h->gicp_ws.work.reserve((size_t)n + 1))) return st;
  int* work = h->gicp_ws.work.p;
  const bool coop = (k <= 64);
  const int ring_cap = coop ? 2 : -1;  // k > 64 does not fit one wave: the per-thread walk finishes everything
  if (coop && nn_coop_enabled()) {   // one wave per point, then one thread per point
    if ((st = h->gicp_ws.work.reserve((size_t)n * k + 1))) return st;
    int* nbr = h->gicp_ws.work.p;
    const long threads = (long)n * 64;
    hipLaunchKernelGGL(gicp_knn_wave_kernel, dim3((unsigned)std::min<long>((threads + 255) / 256, 1 << 20)), dim3(256), 0, h->stream,
                       make_view(grid), cloud.x(), cloud.y(), cloud.z(), n, k, nbr);
    hipLaunchKernelGGL(gicp_cov_from_nbr_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, cloud.x(), cloud.y(), cloud.z(), n, k,
                       eps, nbr, cov.p);
    LSR_HIP(hipGetLastError());
    return LSR_OK;
  }
  LSR_HIP(hipMemsetAsync(work, 0, sizeof(int), h->stream));
  const int spread = (n <= 65536) ? 2 : 1;   // measured on a 30k-point scan: 420 -> 384 us (4: 410, 8: 360)
  const long threads = (long)n * spread;
  hipLaunchKernelGGL(gicp_cov_kernel, dim3((unsigned)((threads + NN_THREADS - 1) / NN_THREADS)), dim3(NN_THREADS), smem, h->stream, make_view(grid),
                     cloud.x(), cloud.y(), cloud.z(), n, k, eps, 2, ring_cap, spread, work, work + 1, cov.p);
  LSR_HIP(hipGetLastError());
  if (coop) {
    hipLaunchKernelGGL(gicp_cov_coop_kernel, dim3(1024), dim3(256), 0, h->stream, make_view(grid), cloud.x(), cloud.y(), cloud.z(),
                       k, eps, work, work + 1, cov.p);
    LSR_HIP(hipGetLastError());
  }
  return LSR_OK;
}

// The 20-NN of a voxel-filtered LiDAR cloud reach ~2 m along far-range scan rings; a 1 m cell keeps them
// inside the first two fine shells (the 1-NN correspondence grid stays at 0.5 m).
constexpr float GICP_COV_CELL = 1.0f;

int ensure_covariances(lsr_handle_s* h) {
  TargetData& t = *h->target;
  int st;
  if ((int)t.n < h->gicp.k || (int)h->source.n < h->gicp.k) {
    set_last_error("GICP: cloud has fewer points than k_correspondences");
    return LSR_ERR_TOO_FEW_POINTS;
  }
  std::unique_lock<std::mutex> lock(t.build_mutex);  // target-side lazy builds are shared state (lsr_share_target)
  if (!t.has_hash) {
    if ((st = nn_build_hash(t.cloud, nn_pick_cell(t.cloud.n, h), t.hash, h->scratch, h->stream))) return st;
    t.has_hash = true;
  }
  if (t.has_cov && (t.cov_k != h->gicp.k || t.cov_eps != h->gicp.gicp_eps) &&
      h->target.use_count() > (long)(1 + (h->spare_target == h->target))) {
    set_last_error("the shared target's covariances were computed with other k_correspondences / gicp_epsilon");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  if (!t.has_cov || t.cov_k != h->gicp.k || t.cov_eps != h->gicp.gicp_eps) {
    if ((st = nn_build_hash(t.cloud, GICP_COV_CELL, h->source_hash, h->scratch, h->stream))) return st;  // borrowed, rebuilt below
    if ((st = compute_covariances(h, t.cloud, h->source_hash, t.cov))) return st;
    t.has_cov = true;
    t.cov_k = h->gicp.k;
    t.cov_eps = h->gicp.gicp_eps;
    h->source_cov_valid = false;
  }
  lock.unlock();
  if (!h->source_cov_valid) {
    if ((st = nn_build_hash(h->source, GICP_COV_CELL, h->source_hash, h->scratch, h->stream))) return st;
    if ((st = compute_covariances(h, h->source, h->source_hash, h->source_cov))) return st;
    h->source_cov_valid = true;
  }
  return LSR_OK;
}

}  // namespace

int gicp_get_covariances(lsr_handle_s* h, int which, double* cov) {
  if (!h->target || h->target->n == 0) return LSR_ERR_NO_TARGET;
  if (!h->has_source) return LSR_ERR_NO_SOURCE;
  int st = ensure_covariances(h);
  if (st) return st;
  if (which < 0 || which > 3) { set_last_error("which must be 0..3"); return LSR_ERR_INVALID_ARGUMENT; }
  if (which >= 2) {  // the sample covariances BEFORE the eigen-regularisation (same neighbours, same FLOAT products)
    const DeviceCloud& cloud = (which == 2) ? h->source : h->target->cloud;
    if ((st = nn_build_hash(cloud, GICP_COV_CELL, h->source_hash, h->scratch, h->stream))) return st;
    if ((st = compute_covariances(h, cloud, h->source_hash, h->gicp_ws.raw_cov, true))) return st;
    LSR_HIP(hipMemcpyAsync(cov, h->gicp_ws.raw_cov.p, sizeof(double) * 9 * cloud.n, hipMemcpyDeviceToHost, h->stream));
    LSR_HIP(hipStreamSynchronize(h->stream));
    h->source_cov_valid = false;  // source_hash was borrowed
    return LSR_OK;
  }
  const DevBuf<double>& c = which == 0 ? h->source_cov : h->target->cov;
  const size_t n = which == 0 ? h->source.n : h->target->n;
  LSR_HIP(hipMemcpyAsync(cov, c.p, sizeof(double) * 9 * n, hipMemcpyDeviceToHost, h->stream));
  LSR_HIP(hipStreamSynchronize(h->stream));
  return LSR_OK;
}

// One GICP align in flight: the launch chain of gicp_align as a resumable object, so that several registrations (a candidate set
// with the backend's stand-alone configuration, graph_based_slam/param/graphbasedslam.yaml:3) can be fed side by side, each on
// its own stream with its own mailbox.
namespace {
struct GicpChain {
  lsr_handle_s* h = nullptr;
  hipStream_t s = nullptr;
  int n = 0, nblocks = 1, updates = 0, spread = 1;
  unsigned int token = 0;
  float thr2 = 0.f;
  bool coop_corr = false, fused = false, ball = false, corr_fused = false, done = false;
  int* d_work = nullptr;
  int* d_shards = nullptr;
  double* d_partials = nullptr;
  IterBlock* d_blk = nullptr;
  PairRec* d_pairs = nullptr;
  long hard_cap = 0;
  unsigned long long last_progress = 0;
  std::chrono::steady_clock::time_point t_begin, t_progress;

  void enqueue_group(int steps) {
    GicpWorkspace& ws = h->gicp_ws;
    const TargetData& t = *h->target;
    if (fused) {   // the correspondence launches work on the block the next step will read
      IterBlock* cur = d_blk + (updates & 1);
      if (corr_fused) {
        // one launch per outer iteration: seeded (or self-seeded) search + what it cannot serve + pair records
        {
          hipLaunchKernelGGL(gicp_corr_seeded_kernel, dim3((unsigned)(((long)n * 16 + 255) / 256)), dim3(256), 0, s, make_view(t.hash),
                             ws.out.x(), ws.out.y(), ws.out.z(), n, cur->T16, cur->Rm, thr2, t.cloud.x(), t.cloud.y(), t.cloud.z(),
                             h->source_cov.p, t.cov.p, &cur->out, ws.last_nn.p, ws.nn_d2.p, d_pairs, d_shards, gicp_ball_cells());
        }
      } else {
      if (ball)
        hipLaunchKernelGGL(gicp_corr_ball_kernel, dim3((unsigned)(((long)n * 16 + 255) / 256)), dim3(256), 0, s, make_view(t.hash),
                           ws.out.x(), ws.out.y(), ws.out.z(), n, cur->T16, thr2, t.cloud.x(), t.cloud.y(), t.cloud.z(), &cur->out,
                           ws.last_nn.p, ws.nn_d2.p, d_work, gicp_ball_cells());
      // the first group is the first outer iteration (every point, one wave each); later groups only run the general search
      // on what the seeded kernel deferred (grid-stride over the list: a small grid, not 7 500 workgroups that exit)
      const unsigned full_grid = (unsigned)(((long)n * 64 + 255) / 256);
      const unsigned gen_grid = (ball && updates > 0) ? std::min(full_grid, 256u) : full_grid;
      hipLaunchKernelGGL(gicp_corr_search_kernel, dim3(gen_grid), dim3(256), 0, s, make_view(t.hash),
                         ws.out.x(), ws.out.y(), ws.out.z(), n, cur->T16, thr2, t.cloud.x(), t.cloud.y(), t.cloud.z(), &cur->out,
                         ws.last_nn.p, ws.nn_d2.p, d_work);
      hipLaunchKernelGGL(gicp_corr_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, cur->Rm, thr2, h->source_cov.p, t.cov.p,
                         t.cloud.x(), t.cloud.y(), t.cloud.z(), ws.last_nn.p, ws.nn_d2.p, d_pairs, &cur->count, &cur->out, d_work);
      }
      for (int it = 0; it < steps; it++) {
        hipLaunchKernelGGL(gicp_step_kernel, dim3(nblocks), dim3(GN_THREADS), 0, s, d_blk, updates, ws.out.x(), ws.out.y(), ws.out.z(), n,
                           d_pairs, d_partials, nblocks, ws.d_mailbox, token, updates + 1, corr_fused ? (const int*)d_shards : (const int*)nullptr);
        updates++;
      }
      return;
    }
    if (coop_corr) {
      hipLaunchKernelGGL(gicp_corr_search_kernel, dim3((unsigned)(((long)n * 64 + 255) / 256)), dim3(256), 0, s, make_view(t.hash),
                         ws.out.x(), ws.out.y(), ws.out.z(), n, d_blk->T16, thr2, t.cloud.x(), t.cloud.y(), t.cloud.z(), &d_blk->out,
                         ws.last_nn.p, ws.nn_d2.p, (const int*)nullptr);
      hipLaunchKernelGGL(gicp_corr_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_blk->Rm, thr2, h->source_cov.p, t.cov.p,
                         t.cloud.x(), t.cloud.y(), t.cloud.z(), ws.last_nn.p, ws.nn_d2.p, d_pairs, &d_blk->count, &d_blk->out, (int*)nullptr);
    } else
    hipLaunchKernelGGL(gicp_corr_kernel, dim3((unsigned)(((long)n * spread + NN_THREADS - 1) / NN_THREADS)), dim3(NN_THREADS), 0, s,
                       make_view(t.hash), ws.out.x(), ws.out.y(), ws.out.z(), n, d_blk->T16, d_blk->Rm, thr2, h->source_cov.p, t.cov.p,
                       t.cloud.x(), t.cloud.y(), t.cloud.z(), d_pairs, &d_blk->count, &d_blk->out, spread, ws.last_nn.p);
    for (int it = 0; it < steps; it++) {
      hipLaunchKernelGGL(gicp_gn_kernel, dim3(nblocks), dim3(GN_THREADS), 0, s, ws.out.x(), ws.out.y(), ws.out.z(), n, d_pairs,
                         &d_blk->st, &d_blk->out, d_partials);
      updates++;
      hipLaunchKernelGGL(gicp_update_kernel, dim3(1), dim3(256), 0, s, d_blk, d_partials, nblocks, ws.d_mailbox, token, updates);
    }
  }

  // everything up to the first three groups of launches
  int begin(lsr_handle_s* handle, const float* guess) {
    h = handle;
    if (!h->target || h->target->n == 0) { set_last_error("align before setInputTarget"); return LSR_ERR_NO_TARGET; }
    if (!h->has_source) { set_last_error("align before setInputSource"); return LSR_ERR_NO_SOURCE; }
    int st = ensure_covariances(h);
    if (st) return st;
    s = h->stream;
    n = (int)h->source.n;
    float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const float* G = guess ? guess : I16;

    // workspace: out cloud | pairs | partials | per-align block {inner state, T16, Rm, outer state, count} | guess
    nblocks = std::max(1, std::min((n + GN_THREADS - 1) / GN_THREADS, 512));
    GicpWorkspace& ws = h->gicp_ws;
    if ((st = ws.out.resize(n))) return st;
    if ((st = ws.pairs.reserve((size_t)n * sizeof(PairRec)))) return st;
    if ((st = ws.buf.reserve((size_t)2 * nblocks * 32 + 64))) return st;   // two banks of partial rows (fused chain)
    if ((st = ws.state.reserve(2 * sizeof(IterBlock) + 256))) return st;   // the block is double buffered by step parity
    if ((st = ws.pin.reserve(sizeof(IterBlock) + 64))) return st;
    d_partials = ws.buf.p;
    d_blk = reinterpret_cast<IterBlock*>(ws.state.p);
    d_pairs = reinterpret_cast<PairRec*>(ws.pairs.p);

    if (!ws.d_mailbox) {
      if ((st = ws.mailbox.reserve(1, hipHostMallocMapped | hipHostMallocCoherent))) return st;
      std::memset(ws.mailbox.p, 0, sizeof(GicpMailbox));
      LSR_HIP(hipHostGetDevicePointer((void**)&ws.d_mailbox, ws.mailbox.p, 0));
    }
    token = ++ws.token;
    if (token == 0) token = ++ws.token;  // 0 is the mailbox's idle value
    t_begin = std::chrono::steady_clock::now();

    // ---- the whole align as ONE upload: outer state at the entry of computeTransformation + the first outer iteration
    IterBlock* hb = reinterpret_cast<IterBlock*>(ws.pin.p);
    std::memset(hb, 0, sizeof(IterBlock));
    OuterState& O = hb->out;
    std::memcpy(O.trans, I16, sizeof(I16));
    std::memcpy(O.prev, I16, sizeof(I16));
    std::memcpy(O.G, G, sizeof(I16));
    O.rot_eps = h->gicp.rot_eps;
    O.trans_eps = h->gicp.trans_eps;
    O.max_iterations = h->gicp.max_iterations;
    O.corr_mark = -1;
    O.token = token;
    hb->st.max_inner = h->gicp.max_inner;
    gicp_begin_outer(*hb);
    thr2 = (float)(h->gicp.max_corr_dist * h->gicp.max_corr_dist);

    // ---- launch chain.  A group = one correspondence pass + `steps` x (accumulate, update); every launch gates itself on
    // the device-side phase, so a group enqueued too early (the previous inner loop still running) or too late (the align
    // over) costs ~2 us per launch and nothing else.  The host keeps groups queued ahead and polls the mailbox.
    spread = (n <= 65536) ? 2 : 1;
    if ((st = ws.last_nn.reserve((size_t)n + 1))) return st;
    updates = 0;
    coop_corr = nn_coop_enabled();
    if (coop_corr && (st = ws.nn_d2.reserve((size_t)n + 1))) return st;
    fused = gicp_fused_enabled();
    // seeded 16-lane search for the outer iterations after the first (env LSR_GICP_BALL=0: the general search every time)
    static const bool ball_on = [] { const char* e = getenv("LSR_GICP_BALL"); return !(e && e[0] == '0'); }();
    ball = fused && ball_on;
    // ... in ONE launch per outer iteration (env LSR_GICP_CORR_FUSED=0: seeded search, general search and pair records as three)
    static const bool corr_fused_on = [] { const char* e = getenv("LSR_GICP_CORR_FUSED"); return !(e && e[0] == '0'); }();
    corr_fused = ball && corr_fused_on;
    d_work = nullptr;
    d_shards = nullptr;
    int* zero_words = nullptr;
    int n_zero = 0;
    if (corr_fused) {   // no work list; the pair counters start an align at zero
      if ((st = ws.count_shards.reserve((size_t)GICP_COUNT_SHARDS * GICP_SHARD_STRIDE))) return st;
      d_shards = ws.count_shards.p;
      zero_words = d_shards; n_zero = GICP_COUNT_SHARDS * GICP_SHARD_STRIDE;
    } else if (ball) {
      if ((st = ws.corr_work.reserve((size_t)n + 2))) return st;
      d_work = ws.corr_work.p;
      zero_words = d_work; n_zero = 1;
    }
    // iteration block + zeroed counters + guess-moved source: one launch (gicp_begin_align_kernel)
    hipLaunchKernelGGL(gicp_begin_align_kernel, dim3((n + 255) / 256), dim3(256), 0, s, h->source.x(), h->source.y(), h->source.z(), n, *hb, d_blk,
                       zero_words, n_zero, ws.out.x(), ws.out.y(), ws.out.z());
    // the first outer iteration typically needs 3-4 Gauss-Newton steps, later ones one or two; the fused chain needs one step
    // more per outer iteration (the step that finds the loop finished accumulates nothing)
    enqueue_group(fused ? 5 : 4);
    enqueue_group(3);
    enqueue_group(3);
    LSR_HIP(hipGetLastError());
    hard_cap = (long)(h->gicp.max_iterations + 2) * (h->gicp.max_inner + 2) + 16;  // update launches an align can need
    last_progress = 0;
    t_progress = std::chrono::steady_clock::now();
    done = false;
    return LSR_OK;
  }

  // one poll of the mailbox: tops the queue up when fewer than two groups are left; sets `done` when the flag is up
  int poll(unsigned long long spins) {
    const GicpMailbox* mb = h->gicp_ws.mailbox.p;
    if (__atomic_load_n(&mb->done, __ATOMIC_ACQUIRE) == token) { done = true; return LSR_OK; }
    const unsigned long long pr = __atomic_load_n(&mb->progress, __ATOMIC_RELAXED);
    const int ran = ((unsigned int)(pr >> 32) == token) ? (int)(unsigned int)pr : 0;
    if (updates - ran < 5) {  // fewer than two groups left in the queue: top up
      if (updates > hard_cap) { set_last_error("GICP launch chain did not finish within its launch cap"); return LSR_ERR_HIP; }
      enqueue_group(3);
      enqueue_group(3);
      LSR_HIP(hipGetLastError());
      return LSR_OK;
    }
    if ((spins & 0x3FFF) == 0 || h->scratch.wait_mode == WAIT_SLEEP) {  // a device that stops making progress must not hang the caller forever
      const auto now = std::chrono::steady_clock::now();
      if (pr != last_progress) { last_progress = pr; t_progress = now; }
      if (std::chrono::duration<double>(now - t_progress).count() > 30.0) {
        set_last_error(std::string("GICP launch chain made no progress for 30 s (stream: ") + hipGetErrorString(hipStreamQuery(s)) + ")");
        return LSR_ERR_HIP;
      }
    }
    return LSR_OK;
  }

  void finish(float* final_T, lsr_result* res) {
    const GicpMailbox* mb = h->gicp_ws.mailbox.p;
    // host clock from the first enqueue to the raised flag (the launches still queued exit at their head)
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    std::memcpy(h->final_T, mb->final_T, sizeof(float) * 16);
    h->converged = mb->converged;
    if (final_T) std::memcpy(final_T, h->final_T, sizeof(float) * 16);
    if (res) {
      res->converged = h->converged;
      res->iterations = mb->nr_iterations;
      res->score = mb->last_cost;
      res->n_evaluations = mb->gn_steps;
      res->n_correspondences = mb->last_cnt;
      res->gpu_ms = ms;
    }
  }
};

void wait_between_polls(int wait_mode) {
  if (wait_mode == WAIT_YIELD) std::this_thread::yield();
  else if (wait_mode == WAIT_SLEEP) std::this_thread::sleep_for(std::chrono::microseconds(20));
  else __builtin_ia32_pause();
}
}  // namespace

int gicp_align(lsr_handle_s* h, const float* guess, float* final_T, lsr_result* res) {
  GicpChain c;
  int st = c.begin(h, guess);
  if (st) return st;
  for (unsigned long long spins = 1; !c.done; spins++) {
    if ((st = c.poll(spins))) return st;
    if (!c.done) wait_between_polls(h->scratch.wait_mode);
  }
  c.finish(final_T, res);
  return LSR_OK;
}

// B independent GICP registrations side by side: every chain on its own object's stream, one host loop feeds them all (a
// chain is a sequence of small dependent launches, far from filling the chip: B of them overlap almost completely).  Objects
// that share a stream still work — their chains then simply queue one after the other.
int gicp_align_batch(lsr_handle_s* const* hs, int B, const float* guesses, float* finals, lsr_result* results) {
  std::vector<GicpChain> chains((size_t)B);
  int st;
  for (int b = 0; b < B; b++)
    if ((st = chains[b].begin(hs[b], guesses ? guesses + 16 * b : nullptr))) {
      for (int a = 0; a < b; a++) (void)hipStreamSynchronize(hs[a]->stream);   // nothing of a failed batch stays in flight
      return st;
    }
  int n_done = 0, first_error = LSR_OK;
  for (unsigned long long spins = 1; n_done < B; spins++) {
    for (int b = 0; b < B; b++) {
      GicpChain& c = chains[b];
      if (c.done) continue;
      st = c.poll(spins);
      if (st) { if (!first_error) first_error = st; c.done = true; n_done++; continue; }
      if (c.done) n_done++;
    }
    if (n_done < B) wait_between_polls(hs[0]->scratch.wait_mode);
  }
  if (first_error) {
    for (int b = 0; b < B; b++) (void)hipStreamSynchronize(hs[b]->stream);
    return first_error;
  }
  for (int b = 0; b < B; b++) chains[b].finish(finals ? finals + 16 * b : nullptr, results ? results + b : nullptr);
  return LSR_OK;
}

}  // namespace lsr
