// NDT kernels for gfx950 (CDNA4).  See ndt.hpp for the decomposition.  Reference behaviour:
// pclomp::NormalDistributionsTransform / pclomp::VoxelGridCovariance as called from
// scanmatcher/src/scanmatcher_component.cpp:105-113,275,307,353 and
// graph_based_slam/src/graph_based_slam_component.cpp:64-72,227,230 — arithmetic restated in
// SURVEY.md §9 (the ndt_omp sources are not vendored in the reference snapshot).
//
// Layout decisions (HBM-side):
//  * source/target clouds are SoA fp32 planes -> every wave-wide load is one coalesced 256-B burst;
//  * the voxel table is a dense int32 cell->slot map plus compact 64-B leaf records
//    {mean.xyz, icov upper triangle} read as three 16-B loads; at the reference resolutions the
//    whole table (<= a few MB) is L2/Infinity-Cache resident, HBM only streams the source planes;
//  * the 29 sums of a pass travel: registers -> quad sum (DPP) -> LDS transpose -> one 256-B partial row per
//    workgroup (plain stores) -> EVERY workgroup of the next launch sums the rows in the same fixed order at its
//    head (deterministic, no floating-point atomics, no inter-workgroup synchronisation inside a launch);
//  * progress and the final result go straight into a pinned host mailbox (NdtMailbox, ndt.hpp): the host feeds
//    the launch chain by polling it, nothing is copied back.
#include "ndt.hpp"

#include <cmath>
#include <cstdlib>

#include "grid_device.hpp"
#include "ndt_point.hpp"
#include "sort.hpp"

namespace lsr {

// ===========================================================================================
// Small host/device maths shared by the controller and the host-side state preparation
// ===========================================================================================
namespace {

__host__ __device__ inline void angle_tables(const double* p, bool with_hessian, int d1_sign, float* jang, float* hang) {
  double cx, cy, cz, sx, sy, sz;
  if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(p[3]); sx = sin(p[3]); }
  if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(p[4]); sy = sin(p[4]); }
  if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(p[5]); sz = sin(p[5]); }
  // rows a..h of eq. 6.19
  jang[0] = (float)(-sx * sz + cx * sy * cz); jang[1] = (float)(-sx * cz - cx * sy * sz); jang[2] = (float)(-cx * cy);
  jang[3] = (float)(cx * sz + sx * sy * cz);  jang[4] = (float)(cx * cz - sx * sy * sz);  jang[5] = (float)(-sx * cy);
  jang[6] = (float)(-sy * cz);                jang[7] = (float)(sy * sz);                 jang[8] = (float)(cy);
  jang[9] = (float)(sx * cy * cz);            jang[10] = (float)(-sx * cy * sz);          jang[11] = (float)(sx * sy);
  jang[12] = (float)(-cx * cy * cz);          jang[13] = (float)(cx * cy * sz);           jang[14] = (float)(-cx * sy);
  jang[15] = (float)(-cy * sz);               jang[16] = (float)(-cy * cz);               jang[17] = 0.f;
  jang[18] = (float)(cx * cz - sx * sy * sz); jang[19] = (float)(-cx * sz - sx * sy * cz); jang[20] = 0.f;
  jang[21] = (float)(sx * cz + cx * sy * sz); jang[22] = (float)(cx * sy * cz - sx * sz);  jang[23] = 0.f;
  if (with_hessian) {
    // rows a2,a3,b2,b3,c2,c3,d1,d2,d3,e1,e2,e3,f1,f2,f3 of eq. 6.21
    hang[0] = (float)(-cx * sz - sx * sy * cz); hang[1] = (float)(-cx * cz + sx * sy * sz); hang[2] = (float)(sx * cy);
    hang[3] = (float)(-sx * sz + cx * sy * cz); hang[4] = (float)(-cx * sy * sz - sx * cz); hang[5] = (float)(-cx * cy);
    hang[6] = (float)(cx * cy * cz);            hang[7] = (float)(-cx * cy * sz);           hang[8] = (float)(cx * sy);
    hang[9] = (float)(sx * cy * cz);            hang[10] = (float)(-sx * cy * sz);          hang[11] = (float)(sx * sy);
    hang[12] = (float)(-sx * cz - cx * sy * sz); hang[13] = (float)(sx * sz - cx * sy * cz); hang[14] = 0.f;
    hang[15] = (float)(cx * cz - sx * sy * sz); hang[16] = (float)(-sx * sy * cz - cx * sz); hang[17] = 0.f;
    hang[18] = (float)(-cy * cz);               hang[19] = (float)(cy * sz);                hang[20] = (float)(d1_sign >= 0 ? sy : -sy);
    hang[21] = (float)(-sx * sy * cz);          hang[22] = (float)(sx * sy * sz);           hang[23] = (float)(sx * cy);
    hang[24] = (float)(cx * sy * cz);           hang[25] = (float)(-cx * sy * sz);          hang[26] = (float)(-cx * cy);
    hang[27] = (float)(sy * sz);                hang[28] = (float)(sy * cz);                hang[29] = 0.f;
    hang[30] = (float)(-sx * cy * sz);          hang[31] = (float)(-sx * cy * cz);          hang[32] = 0.f;
    hang[33] = (float)(cx * cy * sz);           hang[34] = (float)(cx * cy * cz);           hang[35] = 0.f;
    hang[36] = (float)(-cy * cz);               hang[37] = (float)(cy * sz);                hang[38] = 0.f;
    hang[39] = (float)(-cx * sz - sx * sy * cz); hang[40] = (float)(-cx * cz + sx * sy * sz); hang[41] = 0.f;
    hang[42] = (float)(-sx * sz + cx * sy * cz); hang[43] = (float)(-cx * sy * sz - sx * cz); hang[44] = 0.f;
    hang[45] = hang[46] = hang[47] = 0.f;
  }
}

// Eigen 3.4 MatrixBase::eulerAngles(0,1,2) on the fp32 rotation block (host only; used once per align).
inline void euler012_f(const float* M /*col-major 4x4*/, float* res) {
  auto c = [&](int r, int cc) { return M[cc * 4 + r]; };
  const float PI_F = 3.14159265358979323846f;
  res[0] = atan2f(c(1, 2), c(2, 2));
  float c2 = sqrtf(c(0, 0) * c(0, 0) + c(0, 1) * c(0, 1));
  if (res[0] > 0.f) {
    res[0] -= PI_F;
    res[1] = atan2f(-c(0, 2), -c2);
  } else {
    res[1] = atan2f(-c(0, 2), c2);
  }
  float s1 = sinf(res[0]), c1 = cosf(res[0]);
  res[2] = atan2f(s1 * c(2, 0) - c1 * c(1, 0), c1 * c(1, 1) - s1 * c(2, 1));
  res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2];
}

}  // namespace

void ndt_gauss_constants(double resolution, double outlier_ratio, double* d1, double* d2) {
  double c1 = 10 * (1 - outlier_ratio);
  double c2 = outlier_ratio / pow(resolution, 3);
  double d3 = -log(c2);
  *d1 = -log(c1 + c2) - d3;
  *d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / *d1);
}

void ndt_fill_align_constants(NdtState& st, const NdtParamsHost& prm, int n_points) {
  ndt_gauss_constants(prm.resolution, prm.outlier_ratio, &st.d1, &st.d2);
  st.step_max = prm.step_size;
  st.step_min = prm.trans_eps / 2;
  st.eps = prm.trans_eps;
  st.max_iter = prm.max_iterations;
  st.n_points = n_points;
  st.d1_sign = prm.d1_sign;
}

// Host: state at the entry of computeTransformation: output = guess * input,
// p from guess (fp32 Euler XYZ), first pass with Hessian.
void ndt_fill_initial_state(NdtState& st, const float* guess16 /*nullable*/, const NdtParamsHost& prm, int n_points) {
  std::memset(&st, 0, sizeof(st));
  ndt_fill_align_constants(st, prm, n_points);
  float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float* G = guess16 ? guess16 : I16;
  std::memcpy(st.final_T, G, sizeof(I16));
  st.T[0] = G[0]; st.T[1] = G[4]; st.T[2] = G[8];  st.T[3] = G[12];
  st.T[4] = G[1]; st.T[5] = G[5]; st.T[6] = G[9];  st.T[7] = G[13];
  st.T[8] = G[2]; st.T[9] = G[6]; st.T[10] = G[10]; st.T[11] = G[14];
  float e[3];
  euler012_f(G, e);
  st.p[0] = G[12]; st.p[1] = G[13]; st.p[2] = G[14];
  st.p[3] = e[0]; st.p[4] = e[1]; st.p[5] = e[2];
  for (int i = 0; i < 6; i++) st.x_t[i] = st.p[i];
  angle_tables(st.p, true, st.d1_sign, st.jang, st.hang);
  st.want_hessian = 1;
  st.phase = PH_INIT;
  st.done = 0;
}

void ndt_fill_diag_state(NdtState& st, const double* p6, const float* T16, int compute_hessian, const NdtParamsHost& prm,
                         int n_points) {
  std::memset(&st, 0, sizeof(st));
  ndt_fill_align_constants(st, prm, n_points);
  for (int i = 0; i < 6; i++) st.p[i] = st.x_t[i] = p6[i];
  if (T16) {
    st.T[0] = T16[0]; st.T[1] = T16[4]; st.T[2] = T16[8];  st.T[3] = T16[12];
    st.T[4] = T16[1]; st.T[5] = T16[5]; st.T[6] = T16[9];  st.T[7] = T16[13];
    st.T[8] = T16[2]; st.T[9] = T16[6]; st.T[10] = T16[10]; st.T[11] = T16[14];
  } else {
    pose_to_T12(p6, st.T);
  }
  angle_tables(p6, true, st.d1_sign, st.jang, st.hang);
  st.want_hessian = compute_hessian ? 1 : 0;
  st.phase = PH_DIAG;
}

// ===========================================================================================
// K3 + K4: derivative pass with fused controller
// ===========================================================================================
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// The controller works on an LDS image of NdtState: typed LDS pointers let the compiler emit ds_read /
// ds_write instead of flat accesses even though the controller is not inlined into the kernel.
typedef __attribute__((address_space(3))) NdtState LdsState;
typedef __attribute__((address_space(3))) double LdsDouble;

// fp64 division / square root for the More-Thuente interpolation formulas, on ONE lane with the rest of the chip
// waiting: hardware seed (v_rcp_f64 / v_rsq_f64) + two Newton steps + one residual correction = ~10 dependent
// instructions instead of the ~35 of the IEEE sequences (div_scale / div_fmas / div_fixup).  Results are within 1 ulp
// of the correctly rounded ones; badly scaled operands take the IEEE path.
__device__ __forceinline__ double mt_div(double a, double b) {
  if (!(fabs(b) > 1e-290 && fabs(b) < 1e290)) return a / b;  // zero / denormal / huge / NaN divisor: the IEEE sequence
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double mt_sqrt(double x) {
  if (!(x > 1e-290 && x < 1e290)) return sqrt(x);  // zero / negative / denormal / huge / NaN: the IEEE sequence
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  const double s = x * y;
  return fma(fma(-s, s, x), 0.5 * y, s);
}

// ---- More-Thuente helpers ----
__device__ __forceinline__ double mt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t,
                                 double f_t, double g_t) {
  if (f_t > f_l) {
    double z = mt_div(3 * (f_t - f_l), a_t - a_l) - g_t - g_l;
    double w = mt_sqrt(z * z - g_t * g_l);
    double a_c = a_l + mt_div((a_t - a_l) * (w - g_l - z), g_t - g_l + 2 * w);
    double a_q = a_l - mt_div(0.5 * (a_l - a_t) * g_l, g_l - mt_div(f_l - f_t, a_l - a_t));
    if (fabs(a_c - a_l) < fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {
    double z = mt_div(3 * (f_t - f_l), a_t - a_l) - g_t - g_l;
    double w = mt_sqrt(z * z - g_t * g_l);
    double a_c = a_l + mt_div((a_t - a_l) * (w - g_l - z), g_t - g_l + 2 * w);
    double a_s = a_l - mt_div(a_l - a_t, g_l - g_t) * g_l;
    if (fabs(a_c - a_t) >= fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (fabs(g_t) <= fabs(g_l)) {
    double z = mt_div(3 * (f_t - f_l), a_t - a_l) - g_t - g_l;
    double w = mt_sqrt(z * z - g_t * g_l);
    double a_c = a_l + mt_div((a_t - a_l) * (w - g_l - z), g_t - g_l + 2 * w);
    double a_s = a_l - mt_div(a_l - a_t, g_l - g_t) * g_l;
    double a_n = (fabs(a_c - a_t) < fabs(a_s - a_t)) ? a_c : a_s;
    if (a_t > a_l) return fmin(a_t + 0.66 * (a_u - a_t), a_n);
    return fmax(a_t + 0.66 * (a_u - a_t), a_n);
  } else {
    double z = mt_div(3 * (f_t - f_u), a_t - a_u) - g_t - g_u;
    double w = mt_sqrt(z * z - g_t * g_u);
    return a_u + mt_div((a_t - a_u) * (w - g_u - z), g_t - g_u + 2 * w);
  }
}

struct MtInterval { double a_l, f_l, g_l, a_u, f_u, g_u; };
__device__ __forceinline__ bool mt_update_interval(MtInterval& I, double a_t, double f_t, double g_t) {
  if (f_t > I.f_l) {
    I.a_u = a_t; I.f_u = f_t; I.g_u = g_t;
    return false;
  } else if (g_t * (I.a_l - a_t) > 0) {
    I.a_l = a_t; I.f_l = f_t; I.g_l = g_t;
    return false;
  } else if (g_t * (I.a_l - a_t) < 0) {
    I.a_u = I.a_l; I.f_u = I.f_l; I.g_u = I.g_l;
    I.a_l = a_t; I.f_l = f_t; I.g_l = g_t;
    return false;
  }
  return true;
}

// delta = H^{-1} (-g) by Gauss-Jordan elimination with partial pivoting on ONE WAVE: lane 8 r + c holds element (r, c) of the
// augmented 6 x 7 matrix [H | -g] in a register, every step is a handful of cross-lane reads.  The reference calls
// JacobiSVD::solve; for a non-singular 6x6 both give H^{-1} b.  A column whose pivot vanishes is dropped = its unknown set to
// 0, which is the SVD's minimum-norm answer for the degenerate all-zero Hessian of a scan that overlaps no voxel.
// Hu: the 21 values of the upper triangle, row-major (0,0..5) (1,1..5) ... (5,5), g: the gradient, out[0..5] receives delta
// (all three in LDS; the caller fences the wave's LDS traffic around the call).
// Why a wave and not one lane with the matrix in registers (rounds 1-2: ~166 VGPRs, fully unrolled): the register count of a
// kernel is the maximum over everything it calls, for EVERY wave — that one-lane solver capped the derivative kernels at two
// waves per SIMD (one 512-thread workgroup per CU), which is what bounds the passes that have more workgroups than CUs
// (cfg 5, candidate batches).  Here the solver needs a dozen registers; the kernels fit twice the waves.
__device__ __forceinline__ void solve6_wave(const LdsDouble* Hu, const LdsDouble* g, LdsDouble* out) {
  const int lane = threadIdx.x & 63;
  const int r = lane >> 3, c = lane & 7;
  double v = 0.0;
  if (r < 6 && c < 6) {
    const int i = min(r, c), j = max(r, c);
    v = Hu[6 * i - (i * (i - 1)) / 2 + (j - i)];   // row i of the upper triangle starts at 6 i - i (i - 1) / 2
  } else if (r < 6 && c == 6) {
    v = -g[r];
  }
  double scale = (r < 6 && c < 6) ? fabs(v) : 0.0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) scale = fmax(scale, __shfl_xor(scale, m, 64));
  unsigned int dropped = 0u;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    // pivot search over rows k..5 of column k (the same answer in every lane)
    double best = fabs(__shfl(v, k * 8 + k, 64));
    int piv = k;
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double t = fabs(__shfl(v, i * 8 + k, 64));
      if (t > best) { best = t; piv = i; }
    }
    if (!(best > scale * 1e-300) || !(best > 0)) { dropped |= (1u << k); continue; }
    // bring the pivot row to position k
    const double from_piv = __shfl(v, piv * 8 + c, 64), from_k = __shfl(v, k * 8 + c, 64);
    v = (r == k) ? from_piv : ((r == piv) ? from_k : v);
    // eliminate column k from every other row
    const double akk = __shfl(v, k * 8 + k, 64), akj = __shfl(v, k * 8 + c, 64), aik = __shfl(v, r * 8 + k, 64);
    const double f = aik * (1.0 / akk);
    if (r != k && r < 6) v -= f * akj;
  }
  const int jj = min(lane, 5);
  const double num = __shfl(v, jj * 8 + 6, 64), den = __shfl(v, jj * 8 + jj, 64);
  if (lane < 6) out[lane] = ((dropped >> lane) & 1u) ? 0.0 : num / den;
}

// The angular coefficient tables of eq. 6.19 (jang, 24 entries) and eq. 6.21 (hang, 48 entries) as DATA: every entry is
//   s1 * (f[u1] * f[u2]) + s2 * ((f[v1] * f[v2]) * f[v3]),   f = {1, cx, cy, cz, sx, sy, sz, 0},  s in {0, +1, -1}
// so 72 lanes evaluate the 72 entries with one uniform instruction sequence (eight different formulas on one lane
// cost the sum of all of them; eight formulas on eight lanes of one wave cost the same: divergence serialises).
// The products are formed in the same association as the scalar code in angle_tables() ((a*b)*c).
namespace angtab {
enum { ONE = 0, CX = 1, CY = 2, CZ = 3, SX = 4, SY = 5, SZ = 6, NIL = 7, P = 1, M = 2, Z = 0 };
constexpr unsigned int e(int s1, int u1, int u2, int s2, int v1, int v2, int v3) {
  return (unsigned)u1 | ((unsigned)u2 << 3) | ((unsigned)v1 << 6) | ((unsigned)v2 << 9) | ((unsigned)v3 << 12) | ((unsigned)s1 << 15) |
         ((unsigned)s2 << 17);
}
constexpr unsigned int two(int s1, int u1, int u2) { return e(s1, u1, u2, Z, ONE, ONE, ONE); }
constexpr unsigned int three(int s2, int v1, int v2, int v3) { return e(Z, ONE, ONE, s2, v1, v2, v3); }
constexpr unsigned int zero() { return e(Z, ONE, ONE, Z, ONE, ONE, ONE); }
}  // namespace angtab

#define LSR_ANGLE_ENTRIES \
    angtab::e(angtab::M, angtab::SX, angtab::SZ, angtab::P, angtab::CX, angtab::SY, angtab::CZ), \
    angtab::e(angtab::M, angtab::SX, angtab::CZ, angtab::M, angtab::CX, angtab::SY, angtab::SZ), \
    angtab::two(angtab::M, angtab::CX, angtab::CY), \
    angtab::e(angtab::P, angtab::CX, angtab::SZ, angtab::P, angtab::SX, angtab::SY, angtab::CZ), \
    angtab::e(angtab::P, angtab::CX, angtab::CZ, angtab::M, angtab::SX, angtab::SY, angtab::SZ), \
    angtab::two(angtab::M, angtab::SX, angtab::CY), \
    angtab::two(angtab::M, angtab::SY, angtab::CZ), angtab::two(angtab::P, angtab::SY, angtab::SZ), angtab::two(angtab::P, angtab::CY, angtab::ONE), \
    angtab::three(angtab::P, angtab::SX, angtab::CY, angtab::CZ), angtab::three(angtab::M, angtab::SX, angtab::CY, angtab::SZ), \
    angtab::two(angtab::P, angtab::SX, angtab::SY), \
    angtab::three(angtab::M, angtab::CX, angtab::CY, angtab::CZ), angtab::three(angtab::P, angtab::CX, angtab::CY, angtab::SZ), \
    angtab::two(angtab::M, angtab::CX, angtab::SY), \
    angtab::two(angtab::M, angtab::CY, angtab::SZ), angtab::two(angtab::M, angtab::CY, angtab::CZ), angtab::zero(), \
    angtab::e(angtab::P, angtab::CX, angtab::CZ, angtab::M, angtab::SX, angtab::SY, angtab::SZ), \
    angtab::e(angtab::M, angtab::CX, angtab::SZ, angtab::M, angtab::SX, angtab::SY, angtab::CZ), angtab::zero(), \
    angtab::e(angtab::P, angtab::SX, angtab::CZ, angtab::P, angtab::CX, angtab::SY, angtab::SZ), \
    angtab::e(angtab::M, angtab::SX, angtab::SZ, angtab::P, angtab::CX, angtab::SY, angtab::CZ), angtab::zero(), \
    angtab::e(angtab::M, angtab::CX, angtab::SZ, angtab::M, angtab::SX, angtab::SY, angtab::CZ), \
    angtab::e(angtab::M, angtab::CX, angtab::CZ, angtab::P, angtab::SX, angtab::SY, angtab::SZ), angtab::two(angtab::P, angtab::SX, angtab::CY), \
    angtab::e(angtab::M, angtab::SX, angtab::SZ, angtab::P, angtab::CX, angtab::SY, angtab::CZ), \
    angtab::e(angtab::M, angtab::SX, angtab::CZ, angtab::M, angtab::CX, angtab::SY, angtab::SZ), angtab::two(angtab::M, angtab::CX, angtab::CY), \
    angtab::three(angtab::P, angtab::CX, angtab::CY, angtab::CZ), angtab::three(angtab::M, angtab::CX, angtab::CY, angtab::SZ), \
    angtab::two(angtab::P, angtab::CX, angtab::SY), \
    angtab::three(angtab::P, angtab::SX, angtab::CY, angtab::CZ), angtab::three(angtab::M, angtab::SX, angtab::CY, angtab::SZ), \
    angtab::two(angtab::P, angtab::SX, angtab::SY), \
    angtab::e(angtab::M, angtab::SX, angtab::CZ, angtab::M, angtab::CX, angtab::SY, angtab::SZ), \
    angtab::e(angtab::P, angtab::SX, angtab::SZ, angtab::M, angtab::CX, angtab::SY, angtab::CZ), angtab::zero(), \
    angtab::e(angtab::P, angtab::CX, angtab::CZ, angtab::M, angtab::SX, angtab::SY, angtab::SZ), \
    angtab::e(angtab::M, angtab::CX, angtab::SZ, angtab::M, angtab::SX, angtab::SY, angtab::CZ), angtab::zero(), \
    angtab::two(angtab::M, angtab::CY, angtab::CZ), angtab::two(angtab::P, angtab::CY, angtab::SZ), \
    angtab::two(angtab::P, angtab::SY, angtab::ONE), \
    angtab::three(angtab::M, angtab::SX, angtab::SY, angtab::CZ), angtab::three(angtab::P, angtab::SX, angtab::SY, angtab::SZ), \
    angtab::two(angtab::P, angtab::SX, angtab::CY), \
    angtab::three(angtab::P, angtab::CX, angtab::SY, angtab::CZ), angtab::three(angtab::M, angtab::CX, angtab::SY, angtab::SZ), \
    angtab::two(angtab::M, angtab::CX, angtab::CY), \
    angtab::two(angtab::P, angtab::SY, angtab::SZ), angtab::two(angtab::P, angtab::SY, angtab::CZ), angtab::zero(), \
    angtab::three(angtab::M, angtab::SX, angtab::CY, angtab::SZ), angtab::three(angtab::M, angtab::SX, angtab::CY, angtab::CZ), angtab::zero(), \
    angtab::three(angtab::P, angtab::CX, angtab::CY, angtab::SZ), angtab::three(angtab::P, angtab::CX, angtab::CY, angtab::CZ), angtab::zero(), \
    angtab::two(angtab::M, angtab::CY, angtab::CZ), angtab::two(angtab::P, angtab::CY, angtab::SZ), angtab::zero(), \
    angtab::e(angtab::M, angtab::CX, angtab::SZ, angtab::M, angtab::SX, angtab::SY, angtab::CZ), \
    angtab::e(angtab::M, angtab::CX, angtab::CZ, angtab::P, angtab::SX, angtab::SY, angtab::SZ), angtab::zero(), \
    angtab::e(angtab::M, angtab::SX, angtab::SZ, angtab::P, angtab::CX, angtab::SY, angtab::CZ), \
    angtab::e(angtab::M, angtab::SX, angtab::CZ, angtab::M, angtab::CX, angtab::SY, angtab::SZ), angtab::zero(), \
    angtab::zero(), angtab::zero(), angtab::zero()

__device__ __constant__ unsigned int k_angle_entries[72] = {LSR_ANGLE_ENTRIES};
static const unsigned int k_angle_entries_host[72] = {LSR_ANGLE_ENTRIES};


__host__ __device__ inline double angle_entry_value(unsigned int ent, const double* f) {
  const double t1 = f[ent & 7u] * f[(ent >> 3) & 7u];
  const double t2 = (f[(ent >> 6) & 7u] * f[(ent >> 9) & 7u]) * f[(ent >> 12) & 7u];
  const unsigned int s1 = (ent >> 15) & 3u, s2 = (ent >> 17) & 3u;
  const double a = (s1 == 0u) ? 0.0 : ((s1 == 1u) ? t1 : -t1);
  const double b = (s2 == 0u) ? 0.0 : ((s2 == 1u) ? t2 : -t2);
  return a + b;
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() makes hipcc drain vmcnt too, which
// would stall every wave on the voxel-table DMA (global_load_lds) long before its data is needed; the head's
// intermediate barriers only order LDS accesses, the DMA is drained once, by the last __syncthreads() before the points
// are evaluated.
__device__ __forceinline__ void barrier_lds_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Executed by ALL threads of the workgroup (uniform control flow, two barriers).
// lanes 0-2: fp64 sin/cos of the three angles (with the reference's 1e-4 snap), lanes 3-5: fp32 sin/cos (the
// reference composes the point transform from float-cast angles); then lanes 0..23 (0..71 when the Hessian tables
// are refreshed) evaluate one table entry each and the last lane of the workgroup builds T + final_T.
// `ent` = k_angle_entries[tid], fetched with the head loads of the kernel (off the critical path).
template <int THREADS>
__device__ __forceinline__ void build_request(NdtState* S, double* f /*8*/, float* cs_f /*6*/, const unsigned int ent) {
  const int tid = threadIdx.x;
  const int mode = S->pad1;
  if (mode == 0) return;  // uniform: every thread reads the same LDS word
  if (tid < 3) {
    const double a = S->x_t[3 + tid];
    double sn, cn;
    if (fabs(a) < 10e-5) { cn = 1.0; sn = 0.0; } else { sincos(a, &sn, &cn); }
    f[1 + tid] = cn;
    f[4 + tid] = sn;
  } else if (tid < 6) {
    const float a = (float)S->x_t[tid];
    float sn, cn;
    sincosf(a, &sn, &cn);
    cs_f[tid - 3] = cn;
    cs_f[tid] = sn;
  } else if (tid == 6) {
    f[0] = 1.0;
    f[7] = 0.0;
  }
  barrier_lds_only();
  const int nent = (mode & 2) ? 72 : 24;
  if (tid < nent) {
    double val = angle_entry_value(ent, f);
    if (tid == 24 + 20 && S->d1_sign < 0) val = -val;
    if (tid < 24) S->jang[tid] = (float)val; else S->hang[tid - 24] = (float)val;
    if (tid == 0) S->pad1 = 0;
  } else if (tid == THREADS - 1) {
    // fp32 (Translation * Rx * Ry * Rz), as pose_to_T12
    float* T = S->T;
    compose_R12(cs_f[0], cs_f[1], cs_f[2], cs_f[3], cs_f[4], cs_f[5], T);
    T[3] = (float)S->x_t[0]; T[7] = (float)S->x_t[1]; T[11] = (float)S->x_t[2];
    T12_to_colmajor16(T, S->final_T);  // final_transformation_ is assigned before every MT pass
  }
  barrier_lds_only();
}

// Ordering of LDS traffic inside ONE wave: its lanes run in lockstep and the LDS unit executes a wave's operations in
// order, so "lane 0 wrote, every lane reads" needs no workgroup barrier — only that the compiler keeps the order and the
// writes have been issued.
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// build_request() for a caller that runs the whole controller step on wave 0 (quad kernel): same work, the 64 lanes of
// one wave, no workgroup barrier.  ent_a = k_angle_entries[lane], ent_b = k_angle_entries[64 + lane] (lane < 8).
__device__ __forceinline__ void build_request_wave0(NdtState* S, double* f /*8*/, float* cs_f /*6*/, const unsigned int ent_a,
                                                    const unsigned int ent_b) {
  const int lane = threadIdx.x;  // 0..63
  const int mode = S->pad1;
  if (mode == 0) return;  // wave-uniform
  if (lane < 3) {
    const double a = S->x_t[3 + lane];
    double sn, cn;
    if (fabs(a) < 10e-5) { cn = 1.0; sn = 0.0; } else { sincos(a, &sn, &cn); }
    f[1 + lane] = cn;
    f[4 + lane] = sn;
  } else if (lane < 6) {
    const float a = (float)S->x_t[lane];
    float sn, cn;
    sincosf(a, &sn, &cn);
    cs_f[lane - 3] = cn;
    cs_f[lane] = sn;
  } else if (lane == 6) {
    f[0] = 1.0;
    f[7] = 0.0;
  }
  wave_lds_fence();
  const bool with_hang = (mode & 2) != 0;
  if (lane < 24 || with_hang) {
    double val = angle_entry_value(ent_a, f);
    if (lane == 24 + 20 && S->d1_sign < 0) val = -val;
    if (lane < 24) S->jang[lane] = (float)val; else S->hang[lane - 24] = (float)val;
  }
  if (with_hang && lane < 8) S->hang[40 + lane] = (float)angle_entry_value(ent_b, f);  // entries 64..71
  if (lane == 63) {
    // fp32 (Translation * Rx * Ry * Rz), as pose_to_T12
    float* T = S->T;
    compose_R12(cs_f[0], cs_f[1], cs_f[2], cs_f[3], cs_f[4], cs_f[5], T);
    T[3] = (float)S->x_t[0]; T[7] = (float)S->x_t[1]; T[11] = (float)S->x_t[2];
    T12_to_colmajor16(T, S->final_T);  // final_transformation_ is assigned before every MT pass
  }
  if (lane == 0) S->pad1 = 0;
  wave_lds_fence();
}

// K4: consume the sums of the pass that just finished and decide what happens next.
// Runs on one lane of EVERY workgroup (redundantly, same inputs, same result).  sums: [0]=score [1..6]=grad
// [7]=pairs [8..28]=H upper.
// One lane against LDS is a latency machine: a ds_read costs ~100 cycles before its value can be used, and the compiler
// keeps reads and writes through the two LDS pointers in program order.  So each function reads what it needs in batches
// of independent loads, works in registers, and stores at the end (first version: field-by-field LDS access, 28
// serialised read->wait->write round trips for the copy of the sums alone, 1.9 us for the shortest path).
// The line search (9 of 10 passes at cfg 2) and the 6x6 solve are functions of their own (called, not inlined: 50 kernel
// instantiations share them), each small enough to live in the caller-saved half of the register file: the AMDGPU
// calling convention makes a callee spill every callee-saved VGPR it touches to scratch, and one function holding the
// whole state machine plus the register-resident elimination needed 250 of them (3.3 us instead of 1.9).
enum CtlNext : int { CTL_DONE = 0, CTL_NEWTON_BEGIN = 1, CTL_NEWTON_END = 2 };

// Line-search half (computeStepLengthMT): bookkeeping of every pass + the More-Thuente decision after a
// line-search pass.  Returns what the Newton half has to do, CTL_DONE if the next request is already in the state.
__device__ __forceinline__ int ndt_controller_mt(LdsState* S, const LdsDouble* sums) {
  const double mu = 1.e-4, nu = 0.9;
  const int max_step_iterations = 10;
  double sv[8];  // score, gradient, #pairs; the Hessian sums stay in LDS until the Newton solve loads them
#pragma unroll
  for (int k = 0; k < 8; k++) sv[k] = sums[k];
  const int phase = S->phase;
  const bool had_hessian = S->want_hessian != 0;
  const int n_evals = S->n_evals + 1;
  if (phase == PH_INIT || phase == PH_MT_HESS || phase == PH_DIAG) {
    S->n_evals = n_evals;
    S->last_pairs = sv[7];
    if (phase != PH_MT_HESS) {  // the Hessian recomputation leaves score and gradient of the last trial in place
      S->score = sv[0];
#pragma unroll
      for (int i = 0; i < 6; i++) S->g[i] = sv[1 + i];
    }
    if (phase == PH_DIAG) {  // lsr_ndt_derivatives: the sums are the result
      if (had_hessian) {
        double hu[21];
#pragma unroll
        for (int k = 0; k < 21; k++) hu[k] = sums[8 + k];
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = i; j < 6; j++) {
            S->H[i * 6 + j] = hu[k];
            S->H[j * 6 + i] = hu[k];
            k++;
          }
      }
      S->done = 1;
      return CTL_DONE;
    }
    return phase == PH_INIT ? CTL_NEWTON_BEGIN : CTL_NEWTON_END;
  }
  // ---- PH_MT_FIRST / PH_MT_TRIAL: one batch of loads
  int open_interval = S->open_interval, interval_converged = S->interval_converged, step_iterations = S->step_iterations;
  const double step_max = S->step_max, step_min = S->step_min;
  double p[6], dir[6];
#pragma unroll
  for (int i = 0; i < 6; i++) { p[i] = S->p[i]; dir[i] = S->dir[i]; }
  const double phi_0 = S->phi_0, d_phi_0 = S->d_phi_0;
  double a_t = S->a_t;
  MtInterval I = {S->a_l, S->f_l, S->g_l, S->a_u, S->f_u, S->g_u};

  const double score = sv[0];
  const double phi_t = -score;
  double dot = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) dot += sv[1 + i] * dir[i];
  const double d_phi_t = -dot;
  const double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t;
  const double d_psi_t = d_phi_t - mu * d_phi_0;
  if (phase == PH_MT_TRIAL) {
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = 0;
      I.f_l = I.f_l + phi_0 - mu * d_phi_0 * I.a_l;
      I.g_l = I.g_l + mu * d_phi_0;
      I.f_u = I.f_u + phi_0 - mu * d_phi_0 * I.a_u;
      I.g_u = I.g_u + mu * d_phi_0;
    }
    if (open_interval)
      interval_converged = mt_update_interval(I, a_t, psi_t, d_psi_t) ? 1 : 0;
    else
      interval_converged = mt_update_interval(I, a_t, phi_t, d_phi_t) ? 1 : 0;
    step_iterations++;
  }
  int next = CTL_DONE;
  if (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    if (open_interval)
      a_t = mt_trial_value(I.a_l, I.f_l, I.g_l, I.a_u, I.f_u, I.g_u, a_t, psi_t, d_psi_t);
    else
      a_t = mt_trial_value(I.a_l, I.f_l, I.g_l, I.a_u, I.f_u, I.g_u, a_t, phi_t, d_phi_t);
    a_t = fmax(fmin(a_t, step_max), step_min);
#pragma unroll
    for (int i = 0; i < 6; i++) S->x_t[i] = p[i] + dir[i] * a_t;
    S->a_t = a_t;
    S->want_hessian = 0;
    S->phase = PH_MT_TRIAL;
    S->pad1 = 1;  // build T / jang for x_t
  } else if (step_iterations) {
    // computeHessian at x_t: current j_ang, h_ang left over from the last with-Hessian pass; same pose, no new request
    S->want_hessian = 1;
    S->phase = PH_MT_HESS;
  } else {
    next = CTL_NEWTON_END;
  }
  // ---- one batch of stores
  S->n_evals = n_evals;
  S->last_pairs = sv[7];
  S->score = score;
#pragma unroll
  for (int i = 0; i < 6; i++) S->g[i] = sv[1 + i];
  S->open_interval = open_interval;
  S->interval_converged = interval_converged;
  S->step_iterations = step_iterations;
  S->a_l = I.a_l; S->f_l = I.f_l; S->g_l = I.g_l; S->a_u = I.a_u; S->f_u = I.f_u; S->g_u = I.g_u;
  return next;
}

// Newton half (computeTransformation), in two small pieces around the 6x6 solve (solve6_wave) so that nothing but
// the two LDS pointers is live across a call (a callee-saved register costs its user a scratch spill).
// End of a Newton iteration: pose update, convergence test.  Returns CTL_NEWTON_BEGIN, or CTL_DONE after finishing.
__device__ __forceinline__ int ndt_newton_end(LdsState* S) {
  const int nr_iterations = S->nr_iterations, max_iter = S->max_iter;
  const double a_t = S->a_t, eps = S->eps;
  double p[6], dir[6];
#pragma unroll
  for (int i = 0; i < 6; i++) { p[i] = S->p[i]; dir[i] = S->dir[i]; }
  const double score = S->score;
  const int n_points = S->n_points;
  int converged = S->converged;
  if (nr_iterations > max_iter || (nr_iterations && (fabs(a_t) < eps))) converged = 1;
#pragma unroll
  for (int i = 0; i < 6; i++) S->p[i] = p[i] + dir[i] * a_t;
  S->nr_iterations = nr_iterations + 1;
  S->converged = converged;
  if (converged) {
    S->trans_probability = score / (double)n_points;
    S->done = 1;
    return CTL_DONE;
  }
  return CTL_NEWTON_BEGIN;
}

// Start of a Newton iteration, after solve6_wave left delta = -H^{-1} g in `delta`: direction, computeStepLengthMT prologue,
// first step of the line search.  Returns CTL_DONE (request written or align finished) or CTL_NEWTON_END (zero slope).
__device__ __forceinline__ int ndt_newton_begin(LdsState* S, const LdsDouble* delta_lds) {
  const double mu = 1.e-4;
  double delta[6], g[6], p[6];
#pragma unroll
  for (int i = 0; i < 6; i++) { delta[i] = delta_lds[i]; g[i] = S->g[i]; p[i] = S->p[i]; }
  const double score = S->score, step_max = S->step_max, step_min = S->step_min;
  const int n_points = S->n_points;
  double nrm = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) nrm += delta[i] * delta[i];
  nrm = sqrt(nrm);
  if (nrm == 0 || nrm != nrm) {
    S->converged = (nrm == nrm) ? 1 : 0;
    S->trans_probability = score / (double)n_points;
    S->done = 1;
    return CTL_DONE;
  }
  double dir[6];
#pragma unroll
  for (int i = 0; i < 6; i++) dir[i] = delta[i] / nrm;
  const double phi_0 = -score;
  double dot = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) dot += g[i] * dir[i];
  double d_phi_0 = -dot;
  S->phi_0 = phi_0;
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) {
      S->d_phi_0 = d_phi_0;
      S->a_t = 0;
#pragma unroll
      for (int i = 0; i < 6; i++) S->dir[i] = dir[i];
      return CTL_NEWTON_END;
    }
    d_phi_0 = -d_phi_0;
#pragma unroll
    for (int i = 0; i < 6; i++) dir[i] = -dir[i];
  }
  const double a_t = fmax(fmin(nrm, step_max), step_min);
  S->d_phi_0 = d_phi_0;
  S->a_l = 0; S->a_u = 0;
  S->f_l = 0; S->f_u = 0;  // psi(0) = phi_0 - phi_0 - mu*d_phi_0*0
  S->g_l = d_phi_0 - mu * d_phi_0;
  S->g_u = d_phi_0 - mu * d_phi_0;
  S->interval_converged = (step_max - step_min) < 0 ? 1 : 0;
  S->open_interval = 1;
  S->step_iterations = 0;
  S->a_t = a_t;
#pragma unroll
  for (int i = 0; i < 6; i++) { S->dir[i] = dir[i]; S->x_t[i] = p[i] + dir[i] * a_t; }
  S->want_hessian = 1;
  S->phase = PH_MT_FIRST;
  S->pad1 = 3;  // build T / jang / hang for x_t
  return CTL_DONE;
}

// The controller step, executed by the 64 lanes of wave 0 (threadIdx.x < 64): the scalar pieces run on lane 0, the 6x6 solve
// on the whole wave; `next` travels through an SGPR, the wave's own LDS traffic is ordered by wave_lds_fence().
__device__ __forceinline__ void ndt_controller_wave0(LdsState* S, const LdsDouble* sums) {
  const bool lead = (threadIdx.x == 0);
  int next = 0;
  if (lead) next = ndt_controller_mt(S, sums);
  next = __builtin_amdgcn_readfirstlane(next);
  LdsDouble* scratch = const_cast<LdsDouble*>(sums);  // sums[0..7] are consumed by now: delta lands in sums[0..5]
  for (int guard = 0; guard < 8 && next != CTL_DONE; guard++) {
    if (next == CTL_NEWTON_END) {
      if (lead) next = ndt_newton_end(S);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      solve6_wave(sums + 8, S->g, scratch);  // the Hessian in use is always the one of the pass that just finished
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lead) next = ndt_newton_begin(S, scratch);
    }
    next = __builtin_amdgcn_readfirstlane(next);
  }
  if (lead && next != CTL_DONE) {  // unreachable in practice (the a_t == 0 path converges after two rounds)
    S->trans_probability = S->score / (double)S->n_points;
    S->done = 1;
  }
}

#ifdef LSR_TIMING
__device__ long long* g_lsr_timing = nullptr;  // [blocks][16] {wall, shader} pairs
#define LSR_STAMP(k)                                                                   \
  if (threadIdx.x == 0 && g_lsr_timing && blockIdx.y == 0) {                            \
    g_lsr_timing[blockIdx.x * 32 + 2 * (k)] = (long long)wall_clock64();                \
    g_lsr_timing[blockIdx.x * 32 + 2 * (k) + 1] = (long long)clock64();                 \
  }
#define LSR_STAMP_T(k, t)                                                              \
  if ((int)threadIdx.x == (t) && g_lsr_timing && blockIdx.y == 0) {                       \
    g_lsr_timing[blockIdx.x * 32 + 2 * (k)] = (long long)wall_clock64();                \
  }
#ifdef LSR_TIMING_SPAN  // contended atomics in front of the head's loads: distorts the head by microseconds, off by default
#define LSR_SPAN_BEGIN(seq)                                                                                   \
  if (threadIdx.x == 0 && g_lsr_timing && blockIdx.y == 0)                                                     \
    atomicMin((unsigned long long*)&g_lsr_timing[(512 + ((seq) & 255)) * 32 + 0], (unsigned long long)wall_clock64());
#define LSR_SPAN_END(seq)                                                                                     \
  if (threadIdx.x == 0 && g_lsr_timing && blockIdx.y == 0)                                                     \
    atomicMax((unsigned long long*)&g_lsr_timing[(512 + ((seq) & 255)) * 32 + 1], (unsigned long long)wall_clock64());
#else
#define LSR_SPAN_BEGIN(seq)
#define LSR_SPAN_END(seq)
#endif
// controller time by the phase it consumed: rows 800 + phase hold {sum of ticks, count, sum of request-build ticks}
#define LSR_CTL_BEGIN(L)                                                              \
  long long _ctl_t0 = 0; int _ctl_ph = 0;                                              \
  if (threadIdx.x == 0 && g_lsr_timing && blockIdx.x == 0 && blockIdx.y == 0) { _ctl_t0 = (long long)wall_clock64(); _ctl_ph = (L)->phase; }
#define LSR_CTL_END(col)                                                              \
  if (threadIdx.x == 0 && g_lsr_timing && blockIdx.x == 0 && blockIdx.y == 0) {          \
    const long long _t = (long long)wall_clock64();                                     \
    atomicAdd((unsigned long long*)&g_lsr_timing[(800 + _ctl_ph) * 32 + (col)], (unsigned long long)(_t - _ctl_t0)); \
    if ((col) == 0) atomicAdd((unsigned long long*)&g_lsr_timing[(800 + _ctl_ph) * 32 + 1], 1ull);                     \
    _ctl_t0 = _t;                                                                       \
  }
// per-pass phase durations of workgroup 0 by pass type: rows 810 (gradient-only pass) / 811 (with Hessian) hold
// {count, head ticks (entry -> state in LDS), main ticks (request read -> points done), tail ticks, shader cycles entry -> exit,
//  wall ticks entry -> exit}
#define LSR_PASS_BEGIN()                                                               \
  long long _p_t0 = 0, _p_c0 = 0, _p_t1 = 0, _p_t7 = 0, _p_t2 = 0;                         \
  const bool _p_on = (threadIdx.x == 0 && g_lsr_timing && blockIdx.x == 0 && blockIdx.y == 0); \
  if (_p_on) { _p_t0 = (long long)wall_clock64(); _p_c0 = (long long)clock64(); }
#define LSR_PASS_MARK(var) if (_p_on) { var = (long long)wall_clock64(); }
#define LSR_PASS_END(hess)                                                             \
  if (_p_on) {                                                                          \
    const long long _t3 = (long long)wall_clock64(), _c3 = (long long)clock64();         \
    unsigned long long* _r = (unsigned long long*)&g_lsr_timing[(810 + ((hess) ? 1 : 0)) * 32]; \
    atomicAdd(&_r[0], 1ull); atomicAdd(&_r[1], (unsigned long long)(_p_t1 - _p_t0)); atomicAdd(&_r[2], (unsigned long long)(_p_t2 - _p_t7)); \
    atomicAdd(&_r[3], (unsigned long long)(_t3 - _p_t2)); atomicAdd(&_r[4], (unsigned long long)(_c3 - _p_c0)); atomicAdd(&_r[5], (unsigned long long)(_t3 - _p_t0)); \
  }
#else
#define LSR_STAMP(k)
#define LSR_STAMP_T(k, t)
#define LSR_SPAN_BEGIN(seq)
#define LSR_SPAN_END(seq)
#define LSR_CTL_BEGIN(L)
#define LSR_CTL_END(col)
#define LSR_PASS_BEGIN()
#define LSR_PASS_MARK(var)
#define LSR_PASS_END(hess)
#endif

// Values read from the LDS state image are wave-uniform; telling the compiler (v_readfirstlane -> SGPR)
// turns the branches on them into scalar branches and keeps them out of the vector register file.
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uniform_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// sum partner inside a quad of lanes via DPP quad_perm (no LDS traffic): CTRL 0xB1 = [1,0,3,2], 0x4E = [2,3,0,1]
template <int CTRL>
__device__ __forceinline__ double dpp_quad_xor(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <int NOFF>
struct Offsets;
template <>
struct Offsets<1> {
  static __device__ __forceinline__ void get(int o, int& dx, int& dy, int& dz) { dx = dy = dz = 0; }
};
template <>
struct Offsets<7> {
  static __device__ __forceinline__ void get(int o, int& dx, int& dy, int& dz) {
    dx = (o == 1) - (o == 2);
    dy = (o == 3) - (o == 4);
    dz = (o == 5) - (o == 6);
  }
};
template <>
struct Offsets<27> {
  static __device__ __forceinline__ void get(int o, int& dx, int& dy, int& dz) {
    dx = o / 9 - 1;
    dy = (o / 3) % 3 - 1;
    dz = o % 3 - 1;
  }
};

// ===========================================================================================
// The canonical sum of a pass: one input, one answer
// ===========================================================================================
// Every derivative kernel — four lanes per point or one, any workgroup size, any number of workgroups, a registration alone
// or inside a candidate set — returns the SAME 29 doubles for the same request, bit for bit, because the sum is defined on
// the input and not on the launch:
//  * a point's 29 fp32 terms: its neighbours are dealt to four partial sums (partial q takes neighbours q, q + 4, q + 8, ... in
//    that order), the partials are added as (p0 + p1) + (p2 + p3) (ndt_point.hpp forbids fp contraction outside its fmaf calls);
//  * a CHUNK = 64 consecutive source points [64 c, 64 c + 63] (absent points count as zeros): its fp64 total is formed by a
//    fixed tree — gradient-only passes (8 values): eight runs of 8 consecutive points, each summed left to right in fp64, then a
//    butterfly over the runs (xor 1, 2, 4); passes with Hessian (29 values): four runs of 16, then a butterfly (xor 1, 2);
//  * chunk totals are added EXACTLY: each is split into NDT_NBINS signed 31-bit pieces against the fixed binary quanta
//    q_k = 2^(62 - 31 (k + 1)) (ndt.hpp) and the pieces are summed as integers — associative, so neither the order in which
//    chunks arrive nor the workgroup that owns a chunk can change a bit.
// What this buys: lsr_align of one registration (quad kernel) and the same registration inside lsr_align_batch (lane kernel)
// walk the same Newton / More-Thuente trajectory to the same final_T; tests assert array_equal, not a tolerance.
namespace canon {
constexpr int TILE_PITCH = 68;    // floats per row of a staging tile: rows 16-byte aligned, consecutive rows four banks apart
constexpr int TILE_ROWS = 16;     // values staged at a time (a Hessian pass goes through the tile twice)
constexpr int TILE_FLOATS = TILE_ROWS * TILE_PITCH;

__device__ __forceinline__ double quantum(const int k) { return __hiloint2double((1023 + 62 - 31 * (k + 1)) << 20, 0); }
__device__ __forceinline__ double iquantum(const int k) { return __hiloint2double((1023 - 62 + 31 * (k + 1)) << 20, 0); }

// piece k of the exact split of t: the pieces of one value can be formed independently of each other (five lanes, one piece
// each) because t minus its multiple-of-q_(k-1) part is exact in fp64.  |t| >= 2^62 or NaN: no piece, *poison raised.
__device__ __forceinline__ int piece(const double t, const int k, bool* poison) {
  *poison = !(fabs(t) < 4611686018427387904.0);
  if (*poison) return 0;
  double r = t;
  if (k > 0) r = t - trunc(t * iquantum(k - 1)) * quantum(k - 1);   // |r| < q_(k-1) = 2^31 q_k, exact
  return (int)(r * iquantum(k));                                     // truncation toward zero
}

// 8 consecutive fp32 terms at p (16-byte aligned, LDS), lanes l & 7 = run: the chunk total of a gradient-only pass, in every lane
__device__ __forceinline__ double reduce_grad(const float* p) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4);
  double t = (double)a.x;
  t += (double)a.y; t += (double)a.z; t += (double)a.w;
  t += (double)b.x; t += (double)b.y; t += (double)b.z; t += (double)b.w;
  t += dpp_quad_xor<0xB1>(t);    // runs {0<->1, 2<->3, ...}
  t += dpp_quad_xor<0x4E>(t);    // pairs of runs
  t += dpp_quad_xor<0x141>(t);   // row_half_mirror: lane 7 - l of the 8-lane group = the other quad (whose lanes all hold its sum)
  return t;
}
// 16 consecutive fp32 terms at p, lanes l & 3 = run: the chunk total of a pass with Hessian, in every lane of the quad
__device__ __forceinline__ double reduce_hess(const float* p) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4), c = *reinterpret_cast<const f4*>(p + 8),
           d = *reinterpret_cast<const f4*>(p + 12);
  double t = (double)a.x;
  t += (double)a.y; t += (double)a.z; t += (double)a.w;
  t += (double)b.x; t += (double)b.y; t += (double)b.z; t += (double)b.w;
  t += (double)c.x; t += (double)c.y; t += (double)c.z; t += (double)c.w;
  t += (double)d.x; t += (double)d.y; t += (double)d.z; t += (double)d.w;
  t += dpp_quad_xor<0xB1>(t);
  t += dpp_quad_xor<0x4E>(t);
  return t;
}
// piece k of chunk total t of value v -> a workgroup's LDS bins [NDT_NBINS][32] (poison: slot 31 of bin 0, raised by the k = 0 caller)
__device__ __forceinline__ void add_piece_lds(unsigned long long* ibin, const int v, const int k, const double t) {
  bool poison;
  const int m = piece(t, k, &poison);
  if (m != 0) atomicAdd(&ibin[k * 32 + v], (unsigned long long)(long long)m);
  if (poison && k == 0) atomicAdd(&ibin[31], 1ull);
}
}  // namespace canon

// The derivative pass (K3) is preceded, in the same launch, by the controller step (K4) that consumes the PREVIOUS pass.
//
// "Pull" structure: launch number `seq` starts — in EVERY workgroup, redundantly and deterministically — by folding the
// accumulator bank launch seq-1 left behind, advancing the Newton / More-Thuente controller on an LDS image of the state and
// building the next evaluation request; only then does it evaluate its own points and add their chunk sums to its own bank.
// No workgroup ever waits for another one inside a launch: the kernel boundary is the only synchronisation.  State is double
// buffered by launch parity (launch seq reads state[seq & 1]; workgroup 0 writes state[(seq + 1) & 1]), the banks rotate
// (launch seq adds to bank seq % 3, reads bank (seq - 1) % 3 and clears bank (seq + 1) % 3), so a late-starting workgroup
// can never observe a value produced by its own launch.
//  BYVAL: a single-registration launch carries its NdtProblem in the kernel arguments, which removes one dependent memory
//         round trip from the latency chain of every pass.
//  TAB:   where the leaf records live (NdtTableMode).  NDT_TAB_LDS stages the whole valid-voxel table (uint16 cell->slot map +
//         48-byte records) into LDS with wave-wide 16-byte global->LDS DMA issued right after the state has landed: the copy
//         flies while wave 0 runs the controller, and the dependent gathers of a point become ds_read_b128.

// ===========================================================================================
// K3 + K4, "quad" variant for single registrations: FOUR lanes per source point
// ===========================================================================================
// A 30k-point scan is 469 waves of one-lane-per-point work on a chip with 1024 SIMDs, and the time of a pass is the
// dependent-instruction latency of ONE wave (~1000 instructions).  Here a quad of lanes shares a point: lane l takes
// neighbours l and l + 4 of the DIRECT7 neighbourhood (l, l+4, l+8, ... for DIRECT26), the per-point sums A = sum w C q and
// E = sum w (C - d2 Cq Cq^T) are combined inside the quad with DPP adds, and the (angle dependent) Jacobian / Hessian terms
// of the point are then formed by all four lanes (identical values; lane 0 of the quad hands them to the reduction, so the
// quad-sum step of the one-lane kernel disappears).  512-thread workgroups of 128 points put two waves on every SIMD of
// every CU.
// With 235 workgroups the partial-ROW scheme would make every head read twice as many rows; this kernel accumulates the
// workgroup partials into int64 bins instead (NDT_NBINS chunks of 31 bits against fixed quanta, integer atomics, one row
// per shard): exact, order independent => bit-reproducible, and the head reads 10 KiB whatever the number of workgroups.
// Launch seq adds to bank seq % 3, reads bank (seq-1) % 3 and clears bank (seq+1) % 3.
__device__ __forceinline__ float dpp_quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  return v;
}

// min / max over the lanes of a wave (every lane receives the result)
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int m = 4; m <= 32; m <<= 1) v = min(v, __shfl_xor(v, m, 64));   // the four lanes of a quad hold the same value
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int m = 4; m <= 32; m <<= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}
// n / d for small n through a precomputed magic number m = 2^32 / d + 1 (exact for n, d < 65536; d = 1 has no such m)
__device__ __forceinline__ unsigned int div_magic(unsigned int d) { return 0xFFFFFFFFu / d + 1u; }
__device__ __forceinline__ unsigned int div_small(unsigned int n, unsigned int d, unsigned int m) { return d == 1u ? n : __umulhi(n, m); }

// TAB (NdtTableMode) of the quad kernel:
//  NDT_TAB_LDS   the whole valid-voxel table staged into LDS at the head of the launch (tables that fit: res 5.0);
//  NDT_TAB_TILE  per workgroup and per pass, the BOX of grid cells its points touch (bounding box of their centre cells + the
//                one-cell halo of the neighbourhood) is gathered from the dense global table into LDS with global->LDS DMA
//                (48 of the 64 bytes of every record, one 16-byte piece per lane), and the 7 x 3 dependent gathers of a point
//                become ds_read_b128.  The source is ordered by voxel tile at the start of the align (ndt_sort_source), so a
//                workgroup's 128 points are neighbours in space — and stay neighbours under any rigid motion the line search
//                applies — and their box is a few dozen cells.  A box beyond the tile buffer (scattered points) makes that
//                workgroup read the global table directly for that pass, as NDT_TAB_DENSE does for all.
//  NDT_TAB_DENSE / NDT_TAB_COMPACT  records gathered from global memory.
// BYVAL: the problem travels in the kernel arguments (single registrations); otherwise probs[blockIdx.y] (batches).
template <int NOFF, int TAB, int PTS, bool BYVAL, bool KD = false>
__global__ __launch_bounds__(4 * PTS) void ndt_eval_quad_kernel(const NdtProblem pv, const NdtProblem* __restrict__ probs, const int seq) {
  constexpr int THREADS = 4 * PTS;
  // floats per row of the per-point buffers, chosen against the 32-lane groups of ds_*_b32: phase A writes rows
  // ql + 4m at columns pq (4 rows x 8 columns per group): pitch = 8 (mod 32) spreads them over all 32 banks; phase C reads
  // rows cv at columns cseg + SEGS k (2 rows x 16 columns per group): pitch = 16 (mod 32)
  constexpr int PITCH_PT = PTS + 8, PITCH_O = PTS + 4;   // s_o rows 16-byte aligned, four banks apart (phase C reads runs with ds_read_b128)
  constexpr int NT = (NOFF + 3) / 4;       // neighbours per lane
  static_assert(PTS % 64 == 0, "a workgroup batch is a whole number of canonical chunks");
  const NdtProblem& P = BYVAL ? pv : probs[blockIdx.y];
  if ((int)blockIdx.x >= P.nblocks) return;
  const int tid = threadIdx.x, ql = tid & 3, pq = tid >> 2;
  LSR_STAMP(0)
  LSR_STAMP_T(9, THREADS - 64)
  LSR_PASS_BEGIN()
  LSR_SPAN_BEGIN(seq)

  __shared__ float s_pt[14][PITCH_PT];   // phase A -> B: per point {score, #pairs, A (3), E (6), x, y, z}
  __shared__ __attribute__((aligned(16))) float s_o[29][PITCH_O];   // phase B -> C: the 29 per-point terms (phase C reads runs as 16-byte vectors)
  __shared__ double s_bin[NDT_NBINS][32];
  __shared__ unsigned long long s_ibin[NDT_NBINS * 32];   // workgroups that walk several batches collect their pieces here first
  __shared__ double s_sum[NDT_NRED];
  __shared__ double s_lu[8][2];
  __shared__ int s_box[8];               // NDT_TAB_TILE: min (0..2) / max (3..5) centre cell of this workgroup's points
  constexpr int STATE_Q = (int)(sizeof(NdtState) / 16);
  __shared__ uint4 s_state_q[STATE_Q];
  unsigned int* s_state = reinterpret_cast<unsigned int*>(s_state_q);
  extern __shared__ uint4 s_table[];

  const NdtState* __restrict__ Sin = P.st + (seq & 1);
  NdtState* __restrict__ Sout = P.st + ((seq + 1) & 1);

  // ---- head: everything this workgroup needs from memory in one round trip
  const int stride = P.nblocks * PTS;
  int i = blockIdx.x * PTS + pq;
  float x = 0.f, y = 0.f, z = 0.f;
  unsigned int ang_entry = 0u, ang_entry_b = 0u;
  {
    const uint4* gq = reinterpret_cast<const uint4*>(Sin);
    const uint4 stq = (tid < STATE_Q) ? gq[tid] : make_uint4(0u, 0u, 0u, 0u);
    // bins of the previous launch: thread (k, v) folds the NDT_NSHARDS shards of bin k of value v (exact int64 adds)
    long long msum = 0;
    if (seq > 0 && tid < NDT_NBINS * 32) {
      const long long* b = P.bins + (size_t)((seq + 2) % NDT_NBANKS) * NDT_BANK_WORDS + tid;
      long long m[NDT_NSHARDS];
#pragma unroll
      for (int sh = 0; sh < NDT_NSHARDS; sh++) m[sh] = b[sh * (NDT_NBINS * 32)];
      msum = ((m[0] + m[1]) + (m[2] + m[3])) + ((m[4] + m[5]) + (m[6] + m[7]));
    }
    if (tid < NDT_NBINS * 32) {
      const int k = tid >> 5;
      // quantum of bin k: 2^(62 - 31 (k + 1)); an int64 below 2^53 converts exactly
      const double q = __hiloint2double((1023 + 62 - 31 * (k + 1)) << 20, 0);
      s_bin[k][tid & 31] = (double)msum * q;
      s_ibin[tid] = 0ull;
    }
    if (tid < STATE_Q) s_state_q[tid] = stq;
    if (TAB == NDT_TAB_TILE && tid >= THREADS - 6) s_box[THREADS - 1 - tid] = (THREADS - 1 - tid < 3) ? INT_MAX : INT_MIN;
    // issued after the shared lines have landed, on purpose (see the one-lane kernel): these fly across the head's barriers
    if (i < P.n) { x = P.sx[i]; y = P.sy[i]; z = P.sz[i]; }
    if (tid < 64) ang_entry = k_angle_entries[tid];
    if (tid < 8) ang_entry_b = k_angle_entries[64 + tid];
  }
  LSR_STAMP_T(8, 0)
  LSR_STAMP_T(10, THREADS - 64)
  barrier_lds_only();  // not __syncthreads(): the point loads stay in flight
  LdsState* L = (LdsState*)s_state;
  LSR_STAMP(1)
  LSR_PASS_MARK(_p_t1)
  if (P.mailbox != nullptr && blockIdx.x == 0 && tid == 0)  // progress report for the host's launch feeder (relaxed)
    __hip_atomic_store(&P.mailbox->progress, ((unsigned long long)(unsigned int)L->token << 32) | (unsigned int)seq,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (uniform_i(L->done)) {  // finished earlier: keep both state buffers identical so later launches see it too
    if (blockIdx.x == 0 && seq > 0) {
      uint4* gq = reinterpret_cast<uint4*>(Sout);
      if (tid < STATE_Q) gq[tid] = s_state_q[tid];
    }
    return;
  }
  if (TAB == NDT_TAB_LDS) {
    // voxel table -> LDS by DMA, issued by waves 1..7 (wave 0 calls the controller: device functions start with
    // s_waitcnt vmcnt(0)); lands in the shadow of the controller
    const int nchunks = P.lds_bytes >> 10;
    const unsigned char* img = reinterpret_cast<const unsigned char*>(P.lds_image) + (size_t)(tid & 63) * 16;
    unsigned char* dst = reinterpret_cast<unsigned char*>(s_table);
    if (tid >= 64)
      for (int c = (tid >> 6) - 1; c < nchunks; c += THREADS / 64 - 1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (size_t)c * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + (size_t)c * 1024), 16, 0, 0);
  }
  // ---- the controller step lives on WAVE 0 alone: totals, Newton / More-Thuente decision, next request, state write-back —
  // ordered by the wave's own lockstep, no workgroup barrier between them (each s_barrier with 8 waves cost ~0.2-0.3 us and
  // there were four).  The other waves have issued the table DMA and wait at the one barrier below.
  if (tid < 64) {
    if (seq > 0) {
      if (tid < NDT_NRED) {
        // fold the bins smallest quantum first (fixed order); a raised poison slot (overflow / NaN partial) poisons all sums
        double t = (((s_bin[4][tid] + s_bin[3][tid]) + s_bin[2][tid]) + s_bin[1][tid]) + s_bin[0][tid];
        if (s_bin[0][31] != 0.0 || s_bin[1][31] != 0.0) t = __longlong_as_double(0x7FF8000000000000ll);
        s_sum[tid] = t;
      }
      wave_lds_fence();
      LSR_STAMP(6)
      LSR_CTL_BEGIN(L)
      ndt_controller_wave0(L, (const LdsDouble*)s_sum);
      wave_lds_fence();
      LSR_CTL_END(0)
      LSR_STAMP(5)
      build_request_wave0(reinterpret_cast<NdtState*>(s_state), &s_lu[0][0], reinterpret_cast<float*>(&s_lu[4][0]), ang_entry, ang_entry_b);
      LSR_CTL_END(2)
      LSR_STAMP(4)
    }
    if (blockIdx.x == 0) {
      uint4* gq = reinterpret_cast<uint4*>(Sout);
      gq[tid] = s_state_q[tid];
      if (tid + 64 < STATE_Q) gq[tid + 64] = s_state_q[tid + 64];
    }
  }
  __syncthreads();  // the request is complete and the table DMA has landed (vmcnt(0) + barrier)
  if (uniform_i(L->done)) {
    // the controller has just finished this align(): publish the result into the host mailbox, flag last
    if (P.mailbox != nullptr && blockIdx.x == 0 && tid == 0) {
      NdtMailbox* mb = P.mailbox;
#pragma unroll
      for (int k = 0; k < 16; k++) mb->final_T[k] = L->final_T[k];
      mb->converged = L->converged;
      mb->nr_iterations = L->nr_iterations;
      mb->n_evals = L->n_evals;
      mb->trans_probability = L->trans_probability;
      mb->last_pairs = L->last_pairs;
      __threadfence_system();
      __hip_atomic_store(&mb->done, (unsigned int)L->token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (blockIdx.x == 0) {  // clear the bank launch seq + 1 will add to (read last by launch seq - 1, which is complete)
    uint4* zb = reinterpret_cast<uint4*>(P.bins + (size_t)((seq + 1) % NDT_NBANKS) * NDT_BANK_WORDS);
    for (int k = tid; k < NDT_BANK_WORDS / 2; k += THREADS) zb[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  LSR_STAMP(7)
  LSR_PASS_MARK(_p_t7)

  // ---- this launch's request, straight from the LDS image
  const bool hess = uniform_i(L->want_hessian) != 0;
  const double d1d = uniform_d(L->d1);
  const float d2 = uniform_f((float)L->d2);
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = uniform_f(L->T[k]);
  const float leaf = P.leaf;
  LSR_STAMP(11)

  const unsigned short* s_map = reinterpret_cast<const unsigned short*>(s_table);
  const float4* s_rec = reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(s_table) + P.lds_map_bytes);
  const float4* s_tile = reinterpret_cast<const float4*>(s_table);

  // Per batch of PTS points:
  //  A (all 4 PTS lanes, four per point): transform, neighbourhood, pair terms, quad combine;
  //  B the 29 Jacobian / Hessian terms of the point.  With Hessian: the 14 per-point sums go through LDS to ONE lane per point
  //    (two full waves instead of eight quarter-full ones: ~150 instructions per wave); gradient-only passes (three in four)
  //    form their 8 terms right where the sums are, in every lane of the quad — ~30 instructions, no barrier, no LDS round trip;
  //  C: the canonical chunk sums (canon:: above) of the batch's PTS / 64 chunks, straight into the accumulator bank.
  long long* const bank = P.bins + (size_t)(seq % NDT_NBANKS) * NDT_BANK_WORDS + (size_t)(blockIdx.x & (NDT_NSHARDS - 1)) * (NDT_NBINS * 32);
  // a workgroup with ONE batch (a 30k-point scan: all of them) sends its pieces straight to the bank; one that walks several
  // collects them in LDS and sends the sums once (atomics on one address serialise in the L2: cfg 5 would queue twice as many)
  const bool one_batch = (long long)blockIdx.x * PTS + stride >= (long long)P.n;
  for (int base = blockIdx.x * PTS; base < P.n; base += stride) {  // uniform across the workgroup
    // ---- phase A
    {
      const float tx = xform_ref(T[0], T[1], T[2], T[3], x, y, z);
      const float ty = xform_ref(T[4], T[5], T[6], T[7], x, y, z);
      const float tz = xform_ref(T[8], T[9], T[10], T[11], x, y, z);
      const float fx = floorf(tx / leaf), fy = floorf(ty / leaf), fz = floorf(tz / leaf);
      const bool finite_ok = (i < P.n) && (fabsf(fx) < 1.0e9f) && (fabsf(fy) < 1.0e9f) && (fabsf(fz) < 1.0e9f);
      const int ci = finite_ok ? (int)fx : INT_MIN / 2, cj = finite_ok ? (int)fy : INT_MIN / 2, ck = finite_ok ? (int)fz : INT_MIN / 2;

      // ---- NDT_TAB_TILE: the box of cells this batch touches -> LDS
      bool use_tile = false;
      int lo0 = 0, lo1 = 0, lo2 = 0, tdx = 1, tdxy = 1;
      if (TAB == NDT_TAB_TILE) {
        // centre cells that can have a neighbour inside the grid: within one cell of it
        const bool near = (ci >= P.min_b[0] - 1) & (ci <= P.max_b[0] + 1) & (cj >= P.min_b[1] - 1) & (cj <= P.max_b[1] + 1) &
                          (ck >= P.min_b[2] - 1) & (ck <= P.max_b[2] + 1);
        const int mn0 = wave_min_i(near ? ci : INT_MAX), mn1 = wave_min_i(near ? cj : INT_MAX), mn2 = wave_min_i(near ? ck : INT_MAX);
        const int mx0 = wave_max_i(near ? ci : INT_MIN), mx1 = wave_max_i(near ? cj : INT_MIN), mx2 = wave_max_i(near ? ck : INT_MIN);
        if ((tid & 63) == 0 && mn0 != INT_MAX) {
          atomicMin(&s_box[0], mn0); atomicMin(&s_box[1], mn1); atomicMin(&s_box[2], mn2);
          atomicMax(&s_box[3], mx0); atomicMax(&s_box[4], mx1); atomicMax(&s_box[5], mx2);
        }
        barrier_lds_only();
        const int b0 = uniform_i(s_box[0]), b1 = uniform_i(s_box[1]), b2 = uniform_i(s_box[2]);
        const int b3 = uniform_i(s_box[3]), b4 = uniform_i(s_box[4]), b5 = uniform_i(s_box[5]);
        if (b0 == INT_MAX) {
          use_tile = true;   // no point of this batch is near the grid: every neighbour fails the bounds test, nothing is read
        } else {
          lo0 = max(b0 - 1, P.min_b[0]); lo1 = max(b1 - 1, P.min_b[1]); lo2 = max(b2 - 1, P.min_b[2]);
          const int hi0 = min(b3 + 1, P.max_b[0]), hi1 = min(b4 + 1, P.max_b[1]), hi2 = min(b5 + 1, P.max_b[2]);
          const int tdy = hi1 - lo1 + 1, tdz = hi2 - lo2 + 1;
          tdx = hi0 - lo0 + 1;
          tdxy = tdx * tdy;
          const long long ncell = (long long)tdxy * tdz;
          use_tile = ncell * NDT_LDS_REC_BYTES <= (long long)P.tile_bytes;
          if (use_tile) {
            // one 16-byte piece per lane and DMA instruction: piece q = 3 * (cell of the box) + part, LDS address 16 q
            const unsigned int npieces = 3u * (unsigned int)ncell;
            const unsigned int mxy = div_magic((unsigned int)tdxy), mx = div_magic((unsigned int)tdx);
            const unsigned char* recb = reinterpret_cast<const unsigned char*>(P.rec);
            unsigned char* dst = reinterpret_cast<unsigned char*>(s_table);
            for (unsigned int q0 = (unsigned int)(tid & ~63); q0 < npieces; q0 += THREADS) {   // q0: wave-uniform
              const unsigned int q = q0 + (unsigned int)(tid & 63);
              if (q < npieces) {
                const unsigned int cell = (q * 21846u) >> 16;          // q / 3 for q < 32768
                const unsigned int part = q - 3u * cell;
                const unsigned int c = div_small(cell, (unsigned int)tdxy, mxy);
                const unsigned int rem = cell - c * (unsigned int)tdxy;
                const unsigned int b = div_small(rem, (unsigned int)tdx, mx);
                const unsigned int a = rem - b * (unsigned int)tdx;
                const size_t gcell = (size_t)(lo0 - P.min_b[0] + (int)a) + (size_t)(lo1 - P.min_b[1] + (int)b) * P.mul1 +
                                     (size_t)(lo2 - P.min_b[2] + (int)c) * P.mul2;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(recb + gcell * 64 + part * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + (size_t)q0 * 16), 16, 0, 0);
              }
            }
          }
        }
        __syncthreads();   // the tile has landed (vmcnt(0) + barrier); s_box may be reset for the next batch
        if (tid >= THREADS - 6) s_box[THREADS - 1 - tid] = (THREADS - 1 - tid < 3) ? INT_MAX : INT_MIN;
      }

      bool valid[NT];
      int cellv[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const int o = ql + 4 * t;  // this lane's t-th neighbour
        int dx, dy, dz;
        Offsets<NOFF>::get(o, dx, dy, dz);
        const int a = ci + dx, b = cj + dy, c = ck + dz;
        const bool in = (o < NOFF) & (a >= P.min_b[0]) & (a <= P.max_b[0]) & (b >= P.min_b[1]) & (b <= P.max_b[1]) &
                        (c >= P.min_b[2]) & (c <= P.max_b[2]);
        valid[t] = in;
        if (TAB == NDT_TAB_TILE && use_tile) cellv[t] = in ? ((a - lo0) + (b - lo1) * tdx + (c - lo2) * tdxy) : 0;
        else cellv[t] = in ? ((a - P.min_b[0]) + (b - P.min_b[1]) * P.mul1 + (c - P.min_b[2]) * P.mul2) : 0;
      }
      if (KD) {   // KDTREE (a template form of its own: the DIRECT26 kernels keep their registers): of the 27 cells, the leaves whose centroid
                  // the kd-tree's radius search would return
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const int sl = valid[t] ? P.cell_slot[cellv[t]] : -1;
          const float4 cen = P.centroid[sl >= 0 ? sl : 0];
          valid[t] = (sl >= 0) & centroid_in_radius(tx, ty, tz, cen.x, cen.y, cen.z, P.radius2);
        }
      }
      float4 r0[NT], r1[NT], r2[NT];
      if (TAB == NDT_TAB_LDS) {
        int slot[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const int sl = (int)s_map[cellv[t]];
          valid[t] = valid[t] & (sl != 0xFFFF);
          slot[t] = valid[t] ? sl : 0;
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
          r0[t] = s_rec[slot[t] * 3 + 0];
          r1[t] = s_rec[slot[t] * 3 + 1];
          r2[t] = s_rec[slot[t] * 3 + 2];
        }
      } else if (TAB == NDT_TAB_TILE && use_tile) {
#pragma unroll
        for (int t = 0; t < NT; t++) {   // unusable cells of the box hold NaN records: the pair drops itself
          r0[t] = s_tile[cellv[t] * 3 + 0];
          r1[t] = s_tile[cellv[t] * 3 + 1];
          r2[t] = s_tile[cellv[t] * 3 + 2];
        }
      } else {
        size_t ridx[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
          if (TAB != NDT_TAB_COMPACT) {
            ridx[t] = (size_t)cellv[t];
          } else {
            const int sl = P.cell_slot[cellv[t]];
            valid[t] = valid[t] & (sl >= 0);
            ridx[t] = (size_t)(sl >= 0 ? sl : 0);
          }
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
          r0[t] = P.rec[ridx[t] * 4 + 0];
          r1[t] = P.rec[ridx[t] * 4 + 1];
          r2[t] = P.rec[ridx[t] * 4 + 2];
        }
      }

      float score = 0.f, npairs = 0.f;
      float A0 = 0.f, A1 = 0.f, A2 = 0.f;
      float E00 = 0.f, E01 = 0.f, E02 = 0.f, E11 = 0.f, E12 = 0.f, E22 = 0.f;
#pragma unroll
      for (int t = 0; t < NT; t++)
        pair_terms(valid[t], hess, tx, ty, tz, r0[t], r1[t], r2[t], d2, d1d, score, npairs, A0, A1, A2, E00, E01, E02, E11, E12, E22);
      // the quad's four partial sums -> every lane of the quad (fp32, as the reference sums a point's voxels in float)
      score = dpp_quad_sum(score); npairs = dpp_quad_sum(npairs);
      A0 = dpp_quad_sum(A0); A1 = dpp_quad_sum(A1); A2 = dpp_quad_sum(A2);
      if (hess) {
        E00 = dpp_quad_sum(E00); E01 = dpp_quad_sum(E01); E02 = dpp_quad_sum(E02);
        E11 = dpp_quad_sum(E11); E12 = dpp_quad_sum(E12); E22 = dpp_quad_sum(E22);
        // lane l of the quad stores rows l, l + 4, l + 8, (l + 12) of the point's record: one to four stores per lane
        const float row0 = (ql == 0) ? score : (ql == 1) ? npairs : (ql == 2) ? A0 : A1;
        const float row1 = (ql == 0) ? A2 : (ql == 1) ? E00 : (ql == 2) ? E01 : E02;
        const float row2 = (ql == 0) ? E11 : (ql == 1) ? E12 : (ql == 2) ? E22 : x;
        s_pt[ql][pq] = row0;
        s_pt[4 + ql][pq] = row1;
        s_pt[8 + ql][pq] = row2;
        if (ql < 2) s_pt[12 + ql][pq] = (ql == 0) ? y : z;
      } else {
        // gradient-only pass: the 8 terms of the point, formed by every lane of the quad (identical values); lane l stores
        // rows l and l + 4
        float o[29];
        if (npairs != 0.f) {
          point_terms(false, x, y, z, score, npairs, A0, A1, A2, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, L->jang, L->hang, o);
        } else {
#pragma unroll
          for (int k = 0; k < NDT_NRED_GRAD; k++) o[k] = 0.f;
        }
        s_o[ql][pq] = (ql == 0) ? o[0] : (ql == 1) ? o[1] : (ql == 2) ? o[2] : o[3];
        s_o[4 + ql][pq] = (ql == 0) ? o[4] : (ql == 1) ? o[5] : (ql == 2) ? o[6] : o[7];
      }
      i += stride;
      x = 0.f; y = 0.f; z = 0.f;
      if (i < P.n) { x = P.sx[i]; y = P.sy[i]; z = P.sz[i]; }   // next batch's loads fly under phases B and C
    }
    LSR_STAMP(12)
    barrier_lds_only();
    // ---- phase B (passes with Hessian)
    if (hess) {
      if (tid < PTS) {
        const float npairs = s_pt[1][tid];
        float o[29];
        if (npairs != 0.f) {
          point_terms(true, s_pt[11][tid], s_pt[12][tid], s_pt[13][tid], s_pt[0][tid], npairs, s_pt[2][tid], s_pt[3][tid], s_pt[4][tid],
                      s_pt[5][tid], s_pt[6][tid], s_pt[7][tid], s_pt[8][tid], s_pt[9][tid], s_pt[10][tid], L->jang, L->hang, o);
        } else {
#pragma unroll
          for (int k = 0; k < 29; k++) o[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 29; k++) s_o[k][tid] = o[k];
      }
      LSR_STAMP(13)
      barrier_lds_only();
    }
    LSR_STAMP(14)
    // ---- phase C: the float terms of the reference's per-point sums, summed in double chunk by chunk (canon::) and added
    // to the bank piece by piece: the integer pieces of the batch's chunks meet in one lane first (shuffle), so a value
    // costs at most NDT_NBINS atomics per batch
    if (!hess) {
      constexpr int RUNS = PTS / 8;                       // runs of 8 points per value: 8 per chunk
      if (tid < NDT_NRED_GRAD * RUNS) {                   // whole waves
        const int v = tid / RUNS, k = tid & 7;
        const double t = canon::reduce_grad(&s_o[v][8 * (tid % RUNS)]);
        bool poison;
        long long m = canon::piece(t, k < NDT_NBINS ? k : 0, &poison);   // 64-bit from here: two 31-bit pieces may not fit 32 bits
        int pz = poison ? 1 : 0;
        if (PTS == 128) { m += (long long)__shfl_xor((int)m, 8, 64); pz |= __shfl_xor(pz, 8, 64); }   // the batch's second chunk
        if ((tid & (RUNS - 1)) < NDT_NBINS) {
          unsigned long long* dst = one_batch ? reinterpret_cast<unsigned long long*>(bank) : s_ibin;
          if (m != 0) atomicAdd(dst + k * 32 + v, (unsigned long long)m);
          if (pz && k == 0) atomicAdd(dst + 31, 1ull);
        }
      }
    } else {
      constexpr int RUNS = PTS / 16;                      // runs of 16 points per value: 4 per chunk
      if (tid < ((29 * RUNS + 63) & ~63)) {               // whole waves, the lanes beyond value 28 idle along
        const int v = tid / RUNS, sg = tid & 3;
        double t = 0.0;
        if (v < 29) t = canon::reduce_hess(&s_o[v][16 * (tid % RUNS)]);
        bool poison, poison0;
        long long m = canon::piece(t, sg + 1, &poison);   // lanes 0..3 of a chunk take pieces 1..4,
        long long m0 = canon::piece(t, 0, &poison0);      // lane 0 also piece 0 (zero unless the total exceeds 2^31)
        int pz = poison0 ? 1 : 0;
        if (PTS == 128) {   // the batch's second chunk (64-bit sums: two 31-bit pieces may not fit 32 bits)
          m += (long long)__shfl_xor((int)m, 4, 64); m0 += (long long)__shfl_xor((int)m0, 4, 64); pz |= __shfl_xor(pz, 4, 64);
        }
        if (v < 29 && (tid & (RUNS - 1)) < 4) {
          unsigned long long* dst = one_batch ? reinterpret_cast<unsigned long long*>(bank) : s_ibin;
          if (m != 0) atomicAdd(dst + (sg + 1) * 32 + v, (unsigned long long)m);
          if (sg == 0) {
            if (m0 != 0) atomicAdd(dst + v, (unsigned long long)m0);
            if (pz) atomicAdd(dst + 31, 1ull);
          }
        }
      }
    }
    // the next batch writes s_pt / s_o again: with Hessian (and in tile mode) its writes are behind a barrier of the next round;
    // a gradient-only round without the tile barriers writes s_o straight away
    if (!hess && TAB != NDT_TAB_TILE && base + stride < P.n) barrier_lds_only();
  }

  LSR_STAMP(2)
  LSR_PASS_MARK(_p_t2)
  if (!one_batch) {
    __syncthreads();
    if (tid < NDT_NBINS * 32) {
      const unsigned long long m = s_ibin[tid];
      if (m != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(bank + tid), m);
    }
  }
  LSR_STAMP(3)
  LSR_PASS_END(hess)
  LSR_SPAN_END(seq)
}

// ===========================================================================================
// K3 + K4, "lane" variant: ONE lane per source point, wave-private canonical reduction
// ===========================================================================================
// The kernel of candidate SETS (many registrations per launch, blockIdx.y = registration) and of single registrations with
// enough points to fill the chip on their own.  Same arithmetic as the quad kernel to the last bit (canon:: above): a lane
// forms its point's four partial neighbour sums one after the other and adds them in the quad's tree order, the 64 points of a
// wave ARE one canonical chunk, and the chunk is reduced by the wave alone — terms staged through a wave-private LDS tile,
// canon::reduce_grad / reduce_hess, chunk totals split into the exact integer bins — so there is no workgroup barrier
// anywhere after the head, no per-lane fp64 accumulator (round 3's one-lane kernel carried 29 of them = 58 VGPRs and summed a
// lane's points across its batches, which made the answer depend on the launch geometry) and no partial rows.
//  THREADS: 512 or 1024 lanes sharing one LDS image of the voxel table (1024: one workgroup per CU, the table is copied once
//           per CU and pass).
//  nb: workgroups per registration in THIS launch (grid.x) — the answer does not depend on it, so the host widens it from
//      launch to launch as the registrations of a set finish (run_ndt_feeder).
//  tab_bytes: size of the table region at the start of the dynamic LDS (largest image of the set, multiple of 1 KiB); the
//      staging tiles follow it.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4f LdsV4;
typedef __attribute__((address_space(1))) const v4f GlbV4;
typedef __attribute__((address_space(1))) const float GlbFloat;
typedef __attribute__((address_space(1))) const int GlbInt;

//  SPLIT (round 6; single scans that do not fill the chip with one lane per point: cfg 5's 120k points are 1.8 waves per SIMD, 73 % of
//      their cycles waiting on the dependent gather -> exp -> weight chain of seven neighbours one after the other): TWO WAVES per
//      canonical chunk.  Wave 2p takes the point's partial sums 0 and 1 (neighbours 0, 4, 1, 5), wave 2p + 1 partial sums 2 and 3
//      (neighbours 2, 6, 3) — the two HALVES of the canonical per-point tree (p0 + p1) + (p2 + p3) — the odd wave parks its half in its
//      staging tile, one workgroup barrier, the even wave adds the halves, forms the point's terms and reduces the chunk as before.
//      Same bits (the association is the canonical one), half the serial chain per wave, twice the waves: a workgroup covers
//      THREADS / 2 points per trip.
template <int NOFF, int TAB, int THREADS, bool BYVAL, bool SPLIT = false, bool KD = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4))) void ndt_eval_lane_kernel(const NdtProblem pv, const NdtProblem* __restrict__ probs, const int seq,
                                                                const int nb, const int tab_bytes) {
  constexpr int NWAVES = THREADS / 64;
  constexpr int PTS = SPLIT ? THREADS / 2 : THREADS;   // source points a workgroup covers per trip
  constexpr int NT = (NOFF + 3) / 4;                // neighbours per partial sum
  constexpr int GROUP = 4;                          // records fetched per gather round trip
  const NdtProblem& P = BYVAL ? pv : probs[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // a workgroup without points only stays if it is the one that carries the controller state forward
  if (blockIdx.x > 0 && (long long)blockIdx.x * PTS >= (long long)P.n) return;
  const int role = SPLIT ? (wave & 1) : 0;      // SPLIT: which half of the per-point tree this wave forms
  const int cw = SPLIT ? (wave >> 1) : wave;    // the chunk of the trip this wave works on

  __shared__ double s_bin[NDT_NBINS][32];
  __shared__ unsigned long long s_ibin[NDT_NBINS * 32];   // this workgroup's chunk totals, exact (ds_add_u64)
  __shared__ double s_sum[NDT_NRED];
  __shared__ double s_lu[8][2];
  constexpr int STATE_Q = (int)(sizeof(NdtState) / 16);
  static_assert(STATE_Q <= THREADS, "NdtState copy assumes <= THREADS uint4");
  __shared__ uint4 s_state_q[STATE_Q];
  unsigned int* s_state = reinterpret_cast<unsigned int*>(s_state_q);
  extern __shared__ uint4 s_table[];   // [voxel table image: tab_bytes | NWAVES staging tiles of canon::TILE_FLOATS floats]

  const NdtState* __restrict__ Sin = P.st + (seq & 1);
  NdtState* __restrict__ Sout = P.st + ((seq + 1) & 1);

  // ---- head: the same steps as the quad kernel's (bins of the previous launch, controller on wave 0, table DMA by the others)
  const int n = P.n;
  const int stride = nb * PTS;
  int i = blockIdx.x * PTS + cw * 64 + lane;
  float x = 0.f, y = 0.f, z = 0.f;
  unsigned int ang_entry = 0u, ang_entry_b = 0u;
  {
    const uint4* gq = reinterpret_cast<const uint4*>(Sin);
    const uint4 stq = (tid < STATE_Q) ? gq[tid] : make_uint4(0u, 0u, 0u, 0u);
    long long msum = 0;
    if (seq > 0 && tid < NDT_NBINS * 32) {
      const long long* b = P.bins + (size_t)((seq + 2) % NDT_NBANKS) * NDT_BANK_WORDS + tid;
      long long m[NDT_NSHARDS];
#pragma unroll
      for (int sh = 0; sh < NDT_NSHARDS; sh++) m[sh] = b[sh * (NDT_NBINS * 32)];
      msum = ((m[0] + m[1]) + (m[2] + m[3])) + ((m[4] + m[5]) + (m[6] + m[7]));
    }
    if (tid < NDT_NBINS * 32) {
      s_bin[tid >> 5][tid & 31] = (double)msum * canon::quantum(tid >> 5);
      s_ibin[tid] = 0ull;
    }
    if (tid < STATE_Q) s_state_q[tid] = stq;
    if (i < n) { x = P.sx[i]; y = P.sy[i]; z = P.sz[i]; }   // after the shared lines, on purpose (vmcnt retires in order)
    if (tid < 64) ang_entry = k_angle_entries[tid];
    if (tid < 8) ang_entry_b = k_angle_entries[64 + tid];
  }
  barrier_lds_only();
  LdsState* L = (LdsState*)s_state;
  if (P.mailbox != nullptr && blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(&P.mailbox->progress, ((unsigned long long)(unsigned int)L->token << 32) | (unsigned int)seq,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (uniform_i(L->done)) {
    if (blockIdx.x == 0 && seq > 0) {
      uint4* gq = reinterpret_cast<uint4*>(Sout);
      if (tid < STATE_Q) gq[tid] = s_state_q[tid];
    }
    return;
  }
  if (TAB == NDT_TAB_LDS) {
    const int nchunks = P.lds_bytes >> 10;
    const unsigned char* img = reinterpret_cast<const unsigned char*>(P.lds_image) + (size_t)lane * 16;
    unsigned char* dst = reinterpret_cast<unsigned char*>(s_table);
    if (tid >= 64)
      for (int c = wave - 1; c < nchunks; c += NWAVES - 1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (size_t)c * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + (size_t)c * 1024), 16, 0, 0);
  }
  if (tid < 64) {
    if (seq > 0) {
      if (tid < NDT_NRED) {
        double t = (((s_bin[4][tid] + s_bin[3][tid]) + s_bin[2][tid]) + s_bin[1][tid]) + s_bin[0][tid];
        if (s_bin[0][31] != 0.0 || s_bin[1][31] != 0.0) t = __longlong_as_double(0x7FF8000000000000ll);
        s_sum[tid] = t;
      }
      wave_lds_fence();
      ndt_controller_wave0(L, (const LdsDouble*)s_sum);
      wave_lds_fence();
      build_request_wave0(reinterpret_cast<NdtState*>(s_state), &s_lu[0][0], reinterpret_cast<float*>(&s_lu[4][0]), ang_entry, ang_entry_b);
    }
    if (blockIdx.x == 0) {
      uint4* gq = reinterpret_cast<uint4*>(Sout);
      gq[tid] = s_state_q[tid];
      if (tid + 64 < STATE_Q) gq[tid + 64] = s_state_q[tid + 64];
    }
  }
  __syncthreads();  // the request is complete and the table DMA has landed (vmcnt(0) + barrier)
  if (uniform_i(L->done)) {
    if (P.mailbox != nullptr && blockIdx.x == 0 && tid == 0) {
      NdtMailbox* mb = P.mailbox;
#pragma unroll
      for (int k = 0; k < 16; k++) mb->final_T[k] = L->final_T[k];
      mb->converged = L->converged;
      mb->nr_iterations = L->nr_iterations;
      mb->n_evals = L->n_evals;
      mb->trans_probability = L->trans_probability;
      mb->last_pairs = L->last_pairs;
      __threadfence_system();
      __hip_atomic_store(&mb->done, (unsigned int)L->token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (blockIdx.x == 0) {  // clear the bank launch seq + 1 will add to (read last by launch seq - 1, which is complete)
    uint4* zb = reinterpret_cast<uint4*>(P.bins + (size_t)((seq + 1) % NDT_NBANKS) * NDT_BANK_WORDS);
    for (int k = tid; k < NDT_BANK_WORDS / 2; k += THREADS) zb[k] = make_uint4(0u, 0u, 0u, 0u);
  }

  // ---- this launch's request: wave-uniform values in scalar registers
  const bool hess = uniform_i(L->want_hessian) != 0;
  const double d1d = uniform_d(L->d1);
  const float d2 = uniform_f((float)L->d2);
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = uniform_f(L->T[k]);
  const float leaf = P.leaf;
  const int mb0 = P.min_b[0], mb1 = P.min_b[1], mb2 = P.min_b[2];
  const int xb0 = P.max_b[0], xb1 = P.max_b[1], xb2 = P.max_b[2];
  const int mul1 = P.mul1, mul2 = P.mul2;
  const __attribute__((address_space(3))) unsigned short* s_map = (const __attribute__((address_space(3))) unsigned short*)s_table;
  const int map_bytes = P.lds_map_bytes;
  float* s_tile = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(s_table) + tab_bytes) + wave * canon::TILE_FLOATS;
  // pointers read from a problem record in memory are generic to the compiler (flat loads, which also count against the LDS
  // counter): say where they point
  const GlbV4* g_rec = (const GlbV4*)P.rec;
  const GlbInt* g_cell_slot = (const GlbInt*)P.cell_slot;
  const GlbFloat* g_sx = (const GlbFloat*)P.sx;
  const GlbFloat* g_sy = (const GlbFloat*)P.sy;
  const GlbFloat* g_sz = (const GlbFloat*)P.sz;
  (void)g_rec; (void)g_cell_slot; (void)s_map; (void)map_bytes;

  // one canonical chunk per wave (pair of waves) and trip.  SPLIT: the trip count is the WORKGROUP's (every wave meets the barrier
  // inside), a wave whose chunk lies beyond the cloud works on absent points (zeros, no pairs: nothing reaches the bins)
  for (int base = SPLIT ? blockIdx.x * PTS : blockIdx.x * PTS + cw * 64; base < n; base += stride) {
    const bool have = i < n;
    const float px = x, py = y, pz = z;
    const float tx = xform_ref(T[0], T[1], T[2], T[3], px, py, pz);
    const float ty = xform_ref(T[4], T[5], T[6], T[7], px, py, pz);
    const float tz = xform_ref(T[8], T[9], T[10], T[11], px, py, pz);
    i += stride;
    if (i < n) { x = g_sx[i]; y = g_sy[i]; z = g_sz[i]; }   // the next chunk's loads fly under this one's maths
    const float fx = floorf(tx / leaf), fy = floorf(ty / leaf), fz = floorf(tz / leaf);
    const bool finite_ok = have && (fabsf(fx) < 1.0e9f) && (fabsf(fy) < 1.0e9f) && (fabsf(fz) < 1.0e9f);
    const int ci = finite_ok ? (int)fx : INT_MIN / 2, cj = finite_ok ? (int)fy : INT_MIN / 2, ck = finite_ok ? (int)fz : INT_MIN / 2;
    // per-axis bounds tests of the centre cell and its two neighbours, linear index of the centre: a neighbour's cell is
    // centre + dx + dy mul1 + dz mul2 — no multiplication, no branch per neighbour
    bool inx[3], iny[3], inz[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      inx[d] = (ci + d - 1 >= mb0) & (ci + d - 1 <= xb0);
      iny[d] = (cj + d - 1 >= mb1) & (cj + d - 1 <= xb1);
      inz[d] = (ck + d - 1 >= mb2) & (ck + d - 1 <= xb2);
    }
    const int centre = (ci - mb0) + (cj - mb1) * mul1 + (ck - mb2) * mul2;

    // every neighbour's record address first: NOFF cell -> slot lookups in flight at once (one LDS round trip), so that the
    // record gathers below depend on nothing but their own address
    bool nb_ok[NOFF];
    int nb_rec[NOFF];     // LDS table: byte offset of the record inside the table image; global tables: record index
#pragma unroll
    for (int o = 0; o < NOFF; o++) {
      int dx, dy, dz;
      Offsets<NOFF>::get(o, dx, dy, dz);
      const bool in = inx[dx + 1] & iny[dy + 1] & inz[dz + 1];
      const int cell = in ? centre + dx + dy * mul1 + dz * mul2 : 0;
      nb_ok[o] = in;
      nb_rec[o] = cell;
    }
    if (KD) {   // KDTREE (a template form of its own): of the 27 cells, the leaves whose centroid the kd-tree's radius search would return
#pragma unroll
      for (int o = 0; o < NOFF; o++) {
        const int ks = nb_ok[o] ? P.cell_slot[nb_rec[o]] : -1;
        const float4 cen = P.centroid[ks >= 0 ? ks : 0];
        nb_ok[o] = (ks >= 0) & centroid_in_radius(tx, ty, tz, cen.x, cen.y, cen.z, P.radius2);
      }
    }
    if (TAB == NDT_TAB_LDS) {
      int sl[NOFF];
#pragma unroll
      for (int o = 0; o < NOFF; o++) sl[o] = (int)s_map[nb_rec[o]];
#pragma unroll
      for (int o = 0; o < NOFF; o++) {
        nb_ok[o] = nb_ok[o] & (sl[o] != 0xFFFF);    // the LDS table only holds usable leaves
        nb_rec[o] = map_bytes + (nb_ok[o] ? sl[o] : 0) * NDT_LDS_REC_BYTES;
      }
    } else if (TAB == NDT_TAB_COMPACT) {
      int sl[NOFF];
#pragma unroll
      for (int o = 0; o < NOFF; o++) sl[o] = g_cell_slot[nb_rec[o]];
#pragma unroll
      for (int o = 0; o < NOFF; o++) {
        nb_ok[o] = nb_ok[o] & (sl[o] >= 0);
        nb_rec[o] = sl[o] >= 0 ? sl[o] : 0;
      }
    }

    // a neighbour NO lane of the wave can use (the layer above the scan, the one below the ground: a wave's 64 points are
    // neighbours in space) is skipped by the whole wave — a uniform branch; a pair that is not ok leaves every sum as it was,
    // so skipping it changes no bit
    bool nb_any[NOFF];
#pragma unroll
    for (int o = 0; o < NOFF; o++) nb_any[o] = __builtin_amdgcn_ballot_w64(nb_ok[o]) != 0ull;   // the i1 itself: __ballot(int) takes the
                                                                                                 // predicate through a VGPR (v_cndmask 0/1 + v_cmp_ne per neighbour)

    // the four partial sums of the point in the quad kernel's order: partial q takes neighbours q, q + 4, q + 8, ...;
    // halves (0, 1) and (2, 3) are summed first, then the two halves
    float S[11];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (SPLIT && half != role) continue;   // wave-uniform: this wave forms ONE half
      float Pq[2][11];
#pragma unroll
      for (int k = 0; k < 11; k++) { Pq[0][k] = 0.f; Pq[1][k] = 0.f; }
#pragma unroll
      for (int e0 = 0; e0 < 2 * NT; e0 += GROUP) {
        v4f r0[GROUP], r1[GROUP], r2[GROUP];
#pragma unroll
        for (int u = 0; u < GROUP; u++) {
          const int e = e0 + u, ql = e / NT, t = e % NT, o = 2 * half + ql + 4 * t;
          if (e < 2 * NT && o < NOFF) {   // (loaded unconditionally: a branch around the gathers costs more registers than it saves time)
            if (TAB == NDT_TAB_LDS) {
              const LdsV4* rp = (const LdsV4*)((const __attribute__((address_space(3))) unsigned char*)s_table + nb_rec[o]);
              r0[u] = rp[0]; r1[u] = rp[1]; r2[u] = rp[2];
            } else {
              const GlbV4* rp = g_rec + (size_t)nb_rec[o] * 4;   // empty / unusable cells hold NaN records
              r0[u] = rp[0]; r1[u] = rp[1]; r2[u] = rp[2];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < GROUP; u++) {
          const int e = e0 + u, ql = e / NT, t = e % NT, o = 2 * half + ql + 4 * t;
          if (e < 2 * NT && o < NOFF && nb_any[o]) {
            float* A = Pq[ql];
            pair_terms(nb_ok[o], hess, tx, ty, tz, make_float4(r0[u].x, r0[u].y, r0[u].z, r0[u].w), make_float4(r1[u].x, r1[u].y, r1[u].z, r1[u].w),
                       make_float4(r2[u].x, r2[u].y, r2[u].z, r2[u].w), d2, d1d, A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7], A[8], A[9], A[10]);
          }
        }
      }
      // (lane 2 half) + (lane 2 half + 1) of the quad
      if (half == 0 || SPLIT) {
#pragma unroll
        for (int k = 0; k < 11; k++) S[k] = Pq[0][k] + Pq[1][k];
      } else {
#pragma unroll
        for (int k = 0; k < 11; k++) S[k] = S[k] + (Pq[0][k] + Pq[1][k]);
      }
    }
    if (SPLIT) {
      // the halves meet: the odd wave parks (p2 + p3) in its own tile, the even wave adds it to its (p0 + p1) — the canonical order
      if (role == 1) {
#pragma unroll
        for (int k = 0; k < 11; k++) s_tile[k * canon::TILE_PITCH + lane] = S[k];
      }
      __syncthreads();
      if (role == 1) {
        // nothing else to do for this chunk: the even wave carries it from here
        if (base + stride < n) __syncthreads();   // the tile is read before the next trip writes it again
        continue;
      }
      const float* other = s_tile + canon::TILE_FLOATS;   // the partner's tile (wave + 1)
#pragma unroll
      for (int k = 0; k < 11; k++) S[k] = S[k] + other[k * canon::TILE_PITCH + lane];
      if (base + stride < n) __syncthreads();
    }
    // S = {score, #pairs, A0..2, E00, E01, E02, E11, E12, E22}
    // A point without a valid pair contributes zeros.  Its sums S are all zero already; its coordinates are replaced by zeros so
    // that every product below is +-0 whatever the input held (a NaN point has no pair either): the terms come out +-0, the chunk
    // sums and their integer pieces are the same bits as with an explicit "else: zeros" — which cost a branch whose two sides the
    // compiler merged with 60 register moves per point (round 5, ISA of round 4's kernel)
    float o[29];
    {
      const bool any_pair = S[1] != 0.f;
      const float qx = any_pair ? px : 0.f, qy = any_pair ? py : 0.f, qz = any_pair ? pz : 0.f;
      point_terms(hess, qx, qy, qz, S[0], S[1], S[2], S[3], S[4], S[5], S[6], S[7], S[8], S[9], S[10], L->jang, L->hang, o);
    }
    // ---- the chunk's canonical sums -> this workgroup's exact bins
    if (!hess) {
#pragma unroll
      for (int k = 0; k < NDT_NRED_GRAD; k++) s_tile[k * canon::TILE_PITCH + lane] = o[k];
      wave_lds_fence();
      const int v = lane >> 3, g = lane & 7;
      const double t = canon::reduce_grad(s_tile + v * canon::TILE_PITCH + 8 * g);
      if (g < NDT_NBINS) canon::add_piece_lds(s_ibin, v, g, t);
    } else {
#pragma unroll
      for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int k = 0; k < 16; k++)
          if (16 * r + k < 29) s_tile[k * canon::TILE_PITCH + lane] = o[16 * r + k];
        wave_lds_fence();
        const int v = 16 * r + (lane >> 2), sg = lane & 3;
        const double t = canon::reduce_hess(s_tile + (lane >> 2) * canon::TILE_PITCH + 16 * sg);
        if (v < 29) {
          canon::add_piece_lds(s_ibin, v, sg + 1, t);          // lanes 0..3 of the value take bins 1..4,
          if (sg == 0) canon::add_piece_lds(s_ibin, v, 0, t);   // lane 0 also bin 0 (zero unless the total exceeds 2^31)
        }
      }
    }
  }

  // ---- workgroup totals -> the registration's accumulator bank (integer atomics: exact, order independent)
  __syncthreads();
  if (tid < NDT_NBINS * 32) {
    const unsigned long long m = s_ibin[tid];
    if (m != 0ull) {
      long long* bank = P.bins + (size_t)(seq % NDT_NBANKS) * NDT_BANK_WORDS + (size_t)(blockIdx.x & (NDT_NSHARDS - 1)) * (NDT_NBINS * 32);
      atomicAdd(reinterpret_cast<unsigned long long*>(bank + tid), m);
    }
  }
}

#ifdef LSR_TIMING
}  // namespace
extern "C" int lsr_debug_timing_buffer(long long** out) {
  static long long* buf = nullptr;
  if (!buf) {
    if (hipMalloc((void**)&buf, 1024 * 32 * sizeof(long long)) != hipSuccess) return -1;
    (void)hipMemset(buf, 0, 1024 * 32 * sizeof(long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lsr_timing), &buf, sizeof(buf));
  }
  *out = buf;
  return 0;
}
namespace {
#endif

}  // namespace

}  // namespace lsr
// Host-side self check of the table-driven angle coefficients (no device needed): the scalar formulas of
// angle_tables() next to the 72 table entries evaluated exactly as build_request() does on the device.
extern "C" int lsr_debug_angle_tables(const double* p6, int d1_sign, float* jang_ref, float* hang_ref, float* jang_tab, float* hang_tab) {
  if (!p6 || !jang_ref || !hang_ref || !jang_tab || !hang_tab) return LSR_ERR_INVALID_ARGUMENT;
  lsr::angle_tables(p6, true, d1_sign, jang_ref, hang_ref);
  double f[8];
  f[0] = 1.0; f[7] = 0.0;
  for (int k = 0; k < 3; k++) {
    const double a = p6[3 + k];
    if (std::fabs(a) < 10e-5) { f[1 + k] = 1.0; f[4 + k] = 0.0; } else { f[1 + k] = std::cos(a); f[4 + k] = std::sin(a); }
  }
  for (int t = 0; t < 72; t++) {
    double v = lsr::angle_entry_value(lsr::k_angle_entries_host[t], f);
    if (t == 24 + 20 && d1_sign < 0) v = -v;
    if (t < 24) jang_tab[t] = (float)v; else hang_tab[t - 24] = (float)v;
  }
  return LSR_OK;
}
namespace lsr {

// Start of a single align(): the initial controller state travels in the KERNEL ARGUMENTS (1 KiB) and one small launch
// writes both state buffers and clears the accumulator banks — instead of a host-to-device copy (SDMA latency) plus a
// memset in front of the launch chain.
namespace {
__global__ __launch_bounds__(256) void ndt_init_kernel(const NdtState st, NdtState* __restrict__ out, long long* __restrict__ bins) {
  constexpr int STATE_Q = (int)(sizeof(NdtState) / 16);
  const uint4* src = reinterpret_cast<const uint4*>(&st);
  uint4* dst = reinterpret_cast<uint4*>(out);
  for (int k = threadIdx.x; k < 2 * STATE_Q; k += 256) dst[k] = src[k % STATE_Q];
  if (bins) {
    uint4* zb = reinterpret_cast<uint4*>(bins);
    for (int k = threadIdx.x; k < NDT_NBANKS * NDT_BANK_WORDS / 2; k += 256) zb[k] = make_uint4(0u, 0u, 0u, 0u);
  }
}
}  // namespace

int ndt_init_single(const NdtState& st, NdtState* d_state2, long long* d_bins, hipStream_t stream) {
  hipLaunchKernelGGL(ndt_init_kernel, dim3(1), dim3(256), 0, stream, st, d_state2, d_bins);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

// The same for the members of a candidate set: workgroup b takes member b.  Until round 5 a chain started with two host-to-device
// copies and a memset (three blit launches, ~25 us on the stream and three runtime calls on the host per chain).
namespace {
__global__ __launch_bounds__(256) void ndt_init_batch_kernel(const NdtProblem* __restrict__ src_probs, const NdtState* __restrict__ src_states,
                                                             NdtProblem* __restrict__ d_probs, NdtState* __restrict__ d_states,
                                                             long long* __restrict__ d_bins) {
  static_assert(sizeof(NdtProblem) % 4 == 0 && sizeof(NdtState) % 16 == 0, "copied as words / uint4");
  const int b = blockIdx.x;
  constexpr int STATE_Q = (int)(sizeof(NdtState) / 16);
  const uint4* ss = reinterpret_cast<const uint4*>(src_states + 2 * (size_t)b);
  uint4* ds = reinterpret_cast<uint4*>(d_states + 2 * (size_t)b);
  for (int k = threadIdx.x; k < 2 * STATE_Q; k += 256) ds[k] = ss[k];
  const unsigned int* sp = reinterpret_cast<const unsigned int*>(src_probs + b);
  unsigned int* dp = reinterpret_cast<unsigned int*>(d_probs + b);
  for (int k = threadIdx.x; k < (int)(sizeof(NdtProblem) / 4); k += 256) dp[k] = sp[k];
  uint4* zb = reinterpret_cast<uint4*>(d_bins + (size_t)b * NDT_NBANKS * NDT_BANK_WORDS);
  for (int k = threadIdx.x; k < NDT_NBANKS * NDT_BANK_WORDS / 2; k += 256) zb[k] = make_uint4(0u, 0u, 0u, 0u);
}
}  // namespace

int ndt_init_batch(const NdtProblem* src_probs, const NdtState* src_states, NdtProblem* d_probs, NdtState* d_states, long long* d_bins,
                   int n, hipStream_t stream) {
  if (n <= 0) return LSR_OK;
  hipLaunchKernelGGL(ndt_init_batch_kernel, dim3(n), dim3(256), 0, stream, src_probs, src_states, d_probs, d_states, d_bins);
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

template <int NOFF, int TAB, int PTS, bool KD = false>
static int launch_quad_variant(bool byval, dim3 grid, size_t dyn_lds, hipStream_t stream, const NdtProblem& pv, const NdtProblem* d_probs, int seq) {
  static bool allowed[2][64] = {};
  if (dyn_lds > 32 * 1024) {
    int dev = 0;
    LSR_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !allowed[byval ? 1 : 0][dev]) {
      const void* fn = byval ? (const void*)ndt_eval_quad_kernel<NOFF, TAB, PTS, true, KD> : (const void*)ndt_eval_quad_kernel<NOFF, TAB, PTS, false, KD>;
      LSR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NDT_LDS_TABLE_MAX_QUAD));
      allowed[byval ? 1 : 0][dev] = true;
    }
  }
  if (byval) hipLaunchKernelGGL((ndt_eval_quad_kernel<NOFF, TAB, PTS, true, KD>), grid, dim3(4 * PTS), dyn_lds, stream, pv, d_probs, seq);
  else hipLaunchKernelGGL((ndt_eval_quad_kernel<NOFF, TAB, PTS, false, KD>), grid, dim3(4 * PTS), dyn_lds, stream, pv, d_probs, seq);
  return LSR_OK;
}

template <int NOFF, bool KD = false>
static int launch_quad(const NdtLaunchCfg& cfg, bool byval, dim3 grid, hipStream_t stream, const NdtProblem& pv, const NdtProblem* d_probs, int seq) {
  const size_t dyn = (cfg.tab == NDT_TAB_LDS || cfg.tab == NDT_TAB_TILE) ? (size_t)cfg.lds_bytes : 0;
  if (cfg.threads == 64) {  // points per workgroup
    switch (cfg.tab) {
      case NDT_TAB_LDS: return launch_quad_variant<NOFF, NDT_TAB_LDS, 64, KD>(byval, grid, dyn, stream, pv, d_probs, seq);
      case NDT_TAB_TILE:   // (never chosen for KDTREE — choose_table_mode —: that form is not instantiated)
        if constexpr (!KD) return launch_quad_variant<NOFF, NDT_TAB_TILE, 64, KD>(byval, grid, dyn, stream, pv, d_probs, seq);
        else { set_last_error("the tile mode has no KDTREE form"); return LSR_ERR_INVALID_ARGUMENT; }
      case NDT_TAB_COMPACT: return launch_quad_variant<NOFF, NDT_TAB_COMPACT, 64, KD>(byval, grid, dyn, stream, pv, d_probs, seq);
      default: return launch_quad_variant<NOFF, NDT_TAB_DENSE, 64, KD>(byval, grid, dyn, stream, pv, d_probs, seq);
    }
  }
  switch (cfg.tab) {
    case NDT_TAB_LDS: return launch_quad_variant<NOFF, NDT_TAB_LDS, 128, KD>(byval, grid, dyn, stream, pv, d_probs, seq);
    case NDT_TAB_TILE:   // (never chosen for KDTREE — choose_table_mode —: that form is not instantiated)
      if constexpr (!KD) return launch_quad_variant<NOFF, NDT_TAB_TILE, 128, KD>(byval, grid, dyn, stream, pv, d_probs, seq);
      else { set_last_error("the tile mode has no KDTREE form"); return LSR_ERR_INVALID_ARGUMENT; }
    case NDT_TAB_COMPACT: return launch_quad_variant<NOFF, NDT_TAB_COMPACT, 128, KD>(byval, grid, dyn, stream, pv, d_probs, seq);
    default: return launch_quad_variant<NOFF, NDT_TAB_DENSE, 128, KD>(byval, grid, dyn, stream, pv, d_probs, seq);
  }
}

template <int NOFF, int TAB, int THREADS, bool SPLIT = false, bool KD = false>
static int launch_lane_variant(bool byval, dim3 grid, size_t dyn_lds, hipStream_t stream, const NdtProblem& pv, const NdtProblem* d_probs, int seq,
                               int tab_bytes) {
  static bool allowed[2][64] = {};
  if (dyn_lds > 32 * 1024) {
    int dev = 0;
    LSR_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !allowed[byval ? 1 : 0][dev]) {
      const void* fn = byval ? (const void*)ndt_eval_lane_kernel<NOFF, TAB, THREADS, true, SPLIT, KD> : (const void*)ndt_eval_lane_kernel<NOFF, TAB, THREADS, false, SPLIT, KD>;
      // table image + staging tiles, bounded by what a workgroup can have on gfx950 (160 KiB minus the kernel's static LDS)
      const int want = std::min((int)NDT_LDS_TABLE_MAX + ndt_lane_tile_bytes(THREADS), 160 * 1024 - 6 * 1024);
      LSR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, want));
      allowed[byval ? 1 : 0][dev] = true;
    }
  }
  const int nb = (int)grid.x;
  if (byval) hipLaunchKernelGGL((ndt_eval_lane_kernel<NOFF, TAB, THREADS, true, SPLIT, KD>), grid, dim3(THREADS), dyn_lds, stream, pv, d_probs, seq, nb, tab_bytes);
  else hipLaunchKernelGGL((ndt_eval_lane_kernel<NOFF, TAB, THREADS, false, SPLIT, KD>), grid, dim3(THREADS), dyn_lds, stream, pv, d_probs, seq, nb, tab_bytes);
  return LSR_OK;
}

template <int NOFF, bool KD = false>
static int launch_lane(const NdtLaunchCfg& cfg, bool byval, dim3 grid, hipStream_t stream, const NdtProblem& pv, const NdtProblem* d_probs, int seq) {
  const int tab_bytes = (cfg.tab == NDT_TAB_LDS) ? cfg.lds_bytes : 0;
  const size_t dyn = (size_t)tab_bytes + (size_t)ndt_lane_tile_bytes(cfg.threads);
  if (cfg.split && cfg.threads == 512 && !KD) {   // two waves per chunk: the 512-thread form only (what single scans use; the KDTREE form has no such variant)
    switch (cfg.tab) {
      case NDT_TAB_LDS: return launch_lane_variant<NOFF, NDT_TAB_LDS, 512, true>(byval, grid, dyn, stream, pv, d_probs, seq, tab_bytes);
      case NDT_TAB_COMPACT: return launch_lane_variant<NOFF, NDT_TAB_COMPACT, 512, true>(byval, grid, dyn, stream, pv, d_probs, seq, tab_bytes);
      default: return launch_lane_variant<NOFF, NDT_TAB_DENSE, 512, true>(byval, grid, dyn, stream, pv, d_probs, seq, tab_bytes);
    }
  }
  if (cfg.threads == 1024) {
    switch (cfg.tab) {
      case NDT_TAB_LDS: return launch_lane_variant<NOFF, NDT_TAB_LDS, 1024, false, KD>(byval, grid, dyn, stream, pv, d_probs, seq, tab_bytes);
      case NDT_TAB_COMPACT: return launch_lane_variant<NOFF, NDT_TAB_COMPACT, 1024, false, KD>(byval, grid, dyn, stream, pv, d_probs, seq, tab_bytes);
      default: return launch_lane_variant<NOFF, NDT_TAB_DENSE, 1024, false, KD>(byval, grid, dyn, stream, pv, d_probs, seq, tab_bytes);
    }
  }
  switch (cfg.tab) {
    case NDT_TAB_LDS: return launch_lane_variant<NOFF, NDT_TAB_LDS, 512, false, KD>(byval, grid, dyn, stream, pv, d_probs, seq, tab_bytes);
    case NDT_TAB_COMPACT: return launch_lane_variant<NOFF, NDT_TAB_COMPACT, 512, false, KD>(byval, grid, dyn, stream, pv, d_probs, seq, tab_bytes);
    default: return launch_lane_variant<NOFF, NDT_TAB_DENSE, 512, false, KD>(byval, grid, dyn, stream, pv, d_probs, seq, tab_bytes);
  }
}

// Launches seq0 .. seq0+count-1 of the chain (launch seq consumes the bank of launch seq-1).  cfg.max_blocks = grid.x of these
// launches: fixed over a chain of the quad kernel (its workgroups stride by P.nblocks), free from launch to launch for the lane
// kernel (the canonical sum does not depend on it).
int ndt_launch_evals(const NdtProblem* d_probs, const NdtProblem* h_single, const NdtLaunchCfg& cfg, int seq0, int count,
                     hipStream_t stream) {
  const bool byval = (cfg.batch == 1 && h_single != nullptr);
  if (byval ? !h_single->bins : !d_probs) { set_last_error("the derivative kernels need their accumulator banks / problem array"); return LSR_ERR_INVALID_ARGUMENT; }
  dim3 grid(cfg.max_blocks, cfg.batch);
  NdtProblem pv;
  if (byval) pv = *h_single; else std::memset(&pv, 0, sizeof(pv));
  if (cfg.quad) {
    if (cfg.threads != 64 && cfg.threads != 128) { set_last_error("quad kernel: 64 or 128 points per workgroup"); return LSR_ERR_INVALID_ARGUMENT; }
  } else if (cfg.threads != 512 && cfg.threads != 1024) {
    set_last_error("lane kernel: 512 or 1024 threads per workgroup");
    return LSR_ERR_INVALID_ARGUMENT;
  }
  for (int i = 0; i < count; i++) {
    int st;
    if (cfg.quad) {
      switch (cfg.neighborhood) {
        case LSR_DIRECT1: st = launch_quad<1>(cfg, byval, grid, stream, pv, d_probs, seq0 + i); break;
        case LSR_KDTREE: st = launch_quad<27, true>(cfg, byval, grid, stream, pv, d_probs, seq0 + i); break;   // the 27 cells + the kd-tree's radius test (NdtProblem::centroid)
        case LSR_DIRECT26: st = launch_quad<27>(cfg, byval, grid, stream, pv, d_probs, seq0 + i); break;
        default: st = launch_quad<7>(cfg, byval, grid, stream, pv, d_probs, seq0 + i); break;
      }
    } else {
      switch (cfg.neighborhood) {
        case LSR_DIRECT1: st = launch_lane<1>(cfg, byval, grid, stream, pv, d_probs, seq0 + i); break;
        case LSR_KDTREE: st = launch_lane<27, true>(cfg, byval, grid, stream, pv, d_probs, seq0 + i); break;
        case LSR_DIRECT26: st = launch_lane<27>(cfg, byval, grid, stream, pv, d_probs, seq0 + i); break;
        default: st = launch_lane<7>(cfg, byval, grid, stream, pv, d_probs, seq0 + i); break;
      }
    }
    if (st) return st;
  }
  LSR_HIP(hipGetLastError());
  return LSR_OK;
}

}  // namespace lsr
