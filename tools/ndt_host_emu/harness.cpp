// TEST INFRASTRUCTURE.  Host emulation of the NDT derivative pass: csrc/ndt_point.hpp (the per-pair / per-point arithmetic of
// ndt_eval_quad_kernel and ndt_eval_kernel) compiled for the CPU, wrapped around the same point flow as the quad kernel — fmaf
// point transform, floor(x'/leaf) in fp32, four "lanes" per point taking neighbours l and l + 4 of DIRECT7, the quad sum
// (l0 + l1) + (l2 + l3) in fp32, the 29 per-point terms in fp32, accumulation over the points in fp64 — and exported as the
// derivative callback of the CPU oracle's Newton / More-Thuente loop (oracle/ndt_oracle.cpp: DerivCb).
// What it answers without a GPU: how far does the GPU's fp32 OPERATION ORDER alone move a registration away from the
// reference's order (tests/test_ndt_host_emu_cpu.py)?  What it does not reproduce: the device's expf / sincosf
// implementations (glibc's are used; both are within 1 ulp) and the fp64 summation grouping.
// Round 3 used it to find out why cfg-4 candidate 21 ended 2.4 mm from the oracle: with an fmaf-chain point transform and
// fp32-rounded voxel means (round 2's kernels; EMU_R02_ARITH=1 brings them back here) this emulation reproduced the GPU's
// trajectory to 4e-7 m; with the reference's transform order and head + tail means (today's kernels) it ends 3e-9 m from the
// oracle.
//   g++ -O2 -std=c++17 -shared -fPIC -mfma -ffp-contract=off harness.cpp -o libndtemu.so
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define LSR_HOST_EMU 1
struct float4 { float x, y, z, w; };
static inline float __fadd_rn(float a, float b) { return a + b; }   // this file is compiled with -ffp-contract=off
static inline float __fmul_rn(float a, float b) { return a * b; }
#include "../../lidarslam_ros2_amd/csrc/ndt_point.hpp"

extern "C" int lsr_debug_angle_tables(const double* p6, int d1_sign, float* jang_ref, float* hang_ref, float* jang_tab, float* hang_tab);

namespace {
struct Emu {
  // dense table over the grid: 12 floats per cell {mean_hi xyz, c00 | c01 c02 c11 c12 | c22, mean_lo xyz} as the leaf records
  // hold them (grid_device.hpp: leaf_record_dev), valid flag
  std::vector<float> rec;
  std::vector<unsigned char> valid;
  int min_b[3], max_b[3], mul1, mul2;
  float leaf;
  std::vector<float> sx, sy, sz;
  double d1, d2;
  int d1_sign;
  float T[12], jang[24], hang[48];
  long passes = 0;
};

double pass(Emu& E, bool hess, double* grad, double* hmat) {
  const int n = (int)E.sx.size();
  const float d2 = (float)E.d2;
  const double d1d = E.d1;
  const float* T = E.T;
  double acc[29];
  for (int k = 0; k < 29; k++) acc[k] = 0.0;
  static const int off7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  for (int i = 0; i < n; i++) {
    const float x = E.sx[i], y = E.sy[i], z = E.sz[i];
    static const bool r02 = getenv("EMU_R02_ARITH") != nullptr;   // round 2's kernels: fmaf-chain transform, fp32 means
    const float tx = r02 ? fmaf(T[0], x, fmaf(T[1], y, fmaf(T[2], z, T[3]))) : lsr::xform_ref(T[0], T[1], T[2], T[3], x, y, z);
    const float ty = r02 ? fmaf(T[4], x, fmaf(T[5], y, fmaf(T[6], z, T[7]))) : lsr::xform_ref(T[4], T[5], T[6], T[7], x, y, z);
    const float tz = r02 ? fmaf(T[8], x, fmaf(T[9], y, fmaf(T[10], z, T[11]))) : lsr::xform_ref(T[8], T[9], T[10], T[11], x, y, z);
    const float fx = floorf(tx / E.leaf), fy = floorf(ty / E.leaf), fz = floorf(tz / E.leaf);
    const bool finite_ok = (fabsf(fx) < 1.0e9f) && (fabsf(fy) < 1.0e9f) && (fabsf(fz) < 1.0e9f);
    if (!finite_ok) continue;
    const int ci = (int)fx, cj = (int)fy, ck = (int)fz;
    float lane[4][11];
    for (int l = 0; l < 4; l++) {
      float* a = lane[l];
      for (int k = 0; k < 11; k++) a[k] = 0.f;
      for (int t = 0; t < 2; t++) {
        const int o = l + 4 * t;
        if (o >= 7) continue;
        const int ca = ci + off7[o][0], cb = cj + off7[o][1], cc = ck + off7[o][2];
        const bool in = ca >= E.min_b[0] && ca <= E.max_b[0] && cb >= E.min_b[1] && cb <= E.max_b[1] && cc >= E.min_b[2] && cc <= E.max_b[2];
        const size_t cell = in ? (size_t)((ca - E.min_b[0]) + (cb - E.min_b[1]) * E.mul1 + (cc - E.min_b[2]) * E.mul2) : 0;
        const bool ok = in && E.valid[cell];
        const float* r = &E.rec[(ok ? cell : 0) * 12];
        const float4 r0 = {r[0], r[1], r[2], r[3]}, r1 = {r[4], r[5], r[6], r[7]};
        const float4 r2 = {r[8], r02 ? 0.f : r[9], r02 ? 0.f : r[10], r02 ? 0.f : r[11]};
        lsr::pair_terms(ok, hess, tx, ty, tz, r0, r1, r2, d2, d1d, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10]);
      }
    }
    float s[11];
    for (int k = 0; k < 11; k++) s[k] = (lane[0][k] + lane[1][k]) + (lane[2][k] + lane[3][k]);   // the DPP quad sum
    if (s[1] == 0.f) continue;
    float o[29];
    lsr::point_terms(hess, x, y, z, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], s[9], s[10], E.jang, E.hang, o);
    const int nred = hess ? 29 : 8;
    for (int k = 0; k < nred; k++) acc[k] += (double)o[k];
  }
  E.passes++;
  for (int k = 0; k < 6; k++) grad[k] = acc[1 + k];
  if (hess) {
    int k = 8;
    for (int a = 0; a < 6; a++)
      for (int b = a; b < 6; b++) { hmat[a * 6 + b] = acc[k]; hmat[b * 6 + a] = acc[k]; k++; }
  }
  return acc[0];
}
}  // namespace

extern "C" {

// leaves: idx (linear cell index), mean (n x 3), icov (n x 9 row-major) of the USABLE leaves only
void* emu_create(const int* min_b, const int* max_b, float leaf, int n_leaves, const int* idx, const double* mean, const double* icov,
                 const float* src_xyz, int n_src, double d1, double d2, int d1_sign) {
  Emu* E = new Emu();
  for (int k = 0; k < 3; k++) { E->min_b[k] = min_b[k]; E->max_b[k] = max_b[k]; }
  const int dx = max_b[0] - min_b[0] + 1, dy = max_b[1] - min_b[1] + 1, dz = max_b[2] - min_b[2] + 1;
  E->mul1 = dx; E->mul2 = dx * dy;
  E->leaf = leaf;
  E->rec.assign((size_t)dx * dy * dz * 12, 0.f);
  E->valid.assign((size_t)dx * dy * dz, 0);
  for (int l = 0; l < n_leaves; l++) {
    float* r = &E->rec[(size_t)idx[l] * 12];
    const double* m = mean + 3 * l; const double* c = icov + 9 * l;
    r[0] = (float)m[0]; r[1] = (float)m[1]; r[2] = (float)m[2];
    r[3] = (float)c[0]; r[4] = (float)c[1]; r[5] = (float)c[2]; r[6] = (float)c[4]; r[7] = (float)c[5]; r[8] = (float)c[8];
    for (int k = 0; k < 3; k++) r[9 + k] = (float)(m[k] - (double)r[k]);
    E->valid[idx[l]] = 1;
  }
  E->sx.resize(n_src); E->sy.resize(n_src); E->sz.resize(n_src);
  for (int i = 0; i < n_src; i++) { E->sx[i] = src_xyz[3 * i]; E->sy[i] = src_xyz[3 * i + 1]; E->sz[i] = src_xyz[3 * i + 2]; }
  E->d1 = d1; E->d2 = d2; E->d1_sign = d1_sign;
  return E;
}
void emu_destroy(void* e) { delete (Emu*)e; }
long emu_passes(void* e) { return ((Emu*)e)->passes; }

// oracle/ndt_oracle.cpp DerivCb
double emu_deriv_cb(void* user, const double* p, const float* T16, int mode, double* grad, double* hess) {
  Emu& E = *(Emu*)user;
  float jr[24], hr[48], jt[24], ht[48];
  if (mode == 3) {
    // first pass of align(): the point transform is the guess matrix itself (ndt_fill_initial_state)
    E.T[0] = T16[0]; E.T[1] = T16[4]; E.T[2] = T16[8];  E.T[3] = T16[12];
    E.T[4] = T16[1]; E.T[5] = T16[5]; E.T[6] = T16[9];  E.T[7] = T16[13];
    E.T[8] = T16[2]; E.T[9] = T16[6]; E.T[10] = T16[10]; E.T[11] = T16[14];
  } else if (mode != 2) {
    lsr::pose_to_T12(p, E.T);   // build_request: fp32 sines / cosines of the float-cast angles, compose_R12
  }
  if (mode != 2) {
    lsr_debug_angle_tables(p, E.d1_sign, jr, hr, jt, ht);   // the 72-entry table the device evaluates, one entry per lane
    std::memcpy(E.jang, jt, sizeof(jt));
    if (mode != 0) std::memcpy(E.hang, ht, sizeof(ht));     // gradient-only passes leave h_ang as it was (stale on purpose)
  }
  if (mode == 2) {
    double g_unused[6];
    pass(E, true, g_unused, hess);
    return 0.0;
  }
  return pass(E, mode != 0, grad, hess);
}

// one pass at pose p (mode as above): sums for direct comparison with oracle.ndt_derivatives / lsr_ndt_derivatives
double emu_derivatives(void* user, const double* p, const float* T16, int with_hessian, double* grad, double* hess) {
  for (int a = 0; a < 36; a++) hess[a] = 0;
  if (T16) {
    double s = emu_deriv_cb(user, p, T16, 3, grad, hess);
    if (!with_hessian) for (int a = 0; a < 36; a++) hess[a] = 0;
    return s;
  }
  return emu_deriv_cb(user, p, nullptr, with_hessian ? 1 : 0, grad, hess);
}

}  // extern "C"
