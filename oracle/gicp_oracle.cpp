// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// PARITY UNPINNED.  CPU restatement (C++17 + OpenMP) of pclomp::GeneralizedIterativeClosestPoint as
// lidarslam_ros2 calls it (scanmatcher/src/scanmatcher_component.cpp:115-120,315,329,353;
// graph_based_slam/src/graph_based_slam_component.cpp:73-82,181,227,230).  The sources live in the
// un-vendored submodule Thirdparty/ndt_omp_ros2 + PCL 1.12 (gicp.hpp, bfgs.h — a port of GSL's
// vector_bfgs2); this file follows SURVEY.md §9.7 and Segal et al. 2009:
//   * computeCovariances: k-NN (incl. the point itself), single-pass mean/cov with FLOAT products
//     accumulated in double, covariance -> U diag(1,1,eps) U^T;
//   * outer loop: 1-NN within corr_dist, M_i = (R C1 R^T + C2)^-1, inner optimisation, delta test;
//   * inner solver 0 = BFGS (Fletcher line search: rho/sigma/tau1..3, cubic interpolation,
//     |g| < 1e-2 or max_inner iterations) — the reference schedule;
//     inner solver 1 = Gauss-Newton on the same cost with the same stopping rule — what the GPU
//     core runs (north_star asks for 6x6 Hessian / 6x1 gradient accumulation).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "linalg.h"

extern "C" {
void orc_nn_search(void* gp, const float* q, size_t stride_floats, size_t n, const float* T16, int* idx, float* d2,
                   int num_threads);
void orc_knn_search(void* gp, const float* q, size_t stride_floats, size_t n, int k, int* idx, float* d2, int num_threads);
}

namespace {

struct GicpParams {
  double max_corr_dist, trans_eps, rot_eps, gicp_eps;
  int max_iterations, max_inner_iterations, k_correspondences, solver, num_threads;
};
struct GicpResult {
  float final_transformation[16];
  int converged, iterations, n_correspondences;
  double final_cost;
};

typedef double Vec6[6];

struct Problem {
  const float* src;  // guess-transformed source ("output"), packed xyz
  const float* tgt;
  size_t tgt_stride;
  const int* isrc;
  const int* itgt;
  const double* M;  // per SOURCE index, 9 doubles
  int m;
  int threads;
};

// applyState(t = identity, x): R = Rz(x5) Ry(x4) Rx(x3) in float, t = float(x0..2); col-major 4x4
void apply_state_f(const double* x, float* T) {
  float a = (float)x[3], b = (float)x[4], c = (float)x[5];
  float ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cc = std::cos(c), sc = std::sin(c);
  float Rz[9] = {cc, -sc, 0, sc, cc, 0, 0, 0, 1};
  float Ry[9] = {cb, 0, sb, 0, 1, 0, -sb, 0, cb};
  float Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca};
  float A[9], R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float s = 0;
      for (int k = 0; k < 3; k++) s += Rz[i * 3 + k] * Ry[k * 3 + j];
      A[i * 3 + j] = s;
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float s = 0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * Rx[k * 3 + j];
      R[i * 3 + j] = s;
    }
  for (int i = 0; i < 16; i++) T[i] = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[j * 4 + i] = R[i * 3 + j];
  T[12] = (float)x[0]; T[13] = (float)x[1]; T[14] = (float)x[2]; T[15] = 1.f;
}

inline void xform(const float* M, const float* x, float* o) {
  o[0] = M[0] * x[0] + M[4] * x[1] + M[8] * x[2] + M[12];
  o[1] = M[1] * x[0] + M[5] * x[1] + M[9] * x[2] + M[13];
  o[2] = M[2] * x[0] + M[6] * x[1] + M[10] * x[2] + M[14];
}

// dR/dphi, dR/dtheta, dR/dpsi of R = Rz(psi) Ry(theta) Rx(phi) (computeRDerivative)
void r_derivatives(const double* x, double* dPhi, double* dTheta, double* dPsi) {
  double phi = x[3], theta = x[4], psi = x[5];
  double cphi = std::cos(phi), sphi = std::sin(phi), ct = std::cos(theta), st = std::sin(theta), cpsi = std::cos(psi),
         spsi = std::sin(psi);
  double A[9] = {0, sphi * spsi + cphi * cpsi * st, cphi * spsi - cpsi * sphi * st,
                 0, -cpsi * sphi + cphi * spsi * st, -cphi * cpsi - sphi * spsi * st,
                 0, cphi * ct, -ct * sphi};
  double B[9] = {-cpsi * st, cpsi * ct * sphi, cphi * cpsi * ct,
                 -spsi * st, ct * sphi * spsi, cphi * ct * spsi,
                 -ct, -sphi * st, -cphi * st};
  double Cc[9] = {-ct * spsi, -cphi * cpsi - sphi * spsi * st, cpsi * sphi - cphi * spsi * st,
                  cpsi * ct, -cphi * spsi + cpsi * sphi * st, sphi * spsi + cphi * cpsi * st,
                  0, 0, 0};
  std::memcpy(dPhi, A, sizeof(A));
  std::memcpy(dTheta, B, sizeof(B));
  std::memcpy(dPsi, Cc, sizeof(Cc));
}

// f(x) = 1/m sum r^T M r ; g = gradient (OptimizationFunctorWithIndices::fdf)
double cost_grad(const Problem& P, const double* x, double* g /*nullable*/) {
  float T[16];
  apply_state_f(x, T);
  double f = 0, gt[3] = {0, 0, 0}, Rm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma omp parallel for reduction(+ : f) num_threads(P.threads) schedule(static)
  for (int i = 0; i < P.m; i++) {
    const float* ps = P.src + 3 * (size_t)P.isrc[i];
    const float* pt = P.tgt + P.tgt_stride * (size_t)P.itgt[i];
    float pp[3];
    xform(T, ps, pp);
    double res[3] = {(double)(pp[0] - pt[0]), (double)(pp[1] - pt[1]), (double)(pp[2] - pt[2])};
    const double* M = P.M + 9 * (size_t)P.isrc[i];
    double tmp[3];
    for (int a = 0; a < 3; a++) tmp[a] = M[a * 3] * res[0] + M[a * 3 + 1] * res[1] + M[a * 3 + 2] * res[2];
    f += res[0] * tmp[0] + res[1] * tmp[1] + res[2] * tmp[2];
  }
  if (g) {
    for (int i = 0; i < P.m; i++) {  // sequential: deterministic gradient
      const float* ps = P.src + 3 * (size_t)P.isrc[i];
      const float* pt = P.tgt + P.tgt_stride * (size_t)P.itgt[i];
      float pp[3];
      xform(T, ps, pp);
      double res[3] = {(double)(pp[0] - pt[0]), (double)(pp[1] - pt[1]), (double)(pp[2] - pt[2])};
      const double* M = P.M + 9 * (size_t)P.isrc[i];
      double tmp[3];
      for (int a = 0; a < 3; a++) tmp[a] = M[a * 3] * res[0] + M[a * 3 + 1] * res[1] + M[a * 3 + 2] * res[2];
      for (int a = 0; a < 3; a++) gt[a] += tmp[a];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) Rm[a * 3 + b] += (double)ps[a] * tmp[b];  // base_transformation_ = identity
    }
    double s = 2.0 / P.m;
    for (int a = 0; a < 3; a++) g[a] = gt[a] * s;
    for (int a = 0; a < 9; a++) Rm[a] *= s;
    double dA[9], dB[9], dC[9];
    r_derivatives(x, dA, dB, dC);
    auto inner = [&](const double* D) {  // tr(D * Rm)
      double r = 0;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r += D[j * 3 + i] * Rm[i * 3 + j];
      return r;
    };
    g[3] = inner(dA); g[4] = inner(dB); g[5] = inner(dC);
  }
  return f / P.m;
}

// Gauss-Newton step quantities: H = 2/m sum J^T M J, g = 2/m sum J^T M r, J = [I | dR_k p]
void gn_system(const Problem& P, const double* x, double* f, double* g, double* H) {
  float T[16];
  apply_state_f(x, T);
  double dA[9], dB[9], dC[9];
  r_derivatives(x, dA, dB, dC);
  double acc_f = 0, acc_g[6] = {0, 0, 0, 0, 0, 0}, acc_H[36];
  for (int a = 0; a < 36; a++) acc_H[a] = 0;
  for (int i = 0; i < P.m; i++) {
    const float* ps = P.src + 3 * (size_t)P.isrc[i];
    const float* pt = P.tgt + P.tgt_stride * (size_t)P.itgt[i];
    float pp[3];
    xform(T, ps, pp);
    double res[3] = {(double)(pp[0] - pt[0]), (double)(pp[1] - pt[1]), (double)(pp[2] - pt[2])};
    const double* M = P.M + 9 * (size_t)P.isrc[i];
    double p[3] = {ps[0], ps[1], ps[2]};
    double J[3][6];
    for (int a = 0; a < 3; a++) {
      J[a][0] = a == 0; J[a][1] = a == 1; J[a][2] = a == 2;
      J[a][3] = dA[a * 3] * p[0] + dA[a * 3 + 1] * p[1] + dA[a * 3 + 2] * p[2];
      J[a][4] = dB[a * 3] * p[0] + dB[a * 3 + 1] * p[1] + dB[a * 3 + 2] * p[2];
      J[a][5] = dC[a * 3] * p[0] + dC[a * 3 + 1] * p[1] + dC[a * 3 + 2] * p[2];
    }
    double Mr[3], MJ[3][6];
    for (int a = 0; a < 3; a++) {
      Mr[a] = M[a * 3] * res[0] + M[a * 3 + 1] * res[1] + M[a * 3 + 2] * res[2];
      for (int c = 0; c < 6; c++) MJ[a][c] = M[a * 3] * J[0][c] + M[a * 3 + 1] * J[1][c] + M[a * 3 + 2] * J[2][c];
    }
    acc_f += res[0] * Mr[0] + res[1] * Mr[1] + res[2] * Mr[2];
    for (int c = 0; c < 6; c++) {
      acc_g[c] += J[0][c] * Mr[0] + J[1][c] * Mr[1] + J[2][c] * Mr[2];
      for (int d = 0; d < 6; d++) acc_H[c * 6 + d] += J[0][c] * MJ[0][d] + J[1][c] * MJ[1][d] + J[2][c] * MJ[2][d];
    }
  }
  *f = acc_f / P.m;
  for (int c = 0; c < 6; c++) g[c] = 2.0 * acc_g[c] / P.m;
  for (int a = 0; a < 36; a++) H[a] = 2.0 * acc_H[a] / P.m;
}

// ---- BFGS (GSL vector_bfgs2 / PCL bfgs.h) -------------------------------------------------------
struct Bfgs {
  const Problem* P;
  double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step = 1.0;
  int order = 3;
  Vec6 x0, g0, p, dx0, dg0, x_alpha, g_alpha;
  double f_alpha, df_alpha, delta_f, fp0, pnorm, g0norm;
  double f_key, df_key, x_key, g_key;

  static double nrm(const double* v) { double s = 0; for (int i = 0; i < 6; i++) s += v[i] * v[i]; return std::sqrt(s); }
  static double dot(const double* a, const double* b) { double s = 0; for (int i = 0; i < 6; i++) s += a[i] * b[i]; return s; }
  void moveto(double alpha) {
    if (alpha == x_key) return;
    for (int i = 0; i < 6; i++) x_alpha[i] = x0[i] + alpha * p[i];
    x_key = alpha;
  }
  double slope() { return dot(g_alpha, p); }
  double eval_f(double alpha) {
    if (alpha == f_key) return f_alpha;
    moveto(alpha);
    f_alpha = cost_grad(*P, x_alpha, nullptr);
    f_key = alpha;
    return f_alpha;
  }
  double eval_df(double alpha) {
    if (alpha == df_key) return df_alpha;
    moveto(alpha);
    if (alpha != g_key) { cost_grad(*P, x_alpha, g_alpha); g_key = alpha; }
    df_alpha = slope();
    df_key = alpha;
    return df_alpha;
  }
  void eval_fdf(double alpha, double* f, double* df) {
    if (alpha == f_key && alpha == df_key) { *f = f_alpha; *df = df_alpha; return; }
    if (alpha == f_key || alpha == df_key) { *f = eval_f(alpha); *df = eval_df(alpha); return; }
    moveto(alpha);
    f_alpha = cost_grad(*P, x_alpha, g_alpha);
    f_key = alpha; g_key = alpha;
    df_alpha = slope(); df_key = alpha;
    *f = f_alpha; *df = df_alpha;
  }
  void init(const double* x, double* f, double* gradient) {
    delta_f = 0;
    *f = cost_grad(*P, x, gradient);
    for (int i = 0; i < 6; i++) { x0[i] = x[i]; g0[i] = gradient[i]; }
    g0norm = nrm(g0);
    for (int i = 0; i < 6; i++) p[i] = -gradient[i] / g0norm;
    pnorm = nrm(p);
    fp0 = -g0norm;
    for (int i = 0; i < 6; i++) { x_alpha[i] = x0[i]; g_alpha[i] = g0[i]; }
    f_alpha = *f; df_alpha = slope();
    f_key = df_key = x_key = g_key = 0;
  }
  static double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
    double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0));
    double fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
    double c = 2 * (f1 - f0 - fp0);
    double zmin = zl, fmin = fl;
    if (fh < fmin) { zmin = zh; fmin = fh; }
    if (c > 0) {
      double z = -fp0 / c;
      if (z > zl && z < zh) {
        double f = f0 + z * (fp0 + z * (f1 - f0 - fp0));
        if (f < fmin) { zmin = z; fmin = f; }
      }
    }
    return zmin;
  }
  static double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }
  static int solve_quadratic(double a, double b, double c, double* x0, double* x1) {
    if (a == 0) {
      if (b == 0) return 0;
      *x0 = -c / b;
      return 1;
    }
    double disc = b * b - 4 * a * c;
    if (disc > 0) {
      if (b == 0) {
        double r = std::sqrt(-c / a);
        *x0 = -r; *x1 = r;
      } else {
        double sgnb = (b > 0 ? 1 : -1);
        double temp = -0.5 * (b + sgnb * std::sqrt(disc));
        double r1 = temp / a, r2 = c / temp;
        if (r1 < r2) { *x0 = r1; *x1 = r2; } else { *x0 = r2; *x1 = r1; }
      }
      return 2;
    } else if (disc == 0) {
      *x0 = -0.5 * b / a; *x1 = -0.5 * b / a;
      return 2;
    }
    return 0;
  }
  static double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
    double eta = 3 * (f1 - f0) - 2 * fp0 - fp1;
    double xi = fp0 + fp1 - 2 * (f1 - f0);
    double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
    double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl), z0, z1;
    auto check = [&](double z) { double y = cubic(c0, c1, c2, c3, z); if (y < fmin) { zmin = z; fmin = y; } };
    check(zh);
    int n = solve_quadratic(3 * c3, 2 * c2, c1, &z0, &z1);
    if (n == 2) {
      if (z0 > zl && z0 < zh) check(z0);
      if (z1 > zl && z1 < zh) check(z1);
    } else if (n == 1) {
      if (z0 > zl && z0 < zh) check(z0);
    }
    return zmin;
  }
  double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax) {
    double zmin = (xmin - a) / (b - a), zmax = (xmax - a) / (b - a);
    if (zmin > zmax) std::swap(zmin, zmax);
    double z;
    if (order > 2 && std::isfinite(fpb)) z = interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax);
    else z = interp_quad(fa, fpa * (b - a), fb, zmin, zmax);
    return a + z * (b - a);
  }
  // returns 0 success, 1 no progress
  int line_search(double alpha1, double* alpha_new) {
    double f0, fp0l, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
    double alpha = alpha1, alpha_prev = 0.0;
    double a, b, fa, fb, fpa, fpb;
    const int bracket_iters = 100, section_iters = 100;
    int i = 0;
    eval_fdf(0.0, &f0, &fp0l);
    falpha_prev = f0; fpalpha_prev = fp0l;
    a = 0.0; b = alpha; fa = f0; fb = 0.0; fpa = fp0l; fpb = 0.0;
    while (i++ < bracket_iters) {
      falpha = eval_f(alpha);
      if (falpha > f0 + alpha * rho * fp0l || falpha >= falpha_prev) {
        a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
        b = alpha; fb = falpha; fpb = std::numeric_limits<double>::quiet_NaN();
        break;
      }
      fpalpha = eval_df(alpha);
      if (std::fabs(fpalpha) <= -sigma * fp0l) { *alpha_new = alpha; return 0; }
      if (fpalpha >= 0) {
        a = alpha; fa = falpha; fpa = fpalpha;
        b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
        break;
      }
      delta = alpha - alpha_prev;
      alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta);
      alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha; alpha = alpha_next;
    }
    while (i++ < section_iters) {
      delta = b - a;
      alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta);
      falpha = eval_f(alpha);
      if ((a - alpha) * fpa <= 2.220446049250313e-16) return 1;
      if (falpha > f0 + rho * alpha * fp0l || falpha >= fa) {
        b = alpha; fb = falpha; fpb = std::numeric_limits<double>::quiet_NaN();
      } else {
        fpalpha = eval_df(alpha);
        if (std::fabs(fpalpha) <= -sigma * fp0l) { *alpha_new = alpha; return 0; }
        if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
          b = a; fb = fa; fpb = fpa;
          a = alpha; fa = falpha; fpa = fpalpha;
        } else {
          a = alpha; fa = falpha; fpa = fpalpha;
        }
      }
    }
    return 0;
  }
  // 0 success, 1 no progress
  int one_step(double* x, double* f, double* gradient) {
    double alpha = 0.0, alpha1;
    double f0 = *f;
    if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0) return 1;
    if (delta_f < 0) {
      double del = std::max(-delta_f, 10 * 2.220446049250313e-16 * std::fabs(f0));
      alpha1 = std::min(1.0, 2.0 * del / (-fp0));
    } else {
      alpha1 = std::fabs(step);
    }
    int status = line_search(alpha1, &alpha);
    if (status) return status;
    // update_position
    double fa, dfa;
    eval_fdf(alpha, &fa, &dfa);
    *f = fa;
    for (int i = 0; i < 6; i++) { x[i] = x_alpha[i]; gradient[i] = g_alpha[i]; }
    delta_f = *f - f0;
    for (int i = 0; i < 6; i++) { dx0[i] = x[i] - x0[i]; dg0[i] = gradient[i] - g0[i]; }
    double dxg = dot(dx0, gradient), dgg = dot(dg0, gradient), dxdg = dot(dx0, dg0), dgnorm = nrm(dg0), A, B;
    if (dxdg != 0) {
      B = dxg / dxdg;
      A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
    } else {
      B = 0; A = 0;
    }
    for (int i = 0; i < 6; i++) p[i] = gradient[i] - A * dx0[i] - B * dg0[i];
    for (int i = 0; i < 6; i++) { g0[i] = gradient[i]; x0[i] = x[i]; }
    g0norm = nrm(g0);
    pnorm = nrm(p);
    double pg = dot(p, gradient);
    double dir = (pg >= 0.0) ? -1.0 : +1.0;
    for (int i = 0; i < 6; i++) p[i] *= dir / pnorm;
    pnorm = nrm(p);
    fp0 = dot(p, g0);
    // change_direction
    for (int i = 0; i < 6; i++) { x_alpha[i] = x0[i]; g_alpha[i] = g0[i]; }
    x_key = 0; f_alpha = *f; f_key = 0; g_key = 0; df_alpha = slope(); df_key = 0;
    return 0;
  }
};

void mat4_mul_f(const float* A, const float* B, float* Cm) {  // col-major float product
  float T[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      float s = 0;
      for (int k = 0; k < 4; k++) s += A[k * 4 + r] * B[c * 4 + k];
      T[c * 4 + r] = s;
    }
  std::memcpy(Cm, T, sizeof(T));
}

}  // namespace

extern "C" {

// computeCovariances: nn grid is over the cloud itself.
void orc_gicp_covariances(void* nn_grid, const float* pts, size_t stride_f, size_t n, int k, double gicp_eps, double* cov,
                          int num_threads) {
  std::vector<int> idx(n * (size_t)k);
  std::vector<float> d2(n * (size_t)k);
  orc_knn_search(nn_grid, pts, stride_f, n, k, idx.data(), d2.data(), num_threads);
#pragma omp parallel for schedule(static) num_threads(num_threads > 0 ? num_threads : omp_get_max_threads())
  for (long i = 0; i < (long)n; i++) {
    double mean[3] = {0, 0, 0}, c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; j++) {
      const float* p = pts + (size_t)idx[i * k + j] * stride_f;
      mean[0] += p[0]; mean[1] += p[1]; mean[2] += p[2];
      c[0] += (double)(p[0] * p[0]);
      c[3] += (double)(p[1] * p[0]); c[4] += (double)(p[1] * p[1]);
      c[6] += (double)(p[2] * p[0]); c[7] += (double)(p[2] * p[1]); c[8] += (double)(p[2] * p[2]);
    }
    for (int a = 0; a < 3; a++) mean[a] /= (double)k;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b <= a; b++) {
        c[a * 3 + b] /= (double)k;
        c[a * 3 + b] -= mean[a] * mean[b];
        c[b * 3 + a] = c[a * 3 + b];
      }
    if (gicp_eps < 0) {  // inspection: the sample covariance before the regularisation
      for (int a = 0; a < 9; a++) cov[9 * (size_t)i + a] = c[a];
      continue;
    }
    double w[3], V[9];
    orc::sym3_eigen(c, w, V);
    // singular values of a symmetric matrix = |eigenvalues|; descending order of |w|
    int ord[3] = {0, 1, 2};
    std::sort(ord, ord + 3, [&](int a, int b) { return std::fabs(w[a]) > std::fabs(w[b]); });
    double* out = cov + 9 * (size_t)i;
    for (int a = 0; a < 9; a++) out[a] = 0;
    for (int kk = 0; kk < 3; kk++) {
      double v = (kk == 2) ? gicp_eps : 1.0;
      int col = ord[kk];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) out[a * 3 + b] += v * V[a * 3 + col] * V[b * 3 + col];
    }
  }
}

double orc_gicp_cost(const float* src_xyz, const float* tgt_xyz, const double* M, size_t m, const double* x, double* grad,
                     double* unused) {
  std::vector<int> ii(m);
  for (size_t i = 0; i < m; i++) ii[i] = (int)i;
  Problem P{src_xyz, tgt_xyz, 3, ii.data(), ii.data(), M, (int)m, 1};
  return cost_grad(P, x, grad);
}

int orc_gicp_align(void* nn_tgt, const float* tgt, size_t tstride, size_t nt, const double* tgt_cov, void* /*unused*/,
                   const float* src, size_t sstride, size_t ns, const double* src_cov, const float* guess16,
                   const GicpParams* prm, GicpResult* R) {
  const int threads = prm->num_threads > 0 ? prm->num_threads : omp_get_max_threads();
  float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float trans[16], prev[16];
  std::memcpy(trans, I16, sizeof(I16));
  std::memcpy(prev, I16, sizeof(I16));
  std::memset(R, 0, sizeof(*R));
  // output = guess * input
  std::vector<float> out(ns * 3);
  for (size_t i = 0; i < ns; i++) xform(guess16, src + i * sstride, &out[3 * i]);
  std::vector<double> Mv(ns * 9);
  for (size_t i = 0; i < ns; i++) {
    double* M = &Mv[9 * i];
    for (int a = 0; a < 9; a++) M[a] = (a % 4 == 0) ? 1.0 : 0.0;
  }
  const double dist_threshold = prm->max_corr_dist * prm->max_corr_dist;
  int nr_iterations = 0;
  bool converged = false;
  std::vector<int> nn_idx(ns), isrc(ns), itgt(ns);
  std::vector<float> nn_d2(ns);
  double last_cost = 0;
  int last_cnt = 0;
  while (!converged) {
    double TR[16];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += (double)trans[k * 4 + i] * (double)guess16[j * 4 + k];
        TR[j * 4 + i] = s;
      }
    double Rm[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Rm[i * 3 + j] = TR[j * 4 + i];
    orc_nn_search(nn_tgt, out.data(), 3, ns, trans, nn_idx.data(), nn_d2.data(), threads);
    int cnt = 0;
    for (size_t i = 0; i < ns; i++) {
      if (nn_idx[i] < 0 || !((double)nn_d2[i] < dist_threshold)) continue;
      const double* C1 = src_cov + 9 * i;
      const double* C2 = tgt_cov + 9 * (size_t)nn_idx[i];
      double RC[9], tmp[9], Rt[9];
      orc::mat3_mul(Rm, C1, RC);
      orc::mat3_transpose(Rm, Rt);
      orc::mat3_mul(RC, Rt, tmp);
      for (int a = 0; a < 9; a++) tmp[a] += C2[a];
      orc::mat3_inverse(tmp, &Mv[9 * i]);
      isrc[cnt] = (int)i;
      itgt[cnt] = nn_idx[i];
      cnt++;
    }
    last_cnt = cnt;
    std::memcpy(prev, trans, sizeof(prev));
    if (cnt < 4) break;  // NotEnoughPointsException -> caught, loop left with converged_ = false
    Problem P{out.data(), tgt, tstride, isrc.data(), itgt.data(), Mv.data(), cnt, threads};
    double x[6] = {trans[12], trans[13], trans[14], std::atan2((double)trans[6], (double)trans[10]),
                   std::asin(-(double)trans[2]), std::atan2((double)trans[1], (double)trans[0])};
    bool solver_ok = true;
    if (prm->solver == 0) {
      Bfgs bf;
      bf.P = &P;
      double f, grad[6];
      bf.init(x, &f, grad);
      int inner = 0, result = 0;
      const double gradient_tol = 1e-2;
      do {
        inner++;
        result = bf.one_step(x, &f, grad);
        if (result) break;
        result = (Bfgs::nrm(grad) < gradient_tol) ? 2 : 0;  // 2 = Success, 0 = Running
      } while (result == 0 && inner < prm->max_inner_iterations);
      // NoProgress(1) / Success(2) / max iterations -> accept x
      solver_ok = (result == 1 || result == 2 || inner == prm->max_inner_iterations);
      last_cost = f;
    } else {
      double f = 0, g[6], H[36], dx[6];
      for (int inner = 0; inner < prm->max_inner_iterations; inner++) {
        gn_system(P, x, &f, g, H);
        if (Bfgs::nrm(g) < 1e-2) break;
        double neg[6];
        for (int a = 0; a < 6; a++) neg[a] = -g[a];
        orc::svd6_solve(H, neg, dx);
        for (int a = 0; a < 6; a++) x[a] += dx[a];
      }
      last_cost = f;
    }
    if (!solver_ok) break;
    apply_state_f(x, trans);
    double delta = 0;
    for (int k = 0; k < 4; k++)
      for (int l = 0; l < 4; l++) {
        double ratio = (k < 3 && l < 3) ? 1. / prm->rot_eps : 1. / prm->trans_eps;
        double c_delta = ratio * std::fabs((double)prev[l * 4 + k] - (double)trans[l * 4 + k]);
        if (c_delta > delta) delta = c_delta;
      }
    nr_iterations++;
    if (nr_iterations >= prm->max_iterations || delta < 1) {
      converged = true;
      std::memcpy(prev, trans, sizeof(prev));
    }
  }
  mat4_mul_f(prev, guess16, R->final_transformation);
  R->converged = converged ? 1 : 0;
  R->iterations = nr_iterations;
  R->n_correspondences = last_cnt;
  R->final_cost = last_cost;
  return 0;
}

}  // extern "C"
