// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
// under oracle/.  PARITY UNPINNED: the reference's NDT/GICP arithmetic lives in the
// un-vendored submodule Thirdparty/ndt_omp_ros2 (/root/reference/.gitmodules:1-4) + PCL
// 1.12 + Eigen 3.4, none of which are present; the reference ships no tests or golden
// vectors.  These helpers restate the small dense-linear-algebra
// routines those libraries supply (Eigen::SelfAdjointEigenSolver<Matrix3d>,
// Matrix3d::inverse, JacobiSVD<Matrix<double,6,6>>::solve, JacobiSVD<Matrix3d>).
#pragma once
#include <cmath>
#include <cstring>
#include <algorithm>

namespace orc {

// ---- 3x3 helpers (row-major double[9]) -------------------------------------------------
inline void mat3_mul(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      T[i * 3 + j] = s;
    }
  std::memcpy(C, T, sizeof(T));
}
inline void mat3_transpose(const double* A, double* At) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[j * 3 + i] = A[i * 3 + j];
  std::memcpy(At, T, sizeof(T));
}
// General 3x3 inverse by cofactors (what Eigen's fixed-size inverse() does for 3x3).
inline bool mat3_inverse(const double* A, double* Ai) {
  double c00 = A[4] * A[8] - A[5] * A[7];
  double c01 = A[5] * A[6] - A[3] * A[8];
  double c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  double id = 1.0 / det;
  double T[9];
  T[0] = c00 * id;
  T[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  T[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  T[3] = c01 * id;
  T[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  T[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  T[6] = c02 * id;
  T[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  T[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  std::memcpy(Ai, T, sizeof(T));
  return det != 0.0 && std::isfinite(id);
}

// Symmetric 3x3 eigen-decomposition (cyclic Jacobi, reads the LOWER triangle like
// Eigen::SelfAdjointEigenSolver).  Eigenvalues ascending in w[0..2]; eigenvectors are the
// COLUMNS of V (row-major V[i*3+k] = component i of eigenvector k).
inline void sym3_eigen(const double* Ain, double* w, double* V) {
  double A[9];
  A[0] = Ain[0]; A[4] = Ain[4]; A[8] = Ain[8];
  A[1] = A[3] = Ain[3];
  A[2] = A[6] = Ain[6];
  A[5] = A[7] = Ain[7];
  double Q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        double app = A[p * 3 + p], aqq = A[q * 3 + q];
        double theta = (aqq - app) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        // A <- J^T A J
        for (int k = 0; k < 3; k++) {
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double qkp = Q[k * 3 + p], qkq = Q[k * 3 + q];
          Q[k * 3 + p] = c * qkp - s * qkq;
          Q[k * 3 + q] = s * qkp + c * qkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  double d[3] = {A[0], A[4], A[8]};
  std::sort(idx, idx + 3, [&](int a, int b) { return d[a] < d[b]; });
  for (int k = 0; k < 3; k++) {
    w[k] = d[idx[k]];
    for (int i = 0; i < 3; i++) V[i * 3 + k] = Q[i * 3 + idx[k]];
  }
}

// 6x6 least-squares solve via one-sided Jacobi SVD: x = V * S^+ * U^T * b, singular values
// <= (6*eps)*s_max treated as zero — the contract of
// Eigen::JacobiSVD<Matrix<double,6,6>>(H, ComputeFullU|ComputeFullV).solve(b).
inline void svd6_solve(const double* H /*row-major 6x6*/, const double* b, double* x) {
  const int n = 6;
  double U[36], V[36];
  std::memcpy(U, H, sizeof(U));
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < n; k++) {
          alpha += U[k * n + p] * U[k * n + p];
          beta += U[k * n + q] * U[k * n + q];
          gamma += U[k * n + p] * U[k * n + q];
        }
        if (gamma == 0.0) continue;
        if (std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < n; k++) {
          double up = U[k * n + p], uq = U[k * n + q];
          U[k * n + p] = c * up - s * uq;
          U[k * n + q] = s * up + c * uq;
          double vp = V[k * n + p], vq = V[k * n + q];
          V[k * n + p] = c * vp - s * vq;
          V[k * n + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sv[6], smax = 0;
  for (int j = 0; j < n; j++) {
    double s = 0;
    for (int k = 0; k < n; k++) s += U[k * n + j] * U[k * n + j];
    sv[j] = std::sqrt(s);
    smax = std::max(smax, sv[j]);
  }
  const double thr = 6.0 * 2.220446049250313e-16 * smax;
  double y[6];
  for (int j = 0; j < n; j++) {
    if (sv[j] > thr && sv[j] > 0) {
      double d = 0;
      for (int k = 0; k < n; k++) d += U[k * n + j] * b[k];  // (u_j * s_j)^T b
      y[j] = d / (sv[j] * sv[j]);                            // = (u_j^T b) / s_j
    } else {
      y[j] = 0;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int j = 0; j < n; j++) s += V[i * n + j] * y[j];
    x[i] = s;
  }
}

}  // namespace orc
