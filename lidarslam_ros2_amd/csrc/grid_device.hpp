// Device-side leaf mathematics shared by the two voxel-covariance grid builders (sort-based: ndt.hip, counting-sort
// for dense key spaces: grid_dense.hip): Jacobi eigen-decomposition, cofactor inverse, and the finalisation of one
// leaf exactly as pclomp::VoxelGridCovariance does it.
#pragma once
#include <hip/hip_runtime.h>

namespace lsr {

// Symmetric 3x3 eigen-decomposition (cyclic Jacobi), eigenvalues ascending, eigenvectors in columns.
__device__ inline void sym3_eigen_dev(const double* Ain, double* w, double* V) {
  double a00 = Ain[0], a01 = Ain[1], a02 = Ain[2], a11 = Ain[4], a12 = Ain[5], a22 = Ain[8];
  double q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 32; sweep++) {
    double off = a01 * a01 + a02 * a02 + a12 * a12;
    double diag = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
#define LSR_JACOBI(app, aqq, apq, arp, arq, cp, cq)                                   \
  if (apq != 0.0) {                                                                   \
    double theta = (aqq - app) / (2.0 * apq);                                         \
    double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;                                    \
    double npp = app - t * apq, nqq = aqq + t * apq;                                  \
    double nrp = c * arp - s * arq, nrq = s * arp + c * arq;                          \
    app = npp; aqq = nqq; apq = 0.0; arp = nrp; arq = nrq;                            \
    for (int k = 0; k < 3; k++) {                                                     \
      double qp = q[k * 3 + cp], qq = q[k * 3 + cq];                                  \
      q[k * 3 + cp] = c * qp - s * qq;                                                \
      q[k * 3 + cq] = s * qp + c * qq;                                                \
    }                                                                                 \
  }
    LSR_JACOBI(a00, a11, a01, a02, a12, 0, 1)
    LSR_JACOBI(a00, a22, a02, a01, a12, 0, 2)
    LSR_JACOBI(a11, a22, a12, a01, a02, 1, 2)
#undef LSR_JACOBI
  }
  double d[3] = {a00, a11, a22};
  int i0 = 0, i1 = 1, i2 = 2;
  if (d[i0] > d[i1]) { int t = i0; i0 = i1; i1 = t; }
  if (d[i1] > d[i2]) { int t = i1; i1 = i2; i2 = t; }
  if (d[i0] > d[i1]) { int t = i0; i0 = i1; i1 = t; }
  int idx[3] = {i0, i1, i2};
  for (int k = 0; k < 3; k++) {
    w[k] = d[idx[k]];
    for (int i = 0; i < 3; i++) V[i * 3 + k] = q[i * 3 + idx[k]];
  }
}

__device__ inline bool sym3_inverse_dev(const double* A, double* Ai) {
  double c00 = A[4] * A[8] - A[5] * A[7];
  double c01 = A[5] * A[6] - A[3] * A[8];
  double c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  double id = 1.0 / det;
  Ai[0] = c00 * id;
  Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  Ai[3] = c01 * id;
  Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  Ai[6] = c02 * id;
  Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  bool ok = true;
  for (int k = 0; k < 9; k++) ok = ok && isfinite(Ai[k]);
  return ok;
}

// K2 for ONE leaf: mean, single-pass covariance, (n-1)/n, eigenvalue clamp, inverse.  s = {Sx,Sy,Sz,Sxx,Sxy,Sxz,Syy,Syz,Szz}.
// Returns the leaf's point count as the lookups see it (n, or -1 when the covariance is unusable); icov is zeroed
// when the leaf is not usable.
__device__ inline int leaf_finalize_dev(const double* s, int n, int min_points, double eig_mult, double* mean, double* icov, bool* usable) {
  const double nn = (double)n;
  mean[0] = s[0] / nn; mean[1] = s[1] / nn; mean[2] = s[2] / nn;
  for (int k = 0; k < 9; k++) icov[k] = 0.0;
  bool valid = false;
  if (n >= min_points) {
    const double sq[9] = {s[3], s[4], s[5], s[4], s[6], s[7], s[5], s[7], s[8]};
    double cov[9];
    const double f = (nn - 1.0) / nn;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b <= a; b++) {
        double v = ((sq[a * 3 + b] - 2.0 * (s[a] * mean[b])) / nn + mean[a] * mean[b]) * f;
        cov[a * 3 + b] = v;
        cov[b * 3 + a] = v;
      }
    double w[3], V[9];
    sym3_eigen_dev(cov, w, V);
    if (!(w[0] < 0 || w[1] < 0 || w[2] <= 0)) {
      const double lmin = eig_mult * w[2];
      if (w[0] < lmin) {
        w[0] = lmin;
        if (w[1] < lmin) w[1] = lmin;
        // cov = V diag(w) V^T  (V orthonormal: V^-1 = V^T)
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++)
            cov[a * 3 + b] = V[a * 3 + 0] * w[0] * V[b * 3 + 0] + V[a * 3 + 1] * w[1] * V[b * 3 + 1] +
                             V[a * 3 + 2] * w[2] * V[b * 3 + 2];
      }
      valid = sym3_inverse_dev(cov, icov);
      if (!valid) for (int k = 0; k < 9; k++) icov[k] = 0.0;
    }
    if (!valid) n = -1;
  }
  *usable = valid;
  return n;
}

// The 64-byte leaf record the derivative pass reads (ndt_point.hpp: pair_terms): {mean_hi.xyz, c00 | c01 c02 c11 c12 |
// c22, mean_lo.xyz | n, 0, 0, 0}.  mean = mean_hi + mean_lo as an fp32 head + tail pair: the reference subtracts the DOUBLE mean.
// A leaf lookups cannot use (n < min_points, invalid covariance) carries NaN in the three 16-byte pieces the pass reads — the
// pair then drops itself through ndt_omp's own range test — and its point count in the fourth.
__device__ inline void leaf_record_dev(const double* mean, const double* icov, int n, bool usable, float4* __restrict__ rec) {
  if (usable) {
    const float hx = (float)mean[0], hy = (float)mean[1], hz = (float)mean[2];
    rec[0] = make_float4(hx, hy, hz, (float)icov[0]);
    rec[1] = make_float4((float)icov[1], (float)icov[2], (float)icov[4], (float)icov[5]);
    rec[2] = make_float4((float)icov[8], (float)(mean[0] - (double)hx), (float)(mean[1] - (double)hy), (float)(mean[2] - (double)hz));
  } else {
    const float qnan = __int_as_float(0x7FC00000);
    rec[0] = rec[1] = rec[2] = make_float4(qnan, qnan, qnan, qnan);
  }
  rec[3] = make_float4((float)n, 0.f, 0.f, 0.f);
}

}  // namespace lsr
