// The per-pair and per-point arithmetic of the NDT derivative pass (K3), shared by every kernel variant in ndt.hip and —
// compiled for the host — by tools/ndt_host_emu (tests/test_ndt_host_emu_cpu.py): the SAME fp32 operation order on the
// CPU, so that what separates a GPU registration from the CPU oracle's can be measured without a GPU.
// Restates pclomp::NormalDistributionsTransform::computeDerivatives / updateDerivatives (called through align() at
// scanmatcher/src/scanmatcher_component.cpp:353 and graph_based_slam/src/graph_based_slam_component.cpp:230;
// SURVEY.md §9.4-9.5) in the factorised form of DESIGN.md §4.
#pragma once
#ifndef LSR_HOST_EMU
#include <hip/hip_runtime.h>
#endif

namespace lsr {

// fp32 rotation block of (Translation * Rx * Ry * Rz) from the six float sines / cosines, the way the reference composes
// Eigen::Affine3f from the float-cast pose vector: A = Rx * Ry, R = A * Rz, each entry a two-term dot product.  Written
// with explicit fmaf so that the device compiler's contraction choices and the host emulation agree bit for bit.
__host__ __device__ inline void compose_R12(const float cx, const float cy, const float cz, const float sx, const float sy, const float sz,
                                            float* T) {
#pragma clang fp contract(off)
  const float a00 = cy, a02 = sy;
  const float a10 = sx * sy, a11 = cx, a12 = -sx * cy;
  const float a20 = -cx * sy, a21 = sx, a22 = cx * cy;
  T[0] = a00 * cz; T[1] = -a00 * sz; T[2] = a02;
  T[4] = fmaf(a10, cz, a11 * sz); T[5] = fmaf(-a10, sz, a11 * cz); T[6] = a12;
  T[8] = fmaf(a20, cz, a21 * sz); T[9] = fmaf(-a20, sz, a21 * cz); T[10] = a22;
}

// T12 row-major 3x4 from the pose vector (host side: the diagnostic entry point; the device builds it in build_request).
__host__ __device__ inline void pose_to_T12(const double* p, float* T) {
  const float ax = (float)p[3], ay = (float)p[4], az = (float)p[5];
  compose_R12(cosf(ax), cosf(ay), cosf(az), sinf(ax), sinf(ay), sinf(az), T);
  T[3] = (float)p[0]; T[7] = (float)p[1]; T[11] = (float)p[2];
}

__host__ __device__ inline void T12_to_colmajor16(const float* T, float* M) {
  M[0] = T[0]; M[1] = T[4]; M[2] = T[8];  M[3] = 0.f;
  M[4] = T[1]; M[5] = T[5]; M[6] = T[9];  M[7] = 0.f;
  M[8] = T[2]; M[9] = T[6]; M[10] = T[10]; M[11] = 0.f;
  M[12] = T[3]; M[13] = T[7]; M[14] = T[11]; M[15] = 1.f;
}

// Every function of this header opens with `#pragma clang fp contract(off)`: hipcc's default (-ffp-contract=fast-honor-pragmas)
// fuses a product with the addition that consumes it wherever its scheduler likes — differently in two kernels that inline the
// same function (round 3's ISA: xform_ref came out as one fma in one row and two in another) — so the only fused operations
// here are the explicit fmaf() calls, and a point's 29 terms are the same bits in every kernel variant and in the host emulation.
//
// Point transform in the REFERENCE's rounding order — pcl::transformPointCloud: ((m00 x + m01 y) + m02 z) + m03, every product
// and sum rounded to fp32, no fused multiply-add.  The order matters more than it looks: a coordinate of ~50 m carries an
// fp32 rounding error of ~2e-6 m, q = x' - mean is ~1 m, and the gradient of a pass moves by 3e-7 (relative) between this
// order and an fmaf chain — enough to send a registration that walks thirty clamped 0.1 m steps along an ill-conditioned
// Newton direction (BASELINE cfg 4, candidate 21) 2.4 mm away from the reference's result (tests/test_ndt_host_emu_cpu.py).
__device__ __forceinline__ float xform_ref(const float a, const float b, const float c, const float d, const float x, const float y,
                                           const float z) {
#pragma clang fp contract(off)
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, x), __fmul_rn(b, y)), __fmul_rn(c, z)), d);
}

// expf of the device library (OCML) without its two range guards: x * log2(e) split into a rounded product and its error,
// the integer part by rndne, 2^fraction by v_exp_f32, the integer part back by ldexp — operation for operation what expf()
// compiles to, so every result that is not a flushed extreme is the same bit; the guards (x < -103.3 -> 0, x > 88.7 -> inf,
// four instructions per pair, seven pairs per point) only pre-empt what ldexp does anyway: an exponent below -149 gives 0, one
// above 127 inf, NaN stays NaN.  The argument here is -d2 q^T C q / 2 <= 0.
__device__ __forceinline__ float exp_pair(const float x) {
#ifdef LSR_HOST_EMU
  return expf(x);
#else
#pragma clang fp contract(off)
  const float t = x * 0x1.715476p+0f;
  const float r = rintf(t);
  float lo = fmaf(x, 0x1.715476p+0f, -t);
  lo = fmaf(x, 0x1.4ae0bep-26f, lo);
  const float f = (t - r) + lo;
  return ldexpf(__builtin_amdgcn_exp2f(f), (int)r);
#endif
}

// One (point, voxel) pair of eq. 6.9-6.13 in the factorised form: the point Jacobian and second derivatives
// do not depend on the voxel, so a pair only adds to A = sum w C q and E = sum w (C - d2 Cq Cq^T); fp32 per pair with
// ndt_omp's precision recipe: the weight is scaled by the DOUBLE gauss_d1 and rounded back to float.
// Leaf record (48 bytes): r0 = {mean_hi.xyz, c00}, r1 = {c01, c02, c11, c12}, r2 = {c22, mean_lo.xyz}.  The reference subtracts
// the DOUBLE voxel mean from the float point and rounds once, q = (float)((double)x' - mean): the mean travels as an fp32
// head + tail pair (mean_hi = (float)mean, mean_lo = (float)(mean - mean_hi)) and q = (x' - mean_hi) - mean_lo, which equals the
// reference's q to the last bit except for double roundings (a mean rounded to fp32 alone biases every point of a voxel by the
// same ~2e-6 m).  A record of an unusable leaf (no points, n < 6, invalid covariance) is all NaN: q, e and w0 become NaN and
// the range test below drops the pair — no count or flag is read in the hot loop.
// KDTREE neighbourhood (ndt_omp: target_cells_.radiusSearch(x_trans_pt, resolution_, ...) on the kd-tree over the leaves' float
// centroids): FLANN's L2_Simple<float> between the transformed point and a centroid — dx*dx, + dy*dy, + dz*dz, no contraction —
// strictly below the squared radius (RadiusResultSet::addPoint).
__device__ __forceinline__ bool centroid_in_radius(const float tx, const float ty, const float tz, const float cx, const float cy,
                                                   const float cz, const float radius2) {
#pragma clang fp contract(off)
  const float dx = tx - cx, dy = ty - cy, dz = tz - cz;
  float d = dx * dx;
  d = d + dy * dy;
  d = d + dz * dz;
  return d < radius2;
}

__device__ __forceinline__ void pair_terms(const bool leaf_ok, const bool hess, const float tx, const float ty, const float tz,
                                           const float4 r0, const float4 r1, const float4 r2, const float d2, const double d1d,
                                           float& score, float& npairs, float& A0, float& A1, float& A2, float& E00, float& E01,
                                           float& E02, float& E11, float& E12, float& E22) {
#pragma clang fp contract(off)
  const float q0 = (tx - r0.x) - r2.y, q1 = (ty - r0.y) - r2.z, q2 = (tz - r0.z) - r2.w;
  const float c00 = r0.w, c01 = r1.x, c02 = r1.y, c11 = r1.z, c12 = r1.w, c22 = r2.x;
  const float Cq0 = fmaf(c00, q0, fmaf(c01, q1, c02 * q2));
  const float Cq1 = fmaf(c01, q0, fmaf(c11, q1, c12 * q2));
  const float Cq2 = fmaf(c02, q0, fmaf(c12, q1, c22 * q2));
  const float qCq = fmaf(q0, Cq0, fmaf(q1, Cq1, q2 * Cq2));
  const float e = exp_pair(-d2 * qCq * 0.5f);
  const float w0 = d2 * e;
  // ndt_omp drops the whole pair (score included) when d2*e is outside [0,1] or NaN
  const bool ok = leaf_ok & (w0 <= 1.f) & (w0 >= 0.f);
  score += ok ? (float)(-d1d * (double)e) : 0.f;
  npairs += ok ? 1.f : 0.f;
  const float w = (float)((double)w0 * d1d);
  A0 = ok ? fmaf(w, Cq0, A0) : A0;
  A1 = ok ? fmaf(w, Cq1, A1) : A1;
  A2 = ok ? fmaf(w, Cq2, A2) : A2;
  if (hess) {
    const float wd = -w * d2;
    E00 = ok ? E00 + fmaf(wd * Cq0, Cq0, w * c00) : E00;
    E01 = ok ? E01 + fmaf(wd * Cq0, Cq1, w * c01) : E01;
    E02 = ok ? E02 + fmaf(wd * Cq0, Cq2, w * c02) : E02;
    E11 = ok ? E11 + fmaf(wd * Cq1, Cq1, w * c11) : E11;
    E12 = ok ? E12 + fmaf(wd * Cq1, Cq2, w * c12) : E12;
    E22 = ok ? E22 + fmaf(wd * Cq2, Cq2, w * c22) : E22;
  }
}

// The 29 per-point terms (score, 3 + 3 gradient, #pairs, 21 Hessian upper triangle) from the point's A / E sums, the point
// Jacobian J = [I | J3 J4 J5] (eq. 6.18/6.19) and the second-derivative vectors (eq. 6.20/6.21) of the UNTRANSFORMED point.
// o[8..28] are only written when hess.  ja / ha: the 24 / 48 angle coefficients (jang / hang of NdtState; LDS pointers on the device).
template <typename JP, typename HP>
__device__ __forceinline__ void point_terms(const bool hess, const float px, const float py, const float pz, const float score,
                                            const float npairs, const float A0, const float A1, const float A2, const float E00,
                                            const float E01, const float E02, const float E11, const float E12, const float E22,
                                            JP ja, HP ha, float* __restrict__ o) {
#pragma clang fp contract(off)
  const float j_a = fmaf(ja[0], px, fmaf(ja[1], py, ja[2] * pz));
  const float j_b = fmaf(ja[3], px, fmaf(ja[4], py, ja[5] * pz));
  const float j_c = fmaf(ja[6], px, fmaf(ja[7], py, ja[8] * pz));
  const float j_d = fmaf(ja[9], px, fmaf(ja[10], py, ja[11] * pz));
  const float j_e = fmaf(ja[12], px, fmaf(ja[13], py, ja[14] * pz));
  const float j_f = fmaf(ja[15], px, ja[16] * py);
  const float j_g = fmaf(ja[18], px, ja[19] * py);
  const float j_h = fmaf(ja[21], px, ja[22] * py);
  // J3 = (0, a, b), J4 = (c, d, e), J5 = (f, g, h)
  o[0] = score;
  o[1] = A0;
  o[2] = A1;
  o[3] = A2;
  o[4] = fmaf(A1, j_a, A2 * j_b);
  o[5] = fmaf(A0, j_c, fmaf(A1, j_d, A2 * j_e));
  o[6] = fmaf(A0, j_f, fmaf(A1, j_g, A2 * j_h));
  o[7] = npairs;
  if (hess) {
    // E J_k for k = 3,4,5
    const float e3x = fmaf(E01, j_a, E02 * j_b), e3y = fmaf(E11, j_a, E12 * j_b), e3z = fmaf(E12, j_a, E22 * j_b);
    const float e4x = fmaf(E00, j_c, fmaf(E01, j_d, E02 * j_e)), e4y = fmaf(E01, j_c, fmaf(E11, j_d, E12 * j_e)),
                e4z = fmaf(E02, j_c, fmaf(E12, j_d, E22 * j_e));
    const float e5x = fmaf(E00, j_f, fmaf(E01, j_g, E02 * j_h)), e5y = fmaf(E01, j_f, fmaf(E11, j_g, E12 * j_h)),
                e5z = fmaf(E02, j_f, fmaf(E12, j_g, E22 * j_h));
    // second-derivative vectors (eq. 6.20/6.21) dotted with A = sum w C q
    const float ha2 = fmaf(ha[0], px, fmaf(ha[1], py, ha[2] * pz)), ha3 = fmaf(ha[3], px, fmaf(ha[4], py, ha[5] * pz));
    const float hb2 = fmaf(ha[6], px, fmaf(ha[7], py, ha[8] * pz)), hb3 = fmaf(ha[9], px, fmaf(ha[10], py, ha[11] * pz));
    const float hc2 = fmaf(ha[12], px, ha[13] * py), hc3 = fmaf(ha[15], px, ha[16] * py);
    const float hd1 = fmaf(ha[18], px, fmaf(ha[19], py, ha[20] * pz)), hd2 = fmaf(ha[21], px, fmaf(ha[22], py, ha[23] * pz)),
                hd3 = fmaf(ha[24], px, fmaf(ha[25], py, ha[26] * pz));
    const float he1 = fmaf(ha[27], px, ha[28] * py), he2 = fmaf(ha[30], px, ha[31] * py), he3 = fmaf(ha[33], px, ha[34] * py);
    const float hf1 = fmaf(ha[36], px, ha[37] * py), hf2 = fmaf(ha[39], px, ha[40] * py), hf3 = fmaf(ha[42], px, ha[43] * py);
    // upper triangle, row-major: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
    o[8] = E00;  o[9] = E01;  o[10] = E02;
    o[11] = e3x; o[12] = e4x; o[13] = e5x;
    o[14] = E11; o[15] = E12;
    o[16] = e3y; o[17] = e4y; o[18] = e5y;
    o[19] = E22;
    o[20] = e3z; o[21] = e4z; o[22] = e5z;
    o[23] = fmaf(j_a, e3y, j_b * e3z) + fmaf(A1, ha2, A2 * ha3);                       // (3,3)
    o[24] = fmaf(j_a, e4y, j_b * e4z) + fmaf(A1, hb2, A2 * hb3);                       // (3,4)
    o[25] = fmaf(j_a, e5y, j_b * e5z) + fmaf(A1, hc2, A2 * hc3);                       // (3,5)
    o[26] = fmaf(j_c, e4x, fmaf(j_d, e4y, j_e * e4z)) + fmaf(A0, hd1, fmaf(A1, hd2, A2 * hd3));  // (4,4)
    o[27] = fmaf(j_c, e5x, fmaf(j_d, e5y, j_e * e5z)) + fmaf(A0, he1, fmaf(A1, he2, A2 * he3));  // (4,5)
    o[28] = fmaf(j_f, e5x, fmaf(j_g, e5y, j_h * e5z)) + fmaf(A0, hf1, fmaf(A1, hf2, A2 * hf3));  // (5,5)
  }
}

}  // namespace lsr
