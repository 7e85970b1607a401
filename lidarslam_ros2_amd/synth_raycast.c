/* Workload generation only (not on the product path): the ray/primitive intersections of lidarslam_ros2_amd/synth.py's
 * raycast_geometry() in C, IEEE double arithmetic operation for operation (build with -ffp-contract=off, no -ffast-math), so
 * that the clouds are BIT-IDENTICAL to the numpy version's (tests/test_host_cpu.py holds the two against each other) and the
 * committed fixtures stay valid.  numpy's minimum / maximum propagate NaN (0 * inf when a ray is parallel to a slab it starts
 * on): so do min_nan / max_nan below. */
#include <math.h>
#include <stddef.h>

static inline double min_nan(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }   /* np.minimum */
static inline double max_nan(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }   /* np.maximum */

/* d: n x 3 ray directions in the map frame (row-major), o: sensor origin, boxes: nb x 6, cyls: nc x 4 {cx, cy, r, ztop};
 * t_best: n doubles out (inf = no hit) */
void synth_raycast(const double* d, long n, const double* o, double ground_z, const double* boxes, int nb, const double* cyls, int nc,
                   double* t_best) {
  for (long i = 0; i < n; i++) {
    const double dx = d[3 * i], dy = d[3 * i + 1], dz = d[3 * i + 2];
    double best = INFINITY;
    /* ground: tg = (gz - oz) / dz; tg[~(tg > 0)] = inf */
    double tg = (ground_z - o[2]) / dz;
    if (!(tg > 0)) tg = INFINITY;
    best = min_nan(best, tg);
    /* boxes (slab method) */
    const double inv[3] = {1.0 / dx, 1.0 / dy, 1.0 / dz};
    for (int b = 0; b < nb; b++) {
      const double* B = boxes + 6 * b;
      double tn = 0, tf = 0;
      for (int k = 0; k < 3; k++) {
        const double t0 = (B[k] - o[k]) * inv[k], t1 = (B[3 + k] - o[k]) * inv[k];
        const double lo = min_nan(t0, t1), hi = max_nan(t0, t1);
        tn = (k == 0) ? lo : max_nan(tn, lo);     /* np.minimum(t0, t1).max(axis=1) */
        tf = (k == 0) ? hi : min_nan(tf, hi);     /* np.maximum(t0, t1).min(axis=1) */
      }
      const int hit = (tf >= tn) && (tf > 0);
      double tt = (tn > 0) ? tn : tf;
      if (!hit) tt = INFINITY;
      best = min_nan(best, tt);
    }
    /* cylinders (side surface only) */
    const double a = dx * dx + dy * dy;
    for (int c = 0; c < nc; c++) {
      const double* C = cyls + 4 * c;
      const double ox = o[0] - C[0], oy = o[1] - C[1];
      const double bq = ox * dx + oy * dy;
      const double cq = ox * ox + oy * oy - C[2] * C[2];
      const double disc = bq * bq - a * cq;
      int ok = disc > 0;
      const double sq = sqrt(ok ? disc : 0.0);
      double tt = (-bq - sq) / a;
      const double z = o[2] + tt * dz;
      ok = ok && (tt > 0) && (z >= ground_z) && (z <= C[3]);
      if (!ok) tt = INFINITY;
      best = min_nan(best, tt);
    }
    t_best[i] = best;
  }
}
