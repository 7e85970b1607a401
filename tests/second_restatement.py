"""A SECOND, independent restatement of the two optimisation loops of the hot path, in fp64 numpy / scipy — the only pin
available offline for the part of the oracle that no reference vector covers (the C++ oracle and the
GPU controller were written from the same notes by the same hand).  Written from the published algorithms, not from
oracle/*.cpp:

  * NDT: Magnusson 2009 eqs 6.9-6.13 (score, gradient, Hessian of the Gaussian mixture), 6.17-6.21 (pose Jacobian and
    Hessian of T(p) = Trans * Rx * Ry * Rz), Newton's method with the line search of More & Thuente 1994 (the
    trial-value cases 1-4 of section 4 and the interval update of section 2 — on psi(a) = phi(a) - phi(0) - mu phi'(0) a
    while the interval is "open", on phi afterwards), wired as PCL's computeTransformation / computeStepLengthMT do it
    (step clamp [eps/2, step_size], max 10 trials, Hessian recomputed after trials with the h_ang
    of the FIRST pass of the line search, `iter > max_iter` stop rule).
  * GICP: Segal 2009: plane-to-plane covariances (20-NN sample covariance, SVD, singular values -> (1, 1, eps)),
    correspondences by 1-NN inside max_correspondence_distance, Mahalanobis M_i = (C2_j + R C1_i R^T)^-1, the inner
    problem min_x 1/m sum r^T M r solved by scipy's BFGS, PCL's outer stop rule (delta < 1).

Everything is vectorised fp64 except the voxel lookup, which uses the same fp32 floor(x'/leaf) as the reference so that
points on a cell face land in the same voxel.  Test infrastructure only."""
import numpy as np


# ===================================================================================================================
# NDT
# ===================================================================================================================
def gauss_fit(resolution, outlier_ratio=0.55):
    """Magnusson eq. 6.8"""
    c1 = 10.0 * (1.0 - outlier_ratio)
    c2 = outlier_ratio / resolution ** 3
    d3 = -np.log(c2)
    d1 = -np.log(c1 + c2) - d3
    d2 = -2.0 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / d1)
    return d1, d2


def _sc(a, snap):
    if snap and abs(a) < 10e-5:
        return 0.0, 1.0
    return np.sin(a), np.cos(a)


def rot_xyz(rx, ry, rz):
    sx, cx = np.sin(rx), np.cos(rx)
    sy, cy = np.sin(ry), np.cos(ry)
    sz, cz = np.sin(rz), np.cos(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def angle_jacobian(p):
    """d(R x)/d(phi): three 3x3 matrices dR/dphi_k of R = Rx Ry Rz, by differentiating the factors (eq. 6.19 in matrix
    form), with PCL's small-angle snap (|phi| < 1e-4 -> sin 0, cos 1)."""
    sx, cx = _sc(p[3], True)
    sy, cy = _sc(p[4], True)
    sz, cz = _sc(p[5], True)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    dRx = np.array([[0, 0, 0], [0, -sx, -cx], [0, cx, -sx]])
    dRy = np.array([[-sy, 0, cy], [0, 0, 0], [-cy, 0, -sy]])
    dRz = np.array([[-sz, -cz, 0], [cz, -sz, 0], [0, 0, 0]])
    return [dRx @ Ry @ Rz, Rx @ dRy @ Rz, Rx @ Ry @ dRz]


def angle_hessian(p, d1_sign=+1):
    """d2(R x)/(dphi_i dphi_j) as 3x3 matrices (eq. 6.21).  `d1_sign = +1` reproduces the long-standing PCL quirk in the
    (phi_y, phi_y) block: the x-row's z coefficient is +sy where the analytic second derivative has -sy."""
    sx, cx = _sc(p[3], True)
    sy, cy = _sc(p[4], True)
    sz, cz = _sc(p[5], True)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    dRx = np.array([[0, 0, 0], [0, -sx, -cx], [0, cx, -sx]])
    dRy = np.array([[-sy, 0, cy], [0, 0, 0], [-cy, 0, -sy]])
    dRz = np.array([[-sz, -cz, 0], [cz, -sz, 0], [0, 0, 0]])
    ddRx = np.array([[0, 0, 0], [0, -cx, sx], [0, -sx, -cx]])
    ddRy = np.array([[-cy, 0, -sy], [0, 0, 0], [sy, 0, -cy]])
    ddRz = np.array([[-cz, sz, 0], [-sz, -cz, 0], [0, 0, 0]])
    H = [[None] * 3 for _ in range(3)]
    H[0][0] = ddRx @ Ry @ Rz
    H[0][1] = H[1][0] = dRx @ dRy @ Rz
    H[0][2] = H[2][0] = dRx @ Ry @ dRz
    H[1][1] = Rx @ ddRy @ Rz
    H[1][2] = H[2][1] = Rx @ dRy @ dRz
    H[2][2] = Rx @ Ry @ ddRz
    if d1_sign >= 0:
        H[1][1] = H[1][1].copy()
        H[1][1][0, 2] = sy      # analytic value: -sy (first row of Rx ddRy Rz is (-cy cz, cy sz, -sy))
    return H


class VoxelTable:
    """Dense arrays over the grid from an oracle / GPU grid dump (idx, n, mean, icov)."""

    def __init__(self, dump, min_b, max_b, leaf):
        self.leaf = np.float32(leaf)
        self.min_b, self.max_b = np.asarray(min_b, np.int64), np.asarray(max_b, np.int64)
        div = self.max_b - self.min_b + 1
        self.mul = np.array([1, div[0], div[0] * div[1]], np.int64)
        ncell = int(div.prod())
        self.valid = np.zeros(ncell, bool)
        self.mean = np.zeros((ncell, 3))
        self.icov = np.zeros((ncell, 3, 3))
        ok = dump["n"] >= 6
        self.valid[dump["idx"][ok]] = True
        self.mean[dump["idx"][ok]] = dump["mean"][ok]
        self.icov[dump["idx"][ok]] = dump["icov"][ok]
        self.off = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.int64)  # DIRECT7


def ndt_derivatives(tab, src, p, d1, d2, hang_pose=None, d1_sign=+1, with_hessian=True):
    """score, gradient (6), Hessian (6x6) of eq. 6.9-6.13 summed over every (point, DIRECT7 voxel) pair.
    hang_pose: the pose whose second-derivative matrices are used (computeHessian after a line search re-uses the ones of
    the first pass of that line search); None = p."""
    p = np.asarray(p, np.float64)
    x = src.astype(np.float64)
    R = rot_xyz(*p[3:])
    xt = x @ R.T + p[:3]
    ijk = np.floor(xt.astype(np.float32) / tab.leaf).astype(np.int64)                      # (N,3) fp32 lookup
    cell = ijk[:, None, :] + tab.off[None, :, :]                                           # (N,7,3)
    inb = np.all((cell >= tab.min_b) & (cell <= tab.max_b), axis=2)
    lin = ((cell - tab.min_b) * tab.mul).sum(axis=2)
    lin = np.where(inb, lin, 0)
    ok = inb & tab.valid[lin]
    q = xt[:, None, :] - tab.mean[lin]                                                     # (N,7,3)
    C = tab.icov[lin]                                                                      # (N,7,3,3)
    Cq = np.einsum("nkab,nkb->nka", C, q)
    e = np.exp(-d2 * np.einsum("nka,nka->nk", q, Cq) / 2.0)
    w = d2 * e
    ok &= (w >= 0) & (w <= 1)                                                              # ndt_omp drops the whole pair otherwise
    e = np.where(ok, e, 0.0)
    w = np.where(ok, w, 0.0)
    score = float((-d1 * e).sum())
    # point Jacobian J (N,3,6) = [I | dR_k x]
    dR = angle_jacobian(p)
    J = np.zeros((x.shape[0], 3, 6))
    J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1.0
    for k in range(3):
        J[:, :, 3 + k] = x @ dR[k].T
    qCJ = np.einsum("nka,nab->nkb", Cq, J)                                                 # (N,7,6)
    g = d1 * np.einsum("nk,nkb->b", w, qCJ)
    if not with_hessian:
        return score, g, None
    ddR = angle_hessian(p if hang_pose is None else hang_pose, d1_sign)
    H = -d2 * np.einsum("nk,nki,nkj->ij", w, qCJ, qCJ)
    CJ = np.einsum("nkab,nbj->nkaj", C, J)                                                 # (N,7,3,6)
    H += np.einsum("nk,nai,nkaj->ij", w, J, CJ)
    for i in range(3):
        for j in range(3):
            hij = x @ ddR[i][j].T                                                          # (N,3): second derivative of the point
            H[3 + i, 3 + j] += np.einsum("nk,nka,na->", w, Cq, hij)
    return score, g, d1 * H


# ---- More & Thuente 1994 -----------------------------------------------------------------------------------------
def _std_min(a, b):   # std::min(a, b): returns a unless b < a — a NaN first argument survives (PCL clamps with these)
    return b if b < a else a


def _std_max(a, b):   # std::max(a, b): returns a unless a < b
    return b if a < b else a


def mt_trial_value(al, fl, gl, au, fu, gu, at, ft, gt):
    """Section 4, cases 1-4: cubic / quadratic / secant minimisers.  IEEE arithmetic throughout (0/0 -> NaN, as in C++)."""
    al, fl, gl, au, fu, gu, at, ft, gt = (np.float64(v) for v in (al, fl, gl, au, fu, gu, at, ft, gt))
    with np.errstate(all="ignore"):
        return float(_mt_trial_value(al, fl, gl, au, fu, gu, at, ft, gt))


def _mt_trial_value(al, fl, gl, au, fu, gu, at, ft, gt):
    def cubic(a0, f0, g0, a1, f1, g1):
        z = 3.0 * (f1 - f0) / (a1 - a0) - g1 - g0
        w = np.sqrt(z * z - g1 * g0)
        return a0 + (a1 - a0) * (w - g0 - z) / (g1 - g0 + 2.0 * w)

    if ft > fl:                                   # case 1: higher value -> minimum is bracketed
        ac = cubic(al, fl, gl, at, ft, gt)
        aq = al - 0.5 * (al - at) * gl / (gl - (fl - ft) / (al - at))
        return ac if abs(ac - al) < abs(aq - al) else 0.5 * (aq + ac)
    if gt * gl < 0:                               # case 2: lower value, derivatives of opposite sign
        ac = cubic(al, fl, gl, at, ft, gt)
        as_ = al - (al - at) / (gl - gt) * gl
        return ac if abs(ac - at) >= abs(as_ - at) else as_
    if abs(gt) <= abs(gl):                        # case 3: lower value, same sign, derivative decreases
        ac = cubic(al, fl, gl, at, ft, gt)
        as_ = al - (al - at) / (gl - gt) * gl
        an = ac if abs(ac - at) < abs(as_ - at) else as_
        return _std_min(at + 0.66 * (au - at), an) if at > al else _std_max(at + 0.66 * (au - at), an)
    return cubic(au, fu, gu, at, ft, gt)          # case 4 (PCL evaluates the cubic through (a_u, a_t))


def mt_update_interval(I, at, ft, gt):
    """Section 2 ("Updating Algorithm"); returns True when the interval can no longer be updated (converged)."""
    if ft > I["fl"]:
        I["au"], I["fu"], I["gu"] = at, ft, gt
        return False
    if gt * (I["al"] - at) > 0:
        I["al"], I["fl"], I["gl"] = at, ft, gt
        return False
    if gt * (I["al"] - at) < 0:
        I["au"], I["fu"], I["gu"] = I["al"], I["fl"], I["gl"]
        I["al"], I["fl"], I["gl"] = at, ft, gt
        return False
    return True


def ndt_align(tab, src, p0, resolution, trans_eps=0.01, step_size=0.1, max_iterations=35, outlier_ratio=0.55, d1_sign=+1):
    """Newton + More-Thuente as PCL wires them.  Returns dict(p, iterations, converged, trace=[(p, score, step, evals)])."""
    d1, d2 = gauss_fit(resolution, outlier_ratio)
    mu, nu = 1e-4, 0.9
    p = np.asarray(p0, np.float64).copy()
    evals = 0
    score, g, H = ndt_derivatives(tab, src, p, d1, d2, d1_sign=d1_sign)
    evals += 1
    it, converged, trace = 0, False, []
    while not converged:
        U, S, Vt = np.linalg.svd(H)                       # JacobiSVD(...).solve(-g): least squares through the SVD
        delta = Vt.T @ ((U.T @ (-g)) / S)
        n = np.linalg.norm(delta)
        if n == 0 or not np.isfinite(n):
            return dict(p=p, iterations=it, converged=bool(np.isfinite(n)), trace=trace, score=score)
        delta = delta / n
        # ---- computeStepLengthMT
        phi0, dphi0 = -score, -(g @ delta)
        a_t, line_search = 0.0, True
        if dphi0 >= 0:
            if dphi0 == 0:
                line_search = False                        # not a descent direction at all: step length 0
            else:
                dphi0, delta = -dphi0, -delta              # ascent direction: reverse it
        if line_search:
            I = dict(al=0.0, au=0.0, fl=0.0, fu=0.0, gl=dphi0 - mu * dphi0, gu=dphi0 - mu * dphi0)
            a_min, a_max = trans_eps / 2.0, step_size
            interval_converged, open_interval = (a_max - a_min) < 0, True
            a_t = _std_max(_std_min(n, a_max), a_min)
            x_t = p + delta * a_t
            hang_pose = x_t.copy()                         # h_ang is computed here and NOT refreshed by the trials
            score, g, H = ndt_derivatives(tab, src, x_t, d1, d2, d1_sign=d1_sign)
            evals += 1
            phit, dphit = -score, -(g @ delta)
            psit, dpsit = phit - phi0 - mu * dphi0 * a_t, dphit - mu * dphi0
            k = 0
            while not interval_converged and k < 10 and not (psit <= 0 and dphit <= -nu * dphi0):
                if open_interval:
                    a_t = mt_trial_value(I["al"], I["fl"], I["gl"], I["au"], I["fu"], I["gu"], a_t, psit, dpsit)
                else:
                    a_t = mt_trial_value(I["al"], I["fl"], I["gl"], I["au"], I["fu"], I["gu"], a_t, phit, dphit)
                a_t = _std_max(_std_min(a_t, a_max), a_min)
                x_t = p + delta * a_t
                score, g, _ = ndt_derivatives(tab, src, x_t, d1, d2, with_hessian=False)
                evals += 1
                phit, dphit = -score, -(g @ delta)
                psit, dpsit = phit - phi0 - mu * dphi0 * a_t, dphit - mu * dphi0
                if open_interval and psit <= 0 and dpsit >= 0:
                    open_interval = False                  # from psi to phi: f += phi0 - mu dphi0 a, g += mu dphi0
                    I["fl"] += phi0 - mu * dphi0 * I["al"]
                    I["gl"] += mu * dphi0
                    I["fu"] += phi0 - mu * dphi0 * I["au"]
                    I["gu"] += mu * dphi0
                if open_interval:
                    interval_converged = mt_update_interval(I, a_t, psit, dpsit)
                else:
                    interval_converged = mt_update_interval(I, a_t, phit, dphit)
                k += 1
            if k:
                _, _, H = ndt_derivatives(tab, src, x_t, d1, d2, hang_pose=hang_pose, d1_sign=d1_sign)
                evals += 1
        p = p + delta * a_t
        trace.append((p.copy(), score, a_t, evals))
        if it > max_iterations or (it > 0 and abs(a_t) < trans_eps):
            converged = True
        it += 1
    return dict(p=p, iterations=it, converged=True, trace=trace, score=score)


# ===================================================================================================================
# GICP
# ===================================================================================================================
def gicp_covariances(pts, k=20, eps=1e-3):
    """20-NN sample covariance -> SVD -> U diag(1, 1, eps) U^T (Segal 2009, section IV)."""
    from scipy.spatial import cKDTree

    p = pts.astype(np.float64)
    _, idx = cKDTree(p).query(p, k=k)
    nb = p[idx]                                              # (n,k,3)
    mu = nb.mean(axis=1, keepdims=True)
    d = nb - mu
    C = np.einsum("nka,nkb->nab", d, d) / k
    U, S, _ = np.linalg.svd(C)
    return np.einsum("nak,k,nbk->nab", U, np.array([1.0, 1.0, eps]), U)


def rot_zyx(rx, ry, rz):
    sx, cx, sy, cy, sz, cz = np.sin(rx), np.cos(rx), np.sin(ry), np.cos(ry), np.sin(rz), np.cos(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def gicp_align(tgt, src, guess, max_corr_dist=5.0, trans_eps=1e-8, rot_eps=2e-3, k=20, gicp_eps=1e-3, max_iterations=200):
    """Outer GICP loop with PCL's bookkeeping: the source is first moved by `guess`; every outer iteration finds the 1-NN
    of the points moved by the current transformation, builds M_i = (C2 + R C1 R^T)^-1 with the rotation of
    (transformation * guess), minimises f(x) = 1/m sum r^T M r from the current transformation (scipy BFGS, PCL's
    gradient tolerance 1e-2 is NOT imitated: the inner problem is solved to 1e-10) and stops when no element of the
    transformation moved by more than its epsilon (delta < 1).  Returns final 4x4, outer iterations."""
    from scipy.optimize import minimize
    from scipy.spatial import cKDTree

    T = np.asarray(guess, np.float64)
    out = src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]    # the "output" cloud = guess * source
    tq = tgt.astype(np.float64)
    tree = cKDTree(tq)
    C2 = gicp_covariances(tgt, k, gicp_eps)
    C1 = gicp_covariances(src, k, gicp_eps)
    tr = np.eye(4)
    it = 0
    while True:
        moved = out @ tr[:3, :3].T + tr[:3, 3]
        d, j = tree.query(moved)
        sel = d * d < max_corr_dist ** 2
        if sel.sum() < 4:
            return dict(final=tr @ T, iterations=it, converged=False)
        R = (tr @ T)[:3, :3]
        M = np.linalg.inv(C2[j[sel]] + np.einsum("ab,nbc,dc->nad", R, C1[sel], R))
        ps, qs = out[sel], tq[j[sel]]

        def f(x):
            r = ps @ rot_zyx(*x[3:]).T + x[:3] - qs
            return np.einsum("na,nab,nb->", r, M, r) / ps.shape[0]

        x0 = np.r_[tr[:3, 3], np.arctan2(tr[2, 1], tr[2, 2]), np.arcsin(-tr[2, 0]), np.arctan2(tr[1, 0], tr[0, 0])]
        x = minimize(f, x0, method="BFGS", options=dict(gtol=1e-10)).x
        new = np.eye(4)
        new[:3, :3], new[:3, 3] = rot_zyx(*x[3:]), x[:3]
        delta = 0.0
        for a in range(4):
            for b in range(4):
                e = rot_eps if (a < 3 and b < 3) else trans_eps
                delta = max(delta, abs(new[a, b] - tr[a, b]) / e)
        tr = new
        it += 1
        if it >= max_iterations or delta < 1:
            return dict(final=tr @ T, iterations=it, converged=True)
