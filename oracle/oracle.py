"""ORACLE — TEST INFRASTRUCTURE ONLY (ctypes binding of oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (lidarslam_ros2_amd) never does.  PARITY UNPINNED — see ndt_oracle.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


class NdtParams(C.Structure):
    _fields_ = [("resolution", C.c_double), ("step_size", C.c_double), ("outlier_ratio", C.c_double),
                ("trans_eps", C.c_double), ("max_iterations", C.c_int), ("search", C.c_int),
                ("d1_sign", C.c_int), ("num_threads", C.c_int)]


class NdtResult(C.Structure):
    _fields_ = [("final_transformation", C.c_float * 16), ("converged", C.c_int), ("iterations", C.c_int),
                ("trans_probability", C.c_double), ("final_p", C.c_double * 6), ("n_evals", C.c_int),
                ("n_evals_grad", C.c_int), ("n_hessian_recompute", C.c_int)]


class GicpParams(C.Structure):
    _fields_ = [("max_corr_dist", C.c_double), ("trans_eps", C.c_double), ("rot_eps", C.c_double),
                ("gicp_eps", C.c_double), ("max_iterations", C.c_int), ("max_inner_iterations", C.c_int),
                ("k_correspondences", C.c_int), ("solver", C.c_int), ("num_threads", C.c_int)]


class GicpResult(C.Structure):
    _fields_ = [("final_transformation", C.c_float * 16), ("converged", C.c_int), ("iterations", C.c_int),
                ("n_correspondences", C.c_int), ("final_cost", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, dp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.orc_grid_build.restype = vp
        L.orc_grid_build.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_float]
        L.orc_grid_free.argtypes = [vp]
        L.orc_grid_info.argtypes = [vp, ip]
        L.orc_grid_dump.restype = C.c_int
        L.orc_grid_dump.argtypes = [vp, ip, ip, dp, dp, dp]
        L.orc_grid_centroids.restype = C.c_int
        L.orc_grid_centroids.argtypes = [vp, fp]
        L.orc_gauss_constants.argtypes = [C.c_double, C.c_double, dp, dp, dp]
        L.orc_pose_to_matrix.argtypes = [dp, fp]
        L.orc_matrix_to_pose.argtypes = [fp, dp]
        L.orc_ndt_derivatives.restype = C.c_double
        L.orc_ndt_derivatives.argtypes = [vp, fp, C.c_size_t, C.c_size_t, dp, fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_double, dp, dp, C.c_int]
        L.orc_ndt_align.restype = C.c_int
        L.orc_ndt_align.argtypes = [vp, fp, C.c_size_t, C.c_size_t, fp, C.POINTER(NdtParams), C.POINTER(NdtResult),
                                    dp, C.c_int]
        L.orc_ndt_align_cb.restype = C.c_int
        L.orc_ndt_align_cb.argtypes = [vp, fp, C.c_size_t, C.c_size_t, fp, C.POINTER(NdtParams), C.POINTER(NdtResult),
                                       dp, C.c_int, vp, vp]
        L.orc_max_threads.restype = C.c_int
        L.orc_voxel_grid_filter.restype = C.c_int
        L.orc_voxel_grid_filter.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_float, fp]
        L.orc_voxel_grid_filter_xyzi.restype = C.c_int
        L.orc_voxel_grid_filter_xyzi.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_float, C.c_int, fp]
        if hasattr(L, "orc_nn_build"):
            L.orc_nn_build.restype = vp
            L.orc_nn_build.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_float]
            L.orc_nn_free.argtypes = [vp]
            L.orc_nn_search.argtypes = [vp, fp, C.c_size_t, C.c_size_t, fp, ip, fp, C.c_int]
            L.orc_knn_search.argtypes = [vp, fp, C.c_size_t, C.c_size_t, C.c_int, ip, fp, C.c_int]
            L.orc_fitness_score.restype = C.c_double
            L.orc_fitness_score.argtypes = [vp, fp, C.c_size_t, C.c_size_t, fp, C.c_double, C.c_int]
        if hasattr(L, "orc_gicp_covariances"):
            L.orc_gicp_covariances.argtypes = [vp, fp, C.c_size_t, C.c_size_t, C.c_int, C.c_double, dp, C.c_int]
            L.orc_gicp_align.restype = C.c_int
            L.orc_gicp_align.argtypes = [vp, fp, C.c_size_t, C.c_size_t, dp, vp, fp, C.c_size_t, C.c_size_t, dp, fp,
                                         C.POINTER(GicpParams), C.POINTER(GicpResult)]
            L.orc_gicp_cost.restype = C.c_double
            L.orc_gicp_cost.argtypes = [fp, fp, dp, C.c_size_t, dp, dp, dp]
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _f64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def max_threads() -> int:
    return int(lib().orc_max_threads())


def gauss_constants(res: float, outlier: float = 0.55):
    d = [C.c_double() for _ in range(3)]
    lib().orc_gauss_constants(res, outlier, *[C.byref(x) for x in d])
    return tuple(x.value for x in d)


def pose_to_matrix(p) -> np.ndarray:
    p = np.ascontiguousarray(p, np.float64)
    M = np.zeros(16, np.float32)
    lib().orc_pose_to_matrix(_f64p(p), M.ctypes.data_as(C.POINTER(C.c_float)))
    return M.reshape(4, 4).T.copy()


def matrix_to_pose(M) -> np.ndarray:
    Mc = np.ascontiguousarray(np.asarray(M, np.float32).T).reshape(-1)
    p = np.zeros(6, np.float64)
    lib().orc_matrix_to_pose(Mc.ctypes.data_as(C.POINTER(C.c_float)), _f64p(p))
    return p


def voxel_grid_filter_xyzi(pts: np.ndarray, leaf: float, intensity_col: int) -> np.ndarray:
    """pcl::VoxelGrid::filter with downsample_all_data (PCL's default): (m, 4) = leaf means of x, y, z and of the intensity
    found in column `intensity_col` of the input records."""
    a, ap = _f32(pts)
    out = np.zeros((a.shape[0], 4), np.float32)
    n = lib().orc_voxel_grid_filter_xyzi(ap, a.shape[1], a.shape[0], C.c_float(leaf), int(intensity_col),
                                         out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:n].copy()


def voxel_grid_filter(pts: np.ndarray, leaf: float) -> np.ndarray:
    """CPU restatement of pcl::VoxelGrid::filter (centroid per leaf, leaf-index order)."""
    a, ap = _f32(pts)
    out = np.zeros((a.shape[0], 3), np.float32)
    n = lib().orc_voxel_grid_filter(ap, a.shape[1], a.shape[0], C.c_float(leaf), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:n].copy()


class VoxelGridCovariance:
    """CPU restatement of pclomp::VoxelGridCovariance."""

    def __init__(self, pts: np.ndarray, leaf: float):
        self._pts, p = _f32(pts)
        assert self._pts.ndim == 2 and self._pts.shape[1] >= 3
        self.leaf = float(leaf)
        self.h = lib().orc_grid_build(p, self._pts.shape[1], self._pts.shape[0], C.c_float(leaf))
        info = np.zeros(9, np.int32)
        lib().orc_grid_info(self.h, _i32p(info))
        self.min_b, self.max_b = info[0:3].copy(), info[3:6].copy()
        self.n_leaves, self.n_valid, self.overflow = int(info[6]), int(info[7]), bool(info[8])

    def dump(self):
        n = self.n_leaves
        idx, npts = np.zeros(n, np.int32), np.zeros(n, np.int32)
        mean, cov, icov = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3, 3))
        lib().orc_grid_dump(self.h, _i32p(idx), _i32p(npts), _f64p(mean), _f64p(cov), _f64p(icov))
        return dict(idx=idx, n=npts, mean=mean, cov=cov, icov=icov)

    def centroids(self):
        """(n_leaves, 3) fp32: Leaf::centroid of every leaf (float running sum in cloud order / float(n)), sorted by linear index —
        the points of the voxel-centroid kd-tree that the KDTREE neighbourhood search (search=0) queries."""
        c = np.zeros((self.n_leaves, 3), np.float32)
        lib().orc_grid_centroids(self.h, c.ctypes.data_as(C.POINTER(C.c_float)))
        return c

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_grid_free(self.h)
                self.h = None
        except Exception:
            pass


def ndt_derivatives(grid: VoxelGridCovariance, src: np.ndarray, p, *, T=None, compute_hessian=True, search=7,
                    d1_sign=1, num_threads=0, resolution=None, outlier_ratio=0.55, fp64_hessian=False):
    s, sp = _f32(src)
    p = np.ascontiguousarray(p, np.float64)
    g, H = np.zeros(6), np.zeros((6, 6))
    Tp = None
    if T is not None:
        Tc = np.ascontiguousarray(np.asarray(T, np.float32).T).reshape(-1)
        Tp = Tc.ctypes.data_as(C.POINTER(C.c_float))
    score = lib().orc_ndt_derivatives(grid.h, sp, s.shape[1], s.shape[0], _f64p(p), Tp, int(compute_hessian), search,
                                      d1_sign, num_threads, resolution or grid.leaf, outlier_ratio, _f64p(g), _f64p(H),
                                      int(fp64_hessian))
    return score, g, H


def ndt_align(grid: VoxelGridCovariance, src: np.ndarray, guess=None, *, resolution=None, step_size=0.1,
              outlier_ratio=0.55, trans_eps=0.01, max_iterations=35, search=7, d1_sign=1, num_threads=0,
              trace=False, deriv_cb=None):
    """deriv_cb = (function address, user pointer): run the same Newton / More-Thuente loop on another derivative evaluation
    (ndt_oracle.cpp DerivCb; tests/ndt_host_emu.py plugs the GPU kernels' arithmetic compiled for the host in here)."""
    s, sp = _f32(src)
    prm = NdtParams(resolution or grid.leaf, step_size, outlier_ratio, trans_eps, max_iterations, search, d1_sign,
                    num_threads)
    res = NdtResult()
    gp = None
    if guess is not None:
        gc = np.ascontiguousarray(np.asarray(guess, np.float32).T).reshape(-1)
        gp = gc.ctypes.data_as(C.POINTER(C.c_float))
    cap = max_iterations + 4
    tr = np.zeros((cap, 9))
    if deriv_cb is None:
        lib().orc_ndt_align(grid.h, sp, s.shape[1], s.shape[0], gp, C.byref(prm), C.byref(res), _f64p(tr), cap)
    else:
        lib().orc_ndt_align_cb(grid.h, sp, s.shape[1], s.shape[0], gp, C.byref(prm), C.byref(res), _f64p(tr), cap,
                               C.c_void_p(deriv_cb[0]), C.c_void_p(deriv_cb[1]))
    out = dict(final=np.array(res.final_transformation, np.float32).reshape(4, 4).T.copy(),
               converged=bool(res.converged), iterations=int(res.iterations),
               trans_probability=float(res.trans_probability), p=np.array(res.final_p),
               n_evals=int(res.n_evals), n_evals_grad=int(res.n_evals_grad),
               n_hessian_recompute=int(res.n_hessian_recompute))
    if trace:
        out["trace"] = tr[: res.iterations]
    return out


class NearestNeighbour:
    """Exact grid-hash NN over a target cloud (stand-in for pcl::KdTreeFLANN in the CPU path)."""

    def __init__(self, pts: np.ndarray, cell: float = 1.0):
        self._pts, p = _f32(pts)
        self.h = lib().orc_nn_build(p, self._pts.shape[1], self._pts.shape[0], C.c_float(cell))

    def search(self, q: np.ndarray, T=None, num_threads=0):
        qq, qp = _f32(q)
        idx, d2 = np.zeros(qq.shape[0], np.int32), np.zeros(qq.shape[0], np.float32)
        Tp = None
        if T is not None:
            Tc = np.ascontiguousarray(np.asarray(T, np.float32).T).reshape(-1)
            Tp = Tc.ctypes.data_as(C.POINTER(C.c_float))
        lib().orc_nn_search(self.h, qp, qq.shape[1], qq.shape[0], Tp, _i32p(idx), d2.ctypes.data_as(C.POINTER(C.c_float)),
                            num_threads)
        return idx, d2

    def knn(self, q: np.ndarray, k: int, num_threads=0):
        qq, qp = _f32(q)
        idx, d2 = np.zeros((qq.shape[0], k), np.int32), np.zeros((qq.shape[0], k), np.float32)
        lib().orc_knn_search(self.h, qp, qq.shape[1], qq.shape[0], k, _i32p(idx),
                             d2.ctypes.data_as(C.POINTER(C.c_float)), num_threads)
        return idx, d2

    def fitness_score(self, src: np.ndarray, T, max_range: float = float("inf"), num_threads=0) -> float:
        s, sp = _f32(src)
        Tc = np.ascontiguousarray(np.asarray(T, np.float32).T).reshape(-1)
        mr = max_range if np.isfinite(max_range) else 1.7976931348623157e308
        return float(lib().orc_fitness_score(self.h, sp, s.shape[1], s.shape[0],
                                             Tc.ctypes.data_as(C.POINTER(C.c_float)), mr, num_threads))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_nn_free(self.h)
                self.h = None
        except Exception:
            pass


def gicp_raw_covariances(nn: NearestNeighbour, pts: np.ndarray, k: int = 20, num_threads=0):
    """The k-neighbour sample covariances of computeCovariances BEFORE the SVD regularisation (FLOAT products, double sums)."""
    return gicp_covariances(nn, pts, k=k, gicp_eps=-1.0, num_threads=num_threads)


def gicp_covariances(nn: NearestNeighbour, pts: np.ndarray, k: int = 20, gicp_eps: float = 1e-3, num_threads=0):
    p, pp = _f32(pts)
    cov = np.zeros((p.shape[0], 3, 3))
    lib().orc_gicp_covariances(nn.h, pp, p.shape[1], p.shape[0], k, gicp_eps, _f64p(cov), num_threads)
    return cov


def gicp_align(nn_tgt: NearestNeighbour, tgt: np.ndarray, tgt_cov: np.ndarray, src: np.ndarray, src_cov: np.ndarray,
               guess=None, *, max_corr_dist=5.0, trans_eps=1e-8, rot_eps=2e-3, gicp_eps=1e-3, max_iterations=200,
               max_inner_iterations=20, k=20, solver=0, num_threads=0):
    """solver 0 = BFGS (reference schedule), 1 = Gauss-Newton (what the GPU core runs)."""
    t, tp = _f32(tgt)
    s, sp = _f32(src)
    tc = np.ascontiguousarray(tgt_cov, np.float64)
    sc = np.ascontiguousarray(src_cov, np.float64)
    g = np.eye(4, dtype=np.float32) if guess is None else np.asarray(guess, np.float32)
    gc = np.ascontiguousarray(g.T).reshape(-1)
    prm = GicpParams(max_corr_dist, trans_eps, rot_eps, gicp_eps, max_iterations, max_inner_iterations, k, solver,
                     num_threads)
    res = GicpResult()
    lib().orc_gicp_align(nn_tgt.h, tp, t.shape[1], t.shape[0], _f64p(tc), None, sp, s.shape[1], s.shape[0], _f64p(sc),
                         gc.ctypes.data_as(C.POINTER(C.c_float)), C.byref(prm), C.byref(res))
    return dict(final=np.array(res.final_transformation, np.float32).reshape(4, 4).T.copy(),
                converged=bool(res.converged), iterations=int(res.iterations),
                n_correspondences=int(res.n_correspondences), final_cost=float(res.final_cost))


def gicp_cost(src_xyz: np.ndarray, tgt_xyz: np.ndarray, M: np.ndarray, x, with_gradient: bool = True):
    """f(x) = 1/m sum r^T M r over paired points (pair i = src[i], tgt[i]) and its gradient, as
    OptimizationFunctorWithIndices::fdf evaluates them; x = (t, rx, ry, rz), R = Rz Ry Rx."""
    s = np.ascontiguousarray(np.asarray(src_xyz, np.float32)[:, :3])
    t = np.ascontiguousarray(np.asarray(tgt_xyz, np.float32)[:, :3])
    Mc = np.ascontiguousarray(M, np.float64)
    xv = np.ascontiguousarray(x, np.float64)
    g = np.zeros(6)
    fp = C.POINTER(C.c_float)
    f = lib().orc_gicp_cost(s.ctypes.data_as(fp), t.ctypes.data_as(fp), _f64p(Mc), s.shape[0], _f64p(xv),
                            _f64p(g) if with_gradient else None, None)
    return float(f), g


# ---- loop-closure gate: GraphBasedSlamComponent::searchLoop -------------------
def pose_msg_to_matrix(position, orientation) -> np.ndarray:
    """tf2::fromMsg(geometry_msgs::Pose) -> Eigen::Affine3d (graph_based_slam_component.cpp:171,219):
    Translation * Quaterniond(w,x,y,z) with Eigen's un-normalised toRotationMatrix()."""
    x, y, z, w = [float(v) for v in orientation]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    M = np.eye(4)
    M[:3, :3] = [[1 - (tyy + tzz), txy - twz, txz + twy],
                 [txy + twz, 1 - (txx + tzz), tyz - twx],
                 [txz - twy, tyz + twx, 1 - (txx + tyy)]]
    M[:3, 3] = [float(v) for v in position]
    return M


def transform_point_cloud(pts: np.ndarray, M) -> np.ndarray:
    """pcl::transformPointCloud(in, out, Matrix4f) for finite points: fp32 ((m0 x + m1 y) + m2 z) + m3."""
    a = np.asarray(pts, np.float32)[:, :3]
    m = np.asarray(M, np.float32)
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    out = np.empty((a.shape[0], 3), np.float32)
    for r in range(3):
        out[:, r] = ((m[r, 0] * x + m[r, 1] * y) + m[r, 2] * z) + m[r, 3]
    return out


def search_loop(submaps, *, threshold_loop_closure_score=1.0, distance_loop_closure=20.0,
                range_of_searching_loop_closure=20.0, search_submap_num=3, voxel_leaf_size=0.2, top_k=1,
                method="ndt", ndt_resolution=1.0, trans_eps=0.01, max_iterations=100, step_size=0.1, num_threads=0,
                gicp_corr_dist=5.0, gicp_trans_eps=1e-8, gicp_solver=1):
    """CPU restatement of searchLoop() from `latest_submap` on (graph_based_slam_component.cpp:164-252).
    submaps: sequence of dicts {cloud (n,>=3) f32, position (3), orientation (4, xyzw), distance}.
    Returns one dict per evaluated candidate (nearest first): pair_id, fitness_score, accepted, relative_pose, final."""
    n = len(submaps)
    latest = submaps[n - 1]
    init = pose_msg_to_matrix(latest["position"], latest["orientation"])                     # :169-171
    source = transform_point_cloud(latest["cloud"], init.astype(np.float32))                 # :176-181
    lp = np.asarray(latest["position"], np.float64)
    cand = []
    for i, sm in enumerate(submaps):                                                         # :188-205
        dist = float(np.linalg.norm(lp - np.asarray(sm["position"], np.float64)))
        if latest["distance"] - sm["distance"] > distance_loop_closure and dist < range_of_searching_loop_closure:
            cand.append((dist, i))
    cand.sort(key=lambda c: c[0])  # stable: first minimum wins, like the strict `dist < min_dist`
    out = []
    for dist, id_min in cand[:max(1, top_k)]:
        parts = []
        for j in range(2 * search_submap_num + 1):                                          # :209-222
            idx = id_min + j - search_submap_num
            if idx < 0 or idx >= n:  # (the reference guards the lower end only)
                continue
            sm = submaps[idx]
            if len(sm["cloud"]):
                parts.append(transform_point_cloud(sm["cloud"], pose_msg_to_matrix(sm["position"], sm["orientation"]).astype(np.float32)))
        window = np.concatenate(parts) if parts else np.zeros((0, 3), np.float32)
        target = voxel_grid_filter(window, voxel_leaf_size)                                  # :224-226
        if method == "ndt":
            grid = VoxelGridCovariance(target, ndt_resolution)                               # :227
            r = ndt_align(grid, source, None, trans_eps=trans_eps, max_iterations=max_iterations, step_size=step_size,
                          num_threads=num_threads)                                           # :230
        else:
            nn_t = NearestNeighbour(target)
            nn_s = NearestNeighbour(source)
            r = gicp_align(nn_t, target, gicp_covariances(nn_t, target, num_threads=num_threads), source,
                           gicp_covariances(nn_s, source, num_threads=num_threads), None, max_corr_dist=gicp_corr_dist,
                           trans_eps=gicp_trans_eps, max_iterations=max_iterations, solver=gicp_solver, num_threads=num_threads)   # 1 = Gauss-Newton (the device's), 0 = the reference's BFGS
        fitness = NearestNeighbour(target).fitness_score(source, r["final"], num_threads=num_threads)   # :231
        frm = pose_msg_to_matrix(submaps[id_min]["position"], submaps[id_min]["orientation"])
        to = r["final"].astype(np.float64) @ init                                           # :241-243
        inv = np.eye(4)
        inv[:3, :3] = frm[:3, :3].T
        inv[:3, 3] = -frm[:3, :3].T @ frm[:3, 3]                                             # Isometry3d::inverse()
        out.append(dict(pair_id=(id_min, n - 1), fitness_score=float(fitness),
                        accepted=bool(fitness < threshold_loop_closure_score), relative_pose=inv @ to,
                        final=r["final"], iterations=r["iterations"], converged=r["converged"],
                        n_target_points=int(target.shape[0]), candidate_distance=dist))
    return out
