"""Host-side mirror of the `pcl::Registration` surface the reference's two nodes call
(SURVEY.md §8b), on top of the C ABI in include/lidarslam_reg.h.

Method names, argument meaning and defaults follow pclomp / PCL so call sites read like the
reference's: scanmatcher/src/scanmatcher_component.cpp:105-120,275,307,329,353-376 and
graph_based_slam/src/graph_based_slam_component.cpp:64-82,181,227-231.

Clouds may be numpy arrays (host) or torch CUDA tensors (already resident in HBM): shape (n, c)
float32 with c >= 3 and xyz in the first three columns (c = 8 is pcl::PointXYZI's 32-byte record).
4x4 transforms are numpy (4,4) float32, row/col indexed normally (converted to Eigen's
column-major order at the boundary).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _capi as capi

DIRECT7, DIRECT1, DIRECT26, KDTREE = capi.DIRECT7, capi.DIRECT1, capi.DIRECT26, capi.KDTREE


def _is_torch_cuda(x) -> bool:
    return hasattr(x, "is_cuda") and bool(x.is_cuda)


def _cloud_args(cloud, reg=None):
    """-> (pointer, stride_bytes, n, on_device, keepalive).  reg: the registration object that is going to read the cloud —
    for a CUDA tensor its stream is ordered after torch's current stream AFTER any .contiguous() copy has been enqueued there
    (the copy is the buffer the core reads; an event recorded before it would not cover it)."""
    if _is_torch_cuda(cloud):
        import torch

        t = cloud
        if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] < 3:
            raise ValueError("device cloud must be float32 of shape (n, c>=3)")
        if not t.is_contiguous():
            t = t.contiguous()
        if reg is not None:
            _order_after_torch(reg, t)
        return C.c_void_p(t.data_ptr()), t.shape[1] * 4, t.shape[0], True, t
    a = np.asarray(cloud)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("cloud must have shape (n, c>=3)")
    a = np.ascontiguousarray(a, np.float32)
    return C.c_void_p(a.ctypes.data), a.shape[1] * 4, a.shape[0], False, a


def _mat_to_col16(M) -> np.ndarray:
    M = np.asarray(M, np.float32)
    if M.shape != (4, 4):
        raise ValueError("transform must be 4x4")
    return np.ascontiguousarray(M.T).reshape(16)


def _col16_to_mat(v) -> np.ndarray:
    return np.asarray(v, np.float32).reshape(4, 4).T.copy()


def _order_after_torch(reg, cloud):
    """A CUDA tensor may be the result of kernels still running on torch's current stream: make the handle's stream wait
    for that stream (lsr_wait_stream: event record + stream wait, no host wait) before the core reads the tensor."""
    if _is_torch_cuda(cloud):
        import torch

        capi.check(reg._lib.lsr_wait_stream(reg._h, C.c_void_p(torch.cuda.current_stream(cloud.device).cuda_stream)), "lsr_wait_stream")


class Registration:
    """pcl::Registration<PointXYZI, PointXYZI>-shaped base."""

    _method = capi.METHOD_NDT

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = capi.load()
        h = C.c_void_p()
        capi.check(self._lib.lsr_create(self._method, device, C.c_void_p(stream) if stream else None, C.byref(h)),
                   "lsr_create")
        self._h = h
        self._device = device
        self._keep = {}
        self._last = capi.Result()

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.lsr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setters exercised by the reference ---------------------------------------------------
    def _setf(self, key, v, where):
        capi.check(self._lib.lsr_set_f64(self._h, key, float(v)), where)

    def _seti(self, key, v, where):
        capi.check(self._lib.lsr_set_i32(self._h, key, int(v)), where)

    def _getf(self, key):
        v = C.c_double()
        capi.check(self._lib.lsr_get_f64(self._h, key, C.byref(v)), "lsr_get_f64")
        return v.value

    def _geti(self, key):
        v = C.c_int32()
        capi.check(self._lib.lsr_get_i32(self._h, key, C.byref(v)), "lsr_get_i32")
        return v.value

    def setTransformationEpsilon(self, eps: float):  # scanmatcher_component.cpp:108,119
        self._setf(capi.TRANSFORMATION_EPSILON, eps, "setTransformationEpsilon")

    def getTransformationEpsilon(self) -> float:
        return self._getf(capi.TRANSFORMATION_EPSILON)

    def setMaximumIterations(self, n: int):  # graph_based_slam_component.cpp:66,77
        self._seti(capi.MAX_ITERATIONS, n, "setMaximumIterations")

    def getMaximumIterations(self) -> int:
        return self._geti(capi.MAX_ITERATIONS)

    def setMaxCorrespondenceDistance(self, d: float):  # scanmatcher_component.cpp:118
        self._setf(capi.MAX_CORRESPONDENCE_DISTANCE, d, "setMaxCorrespondenceDistance")

    def setEuclideanFitnessEpsilon(self, eps: float):  # graph_based_slam_component.cpp:80
        self._setf(capi.EUCLIDEAN_FITNESS_EPSILON, eps, "setEuclideanFitnessEpsilon")

    def setRANSACIterations(self, n: int):  # graph_based_slam_component.cpp:81
        self._seti(capi.RANSAC_ITERATIONS, n, "setRANSACIterations")

    # -- clouds ------------------------------------------------------------------------------
    def setInputTarget(self, cloud):  # scanmatcher_component.cpp:275,307,315; graph_based_slam_component.cpp:227
        p, stride, n, dev, keep = _cloud_args(cloud, self)
        fn = self._lib.lsr_set_input_target_device if dev else self._lib.lsr_set_input_target
        capi.check(fn(self._h, p, stride, n), "setInputTarget")
        self._keep["target"] = None  # the core keeps its own SoA copy in HBM

    def setInputTargetFrames(self, frames, poses):
        """Submap assembly on the device: frame f transformed by poses[f] (4x4), concatenated, then
        setInputTarget (scanmatcher_component.cpp:449-464,307).  Frames: host arrays or CUDA tensors, same layout."""
        args = [_cloud_args(f) for f in frames]
        dev_frames = [a[4] for a in args if _is_torch_cuda(a[4])]
        if dev_frames:
            _order_after_torch(self, dev_frames[-1])   # one ordering, after every .contiguous() copy has been enqueued
        dev = args[0][3]
        if any(a[3] != dev for a in args) or any(a[1] != args[0][1] for a in args):
            raise ValueError("frames must all be host or all device, with one record stride")
        nf = len(args)
        ptrs = (C.c_void_p * nf)(*[a[0] for a in args])
        counts = (C.c_size_t * nf)(*[a[2] for a in args])
        P = np.ascontiguousarray(np.stack([_mat_to_col16(p) for p in poses]), np.float32)
        capi.check(self._lib.lsr_set_input_target_frames(self._h, nf, ptrs, counts, args[0][1],
                                                         P.ctypes.data_as(C.POINTER(C.c_float)), 1 if dev else 0),
                   "setInputTargetFrames")
        self._n_target = int(sum(a[2] for a in args))

    def setInputSource(self, cloud):  # scanmatcher_component.cpp:329; graph_based_slam_component.cpp:181
        p, stride, n, dev, keep = _cloud_args(cloud, self)
        fn = self._lib.lsr_set_input_source_device if dev else self._lib.lsr_set_input_source
        capi.check(fn(self._h, p, stride, n), "setInputSource")
        self._n_source = n
        self._keep["source"] = keep if dev else None  # device upload is asynchronous on the handle's stream

    def setInputSourceFiltered(self, cloud, leaf: float) -> int:
        """pcl::VoxelGrid(leaf).filter + setInputSource on the device (scanmatcher_component.cpp:324-329);
        returns the number of points kept."""
        p, stride, n, dev, keep = _cloud_args(cloud, self)
        n_out = C.c_size_t()
        capi.check(self._lib.lsr_set_input_source_filtered(self._h, p, stride, n, C.c_float(leaf), 1 if dev else 0,
                                                           C.byref(n_out)), "setInputSourceFiltered")
        self._n_source = int(n_out.value)
        return self._n_source

    def setInputSourceFrontend(self, cloud, scan_min_range: float, scan_max_range: float, vg_size_for_input: float) -> int:
        """Range filter (scanmatcher_component.cpp:210-218) + VoxelGrid (:324-328) + setInputSource (:329) on the
        device; returns the number of points kept."""
        p, stride, n, dev, keep = _cloud_args(cloud, self)
        n_out = C.c_size_t()
        capi.check(self._lib.lsr_set_input_source_frontend(self._h, p, stride, n, float(scan_min_range), float(scan_max_range),
                                                           C.c_float(vg_size_for_input), 1 if dev else 0, C.byref(n_out)),
                   "setInputSourceFrontend")
        self._n_source = int(n_out.value)
        return self._n_source

    # -- sensor_msgs/PointCloud2 codec ----------------------------------------------
    @staticmethod
    def _layout(point_step, offsets):
        ox, oy, oz, oi = offsets
        return capi.Pc2Layout(int(point_step), int(ox), int(oy), int(oz), int(-1 if oi is None else oi))

    def setInputSourcePointCloud2(self, data, n_points: int, point_step: int, offsets, scan_min_range: float, scan_max_range: float,
                                  vg_size_for_input: float) -> int:
        """pcl::fromROSMsg + range filter + VoxelGrid + setInputSource from a raw PointCloud2 `data` buffer (bytes / uint8 numpy
        array / CUDA uint8 tensor); offsets = (x, y, z, intensity or None) in bytes.  Returns the number of points kept."""
        lay = self._layout(point_step, offsets)
        n_out = C.c_size_t()
        if _is_torch_cuda(data):
            _order_after_torch(self, data)
            ptr, dev, keep = C.c_void_p(data.data_ptr()), 1, data
        else:
            keep = np.frombuffer(data, np.uint8) if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data).view(np.uint8)
            ptr, dev = C.c_void_p(keep.ctypes.data), 0
        capi.check(self._lib.lsr_set_input_source_pc2(self._h, ptr, int(n_points), C.byref(lay), float(scan_min_range), float(scan_max_range),
                                                      C.c_float(vg_size_for_input), dev, C.byref(n_out)), "setInputSourcePointCloud2")
        self._n_source = int(n_out.value)
        return self._n_source

    def voxelFilterForm(self) -> int:
        """Which form the last VoxelGrid filter on this object took (LSR_VOXEL_FILTER_FORM): 1 = grid dimensions on the host, 2 = on the
        device (one host wait per scan), 3 = the device form came back flagged and the host form ran; 0 = none yet."""
        return self._geti(capi.VOXEL_FILTER_FORM)

    def getInputSourcePointCloud2(self, point_step: int = 32, offsets=(0, 4, 8, 16)) -> np.ndarray:
        """pcl::toROSMsg of the current input source: (n, point_step) uint8 records (default: pcl::PointXYZI's layout)."""
        lay = self._layout(point_step, offsets)
        out = np.zeros((max(self._n_source, 1), point_step), np.uint8)
        n_out = C.c_size_t()
        capi.check(self._lib.lsr_get_source_pc2(self._h, C.c_void_p(out.ctypes.data), out.shape[0], C.byref(lay), C.byref(n_out)),
                   "getInputSourcePointCloud2")
        return out[: n_out.value].copy()

    def getInputSourceDeviceRecords(self):
        """The current input source as pcl::PointXYZI records in HBM: an (n, 8) fp32 CUDA tensor (x y z _ intensity _ _ _), e.g. a
        keyframe for setInputTargetFrames that never leaves the device (lsr_get_source_pc2_device)."""
        import torch

        lay = self._layout(32, (0, 4, 8, 16))
        # on the OBJECT's device (not torch's current one), and the object's stream ordered behind whatever torch's stream may
        # still be doing with the block the caching allocator hands out
        dev = torch.device("cuda", self._device)
        out = torch.empty((max(self._n_source, 1), 8), dtype=torch.float32, device=dev)
        capi.check(self._lib.lsr_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "lsr_wait_stream")
        n_out = C.c_size_t()
        capi.check(self._lib.lsr_get_source_pc2_device(self._h, C.c_void_p(out.data_ptr()), out.shape[0], C.byref(lay), C.byref(n_out)),
                   "getInputSourceDeviceRecords")
        return out[: n_out.value]

    def voxelGridFilterPointCloud2(self, data, n_points: int, point_step: int, offsets, leaf: float, out_point_step: int = 32,
                                   out_offsets=(0, 4, 8, 16)) -> np.ndarray:
        """pcl::VoxelGrid(leaf).filter on a PointCloud2 payload (host), all four fields averaged per leaf; -> (m, out_point_step) uint8."""
        li, lo = self._layout(point_step, offsets), self._layout(out_point_step, out_offsets)
        src = np.frombuffer(data, np.uint8) if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data).view(np.uint8)
        out = np.zeros((max(int(n_points), 1), out_point_step), np.uint8)
        n_out = C.c_size_t()
        capi.check(self._lib.lsr_voxel_grid_filter_pc2(self._h, C.c_void_p(src.ctypes.data), int(n_points), C.byref(li), C.c_float(leaf),
                                                       C.c_void_p(out.ctypes.data), out.shape[0], C.byref(lo), C.byref(n_out)),
                   "voxelGridFilterPointCloud2")
        return out[: n_out.value].copy()

    def voxelGridFilter(self, cloud, leaf: float) -> np.ndarray:
        """Stand-alone pcl::VoxelGrid(leaf).filter on the device; host (n,c>=3) in, (m,3) fp32 out."""
        p, stride, n, dev, keep = _cloud_args(cloud)
        if dev:
            raise ValueError("voxelGridFilter takes a host array")
        out = np.zeros((max(n, 1), 3), np.float32)
        n_out = C.c_size_t()
        capi.check(self._lib.lsr_voxel_grid_filter(self._h, p, stride, n, C.c_float(leaf), C.c_void_p(out.ctypes.data), 12,
                                                   out.shape[0], C.byref(n_out)), "voxelGridFilter")
        return out[: n_out.value].copy()

    def shareTargetOf(self, other: "Registration"):
        """Register against the target already resident in `other` (N keyframes vs one submap)."""
        capi.check(self._lib.lsr_share_target(self._h, other._h), "shareTargetOf")

    # -- align + accessors -------------------------------------------------------------------
    def align(self, guess=None, output: bool = False):
        """registration_->align(output, guess) (scanmatcher_component.cpp:353).  Returns the
        transformed source as (n,3) fp32 when output=True, else None (both reference callers
        discard it)."""
        g = _mat_to_col16(guess) if guess is not None else None
        gp = g.ctypes.data_as(C.POINTER(C.c_float)) if g is not None else None
        fin = np.zeros(16, np.float32)
        out = None
        outp, stride = None, 0
        if output:
            out = np.zeros((self._n_source, 3), np.float32)
            outp, stride = C.c_void_p(out.ctypes.data), 12
        capi.check(self._lib.lsr_align(self._h, gp, fin.ctypes.data_as(C.POINTER(C.c_float)), C.byref(self._last), outp,
                                       stride), "align")
        return out

    def getFinalTransformation(self) -> np.ndarray:  # scanmatcher_component.cpp:356
        fin = np.zeros(16, np.float32)
        capi.check(self._lib.lsr_get_final_transformation(self._h, fin.ctypes.data_as(C.POINTER(C.c_float))),
                   "getFinalTransformation")
        return _col16_to_mat(fin)

    def hasConverged(self) -> bool:  # scanmatcher_component.cpp:375
        v = C.c_int32()
        capi.check(self._lib.lsr_has_converged(self._h, C.byref(v)), "hasConverged")
        return bool(v.value)

    def getFitnessScore(self, max_range: float = 1.7976931348623157e308) -> float:  # graph_based_slam_component.cpp:231
        v = C.c_double()
        capi.check(self._lib.lsr_get_fitness_score(self._h, float(max_range), C.byref(v)), "getFitnessScore")
        return v.value

    # -- extras (not part of the PCL surface) -------------------------------------------------
    @property
    def last_result(self) -> dict:
        r = self._last
        return dict(converged=bool(r.converged), iterations=int(r.iterations), score=float(r.score),
                    n_evaluations=int(r.n_evaluations), n_correspondences=int(r.n_correspondences),
                    gpu_ms=float(r.gpu_ms))

    def getFinalNumIteration(self) -> int:
        return int(self._last.iterations)

    def nearestNeighbors(self, T=None):
        idx = np.zeros(self._n_source, np.int32)
        d2 = np.zeros(self._n_source, np.float32)
        t = _mat_to_col16(T) if T is not None else None
        capi.check(self._lib.lsr_nearest_neighbors(self._h, t.ctypes.data_as(C.POINTER(C.c_float)) if t is not None else None,
                                                   idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                                   d2.ctypes.data_as(C.POINTER(C.c_float))), "nearestNeighbors")
        return idx, d2

    def setTuning(self, workgroup: Optional[int] = None, table_mode: Optional[int] = None, grid_builder: Optional[int] = None,
                  wait_mode: Optional[int] = None, quad: Optional[int] = None, sort: Optional[int] = None, split: Optional[int] = None):
        """Tuning keys of the core (no effect on results: every derivative kernel returns the same bits): NDT workgroup (0 auto;
        quad kernel: 64 / 128 points, lane kernel: 512 / 1024 threads), kernel (quad: -1 auto, 0 lane kernel, 1 quad kernel),
        where the derivative pass reads the voxel table (-1 auto, 0 dense global, 1 compact global, 2 LDS, 3 tile), the
        grid builder (0 auto, 1 radix sort), how the calling thread waits (0 spin, 1 yield, 2 sleep), source ordering by
        voxel tile (-1 auto, 0 never, 1 also for global-table gathers), two waves per chunk in the 512-thread lane kernel of a single
        registration (split: -1 auto, 0, 1)."""
        if workgroup is not None:
            self._seti(capi.NDT_WORKGROUP, workgroup, "setTuning(workgroup)")
        if table_mode is not None:
            self._seti(capi.NDT_TABLE_MODE, table_mode, "setTuning(table_mode)")
        if grid_builder is not None:
            self._seti(capi.GRID_BUILDER, grid_builder, "setTuning(grid_builder)")
        if wait_mode is not None:
            self._seti(capi.WAIT_MODE, wait_mode, "setTuning(wait_mode)")
        if quad is not None:
            self._seti(capi.NDT_QUAD, quad, "setTuning(quad)")
        if sort is not None:
            self._seti(capi.NDT_SORT, sort, "setTuning(sort)")
        if split is not None:
            self._seti(capi.NDT_SPLIT, split, "setTuning(split)")

    def setProfiling(self, on: bool):
        self._seti(capi.PROFILE, 1 if on else 0, "setProfiling")

    def getProfile(self, reset: bool = False) -> dict:
        p = capi.Profile()
        capi.check(self._lib.lsr_get_profile(self._h, C.byref(p), 1 if reset else 0), "getProfile")
        return dict(deriv_ms_total=p.deriv_ms_total, deriv_launches=p.deriv_launches, deriv_points=p.deriv_points,
                    deriv_pairs=p.deriv_pairs)


class NormalDistributionsTransform(Registration):
    """pclomp::NormalDistributionsTransform<PointXYZI,PointXYZI> (scanmatcher_component.cpp:105-113)."""

    _method = capi.METHOD_NDT

    def setResolution(self, res: float):  # scanmatcher_component.cpp:107
        self._setf(capi.RESOLUTION, res, "setResolution")

    def getResolution(self) -> float:
        return self._getf(capi.RESOLUTION)

    def setStepSize(self, s: float):
        self._setf(capi.STEP_SIZE, s, "setStepSize")

    def getStepSize(self) -> float:
        return self._getf(capi.STEP_SIZE)

    def setOulierRatio(self, r: float):  # (sic) PCL spells it this way
        self._setf(capi.OUTLIER_RATIO, r, "setOulierRatio")

    setOutlierRatio = setOulierRatio

    def setNeighborhoodSearchMethod(self, method: int):  # scanmatcher_component.cpp:110
        self._seti(capi.NEIGHBORHOOD, method, "setNeighborhoodSearchMethod")

    def setNumThreads(self, n: int):  # scanmatcher_component.cpp:111 — CPU hint, accepted and ignored
        self._seti(capi.NUM_THREADS, n, "setNumThreads")

    def setHessianD1Sign(self, sign: int):
        self._seti(capi.HESSIAN_D1_SIGN, sign, "setHessianD1Sign")

    def getTransformationProbability(self) -> float:
        return float(self._last.score)

    # inspection ----------------------------------------------------------------------------
    def gridInfo(self) -> dict:
        info = np.zeros(8, np.int32)
        capi.check(self._lib.lsr_ndt_grid_info(self._h, info.ctypes.data_as(C.POINTER(C.c_int32))), "gridInfo")
        return dict(min_b=info[0:3].copy(), max_b=info[3:6].copy(), n_leaves=int(info[6]), n_valid=int(info[7]))

    def gridDump(self) -> dict:
        n = self.gridInfo()["n_leaves"]
        idx, npts = np.zeros(n, np.int32), np.zeros(n, np.int32)
        mean, icov = np.zeros((n, 3)), np.zeros((n, 3, 3))
        capi.check(self._lib.lsr_ndt_grid_dump(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                               npts.ctypes.data_as(C.POINTER(C.c_int32)),
                                               mean.ctypes.data_as(C.POINTER(C.c_double)),
                                               icov.ctypes.data_as(C.POINTER(C.c_double))), "gridDump")
        return dict(idx=idx, n=npts, mean=mean, icov=icov)

    def gridCentroids(self) -> np.ndarray:
        """(n_leaves, 3) fp32, in gridDump's order: the leaves' FLOAT centroids (the points of the voxel-centroid kd-tree that the
        KDTREE neighbourhood searches); NaN for leaves outside the kd-tree (fewer than 6 points)."""
        c = np.zeros((self.gridInfo()["n_leaves"], 3), np.float32)
        capi.check(self._lib.lsr_ndt_grid_centroids(self._h, c.ctypes.data_as(C.POINTER(C.c_float))), "gridCentroids")
        return c

    def derivatives(self, p, T=None, compute_hessian: bool = True):
        p = np.ascontiguousarray(p, np.float64)
        t = _mat_to_col16(T) if T is not None else None
        score = C.c_double()
        g, H = np.zeros(6), np.zeros((6, 6))
        capi.check(self._lib.lsr_ndt_derivatives(self._h, p.ctypes.data_as(C.POINTER(C.c_double)),
                                                 t.ctypes.data_as(C.POINTER(C.c_float)) if t is not None else None,
                                                 1 if compute_hessian else 0, C.byref(score),
                                                 g.ctypes.data_as(C.POINTER(C.c_double)),
                                                 H.ctypes.data_as(C.POINTER(C.c_double))), "derivatives")
        return score.value, g, H


class GeneralizedIterativeClosestPoint(Registration):
    """pclomp::GeneralizedIterativeClosestPoint<PointXYZI,PointXYZI> (scanmatcher_component.cpp:115-120)."""

    _method = capi.METHOD_GICP

    def setRotationEpsilon(self, eps: float):
        self._setf(capi.ROTATION_EPSILON, eps, "setRotationEpsilon")

    def setCorrespondenceRandomness(self, k: int):
        self._seti(capi.K_CORRESPONDENCES, k, "setCorrespondenceRandomness")

    def setMaximumOptimizerIterations(self, n: int):
        self._seti(capi.MAX_INNER_ITERATIONS, n, "setMaximumOptimizerIterations")

    def covariances(self, which: str, raw: bool = False) -> np.ndarray:
        """Per-point covariances; raw=True: the k-neighbour sample covariance before the eigen-regularisation."""
        w = (0 if which == "source" else 1) + (2 if raw else 0)
        n = self._n_source if w in (0, 2) else self._n_target
        cov = np.zeros((n, 3, 3))
        capi.check(self._lib.lsr_gicp_covariances(self._h, w, cov.ctypes.data_as(C.POINTER(C.c_double))), "covariances")
        return cov

    def setInputTarget(self, cloud):
        super().setInputTarget(cloud)
        self._n_target = int(np.shape(cloud)[0]) if not _is_torch_cuda(cloud) else int(cloud.shape[0])


def set_input_target_batch(regs: Sequence[Registration], clouds):
    """setInputTarget of every candidate of a set with the builds overlapped on the device (lsr_set_input_target_batch;
    graph_based_slam_component.cpp:181-227 per candidate).  regs[b] receives clouds[b]; all host arrays or all CUDA tensors."""
    lib = capi.load()
    B = len(regs)
    if len(clouds) != B:
        raise ValueError("one cloud per registration object")
    args = [_cloud_args(c, r) for r, c in zip(regs, clouds)]
    if B and (any(a[3] != args[0][3] for a in args) or any(a[1] != args[0][1] for a in args)):
        raise ValueError("clouds must all be host or all device, with one record stride")
    hs = (C.c_void_p * B)(*[r._h for r in regs])
    ptrs = (C.c_void_p * B)(*[a[0] for a in args])
    counts = (C.c_size_t * B)(*[a[2] for a in args])
    capi.check(lib.lsr_set_input_target_batch(hs, B, ptrs, counts, args[0][1] if B else 12, 1 if (B and args[0][3]) else 0),
               "set_input_target_batch")
    for r, a in zip(regs, args):
        r._keep["target"] = None
        if hasattr(r, "_n_target"):
            r._n_target = a[2]


def set_input_source_batch(regs: Sequence[Registration], clouds):
    """setInputSource of every candidate of a set in shared launches (lsr_set_input_source_batch)."""
    lib = capi.load()
    B = len(regs)
    if len(clouds) != B:
        raise ValueError("one cloud per registration object")
    args = [_cloud_args(c, r) for r, c in zip(regs, clouds)]
    if B and (any(a[3] != args[0][3] for a in args) or any(a[1] != args[0][1] for a in args)):
        raise ValueError("clouds must all be host or all device, with one record stride")
    hs = (C.c_void_p * B)(*[r._h for r in regs])
    ptrs = (C.c_void_p * B)(*[a[0] for a in args])
    counts = (C.c_size_t * B)(*[a[2] for a in args])
    capi.check(lib.lsr_set_input_source_batch(hs, B, ptrs, counts, args[0][1] if B else 12, 1 if (B and args[0][3]) else 0),
               "set_input_source_batch")
    for r, a in zip(regs, args):
        r._n_source = a[2]
        r._keep["source"] = a[4] if a[3] else None


def fitness_score_batch(regs: Sequence[Registration], max_range: float = 1.7976931348623157e308):
    """getFitnessScore of every candidate of a set, all searches enqueued before the first wait (lsr_get_fitness_score_batch)."""
    lib = capi.load()
    B = len(regs)
    hs = (C.c_void_p * B)(*[r._h for r in regs])
    out = (C.c_double * max(B, 1))()
    capi.check(lib.lsr_get_fitness_score_batch(hs, B, float(max_range), out), "fitness_score_batch")
    return [float(out[b]) for b in range(B)]


def align_batch(regs: Sequence[Registration], guesses=None):
    """Advance B registrations together in shared launches (BASELINE.json cfg 4).  Returns
    (finals (B,4,4) fp32, list of result dicts)."""
    lib = capi.load()
    B = len(regs)
    hs = (C.c_void_p * B)(*[r._h for r in regs])
    g = None
    if guesses is not None:
        g = np.ascontiguousarray(np.stack([_mat_to_col16(x) for x in guesses]), np.float32)
    fin = np.zeros((B, 16), np.float32)
    res = (capi.Result * B)()
    capi.check(lib.lsr_align_batch(hs, B, g.ctypes.data_as(C.POINTER(C.c_float)) if g is not None else None,
                                   fin.ctypes.data_as(C.POINTER(C.c_float)), res), "align_batch")
    finals = np.stack([_col16_to_mat(fin[b]) for b in range(B)])
    out = []
    for b, r in enumerate(regs):
        r._last = capi.Result.from_buffer_copy(res[b])
        out.append(r.last_result)
    return finals, out


def align_fitness_batch(regs: Sequence[Registration], guesses=None, max_range: float = 1.7976931348623157e308):
    """align() followed by getFitnessScore() for every candidate of a set in one call (graph_based_slam_component.cpp:230-231):
    returns (finals (B,4,4) fp32, list of result dicts, fitness scores (B,) fp64) — what align_batch + fitness_score_batch return,
    with the searches of the candidates that finish early running under the launch chain of the others."""
    lib = capi.load()
    B = len(regs)
    hs = (C.c_void_p * B)(*[r._h for r in regs])
    g = None
    if guesses is not None:
        g = np.ascontiguousarray(np.stack([_mat_to_col16(x) for x in guesses]), np.float32)
    fin = np.zeros((B, 16), np.float32)
    res = (capi.Result * B)()
    fit = np.zeros(B, np.float64)
    capi.check(lib.lsr_align_fitness_batch(hs, B, g.ctypes.data_as(C.POINTER(C.c_float)) if g is not None else None,
                                           fin.ctypes.data_as(C.POINTER(C.c_float)), res, C.c_double(max_range),
                                           fit.ctypes.data_as(C.POINTER(C.c_double))), "align_fitness_batch")
    finals = np.stack([_col16_to_mat(fin[b]) for b in range(B)])
    out = []
    for b, r in enumerate(regs):
        r._last = capi.Result.from_buffer_copy(res[b])
        out.append(r.last_result)
    return finals, out, fit
