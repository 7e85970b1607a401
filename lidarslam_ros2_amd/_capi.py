"""ctypes binding of liblidarslam_reg.so (include/lidarslam_reg.h).

The shared library is built in-tree by `__graft_entry__.build()` / `lidarslam_ros2_amd.build`.
There is no Python or CPU fallback: if the library is missing or no gfx950 device is visible,
the first compute call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("LSR_LIB_NAME", "liblidarslam_reg.so"))

# enums (mirror include/lidarslam_reg.h)
OK = 0
METHOD_NDT, METHOD_GICP = 0, 1
KDTREE, DIRECT26, DIRECT7, DIRECT1 = 0, 1, 2, 3
(RESOLUTION, TRANSFORMATION_EPSILON, STEP_SIZE, OUTLIER_RATIO, MAX_CORRESPONDENCE_DISTANCE, ROTATION_EPSILON,
 EUCLIDEAN_FITNESS_EPSILON, GICP_EPSILON) = range(8)
(MAX_ITERATIONS, NEIGHBORHOOD, NUM_THREADS, K_CORRESPONDENCES, MAX_INNER_ITERATIONS, RANSAC_ITERATIONS,
 HESSIAN_D1_SIGN, PROFILE, NDT_WORKGROUP, NDT_TABLE_MODE, GRID_BUILDER, WAIT_MODE, NDT_QUAD, NDT_SORT, VOXEL_FILTER_FORM, NDT_SPLIT) = range(32, 48)

EXPORTED_SYMBOLS = [
    "lsr_version", "lsr_status_string", "lsr_last_error", "lsr_device_count", "lsr_create", "lsr_destroy",
    "lsr_set_f64", "lsr_set_i32", "lsr_get_f64", "lsr_get_i32", "lsr_set_input_target", "lsr_set_input_target_device",
    "lsr_set_input_target_frames", "lsr_set_input_source", "lsr_set_input_source_device", "lsr_set_input_source_filtered", "lsr_set_input_source_frontend", "lsr_voxel_grid_filter",
    "lsr_share_target", "lsr_wait_stream", "lsr_align", "lsr_align_batch",
    "lsr_get_final_transformation", "lsr_has_converged", "lsr_get_fitness_score", "lsr_search_loop", "lsr_ndt_grid_info",
    "lsr_ndt_grid_dump", "lsr_ndt_grid_centroids", "lsr_ndt_derivatives", "lsr_gicp_covariances", "lsr_nearest_neighbors", "lsr_get_profile",
    "lsr_debug_angle_tables", "lsr_set_input_source_pc2", "lsr_get_source_pc2", "lsr_voxel_grid_filter_pc2", "lsr_shard_range", "lsr_comm_unique_id", "lsr_comm_create", "lsr_comm_destroy", "lsr_align_batch_sharded",
    "lsr_shard_plan", "lsr_align_batch_planned", "lsr_align_fitness_batch",
    "lsr_set_input_target_batch", "lsr_set_input_source_batch", "lsr_get_fitness_score_batch", "lsr_set_input_target_bcast", "lsr_get_source_pc2_device",
    "lsr_comm_all_gather_records",
]


class Result(C.Structure):
    _fields_ = [("converged", C.c_int32), ("iterations", C.c_int32), ("score", C.c_double),
                ("n_evaluations", C.c_int32), ("n_correspondences", C.c_int32), ("gpu_ms", C.c_double)]


class Profile(C.Structure):
    _fields_ = [("deriv_ms_total", C.c_double), ("deriv_launches", C.c_int64), ("deriv_points", C.c_int64),
                ("deriv_pairs", C.c_int64)]


class SubMap(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("orientation", C.c_double * 4), ("distance", C.c_double),
                ("cloud", C.c_void_p), ("n_points", C.c_size_t)]


class LoopParams(C.Structure):
    _fields_ = [("threshold_loop_closure_score", C.c_double), ("distance_loop_closure", C.c_double),
                ("range_of_searching_loop_closure", C.c_double), ("search_submap_num", C.c_int),
                ("voxel_leaf_size", C.c_float), ("top_k", C.c_int), ("reserved", C.c_int)]


class LoopEdge(C.Structure):
    _fields_ = [("id_from", C.c_int), ("id_to", C.c_int), ("accepted", C.c_int), ("converged", C.c_int),
                ("iterations", C.c_int), ("n_target_points", C.c_int), ("candidate_distance", C.c_double),
                ("fitness_score", C.c_double), ("relative_pose", C.c_double * 16), ("final_transformation", C.c_float * 16)]


class Pc2Layout(C.Structure):
    _fields_ = [("point_step", C.c_uint32), ("offset_x", C.c_uint32), ("offset_y", C.c_uint32), ("offset_z", C.c_uint32),
                ("offset_intensity", C.c_int32)]


class ShardRecord(C.Structure):
    _fields_ = [("T", C.c_float * 12), ("score", C.c_float), ("iterations", C.c_float), ("converged", C.c_float),
                ("fitness", C.c_float)]


class RegistrationError(RuntimeError):
    def __init__(self, status: int, where: str):
        lib = load()
        msg = lib.lsr_status_string(status).decode()
        detail = lib.lsr_last_error().decode()
        super().__init__(f"{where}: {msg} ({status}){': ' + detail if detail else ''}")
        self.status = status


_lib = None


def load() -> C.CDLL:
    """Load the HIP library; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). lidarslam_ros2_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, fp, dp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.lsr_version.restype = C.c_char_p
    L.lsr_status_string.restype = C.c_char_p
    L.lsr_status_string.argtypes = [C.c_int]
    L.lsr_last_error.restype = C.c_char_p
    L.lsr_device_count.argtypes = [ip]
    L.lsr_create.argtypes = [C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.lsr_destroy.argtypes = [vp]
    L.lsr_set_f64.argtypes = [vp, C.c_int, C.c_double]
    L.lsr_set_i32.argtypes = [vp, C.c_int, C.c_int]
    L.lsr_get_f64.argtypes = [vp, C.c_int, dp]
    L.lsr_get_i32.argtypes = [vp, C.c_int, ip]
    for name in ("lsr_set_input_target", "lsr_set_input_target_device", "lsr_set_input_source",
                 "lsr_set_input_source_device"):
        getattr(L, name).argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    L.lsr_share_target.argtypes = [vp, vp]
    L.lsr_wait_stream.argtypes = [vp, vp]
    L.lsr_set_input_target_frames.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, fp, C.c_int]
    L.lsr_set_input_source_filtered.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_float, C.c_int, C.POINTER(C.c_size_t)]
    L.lsr_set_input_source_frontend.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_float, C.c_int,
                                                C.POINTER(C.c_size_t)]
    L.lsr_voxel_grid_filter.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_float, vp, C.c_size_t, C.c_size_t,
                                        C.POINTER(C.c_size_t)]
    L.lsr_align.argtypes = [vp, fp, fp, C.POINTER(Result), vp, C.c_size_t]
    L.lsr_align_batch.argtypes = [C.POINTER(vp), C.c_int, fp, fp, C.POINTER(Result)]
    L.lsr_get_final_transformation.argtypes = [vp, fp]
    L.lsr_has_converged.argtypes = [vp, ip]
    L.lsr_get_fitness_score.argtypes = [vp, C.c_double, dp]
    L.lsr_get_fitness_score_batch.argtypes = [C.POINTER(vp), C.c_int, C.c_double, dp]
    L.lsr_set_input_target_batch.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, C.c_int]
    L.lsr_set_input_source_batch.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, C.c_int]
    L.lsr_search_loop.argtypes = [vp, C.POINTER(SubMap), C.c_int, C.c_size_t, C.c_int, C.POINTER(LoopParams),
                                  C.POINTER(LoopEdge), C.c_int, ip]
    L.lsr_ndt_grid_info.argtypes = [vp, ip]
    L.lsr_ndt_grid_dump.argtypes = [vp, ip, ip, dp, dp]
    L.lsr_ndt_grid_centroids.argtypes = [vp, fp]
    L.lsr_ndt_derivatives.argtypes = [vp, dp, fp, C.c_int, dp, dp, dp]
    L.lsr_gicp_covariances.argtypes = [vp, C.c_int, dp]
    L.lsr_nearest_neighbors.argtypes = [vp, fp, ip, fp]
    L.lsr_get_profile.argtypes = [vp, C.POINTER(Profile), C.c_int]
    L.lsr_debug_angle_tables.argtypes = [dp, C.c_int, fp, fp, fp, fp]
    L.lsr_set_input_source_pc2.argtypes = [vp, vp, C.c_size_t, C.POINTER(Pc2Layout), C.c_double, C.c_double, C.c_float, C.c_int,
                                           C.POINTER(C.c_size_t)]
    L.lsr_get_source_pc2.argtypes = [vp, vp, C.c_size_t, C.POINTER(Pc2Layout), C.POINTER(C.c_size_t)]
    L.lsr_get_source_pc2_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(Pc2Layout), C.POINTER(C.c_size_t)]
    L.lsr_voxel_grid_filter_pc2.argtypes = [vp, vp, C.c_size_t, C.POINTER(Pc2Layout), C.c_float, vp, C.c_size_t, C.POINTER(Pc2Layout),
                                            C.POINTER(C.c_size_t)]
    L.lsr_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip]
    L.lsr_shard_range.restype = None
    L.lsr_comm_unique_id.argtypes = [vp]
    L.lsr_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.lsr_comm_destroy.argtypes = [vp]
    L.lsr_align_batch_sharded.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, fp, C.c_int, C.POINTER(ShardRecord)]
    L.lsr_set_input_target_bcast.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
    L.lsr_align_fitness_batch.argtypes = [C.POINTER(vp), C.c_int, fp, fp, C.POINTER(Result), C.c_double, dp]
    i32p = C.POINTER(C.c_int32)
    L.lsr_shard_plan.argtypes = [C.c_int, dp, C.c_int, i32p, i32p, i32p]
    L.lsr_align_batch_planned.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, i32p, i32p, fp, C.c_int, C.POINTER(ShardRecord)]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int or name not in ("lsr_version", "lsr_status_string", "lsr_last_error"):
            if name not in ("lsr_version", "lsr_status_string", "lsr_last_error", "lsr_shard_range"):
                fn.restype = C.c_int
    _lib = L
    return L


def check(status: int, where: str) -> None:
    if status != OK:
        raise RegistrationError(status, where)
