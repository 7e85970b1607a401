"""Loop-closure gate: the compute half of GraphBasedSlamComponent::searchLoop()
(graph_based_slam_component.cpp:164-252) behind `lsr_search_loop`.

`SubMap` mirrors lidarslam_msgs/msg/SubMap (distance, pose, cloud); `LoopClosureParams` the node parameters
searchLoop() reads (graph_based_slam_component.cpp:23-38, same names and defaults).  `search_loop` returns the
`LoopEdge`s the reference would push into `loop_edges_` (accepted ones) plus the rejected evaluations; the pose
graph optimisation that follows (doPoseAdjustment, g2o) is not part of the hot path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

from . import _capi
from .registration import Registration, _cloud_args, _is_torch_cuda, _order_after_torch


@dataclass
class SubMap:
    """lidarslam_msgs/msg/SubMap.msg: `distance`, `pose` (position + quaternion x,y,z,w), `cloud` (pose-local xyz)."""
    cloud: object  # (n, >=3) float32 numpy array or CUDA tensor
    position: Sequence[float]
    orientation: Sequence[float] = (0.0, 0.0, 0.0, 1.0)
    distance: float = 0.0


@dataclass
class LoopClosureParams:
    threshold_loop_closure_score: float = 1.0       # graph_based_slam_component.cpp:31
    distance_loop_closure: float = 20.0             # :33
    range_of_searching_loop_closure: float = 20.0   # :35
    search_submap_num: int = 3                      # :37
    voxel_leaf_size: float = 0.2                    # :23
    top_k: int = 1                                  # 1 = the reference (nearest candidate only)


@dataclass
class LoopEdge:
    pair_id: tuple                # (candidate index, num_submaps - 1)   :240
    relative_pose: np.ndarray     # 4x4 fp64: from^-1 * (final * init)   :241-245
    fitness_score: float          # :231
    accepted: bool                # fitness_score < threshold            :233
    final_transformation: np.ndarray = field(repr=False, default=None)
    converged: bool = False
    iterations: int = 0
    n_target_points: int = 0
    candidate_distance: float = 0.0


def search_loop(registration: Registration, submaps: Sequence[SubMap], params: LoopClosureParams = LoopClosureParams()) -> List[LoopEdge]:
    """All candidate evaluations, nearest candidate first (empty list = no candidate passed the distance gates)."""
    lib = _capi.load()
    n = len(submaps)
    if n == 0:
        return []
    arr = (_capi.SubMap * n)()
    keep = []
    on_device, stride = None, None
    for i, sm in enumerate(submaps):
        ptr, st, cnt, dev, holder = _cloud_args(sm.cloud)
        keep.append(holder)
        if on_device is None:
            on_device, stride = dev, st
        elif cnt and (dev != on_device or st != stride):
            raise ValueError("all submap clouds must share residency (host/device) and point stride")
        arr[i].position[:] = [float(v) for v in sm.position]
        arr[i].orientation[:] = [float(v) for v in sm.orientation]
        arr[i].distance = float(sm.distance)
        arr[i].cloud = ptr.value
        arr[i].n_points = cnt
    # ONE ordering of the handle's stream after torch's current stream, after every .contiguous() copy above has been enqueued
    # there (an event per submap was 21 x (event record + stream wait) = 0.1 ms of a 0.55 ms call)
    dev_holders = [h for h in keep if _is_torch_cuda(h)]
    if dev_holders:
        _order_after_torch(registration, dev_holders[-1])
    cp = _capi.LoopParams(params.threshold_loop_closure_score, params.distance_loop_closure,
                          params.range_of_searching_loop_closure, params.search_submap_num, params.voxel_leaf_size,
                          params.top_k, 0)
    cap = max(1, params.top_k)
    edges = (_capi.LoopEdge * cap)()
    n_eval = C.c_int32(0)
    _capi.check(lib.lsr_search_loop(registration._h, arr, n, stride, int(bool(on_device)), C.byref(cp), edges, cap,
                                    C.byref(n_eval)), "searchLoop")
    if n_eval.value and hasattr(registration, "_n_target"):
        registration._n_target = int(edges[0].n_target_points)   # the object now holds the nearest candidate's window as its target
    out = []
    for e in edges[:n_eval.value]:
        out.append(LoopEdge(pair_id=(e.id_from, e.id_to),
                            relative_pose=np.array(e.relative_pose[:], np.float64).reshape(4, 4, order="F"),
                            fitness_score=e.fitness_score, accepted=bool(e.accepted),
                            final_transformation=np.array(e.final_transformation[:], np.float32).reshape(4, 4, order="F"),
                            converged=bool(e.converged), iterations=e.iterations, n_target_points=e.n_target_points,
                            candidate_distance=e.candidate_distance))
    return out
