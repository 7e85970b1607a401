"""The frontend's per-scan sequence, host side: what ScanMatcherComponent::receiveCloud and ::updateMap do around the
registration object (scanmatcher/src/scanmatcher_component.cpp:296-356 and :436-481), written against the method names of
`lidarslam_ros2_amd.registration` so that the same loop drives the gfx950 core (bench.py `frontend_stream`,
tests/test_frontend_stream_gpu.py) and — through an adapter with the same methods — the CPU oracle the tests compare with.

    per scan (receiveCloud)      raw PointCloud2 payload -> range filter (:210-218) -> VoxelGrid(vg_size_for_input) (:324-328) ->
                                 setInputSource (:329) -> align(previous pose) (:353) -> getFinalTransformation (:356)
    every trans_for_mapupdate m  (updateMap) VoxelGrid(vg_size_for_map) of the scan (:442-446), kept with the registered pose as the newest
                                 submap (:466-478); target = that scan + the num_targeted_cloud - 1 submaps before it, each moved by its
                                 pose and concatenated (:448-464); setInputTarget at the start of the next callback (:304-307)

Nothing here computes: every step is one call into the registration object.  The reference runs updateMap on a worker thread and
applies the new target at the next callback; a replay has no second thread, so the update is timed on its own and reported next to
the per-scan latency (bench.py)."""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

PC2_XYZI = (32, (0, 4, 8, 16))   # pcl::PointXYZI as pcl::toROSMsg lays it out: point_step 32, x@0 y@4 z@8 intensity@16


def as_pc2_payload(xyz: np.ndarray, intensity: Optional[np.ndarray] = None) -> np.ndarray:
    """(n,3) fp32 -> (n,32) uint8 PointCloud2 payload in pcl::PointXYZI's layout."""
    n = int(xyz.shape[0])
    rec = np.zeros((n, 8), np.float32)
    rec[:, :3] = xyz
    if intensity is not None:
        rec[:, 4] = intensity
    return rec.view(np.uint8).reshape(n, 32)


def _records(payload: np.ndarray) -> np.ndarray:
    """(m,32) uint8 PointXYZI payload -> the same bytes as (m,8) fp32 records (what setInputTargetFrames takes)."""
    return np.ascontiguousarray(payload).view(np.float32).reshape(-1, 8)


@dataclass
class FrontendParams:   # scanmatcher_component.cpp:36-60 (declare_parameter defaults; vg sizes as BASELINE cfg 1/2 uses them)
    vg_size_for_input: float = 0.2
    vg_size_for_map: float = 0.1
    trans_for_mapupdate: float = 1.5
    scan_min_range: float = 0.1
    scan_max_range: float = 100.0
    num_targeted_cloud: int = 10


@dataclass
class FrontendResult:
    poses: List[np.ndarray] = field(default_factory=list)          # registered pose of every scan (4x4)
    iterations: List[int] = field(default_factory=list)
    points_kept: List[int] = field(default_factory=list)
    scan_seconds: List[float] = field(default_factory=list)         # scan in -> pose out (preprocess + setInputSource + align)
    update_seconds: List[float] = field(default_factory=list)       # map updates: VoxelGrid(map) + assembly + setInputTarget
    update_at: List[int] = field(default_factory=list)              # index of the scan after which each update ran


class FrontendReplay:
    """`reg`: a registration object (lidarslam_ros2_amd.NormalDistributionsTransform or an adapter with the same methods).
    `to_device` (optional): maps a kept submap — (m,8) fp32 pcl::PointXYZI records on the host — to what `reg.setInputTargetFrames`
    should be given (e.g. a CUDA tensor, so that the keyframes stay resident in HBM); identity when None."""

    def __init__(self, reg, params: FrontendParams | None = None, to_device=None, mapper=None):
        self.reg = reg
        self.p = params or FrontendParams()
        self.to_device = to_device or (lambda a: a)
        # `mapper` (optional): a second registration object whose input-source slot serves as the map side's filter — with it and a
        # device-resident payload the new keyframe (range filter + VoxelGrid(vg_size_for_map)) is produced and kept in HBM
        self.mapper = mapper
        self.submaps: list = []          # [(payload as given to setInputTargetFrames, pose 4x4 f64)]
        self.pose = np.eye(4)
        self.key_position = np.zeros(3)

    def initialise(self, frames_xyz, frame_poses, pose0):
        """The map the drive starts from: the keyframes so far (sensor frame, already VoxelGrid(vg_size_for_map)-filtered) and their poses."""
        self.submaps = [(self.to_device(_records(as_pc2_payload(f))), np.asarray(P, np.float64)) for f, P in zip(frames_xyz, frame_poses)]
        self.submaps = self.submaps[-self.p.num_targeted_cloud:]
        self._set_target()
        self.pose = np.asarray(pose0, np.float64)
        self.key_position = np.asarray(frame_poses[-1], np.float64)[:3, 3].copy()

    def _set_target(self):
        # newest first, as updateMap concatenates (:448-464); the voxel grid does not depend on the order
        window = self.submaps[-self.p.num_targeted_cloud:][::-1]
        self.reg.setInputTargetFrames([w[0] for w in window], [w[1] for w in window])

    def receive_cloud(self, payload, n_points: int, out: FrontendResult, payload_host=None):
        """One LiDAR message.  `payload`: the raw PointCloud2 data (host array or CUDA tensor); `payload_host`: a host copy for the map
        update (the reference's callback holds the cloud on the host anyway); defaults to `payload`."""
        step, offs = PC2_XYZI
        t0 = time.perf_counter()
        kept = self.reg.setInputSourcePointCloud2(payload, n_points, step, offs, self.p.scan_min_range, self.p.scan_max_range,
                                                  self.p.vg_size_for_input)
        self.reg.align(self.pose.astype(np.float32))
        T = np.asarray(self.reg.getFinalTransformation(), np.float64)
        t1 = time.perf_counter()
        self.pose = T
        out.poses.append(T); out.points_kept.append(int(kept)); out.scan_seconds.append(t1 - t0)
        out.iterations.append(int(self.reg.getFinalNumIteration()))
        # displacement since the last map update (:412-424): trans_ >= trans_for_mapupdate_
        if float(np.linalg.norm(T[:3, 3] - self.key_position)) >= self.p.trans_for_mapupdate:
            t2 = time.perf_counter()
            if self.mapper is not None and hasattr(payload, "is_cuda") and payload.is_cuda:
                # the whole map side on the device: range filter + VoxelGrid(vg_size_for_map) into the mapper's source slot, from there
                # into a keyframe buffer in HBM (lsr_set_input_source_pc2 + lsr_get_source_pc2_device)
                self.mapper.setInputSourcePointCloud2(payload, n_points, step, offs, self.p.scan_min_range, self.p.scan_max_range,
                                                      self.p.vg_size_for_map)
                self.submaps.append((self.mapper.getInputSourceDeviceRecords(), T.copy()))
                self.submaps = self.submaps[-self.p.num_targeted_cloud:]
                self._set_target()
                self.key_position = T[:3, 3].copy()
                out.update_seconds.append(time.perf_counter() - t2)
                out.update_at.append(len(out.poses) - 1)
                return
            host = np.asarray(payload if payload_host is None else payload_host).reshape(n_points, step)
            # updateMap filters the cloud the callback received, i.e. AFTER the subscription's range filter (:210-218: horizontal range,
            # open interval, in double) — a host-side mask here, as in the reference
            xy = host[:, :8].copy().view(np.float32).astype(np.float64)
            r = np.sqrt(xy[:, 0] ** 2 + xy[:, 1] ** 2)
            ranged = np.ascontiguousarray(host[(self.p.scan_min_range < r) & (r < self.p.scan_max_range)])
            filtered = self.reg.voxelGridFilterPointCloud2(ranged, int(ranged.shape[0]), step, offs, self.p.vg_size_for_map)
            self.submaps.append((self.to_device(_records(filtered)), T.copy()))
            self.submaps = self.submaps[-self.p.num_targeted_cloud:]
            self._set_target()
            self.key_position = T[:3, 3].copy()
            out.update_seconds.append(time.perf_counter() - t2)
            out.update_at.append(len(out.poses) - 1)
