"""The frontend's per-scan sequence, host side: what ScanMatcherComponent::receiveCloud and ::updateMap do around the
registration object (scanmatcher/src/scanmatcher_component.cpp:296-356 and :436-481), written against the method names of
`lidarslam_ros2_amd.registration` so that the same loop drives the gfx950 core (bench.py `frontend_stream`,
tests/test_frontend_stream_gpu.py) and — through an adapter with the same methods — the CPU oracle the tests compare with.

    per scan (receiveCloud)      raw PointCloud2 payload -> range filter (:210-218) -> VoxelGrid(vg_size_for_input) (:324-328) ->
                                 setInputSource (:329) -> align(previous pose) (:353) -> getFinalTransformation (:356)
    every trans_for_mapupdate m  (updateMap) VoxelGrid(vg_size_for_map) of the scan (:442-446), kept with the registered pose as the newest
                                 submap (:466-478); target = that scan + the num_targeted_cloud - 1 submaps before it, each moved by its
                                 pose and concatenated (:448-464); setInputTarget at the start of the next callback (:304-307)

Nothing here computes: every step is one call into the registration object.

The reference runs updateMap on a worker thread (:427-434) and takes the new target over at the start of a later callback
(:298-320: `mapping_future_.wait_for(0s)` — whenever the thread happens to be done).  Two things model that here:
  * `async_update=True` + a `builder` object: the map side — filter of the new keyframe (`mapper`), assembly of the window AND the
    voxel-grid build (`builder.setInputTargetFrames`, on the builder's own stream) — runs on a worker thread while the callback thread
    registers the next scans; the hand-over is `reg.shareTargetOf(builder)` (lsr_share_target: a pointer swap, the old target goes back
    to the builder for recycling).  The reference builds the voxel grid INSIDE the callback (`registration_->setInputTarget`, :307);
    here the callback never builds a grid.
  * `swap_lag`: the reference's hand-over is racy (it depends on how long the thread took); a replay must be deterministic, so the
    lag is a parameter — the target of an update triggered by scan k is in place for scan k + 1 + swap_lag (0 = before the very next
    scan, which is what a 10 Hz sensor sees with a map side of a fraction of a millisecond; 1 = one scan later, i.e. the next scan is
    registered WHILE the map side runs).  The callback that is due waits for the worker (`swap_wait_seconds`).  While an update is
    pending no new one is triggered (`!mapping_flag_`, :427).  The serial replay with the same lag gives the same poses bit for bit.
"""
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
    scan_seconds: List[float] = field(default_factory=list)         # scan in -> pose out (hand-over of a due target + preprocess + setInputSource + align)
    update_seconds: List[float] = field(default_factory=list)       # map updates: VoxelGrid(map) + assembly + setInputTarget (on the worker thread when asynchronous)
    update_at: List[int] = field(default_factory=list)              # index of the scan after which each update was triggered
    swap_wait_seconds: List[float] = field(default_factory=list)    # asynchronous replay: what the due callback waited for the worker + the pointer swap
    swap_at: List[int] = field(default_factory=list)                # index of the scan whose callback took each new target over


class FrontendReplay:
    """`reg`: a registration object (lidarslam_ros2_amd.NormalDistributionsTransform or an adapter with the same methods).
    `to_device` (optional): maps a kept submap — (m,8) fp32 pcl::PointXYZI records on the host — to what `reg.setInputTargetFrames`
    should be given (e.g. a CUDA tensor, so that the keyframes stay resident in HBM); identity when None."""

    def __init__(self, reg, params: FrontendParams | None = None, to_device=None, mapper=None, builder=None, async_update: bool = False,
                 swap_lag: int = 0):
        self.reg = reg
        self.p = params or FrontendParams()
        self.to_device = to_device or (lambda a: a)
        # `mapper` (optional): a second registration object whose input-source slot serves as the map side's filter — with it and a
        # device-resident payload the new keyframe (range filter + VoxelGrid(vg_size_for_map)) is produced and kept in HBM
        self.mapper = mapper
        # `builder` (optional): the object that builds the targets (same method and resolution as `reg`, its own stream); `reg` takes
        # them over with shareTargetOf.  Without it `reg` builds its own targets.
        self.builder = builder
        self.async_update = bool(async_update)
        if self.async_update and builder is None:
            raise ValueError("an asynchronous map update needs a builder object (the callback's object must not build targets)")
        self.swap_lag = int(swap_lag)
        self._pool = None
        self._pending = None             # (due scan index, future or job, trigger time)
        self._n_scans = 0
        self.submaps: list = []          # [(payload as given to setInputTargetFrames, pose 4x4 f64)]
        self.pose = np.eye(4)
        self.key_position = np.zeros(3)

    def initialise(self, frames_xyz, frame_poses, pose0):
        """The map the drive starts from: the keyframes so far (sensor frame, already VoxelGrid(vg_size_for_map)-filtered) and their poses."""
        self.close()
        self.submaps = [(self.to_device(_records(as_pc2_payload(f))), np.asarray(P, np.float64)) for f, P in zip(frames_xyz, frame_poses)]
        self.submaps = self.submaps[-self.p.num_targeted_cloud:]
        self._set_target()
        self._hand_over()
        self.pose = np.asarray(pose0, np.float64)
        self.key_position = np.asarray(frame_poses[-1], np.float64)[:3, 3].copy()
        self._n_scans = 0

    def close(self):
        """Drains a pending update and stops the worker thread."""
        if self._pending is not None and hasattr(self._pending[1], "result"):
            self._pending[1].result()
        self._pending = None
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def _set_target(self):
        # newest first, as updateMap concatenates (:448-464); the voxel grid does not depend on the order
        window = self.submaps[-self.p.num_targeted_cloud:][::-1]
        (self.builder or self.reg).setInputTargetFrames([w[0] for w in window], [w[1] for w in window])

    def _hand_over(self):
        if self.builder is not None:
            self.reg.shareTargetOf(self.builder)

    def _update_job(self, payload, n_points, payload_host, T):
        """updateMap (:436-481) up to and including the target of the next scans; -> seconds."""
        step, offs = PC2_XYZI
        t2 = time.perf_counter()
        if self.mapper is not None and hasattr(payload, "is_cuda") and payload.is_cuda:
            # the whole map side on the device: range filter + VoxelGrid(vg_size_for_map) into the mapper's source slot, from there
            # into a keyframe buffer in HBM (lsr_set_input_source_pc2 + lsr_get_source_pc2_device)
            self.mapper.setInputSourcePointCloud2(payload, n_points, step, offs, self.p.scan_min_range, self.p.scan_max_range,
                                                  self.p.vg_size_for_map)
            keyframe = self.mapper.getInputSourceDeviceRecords()
        else:
            host = np.asarray(payload if payload_host is None else payload_host).reshape(n_points, step)
            # updateMap filters the cloud the callback received, i.e. AFTER the subscription's range filter (:210-218: horizontal range,
            # open interval, in double) — a host-side mask here, as in the reference
            xy = host[:, :8].copy().view(np.float32).astype(np.float64)
            r = np.sqrt(xy[:, 0] ** 2 + xy[:, 1] ** 2)
            ranged = np.ascontiguousarray(host[(self.p.scan_min_range < r) & (r < self.p.scan_max_range)])
            filtered = (self.builder or self.reg).voxelGridFilterPointCloud2(ranged, int(ranged.shape[0]), step, offs, self.p.vg_size_for_map)
            keyframe = self.to_device(_records(filtered))
        self.submaps.append((keyframe, T.copy()))
        self.submaps = self.submaps[-self.p.num_targeted_cloud:]
        self._set_target()
        return time.perf_counter() - t2

    def _settle(self, out: FrontendResult, force: bool = False):
        """Start of a callback (:298-320): a target that is due is taken over (the callback waits for the worker if it must)."""
        if self._pending is None:
            return
        due, job, _ = self._pending
        if not force and self._n_scans < due:
            return
        t0 = time.perf_counter()
        if hasattr(job, "result"):
            out.update_seconds.append(job.result())      # the worker's own clock
        else:
            out.update_seconds.append(job())             # serial replay: the update runs here, between two scans
            t0 = time.perf_counter()
        self._hand_over()
        out.swap_wait_seconds.append(time.perf_counter() - t0)
        out.swap_at.append(self._n_scans)
        self._pending = None

    def receive_cloud(self, payload, n_points: int, out: FrontendResult, payload_host=None):
        """One LiDAR message.  `payload`: the raw PointCloud2 data (host array or CUDA tensor); `payload_host`: a host copy for the map
        update (the reference's callback holds the cloud on the host anyway); defaults to `payload`."""
        step, offs = PC2_XYZI
        t0 = time.perf_counter()
        if self.async_update:
            self._settle(out)            # inside the scan's clock: what the callback really waits
        kept = self.reg.setInputSourcePointCloud2(payload, n_points, step, offs, self.p.scan_min_range, self.p.scan_max_range,
                                                  self.p.vg_size_for_input)
        self.reg.align(self.pose.astype(np.float32))
        T = np.asarray(self.reg.getFinalTransformation(), np.float64)
        t1 = time.perf_counter()
        self.pose = T
        out.poses.append(T); out.points_kept.append(int(kept)); out.scan_seconds.append(t1 - t0)
        out.iterations.append(int(self.reg.getFinalNumIteration()))
        self._n_scans += 1
        # displacement since the last map update (:412-427): trans_ >= trans_for_mapupdate_ && !mapping_flag_
        if self._pending is None and float(np.linalg.norm(T[:3, 3] - self.key_position)) >= self.p.trans_for_mapupdate:
            self.key_position = T[:3, 3].copy()
            out.update_at.append(len(out.poses) - 1)
            job = (lambda p=payload, n=n_points, h=payload_host, TT=T.copy(): self._update_job(p, n, h, TT))
            if self.async_update:
                if self._pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="lsr-map")
                job = self._pool.submit(job)
            self._pending = (self._n_scans + self.swap_lag, job, t1)
        if not self.async_update:
            self._settle(out)            # the serial replay does the due update between two scans, on its own clock (update_seconds)

    def finish(self, out: FrontendResult):
        """End of a drive: an update still pending is completed (its time is reported, nothing registers against it)."""
        self._settle(out, force=True)
        self.close()
