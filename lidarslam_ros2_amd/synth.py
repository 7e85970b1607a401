"""Deterministic synthetic LiDAR workloads for the registration hot path.

The reference ships no data (its demo bags are external downloads, /root/reference/README.md:
123-165), so the benchmark/test inputs are ray-cast here: a VLP-32-like or 64-line sensor
driving along +x through an analytic world (ground plane, axis-aligned boxes, vertical
cylinders).  The frontend's data flow is mirrored: every scan is voxel-downsampled
(`vg_size_for_map` for keyframes that go into the target submap,
scanmatcher_component.cpp:443-464; `vg_size_for_input` for the source scan,
scanmatcher_component.cpp:324-329), keyframes are `trans_for_mapupdate` = 1.5 m apart
(scanmatcher_component.cpp:34,423) and the guess is the previous scan's pose
(scanmatcher_component.cpp:331,353).

This module is workload generation only (numpy, host side); it is not on the product path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

WORLD_SEED = 20240924


@dataclass
class World:
    boxes: np.ndarray      # (B, 6) xmin,ymin,zmin,xmax,ymax,zmax
    cyls: np.ndarray       # (C, 4) cx, cy, r, ztop  (base at ground)
    ground_z: float = -1.8


def make_world(seed: int = WORLD_SEED, n_boxes: int = 70, n_cyls: int = 60, half_extent: float = 80.0,
               corridor: float = 6.0, x_shift: float = 0.0) -> World:
    """70 boxes (footprint U(3,25) m, height U(3,15) m) + 60 cylinders placed U(-80,80)^2 around
    (x_shift, 0), keeping a `corridor`-wide lane along +x free.  (SURVEY.md §8d proposed 40/30;
    densified, together with azimuth_oversample=3, so a VoxelGrid(0.2) scan keeps >= 30 000 and a
    64-line VoxelGrid(0.1) scan >= 120 000 points — the survey allows tuning, never padding.)"""
    rng = np.random.default_rng(seed)
    gz = -1.8
    boxes = []
    while len(boxes) < n_boxes:
        cx, cy = rng.uniform(-half_extent, half_extent, 2)
        sx, sy = rng.uniform(3.0, 25.0, 2)
        h = rng.uniform(3.0, 15.0)
        y0, y1 = cy - sy / 2, cy + sy / 2
        if y0 < corridor / 2 and y1 > -corridor / 2:
            continue
        boxes.append([cx - sx / 2 + x_shift, y0, gz, cx + sx / 2 + x_shift, y1, gz + h])
    cyls = []
    while len(cyls) < n_cyls:
        cx, cy = rng.uniform(-half_extent, half_extent, 2)
        r = rng.uniform(0.15, 0.5)
        h = rng.uniform(3.0, 8.0)
        if abs(cy) - r < corridor / 2:
            continue
        cyls.append([cx + x_shift, cy, r, gz + h])
    return World(np.asarray(boxes, np.float64), np.asarray(cyls, np.float64), gz)


@dataclass
class Sensor:
    n_beams: int
    elev_min_deg: float
    elev_max_deg: float
    n_azimuth: int
    range_noise: float = 0.02
    max_range: float = 100.0
    min_range: float = 0.5
    _dirs: np.ndarray | None = field(default=None, repr=False)

    def directions(self) -> np.ndarray:
        if self._dirs is None:
            el = np.deg2rad(np.linspace(self.elev_min_deg, self.elev_max_deg, self.n_beams))
            az = np.linspace(0.0, 2 * np.pi, self.n_azimuth, endpoint=False)
            ce, se = np.cos(el), np.sin(el)
            d = np.stack([np.outer(np.cos(az), ce), np.outer(np.sin(az), ce), np.outer(np.ones_like(az), se)], -1)
            self._dirs = d.reshape(-1, 3)
        return self._dirs


def vlp32() -> Sensor:
    return Sensor(32, -25.0, 15.0, 1800)


def hdl64() -> Sensor:
    return Sensor(64, -24.8, 2.0, 2048)


def pose_matrix(x: float, y: float, z: float, yaw: float, roll: float = 0.0, pitch: float = 0.0) -> np.ndarray:
    cr, sr, cp, sp, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rx @ Ry @ Rz
    T[:3, 3] = (x, y, z)
    return T


def trajectory_pose(x: float) -> np.ndarray:
    """Pose of the sensor after travelling x metres: straight +x, yaw = 0.02 sin(0.1 x)."""
    return pose_matrix(x, 0.0, 0.0, 0.02 * math.sin(0.1 * x))


_RAYCAST_C = None   # ctypes handle of synth_raycast.c (False: unavailable, numpy is used)


def _raycast_c():
    """The intersection loop compiled from synth_raycast.c (gcc -O2 -ffp-contract=off: the same IEEE double operations as the
    numpy code below, bit-identical results, ~50x faster).  Built next to this file on first use; any failure falls back to
    numpy silently — the clouds are the same either way.  LSR_SYNTH_NUMPY=1 forces the numpy path."""
    global _RAYCAST_C
    if _RAYCAST_C is not None:
        return _RAYCAST_C or None
    _RAYCAST_C = False
    import os
    if os.environ.get("LSR_SYNTH_NUMPY"):
        return None
    try:
        import ctypes as C
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        src, so = os.path.join(here, "synth_raycast.c"), os.path.join(here, "libsynth_raycast.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            tmp = so + ".%d.tmp" % os.getpid()
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", src, "-o", tmp, "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.replace(tmp, so)
        lib = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        lib.synth_raycast.argtypes = [dp, C.c_long, dp, C.c_double, dp, C.c_int, dp, C.c_int, dp]
        lib.synth_raycast.restype = None
        _RAYCAST_C = lib
    except Exception:
        _RAYCAST_C = False
    return _RAYCAST_C or None


def raycast_geometry(world: World, sensor: Sensor, T: np.ndarray):
    """The deterministic (and expensive) half of one revolution from pose T (sensor->map): ray/primitive intersections.
    Returns (keep, t) — the boolean mask of rays that hit something inside the sensor's range and their ranges.  Has no
    random state, so the scans of a workload can be intersected in parallel worker processes (make_case(pool=...))."""
    d_s = sensor.directions()
    R, o = T[:3, :3], T[:3, 3]
    d = d_s @ R.T
    n = d.shape[0]
    lib = _raycast_c()
    if lib is not None:
        import ctypes as C
        dp = C.POINTER(C.c_double)
        dc, oc = np.ascontiguousarray(d, np.float64), np.ascontiguousarray(o, np.float64)
        bx, cy = np.ascontiguousarray(world.boxes, np.float64), np.ascontiguousarray(world.cyls, np.float64)
        t_best = np.empty(n, np.float64)
        lib.synth_raycast(dc.ctypes.data_as(dp), n, oc.ctypes.data_as(dp), float(world.ground_z), bx.ctypes.data_as(dp), int(bx.shape[0]),
                          cy.ctypes.data_as(dp), int(cy.shape[0]), t_best.ctypes.data_as(dp))
        keep = np.isfinite(t_best) & (t_best < sensor.max_range) & (t_best > sensor.min_range)
        return keep, t_best[keep]
    t_best = np.full(n, np.inf)
    # ground
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (world.ground_z - o[2]) / d[:, 2]
    tg[~(tg > 0)] = np.inf
    t_best = np.minimum(t_best, tg)
    # boxes (slab method)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
    for b in world.boxes:
        t0 = (b[:3] - o) * inv
        t1 = (b[3:] - o) * inv
        tn = np.minimum(t0, t1).max(axis=1)
        tf = np.maximum(t0, t1).min(axis=1)
        hit = (tf >= tn) & (tf > 0)
        tt = np.where(tn > 0, tn, tf)
        tt[~hit] = np.inf
        t_best = np.minimum(t_best, tt)
    # cylinders (side surface only)
    a = d[:, 0] ** 2 + d[:, 1] ** 2
    for c in world.cyls:
        ox, oy = o[0] - c[0], o[1] - c[1]
        bq = ox * d[:, 0] + oy * d[:, 1]
        cq = ox * ox + oy * oy - c[2] ** 2
        disc = bq * bq - a * cq
        ok = disc > 0
        sq = np.sqrt(np.where(ok, disc, 0.0))
        with np.errstate(divide="ignore", invalid="ignore"):
            tt = (-bq - sq) / a
        z = o[2] + tt * d[:, 2]
        ok &= (tt > 0) & (z >= world.ground_z) & (z <= c[3])
        tt = np.where(ok, tt, np.inf)
        t_best = np.minimum(t_best, tt)
    keep = np.isfinite(t_best) & (t_best < sensor.max_range) & (t_best > sensor.min_range)
    return keep, t_best[keep]


def raycast_noise(sensor: Sensor, keep: np.ndarray, t: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """The random half: range noise and surface roughness, drawn from `rng` in a fixed order.  Sensor-frame fp32 (n,3)."""
    d_s = sensor.directions()
    rr = t + rng.normal(0.0, sensor.range_noise, int(keep.sum()))
    pts = d_s[keep] * rr[:, None]
    pts[:, 2] += rng.normal(0.0, 0.01, pts.shape[0])  # surface roughness
    return pts.astype(np.float32)


def raycast(world: World, sensor: Sensor, T: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """One revolution from pose T (sensor->map).  Returns hit points in the SENSOR frame, fp32 (n,3)."""
    keep, t = raycast_geometry(world, sensor, T)
    return raycast_noise(sensor, keep, t, rng)


def _geometry_job(args):
    world, sensor_args, T = args
    return raycast_geometry(world, Sensor(*sensor_args), T)


def raycast_many(world: World, sensor: Sensor, poses, rng: np.random.Generator, pool=None) -> list:
    """Scans from `poses`, in order, consuming `rng` exactly as successive raycast() calls would; the intersections run in
    `pool` (anything with a .map, e.g. multiprocessing.Pool) when one is given."""
    if pool is None:
        return [raycast(world, sensor, T, rng) for T in poses]
    sargs = (sensor.n_beams, sensor.elev_min_deg, sensor.elev_max_deg, sensor.n_azimuth, sensor.range_noise, sensor.max_range,
             sensor.min_range)
    geo = pool.map(_geometry_job, [(world, sargs, T) for T in poses])
    return [raycast_noise(sensor, keep, t, rng) for keep, t in geo]


def voxel_downsample(pts: np.ndarray, leaf: float) -> np.ndarray:
    """Host-side stand-in for pcl::VoxelGrid::filter (centroid per leaf, output ordered by
    leaf index, x fastest) used only to PREPARE workloads (scanmatcher_component.cpp:324-328)."""
    pts = np.asarray(pts, np.float32)
    if pts.shape[0] == 0:
        return pts.reshape(0, 3)
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(pts * inv).astype(np.int64)
    mn = ijk.min(axis=0)
    ijk -= mn
    div = ijk.max(axis=0) + 1
    key = ijk[:, 0] + div[0] * (ijk[:, 1] + div[1] * ijk[:, 2])
    uniq, inv_idx, cnt = np.unique(key, return_inverse=True, return_counts=True)
    out = np.zeros((uniq.shape[0], 3), np.float64)
    for k in range(3):
        out[:, k] = np.bincount(inv_idx, weights=pts[:, k].astype(np.float64), minlength=uniq.shape[0])
    out /= cnt[:, None]
    return out.astype(np.float32)


def transform_points(T: np.ndarray, pts: np.ndarray) -> np.ndarray:
    T = np.asarray(T, np.float32)
    return (pts @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


def subsample_exact(pts: np.ndarray, n: int, seed: int) -> np.ndarray:
    """Keep exactly n points (seeded choice, original order preserved — PCL's VoxelGrid output is
    ordered by leaf index, i.e. spatially coherent, and a real pipeline would not shuffle it)."""
    if pts.shape[0] < n:
        raise ValueError(f"scan has {pts.shape[0]} points after filtering, need >= {n}; tune the generator")
    rng = np.random.default_rng(seed)
    keep = np.sort(rng.permutation(pts.shape[0])[:n])
    return pts[keep]


@dataclass
class RegistrationCase:
    target: np.ndarray       # (M,3) fp32 map-frame submap
    source: np.ndarray       # (N,3) fp32 sensor-frame scan
    guess: np.ndarray        # (4,4) fp32
    truth: np.ndarray        # (4,4) fp64 ground-truth source->map pose
    name: str = ""
    # the parts the case was assembled from (keep_parts=True): what the frontend holds BEFORE its own preprocessing
    raw_source: "np.ndarray | None" = None      # the source scan as the sensor delivers it (no VoxelGrid, no sub-sampling)
    frames: "list | None" = None                # the keyframe clouds, VoxelGrid(vg_map)-filtered, each in its OWN sensor frame
    frame_poses: "list | None" = None           # their poses (4x4): target = concat(pose_k * frame_k)


def make_case(*, sensor: Sensor | None = None, world: World | None = None, n_keyframes: int = 10,
              start_x: float = 0.0, keyframe_spacing: float = 1.5, scan_spacing: float = 0.5,
              vg_map: float = 0.1, vg_input: float = 0.2, n_source: int | None = 30000,
              vg_target: float | None = None, guess_perturb: tuple | None = None, seed: int = 0,
              azimuth_oversample: int = 1, name: str = "", pool=None, keep_parts: bool = False) -> RegistrationCase:
    """Frontend-style case: target = n_keyframes scans (each VoxelGrid(vg_map), moved to the map
    frame, concatenated without re-filtering: scanmatcher_component.cpp:452-464); source = the
    next scan, VoxelGrid(vg_input) then cut to exactly n_source points; guess = pose of the
    previous scan (or truth perturbed by guess_perturb=(dx,dy,dyaw))."""
    sensor = sensor or vlp32()
    if azimuth_oversample != 1:
        sensor = Sensor(sensor.n_beams, sensor.elev_min_deg, sensor.elev_max_deg, sensor.n_azimuth * azimuth_oversample,
                        sensor.range_noise, sensor.max_range, sensor.min_range)
    world = world or make_world()
    rng = np.random.default_rng(WORLD_SEED + 7919 * seed + 1)
    x_last = start_x + keyframe_spacing * (n_keyframes - 1)
    x_src = x_last + scan_spacing
    T_src = trajectory_pose(x_src)
    poses = [trajectory_pose(start_x + keyframe_spacing * k) for k in range(n_keyframes)] + [T_src]
    scans = raycast_many(world, sensor, poses, rng, pool)   # `pool` only parallelises the intersections: same clouds
    chunks, frames = [], []
    for k in range(n_keyframes):
        frames.append(voxel_downsample(scans[k], vg_map))
        chunks.append(transform_points(poses[k], frames[k]))
    target = np.concatenate(chunks, 0)
    if vg_target is not None:  # GICP frontend path re-filters the target (scanmatcher_component.cpp:309-315)
        target = voxel_downsample(target, vg_target)
    src = voxel_downsample(scans[n_keyframes], vg_input)
    if n_source is not None:
        src = subsample_exact(src, n_source, seed=WORLD_SEED + seed)
    if guess_perturb is None:
        guess = trajectory_pose(x_last)
    else:
        dx, dy, dyaw = guess_perturb
        guess = pose_matrix(x_src + dx, dy, 0.0, 0.02 * math.sin(0.1 * x_src) + dyaw)
    if keep_parts:
        return RegistrationCase(target, src, guess.astype(np.float32), T_src, name, raw_source=scans[n_keyframes], frames=frames,
                                frame_poses=[np.asarray(P, np.float64) for P in poses[:n_keyframes]])
    return RegistrationCase(target, src, guess.astype(np.float32), T_src, name)


# ---- BASELINE.json configs ---------------------------------------------------------------
def cfg_ndt_30k(seed: int = 0, start_x: float = 0.0, guess_perturb=None, world: World | None = None, pool=None,
                keep_parts: bool = False) -> RegistrationCase:
    """cfg 1/2: 30k-pt VLP-32 scan (vg 0.2) vs 10-frame submap (vg 0.1)."""
    return make_case(sensor=vlp32(), n_keyframes=10, vg_map=0.1, vg_input=0.2, n_source=30000, seed=seed,
                     start_x=start_x, guess_perturb=guess_perturb, world=world, azimuth_oversample=3,
                     name="ndt_30k_vs_10frame", pool=pool, keep_parts=keep_parts)


def cfg_scan_stream(n_scans: int, seed: int = 0, pool=None, world: World | None = None) -> list:
    """A stream of DIFFERENT cfg-1/2 scans against the cfg_ndt_30k(seed) submap: scan j is taken 0.5, 1.0 or 1.5 m past
    the last keyframe (the frontend registers about three scans per map update, trans_for_mapupdate = 1.5 m) with its own
    noise realisation and sub-sample; guess = the true pose 0.5 m earlier.  Returns [(source (30000,3) f32, guess 4x4 f32,
    truth 4x4 f64)]; scan 0 is NOT cfg_ndt_30k's own source (different random stream)."""
    sensor = vlp32()
    sensor = Sensor(sensor.n_beams, sensor.elev_min_deg, sensor.elev_max_deg, sensor.n_azimuth * 3, sensor.range_noise,
                    sensor.max_range, sensor.min_range)
    world = world or make_world()
    x_last = 1.5 * 9
    xs = [x_last + 0.5 * (1 + j % 3) for j in range(n_scans)]
    poses = [trajectory_pose(x) for x in xs]
    sargs = (sensor.n_beams, sensor.elev_min_deg, sensor.elev_max_deg, sensor.n_azimuth, sensor.range_noise, sensor.max_range,
             sensor.min_range)
    jobs = [(world, sargs, T) for T in poses]
    geo = pool.map(_geometry_job, jobs) if pool is not None else [_geometry_job(j) for j in jobs]
    out = []
    for j, (keep, t) in enumerate(geo):
        rng = np.random.default_rng([WORLD_SEED, 40503, seed, j])
        src = subsample_exact(voxel_downsample(raycast_noise(sensor, keep, t, rng), 0.2), 30000, seed=WORLD_SEED + 131 * seed + j)
        out.append((src, trajectory_pose(xs[j] - 0.5).astype(np.float32), poses[j]))
    return out


def cfg_frontend_drive(n_scans: int, seed: int = 0, pool=None, world: World | None = None, n_keyframes: int = 10) -> dict:
    """The frontend's input over a drive (scanmatcher_component.cpp:296-356,436-481): the cfg-1/2 map of ten keyframes (each
    VoxelGrid(0.1)-filtered in its OWN sensor frame, with its pose) followed by n_scans RAW scans — as the sensor delivers them: no
    range filter, no VoxelGrid, ~147k points — taken every 0.5 m further along the trajectory, so that a frontend with
    trans_for_mapupdate = 1.5 m updates its map on every third scan.  Returns {frames, frame_poses, scans [(n,3) f32 sensor frame],
    truths [4x4 f64], guess0 (pose of the last keyframe)}."""
    if n_keyframes == 10:
        base = cfg_ndt_30k(seed=seed, pool=pool, keep_parts=True, world=world)
    else:   # e.g. the 20-frame window of the reference's lidarslam.yaml (num_targeted_cloud: 20)
        base = make_case(sensor=vlp32(), n_keyframes=n_keyframes, vg_map=0.1, vg_input=0.2, n_source=30000, seed=seed, world=world,
                         azimuth_oversample=3, name=f"ndt_30k_vs_{n_keyframes}frame", pool=pool, keep_parts=True)
    sensor = vlp32()
    sensor = Sensor(sensor.n_beams, sensor.elev_min_deg, sensor.elev_max_deg, sensor.n_azimuth * 3, sensor.range_noise,
                    sensor.max_range, sensor.min_range)
    world = world or make_world()
    x_last = 1.5 * (n_keyframes - 1)
    xs = [x_last + 0.5 * (1 + j) for j in range(n_scans)]
    poses = [trajectory_pose(x) for x in xs]
    sargs = (sensor.n_beams, sensor.elev_min_deg, sensor.elev_max_deg, sensor.n_azimuth, sensor.range_noise, sensor.max_range,
             sensor.min_range)
    jobs = [(world, sargs, T) for T in poses]
    geo = pool.map(_geometry_job, jobs) if pool is not None else [_geometry_job(j) for j in jobs]
    scans = []
    for j, (keep, t) in enumerate(geo):
        rng = np.random.default_rng([WORLD_SEED, 60607, seed, j])
        scans.append(raycast_noise(sensor, keep, t, rng))
    return {"frames": base.frames, "frame_poses": base.frame_poses, "scans": scans, "truths": [np.asarray(P, np.float64) for P in poses],
            "guess0": np.asarray(trajectory_pose(x_last), np.float64)}


def cfg_gicp_30k(seed: int = 0, pool=None) -> RegistrationCase:
    """cfg 3: same source; target additionally VoxelGrid(0.2) (scanmatcher_component.cpp:309-315)."""
    return make_case(sensor=vlp32(), n_keyframes=10, vg_map=0.1, vg_input=0.2, n_source=30000, vg_target=0.2,
                     seed=seed, azimuth_oversample=3, name="gicp_30k_vs_10frame", pool=pool)


def cfg_loop_candidate(c: int, pool=None) -> RegistrationCase:
    """cfg 4: candidate c starts 3*c m along the route; guess perturbed U(-1,1) m xy, U(-3,3) deg yaw."""
    rng = np.random.default_rng(1000 + c)
    dx, dy = rng.uniform(-1, 1, 2)
    dyaw = math.radians(rng.uniform(-3, 3))
    return cfg_ndt_30k(seed=100 + c, start_x=3.0 * c, guess_perturb=(dx, dy, dyaw),
                       world=make_world(x_shift=3.0 * c), pool=pool)


def cfg_dense_120k(seed: int = 0, pool=None) -> RegistrationCase:
    """cfg 5: 64-line scan (vg 0.1) cut to 120k vs 20-frame submap."""
    return make_case(sensor=hdl64(), n_keyframes=20, vg_map=0.1, vg_input=0.1, n_source=120000, seed=seed,
                     azimuth_oversample=3, name="ndt_120k_vs_20frame", pool=pool)


def small_case(n_source: int = 2000, n_keyframes: int = 3, seed: int = 0, guess_perturb=None,
               vg_input: float = 0.4, vg_map: float = 0.2) -> RegistrationCase:
    """Small case for CPU-side oracle tests (seconds, not minutes)."""
    sensor = Sensor(16, -20.0, 12.0, 600)
    return make_case(sensor=sensor, n_keyframes=n_keyframes, vg_map=vg_map, vg_input=vg_input, n_source=n_source,
                     seed=seed, guess_perturb=guess_perturb, name="small")


def as_pointxyzi(pts: np.ndarray) -> np.ndarray:
    """Pack (n,3) fp32 into pcl::PointXYZI's 32-byte layout: x,y,z,1,intensity,pad*3."""
    out = np.zeros((pts.shape[0], 8), np.float32)
    out[:, :3] = pts
    out[:, 3] = 1.0
    return out


# ---- loop-closure route ------------------------------------------------------
def make_loop_route(spacing: float = 3.0, length: float = 24.0, lane: float = 1.25, sensor: Sensor | None = None,
                    vg_map: float = 0.2, drift: tuple = (0.35, -0.25, 0.04, 0.01), seed: int = 0,
                    world: World | None = None, pool=None) -> list:
    """A route that returns to its start, as a lidarslam_msgs/MapArray stand-in: out along the corridor on the lane
    y = -lane, a turn, back on y = +lane, and a final submap next to the first one.  Each submap is one scan
    (VoxelGrid(vg_map), pose-local coordinates) with its ESTIMATED pose; the estimate drifts linearly with travelled
    distance up to `drift` = (dx, dy, dz, dyaw) at the end, which is what a loop edge has to correct.
    Returns a list of dicts: cloud (n,3) f32, position (3), orientation (x,y,z,w), distance, truth (4x4 f64).
    `pool` only parallelises the ray intersections (same clouds)."""
    sensor = sensor or Sensor(32, -25.0, 15.0, 900)
    world = world or make_world()
    rng = np.random.default_rng(WORLD_SEED + 104729 * seed + 5)
    n_leg = int(round(length / spacing))
    way = [(spacing * k, -lane, 0.0) for k in range(n_leg + 1)]                      # out, heading +x
    way += [(length + 0.5 * spacing, 0.0, 0.5 * math.pi)]                            # turn
    way += [(length - spacing * k, lane, math.pi) for k in range(n_leg + 1)]         # back, heading -x
    way += [(-0.5 * spacing, 0.2, 1.5 * math.pi), (0.4 * spacing, -lane + 0.3, 2.0 * math.pi)]  # turn, re-visit of the start
    out = []
    dist = 0.0
    for k, (x, y, yaw) in enumerate(way):
        if k:
            dist += math.hypot(x - way[k - 1][0], y - way[k - 1][1])
        out.append(dict(xyyaw=(x, y, yaw), distance=dist))
    total = max(dist, 1e-9)
    scans = raycast_many(world, sensor, [pose_matrix(sm["xyyaw"][0], sm["xyyaw"][1], 0.0, sm["xyyaw"][2]) for sm in out], rng, pool)
    for sm, scan in zip(out, scans):
        x, y, yaw = sm.pop("xyyaw")
        T = pose_matrix(x, y, 0.0, yaw)
        f = sm["distance"] / total
        est_yaw = yaw + drift[3] * f
        sm["truth"] = T
        sm["cloud"] = voxel_downsample(scan, vg_map)
        sm["position"] = (x + drift[0] * f, y + drift[1] * f, drift[2] * f)
        sm["orientation"] = (0.0, 0.0, math.sin(0.5 * est_yaw), math.cos(0.5 * est_yaw))
    return out


def cfg_loop_route_full(pool=None, length: float = 54.0) -> list:
    """The backend's input at the size the frontend produces it: every submap is a full VLP-32 revolution filtered at
    vg_size_for_map = 0.1 (~110k points, pose-local), one every 3 m along a there-and-back route long enough for the reference's own
    gate (lidarslam.yaml: distance_loop_closure 100 m -> > 100 m of travel between the two visits of the start)."""
    s = vlp32()
    sensor = Sensor(s.n_beams, s.elev_min_deg, s.elev_max_deg, s.n_azimuth * 3, s.range_noise, s.max_range, s.min_range)
    return make_loop_route(spacing=3.0, length=length, sensor=sensor, vg_map=0.1, pool=pool)
