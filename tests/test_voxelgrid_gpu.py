"""GPU parity for the 'next' row N1: pcl::VoxelGrid::filter on the device vs the oracle."""
import numpy as np
import pytest

from lidarslam_ros2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def scan():
    rng = np.random.default_rng(0)
    return synth.raycast(synth.make_world(), synth.vlp32(), synth.trajectory_pose(3.0), rng)   # raw ~45k-point scan


@pytest.mark.parametrize("leaf", [0.1, 0.2, 0.5, 2.0])
def test_voxel_grid_filter_matches_oracle(O, scan, leaf):
    from lidarslam_ros2_amd import NormalDistributionsTransform

    r = NormalDistributionsTransform(device=0)
    pts = np.concatenate([scan, np.array([[np.nan, 0, 0], [1, np.inf, 2]], np.float32)])  # non-finite points are dropped
    got = r.voxelGridFilter(synth.as_pointxyzi(pts), leaf)
    ref = O.voxel_grid_filter(pts, leaf)
    assert got.shape == ref.shape                     # same occupied-leaf set, same (leaf index) order
    # fp64-accumulated centroid rounded to fp32 vs the reference's fp32 accumulation: a few ulp of the coordinate
    assert np.abs(got - ref).max() <= 4e-6 * max(1.0, float(np.abs(ref).max()))
    # and the numpy stand-in used by the workload generator agrees on the leaf set
    assert synth.voxel_downsample(scan, leaf).shape[0] == ref.shape[0]


def test_filtered_source_feeds_align(O, scan):
    """Frontend flow (scanmatcher_component.cpp:324-353): filter on device, register, same pose as filtering on the
    host first."""
    from lidarslam_ros2_amd import NormalDistributionsTransform
    from lidarslam_ros2_amd.posemath import pose_delta

    case = synth.small_case(n_source=2000, n_keyframes=3)
    raw = synth.raycast(synth.make_world(), synth.Sensor(16, -20.0, 12.0, 600), case.truth, np.random.default_rng(5))
    a, b = NormalDistributionsTransform(device=0), NormalDistributionsTransform(device=0)
    for r in (a, b):
        r.setResolution(5.0)
        r.setTransformationEpsilon(0.01)
        r.setInputTarget(case.target)
    n = a.setInputSourceFiltered(synth.as_pointxyzi(raw), 0.4)
    host_filtered = O.voxel_grid_filter(raw, 0.4)
    assert n == host_filtered.shape[0]
    b.setInputSource(host_filtered)
    a.align(case.guess)
    b.align(case.guess)
    dt, ang = pose_delta(a.getFinalTransformation(), b.getFinalTransformation())
    assert dt < 1e-4 and ang < 1e-5
    assert pose_delta(a.getFinalTransformation(), case.truth)[0] < 0.1


def test_submap_assembly_on_device_matches_host_assembly(O):
    """'Next' row N2: transformPointCloud per keyframe + concat on the device == assembling on the host
    (scanmatcher_component.cpp:449-464) — same voxel table, bit-exact leaf set."""
    from lidarslam_ros2_amd import NormalDistributionsTransform

    world, sensor = synth.make_world(), synth.Sensor(16, -20.0, 12.0, 600)
    rng = np.random.default_rng(3)
    frames, poses = [], []
    for k in range(4):
        T = synth.trajectory_pose(1.5 * k)
        frames.append(synth.voxel_downsample(synth.raycast(world, sensor, T, rng), 0.2))
        poses.append(T.astype(np.float32))
    # host assembly with pcl::transformPointCloud's fp32 arithmetic: ((m00 x + m01 y) + m02 z) + m03
    chunks = []
    for f, T in zip(frames, poses):
        cols = [((T[r, 0] * f[:, 0] + T[r, 1] * f[:, 1]) + T[r, 2] * f[:, 2]) + T[r, 3] for r in range(3)]
        chunks.append(np.stack(cols, 1).astype(np.float32))
    host = np.concatenate(chunks)
    a, b = NormalDistributionsTransform(device=0), NormalDistributionsTransform(device=0)
    for r in (a, b):
        r.setResolution(2.0)
    a.setInputTargetFrames([synth.as_pointxyzi(f) for f in frames], poses)
    b.setInputTarget(host)
    da, db = a.gridDump(), b.gridDump()
    assert np.array_equal(da["idx"], db["idx"]) and np.array_equal(da["n"], db["n"])
    assert np.array_equal(da["mean"], db["mean"]) and np.array_equal(da["icov"], db["icov"])   # bit-identical clouds
    ref = O.VoxelGridCovariance(host, 2.0)
    assert a.gridInfo()["n_valid"] == ref.n_valid


def test_frontend_preprocessing_range_filter_then_voxelgrid(O, scan):
    """'Next' row N4: min-max range filter (scanmatcher_component.cpp:210-218) + VoxelGrid + setInputSource in one
    device call == doing the two filters on the host."""
    from lidarslam_ros2_amd import NormalDistributionsTransform

    rmin, rmax, leaf = 2.0, 60.0, 0.2
    r64 = np.sqrt(scan[:, 0].astype(np.float64) ** 2 + scan[:, 1].astype(np.float64) ** 2)
    kept = scan[(rmin < r64) & (r64 < rmax)]
    assert 0 < kept.shape[0] < scan.shape[0]
    ref = O.voxel_grid_filter(kept, leaf)
    case = synth.small_case(n_source=1000, n_keyframes=2)
    a = NormalDistributionsTransform(device=0)
    a.setResolution(5.0)
    a.setInputTarget(case.target)
    n = a.setInputSourceFrontend(synth.as_pointxyzi(scan), rmin, rmax, leaf)   # PointCloud2-style 32-byte records
    assert n == ref.shape[0]
    out = a.align(np.eye(4, dtype=np.float32), output=True)                   # output = T * filtered source
    T = a.getFinalTransformation()
    back = (out - T[:3, 3]) @ T[:3, :3]                                        # undo T: recovers the filtered cloud
    assert np.abs(back - ref).max() < 2e-3
