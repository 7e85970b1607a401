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


@pytest.mark.parametrize("leaf", [0.1, 0.2, 0.5, 2.0, 8.0])
def test_voxel_grid_filter_matches_oracle(O, scan, leaf):
    from lidarslam_ros2_amd import NormalDistributionsTransform

    r = NormalDistributionsTransform(device=0)
    pts = np.concatenate([scan, np.array([[np.nan, 0, 0], [1, np.inf, 2]], np.float32)])  # non-finite points are dropped
    got = r.voxelGridFilter(synth.as_pointxyzi(pts), leaf)
    ref = O.voxel_grid_filter(pts, leaf)
    assert got.shape == ref.shape                     # same occupied-leaf set, same (leaf index) order
    # the centroid is accumulated in FLOAT over the leaf's points in ascending index, exactly as the CPU restatement of
    # pcl::CentroidPoint does: bit-identical
    assert np.array_equal(got, ref)
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


# ---- N4: sensor_msgs/PointCloud2 codec (arbitrary field offsets, intensity carried through the VoxelGrid) ------------
def _pc2_payload(xyz, intensity, point_step, offsets):
    """A PointCloud2 `data` buffer: float32 fields at byte offsets (x, y, z, intensity), junk in every other byte."""
    n = xyz.shape[0]
    rng = np.random.default_rng(9)
    buf = rng.integers(0, 255, (n, point_step), dtype=np.uint8)
    f = buf.view(np.float32).reshape(n, point_step // 4)
    for k in range(3):
        f[:, offsets[k] // 4] = xyz[:, k]
    if offsets[3] is not None:
        f[:, offsets[3] // 4] = intensity
    return buf


@pytest.mark.parametrize("point_step,offsets", [(32, (0, 4, 8, 16)), (24, (4, 12, 20, 0)), (16, (0, 4, 8, None)), (48, (16, 20, 24, 40))])
def test_pointcloud2_frontend_and_writer_match_oracle(O, scan, point_step, offsets):
    """fromROSMsg (any field offsets) + range filter + VoxelGrid with downsample_all_data + setInputSource, then toROSMsg of
    the filtered cloud: coordinates AND intensity bit-identical to the CPU restatement, junk bytes never leak."""
    from lidarslam_ros2_amd import NormalDistributionsTransform

    rng = np.random.default_rng(4)
    inten = rng.uniform(0, 255, scan.shape[0]).astype(np.float32)
    payload = _pc2_payload(scan, inten, point_step, offsets)
    rmin, rmax, leaf = 2.0, 60.0, 0.3
    r = NormalDistributionsTransform(device=0)
    n = r.setInputSourcePointCloud2(payload, scan.shape[0], point_step, offsets, rmin, rmax, leaf)
    # CPU: the reference's filter (scanmatcher_component.cpp:210-218, double arithmetic), then the VoxelGrid restatement
    rr = np.sqrt(scan[:, 0].astype(np.float64) ** 2 + scan[:, 1].astype(np.float64) ** 2)
    keep = (rmin < rr) & (rr < rmax)
    rec = np.c_[scan[keep], inten[keep] if offsets[3] is not None else np.zeros(keep.sum(), np.float32)].astype(np.float32)
    ref = O.voxel_grid_filter_xyzi(rec, leaf, 3)
    assert n == ref.shape[0]
    out = r.getInputSourcePointCloud2()                       # pcl::PointXYZI layout: x@0 y@4 z@8 intensity@16, step 32
    f = out.view(np.float32).reshape(n, 8)
    assert np.array_equal(f[:, :3], ref[:, :3])
    assert np.array_equal(f[:, 4], ref[:, 3])
    assert not f[:, 3].any() and not f[:, 5:].any()           # bytes outside the four fields are zero
    # writer with another layout round-trips through the reader
    out2 = r.getInputSourcePointCloud2(point_step=20, offsets=(8, 0, 4, 16))
    g = out2.view(np.float32).reshape(n, 5)
    assert np.array_equal(g[:, [2, 0, 1, 4]], ref)


def test_pointcloud2_stand_alone_filter_and_bad_layouts(O, scan):
    from lidarslam_ros2_amd import NormalDistributionsTransform, _capi

    r = NormalDistributionsTransform(device=0)
    inten = np.linspace(0, 1, scan.shape[0], dtype=np.float32)
    payload = _pc2_payload(scan, inten, 32, (0, 4, 8, 16))
    out = r.voxelGridFilterPointCloud2(payload, scan.shape[0], 32, (0, 4, 8, 16), 0.5)
    ref = O.voxel_grid_filter_xyzi(np.c_[scan, inten].astype(np.float32), 0.5, 3)
    f = out.view(np.float32).reshape(-1, 8)
    assert f.shape[0] == ref.shape[0] and np.array_equal(f[:, :3], ref[:, :3]) and np.array_equal(f[:, 4], ref[:, 3])
    for step, offs in ((32, (0, 4, 30, 16)), (30, (0, 4, 8, 16)), (32, (0, 4, 8, 33)), (32, (2, 4, 8, 16))):
        with pytest.raises(_capi.RegistrationError) as ei:
            r.setInputSourcePointCloud2(payload, 10, step, offs, 0.0, 100.0, 0.2)
        assert ei.value.status == -1


def test_device_side_grid_dimensions_give_the_same_filter(O):
    """From the second scan of an object on, the VoxelGrid filter behind lsr_set_input_source_pc2 / _frontend works out its grid
    dimensions on the DEVICE (one host wait per scan instead of two): same leaf set, same order, bit-identical centroids — through
    clouds that need as many key bits as the last one, more (the sort was planned too short: flagged, host form), fewer, a scan the
    range filter rejects completely, a leaf size whose index space overflows (PCL's error), and back."""
    import os
    from lidarslam_ros2_amd import NormalDistributionsTransform, _capi

    if os.environ.get("LSR_VG_SORT", "").startswith("r") or os.environ.get("LSR_VG_DEVICE_DIMS", "") == "0":
        pytest.skip("an A/B switch of this process keeps the filter on the host-dimension form")
    step, offs = 32, (0, 4, 8, 16)
    r = NormalDistributionsTransform(device=0)
    assert r.voxelFilterForm() == 0

    def run(n, extent, seed, leaf, rmin=0.0, rmax=1.0e4):
        pts = _dense_cloud(n, extent, seed)
        pts[::613] = np.nan
        inten = np.random.default_rng(seed).uniform(0, 255, n).astype(np.float32)
        got_n = r.setInputSourcePointCloud2(_pc2_payload(pts, inten, step, offs), n, step, offs, rmin, rmax, leaf)
        rr = np.sqrt(pts[:, 0].astype(np.float64) ** 2 + pts[:, 1].astype(np.float64) ** 2)
        keep = (rmin < rr) & (rr < rmax)                                  # NaN x: rejected here, dropped there
        ref = O.voxel_grid_filter_xyzi(np.c_[pts[keep], inten[keep]].astype(np.float32), leaf, 3) if keep.any() else np.zeros((0, 4), np.float32)
        assert got_n == ref.shape[0], (n, extent, seed, leaf, r.voxelFilterForm())
        if got_n:
            f = r.getInputSourcePointCloud2().view(np.float32).reshape(got_n, 8)
            assert np.array_equal(f[:, :3], ref[:, :3]) and np.array_equal(f[:, 4], ref[:, 3]), (n, extent, seed, leaf, r.voxelFilterForm())
        return r.voxelFilterForm()

    assert run(60000, 40.0, 1, 0.2) == 1               # first scan of the object: dimensions on the host
    assert run(60000, 40.0, 2, 0.2) == 2               # same extent: device form
    assert run(61000, 39.0, 3, 0.2) == 2
    assert run(147443, 95.0, 4, 0.2) == 3              # a larger index space than the sort was planned for: flagged, host form ran
    assert run(147443, 95.0, 5, 0.2) == 2              # ... which renewed the plan
    assert run(30000, 10.0, 6, 0.2) == 2               # fewer bits than planned: still ordered by the whole key
    assert run(30000, 10.0, 7, 0.2, rmin=1.0e5, rmax=2.0e5) == 2 and r.getInputSourcePointCloud2().shape[0] == 0   # nothing passes the range filter
    assert run(30000, 10.0, 8, 0.2, rmin=2.0, rmax=9.0) == 2
    with pytest.raises(_capi.RegistrationError) as ei:  # PCL: "Leaf size is too small for the input dataset", found by the device form
        run(30000, 3000.0, 9, 0.2)
    assert ei.value.status == -7 and r.voxelFilterForm() == 3   # LSR_ERR_INDEX_OVERFLOW
    assert run(60000, 40.0, 10, 0.2) in (2, 3)
    assert run(60000, 40.0, 11, 0.2) == 2
    assert run(60000, 40.0, 15, 0.4) == 1              # another leaf size is another index space: the plan is not carried over
    assert run(60000, 40.0, 16, 0.4) == 2
    with pytest.raises(_capi.RegistrationError) as ei:  # the same error from the host form (first scan at this leaf size)
        run(30000, 95.0, 17, 1.0e-4)
    assert ei.value.status == -7 and r.voxelFilterForm() == 1
    assert run(60000, 40.0, 18, 0.4) == 2
    assert run(610001, 95.0, 13, 0.2) == 1             # (back to leaf 0.2: host form) beyond the fused sort's 256 workgroups: key kernel and histogram stay two launches
    assert run(610001, 95.0, 14, 0.2) == 2
    # strided xyz records (lsr_set_input_source_frontend) take the same path
    pts = _dense_cloud(50000, 40.0, 12)
    n1 = r.setInputSourceFrontend(synth.as_pointxyzi(pts), 1.0, 35.0, 0.2)
    rr = np.sqrt(pts[:, 0].astype(np.float64) ** 2 + pts[:, 1].astype(np.float64) ** 2)
    ref = O.voxel_grid_filter(pts[(1.0 < rr) & (rr < 35.0)], 0.2)
    assert n1 == ref.shape[0] and r.voxelFilterForm() == 2
    assert np.array_equal(r.getInputSourcePointCloud2().view(np.float32).reshape(n1, 8)[:, :3], ref)


def test_device_side_grid_dimensions_equal_the_host_form():
    """A/B in child processes: LSR_VG_DEVICE_DIMS=0 keeps every scan on the host-dimension form; the scans of a short stream come out
    bit-identical either way."""
    import os
    import subprocess
    import sys
    import tempfile

    if os.environ.get("LSR_VG_SORT", "").startswith("r") or os.environ.get("LSR_VG_DEVICE_DIMS", "") == "0":
        pytest.skip("an A/B switch of this process keeps the filter on the host-dimension form")
    code = ("import numpy as np, sys\n"
            "sys.path.insert(0, %r)\n"
            "from lidarslam_ros2_amd import NormalDistributionsTransform, synth\n"
            "r = NormalDistributionsTransform(device=0)\n"
            "outs, forms = [], []\n"
            "for k in range(4):\n"
            "    rng = np.random.default_rng(20 + k)\n"
            "    pts = rng.uniform(-70, 70, (120000, 3)).astype(np.float32); pts[:, 2] *= 0.05\n"
            "    n = r.setInputSourceFrontend(synth.as_pointxyzi(pts), 0.5, 60.0, 0.2)\n"
            "    outs.append(r.getInputSourcePointCloud2()); forms.append(r.voxelFilterForm())\n"
            "np.savez(sys.argv[1], forms=np.array(forms), **{'o%%d' %% k: o for k, o in enumerate(outs)})\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    with tempfile.TemporaryDirectory() as d:
        for name, env in (("device", {}), ("host", {"LSR_VG_DEVICE_DIMS": "0"})):
            path = os.path.join(d, name + ".npz")
            subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, **env), timeout=300)
            res.append(dict(np.load(path)))
    assert list(res[0]["forms"]) == [1, 2, 2, 2] and list(res[1]["forms"]) == [1, 1, 1, 1]
    for k in range(4):
        assert np.array_equal(res[0]["o%d" % k], res[1]["o%d" % k])


# ---- the hand-written stable LSD radix sort behind N1 (csrc/lsd_sort.hip) ------------------------------------------------
def _dense_cloud(n, extent, seed):
    """n points: half of them uniform in a box of `extent`, half piled into a few leaves (hot digits in both passes)."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-extent, extent, (n // 2, 3)).astype(np.float32)
    a[:, 2] *= 0.1
    hot = rng.uniform(-1.0, 1.0, (n - n // 2, 3)).astype(np.float32) * np.float32(0.3)
    pts = np.concatenate([a, hot])
    return pts[rng.permutation(n)]


@pytest.mark.parametrize("n,extent,leaf,what", [
    (147443, 95.0, 0.2, "a raw scan: two passes, 1024-point workgroups"),
    (700001, 95.0, 0.2, "a map-side cloud: 4096-point workgroups, ragged tail"),
    (90000, 60.0, 0.05, "30-bit leaf index: three passes of 10 bits"),
    (300000, 100.0, 0.08, "29-bit leaf index on 300k points"),
    (5000, 2.0, 0.5, "a handful of leaves: one short pass, every wave full of equal digits"),
    (63, 2.0, 0.5, "less than one wave"),
])
def test_lsd_sort_path_is_bit_identical_to_the_oracle(O, n, extent, leaf, what):
    """pcl::VoxelGrid on the device through the hand-written LSD sort: same leaf set, same leaf-index order, float centroids
    bit-identical (the sums of a leaf run over its points in ascending index — only a STABLE sort gives that)."""
    from lidarslam_ros2_amd import NormalDistributionsTransform

    pts = _dense_cloud(n, extent, seed=n)
    pts[::977] = np.nan                                   # non-finite points: the sentinel run, dropped
    r = NormalDistributionsTransform(device=0)
    got = r.voxelGridFilter(synth.as_pointxyzi(pts), leaf)
    ref = O.voxel_grid_filter(pts, leaf)
    assert got.shape == ref.shape, what
    assert np.array_equal(got, ref), what


def test_lsd_sort_path_equals_the_rocprim_path():
    """A/B: the same filter through rocPRIM's radix sort + run_length_encode + scan (LSR_VG_SORT=rocprim, rounds 1-4) in a child
    process: bit-identical output."""
    import os
    import subprocess
    import sys
    import tempfile

    code = ("import numpy as np, sys\n"
            "sys.path.insert(0, %r)\n"
            "from lidarslam_ros2_amd import NormalDistributionsTransform, synth\n"
            "rng = np.random.default_rng(11)\n"
            "pts = rng.uniform(-60, 60, (200000, 3)).astype(np.float32); pts[:, 2] *= 0.05\n"
            "out = NormalDistributionsTransform(device=0).voxelGridFilter(synth.as_pointxyzi(pts), 0.25)\n"
            "np.save(sys.argv[1], out)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for name, env in (("lsd", {}), ("rocprim", {"LSR_VG_SORT": "rocprim"})):
            path = os.path.join(d, name + ".npy")
            subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, **env), timeout=300)
            outs.append(np.load(path))
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1])
