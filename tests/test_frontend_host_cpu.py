"""CPU: the host side of the frontend loop (lidarslam_ros2_amd/frontend.py — receiveCloud / updateMap of
scanmatcher_component.cpp:296-356,436-481) against a scripted registration object: which calls are made, in which order, with which
arguments.  No device, no oracle: the numerics of the loop are held by tests/test_frontend_stream_gpu.py."""
import numpy as np

from lidarslam_ros2_amd.frontend import PC2_XYZI, FrontendParams, FrontendReplay, FrontendResult, as_pc2_payload


class ScriptedRegistration:
    """Records every call; align() answers with the next pose of a script; the map-side filter is the identity, so that what reaches
    setInputTargetFrames shows what the loop itself did to the cloud (the range mask)."""

    def __init__(self, poses):
        self.script = [np.asarray(P, np.float64) for P in poses]
        self.calls = []
        self.targets = []          # (frames, poses) of every setInputTargetFrames
        self.guesses = []
        self.k = 0
        self.final = np.eye(4)

    def setInputSourcePointCloud2(self, data, n_points, point_step, offsets, rmin, rmax, leaf):
        self.calls.append(("source", int(n_points), int(point_step), tuple(offsets), float(rmin), float(rmax), float(leaf)))
        return int(n_points) // 4

    def align(self, guess):
        self.calls.append(("align",))
        self.guesses.append(np.asarray(guess, np.float64))
        self.final = self.script[self.k]
        self.k += 1

    def getFinalTransformation(self):
        return self.final

    def getFinalNumIteration(self):
        return 6

    def voxelGridFilterPointCloud2(self, data, n_points, point_step, offsets, leaf, out_point_step=32, out_offsets=(0, 4, 8, 16)):
        self.calls.append(("map_filter", int(n_points), float(leaf)))
        return np.asarray(data).reshape(int(n_points), point_step).copy()

    def setInputTargetFrames(self, frames, poses):
        self.calls.append(("target", len(frames)))
        self.targets.append(([f if isinstance(f, tuple) else np.asarray(f) for f in frames], [np.asarray(P, np.float64) for P in poses]))


def _pose(x):
    T = np.eye(4)
    T[0, 3] = x
    return T


def test_payload_layout_is_pcl_pointxyzi():
    xyz = np.arange(12, dtype=np.float32).reshape(4, 3)
    inten = np.array([7, 8, 9, 10], np.float32)
    p = as_pc2_payload(xyz, inten)
    assert p.shape == (4, 32) and p.dtype == np.uint8 and PC2_XYZI == (32, (0, 4, 8, 16))
    f = p.view(np.float32).reshape(4, 8)
    assert np.array_equal(f[:, :3], xyz) and np.array_equal(f[:, 4], inten)
    assert not f[:, 3].any() and not f[:, 5:].any()          # padding bytes are zero, as pcl::toROSMsg leaves them


def test_receive_cloud_and_update_map_make_the_reference_calls_in_the_reference_order():
    prm = FrontendParams()
    # twelve keyframes 1.5 m apart to start from; seven scans 0.5 m apart: map updates after the scans that are >= 1.5 m from the last keyframe
    frames = [np.full((5, 3), float(k), np.float32) for k in range(12)]
    frame_poses = [_pose(1.5 * k) for k in range(12)]
    x0 = 1.5 * 11
    truth = [_pose(x0 + 0.5 * (j + 1)) for j in range(7)]
    reg = ScriptedRegistration(truth)
    fr = FrontendReplay(reg, prm)
    fr.initialise(frames, frame_poses, _pose(x0))
    # the initial target: the newest num_targeted_cloud keyframes, newest first (updateMap concatenates the new one first, :448-464)
    assert reg.calls == [("target", prm.num_targeted_cloud)]
    f0, p0 = reg.targets[0]
    assert [float(f[0, 0]) for f in f0] == [float(k) for k in range(11, 1, -1)]
    assert [P[0, 3] for P in p0] == [1.5 * k for k in range(11, 1, -1)]
    assert all(f.shape == (5, 8) and f.dtype == np.float32 for f in f0)      # (m,8) fp32 pcl::PointXYZI records

    out = FrontendResult()
    rng = np.random.default_rng(0)
    for j in range(7):
        xyz = rng.uniform(-30, 30, (40, 3)).astype(np.float32)
        xyz[0] = (0.01, 0.0, 1.0)          # closer than scan_min_range: masked before the map-side filter
        xyz[1] = (150.0, 0.0, 1.0)         # beyond scan_max_range
        xyz[2] = (0.0, 0.05, 99.0)         # the range is HORIZONTAL (:212): z does not rescue it
        fr.receive_cloud(as_pc2_payload(xyz), 40, out)
    assert len(out.poses) == 7 and out.iterations == [6] * 7 and out.points_kept == [10] * 7
    assert out.update_at == [2, 5]                                            # 1.5 m after the last keyframe, then 1.5 m after that one
    # the guess of every align is the previous scan's pose (:353), the first one the pose handed to initialise
    assert np.allclose(reg.guesses[0], _pose(x0)) and all(np.allclose(reg.guesses[j], truth[j - 1]) for j in range(1, 7))
    # per scan: source (range filter + VoxelGrid(vg_size_for_input) in one call) then align; a map update = filter + target, after the align
    want = [("target", 10)]
    for j in range(7):
        want += [("source", 40, 32, (0, 4, 8, 16), prm.scan_min_range, prm.scan_max_range, prm.vg_size_for_input), ("align",)]
        if j in (2, 5):
            want += [("map_filter", 37, prm.vg_size_for_map), ("target", 10)]   # 40 points minus the three the range mask drops
    assert reg.calls == want
    # the new keyframe leads the window with the pose the scan was registered at; the window keeps num_targeted_cloud frames
    f1, p1 = reg.targets[1]
    assert len(f1) == 10 and np.allclose(p1[0], truth[2]) and f1[0].shape == (37, 8)
    assert [P[0, 3] for P in p1[1:]] == [1.5 * k for k in range(11, 2, -1)]
    f2, p2 = reg.targets[2]
    assert np.allclose(p2[0], truth[5]) and np.allclose(p2[1], truth[2]) and [P[0, 3] for P in p2[2:]] == [1.5 * k for k in range(11, 3, -1)]


def test_keyframes_go_through_to_device_and_the_mapper_needs_a_device_payload():
    seen = []

    def to_device(a):
        seen.append(a.shape)
        return ("resident", a)

    frames = [np.zeros((3, 3), np.float32) for _ in range(3)]
    reg = ScriptedRegistration([_pose(2.0)])
    fr = FrontendReplay(reg, FrontendParams(), to_device=to_device, mapper=object())   # a mapper is only used with CUDA payloads
    fr.initialise(frames, [_pose(0.0)] * 3, _pose(0.0))
    assert seen == [(3, 8)] * 3
    out = FrontendResult()
    fr.receive_cloud(as_pc2_payload(np.full((6, 3), 5.0, np.float32)), 6, out)         # host payload: the host path, through to_device
    assert out.update_at == [0] and seen[-1] == (6, 8)
    assert reg.targets[-1][0][0][0] == "resident"


class ScriptedShared(ScriptedRegistration):
    """The callback's object of an asynchronous replay: it never builds a target, it takes the builder's over."""

    def __init__(self, poses, log):
        super().__init__(poses)
        self.log = log

    def shareTargetOf(self, owner):
        self.calls.append(("share",))
        self.log.append(("share", len(owner.targets)))        # which of the builder's targets is in place from here on

    def align(self, guess):
        super().align(guess)
        self.log.append(("align", self.k - 1))


def _drive(fr, n=9):
    out = FrontendResult()
    rng = np.random.default_rng(1)
    for j in range(n):
        fr.receive_cloud(as_pc2_payload(rng.uniform(-30, 30, (20, 3)).astype(np.float32)), 20, out)
    fr.finish(out)
    return out


def test_asynchronous_map_update_runs_on_the_builder_and_is_handed_over_at_the_due_callback():
    """updateMap on a worker thread (scanmatcher_component.cpp:427-434), the new target taken over at the start of a later callback
    (:298-320): the callback's object never filters a keyframe or builds a target; with swap_lag = L the target triggered by scan k
    is in place for scan k + 1 + L — in the asynchronous replay and in the serial one alike; no update is triggered while one is
    pending (`!mapping_flag_`)."""
    prm = FrontendParams()
    frames = [np.full((5, 3), float(k), np.float32) for k in range(12)]
    frame_poses = [_pose(1.5 * k) for k in range(12)]
    x0 = 1.5 * 11
    truth = [_pose(x0 + 0.8 * (j + 1)) for j in range(9)]     # 0.8 m per scan: an update is due after every second scan
    for lag in (0, 1, 2):
        logs = []
        for asynchronous in (True, False):
            log = []
            reg = ScriptedShared(truth, log)
            builder = ScriptedRegistration([])
            fr = FrontendReplay(reg, prm, builder=builder, async_update=asynchronous, swap_lag=lag)
            fr.initialise(frames, frame_poses, _pose(x0))
            out = _drive(fr)
            assert not any(c[0] in ("target", "map_filter") for c in reg.calls)                 # the callback's object builds nothing
            assert [c[0] for c in builder.calls].count("target") == 1 + len(out.update_at)       # initial target + one per update
            assert len(out.update_seconds) == len(out.update_at) == len(out.swap_wait_seconds)
            logs.append((log, out.update_at, [P.copy() for _, P in fr.submaps]))
            # the hand-over of update u (triggered after scan k = update_at[u]) comes right before align number k + 1 + lag
            aligns_before_share = []
            n_align = 0
            for e in log[1:]:                   # log[0] is the initial hand-over
                if e[0] == "align":
                    n_align += 1
                else:
                    aligns_before_share.append(n_align)
            due = [k + 1 + lag for k in out.update_at]
            assert aligns_before_share == [min(d, 9) for d in due], (lag, asynchronous, aligns_before_share, due)
        assert logs[0][0] == logs[1][0] and logs[0][1] == logs[1][1]                              # same schedule, threaded or not
        assert all(np.array_equal(a, b) for a, b in zip(logs[0][2], logs[1][2]))                  # ... and the same map window at the end
    # lag 2 with an update due every second scan: the pending flag suppresses triggers (fewer updates than with lag 0)
    reg = ScriptedShared(truth, [])
    fr = FrontendReplay(reg, prm, builder=ScriptedRegistration([]), async_update=True, swap_lag=0)
    fr.initialise(frames, frame_poses, _pose(x0))
    n0 = len(_drive(fr).update_at)
    reg = ScriptedShared(truth, [])
    fr = FrontendReplay(reg, prm, builder=ScriptedRegistration([]), async_update=True, swap_lag=3)
    fr.initialise(frames, frame_poses, _pose(x0))
    n3 = len(_drive(fr).update_at)
    assert n3 < n0
