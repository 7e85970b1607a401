"""The frontend loop as the reference runs it (scanmatcher_component.cpp:296-356 receiveCloud, :436-481 updateMap) over a stream of
RAW scans with map updates, on the gfx950 core and on the CPU oracle: raw PointCloud2 payload -> range filter -> VoxelGrid(0.2) ->
setInputSource -> align(previous pose); every 1.5 m VoxelGrid(0.1) of the scan + assembly of the last ten submaps + setInputTarget.
Both pipelines feed on their OWN previous poses and their OWN maps, so the comparison holds the whole sequence — preprocessing,
registration, map assembly — to the north_star bar on every scan (drift included)."""
import numpy as np
import pytest

from lidarslam_ros2_amd import synth
from lidarslam_ros2_amd.frontend import FrontendParams, FrontendReplay, FrontendResult, as_pc2_payload
from lidarslam_ros2_amd.posemath import pose_delta

pytestmark = pytest.mark.gpu

N_SCANS = 12   # four map updates


@pytest.fixture(scope="module")
def drive():
    import multiprocessing as mp
    import os

    with mp.get_context("spawn").Pool(min(32, len(os.sched_getaffinity(0)))) as p:
        return synth.cfg_frontend_drive(N_SCANS, pool=p)


def _replay(reg, drive, to_device=None, device_payloads=False, mapper=None, builder=None, async_update=False, swap_lag=0):
    import torch

    if device_payloads and to_device is None:   # a device run keeps its keyframes in HBM whichever way they are produced
        to_device = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    fr = FrontendReplay(reg, FrontendParams(), to_device=to_device, mapper=mapper, builder=builder, async_update=async_update, swap_lag=swap_lag)
    fr.initialise(drive["frames"], drive["frame_poses"], drive["guess0"])
    out = FrontendResult()
    for scan in drive["scans"]:
        host = as_pc2_payload(scan)
        payload = torch.from_numpy(host).cuda() if device_payloads else host
        fr.receive_cloud(payload, int(scan.shape[0]), out, payload_host=host)
    fr.finish(out)
    return out


def _ndt():
    from lidarslam_ros2_amd import DIRECT7, NormalDistributionsTransform

    r = NormalDistributionsTransform(device=0)   # its own stream
    r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(35); r.setNeighborhoodSearchMethod(DIRECT7)
    return r


def test_map_update_on_a_worker_thread_gives_the_serial_replay_bit_for_bit(drive):
    """The map side off the scan path (scanmatcher_component.cpp:427-434 runs updateMap on a thread, :298-320 takes the target over
    at a later callback): a worker thread filters the keyframe (mapper), assembles the window and builds the voxel grid (builder, own
    stream) while the callback thread registers scans on `reg`; the hand-over is lsr_share_target.  For every hand-over lag the
    threaded replay returns the bits of the serial replay with that lag; lag 0 is round 5's replay (update before the next scan)."""
    plain = _replay(_ndt(), drive, device_payloads=True, mapper=_ndt())
    for lag in (0, 1):
        serial = _replay(_ndt(), drive, device_payloads=True, mapper=_ndt(), builder=_ndt(), swap_lag=lag)
        for rep in range(2):   # twice: the second drive recycles the targets of the first (lsr_share_target hands them back)
            reg, mapper, builder = _ndt(), _ndt(), _ndt()
            thr = _replay(reg, drive, device_payloads=True, mapper=mapper, builder=builder, async_update=True, swap_lag=lag)
            assert thr.update_at == serial.update_at and len(thr.update_at) >= 3
            assert thr.points_kept == serial.points_kept and thr.iterations == serial.iterations
            assert all(np.array_equal(a, b) for a, b in zip(thr.poses, serial.poses)), lag
            assert len(thr.update_seconds) == len(thr.update_at) == len(thr.swap_wait_seconds)
        if lag == 0:
            assert all(np.array_equal(a, b) for a, b in zip(serial.poses, plain.poses))
    # the lag changes which scans see the new keyframe, not where the drive ends up
    for a, t in zip(serial.poses, drive["truths"]):
        dt, ang = pose_delta(a, t)
        assert dt <= 0.05 and ang <= 2e-3


def test_frontend_stream_with_a_delayed_hand_over_matches_the_oracle(drive):
    """swap_lag = 1 (the target of an update triggered by scan k serves from scan k + 2 on — the next scan is registered while the map
    side runs) on the gfx950 core with the worker thread, and on the CPU oracle serially: every scan inside the bar."""
    from frontend_oracle import OracleFrontendRegistration

    gpu = _replay(_ndt(), drive, device_payloads=True, mapper=_ndt(), builder=_ndt(), async_update=True, swap_lag=1)
    cpu = _replay(OracleFrontendRegistration(5.0, 0.01, 35), drive, swap_lag=1)
    assert gpu.update_at == cpu.update_at and gpu.points_kept == cpu.points_kept and gpu.iterations == cpu.iterations
    for j, (a, b) in enumerate(zip(gpu.poses, cpu.poses)):
        dt, ang = pose_delta(a, b)
        assert dt <= 1e-3 and ang <= 1e-4, (j, dt, ang)


def test_frontend_stream_matches_the_oracle_on_every_scan(drive):
    import torch

    from lidarslam_ros2_amd import DIRECT7, NormalDistributionsTransform
    from frontend_oracle import OracleFrontendRegistration

    ndt = NormalDistributionsTransform(device=0)
    ndt.setResolution(5.0); ndt.setTransformationEpsilon(0.01); ndt.setMaximumIterations(35); ndt.setNeighborhoodSearchMethod(DIRECT7)
    mapper = NormalDistributionsTransform(device=0)   # its source slot filters the new keyframes, which then never leave HBM
    gpu = _replay(ndt, drive, to_device=lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda(), device_payloads=True, mapper=mapper)
    cpu = _replay(OracleFrontendRegistration(5.0, 0.01, 35), drive)
    assert gpu.update_at == cpu.update_at and len(gpu.update_at) >= 3, (gpu.update_at, cpu.update_at)   # the same scans became keyframes
    assert gpu.points_kept == cpu.points_kept                                                            # N1/N4: same filtered scans
    worst = (0.0, 0.0)
    for j, (a, b) in enumerate(zip(gpu.poses, cpu.poses)):
        dt, ang = pose_delta(a, b)
        assert dt <= 1e-3 and ang <= 1e-4, (j, dt, ang)
        worst = (max(worst[0], dt), max(worst[1], ang))
    assert gpu.iterations == cpu.iterations, (gpu.iterations, cpu.iterations)
    # and the stream tracks the ground truth (a frontend that drifted would still agree with an oracle that drifted the same way)
    for j, (a, t) in enumerate(zip(gpu.poses, drive["truths"])):
        dt, ang = pose_delta(a, t)
        assert dt <= 0.05 and ang <= 2e-3, (j, dt, ang)
    print("frontend stream: worst GPU-vs-oracle pose difference over %d scans: %.2e m %.2e rad" % (len(gpu.poses), worst[0], worst[1]))
    # ... and against the REFERENCE's own run of the same drive, once oracle/ref_recipe has dumped it (make -C oracle ref)
    from golden_fixtures import load_reference
    ref = load_reference("frontend_stream")
    if ref is not None and len(ref["poses"]) == len(gpu.poses):
        assert gpu.update_at == ref["update_at"].tolist() and gpu.points_kept == ref["points_kept"].tolist()
        for j, (a, b) in enumerate(zip(gpu.poses, ref["poses"])):
            dt, ang = pose_delta(a, b)
            assert dt <= 1e-3 and ang <= 1e-4, ("reference", j, dt, ang)


def test_host_and_device_payloads_give_the_same_stream(drive):
    """The raw payload handed over as a CUDA tensor — with the keyframes produced and kept in HBM (lsr_get_source_pc2_device), and
    with the keyframes taken through the host — or as a host buffer (PCIe-inclusive path): bit-identical poses, all three."""
    from lidarslam_ros2_amd import DIRECT7, NormalDistributionsTransform

    outs = []
    for dev, resident in ((True, True), (True, False), (False, False)):
        ndt = NormalDistributionsTransform(device=0)
        ndt.setResolution(5.0); ndt.setTransformationEpsilon(0.01); ndt.setMaximumIterations(35); ndt.setNeighborhoodSearchMethod(DIRECT7)
        outs.append(_replay(ndt, drive, device_payloads=dev, mapper=NormalDistributionsTransform(device=0) if resident else None))
    assert outs[0].update_at == outs[1].update_at == outs[2].update_at
    for a, b, c in zip(outs[0].poses, outs[1].poses, outs[2].poses):
        assert np.array_equal(a, b) and np.array_equal(a, c)
