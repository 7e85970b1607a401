"""The parameter sets the reference SHIPS, at full size (VERDICT r05 missing #4) — not BASELINE's 5 m / 10-frame configuration:

  lidarslam/param/lidarslam.yaml:5-17    frontend: NDT, ndt_resolution 2.0, vg_size_for_input 0.5, vg_size_for_map 0.1, scan range 1..200 m,
                                         num_targeted_cloud 20 (a 20-frame window of full VLP-32 keyframes, ~1.3 M target points)
  lidarslam/param/lidarslam.yaml:30-41   backend: NDT, ndt_resolution 1.0, voxel_leaf_size 0.1, threshold 0.7, distance_loop_closure 100,
                                         range 20, search_submap_num 2
  graph_based_slam/param/graphbasedslam.yaml:3-7   backend: GICP, voxel_leaf_size 0.2, threshold 1.5, distance_loop_closure 30

each through the gfx950 core and through the CPU oracle on the same inputs, held to north_star's 1e-3 m / 1e-4 rad."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from lidarslam_ros2_amd import synth
from lidarslam_ros2_amd.frontend import FrontendParams, FrontendReplay, FrontendResult, as_pc2_payload
from lidarslam_ros2_amd.posemath import pose_delta

pytestmark = pytest.mark.gpu
TOL_T, TOL_R = 1e-3, 1e-4


def _pool():
    return mp.get_context("spawn").Pool(min(32, len(os.sched_getaffinity(0))))


@pytest.fixture(scope="module")
def drive20():
    with _pool() as p:
        return synth.cfg_frontend_drive(9, pool=p, n_keyframes=20)   # two map updates


@pytest.fixture(scope="module")
def route_full():
    with _pool() as p:
        return synth.cfg_loop_route_full(pool=p)


def test_frontend_stream_at_the_shipped_lidarslam_yaml(drive20):
    import torch

    from frontend_oracle import OracleFrontendRegistration
    from lidarslam_ros2_amd import DIRECT7, NormalDistributionsTransform

    prm = FrontendParams(vg_size_for_input=0.5, vg_size_for_map=0.1, trans_for_mapupdate=1.5, scan_min_range=1.0, scan_max_range=200.0, num_targeted_cloud=20)

    def replay(reg, device, **kw):
        to_dev = (lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()) if device else None
        fr = FrontendReplay(reg, prm, to_device=to_dev, **kw)
        fr.initialise(drive20["frames"], drive20["frame_poses"], drive20["guess0"])
        out = FrontendResult()
        for scan in drive20["scans"]:
            host = as_pc2_payload(scan)
            fr.receive_cloud(torch.from_numpy(host).cuda() if device else host, int(scan.shape[0]), out, payload_host=host)
        fr.finish(out)
        return out

    def ndt():
        r = NormalDistributionsTransform(device=0)
        r.setResolution(2.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(35); r.setNeighborhoodSearchMethod(DIRECT7)
        return r

    gpu = replay(ndt(), True, mapper=ndt(), builder=ndt(), async_update=True)
    assert len(drive20["frames"]) == 20 and len(gpu.update_at) >= 2
    cpu = replay(OracleFrontendRegistration(2.0, 0.01, 35), False)
    assert gpu.update_at == cpu.update_at and gpu.points_kept == cpu.points_kept
    for j, (a, b) in enumerate(zip(gpu.poses, cpu.poses)):
        dt, ang = pose_delta(a, b)
        assert dt <= TOL_T and ang <= TOL_R, (j, dt, ang)
    assert gpu.iterations == cpu.iterations, (gpu.iterations, cpu.iterations)
    for j, (a, t) in enumerate(zip(gpu.poses, drive20["truths"])):
        dt, ang = pose_delta(a, t)
        assert dt <= 0.05 and ang <= 2e-3, (j, dt, ang)


def _submaps(route):
    import torch

    from lidarslam_ros2_amd import SubMap

    return [SubMap(torch.from_numpy(synth.as_pointxyzi(s["cloud"])).cuda(), s["position"], s["orientation"], s["distance"]) for s in route]


def test_loop_gate_at_the_shipped_lidarslam_yaml_ndt(route_full):
    from lidarslam_ros2_amd import LoopClosureParams, NormalDistributionsTransform, search_loop
    from oracle import oracle

    lp = dict(threshold_loop_closure_score=0.7, distance_loop_closure=100.0, range_of_searching_loop_closure=20.0, search_submap_num=2, voxel_leaf_size=0.1)
    assert route_full[-1]["distance"] > 100.0 and np.median([s["cloud"].shape[0] for s in route_full]) > 50000
    back = NormalDistributionsTransform(0)   # graph_based_slam_component.cpp:64-72
    back.setMaximumIterations(100); back.setResolution(1.0); back.setTransformationEpsilon(0.01)
    edges = search_loop(back, _submaps(route_full), LoopClosureParams(**lp))
    ref = oracle.search_loop(route_full, **lp, ndt_resolution=1.0, trans_eps=0.01, max_iterations=100, num_threads=min(64, oracle.max_threads()))
    assert len(edges) == len(ref) == 1
    e, o = edges[0], ref[0]
    assert e.pair_id == o["pair_id"] and e.n_target_points == o["n_target_points"] and e.accepted == o["accepted"]
    dt, dr = pose_delta(e.final_transformation, o["final"])
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    assert e.fitness_score == pytest.approx(o["fitness_score"], rel=1e-3)
    assert e.iterations == o["iterations"]
    truth = np.linalg.inv(route_full[e.pair_id[0]]["truth"]) @ route_full[-1]["truth"]
    dt, dr = pose_delta(e.relative_pose, truth)
    # the edge removes most of the 0.43 m of drift; with 1 m cells NDT settles 0.12 m from the truth — on the oracle exactly as here
    # (the GICP set below ends within 6 mm on the same route): a property of the method at this resolution, not of the device path
    assert e.accepted and dt < 0.2 and dr < 2e-3, (e.accepted, e.fitness_score, dt, dr)


def test_loop_gate_at_the_shipped_graphbasedslam_yaml_gicp(route_full):
    from lidarslam_ros2_amd import GeneralizedIterativeClosestPoint, LoopClosureParams, search_loop
    from oracle import oracle

    lp = dict(threshold_loop_closure_score=1.5, distance_loop_closure=30.0, range_of_searching_loop_closure=20.0, search_submap_num=3, voxel_leaf_size=0.2)
    gicp = GeneralizedIterativeClosestPoint(0)   # graph_based_slam_component.cpp:74-82
    gicp.setMaxCorrespondenceDistance(30); gicp.setMaximumIterations(100); gicp.setTransformationEpsilon(1e-8)
    gicp.setEuclideanFitnessEpsilon(1e-6); gicp.setRANSACIterations(0)
    edges = search_loop(gicp, _submaps(route_full), LoopClosureParams(**lp))
    e = edges[0]
    assert len(edges) == 1
    # against the oracle with the device's inner solver (Gauss-Newton: north_star) AND with the reference's own (BFGS)
    for solver, tol_t, tol_r in ((1, 1e-4, 1e-5), (0, TOL_T, TOL_R)):
        ref = oracle.search_loop(route_full, **lp, method="gicp", gicp_corr_dist=30.0, gicp_trans_eps=1e-8, max_iterations=100, gicp_solver=solver,
                                 num_threads=min(64, oracle.max_threads()))
        assert len(ref) == 1
        o = ref[0]
        assert e.pair_id == o["pair_id"] and e.n_target_points == o["n_target_points"] and e.accepted == o["accepted"]
        dt, dr = pose_delta(e.final_transformation, o["final"])
        print("gicp gate vs oracle solver", solver, ": dt", dt, "dr", dr, "fitness", e.fitness_score, o["fitness_score"])
        assert dt <= tol_t and dr <= tol_r, (solver, dt, dr)
        assert e.fitness_score == pytest.approx(o["fitness_score"], rel=1e-3)
    truth = np.linalg.inv(route_full[e.pair_id[0]]["truth"]) @ route_full[-1]["truth"]
    dt, dr = pose_delta(e.relative_pose, truth)
    assert e.accepted and dt < 0.03 and dr < 2e-3, (e.accepted, e.fitness_score, dt, dr)
