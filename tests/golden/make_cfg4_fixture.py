"""Generates tests/golden/cfg4_candidates_oracle.npz: the CPU oracle's result for the 64 loop-closure candidates of
BASELINE cfg 4 (synth.cfg_loop_candidate(0..63); NDT res 5.0, DIRECT7, transformation_epsilon 0.01, max_iterations 100 —
the backend's settings, graph_based_slam_component.cpp:64-72) and the fitness score at the oracle's pose.
bench.py holds every candidate of its cfg4_loop_batch leg to this file (pose, Newton iterations, fitness).

    python tests/golden/make_cfg4_fixture.py          # ~10 min on 8 cores
"""
import multiprocessing as mp
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lidarslam_ros2_amd import synth  # noqa: E402


def job(c):
    from oracle import oracle as O

    k = synth.cfg_loop_candidate(c)
    ref = O.ndt_align(O.VoxelGridCovariance(k.target, 5.0), k.source, k.guess, resolution=5.0, trans_eps=0.01, max_iterations=100,
                      num_threads=1)
    fit = O.NearestNeighbour(k.target, 1.0).fitness_score(k.source, ref["final"], num_threads=1)
    return c, np.asarray(ref["final"], np.float64), int(ref["iterations"]), bool(ref["converged"]), float(fit), np.asarray(k.truth, np.float64)


if __name__ == "__main__":
    with mp.get_context("fork").Pool(len(os.sched_getaffinity(0))) as p:
        out = p.map(job, range(64), chunksize=1)
    out.sort(key=lambda r: r[0])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg4_candidates_oracle.npz"),
                        final=np.stack([r[1] for r in out]), iterations=np.array([r[2] for r in out], np.int32),
                        converged=np.array([r[3] for r in out], np.bool_), fitness=np.array([r[4] for r in out], np.float64),
                        truth=np.stack([r[5] for r in out]))
    print("iterations", [r[2] for r in out])
