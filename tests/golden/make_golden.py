"""Writes tests/golden/ndt_small_golden.npz from the CPU oracle (the reference ships no golden vectors
and cannot be built here — SURVEY.md §8c — so these pin the oracle and give the GPU tests a fixture
that does not depend on the oracle library being rebuilt identically)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidarslam_ros2_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

n_source, n_keyframes, res = 3000, 3, 4.0
case = synth.small_case(n_source=n_source, n_keyframes=n_keyframes)
grid = O.VoxelGridCovariance(case.target, res)
p = O.matrix_to_pose(case.guess) + np.array([0.12, -0.05, 0.02, 0.003, -0.004, 0.008])
s, g, H = O.ndt_derivatives(grid, case.source, p, resolution=res, num_threads=1)
a = O.ndt_align(grid, case.source, case.guess, resolution=res, trans_eps=0.01, num_threads=1)
b = O.ndt_align(grid, case.source, case.guess, resolution=res, trans_eps=1e-6, max_iterations=30, num_threads=1)
d = grid.dump()
# the fourth pclomp neighbourhood (KDTREE: radius search over the leaves' FLOAT centroids), round 6
sk, gk, Hk = O.ndt_derivatives(grid, case.source, p, resolution=res, search=0, num_threads=1)
ak = O.ndt_align(grid, case.source, case.guess, resolution=res, trans_eps=0.01, search=0, num_threads=1)
cen = grid.centroids()
cen[d["n"] < 6] = np.nan     # leaves that are not in the kd-tree
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ndt_small_golden.npz"),
                    n_source=n_source, n_keyframes=n_keyframes, res=res, n_target=case.target.shape[0],
                    source=case.source, guess=case.guess, p=p, score=s, grad=g, hess=H,
                    final_eps001=a["final"], iters_eps001=a["iterations"], final_tight=b["final"], iters_tight=b["iterations"],
                    leaf_idx=d["idx"], leaf_n=d["n"], min_b=grid.min_b, max_b=grid.max_b,
                    score_kdtree=sk, grad_kdtree=gk, hess_kdtree=Hk, final_kdtree=ak["final"], iters_kdtree=ak["iterations"], leaf_centroid=cen)
print("wrote golden:", s, a["iterations"], b["iterations"], len(d["idx"]))
