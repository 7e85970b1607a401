"""Writes tests/golden/gicp_small_golden.npz from the CPU oracle: GICP covariances, both inner solvers' final poses,
the fitness score and a pcl::VoxelGrid result on a small seeded case.  The reference ships no golden vectors and cannot
be built here, so this fixture pins the oracle against drift; regenerate only on purpose."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidarslam_ros2_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

n_source, n_keyframes, leaf = 2500, 3, 0.4
case = synth.small_case(n_source=n_source, n_keyframes=n_keyframes)
tgt = O.voxel_grid_filter(case.target, leaf)
nn_t, nn_s = O.NearestNeighbour(tgt), O.NearestNeighbour(case.source)
ct, cs = O.gicp_covariances(nn_t, tgt, num_threads=1), O.gicp_covariances(nn_s, case.source, num_threads=1)
bfgs = O.gicp_align(nn_t, tgt, ct, case.source, cs, case.guess, solver=0, num_threads=1)
gn = O.gicp_align(nn_t, tgt, ct, case.source, cs, case.guess, solver=1, num_threads=1)
fit = nn_t.fitness_score(case.source, gn["final"], num_threads=1)
idx, d2 = nn_t.search(case.source, case.guess, num_threads=1)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gicp_small_golden.npz"),
                    n_source=n_source, n_keyframes=n_keyframes, leaf=leaf, n_target_raw=case.target.shape[0],
                    n_target=tgt.shape[0], target_head=tgt[:64], source=case.source, guess=case.guess,
                    cov_src_head=cs[:200], cov_tgt_head=ct[:200], nn_idx=idx, nn_d2=d2,
                    final_bfgs=bfgs["final"], iters_bfgs=bfgs["iterations"], final_gn=gn["final"], iters_gn=gn["iterations"],
                    n_corr=gn["n_correspondences"], fitness=fit)
print("wrote gicp golden:", tgt.shape, bfgs["iterations"], gn["iterations"], gn["n_correspondences"], fit)
