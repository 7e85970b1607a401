"""Frontend latency probe (cfg 1 style: eps 0.01, converges in a few iterations): wall time of
setInputSource(device)+align over several different scans, with the launch-chunk schedule under test."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, DIRECT7, synth

cases = [synth.cfg_ndt_30k(seed=0)] * 3
ndt = NormalDistributionsTransform(0); ndt.setResolution(5.0); ndt.setTransformationEpsilon(0.01); ndt.setNeighborhoodSearchMethod(DIRECT7)
ndt.setInputTarget(torch.from_numpy(synth.as_pointxyzi(cases[0].target)).cuda())
srcs = [torch.from_numpy(synth.as_pointxyzi(c.source)).cuda() for c in cases]
# perturbed guesses to get a spread of iteration counts
rng = np.random.default_rng(0)
guesses = []
for c in cases:
    for k in range(4):
        g = c.guess.copy(); g[:2, 3] += rng.uniform(-0.3, 0.3, 2).astype(np.float32) * k / 3
        guesses.append(g)
for _ in range(3):
    ndt.setInputSource(srcs[0]); ndt.align(guesses[0])
rows = []
for rep in range(9):
    for i, g in enumerate(guesses):
        s = srcs[0]  # same target frame; all guesses are for scans near keyframe 10
        t0 = time.perf_counter(); ndt.setInputSource(s); ndt.align(g); t = time.perf_counter() - t0
        r = ndt.last_result
        rows.append((i, r["n_evaluations"], r["iterations"], r["gpu_ms"], t * 1e3))
rows = np.array(rows)
for i in range(len(guesses)):
    m = rows[rows[:, 0] == i]
    print("guess %2d evals %3d iters %2d gpu %.3f ms wall median %.3f ms" % (i, m[0, 1], m[0, 2], np.median(m[:, 3]), np.median(m[:, 4])))
print("ALL: wall median %.3f ms mean %.3f ms ; gpu mean %.3f" % (np.median(rows[:, 4]), rows[:, 4].mean(), rows[:, 3].mean()), "chunks", os.environ.get("LSR_CHUNK0"), os.environ.get("LSR_CHUNK"))
