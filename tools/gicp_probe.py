"""cfg 3 GICP registration timing (setInputSource + align) on HBM-resident clouds."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
def _make():
    with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
        return synth.cfg_gicp_30k(pool=pool)
from _cache import cached
gc = cached("probe_cfg_gicp_30k", _make)
import torch
from lidarslam_ros2_amd import GeneralizedIterativeClosestPoint
from lidarslam_ros2_amd.posemath import pose_delta
g = GeneralizedIterativeClosestPoint(0); g.setMaxCorrespondenceDistance(5.0); g.setTransformationEpsilon(1e-8)
tgt = torch.from_numpy(synth.as_pointxyzi(gc.target)).cuda(); src = torch.from_numpy(synth.as_pointxyzi(gc.source)).cuda()
g.setInputTarget(tgt); g.setInputSource(src); g.align(gc.guess)
ts, ta = [], []
for _ in range(20):
    t0 = time.perf_counter(); g.setInputSource(src); t1 = time.perf_counter(); g.align(gc.guess); t2 = time.perf_counter()
    ts.append(t1 - t0); ta.append(t2 - t1)
print("GICP cfg3: setInputSource %.3f ms (enqueue only), align %.3f ms, total median %.3f ms" % (1e3 * np.median(ts), 1e3 * np.median(ta), 1e3 * np.median(np.add(ts, ta))), g.last_result,
      pose_delta(g.getFinalTransformation(), gc.truth), flush=True)
