"""Frontend source preprocessing (range filter + VoxelGrid(0.2) + setInputSource) on a raw 147k-point scan, and the loop gate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
from _cache import cached
def _c2():
    with mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0)))) as p: return synth.cfg_ndt_30k(pool=p, keep_parts=True)
case = cached("probe_cfg_ndt_30k_parts", _c2)
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform
raw = torch.from_numpy(synth.as_pointxyzi(case.raw_source)).cuda(); torch.cuda.synchronize()
r = NormalDistributionsTransform(0); r.setResolution(5.0)
for _ in range(5): n = r.setInputSourceFrontend(raw, 0.1, 100.0, 0.2)
ts, tr = [], []   # to the end of the device work / until the call returns (the centroid launch may still be running)
for _ in range(40):
    t0 = time.perf_counter(); n = r.setInputSourceFrontend(raw, 0.1, 100.0, 0.2); t1 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); tr.append(t1 - t0)
print("preprocess: %d -> %d points, median %.1f us p10 %.1f (call returns after %.1f) | voxel filter form %d (2 = grid dimensions on the device) | LSR_VG_DEVICE_DIMS=%s LSR_VG_SORT=%s" %
      (raw.shape[0], n, 1e6 * np.median(ts), 1e6 * np.percentile(ts, 10), 1e6 * np.median(tr), r.voxelFilterForm(), os.environ.get("LSR_VG_DEVICE_DIMS", "-"), os.environ.get("LSR_VG_SORT", "-")), flush=True)
