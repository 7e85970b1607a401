"""Round 4: setInputTarget on the cfg-5 submap (2.5M points) at ndt_resolution 2.0 (22 113 cells: sort-based builder) and 5.0."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lidarslam_ros2_amd import NormalDistributionsTransform, synth
from _cache import cached
c = cached("probe_cfg5", synth.cfg_dense_120k)
import torch
tgt = torch.from_numpy(synth.as_pointxyzi(c.target)).cuda()
torch.cuda.synchronize()
for res in (2.0, 5.0):
    ndt = NormalDistributionsTransform(0); ndt.setResolution(res)
    ndt.setInputTarget(tgt)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); ndt.setInputTarget(tgt); ts.append(time.perf_counter() - t0)
    print(f"cfg5 target ({tgt.shape[0]} pts) res {res}: setInputTarget median {1e3 * np.median(ts):.3f} ms min {1e3 * np.min(ts):.3f} ms", ndt.gridInfo(), flush=True)
