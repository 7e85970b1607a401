"""rocprofv3 target: setInputTarget (K1/K2) on the HBM-resident cfg-2 submap, both builders."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, synth
import multiprocessing as mp
def _make():
    with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
        return synth.cfg_ndt_30k(pool=pool)
from _cache import cached
case = cached("probe_cfg_ndt_30k", _make)
tgt = torch.from_numpy(synth.as_pointxyzi(case.target)).cuda()
torch.cuda.synchronize()
for builder in (0, 1):
    ndt = NormalDistributionsTransform(0); ndt.setResolution(5.0); ndt.setTuning(grid_builder=builder)
    ndt.setInputTarget(tgt)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); ndt.setInputTarget(tgt); ts.append(time.perf_counter() - t0)
    print(f"setInputTarget builder {builder}: median {1e3 * np.median(ts):.3f} ms min {1e3 * np.min(ts):.3f} ms", ndt.gridInfo(), flush=True)
