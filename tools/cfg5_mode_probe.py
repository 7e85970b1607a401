"""Round 5: the cfg-5 pass (120k-point scan, ndt_resolution RES, default 2.0) under a chosen table mode / kernel form —
LSR_NDT_TABLE_MODE (0 dense global table, 3 per-workgroup tile staged in LDS) and LSR_NDT_QUAD are read by the library itself.
Prints microseconds per derivative pass; run under rocprofv3 --pmc by tools/pmc_cfg5.sh."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
from _cache import cached
def _make5():
    with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
        return synth.cfg_dense_120k(pool=pool)
dense = cached("probe_cfg_dense_120k", _make5)
from lidarslam_ros2_amd import NormalDistributionsTransform
res = float(os.environ.get("RES", "2.0"))
d5 = NormalDistributionsTransform(0); d5.setResolution(res); d5.setTransformationEpsilon(0.0); d5.setMaximumIterations(10)
d5.setInputTarget(dense.target); d5.setInputSource(dense.source)
for i in range(3): d5.align(dense.guess)
d5.setProfiling(True); d5.getProfile(reset=True)
for i in range(4): d5.align(dense.guess)
p = d5.getProfile(reset=True)
print("cfg5 res %.1f table_mode %s quad %s: %.2f us per pass over %d passes | %s" % (res, os.environ.get("LSR_NDT_TABLE_MODE", "auto"), os.environ.get("LSR_NDT_QUAD", "auto"),
      1e3 * p["deriv_ms_total"] / max(1, p["deriv_launches"]), p["deriv_launches"], d5.gridInfo()["n_valid"]), flush=True)
