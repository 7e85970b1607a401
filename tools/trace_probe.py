"""rocprofv3 target: chained aligns of one cfg-2 scan, a batch of 16, and one cfg-5 scan (120k points, res 2.0: the pass
that gathers from the dense global table) — the launches tools/pmc_ndt_parse.py tells apart by their grid size."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
def _make():
    with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
        return synth.cfg_ndt_30k(pool=pool)
from _cache import cached
case = cached("probe_cfg_ndt_30k", _make)
from lidarslam_ros2_amd import NormalDistributionsTransform, align_batch
ndt = NormalDistributionsTransform(0); ndt.setResolution(5.0); ndt.setTransformationEpsilon(0.0); ndt.setMaximumIterations(30)
ndt.setInputTarget(case.target); ndt.setInputSource(case.source)
for i in range(4): ndt.align(case.guess)
print(ndt.last_result)
B = int(os.environ.get("LSR_B", "16"))
regs = [ndt] + [NormalDistributionsTransform(0) for _ in range(B - 1)]
for r in regs[1:]:
    r.setResolution(5.0); r.setTransformationEpsilon(0.0); r.setMaximumIterations(30); r.shareTargetOf(ndt)
for r in regs: r.setInputSource(case.source)
for i in range(3): align_batch(regs, [case.guess] * B)

if os.environ.get("LSR_PROBE_CFG5", "1") == "1":
    def _make5():
        with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
            return synth.cfg_dense_120k(pool=pool)
    dense = cached("probe_cfg_dense_120k", _make5)
    d5 = NormalDistributionsTransform(0); d5.setResolution(2.0); d5.setTransformationEpsilon(0.01)
    d5.setInputTarget(dense.target); d5.setInputSource(dense.source)
    for i in range(6): d5.align(dense.guess)
    print(d5.last_result)
