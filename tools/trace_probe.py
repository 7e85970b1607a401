"""rocprofv3 target: diagnostic derivative passes then chained aligns (single + batch)."""
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
