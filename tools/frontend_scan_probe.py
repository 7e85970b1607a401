"""One frontend scan at the reference's settings — lsr_set_input_source_pc2 (raw 147k-point payload in HBM) + lsr_align (eps 0.01) — a
few dozen times with a pause between them, for a kernel timeline of ONE scan (tools/timeline.py, GAP_US below the pause).  NOPAUSE=1:
back to back (the next scan's first kernel then queues behind the launches the previous align left in its stream)."""
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
from lidarslam_ros2_amd import NormalDistributionsTransform, DIRECT7
from lidarslam_ros2_amd.frontend import as_pc2_payload
raw = case.raw_source
payload = torch.from_numpy(as_pc2_payload(raw)).cuda(); torch.cuda.synchronize()
r = NormalDistributionsTransform(0); r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(35); r.setNeighborhoodSearchMethod(DIRECT7)
r.setInputTarget(torch.from_numpy(synth.as_pointxyzi(case.target)).cuda())
g = np.asarray(case.guess, np.float32)
ts, ta = [], []
for _ in range(int(os.environ.get("REPS", "30"))):
    if os.environ.get("NOPAUSE") != "1":
        torch.cuda.synchronize(); time.sleep(0.002)
    t0 = time.perf_counter(); r.setInputSourcePointCloud2(payload, raw.shape[0], 32, (0, 4, 8, 16), 0.1, 100.0, 0.2); t1 = time.perf_counter(); r.align(g); t2 = time.perf_counter()
    ts.append(t1 - t0); ta.append(t2 - t1)
print("frontend scan: source %.1f us + align %.1f us = %.1f us (medians; %d Newton iterations)" %
      (1e6 * np.median(ts[3:]), 1e6 * np.median(ta[3:]), 1e6 * np.median(np.add(ts, ta)[3:]), r.getFinalNumIteration()), flush=True)
