"""Round-2 A/B probe: cfg 2 / cfg 1 registrations through every launch variant of the derivative pass, and
setInputTarget through both grid builders (HBM-resident submap)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, synth
import multiprocessing as mp
with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
    case = synth.cfg_ndt_30k(pool=pool)
tgt = torch.from_numpy(synth.as_pointxyzi(case.target)).cuda()
src = torch.from_numpy(synth.as_pointxyzi(case.source)).cuda()
torch.cuda.synchronize()
for builder in (0, 1):
    ndt = NormalDistributionsTransform(0); ndt.setResolution(5.0); ndt.setTuning(grid_builder=builder)
    ndt.setInputTarget(tgt)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); ndt.setInputTarget(tgt); ts.append(time.perf_counter() - t0)
    print(f"setInputTarget builder {builder}: median {1e3 * np.median(ts):.3f} ms min {1e3 * np.min(ts):.3f} ms", ndt.gridInfo(), flush=True)
for quad, wg, tab in ((1, 128, 2), (1, 64, 2), (1, 128, 0), (0, 256, 2), (0, 256, 0)):
    for eps, mi, name, reps in ((0.0, 30, "cfg2", 10), (0.01, 35, "cfg1", 40)):
        ndt = NormalDistributionsTransform(0); ndt.setResolution(5.0); ndt.setTransformationEpsilon(eps); ndt.setMaximumIterations(mi)
        ndt.setTuning(workgroup=wg, table_mode=tab, quad=quad)
        ndt.setInputTarget(tgt); ndt.setInputSource(src)
        for _ in range(3): ndt.align(case.guess)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); ndt.setInputSource(src); ndt.align(case.guess); ts.append(time.perf_counter() - t0)
        r = ndt.last_result
        print(f"quad {quad} wg {wg} tab {tab} {name}: median {1e3 * np.median(ts):.3f} ms ({1e6 * np.median(ts) / max(1, r['n_evaluations']):.2f} us/pass) "
              f"min {1e3 * np.min(ts):.3f} ms  it {r['iterations']} passes {r['n_evaluations']} score {r['score']:.9f}", flush=True)
