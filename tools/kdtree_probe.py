"""cfg 2 (30 000-point scan vs 661 519-point submap, res 5.0) with each pclomp neighbourhood: centroid build time (KDTREE, once per
target) and align time — for DESIGN.md §4 "KDTREE" (not a bench leg: the reference never selects it)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import multiprocessing as mp
import lidarslam_ros2_amd as L
from lidarslam_ros2_amd import synth
from _cache import cached
def _make():
    with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
        return synth.cfg_ndt_30k(pool=pool)
case = cached("probe_cfg2_case", _make)
for name in ("DIRECT7", "DIRECT26", "KDTREE"):
    r = L.NormalDistributionsTransform(0); r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(35)
    r.setInputTarget(case.target); r.setInputSource(case.source)
    t0 = time.perf_counter(); r.setNeighborhoodSearchMethod(getattr(L, name)); r.derivatives(np.zeros(6)); t1 = time.perf_counter()
    ts = []
    for _ in range(12):
        t = time.perf_counter(); r.align(case.guess); ts.append(time.perf_counter() - t)
    res = r.last_result
    print(f"{name:9s}: first derivative call {1e3 * (t1 - t0):.2f} ms (KDTREE: + centroids) | align median {1e3 * np.median(ts[2:]):.3f} ms, "
          f"{res['iterations']} iterations, {res['n_evaluations']} passes -> {1e6 * np.median(ts[2:]) / max(1, res['n_evaluations']):.1f} us per pass", flush=True)
