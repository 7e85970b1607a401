"""BASELINE cfg 5: 120k-pt 64-line scan vs 20-frame submap, NDT res 2.0 — GPU timing + CPU oracle timing."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import NormalDistributionsTransform, synth, align_batch
from lidarslam_ros2_amd.posemath import pose_delta
from oracle import oracle as O
t0 = time.time(); c = synth.cfg_dense_120k(); print("gen %.1fs" % (time.time() - t0), c.target.shape, c.source.shape, flush=True)
for eps, mi in ((0.01, 35), (0.0, 30)):
    r = NormalDistributionsTransform(0); r.setResolution(2.0); r.setTransformationEpsilon(eps); r.setMaximumIterations(mi)
    t0 = time.perf_counter(); r.setInputTarget(c.target); t1 = time.perf_counter(); r.setInputTarget(c.target); t2 = time.perf_counter()
    r.setInputSource(c.source)
    for _ in range(2): r.align(c.guess)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r.setInputSource(c.source); r.align(c.guess); ts.append(time.perf_counter() - t0)
    res = r.last_result
    print("eps", eps, "setInputTarget %.2f ms (warm %.2f) grid" % ((t1-t0+0)*0 + (t2-t1)*1e3, (t2-t1)*1e3), r.gridInfo()["n_valid"],
          "| GPU align %.3f ms, %d iterations, %d passes, %.1f us/pass" % (np.median(ts)*1e3, res["iterations"], res["n_evaluations"], np.median(ts)*1e6/res["n_evaluations"]), flush=True)
    g = O.VoxelGridCovariance(c.target, 2.0)
    tc = time.perf_counter(); ref = O.ndt_align(g, c.source, c.guess, resolution=2.0, trans_eps=eps, max_iterations=mi, num_threads=64); tc = time.perf_counter() - tc
    print("   CPU oracle (64 threads) %.1f ms, %d iterations; GPU vs CPU" % (tc*1e3, ref["iterations"]), pose_delta(r.getFinalTransformation(), ref["final"]), "vs truth", pose_delta(r.getFinalTransformation(), c.truth), flush=True)
