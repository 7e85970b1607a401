"""Ad-hoc GPU probe: cfg 1/2 sized NDT registration, timing + parity vs the CPU oracle."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import NormalDistributionsTransform, DIRECT7, synth, align_batch
from oracle import oracle as O

from lidarslam_ros2_amd.posemath import pose_delta
t0 = time.time(); case = synth.cfg_ndt_30k(); print("gen %.1fs" % (time.time() - t0), case.target.shape, case.source.shape, flush=True)
res = 5.0
tgt = synth.as_pointxyzi(case.target); src = synth.as_pointxyzi(case.source)
for eps, mi in ((0.01, 35), (0.0, 30)):
    ndt = NormalDistributionsTransform(0); ndt.setResolution(res); ndt.setTransformationEpsilon(eps); ndt.setMaximumIterations(mi)
    ndt.setNeighborhoodSearchMethod(DIRECT7)
    t0 = time.time(); ndt.setInputTarget(tgt); t1 = time.time(); ndt.setInputTarget(tgt); t2 = time.time()
    print("setInputTarget first %.2f ms, second %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), ndt.gridInfo())
    ndt.setInputSource(src)
    for _ in range(3): ndt.align(case.guess)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); ndt.setInputSource(src); ndt.align(case.guess); ts.append(time.perf_counter() - t0)
    r = ndt.last_result
    print("eps", eps, "GPU align median %.3f ms  min %.3f ms" % (np.median(ts) * 1e3, np.min(ts) * 1e3), r)
    T = ndt.getFinalTransformation()
    t0 = time.time(); g = O.VoxelGridCovariance(case.target, res); t1 = time.time()
    ref = O.ndt_align(g, case.source, case.guess, resolution=res, trans_eps=eps, max_iterations=mi); t2 = time.time()
    print("CPU oracle grid %.1f ms align %.1f ms threads %d iters %d evals %d/%d/%d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, O.max_threads(), ref["iterations"], ref["n_evals"], ref["n_evals_grad"], ref["n_hessian_recompute"]))
    print("pose delta GPU vs oracle:", pose_delta(T, ref["final"]), " vs truth:", pose_delta(T, case.truth), flush=True)
    # profile
    ndt.setProfiling(True); ndt.getProfile(reset=True); ndt.align(case.guess); p = ndt.getProfile(); ndt.setProfiling(False)
    print("profile", p, "avg deriv us", p["deriv_ms_total"] / max(1, p["deriv_launches"]) * 1e3)
# batch
B = 16
lead = NormalDistributionsTransform(0); lead.setResolution(res); lead.setTransformationEpsilon(0.0); lead.setMaximumIterations(30)
lead.setInputTarget(tgt)
regs = [lead] + [NormalDistributionsTransform(0) for _ in range(B - 1)]
for r in regs[1:]:
    r.setResolution(res); r.setTransformationEpsilon(0.0); r.setMaximumIterations(30); r.shareTargetOf(lead)
for r in regs: r.setInputSource(src)
gs = [case.guess] * B
for _ in range(2): align_batch(regs, gs)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); finals, results = align_batch(regs, gs); ts.append(time.perf_counter() - t0)
print("batch", B, "median %.3f ms -> %.0f reg/s" % (np.median(ts) * 1e3, B / np.median(ts)), results[0])
