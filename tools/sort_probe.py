import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import NormalDistributionsTransform, synth, align_batch
case = synth.cfg_ndt_30k()
def mk():
    r = NormalDistributionsTransform(0); r.setResolution(5.0); r.setTransformationEpsilon(0.0); r.setMaximumIterations(30); return r
lead = mk(); lead.setInputTarget(case.target)
G = case.guess
pt = case.source @ G[:3, :3].T + G[:3, 3]
c = np.floor(pt / 5.0).astype(np.int64)
key = (c[:, 2] - c[:, 2].min()) * 10000 * 10000 + (c[:, 1] - c[:, 1].min()) * 10000 + (c[:, 0] - c[:, 0].min())
order_cell = np.argsort(key, kind="stable")
rng = np.random.default_rng(0)
variants = {"voxelgrid-order": case.source, "cell-sorted": case.source[order_cell], "shuffled": case.source[rng.permutation(len(case.source))]}
for name, src in variants.items():
    for B in (1, 16):
        regs = [lead] + [mk() for _ in range(B - 1)]
        for r in regs[1:]: r.shareTargetOf(lead)
        for r in regs: r.setInputSource(src)
        gs = [G] * B
        for _ in range(2): align_batch(regs, gs)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); f, res = align_batch(regs, gs); ts.append(time.perf_counter() - t0)
        print("%-16s B=%2d median %.3f ms  %.2f us/pass (%d passes)" % (name, B, np.median(ts)*1e3, np.median(ts)*1e6/res[0]['n_evaluations'], res[0]['n_evaluations']), flush=True)
