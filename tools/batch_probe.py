import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import NormalDistributionsTransform, synth, align_batch
case = synth.cfg_ndt_30k()
def mk():
    r = NormalDistributionsTransform(0); r.setResolution(5.0); r.setTransformationEpsilon(0.0); r.setMaximumIterations(30); return r
lead = mk(); lead.setInputTarget(case.target)
for B in [int(b) for b in os.environ.get("LSR_BS", "1,4,16,64").split(",")]:
    regs = [lead] + [mk() for _ in range(B - 1)]
    for r in regs[1:]: r.shareTargetOf(lead)
    for r in regs: r.setInputSource(case.source)
    gs = [case.guess] * B
    for _ in range(2): align_batch(regs, gs)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); f, res = align_batch(regs, gs); ts.append(time.perf_counter() - t0)
    print("B=%d median %.3f ms -> %.0f reg/s, %.1f us/pass (%d passes)" % (B, np.median(ts)*1e3, B/np.median(ts), np.median(ts)*1e6/res[0]['n_evaluations'], res[0]['n_evaluations']), flush=True)
