"""Round 4: cfg 5 (120k-pt scan, res 2.0 / 1.0) pass time through the quad kernel and the lane kernel (same bits)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lidarslam_ros2_amd import NormalDistributionsTransform, synth
from _cache import cached
c = cached("probe_cfg5", synth.cfg_dense_120k)
import torch
tgt = torch.from_numpy(synth.as_pointxyzi(c.target)).cuda(); src = torch.from_numpy(synth.as_pointxyzi(c.source)).cuda()
ref = {}
for res in (2.0, 1.0):
    for quad, wg in ((1, 0), (0, 1024), (0, 512)):
        r = NormalDistributionsTransform(0); r.setResolution(res); r.setTransformationEpsilon(0.0); r.setMaximumIterations(30)
        r.setTuning(quad=quad, workgroup=wg)
        r.setInputTarget(tgt); r.setInputSource(src)
        for _ in range(2): r.align(c.guess)
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r.align(c.guess); ts.append(time.perf_counter() - t0)
        lr = r.last_result; T = r.getFinalTransformation()
        same = (res not in ref) or np.array_equal(ref[res], T); ref.setdefault(res, T)
        print(f"res {res} quad {quad} wg {wg}: align {1e3*np.median(ts):.3f} ms, {lr['n_evaluations']} passes, {1e6*np.median(ts)/lr['n_evaluations']:.2f} us/pass, same bits as quad: {same}", flush=True)
