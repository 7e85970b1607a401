"""Per-stage timing of what the loop gate does with one candidate (52k-pt window target, 18k-pt source): setInputTarget,
setInputSource, align, getFitnessScore, each on its own — runs unchanged in the round-2 tree for A/B."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import synth
from _cache import cached
route = cached("probe_loop_route", synth.make_loop_route)
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform
rng = np.random.default_rng(3)
tgt_np = np.concatenate([s["cloud"][:, :3] for s in route[:5]])[:52000].astype(np.float32)
src_np = route[20]["cloud"][:18000, :3].astype(np.float32)
tgt = torch.from_numpy(synth.as_pointxyzi(tgt_np)).cuda(); src = torch.from_numpy(synth.as_pointxyzi(src_np)).cuda()
r = NormalDistributionsTransform(device=0); r.setMaximumIterations(100); r.setResolution(5.0); r.setTransformationEpsilon(0.01)
T = {k: [] for k in ("target", "source", "align", "fitness")}
for it in range(45):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); r.setInputTarget(tgt); t1 = time.perf_counter(); r.setInputSource(src); t2 = time.perf_counter()
    r.align(np.eye(4, dtype=np.float32)); t3 = time.perf_counter(); f = r.getFitnessScore(); t4 = time.perf_counter()
    if it >= 5:
        T["target"].append(t1 - t0); T["source"].append(t2 - t1); T["align"].append(t3 - t2); T["fitness"].append(t4 - t3)
print("stages (median us): " + " | ".join("%s %.1f" % (k, 1e6 * np.median(v)) for k, v in T.items()) + " | iterations %d fitness %.5f" % (r.last_result["iterations"], f), flush=True)
