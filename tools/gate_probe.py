"""Loop-closure gate (lsr_search_loop, top_k = 1) timing: the bench's loop_gate leg on its own."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import synth
from _cache import cached
route = cached("probe_loop_route", synth.make_loop_route)
import torch
from lidarslam_ros2_amd import LoopClosureParams, NormalDistributionsTransform, SubMap, search_loop
sms = [SubMap(torch.from_numpy(synth.as_pointxyzi(s["cloud"])).cuda(), s["position"], s["orientation"], s["distance"]) for s in route]
lp = dict(threshold_loop_closure_score=1.0, distance_loop_closure=20.0, range_of_searching_loop_closure=10.0, search_submap_num=2, voxel_leaf_size=0.2)
back = NormalDistributionsTransform(device=0)
back.setMaximumIterations(100); back.setResolution(5.0); back.setTransformationEpsilon(0.01)
for _ in range(3):
    edges = search_loop(back, sms, LoopClosureParams(**lp))
torch.cuda.synchronize()
ts = []
for _ in range(40):
    t0 = time.perf_counter(); edges = search_loop(back, sms, LoopClosureParams(**lp)); ts.append(time.perf_counter() - t0)
print("loop gate: median %.3f ms p10 %.3f p90 %.3f | fitness %.6f accepted %s | env NN_FROM_GRID=%s" % (
    1e3 * np.median(ts), 1e3 * np.percentile(ts, 10), 1e3 * np.percentile(ts, 90), edges[0].fitness_score, edges[0].accepted,
    os.environ.get("LSR_NN_FROM_GRID", "1")), flush=True)
