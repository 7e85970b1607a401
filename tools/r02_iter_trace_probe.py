"""Per-iteration trajectory of the GPU NDT registration of chosen cfg-4 candidates: final pose and derivative-pass count
with max_iterations = 1, 2, ... (the controller is deterministic, so run j reproduces the first j iterations of run j+1)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
CANDS = [int(x) for x in os.environ.get("CANDS", "21,34").split(",")]
out = {}
for c in CANDS:
    with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
        k = synth.cfg_loop_candidate(c, pool=pool)
    from lidarslam_ros2_amd import NormalDistributionsTransform
    r = NormalDistributionsTransform(0); r.setResolution(5.0); r.setTransformationEpsilon(0.01)
    r.setInputTarget(k.target); r.setInputSource(k.source)
    rows = []
    for j in range(1, 40):
        r.setMaximumIterations(j)
        r.align(k.guess)
        res = r.last_result
        rows.append({"max_it": j, "it": res["iterations"], "evals": res["n_evaluations"], "conv": bool(res["converged"]), "score": res["score"],
                     "T": np.asarray(r.getFinalTransformation(), np.float64).tolist()})
        if res["iterations"] < j:
            break
    out[str(c)] = rows
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r02_iter_trace.json"), "w"))
print("ok", {c: len(v) for c, v in out.items()})
