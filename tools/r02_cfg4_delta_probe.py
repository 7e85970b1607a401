"""cfg 4: per-candidate difference between the GPU registrations (batched chain, and one by one) and the CPU-oracle fixture."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
from lidarslam_ros2_amd.posemath import pose_delta
def job(c):
    k = synth.cfg_loop_candidate(c); return k.target, k.source, k.guess, k.truth
with mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0)))) as pool:
    cands = pool.map(job, range(64), chunksize=1)
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, align_batch
from lidarslam_ros2_amd.registration import set_input_target_batch
fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfg4_candidates_oracle.npz"))
regs = []
for t, s, g, tr in cands:
    r = NormalDistributionsTransform(0); r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(100); regs.append(r)
set_input_target_batch(regs, [c[0] for c in cands])
for r, c in zip(regs, cands): r.setInputSource(c[1])
finals, res = align_batch(regs, [c[2] for c in cands])
single = []
for r, c in zip(regs, cands):
    r.align(c[2]); single.append((r.getFinalTransformation().copy(), dict(r.last_result)))
for c in range(64):
    db = pose_delta(finals[c], fx["final"][c]); ds = pose_delta(single[c][0], fx["final"][c]); dbs = pose_delta(finals[c], single[c][0])
    if max(db[0], ds[0]) > 2e-4 or res[c]["iterations"] != fx["iterations"][c] or single[c][1]["iterations"] != fx["iterations"][c]:
        print(f"cand {c}: batch-vs-cpu {db[0]:.2e} m {db[1]:.2e} rad | single-vs-cpu {ds[0]:.2e} {ds[1]:.2e} | batch-vs-single {dbs[0]:.2e} | "
              f"it cpu {fx['iterations'][c]} batch {res[c]['iterations']} single {single[c][1]['iterations']} | evals batch {res[c]['n_evaluations']} single {single[c][1]['n_evaluations']} | "
              f"err-vs-truth cpu {pose_delta(fx['final'][c], fx['truth'][c])[0]:.3f} batch {pose_delta(finals[c], fx['truth'][c])[0]:.3f}", flush=True)
print("done")
