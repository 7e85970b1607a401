"""cfg 4 stage breakdown: per-candidate setInputTarget / setInputSource, the shared-launch batch align, getFitnessScore."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
NC = int(os.environ.get("NC", "16"))
def job(c):
    k = synth.cfg_loop_candidate(c); return k.target, k.source, k.guess, k.truth
with mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0)))) as pool:
    cands = pool.map(job, range(NC), chunksize=1)
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, _capi, align_batch
from lidarslam_ros2_amd.registration import set_input_target_batch, fitness_score_batch
OWN = os.environ.get('OWN_STREAMS', '1') == '1'
lib = _capi.load()
st = torch.cuda.current_stream().cuda_stream
regs, tg, sr = [], [], []
for t, s, g, tr in cands:
    r = NormalDistributionsTransform(0, stream=(None if OWN else st)); r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(100)
    regs.append(r); tg.append(torch.from_numpy(synth.as_pointxyzi(t)).cuda()); sr.append(torch.from_numpy(synth.as_pointxyzi(s)).cuda())
guesses = [c[2] for c in cands]
for mode in (os.environ.get("LSR_NDT_TABLE_MODE", "auto"),):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if OWN: set_input_target_batch(regs, tg)
        else:
            for r, t in zip(regs, tg): r.setInputTarget(t)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for r, s in zip(regs, sr): r.setInputSource(s)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        finals, res = align_batch(regs, guesses)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        fit = fitness_score_batch(regs) if OWN else [r.getFitnessScore() for r in regs]
        torch.cuda.synchronize(); t4 = time.perf_counter()
    its = [x["iterations"] for x in res]; ev = [x["n_evaluations"] for x in res]
    print(f"cfg4 x{NC} own_streams={OWN} fit0={fit[0]:.6f}: setInputTarget {1e3*(t1-t0):.2f} ms | setInputSource {1e3*(t2-t1):.2f} | batch align {1e3*(t3-t2):.2f} (max passes {max(ev)}, sum {sum(ev)}, iters max {max(its)}) | "
          f"fitness {1e3*(t4-t3):.2f} | total {1e3*(t4-t0):.2f} ms = {NC/(t4-t0):.0f} reg/s", flush=True)
