"""cfg 4 stage breakdown (round 3): the group entries of a candidate set — lsr_set_input_target_batch, lsr_set_input_source_batch,
lsr_align_batch, lsr_get_fitness_score_batch — timed one after the other for NC candidates (host clock, device idle between)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
from _cache import cached
NC = int(os.environ.get("NC", "64"))
def job(c):
    k = synth.cfg_loop_candidate(c); return k.target, k.source, k.guess, k.truth
def make():
    with mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0)))) as pool:
        return pool.map(job, range(NC), chunksize=1)
cands = cached("probe_cfg4_%d" % NC, make)
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, align_batch
from lidarslam_ros2_amd.registration import set_input_target_batch, set_input_source_batch, fitness_score_batch
regs, tg, sr = [], [], []
for t, s, g, tr in cands:
    r = NormalDistributionsTransform(0); r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(100)
    regs.append(r); tg.append(torch.from_numpy(synth.as_pointxyzi(t)).cuda()); sr.append(torch.from_numpy(synth.as_pointxyzi(s)).cuda())
guesses = [c[2] for c in cands]
for sub in sorted({NC, 8, 16}):
    R, T, S, G = regs[:sub], tg[:sub], sr[:sub], guesses[:sub]
    best = None
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        set_input_target_batch(R, T)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        set_input_source_batch(R, S)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        finals, res = align_batch(R, G)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        fit = fitness_score_batch(R)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        row = (t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3)
        if rep and (best is None or row[0] < best[0]): best = row
    ev = [x["n_evaluations"] for x in res]
    print(f"cfg4 x{sub}: setInputTarget {1e3*best[1]:.2f} ms | setInputSource {1e3*best[2]:.2f} | batch align {1e3*best[3]:.2f} (max passes {max(ev)}, sum {sum(ev)}) | "
          f"fitness {1e3*best[4]:.2f} | total {1e3*best[0]:.2f} ms = {sub/best[0]:.0f} reg/s  fit0={fit[0]:.6f}", flush=True)
