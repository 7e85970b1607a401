"""Round 4: the shared launch chain of a cfg-4 candidate set on its own — targets and sources resident, lsr_align_batch timed by
the host clock; run under `rocprofv3 --kernel-trace` the per-launch durations of the last set are listed by tools/chain_parse.py.
NC = members (64), REPS = timed aligns."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
from _cache import cached
NC = int(os.environ.get("NC", "64"))
REPS = int(os.environ.get("REPS", "5"))
def job(c):
    k = synth.cfg_loop_candidate(c); return k.target, k.source, k.guess, k.truth
def make():
    with mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0)))) as pool:
        return pool.map(job, range(64), chunksize=1)
cands = cached("probe_cfg4_64", make)[:NC]
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, align_batch
from lidarslam_ros2_amd.registration import set_input_target_batch, set_input_source_batch
regs, tg, sr = [], [], []
for t, s, g, tr in cands:
    r = NormalDistributionsTransform(0); r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(100)
    regs.append(r); tg.append(torch.from_numpy(synth.as_pointxyzi(t)).cuda()); sr.append(torch.from_numpy(synth.as_pointxyzi(s)).cuda())
guesses = [c[2] for c in cands]
set_input_target_batch(regs, tg); set_input_source_batch(regs, sr)
torch.cuda.synchronize()
ts, lib_ms = [], []
for rep in range(REPS + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    finals, res = align_batch(regs, guesses)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    lib_ms.append(max(x["gpu_ms"] for x in res))   # the library's own host clock: state upload -> last `done` flag (no Python in it)
ev = [x["n_evaluations"] for x in res]
pts = sum(len(c[1]) for c in cands)
print(f"chain x{NC}: align_batch best {1e3*min(ts[1:]):.3f} ms median {1e3*sorted(ts[1:])[len(ts[1:])//2]:.3f} ms (Python wrapper's clock; inside the library, "
      f"state upload -> last done flag: best {min(lib_ms[1:]):.3f} ms median {sorted(lib_ms[1:])[len(lib_ms[1:])//2]:.3f} ms) | passes max {max(ev)} sum {sum(ev)} | "
      f"{1e3*min(ts[1:])/sum(ev)*1e3:.2f} us per member-pass | mean source {pts/NC:.0f} pts", flush=True)
