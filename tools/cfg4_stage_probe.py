"""cfg 4, 64 candidates, the C-ABI batch entries called directly (as bench.py does): wall time per entry."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
from _cache import cached
N = int(os.environ.get("NC", "64"))
def _make():
    with mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0)))) as p:
        return [synth.cfg_loop_candidate(c, pool=p) for c in range(N)]
cands = cached("probe_cfg4_cases_%d" % N, _make)
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, _capi
lib = _capi.load()
regs, tg, sr, gs = [], [], [], []
for k in cands:
    r = NormalDistributionsTransform(device=0); r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(100)
    regs.append(r); tg.append(torch.from_numpy(synth.as_pointxyzi(k.target)).cuda()); sr.append(torch.from_numpy(synth.as_pointxyzi(k.source)).cuda())
    gs.append(np.ascontiguousarray(np.asarray(k.guess, np.float32).T).reshape(16))
torch.cuda.synchronize()
hs = (C.c_void_p * N)(*[r._h for r in regs])
tptr = (C.c_void_p * N)(*[C.c_void_p(t.data_ptr()) for t in tg]); tcnt = (C.c_size_t * N)(*[int(t.shape[0]) for t in tg])
sptr = (C.c_void_p * N)(*[C.c_void_p(t.data_ptr()) for t in sr]); scnt = (C.c_size_t * N)(*[int(t.shape[0]) for t in sr])
G = np.ascontiguousarray(np.stack(gs), np.float32); fptr = C.POINTER(C.c_float)
finals = np.zeros((N, 16), np.float32); res = (_capi.Result * N)(); fit = (C.c_double * N)()
T = {k: [] for k in ("target", "source", "align", "fitness")}
for it in range(8):
    t0 = time.perf_counter(); _capi.check(lib.lsr_set_input_target_batch(hs, N, tptr, tcnt, 32, 1), "t")
    t1 = time.perf_counter(); _capi.check(lib.lsr_set_input_source_batch(hs, N, sptr, scnt, 32, 1), "s")
    t2 = time.perf_counter(); _capi.check(lib.lsr_align_batch(hs, N, G.ctypes.data_as(fptr), finals.ctypes.data_as(fptr), res), "a")
    t3 = time.perf_counter(); _capi.check(lib.lsr_get_fitness_score_batch(hs, N, C.c_double(1.7976931348623157e308), fit), "f")
    t4 = time.perf_counter()
    if it >= 2:
        T["target"].append(t1 - t0); T["source"].append(t2 - t1); T["align"].append(t3 - t2); T["fitness"].append(t4 - t3)
tf = []
for it in range(8):
    _capi.check(lib.lsr_set_input_target_batch(hs, N, tptr, tcnt, 32, 1), "t"); _capi.check(lib.lsr_set_input_source_batch(hs, N, sptr, scnt, 32, 1), "s")
    t0 = time.perf_counter()
    _capi.check(lib.lsr_align_fitness_batch(hs, N, G.ctypes.data_as(fptr), finals.ctypes.data_as(fptr), res, C.c_double(1.7976931348623157e308), fit), "af")
    tf.append(time.perf_counter() - t0)
print("cfg4 x%d lsr_align_fitness_batch (median ms): %.3f  (align + fitness as two calls: %.3f)" % (N, 1e3 * np.median(tf[2:]), 1e3 * (np.median(T["align"]) + np.median(T["fitness"]))), flush=True)
print("cfg4 x%d C entries (median ms): " % N + " | ".join("%s %.3f" % (k, 1e3 * np.median(v)) for k, v in T.items()) +
      " | total %.3f" % (1e3 * sum(np.median(v) for v in T.values())), flush=True)
