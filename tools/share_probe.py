"""Round 5: ONE share of the cfg-4 candidate set (what one GPU of an 8-GPU node runs: NC = 8 candidates starting at FIRST) through the
C-ABI batch entries, stage by stage and as the whole per-share path bench.py's `projected_8gpu` times
(lsr_set_input_target_batch + lsr_set_input_source_batch + lsr_align_batch_sharded with a one-rank communicator).
MODE=chain: only lsr_align_batch REPS times (targets and sources resident) — the run tools/chain_parse.py reads under
`rocprofv3 --kernel-trace`.  MODE=share: only the whole path REPS times (for a kernel trace of a whole share)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
from _cache import cached
N = int(os.environ.get("NC", "8"))
FIRST = int(os.environ.get("FIRST", "0"))
REPS = int(os.environ.get("REPS", "8"))
MODE = os.environ.get("MODE", "stages")
def _make():
    with mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0)))) as p:
        return [synth.cfg_loop_candidate(c, pool=p) for c in range(FIRST, FIRST + N)]
cands = cached("probe_cfg4_share_%d_%d" % (FIRST, N), _make)
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
BIG = C.c_double(1.7976931348623157e308)
comm = C.c_void_p()
_capi.check(lib.lsr_comm_create(None, 0, 1, 0, C.byref(comm)), "lsr_comm_create")
recs = (_capi.ShardRecord * N)()

def stage_all():
    _capi.check(lib.lsr_set_input_target_batch(hs, N, tptr, tcnt, 32, 1), "t")
    _capi.check(lib.lsr_set_input_source_batch(hs, N, sptr, scnt, 32, 1), "s")

def whole():
    t0 = time.perf_counter()
    stage_all()
    _capi.check(lib.lsr_align_batch_sharded(comm, hs, N, N, G.ctypes.data_as(fptr), 1, recs), "a")
    return time.perf_counter() - t0

if MODE == "chain":
    stage_all()
    ts = []
    for it in range(REPS + 1):
        t0 = time.perf_counter(); _capi.check(lib.lsr_align_batch(hs, N, G.ctypes.data_as(fptr), finals.ctypes.data_as(fptr), res), "a"); ts.append(time.perf_counter() - t0)
    ev = [int(r.n_evaluations) for r in res]
    print("chain x%d first %d: align_batch best %.3f median %.3f ms | passes %s (max %d sum %d) | launches x %.2f us" %
          (N, FIRST, 1e3 * min(ts[1:]), 1e3 * float(np.median(ts[1:])), ev, max(ev), sum(ev), 1e6 * min(ts[1:]) / (max(ev) + 1)), flush=True)
    sys.exit(0)
if MODE == "fitness":   # the fitness stage alone, REPS times (for a kernel trace of its launches)
    stage_all()
    _capi.check(lib.lsr_align_batch(hs, N, G.ctypes.data_as(fptr), finals.ctypes.data_as(fptr), res), "a")
    ts = []
    for it in range(REPS + 2):
        t0 = time.perf_counter(); _capi.check(lib.lsr_get_fitness_score_batch(hs, N, BIG, fit), "f"); ts.append(time.perf_counter() - t0)
    print("fitness x%d first %d: best %.3f median %.3f ms" % (N, FIRST, 1e3 * min(ts[2:]), 1e3 * float(np.median(ts[2:]))), flush=True)
    sys.exit(0)
if MODE == "fitness_each":   # the fitness stage of every member on its own (which member makes a share's stage long)
    stage_all()
    _capi.check(lib.lsr_align_batch(hs, N, G.ctypes.data_as(fptr), finals.ctypes.data_as(fptr), res), "a")
    for b in range(N):
        one = (C.c_void_p * 1)(hs[b]); f1 = (C.c_double * 1)()
        ts = []
        for it in range(REPS + 2):
            t0 = time.perf_counter(); _capi.check(lib.lsr_get_fitness_score_batch(one, 1, BIG, f1), "f"); ts.append(time.perf_counter() - t0)
        print("member %d: fitness alone best %.3f median %.3f ms | score %.4f" % (FIRST + b, 1e3 * min(ts[2:]), 1e3 * float(np.median(ts[2:])), f1[0]), flush=True)
    sys.exit(0)
if MODE == "share":
    ts = [whole() for _ in range(REPS + 2)][2:]
    print("share x%d first %d: whole path best %.3f median %.3f ms" % (N, FIRST, 1e3 * min(ts), 1e3 * float(np.median(ts))), flush=True)
    sys.exit(0)

T = {k: [] for k in ("target", "source", "align", "fitness")}
for it in range(REPS + 2):
    t0 = time.perf_counter(); _capi.check(lib.lsr_set_input_target_batch(hs, N, tptr, tcnt, 32, 1), "t")
    t1 = time.perf_counter(); _capi.check(lib.lsr_set_input_source_batch(hs, N, sptr, scnt, 32, 1), "s")
    t2 = time.perf_counter(); _capi.check(lib.lsr_align_batch(hs, N, G.ctypes.data_as(fptr), finals.ctypes.data_as(fptr), res), "a")
    t3 = time.perf_counter(); _capi.check(lib.lsr_get_fitness_score_batch(hs, N, BIG, fit), "f")
    t4 = time.perf_counter()
    if it >= 2:
        T["target"].append(t1 - t0); T["source"].append(t2 - t1); T["align"].append(t3 - t2); T["fitness"].append(t4 - t3)
tf = []
for it in range(REPS + 2):
    stage_all()
    t0 = time.perf_counter()
    _capi.check(lib.lsr_align_fitness_batch(hs, N, G.ctypes.data_as(fptr), finals.ctypes.data_as(fptr), res, BIG, fit), "af")
    tf.append(time.perf_counter() - t0)
tw = [whole() for _ in range(REPS + 2)][2:]
ev = [int(r.n_evaluations) for r in res]; its = [int(r.iterations) for r in res]
print("share x%d first %d | median ms: " % (N, FIRST) + " | ".join("%s %.3f" % (k, 1e3 * np.median(v)) for k, v in T.items()) +
      " | align+fitness (one call) %.3f | whole path best %.3f median %.3f | passes %s iterations %s | fitness bits %s" %
      (1e3 * np.median(tf[2:]), 1e3 * min(tw), 1e3 * float(np.median(tw)), ev, its,
       "%016x" % (int(np.bitwise_xor.reduce(np.frombuffer(np.array(list(fit), np.float64).tobytes(), np.uint64)))), ), flush=True)
