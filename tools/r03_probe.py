"""Round-3 A/B probe (one process, workloads cached under LSR_BENCH_CACHE_DIR):
  cfg 2 / cfg 1 single registrations through the quad kernel (LDS table) and the tile path forced onto the same workload;
  cfg 5 (120k-pt scan vs 20-frame submap, res 2.0) and a res-1.0 backend-style case through tile / dense / compact tables;
  a batch of 8 registrations at res 2.0 (tile path, quad kernel over the batch) against the same eight one by one."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import multiprocessing as mp
from lidarslam_ros2_amd import NormalDistributionsTransform, synth, align_batch
from lidarslam_ros2_amd.posemath import pose_delta
from _cache import cached

def _pool():
    return mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0))))
def _c2():
    with _pool() as p: return synth.cfg_ndt_30k(pool=p)
def _c5():
    with _pool() as p: return synth.cfg_dense_120k(pool=p)
case = cached("probe_cfg_ndt_30k", _c2)
dense = cached("probe_cfg_dense_120k", _c5)
dev = lambda a: torch.from_numpy(synth.as_pointxyzi(a)).cuda()
tgt, src = dev(case.target), dev(case.source)
dtgt, dsrc = dev(dense.target), dev(dense.source)
torch.cuda.synchronize()
TAB = {0: "dense", 1: "compact", 2: "lds", 3: "tile", -1: "auto"}

def run(name, tgt_t, src_t, guess, truth, res, eps, mi, tab, reps=8, quad=-1, sort=-1):
    ndt = NormalDistributionsTransform(0); ndt.setResolution(res); ndt.setTransformationEpsilon(eps); ndt.setMaximumIterations(mi)
    ndt.setTuning(table_mode=tab, quad=quad, sort=sort)
    t0 = time.perf_counter(); ndt.setInputTarget(tgt_t); t_tgt = time.perf_counter() - t0
    tt = []
    for _ in range(5):
        t0 = time.perf_counter(); ndt.setInputTarget(tgt_t); tt.append(time.perf_counter() - t0)
    ndt.setInputSource(src_t)
    for _ in range(2): ndt.align(guess)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ndt.setInputSource(src_t); ndt.align(guess); ts.append(time.perf_counter() - t0)
    ndt.setProfiling(True); ndt.getProfile(reset=True); ndt.align(guess); p = ndt.getProfile(reset=True); ndt.setProfiling(False)
    r = ndt.last_result
    T = ndt.getFinalTransformation()
    print(f"{name:10s} tab {TAB[tab]:7s} sort {sort:2d}: {1e3 * np.median(ts):8.3f} ms  {r['iterations']:3d} it {r['n_evaluations']:4d} passes  "
          f"{1e6 * np.median(ts) / max(1, r['n_evaluations']):7.2f} us/pass wall, {1e3 * p['deriv_ms_total'] / max(1, p['deriv_launches']):7.2f} us/pass events | "
          f"setInputTarget {1e3 * np.median(tt):.3f} ms | vs truth %.2e m %.2e rad | grid {ndt.gridInfo()['n_valid']} valid" % pose_delta(T, truth), flush=True)
    return T

print("== cfg 2 (eps 0, 30 it) / cfg 1 (eps 0.01)", flush=True)
for tab in (-1, 3, 0):
    T2 = run("cfg2", tgt, src, case.guess, case.truth, 5.0, 0.0, 30, tab)
    T1 = run("cfg1", tgt, src, case.guess, case.truth, 5.0, 0.01, 35, tab, reps=30)
print("== cfg 5 (res 2.0)", flush=True)
ref = None
for tab, sort in ((-1, -1), (0, 0), (0, 1), (1, 0)):
    T = run("cfg5", dtgt, dsrc, dense.guess, dense.truth, 2.0, 0.01, 35, tab, sort=sort)
    if ref is None: ref = T
    else: print("           vs auto: %.2e m %.2e rad" % pose_delta(T, ref), flush=True)
for tab, sort in ((-1, -1), (0, 0), (0, 1)):
    run("cfg5 30it", dtgt, dsrc, dense.guess, dense.truth, 2.0, 0.0, 30, tab, reps=4, sort=sort)
print("== res 1.0 / 1.5 on the 20-frame submap (the reference's backend / tukuba resolutions), 120k-pt scan", flush=True)
for res in (1.0, 1.5):
    ref = None
    for tab, sort in ((-1, -1), (0, 0), (0, 1)):
        try:
            T = run(f"res{res}", dtgt, dsrc, dense.guess, dense.truth, res, 0.0, 10, tab, reps=4, sort=sort)
            if ref is None: ref = T
            else: print("           vs auto: %.2e m %.2e rad" % pose_delta(T, ref), flush=True)
        except Exception as e:
            print("res", res, "tab", tab, "FAILED", repr(e), flush=True)
print("== batch of 8 at res 2.0 on the 10-frame submap (30k-pt scans from 8 guesses)", flush=True)
rng = np.random.default_rng(5)
guesses = []
for b in range(8):
    G = np.array(case.guess, np.float64); G[:3, 3] += rng.uniform(-0.3, 0.3, 3); guesses.append(G.astype(np.float32))
for tab, sort in ((-1, -1), (0, 0), (0, 1)):
    regs = []
    for b in range(8):
        r = NormalDistributionsTransform(0); r.setResolution(2.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(35); r.setTuning(table_mode=tab, sort=sort)
        if b == 0: r.setInputTarget(tgt)
        else: r.shareTargetOf(regs[0])
        r.setInputSource(src); regs.append(r)
    for _ in range(2): finals, results = align_batch(regs, guesses)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); finals, results = align_batch(regs, guesses); ts.append(time.perf_counter() - t0)
    ones = []
    for b in range(8):
        regs[b].align(guesses[b]); ones.append(regs[b].getFinalTransformation())
    t0 = time.perf_counter()
    for b in range(8): regs[b].align(guesses[b])
    t_one = time.perf_counter() - t0
    d = max(pose_delta(finals[b], ones[b])[0] for b in range(8))
    print(f"batch8 res2 tab {TAB[tab]:7s} sort {sort:2d}: batch {1e3 * np.median(ts):.3f} ms, one by one {1e3 * t_one:.3f} ms, passes {[x['n_evaluations'] for x in results]}, "
          f"batch vs single max {d:.2e} m, vs truth {max(pose_delta(finals[b], case.truth)[0] for b in range(8)):.2e} m", flush=True)
