"""GPU tests at BASELINE.json's full sizes (cfg 1/2: 30k-pt scan vs 10-frame submap; cfg 3: GICP on the
same scan; cfg 4: candidate batch; cfg 5: 120k-pt 64-line scan vs 20-frame submap).  Direct oracle
comparison where the oracle finishes in seconds, plus size-independent properties: recovery of a known
rigid offset, invariance to source permutation, idempotence (re-aligning from the answer stays put)."""
import numpy as np
import pytest

from lidarslam_ros2_amd import synth
from lidarslam_ros2_amd.posemath import pose_delta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def pool():
    """Worker processes for the ray casting of the full-size workloads (same clouds as the sequential generator)."""
    import multiprocessing as mp
    import os

    with mp.get_context("spawn").Pool(min(32, len(os.sched_getaffinity(0)))) as p:   # spawn: this process may hold a GPU context
        yield p


@pytest.fixture(scope="module")
def cfg12(pool):
    return synth.cfg_ndt_30k(pool=pool)


def make_ndt(res, eps, max_iter=None):
    from lidarslam_ros2_amd import DIRECT7, NormalDistributionsTransform

    r = NormalDistributionsTransform(device=0)
    r.setResolution(res)
    r.setTransformationEpsilon(eps)
    r.setNeighborhoodSearchMethod(DIRECT7)
    if max_iter is not None:
        r.setMaximumIterations(max_iter)
    return r


@pytest.mark.parametrize("eps,max_iter", [(0.01, None), (0.0, 30)])
def test_cfg1_cfg2_match_oracle(O, cfg12, eps, max_iter):
    """cfg 1 (reference settings, eps 0.01) and cfg 2 (fixed 30 iterations) on the full workload."""
    c = cfg12
    ndt = make_ndt(5.0, eps, max_iter)
    ndt.setInputTarget(synth.as_pointxyzi(c.target))
    ndt.setInputSource(synth.as_pointxyzi(c.source))
    ndt.align(c.guess)
    T = ndt.getFinalTransformation()
    ref = O.ndt_align(O.VoxelGridCovariance(c.target, 5.0), c.source, c.guess, resolution=5.0, trans_eps=eps,
                      max_iterations=max_iter or 35, num_threads=min(32, O.max_threads()))
    dt, ang = pose_delta(T, ref["final"])
    assert dt <= 1e-3 and ang <= 1e-4, (dt, ang)
    assert ndt.getFinalNumIteration() == ref["iterations"]
    assert ndt.hasConverged() and ref["converged"]
    gt_dt, gt_ang = pose_delta(T, c.truth)
    assert gt_dt < 0.05 and gt_ang < 2e-3
    # voxel table of the 661k-point submap: same leaf set as the CPU path
    g = O.VoxelGridCovariance(c.target, 5.0)
    info = ndt.gridInfo()
    assert info["n_leaves"] == g.n_leaves and info["n_valid"] == g.n_valid


def test_cfg2_properties(cfg12):
    c = cfg12
    ndt = make_ndt(5.0, 1e-6, 30)
    ndt.setInputTarget(c.target)
    ndt.setInputSource(c.source)
    ndt.align(c.guess)
    T = ndt.getFinalTransformation()
    # permutation invariance: only the fp64 summation order changes
    rng = np.random.default_rng(0)
    ndt.setInputSource(c.source[rng.permutation(c.source.shape[0])])
    ndt.align(c.guess)
    dt, ang = pose_delta(ndt.getFinalTransformation(), T)
    assert dt < 1e-4 and ang < 1e-5
    # idempotence: starting from the answer, the answer does not move
    ndt.align(T)
    dt, ang = pose_delta(ndt.getFinalTransformation(), T)
    print("cfg 2 idempotence: %.2e m %.2e rad" % (dt, ang))
    assert dt < 1e-3 and ang < 1e-4   # the north_star bar itself
    # rigid-motion recovery: move the source by a known transform, the estimate moves by its inverse
    D = synth.pose_matrix(0.15, -0.1, 0.02, 0.01).astype(np.float32)
    moved = (c.source - D[:3, 3]) @ D[:3, :3]          # D^-1 applied to the scan
    ndt.setInputSource(moved.astype(np.float32))
    ndt.align((c.guess.astype(np.float64) @ D.astype(np.float64)).astype(np.float32))
    dt, ang = pose_delta(ndt.getFinalTransformation(), T.astype(np.float64) @ D.astype(np.float64))
    assert dt < 5e-3 and ang < 5e-4


def test_cfg3_gicp_matches_oracle(O, pool):
    from lidarslam_ros2_amd import GeneralizedIterativeClosestPoint

    c = synth.cfg_gicp_30k(pool=pool)
    g = GeneralizedIterativeClosestPoint(device=0)
    g.setMaxCorrespondenceDistance(5.0)
    g.setTransformationEpsilon(1e-8)
    g.setInputTarget(synth.as_pointxyzi(c.target))
    g.setInputSource(synth.as_pointxyzi(c.source))
    g.align(c.guess)
    T = g.getFinalTransformation()
    th = min(32, O.max_threads())
    nt, ns = O.NearestNeighbour(c.target, 1.0), O.NearestNeighbour(c.source, 1.0)
    ct, cs = O.gicp_covariances(nt, c.target, num_threads=th), O.gicp_covariances(ns, c.source, num_threads=th)
    ref = O.gicp_align(nt, c.target, ct, c.source, cs, c.guess, max_corr_dist=5.0, trans_eps=1e-8, solver=0, num_threads=th)
    dt, ang = pose_delta(T, ref["final"])           # vs the reference schedule (BFGS)
    assert dt <= 1e-3 and ang <= 1e-4, (dt, ang, g.last_result, ref)
    assert g.hasConverged()
    gt_dt, gt_ang = pose_delta(T, c.truth)
    assert gt_dt < 0.03 and gt_ang < 1e-3


def test_cfg4_candidate_batch_sharded_single_rank(O, pool):
    """One GPU's share of cfg 4 (64 candidates / 8 GPUs = 8): per candidate setInputTarget + setInputSource + align +
    getFitnessScore (graph_based_slam_component.cpp:181-231), advanced as ONE batch through the C ABI's sharded entry point
    (lsr_align_batch_sharded, one-rank communicator) and through the torch.distributed layer.  EVERY candidate is held to the
    oracle: pose inside the north_star bar, same number of Newton iterations, same fitness score."""
    from lidarslam_ros2_amd import align_batch
    from lidarslam_ros2_amd.sharding import Comm, align_batch_sharded, pack_record, register_sharded

    n_cand = 8
    cases = [synth.cfg_loop_candidate(c, pool=pool) for c in range(n_cand)]

    def make_regs(indices):
        regs = []
        for i in indices:
            r = make_ndt(5.0, 0.01, 100)               # backend: setMaximumIterations(100)
            r.setInputTarget(cases[i].target)
            r.setInputSource(cases[i].source)
            regs.append(r)
        return regs

    def register_local(indices):
        regs = make_regs(indices)
        finals, results = align_batch(regs, [cases[i].guess for i in indices])
        return [pack_record(finals[k], results[k]["score"], results[k]["iterations"], results[k]["converged"],
                            regs[k].getFitnessScore()) for k in range(len(indices))]

    out = register_sharded(n_cand, register_local)
    comm = Comm(0, 1, 0)
    out_c = align_batch_sharded(comm, make_regs(range(n_cand)), n_cand, [c.guess for c in cases], with_fitness=True)
    comm.close()
    assert len(out) == n_cand and len(out_c) == n_cand
    threads = min(32, O.max_threads())
    for i in range(n_cand):
        r, rc = out[i], out_c[i]
        assert np.array_equal(r["T"], rc["T"]) and r["iterations"] == rc["iterations"]       # both layers drive the same core
        dt, ang = pose_delta(r["T"], cases[i].truth)
        assert r["converged"] and dt < 0.08 and ang < 3e-3, (i, dt, ang)
        assert r["fitness"] < 0.2                      # a closed loop passes the reference's gate (score < 0.3 default region)
        ref = O.ndt_align(O.VoxelGridCovariance(cases[i].target, 5.0), cases[i].source, cases[i].guess, resolution=5.0,
                          trans_eps=0.01, max_iterations=100, num_threads=threads)
        dt, ang = pose_delta(r["T"], ref["final"])
        assert dt <= 1e-3 and ang <= 1e-4, (i, dt, ang)
        assert r["iterations"] == ref["iterations"], i
        fit = O.NearestNeighbour(cases[i].target, 1.0).fitness_score(cases[i].source, ref["final"], num_threads=threads)
        assert abs(r["fitness"] - fit) <= 1e-4 * fit, i


def test_cfg5_dense_scan_matches_oracle(O, pool):
    c = synth.cfg_dense_120k(pool=pool)
    assert c.source.shape == (120000, 3)
    ndt = make_ndt(2.0, 0.01)
    ndt.setInputTarget(c.target)
    ndt.setInputSource(c.source)
    ndt.align(c.guess)
    T = ndt.getFinalTransformation()
    g = O.VoxelGridCovariance(c.target, 2.0)
    info = ndt.gridInfo()
    assert info["n_leaves"] == g.n_leaves and info["n_valid"] == g.n_valid
    ref = O.ndt_align(g, c.source, c.guess, resolution=2.0, trans_eps=0.01, num_threads=min(32, O.max_threads()))
    dt, ang = pose_delta(T, ref["final"])
    assert dt <= 1e-3 and ang <= 1e-4, (dt, ang, ndt.last_result, ref["iterations"])
    assert ndt.getFinalNumIteration() == ref["iterations"]


def test_cfg5_quad_and_lane_kernels_return_the_same_bits(pool):
    """cfg 5 through every kernel: the lane kernel (automatic for >= 65 536 points; 512 and 1024 threads) and the quad kernel
    forced (its workgroups walk several 128-point batches here and collect their integer pieces in LDS first) — the same
    final_T after the same number of passes, bit for bit, at res 2.0 (dense global table) and res 1.0."""
    c = synth.cfg_dense_120k(pool=pool)
    for res in (2.0, 1.0):
        ref = None
        for quad, wg in ((-1, 0), (1, 0), (0, 1024), (0, 512)):
            ndt = make_ndt(res, 0.01)
            ndt.setTuning(quad=quad, workgroup=wg)
            ndt.setInputTarget(c.target)
            ndt.setInputSource(c.source)
            ndt.align(c.guess)
            got = (ndt.getFinalTransformation(), ndt.getFinalNumIteration(), ndt.last_result["n_evaluations"])
            if ref is None:
                ref = got
            assert np.array_equal(got[0], ref[0]) and got[1:] == ref[1:], (res, quad, wg)
            ndt.close()


def test_cfg4_hard_candidates_match_the_cpu_fixture(pool):
    """The four cfg-4 candidates on which NDT with the backend's settings stops in a local optimum (0.2 - 1.0 m from the truth):
    the GPU path must stop in the SAME place after the SAME number of Newton iterations as the CPU oracle
    (tests/golden/cfg4_candidates_oracle.npz, generated by tests/golden/make_cfg4_fixture.py; bench.py checks all 64)."""
    import os

    from lidarslam_ros2_amd import align_batch
    from lidarslam_ros2_amd.registration import fitness_score_batch, set_input_target_batch

    from golden_fixtures import load_golden

    fx, origin = load_golden("cfg4_candidates_oracle")
    print("[golden] cfg-4 hard candidates held to the %s fixture" % origin)
    hard = [11, 30, 43, 48]
    cases = [synth.cfg_loop_candidate(c, pool=pool) for c in hard]
    regs = [make_ndt(5.0, 0.01, 100) for _ in hard]
    set_input_target_batch(regs, [c.target for c in cases])
    for r, c in zip(regs, cases):
        r.setInputSource(c.source)
    finals, results = align_batch(regs, [c.guess for c in cases])
    fits = fitness_score_batch(regs)
    for k, c in enumerate(hard):
        dt, ang = pose_delta(finals[k], fx["final"][c])
        assert dt <= 1e-3 and ang <= 1e-4, (c, dt, ang)
        assert results[k]["iterations"] == int(fx["iterations"][c]), c
        assert abs(fits[k] - fx["fitness"][c]) <= 1e-4 * fx["fitness"][c], c
        assert pose_delta(fx["final"][c], fx["truth"][c])[0] > 0.15      # these really are the local-optimum cases


# ---- cfg 4: ALL 64 candidates against the committed CPU-oracle fixtures -------------------------------------------------
# tests/golden/cfg4_candidates_oracle.npz (eps 0.01, the backend's schedule; make_cfg4_fixture.py) and
# tests/golden/cfg4_candidates_oracle_tight.npz (eps 1e-6; make_cfg4_tight_fixture.py).
# Round 3 had to bound candidates 18 / 34 by a multiple of the CPU path's own spread: the batch path ended 1.4 mm from the
# fixture.  The cause was not conditioning but the compiler fusing the point transform differently in the two kernels; with the
# fp32 operation order pinned (ndt_point.hpp) and the canonical sums (ndt.hip: canon) every candidate is within 5e-5 m of the CPU
# result on BOTH paths — which are now one path: the staged batch and the one-by-one loop return the same bits.
BAR_T, BAR_R = 1e-3, 1e-4          # north_star: <= 1e-3 m translation, <= 1e-4 rad rotation
TIGHT_IT_SLACK = 16         # Newton iterations a tight (eps 1e-6) registration may differ from the CPU fixture by (measured max: 13)
TIGHT_IT_MISMATCHES = 12    # candidates of the 64 whose tight iteration count may differ at all (measured: 7)


def _fit_tol(dt, ang, fit):
    """getFitnessScore is the mean squared nearest-neighbour distance at the registered pose: moving the points by d changes
    it by about 2 sqrt(fit) d, i.e. by 2 d / sqrt(fit) relative; a pose difference (dt, ang) moves a scan point at range R by
    up to dt + ang R (R ~ 30 m for these scans).  1e-4 (the bar for equal poses) + twice that estimate."""
    return 1e-4 + 4.0 * (dt + 30.0 * ang) / float(np.sqrt(fit))


def _dump(name, rows):
    """per-candidate numbers of a parity run -> gpurun_out/ (scratch; read back after a GPU session)"""
    import json
    import os

    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        json.dump(rows, open(os.path.join(d, name), "w"))
    except OSError:
        pass


@pytest.fixture(scope="module")
def cfg4_all(pool):
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    from golden_fixtures import load_golden

    fx, origin = load_golden("cfg4_candidates_oracle")   # eps 0.01: the reference's own dump (oracle/ref_recipe) when there is one
    print("[golden] all 64 cfg-4 candidates held to the %s fixture" % origin)
    fxt = np.load(os.path.join(here, "golden", "cfg4_candidates_oracle_tight.npz"))
    cases = pool.map(synth.cfg_loop_candidate, range(64), chunksize=1)
    return fx, fxt, cases


def _register_all(cases, eps, batched):
    """-> (finals, iterations, fitness) of all candidates; batched: the staged C-ABI entries (lsr_set_input_target_batch,
    one shared launch chain, lsr_get_fitness_score_batch), else the reference's loop, one candidate after the other
    (graph_based_slam_component.cpp:181-231)."""
    from lidarslam_ros2_amd import align_batch
    from lidarslam_ros2_amd.registration import fitness_score_batch, set_input_target_batch

    regs = [make_ndt(5.0, eps, 100) for _ in cases]
    if batched:
        set_input_target_batch(regs, [c.target for c in cases])
        for r, c in zip(regs, cases):
            r.setInputSource(c.source)
        finals, results = align_batch(regs, [c.guess for c in cases])
        fits = fitness_score_batch(regs)
        its = [res["iterations"] for res in results]
    else:
        finals, its, fits = [], [], []
        for r, c in zip(regs, cases):
            r.setInputTarget(c.target)
            r.setInputSource(c.source)
            r.align(c.guess)
            finals.append(r.getFinalTransformation())
            its.append(r.getFinalNumIteration())
            fits.append(r.getFitnessScore())
    for r in regs:
        r.close()
    return finals, its, fits


@pytest.fixture(scope="module")
def cfg4_runs(cfg4_all):
    """(eps, batched) -> (finals, iterations, fitness) of all 64 candidates, computed once"""
    fx, fxt, cases = cfg4_all
    return {(eps, batched): _register_all(cases, eps, batched) for eps in (0.01, 1e-6) for batched in (True, False)}


@pytest.mark.parametrize("eps", [0.01, 1e-6], ids=["backend-schedule", "tight"])
def test_cfg4_staged_batch_and_one_by_one_return_the_same_bits(cfg4_runs, eps):
    """One input, one answer at BASELINE size: the 64 candidates registered through the staged batch entries (lane kernel, shared
    launch chain, launches widened as members finish) and one after the other (quad kernel) end at the SAME final_T after the
    same number of Newton iterations; the fitness scores then differ only by the fp64 order of the mean."""
    fb, ib, sb = cfg4_runs[(eps, True)]
    fs, is_, ss = cfg4_runs[(eps, False)]
    for c in range(64):
        assert np.array_equal(fb[c], fs[c]), (c, pose_delta(fb[c], fs[c]))
        assert ib[c] == is_[c], c
        assert abs(sb[c] - ss[c]) <= 1e-12 * ss[c], c


@pytest.mark.parametrize("batched", [True, False], ids=["staged-batch", "one-by-one"])
def test_cfg4_all_64_candidates_match_the_cpu_fixture_tight(cfg4_all, cfg4_runs, batched):
    """Tight mode (transformation_epsilon 1e-6, max_iterations 100): both sides reach the optimum of their candidate, so EVERY
    one of the 64 must agree inside the north_star bar — the ill-conditioned ones (18, 21, 34) included — with the fitness score
    at that pose equal to 1e-4.  Iteration counts are BOUNDED here, not equal: at a 1e-6 step threshold the loop stops on the
    noise floor of the line search (the CPU emulation of the GPU's arithmetic, tests/ndt_host_emu.py, and the oracle differ
    by up to a dozen iterations there while ending 1e-4 m apart at most).  Measured on MI355X (rounds 4 and 5): 57 of 64 equal,
    the others off by 1, 1, 1, 2, 2, 2 and 13 — so at most TIGHT_IT_SLACK iterations on any candidate and at most
    TIGHT_IT_MISMATCHES candidates that differ at all; a controller that wanders fails both.  At eps 0.01
    the counts are compared for equality (next test)."""
    fx, fxt, cases = cfg4_all
    finals, its, fits = cfg4_runs[(1e-6, batched)]
    bad, rows = {}, []
    for c in range(64):
        dt, ang = pose_delta(finals[c], fxt["final_tight"][c])
        fit_rel = abs(fits[c] - fxt["fitness_tight"][c]) / fxt["fitness_tight"][c]
        rows.append([c, dt, ang, float(fit_rel), int(its[c]), int(fxt["iterations_tight"][c])])
        if (dt > BAR_T or ang > BAR_R or fit_rel > _fit_tol(dt, ang, fxt["fitness_tight"][c]) or its[c] > 102 or
                abs(int(its[c]) - int(fxt["iterations_tight"][c])) > TIGHT_IT_SLACK):
            bad[c] = (dt, ang, fit_rel, its[c], int(fxt["iterations_tight"][c]))
    _dump("cfg4_parity_tight_%s.json" % ("batch" if batched else "single"), rows)
    assert not bad, bad
    differ = [r[0] for r in rows if r[4] != r[5]]
    assert len(differ) <= TIGHT_IT_MISMATCHES, differ


@pytest.mark.parametrize("batched", [True, False], ids=["staged-batch", "one-by-one"])
def test_cfg4_all_64_candidates_match_the_cpu_fixture(cfg4_all, cfg4_runs, batched):
    """The backend's own schedule (eps 0.01, max_iterations 100; graph_based_slam_component.cpp:64-72): the same number of
    Newton iterations on all 64, EVERY pose inside the north_star bar (measured: worst 4.9e-5 m / 4.9e-6 rad — a twentieth of
    it), fitness to 1e-4 at that pose.  No named exceptions, no spread factor."""
    fx, fxt, cases = cfg4_all
    finals, its, fits = cfg4_runs[(0.01, batched)]
    bad, rows = {}, []
    for c in range(64):
        dt, ang = pose_delta(finals[c], fx["final"][c])
        fit_rel = abs(fits[c] - fx["fitness"][c]) / fx["fitness"][c]
        rows.append([c, dt, ang, float(fit_rel), int(its[c]), int(fx["iterations"][c])])
        if its[c] != int(fx["iterations"][c]):
            bad[c] = ("iterations", its[c], int(fx["iterations"][c]))
        elif dt > BAR_T or ang > BAR_R or fit_rel > _fit_tol(dt, ang, fx["fitness"][c]):
            bad[c] = (dt, ang, fit_rel)
    _dump("cfg4_parity_eps001_%s.json" % ("batch" if batched else "single"), rows)
    assert not bad, bad
