"""GPU parity tests for the NDT path: HIP kernels (through the C ABI) vs the CPU oracle on the same
seeded inputs.  Tolerances follow BASELINE.json north_star: final pose <= 1e-3 m / <= 1e-4 rad."""
import numpy as np
import pytest

from lidarslam_ros2_amd import synth
from lidarslam_ros2_amd.posemath import pose_delta

pytestmark = pytest.mark.gpu

POSE_T_TOL = 1e-3   # metres   (north_star)
POSE_R_TOL = 1e-4   # radians  (north_star)


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def case():
    return synth.small_case(n_source=4500, n_keyframes=4)


def make_ndt(res, eps=0.01, max_iter=None):
    from lidarslam_ros2_amd import DIRECT7, NormalDistributionsTransform

    ndt = NormalDistributionsTransform(device=0)
    ndt.setResolution(res)
    ndt.setTransformationEpsilon(eps)
    ndt.setNeighborhoodSearchMethod(DIRECT7)
    if max_iter is not None:
        ndt.setMaximumIterations(max_iter)
    return ndt


@pytest.mark.parametrize("res", [5.0, 2.0, 1.0])
def test_voxel_grid_matches_oracle(O, case, res):
    ndt = make_ndt(res)
    ndt.setInputTarget(synth.as_pointxyzi(case.target))
    info = ndt.gridInfo()
    ref = O.VoxelGridCovariance(case.target, res)
    assert np.array_equal(info["min_b"], ref.min_b) and np.array_equal(info["max_b"], ref.max_b)
    assert info["n_leaves"] == ref.n_leaves
    d, r = ndt.gridDump(), ref.dump()
    # voxel SET and membership counts are integer work: bit-exact
    assert np.array_equal(d["idx"], r["idx"])
    assert np.array_equal(d["n"], r["n"])
    assert info["n_valid"] == ref.n_valid
    # fp64 sums in a different (tree) order: means to 1e-9 m, inverse covariances to 1e-6 relative
    assert np.abs(d["mean"] - r["mean"]).max() < 1e-9
    valid = r["n"] >= 6
    num = np.abs(d["icov"][valid] - r["icov"][valid]).max(axis=(1, 2))
    den = np.abs(r["icov"][valid]).max(axis=(1, 2))
    assert (num / den).max() < 1e-6


@pytest.mark.parametrize("res", [5.0, 2.0])
@pytest.mark.parametrize("hess", [True, False])
def test_derivative_pass_matches_oracle(O, case, res, hess):
    ndt = make_ndt(res)
    ndt.setInputTarget(synth.as_pointxyzi(case.target))
    ndt.setInputSource(synth.as_pointxyzi(case.source))
    ref = O.VoxelGridCovariance(case.target, res)
    rng = np.random.default_rng(3)
    for trial in range(3):
        p = O.matrix_to_pose(case.guess) + np.r_[rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.02, 0.02, 3)]
        s, g, H = ndt.derivatives(p, compute_hessian=hess)
        rs, rg, rH = O.ndt_derivatives(ref, case.source, p, compute_hessian=hess, resolution=res)
        assert abs(s - rs) <= 1e-5 * abs(rs)
        assert np.abs(g - rg).max() <= 2e-5 * np.abs(rg).max()
        if hess:
            assert np.abs(H - rH).max() <= 2e-5 * np.abs(rH).max()


@pytest.mark.parametrize("res,eps,max_iter", [(5.0, 0.01, None), (3.0, 0.01, None), (5.0, 1e-6, 30), (3.0, 1e-6, 30)])
def test_align_matches_oracle(O, case, res, eps, max_iter):
    """Same-schedule (eps 0.01) and tight (eps 1e-6, max 30 iterations) modes."""
    ndt = make_ndt(res, eps, max_iter)
    ndt.setInputTarget(synth.as_pointxyzi(case.target))
    ndt.setInputSource(synth.as_pointxyzi(case.source))
    out = ndt.align(case.guess, output=True)
    T = ndt.getFinalTransformation()
    ref = O.ndt_align(O.VoxelGridCovariance(case.target, res), case.source, case.guess, resolution=res, trans_eps=eps,
                      max_iterations=max_iter or 35)
    dt, ang = pose_delta(T, ref["final"])
    assert ndt.hasConverged() == ref["converged"]
    assert dt <= POSE_T_TOL and ang <= POSE_R_TOL, (dt, ang, ndt.last_result, ref["iterations"])
    # output cloud = final transformation applied to the source
    exp = case.source @ T[:3, :3].T + T[:3, 3]
    assert np.abs(out - exp).max() < 1e-4
    # and it actually registered: within a few cm of ground truth on this small scene
    gt_dt, _ = pose_delta(T, case.truth)
    assert gt_dt < 0.1


def test_identity_guess_and_no_overlap(O, case):
    ndt = make_ndt(5.0)
    ndt.setInputTarget(synth.as_pointxyzi(case.target))
    # source far away from every voxel: zero score, zero step -> converged at the guess
    far = case.source + np.float32(5000.0)
    ndt.setInputSource(far)
    ndt.align()
    assert np.allclose(ndt.getFinalTransformation(), np.eye(4))
    assert ndt.hasConverged()
    assert ndt.getFinalNumIteration() == 0


@pytest.mark.parametrize("eps,max_iter", [(0.01, 35), (1e-6, 60)])
def test_batch_equals_single(O, case, eps, max_iter):
    """One input, one answer: a registration inside lsr_align_batch (lane kernel, one lane per point, launches widened as
    members finish; seven members = two independent launch chains of 3 + 4 on two streams, the subset of three = one chain)
    returns the SAME final_T and iteration counts as the same registration through lsr_align (quad kernel, four lanes per
    point) — bit for bit, at the reference's schedule and at the tight one.  The sum of a pass is defined on the input
    (csrc/ndt.hip: canon), not on the launch."""
    from lidarslam_ros2_amd import align_batch

    res = 5.0
    lead = make_ndt(res, eps=eps, max_iter=max_iter)
    lead.setInputTarget(synth.as_pointxyzi(case.target))
    regs, guesses, singles = [], [], []
    rng = np.random.default_rng(11)
    B = 7
    for b in range(B):
        r = lead if b == 0 else make_ndt(res, eps=eps, max_iter=max_iter)
        if b:
            r.shareTargetOf(lead)
        n = 4500 - 251 * b
        r.setInputSource(case.source[:n])
        g = case.guess.copy()
        g[:3, 3] += rng.uniform(-0.2, 0.2, 3).astype(np.float32)
        regs.append(r)
        guesses.append(g)
    for r, g in zip(regs, guesses):
        r.align(g)
        singles.append((r.getFinalTransformation(), r.getFinalNumIteration(), r.last_result["n_evaluations"]))
    finals, results = align_batch(regs, guesses)
    for b in range(B):
        assert np.array_equal(finals[b], singles[b][0]), (b, pose_delta(finals[b], singles[b][0]))
        assert results[b]["iterations"] == singles[b][1], b
        assert results[b]["n_evaluations"] == singles[b][2], b
    # any subset, any order: the same bits again
    finals2, results2 = align_batch(regs[::-1][:3], guesses[::-1][:3])
    for k, b in enumerate((B - 1, B - 2, B - 3)):
        assert np.array_equal(finals2[k], singles[b][0]) and results2[k]["iterations"] == singles[b][1]


def test_back_to_back_aligns_of_different_lengths_do_not_interfere(case):
    """The launch chain of an align() is fed by polling a host mailbox and may leave a few queued launches that
    exit at their head.  A mixed sequence of long, short and zero-iteration aligns — single and
    batched, growing and shrinking the batch — on ONE handle must give exactly what fresh handles give."""
    from lidarslam_ros2_amd import align_batch

    res = 5.0
    tgt = synth.as_pointxyzi(case.target)
    src = synth.as_pointxyzi(case.source)
    schedule = [(0.01, 35), (0.0, 5), (0.01, 0), (0.0, 12), (1e-6, 35), (0.01, 1), (0.0, 0), (0.01, 35)]

    def run(ndt, eps, mi, guess):
        ndt.setTransformationEpsilon(eps)
        ndt.setMaximumIterations(mi)
        ndt.setInputSource(src)
        ndt.align(guess)
        r = ndt.last_result
        return ndt.getFinalTransformation().copy(), r["iterations"], r["n_evaluations"], bool(ndt.hasConverged())

    fresh = []
    for eps, mi in schedule:
        n = make_ndt(res)
        n.setInputTarget(tgt)
        fresh.append(run(n, eps, mi, case.guess))
    one = make_ndt(res)
    one.setInputTarget(tgt)
    peers = [make_ndt(res) for _ in range(4)]
    for p in peers:
        p.shareTargetOf(one)
        p.setInputSource(src)
    for rep in range(3):
        for k, (eps, mi) in enumerate(schedule):
            got = run(one, eps, mi, case.guess)
            assert np.array_equal(got[0], fresh[k][0]) and got[1:] == fresh[k][1:], (rep, k)
            if k % 3 == rep % 3:   # interleave batches of changing size on the same lead handle / stream
                B = 2 + (k + rep) % 4
                regs = [one] + peers[:B - 1]
                for r in regs:
                    r.setTransformationEpsilon(0.01)
                    r.setMaximumIterations(35)
                finals, results = align_batch(regs, [case.guess] * B)
                for b in range(1, B):
                    assert np.array_equal(finals[b], finals[1]) and results[b]["iterations"] == results[1]["iterations"]


def test_gpu_matches_the_committed_golden_fixture():
    """tests/golden/ndt_small_golden.npz (written by tests/golden/make_golden.py from the oracle, committed): the HIP
    path against the fixture itself, independent of the oracle library being rebuilt identically on this box."""
    import os

    from golden_fixtures import load_golden

    gold, origin = load_golden("ndt_small_golden")   # the reference's own dump (oracle/ref_recipe) when there is one
    print("[golden] the HIP path is held to the %s fixture" % origin)
    case = synth.small_case(n_source=int(gold["n_source"]), n_keyframes=int(gold["n_keyframes"]))
    assert case.target.shape[0] == int(gold["n_target"]) and np.array_equal(case.source, gold["source"])
    res = float(gold["res"])
    ndt = make_ndt(res)
    ndt.setInputTarget(synth.as_pointxyzi(case.target))
    ndt.setInputSource(synth.as_pointxyzi(case.source))
    info, d = ndt.gridInfo(), ndt.gridDump()
    # voxel set and membership counts: bit-exact
    assert np.array_equal(info["min_b"], gold["min_b"]) and np.array_equal(info["max_b"], gold["max_b"])
    assert np.array_equal(d["idx"], gold["leaf_idx"]) and np.array_equal(d["n"], gold["leaf_n"])
    # one derivative pass at the fixture's pose
    s, g, H = ndt.derivatives(gold["p"], compute_hessian=True)
    assert abs(s - float(gold["score"])) <= 1e-5 * abs(float(gold["score"]))
    assert np.abs(g - gold["grad"]).max() <= 2e-5 * np.abs(gold["grad"]).max()
    assert np.abs(H - gold["hess"]).max() <= 2e-5 * np.abs(gold["hess"]).max()
    # align(): the reference's schedule and the tight one
    for eps, mi, key in ((0.01, 35, "eps001"), (1e-6, 30, "tight")):
        ndt.setTransformationEpsilon(eps)
        ndt.setMaximumIterations(mi)
        ndt.align(gold["guess"])
        dt, ang = pose_delta(ndt.getFinalTransformation(), gold["final_" + key])
        assert dt <= POSE_T_TOL and ang <= POSE_R_TOL, (key, dt, ang)
        # the backend's schedule stops on a 0.01 m step: same count; at 1e-6 the loop ends on the noise floor of the line search,
        # where the CPU's fp64 summation order and the device's canonical one may part by an iteration
        if key == "eps001":
            assert ndt.getFinalNumIteration() == int(gold["iters_" + key])
        else:
            assert abs(ndt.getFinalNumIteration() - int(gold["iters_" + key])) <= 2
    # the KDTREE neighbourhood: the kd-tree's centroids bit for bit, a derivative pass, the reference's schedule
    import lidarslam_ros2_amd as L
    ndt.setNeighborhoodSearchMethod(L.KDTREE)
    cen = ndt.gridCentroids()
    in_tree = ~np.isnan(gold["leaf_centroid"][:, 0])
    assert np.array_equal(np.isnan(cen[:, 0]), ~in_tree) and np.array_equal(cen[in_tree], gold["leaf_centroid"][in_tree])
    s, g, H = ndt.derivatives(gold["p"], compute_hessian=True)
    assert abs(s - float(gold["score_kdtree"])) <= 1e-5 * abs(float(gold["score_kdtree"]))
    assert np.abs(g - gold["grad_kdtree"]).max() <= 2e-5 * np.abs(gold["grad_kdtree"]).max()
    assert np.abs(H - gold["hess_kdtree"]).max() <= 2e-5 * np.abs(gold["hess_kdtree"]).max()
    ndt.setTransformationEpsilon(0.01)
    ndt.setMaximumIterations(35)
    ndt.align(gold["guess"])
    dt, ang = pose_delta(ndt.getFinalTransformation(), gold["final_kdtree"])
    assert dt <= POSE_T_TOL and ang <= POSE_R_TOL and ndt.getFinalNumIteration() == int(gold["iters_kdtree"])


# ---- launch variants of the derivative pass and the two grid builders -----------------------------------------------
# (quad: 1 = four lanes per point (workgroup = points per workgroup: 0 auto / 64 / 128), 0 = lane kernel, one lane per point
#  (workgroup = threads: 512 / 1024); table mode: 0 dense global, 1 compact global, 2 LDS)
#  a fourth entry: split = 1 — two waves per chunk in the 512-thread lane kernel (round 6)
VARIANTS = [(1, 0, 2), (1, 0, 0), (1, 0, 1), (1, 64, 2), (0, 1024, 0), (0, 1024, 1), (0, 1024, 2), (0, 512, 0), (0, 512, 2),
            (0, 512, 0, 1), (0, 512, 1, 1), (0, 512, 2, 1)]


def _tune(ndt, v):
    quad, workgroup, table_mode = v[:3]
    ndt.setTuning(workgroup=workgroup, table_mode=table_mode, quad=quad, split=(v[3] if len(v) > 3 else 0))


@pytest.mark.parametrize("variant", VARIANTS)
def test_every_launch_variant_matches_the_oracle(O, case, variant):
    """The same derivative pass and the same align through every kernel instantiation (kernel x workgroup size x where the
    leaf records are read from): oracle tolerances for one pass, north_star bar and equal iteration counts for align."""
    res = 5.0
    ndt = make_ndt(res)
    _tune(ndt, variant)
    ndt.setInputTarget(synth.as_pointxyzi(case.target))
    ndt.setInputSource(synth.as_pointxyzi(case.source))
    ref = O.VoxelGridCovariance(case.target, res)
    rng = np.random.default_rng(11)
    for hess in (True, False):
        p = O.matrix_to_pose(case.guess) + np.r_[rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.02, 0.02, 3)]
        s, g, H = ndt.derivatives(p, compute_hessian=hess)
        rs, rg, rH = O.ndt_derivatives(ref, case.source, p, compute_hessian=hess, resolution=res)
        assert abs(s - rs) <= 1e-5 * abs(rs)
        assert np.abs(g - rg).max() <= 2e-5 * np.abs(rg).max()
        if hess:
            assert np.abs(H - rH).max() <= 2e-5 * np.abs(rH).max()
    for eps, max_iter in ((0.01, 35), (1e-6, 30)):
        ndt.setTransformationEpsilon(eps)
        ndt.setMaximumIterations(max_iter)
        ndt.align(case.guess)
        r = O.ndt_align(ref, case.source, case.guess, resolution=res, trans_eps=eps, max_iterations=max_iter)
        dt, ang = pose_delta(ndt.getFinalTransformation(), r["final"])
        assert dt <= POSE_T_TOL and ang <= POSE_R_TOL
        assert ndt.getFinalNumIteration() == r["iterations"]


@pytest.mark.parametrize("neighborhood", ["DIRECT7", "DIRECT1", "DIRECT26", "KDTREE"])
def test_launch_variants_agree_with_each_other(case, neighborhood):
    """One input, one answer: every kernel (four lanes per point / one), every workgroup size, every table form returns the
    same score, gradient and Hessian — and the same registration — bit for bit: a point's terms are formed in one fp32
    operation order (ndt_point.hpp), chunks of 64 points are summed by one fixed fp64 tree, chunk totals are added as exact
    integers (csrc/ndt.hip: canon)."""
    import lidarslam_ros2_amd as L

    res = 5.0
    out, aligned = {}, {}
    for v in VARIANTS:
        ndt = make_ndt(res)
        ndt.setNeighborhoodSearchMethod(getattr(L, neighborhood))
        _tune(ndt, v)
        ndt.setInputTarget(synth.as_pointxyzi(case.target))
        ndt.setInputSource(synth.as_pointxyzi(case.source))
        res_v = []
        for hess, p in ((True, np.r_[0.3, -0.2, 0.05, 0.01, -0.015, 0.02]), (False, np.r_[-0.1, 0.25, 0.0, -0.02, 0.01, 0.03])):
            res_v.append(ndt.derivatives(p, compute_hessian=hess))
            again = ndt.derivatives(p, compute_hessian=hess)
            assert res_v[-1][0] == again[0] and np.array_equal(res_v[-1][1], again[1]) and np.array_equal(res_v[-1][2], again[2])
        out[v] = res_v
        ndt.align(case.guess)
        aligned[v] = (ndt.getFinalTransformation(), ndt.getFinalNumIteration(), ndt.last_result["n_evaluations"])
    ref = out[VARIANTS[0]]
    for v, res_v in out.items():
        for (s, g, H), (s0, g0, H0) in zip(res_v, ref):
            assert s == s0 and np.array_equal(g, g0) and np.array_equal(H, H0), v
        assert np.array_equal(aligned[v][0], aligned[VARIANTS[0]][0]) and aligned[v][1:] == aligned[VARIANTS[0]][1:], v


@pytest.mark.parametrize("res", [5.0, 3.0])
def test_counting_sort_builder_equals_radix_sort_builder(case, res):
    """K1/K2 by the counting-sort builder (dense key spaces) and by the radix-sort builder: identical voxel set, counts and
    bbox; means / inverse covariances differ only by the fp64 summation order."""
    tgt = synth.as_pointxyzi(case.target)
    tgt[7::97, 0] = np.nan     # a few non-finite points: dropped by both builders
    a, b = make_ndt(res), make_ndt(res)
    b.setTuning(grid_builder=1)
    a.setInputTarget(tgt)
    b.setInputTarget(tgt)
    ia, ib = a.gridInfo(), b.gridInfo()
    assert np.array_equal(ia["min_b"], ib["min_b"]) and np.array_equal(ia["max_b"], ib["max_b"])
    assert ia["n_leaves"] == ib["n_leaves"] and ia["n_valid"] == ib["n_valid"]
    da, db = a.gridDump(), b.gridDump()
    assert np.array_equal(da["idx"], db["idx"]) and np.array_equal(da["n"], db["n"])
    assert np.abs(da["mean"] - db["mean"]).max() < 1e-11
    valid = da["n"] >= 6
    num = np.abs(da["icov"][valid] - db["icov"][valid]).max(axis=(1, 2))
    den = np.abs(db["icov"][valid]).max(axis=(1, 2))
    assert (num / den).max() < 1e-8
    # rebuilding gives bit-identical results (deterministic summation order)
    a.setInputTarget(tgt)
    da2 = a.gridDump()
    assert np.array_equal(da["mean"], da2["mean"]) and np.array_equal(da["icov"], da2["icov"])
    # the neighbour grid refined from the counting sort's voxel order answers like the stand-alone one (radix-sorted target)
    a.setInputSource(case.source); b.setInputSource(case.source)
    ia_, da_ = a.nearestNeighbors(case.guess)
    ib_, db_ = b.nearestNeighbors(case.guess)
    assert np.array_equal(ia_, ib_) and np.array_equal(da_, db_)


@pytest.mark.parametrize("wait_mode", [1, 2])
def test_wait_modes_give_the_same_result(O, case, wait_mode):
    res = 5.0
    a, b = make_ndt(res), make_ndt(res)
    b.setTuning(wait_mode=wait_mode)
    for r in (a, b):
        r.setInputTarget(synth.as_pointxyzi(case.target))
        r.setInputSource(synth.as_pointxyzi(case.source))
        r.align(case.guess)
    assert np.array_equal(a.getFinalTransformation(), b.getFinalTransformation())
    assert a.getFinalNumIteration() == b.getFinalNumIteration()


def test_target_batch_and_fitness_batch_equal_the_single_calls(case):
    """lsr_set_input_target_batch / lsr_get_fitness_score_batch stage the same work as the per-object calls so that the
    builds overlap on the device: grids, registrations and scores must be IDENTICAL to the one-at-a-time results."""
    import torch

    from lidarslam_ros2_amd import align_batch
    from lidarslam_ros2_amd.registration import fitness_score_batch, set_input_target_batch

    rng = np.random.default_rng(5)
    B = 5
    targets, sources, guesses = [], [], []
    for b in range(B):   # distinct targets: a different subset of the submap each, shifted
        keep = rng.random(case.target.shape[0]) < (0.6 + 0.08 * b)
        shift = np.array([0.3 * b, -0.2 * b, 0.0], np.float32)
        targets.append(synth.as_pointxyzi(case.target[keep] + shift))
        sources.append(synth.as_pointxyzi(case.source + shift))
        guesses.append(case.guess.copy())
        guesses[-1][:3, 3] += shift
    singles = []
    for b in range(B):
        r = make_ndt(2.0, 0.01, 40)
        r.setInputTarget(targets[b]); r.setInputSource(sources[b]); r.align(guesses[b])
        singles.append((r.gridDump(), r.getFinalTransformation().copy(), r.getFinalNumIteration(), r.getFitnessScore()))
    for on_device in (False, True):
        regs = [make_ndt(2.0, 0.01, 40) for _ in range(B)]
        clouds = [torch.from_numpy(t).cuda() for t in targets] if on_device else targets
        torch.cuda.synchronize()
        set_input_target_batch(regs, clouds)
        for b, r in enumerate(regs):
            d = r.gridDump()
            assert np.array_equal(d["idx"], singles[b][0]["idx"]) and np.array_equal(d["n"], singles[b][0]["n"])
            assert np.array_equal(d["mean"], singles[b][0]["mean"]) and np.array_equal(d["icov"], singles[b][0]["icov"])
            r.setInputSource(torch.from_numpy(sources[b]).cuda() if on_device else sources[b])
        finals, results = align_batch(regs, guesses)
        fits = fitness_score_batch(regs)
        for b in range(B):
            assert np.array_equal(finals[b], singles[b][1]), (b, pose_delta(finals[b], singles[b][1]))   # one input, one answer
            assert results[b]["iterations"] == singles[b][2]
            assert abs(fits[b] - regs[b].getFitnessScore()) == 0.0           # same kernels, same order
            # same pose, same neighbours: what is left between the group search and the single one is the fp64 order of the mean
            assert abs(fits[b] - singles[b][3]) <= 1e-12 * singles[b][3]
    # a second batch on the same objects recycles the target buffers
    set_input_target_batch(regs, clouds)
    assert regs[0].gridInfo()["n_leaves"] == len(singles[0][0]["idx"])
    with pytest.raises(Exception):
        set_input_target_batch([regs[0], regs[0]], clouds[:2])


def test_ragged_candidate_sets(case):
    """A candidate set whose members differ: sources from 60 to 4500 points, an EMPTY source (nothing to match: pose = guess, as the
    single call answers), a source with non-finite points, targets of different size — every member must come out as its own
    single registration would; a member WITHOUT target points makes the set's align fail with NO_TARGET and leaves the objects
    usable (the next set on the same objects works)."""
    from lidarslam_ros2_amd import _capi, align_batch
    from lidarslam_ros2_amd.registration import fitness_score_batch, set_input_source_batch, set_input_target_batch

    rng = np.random.default_rng(5)
    sources = [case.source, case.source[:60], np.zeros((0, 3), np.float32), case.source[::3],
               np.concatenate([case.source[:2000], np.array([[np.nan, 0, 0], [0, np.inf, 0]], np.float32)]), case.source[:1000]]
    targets = [case.target, case.target[::2], case.target, case.target[: len(case.target) // 3], case.target, case.target[::5]]
    guesses = [case.guess.copy() for _ in sources]
    for g in guesses:
        g[:3, 3] += rng.normal(0, 0.05, 3).astype(np.float32)
    regs = [make_ndt(3.0) for _ in sources]
    set_input_target_batch(regs, targets)
    set_input_source_batch(regs, sources)
    finals, res = align_batch(regs, guesses)
    fits = fitness_score_batch([r for k, r in enumerate(regs) if len(sources[k])])
    k_fit = 0
    for k in range(len(sources)):
        one = make_ndt(3.0)
        one.setInputTarget(targets[k]); one.setInputSource(sources[k]); one.align(guesses[k])
        if len(sources[k]) == 0:
            assert np.array_equal(finals[k], guesses[k]) and res[k]["iterations"] == 0
            continue
        assert np.array_equal(finals[k], one.getFinalTransformation()), (k, pose_delta(finals[k], one.getFinalTransformation()))
        assert res[k]["iterations"] == one.getFinalNumIteration(), k
        f1 = one.getFitnessScore()
        assert abs(fits[k_fit] - f1) <= 1e-12 * f1, (k, fits[k_fit], f1)
        k_fit += 1
    # a member without target points: the set's align refuses like the single call does, nothing wedges
    set_input_target_batch(regs, [np.zeros((0, 3), np.float32)] + targets[1:])
    with pytest.raises(_capi.RegistrationError) as ei:
        align_batch(regs, guesses)
    assert ei.value.status == -4                                   # LSR_ERR_NO_TARGET
    set_input_target_batch(regs, targets)
    finals2, _ = align_batch(regs, guesses)
    for k in range(len(sources)):
        assert np.array_equal(finals2[k], finals[k]), k            # same objects, same inputs: the same answer again


def test_align_fitness_batch_equals_the_two_calls(case):
    """lsr_align_fitness_batch (align + getFitnessScore of a candidate set in one call, the searches of early finishers under the
    launch chain of the others) returns exactly what lsr_align_batch followed by lsr_get_fitness_score_batch return — same poses
    bit for bit, same scores — for sets of 1, 5 and 19 members with sources of different size and guesses of different quality
    (members finish between 1 and ~15 Newton iterations apart)."""
    from lidarslam_ros2_amd import align_batch
    from lidarslam_ros2_amd.registration import (align_fitness_batch, fitness_score_batch, set_input_source_batch,
                                                 set_input_target_batch)

    rng = np.random.default_rng(17)
    for B in (1, 5, 19):
        sources = [case.source[:: 1 + (k % 4)] for k in range(B)]
        targets = [case.target if k % 3 else case.target[::2] for k in range(B)]
        guesses = []
        for k in range(B):
            g = case.guess.copy()
            g[:3, 3] += rng.normal(0, 0.02 + 0.1 * (k % 5), 3).astype(np.float32)
            guesses.append(g)
        a, b = [make_ndt(3.0) for _ in range(B)], [make_ndt(3.0) for _ in range(B)]
        for regs in (a, b):
            set_input_target_batch(regs, targets)
            set_input_source_batch(regs, sources)
        f1, r1 = align_batch(a, guesses)
        s1 = fitness_score_batch(a)
        f2, r2, s2 = align_fitness_batch(b, guesses)
        for k in range(B):
            assert np.array_equal(f1[k], f2[k]), (B, k)
            assert r1[k]["iterations"] == r2[k]["iterations"] and r1[k]["converged"] == r2[k]["converged"]
            assert s2[k] == pytest.approx(s1[k], rel=1e-12), (B, k)
            assert b[k].getFitnessScore() == pytest.approx(s1[k], rel=1e-12)      # the objects are left as after the two calls


def test_hand_written_sort_builders_equal_the_rocprim_builders(tmp_path):
    """Round 6: the radix target builder (key spaces beyond the counting sort: cfg 5, the reference's 1-2 m resolutions) and the NN grid
    builder run on the hand-written LSD sort + run table + scans of csrc/lsd_sort.hip; rocPRIM's sort / run_length_encode / scan stay
    behind LSR_TARGET_SORT=rocprim / LSR_NN_SORT=rocprim.  Child processes, one per setting: the voxel grids (leaf set, counts, fp64
    means and inverse covariances), a registration, the NN answers and the GICP covariances are bit-identical."""
    import os
    import subprocess
    import sys

    code = ("import numpy as np, sys\n"
            "sys.path.insert(0, %r)\n"
            "from lidarslam_ros2_amd import NormalDistributionsTransform, GeneralizedIterativeClosestPoint, DIRECT7, synth\n"
            "case = synth.small_case(n_source=4000, n_keyframes=4, seed=3)\n"
            "tgt = synth.as_pointxyzi(case.target); tgt[5::89, 1] = np.nan\n"
            "out = {}\n"
            "for tag, res, builder in (('r1', 1.0, 0), ('r07', 0.7, 0), ('r5_forced', 5.0, 1)):\n"
            "    r = NormalDistributionsTransform(device=0); r.setResolution(res); r.setTransformationEpsilon(0.01); r.setNeighborhoodSearchMethod(DIRECT7)\n"
            "    r.setTuning(grid_builder=builder); r.setInputTarget(tgt); r.setInputSource(case.source); r.align(case.guess)\n"
            "    d = r.gridDump(); info = r.gridInfo()\n"
            "    for k in ('idx', 'n', 'mean', 'icov'): out[tag + '_' + k] = d[k]\n"
            "    out[tag + '_T'] = r.getFinalTransformation(); out[tag + '_leaves'] = np.array([info['n_leaves'], info['n_valid']])\n"
            "for tag, builder in (('nn_bucket', 0), ('nn_sort', 1)):\n"
            "    g = GeneralizedIterativeClosestPoint(device=0); g.setTuning(grid_builder=builder); g.setInputTarget(case.target); g.setInputSource(case.source)\n"
            "    idx, d2 = g.nearestNeighbors(case.guess); g.align(case.guess)\n"
            "    out[tag + '_idx'] = idx; out[tag + '_d2'] = d2; out[tag + '_T'] = g.getFinalTransformation(); out[tag + '_cov'] = g.covariances('target')\n"
            "    out[tag + '_fit'] = np.array([g.getFitnessScore()])\n"
            "np.savez(sys.argv[1], **out)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, env in (("hand", {}), ("rocprim", {"LSR_TARGET_SORT": "rocprim", "LSR_NN_SORT": "rocprim"})):
        path = str(tmp_path / (name + ".npz"))
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, **env), timeout=600)
        outs.append(np.load(path))
    a, b = outs
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 28
    for k in a.files:
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True), k
    assert int(a["r1_leaves"][0]) > 16383 or int(a["r07_leaves"][0]) > 3000   # the general builder really ran on a large key space
