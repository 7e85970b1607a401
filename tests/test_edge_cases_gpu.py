"""GPU edge cases of the C ABI: empty / ragged / non-finite inputs, record strides, device-pointer inputs,
neighbourhood variants, lazy grid rebuild, error codes (the reference has no tests — SURVEY.md §4 — these
pin the behaviour SURVEY.md §8b/§9 describes)."""
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
def case():
    return synth.small_case(n_source=3000, n_keyframes=3)


def make_ndt(res=5.0, eps=0.01):
    from lidarslam_ros2_amd import NormalDistributionsTransform

    r = NormalDistributionsTransform(device=0)
    r.setResolution(res)
    r.setTransformationEpsilon(eps)
    return r


def test_record_strides_and_device_pointers_are_equivalent(case):
    import torch

    finals = []
    for fmt in ("xyz12", "xyzw16", "xyzi32", "device32"):
        r = make_ndt()
        if fmt == "xyz12":
            t, s = case.target, case.source
        elif fmt == "xyzw16":
            t, s = np.c_[case.target, np.ones(len(case.target), np.float32)], np.c_[case.source, np.ones(len(case.source), np.float32)]
        else:
            t, s = synth.as_pointxyzi(case.target), synth.as_pointxyzi(case.source)
        if fmt == "device32":
            t, s = torch.from_numpy(t).cuda(), torch.from_numpy(s).cuda()
        r.setInputTarget(t)
        r.setInputSource(s)
        r.align(case.guess)
        finals.append(r.getFinalTransformation())
    for f in finals[1:]:
        assert np.array_equal(f, finals[0])      # same SoA planes in HBM -> bit-identical registration


def test_empty_and_missing_inputs(case):
    from lidarslam_ros2_amd import _capi

    r = make_ndt()
    with pytest.raises(_capi.RegistrationError) as ei:
        r.align()
    assert ei.value.status == -4                                   # LSR_ERR_NO_TARGET
    r.setInputTarget(np.zeros((0, 3), np.float32))                 # empty target: accepted, align refuses
    r.setInputSource(case.source)
    with pytest.raises(_capi.RegistrationError) as ei:
        r.align()
    assert ei.value.status == -4
    r.setInputTarget(case.target)
    r.setInputSource(np.zeros((0, 3), np.float32))                 # empty source: nothing to match, pose = guess
    r.align(case.guess)
    assert np.array_equal(r.getFinalTransformation(), case.guess)
    assert r.getFinalNumIteration() == 0
    with pytest.raises(_capi.RegistrationError):
        r.setResolution(0.0)
    with pytest.raises(ValueError):
        r.setInputSource(np.zeros((5, 2), np.float32))             # fewer than 3 columns


def test_nonfinite_points_are_ignored(O, case):
    src = np.concatenate([case.source, np.array([[np.nan, 0, 0], [0, np.inf, 0], [1e30, 1e30, 1e30]], np.float32)])
    tgt = np.concatenate([case.target, np.array([[np.nan, np.nan, np.nan]], np.float32)])
    a, b = make_ndt(), make_ndt()
    a.setInputTarget(case.target); a.setInputSource(case.source); a.align(case.guess)
    b.setInputTarget(tgt); b.setInputSource(src); b.align(case.guess)
    dt, ang = pose_delta(a.getFinalTransformation(), b.getFinalTransformation())
    assert dt < 1e-6 and ang < 1e-7
    # trans_probability is score / N with N = all source points, as in the reference
    assert b.getTransformationProbability() == pytest.approx(a.getTransformationProbability() * len(case.source) / len(src), rel=1e-6)


@pytest.mark.parametrize("method,search", [("DIRECT1", 1), ("DIRECT7", 7), ("DIRECT26", 26), ("KDTREE", 0)])
def test_neighbourhood_variants_match_oracle(O, case, method, search):
    import lidarslam_ros2_amd as L

    r = make_ndt(4.0)
    r.setNeighborhoodSearchMethod(getattr(L, method))
    r.setInputTarget(case.target)
    r.setInputSource(case.source)
    grid = O.VoxelGridCovariance(case.target, 4.0)
    p = O.matrix_to_pose(case.guess) + np.array([0.1, -0.05, 0.02, 0.002, -0.003, 0.004])
    s, g, H = r.derivatives(p)
    rs, rg, rH = O.ndt_derivatives(grid, case.source, p, resolution=4.0, search=search)
    assert abs(s - rs) <= 1e-5 * abs(rs)
    assert np.abs(g - rg).max() <= 2e-5 * np.abs(rg).max() and np.abs(H - rH).max() <= 2e-5 * np.abs(rH).max()
    r.align(case.guess)
    ref = O.ndt_align(grid, case.source, case.guess, resolution=4.0, search=search)
    dt, ang = pose_delta(r.getFinalTransformation(), ref["final"])
    assert dt <= 1e-3 and ang <= 1e-4


@pytest.mark.parametrize("res,force_radix", [(4.0, False), (2.0, False), (4.0, True)])
def test_kdtree_centroids_are_the_float_running_sums(O, case, res, force_radix):
    """The voxel-centroid kd-tree of the KDTREE neighbourhood holds Leaf::centroid — a FLOAT running sum of the leaf's points in cloud
    order over (float) count —, not the fp64 mean: bit for bit the oracle's, whichever builder made the grid."""
    r = make_ndt(res)
    if force_radix:
        r.setTuning(grid_builder=1)
    r.setInputTarget(case.target)
    grid = O.VoxelGridCovariance(case.target, res)
    d, dump = r.gridDump(), grid.dump()
    assert np.array_equal(d["idx"], dump["idx"])
    cen, ref = r.gridCentroids(), grid.centroids()
    in_tree = dump["n"] >= 6
    assert in_tree.sum() > 20
    assert np.array_equal(cen[in_tree], ref[in_tree])
    assert np.isnan(cen[~in_tree]).all()
    assert np.abs(ref[in_tree] - dump["mean"][in_tree]).max() > 0            # ... and they are not the rounded fp64 means


def test_kdtree_neighbourhood_sets_and_batches(O, case):
    """KDTREE after the target was set (centroids built on first use), inside a candidate set (lane kernel) and alone (quad kernel):
    the same registration bit for bit, and the oracle's."""
    import lidarslam_ros2_amd as L
    from lidarslam_ros2_amd.registration import align_batch

    regs = []
    for _ in range(3):
        r = make_ndt(4.0)
        r.setInputTarget(case.target)
        r.setInputSource(case.source)
        r.setNeighborhoodSearchMethod(L.KDTREE)                              # after setInputTarget: built lazily
        regs.append(r)
    regs[0].align(case.guess)
    alone = (regs[0].getFinalTransformation(), regs[0].getFinalNumIteration())
    finals, results = align_batch(regs, [case.guess] * 3)
    for k in range(3):
        assert np.array_equal(np.asarray(finals[k]), alone[0]) and int(results[k]["iterations"]) == alone[1]
    grid = O.VoxelGridCovariance(case.target, 4.0)
    ref = O.ndt_align(grid, case.source, case.guess, resolution=4.0, search=0)
    dt, ang = pose_delta(alone[0], ref["final"])
    assert dt <= 1e-3 and ang <= 1e-4 and alone[1] == ref["iterations"]
    # the DIRECT26 answer is another one: the radius test drops neighbours
    regs[1].setNeighborhoodSearchMethod(L.DIRECT26)
    p = O.matrix_to_pose(case.guess)
    assert regs[1].derivatives(p)[0] != regs[0].derivatives(p)[0]


def test_hessian_d1_sign_option_matches_oracle(O, case):
    grid = O.VoxelGridCovariance(case.target, 5.0)
    p = O.matrix_to_pose(case.guess) + np.array([0.05, 0.02, -0.01, 0.003, 0.02, -0.004])
    for sign in (+1, -1):
        r = make_ndt()
        r.setHessianD1Sign(sign)
        r.setInputTarget(case.target)
        r.setInputSource(case.source)
        _, _, H = r.derivatives(p)
        _, _, rH = O.ndt_derivatives(grid, case.source, p, resolution=5.0, d1_sign=sign)
        assert np.abs(H - rH).max() <= 2e-5 * np.abs(rH).max()


def test_resolution_change_rebuilds_grid_lazily(O, case):
    r = make_ndt(5.0)
    r.setInputTarget(case.target)
    n5 = r.gridInfo()["n_leaves"]
    r.setResolution(2.0)                                            # pclomp re-inits the grid on the next use
    info = r.gridInfo()
    assert info["n_leaves"] == O.VoxelGridCovariance(case.target, 2.0).n_leaves != n5
    r.setInputSource(case.source)
    r.align(case.guess)
    ref = O.ndt_align(O.VoxelGridCovariance(case.target, 2.0), case.source, case.guess, resolution=2.0)
    dt, ang = pose_delta(r.getFinalTransformation(), ref["final"])
    assert dt <= 1e-3 and ang <= 1e-4


def test_max_iterations_zero_and_repeatability(O, case):
    r = make_ndt(5.0, 1e-6)
    r.setMaximumIterations(0)
    r.setInputTarget(case.target)
    r.setInputSource(case.source)
    r.align(case.guess)
    ref = O.ndt_align(O.VoxelGridCovariance(case.target, 5.0), case.source, case.guess, resolution=5.0, trans_eps=1e-6,
                      max_iterations=0)
    assert r.getFinalNumIteration() == ref["iterations"] == 2      # `iter > max_iter` is strict: two passes
    dt, ang = pose_delta(r.getFinalTransformation(), ref["final"])
    assert dt <= 1e-3 and ang <= 1e-4
    # run-to-run reproducibility: fixed-order reductions, no float atomics
    T1 = r.getFinalTransformation()
    r.align(case.guess)
    assert np.array_equal(T1, r.getFinalTransformation())


def test_compact_leaf_table_path_for_huge_extents(O, case):
    """Targets whose dense cell table would exceed 4 Mi cells use the compact table + cell->slot indirection
    (ndt_build_grid): same answers as the oracle, also for a registration 3.9 km from the origin."""
    off = np.float32([3000.0, 2500.0, 0.0])
    tgt = np.concatenate([case.target, case.target + off])               # a second copy of the scene ~3.9 km away
    res = 1.5
    r = make_ndt(res)
    r.setInputTarget(tgt)
    info = r.gridInfo()
    cells = np.prod((info["max_b"] - info["min_b"] + 1).astype(np.int64))
    assert cells > 4 * 2**20                                             # really on the compact path
    ref_grid = O.VoxelGridCovariance(tgt, res)
    assert info["n_leaves"] == ref_grid.n_leaves and info["n_valid"] == ref_grid.n_valid
    r.setInputSource(case.source)
    for shift in (np.zeros(3, np.float32), off):
        G = case.guess.copy()
        G[:3, 3] += shift
        p = O.matrix_to_pose(G)
        s, g, H = r.derivatives(p)
        rs, rg, rH = O.ndt_derivatives(ref_grid, case.source, p, resolution=res)
        tol = 1e-5 if not shift.any() else 1e-3      # 3.9 km away the fp32 leaf means carry ~1e-4 m of rounding
        assert abs(s - rs) <= tol * abs(rs) and np.abs(g - rg).max() <= 20 * tol * np.abs(rg).max()
        r.align(G)
        ref = O.ndt_align(ref_grid, case.source, G, resolution=res)
        dt, ang = pose_delta(r.getFinalTransformation(), ref["final"])
        assert dt <= 1e-3 and ang <= 1e-4, (shift, dt, ang)
        assert r.hasConverged() == ref["converged"]
    # the KDTREE neighbourhood on the compact table (centroids indexed by leaf slot, not by cell)
    import lidarslam_ros2_amd as L
    r.setNeighborhoodSearchMethod(L.KDTREE)
    dump = ref_grid.dump()
    in_tree = dump["n"] >= 6
    assert np.array_equal(r.gridCentroids()[in_tree], ref_grid.centroids()[in_tree])
    p = O.matrix_to_pose(case.guess)
    s, g, H = r.derivatives(p)
    rs, rg, rH = O.ndt_derivatives(ref_grid, case.source, p, resolution=res, search=0)
    assert abs(s - rs) <= 1e-5 * abs(rs) and np.abs(g - rg).max() <= 2e-4 * np.abs(rg).max()


def test_points_exactly_on_voxel_faces(O, case):
    """SURVEY.md §8c KAT 7: source points whose transformed coordinates sit exactly on leaf faces take the
    reference's cell assignment (fp32 floor(x'/leaf) at lookup, floor(x*inv_leaf) at build)."""
    res = 2.5
    grid = O.VoxelGridCovariance(case.target, res)
    lo, hi = grid.min_b * res, (grid.max_b + 1) * res
    rng = np.random.default_rng(2)
    n = 4000
    pts = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    # snap one (random) coordinate of every point onto a face k*res — exactly representable in fp32
    ax = rng.integers(0, 3, n)
    pts[np.arange(n), ax] = (np.round(pts[np.arange(n), ax] / res) * res).astype(np.float32)
    r = make_ndt(res)
    r.setInputTarget(case.target)
    r.setInputSource(pts)
    p = np.zeros(6)
    s, g, H = r.derivatives(p, T=np.eye(4, dtype=np.float32))
    rs, rg, rH = O.ndt_derivatives(grid, pts, p, T=np.eye(4, dtype=np.float32), resolution=res)
    assert rs > 0
    assert abs(s - rs) <= 1e-5 * abs(rs)
    assert np.abs(g - rg).max() <= 2e-5 * np.abs(rg).max() and np.abs(H - rH).max() <= 2e-5 * np.abs(rH).max()


def test_d1_quirk_does_not_move_the_fixed_point(case):
    """the h_ang d1 sign only perturbs the Hessian (step direction), never the gradient, so in
    tight-epsilon mode both settings converge to the same pose."""
    finals = []
    for sign in (+1, -1):
        r = make_ndt(5.0, 1e-6)
        r.setMaximumIterations(30)
        r.setHessianD1Sign(sign)
        r.setInputTarget(case.target)
        r.setInputSource(case.source)
        r.align(case.guess)
        finals.append(r.getFinalTransformation())
    dt, ang = pose_delta(finals[0], finals[1])
    assert dt <= 1e-3 and ang <= 1e-4


def test_objects_are_created_and_destroyed_by_the_hundred(case):
    """The backend creates and drops registration objects as loop candidates come and go: 100 objects in turn — each sets a target,
    registers, scores (lazy neighbour grid), every fifth also leads a candidate set (side stream, chain streams, their events) — and is
    destroyed again (lsr_destroy releases streams, events, pinned mailboxes).  The last object must answer like the first."""
    import gc

    from lidarslam_ros2_amd import NormalDistributionsTransform
    from lidarslam_ros2_amd.registration import align_fitness_batch

    first = None
    for k in range(100):
        r = NormalDistributionsTransform(device=0)
        r.setResolution(5.0)
        r.setTransformationEpsilon(0.01)
        r.setInputTarget(case.target)
        r.setInputSource(case.source)
        r.align(case.guess)
        got = (np.array(r.getFinalTransformation()), r.getFinalNumIteration(), r.getFitnessScore())
        if k % 5 == 0:   # a set of two led by this object
            mate = NormalDistributionsTransform(device=0)
            mate.setResolution(5.0)
            mate.setTransformationEpsilon(0.01)
            mate.setInputTarget(case.target)
            mate.setInputSource(case.source)
            finals, _, fit = align_fitness_batch([r, mate], [case.guess, case.guess])
            assert np.array_equal(finals[0], finals[1]) and fit[0] == fit[1]
            del mate
        if first is None:
            first = got
        assert np.array_equal(got[0], first[0]) and got[1] == first[1] and got[2] == first[2], k
        del r
        gc.collect()
