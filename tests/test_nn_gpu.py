"""GPU parity: kd-tree-free NN grid + getFitnessScore vs the CPU oracle (exact NN => bit-exact indices
and fp32 squared distances)."""
import numpy as np
import pytest

from lidarslam_ros2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def case():
    return synth.small_case(n_source=4500, n_keyframes=4)


def make_ndt(res=5.0):
    from lidarslam_ros2_amd import NormalDistributionsTransform

    r = NormalDistributionsTransform(device=0)
    r.setResolution(res)
    r.setTransformationEpsilon(0.01)
    return r


def test_nearest_neighbours_exact_vs_oracle(O, case):
    r = make_ndt()
    r.setInputTarget(case.target)
    # include far outliers (forces the coarse-shell phase) and points outside the grid on every side
    rng = np.random.default_rng(4)
    extra = rng.uniform(-300, 300, (200, 3)).astype(np.float32)
    src = np.concatenate([case.source, extra, case.source[:50] + np.float32([0, 0, 40])])
    r.setInputSource(src)
    nn = O.NearestNeighbour(case.target, cell=1.0)
    for T in (None, case.guess, case.truth.astype(np.float32)):
        idx, d2 = r.nearestNeighbors(T)
        ridx, rd2 = nn.search(src, T)
        assert np.array_equal(d2, rd2)          # same fp32 arithmetic, exact search: bit-exact distances
        same = idx == ridx
        # identical points in the target (duplicates across keyframes) may tie; both pick the lowest index
        assert same.all()


def test_fitness_score_matches_oracle(O, case):
    r = make_ndt()
    r.setInputTarget(synth.as_pointxyzi(case.target))
    r.setInputSource(synth.as_pointxyzi(case.source))
    r.align(case.guess)
    T = r.getFinalTransformation()
    nn = O.NearestNeighbour(case.target, cell=1.0)
    for max_range in (float("inf"), 1.0, 0.04):
        fs = r.getFitnessScore() if np.isinf(max_range) else r.getFitnessScore(max_range)
        ref = nn.fitness_score(case.source, T, max_range)
        assert fs == pytest.approx(ref, rel=1e-12)
    # loop-closure gate semantics (graph_based_slam_component.cpp:231-233): a good alignment scores low
    assert r.getFitnessScore() < 0.5


def test_fitness_score_no_overlap_and_errors(O, case):
    from lidarslam_ros2_amd import _capi

    r = make_ndt()
    with pytest.raises(_capi.RegistrationError) as ei:
        r.getFitnessScore()
    assert ei.value.status == -4  # LSR_ERR_NO_TARGET
    r.setInputTarget(case.target)
    with pytest.raises(_capi.RegistrationError) as ei:
        r.getFitnessScore()
    assert ei.value.status == -5  # LSR_ERR_NO_SOURCE
    r.setInputSource(case.source + np.float32(1000.0))
    r.align()
    # nothing within 1 m^2: PCL returns std::numeric_limits<double>::max()
    assert r.getFitnessScore(1.0) == 1.7976931348623157e308


def test_bucket_built_grid_equals_sort_built_grid(case):
    """The NN grid of a dense key space is built by bucketing (integer histogram + scan + scatter through the counters, the
    order inside a fine cell left to the atomics), larger ones by a radix sort: both must answer every query identically —
    candidates are ranked by (distance, original index), never by their position in a cell."""
    from lidarslam_ros2_amd import GeneralizedIterativeClosestPoint

    out = []
    for builder in (0, 1):
        g = GeneralizedIterativeClosestPoint(device=0)
        g.setTuning(grid_builder=builder)
        g.setInputTarget(case.target)
        g.setInputSource(case.source)
        T = case.guess
        idx, d2 = g.nearestNeighbors(T)
        g.align(case.guess)
        out.append((idx, d2, g.getFitnessScore(), g.getFinalTransformation(), g.covariances("source")))
        # repeated builds answer identically too (the atomics may order a cell differently every time)
        g.setInputTarget(case.target)
        idx2, d22 = g.nearestNeighbors(T)
        assert np.array_equal(idx, idx2) and np.array_equal(d2, d22)
    a, b = out
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2] == b[2]
    assert np.array_equal(a[3], b[3])
    assert np.array_equal(a[4], b[4])


@pytest.mark.parametrize("res", [5.0, 2.0, 1.0, 3.7])
def test_neighbour_grid_refined_from_the_voxel_grid_is_exact(O, case, res):
    """An NDT object's neighbour grid is a refinement of its voxel grid (fine cell = ndt_resolution / 8, coarse cells = the NDT
    voxels, built by ordering every voxel's points by fine cell): the answers must equal the oracle's — and the stand-alone
    builder's (a GICP object on the same clouds) — bit for bit at every resolution, cell edges that are not powers of two
    included, for far outliers and for queries outside the grid."""
    from lidarslam_ros2_amd import GeneralizedIterativeClosestPoint

    r = make_ndt(res)
    r.setInputTarget(case.target)
    rng = np.random.default_rng(7)
    extra = rng.uniform(-300, 300, (200, 3)).astype(np.float32)
    src = np.concatenate([case.source, extra, case.source[:50] + np.float32([0, 0, 40]), case.target[:300]])   # exact hits too
    r.setInputSource(src)
    g = GeneralizedIterativeClosestPoint(device=0)
    g.setInputTarget(case.target)
    g.setInputSource(src)
    nn = O.NearestNeighbour(case.target, cell=1.0)
    for T in (None, case.guess):
        idx, d2 = r.nearestNeighbors(T)
        ridx, rd2 = nn.search(src, T)
        gidx, gd2 = g.nearestNeighbors(T)
        assert np.array_equal(d2, rd2) and np.array_equal(idx, ridx)
        assert np.array_equal(d2, gd2) and np.array_equal(idx, gidx)


def test_group_entries_equal_the_single_ones(case):
    """lsr_set_input_target_batch / lsr_set_input_source_batch / lsr_get_fitness_score_batch serve a candidate set with GROUP
    launches (up to 16 members per launch): every member must end up exactly where the one-by-one calls put it — same voxel
    grid, same registration, same fitness score, same neighbours."""
    from lidarslam_ros2_amd import align_batch
    from lidarslam_ros2_amd.registration import fitness_score_batch, set_input_source_batch, set_input_target_batch

    rng = np.random.default_rng(3)
    B = 19     # more than one group
    targets, sources, guesses = [], [], []
    for b in range(B):
        keep = rng.random(case.target.shape[0]) < 0.9 - 0.02 * (b % 5)
        targets.append(np.ascontiguousarray(case.target[keep] + np.float32([0.37 * b, -0.21 * b, 0.0])))
        sources.append(np.ascontiguousarray(case.source[: 4400 - 97 * b] + np.float32([0.37 * b, -0.21 * b, 0.0])))
        G = case.guess.copy(); G[:3, 3] += np.float32([0.37 * b, -0.21 * b, 0.0]) - G[:3, :3] @ np.float32([0.37 * b, -0.21 * b, 0.0])
        guesses.append(G)
    singles = []
    for b in range(B):
        r = make_ndt(4.0)
        r.setInputTarget(targets[b]); r.setInputSource(sources[b]); r.align(guesses[b])
        singles.append((r.gridInfo(), r.getFinalTransformation(), r.getFinalNumIteration(), r.getFitnessScore(), r.nearestNeighbors(guesses[b])))
    regs = [make_ndt(4.0) for _ in range(B)]
    set_input_target_batch(regs, targets)
    set_input_source_batch(regs, sources)
    finals, results = align_batch(regs, guesses)
    fits = fitness_score_batch(regs)
    for b in range(B):
        info, T, it, fit, (idx, d2) = singles[b]
        gi = regs[b].gridInfo()
        assert gi["n_valid"] == info["n_valid"] and gi["n_leaves"] == info["n_leaves"]
        assert np.array_equal(gi["min_b"], info["min_b"]) and np.array_equal(gi["max_b"], info["max_b"])
        bidx, bd2 = regs[b].nearestNeighbors(guesses[b])
        assert np.array_equal(bidx, idx) and np.array_equal(bd2, d2)
        # the batch registers with the one-lane kernel, the single call with the quad kernel: same optimum, fp32 association differs
        from lidarslam_ros2_amd.posemath import pose_delta
        dt, ang = pose_delta(finals[b], T)
        assert dt < 1e-3 and ang < 1e-4, (b, dt, ang)
        if dt == 0.0:
            assert fits[b] == fit
        # the fitness of the BATCH pose, evaluated one by one on the same object, must equal the group result exactly
        assert regs[b].getFitnessScore() == fits[b], b
    # device-resident clouds through the same entries
    import torch
    regs2 = [make_ndt(4.0) for _ in range(B)]
    tt = [torch.from_numpy(synth.as_pointxyzi(t)).cuda() for t in targets]
    ss = [torch.from_numpy(synth.as_pointxyzi(s)).cuda() for s in sources]
    set_input_target_batch(regs2, tt)
    set_input_source_batch(regs2, ss)
    finals2, _ = align_batch(regs2, guesses)
    assert np.array_equal(finals2, finals)
    assert fitness_score_batch(regs2) == fits
