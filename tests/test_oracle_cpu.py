"""CPU tests that pin the oracle (the reference ships no tests or golden vectors — SURVEY.md §4 — so
these known-answer / property tests are what anchors it; parity stays "unpinned" in the strict
sense)."""
import os

import numpy as np
import pytest

from lidarslam_ros2_amd import synth
from lidarslam_ros2_amd.posemath import pose_delta
from oracle import oracle as O

from ndt_numpy import NumpyNdt, rot_xyz

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ndt_small_golden.npz")


def test_gauss_constants_kat():
    # SURVEY.md §9.1 table (Magnusson eq. 6.8, outlier ratio 0.55)
    table = {5.0: (-6.9312054349, 0.149546508997, 5.42615073806), 2.0: (-4.19651818695, 0.248478510124, 2.67727854244),
             1.5: (-3.35388340013, 0.30722693662, 1.81423232508), 1.0: (-2.21722524404, 0.433123004704, 0.597837000756)}
    for res, exp in table.items():
        got = O.gauss_constants(res)
        assert np.allclose(got, exp, rtol=1e-10, atol=1e-11)


def test_single_voxel_analytic():
    """One leaf, hand-computed: mean, single-pass covariance with the (n-1)/n factor (sic), inverse."""
    rng = np.random.default_rng(1)
    pts = (rng.normal(0, 0.4, (50, 3)) + np.array([2.5, 2.5, 2.5])).astype(np.float32)
    g = O.VoxelGridCovariance(pts, 5.0)
    d = g.dump()
    assert g.n_leaves == 1 and g.n_valid == 1 and d["n"][0] == 50
    p64 = pts.astype(np.float64)
    mean = p64.mean(0)
    cov = (p64.T @ p64 - 2 * np.outer(p64.sum(0), mean)) / 50 + np.outer(mean, mean)
    cov *= (50 - 1.0) / 50
    assert np.allclose(d["mean"][0], mean, atol=1e-12)
    assert np.allclose(d["cov"][0], cov, atol=1e-12)
    assert np.allclose(d["icov"][0], np.linalg.inv(cov), rtol=1e-9)


def test_eigenvalue_clamp_on_thin_voxel():
    """A nearly planar leaf: the two small eigenvalues are raised to 0.01 * largest (eq. 6.11)."""
    rng = np.random.default_rng(2)
    pts = np.c_[rng.uniform(0.5, 4.5, (200, 2)), 2.0 + rng.normal(0, 1e-3, 200)].astype(np.float32)
    d = O.VoxelGridCovariance(pts, 5.0).dump()
    w = np.linalg.eigvalsh(d["cov"][0])
    assert w[0] == pytest.approx(0.01 * w[2], rel=1e-9)
    assert np.allclose(d["icov"][0] @ d["cov"][0], np.eye(3), atol=1e-9)


def test_min_points_rule_and_membership_vs_bruteforce():
    case = synth.small_case(n_source=1000, n_keyframes=2)
    leaf = 2.0
    g = O.VoxelGridCovariance(case.target, leaf)
    d = g.dump()
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(case.target * inv).astype(np.int64)
    mn, mx = ijk.min(0), ijk.max(0)
    assert np.array_equal(mn, g.min_b) and np.array_equal(mx, g.max_b)
    div = mx - mn + 1
    key = (ijk - mn) @ np.array([1, div[0], div[0] * div[1]])
    uk, cnt = np.unique(key, return_counts=True)
    assert np.array_equal(uk, d["idx"])
    # leaves with < 6 points stay in the map but are unusable; invalidated leaves carry -1
    assert np.array_equal(np.where(d["n"] >= 0, d["n"], cnt), cnt)
    assert g.n_valid == int((d["n"] >= 6).sum())
    assert (cnt < 6).any() and (cnt >= 6).any()


def test_voxel_build_permutation_invariance():
    case = synth.small_case(n_source=1000, n_keyframes=2)
    rng = np.random.default_rng(5)
    a = O.VoxelGridCovariance(case.target, 3.0).dump()
    b = O.VoxelGridCovariance(case.target[rng.permutation(case.target.shape[0])], 3.0).dump()
    assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["n"], b["n"])
    assert np.abs(a["mean"] - b["mean"]).max() < 1e-10


def test_nonfinite_target_points_are_skipped():
    case = synth.small_case(n_source=1000, n_keyframes=2)
    t = case.target.copy()
    t2 = np.concatenate([t, np.array([[np.nan, 0, 0], [np.inf, 1, 1]], np.float32)])
    a, b = O.VoxelGridCovariance(t, 3.0).dump(), O.VoxelGridCovariance(t2, 3.0).dump()
    assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["n"], b["n"])


@pytest.fixture(scope="module")
def small():
    case = synth.small_case(n_source=600, n_keyframes=2)
    res = 4.0
    grid = O.VoxelGridCovariance(case.target, res)
    d1, d2, _ = O.gauss_constants(res)
    ref = NumpyNdt(grid.dump(), grid.min_b, grid.max_b, res, d1, d2)
    return case, res, grid, ref


def test_gradient_matches_fp64_finite_differences(small):
    case, res, grid, ref = small
    p = O.matrix_to_pose(case.guess) + np.array([0.11, -0.07, 0.03, 0.004, -0.006, 0.01])
    s, g, H = O.ndt_derivatives(grid, case.source, p, resolution=res)
    s64, g64 = ref.score_grad(case.source, p)
    # oracle (fp32 pair maths) vs fp64 analytic
    assert abs(s - s64) <= 2e-5 * abs(s64)
    assert np.abs(g - g64).max() <= 1e-4 * np.abs(g64).max()
    # fp64 analytic gradient vs central differences of the fp64 score
    fd = np.zeros(6)
    for i in range(6):
        h = 1e-6
        pp, pm = p.copy(), p.copy()
        pp[i] += h
        pm[i] -= h
        fd[i] = (ref.score_grad(case.source, pp)[0] - ref.score_grad(case.source, pm)[0]) / (2 * h)
    assert np.abs(fd - g64).max() <= 1e-4 * np.abs(g64).max()


def test_hessian_matches_finite_differences_and_d1_quirk(small):
    case, res, grid, ref = small
    p = O.matrix_to_pose(case.guess) + np.array([0.05, 0.02, -0.01, 0.003, 0.02, -0.004])
    _, _, H_an = O.ndt_derivatives(grid, case.source, p, resolution=res, d1_sign=-1)
    _, _, H_up = O.ndt_derivatives(grid, case.source, p, resolution=res, d1_sign=+1)
    Hfd = np.zeros((6, 6))
    for i in range(6):
        h = 1e-6
        pp, pm = p.copy(), p.copy()
        pp[i] += h
        pm[i] -= h
        Hfd[i] = (ref.score_grad(case.source, pp)[1] - ref.score_grad(case.source, pm)[1]) / (2 * h)
    scale = np.abs(Hfd).max()
    assert np.abs(H_an - Hfd).max() <= 2e-4 * scale          # analytic sign: a true Hessian
    assert np.abs(H_an - H_an.T).max() <= 1e-6 * scale         # fp32 pair maths: (i,j) and (j,i) round separately
    diff = np.abs(H_up - H_an)
    assert diff[4, 4] > 0                                      # upstream "+sy" quirk ...
    diff[4, 4] = 0
    assert diff.max() <= 1e-12 * scale                         # ... touches H[ry,ry] only
    # the fp64 computeHessian path agrees with the fp32 computeDerivatives path
    _, _, H64 = O.ndt_derivatives(grid, case.source, p, resolution=res, fp64_hessian=True)
    assert np.abs(H64 - H_up).max() <= 1e-4 * scale


@pytest.mark.parametrize("search", [1, 26])
def test_direct1_and_direct26_match_fp64_numpy(small, search):
    """The other two pclomp neighbourhoods (DIRECT1: own cell; DIRECT26: the whole 3x3x3 block) against the fp64
    numpy restatement: same voxel sets, same score and gradient."""
    case, res, grid, _ = small
    d1, d2, _ = O.gauss_constants(res)
    ref = NumpyNdt(grid.dump(), grid.min_b, grid.max_b, res, d1, d2, search=search)
    p = O.matrix_to_pose(case.guess) + np.array([-0.06, 0.09, 0.02, -0.003, 0.005, 0.008])
    s, g, _ = O.ndt_derivatives(grid, case.source, p, resolution=res, search=search)
    s64, g64 = ref.score_grad(case.source, p)
    assert abs(s - s64) <= 2e-5 * abs(s64)
    assert np.abs(g - g64).max() <= 1e-4 * np.abs(g64).max()


def test_kdtree_neighbourhood_is_a_radius_search_over_the_leaf_centroids(small):
    """pclomp's KDTREE neighbourhood = radiusSearch(x', resolution) on the kd-tree of the leaves' float centroids.  The oracle restates
    it as the 27 cells around the point's cell + the radius test; the numpy model searches ALL centroids by brute force: the same
    voxel sets (score and gradient agree as for the DIRECT methods), strictly between DIRECT7's and DIRECT26's."""
    case, res, grid, _ = small
    d1, d2, _ = O.gauss_constants(res)
    cen = grid.centroids()
    dump = grid.dump()
    ok = dump["n"] >= 6
    # Leaf::centroid is a float running sum over float(n): close to the fp64 mean, not equal to its rounding
    assert np.abs(cen[ok] - dump["mean"][ok]).max() < 1e-3 and (cen[ok] != dump["mean"][ok].astype(np.float32)).any()
    ref = NumpyNdt(dump, grid.min_b, grid.max_b, res, d1, d2, search=0, centroids=cen)
    for dp in (np.array([-0.06, 0.09, 0.02, -0.003, 0.005, 0.008]), np.array([0.3, -0.2, 0.05, 0.01, -0.015, 0.02])):
        p = O.matrix_to_pose(case.guess) + dp
        s, g, _ = O.ndt_derivatives(grid, case.source, p, resolution=res, search=0)
        s64, g64 = ref.score_grad(case.source, p)
        assert abs(s - s64) <= 2e-5 * abs(s64)
        assert np.abs(g - g64).max() <= 1e-4 * np.abs(g64).max()
        s7 = O.ndt_derivatives(grid, case.source, p, resolution=res, search=7)[0]
        s27 = O.ndt_derivatives(grid, case.source, p, resolution=res, search=26)[0]
        assert s7 != s and s < s27
    # a registration with it converges to the same place as with DIRECT7 (the reference's choice) on this easy case
    a = O.ndt_align(grid, case.source, case.guess, resolution=res, search=0)
    b = O.ndt_align(grid, case.source, case.guess, resolution=res, search=7)
    dt, ang = pose_delta(a["final"], b["final"])
    assert a["converged"] and dt < 0.05 and ang < 0.01


def test_direct7_boundary_cases(small):
    case, res, grid, ref = small
    far = (case.source + np.float32(1e4)).astype(np.float32)     # outside the bbox: zero neighbours
    s, g, H = O.ndt_derivatives(grid, far, np.zeros(6), resolution=res)
    assert s == 0 and not g.any() and not H.any()
    # DIRECT1 <= DIRECT7 <= DIRECT26 in score (more voxels contribute)
    p = O.matrix_to_pose(case.guess)
    s1 = O.ndt_derivatives(grid, case.source, p, resolution=res, search=1)[0]
    s7 = O.ndt_derivatives(grid, case.source, p, resolution=res, search=7)[0]
    s27 = O.ndt_derivatives(grid, case.source, p, resolution=res, search=26)[0]
    assert 0 < s1 <= s7 <= s27


def test_euler_xyz_quirk_roundtrip():
    """Eigen's eulerAngles(0,1,2) returns roll in [0, pi]: a small negative roll comes back as the
    (pi + r, pi - p, pi + y)-style alias; the matrix rebuilt from it is the same rotation."""
    for roll in (0.03, -0.03):
        T = synth.pose_matrix(1.0, -2.0, 0.5, 0.2, roll=roll, pitch=-0.05).astype(np.float32)
        p = O.matrix_to_pose(T)
        assert 0 <= p[3] <= np.pi + 1e-6
        if roll < 0:
            assert p[3] > 3.0
        dt, ang = pose_delta(O.pose_to_matrix(p), T)
        assert dt < 1e-6 and ang < 2e-6
        assert np.allclose(rot_xyz(*p[3:]), T[:3, :3], atol=2e-6)


def test_identity_and_recovery():
    case = synth.small_case(n_source=2000, n_keyframes=3)
    res = 5.0
    grid = O.VoxelGridCovariance(case.target, res)
    # source taken from the target itself, identity guess: stays at identity
    sub = case.target[::7][:3000]
    r = O.ndt_align(grid, sub, None, resolution=res, trans_eps=0.01)
    dt, ang = pose_delta(r["final"], np.eye(4))
    assert r["converged"] and dt < 0.03 and ang < 2e-3
    # frontend-style case: guess = previous scan pose (0.5 m off), converges to within NDT accuracy
    for eps, mi in ((0.01, 35), (1e-6, 30)):
        r = O.ndt_align(grid, case.source, case.guess, resolution=res, trans_eps=eps, max_iterations=mi)
        dt, ang = pose_delta(r["final"], case.truth)
        assert r["converged"] and dt < 0.06 and ang < 5e-3
    # thread-count invariance (index-ordered final sum)
    a = O.ndt_align(grid, case.source, case.guess, resolution=res, num_threads=1)
    b = O.ndt_align(grid, case.source, case.guess, resolution=res, num_threads=3)
    assert np.array_equal(a["final"], b["final"]) and a["iterations"] == b["iterations"]


def test_nn_oracle_exact_vs_bruteforce():
    rng = np.random.default_rng(9)
    tgt = rng.uniform(-10, 10, (3000, 3)).astype(np.float32)
    q = rng.uniform(-14, 14, (400, 3)).astype(np.float32)
    nn = O.NearestNeighbour(tgt, cell=1.5)
    idx, d2 = nn.search(q)
    D = ((q[:, None, :] - tgt[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(idx, D.argmin(1))
    assert np.allclose(d2, D.min(1), rtol=1e-6)
    kidx, kd2 = nn.knn(q[:50], 20)
    assert np.array_equal(np.sort(kidx, 1), np.sort(np.argsort(D[:50], 1)[:, :20], 1))
    assert np.all(np.diff(kd2, axis=1) >= 0)
    T = synth.pose_matrix(0.3, -0.2, 0.1, 0.05).astype(np.float32)
    fs = nn.fitness_score(q, T)
    qt = q @ T[:3, :3].T + T[:3, 3]
    Dt = ((qt[:, None, :] - tgt[None, :, :]) ** 2).sum(-1).min(1)
    assert fs == pytest.approx(Dt.mean(), rel=1e-5)
    assert nn.fitness_score(q, T, max_range=1.0) == pytest.approx(Dt[Dt <= 1.0].mean(), rel=1e-5)


def test_oracle_reproduces_golden():
    """tests/golden/ndt_small_golden.npz was written by tests/golden/make_golden.py from this oracle;
    it pins the oracle (and the GPU tests' expectations) against silent drift."""
    from golden_fixtures import load_golden

    gold, origin = load_golden("ndt_small_golden")   # the reference's own dump when oracle/ref_recipe has produced one
    print("[golden] test_oracle_reproduces_golden checks the oracle against the %s fixture" % origin)
    case = synth.small_case(n_source=int(gold["n_source"]), n_keyframes=int(gold["n_keyframes"]))
    assert np.array_equal(case.source, gold["source"]) and np.array_equal(case.guess, gold["guess"])
    assert case.target.shape[0] == int(gold["n_target"])
    grid = O.VoxelGridCovariance(case.target, float(gold["res"]))
    s, g, H = O.ndt_derivatives(grid, case.source, gold["p"], resolution=float(gold["res"]), num_threads=1)
    assert s == pytest.approx(float(gold["score"]), rel=1e-12)
    assert np.allclose(g, gold["grad"], rtol=1e-10, atol=1e-9) and np.allclose(H, gold["hess"], rtol=1e-10, atol=1e-7)
    r = O.ndt_align(grid, case.source, case.guess, resolution=float(gold["res"]), trans_eps=0.01, num_threads=1)
    assert np.allclose(r["final"], gold["final_eps001"], atol=1e-6) and r["iterations"] == int(gold["iters_eps001"])
    # KDTREE neighbourhood (search = 0): the centroids of the kd-tree bit for bit, one derivative pass, one registration
    cen = grid.centroids()
    in_tree = ~np.isnan(gold["leaf_centroid"][:, 0])
    assert np.array_equal(in_tree, gold["leaf_n"] >= 6) and np.array_equal(cen[in_tree], gold["leaf_centroid"][in_tree])
    s, g, H = O.ndt_derivatives(grid, case.source, gold["p"], resolution=float(gold["res"]), search=0, num_threads=1)
    assert s == pytest.approx(float(gold["score_kdtree"]), rel=1e-12)
    assert np.allclose(g, gold["grad_kdtree"], rtol=1e-10, atol=1e-9) and np.allclose(H, gold["hess_kdtree"], rtol=1e-10, atol=1e-7)
    r = O.ndt_align(grid, case.source, case.guess, resolution=float(gold["res"]), trans_eps=0.01, search=0, num_threads=1)
    assert np.allclose(r["final"], gold["final_kdtree"], atol=1e-6) and r["iterations"] == int(gold["iters_kdtree"])


def test_fixture_loader_prefers_a_reference_dump(tmp_path, monkeypatch):
    """tests/golden_fixtures.py: a `ref_<name>.npz` (what oracle/ref_recipe writes from the REFERENCE's own pclomp code) wins over the
    oracle's `<name>.npz`, arrays the dump does not hold fall back to the oracle fixture, and the import step of the recipe writes the
    schema the tests read.  Exercised with the oracle's own numbers routed through the recipe's JSON -> npz step."""
    import json
    import subprocess
    import sys

    from golden_fixtures import load_golden

    gold = np.load(GOLDEN)
    monkeypatch.setenv("LSR_GOLDEN_DIR", str(tmp_path))
    g0, origin0 = load_golden("ndt_small_golden")
    assert origin0 == "oracle" and float(g0["score"]) == float(gold["score"])
    col = lambda M: np.asarray(M, np.float64).T.reshape(-1).tolist()   # the dumper prints Eigen's column-major storage
    res = {"ndt_small": {"score": float(gold["score"]) + 1.0, "grad": gold["grad"].tolist(), "hess": gold["hess"].reshape(-1).tolist(),
                         "final_eps001": col(gold["final_eps001"]), "iters_eps001": int(gold["iters_eps001"]),
                         "final_tight": col(gold["final_tight"]), "iters_tight": int(gold["iters_tight"]),
                         "leaf_idx": gold["leaf_idx"].tolist(), "leaf_n": gold["leaf_n"].tolist(),
                         "min_b": gold["min_b"].tolist(), "max_b": gold["max_b"].tolist()}}
    rj = tmp_path / "results.json"
    rj.write_text(json.dumps(res))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "oracle", "ref_recipe", "import_results.py"), str(rj), str(tmp_path)])
    g1, origin1 = load_golden("ndt_small_golden")
    assert origin1 == "reference"
    assert float(g1["score"]) == float(gold["score"]) + 1.0                       # the dump's value, not the oracle's
    assert np.array_equal(g1["final_eps001"], gold["final_eps001"]) and int(g1["iters_tight"]) == int(gold["iters_tight"])
    assert np.array_equal(g1["source"], gold["source"]) and float(g1["res"]) == float(gold["res"])   # inputs: from the oracle fixture


# ---- loop-closure gate ---------------------------------------------------------
def test_pose_msg_to_matrix_is_the_quaternion_rotation():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(5)
    for _ in range(5):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        t = rng.normal(size=3)
        M = O.pose_msg_to_matrix(t, q)
        assert np.allclose(M[:3, :3], Rotation.from_quat(q).as_matrix(), atol=1e-14)
        assert np.array_equal(M[:3, 3], t) and np.array_equal(M[3], [0, 0, 0, 1])


def test_search_loop_oracle_closes_a_synthetic_loop():
    route = synth.make_loop_route(sensor=synth.Sensor(16, -20.0, 12.0, 450), vg_map=0.4)
    nt = min(16, O.max_threads())
    kw = dict(distance_loop_closure=20.0, range_of_searching_loop_closure=10.0, search_submap_num=2, voxel_leaf_size=0.4,
              ndt_resolution=5.0, num_threads=nt)
    out = O.search_loop(route, **kw)
    assert len(out) == 1 and out[0]["accepted"] and out[0]["pair_id"][1] == len(route) - 1
    # candidates respect both gates (graph_based_slam_component.cpp:195-196) and the nearest one is taken (:199)
    lp = np.asarray(route[-1]["position"])
    d = [np.linalg.norm(lp - np.asarray(s["position"])) for s in route]
    ok = [i for i, s in enumerate(route) if route[-1]["distance"] - s["distance"] > 20.0 and d[i] < 10.0]
    assert out[0]["pair_id"][0] == min(ok, key=lambda i: d[i])
    # the edge recovers the true relative pose although the estimates drifted by ~0.4 m
    truth = np.linalg.inv(route[out[0]["pair_id"][0]]["truth"]) @ route[-1]["truth"]
    dt, dr = pose_delta(out[0]["relative_pose"], truth)
    assert dt < 0.05 and dr < 3e-3, (dt, dr)
    assert O.search_loop(route, **dict(kw, distance_loop_closure=1e6)) == []
    top3 = O.search_loop(route, **dict(kw, top_k=3))
    assert len(top3) == 3 and top3[0]["pair_id"] == out[0]["pair_id"]
    assert [e["candidate_distance"] for e in top3] == sorted(e["candidate_distance"] for e in top3)
