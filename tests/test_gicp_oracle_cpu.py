"""CPU tests that pin the GICP / VoxelGrid / fitness-score parts of the oracle against INDEPENDENT fp64 numpy
restatements and closed forms (SURVEY.md §8c KAT 8 and §9.7-9.8).  The reference ships no tests for this path
(SURVEY.md §4), so these are what anchors the oracle the GPU parity tests compare with."""
import numpy as np
import pytest

from lidarslam_ros2_amd import synth
from lidarslam_ros2_amd.posemath import pose_delta
from oracle import oracle as O


# ---- independent fp64 restatement of the GICP cost (Segal 2009 eq. 2; PCL applyState: R = Rz Ry Rx) ----------
def rot_zyx(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def np_cost(src, tgt, M, x):
    R = rot_zyx(*x[3:])
    r = src.astype(np.float64) @ R.T + x[:3] - tgt.astype(np.float64)
    return float(np.einsum("ia,iab,ib->", r, M, r) / len(src))


def random_spd(rng, n):
    A = rng.normal(size=(n, 3, 3))
    return A @ A.transpose(0, 2, 1) + 0.5 * np.eye(3)


def test_gicp_cost_and_gradient_vs_fp64_numpy():
    rng = np.random.default_rng(7)
    n = 400
    src = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    tgt = (src + rng.normal(0, 0.2, (n, 3))).astype(np.float32)
    M = random_spd(rng, n)
    for x in (np.zeros(6), np.array([0.3, -0.2, 0.1, 0.02, -0.03, 0.05]), np.array([-1.0, 0.5, 0.2, -0.4, 0.3, 1.2])):
        f, g = O.gicp_cost(src, tgt, M, x)
        # the oracle transforms points in fp32 like the reference (applyState -> Matrix4f): 1e-5 relative
        assert f == pytest.approx(np_cost(src, tgt, M, x), rel=2e-5)
        fd = np.zeros(6)
        for k in range(6):
            h = 1e-6
            e = np.zeros(6)
            e[k] = h
            fd[k] = (np_cost(src, tgt, M, x + e) - np_cost(src, tgt, M, x - e)) / (2 * h)
        assert np.allclose(g, fd, rtol=2e-4, atol=2e-4 * np.abs(fd).max())


def test_plane_covariance_is_diag_1_1_eps_in_plane_frame():
    """A perfect plane: every regularised covariance is U diag(1, 1, eps) U^T with U's third axis = the normal."""
    rng = np.random.default_rng(3)
    nrm = np.array([0.3, -0.5, 0.81])
    nrm /= np.linalg.norm(nrm)
    u = np.cross(nrm, [1.0, 0, 0])
    u /= np.linalg.norm(u)
    v = np.cross(nrm, u)
    ab = rng.uniform(-5, 5, (1500, 2))
    pts = (ab[:, :1] * u + ab[:, 1:] * v + 2.0 * nrm)
    # fp32 storage puts the points ~1e-7 off the plane: the smallest eigen-direction is still the normal
    cov = O.gicp_covariances(O.NearestNeighbour(pts), pts, k=20, gicp_eps=1e-3)
    assert np.allclose(cov @ nrm, 1e-3 * nrm, atol=2e-4)
    assert np.allclose(cov @ u, u, atol=2e-4) and np.allclose(cov @ v, v, atol=2e-4)
    w = np.linalg.eigvalsh(cov)
    assert np.allclose(w, [1e-3, 1.0, 1.0], atol=1e-9)


def test_covariances_vs_bruteforce_numpy():
    """k-NN (the point itself included) -> single-pass mean / covariance -> eigen -> diag(1, 1, eps)."""
    rng = np.random.default_rng(5)
    pts = rng.uniform(-3, 3, (600, 3)).astype(np.float32)
    k, eps = 20, 1e-3
    cov = O.gicp_covariances(O.NearestNeighbour(pts), pts, k=k, gicp_eps=eps)
    p64 = pts.astype(np.float64)
    d2 = ((p64[:, None, :] - p64[None, :, :]) ** 2).sum(-1)
    for i in range(0, 600, 37):
        nb = p64[np.argsort(d2[i], kind="stable")[:k]]
        c = np.cov(nb.T, bias=True)
        w, V = np.linalg.eigh(c)       # ascending: the smallest direction gets eps
        ref = V @ np.diag([eps, 1.0, 1.0]) @ V.T
        assert np.allclose(cov[i], ref, atol=5e-5)


def test_knn_is_sorted_and_exact_vs_bruteforce():
    rng = np.random.default_rng(11)
    pts = rng.uniform(-20, 20, (3000, 3)).astype(np.float32)
    q = rng.uniform(-25, 25, (200, 3)).astype(np.float32)   # some queries outside the cloud's bounding box
    idx, d2 = O.NearestNeighbour(pts, cell=1.0).knn(q, 8)
    diff = q[:, None, :] - pts[None, :, :]
    bd2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]  # fp32, PCL order
    order = np.lexsort((np.broadcast_to(np.arange(3000), bd2.shape), bd2), axis=1)[:, :8]
    assert np.array_equal(idx, order)
    assert np.array_equal(d2, np.take_along_axis(bd2, order, 1))
    assert np.all(np.diff(d2, axis=1) >= 0)


def test_identical_patches_zero_cost_and_identity():
    rng = np.random.default_rng(13)
    pts = np.c_[rng.uniform(-4, 4, (800, 2)), 0.05 * rng.normal(size=800)].astype(np.float32)
    nn = O.NearestNeighbour(pts)
    cov = O.gicp_covariances(nn, pts)
    for solver in (0, 1):
        r = O.gicp_align(nn, pts, cov, pts, cov, None, solver=solver)
        assert r["converged"] and r["n_correspondences"] == 800
        assert r["final_cost"] == pytest.approx(0.0, abs=1e-12)
        assert np.allclose(r["final"], np.eye(4), atol=1e-7)


def test_bfgs_and_gauss_newton_reach_the_same_pose_and_recover_truth():
    case = synth.small_case(n_source=3000, n_keyframes=3)
    tgt = O.voxel_grid_filter(case.target, 0.4)
    nn_t, nn_s = O.NearestNeighbour(tgt), O.NearestNeighbour(case.source)
    ct, cs = O.gicp_covariances(nn_t, tgt), O.gicp_covariances(nn_s, case.source)
    bfgs = O.gicp_align(nn_t, tgt, ct, case.source, cs, case.guess, solver=0)
    gn = O.gicp_align(nn_t, tgt, ct, case.source, cs, case.guess, solver=1)
    assert bfgs["converged"] and gn["converged"]
    dt, da = pose_delta(bfgs["final"], gn["final"])
    assert dt < 1e-3 and da < 1e-4          # north_star's parity bar between the two inner solvers
    dt, da = pose_delta(gn["final"], case.truth)
    assert dt < 0.05 and da < 5e-3          # and both sit on the true pose (sensor noise 2 cm)


def test_max_correspondence_distance_gates_pairs():
    rng = np.random.default_rng(17)
    tgt = rng.uniform(-5, 5, (500, 3)).astype(np.float32)
    src = (tgt[:200] + np.array([0.0, 0.0, 30.0], np.float32))   # 30 m away: no pair within 5 m
    nn_t = O.NearestNeighbour(tgt)
    ct, cs = O.gicp_covariances(nn_t, tgt), O.gicp_covariances(O.NearestNeighbour(src), src)
    r = O.gicp_align(nn_t, tgt, ct, src, cs, None, max_corr_dist=5.0)
    assert not r["converged"] and r["n_correspondences"] == 0 and r["iterations"] == 0
    assert np.allclose(r["final"], np.eye(4))
    r = O.gicp_align(nn_t, tgt, ct, src, cs, None, max_corr_dist=100.0)
    assert r["n_correspondences"] == 200


# ---- pcl::VoxelGrid::filter and getFitnessScore restatements --------------------------------------------
def test_voxel_grid_filter_vs_bruteforce_numpy():
    rng = np.random.default_rng(19)
    pts = rng.uniform(-7, 9, (5000, 3)).astype(np.float32)
    pts[::97] = np.nan                                            # non-finite points are dropped
    for leaf in (0.5, 2.0):
        out = O.voxel_grid_filter(pts, leaf)
        ok = np.isfinite(pts).all(1)
        p = pts[ok]
        inv = np.float32(1.0) / np.float32(leaf)
        mn = np.floor(p.min(0) * inv).astype(np.int64)
        mx = np.floor(p.max(0) * inv).astype(np.int64)
        div = mx - mn + 1
        ijk = (np.floor(p * inv) - mn.astype(np.float32)).astype(np.int64)
        key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
        uniq, inverse = np.unique(key, return_inverse=True)       # ascending leaf index = PCL's output order
        assert out.shape[0] == len(uniq)
        cen = np.zeros((len(uniq), 3))
        np.add.at(cen, inverse, p.astype(np.float64))
        cen /= np.bincount(inverse)[:, None]
        assert np.allclose(out, cen, atol=2e-6)
        # idempotence-like property: every centroid falls into its own leaf, so filtering again keeps the count
        assert O.voxel_grid_filter(out, leaf).shape[0] == out.shape[0]


def test_fitness_score_vs_bruteforce_and_max_range():
    rng = np.random.default_rng(23)
    tgt = rng.uniform(-10, 10, (4000, 3)).astype(np.float32)
    src = rng.uniform(-12, 12, (700, 3)).astype(np.float32)
    T = synth.pose_matrix(0.4, -0.3, 0.1, 0.05, 0.01, -0.02).astype(np.float32)
    nn = O.NearestNeighbour(tgt)
    moved = O.transform_point_cloud(src, T)
    d2 = ((moved[:, None, :].astype(np.float64) - tgt[None, :, :].astype(np.float64)) ** 2).sum(-1).min(1)
    assert nn.fitness_score(src, T) == pytest.approx(d2.mean(), rel=1e-6)
    mr = 0.5
    keep = d2 <= mr
    assert keep.any() and not keep.all()
    assert nn.fitness_score(src, T, max_range=mr) == pytest.approx(d2[keep].mean(), rel=1e-6)


def test_oracle_reproduces_the_gicp_golden_fixture():
    """tests/golden/gicp_small_golden.npz was written by tests/golden/make_golden_gicp.py from this oracle: any drift of
    the restatement (k-NN order, covariance recipe, either inner solver, the fitness sum, the voxel filter) shows here.
    Integer work bit-exact; floating point to the last few ulps (thread count and summation order are fixed)."""
    import os

    from golden_fixtures import load_golden

    gold, origin = load_golden("gicp_small_golden")
    print("[golden] the GICP oracle is checked against the %s fixture" % origin)
    case = synth.small_case(n_source=int(gold["n_source"]), n_keyframes=int(gold["n_keyframes"]))
    assert case.target.shape[0] == int(gold["n_target_raw"]) and np.array_equal(case.source, gold["source"])
    tgt = O.voxel_grid_filter(case.target, float(gold["leaf"]))
    assert tgt.shape[0] == int(gold["n_target"]) and np.array_equal(tgt[:64], gold["target_head"])
    nn_t, nn_s = O.NearestNeighbour(tgt), O.NearestNeighbour(case.source)
    idx, d2 = nn_t.search(case.source, gold["guess"], num_threads=1)
    assert np.array_equal(idx, gold["nn_idx"]) and np.array_equal(d2, gold["nn_d2"])
    ct, cs = O.gicp_covariances(nn_t, tgt, num_threads=1), O.gicp_covariances(nn_s, case.source, num_threads=1)
    assert np.allclose(cs[:200], gold["cov_src_head"], rtol=0, atol=1e-13)
    assert np.allclose(ct[:200], gold["cov_tgt_head"], rtol=0, atol=1e-13)
    for solver, key in ((0, "bfgs"), (1, "gn")):
        r = O.gicp_align(nn_t, tgt, ct, case.source, cs, gold["guess"], solver=solver, num_threads=1)
        assert r["iterations"] == int(gold["iters_" + key])
        assert np.allclose(r["final"], gold["final_" + key], rtol=0, atol=1e-6)
        if solver == 1:
            assert r["n_correspondences"] == int(gold["n_corr"])
            fit = nn_t.fitness_score(case.source, r["final"], num_threads=1)
            assert fit == pytest.approx(float(gold["fitness"]), rel=1e-9)
