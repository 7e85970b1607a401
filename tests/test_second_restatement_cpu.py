"""The C++ oracle held to a second, independent restatement (tests/second_restatement.py: fp64 numpy Newton +
More-Thuente for NDT, numpy/scipy GICP) — iteration by iteration for NDT, per outer loop result for GICP.  This is the
only pin available offline for the optimisation loops (the reference ships no vectors and its NDT/GICP sources are an
un-vendored submodule)."""
import numpy as np
import pytest

from lidarslam_ros2_amd import synth
from lidarslam_ros2_amd.posemath import pose_delta
from tests import second_restatement as R2


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="module")
def case():
    return synth.small_case(n_source=2000, n_keyframes=3)


def _table(O, case, res):
    grid = O.VoxelGridCovariance(case.target, res)
    return grid, R2.VoxelTable(grid.dump(), grid.min_b, grid.max_b, res)


def _compare_traces(mine, ref, eps):
    """Per Newton iteration: pose, accepted step, score, derivative passes so far.  A last iteration whose step sits on the
    lower clamp eps/2 is a line search between identical points, decided by rounding noise (fp32 pairs in the oracle, fp64
    here): its number of trial passes is not compared."""
    n = len(mine["trace"])
    for k, (p, score, step, evals) in enumerate(mine["trace"]):
        rp, rscore, rstep, revals = ref["trace"][k][:6], ref["trace"][k][6], ref["trace"][k][7], ref["trace"][k][8]
        on_clamp = (k == n - 1) and abs(rstep - eps / 2) < 1e-12
        assert abs(step - rstep) <= 2e-6 + 1e-3 * abs(rstep), k
        if on_clamp:   # a step of length eps/2 along a Newton direction computed from a gradient that is rounding noise
            assert np.abs(p - rp).max() <= 2.0 * rstep, k
            assert abs(score - rscore) <= 1e-3 * abs(rscore), k
        else:
            assert abs(score - rscore) <= 1e-5 * abs(rscore), k
            assert np.abs(p[:3] - rp[:3]).max() < 5e-5 and np.abs(p[3:] - rp[3:]).max() < 5e-6, k   # fp32 pairs (oracle) vs fp64 (here)
            assert evals == int(revals), k


def test_gauss_constants_second_restatement(O):
    for res in (5.0, 2.0, 1.0):
        d1, d2 = R2.gauss_fit(res)
        o1, o2, _ = O.gauss_constants(res)
        assert abs(d1 - o1) < 1e-12 and abs(d2 - o2) < 1e-12


@pytest.mark.parametrize("res", [5.0, 3.0])
def test_one_derivative_pass_with_hessian_second_restatement(O, case, res):
    """Score, gradient AND Hessian (including the h_ang d1 quirk, both settings) of one pass: the oracle computes pairs in
    fp32, the restatement in fp64 -> agreement at fp32 level."""
    grid, tab = _table(O, case, res)
    d1, d2 = R2.gauss_fit(res)
    rng = np.random.default_rng(2)
    for sign in (+1, -1):
        p = O.matrix_to_pose(case.guess) + np.r_[rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.02, 0.02, 3)]
        s, g, H = R2.ndt_derivatives(tab, case.source, p, d1, d2, d1_sign=sign)
        rs, rg, rH = O.ndt_derivatives(grid, case.source, p, resolution=res, d1_sign=sign)
        assert abs(s - rs) <= 2e-6 * abs(rs)
        assert np.abs(g - rg).max() <= 2e-5 * np.abs(rg).max()
        assert np.abs(H - rH).max() <= 2e-5 * np.abs(rH).max()
        assert np.allclose(H, H.T, rtol=0, atol=1e-9 * np.abs(H).max())


@pytest.mark.parametrize("res,eps,max_iter", [(5.0, 0.01, 35), (3.0, 0.01, 35), (5.0, 1e-4, 30)])
def test_newton_more_thuente_trace_second_restatement(O, case, res, eps, max_iter):
    """Every Newton iteration of the oracle against the independent loop: same number of iterations and derivative passes,
    same accepted step lengths, same poses (fp32-vs-fp64 pair arithmetic apart)."""
    grid, tab = _table(O, case, res)
    p0 = O.matrix_to_pose(case.guess)
    mine = R2.ndt_align(tab, case.source, p0, res, trans_eps=eps, max_iterations=max_iter)
    ref = O.ndt_align(grid, case.source, case.guess, resolution=res, trans_eps=eps, max_iterations=max_iter, trace=True)
    assert mine["iterations"] == ref["iterations"]
    _compare_traces(mine, ref, eps)
    dt, ang = pose_delta(O.pose_to_matrix(mine["p"]), ref["final"])
    assert dt < 1e-4 and ang < 1e-5


def test_line_search_with_trials_second_restatement(O, case):
    """A start far enough from the optimum that the first step is clamped and More-Thuente has to run trial steps
    (gradient-only passes) and recompute the Hessian with the stale h_ang: the pass counts must still agree."""
    res = 5.0
    grid, tab = _table(O, case, res)
    G = case.guess.copy()
    G[:3, 3] += np.array([0.9, -0.7, 0.1], np.float32)
    p0 = O.matrix_to_pose(G)
    mine = R2.ndt_align(tab, case.source, p0, res, trans_eps=1e-3, max_iterations=35)
    ref = O.ndt_align(grid, case.source, G, resolution=res, trans_eps=1e-3, max_iterations=35, trace=True)
    assert ref["n_evals_grad"] > 0, "this start was chosen to exercise trial steps"
    assert mine["iterations"] == ref["iterations"]
    _compare_traces(mine, ref, 1e-3)


def test_gicp_covariances_second_restatement(O, case):
    """numpy SVD regularisation vs the oracle's Jacobi one (bulk tight, ill-conditioned normals loose), k-NN by cKDTree."""
    src = case.source
    mine = R2.gicp_covariances(src)
    ref = O.gicp_covariances(O.NearestNeighbour(src, 1.0), src)
    err = np.abs(mine - ref).max(axis=(1, 2))
    assert np.quantile(err, 0.99) < 1e-5 and err.max() < 5e-2


def test_gicp_outer_loop_second_restatement(O):
    """The whole GICP registration by an independent implementation (cKDTree correspondences, numpy Mahalanobis matrices,
    scipy BFGS solved tightly, PCL's delta stop rule) against the oracle's BFGS schedule and its Gauss-Newton variant:
    final poses inside the north_star bar."""
    case = synth.small_case(n_source=2500, n_keyframes=3, seed=1)
    tgt = O.voxel_grid_filter(case.target, 0.3)
    nn_t, nn_s = O.NearestNeighbour(tgt, 1.0), O.NearestNeighbour(case.source, 1.0)
    ct, cs = O.gicp_covariances(nn_t, tgt), O.gicp_covariances(nn_s, case.source)
    mine = R2.gicp_align(tgt, case.source, case.guess, max_corr_dist=5.0, trans_eps=1e-8)
    for solver in (0, 1):
        ref = O.gicp_align(nn_t, tgt, ct, case.source, cs, case.guess, max_corr_dist=5.0, trans_eps=1e-8, solver=solver)
        dt, ang = pose_delta(mine["final"], ref["final"])
        assert dt <= 1e-3 and ang <= 1e-4, (solver, dt, ang)
    dt, ang = pose_delta(mine["final"], case.truth)
    assert dt < 0.05 and ang < 2e-3
