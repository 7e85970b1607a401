"""The GPU derivative pass's arithmetic on the CPU (tests/ndt_host_emu.py: lidarslam_ros2_amd/csrc/ndt_point.hpp compiled for
the host, plugged into the oracle's Newton / More-Thuente loop) against the oracle's own per-pair arithmetic — no GPU.

What this pins: the factorised form of the kernels (A = sum w C q, E = sum w (C - d2 Cq Cq^T), fp32 sums over a point's voxels,
29 per-point terms), the reference-order point transform and the head + tail voxel means reproduce the reference's recipe
 to ~1e-8 on a whole pass and to the same Newton trajectory on a whole registration; and that the two
shortcuts round 2's kernels took (fmaf-chain transform, fp32-rounded means) were what separated the GPU from the oracle."""
import os

import numpy as np
import pytest

from lidarslam_ros2_amd import synth
from lidarslam_ros2_amd.posemath import pose_delta

import ndt_host_emu as E


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def case():
    return synth.small_case(n_source=4000, n_keyframes=3)


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


def test_one_pass_matches_the_oracle_arithmetic(O, case):
    res = 3.0
    g = O.VoxelGridCovariance(case.target, res)
    e = E.Emu(O, g, case.source, res)
    p0 = O.matrix_to_pose(case.guess)
    for p, T in ((p0, case.guess), (p0 + np.array([0.05, -0.03, 0.01, 0.002, -0.001, 0.004]), None)):
        s1, g1, H1 = O.ndt_derivatives(g, case.source, p, T=T, resolution=res)
        s2, g2, H2 = e.derivatives(p, T=T)
        assert abs(s1 - s2) <= 1e-8 * abs(s1)
        assert _rel(g2, g1) <= 5e-8 and _rel(H2, H1) <= 5e-8      # measured ~4e-9 / ~1e-9: expf + fp32 association noise only
        s3, g3, _ = e.derivatives(p, T=T, with_hessian=False)
        assert s3 == s2 and np.array_equal(g3, g2)               # gradient-only passes form the same 8 terms


def test_round2_shortcuts_were_the_gap(O, case, monkeypatch):
    """fmaf-chain transform + fp32 means (EMU_R02_ARITH, read when the emulation library is first used by a process): the
    gradient of a pass moves by ~1e-7..1e-6 relative — two orders more than with the reference's order.  Runs in a
    subprocess so that the switch is seen."""
    import subprocess
    import sys

    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from lidarslam_ros2_amd import synth; from oracle import oracle as O; import ndt_host_emu as E\n"
            "c = synth.small_case(n_source=4000, n_keyframes=3); g = O.VoxelGridCovariance(c.target, 3.0)\n"
            "p = O.matrix_to_pose(c.guess); e = E.Emu(O, g, c.source, 3.0)\n"
            "s1, g1, H1 = O.ndt_derivatives(g, c.source, p, T=c.guess, resolution=3.0); s2, g2, H2 = e.derivatives(p, T=c.guess)\n"
            "print(float(np.abs(g1 - g2).max() / np.abs(g1).max()))\n") % (E.ROOT, os.path.join(E.ROOT, "tests"))
    env = dict(os.environ, EMU_R02_ARITH="1")
    r02 = float(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, check=True).stdout.split()[-1])
    env.pop("EMU_R02_ARITH")
    now = float(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, check=True).stdout.split()[-1])
    assert now <= 5e-8 and r02 >= 10 * now, (now, r02)


@pytest.mark.parametrize("eps,max_iter", [(0.01, 35), (1e-6, 60)])
def test_whole_registration_follows_the_oracle(O, case, eps, max_iter):
    res = 3.0
    g = O.VoxelGridCovariance(case.target, res)
    ref = O.ndt_align(g, case.source, case.guess, resolution=res, trans_eps=eps, max_iterations=max_iter, trace=True)
    emu = E.ndt_align(O, g, case.source, case.guess, res, trans_eps=eps, max_iterations=max_iter, trace=True)
    dt, ang = pose_delta(ref["final"], emu["final"])
    assert dt <= 1e-5 and ang <= 1e-6, (dt, ang)
    if eps >= 1e-3:
        assert emu["iterations"] == ref["iterations"]
        n = ref["iterations"]
        assert np.abs(ref["trace"][:n, :6] - emu["trace"][:n, :6]).max() <= 1e-5      # pose after every Newton iteration
        assert np.array_equal(ref["trace"][:n, 8], emu["trace"][:n, 8])               # derivative passes so far: same line searches
