"""CPU tests of the DEVICE nearest-neighbour search code (lidarslam_ros2_amd/csrc/nn_device.hpp) compiled for the host
(tools/nn_host_emu/harness.cpp): the per-thread walk with its row pruning and x clipping against brute force — indices and
fp32 distances bit for bit, ties by lowest index —, and the slot enumeration of the wave-cooperative search: the segments of
a shell cover the shell's cells exactly once, and pruning / clipping never drop a point that could still matter."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    d = tmp_path_factory.mktemp("nn_host_emu")
    src = open(os.path.join(ROOT, "lidarslam_ros2_amd", "csrc", "nn_device.hpp")).read()
    src = src.replace('#include "common.hpp"', "").replace("\ninline NNGridView make_view", "\nstatic inline NNGridView make_view")
    open(d / "nn_device_emu.hpp", "w").write(src)
    harness = open(os.path.join(ROOT, "tools", "nn_host_emu", "harness.cpp")).read()
    open(d / "harness.cpp", "w").write(harness)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", str(d / "libnnh.so"), str(d / "harness.cpp")])
    return C.CDLL(str(d / "libnnh.so"))


def build_grid(pts, cell):
    """The two-level grid of nn.hip, built with numpy (points ordered by (coarse cell, fine cell), stable)."""
    inv = np.float32(1.0) / np.float32(cell)
    f = np.floor(pts * inv).astype(np.int64)
    f0, f1 = f.min(0), f.max(0)
    org = np.where(f0 >= 0, f0 & ~7, -(((-f0) + 7) & ~7))
    cdim = ((f1 - org) >> 3) + 1
    rel = f - org
    cc = rel >> 3
    clin = cc[:, 0] + cdim[0] * (cc[:, 1] + cdim[1] * cc[:, 2])
    fine = (rel[:, 0] & 7) | ((rel[:, 1] & 7) << 3) | ((rel[:, 2] & 7) << 6)
    order = np.argsort(clin * 512 + fine, kind="stable").astype(np.int32)
    ublk = np.unique(clin)
    coarse_block = -np.ones(int(np.prod(cdim)), np.int32)
    coarse_block[ublk] = np.arange(len(ublk), dtype=np.int32)
    blk_of = coarse_block[clin[order]]
    fkey = blk_of.astype(np.int64) * 513 + fine[order]
    fine_start = np.searchsorted(fkey, np.arange(len(ublk) * 513), side="left").astype(np.int32)
    block_off = np.searchsorted(blk_of, np.arange(len(ublk) + 1), side="left").astype(np.int32)
    s = pts[order]
    return dict(cell=np.float32(cell), org=org.astype(np.int32), cdim=cdim.astype(np.int32), coarse_block=coarse_block, block_off=block_off,
                fine_start=fine_start, sx=np.ascontiguousarray(s[:, 0]), sy=np.ascontiguousarray(s[:, 1]), sz=np.ascontiguousarray(s[:, 2]),
                order=order, rel=rel, pts=pts)


def P(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def grid_args(g):
    return (C.c_float(float(g["cell"])), P(g["org"], C.c_int), P(g["cdim"], C.c_int), P(g["coarse_block"], C.c_int), P(g["block_off"], C.c_int),
            P(g["fine_start"], C.c_int), P(g["sx"], C.c_float), P(g["sy"], C.c_float), P(g["sz"], C.c_float), P(g["order"], C.c_int), len(g["order"]))


def brute_nn1(pts, q):
    """(distance, index) minimum with the device's fp32 arithmetic: ((dx*dx + dy*dy) + dz*dz), no contraction."""
    d = (pts - q[None, :]).astype(np.float32)
    d2 = ((d[:, 0] * d[:, 0]).astype(np.float32) + (d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
    d2 = (d2 + (d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)
    j = int(np.lexsort((np.arange(len(pts)), d2))[0])
    return j, d2[j]


def clouds():
    from lidarslam_ros2_amd import synth

    rng = np.random.default_rng(3)
    case = synth.small_case(n_source=1500, n_keyframes=2)
    dup = rng.uniform(-4, 4, (300, 3)).astype(np.float32)
    dup = np.vstack([dup, dup[:150]])            # exact duplicates: ties broken by the lowest index
    lattice = (np.stack(np.meshgrid(*[np.arange(-6, 6)] * 3, indexing="ij"), -1).reshape(-1, 3) * 0.5).astype(np.float32)   # points ON cell faces
    return [("uniform", rng.uniform(-12, 12, (4000, 3)).astype(np.float32)), ("scan", case.target[:6000].astype(np.float32)), ("duplicates", dup),
            ("lattice", lattice)]


@pytest.mark.parametrize("cell", [0.5, 0.3, 1.0])
def test_per_thread_walk_with_pruning_matches_brute_force(emu, cell):
    rng = np.random.default_rng(11)
    for name, pts in clouds():
        g = build_grid(pts, cell)
        q = np.vstack([pts[rng.integers(0, len(pts), 150)] + rng.normal(0, 0.2, (150, 3)).astype(np.float32),
                       rng.uniform(-15, 15, (50, 3)).astype(np.float32), pts[:30]]).astype(np.float32)      # near, anywhere (far outliers), exact hits
        qx, qy, qz = [np.ascontiguousarray(q[:, k]) for k in range(3)]
        idx = np.zeros(len(q), np.int32)
        d2 = np.zeros(len(q), np.float32)
        for fine_rings in (0, 1):
            emu.run_nn1(*grid_args(g), P(qx, C.c_float), P(qy, C.c_float), P(qz, C.c_float), len(q), fine_rings, C.c_float(np.inf),
                        P(idx, C.c_int), P(d2, C.c_float))
            for i in range(len(q)):
                j, dj = brute_nn1(pts, q[i])
                assert idx[i] == j and d2[i] == dj, (name, cell, fine_rings, i, idx[i], j, d2[i], dj)
        # with a distance gate: the answer inside the gate is unchanged, nothing is invented outside it
        gate = np.float32(0.25)
        emu.run_nn1(*grid_args(g), P(qx, C.c_float), P(qy, C.c_float), P(qz, C.c_float), len(q), 1, C.c_float(gate), P(idx, C.c_int), P(d2, C.c_float))
        for i in range(len(q)):
            j, dj = brute_nn1(pts, q[i])
            if dj <= gate:
                assert idx[i] == j and d2[i] == dj, (name, cell, i)


@pytest.mark.parametrize("cell", [0.5, 0.37])
def test_shell_slots_cover_their_cells_exactly_once_and_pruning_is_conservative(emu, cell):
    rng = np.random.default_rng(5)
    for name, pts in clouds()[:2]:
        g = build_grid(pts, cell)
        rel = g["rel"][g["order"]]                                   # fine coordinates of the cell-ordered points
        sorted_pts = pts[g["order"]]
        for q in np.vstack([pts[rng.integers(0, len(pts), 12)] + rng.normal(0, 0.3, (12, 3)).astype(np.float32),
                            rng.uniform(-13, 13, (4, 3)).astype(np.float32)]).astype(np.float32):
            fq = np.floor(q * (np.float32(1.0) / np.float32(cell))).astype(np.int64) - g["org"]
            d = (sorted_pts - q[None, :]).astype(np.float64)
            dist2 = (d * d).sum(1)
            for r in range(0, 5):
                out = np.zeros(2 * 4 * (2 * r + 1) ** 2, np.int32)
                n_slots = emu.run_shell_segments(*grid_args(g), C.c_float(q[0]), C.c_float(q[1]), C.c_float(q[2]), r, 0, C.c_float(np.inf),
                                                 C.c_float(np.inf), P(out, C.c_int))
                assert n_slots == 4 * (2 * r + 1) ** 2
                got = np.concatenate([np.arange(out[2 * s], out[2 * s] + out[2 * s + 1]) for s in range(n_slots)] or [np.zeros(0, np.int64)])
                cheb = np.abs(rel - fq[None, :]).max(1)
                want = np.nonzero(cheb == r)[0]
                assert len(got) == len(np.unique(got)), (name, r, "a point was handed out twice")
                assert np.array_equal(np.sort(got), want), (name, cell, r, len(got), len(want))
                # pruned / clipped: every point of the shell that beats `worst` (ties included) must still be there
                for worst in (np.float32(0.01), np.float32(0.2), np.float32(1.5)):
                    emu.run_shell_segments(*grid_args(g), C.c_float(q[0]), C.c_float(q[1]), C.c_float(q[2]), r, 1, C.c_float(worst),
                                           C.c_float(np.inf), P(out, C.c_int))
                    kept = np.concatenate([np.arange(out[2 * s], out[2 * s] + out[2 * s + 1]) for s in range(n_slots)] or [np.zeros(0, np.int64)])
                    assert len(kept) == len(np.unique(kept))
                    assert set(kept.tolist()) <= set(want.tolist())
                    must = want[dist2[want] <= float(worst) * (1.0 + 1e-6)]
                    assert set(must.tolist()) <= set(kept.tolist()), (name, cell, r, float(worst))


@pytest.mark.parametrize("cell", [0.5, 0.37])
def test_ball_cell_range_holds_every_point_of_the_ball(emu, cell):
    """The seeded GICP correspondence search reads ONLY the cells ball_cell_range() names: every point within the seed's
    distance (ties included) must lie in that box, whatever the cell size and wherever the query sits in its cell."""
    rng = np.random.default_rng(9)
    for name, pts in clouds()[:2] + [clouds()[3]]:
        g = build_grid(pts, cell)
        rel = g["rel"]
        for _ in range(300):
            seed = int(rng.integers(0, len(pts)))
            q = (pts[seed] + rng.normal(0, rng.choice([0.02, 0.1, 0.3]), 3)).astype(np.float32)
            j, d2 = brute_nn1(pts, q)
            dseed = brute_nn1(pts[seed:seed + 1], q)[1]           # the seed's distance with the device arithmetic
            out = np.zeros(6, np.int32)
            ok = emu.run_ball_range(C.c_float(float(g["cell"])), P(g["org"], C.c_int), P(g["cdim"], C.c_int), C.c_float(q[0]), C.c_float(q[1]),
                                    C.c_float(q[2]), C.c_float(dseed), 3, P(out, C.c_int))
            if not ok:
                continue                                          # deferred to the general search
            lo, hi = out[:3], out[3:]
            d = (pts - q[None, :]).astype(np.float32)
            dd = ((d[:, 0] * d[:, 0]).astype(np.float32) + (d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
            dd = (dd + (d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)
            inside = np.nonzero(dd <= dseed)[0]
            assert len(inside) >= 1
            assert np.all((rel[inside] >= lo[None, :]) & (rel[inside] <= hi[None, :])), (name, cell, q, lo, hi)
            assert np.all(hi - lo <= 2)
            assert (rel[j] >= lo).all() and (rel[j] <= hi).all()   # in particular the true nearest neighbour
