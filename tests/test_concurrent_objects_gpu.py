"""Two registration objects driven from two threads at the same time — what the reference's MultiThreadedExecutor does with the
frontend's and the backend's object (lidarslam/src/lidarslam.cpp:12-17; INTEGRATION.md: one caller per object, objects run
concurrently: each owns its HIP stream, its buffers and its host mailbox).  ctypes releases the GIL inside the C entries, so the
threads really are inside the library together.  Every result must equal, bit for bit, what the same object returns on its own."""
import threading

import numpy as np
import pytest

from lidarslam_ros2_amd import synth

pytestmark = pytest.mark.gpu

ROUNDS = 40


def _frontend_job():
    """scan preprocessing (N1: range filter + VoxelGrid) + NDT align at the reference's settings, as receiveCloud does per scan"""
    from lidarslam_ros2_amd import DIRECT7, NormalDistributionsTransform
    from lidarslam_ros2_amd.frontend import as_pc2_payload

    case = synth.small_case(n_source=3000, n_keyframes=3)
    raw = synth.raycast(synth.make_world(), synth.Sensor(16, -20.0, 12.0, 900), case.truth, np.random.default_rng(7))
    payload = as_pc2_payload(raw)
    r = NormalDistributionsTransform(device=0)
    r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(35); r.setNeighborhoodSearchMethod(DIRECT7)
    r.setInputTarget(case.target)

    def once():
        n = r.setInputSourcePointCloud2(payload, raw.shape[0], 32, (0, 4, 8, 16), 0.5, 80.0, 0.4)
        r.align(case.guess)
        return (n, np.array(r.getFinalTransformation()), r.getFinalNumIteration(), r.getInputSourcePointCloud2().tobytes())
    return once


def _backend_job():
    """a loop-closure candidate as searchLoop() treats it: setInputTarget + setInputSource + align + getFitnessScore, a new target each time"""
    from lidarslam_ros2_amd import NormalDistributionsTransform

    cases = [synth.small_case(n_source=2500, n_keyframes=4, seed=s) for s in (11, 12)]
    r = NormalDistributionsTransform(device=0)
    r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(100)
    state = {"k": 0}

    def once():
        c = cases[state["k"] % 2]
        state["k"] += 1
        r.setInputTarget(c.target)
        r.setInputSource(c.source)
        r.align(c.guess)
        return (state["k"] % 2, np.array(r.getFinalTransformation()), r.getFinalNumIteration(), r.getFitnessScore())
    return once


def _gicp_job():
    from lidarslam_ros2_amd import GeneralizedIterativeClosestPoint

    c = synth.small_case(n_source=2000, n_keyframes=3, seed=21)
    g = GeneralizedIterativeClosestPoint(device=0)
    g.setMaxCorrespondenceDistance(5.0); g.setTransformationEpsilon(1e-8); g.setMaximumIterations(30)
    g.setInputTarget(c.target)

    def once():
        g.setInputSource(c.source)
        g.align(c.guess)
        return (np.array(g.getFinalTransformation()), g.getFitnessScore())
    return once


def _candidate_set_job():
    """a candidate set through the staged batch entries (lsr_set_input_target_batch / _source_batch / lsr_align_fitness_batch): its lead
    object owns a side stream and chain streams of its own"""
    from lidarslam_ros2_amd import NormalDistributionsTransform
    from lidarslam_ros2_amd.registration import align_fitness_batch, set_input_source_batch, set_input_target_batch

    cases = [synth.small_case(n_source=2000, n_keyframes=3, seed=s) for s in (31, 32, 33, 34, 35, 36)]
    regs = []
    for _ in cases:
        r = NormalDistributionsTransform(device=0)
        r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(100)
        regs.append(r)

    def once():
        set_input_target_batch(regs, [synth.as_pointxyzi(c.target) for c in cases])
        set_input_source_batch(regs, [synth.as_pointxyzi(c.source) for c in cases])
        finals, res, fit = align_fitness_batch(regs, [c.guess for c in cases])
        return (np.array(finals), tuple(int(x["iterations"]) for x in res), np.array(fit))
    return once


def _same(a, b):
    if isinstance(a, tuple):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return np.array_equal(a, b)
    return a == b


def test_objects_driven_from_four_threads_return_what_they_return_alone():
    jobs = [_frontend_job(), _backend_job(), _gicp_job(), _candidate_set_job()]
    # alone, one after the other: the reference answers (two rounds each: the backend job alternates between two candidates)
    alone = [[job() for _ in range(2)] for job in jobs]
    for ref in alone[0::2]:
        assert _same(ref[0], ref[1])                       # same input twice: same bits (frontend, GICP)
    results = [[] for _ in jobs]
    errors = []
    start = threading.Barrier(len(jobs))

    def run(k):
        try:
            start.wait()
            for _ in range(ROUNDS):
                results[k].append(jobs[k]())
        except Exception as e:   # noqa: BLE001 — reported by the assertion below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert all(len(r) == ROUNDS for r in results)
    for i, got in enumerate(results[0]):
        assert _same(got, alone[0][0]), ("frontend object, round", i)
    for i, got in enumerate(results[2]):
        assert _same(got, alone[2][0]), ("GICP object, round", i)
    assert _same(alone[3][0], alone[3][1])
    for i, got in enumerate(results[3]):
        assert _same(got, alone[3][0]), ("candidate set, round", i)
    for i, got in enumerate(results[1]):                    # rounds continue the alternation the two solo rounds started
        assert _same(got, alone[1][i % 2]), ("backend object, round", i)


def test_objects_sharing_one_target_from_three_threads():
    """N keyframes against ONE submap (lsr_share_target), every keyframe's object on a thread of its own: the lazy builds on the shared
    target (voxel grid, neighbour grid for getFitnessScore) are behind the target's mutex; every object returns what it returns alone."""
    from lidarslam_ros2_amd import NormalDistributionsTransform

    base = synth.small_case(n_source=2500, n_keyframes=4, seed=41)
    world = synth.make_world()
    owner = NormalDistributionsTransform(device=0)
    owner.setResolution(5.0)
    owner.setInputTarget(base.target)
    sources = [base.source] + [synth.voxel_downsample(synth.raycast(world, synth.Sensor(16, -20.0, 12.0, 600), base.truth, np.random.default_rng(50 + k)), 0.4)
                               for k in range(2)]

    def make(src):
        r = NormalDistributionsTransform(device=0)
        r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(60)
        r.shareTargetOf(owner)

        def once():
            r.setInputSource(src)
            r.align(base.guess)
            return (np.array(r.getFinalTransformation()), r.getFinalNumIteration(), r.getFitnessScore())
        return once

    # the solo answers come from objects of their own, so that the threads below meet a target whose lazy builds have NOT run for them
    alone = [make(s)() for s in sources]
    owner2 = NormalDistributionsTransform(device=0)
    owner2.setResolution(5.0)
    owner2.setInputTarget(base.target)
    owner_keep, owner = owner, owner2          # fresh target: grid and neighbour grid are built under the threads
    jobs = [make(s) for s in sources]
    results = [[] for _ in jobs]
    errors = []
    start = threading.Barrier(len(jobs))

    def run(k):
        try:
            start.wait()
            for _ in range(ROUNDS):
                results[k].append(jobs[k]())
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    for k in range(len(jobs)):
        assert len(results[k]) == ROUNDS
        for i, got in enumerate(results[k]):
            assert _same(got, alone[k]), ("object", k, "round", i)
    del owner_keep


def test_a_member_of_one_set_leads_the_next_set():
    """The candidate sets rotate: the object that led a set is a member of the next, a member leads it — what one set left in flight
    on its lead's stream for a member (lsr_set_input_source_batch defers the members' own streams behind it) has to be honoured
    when that member's stream becomes the lead stream of the following call."""
    from lidarslam_ros2_amd import NormalDistributionsTransform
    from lidarslam_ros2_amd.registration import align_fitness_batch, set_input_source_batch, set_input_target_batch

    cases = [synth.small_case(n_source=2000, n_keyframes=3, seed=s) for s in (41, 42, 43, 44)]
    regs = []
    for _ in cases:
        r = NormalDistributionsTransform(device=0)
        r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(100)
        regs.append(r)
    set_input_target_batch(regs, [synth.as_pointxyzi(c.target) for c in cases])
    sources = [synth.as_pointxyzi(c.source) for c in cases]
    alone = []
    for r, c in zip(regs, cases):
        r.setInputSource(c.source); r.align(c.guess)
        alone.append((np.array(r.getFinalTransformation()), r.getFinalNumIteration(), r.getFitnessScore()))
    order = list(range(len(regs)))
    for rnd in range(12):
        # every object gets ANOTHER object's source first (from a set led by order[0]) …
        wrong = order[1:] + order[:1]
        set_input_source_batch([regs[i] for i in order], [sources[j] for j in wrong])
        # … and its own one right after, from a set led by what was a member a moment ago
        order = order[1:] + order[:1]
        set_input_source_batch([regs[i] for i in order], [sources[i] for i in order])
        finals, res, fit = align_fitness_batch([regs[i] for i in order], [cases[i].guess for i in order])
        for k, i in enumerate(order):
            assert np.array_equal(np.asarray(finals[k]), alone[i][0]), (rnd, i)
            assert int(res[k]["iterations"]) == alone[i][1] and float(fit[k]) == pytest.approx(alone[i][2], rel=1e-12), (rnd, i)
