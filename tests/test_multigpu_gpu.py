"""Multi-GPU path: one process per GPU, static shard of a candidate batch, one RCCL all-gather
of 64-byte result records.  With two devices the ranks take one each and the collectives are RCCL's; on a ONE-device box
(round 6) the two ranks share device 0 and csrc/comm.hip binds tests/cpp/stub_ccl.cpp through LSR_RCCL_LIB — RCCL refuses two
ranks on one device — so every world > 1 line of comm.hip executes there too: no test of this file skips for want of a GPU."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    two = torch.cuda.device_count() >= 2
    dev = rank if two else 0
    torch.cuda.set_device(dev)
    if two:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:   # both ranks on the one device: registrations on the GPU, the record exchange on the host (RCCL refuses this layout)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lidarslam_ros2_amd import NormalDistributionsTransform, align_batch, synth
    from lidarslam_ros2_amd.sharding import pack_record, register_sharded

    n_cand = 6
    cases = [synth.small_case(n_source=2500, n_keyframes=3, seed=c) for c in range(n_cand)]

    def register_local(indices):
        regs = []
        for i in indices:
            r = NormalDistributionsTransform(device=dev)
            r.setResolution(5.0)
            r.setTransformationEpsilon(0.01)
            r.setMaximumIterations(100)
            r.setInputTarget(cases[i].target)
            r.setInputSource(cases[i].source)
            regs.append(r)
        finals, results = align_batch(regs, [cases[i].guess for i in indices])
        return [pack_record(finals[k], results[k]["score"], results[k]["iterations"], results[k]["converged"],
                            regs[k].getFitnessScore()) for k in range(len(indices))]

    res = register_sharded(n_cand, register_local, device=torch.device("cuda", dev) if two else None)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.stack([r["T"] for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_candidates_rccl_world2(tmp_path):
    import torch.multiprocessing as mp

    from lidarslam_ros2_amd import synth
    from lidarslam_ros2_amd.posemath import pose_delta

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b)
    for c in range(6):
        truth = synth.small_case(n_source=2500, n_keyframes=3, seed=c).truth
        dt, ang = pose_delta(a[c], truth)
        assert dt < 0.1 and ang < 5e-3


def _make_regs(cases, device):
    from lidarslam_ros2_amd import NormalDistributionsTransform

    regs = []
    for c in cases:
        r = NormalDistributionsTransform(device=device)
        r.setResolution(5.0)
        r.setTransformationEpsilon(0.01)
        r.setMaximumIterations(100)
        r.setInputTarget(c.target)
        r.setInputSource(c.source)
        regs.append(r)
    return regs


def test_c_abi_sharded_batch_with_a_one_rank_communicator():
    """lsr_align_batch_sharded through a one-rank communicator (no RCCL involved): the records equal what lsr_align_batch
    plus getFitnessScore give, and every candidate lands on its ground truth."""
    from lidarslam_ros2_amd import align_batch, synth
    from lidarslam_ros2_amd.posemath import pose_delta
    from lidarslam_ros2_amd.sharding import Comm, align_batch_sharded

    cases = [synth.small_case(n_source=2500, n_keyframes=3, seed=c) for c in range(5)]
    regs = _make_regs(cases, 0)
    comm = Comm(0, 1, 0)
    res = align_batch_sharded(comm, regs, len(cases), [c.guess for c in cases], with_fitness=True)
    finals, ref = align_batch(_make_regs(cases, 0), [c.guess for c in cases])
    for k, c in enumerate(cases):
        assert np.array_equal(res[k]["T"], finals[k])
        assert res[k]["iterations"] == ref[k]["iterations"] and res[k]["converged"] == ref[k]["converged"]
        assert abs(res[k]["fitness"] - regs[k].getFitnessScore()) <= 1e-6 * abs(res[k]["fitness"])
        dt, ang = pose_delta(res[k]["T"], c.truth)
        assert dt < 0.1 and ang < 5e-3
    comm.close()


def test_c_abi_planned_batch_with_a_one_rank_communicator():
    """lsr_align_batch_planned: members of different size handed over longest first (lsr_shard_plan), records back in batch
    order and equal to the block-partition call's; a plan that is not a permutation is refused."""
    from lidarslam_ros2_amd import synth
    from lidarslam_ros2_amd._capi import RegistrationError
    from lidarslam_ros2_amd.sharding import Comm, ShardPlan, align_batch_sharded, c_shard_plan, registration_cost

    sizes = [1500, 4000, 2500, 3000, 2000]
    cases = [synth.small_case(n_source=n, n_keyframes=3, seed=c) for c, n in enumerate(sizes)]
    plan = c_shard_plan([registration_cost(len(c.target), len(c.source)) for c in cases], 1)
    assert sorted(plan.order.tolist()) == list(range(5)) and plan.order.tolist() != list(range(5))
    comm = Comm(0, 1, 0)
    regs = _make_regs(cases, 0)
    a = align_batch_sharded(comm, regs, len(cases), [c.guess for c in cases], with_fitness=True)
    regs2 = _make_regs([cases[i] for i in plan.items(0)], 0)
    b = align_batch_sharded(comm, regs2, len(cases), [cases[i].guess for i in plan.items(0)], with_fitness=True, plan=plan)
    for k in range(len(cases)):
        assert np.array_equal(a[k]["T"], b[k]["T"]) and a[k]["iterations"] == b[k]["iterations"]
        assert a[k]["fitness"] == pytest.approx(b[k]["fitness"], rel=1e-6)
    bad = ShardPlan(plan.owner, [0, 0, 1, 2, 3], plan.rank_first)
    with pytest.raises(RegistrationError):
        align_batch_sharded(comm, regs2, len(cases), None, plan=bad)
    comm.close()


def _worker_c_abi(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = rank if torch.cuda.device_count() >= 2 else 0   # one device: LSR_RCCL_LIB (set by the test) carries the collectives
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # only carries the 128-byte ncclUniqueId
    from lidarslam_ros2_amd import NormalDistributionsTransform, synth
    from lidarslam_ros2_amd.sharding import Comm, align_batch_sharded, c_shard_plan, c_shard_range, set_input_target_bcast

    n_cand = 7
    box = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = Comm(rank, world, dev, box[0])
    mine = c_shard_range(n_cand, world, rank)
    cases = [synth.small_case(n_source=2500, n_keyframes=3, seed=c) for c in mine]
    res = align_batch_sharded(comm, _make_regs(cases, dev), n_cand, [c.guess for c in cases], with_fitness=True)
    np.save(os.path.join(out_dir, f"c_rank{rank}.npy"), np.stack([np.r_[r["T"].reshape(-1), r["fitness"], r["iterations"]] for r in res]))
    # the longest-first plan through the same communicator (lsr_align_batch_planned): the same table, in batch order
    sizes = [1500 + 400 * ((3 * c) % 5) for c in range(n_cand)]
    plan = c_shard_plan([float(n) for n in sizes], world)
    pc = [synth.small_case(n_source=2500, n_keyframes=3, seed=c) for c in plan.items(rank)]
    res_p = align_batch_sharded(comm, _make_regs(pc, dev), n_cand, [c.guess for c in pc], with_fitness=True, plan=plan)
    np.save(os.path.join(out_dir, f"p_rank{rank}.npy"), np.stack([np.r_[r["T"].reshape(-1), r["fitness"], r["iterations"]] for r in res_p]))
    # "N keyframes vs. one submap": rank 1 holds the submap as a CUDA tensor, both ranks register candidate 5's scan against it
    c5 = synth.small_case(n_source=2500, n_keyframes=3, seed=5)
    reg = NormalDistributionsTransform(device=dev)
    reg.setResolution(5.0); reg.setTransformationEpsilon(0.01); reg.setMaximumIterations(100)
    cloud = torch.from_numpy(synth.as_pointxyzi(c5.target)).cuda() if rank == 1 else None
    set_input_target_bcast(comm, reg, cloud, root=1)
    reg.setInputSource(c5.source)
    reg.align(c5.guess)
    np.save(os.path.join(out_dir, f"b_rank{rank}.npy"), reg.getFinalTransformation())
    # the record all-gather on its own
    g = comm.all_gather_records(np.full((4, 16), float(rank + 1), np.float32))
    assert g.shape == (world, 4, 16) and all(np.all(g[r] == r + 1) for r in range(world))
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def test_c_abi_sharded_batch_rccl_world2(tmp_path, monkeypatch):
    """Two ranks, the exchanges done by the C core itself (lsr_comm_create(world = 2), ncclAllGather behind lsr_align_batch_sharded
    / _planned, the ncclBroadcast chain of lsr_set_input_target_bcast from a non-zero root with a device-resident cloud,
    lsr_comm_all_gather_records).  Two devices: RCCL; one device: both ranks on it, collectives from the stub library."""
    import torch
    import torch.multiprocessing as mp

    from ccl_stub import build_stub
    from lidarslam_ros2_amd import NormalDistributionsTransform, synth
    from lidarslam_ros2_amd.posemath import pose_delta

    if torch.cuda.device_count() < 2:
        monkeypatch.setenv("LSR_RCCL_LIB", build_stub())   # inherited by the spawned ranks
    mp.spawn(_worker_c_abi, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "c_rank0.npy"), np.load(tmp_path / "c_rank1.npy")
    assert np.array_equal(a, b)
    for c in range(7):
        truth = synth.small_case(n_source=2500, n_keyframes=3, seed=c).truth
        dt, ang = pose_delta(a[c, :16].reshape(4, 4), truth)
        assert dt < 0.1 and ang < 5e-3
    # the plan does not change a bit, on either rank
    assert np.array_equal(np.load(tmp_path / "p_rank0.npy")[:, :16], a[:, :16]) and np.array_equal(np.load(tmp_path / "p_rank1.npy"), np.load(tmp_path / "p_rank0.npy"))
    # ... and the table is what ONE process computes (one input, one answer)
    cases = [synth.small_case(n_source=2500, n_keyframes=3, seed=c) for c in range(7)]
    from lidarslam_ros2_amd import align_batch
    finals, _ = align_batch(_make_regs(cases, 0), [c.guess for c in cases])
    assert np.array_equal(np.stack(finals).reshape(7, 16), a[:, :16])
    # the broadcast submap: both ranks built the same grid from the same bytes and landed where a plain setInputTarget lands
    b0, b1 = np.load(tmp_path / "b_rank0.npy"), np.load(tmp_path / "b_rank1.npy")
    ref = NormalDistributionsTransform(device=0)
    ref.setResolution(5.0); ref.setTransformationEpsilon(0.01); ref.setMaximumIterations(100)
    ref.setInputTarget(cases[5].target); ref.setInputSource(cases[5].source); ref.align(cases[5].guess)
    assert np.array_equal(b0, b1) and np.array_equal(b0, ref.getFinalTransformation())


_RCCL1_CODE = r"""
import numpy as np
from lidarslam_ros2_amd import NormalDistributionsTransform, DIRECT7, synth
from lidarslam_ros2_amd.sharding import Comm, align_batch_sharded
def make(cases):
    regs = []
    for c in cases:
        r = NormalDistributionsTransform(device=0); r.setResolution(3.0); r.setTransformationEpsilon(0.01); r.setNeighborhoodSearchMethod(DIRECT7)
        r.setInputTarget(c.target); r.setInputSource(c.source); regs.append(r)
    return regs
cases = [synth.small_case(n_source=2500, n_keyframes=3, seed=c) for c in range(5)]
plain = Comm(0, 1, 0)                        # no RCCL: records copied
a = align_batch_sharded(plain, make(cases), len(cases), [c.guess for c in cases], with_fitness=True)
plain.close()
real = Comm(0, 1, 0, Comm.unique_id())       # ncclGetUniqueId + ncclCommInitRank(nranks = 1): the records go through ncclAllGather
b = align_batch_sharded(real, make(cases), len(cases), [c.guess for c in cases], with_fitness=True)
for x, y in zip(a, b):
    assert np.array_equal(x["T"], y["T"]) and x["iterations"] == y["iterations"] and x["fitness"] == y["fitness"] and x["converged"] == y["converged"]
# the cost-aware plan through the same collective: members handed over longest first, table back in batch order
from lidarslam_ros2_amd.sharding import c_shard_plan
plan = c_shard_plan([float(len(c.source)) * (1 + k % 3) for k, c in enumerate(cases)], 1)
order = plan.items(0)
c = align_batch_sharded(real, make([cases[i] for i in order]), len(cases), [cases[i].guess for i in order], with_fitness=True, plan=plan)
for x, y in zip(a, c):
    assert np.array_equal(x["T"], y["T"]) and x["iterations"] == y["iterations"] and x["converged"] == y["converged"]
    assert abs(x["fitness"] - y["fitness"]) <= 1e-6 * abs(x["fitness"])
# the target broadcast (lsr_set_input_target_bcast) on one rank — RCCL-free communicator and size-1 RCCL communicator —: nothing to exchange, the cloud goes to setInputTarget and must give the voxel grid and the pose of a plain
# setInputTarget (the world > 1 path, ncclBroadcast of header + records, needs two devices: tests/cpp/two_rank_comm.cpp)
from lidarslam_ros2_amd.sharding import set_input_target_bcast
ref = make(cases[:1])[0]
ref.align(cases[0].guess)
one = Comm(0, 1, 0)
for comm, cloud in ((one, synth.as_pointxyzi(cases[0].target)), (real, synth.as_pointxyzi(cases[0].target))):   # (no torch in this process:
    # torch brings its own RCCL into the address space, and two copies of it do not survive ncclCommDestroy)
    r = NormalDistributionsTransform(0); r.setResolution(3.0); r.setTransformationEpsilon(0.01); r.setNeighborhoodSearchMethod(DIRECT7)
    set_input_target_bcast(comm, r, cloud, root=0)
    r.setInputSource(cases[0].source); r.align(cases[0].guess)
    assert np.array_equal(r.getFinalTransformation(), ref.getFinalTransformation())
    ga, gb = r.gridDump(), ref.gridDump()
    assert np.array_equal(ga["idx"], gb["idx"]) and np.array_equal(ga["mean"], gb["mean"])
real.close(); one.close()
print("RCCL1 OK")
"""


def test_c_abi_sharded_batch_through_a_real_rccl_communicator_of_one_rank():
    """RCCL on hardware with the one GPU there is: a communicator of size 1 created from an ncclUniqueId takes the same code
    path as N ranks (dlopen of librccl, ncclCommInitRank, ncclAllGather of the 64-byte records on the communicator's stream,
    unpacking by lsr_shard_range) and must return what the RCCL-free one-rank communicator returns.  (Two ranks on one device
    are refused by RCCL; the two-GPU tests above need a second device.)"""
    import subprocess

    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", _RCCL1_CODE], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and "RCCL1 OK" in p.stdout, (p.stdout[-2500:], p.stderr[-3000:])


@pytest.mark.gpu
def test_c_level_two_process_launcher_runs_the_rccl_path(tmp_path):
    """csrc/comm.hip with world = 2 through the C ABI alone: tests/cpp/two_rank_comm.cpp forks, rank 0
    hands the id to rank 1 over a pipe, and both walk through every world > 1 entry — block and planned batches, the target
    broadcast from either root (host and device-resident cloud), the record all-gather, a failed share that still joins, a refused
    broadcast — ending with the table one process computes for all six.  One device: both ranks on it, collectives from the stub."""
    import os
    import subprocess

    import torch

    from ccl_stub import two_rank_env

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "lidarslam_ros2_amd")
    exe = str(tmp_path / "two_rank_comm")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "two_rank_comm.cpp"), "-o", exe, "-L" + libdir, "-llidarslam_reg", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=two_rank_env(torch.cuda.device_count()))
    out = r.stdout
    assert not out.startswith("SKIP"), out
    assert r.returncode == 0 and "TWO_RANK ok=1 converged=6/6" in out, (out, r.stderr[-2000:])
