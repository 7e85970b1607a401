"""Multi-GPU path: one process per GPU, static shard of a candidate batch, one RCCL all-gather
of 64-byte result records.  Needs >= 2 GPUs — skipped on the 1-GPU boxes; the same sharding code runs in
tests/test_host_cpu.py with gloo, and bench.py's RCCL calls are exercised on one GPU with
LSR_BENCH_FORCE_DIST=1."""
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
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from lidarslam_ros2_amd import NormalDistributionsTransform, align_batch, synth
    from lidarslam_ros2_amd.sharding import pack_record, register_sharded

    n_cand = 6
    cases = [synth.small_case(n_source=2500, n_keyframes=3, seed=c) for c in range(n_cand)]

    def register_local(indices):
        regs = []
        for i in indices:
            r = NormalDistributionsTransform(device=rank)
            r.setResolution(5.0)
            r.setTransformationEpsilon(0.01)
            r.setMaximumIterations(100)
            r.setInputTarget(cases[i].target)
            r.setInputSource(cases[i].source)
            regs.append(r)
        finals, results = align_batch(regs, [cases[i].guess for i in indices])
        return [pack_record(finals[k], results[k]["score"], results[k]["iterations"], results[k]["converged"],
                            regs[k].getFitnessScore()) for k in range(len(indices))]

    res = register_sharded(n_cand, register_local, device=torch.device("cuda", rank))
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.stack([r["T"] for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_candidates_rccl_world2(tmp_path):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
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
