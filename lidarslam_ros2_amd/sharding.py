"""Multi-GPU sharding of a batch of independent registrations.

The path shards at registration granularity: each (target, source, guess) triple is independent
(the loop-closure candidate scan of graph_based_slam_component.cpp:190-231 generalised from arg-min
to top-k, or N keyframes against a submap).  One process per GPU, static block partition of the
batch — or, where the members differ in size, a cost-aware longest-first plan (shard_plan) — no collective on the data path; the only exchange is ONE all-gather of fixed 64-byte result
records (3x4 pose fp32, score, iterations, converged, fitness) — `torch.distributed` backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  Splitting ONE registration over GPUs would need
a 29-double all-reduce per derivative pass (hundreds per align): pure latency, so a single
registration is "replicas only".
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np

RECORD_FLOATS = 16  # 64 bytes: T[:3,:4] (12) | score | iterations | converged | fitness


def shard_range(n_items: int, world: int, rank: int) -> range:
    """Static block partition: the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


class ShardPlan:
    """owner[i] = rank of item i; order = the batch regrouped rank by rank, each rank's items longest first; rank r owns
    order[rank_first[r]:rank_first[r + 1]] (lsr_shard_plan of the C ABI computes the same arrays)."""

    def __init__(self, owner, order, rank_first):
        self.owner = np.asarray(owner, np.int32)
        self.order = np.asarray(order, np.int32)
        self.rank_first = np.asarray(rank_first, np.int32)

    @property
    def world(self) -> int:
        return len(self.rank_first) - 1

    def items(self, rank: int) -> List[int]:
        return [int(i) for i in self.order[self.rank_first[rank]:self.rank_first[rank + 1]]]

    def loads(self, costs) -> np.ndarray:
        c = np.asarray(costs, np.float64)
        return np.array([c[self.items(r)].sum() for r in range(self.world)])


def block_plan(n_items: int, world: int) -> ShardPlan:
    """shard_range as a plan: order = identity."""
    owner = np.zeros(n_items, np.int32)
    first = [0]
    for r in range(world):
        rr = shard_range(n_items, world, r)
        owner[rr.start:rr.stop] = r
        first.append(rr.stop)
    return ShardPlan(owner, np.arange(n_items, dtype=np.int32), first)


def shard_plan(costs, world: int) -> ShardPlan:
    """Longest-processing-time-first: items by cost descending (ties: lower index first), each to the rank with the least
    load so far (ties: lower rank).  Within 4/3 - 1/(3 world) of the best makespan; costs within 2 % of each other give the block
    partition (differences below the noise of the cost model should not reshuffle a set).  The ring
    gate's candidate sets are the case it is for: targets from a few thousand to 661 k points in one set."""
    c = np.asarray(costs, np.float64)
    if c.ndim != 1 or not np.all(np.isfinite(c)) or np.any(c < 0):
        raise ValueError("costs must be a 1-D array of finite non-negative numbers")
    n = len(c)
    if n == 0 or float(c.max() - c.min()) <= 0.02 * float(c.max()):
        return block_plan(n, world)   # costs the model cannot tell apart are ties: the block partition (lsr_shard_plan does the same)
    by_cost = np.argsort(-c, kind="stable")
    load = np.zeros(world)
    owner = np.zeros(n, np.int32)
    lists: List[List[int]] = [[] for _ in range(world)]
    for i in by_cost:
        r = int(np.argmin(load))     # first minimum = lowest rank among ties
        owner[i] = r
        load[r] += c[i]
        lists[r].append(int(i))
    first = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    order = np.array([i for x in lists for i in x], np.int32)
    return ShardPlan(owner, order, first)


def registration_cost(n_target: int, n_source: int, passes: float = 30.0) -> float:
    """Work of one NDT candidate in point visits: the target is read a handful of times by the voxel-grid and neighbour-grid
    builds, the source once per derivative pass and once by the fitness search."""
    return 6.0 * float(n_target) + (float(passes) + 4.0) * float(n_source)


def pack_record(T: np.ndarray, score: float, iterations: int, converged: bool, fitness: float = float("nan")) -> np.ndarray:
    rec = np.zeros(RECORD_FLOATS, np.float32)
    rec[:12] = np.asarray(T, np.float32)[:3, :4].reshape(-1)
    rec[12:] = (score, iterations, 1.0 if converged else 0.0, fitness)
    return rec


def unpack_record(rec: np.ndarray) -> dict:
    T = np.eye(4, dtype=np.float32)
    T[:3, :4] = np.asarray(rec[:12], np.float32).reshape(3, 4)
    return dict(T=T, score=float(rec[12]), iterations=int(round(float(rec[13]))), converged=bool(rec[14] > 0.5),
                fitness=float(rec[15]))


def all_gather_records(local: np.ndarray, n_items: int, device=None, plan: ShardPlan | None = None) -> np.ndarray:
    """All-gather the per-rank record blocks into the full (n_items, 16) table, in batch order.  `local` is in the order of
    this rank's share (plan.items(rank), or shard_range without a plan).

    Ranks may own different counts; blocks are padded to the largest shard so a single fixed-size
    all_gather suffices (4 KiB for 64 candidates: latency-bound, ring order irrelevant)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        assert local.shape[0] == n_items
        table = np.zeros((n_items, RECORD_FLOATS), np.float32)
        table[plan.items(0) if plan is not None else slice(None)] = np.asarray(local, np.float32).reshape(n_items, RECORD_FLOATS)
        return table
    world, rank = dist.get_world_size(), dist.get_rank()
    if plan is None:
        plan = block_plan(n_items, world)
    assert plan.world == world and len(plan.order) == n_items
    max_count = max(1, max(len(plan.items(r)) for r in range(world)))
    buf = torch.zeros((max_count, RECORD_FLOATS), dtype=torch.float32)
    mine = plan.items(rank)
    if len(mine):
        buf[: len(mine)] = torch.from_numpy(np.asarray(local, np.float32).reshape(len(mine), RECORD_FLOATS))
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    table = np.zeros((n_items, RECORD_FLOATS), np.float32)
    for r in range(world):
        rr = plan.items(r)
        if len(rr):
            table[rr] = out[r][: len(rr)].cpu().numpy()
    return table


def register_sharded(n_items: int, register_local: Callable[[Sequence[int]], List[np.ndarray]], device=None, costs=None) -> List[dict]:
    """Run `register_local(indices)` on this rank's shard (it returns one packed record per index) and
    return the full list of results on every rank.  With `costs` (one per item, the same on every rank) the shard is the
    longest-first plan instead of the block partition, and `indices` arrive longest first."""
    import torch.distributed as dist

    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    plan = shard_plan(costs, world) if costs is not None else block_plan(n_items, world)
    mine = plan.items(rank)
    recs = register_local(list(mine))
    local = np.stack(recs).astype(np.float32) if len(recs) else np.zeros((0, RECORD_FLOATS), np.float32)
    table = all_gather_records(local, n_items, device=device, plan=plan)
    return [unpack_record(table[i]) for i in range(n_items)]


def broadcast_target(reg, cloud, src: int = 0, device=None):
    """"N keyframes vs. one submap" across ranks: rank `src` holds the target cloud — (n, c>=3) fp32, what
    scanmatcher_component.cpp:307 hands to registration_->setInputTarget — and every rank sets it as the input target of its own
    registration object.  Two torch.distributed broadcasts (shape, then the records: "nccl" = RCCL over xGMI on GPUs, "gloo" in the
    CPU tests); the voxel grid is built redundantly per rank.  Returns the cloud as this rank received it.  Without an initialised
    process group (one rank) the cloud goes straight through."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if reg is not None:
            reg.setInputTarget(cloud)
        return cloud
    rank = dist.get_rank()
    shape = torch.zeros(2, dtype=torch.int64)
    if rank == src:
        a = np.ascontiguousarray(np.asarray(cloud, np.float32))
        shape[0], shape[1] = a.shape[0], a.shape[1]
    if device is not None:
        shape = shape.to(device)
    dist.broadcast(shape, src=src)
    n, c = int(shape[0]), int(shape[1])
    buf = torch.from_numpy(a) if rank == src else torch.empty((n, c), dtype=torch.float32)
    if device is not None:
        buf = buf.to(device)
    dist.broadcast(buf, src=src)
    if reg is not None:
        reg.setInputTarget(buf if device is not None else buf.numpy())
    return buf


def set_input_target_bcast(comm: "Comm", reg, cloud=None, root: int = 0):
    """lsr_set_input_target_bcast of the C ABI: the same through the core's own RCCL communicator (one ncclBroadcast of the records).
    `cloud` is read on rank `root` only (host array or CUDA tensor)."""
    import ctypes as C

    from . import _capi as capi
    from .registration import _cloud_args

    p, stride, n, dev = None, 12, 0, 0
    keep = None
    if comm.rank == root:
        p, stride, n, dev, keep = _cloud_args(cloud, reg)
    capi.check(capi.load().lsr_set_input_target_bcast(comm._h, reg._h, p, stride, n, 1 if dev else 0, root), "lsr_set_input_target_bcast")
    del keep


# ---- the same exchange at the C ABI: lsr_comm_* / lsr_align_batch_sharded (include/lidarslam_reg.h) -------------------
class Comm:
    """One rank's RCCL communicator of the C core.  `unique_id` (128 bytes, from Comm.unique_id() on rank 0, handed to the
    other ranks by any channel: here torch.distributed's store) may be None for a one-rank communicator."""

    def __init__(self, rank: int, world: int, device: int, unique_id: bytes | None = None):
        import ctypes as C

        from . import _capi as capi

        self._lib = capi.load()
        self.rank, self.world = rank, world
        h = C.c_void_p()
        ident = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        capi.check(self._lib.lsr_comm_create(ident, rank, world, device, C.byref(h)), "lsr_comm_create")
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _capi as capi

        ident = (C.c_char * 128)()
        capi.check(capi.load().lsr_comm_unique_id(ident), "lsr_comm_unique_id")
        return bytes(ident)

    def all_gather_records(self, local: np.ndarray) -> np.ndarray:
        """lsr_comm_all_gather_records: `local` = (count, 16) fp32 records of this rank (same count on every rank); returns
        (world, count, 16), rank-major — the pose all-gather on its own."""
        import ctypes as C

        from . import _capi as capi

        a = np.ascontiguousarray(np.asarray(local, np.float32).reshape(-1, RECORD_FLOATS))
        out = np.zeros((self.world, a.shape[0], RECORD_FLOATS), np.float32)
        capi.check(self._lib.lsr_comm_all_gather_records(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0], out.ctypes.data_as(C.c_void_p)),
                   "lsr_comm_all_gather_records")
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lsr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def c_shard_range(n_items: int, world: int, rank: int) -> range:
    import ctypes as C

    from . import _capi as capi

    f, n = C.c_int(), C.c_int()
    capi.load().lsr_shard_range(n_items, world, rank, C.byref(f), C.byref(n))
    return range(f.value, f.value + n.value)


def c_shard_plan(costs, world: int) -> ShardPlan:
    """lsr_shard_plan of the C ABI (device-free)."""
    import ctypes as C

    from . import _capi as capi

    n = len(costs) if costs is not None and not isinstance(costs, int) else int(costs or 0)
    c = np.ascontiguousarray(costs, np.float64) if not isinstance(costs, int) else None
    owner, order, first = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), np.zeros(world + 1, np.int32)
    i32p = C.POINTER(C.c_int32)
    capi.check(capi.load().lsr_shard_plan(n, c.ctypes.data_as(C.POINTER(C.c_double)) if c is not None else None, world,
                                          owner.ctypes.data_as(i32p), order.ctypes.data_as(i32p), first.ctypes.data_as(i32p)), "lsr_shard_plan")
    return ShardPlan(owner[:n], order[:n], first)


def align_batch_sharded(comm: Comm, local_regs, n_items: int, local_guesses=None, with_fitness: bool = True,
                        plan: ShardPlan | None = None) -> List[dict]:
    """lsr_align_batch_sharded / lsr_align_batch_planned: this rank registers `local_regs` (its share of the n_items
    registrations — lsr_shard_range, or plan.items(rank) in that order — targets and sources set) in shared launches; one
    ncclAllGather of 64-byte records; every rank returns all n_items results in batch order."""
    import ctypes as C

    from . import _capi as capi

    lib = capi.load()
    nloc = len(local_regs)
    hs = (C.c_void_p * max(nloc, 1))(*[r._h for r in local_regs])
    g = None
    if local_guesses is not None and nloc:
        g = np.ascontiguousarray(np.stack([np.ascontiguousarray(np.asarray(x, np.float32).T).reshape(16) for x in local_guesses]), np.float32)
    recs = (capi.ShardRecord * n_items)()
    gp = g.ctypes.data_as(C.POINTER(C.c_float)) if g is not None else None
    if plan is None:
        capi.check(lib.lsr_align_batch_sharded(comm._h, hs, nloc, n_items, gp, 1 if with_fitness else 0, recs), "lsr_align_batch_sharded")
    else:
        i32p = C.POINTER(C.c_int32)
        order, first = np.ascontiguousarray(plan.order, np.int32), np.ascontiguousarray(plan.rank_first, np.int32)
        capi.check(lib.lsr_align_batch_planned(comm._h, hs, nloc, n_items, order.ctypes.data_as(i32p), first.ctypes.data_as(i32p), gp,
                                               1 if with_fitness else 0, recs), "lsr_align_batch_planned")
    out = []
    for r in recs:
        T = np.eye(4, dtype=np.float32)
        T[:3, :4] = np.asarray(r.T, np.float32).reshape(3, 4)
        out.append(dict(T=T, score=float(r.score), iterations=int(round(r.iterations)), converged=bool(r.converged > 0.5),
                        fitness=float(r.fitness)))
    return out
