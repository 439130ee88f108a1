"""Multi-GPU sharding of a batch of independent QPs.

QP instances do not interact, so the partition is a contiguous slice of the
batch per rank and the iteration needs NO collective: each rank runs the 1-GPU
path on its slice.  RCCL (torch.distributed backend "nccl") is used only for
the global stop / adaptive-rho decision of solve_sharded (ONE all-reduce, SUM,
of three doubles per checked iteration: QPs still unconverged, sum r^2, sum
s^2) and for gathering per-QP results; both are off the hot loop.
"""
from __future__ import annotations

from typing import Tuple

from .problems import Problem


def shard_bounds(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `rank`; the first `batch % world_size` ranks
    get one extra QP.  Empty slices are allowed (start == stop)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, extra = divmod(batch, world_size)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    return start, stop


def shard_problem(problem: Problem, world_size: int, rank: int) -> Problem:
    start, stop = shard_bounds(problem.batch, world_size, rank)
    return problem.slice(start, stop)


def gather_batch(local, batch: int, group=None):
    """All-gather per-QP rows (first axis = this rank's shard) into the full
    batch on every rank.  Works on CPU tensors over gloo and on GPU tensors over
    RCCL (backend "nccl").  Shards may differ in size by one QP (shard_bounds),
    so rows are padded to the largest shard for the collective and trimmed after.
    Off the hot loop: used once per solve to collect results."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_bounds(batch, world, r) for r in range(world)]
    max_rows = max(b - a for a, b in sizes)
    t = torch.as_tensor(local)
    pad = torch.zeros((max_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, sizes)], dim=0)


def global_residual_max(r_max: float, s_max: float, device="cpu", group=None):
    """Helper for callers that only want the worst residual pair of the whole batch (reporting):
    one MAX all-reduce of two doubles.  solve_sharded does NOT use it -- its stop decision is the
    3-double SUM described there."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([r_max, s_max], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0]), float(t[1])


def solve_sharded(solver, global_batch: int, group=None, device="cpu"):
    """Multi-GPU admm_solve: every rank holds a Solver over its shard (shard_problem) and calls
    this collectively.  The iterations need no communication; at every checked iteration ONE
    all-reduce (SUM) of three doubles -- unconverged QPs, sum r^2, sum s^2 -- makes the stop
    decision and the adaptive-rho decision global, so every QP runs exactly the iterations it would
    run in an unsharded solve of the whole batch (with adaptive rho: up to the summation order of
    R and S).  `device` = where the 3-double tensor lives ("cpu" for gloo, "cuda:i" for RCCL).
    Returns this rank's SolveInfo (per-QP arrays are the shard's; use gather_batch to assemble)."""
    import torch
    import torch.distributed as dist
    opt = solver.options
    adaptive = opt.adapt_interval > 0
    solver.solve_begin()
    while True:
        it, nconv, R, S = solver.solve_step(sums=adaptive)
        t = torch.tensor([float(solver.batch - nconv), R, S], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        left, Rg, Sg = float(t[0]), float(t[1]), float(t[2])
        if left == 0 or it >= opt.max_iter:
            break
        if adaptive:
            solver.solve_adapt(Rg, Sg)
    return solver.solve_end()


# ---------------------------------------------------------------------------------------------------------------------
# Time sharding (include/admm_hip.h ABI v7): ONE batch whose horizon is cut into segments that live on
# different ranks -- the "shooting segments" of BASELINE.json's north_star.  The segment algebra couples them exactly
# through 2 n numbers per segment and QP, completed by an all-gather before every segment scan: the one place where RCCL
# carries data in this library.
# ---------------------------------------------------------------------------------------------------------------------

class _DevView:
    """A device pointer as something torch can wrap without copying (__cuda_array_interface__)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def device_tensor(ptr: int, count: int, device):
    """fp64 torch tensor over `count` doubles of device memory at `ptr` (no copy; the memory belongs to the handle)."""
    import torch
    return torch.as_tensor(_DevView(ptr, count), device=device)


def make_exchange(group=None, device="cuda:0"):
    """The admm_exchange_fn of a time-sharded handle over torch.distributed: RCCL (backend "nccl") enqueues the all-gather
    behind the library's own HIP stream, so nothing waits on the host; any other backend (gloo: several ranks on one GPU, CPU
    transport) synchronises the stream and moves the slices through host memory.  Returns (ctypes callback, stats dict)."""
    import torch
    import torch.distributed as dist
    from . import _abi
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nccl = dist.get_backend(group) == "nccl"
    stats = {"calls": 0, "doubles": 0}

    def exchange(ctx, stream, op, buf, count):
        try:
            if op != _abi.EXCHANGE_ALLGATHER:
                return 2
            full = device_tensor(buf, count * world, device)
            mine = full[rank * count:(rank + 1) * count]
            ext = torch.cuda.ExternalStream(stream, device=device)
            stats["calls"] += 1
            stats["doubles"] += int(count) * world
            with torch.cuda.stream(ext):
                if nccl:
                    dist.all_gather_into_tensor(full, mine, group=group)        # in place: `mine` is its own slice of `full`
                else:
                    ext.synchronize()
                    host = mine.cpu()
                    outs = [torch.empty_like(host) for _ in range(world)]
                    dist.all_gather(outs, host, group=group)
                    for r, t in enumerate(outs):
                        if r != rank:
                            full[r * count:(r + 1) * count].copy_(t)
                    ext.synchronize()
            return 0
        except Exception:           # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    return _abi.EXCHANGE_FN(exchange), stats


class TimeShardedSolver:
    """One batch of QPs, the horizon's segments spread over the ranks of a torch.distributed group (one process per GPU).
    Every rank passes the GLOBAL problem; iterate / run / solve are collective calls (same arguments on every rank) and
    return the same residuals, iteration counts and stop decisions everywhere; get() assembles the full vectors from the
    ranks' windows."""

    def __init__(self, problem: Problem, options=None, group=None, device="cuda:0"):
        import torch.distributed as dist
        from .solver import Solver
        self.group, self.device = group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._fn, self.exchange_stats = make_exchange(group, device)
        self.solver = Solver(problem, options, timeshard=(self.rank, self.world, self._fn))
        self.window = self.solver.window()

    def close(self):
        self.solver.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def iterate(self, iters: int):
        self.solver.iterate(iters)

    def run(self, iters: int, residual_every: int = 0):
        self.solver.run(iters, residual_every)

    def solve(self):
        return self.solver.solve()

    def residuals(self):
        return self.solver.residuals()

    def get(self):
        """(w, z, y) of the whole horizon on every rank: each rank contributes the rows of its own stages."""
        import numpy as np
        import torch.distributed as dist
        nb = self.solver.problem.nb
        lo, hi = self.window["stage_lo"] * nb, self.window["stage_hi"] * nb
        local = [a[:, lo:hi].copy() for a in self.solver.get()]
        parts = [None] * self.world
        dist.all_gather_object(parts, local, group=self.group)
        return tuple(np.concatenate([p[i] for p in parts], axis=1) for i in range(3))
