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
