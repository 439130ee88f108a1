"""Multi-GPU sharding of a batch of independent QPs.

QP instances do not interact, so the partition is a contiguous slice of the
batch per rank and the iteration needs NO collective: each rank runs the 1-GPU
path on its slice.  RCCL (torch.distributed backend "nccl") is used only for
the optional global stop decision (one MAX all-reduce of two doubles per check)
and for gathering per-QP results; both are off the hot loop.
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
