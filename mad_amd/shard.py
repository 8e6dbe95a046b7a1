"""Work partitioning across the GPUs of one node: independent images, no data-path collective.

Mirrors the split of ``/root/reference/generate_translation.py:100-118``: ``len // n`` items per
worker, contiguous, the remainder to the last worker, and the output index offset
``rank * (len // n)``.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple, TypeVar

T = TypeVar("T")


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world size {world_size}")
    per = n_items // world_size
    lo = rank * per
    hi = (rank + 1) * per if rank < world_size - 1 else n_items
    return lo, hi


def shard_list(items: Sequence[T], world_size: int, rank: int) -> Tuple[List[T], int]:
    """-> (this rank's items, index offset of its first item)."""
    lo, hi = shard_bounds(len(items), world_size, rank)
    return list(items[lo:hi]), lo


def job_throughput(local_items: int, local_seconds: float, device=None):
    """Whole-job items/s for one-process-per-GPU runs: items summed over ranks divided by the slowest
    rank's time.  Uses the default process group when one is initialised (RCCL on GPUs, gloo on CPU);
    a single process needs none.  Returns (items_total, seconds_max, items_per_second)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
        n = torch.tensor([float(local_items)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        local_seconds, local_items = float(t.item()), int(n.item())
    return local_items, local_seconds, local_items / local_seconds
