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
