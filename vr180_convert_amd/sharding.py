"""Multi-GPU partitioning of the hot path: one process per GPU, no data-path collective.

Every (frame, eye) is an independent unit (reference remapper.py:388-398 loops over them
serially); ranks take a contiguous, balanced block of FRAMES so that both eyes of a frame -- which
end up in one side-by-side output buffer -- stay on one GPU.  SURVEY.md 8e.
"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world_size: int) -> range:
    """Contiguous balanced block of ``range(n_items)`` owned by ``rank`` (first ranks get the
    remainder).  Blocks of all ranks partition ``range(n_items)`` exactly."""
    if world_size < 1 or not 0 <= rank < world_size or n_items < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_sizes(n_items: int, world_size: int) -> list[int]:
    return [len(shard_range(n_items, r, world_size)) for r in range(world_size)]
