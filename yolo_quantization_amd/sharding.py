"""Image sharding of the multi-GPU path: independent images, contiguous equal shards, no data-path collective
(SURVEY.md 8e).  bench.py uses weak scaling (a fixed per-rank batch); a fixed global batch splits as below."""


def shard_range(total_images: int, rank: int, world: int) -> tuple[int, int]:
    """[start, stop) of the images owned by `rank`: sizes differ by at most one, union is [0, total), disjoint."""
    base, extra = divmod(total_images, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)
