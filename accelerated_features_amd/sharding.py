"""Frame sharding for multi-GPU runs.

Frames (and frame pairs) are independent and the model is 6 MB, so every GPU holds a replica
and processes a contiguous chunk of the work list: no data-path collective exists.  One
process per GPU; torch.distributed (RCCL) is used only for the start/stop barrier and the
max-over-ranks timing of the benchmark.
"""


def shard_range(n_items, rank, world, multiple=1):
    """Contiguous [begin, end) of `n_items` work units for `rank` of `world`.

    Units are spread as evenly as possible (sizes differ by at most `multiple`); `multiple`
    keeps chunk boundaries on multiples of that many units (2 for frame pairs)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if n_items % multiple:
        raise ValueError(f"{n_items} items is not a multiple of {multiple}")
    groups = n_items // multiple
    base, extra = divmod(groups, world)
    begin = rank * base + min(rank, extra)
    end = begin + base + (1 if rank < extra else 0)
    return begin * multiple, end * multiple
