"""Ray sharding for multi-GPU runs: rays are independent units, so a batch is cut
into contiguous slices, one per rank; the scene is replicated and there is no data-path collective.
The only exchange is a sum of the batch counters (hits, ...)."""
from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of n_total rays owned by `rank`; slices tile the batch exactly and
    differ in length by at most one ray."""
    if world < 1 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard arguments")
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    return lo, hi


def reduce_counters(counters, dist=None):
    """Sum a counters tensor over all ranks (RCCL on GPU, gloo on CPU); in place, returns it."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counters)
    return counters
