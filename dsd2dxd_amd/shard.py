"""How a batch of independent DSD files is split over ranks (one process per GPU).

Files never interact -- the FIR output depends only on a channel's own bit history and the dither
generator is keyed per (seed, channel, index) -- so the split needs no data-path collective.  This
is the reference's Rayon `into_par_iter` over files (/root/reference/src/main.rs:280-300) with a
rank in place of a worker thread.
"""


def shard_range(total, world, rank):
    """Contiguous, balanced [begin, end) of `total` items for `rank` of `world` (sizes differ by <= 1)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_by_bytes(sizes, world):
    """Greedy longest-first assignment for files of unequal length: returns a list of index lists,
    one per rank, minimising the largest byte total (LPT rule)."""
    order = sorted(range(len(sizes)), key=lambda i: -sizes[i])
    loads = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: loads[k])
        out[r].append(i)
        loads[r] += sizes[i]
    for lst in out:
        lst.sort()
    return out
