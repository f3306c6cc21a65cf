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


def shard_channels(channels, world, rank):
    """(channel_first, channel_count) of ONE multichannel stream for `rank` of `world`: contiguous,
    balanced channel ranges (SURVEY.md 8e / BASELINE config 5: 8 channels -> one channel per GPU).  A rank
    beyond the channel count gets (0, 0) and sits the job out.  The pair feeds d2d_params.channel_first /
    channel_count: every rank reads the whole interleaved input and converts its own columns."""
    begin, end = shard_range(channels, world, rank)
    if end == begin:
        return 0, 0
    return begin, end - begin


def merge_channel_frames(parts, sample_bytes):
    """Interleave per-rank PCM back into full-width frames.  `parts`: (channel_first, channel_count,
    uint8 array of frames * channel_count * sample_bytes bytes) per rank; the channel ranges must tile
    the stream's channels and all parts hold the same number of frames."""
    import numpy as np
    parts = sorted((p for p in parts if p[1] > 0), key=lambda p: p[0])
    frames = np.asarray(parts[0][2]).size // (parts[0][1] * sample_bytes)
    cols = []
    expect = parts[0][0]
    for first, count, buf in parts:
        if first != expect:
            raise ValueError("channel ranges do not tile")
        a = np.asarray(buf, dtype=np.uint8)
        if a.size != frames * count * sample_bytes:
            raise ValueError("parts disagree on the frame count")
        cols.append(a.reshape(frames, count * sample_bytes))
        expect = first + count
    return np.concatenate(cols, axis=1).reshape(-1)
