"""The N>1 path on CPU: two gloo ranks shard a batch of files, adopt rank 0's table blob and reduce
their timings/counts the way bench.py does with RCCL.  The conversion itself is the oracle here
(this container has no GPU); what is covered is the host-side protocol."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from dsd2dxd_amd.shard import shard_by_bytes, shard_range
    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            got = [shard_range(total, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(got[:-1], got[1:]))
            sizes = [e - b for b, e in got]
            assert max(sizes) - min(sizes) <= 1
    parts = shard_by_bytes([10, 1, 1, 1, 9, 8, 2], 3)
    assert sorted(i for p in parts for i in p) == list(range(7))
    loads = [sum([10, 1, 1, 1, 9, 8, 2][i] for i in p) for p in parts]
    assert max(loads) <= 12


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dsd2dxd_amd.shard import shard_range
    from helpers import pack_layout, synth
    from oracle import oracle as O
    n_files, nbytes = 5, 4096 * 4
    kw = dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E",
              bit_depth=24, dither="T", seed=206)
    # "tables": rank 0 owns the blob, the others adopt it (bench.py: d2d_tables_export/import + RCCL broadcast)
    blob = torch.zeros(64, dtype=torch.uint8)
    if rank == 0:
        blob = torch.arange(64, dtype=torch.uint8)
    dist.broadcast(blob, src=0)
    assert blob.tolist() == list(range(64))
    b, e = shard_range(n_files, world, rank)
    sums = []
    for f in range(b, e):
        buf = pack_layout([synth("sine", nbytes, seed=10 + f), synth("pink", nbytes, seed=20 + f, amp=0.098)], "P", 4096)
        pcm, fr = O.Oracle(**kw).translate(buf)
        sums.append((f, int(pcm.astype(np.uint64).sum()), fr))
    cnt = torch.tensor([sum(s[2] for s in sums)], dtype=torch.int64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, sums)
    if rank == 0:
        q.put((int(cnt.item()), float(t.item()), gathered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sharded_batch_equals_single_rank():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import pack_layout, synth
    from oracle import oracle as O
    O.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, tmax, gathered = q.get(timeout=100)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    n_files, nbytes = 5, 4096 * 4
    kw = dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E",
              bit_depth=24, dither="T", seed=206)
    ref = []
    for f in range(n_files):
        buf = pack_layout([synth("sine", nbytes, seed=10 + f), synth("pink", nbytes, seed=20 + f, amp=0.098)], "P", 4096)
        pcm, fr = O.Oracle(**kw).translate(buf)
        ref.append((f, int(pcm.astype(np.uint64).sum()), fr))
    got = sorted(x for part in gathered for x in part)
    assert got == ref                               # 1 GPU == N GPUs: same files, same bytes
    assert total == sum(r[2] for r in ref)
    assert abs(tmax - 0.2) < 1e-12                  # MAX over ranks, as bench.py times a step


def test_shard_channels_tiles_and_merge_restores_frames():
    from dsd2dxd_amd.shard import merge_channel_frames, shard_channels
    for channels in (1, 2, 5, 6, 8):
        for world in (1, 2, 3, 8, 10):
            got = [shard_channels(channels, world, r) for r in range(world)]
            used = [(f, c) for f, c in got if c]
            assert sum(c for _, c in used) == channels and used[0][0] == 0
            assert all(a[0] + a[1] == b[0] for a, b in zip(used[:-1], used[1:]))
            assert max(c for _, c in used) - min(c for _, c in used) <= 1
    rng = np.random.default_rng(1)
    frames, chn, sb = 37, 5, 3
    full = rng.integers(0, 256, size=(frames, chn, sb), dtype=np.uint8)
    parts = [(f, c, full[:, f:f + c, :].reshape(-1).copy()) for f, c in (shard_channels(chn, 3, r) for r in range(3))]
    assert np.array_equal(merge_channel_frames(parts[::-1], sb), full.reshape(-1))
    with pytest.raises(ValueError):
        merge_channel_frames([parts[0], parts[2]], sb)


def _channel_worker(rank, world, port, q):
    """BASELINE config 5's split: ONE multichannel stream, a channel range per rank.  The conversion is the
    oracle's (no GPU here) restricted to the rank's columns -- what a subset engine emits, see
    tests/test_gpu_api.py::test_channel_subsets_reproduce_the_full_conversion."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dsd2dxd_amd.shard import merge_channel_frames, shard_channels
    from helpers import pack_layout, random_bytes
    from oracle import oracle as O
    chn, nbytes = 5, 4096 * 2 + 77
    kw = dict(dsd_rate=1, output_rate=96000, channels=chn, fmt="I", endianness="M", block_size=1, filter="E", bit_depth=24, dither="T", seed=9)
    buf = pack_layout([random_bytes(nbytes, 40 + c) for c in range(chn)], "I", 1)     # every rank reads the whole stream
    pcm, fr = O.Oracle(**kw).translate(buf)
    first, count = shard_channels(chn, world, rank)
    mine = pcm[:fr * chn * 3].reshape(fr, chn, 3)[:, first:first + count, :].reshape(-1).copy()
    gathered = [None] * world
    dist.all_gather_object(gathered, (first, count, mine))
    if rank == 0:
        q.put((np.array_equal(merge_channel_frames(gathered, 3), pcm[:fr * chn * 3]), [g[:2] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_split_one_stream_by_channel():
    from oracle import oracle as O
    O.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_channel_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, ranges = q.get(timeout=100)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert same and ranges == [(0, 3), (3, 2)]


def _bench(*args, **env):
    import subprocess
    e = dict(os.environ, D2D_BENCH_STUB="1", D2D_BENCH_TIMEOUT="120", **env)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True, timeout=300)


def test_bench_starts_its_ranks_and_prints_rank0s_line():
    """bench.py --gpus 2 with no launcher: the parent starts two ranks (here the stub step over gloo: no GPU), prints rank 0's JSON
    line and nothing else, exits 0"""
    import json
    r = _bench("--gpus", "2", "--steps", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["ms_per_step"] >= 2.0      # MAX over ranks: rank 1 sleeps 2 ms per step


def test_bench_ends_the_run_when_a_rank_dies_early():
    """a rank that dies before the rendezvous (bad device ordinal, refused table import ...) must not leave rank 0 waiting in
    init_process_group: the parent terminates the others, repeats the failed rank's stderr and exits with its code"""
    import time
    t0 = time.monotonic()
    r = _bench("--gpus", "2", "--steps", "3", D2D_BENCH_STUB_FAIL_RANK="1")
    assert r.returncode == 7
    assert time.monotonic() - t0 < 60
    assert "rank 1" in r.stderr and "failing on request" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
