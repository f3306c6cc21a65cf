"""1 GPU == N GPUs with REAL engines: two engine processes on the one GPU of the test box share a
batch -- by files (BASELINE config 4's split, /root/reference/src/main.rs:279-300) and by channels of one stream
(config 5's split) -- with rank 0's table blob broadcast over gloo and imported, exactly as bench.py --gpus N
does with RCCL.  The union of the shards must equal a single engine's conversion and the oracle's."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(mode, world, tmp_path, env=None):
    port = _free_port()
    outs = [str(tmp_path / f"{mode}_{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), mode, str(r), str(world), str(port), outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **(env or {}))) for r in range(world)]
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se[-3000:]
    return [np.load(o) for o in outs]


@pytest.mark.timeout(300)
def test_two_engine_processes_share_a_batch_by_files(engine_lib, oracle_mod, tmp_path):
    import shard_worker as W
    from dsd2dxd_amd.shard import shard_range
    parts = _run_ranks("files", 2, tmp_path)
    assert "mfma" in str(parts[0]["kernel"]) or "d2d_fir_mx" in str(parts[0]["kernel"])
    e1 = engine_lib.Engine(n_files=1, kernel=2, **W.KW_FILES)
    for rank, part in enumerate(parts):
        b, e = shard_range(W.N_FILES, 2, rank)
        for i, f in enumerate(range(b, e)):
            buf = W.file_bytes(f)
            want, fr = oracle_mod.Oracle(**W.KW_FILES).translate(buf)
            assert np.array_equal(part["pcm%d" % i], want[:fr * 6]), (rank, f)
            e1.reset()
            single, fr1 = e1.translate(buf)
            assert fr1 == fr and np.array_equal(part["pcm%d" % i], single)
            assert [e1.peak(c) for c in range(2)] == list(part["peaks"][i])


@pytest.mark.timeout(300)
def test_two_engine_processes_split_one_stream_by_channel(engine_lib, oracle_mod, tmp_path):
    import shard_worker as W
    from dsd2dxd_amd.shard import merge_channel_frames, shard_channels
    parts = _run_ranks("channels", 2, tmp_path)
    buf = W.stream_bytes()
    o = oracle_mod.Oracle(**W.KW_CHANNELS)
    want, fr = o.translate(buf)
    merged = merge_channel_frames([(*shard_channels(W.CHN, 2, r), parts[r]["pcm0"]) for r in range(2)], 3)
    assert np.array_equal(merged, want[:fr * W.CHN * 3])
    full = engine_lib.Engine(n_files=1, kernel=2, **W.KW_CHANNELS)
    single, fr1 = full.translate(buf)
    assert fr1 == fr and np.array_equal(merged, single)
    for r in range(2):
        first, count = shard_channels(W.CHN, 2, r)
        assert list(parts[r]["peaks"][0]) == [full.peak(first + c) for c in range(count)] == [o.peak(first + c) for c in range(count)]


@pytest.mark.parametrize("rate", [88200, 96000])
def test_uneven_channel_shards_keep_their_own_tables(engine_lib, oracle_mod, tmp_path, rate):
    """six channels over four ranks = 2, 2, 1, 1.  At 88.2 kHz the two-channel ranks run a pipelined stereo kernel, the one-channel ranks the
    two-group kernel with ANOTHER tap-table variant of the same size: the blob's header names the variant, the import
    is refused, the rank keeps the tables it built.  At 96 kHz one composed polyphase table serves every channel count: every rank adopts
    rank 0's.  Either way the union is the oracle's conversion"""
    import shard_worker as W
    from dsd2dxd_amd.shard import merge_channel_frames, shard_channels
    parts = _run_ranks("channels", 4, tmp_path, env={"D2D_SHARD_RATE": str(rate)})
    counts = [shard_channels(W.CHN, 4, r)[1] for r in range(4)]
    assert counts == [2, 2, 1, 1]
    if rate == 88200:
        assert [bool(p["adopted"]) for p in parts] == [True, True, False, False]
        assert str(parts[0]["kernel"]) != str(parts[2]["kernel"])
    else:
        assert all(bool(p["adopted"]) for p in parts)
        assert all("d2d_fir_px_kernel" in str(p["kernel"]) for p in parts)
    buf = W.stream_bytes()
    want, fr = oracle_mod.Oracle(**dict(W.KW_CHANNELS, output_rate=rate)).translate(buf)
    merged = merge_channel_frames([(*shard_channels(W.CHN, 4, r), parts[r]["pcm0"]) for r in range(4)], 3)
    assert np.array_equal(merged, want[:fr * W.CHN * 3])


@pytest.mark.timeout(300)
def test_rccl_path_with_one_rank(engine_lib, oracle_mod, tmp_path):
    """the collective path of the multi-GPU bench on the one GPU of the test box: a fresh process joins a one-rank process group over RCCL
    (backend "nccl", device_id), the table blob is exported, broadcast as a DEVICE tensor, imported, and the conversion that follows is the
    oracle's; then bench.py --force-dist runs its own broadcast / barrier / MAX-reduce lines once (no scaling number is claimed from it)"""
    import json
    import shard_worker as W
    parts = _run_ranks("files", 1, tmp_path, env={"D2D_SHARD_BACKEND": "nccl", "D2D_SHARD_IMPORT_OWN": "1"})
    assert bool(parts[0]["adopted"])
    for f in range(W.N_FILES):
        want, fr = oracle_mod.Oracle(**W.KW_FILES).translate(W.file_bytes(f))
        assert np.array_equal(parts[0]["pcm%d" % f], want[:fr * 6]), f
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--files", "2", "--seconds", "2", "--steps", "2", "--warmup", "1",
                        "--reps", "1", "--sustain", "0", "--no-cpu-baseline", "--no-pcie"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["collectives"]["backend"] == "nccl" and line["config"]["collectives"]["table_blob_bytes"] > 0
    assert line["value"] > 0
