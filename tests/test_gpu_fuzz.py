"""Seeded random configurations: every (rate, filter, depth, dither, layout, channel count, call
pattern) the engine accepts must match the oracle bit for bit, with both kernels."""
import numpy as np
import pytest

from helpers import pack_layout, random_bytes

pytestmark = pytest.mark.gpu

RATES = [(1, 88200), (1, 176400), (1, 352800), (2, 88200), (2, 176400), (2, 352800), (2, 705600), (4, 88200), (4, 352800),
         (4, 1411200), (8, 352800), (1, 96000), (1, 192000), (1, 384000), (2, 96000), (4, 192000), (8, 96000)]


def _case(seed):
    rng = np.random.default_rng(seed)
    dsd_rate, out_rate = RATES[rng.integers(len(RATES))]
    filt = "E"
    if out_rate % 44100 == 0 and out_rate <= 352800:
        if dsd_rate == 1:
            filt = rng.choice(["E", "X"] + (["D"] if out_rate == 352800 else []))
        elif dsd_rate == 2:
            filt = rng.choice(["E", "C"])
    bits = int(rng.choice([16, 20, 24, 32]))
    dither = str(rng.choice(["T", "R", "X", "F"]))
    channels = int(rng.choice([1, 2, 2, 2, 3, 5, 6, 8]))
    fmt = str(rng.choice(["P", "P", "I"]))
    block = int(rng.choice([4096, 4096, 1024, 256, 48, 7])) if fmt == "P" else 1
    endian = str(rng.choice(["L", "M"]))
    level = float(rng.choice([0.0, 0.0, -3.0, 4.0, -0.5]))
    if out_rate % 44100 == 0 and rng.integers(5) == 0:     # drawn last so the other parameters of a seed stay what they were
        dither = "N"                                        # the noise-shaped extension (44.1 kHz family only)
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=channels, fmt=fmt, endianness=endian, block_size=block,
              filter=str(filt), bit_depth=bits, dither=dither, seed=int(rng.integers(1 << 30)), level_db=level)
    total = int(rng.integers(2000, 30000))
    ncuts = int(rng.integers(1, 6))
    cuts = sorted(set([0, total] + [int(x) for x in rng.integers(0, total, ncuts)]))
    if rng.random() < 0.3:
        cuts.insert(1, cuts[1])            # a zero-length call
    return kw, total, cuts


import os
_EXTRA = int(os.environ.get("D2D_FUZZ_EXTRA", "0"))      # a longer one-off run: D2D_FUZZ_EXTRA=400 adds that many seeds to each family


@pytest.mark.parametrize("seed", list(range(40 + _EXTRA)) + [1000 + i for i in range(16 + _EXTRA)])
def test_random_configuration(engine_lib, oracle_mod, seed):
    kw, total, cuts = _case(seed)
    if seed >= 1000:                                          # the second family of seeds: the 32-bit tap grid where it is defined
        if kw["output_rate"] % 44100 or kw["dither"] == "N":
            pytest.skip("32-bit taps: 44.1k family, dither T/R/F/X")
        kw["tap_bits"] = 32
    chans = [random_bytes(total, 1000 * seed + c) for c in range(kw["channels"])]
    bufs = [pack_layout([ch[a:b] for ch in chans], kw["fmt"], kw["block_size"]) if b > a else np.zeros(0, np.uint8)
            for a, b in zip(cuts[:-1], cuts[1:])]
    o = oracle_mod.Oracle(**kw)
    want = []
    for b in bufs:
        r, rf = o.translate(b)
        want.append(r[:rf * o.frame_bytes].copy())
    want = np.concatenate(want) if want else np.zeros(0, np.uint8)
    for kernel in (1, 2):
        e = engine_lib.Engine(kernel=kernel, **kw)
        got = np.concatenate([e.translate(b)[0] for b in bufs])
        assert np.array_equal(got, want), (kernel, kw)
        assert e.peak_dbfs() == o.peak_dbfs() or (np.isinf(e.peak_dbfs()) and np.isinf(o.peak_dbfs()))


@pytest.mark.parametrize("seed", [2000 + i for i in range(24 + _EXTRA // 10)])
def test_random_long_streams(engine_lib, oracle_mod, seed):
    """the third family: streams long enough for every wave of the pipelined kernels to walk several tiles in its fixed-order loop (the first
    two families' calls are a tile or two: careful tiles mostly) -- mono pairs, stereo, 5.1 whole frames, the one-pass 32-bit taps, the composed
    48k kernels, planar power-of-two blocks or byte-interleaved, two or three ragged calls"""
    rng = np.random.default_rng(seed)
    dsd_rate, out_rate = [(1, 88200), (2, 88200), (4, 176400), (4, 88200), (1, 352800), (1, 176400), (1, 96000), (1, 192000), (2, 384000)][rng.integers(9)]
    channels = int(rng.choice([1, 2, 2, 6]))
    fmt = str(rng.choice(["P", "P", "I"]))
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=channels, fmt=fmt, endianness=str(rng.choice(["L", "M"])), block_size=4096 if fmt == "P" else 1,
              filter="E", bit_depth=int(rng.choice([16, 24, 24, 32])), dither=str(rng.choice(["T", "R", "X"])), seed=int(rng.integers(1 << 30)),
              level_db=float(rng.choice([0.0, 0.0, 0.0, -3.0])))
    if out_rate % 44100 == 0 and rng.integers(4) == 0:
        kw["tap_bits"] = 32
    total = 4096 * int(rng.integers(40, 150)) * dsd_rate
    c1 = 4096 * int(rng.integers(1, 30)) * dsd_rate
    cuts = [0, c1, total - (0 if fmt == "P" else int(rng.integers(0, 500))), total]
    chans = [random_bytes(total, 77 * seed + c) if c % 2 else np.where(np.arange(total) % 9 < 5, 0xA5, 0x5A).astype(np.uint8) ^ random_bytes(total, 78 * seed + c) // 64
             for c in range(channels)]
    o = oracle_mod.Oracle(**kw)
    e = engine_lib.Engine(kernel=2, **kw)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b <= a:
            continue
        buf = pack_layout([ch[a:b] for ch in chans], fmt, kw["block_size"])
        g, gf = e.translate(buf)
        w, wf = o.translate(buf)
        assert gf == wf and np.array_equal(g, w[:wf * o.frame_bytes]), (kw, a, b, e.kernel_name())
    assert [e.peak(c) for c in range(channels)] == [o.peak(c) for c in range(channels)]
