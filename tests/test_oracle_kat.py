"""CPU known-answer tests that pin the oracle (oracle/d2d_oracle.c) from first principles.

The reference holds no golden vectors for this path (parity unpinned), so the oracle
is pinned by what can be derived without it: the fixtures' measured properties (bit order, tone
level), the impulse response against the committed tap tables, exact invariances of the arithmetic,
and the statistics of the dither.  Tests that read /root/reference/test/* skip when that directory
is absent (it never travels to the GPU box).
"""
import os

import numpy as np
import pytest

from helpers import BITREV, decode_pcm, pack_layout, random_bytes, synth

REF = "/root/reference/test"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference fixtures not present")


def band_residual_db(y, fs, f0=1000.0, lo=2000.0, hi=20000.0):
    """power between lo and hi relative to the f0 tone, in dB"""
    y = y - y.mean()
    w = np.hanning(len(y))
    Y = np.abs(np.fft.rfft(y * w)) ** 2
    f = np.fft.rfftfreq(len(y), 1.0 / fs)
    tone = Y[(f > f0 - 200) & (f < f0 + 200)].sum()
    band = Y[(f >= lo) & (f <= hi)].sum()
    return 10 * np.log10(band / tone)


def tone_fit(y, fs, f0=1000.0):
    n = np.arange(len(y))
    w = 2 * np.pi * f0 / fs
    A = np.stack([np.sin(w * n), np.cos(w * n), np.ones_like(n, dtype=float)], 1)
    c = np.linalg.lstsq(A, y, rcond=None)[0]
    return float(np.hypot(c[0], c[1])), float(c[2])


@needs_ref
def test_bit_order_discriminator_lsb_fixture(oracle_mod):
    """1kHz_mono_p.dsd is LSB-first (DSF payload): right order -> clean 2-20 kHz band, wrong -> ~-36 dB"""
    raw = np.fromfile(os.path.join(REF, "1kHz_mono_p.dsd"), dtype=np.uint8)[:4096 * 200]
    res = {}
    for e in "LM":
        o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=1, fmt="P", endianness=e, bit_depth=32, dither="X")
        _, fr, f = o.translate(raw, want_f64=True)
        res[e] = band_residual_db(f[2000:, 0], 88200.0)
    assert res["L"] < -80.0
    assert res["M"] > -45.0


@needs_ref
def test_bit_order_discriminator_msb_fixture(oracle_mod):
    """1kHz_stereo_i.dsd is byte-interleaved, MSB-first (DFF payload); L == R bit-identically"""
    raw = np.fromfile(os.path.join(REF, "1kHz_stereo_i.dsd"), dtype=np.uint8)[:2 * 4096 * 100]
    res = {}
    for e in "LM":
        o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=2, fmt="I", endianness=e, bit_depth=32, dither="X")
        _, fr, f = o.translate(raw, want_f64=True)
        assert np.array_equal(f[:, 0], f[:, 1])
        res[e] = band_residual_db(f[2000:, 0], 88200.0)
    assert res["M"] < -80.0
    assert res["L"] > -45.0


@needs_ref
@pytest.mark.parametrize("name,dsd_rate,channels,amp_db", [
    ("1kHz_mono_p.dsd", 1, 1, -9.23), ("1kHz_mono_128.dsf", 2, 1, -9.07), ("1kHz_stereo_128.dsf", 2, 2, -9.07)])
def test_tone_level_matches_fixture_measurement(oracle_mod, name, dsd_rate, channels, amp_db):
    """unity DC gain: the 1 kHz tone comes out at the level measured in the DSD stream"""
    raw = np.fromfile(os.path.join(REF, name), dtype=np.uint8)
    if name.endswith(".dsf"):
        raw = raw[92:]                                   # DSF: data chunk payload starts at byte 92
    raw = raw[:channels * 4096 * 120]
    o = oracle_mod.Oracle(dsd_rate=dsd_rate, output_rate=88200, channels=channels, fmt="P", endianness="L",
                          block_size=4096, bit_depth=32, dither="X")
    _, fr, f = o.translate(raw, want_f64=True)
    for c in range(channels):
        amp, dc = tone_fit(f[2000:, c], 88200.0)
        assert abs(20 * np.log10(amp) - amp_db) < 0.1
        assert abs(dc) < 2e-3
    if channels == 2:
        assert np.array_equal(f[:, 0], f[:, 1])          # the stereo fixtures are dual-mono


@pytest.mark.parametrize("filt,dsd_rate,out_rate", [("E", 1, 88200), ("E", 1, 352800), ("D", 1, 352800), ("X", 1, 176400),
                                                     ("C", 2, 88200), ("E", 4, 88200)])
@pytest.mark.parametrize("endian", ["M", "L"])
def test_impulse_recovers_taps(oracle_mod, filt, dsd_rate, out_rate, endian):
    """impulse_mono_toggle.dsd idea: idle 0xAA with ONE flipped bit -> (response - idle response) = 2*h at
    the phase the bit lands on; must equal the committed taps exactly (dyadic taps: exact in f64)."""
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=1, fmt="P", endianness=endian, bit_depth=32,
              dither="X", filter=filt)
    nbytes = 4096 * 4
    idle = np.full(nbytes, 0xAA, dtype=np.uint8)
    hit = idle.copy()
    byte, bit = 8192, 3
    hit[byte] ^= (1 << bit)
    o1, o2 = oracle_mod.Oracle(**kw), oracle_mod.Oracle(**kw)
    _, _, f1 = o1.translate(idle, want_f64=True)
    _, _, f2 = o2.translate(hit, want_f64=True)
    info = o1.info()
    M, N = info["M"], info["ntaps"]
    taps = o1.taps()
    assert abs(taps.sum() - 1.0) == 0.0 and np.array_equal(taps, taps[::-1])
    t = byte * 8 + (7 - bit if endian == "M" else bit)           # time index of the flipped bit
    was_one = bool((0xAA >> bit) & 1)
    diff = f2[:, 0] - f1[:, 0]
    expect = np.zeros_like(diff)
    for n in range(len(diff)):
        j = t - ((n + 1) * M - N)                                # y[n] = sum_j h[j] s[(n+1)M - N + j]
        if 0 <= j < N:
            expect[n] = (-2.0 if was_one else 2.0) * taps[j]
    assert np.array_equal(diff, expect)


def test_direct_form_equals_lut_form(oracle_mod):
    raw = pack_layout([random_bytes(4096 * 3, 1), random_bytes(4096 * 3, 2)], "P", 4096)
    for out_rate in (88200, 352800, 96000):
        outs = []
        for mode in (0, 1):
            o = oracle_mod.Oracle(dsd_rate=1, output_rate=out_rate, channels=2, fmt="P", endianness="L", bit_depth=24,
                                  dither="T", seed=3, fir_mode=mode)
            outs.append(o.translate(raw)[0])
        assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("out_rate", [88200, 96000])
def test_layout_and_bit_order_invariances(oracle_mod, out_rate):
    """planar(any block) == interleaved of the same bits; MSB file == bit-reversed LSB file;
    streaming in ragged calls == one shot"""
    nbytes = 4096 * 5
    chans = [random_bytes(nbytes, 7), random_bytes(nbytes, 8)]
    base_kw = dict(dsd_rate=1, output_rate=out_rate, channels=2, bit_depth=24, dither="T", seed=1)

    def run(fmt, endian, block, chans_, cuts=None):
        o = oracle_mod.Oracle(fmt=fmt, endianness=endian, block_size=block, **base_kw)
        cuts = cuts or [0, nbytes]
        out = [o.translate(pack_layout([c[a:b] for c in chans_], fmt, block))[0] for a, b in zip(cuts[:-1], cuts[1:])]
        return np.concatenate(out), o
    ref, o_ref = run("P", "L", 4096, chans)
    for fmt, block in (("P", 512), ("P", 100), ("I", 1), ("P", 4096 * 5)):
        assert np.array_equal(run(fmt, "L", block, chans)[0], ref)
    rev = [BITREV[c] for c in chans]
    assert np.array_equal(run("P", "M", 4096, rev)[0], ref)
    out, o2 = run("P", "L", 4096, chans, cuts=[0, 1, 4096, 4099, 9000, 9000, nbytes])
    assert np.array_equal(out, ref)
    assert o2.peak(0) == o_ref.peak(0) and o2.peak_dbfs() == o_ref.peak_dbfs()


def test_output_length_rule(oracle_mod):
    """frames = floor(total bits / M) for 44.1k multiples; ceil(n_x * L / 147) for the 48k cascade"""
    for out_rate, M in ((88200, 32), (176400, 16), (352800, 8)):
        o = oracle_mod.Oracle(dsd_rate=1, output_rate=out_rate, channels=1, fmt="P", endianness="L", bit_depth=24)
        total = 0
        frames = 0
        for n in (1, 3, 4096, 77, 0, 5000):
            frames += o.translate(random_bytes(n, n))[1]
            total += n
            assert frames == total * 8 // M
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=96000, channels=1, fmt="P", endianness="L", bit_depth=24)
    _, fr = o.translate(random_bytes(4096 * 2, 5))
    nx = 4096 * 2                       # stage A decimates by 8: one sample per byte
    assert fr == -(-nx * 40 // 147)


def test_known_levels(oracle_mod):
    """all ones -> +1 exactly (taps sum to 1), clipped to 2^23-1; all zeros -> -2^23; -6.0206 dB halves it"""
    ones = np.full(4096 * 2, 0xFF, dtype=np.uint8)
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=1, fmt="P", endianness="M", bit_depth=24, dither="X")
    pcm, fr, f = o.translate(ones, want_f64=True)
    assert (f[100:, 0] == 1.0).all()
    assert (decode_pcm(pcm, 24, 1)[100:, 0] == 8388607).all()
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=1, fmt="P", endianness="M", bit_depth=16, dither="X",
                          level_db=20 * np.log10(0.5))
    pcm, fr = o.translate(np.zeros(4096 * 2, dtype=np.uint8))
    assert (decode_pcm(pcm, 16, 1)[100:, 0] == -16384).all()
    # 20 bit rides in 24: value << 4
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=1, fmt="P", endianness="M", bit_depth=20, dither="X")
    pcm, fr = o.translate(ones)
    assert (decode_pcm(pcm, 24, 1)[100:, 0] == (524287 << 4)).all()


def test_dither_statistics_and_determinism(oracle_mod):
    """TPDF: error mean 0, variance 1/6 (+1/12 rounding) LSB^2, triangular; rectangular: 1/12 (+1/12);
    a fixed seed reproduces the bytes, another seed does not"""
    nbytes = 4096 * 64
    raw = synth("sine", nbytes, seed=9, amp=0.3, freq=997.0)
    base = dict(dsd_rate=1, output_rate=352800, channels=1, fmt="P", endianness="L", bit_depth=16)
    _, _, f = oracle_mod.Oracle(dither="X", **base).translate(raw, want_f64=True)
    exact = f[:, 0] * 32768.0
    for d, var in (("T", 1 / 6 + 1 / 12), ("R", 1 / 12 + 1 / 12)):
        pcm, fr = oracle_mod.Oracle(dither=d, seed=42, **base).translate(raw)
        err = decode_pcm(pcm, 16, 1)[:, 0] - exact
        assert abs(err.mean()) < 0.01
        assert abs(err.var() - var) < 0.01
        pcm2, _ = oracle_mod.Oracle(dither=d, seed=42, **base).translate(raw)
        pcm3, _ = oracle_mod.Oracle(dither=d, seed=43, **base).translate(raw)
        assert np.array_equal(pcm, pcm2) and not np.array_equal(pcm, pcm3)
    # the generator itself: uniform 32-bit words, no DC in the two dither forms
    r = np.array([oracle_mod.rng(206, 1, n) for n in range(20000)], dtype=np.uint64)
    assert abs(r.mean() / 2 ** 32 - 0.5) < 0.01
    tp = ((r & 0xFFFF) + (r >> 16) + 1) / 65536.0 - 1.0
    assert abs(tp.mean()) < 0.01 and abs(tp.var() - 1 / 6) < 0.01 and tp.min() > -1 and tp.max() < 1


def test_float_dither_stays_within_half_ulp(oracle_mod):
    raw = synth("sine", 4096 * 8, seed=2)
    base = dict(dsd_rate=1, output_rate=352800, channels=1, fmt="P", endianness="L", bit_depth=32)
    a, _, f = oracle_mod.Oracle(dither="X", **base).translate(raw, want_f64=True)
    b, _ = oracle_mod.Oracle(dither="F", seed=5, **base).translate(raw)
    fa, fb = decode_pcm(a, 32, 1)[:, 0], decode_pcm(b, 32, 1)[:, 0]
    assert np.array_equal(fa, f[:, 0].astype(np.float32))
    ulp = np.spacing(np.abs(fa).astype(np.float32))
    assert (np.abs(fb.astype(np.float64) - f[:, 0]) <= 1.01 * ulp).all()
    assert not np.array_equal(fa, fb)


def test_parameter_errors(oracle_mod):
    """the combinations the CLI documents as unavailable (src/main.rs:62-67,85-92) are refused"""
    bad = [dict(dsd_rate=1, output_rate=705600), dict(dsd_rate=8, output_rate=88200), dict(dsd_rate=2, output_rate=1411200),
           dict(dsd_rate=2, output_rate=88200, filter="X"), dict(dsd_rate=1, output_rate=88200, filter="D"),
           dict(dsd_rate=1, output_rate=88200, filter="C"), dict(dsd_rate=1, output_rate=96000, filter="X"),
           dict(dsd_rate=1, output_rate=44100), dict(dsd_rate=3, output_rate=88200), dict(output_rate=88200, bit_depth=8),
           dict(output_rate=88200, dither="Q")]
    for kw in bad:
        with pytest.raises(oracle_mod.OracleError):
            oracle_mod.Oracle(**kw)
    for kw in [dict(dsd_rate=2, output_rate=705600), dict(dsd_rate=4, output_rate=1411200), dict(dsd_rate=8, output_rate=352800),
               dict(dsd_rate=1, output_rate=352800, filter="D"), dict(dsd_rate=2, output_rate=176400, filter="C")]:
        oracle_mod.Oracle(**kw)


def test_noise_shaped_dither_moves_the_error_out_of_band(oracle_mod):
    """'N' (extension): TPDF dither inside a second-order error-feedback loop with NTF (1 - z^-1)^2.  The
    requantisation error against the float conversion falls at low frequencies and rises towards Nyquist;
    the loop follows w = x - (2 e1 - e2), r = round(w + d), e = r - w exactly."""
    n = 4096 * 16
    buf = pack_layout([synth("sine", n, seed=1, amp=0.3), synth("sine", n, seed=2, amp=0.3, freq=3000.0)], "P", 4096)
    kw = dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, seed=3)
    ref, rf = oracle_mod.Oracle(dither="X", bit_depth=32, **kw).translate(buf)
    x = decode_pcm(ref[:rf * 8], 32, 2).astype(np.float64) * 32768.0

    def err(d):
        r, rf2 = oracle_mod.Oracle(dither=d, bit_depth=16, **kw).translate(buf)
        return decode_pcm(r[:rf2 * 4], 16, 2).astype(np.float64) - x

    def band(e, lo, hi):
        E = np.abs(np.fft.rfft(e[2000:, 0] * np.hanning(len(e) - 2000))) ** 2
        f = np.fft.rfftfreq(len(e) - 2000, 1 / 88200.0)
        return 10 * np.log10(E[(f > lo) & (f < hi)].mean())
    et, en = err("T"), err("N")
    assert band(en, 200, 4000) < band(et, 200, 4000) - 20          # in-band noise at least 20 dB down
    assert band(en, 30000, 44100) > band(et, 30000, 44100) + 6     # paid for near Nyquist
    assert abs(en[2000:].mean()) < 0.02                              # no DC offset
    # the loop itself, replayed in numpy on channel 0 from the oracle's own pre-dither samples
    o = oracle_mod.Oracle(dither="N", bit_depth=16, **kw)
    r, rf2, pre = o.translate(buf, want_f64=True)
    got = decode_pcm(r[:rf2 * 4], 16, 2)[:, 0]
    xs = pre.reshape(-1, 2)[:, 0] * 32768.0
    e1 = e2 = 0.0
    for i in range(400):                                              # (the loop restarts at multiples of 8192)
        z = oracle_mod.rng(3, 0, i)
        d = ((z & 0xFFFF) + (z >> 16) + 1) * 2.0 ** -16 - 1.0
        w = xs[i] - (2.0 * e1 - e2)
        q = w + d
        rr = np.floor(q + 0.5) if q >= 0 else np.ceil(q - 0.5)
        e2, e1 = e1, rr - w
        assert got[i] == int(rr), i


@pytest.mark.parametrize("kw", [
    dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=206),
    dict(dsd_rate=1, output_rate=352800, channels=3, fmt="I", endianness="M", block_size=1, filter="D", bit_depth=32, dither="F", seed=1),
    dict(dsd_rate=2, output_rate=176400, channels=1, fmt="P", endianness="M", block_size=100, filter="C", bit_depth=16, dither="R", seed=2, level_db=-3.0),
    dict(dsd_rate=1, output_rate=96000, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=3),
])
def test_streaming_cpu_path_equals_the_oracle(oracle_mod, kw):
    """orc_translate_stream (what bench.py times as the CPU baseline) produces the oracle's bytes, peaks and state,
    call after call, whatever the layout; configurations it does not cover (the 48k cascade) fall back to the oracle."""
    C_ = kw["channels"]
    blk = kw["block_size"] if kw["fmt"] == "P" else 1
    n = blk * (9000 // blk + 3)
    chans = [random_bytes(n, 700 + c) for c in range(C_)]
    cuts = [0, blk * 2, blk * 2, blk * (n // blk - 1), n]
    a, b = oracle_mod.Oracle(**kw), oracle_mod.Oracle(**kw)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        buf = pack_layout([ch[lo:hi] for ch in chans], kw["fmt"], blk) if hi > lo else np.zeros(0, np.uint8)
        ra, fa = a.translate(buf)
        rb, fbb = b.translate_stream(buf)
        assert fa == fbb and np.array_equal(ra[:fa * a.frame_bytes], rb[:fbb * b.frame_bytes])
    assert [a.peak(c) for c in range(C_)] == [b.peak(c) for c in range(C_)]
